"""Whole-batch HBM GB/s and MFMA-busy against the gfx950 peaks, from rocprofv3 PMC passes over bench.py.
    python tools/counters_vs_peak.py <traffic.json> <mfma_counter_collection.csv> <batches> <ms_per_batch> out.json
traffic.json: tools/pmc_traffic.py output (FETCH_SIZE x2 + WRITE_SIZE per launch); the MFMA csv needs SQ_VALU_MFMA_BUSY_CYCLES
(sum of busy cycles over the 1024 SIMD matrix pipes)."""
import collections
import csv
import json
import sys

HBM_PEAK_GBS, SIMDS, CLK_GHZ = 8000.0, 256 * 4, 2.4


def main(traffic, mfma_csv, batches, ms, out):
    batches, ms = float(batches), float(ms)
    t = json.load(open(traffic))["kernels"]
    skip = ("synth_kernel", "split_rows", "gather_pack", "w2_tile_pack", "rope_", "copyBuffer", "fillBuffer")
    hbm = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in t.items() if not any(s in k for s in skip)) / batches
    busy = collections.Counter()
    with open(mfma_csv) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                busy[r["Kernel_Name"].split("(")[0][:60]] += float(r["Counter_Value"])
    tot_busy = sum(busy.values()) / batches
    cycles = ms * 1e-3 * CLK_GHZ * 1e9
    res = {"per": "one 8 x 10 s batch", "ms_per_batch": ms,
           "hbm_bytes": round(hbm), "hbm_GBs": round(hbm / (ms * 1e-3) / 1e9, 1), "hbm_peak_GBs": HBM_PEAK_GBS,
           "hbm_frac_of_peak": round(hbm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "mfma_busy_cycles": round(tot_busy), "mfma_busy_frac": round(tot_busy / (SIMDS * cycles), 4),
           "note": "MFMA busy = sum of SQ_VALU_MFMA_BUSY_CYCLES over all kernels / (1024 SIMDs x batch time x 2.4 GHz nominal); "
                   "HBM = sum over kernels of (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, weight-packing kernels excluded",
           "top_mfma_kernels": [{"kernel": k, "share": round(v / max(sum(busy.values()), 1), 3)} for k, v in busy.most_common(6)]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
