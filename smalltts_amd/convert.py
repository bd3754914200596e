"""Real-weight ingestion: turn the reference's artefacts into this build's flat weight file.

Two sources, both optional at run time because neither exists offline:

* a training checkpoint ``dmd_checkpoints/checkpoint_latest.pt`` whose ``"student_model"`` entry is the
  ``DiTModel(64).state_dict()`` (reference ``train/distill.py:468-479``; wrapper prefixes stripped as in
  ``distill.py:47-54``) — fully checked against :func:`smalltts_amd.weights.dit_param_specs`;
* the four ONNX files the reference downloads (``assets/ensure.py``): their *initialisers* carry the weights.
  ``onnx`` is not a dependency: :func:`read_onnx_initializers` walks the protobuf wire format directly
  (ModelProto.graph = 7, GraphProto.initializer = 5, TensorProto dims = 1 / data_type = 2 / float_data = 4 /
  int64_data = 7 / name = 8 / raw_data = 9).  Tensors are matched to this build's parameter names by name;
  exporters rename MatMul weights (``onnx::MatMul_123``, stored transposed), so everything that cannot be matched
  by name is listed in the report instead of being guessed — codec parity stays "unpinned" until a name map for
  the exported codec graph has been checked against real files.

CLI:  python -m smalltts_amd.convert --checkpoint ckpt.pt --out weights.smtts
      python -m smalltts_amd.convert --onnx condition_encoder.onnx denoiser.onnx --out weights.smtts [--allow-partial]
      python -m smalltts_amd.convert --onnx decoder.onnx encoder.onnx --codec [--codec-spec spec.json] [--map-by-position]
                                     --out codec.smtts --report codec_report.json
The last form is the one-command path to pinning the codec the day the files are available: the report lists, per shape
signature, what the exported graph holds against this build's CodecSpec inventory (missing biases, extra norms, other depths).
"""
from __future__ import annotations

import argparse
import json
import struct
import sys
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .weights import (CodecSpec, clean_state_dict_keys, codec_decoder_param_specs, codec_encoder_param_specs, dit_param_specs,
                      save_weight_file)


# ----------------------------------------------------------------------------------------------
# checking a name -> array dict against the parameter inventory
# ----------------------------------------------------------------------------------------------
class ConversionReport(dict):
    """{"matched": n, "missing": [...], "unexpected": [...], "shape_mismatch": [(name, got, want)]}"""

    @property
    def ok(self) -> bool:
        return not self["missing"] and not self["shape_mismatch"]

    def summary(self) -> str:
        return (f"matched {self['matched']}, missing {len(self['missing'])}, shape mismatches "
                f"{len(self['shape_mismatch'])}, unexpected {len(self['unexpected'])}")


def check_against_specs(tensors: Dict[str, np.ndarray], specs: Iterable[Tuple[str, Tuple[int, ...]]]) -> ConversionReport:
    want = dict(specs)
    rep = ConversionReport(matched=0, missing=[], unexpected=[], shape_mismatch=[])
    for name, shape in want.items():
        if name not in tensors:
            rep["missing"].append(name)
        elif tuple(tensors[name].shape) != tuple(shape):
            rep["shape_mismatch"].append((name, tuple(tensors[name].shape), tuple(shape)))
        else:
            rep["matched"] += 1
    rep["unexpected"] = sorted(k for k in tensors if k not in want)
    return rep


# ----------------------------------------------------------------------------------------------
# training checkpoint -> weight file
# ----------------------------------------------------------------------------------------------
def state_dict_from_checkpoint(obj, key: Optional[str] = None) -> Dict[str, np.ndarray]:
    """Pull the DiTModel state_dict out of a loaded checkpoint object and normalise its keys."""
    if isinstance(obj, dict):
        for k in ([key] if key else []) + ["student_model", "model", "state_dict"]:
            if k in obj and isinstance(obj[k], dict):
                obj = obj[k]
                break
    if not isinstance(obj, dict):
        raise ValueError("checkpoint does not contain a state_dict")
    out = {}
    for k, v in clean_state_dict_keys(obj).items():
        if not hasattr(v, "shape"):
            continue
        arr = v.detach().cpu().float().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
        out[k] = np.asarray(arr, dtype=np.float32, order="C")  # (ascontiguousarray would turn 0-d scalars into shape (1,))
    return out


def convert_checkpoint(path: str, out: str, key: Optional[str] = None, allow_partial: bool = False) -> ConversionReport:
    import torch
    ck = torch.load(path, map_location="cpu", weights_only=True)
    tensors = state_dict_from_checkpoint(ck, key)
    rep = check_against_specs(tensors, dit_param_specs())
    if not rep.ok and not allow_partial:
        raise ValueError(f"{path}: not a DiTModel(64) state_dict — {rep.summary()}; first problems: "
                         f"{(rep['missing'] + [m[0] for m in rep['shape_mismatch']])[:5]}")
    keep = {n: tensors[n] for n, _ in dit_param_specs() if n in tensors}
    save_weight_file(out, keep)
    return rep


# ----------------------------------------------------------------------------------------------
# ONNX initialisers without the onnx package
# ----------------------------------------------------------------------------------------------
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    val, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _fields(buf: bytes):
    """Yield (field_number, wire_type, value) for one protobuf message; length-delimited values are memoryviews."""
    pos, n = 0, len(buf)
    mv = memoryview(buf)
    while pos < n:
        tag, pos = _varint(buf, pos)
        fno, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = mv[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = mv[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = mv[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, val


_ONNX_DTYPES = {1: np.float32, 7: np.int64, 10: np.float16, 11: np.float64, 6: np.int32, 9: np.bool_}


def _tensor_proto(buf: bytes) -> Tuple[str, Optional[np.ndarray]]:
    dims: List[int] = []
    dtype, name, raw = 1, "", None
    floats: List[float] = []
    ints: List[int] = []
    external = False
    for fno, wt, val in _fields(buf):
        if fno == 1:  # dims: repeated int64, packed or not
            if wt == 2:
                b, p = bytes(val), 0
                while p < len(b):
                    d, p = _varint(b, p)
                    dims.append(d)
            else:
                dims.append(val)
        elif fno == 2:
            dtype = val
        elif fno == 4:  # float_data (packed)
            floats.extend(np.frombuffer(bytes(val), dtype="<f4").tolist() if wt == 2 else [struct.unpack("<f", bytes(val))[0]])
        elif fno == 7:  # int64_data
            if wt == 2:
                b, p = bytes(val), 0
                while p < len(b):
                    d, p = _varint(b, p)
                    ints.append(d - (1 << 64) if d >= 1 << 63 else d)
            else:
                ints.append(val)
        elif fno == 8:
            name = bytes(val).decode()
        elif fno == 9:
            raw = bytes(val)
        elif fno == 14 and val == 1:  # data_location = EXTERNAL
            external = True
    if external:
        return name, None
    np_dtype = _ONNX_DTYPES.get(dtype)
    if np_dtype is None:
        return name, None
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np.dtype(np_dtype).newbyteorder("<"))
    elif floats:
        arr = np.asarray(floats, dtype=np.float32)
    elif ints:
        arr = np.asarray(ints, dtype=np.int64)
    else:
        arr = np.zeros(0, dtype=np_dtype)
    return name, arr.reshape(dims) if dims else arr.reshape(())


def read_onnx_initializers(path: str) -> Dict[str, np.ndarray]:
    """name -> array for every initialiser stored inside the .onnx file (external-data tensors are skipped)."""
    with open(path, "rb") as f:
        model = f.read()
    out: Dict[str, np.ndarray] = {}
    for fno, wt, val in _fields(model):
        if fno == 7 and wt == 2:  # ModelProto.graph
            for gno, gwt, gval in _fields(bytes(val)):
                if gno == 5 and gwt == 2:  # GraphProto.initializer
                    name, arr = _tensor_proto(bytes(gval))
                    if arr is not None:
                        out[name] = arr
    return out


def _signature(shape: Sequence[int]) -> Tuple[int, ...]:
    """Layout-insensitive shape signature: size-1 axes dropped (a depthwise conv weight is (C, 1, K) in torch and (C, K) here),
    2-D shapes order-insensitive (exporters store Linear weights as MatMul operands [in, out] = W^T)."""
    sq = tuple(int(d) for d in shape if int(d) != 1)
    return tuple(sorted(sq)) if len(sq) == 2 else sq


def _fit(arr: np.ndarray, want: Tuple[int, ...], name: str) -> Optional[np.ndarray]:
    """Bring an initialiser with the right signature into the inventory's layout (reshape size-1 axes, transpose MatMul
    operands).  Square 2-D tensors are ambiguous: transposed iff the initialiser is a MatMul operand by name."""
    if tuple(arr.shape) == tuple(want):
        if len(want) == 2 and want[0] == want[1] and "matmul" in name.lower():
            return np.ascontiguousarray(arr.T)
        return arr
    sq = arr.reshape([d for d in arr.shape if d != 1] or [1]) if arr.ndim else arr
    wsq = tuple(d for d in want if d != 1)
    if tuple(sq.shape) == wsq:
        return np.ascontiguousarray(sq).reshape(want)
    if sq.ndim == 2 and tuple(sq.shape[::-1]) == wsq:
        return np.ascontiguousarray(sq.T).reshape(want)
    return None


def shape_signature_report(inits: Dict[str, np.ndarray], specs: Iterable[Tuple[str, Tuple[int, ...]]]) -> dict:
    """Match ONNX initialisers to the parameter inventory: by (cleaned) name first, then BY POSITION inside each shape
    signature — the k-th still-unmatched initialiser of a signature (file order = graph order of the exporter) against the
    k-th still-unmatched parameter of that signature (inventory order = module registration order).  Position matches are
    unverified by construction and reported as such; signatures whose counts differ are listed with both sides so that the
    delta between the exported graph and this build's spec (a missing bias, an extra norm, another depth) can be read off."""
    specs = list(specs)
    want = dict(specs)
    rep = {"by_name": {}, "by_position": {}, "count_mismatch": [], "expected_only": [], "onnx_only": []}
    left_onnx = []
    for name, arr in inits.items():
        if arr.dtype.kind != "f":
            continue
        if name in want and _signature(arr.shape) == _signature(want[name]):
            rep["by_name"][name] = name
        else:
            left_onnx.append(name)
    left_exp = [n for n, _ in specs if n not in rep["by_name"]]
    groups: Dict[Tuple[int, ...], List[List[str]]] = {}
    for n in left_exp:
        groups.setdefault(_signature(want[n]), [[], []])[0].append(n)
    for n in left_onnx:
        groups.setdefault(_signature(inits[n].shape), [[], []])[1].append(n)
    for sig, (exp, got) in sorted(groups.items(), key=lambda kv: (len(kv[0]), kv[0])):
        if exp and got and len(exp) == len(got):
            for e_, g_ in zip(exp, got):
                rep["by_position"][g_] = e_
        elif exp and got:
            rep["count_mismatch"].append({"signature": list(sig), "expected": len(exp), "onnx": len(got),
                                          "expected_names": exp[:6], "onnx_names": got[:6]})
        elif exp:
            rep["expected_only"].append({"signature": list(sig), "count": len(exp), "names": exp[:6]})
        else:
            rep["onnx_only"].append({"signature": list(sig), "count": len(got), "names": got[:6]})
    rep["summary"] = (f"{len(rep['by_name'])} by name, {len(rep['by_position'])} by position (unverified), "
                      f"{len(rep['count_mismatch'])} signatures with differing counts, "
                      f"{sum(e['count'] for e in rep['expected_only'])} parameters without a candidate, "
                      f"{sum(e['count'] for e in rep['onnx_only'])} initialisers without a parameter")
    return rep


def convert_onnx(paths: Sequence[str], out: str, codec: Optional[CodecSpec] = None, allow_partial: bool = False,
                 map_by_position: bool = False, parts: Optional[Sequence[str]] = None):
    """Merge the float initialisers of the given files and keep those that are parameters of this build: by name (exact or
    transposed 2-D shape) and, with map_by_position, by position inside each shape signature (shape_signature_report).
    `parts` names the inventories the files are matched against — "dit" (condition_encoder.onnx + denoiser.onnx; default
    without `codec`), "decoder" / "encoder" (codec/decoder.onnx, codec/encoder.onnx; default both with `codec`): matching a file
    against an inventory it does not hold would only add signature collisions.  The report's `signature` entry lists every delta."""
    merged: Dict[str, np.ndarray] = {}
    for p in paths:
        for name, arr in read_onnx_initializers(p).items():
            if arr.dtype.kind == "f":
                merged[name] = np.ascontiguousarray(arr, dtype=np.float32)
    merged = {k: v for k, v in clean_state_dict_keys(merged).items()}
    parts = tuple(parts) if parts else (("decoder", "encoder") if codec else ("dit",))
    if ("decoder" in parts or "encoder" in parts) and codec is None:
        codec = CodecSpec()
    specs = []
    if "dit" in parts:
        specs += list(dit_param_specs())
    if "decoder" in parts:
        specs += codec_decoder_param_specs(codec)
    if "encoder" in parts:
        specs += codec_encoder_param_specs(codec)
    want = dict(specs)
    sig = shape_signature_report(merged, specs)
    tensors: Dict[str, np.ndarray] = {}
    for name in sig["by_name"]:
        fit = _fit(merged[name], want[name], name)
        if fit is not None:
            tensors[name] = fit
    if map_by_position:
        for src, dst in sig["by_position"].items():
            fit = _fit(merged[src], want[dst], src)
            if fit is not None:
                tensors[dst] = fit
    rep = check_against_specs(tensors, specs)
    rep["unexpected"] = sorted(k for k in merged if k not in tensors and k not in sig["by_position"])
    rep["signature"] = sig
    if not rep.ok and not allow_partial:
        raise ValueError(f"ONNX initialisers do not cover the parameter inventory — {rep.summary()}; signature match: "
                         f"{sig['summary']}" + ("" if map_by_position else " (re-run with --map-by-position to apply the position matches)"))
    keep = {n: tensors[n] for n, s_ in specs if n in tensors}
    save_weight_file(out, keep, codec)
    return rep


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--checkpoint", help="training checkpoint (.pt) holding the DiTModel state_dict")
    src.add_argument("--onnx", nargs="+", help=".onnx files whose initialisers hold the weights")
    ap.add_argument("--key", default=None, help="checkpoint entry to use (default: student_model / model)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--allow-partial", action="store_true", help="write what matched instead of failing")
    ap.add_argument("--report", default=None, help="write the match report as JSON")
    ap.add_argument("--codec", action="store_true", help="--onnx: include the codec inventory (decoder.onnx / encoder.onnx)")
    ap.add_argument("--codec-spec", default=None, help="JSON file with CodecSpec fields (default: the built-in VibeVoice-shaped spec)")
    ap.add_argument("--parts", default=None, help="--onnx: inventories to match, comma separated: dit | decoder | encoder "
                                                  "(default: dit, or decoder,encoder with --codec)")
    ap.add_argument("--map-by-position", action="store_true",
                    help="--onnx: also take initialisers matched by position inside their shape signature (unverified; see the report)")
    a = ap.parse_args(argv)
    codec = None
    if a.codec or a.codec_spec:
        codec = CodecSpec(**json.load(open(a.codec_spec))) if a.codec_spec else CodecSpec()
    rep = (convert_checkpoint(a.checkpoint, a.out, a.key, a.allow_partial) if a.checkpoint
           else convert_onnx(a.onnx, a.out, codec, a.allow_partial, a.map_by_position, a.parts.split(",") if a.parts else None))
    print(rep.summary())
    if "signature" in rep:
        sg = rep["signature"]
        print("signature match:", sg["summary"])
        for row in sg["count_mismatch"]:
            print(f"  shape {tuple(row['signature'])}: this build expects {row['expected']} ({', '.join(row['expected_names'][:3])} ...), "
                  f"the file has {row['onnx']} ({', '.join(row['onnx_names'][:3])} ...)")
        for row in sg["expected_only"]:
            print(f"  shape {tuple(row['signature'])}: {row['count']} parameter(s) with no initialiser of that shape, e.g. {row['names'][0]}")
        for row in sg["onnx_only"]:
            print(f"  shape {tuple(row['signature'])}: {row['count']} initialiser(s) this build has no parameter for, e.g. {row['names'][0]}")
    if a.report:
        with open(a.report, "w") as f:
            json.dump(rep, f, indent=1)
    return 0 if rep.ok or a.allow_partial else 1


if __name__ == "__main__":
    sys.exit(main())
