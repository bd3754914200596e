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
"""
from __future__ import annotations

import argparse
import json
import struct
import sys
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .weights import (CodecSpec, all_param_specs, clean_state_dict_keys, dit_param_specs, save_weight_file)


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


def convert_onnx(paths: Sequence[str], out: str, codec: Optional[CodecSpec] = None, allow_partial: bool = False):
    """Merge the initialisers of the given files; keep those whose names are parameters of this build."""
    merged: Dict[str, np.ndarray] = {}
    for p in paths:
        for name, arr in read_onnx_initializers(p).items():
            if arr.dtype.kind == "f":
                merged[name] = np.ascontiguousarray(arr, dtype=np.float32)
    merged = {k: v for k, v in clean_state_dict_keys(merged).items()}
    specs = list(dit_param_specs()) + ([s for s in all_param_specs(codec) if s[0] not in dict(dit_param_specs())] if codec else [])
    rep = check_against_specs(merged, specs)
    # exporters fold Linear weights into MatMul initialisers stored as [in, out]: accept an exact transposed shape match by name
    for name, got, want in list(rep["shape_mismatch"]):
        if len(want) == 2 and tuple(reversed(got)) == tuple(want):
            merged[name] = np.ascontiguousarray(merged[name].T)
            rep["shape_mismatch"].remove((name, got, want))
            rep["matched"] += 1
    if not rep.ok and not allow_partial:
        raise ValueError(f"ONNX initialisers do not cover the parameter inventory by name — {rep.summary()}; "
                         f"unmatched initialisers (first 5): {rep['unexpected'][:5]}")
    keep = {n: merged[n] for n, s in specs if n in merged and tuple(merged[n].shape) == tuple(s)}
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
    a = ap.parse_args(argv)
    rep = (convert_checkpoint(a.checkpoint, a.out, a.key, a.allow_partial) if a.checkpoint
           else convert_onnx(a.onnx, a.out, None, a.allow_partial))
    print(rep.summary())
    if a.report:
        with open(a.report, "w") as f:
            json.dump(rep, f, indent=1)
    return 0 if rep.ok or a.allow_partial else 1


if __name__ == "__main__":
    sys.exit(main())
