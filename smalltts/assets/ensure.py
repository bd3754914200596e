"""`from smalltts.assets.ensure import ensure_assets` (reference src/smalltts/assets/ensure.py:21-40 downloads the ONNX files
from HuggingFace into ./assets/<folder>).  This build neither downloads (it is meant to run without network) nor reads ONNX
files at run time: weights come from one converted flat file (`python -m smalltts_amd.convert`, SMALLTTS_WEIGHTS or
assets/smalltts.smtts).  ensure_assets therefore only CHECKS, so that the reference's scripts run unchanged and fail with
an instruction instead of a stack trace."""
import os
from typing import Iterable

from smalltts_amd.api import DEFAULT_WEIGHTS


def ensure_assets(folders: Iterable[str] = ("codec", "dmd"), root: str = "assets") -> str:
    src = os.environ.get("SMALLTTS_WEIGHTS", DEFAULT_WEIGHTS)
    if src.startswith("synthetic") or all(os.path.exists(p) for p in src.split("+") if not p.startswith("synthetic")):
        return src
    raise FileNotFoundError(
        f"weights {src!r} not found.  Convert the released files once with `python -m smalltts_amd.convert --checkpoint "
        f"<dmd_checkpoints/checkpoint_latest.pt> --out {DEFAULT_WEIGHTS}` (or --onnx <assets/dmd/*.onnx assets/codec/*.onnx>), "
        "or set SMALLTTS_WEIGHTS (e.g. synthetic:0 for seeded random weights).")
