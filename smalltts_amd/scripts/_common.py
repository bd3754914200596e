"""Shared argument handling for the CLI surface (tryme / clone / batch)."""
import argparse
import json

import numpy as np

from ..audio import read_wav, resample_hq
from ..phonemes import get_token_ids, parse_tokens_arg


def add_engine_args(ap: argparse.ArgumentParser) -> None:
    ap.add_argument("--weights", default=None, help="weight file | checkpoint .pt | synthetic:<seed>")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--precision", default="f16", help="f16 (mixed, default) | bf16x3 (fp32-class) | bf16, optionally with site overrides")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--tokens", default=None, help="pre-tokenised phoneme ids (comma/space separated or JSON)")
    ap.add_argument("--tokenizer", default="espeak", choices=["espeak", "chars"])


def tokens_for(args, text: str):
    if args.tokens:
        if args.tokens.strip().startswith("["):
            return [int(t) for t in json.loads(args.tokens)]
        return parse_tokens_arg(args.tokens)
    return get_token_ids(text, backend=args.tokenizer)


def load_reference_wav(path: str, engine=None) -> np.ndarray:
    """-> mono float32 @ 24 kHz, shape (1, 1, S) (clone.py:29-33).  With an engine the resampling runs on the
    device (smtts_resample_poly); the host numpy restatement is the fallback-free CPU path of the same filter."""
    y, sr = read_wav(path)
    if y.ndim == 2:
        y = y.mean(axis=1)
    if engine is not None:
        y = engine.resample(y.astype(np.float32), sr, 24_000).cpu().numpy()
    else:
        y = resample_hq(y.astype(np.float32), sr, 24_000)
    return y[None, None, :]
