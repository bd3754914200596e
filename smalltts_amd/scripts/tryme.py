"""python -m smalltts_amd.scripts.tryme [text]  ->  out/tryme.wav   (reference src/scripts/tryme.py)"""
import argparse
from pathlib import Path

import numpy as np

from ..api import SmallTTS, estimate_duration
from ..audio import write_wav_pcm16
from ._common import add_engine_args, tokens_for

DEFAULT_TEXT = "hello this is small brain speaking, thanks for trying this model out and have fun"


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("text", nargs="?", default=DEFAULT_TEXT)
    ap.add_argument("--ref-latents", default="assets/tryme/latents.npy", help="(R,64) float32 .npy reference voice")
    ap.add_argument("--out", default="out/tryme.wav")
    add_engine_args(ap)
    args = ap.parse_args(argv)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    print("loading model")
    model = SmallTTS(weights=args.weights, device=args.device, precision=args.precision, num_steps=args.steps,
                     seed=args.seed)
    if Path(args.ref_latents).exists():
        ref = np.load(args.ref_latents).astype(np.float32)
    else:
        print(f"{args.ref_latents} not found: using a seeded random reference voice")
        ref = np.random.default_rng(0).standard_normal((15, 64)).astype(np.float32)
    tokens = tokens_for(args, args.text)
    duration = estimate_duration(args.text)
    print(f"generating ({duration:.1f}s estimated)")
    audio = model.synthesize(ref, tokens, duration)
    write_wav_pcm16(args.out, audio.squeeze(), 24_000)
    print(args.out)


if __name__ == "__main__":
    main()
