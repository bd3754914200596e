"""python -m smalltts_amd.scripts.interactive [--wav ref.wav]   (reference src/scripts/infer/interactive.py:17-60)

Type a line, get speech: every line is tokenised, given estimate_duration(line) seconds and synthesised with the loaded
reference voice; prints generation time and the real-time factor like the reference (`gen 0.02s, 480.0x rt`).  The reference
plays through `sounddevice` and draws with `rich`; both are optional here — without sounddevice every utterance is written to
out/interactive_NNN.wav instead."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

from ..api import Encoder, SmallTTS, estimate_duration
from ..audio import write_wav_pcm16
from ._common import add_engine_args, load_reference_wav, tokens_for


def main(argv=None, lines=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--wav", type=str, help="reference audio file")
    ap.add_argument("--ref-latents", default="assets/tryme/latents.npy", help="(R,64) float32 .npy reference voice (no --wav)")
    ap.add_argument("--outdir", default="out")
    add_engine_args(ap)
    args = ap.parse_args(argv)
    try:
        import sounddevice as sd
    except Exception:
        sd = None
    print("smalltts interactive\ntype and press enter. ctrl-c to exit.\nloading model")
    t0 = time.perf_counter()
    kw = dict(weights=args.weights, device=args.device, precision=args.precision)
    model = SmallTTS(num_steps=args.steps, seed=args.seed, **kw)
    if args.wav:
        enc = Encoder(**kw)
        x = torch.from_numpy(load_reference_wav(args.wav, enc.engine))
        ref_latents = enc.encode_reference(x)[0].cpu().numpy()        # cached per voice (interactive.py:34)
    elif Path(args.ref_latents).exists():
        ref_latents = np.load(args.ref_latents).astype(np.float32)
    else:
        print(f"{args.ref_latents} not found: using a seeded random reference voice")
        ref_latents = np.random.default_rng(0).standard_normal((15, 64)).astype(np.float32)
    Path(args.outdir).mkdir(parents=True, exist_ok=True)
    first, n = True, 0
    src = iter(lines) if lines is not None else None
    while True:
        try:
            s = (next(src) if src is not None else input(">> ")).strip()
        except (EOFError, KeyboardInterrupt, StopIteration):
            break
        if not s:
            continue
        st = time.perf_counter()
        tokens = tokens_for(args, s)
        audio = model.synthesize(ref_latents, tokens, estimate_duration(s))
        dt = time.perf_counter() - st
        dur = audio.shape[-1] / 24_000.0
        rtf = dur / dt if dt > 0 else 0.0
        if first:
            print(f"gen {dt:.2f}s (+{time.perf_counter() - t0 - dt:.2f}s warmup), {rtf:.1f}x rt")
            first = False
        else:
            print(f"gen {dt:.2f}s, {rtf:.1f}x rt")
        a = audio.squeeze()
        if sd is not None:
            sd.play(a, 24_000)
            sd.wait()
        else:
            path = str(Path(args.outdir) / f"interactive_{n:03d}.wav")
            write_wav_pcm16(path, a, 24_000)
            print(path)
        n += 1
    return n


if __name__ == "__main__":
    sys.exit(0 if main() >= 0 else 1)
