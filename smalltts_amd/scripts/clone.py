"""python -m smalltts_amd.scripts.clone --wav ref.wav --text "..." [--duration S] [--out out/clone.wav]
(reference src/scripts/infer/clone.py: read wav -> mono -> 24 kHz -> codec encode -> synthesize)"""
import argparse
from pathlib import Path

import torch

from ..api import Encoder, SmallTTS, estimate_duration
from ..audio import write_wav_pcm16
from ._common import add_engine_args, load_reference_wav, tokens_for


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--wav", required=True, help="reference audio file")
    ap.add_argument("--text", required=True, help="text to speak")
    ap.add_argument("--duration", type=float, default=None, help="duration in seconds (auto if omitted)")
    ap.add_argument("--out", default="out/clone.wav")
    add_engine_args(ap)
    args = ap.parse_args(argv)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    print("loading")
    kw = dict(weights=args.weights, device=args.device, precision=args.precision)
    enc = Encoder(**kw)
    x = load_reference_wav(args.wav, enc.engine)
    print("encoding reference audio")
    ref_latents = enc.encode_reference(torch.from_numpy(x))[0].numpy()
    tts = SmallTTS(num_steps=args.steps, seed=args.seed, **kw)
    tokens = tokens_for(args, args.text)
    duration = args.duration or estimate_duration(args.text)
    print(f"generating ({duration:.1f}s)")
    audio = tts.synthesize(ref_latents, tokens, duration)
    write_wav_pcm16(args.out, audio.squeeze(), 24_000)
    print(args.out)


if __name__ == "__main__":
    main()
