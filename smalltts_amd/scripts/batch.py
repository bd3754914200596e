"""python -m smalltts_amd.scripts.batch  (reference src/scripts/infer/batch.py): reads
assets/test_audio/transcriptions.json ({"filename": ...} items), clones each voice with one of four
fixed texts and writes out/<stem>_gen.wav.  Unlike the reference's sequential loop, all files are
encoded and synthesised as ONE padded batch on the GPU."""
import argparse
import json
from pathlib import Path

import numpy as np
import torch

from ..api import Encoder, SmallTTS, estimate_duration
from ..audio import write_wav_pcm16
from ..phonemes import get_token_ids
from ._common import add_engine_args, load_reference_wav

TEXTS = [
    "Hello world, I am small tts, and I am talking!",
    "I can clone any voice and emotion.",
    "I have an ONNX export and run very fast.",
    "Woah, this is awesome I can do any character!",
]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="assets/test_audio")
    ap.add_argument("--out", default="out")
    add_engine_args(ap)
    args = ap.parse_args(argv)
    td = Path(args.dir)
    with open(td / "transcriptions.json") as f:
        items = json.load(f)
    files = [td / it["filename"] for it in items]
    outdir = Path(args.out)
    outdir.mkdir(parents=True, exist_ok=True)
    kw = dict(weights=args.weights, device=args.device, precision=args.precision)
    enc, tts = Encoder(**kw), SmallTTS(num_steps=args.steps, seed=args.seed, **kw)
    pairs = list(zip(files, TEXTS))
    refs = []
    for fpath, _ in pairs:
        refs.append(enc.encode_reference(torch.from_numpy(load_reference_wav(str(fpath), enc.engine)))[0].numpy())
    toks = [get_token_ids(t, backend=args.tokenizer) for _, t in pairs]
    durs = [estimate_duration(t) for _, t in pairs]
    audios = tts.synthesize_batch(refs, toks, durs)
    for i, ((fpath, _), audio) in enumerate(zip(pairs, audios)):
        out_path = outdir / f"{fpath.stem}_gen.wav"
        write_wav_pcm16(str(out_path), np.asarray(audio).squeeze(), 24_000)
        print(f"[{i + 1}/{len(pairs)}] {fpath.name} -> {out_path}")


if __name__ == "__main__":
    main()
