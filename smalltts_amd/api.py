"""Drop-in Python API of the reference (`smalltts.SmallTTS`, `estimate_duration`, codec `Encoder` /
`Decoder`) on top of the gfx950 engine.

Mirrors reference `src/smalltts/infer/onnx.py` (class SmallTTS :50-159, constants :11-14,
estimate_duration :17-18) and `src/smalltts/codec/onnx.py` (Encoder/Decoder :34-75): same positional
arguments, argument meaning, return types and error-by-exception behaviour.  The three ONNX path
arguments are accepted for call compatibility; weights come from `weights=` (keyword-only
addition): a flat weight file written by `smalltts_amd.weights.save_weight_file`, a torch
checkpoint holding the reference's state_dict (optionally under "student_model",
distill.py:468-479), or "synthetic:<seed>" (seeded random weights — there are no released weights
offline).  Unlike the reference, `forward` runs all utterances as ONE padded batch on the GPU.
"""
from __future__ import annotations

import collections
import hashlib
import os
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from .engine import DEFAULT_PRECISION, HipEngine
from .weights import DEFAULT_CODEC, CodecSpec, all_param_specs, load_weight_file

SAMPLE_RATE = 24_000
HOP_SIZE = 3_200
NUM_STEPS = 4
CHARS_PER_SECOND = 11.5
DEFAULT_WEIGHTS = os.environ.get("SMALLTTS_WEIGHTS", "assets/smalltts.smtts")

_ENGINES: Dict[tuple, HipEngine] = {}


def estimate_duration(text: str, min_sec: float = 0.5, max_sec: float = 30.0) -> float:
    return max(min_sec, min(len(text) / CHARS_PER_SECOND, max_sec))


def _split_sources(weights) -> List[str]:
    if isinstance(weights, (list, tuple)):
        return [str(w) for w in weights]
    return [w for w in str(weights).split("+") if w]


def _validated(tensors: Dict[str, object], source: str, codec: CodecSpec) -> Dict[str, object]:
    """Reject a tensor whose shape is not the inventory's BEFORE it reaches the packers: the kernels assume the model's
    strides (960 / 2400 / 512 ...), so a checkpoint of another model size must be a clean error naming the tensor."""
    want = dict(all_param_specs(codec))
    bad = [(k, tuple(v.shape), want[k]) for k, v in tensors.items() if k in want and tuple(v.shape) != tuple(want[k])]
    if bad:
        k, got, exp = bad[0]
        raise ValueError(f"{source}: tensor {k!r} has shape {got}, this build expects {exp} "
                         f"({len(bad)} mismatching tensor{'s' if len(bad) > 1 else ''}) — not a DiTModel(64) / CodecSpec checkpoint")
    unknown = [k for k in tensors if k not in want]
    if unknown and len(unknown) == len(tensors):
        raise ValueError(f"{source}: none of its {len(tensors)} tensors is a parameter of this build (first: {unknown[0]!r})")
    # A codec.* tensor the engine WOULD apply (biases, layer scales and the final norm are optional at run time) but that the
    # CodecSpec in force says is absent must not vanish silently: the audio would come out wrong with no error.
    full = dict(all_param_specs(CodecSpec(**{**codec.to_dict(), "conv_bias": True, "ffn_bias": True, "layer_scale": True, "final_norm": True})))
    dropped = [k for k in unknown if k in full]
    if dropped:
        raise ValueError(f"{source}: tensor {dropped[0]!r} (+{len(dropped) - 1} more) is an optional codec parameter that this "
                         "CodecSpec switches off (conv_bias / ffn_bias / layer_scale / final_norm) — it would be ignored; "
                         "load it with the matching spec (convert.py --codec-spec, or a .smtts file that carries its spec)")
    return {k: v for k, v in tensors.items() if k in want}


def _load_weights_into(eng: HipEngine, weights, parts: Sequence[str]) -> None:
    """`weights`: one source or several joined with '+' (or a list), applied in order:
      * ``*.pt / *.pth / *.ckpt``  torch checkpoint holding the reference's ``DiTModel`` state_dict, bare or under
        ``"student_model"`` / ``"model"`` with the wrapper prefixes of distill.py:47-54 (codec.* keys are taken too);
      * any other path             flat weight file (`weights.save_weight_file`, written by `smalltts_amd.convert`);
      * ``synthetic[:seed]``       seeded random weights for every requested part NOT provided by an earlier source.
    e.g. ``"dmd.pt+codec.smtts"`` (DiT checkpoint + separately converted codec) or ``"dmd.pt+synthetic:7"``."""
    have: set = set()
    prefixes = {"dit": ("velocity.",), "decoder": ("codec.decoder.",), "encoder": ("codec.encoder.",)}
    codec = eng.codec_spec

    def note(names):
        for part, pre in prefixes.items():
            if any(n.startswith(pre) for n in names):
                have.add(part)

    for src in _split_sources(weights):
        if src.startswith("synthetic"):
            seed = int(src.split(":", 1)[1]) if ":" in src else 0
            todo = [p for p in parts if p not in have]
            if todo:
                eng.load_synthetic(seed, parts=todo)
                have.update(todo)
            continue
        if not os.path.exists(src):
            raise FileNotFoundError(
                f"weight file {src!r} not found. The reference downloads its ONNX weights from HuggingFace "
                "(assets/ensure.py); offline, pass weights='synthetic:<seed>' or a converted weight file.")
        if src.endswith((".pt", ".pth", ".ckpt")):
            from .convert import state_dict_from_checkpoint
            tensors = state_dict_from_checkpoint(torch.load(src, map_location="cpu", weights_only=True))
        else:
            tensors, cdict = load_weight_file(src)
            if cdict:
                codec = CodecSpec(**cdict)
                eng.set_codec_spec(codec)
        tensors = _validated(tensors, src, codec)
        eng.load_state_dict(tensors)
        note(tensors)
    eng.finalize()
    # real weights (anything but the seeded recipe): hold the preset to split-bf16 on a probe batch once, demote what drifts
    if any(not src.startswith("synthetic") for src in _split_sources(weights)) and os.environ.get("SMTTS_CALIBRATE", "1") != "0":
        eng.calibrate()


def get_engine(weights: Optional[str] = None, device: int = 0, precision: str = DEFAULT_PRECISION,
               parts: Sequence[str] = ("dit", "decoder", "encoder")) -> HipEngine:
    """One engine (one weight copy) per (weights, device, parts); shared by SmallTTS/Encoder/Decoder."""
    weights = weights or DEFAULT_WEIGHTS
    key = ("+".join(_split_sources(weights)), int(device), tuple(sorted(parts)))
    eng = _ENGINES.get(key)
    if eng is None:
        eng = HipEngine(device, precision)
        _load_weights_into(eng, weights, parts)
        _ENGINES[key] = eng
    if eng.precision != precision:
        eng.set_precision(precision)
    return eng


def _frames(duration_sec: float) -> int:
    return max(1, int(duration_sec * SAMPLE_RATE / HOP_SIZE))  # floor, infer/onnx.py:84


class SmallTTS:
    """DMD few-step synthesis: condition-encode -> n-step sampler -> codec decode, all on one MI355X."""

    def __init__(self, cond_encoder_path: str = "assets/dmd/condition_encoder.onnx",
                 denoiser_path: str = "assets/dmd/denoiser.onnx",
                 codec_decoder_path: str = "assets/codec/decoder.onnx",
                 providers: Optional[Iterable[str]] = None, *, weights: Optional[str] = None, device: int = 0,
                 precision: str = DEFAULT_PRECISION, num_steps: int = NUM_STEPS, seed: Optional[int] = None,
                 engine: Optional[HipEngine] = None, device_ids: Optional[Sequence[int]] = None) -> None:
        """Keyword-only additions to the reference signature (infer/onnx.py:53-59): `weights`, `device`, `precision`,
        `num_steps`, `seed`, and `device_ids` — the GPUs `synthesize_sharded` spreads a request list over from THIS process
        (one weight replica, engine and host thread per GPU; under torch.distributed the process group is used instead)."""
        if device_ids:
            device = int(device_ids[0])
        self.engine = engine or get_engine(weights, device, precision, parts=("dit", "decoder", "encoder"))
        if not (self.engine.has("dit") and self.engine.has("decoder")):
            raise RuntimeError("SmallTTS needs DiT and codec-decoder weights")
        self.num_steps = int(num_steps)
        self._seed = seed
        self._rng = np.random.default_rng(seed) if seed is not None else None
        self._replicas: List["SmallTTS"] = [self]
        for d in list(device_ids or [])[1:]:
            # a repeated device id is its own replica (own engine + weights): the threads must not share an engine
            eng = (HipEngine(int(d), precision) if int(d) in [r.engine.device_index for r in self._replicas]
                   else get_engine(weights, int(d), precision, parts=("dit", "decoder", "encoder")))
            if not eng.has("dit"):
                _load_weights_into(eng, weights or DEFAULT_WEIGHTS, ("dit", "decoder", "encoder"))
            # replica seeds are derived from (seed, replica index): with one shared seed every shard would draw the same noise
            # for its k-th batch (correlated outputs across GPUs)
            rseed = None if seed is None else int(np.random.SeedSequence([int(seed), len(self._replicas)]).generate_state(1, np.uint64)[0] >> 1)
            self._replicas.append(SmallTTS(engine=eng, num_steps=num_steps, seed=rseed))

    def _next_seed(self) -> int:
        # the reference draws noise from numpy's global RNG (infer/onnx.py:104); seeding numpy (or seed=)
        # therefore makes runs reproducible here too, while the normals themselves come from the GPU
        if self._rng is not None:
            return int(self._rng.integers(0, 2 ** 63 - 1))
        return int(np.random.randint(0, 2 ** 31 - 1)) * 2654435761 % (2 ** 63)

    def synthesize_batch(self, ref_latents: Sequence[np.ndarray], phoneme_ids: Sequence[Sequence[int]],
                         durations, *, noise: Optional[np.ndarray] = None, return_latents: bool = False,
                         frames: Optional[Sequence[int]] = None, _defer: bool = False):
        """Batched synthesize: per-utterance (R_i,64) refs, token lists and durations -> list of (1, samples).
        `frames` overrides the per-utterance frame counts (default floor(duration * 7.5), infer/onnx.py:84; the HTTP server
        rounds up like the reference's Rust server, pipeline.rs:66)."""
        B = len(ref_latents)
        if B == 0:
            return []
        if np.isscalar(durations):
            durations = [float(durations)] * B
        ns = [int(f) for f in frames] if frames is not None else [_frames(d) for d in durations]
        rs = [int(np.asarray(r).shape[0]) for r in ref_latents]
        ps = [len(p) for p in phoneme_ids]
        Rm, Pm, Nm = max(max(rs), 1), max(max(ps), 1), max(ns)
        ref = np.zeros((B, Rm, 64), np.float32)
        ids = np.zeros((B, Pm), np.int64)
        pm = np.zeros((B, Pm), bool)
        mask = np.zeros((B, Nm), bool)
        for b in range(B):
            ref[b, :rs[b]] = np.asarray(ref_latents[b], np.float32)
            ids[b, :ps[b]] = np.asarray(list(phoneme_ids[b]), np.int64)
            pm[b, :ps[b]] = True
            mask[b, :ns[b]] = True
        eng = self.engine
        seed = self._next_seed()

        def run():
            cache = eng.cond_encode(ref, np.asarray(rs, np.int64), ids, pm)
            x_ = eng.sample(cache, mask, num_steps=self.num_steps, noise=noise, seed=seed)
            return eng.codec_decode(x_), x_                    # (B, 1, HOP * Nm); causal => prefixes are exact

        audio, x = run()
        if _defer:                                             # synthesize_batches: stay on the device / stream
            return audio, x, ns, run
        audio = audio.cpu().numpy()
        if eng.check_fp16_range("synthesize"):                 # an fp16 operand clipped: the site is split-bf16 now, run again
            audio, x = run()
            audio = audio.cpu().numpy()
        outs = [audio[b, :, : HOP_SIZE * ns[b]] for b in range(B)]
        if return_latents:
            xl = x.cpu().numpy()
            return outs, [xl[b, : ns[b]] for b in range(B)]
        return outs

    def synthesize_batches(self, batches: Sequence[tuple], in_flight: int = 3, release_workspaces: bool = False) -> List[list]:
        """Several independent batches, `in_flight` of them overlapping on the GPU.

        batches: [(ref_latents, phoneme_ids, durations), ...] as for synthesize_batch.  Batch i runs whole on HIP stream
        i % in_flight with its own workspace, so one batch's latency-bound phases (condition encoders, DiT) fill the
        CUs another batch's kernels leave idle (bench.py: 16 ms per 8 x 10 s batch against 20 ms one at a time).
        The engine runs in throughput tuning meanwhile; results equal a loop of synthesize_batch under that tuning bit for bit."""
        eng = self.engine
        if in_flight <= 1 or len(batches) <= 1:
            return [self.synthesize_batch(*b) for b in batches]
        dev = eng.device
        cur = torch.cuda.current_stream(dev)
        streams = [torch.cuda.Stream(dev) for _ in range(min(in_flight, len(batches)))]
        for st in streams:
            st.wait_stream(cur)
        pending = []
        # throughput tuning: unsplit GEMMs, capped persistent codec grids, no engine side stream (it would serialise the text encoders of all
        # batches in flight); the caller's mode is restored afterwards
        prev_tuning = eng.set_tuning("throughput")
        try:
            for i, (refs, toks, durs) in enumerate(batches):
                with torch.cuda.stream(streams[i % len(streams)]):
                    eng.use_workspace(f"batch{i % len(streams)}")
                    pending.append(self.synthesize_batch(refs, toks, durs, _defer=True))
        finally:
            eng.use_workspace(None)
            eng.set_tuning(prev_tuning)
        for st in streams:
            cur.wait_stream(st)
        torch.cuda.synchronize(dev)
        if eng.check_fp16_range("synthesize_batches"):         # clipped somewhere: every batch again, one at a time, at the demoted precision
            pending = [(*run(), ns, run) for _a, _x, ns, run in pending]
        outs = []
        for audio, _, ns, _run in pending:
            a = audio.cpu().numpy()
            outs.append([a[b, :, : HOP_SIZE * ns[b]] for b in range(len(ns))])
        if release_workspaces:
            torch.cuda.synchronize(dev)
            eng.release_workspaces()
        return outs

    def synthesize_sharded(self, ref_latents: Sequence[np.ndarray], phoneme_ids: Sequence[Sequence[int]],
                           duration_sec: float, *, max_batch: int = 8) -> np.ndarray:
        """Data-parallel synthesis of a request list with a common duration -> (n, 1, samples) fp32 on the host.

        * under torch.distributed (one process per GPU, e.g. `python -m torch.distributed.run --nproc-per-node 8`): every rank
          passes the same list, synthesises its contiguous shard and receives the whole batch after ONE RCCL all-gather
          (`smalltts_amd.parallel.ShardContext`; the configuration `bench.py --gpus N` measures);
        * otherwise over `device_ids` from this process (one engine + one host thread per GPU, gathered on the host);
        * with one GPU: plain batches of `max_batch`."""
        from . import parallel
        n = len(ref_latents)
        S = _frames(duration_sec) * HOP_SIZE

        def run_span(tts: "SmallTTS", lo: int, hi: int) -> list:
            outs: list = []
            with torch.cuda.device(tts.engine.device):
                for s0 in range(lo, hi, max_batch):
                    s1 = min(hi, s0 + max_batch)
                    outs.extend(tts.synthesize_batch(list(ref_latents[s0:s1]), list(phoneme_ids[s0:s1]), duration_sec))
            return outs

        ctx = parallel.ShardContext.current()
        if ctx.world > 1:
            full = parallel.synthesize_sharded(lambda r, p, d: run_span(self, *ctx.my_shard(n)), ref_latents, phoneme_ids,
                                               duration_sec, ctx=ctx)
            return full.cpu().numpy()
        if len(self._replicas) > 1:
            outs = parallel.run_shards_in_threads([lambda lo, hi, t=t: run_span(t, lo, hi) for t in self._replicas], n)
        else:
            outs = run_span(self, 0, n)
        return np.stack(outs) if outs else np.zeros((0, 1, S), np.float32)

    def synthesize(self, ref_latents: np.ndarray, phoneme_ids: list, duration_sec: float) -> np.ndarray:
        """ref_latents (T,64) f32, phoneme ids, duration -> audio (1, samples) f32 @ 24 kHz."""
        return self.synthesize_batch([ref_latents], [phoneme_ids], [duration_sec])[0]

    def forward(self, conditionings: List[torch.Tensor], transcriptions: list, texts: list,
                duration_sec: float = 3.0) -> List[torch.Tensor]:
        from .phonemes import get_token_ids
        toks = []
        for trans, text in zip(transcriptions, texts):
            a = get_token_ids(trans) if isinstance(trans, str) else list(map(int, trans))
            b = get_token_ids(text) if isinstance(text, str) else list(map(int, text))
            toks.append(a + b)                                 # transcription + text, infer/onnx.py:144-152
        n = min(len(conditionings), len(toks))
        refs = [c.detach().cpu().numpy().astype(np.float32) for c in conditionings[:n]]
        outs = self.synthesize_batch(refs, toks[:n], duration_sec)
        return [torch.from_numpy(np.ascontiguousarray(o)) for o in outs]

    __call__ = forward


class _CodecRunner:
    _part = ""

    def __init__(self, path: str, providers: Optional[Iterable[str]] = None, *, weights: Optional[str] = None,
                 device: int = 0, precision: str = DEFAULT_PRECISION, engine: Optional[HipEngine] = None) -> None:
        self.engine = engine or get_engine(weights, device, precision, parts=("dit", "decoder", "encoder"))
        if not self.engine.has(self._part):
            raise RuntimeError(f"codec {self._part} weights are not loaded")


class Decoder(_CodecRunner):
    _part = "decoder"

    def __init__(self, path: str = "assets/codec/decoder.onnx", providers: Optional[Iterable[str]] = None, **kw):
        super().__init__(path, providers, **kw)

    def decode(self, latents: torch.Tensor) -> torch.Tensor:
        """latents f32 (batch, T, 64) -> audio f32 (batch, 1, 3200*T), returned on the CPU like the reference."""
        return self.engine.codec_decode(latents.detach()).cpu()


class Encoder(_CodecRunner):
    _part = "encoder"

    def __init__(self, path: str = "assets/codec/encoder.onnx", providers: Optional[Iterable[str]] = None, **kw):
        super().__init__(path, providers, **kw)

    def encode(self, audio: torch.Tensor) -> torch.Tensor:
        """audio f32 (batch, 1, time) @ 24 kHz -> latents f32 (batch, time // 3200, 64), on the CPU."""
        return self.engine.codec_encode(audio.detach()).cpu()

    # reference voices are encoded once per voice (clone.py:36, interactive.py:34): latents cached by content hash
    _ref_cache: "collections.OrderedDict[bytes, torch.Tensor]" = None  # type: ignore[assignment]
    REF_CACHE_ENTRIES = 64

    def encode_reference(self, audio: torch.Tensor) -> torch.Tensor:
        """Like encode() for ONE reference clip (1, 1, time), memoised on the samples' digest."""
        if Encoder._ref_cache is None:
            Encoder._ref_cache = collections.OrderedDict()
        a = audio.detach().to(torch.float32).contiguous().cpu()
        key = hashlib.blake2b(a.numpy().tobytes(), digest_size=16, person=str(tuple(a.shape)).encode()[:16]).digest()
        key += str(id(self.engine)).encode()
        hit = Encoder._ref_cache.get(key)
        if hit is not None:
            Encoder._ref_cache.move_to_end(key)
            return hit
        lat = self.encode(a)
        Encoder._ref_cache[key] = lat
        while len(Encoder._ref_cache) > self.REF_CACHE_ENTRIES:
            Encoder._ref_cache.popitem(last=False)
        return lat
