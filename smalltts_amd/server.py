"""HTTP front-end for the synthesis path: the reference's `POST /synthesize?duration=N` without the x402 paywall.

Mirrors `src/server/src/main.rs:58-80,103-156`: multipart form with an `audio` part (reference voice, WAV) and a `text` part,
`duration` in the query; answers `audio/wav` (mono, 24 kHz, PCM16, `audio.rs:22-37`).  Same status codes and messages for the
same mistakes: 400 "missing 'audio'" / "missing 'text'" / "audio decode failed: ...", 500 "phonemize failed: ..." /
"inference failed: ...", 413 over the 2 MiB body limit (main.rs:87), `GET /health` -> "ok".  The frame count rounds UP like the
Rust pipeline (`pipeline.rs:66`), unlike the Python API's floor.

What is different — and the point: the reference holds one mutex around the whole inference (main.rs:24,137-146), so requests
run one at a time at batch 1.  Here handler threads only parse and wait; ONE dispatcher thread owns the engine (it is
single-threaded by contract) and
  * packs the requests that arrive within `window_ms` (up to `max_batch`) into ONE padded batch with mixed reference lengths,
    token counts and durations (`SmallTTS.synthesize_batch`),
  * keeps up to `in_flight` such batches running on their own HIP streams / workspaces in the engine's throughput tuning,
  * encodes each reference voice once (content-hash cache, `Encoder.encode_reference`),
while a completer thread waits for each batch's event, copies the audio out and wakes the handlers.
Build additions to the request: an optional `tokens` part (phoneme ids as JSON or comma separated: no espeak needed) and an
optional `seed` query parameter (per-request noise stream, so a result does not depend on which batch the request rode in).

    python -m smalltts_amd.server --port 3000 --weights assets/smalltts.smtts
"""
from __future__ import annotations

import argparse
import io
import json
import math
import queue
import struct
import threading
import time
import urllib.parse
from concurrent.futures import Future
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import List, Optional, Tuple

import numpy as np

RANGE_CHECK_BATCHES = 16      # the fp16 saturation counters are read at least this often (batches) ...
RANGE_CHECK_SECONDS = 5.0     # ... and at least this often (seconds), also when the queue never runs dry

BODY_LIMIT = 2 * 1024 * 1024     # main.rs:87 RequestBodyLimitLayer
SAMPLE_RATE, HOP, LATENT = 24_000, 3_200, 64


class HttpError(Exception):
    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code, self.msg = code, msg


# ----------------------------------------------------------------------------------------------------------------------
# request parsing (no GPU)
# ----------------------------------------------------------------------------------------------------------------------
def parse_multipart(content_type: str, body: bytes) -> dict:
    """multipart/form-data -> {field name: bytes}."""
    m = [p.strip() for p in content_type.split(";")]
    if not m or m[0].lower() != "multipart/form-data":
        raise HttpError(400, "expected multipart/form-data")
    boundary = next((p.split("=", 1)[1].strip('"') for p in m[1:] if p.lower().startswith("boundary=")), None)
    if not boundary:
        raise HttpError(400, "multipart boundary missing")
    out = {}
    for part in body.split(b"--" + boundary.encode())[1:]:
        if part.startswith(b"--"):
            break
        head, sep, data = part.lstrip(b"\r\n").partition(b"\r\n\r\n")
        if not sep:
            continue
        name = None
        for line in head.decode("utf-8", "replace").split("\r\n"):
            if line.lower().startswith("content-disposition"):
                for item in line.split(";"):
                    k, _, v = item.strip().partition("=")
                    if k == "name":
                        name = v.strip('"')
        if name:
            out[name] = data[:-2] if data.endswith(b"\r\n") else data
    return out


def decode_wav_bytes(data: bytes) -> Tuple[np.ndarray, int]:
    """-> (mono float32, sample rate); averaging channels like audio.rs:76-84."""
    import os
    import tempfile
    from .audio import read_wav
    fd, path = tempfile.mkstemp(suffix=".wav")
    try:
        with os.fdopen(fd, "wb") as f:
            f.write(data)
        y, sr = read_wav(path)
    finally:
        os.unlink(path)
    if y.ndim == 2:
        y = y.mean(axis=1)
    return y.astype(np.float32), int(sr)


def encode_wav(samples: np.ndarray, sample_rate: int = SAMPLE_RATE) -> bytes:
    """mono PCM16 WAV; `(s.clamp(-1, 1) * i16::MAX) as i16` truncates toward zero (audio.rs:33)."""
    pcm = np.trunc(np.clip(np.asarray(samples, np.float32).reshape(-1), -1.0, 1.0) * 32767.0).astype("<i2").tobytes()
    return (b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16)
            + b"data" + struct.pack("<I", len(pcm)) + pcm)


def frames_for(duration: float) -> int:
    return max(1, int(math.ceil(duration * SAMPLE_RATE / HOP)))   # pipeline.rs:66


MAX_FRAMES = 4096   # the engine's rope tables (dit.py:139): frames per utterance, tokens per text
MAX_TOKENS = 4096
MAX_REF_FRAMES = 1024   # reference voice: 137 s at 24 kHz (a 2 MiB body holds 44 s of 24 kHz PCM16, more at lower rates)
# A padded batch costs B x max(frames) (x max(tokens), x max(ref frames)) whatever its members asked for: the packer bounds THAT,
# not the request count.  2400 padded frames = 32 utterances of 10 s: ~10 GB of codec workspace per batch in flight; one request
# above the budget (up to MAX_FRAMES) runs alone.  (24 x 4096 frames would have been ~100 GB and a 500 for all 24.)
PACK_FRAMES = 2400
PACK_TOKENS = 8192
PACK_REF_FRAMES = 2400
PACK_LENGTH_RATIO = 4.0   # frames of the longest / shortest member of one batch: short requests do not wait on long ones


def plan_batches(ns, ps, rs, max_pack, frames=PACK_FRAMES, tokens=PACK_TOKENS, ref_frames=PACK_REF_FRAMES, ratio=PACK_LENGTH_RATIO):
    """Greedy partition, in arrival order, of requests with frame counts `ns`, token counts `ps` and reference frame counts `rs`
    into batches whose PADDED work stays inside the budgets.  Returns lists of indices; every index appears exactly once."""
    groups, cur = [], []
    for i in range(len(ns)):
        trial = cur + [i]
        n_hi, n_lo = max(ns[j] for j in trial), min(ns[j] for j in trial)
        fits = (len(trial) <= max_pack and len(trial) * n_hi <= frames and len(trial) * max(ps[j] for j in trial) <= tokens
                and len(trial) * max(rs[j] for j in trial) <= ref_frames and n_hi <= ratio * n_lo)
        if cur and not fits:
            groups.append(cur)
            cur = [i]
        else:
            cur = trial
    if cur:
        groups.append(cur)
    return groups


def validate_request(duration: float, tokens) -> int:
    """Per-request limits, checked BEFORE a request may join a shared batch (the reference server runs batch 1, so a bad request
    only hurts itself there; here it would take its batch-mates down with it).  Returns the frame count; raises HttpError(400)."""
    if not isinstance(duration, (int, float)) or not math.isfinite(duration) or duration <= 0:
        raise HttpError(400, "invalid `duration`: must be a finite number of seconds > 0")
    if duration * SAMPLE_RATE / HOP > MAX_FRAMES:
        raise HttpError(400, f"invalid `duration`: at most {MAX_FRAMES * HOP / SAMPLE_RATE:.0f} s ({MAX_FRAMES} frames)")
    if not 1 <= len(tokens) <= MAX_TOKENS:
        raise HttpError(400, f"invalid text: {len(tokens)} tokens (1 .. {MAX_TOKENS} allowed)")
    return frames_for(duration)


class Request:
    __slots__ = ("wav", "sr", "tokens", "duration", "seed", "future", "t_in")

    def __init__(self, wav, sr, tokens, duration, seed):
        self.wav, self.sr, self.tokens, self.duration, self.seed = wav, sr, tokens, duration, seed
        self.future: Future = Future()
        self.t_in = time.perf_counter()


# ----------------------------------------------------------------------------------------------------------------------
# the batching dispatcher (owns the engine)
# ----------------------------------------------------------------------------------------------------------------------
class Batcher:
    def __init__(self, tts, encoder, max_batch: int = 8, window_ms: float = 4.0, in_flight: int = 3, num_steps: int = 4,
                 max_pack: int = 24):
        import torch
        self.torch = torch
        self.tts, self.enc, self.eng = tts, encoder, tts.engine
        self.max_batch, self.window, self.in_flight, self.steps = int(max_batch), window_ms * 1e-3, max(1, int(in_flight)), num_steps
        # When the queue is deep the dispatcher packs more than max_batch utterances into one padded batch (the DiT's GEMMs run
        # ~2.8x more efficiently at 1800 rows than at 600, NOTEBOOK §5); it never WAITS for more than max_batch.
        self.max_pack = max(int(max_pack), self.max_batch)
        self.q: "queue.Queue[Optional[Request]]" = queue.Queue()
        self.done_q: "queue.Queue" = queue.Queue()
        self.stats = {"requests": 0, "batches": 0, "max_batch_seen": 0, "ref_cache_hits": 0}
        self._slots = threading.Semaphore(self.in_flight)
        self._threads = [threading.Thread(target=self._dispatch, daemon=True), threading.Thread(target=self._complete, daemon=True)]
        for t in self._threads:
            t.start()

    def submit(self, req: Request) -> Future:
        self.q.put(req)
        return req.future

    def close(self):
        self.q.put(None)
        for t in self._threads:
            t.join(timeout=30)

    # -- dispatcher thread: every engine call happens here ---------------------------------------------------------------
    def _gather(self) -> Optional[List[Request]]:
        first = self.q.get()
        if first is None:
            return None
        reqs, deadline = [first], time.perf_counter() + self.window
        while len(reqs) < self.max_batch:
            left = deadline - time.perf_counter()
            try:
                r = self.q.get(timeout=max(left, 0.0)) if left > 0 else self.q.get_nowait()
            except queue.Empty:
                break
            if r is None:
                self.q.put(None)
                break
            reqs.append(r)
        while len(reqs) < self.max_pack:   # deep queue: take what is already waiting, without waiting for more
            try:
                r = self.q.get_nowait()
            except queue.Empty:
                break
            if r is None:
                self.q.put(None)
                break
            reqs.append(r)
        return reqs

    def _ref_latents(self, r: Request):
        torch = self.torch
        y = r.wav
        if r.sr != SAMPLE_RATE:
            y = self.eng.resample(y, r.sr, SAMPLE_RATE).cpu().numpy()
        n = (len(y) // HOP) * HOP
        if n == 0:
            raise HttpError(400, "audio decode failed: reference shorter than one codec frame (3200 samples at 24 kHz)")
        if n // HOP > MAX_REF_FRAMES:
            raise HttpError(400, f"audio decode failed: reference longer than {MAX_REF_FRAMES * HOP / SAMPLE_RATE:.0f} s")
        before = len(type(self.enc)._ref_cache or {})
        lat = self.enc.encode_reference(torch.from_numpy(np.ascontiguousarray(y[:n]))[None, None])
        if len(type(self.enc)._ref_cache or {}) == before:
            self.stats["ref_cache_hits"] += 1
        return lat[0].numpy()

    def _dispatch(self):
        torch = self.torch
        dev = self.eng.device
        streams = [torch.cuda.Stream(dev) for _ in range(self.in_flight)]
        prev = self.eng.set_tuning("throughput")
        i = 0
        last_check_i, last_check_t = 0, time.monotonic()
        try:
            while True:
                reqs = self._gather()
                if reqs is None:
                    break
                ok: List[Request] = []
                refs, ns = [], []
                for r in reqs:
                    try:
                        n = validate_request(r.duration, r.tokens)   # (the handler checked already; a direct submit() may not have)
                        refs.append(self._ref_latents(r))
                        ns.append(n)
                        ok.append(r)
                    except HttpError as e:
                        r.future.set_exception(e)
                    except Exception as e:   # engine errors while encoding this voice
                        r.future.set_exception(HttpError(500, f"inference failed: {e}"))
                if not ok:
                    continue
                # padded-work budgets, not a request count: one long request neither blows the workspace up for 23 batch-mates
                # nor makes short requests wait for it (plan_batches); the groups go out back to back, each its own batch in flight
                for grp in plan_batches(ns, [len(r.tokens) for r in ok], [int(x.shape[0]) for x in refs], self.max_pack):
                    g_ok, g_refs, g_ns = [ok[j] for j in grp], [refs[j] for j in grp], [ns[j] for j in grp]
                    self._slots.acquire()                       # at most in_flight batches on the GPU
                    slot = i % self.in_flight
                    i += 1
                    try:
                        with torch.cuda.stream(streams[slot]):
                            self.eng.use_workspace(f"srv{slot}")
                            # per-request noise streams: a request's result does not depend on the batch it rides in
                            noise = torch.zeros(self.steps, len(g_ok), max(g_ns), LATENT, device=dev)
                            for b, r in enumerate(g_ok):
                                for s_ in range(self.steps):
                                    noise[s_, b, :g_ns[b]] = self.eng.randn(g_ns[b] * LATENT, r.seed, s_).view(g_ns[b], LATENT)
                            audio, _, _, _ = self.tts.synthesize_batch(g_refs, [r.tokens for r in g_ok], [r.duration for r in g_ok],
                                                                       noise=noise, frames=g_ns, _defer=True)
                            ev = torch.cuda.Event()
                            ev.record()
                        self.eng.use_workspace(None)
                        self.stats["requests"] += len(g_ok)
                        self.stats["batches"] += 1
                        self.stats["max_batch_seen"] = max(self.stats["max_batch_seen"], len(g_ok))
                        self.stats["max_padded_frames"] = max(self.stats.get("max_padded_frames", 0), len(g_ok) * max(g_ns))
                        self.done_q.put((ev, audio, g_ns, g_ok))
                    except Exception as e:
                        self.eng.use_workspace(None)
                        self._slots.release()
                        for r in g_ok:
                            r.future.set_exception(HttpError(500, f"inference failed: {e}"))
                # fp16 range guard (engine.check_fp16_range): the counters are read when the queue has run dry — a device
                # synchronisation the busy path should not pay per batch.  A site that clamped runs split-bf16 from then on; the
                # requests that hit it were answered with saturated (finite) operands, which is what the warning is for.
                # Under sustained load the queue is rarely empty — the regime where the guard matters most — so the counters are
                # also read every RANGE_CHECK_BATCHES batches / RANGE_CHECK_SECONDS seconds whatever the queue holds.
                # The read synchronises the device, i.e. drains the batches in flight: once per 16 batches / 5 s that is < 1 % of a loaded
                # server's time (one ~8 ms pipeline refill per >= 130 ms of work) — accepted, stated here.
                now = time.monotonic()
                if self.q.empty() or i - last_check_i >= RANGE_CHECK_BATCHES or now - last_check_t >= RANGE_CHECK_SECONDS:
                    last_check_i, last_check_t = i, now
                    try:
                        self.stats["range_checks"] = self.stats.get("range_checks", 0) + 1
                        if self.eng.check_fp16_range("server"):
                            self.stats["range_demotions"] = sorted(self.eng._demoted)
                        self.stats["precision"] = self.eng.precision_in_force()
                    except Exception as e:   # a failing guard must be visible (/stats), not silently skipped
                        self.stats["range_check_errors"] = self.stats.get("range_check_errors", 0) + 1
                        self.stats["range_check_last_error"] = repr(e)
        finally:
            self.eng.set_tuning(prev)
            self.done_q.put(None)

    # -- completer thread: waits for a batch, copies out, wakes the handlers ---------------------------------------------
    def _complete(self):
        while True:
            item = self.done_q.get()
            if item is None:
                break
            ev, audio, ns, reqs = item
            try:
                ev.synchronize()
                host = audio.cpu().numpy()
                for b, r in enumerate(reqs):
                    r.future.set_result(host[b, 0, : HOP * ns[b]].copy())
            except Exception as e:
                for r in reqs:
                    if not r.future.done():
                        r.future.set_exception(HttpError(500, f"inference failed: {e}"))
            finally:
                self._slots.release()


# ----------------------------------------------------------------------------------------------------------------------
# HTTP surface
# ----------------------------------------------------------------------------------------------------------------------
def make_handler(batcher: Batcher, tokenizer: str = "espeak"):
    from .phonemes import get_token_ids, parse_tokens_arg

    class Handler(BaseHTTPRequestHandler):
        protocol_version = "HTTP/1.1"

        def log_message(self, fmt, *args):   # quiet by default (the reference logs through tracing at info)
            pass

        def _send(self, code: int, body: bytes, ctype: str = "text/plain; charset=utf-8"):
            self.send_response(code)
            self.send_header("content-type", ctype)
            self.send_header("content-length", str(len(body)))
            self.send_header("access-control-allow-origin", "*")   # CorsLayer Any (main.rs:88-94)
            self.end_headers()
            self.wfile.write(body)

        def do_GET(self):
            path = urllib.parse.urlsplit(self.path).path
            if path == "/health":
                return self._send(200, b"ok")
            if path == "/stats":
                return self._send(200, json.dumps(batcher.stats).encode(), "application/json")
            self._send(404, b"not found")

        def do_POST(self):
            url = urllib.parse.urlsplit(self.path)
            if url.path != "/synthesize":
                return self._send(404, b"not found")
            try:
                q = urllib.parse.parse_qs(url.query)
                n = int(self.headers.get("content-length") or 0)
                # The body is taken off the socket before ANY answer: a client that writes its whole request before it reads
                # (http.client, curl) would otherwise see EPIPE / a reset instead of the status.  Over the limit it is discarded,
                # a bounded amount of it; beyond that the connection is simply closed after the answer.
                if n > BODY_LIMIT:
                    left = min(n, 8 * BODY_LIMIT)
                    while left > 0:
                        chunk = self.rfile.read(min(left, 1 << 16))
                        if not chunk:
                            break
                        left -= len(chunk)
                    self.close_connection = True
                    raise HttpError(413, "length limit exceeded")
                body = self.rfile.read(n)
                try:
                    duration = float(q["duration"][0])
                except Exception:
                    raise HttpError(400, "Failed to deserialize query string: missing field `duration`")
                fields = parse_multipart(self.headers.get("content-type", ""), body)
                if "audio" not in fields:
                    raise HttpError(400, "missing 'audio'")
                if "text" not in fields and "tokens" not in fields:
                    raise HttpError(400, "missing 'text'")
                try:
                    wav, sr = decode_wav_bytes(fields["audio"])
                except Exception as e:
                    raise HttpError(400, f"audio decode failed: {e}")
                try:
                    if "tokens" in fields:
                        tokens = parse_tokens_arg(fields["tokens"].decode())
                    else:
                        tokens = get_token_ids(fields["text"].decode("utf-8"), backend=tokenizer)
                except Exception as e:
                    raise HttpError(500, f"phonemize failed: {e}")
                validate_request(duration, tokens)   # 400 for the offender alone, before it can join a batch
                seed = int(q["seed"][0]) if "seed" in q else int.from_bytes(np.random.bytes(7), "little")
                fut = batcher.submit(Request(wav, sr, tokens, duration, seed))
                audio = fut.result(timeout=120)
                self._send(200, encode_wav(audio), "audio/wav")
            except HttpError as e:
                self._send(e.code, e.msg.encode())
            except Exception as e:   # pragma: no cover
                self._send(500, f"inference failed: {e}".encode())

    return Handler


def serve(host: str = "0.0.0.0", port: int = 3000, weights: Optional[str] = None, device: int = 0, precision: Optional[str] = None,
          max_batch: int = 8, window_ms: float = 4.0, in_flight: int = 3, tokenizer: str = "espeak", ready: Optional[threading.Event] = None,
          max_pack: int = 24):
    """Blocking; returns the ThreadingHTTPServer after shutdown.  `ready` is set once the socket is listening (tests)."""
    from .api import Encoder, SmallTTS
    from .engine import DEFAULT_PRECISION
    kw = dict(weights=weights, device=device, precision=precision or DEFAULT_PRECISION)
    tts = SmallTTS(**kw)
    enc = Encoder(**kw)
    batcher = Batcher(tts, enc, max_batch, window_ms, in_flight, tts.num_steps, max_pack)
    class _Server(ThreadingHTTPServer):
        request_queue_size = 256   # listen backlog: bursts of concurrent clients (the default of 5 resets connections)
    httpd = _Server((host, port), make_handler(batcher, tokenizer))
    httpd.daemon_threads = True
    httpd.batcher = batcher
    if ready is not None:
        ready.httpd = httpd
        ready.set()
    try:
        httpd.serve_forever(poll_interval=0.05)
    finally:
        batcher.close()
        httpd.server_close()
    return httpd


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="smalltts HTTP server: POST /synthesize?duration=N (multipart audio + text) -> audio/wav")
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=3000)
    ap.add_argument("--weights", default=None)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--precision", default=None)
    ap.add_argument("--max-batch", type=int, default=8)
    ap.add_argument("--window-ms", type=float, default=4.0, help="how long the dispatcher waits for more requests to share a batch")
    ap.add_argument("--in-flight", type=int, default=3, help="batches kept running concurrently on the GPU")
    ap.add_argument("--max-pack", type=int, default=24, help="utterances per padded batch when the queue is deep (>= --max-batch)")
    ap.add_argument("--tokenizer", default="espeak", choices=["espeak", "chars"])
    a = ap.parse_args(argv)
    print(f"listening on {a.host}:{a.port}")
    serve(a.host, a.port, a.weights, a.device, a.precision, a.max_batch, a.window_ms, a.in_flight, a.tokenizer, max_pack=a.max_pack)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
