"""GPU: the request front-end end to end: 16 concurrent POST /synthesize requests with different voices,
token counts and durations are packed into padded batches, kept in flight on several streams, and every client gets the audio
`SmallTTS.synthesize_batch` produces for its request alone with the same per-request noise stream."""
import http.client
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

from smalltts_amd import server as S
from smalltts_amd.weights import CodecSpec

pytestmark = pytest.mark.gpu
SPEC = CodecSpec(n_filters=8, ratios=(8, 5, 5, 4, 2, 2), dec_depths=(1, 1, 1, 1, 1, 1, 1))


def snr_db(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return 10 * np.log10((ref ** 2).sum() / max(((got - ref) ** 2).sum(), 1e-300))


def _post(port, wav, tokens, duration, seed):
    bd = "----t"
    body = (f"--{bd}\r\nContent-Disposition: form-data; name=\"audio\"; filename=\"r.wav\"\r\n\r\n".encode() + wav + b"\r\n"
            + f"--{bd}\r\nContent-Disposition: form-data; name=\"tokens\"\r\n\r\n{tokens}\r\n--{bd}--\r\n".encode())
    c = http.client.HTTPConnection("127.0.0.1", port, timeout=120)
    c.request("POST", f"/synthesize?duration={duration}&seed={seed}", body=body,
              headers={"content-type": f"multipart/form-data; boundary={bd}"})
    r = c.getresponse()
    data = r.read()
    c.close()
    return r.status, data


@pytest.mark.parametrize("n_req,max_pack", [(16, 8), (48, 24)])
def test_concurrent_requests_get_their_own_audio(n_req, max_pack):
    """16 requests packed into batches of at most 8 (the headline configuration), and 48 requests with a deep queue: the dispatcher
    then packs up to 24 utterances into one padded batch (the DiT's GEMMs run ~2.8x more efficiently at 1800 rows than at 600)."""
    import time
    from smalltts_amd.api import Encoder, SmallTTS
    from smalltts_amd.engine import HipEngine
    eng = HipEngine(0, "bf16x3")
    eng.load_synthetic(5, parts=("dit", "decoder", "encoder"), codec_spec=SPEC)
    eng.finalize()
    tts, enc = SmallTTS(engine=eng, seed=0), Encoder(engine=eng)
    batcher = S.Batcher(tts, enc, max_batch=8, window_ms=30.0, in_flight=3, num_steps=4, max_pack=max_pack)
    from http.server import ThreadingHTTPServer

    class Srv(ThreadingHTTPServer):
        request_queue_size = 256          # 48 clients connect at once: the default backlog of 5 resets some of them
    httpd = Srv(("127.0.0.1", 0), S.make_handler(batcher, tokenizer="chars"))
    httpd.daemon_threads = True
    th = threading.Thread(target=httpd.serve_forever, kwargs={"poll_interval": 0.02}, daemon=True)
    th.start()
    port = httpd.server_address[1]
    rng = np.random.default_rng(0)
    reqs = []
    for i in range(n_req):
        sr = (16000, 24000, 44100)[i % 3]
        t = np.arange(int((0.5 + 0.1 * (i % 4)) * sr)) / sr
        voice = 0.4 * np.sin(2 * np.pi * (220 + 40 * (i % 5)) * t) + 0.02 * rng.standard_normal(t.size)   # 5 distinct voices x 3 rates
        wav = S.encode_wav(voice, sr) if i < 8 else reqs[i % 8][0]      # later requests re-use the first eight voices (cache)
        reqs.append((wav, [int(v) for v in rng.integers(1, 198, size=4 + i % 7)], round(0.3 + 0.17 * (i % 6), 2), 1000 + i))
    try:
        t0 = time.perf_counter()
        with ThreadPoolExecutor(n_req) as pool:
            outs = list(pool.map(lambda r: _post(port, *r), reqs))
        dt = time.perf_counter() - t0
    finally:
        httpd.shutdown()
        httpd.server_close()
        batcher.close()
    st = batcher.stats
    print(f"\n[server] {n_req} concurrent requests, max_pack {max_pack}: {n_req / dt:.1f} requests/s, {st['batches']} batches, "
          f"largest {st['max_batch_seen']}")
    assert st["requests"] == n_req and st["batches"] < n_req and st["max_batch_seen"] > 1, st      # requests really shared batches
    assert st["max_batch_seen"] <= max_pack and (max_pack == 8 or st["max_batch_seen"] > 8), st   # a deep queue is packed beyond 8
    # the fp16 range counters were read (whatever the queue held) and /stats states the precision actually in force
    assert st["range_checks"] >= 1 and st["precision"]["preset"] and st["precision"]["demoted"] == {}, st
    for (wav, toks, dur, seed), (code, data) in zip(reqs, outs):
        assert code == 200, data[:200]
        n = S.frames_for(dur)
        pcm = np.frombuffer(data[44:], "<i2")
        assert data[:4] == b"RIFF" and pcm.size == 3200 * n
        # the same request alone through the Python API, same per-request noise stream
        y, sr = S.decode_wav_bytes(wav)
        if sr != 24000:
            y = eng.resample(y, sr, 24000).cpu().numpy()
        lat = enc.encode(torch.from_numpy(np.ascontiguousarray(y[: len(y) // 3200 * 3200]))[None, None])[0].numpy()
        noise = torch.stack([eng.randn(n * 64, seed, s).view(1, n, 64) for s in range(4)]).cpu().numpy()
        want = tts.synthesize_batch([lat], [toks], [dur], noise=noise, frames=[n])[0]
        got = pcm.astype(np.float32) / 32767.0
        ref = np.trunc(np.clip(want.reshape(-1), -1, 1) * 32767.0) / 32767.0
        # batch-of-k vs alone and throughput vs latency tuning differ by fp32 summation order; PCM16 truncation can flip an LSB
        assert snr_db(got, ref) > 55.0, snr_db(got, ref)
    assert st["ref_cache_hits"] > 0
    eng.close()
