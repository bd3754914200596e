"""CPU: the HTTP surface of smalltts_amd.server (routing, multipart parsing, status codes and messages of the reference's
axum handler, src/server/src/main.rs:103-173) against a stand-in batcher — no engine, no GPU.  The reference's own e2e test
stubs the handler the same way (src/server/tests/e2e.rs:56-58)."""
import http.client
import json
import threading
from concurrent.futures import Future
from http.server import ThreadingHTTPServer

import numpy as np
import pytest

from smalltts_amd import server as S


class FakeBatcher:
    def __init__(self):
        self.stats = {"requests": 0}
        self.seen = []

    def submit(self, req):
        self.seen.append(req)
        self.stats["requests"] += 1
        f = Future()
        f.set_result(np.linspace(-1.2, 1.2, S.HOP * S.frames_for(req.duration), dtype=np.float32))
        return f


@pytest.fixture()
def srv():
    b = FakeBatcher()
    httpd = ThreadingHTTPServer(("127.0.0.1", 0), S.make_handler(b, tokenizer="chars"))
    t = threading.Thread(target=httpd.serve_forever, kwargs={"poll_interval": 0.02}, daemon=True)
    t.start()
    yield httpd.server_address[1], b
    httpd.shutdown()
    httpd.server_close()


def _multipart(fields, boundary="----smtts"):
    body = b""
    for name, (data, fname) in fields.items():
        body += f"--{boundary}\r\nContent-Disposition: form-data; name=\"{name}\"".encode()
        body += (f"; filename=\"{fname}\"\r\nContent-Type: application/octet-stream".encode() if fname else b"") + b"\r\n\r\n"
        body += data + b"\r\n"
    return body + f"--{boundary}--\r\n".encode(), f"multipart/form-data; boundary={boundary}"


def _req(port, method, path, body=None, ctype=None):
    c = http.client.HTTPConnection("127.0.0.1", port, timeout=10)
    c.request(method, path, body=body, headers={"content-type": ctype} if ctype else {})
    r = c.getresponse()
    data = r.read()
    c.close()
    return r.status, dict(r.getheaders()), data


def test_health_and_routing(srv):
    port, _ = srv
    assert _req(port, "GET", "/health")[::2] == (200, b"ok")
    assert _req(port, "GET", "/nope")[0] == 404 and _req(port, "POST", "/nope", b"x")[0] == 404
    assert json.loads(_req(port, "GET", "/stats")[2]) == {"requests": 0}


def test_synthesize_roundtrip_and_wav_encoding(srv):
    port, b = srv
    ref = S.encode_wav(0.25 * np.sin(np.arange(16000) / 7.0), 16000)
    body, ct = _multipart({"audio": (ref, "r.wav"), "text": ("it costs $5".encode(), None)})
    st, hdr, wav = _req(port, "POST", "/synthesize?duration=1.05&seed=7", body, ct)
    assert st == 200 and hdr["content-type"] == "audio/wav" and hdr["access-control-allow-origin"] == "*"
    req = b.seen[0]
    assert req.sr == 16000 and req.wav.shape == (16000,) and req.seed == 7 and req.duration == 1.05
    from smalltts_amd.phonemes import decode_token_ids
    assert decode_token_ids(req.tokens).split() == ["it", "costs", "five", "dollars"]      # normalised, then tokenised
    assert S.frames_for(1.05) == 8                                                           # ceil like pipeline.rs:66 (floor would be 7)
    assert wav[:4] == b"RIFF" and wav[8:16] == b"WAVEfmt " and len(wav) == 44 + 2 * 3200 * 8
    pcm = np.frombuffer(wav[44:], "<i2")
    assert pcm[0] == -32767 and pcm[-1] == 32767                                             # clamp, * i16::MAX, truncate (audio.rs:33)
    x = np.linspace(-1.2, 1.2, 3200 * 8, dtype=np.float32)
    assert np.array_equal(pcm, np.trunc(np.clip(x, -1, 1) * 32767).astype(np.int16))
    # pre-tokenised request (build addition): no text needed
    body, ct = _multipart({"audio": (ref, "r.wav"), "tokens": (b"[3, 5, 8]", None)})
    assert _req(port, "POST", "/synthesize?duration=0.2", body, ct)[0] == 200 and b.seen[1].tokens == [3, 5, 8]


def test_error_paths_match_the_reference_handler(srv):
    port, _ = srv
    ref = S.encode_wav(np.zeros(4800), 24000)
    body, ct = _multipart({"text": (b"hi", None)})
    assert _req(port, "POST", "/synthesize?duration=1", body, ct)[::2] == (400, b"missing 'audio'")
    body, ct = _multipart({"audio": (ref, "r.wav")})
    assert _req(port, "POST", "/synthesize?duration=1", body, ct)[::2] == (400, b"missing 'text'")
    body, ct = _multipart({"audio": (b"not a wav file", "r.wav"), "text": (b"hi", None)})
    st, _, msg = _req(port, "POST", "/synthesize?duration=1", body, ct)
    assert st == 400 and msg.startswith(b"audio decode failed: ")
    body, ct = _multipart({"audio": (ref, "r.wav"), "text": (b"hi", None)})
    assert _req(port, "POST", "/synthesize", body, ct)[0] == 400                            # duration is required (Query<SynthesizeParams>)
    assert _req(port, "POST", "/synthesize?duration=1", b"x" * 10, "text/plain")[0] == 400
    big, ct = _multipart({"audio": (b"\0" * (S.BODY_LIMIT + 1), "r.wav"), "text": (b"hi", None)})
    assert _req(port, "POST", "/synthesize?duration=1", big, ct)[0] == 413                  # RequestBodyLimitLayer 2 MiB


def test_bad_duration_or_token_count_is_a_400_for_the_offender_alone(srv):
    """limits are checked before a request may join a shared batch — nan / inf / huge durations and empty or
    over-long token lists never reach the batcher (where they would have failed every request of the batch with a 500)."""
    port, b = srv
    ref = S.encode_wav(np.zeros(4800), 24000)
    body, ct = _multipart({"audio": (ref, "r.wav"), "text": (b"hi", None)})
    for d in ("nan", "inf", "-inf", "-1", "0", "1e9", "600"):
        st, _, msg = _req(port, "POST", f"/synthesize?duration={d}", body, ct)
        assert st == 400 and msg.startswith(b"invalid `duration`"), (d, st, msg)
    body, ct = _multipart({"audio": (ref, "r.wav"), "tokens": (str(list(range(1, 4098))).encode(), None)})
    st, _, msg = _req(port, "POST", "/synthesize?duration=1", body, ct)
    assert st == 400 and msg.startswith(b"invalid text: 4097 tokens")
    body, ct = _multipart({"audio": (ref, "r.wav"), "tokens": (b"[]", None)})
    assert _req(port, "POST", "/synthesize?duration=1", body, ct)[0] == 400
    assert b.seen == []                                                  # none of them was submitted
    assert S.validate_request(546.0, [1]) == S.MAX_FRAMES - 1 and S.validate_request(0.01, [1] * 4096) == 1


def test_dispatcher_packs_a_deep_queue_but_never_waits_beyond_max_batch():
    """Batcher._gather: up to max_batch inside the window, then whatever is ALREADY queued up to max_pack."""
    import queue
    g = S.Batcher.__new__(S.Batcher)
    g.q, g.max_batch, g.max_pack, g.window = queue.Queue(), 8, 24, 0.002
    for i in range(30):
        g.q.put(i)
    assert g._gather() == list(range(24)) and g._gather() == list(range(24, 30))
    for i in range(3):
        g.q.put(i)
    assert g._gather() == [0, 1, 2]
    g.q.put(7); g.q.put(None)
    assert g._gather() == [7] and g._gather() is None


def test_packing_is_bounded_by_padded_work_not_by_request_count():
    """a padded batch costs B x max(frames); the packer bounds that (and tokens, reference frames, the length
    ratio inside a batch) instead of packing 24 requests whatever their lengths."""
    from smalltts_amd.server import PACK_FRAMES, PACK_LENGTH_RATIO, plan_batches
    # 24 ordinary 10-s requests: one batch (24 x 75 = 1800 padded frames)
    assert plan_batches([75] * 24, [30] * 24, [15] * 24, 24) == [list(range(24))]
    # one maximal request among 23 short ones: it rides alone, the short ones share batches without it
    ns = [75] * 10 + [4096] + [75] * 13
    groups = plan_batches(ns, [30] * 24, [15] * 24, 24)
    assert sorted(i for g in groups for i in g) == list(range(24))          # nobody lost, nobody twice
    assert [10] in groups
    for g in groups:
        n_hi, n_lo = max(ns[i] for i in g), min(ns[i] for i in g)
        assert len(g) == 1 or (len(g) * n_hi <= PACK_FRAMES and n_hi <= PACK_LENGTH_RATIO * n_lo), (g, n_hi, n_lo)
    # arrival order is kept inside and across groups
    flat = [i for g in groups for i in g]
    assert flat == sorted(flat)
    # token and reference budgets bind too
    assert len(plan_batches([75] * 8, [4096] * 8, [15] * 8, 24)) == 4       # 2 x 4096 tokens per batch
    assert len(plan_batches([75] * 8, [30] * 8, [1024] * 8, 24)) == 4       # 2 x 1024 reference frames per batch
    # short requests do not wait on a long one: 1-s and 30-s requests never share a batch
    groups = plan_batches([8, 225, 8, 8], [10] * 4, [15] * 4, 24)
    assert all(len({ns_ > 100 for ns_ in ([8, 225, 8, 8][i] for i in g)}) == 1 for g in groups)
