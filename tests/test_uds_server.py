"""not-gpu tier: the UDS front end (dusk_blindbidproof_amd/server/) against a STUB engine -- framing, opcode dispatch, the
reference's error behaviour (src/futures/main.rs:64-110: prove error -> nothing written, verify failure or malformed -> [0x00],
unknown opcode -> nothing written) and micro-batching of concurrent connections through the product's call combiner.
The GPU tier runs the same client against the real engine (tests/test_gpu_uds.py)."""
import os
import re
import signal
import subprocess
import tempfile
import threading
import time

import pytest

from tests import uds_client as uc


@pytest.fixture(scope="module")
def server(built):
    built.build_server()
    stub = built.build_stub_engine()
    d = tempfile.mkdtemp(prefix="bbp-uds-")
    path = os.path.join(d, "sock")
    err = open(os.path.join(d, "log"), "w+")
    p = subprocess.Popen([built.SERVER_BIN, "-b", path, "-l", "info", "--engine", stub, "--window-us", "2000"], stderr=err)
    for _ in range(200):
        if os.path.exists(path):
            break
        time.sleep(0.02)
    assert os.path.exists(path), "server did not bind"
    yield {"path": path, "proc": p, "log": err}
    if p.poll() is None:
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=10)


def _bid(i, n):
    s7 = b"".join(bytes([(7 * i + k) & 0xff]) * 31 + b"\x01" for k in range(7))
    pub = b"".join(bytes([(11 * i + j) & 0xff]) * 31 + b"\x02" for j in range(n))
    return s7, pub, i % n


def test_prove_then_verify_over_the_socket(server):
    s7, pub, toggle = _bid(3, 8)
    blob = uc.prove(server["path"], s7, pub, toggle)
    assert blob is not None
    proof, c, t = uc.decode_proof(blob)
    assert len(proof) == 1121 and proof[0] == 0 and len(c) == 4 and len(t) == 8 and all(len(x) == 32 for x in c + t)
    score, z_img, seed = s7[128:160], s7[160:192], s7[192:224]
    assert uc.verify(server["path"], blob, score, z_img, seed, pub) == b"\x01"
    bad = bytearray(blob)
    bad[100] ^= 1
    assert uc.verify(server["path"], bytes(bad), score, z_img, seed, pub) == b"\x00"          # verification failure
    assert uc.verify(server["path"], blob, z_img, z_img, seed, pub) == b"\x00"                 # wrong score


def test_verify_answers_0x00_to_anything_malformed(server):
    """main.rs:94-101: `Verify::try_from_reader_variables(..).and_then(verify).is_ok()` -- parse errors are a 0x00 reply too."""
    s7, pub, toggle = _bid(5, 3)
    blob = uc.prove(server["path"], s7, pub, toggle)
    score, z_img, seed = s7[128:160], s7[160:192], s7[192:224]
    for frame in (uc.tlv(b"\x02"),                                                   # nothing after the opcode
                  uc.tlv(b"\x02" + b"\x09garbage"),                                  # not even a TLV header
                  uc.tlv(b"\x02" + uc.tlv(blob) + uc.tlv(score)),                    # truncated
                  uc.tlv(b"\x02" + uc.tlv(blob) + uc.tlv(score[:31]) + uc.tlv(z_img) + uc.tlv(seed) + uc.tlv_list([pub[:32]] * 3)),  # 31-byte scalar
                  uc.tlv(b"\x02" + uc.tlv(blob) + uc.tlv(score) + uc.tlv(z_img) + uc.tlv(seed) + uc.tlv_list([pub[:31]] * 3)),       # 31-byte list item
                  uc.tlv(b"\x02" + uc.tlv(blob) + uc.tlv(score) + uc.tlv(z_img) + uc.tlv(seed) + uc.tlv_list([pub[:32]] * 2))):      # list shorter than t_c
        c = uc.Conn(server["path"])
        c.send(frame)
        assert c.recv_frame() == b"\x00"
        c.close()
    # a commitment that is not 32 bytes (proof.rs:158-162) and a blob with three commitments
    proof, cm, t = uc.decode_proof(blob)
    for forged in (uc.tlv(proof) + uc.tlv_list([cm[0][:31]] + cm[1:]) + uc.tlv_list(t), uc.tlv(proof) + uc.tlv_list(cm[:3]) + uc.tlv_list(t)):
        assert uc.verify(server["path"], forged, score, z_img, seed, pub) == b"\x00"
    # extra public-list entries beyond the toggle commitments are never read by the gadget (gadgets.rs:97-131)
    assert uc.verify(server["path"], blob, score, z_img, seed, pub + pub[:32]) == b"\x01"


def test_prove_errors_write_nothing(server):
    """main.rs:86-92, 15-25: any prove-side error resolves to Message::Error -- the peer sees the connection close, no payload."""
    s7, pub, toggle = _bid(9, 4)
    nc = bytearray(s7)
    nc[31] = 0xff                                                              # non-canonical d: serde Scalar refuses it
    bad_frames = [uc.prove_request(bytes(nc), pub, toggle),
                  uc.prove_request(s7, pub, 4),                               # toggle >= N: typed API cannot express it
                  uc.tlv(b"\x01" + b"".join(uc.tlv(s7[32 * i:32 * i + 32]) for i in range(7)) + uc.tlv_list([pub[:31]]) + uc.tlv(bytes(8))),  # 31-byte bid
                  uc.tlv(b"\x01" + b"".join(uc.tlv(s7[32 * i:32 * i + 32]) for i in range(6))),                                             # six scalars
                  uc.tlv(b"\x01" + b"".join(uc.tlv(s7[32 * i:32 * i + 32]) for i in range(7)) + uc.tlv_list([]) + uc.tlv(bytes(8))),       # empty list
                  uc.tlv(b"\x07hello"),                                       # undefined operation code (main.rs:102-105)
                  uc.tlv(b""),                                                # empty request
                  b"\x03\x00\x00\x00"]                                        # not a frame header at all
    for f in bad_frames:
        c = uc.Conn(server["path"])
        c.send(f)
        assert c.recv_frame() is None
        c.close()
    assert uc.prove(server["path"], s7, pub, toggle) is not None               # and the server is still serving


def test_slow_partial_and_abandoned_clients(server):
    """A frame that trickles in a few bytes at a time is served; clients that hang up mid-header, mid-payload or right after
    sending (never reading the reply) cost the server nothing: it keeps answering everyone else."""
    s7, pub, toggle = _bid(77, 4)
    frame = uc.prove_request(s7, pub, toggle)
    c = uc.Conn(server["path"])
    for i in range(0, len(frame), 37):
        c.send(frame[i:i + 37])
        time.sleep(0.002)
    blob = c.recv_frame()
    assert blob is not None and len(uc.decode_proof(blob)[2]) == 4
    c.close()
    for cut in (1, 2, 50, len(frame) - 1):                      # hang up mid-header / mid-payload
        c = uc.Conn(server["path"])
        c.send(frame[:cut])
        c.close()
    for _ in range(20):                                          # fire and forget: the reply has nowhere to go
        c = uc.Conn(server["path"])
        c.send(frame)
        c.close()
    idle = [uc.Conn(server["path"]) for _ in range(50)]         # connected, silent
    assert uc.verify(server["path"], blob, s7[128:160], s7[160:192], s7[192:224], pub) == b"\x01"
    for c in idle:
        c.close()
    assert uc.prove(server["path"], s7, pub, toggle) is not None


def test_one_connection_may_carry_several_requests(server):
    s7, pub, toggle = _bid(21, 2)
    c = uc.Conn(server["path"])
    for _ in range(3):
        c.send(uc.prove_request(s7, pub, toggle))
        blob = c.recv_frame()
        c.send(uc.verify_request(blob, s7[128:160], s7[160:192], s7[192:224], pub))
        assert c.recv_frame() == b"\x01"
    c.close()


def test_concurrent_connections_are_micro_batched(server):
    """48 connections at once: every reply correct, and the device-call count the server logs at shutdown is far below the
    request count -- concurrent connections became batches (this must be the LAST test: it stops the server to read the log)."""
    T, per, N = 48, 4, 8
    errors, replies = [], {}

    def worker(t):
        try:
            for j in range(per):
                s7, pub, toggle = _bid(100 + t * per + j, N)
                blob = uc.prove(server["path"], s7, pub, toggle)
                ok = uc.verify(server["path"], blob, s7[128:160], s7[160:192], s7[192:224], pub)
                replies[(t, j)] = (blob is not None and len(uc.decode_proof(blob)[2]) == N, ok)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:3]
    assert len(replies) == T * per and all(v == (True, b"\x01") for v in replies.values())
    server["proc"].send_signal(signal.SIGTERM)
    server["proc"].wait(timeout=10)
    server["log"].seek(0)
    log = server["log"].read()
    m = re.search(r"served (\d+) requests \((\d+) errors\) in (\d+) device calls, largest batch (\d+)", log)
    assert m, log[-500:]
    served, errs, calls, biggest = map(int, m.groups())
    assert served >= 2 * T * per and calls < served // 2 and biggest >= 4, m.group(0)


def test_combiner_keeps_two_batches_in_flight(built):
    """The call combiner (csrc/submit.cpp, the product's own code behind the stub) lets TWO combined calls run at once -- what
    keeps the engine's cross-call pipeline full -- never more, loses no request, and bounds every batch by max_batch."""
    import ctypes
    L = ctypes.CDLL(built.build_stub_engine())
    h = ctypes.c_void_p()
    L.bbp_init.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
    assert L.bbp_init(0, ctypes.byref(h)) == 0
    L.bbp_set_batching.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    L.bbp_set_batching(h, 100, 8)
    L.bbp_prove.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.bbp_verify.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    L.stub_max_concurrency.argtypes = [ctypes.c_void_p]
    T, per, errors = 48, 6, []

    def worker(t):
        try:
            for j in range(per):
                n = 2 + (t + j) % 3                                        # three list lengths = three request classes in the queue
                s7, pub, toggle = _bid(t * per + j, n)
                out = (ctypes.c_uint8 * (1121 + 32 * (4 + n)))()
                assert L.bbp_prove(h, s7, pub, n, toggle, None, out, None) == 0
                assert L.bbp_verify(h, bytes(out), len(out), s7[128:160], s7[160:192], s7[192:224], pub, n) == 0
                bad = bytearray(bytes(out))
                bad[7] ^= 1
                assert L.bbp_verify(h, bytes(bad), len(out), s7[128:160], s7[160:192], s7[192:224], pub, n) == 1
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:3]
    calls, reqs, biggest = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint32()
    L.bbp_batching_stats.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 3
    L.bbp_batching_stats(h, ctypes.byref(calls), ctypes.byref(reqs), ctypes.byref(biggest))
    assert reqs.value == T * per * 3 and biggest.value <= 8 and calls.value < reqs.value
    assert L.stub_max_concurrency(h) == 2


def test_combiner_under_thread_sanitizer(built):
    """csrc/submit.cpp compiled with -fsanitize=thread behind a stand-in engine: 48 threads, three request classes, prove and
    verify mixed, a batching window and a stagger -- every answer right, one class per batch, max_batch respected, at most two
    batches in flight, and no data race reported by the sanitizer."""
    p = subprocess.run([built.build_combiner_tsan()], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr[-3000:]
    assert "WARNING: ThreadSanitizer" not in p.stderr, p.stderr[-3000:]
    assert "wrong 0, bad batches 0" in p.stdout, p.stdout


def test_combiner_waits_out_a_lopsided_pair_of_batches(built):
    """Closed-loop callers (every reply answered by the next request) that start with a small batch in flight beside a large one:
    without the lopsided-pair rule that state perpetuates itself -- the small batch's callers come back early and are sent again
    at once, 100 / 500 for the whole run -- with it the few wait for the large batch's callers and the burst is cut in half: equal
    batches from then on (csrc/submit.cpp; measured through the UDS server in profiles/r03_uds_lopsided_ab.jsonl)."""
    import re
    exe = built.build_combiner_rules()
    def once():
        res = {}
        for rule in (1, 0):
            p = subprocess.run([exe, str(rule)], capture_output=True, text=True, timeout=120)
            assert p.returncode == 0, p.stdout + p.stderr[-2000:]
            m = re.search(r"RESULT min (\d+) max (\d+) wrong 0 live 0", p.stdout)
            assert m, p.stdout
            res[rule] = (int(m.group(1)), int(m.group(2)))
        return res
    # the harness is 600 real threads timed in hundreds of microseconds: on a loaded 8-core box one run in a few dozen lands a batch
    # boundary elsewhere.  Correctness (wrong 0, live 0) is asserted in every attempt; the batch-size pattern in one of three.
    for attempt in range(3):
        res = once()
        if res[1][0] >= 200 and res[1][1] <= 400 and res[0][0] * 2 < res[0][1]:
            break
    assert res[1][0] >= 200 and res[1][1] <= 400, res   # 600 callers in two equal batches
    assert res[0][0] * 2 < res[0][1], res               # (the harness does reproduce the state the rule is for)


def test_connections_are_dealt_over_several_device_contexts(built):
    """--devices a,b,..: one engine context per GPU, connections round-robin (independent bids: no cross-GPU traffic on this
    path).  With the stub both contexts are fakes; the plumbing -- two contexts created, both used, statistics summed -- is real."""
    built.build_server()
    stub = built.build_stub_engine()
    d = tempfile.mkdtemp(prefix="bbp-uds-md-")
    path, log = os.path.join(d, "sock"), open(os.path.join(d, "log"), "w+")
    p = subprocess.Popen([built.SERVER_BIN, "-b", path, "--engine", stub, "--devices", "0,1", "--window-us", "0"], stderr=log)
    try:
        for _ in range(200):
            if os.path.exists(path):
                break
            time.sleep(0.02)
        for i in range(6):                                                     # six connections: three per context
            s7, pub, toggle = _bid(40 + i, 3)
            blob = uc.prove(path, s7, pub, toggle)
            assert uc.verify(path, blob, s7[128:160], s7[160:192], s7[192:224], pub) == b"\x01"
    finally:
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=10)
    log.seek(0)
    text = log.read()
    assert "2 device context(s)" in text
    m = re.search(r"served (\d+) requests \((\d+) errors\) in (\d+) device calls", text)
    assert m and int(m.group(1)) == 12 and int(m.group(3)) == 12              # window 0, sequential client: one call per request


def test_devices_all_serves_from_a_pool_over_every_gpu(built):
    """--devices all = bbp_init(-1) = bbp_init_all: the stub pretends the node has two GPUs."""
    built.build_server()
    stub = built.build_stub_engine()
    d = tempfile.mkdtemp(prefix="bbp-uds-all-")
    path, log = os.path.join(d, "sock"), open(os.path.join(d, "log"), "w+")
    p = subprocess.Popen([built.SERVER_BIN, "-b", path, "--engine", stub, "--devices", "all", "--window-us", "0"], stderr=log)
    try:
        for _ in range(200):
            if os.path.exists(path):
                break
            time.sleep(0.02)
        for i in range(4):
            s7, pub, toggle = _bid(60 + i, 2)
            blob = uc.prove(path, s7, pub, toggle)
            assert uc.verify(path, blob, s7[128:160], s7[160:192], s7[192:224], pub) == b"\x01"
    finally:
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=10)
    log.seek(0)
    text = log.read()
    assert "2 device context(s)" in text and re.search(r"device context 1 \(device 1\): \d+ device calls", text), text[-600:]


def test_device_failure_is_fatal_and_never_answers_rejected(built):
    """once the engine's health word is set every call returns BBP_ERR_DEVICE.  A verify that was not judged must
    not be answered 0x00 ("rejected"), and the server must not stay up as a reject-all: it drops the connection, stops accepting,
    drains and exits non-zero so a supervisor starts a fresh process."""
    built.build_server()
    stub = built.build_stub_engine()
    d = tempfile.mkdtemp(prefix="bbp-uds-dead-")
    path, log = os.path.join(d, "sock"), open(os.path.join(d, "log"), "w+")
    env = dict(os.environ, STUB_DEVICE_FAIL_AFTER="3")  # calls 1 and 2 work, the third reports a dead device
    p = subprocess.Popen([built.SERVER_BIN, "-b", path, "--engine", stub, "--window-us", "0"], stderr=log, env=env)
    try:
        for _ in range(200):
            if os.path.exists(path):
                break
            time.sleep(0.02)
        s7, pub, toggle = _bid(21, 4)
        blob = uc.prove(path, s7, pub, toggle)                                                     # call 1
        assert blob is not None
        assert uc.verify(path, blob, s7[128:160], s7[160:192], s7[192:224], pub) == b"\x01"        # call 2
        reply = uc.verify(path, blob, s7[128:160], s7[160:192], s7[192:224], pub)                  # call 3: device failure
        assert reply in (None, b""), reply                                                         # nothing written -- in particular no 0x00
        assert p.wait(timeout=15) == 3
    finally:
        if p.poll() is None:
            p.kill()
            p.wait()
    log.seek(0)
    text = log.read()
    assert "device failure" in text and "exiting with status 3" in text, text[-600:]


def test_cli_mirrors_the_reference_flags(built):
    built.build_server()
    p = subprocess.run([built.SERVER_BIN, "-l", "loud"], capture_output=True, text=True)
    assert p.returncode == 2 and "invalid log level" in p.stderr           # clap possible_values (src/main.rs:33)
    p = subprocess.run([built.SERVER_BIN, "--engine", "/nonexistent.so", "-b", "/tmp/bbp-nope"], capture_output=True, text=True)
    assert p.returncode == 1 and "cannot load engine" in p.stderr


def test_load_generator_closed_and_open_loop_against_the_stub(built):
    """tools/uds_bench.py --stub: the measurement plumbing of configs[4] through the socket on a CPU box -- server (epoll front end,
    asynchronous engine calls, --reserve, a two-member stub pool), bbp-uds-loadgen in closed-loop and in open-loop (Poisson) mode.
    Not a measurement: every op must complete, none may fail or be rejected, and the open-loop run must finish its arrivals."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "uds_bench.py")
    closed = subprocess.run([sys.executable, tool, "--stub", "--connections", "96", "--ops", "1536", "--devices", "0,1"], capture_output=True, text=True, timeout=120)
    assert closed.returncode == 0, closed.stderr[-800:]
    d = json.loads(closed.stdout.strip().splitlines()[-1])
    assert d["mode"] == "closed-loop" and d["ops"] == 1536 and d["failed"] == 0 and d["rejected"] == 0
    assert d["server"]["errors"] == 0 and d["server"]["device_calls"] < d["server"]["requests"]
    opened = subprocess.run([sys.executable, tool, "--stub", "--connections", "512", "--rate", "3000", "--duration", "1.5"], capture_output=True, text=True,
                            timeout=120)
    assert opened.returncode == 0, opened.stderr[-800:]
    d = json.loads(opened.stdout.strip().splitlines()[-1])
    assert d["mode"].startswith("open-loop") and d["failed"] == 0 and d["rejected"] == 0
    assert d["ops"] == d["arrivals"] and 0.7 * 3000 * 1.5 < d["arrivals"] < 1.3 * 3000 * 1.5
    assert d["op_latency_ms"]["p50"] > 0 and d["max_client_backlog"] == 0
