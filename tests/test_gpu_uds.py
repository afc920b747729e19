"""GPU tier, end to end through the reference's IPC surface: bbp-uds-server with the REAL engine on cuda:0, concurrent
connections speaking the opcode-1 / opcode-2 protocol of src/futures/main.rs:64-110.  The server draws its own entropy (the wire
carries none, like the reference), so parity here is cross-acceptance: every proof that comes back over the socket is accepted by
the C oracle's verifier and by the server's own opcode 2, tampered ones are refused, error cases write nothing."""
import os
import re
import signal
import subprocess
import tempfile
import threading
import time

import pytest

from tests import oracle_c, uds_client as uc
from tests.test_gpu_prove_verify import _synth_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oc(built):
    return oracle_c.load(built.build_oracle())


@pytest.fixture(scope="module")
def server(built, bbp):
    built.build_server()
    d = tempfile.mkdtemp(prefix="bbp-uds-gpu-")
    path = os.path.join(d, "sock")
    err = open(os.path.join(d, "log"), "w+")
    p = subprocess.Popen([built.SERVER_BIN, "-b", path, "-l", "info", "--engine", bbp.lib_path, "--device", "0", "--window-us", "500"], stderr=err)
    for _ in range(1500):
        if os.path.exists(path) or p.poll() is not None:
            break
        time.sleep(0.02)
    assert os.path.exists(path), "server did not bind: " + open(err.name).read()[-800:]
    yield {"path": path, "proc": p, "log": err}
    if p.poll() is None:
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=30)


def test_concurrent_prove_verify_through_the_socket(ctx, oc, bbp, server):
    N, T, per = 8, 12, 2
    ins, _, vins = _synth_batch(ctx, T * per, N, seed=8088)
    out, errors = {}, []

    def worker(t):
        try:
            for j in range(per):
                i = t * per + j
                s7, pub, toggle = ins[i][:224], ins[i][224:224 + 32 * N], int.from_bytes(ins[i][-8:], "little")
                blob = uc.prove(server["path"], s7, pub, toggle)
                ok = uc.verify(server["path"], blob, *vins[i])
                bad = bytearray(blob)
                bad[300 + i] ^= 0x04
                rej = uc.verify(server["path"], bytes(bad), *vins[i])
                out[i] = (blob, ok, rej)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:3]
    seen = set()
    for i in range(T * per):
        blob, ok, rej = out[i]
        proof, c, t = uc.decode_proof(blob)
        assert len(proof) == 1121 and len(c) == 4 and len(t) == N
        record = proof + b"".join(c) + b"".join(t)
        assert oc.verify(record, *vins[i]) == 0, i          # the oracle accepts what came over the wire
        assert ctx.verify(record, *vins[i]) == 0
        assert ok == b"\x01" and rej == b"\x00", i
        seen.add(proof)
    assert len(seen) == T * per                              # OS entropy: no two proofs alike


def test_wire_error_behaviour_with_the_real_engine(ctx, server):
    N = 3
    ins, _, vins = _synth_batch(ctx, 2, N, seed=8089)
    s7, pub = ins[0][:224], ins[0][224:224 + 32 * N]
    nc = bytearray(s7)
    nc[32:64] = b"\xff" * 32                                  # k >= l: the reference's serde Scalar rejects it -> no payload
    assert uc.prove(server["path"], bytes(nc), pub, 0) is None
    assert uc.prove(server["path"], s7, pub, N) is None       # toggle >= N
    # a toggle that points at someone else's slot: the reference proves a false statement without complaint; nobody accepts it
    blob = uc.prove(server["path"], s7, pub, 1)
    assert blob is not None and uc.verify(server["path"], blob, *vins[0]) == b"\x00"
    good = uc.prove(server["path"], s7, pub, 0)
    assert uc.verify(server["path"], good, *vins[0]) == b"\x01"
    plus_l = (int.from_bytes(vins[0][0], "little") + 2**252 + 27742317777372353535851937790883648493).to_bytes(32, "little")
    assert uc.verify(server["path"], good, plus_l, *vins[0][1:]) == b"\x00"   # non-canonical score: FormatError -> 0x00
    c = uc.Conn(server["path"])
    c.send(uc.tlv(b"\x09"))
    assert c.recv_frame() is None                             # undefined operation code
    c.close()


def test_server_log_reports_batched_device_calls(server):
    """Last test of the module: stop the server and read its shutdown line -- concurrent connections shared device batches."""
    server["proc"].send_signal(signal.SIGTERM)
    server["proc"].wait(timeout=30)
    server["log"].seek(0)
    log = server["log"].read()
    m = re.search(r"served (\d+) requests \((\d+) errors\) in (\d+) device calls, largest batch (\d+)", log)
    assert m, log[-800:]
    served, errs, calls, biggest = map(int, m.groups())
    assert served >= 72 and calls < served and biggest >= 2, m.group(0)
