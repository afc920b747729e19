"""not-gpu tier: the multi-rank path (block split of the batch by proof index, no data-path collective, one final gather of
fixed-stride records / flags to rank 0) exercised with world_size = 2 over gloo.  The per-rank compute is stood in for by the
C oracle here (no GPU in this tier); what is under test is the partitioning and the gather, which bench.py reuses verbatim."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, N, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dusk_blindbidproof_amd import sharding
    import __graft_entry__ as ge
    from tests import oracle_c
    import json
    oc = oracle_c.load(ge.build_oracle())
    c = json.load(open(os.path.join(ROOT, "tests", "golden", "proofs_small.json")))["small"][0]
    lo, hi = sharding.shard_range(total, rank, world)
    # every "proof" of the global batch is the same small-circuit fixture with a different rng seed byte
    recs = []
    s7 = b"".join(bytes.fromhex(c[k]) for k in ["d", "k", "y", "y_inv", "q", "z_img", "seed"])
    pub = b"".join(bytes.fromhex(p) for p in c["pub_list"])
    for i in range(lo, hi):
        ent = bytearray(bytes.fromhex(c["entropy"]))
        ent[-1] = i
        rc, rec = oc.prove(s7, pub, c["toggle"], bytes(ent), c["rounds"], c["cap"])
        assert rc == 0
        recs.append(rec)
    stride = len(recs[0]) if recs else sharding.small_record_len(c)
    local = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8) if recs else torch.empty(0, dtype=torch.uint8)
    gathered = sharding.gather_records(dist, local, stride, total, rank, world)
    flags = torch.tensor([oc.verify(r, bytes.fromhex(c["q"]), bytes.fromhex(c["z_img"]), bytes.fromhex(c["seed"]), pub, c["rounds"], c["cap"])
                          for r in recs], dtype=torch.int32)
    gflags = sharding.gather_records(dist, flags.view(torch.uint8), 4, total, rank, world)
    if rank == 0:
        q.put((bytes(gathered.numpy().tobytes()), gflags.view(torch.int32).tolist(), stride))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 2, 1])
def test_two_rank_shard_and_gather(built, total):
    world, N = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    data, flags, stride = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len(data) == total * stride and flags == [0] * total
    # records arrive in global proof order: record i carries rng seed byte i -> all distinct, and equal to a direct run
    recs = [data[i * stride:(i + 1) * stride] for i in range(total)]
    assert len(set(recs)) == total


def test_shard_range_covers_everything():
    sys.path.insert(0, ROOT)
    from dusk_blindbidproof_amd import sharding
    for total in (0, 1, 7, 8, 1024, 65536, 1000003):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
