"""Batch sharding across the GPUs of a node: one process per GPU, contiguous block split by proof index, tables
replicated per rank, NO data-path collective; the only exchange is the final gather of fixed-stride proof records (or
4-byte verify flags) to rank 0 over torch.distributed (RCCL over xGMI on the GPU box, gloo in CPU tests).

Mirrors the reference's concurrency model -- independent requests, one per dusk-uds worker (src/main.rs:55,
src/futures/main.rs:52-56) -- as independent proofs per rank.  SURVEY.md 8e.
"""
import torch


def shard_range(total, rank, world):
    """[lo, hi) of the global batch owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def small_record_len(case):
    return case["proof_len"] + 32 * (4 + case["N"])


def gather_records(dist, local, stride, total, rank, world):
    """Gather every rank's local records (uint8 tensor, len = n_local * stride) to rank 0 in global proof order.
    Shards may differ by one record, so each rank pads to the largest shard (gather needs equal sizes)."""
    base, rem = divmod(total, world)
    cap = (base + (1 if rem else 0)) * stride
    buf = torch.zeros(cap, dtype=torch.uint8, device=local.device)
    buf[:local.numel()] = local
    if world == 1:
        return buf[:total * stride]
    if dist.get_backend() != "nccl":
        buf = buf.cpu()  # gloo: host tensors
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, outs, dst=0)
    if rank != 0:
        return None
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(outs[r][:(hi - lo) * stride])
    return torch.cat(parts)
