"""Multi-GPU layout of the hot path: channels are independent, so they shard statically across
ranks (one process per GPU); the only exchange makes each raw IQ block available on every rank: a
broadcast from the ingest rank, or - when the capture already lies striped across the GPUs - an
all-gather of the stripes (RCCL over xGMI on GPUs, gloo in the CPU tests).  Frames never cross
GPUs: every rank drains its own and rank 0 may gather them (tiny) for a deterministic merge.

The reference's equivalent is "one pthread per channel over a shared sbuf" (src/dumpvdl2.c:117-135,
src/demod.c:300-301,342-346).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_channels(nchan: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous static partition: rank r owns channels [first, first+count)."""
    base, extra = divmod(nchan, world)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first, count


def broadcast_block(tensor, src: int = 0, group=None):
    """In-place broadcast of one raw IQ block (uint8/int16 tensor, same shape on every rank)."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        # raw bytes: ncclUint8 / gloo uint8 exist on every backend (int16 does not on gloo)
        dist.broadcast(tensor.view(torch.uint8) if tensor.dtype != torch.uint8 else tensor, src=src, group=group)
    return tensor


def stripe_of(nbytes: int, world: int, rank: int) -> Tuple[int, int]:
    """Byte range [first, first+count) of a raw block that rank `rank` holds when the capture is striped over the GPUs
    (equal stripes; the block length must be a multiple of the world size)."""
    if nbytes % world:
        raise ValueError(f"block of {nbytes} bytes does not split into {world} equal stripes")
    return rank * (nbytes // world), nbytes // world


def allgather_block(out, stripe, group=None, async_op: bool = False):
    """Assemble one raw IQ block on every rank from the ranks' stripes (the capture lives striped across the GPUs' HBM).
    Same end state as broadcast_block(), but the (N-1)/N of the block a GPU is missing arrives over all of its xGMI links at
    once instead of down one broadcast tree.  `out`: the full block (uint8 view is taken), `stripe`: this rank's part."""
    import torch
    import torch.distributed as dist
    o = out.view(torch.uint8) if out.dtype != torch.uint8 else out
    s = stripe.view(torch.uint8) if stripe.dtype != torch.uint8 else stripe
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        o.copy_(s)
        return None
    assert o.numel() == s.numel() * dist.get_world_size(group)
    return dist.all_gather_into_tensor(o, s, group=group, async_op=async_op)


class BlockExchange:
    """The path's only exchange, as bench.py drives it: puts the next raw IQ block on every rank while the current one is being
    demodulated.  `mode` "allgather": every rank keeps its stripe of the capture resident and the block is rebuilt with one
    all-gather; "broadcast": rank `src` sends it whole.  `block` is the block as it already lies on this rank (any rank's copy
    is complete after the initial broadcast); on a backend/shape that cannot all-gather, or if the dry run does not rebuild
    the block bit for bit on every rank, the exchange falls back to broadcast on all ranks together."""

    def __init__(self, block, mode: str = "allgather", src: int = 0, group=None, scratch=None):
        import torch
        import torch.distributed as dist
        self.group, self.src, self.mode, self.stripe = group, src, "broadcast", None
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        nbytes = block.numel() * block.element_size()
        if mode == "allgather" and self.world > 1 and nbytes % self.world == 0:
            b0, nb = stripe_of(nbytes, self.world, rank)
            self.stripe = block.view(torch.uint8)[b0:b0 + nb].clone()
            ok = 1
            try:
                dst = scratch if scratch is not None else torch.empty_like(block)
                w = allgather_block(dst, self.stripe, group=group, async_op=True)
                w.wait()
                if block.is_cuda:
                    torch.cuda.synchronize()
                ok = int(torch.equal(dst.view(torch.uint8), block.view(torch.uint8)))
            except Exception:                  # noqa: BLE001 - any backend complaint means "use the other exchange"
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=block.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()):
                self.mode = "allgather"
            else:
                self.stripe = None

    def start(self, dst):
        """Begin filling `dst` (full-size block buffer) on every rank; returns the async work handle (None for one rank)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return None
        if self.mode == "allgather":
            return allgather_block(dst, self.stripe, group=self.group, async_op=True)
        return dist.broadcast(dst.view(torch.uint8) if dst.dtype != torch.uint8 else dst, src=self.src, group=self.group, async_op=True)


def merge_frames(per_rank: Sequence[Sequence[dict]]) -> List[dict]:
    """Deterministic global order of frames from all shards: (end_sample, chan, idx)."""
    out = [f for fr in per_rank for f in fr]
    out.sort(key=lambda f: (f["end_sample"], f["chan"], f["idx"]))
    return out


def gather_frames(frames: Sequence[dict], dst: int = 0, group=None):
    """Collect every rank's frame list on `dst` (python objects; a few KB per second of signal)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return merge_frames([frames])
    world = dist.get_world_size(group)
    bucket = [None] * world if dist.get_rank(group) == dst else None
    dist.gather_object(list(frames), bucket, dst=dst, group=group)
    return merge_frames(bucket) if bucket is not None else None
