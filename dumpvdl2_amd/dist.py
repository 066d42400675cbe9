"""Multi-GPU layout of the hot path: channels are independent, so they shard statically across
ranks (one process per GPU, rank r owns channels [r*C/N, (r+1)*C/N)); the only exchange makes each
raw IQ block available on every rank: a broadcast from the ingest rank (north_star's literal form),
or - when the capture lies striped across the ranks (each rank ingests 1/N of every block over its
own PCIe link, or holds it in HBM) - an all-gather of the stripes (RCCL over xGMI on GPUs, gloo in
the CPU tests).  Frames never cross GPUs: every rank drains its own and rank 0 may gather them
(tiny) for a deterministic merge.

The reference's equivalent is "one pthread per channel over a shared sbuf" (src/dumpvdl2.c:117-135,
src/demod.c:300-301,342-346).
"""
from __future__ import annotations

import time
from typing import List, Optional, Sequence, Tuple


def shard_channels(nchan: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous static partition: rank r owns channels [first, first+count)."""
    base, extra = divmod(nchan, world)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first, count


def _u8(t):
    import torch
    return t if t.dtype == torch.uint8 else t.view(torch.uint8)


def broadcast_block(tensor, src: int = 0, group=None, async_op: bool = False):
    """In-place broadcast of one raw IQ block (uint8/int16 tensor, same shape on every rank)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        # raw bytes: ncclUint8 / gloo uint8 exist on every backend (int16 does not on gloo)
        return dist.broadcast(_u8(tensor), src=src, group=group, async_op=async_op)
    return None


def stripe_of(nbytes: int, world: int, rank: int) -> Tuple[int, int]:
    """Byte range [first, first+count) of a raw block that rank `rank` holds when the capture is striped over the ranks
    (equal stripes; the block length must be a multiple of the world size)."""
    if nbytes % world:
        raise ValueError(f"block of {nbytes} bytes does not split into {world} equal stripes")
    return rank * (nbytes // world), nbytes // world


def allgather_block(out, stripe, group=None, async_op: bool = False):
    """Assemble one raw IQ block on every rank from the ranks' stripes.  Same end state as broadcast_block(), but the
    (N-1)/N of the block a GPU is missing arrives over all of its xGMI links at once instead of down one broadcast tree.
    `out`: the full block (uint8 view is taken), `stripe`: this rank's part."""
    import torch.distributed as dist
    o, s = _u8(out), _u8(stripe)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        o.copy_(s)
        return None
    assert o.numel() == s.numel() * dist.get_world_size(group)
    return dist.all_gather_into_tensor(o, s, group=group, async_op=async_op)


class ShardedFeeder:
    """The data path of one rank in a sharded run, as bench.py drives it (and tests/test_dist_cpu.py, with gloo):

        step i:   start putting block i+1 on this rank        (exchange stream; overlaps the demodulation of block i)
                  feed block i to this rank's receiver         (its channeliser waits for the exchange of block i only)
                  drain the frames this rank's channels gave

    `source`  "host": the capture is in page-locked host memory and crosses PCIe inside every step - the whole block on
              rank `src` (mode "broadcast") or each rank's 1/N stripe over its own link (mode "allgather");
              "hbm": it already lies in device memory (whole on `src`, or striped).
    `mode`    "broadcast" | "allgather" (see module docstring).
    With one rank there is no exchange: the block is fed from host memory (`vdl2hip_feed_pinned`) or from the resident
    device copy.

    `pairs`   the exchange buffers are laid out two by two, so that step(pair=True) can hand the receiver TWO consecutive blocks as
              one feed: what a feed costs a rank-sized receiver beyond its channeliser - the chain walk, scans, check: 2-3 ms - is paid
              per feed, not per block (32 of 256 channels: 1.3-2.0 ms per 16 s block one at a time, 1.1-1.25 two at a time: bench.py,
              projected_scaling.two_blocks_per_feed).  The first block of a pair is only exchanged; the second step feeds both; flush()
              feeds a block left without a partner.  Results come one block later.

    `rx` needs feed_tensor(t) (block resident on this rank's device), feed_pinned_tensor(t) (host, world 1 only) and
    drain_packed(); `front_stream` is the receiver's front stream as a torch stream (None on CPU)."""

    def __init__(self, rx, block_host, world: int, rank: int, mode: str = "broadcast", source: str = "host", src: int = 0,
                 device=None, group=None, nbuf: int = 3, front_stream=None, pairs: bool = False):
        import torch
        self.torch = torch
        self.rx, self.world, self.rank, self.mode, self.source, self.src, self.group = rx, world, rank, mode, source, src, group
        self.device = device if device is not None else torch.device("cpu")
        self.cuda = self.device.type == "cuda"
        self.front = front_stream
        self.i = 0
        host = _u8(block_host)
        self.nbytes = host.numel()
        assert mode in ("broadcast", "allgather") and source in ("host", "hbm")
        if mode == "allgather" and world > 1 and self.nbytes % world:
            raise ValueError("all-gather needs a block length that is a multiple of the world size")
        pin = (lambda t: t.pin_memory()) if self.cuda else (lambda t: t)
        self.host_block = self.host_stripe = self.dev_block = self.dev_stripe = None
        if world == 1:
            if source == "host":
                self.host_block = pin(host)
            else:
                self.dev_block = host.to(self.device)
            self.bufs = []
            return
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.pairs = bool(pairs)
        self.held = None                     # pair mode: the buffer index of a pair's first block, exchanged and not yet fed
        if self.pairs:
            # nbuf PAIRS: bufs[2p], bufs[2p + 1] are the two halves of one allocation
            self.bufs2 = [torch.empty(2 * self.nbytes, dtype=torch.uint8, device=self.device) for _ in range(nbuf)]
            self.bufs = [b[h * self.nbytes:(h + 1) * self.nbytes] for b in self.bufs2 for h in (0, 1)]
            nbuf = 2 * nbuf
        else:
            self.bufs = [torch.empty(self.nbytes, dtype=torch.uint8, device=self.device) for _ in range(nbuf)]
        self.ready = [None] * nbuf           # exchange of the block in bufs[k] complete
        self.consumed = [None] * nbuf        # channeliser that read bufs[k] complete
        if mode == "allgather":
            b0, nb = stripe_of(self.nbytes, world, rank)
            if source == "host":
                self.host_stripe = pin(host[b0:b0 + nb].clone())
                self.dev_stripe = torch.empty(nb, dtype=torch.uint8, device=self.device)
            else:
                self.dev_stripe = host[b0:b0 + nb].to(self.device)
        elif rank == src:
            if source == "host":
                self.host_block = pin(host)
            else:
                self.dev_block = host.to(self.device)
                if self.pairs:
                    self.dev_block2 = torch.cat([self.dev_block, self.dev_block])    # (the source rank reads its resident copy: twice in a row)
        self._start_exchange(0)

    # -- one exchange: block -> bufs[k] on every rank (asynchronous on GPUs: queued on the side stream) --
    def _start_exchange(self, k: int):
        torch = self.torch
        dst = self.bufs[k]
        if self.cuda:
            ctx = torch.cuda.stream(self.side)
            ctx.__enter__()
            if self.consumed[k] is not None:
                self.side.wait_event(self.consumed[k])     # the channeliser that last read this buffer
        try:
            root_direct = False
            if self.mode == "allgather":
                if self.source == "host":
                    self.dev_stripe.copy_(self.host_stripe, non_blocking=True)      # 1/N of the block over this rank's PCIe link
                w = allgather_block(dst, self.dev_stripe, group=self.group, async_op=True)
            else:
                if self.rank == self.src:
                    if self.source == "host":
                        dst.copy_(self.host_block, non_blocking=True)              # the whole block over the ingest rank's link
                        w = broadcast_block(dst, src=self.src, group=self.group, async_op=True)
                    else:
                        w = broadcast_block(self.dev_block, src=self.src, group=self.group, async_op=True)
                        root_direct = True                                         # the source rank reads its resident copy
                else:
                    w = broadcast_block(dst, src=self.src, group=self.group, async_op=True)
            if w is not None:
                w.wait()                       # GPUs: the side stream waits for the collective; gloo: the host does
            self.root_direct = root_direct
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(self.side)
                self.ready[k] = ev
        finally:
            if self.cuda:
                ctx.__exit__(None, None, None)

    def step(self, pair: bool = False):
        """Feed block i (every block carries the same bytes: a benchmark loop), start the exchange of block i+1, drain.
        pair=True (a feeder made with pairs=True): the first block of a pair is exchanged and held, the second step feeds both."""
        if self.world == 1:
            if self.source == "host":
                self.rx.feed_pinned_tensor(self.host_block)
            else:
                self.rx.feed_tensor(self.dev_block)
            self.i += 1
            return self.rx.drain_packed()
        nb = len(self.bufs)
        k = self.i % nb
        direct = getattr(self, "root_direct", False)
        self._start_exchange((self.i + 1) % nb)
        pair = pair and self.pairs
        if pair and self.held is None and k % 2 == 0:
            self.held = k                    # the first half of a pair: wait for its partner
            self.i += 1
            return self.rx.drain_packed()
        if self.held is not None and not (pair and k == self.held + 1):
            self._feed_held()                # (pair mode was left between the two halves)
        both = self.held is not None
        ks = [self.held, k] if both else [k]
        if self.cuda:
            for j in ks:
                if self.ready[j] is not None:
                    self.front.wait_event(self.ready[j])
        if both:
            self.rx.feed_tensor(self.dev_block2 if (direct and self.dev_block is not None) else self.bufs2[k // 2])
        else:
            self.rx.feed_tensor(self.dev_block if (direct and self.dev_block is not None) else self.bufs[k])
        if self.cuda:
            ev = self.front.record_event()
            for j in ks:
                self.consumed[j] = ev
        self.held = None
        self.i += 1
        return self.rx.drain_packed()

    def _feed_held(self):
        k, direct = self.held, getattr(self, "root_direct", False)
        if self.cuda and self.ready[k] is not None:
            self.front.wait_event(self.ready[k])
        self.rx.feed_tensor(self.dev_block if (direct and self.dev_block is not None) else self.bufs[k])
        if self.cuda:
            self.consumed[k] = self.front.record_event()
        self.held = None

    def flush(self):
        """pair mode: feed a block that has been exchanged and is still waiting for its partner (before the last drain of a region)"""
        if self.world > 1 and self.held is not None:
            self._feed_held()

    def finish(self):
        """wait for the exchange left in flight by the last step (collectives must complete on every rank)"""
        if self.cuda and self.world > 1:
            self.side.synchronize()

    def current_block(self):
        """the block this rank would feed next, for checks (bytes must equal the capture on every rank)"""
        if self.world == 1:
            return self.host_block if self.host_block is not None else self.dev_block
        self.finish()
        if getattr(self, "root_direct", False) and self.dev_block is not None:
            return self.dev_block
        return self.bufs[self.i % len(self.bufs)]


def time_exchange(block_host, world: int, rank: int, mode: str, source: str, device, group=None, iters: int = 4, src: int = 0):
    """Seconds per block of the bare exchange (no demodulation), max over ranks - what `--exchange auto` compares."""
    import torch
    import torch.distributed as dist

    class _Null:
        def feed_tensor(self, t): pass
        def feed_pinned_tensor(self, t): pass
        def drain_packed(self): return None

    front = torch.cuda.current_stream(device) if device.type == "cuda" else None
    f = ShardedFeeder(_Null(), block_host, world, rank, mode=mode, source=source, src=src, device=device, group=group, front_stream=front)
    f.step(); f.finish()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    dist.barrier(group=group)
    t0 = time.perf_counter()
    for _ in range(iters):
        f.step()
    f.finish()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / iters
    ok = int(torch.equal(_u8(f.current_block()).cpu(), _u8(block_host)))
    t = torch.tensor([dt, float(1 - ok)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t[0].item()), t[1].item() == 0.0


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
