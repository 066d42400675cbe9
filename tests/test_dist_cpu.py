"""world_size-2 CPU test (gloo) of the multi-GPU layout: channel sharding, the IQ-block broadcast
and the frame merge.  The per-rank decoder here is the oracle (the HIP library needs a GPU); what
is under test is dumpvdl2_amd/dist.py, which bench.py drives identically with RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_channels_partitions():
    from dumpvdl2_amd.dist import shard_channels
    for n in (1, 7, 8, 64, 255, 256):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_channels(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dumpvdl2_amd import dist as vd
    from oracle import pyoracle as po
    import cases
    cfg, iq, _, gold = cases.load("config2_1s")
    block = torch.from_numpy(iq.copy()) if rank == 0 else torch.zeros(iq.size, dtype=torch.int16)
    vd.broadcast_block(block, src=0)
    # the other exchange: every rank holds one stripe of the capture, an all-gather rebuilds the block everywhere
    first_b, count_b = vd.stripe_of(block.numel() * 2, world, rank)
    stripe = block.view(torch.uint8)[first_b:first_b + count_b].clone()
    rebuilt = torch.zeros_like(block)
    w = vd.allgather_block(rebuilt, stripe, async_op=True)
    w.wait()
    assert torch.equal(rebuilt, torch.from_numpy(iq))
    block = rebuilt
    # the exchange object bench.py uses, in both modes: the "next block" lands complete in the destination on every rank
    for mode in ("allgather", "broadcast"):
        ex = vd.BlockExchange(block, mode=mode, src=0)
        assert ex.mode == mode
        nxt = torch.zeros_like(block)
        if mode == "broadcast" and rank == 0:
            nxt.copy_(block)                                             # broadcast: the source rank's buffer holds the block
        ex.start(nxt).wait()
        assert torch.equal(nxt, torch.from_numpy(iq)), mode
    odd = torch.arange(7, dtype=torch.uint8)                            # 7 bytes do not split into 2 stripes -> broadcast
    assert vd.BlockExchange(odd, mode="allgather").mode == "broadcast"
    first, count = vd.shard_channels(len(cfg.freqs), world, rank)
    o = po.Oracle(cfg.centerfreq, list(cfg.freqs)[first:first + count], oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
    o.process(block.numpy().view(np.uint8), block_bytes=1 << 24)
    fr = o.frames()
    for f in fr:
        f["chan"] += first
    merged = vd.gather_frames(fr, dst=0)
    if rank == 0:
        q.put([(f["chan"], f["burst_ord"], f["idx"], f["octets"]) for f in merged])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_broadcast_merge(oracle_mod):
    import cases, hashlib
    cfg, iq, _, gold = cases.load("config2_1s")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue, time
    merged = None
    t0 = time.time()
    while merged is None and time.time() - t0 < 240:
        try:
            merged = q.get(timeout=2)
        except queue.Empty:
            assert all(p.exitcode in (None, 0) for p in procs), "a rank died"
    assert merged is not None
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got = sorted((c, b, i, hashlib.sha1(o).hexdigest()) for c, b, i, o in merged)
    want = sorted((f["chan"], f["burst_ord"], f["idx"], f["sha1"]) for f in gold["frames"])
    assert got == want
    ends = [m for m in merged]
    assert len(ends) == len(gold["frames"])
