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


class OracleRx:
    """stands in for vdl2hip.Receiver on a CPU rank: same three calls the feeder makes, decoding with the oracle"""

    def __init__(self, po, cfg, first, count):
        self.o = po.Oracle(cfg.centerfreq, list(cfg.freqs)[first:first + count], oversample=cfg.oversample, max_ppm=cfg.rx_max_ppm)
        self.first, self.blocks = first, 0

    def feed_tensor(self, t):
        self.o.process(t.numpy().view(np.uint8), block_bytes=1 << 24)
        self.blocks += 1

    feed_pinned_tensor = feed_tensor

    def drain_packed(self):
        fr = self.o.frames()
        for f in fr:
            f["chan"] += self.first
        return fr


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dumpvdl2_amd import dist as vd
    from oracle import pyoracle as po
    import cases
    cfg, iq, _, gold = cases.load("config2_1s")
    host = torch.from_numpy(iq.copy())
    cpu = torch.device("cpu")
    first, count = vd.shard_channels(len(cfg.freqs), world, rank)
    # bench.py's step function, in every exchange form: after each step the block this rank would feed next is the capture,
    # bit for bit, and the rank has decoded exactly its own channels of one block
    merged_by_mode = {}
    for mode in ("broadcast", "allgather"):
        for source in ("host", "hbm"):
            # ranks other than the source start from garbage in broadcast mode: only what the exchange delivers counts
            mine = host if (mode == "allgather" or rank == 0) else torch.randint(-99, 99, host.shape, dtype=torch.int16)
            rx = OracleRx(po, cfg, first, count)
            f = vd.ShardedFeeder(rx, mine, world, rank, mode=mode, source=source, device=cpu)
            fr = f.step()
            assert rx.blocks == 1
            assert torch.equal(vd._u8(f.current_block()), vd._u8(host)), (mode, source)
            assert all(first <= x["chan"] < first + count for x in fr)
            merged_by_mode[(mode, source)] = vd.gather_frames(fr, dst=0)
            for _ in range(4):                 # the ring of buffers wraps: blocks keep arriving complete
                f._start_exchange((f.i + 1) % len(f.bufs)); f.i += 1
                assert torch.equal(vd._u8(f.current_block()), vd._u8(host)), (mode, source)
            f.finish()
    # blocks two at a time (ShardedFeeder(pairs=True), step(pair=True)): the first block of a pair is exchanged and held, the second step feeds
    # both as ONE tensor of twice the length, flush() feeds a block left without a partner, and a step without `pair` in between feeds what
    # is held on its own - whatever the exchange form, every byte fed is the capture's
    class ByteRx:
        def __init__(self): self.fed = []
        def feed_tensor(self, t): self.fed.append(vd._u8(t).clone())
        feed_pinned_tensor = feed_tensor
        def drain_packed(self): return (0, None, None)
    small = host[: 2 * 4096 * world].clone()
    one = vd._u8(small)
    for mode in ("broadcast", "allgather"):
        for source in ("host", "hbm"):
            mine = small if (mode == "allgather" or rank == 0) else torch.randint(-99, 99, small.shape, dtype=torch.int16)
            rx = ByteRx()
            f = vd.ShardedFeeder(rx, mine, world, rank, mode=mode, source=source, device=cpu, pairs=True)
            f.step(pair=True); assert len(rx.fed) == 0 and f.held == 0
            f.step(pair=True); assert len(rx.fed) == 1 and f.held is None
            f.step(pair=True); f.step(pair=True)                      # the second pair
            f.step(pair=True); assert f.held == 4                      # a first half ...
            f.step()                                                   # ... then a step that does not pair: the held block goes on its own, then this one
            f.step(pair=True); f.flush(); f.flush()                    # a block left without a partner (slot 0 again: the ring of 6 has wrapped)
            f.step(); f.finish()
            sizes = [t.numel() // one.numel() for t in rx.fed]
            assert sizes == [2, 2, 1, 1, 1, 1], (mode, source, sizes)
            for t in rx.fed:
                assert torch.equal(t, one if t.numel() == one.numel() else torch.cat([one, one])), (mode, source)
    secs, ok = vd.time_exchange(host, world, rank, "allgather", "host", cpu, iters=2)
    assert ok and secs > 0
    secs, ok = vd.time_exchange(host, world, rank, "broadcast", "hbm", cpu, iters=2)
    assert ok and secs > 0
    with pytest.raises(ValueError):
        vd.ShardedFeeder(OracleRx(po, cfg, first, count), torch.arange(7, dtype=torch.uint8), world, rank, mode="allgather", device=cpu)
    if rank == 0:
        ref = merged_by_mode[("broadcast", "host")]
        key = lambda m: [(f["chan"], f["burst_ord"], f["idx"], f["octets"]) for f in m]
        assert all(key(m) == key(ref) for m in merged_by_mode.values())
        q.put(key(ref))
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_feeder_needs_no_process_group(oracle_mod):
    """world 1: the same step function feeds from host memory or from the resident copy, no collective involved"""
    import cases
    from dumpvdl2_amd import dist as vd
    cfg, iq, _, gold = cases.load("config2_1s")
    for source in ("host", "hbm"):
        rx = OracleRx(oracle_mod, cfg, 0, len(cfg.freqs))
        f = vd.ShardedFeeder(rx, torch.from_numpy(iq.copy()), 1, 0, mode="broadcast", source=source)
        fr = f.step()
        cases.check_against_golden(fr, None, gold, 1e-3, 1e-4, "one-rank feeder")


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
