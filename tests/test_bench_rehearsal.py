"""The N > 1 control flow of bench.py exercised by the suite (not by hand): the driver's multi-GPU command line with every
rank on GPU 0 and gloo instead of RCCL as the transport (VDL2_BENCH_REHEARSAL=1) - sharding, both exchange forms timed bare and
demodulating, the parity gate on the merged frames of all ranks, the JSON.  Its numbers mean nothing; what RCCL itself does is
covered by tests/test_gpu_parity.py::test_group_over_two_real_gpus and by the driver's own N = 2, 4, 8 runs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,pair", [(2, None), (2, "1"), (8, None)])
def test_bench_multi_rank_rehearsal(world, pair):
    """pair "1": the ranks take the exchanged blocks two per feed in the timed loops (what ranks of <= 64 channels do by themselves: 8 GPUs;
    here forced, VDL2_BENCH_PAIR=1, with K = 3 - a block left without a partner at the end of every region)"""
    import socket
    env = dict(os.environ, VDL2_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    env.pop("VDL2_BENCH_PAIR", None)
    if pair:
        env["VDL2_BENCH_PAIR"] = pair
    r = None
    for attempt in range(2):           # a rendezvous on a fresh box can fail once (port in TIME_WAIT, slow first import of torch)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "2", "--repeats", "2", "--duration", "2.0"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        if r.returncode == 0:
            break
        out_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, f"rehearsal_failure_{attempt}.txt"), "w") as f:
                f.write(r.stdout + "\n==== stderr ====\n" + r.stderr)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == world and j["scaling"] == "strong" and j["repeats"] == 2 and len(j["ms_per_step_all_repeats"]) == 2
    c = j["config"]
    assert c["channels_total"] == 256 and c["channels_per_gpu"] == 256 // world
    ex = c["exchange"]
    for k in ("broadcast_host_ms", "broadcast_hbm_ms", "allgather_host_ms", "allgather_hbm_ms"):
        assert ex[k] is not None and ex[k] > 0
    assert ex["chosen"] in ("broadcast", "allgather") and ex["uses_rccl"] is False and len(ex["h2d_whole_block_ms_by_rank"]) == world
    assert set(c["by_exchange"]) == {"broadcast", "allgather"}           # the demodulating value of BOTH forms
    for v in c["by_exchange"].values():
        assert v["value"] > 0 and v["ms_per_step"] > 0
    assert len(c["rank_ms_per_step"]) == world
    v = c["verified"]
    assert v["frames_and_integer_metadata_identical"] and v["oracle_parity_within_tolerance"] and v["channels_with_frames"] > 256 // world
    assert "REHEARSAL" in c["parallelism"] and c["blocks_per_feed"] == (2 if (pair or 256 // world <= 64) else 1)     # (world 8: ranks of 32 channels pair by themselves)
    assert j["roofline"]["bound"] == "valu" and j["roofline"]["hbm_algorithmic"]["frac"] > 0
