hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DVDL2_EXPERIMENTS -o /tmp/vdl2hip_exp.so dumpvdl2_amd/csrc/vdl2hip.hip 2>/dev/null
export VDL2HIP_LIB=/tmp/vdl2hip_exp.so LAGS=5 VDL2HIP_GAPS=1
python - <<'P' 2>&1 | tail -40
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from dumpvdl2_amd import synth, vdl2hip, workloads
cfg = workloads.config4(16.0)
path = "/tmp/vdl2_config4_16.npy"
iq = np.load(path) if os.path.exists(path) else synth.synthesize(cfg)[0]
nbytes = iq.size * 2
dev = torch.from_numpy(iq).to("cuda:0")
rx = vdl2hip.Receiver(cfg.centerfreq, list(cfg.freqs), cfg.oversample, vdl2hip.FMT_S16LE, cfg.rx_max_ppm, max_block_bytes=nbytes, chan_first=96, chan_count=32)
rx.set_profiling(2)
rx.set_drain_lag(int(os.environ.get("LAG","5")))
for _ in range(int(os.environ.get("NSTEP","30"))):
    rx.feed_device(dev.data_ptr(), nbytes); rx.drain_packed()
rx.set_drain_lag(0); rx.drain_packed()
P
