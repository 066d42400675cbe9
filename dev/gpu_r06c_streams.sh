#!/bin/bash
# r06c: a rank-sized receiver (32 of config4's channels) with the walks of one / two feeds ahead of a check, three scan streams or one per
# slot in flight (dev/_variants/libvdl2hip_pre6.so: -DVDL2_SIDE_PRE=6), 4 or 8 hardware queues per priority class.  One box, one call.
cd "$(dirname "$0")/.."
O=gpurun_out/r06c_streams.txt; : > $O
run() { # label, env...
	local label=$1; shift
	env "$@" timeout 500 python dev/gpu_shard32.py --ranks 0,3,7 --steps ${STEPS:-60} --repeats 2 --sweep "VDL2HIP_WALK_AHEAD=1,2" 2>&1 >/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    print('$label', j['env'], 'rank', j['rank'], j.get('ms_per_step', j.get('error')))
" | tee -a $O
}
V=$PWD/dev/_variants/libvdl2hip_pre6.so
run "pre3 hwq4" A=1
run "pre6 hwq4" VDL2HIP_LIB=$V
run "pre6 hwq8" VDL2HIP_LIB=$V GPU_MAX_HW_QUEUES=8
run "pre3 hwq8" GPU_MAX_HW_QUEUES=8
