cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tl
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_tl -o r -- python $R/dev/gpu_shard32.py --ranks ${RANK:-3} --steps 40 --repeats 1 > /tmp/prof_tl.log 2>&1
DB=$(find /tmp/prof_tl -name "*.db" | head -1)
python $R/dev/gpu_timeline.py $DB > $R/gpurun_out/r06w_timeline_shard${RANK:-3}.txt 2>&1
tail -3 /tmp/prof_tl.log | cut -c1-600
