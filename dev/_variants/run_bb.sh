cd /root/repo
for e in "A=1" "VDL2HIP_WALK_AHEAD=1" "VDL2HIP_WALK_AHEAD=2" "VDL2HIP_WALK_AHEAD=1 VDL2HIP_REF_PRESCAN=1" "VDL2HIP_REF_PRESCAN=1"; do
  echo "== $e"; env $e python dev/gpu_block_batch.py config4 8 16,32 1,2,4 2>&1 | grep -v amdgpu.ids
done
