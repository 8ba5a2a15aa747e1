cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5b
for spec in "X=1" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=1" "HSA_ENABLE_SDMA=0" "ROCR_VISIBLE_DEVICES=0" "HIP_VISIBLE_DEVICES=0" "HSA_XNACK=0" "AMD_SERIALIZE_KERNEL=0 HIP_INITIAL_DM_SIZE=0" "HSA_TOOLS_LIB= ROCP_TOOL_LIB=" ; do
  for r in 1 2 3; do echo -n "$spec: "; env $spec ./tools/ubench/hip_init; done
done > gpurun_out/r5b/hip_init.txt 2>&1
cat gpurun_out/r5b/hip_init.txt
env | grep -i -E "^(HSA|HIP|ROC|AMD|GPU)" 
nproc; rocminfo 2>/dev/null | grep -c "Name:.*gfx"
RUNS=2 python3 tools/launcher_timing.py 2>&1 | grep -E "run|write_png|copy to|total"
