cd "$GRAFT_REPO_ROOT"
REPS=2 bash tools/ab_variants.sh r5exp "--large-steps 0" "RT_LIB=raytracinggpu_amd/exp/maxilp.so --large-steps 0" "RT_LIB=raytracinggpu_amd/exp/maxmem.so --large-steps 0" "RT_LIB=raytracinggpu_amd/exp/nobranch.so --large-steps 0" "RT_TRAV_WAVES=3 --large-steps 0" "RT_TRAVQ_LOW=40 --large-steps 0" "RT_ADV_BLOCK=256 --large-steps 0" > gpurun_out/r5exp.txt 2>&1
cat gpurun_out/r5exp.txt
