cd "$GRAFT_REPO_ROOT"
REPS=2 bash tools/ab_variants.sh qwbpc "RT_TRAVQ_QW=1 --large-steps 0" "RT_TRAVQ_QW=1 RT_LIB=raytracinggpu_amd/exp/qw_s460.so --large-steps 0" "RT_TRAVQ_QW=1 RT_TRAVQ_BPC5=1 RT_LIB=raytracinggpu_amd/exp/qw_s460.so --large-steps 0" "RT_TRAVQ_QW=1 RT_TRAVQ_BPC5=1 RT_TRAVQ_LOW=32 RT_LIB=raytracinggpu_amd/exp/qw_s460.so --large-steps 0" "RT_TRAVQ_QW=1 RT_PARTS=3 --large-steps 0" "RT_TRAVQ_QW=1 RT_ADV_BLOCK=128 --large-steps 0" > gpurun_out/qwbpc.txt 2>&1
cat gpurun_out/qwbpc.txt
