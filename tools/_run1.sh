cd "$GRAFT_REPO_ROOT"
REPS=2 bash tools/ab_variants.sh qwlow "RT_TRAVQ_QW=0 --large-steps 0" "RT_TRAVQ_QW=1 --large-steps 0" "RT_TRAVQ_QW=1 RT_TRAVQ_LOW=32 --large-steps 0" "RT_TRAVQ_QW=1 RT_TRAVQ_LOW=64 --large-steps 0" "RT_TRAVQ_QW=1 RT_TRAVQ_LOW=96 --large-steps 0" "RT_TRAVQ_QW=1 RT_TRAVQ_MINFREE=8 --large-steps 0" "RT_TRAVQ_QW=1 RT_TRAVQ_MINFREE=32 --large-steps 0" "RT_TRAVQ_QW=1 RT_TRAVQ_OVERSUB=3 --large-steps 0" > gpurun_out/qwlow.txt 2>&1
cat gpurun_out/qwlow.txt
bash tools/pmc_ab.sh pmcqw "RT_TRAVQ_QW=0" "RT_TRAVQ_QW=1" > gpurun_out/pmcqw.txt 2>&1
tail -5 gpurun_out/pmcqw.txt
