#!/bin/bash
cd "$GRAFT_REPO_ROOT"
REPS=2 STEPS=30 bash tools/ab_variants.sh parts "--large-steps 0" "RT_PARTS=3 --large-steps 0" "RT_PARTS=4 --large-steps 0" "RT_PARTS=3 GPU_MAX_HW_QUEUES=8 --large-steps 0" "RT_PARTS=4 GPU_MAX_HW_QUEUES=8 --large-steps 0" "RT_PARTS=3 RT_PART_PRIO=1 --large-steps 0" "RT_PARTS=4 RT_PART_PRIO=1 GPU_MAX_HW_QUEUES=8 --large-steps 0"
