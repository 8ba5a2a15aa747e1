#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for ct in 0.7 1.0 1.6 2.5; do
  echo "== RT_LBVH_CT=$ct"
  RT_LBVH_CT=$ct BIG_N=161,513,1025 timeout -k 10 400 python3 tools/big_mesh_bench.py 2>&1 | grep -E "LBVH|rror" | sed -E 's/device build.*call [0-9.]+ s\), //' || exit 1
done
