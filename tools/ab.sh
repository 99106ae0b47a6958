#!/bin/bash
# A/B of k_fused builds on one box: product lib, then every gpurun_<name>.so given
cd ${GRAFT_REPO_ROOT:-/root/repo}
B=${B:-262144}
echo "product:"; timeout -k 10 120 python tools/dev_bench.py $B fused 2>&1 | grep "^fused"
for v in "$@"; do echo "$v:"; OFX_LIB=$PWD/gpurun_$v.so timeout -k 10 120 python tools/dev_bench.py $B fused 2>&1 | grep "^fused"; done
echo "product again:"; timeout -k 10 120 python tools/dev_bench.py $B fused 2>&1 | grep "^fused"
