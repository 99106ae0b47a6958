#!/bin/bash
# A/B of k_fused25 builds on one box: product lib, then every gpurun_f25_*.so given
cd ${GRAFT_REPO_ROOT:-/root/repo}
B=${B:-262144}
echo "product:"; timeout -k 10 120 python tools/dev_bench_fused25.py $B fused
for v in "$@"; do echo "$v:"; OFX_LIB=$PWD/gpurun_f25_$v.so timeout -k 10 120 python tools/dev_bench_fused25.py $B fused; done
echo "product again:"; timeout -k 10 120 python tools/dev_bench_fused25.py $B fused
