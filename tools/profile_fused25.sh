#!/bin/bash
# Profile artefacts of a FUSED kernel other than the headline one on the GPU box, run from the repo
# root through gpurun:  kernel stats of the bench command, separate FETCH_SIZE / WRITE_SIZE passes,
# SQ counters.   tools/profile_fused25.sh [samples = 25000] [traces = 1048576]
# (25000: k_fused25, 4096: k_wave).  Output: gpurun_out/prof<samples>/ (25000: prof25);
# tools/make_profiles25.py turns it into profiles/rNN_*.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
S=${1:-25000}
T=${2:-1048576}
O=$R/gpurun_out/prof$S
[ "$S" = 25000 ] && O=$R/gpurun_out/prof25
rm -rf $O; mkdir -p $O
cd $R
BENCH="python3 bench.py --samples $S --traces $T --config 1"
timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $BENCH --steps 5 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
echo "stats done"
for p in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 180 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $O/pmc_$p -o run -- $BENCH --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$p.log 2>&1
done
echo "pmc done"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq_$i -o run -- $BENCH --steps 1 --warmup 1 --no-cpu-baseline > $O/sq_$i.log 2>&1
done
echo "sq done"
