#!/bin/bash
# Collect the profile artefacts of a round on the GPU box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the bench command          -> kernel average duration
#   2. separate --pmc passes FETCH_SIZE / WRITE_SIZE of the same command -> HBM traffic
#   3. the same FETCH_SIZE pass on the load-only build (known 131072 B/trace) -> calibration of (2)
#   4. SQ instruction / wait counters
# Output: gpurun_out/prof/ ; tools/make_profiles.py turns it into profiles/rNN_*.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --steps 5 --warmup 1 > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1
done
if [ -f $R/gpurun_LOADONLY.so ]; then
  export OFX_LIB=$R/gpurun_LOADONLY.so
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_FETCH_SIZE -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/cal_FETCH_SIZE.log 2>&1
  unset OFX_LIB
fi
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq_$i -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/sq_$i.log 2>&1
done
python3 tools/clock_probe.py 262144 5 > $O/clock_probe.txt 2>&1
echo done
