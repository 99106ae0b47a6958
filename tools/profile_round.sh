#!/bin/bash
# Collect the profile artefacts of a round on the GPU box (run from the repo root through gpurun):
#   per bench config (1, 2, 3):
#     1. rocprofv3 --kernel-trace --stats of the bench command              -> kernel average duration
#     2. separate --pmc passes FETCH_SIZE / WRITE_SIZE of the same command  -> HBM traffic
#   3. the same FETCH_SIZE pass on the load-only build (known 131072 B/trace) -> calibration of (2)
#   4. SQ instruction / wait counters of config 1
#   5. clock / power samples while the kernel runs
# Output: gpurun_out/prof/ ; tools/make_profiles.py turns it into profiles/rNN_*.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd $R
for c in 1 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c$c -o run -- python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline > $O/stats_c$c.log 2>&1
  echo "stats config $c done"
  for p in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $p --kernel-trace --output-format csv -d $O/pmc_${p}_c$c -o run -- python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_${p}_c$c.log 2>&1
  done
  echo "pmc config $c done"
done
if [ -f $R/gpurun_loadonly.so ]; then
  export OFX_LIB=$R/gpurun_loadonly.so
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_FETCH_SIZE -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/cal_FETCH_SIZE.log 2>&1
  unset OFX_LIB
  echo "calibration done"
fi
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq_$i -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/sq_$i.log 2>&1
done
echo "sq done"
python3 tools/clock_probe.py 262144 5 > $O/clock_probe.txt 2>&1
echo done
