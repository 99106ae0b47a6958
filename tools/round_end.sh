#!/bin/bash
# Everything the committed round summaries are made from, in one GPU call (run from the repo root through
# gpurun; ~6 minutes):  tools/round_end.sh  ->  gpurun_out/prof, prof25, parity_report.json, fuzz logs.
# Afterwards here: python tools/make_profiles.py N; python tools/make_profiles25.py N; python tools/make_profiles25.py N 4096 4194304
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
bash tools/profile_round.sh > gpurun_out/round_prof.log 2>&1
echo "profile_round done"
bash tools/profile_fused25.sh > gpurun_out/round_prof25.log 2>&1
echo "profile 25000 done"
# (the 4096-sample passes -- tools/profile_fused25.sh 4096 4194304 -- are run in a call of their own: one
#  --pmc pass of that command stopped answering once at the end of a long session of profiler runs)
python3 tools/parity_report.py 8192 gpurun_out/parity_report.json > gpurun_out/parity_report.log 2>&1
echo "parity report done"
for m in fused fused25 wave general trigger nxm; do
  if [ $m = general ]; then a=""; else a=$m; fi
  timeout -k 10 400 python3 tools/fuzz_engines.py 60 9300 $a > gpurun_out/fuzz_$m.log 2>&1 || true
  tail -1 gpurun_out/fuzz_$m.log
done
