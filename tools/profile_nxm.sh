cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nxm -o run -- python3 tools/dev_bench_nxm.py > gpurun_out/prof_nxm.log 2>&1
grep -h "events/s" gpurun_out/prof_nxm.log
