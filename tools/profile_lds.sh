cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lds -o run -- python3 tools/dev_bench_lds.py 25000 > gpurun_out/prof_lds.log 2>&1
grep -h "lds \|rocfft " gpurun_out/prof_lds.log
