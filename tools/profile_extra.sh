cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lds -o run -- python3 tools/dev_bench_lds.py 25000 > gpurun_out/prof_lds.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trig -o run -- python3 tools/dev_bench_trigger.py 32768 75000000 > gpurun_out/prof_trig.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg -o run -- python3 tools/dev_bench_configs.py 131072 > gpurun_out/prof_cfg.log 2>&1
grep -h "lds \|rocfft \|GPU trigger\|config" gpurun_out/prof_lds.log gpurun_out/prof_trig.log gpurun_out/prof_cfg.log
