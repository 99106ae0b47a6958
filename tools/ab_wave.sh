#!/bin/bash
# A/B of builds of the 4096- / 8192-sample kernels on one box: tools/ab_wave.sh [-n samples] lib1.so lib2.so ...
# ("-" = the product library); prints traces/s of bench configs 1 and 2 per build.
R=${GRAFT_REPO_ROOT:-$PWD}
NS=4096
if [ "$1" = "-n" ]; then NS=$2; shift 2; fi
TR=$((8589934592 / NS))          # 32 GiB of traces
for lib in "$@"; do
  for c in 1 2; do
    if [ "$lib" = "-" ]; then unset OFX_LIB; else export OFX_LIB=$R/$lib; fi
    v=$(timeout -k 10 300 python3 bench.py --samples $NS --traces $TR --engine fused --config $c --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']/1e6,1))")
    echo "$lib config $c: $v M traces/s"
  done
done
