#!/bin/bash
# Compact per-kernel resource table (VGPRs, scratch bytes, LDS) of one HIP source:
#   tools/kernel_resources.sh detprocess_amd/csrc/ofx_fused.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 -fPIC --offload-arch=gfx950 \
    -I"$(dirname "$0")/../include" -I"$(dirname "$src")" "$@" -c "$src" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 |
awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /LDS Size/ {print name, "vgpr="v, "scratch="s, "lds="$(NF-1)}' |
sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | c++filt | sed 's/(anonymous namespace):://; s/(.*)//'
