#!/bin/bash
# registers / scratch of the quick (FEAT = 0) build of k_fused25, and its per-phase instruction mix
cd /root/repo
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 --offload-arch=gfx950 -Iinclude -S --cuda-device-only -DOFX_QUICK "$@" detprocess_amd/csrc/ofx_fused25.hip -o /tmp/f25.s 2>&1 | grep -E "error" 
grep -E "^\s*\.(vgpr_count|private_segment_fixed_size|vgpr_spill_count):" /tmp/f25.s | paste - - - | head -3
python tools/isa_phases.py /tmp/f25.s 0 _ZN12_GLOBAL__N_19k_fused25ILi
