#!/bin/bash
# Reproduces, without a GPU, the register-allocator miscompile behind round 2's wrong
# k_fused25<2, true> / <6, true> from this repository's own history:
# the kernel source of commit 5a80a2d^ is compiled to gfx950 assembly with the product flags and
# tools/isa_hazards.py reports the two flow blocks whose live-range-split copies
#     v_mov_b64_e32 v[112:113], v[84:85]      (<2, true>)     v_mov_b64_e32 v[98:99], v[92:93]   (<6, true>)
# sit, behind a rematerialised `s_movk_i32 s0, 0x2800`, ABOVE `s_or_saveexec_b64 s[8:9], s[8:9]` --
# i.e. they run with the EXEC of the `tl != 0` side of `tbh = (tl == 0) ? tb0hi : tb` only, and
# thread 0 keeps stale registers.  (hipcc of ROCm 7.2.0; SIInstrInfo::isBasicBlockPrologue stops at
# the scalar instruction, so SplitKit's insertion point for vector copies is the block start.)
# The same source with -mllvm -disable-machine-sink happens to allocate differently and is clean: on
# the GPU the first build fails tools/probe_bins.py (thread 0's bins only) and the second passes --
# as do five more flag / source variants, flagged instantiation by flagged instantiation
# (profiles/r03_miscompile_bisect.md).  What avoids it in the product: no scalar load behind a
# per-lane select in the register-heavy kernels (Tabs25::tb0hi), and `make check` on every build.
set -e
cd "$(dirname "$0")/../.."
T=$(mktemp -d)
git archive 5a80a2d^ detprocess_amd/csrc include | tar -x -C "$T"
F="-O3 -fno-slp-vectorize -std=c++17 --offload-arch=gfx950 -I$T/include -S --cuda-device-only"
/opt/rocm/bin/hipcc $F "$T/detprocess_amd/csrc/ofx_fused25.hip" -o "$T/bad.s" 2>/dev/null &
/opt/rocm/bin/hipcc $F -mllvm -disable-machine-sink "$T/detprocess_amd/csrc/ofx_fused25.hip" -o "$T/nosink.s" 2>/dev/null &
wait
python3 tools/isa_hazards.py "$T/bad.s" "$T/nosink.s" | c++filt | cut -c1-160
rm -rf "$T"
