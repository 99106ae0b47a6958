"""The static ISA check (tools/isa_hazards.py) that guards the build against the register-allocator
miscompile behind round 2's wrong k_fused25<2, true> / <6, true>: split
copies placed above the instruction that restores EXEC at the top of a flow block."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "isa_hazards.py")
GOLD = os.path.join(ROOT, "tests", "golden")


def _run(*paths):
    return subprocess.run([sys.executable, TOOL, *paths], capture_output=True, text=True)


def test_checker_flags_the_miscompiled_flow_block():
    """The flow block of `tbh = (tl == 0) ? tb0hi : tb` exactly as hipcc 7.2 emitted it for the
    failing instantiation (two v_mov_b64 and a rematerialised s_movk_i32 above s_or_saveexec_b64)
    is reported; the same block with the copies behind the EXEC restore is not."""
    bad = _run(os.path.join(GOLD, "isa_execprologue_bad.s"))
    assert bad.returncode == 1 and "[execprologue]" in bad.stdout, bad.stdout
    assert "v_mov_b64_e32 v[112:113], v[84:85]" in bad.stdout
    good = _run(os.path.join(GOLD, "isa_execprologue_good.s"))
    assert good.returncode == 0, good.stdout


def test_checker_models_wait_states_and_counters(tmp_path):
    """The other passes on hand-written snippets: a DPP read one wait state after the VALU write of
    its source, a transcendental result used at once, a load result read with its vmcnt pending."""
    cases = {
        "dpp": ("k:\n\tv_add_f32_e32 v1, v2, v3\n\tv_mov_b32_dpp v4, v1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_endpgm\n", True),
        "dpp_ok": ("k:\n\tv_add_f32_e32 v1, v2, v3\n\ts_nop 1\n\tv_mov_b32_dpp v4, v1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_endpgm\n", False),
        "trans": ("k:\n\tv_rcp_f32_e32 v1, v2\n\tv_mul_f32_e32 v3, v1, v1\n\ts_endpgm\n", True),
        "trans_ok": ("k:\n\tv_rcp_f32_e32 v1, v2\n\ts_nop 0\n\tv_mul_f32_e32 v3, v1, v1\n\ts_endpgm\n", False),
        "vmcnt": ("k:\n\tbuffer_load_dword v1, v0, s[0:3], 0 offen\n\tbuffer_load_dword v2, v0, s[0:3], 0 offen offset:4\n"
                  "\ts_waitcnt vmcnt(1)\n\tv_add_f32_e32 v3, v2, v2\n\ts_endpgm\n", True),
        "vmcnt_ok": ("k:\n\tbuffer_load_dword v1, v0, s[0:3], 0 offen\n\tbuffer_load_dword v2, v0, s[0:3], 0 offen offset:4\n"
                     "\ts_waitcnt vmcnt(1)\n\tv_add_f32_e32 v3, v1, v1\n\ts_endpgm\n", False),
        "descriptor": ("k:\n\ts_mov_b32 s2, 0x800\n.L1:\n\tbuffer_load_dword v1, v0, s[0:3], 0 offen\n"
                       "\tv_cmp_lt_i32_e64 s[2:3], s4, v0\n\ts_cbranch_scc1 .L1\n\ts_endpgm\n", True),
    }
    for name, (text, flagged) in cases.items():
        f = tmp_path / (name + ".s")
        f.write_text(text)
        r = _run(str(f))
        assert (r.returncode == 1) == flagged, (name, r.stdout)


def test_every_kernel_of_the_build_passes_the_check():
    """The gfx950 assembly the Makefile keeps beside the objects (--save-temps=obj) of EVERY HIP source
    is clean: `make check`.  __graft_entry__.build() runs the same."""
    build = os.path.join(ROOT, "detprocess_amd", "csrc", "build")
    files = sorted(glob.glob(os.path.join(build, "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if not files:
        pytest.skip("no build tree here (the .so travels without it): run `make -C detprocess_amd/csrc`")
    srcs = glob.glob(os.path.join(ROOT, "detprocess_amd", "csrc", "*.hip"))
    assert len(files) == len(srcs), (len(files), len(srcs))
    r = _run(*files)
    assert r.returncode == 0, r.stdout[-4000:]
