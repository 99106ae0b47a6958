#!/usr/bin/env python3
"""Static check of gfx950 ISA text for data hazards the hardware does NOT interlock and that the
compiler's hazard recognizer cannot see when one side of the pair sits in an inline-asm statement
(LLVM's GCNHazardRecognizer skips INLINEASM both as producer -- `isVALU()` is false for it -- and as
consumer, apart from the dst_sel forwarding case).

    hipcc -O3 -fno-slp-vectorize -std=c++17 --offload-arch=gfx950 -Iinclude -S --cuda-device-only \
          detprocess_amd/csrc/ofx_fused25.hip -o /tmp/f25.s
    python tools/isa_hazards.py /tmp/f25.s [more.s ...]

Rules (CDNA3 ISA guide section 4.5 "manually inserted wait states", also valid for gfx950):
  trans   VALU transcendental (exp log rcp rsq sqrt sin cos) -> non-trans VALU reading the result : 1 wait state
  dpp     VALU writes a VGPR -> VALU DPP reads that VGPR                                        : 2 wait states
  execdpp VALU writes EXEC -> VALU DPP op                                                       : 5 wait states
  sgprvm  VALU writes an SGPR -> VMEM reads that SGPR                                           : 5 wait states
  lanesel VALU writes SGPR/VCC -> v_readlane / v_writelane using it as lane select              : 4 wait states
One instruction = one wait state, `s_nop n` = n + 1.  The scan is linear over the text of each
function (labels do not reset it: a fall-through is a legal path; a taken branch costs >= 1 wait
state more, never less), so a report is a real violation on the fall-through path.

Second pass -- `clobber`: integrity of scalar operands that must be loop-invariant data.  The function
is cut into basic blocks (labels, s_branch / s_cbranch_*, s_endpgm), reaching definitions of every
SGPR are iterated to a fixed point, and for every buffer_* instruction the four descriptor words and
the scalar offset, and for every s_load / s_buffer_load / global saddr the base pair, are checked: a
reaching definition that is a lane-mask producer (v_cmp*, s_{and,or,xor,andn2,orn2,nor,...}_b64,
s_cselect_b64, s_*_saveexec_b64) means the register allocator has handed part of a live descriptor
to a compare result.  This is the miscompile found in k_fused25<2, true> / <6, true> (round 2's
"wrong instantiation": the tbase descriptor s[48:51], built once in the prologue, had words 2-3
overwritten by `v_cmp_lt_i32_e64 s[50:51], ...` in the tail of the slot loop, and s50 doubled as the
0x800 row offset of two more loads).

Third pass -- `waitcnt`: every read (or overwrite) of a register that is the destination of a vector
memory load or an LDS read still in flight.  Model of SIInsertWaitcnts for gfx9: vmcnt counts loads
and stores, returns in order, so a load is retired by `s_waitcnt vmcnt(N)` iff at least N vector
memory instructions were issued after it; lgkmcnt likewise for LDS, except that scalar loads return
out of order, so while one is pending only lgkmcnt(0) retires anything.  Pending sets are merged at
joins (union, fewest later issues), iterated to a fixed point.

Fourth pass -- `execprologue`: a vector instruction (VALU, LDS, VMEM, scratch) that sits in a basic
block ABOVE the instruction that re-widens EXEC at the top of that block (`s_or_b64 exec, exec, s[..]`
of an end-of-if, `s_or_saveexec_b64` of an else).  Such a block is entered with the EXEC of one branch
side only, so the instruction runs for a subset of the lanes although the register allocator inserted
it (a live-range-split copy or a reload) as whole-wave code.  (`s_andn2_saveexec_b64` is the fused
form of an else entry and counts as well.)  This is the cause of round 2's wrong
k_fused25<2, true> / <6, true>: greedy RA rematerialised a scalar constant (`s_movk_i32`) at the very
top of the flow block of `tbh = (tl == 0) ? tb0hi : tb`; SIInstrInfo::isBasicBlockPrologue() stops
scanning at that instruction, so the split copies `v_mov_b64 v[112:113], v[84:85]` and
`v_mov_b64 v[110:111], v[80:81]` landed above `s_or_saveexec_b64` and lane 0 (thread 0, the lane the
branch had masked off) never received its values.
Exit status 1 if anything is found.
"""
import os
import re
import sys

TRANS = re.compile(r"^v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)_(f16|f32|f64|legacy_f32)")
REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc|exec|m0)(_lo|_hi)?\b")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        elif m.group(3):
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
        else:
            out.add((m.group(6), 0))
    return out


def parse(line):
    t = line.split(";")[0].split("//")[0].strip()
    if not t or t.endswith(":") or t.startswith("."):
        return None
    parts = t.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    # re-join register ranges split on ',' -- none on AMDGPU (ranges use ':'), modifiers are space separated
    return op, ops, t


def classify(op, ops):
    """-> (defs, uses, flags)"""
    flags = set()
    if op.startswith("v_"):
        flags.add("valu")
        if TRANS.match(op):
            flags.add("trans")
    joined = " ".join(ops)
    if "dpp" in op or "row_" in joined or "quad_perm" in joined or "wave_" in joined or "row_bcast" in joined:
        if op.startswith("v_"):
            flags.add("dpp")
    if op.startswith(("buffer_", "global_", "flat_", "scratch_", "image_", "tbuffer_")):
        flags.add("vmem")
    ndst = 0
    if op.startswith("v_"):
        ndst = 1
        if "_co_" in op or op.startswith("v_div_scale") or op.startswith("v_mad_u64") or op.startswith("v_mad_i64"):
            ndst = 2
        if op.startswith("v_cmpx"):
            return {("exec", 0)} | (regs(ops[0]) if ops and not ops[0].startswith("v") else set()), \
                   set().union(*[regs(o) for o in ops]) if ops else set(), flags
        if op.startswith(("v_nop", "v_swap")):
            ndst = 0
    elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch",
                                                     "s_endpgm", "s_setprio", "s_sleep", "s_store", "s_dcache",
                                                     "s_cmp", "s_bitcmp", "s_setreg", "s_sendmsg", "s_set_gpr")):
        ndst = 1
    elif op.startswith("ds_"):
        ndst = 1 if ("read" in op or "_rtn" in op or "bpermute" in op or "permute" in op or "swizzle" in op) else 0
    elif "vmem" in flags:
        ndst = 1 if ("load" in op or "_rtn" in op) and "lds" not in joined else 0
    defs = set()
    for o in ops[:ndst]:
        defs |= regs(o.split()[0] if o else o)
    uses = set()
    for o in ops[ndst:]:
        uses |= regs(o)
    if op.startswith("v_") and any(x in op for x in ("fmac", "mac_", "v_pk_fmac", "v_movrel", "v_writelane", "v_dot")) \
            or (op.startswith("v_") and ("dpp" in flags) and "bound_ctrl" not in joined):
        uses |= defs                      # dst is also a source (accumulator / old value of a DPP op)
    if op.startswith(("v_cndmask",)) and len(ops) == 3:
        uses.add(("vcc", 0))
    if op.startswith(("v_addc", "v_subb", "v_subbrev", "v_div_fmas")):
        uses.add(("vcc", 0))
    return defs, uses, flags


def scan(path):
    found = []
    fn = None
    hist = []          # (waitstates_ago, defs, flags, text, lineno)  newest last
    with open(path) as fh:
        for ln, line in enumerate(fh, 1):
            s = line.split(";")[0].strip()
            if s.endswith(":") and not s.startswith("."):
                if s.startswith("_Z") or s[0].isalpha():
                    fn = s[:-1]
                    hist = []
                continue
            p = parse(line)
            if p is None:
                continue
            op, ops, text = p
            if op == "s_nop":
                n = int(ops[0], 0) + 1 if ops else 1
                hist = [(a + n, d, f, t, l) for a, d, f, t, l in hist if a + n <= 8]
                continue
            defs, uses, flags = classify(op, ops)
            lane_sel = set()
            if op.startswith(("v_readlane", "v_writelane")) and len(ops) >= 3:
                lane_sel = regs(ops[2])
            for ago, pdefs, pflags, ptext, pl in hist:
                ws = ago                      # wait states between producer and this instruction
                if "valu" not in pflags:
                    continue
                hit = None
                if "trans" in pflags and "valu" in flags and "trans" not in flags and ws < 1 and (pdefs & uses):
                    hit = "trans"
                pv = {r for r in pdefs if r[0] == "v"}
                if "dpp" in flags and ws < 2 and (pv & uses):
                    hit = "dpp"
                if "dpp" in flags and ws < 5 and ("exec", 0) in pdefs:
                    hit = "execdpp"
                ps = {r for r in pdefs if r[0] in ("s", "vcc")}
                if "vmem" in flags and ws < 5 and (ps & uses):
                    hit = "sgprvm"
                if lane_sel and ws < 4 and (ps & lane_sel):
                    hit = "lanesel"
                if hit:
                    found.append((path, fn, hit, pl, ptext, ln, text, ws))
            hist = [(a + 1, d, f, t, l) for a, d, f, t, l in hist if a + 1 <= 8]
            hist.append((0, defs, flags, text, ln))
    return found


CONSTMIX = False      # --constmix: also report uses reached by disagreeing constant moves (noisy)
MASKDEF = re.compile(r"^(v_cmpx?_|s_(and|or|xor|andn2|orn2|nor|nand|xnor)_b64|s_cselect_b64|s_\w+_saveexec_b64|s_not_b64)")
TERM = ("s_branch", "s_cbranch_", "s_endpgm", "s_setpc", "s_swappc")


def sregs_of(tok):
    return {r for r in regs(tok) if r[0] == "s"}


def clobber_scan(path):
    """reaching definitions of SGPRs over the CFG of each function -> descriptor / base integrity"""
    found = []
    funcs = []                      # (name, [(lineno, label-or-None, parsed)])
    cur = None
    with open(path) as fh:
        for ln, line in enumerate(fh, 1):
            s = line.split(";")[0].strip()
            if s.endswith(":") and not s.startswith("//"):
                lab = s[:-1]
                if lab.startswith(".L"):
                    if cur is not None:
                        cur[1].append((ln, lab, None))
                elif not lab.startswith("."):
                    cur = (lab, [])
                    funcs.append(cur)
                continue
            if s.startswith(".amdhsa_kernel") or s.startswith(".section"):
                cur = None
                continue
            if cur is None:
                continue
            p = parse(line)
            if p is None:
                continue
            cur[1].append((ln, None, p))
    for name, items in funcs:
        # basic blocks
        blocks = []                 # each: dict(label, insts=[(ln, op, ops, text)], succ=[])
        blk = {"label": None, "insts": [], "succ": []}
        for ln, lab, p in items:
            if lab is not None:
                if blk["insts"] or blk["label"] is not None:
                    blocks.append(blk)
                blk = {"label": lab, "insts": [], "succ": []}
                continue
            op, ops, text = p
            blk["insts"].append((ln, op, ops, text))
            if op.startswith(TERM):
                blocks.append(blk)
                blk = {"label": None, "insts": [], "succ": []}
        if blk["insts"] or blk["label"] is not None:
            blocks.append(blk)
        if not blocks:
            continue
        index = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
        for i, b in enumerate(blocks):
            last = b["insts"][-1] if b["insts"] else None
            fall = True
            if last is not None:
                op, ops = last[1], last[2]
                if op == "s_branch":
                    fall = False
                    if ops and ops[0] in index:
                        b["succ"].append(index[ops[0]])
                elif op.startswith("s_cbranch_"):
                    if ops and ops[-1] in index:
                        b["succ"].append(index[ops[-1]])
                elif op.startswith(("s_endpgm", "s_setpc")):
                    fall = False
            if fall and i + 1 < len(blocks):
                b["succ"].append(i + 1)
        # definitions: id -> (ln, text, is_mask, const); per-instruction register defs (SGPRs and VGPRs)
        defs = []
        for b in blocks:
            b["d"] = []
            b["u"] = []
            for ln, op, ops, text in b["insts"]:
                d, u, fl = classify(op, ops)
                sd = {r for r in d if r[0] in ("s", "v")}
                if op.startswith("v_cmp") and ops:
                    sd |= sregs_of(ops[0])
                if op.startswith(("v_readlane", "v_readfirstlane")) and ops:
                    sd |= sregs_of(ops[0])
                did = None
                if sd:
                    did = len(defs)
                    const = None
                    if op in ("v_mov_b32_e32", "s_mov_b32", "s_movk_i32", "v_bfrev_b32_e32", "s_brev_b32") \
                            and len(ops) == 2 and not REG.search(ops[1]):
                        const = op[:6] + ops[1]
                    defs.append((ln, text, bool(MASKDEF.match(op)), const))
                b["d"].append((sd, did))
                b["u"].append({r for r in u if r[0] in ("s", "v")})
        NS = 128
        # block transfer: gen[r] = last def id in block, kill = regs defined
        for b in blocks:
            gen = {}
            for sd, did in b["d"]:
                for r in sd:
                    gen[r] = did
            b["gen"] = gen
            b["in"] = {}
        changed = True
        preds = [[] for _ in blocks]
        for i, b in enumerate(blocks):
            for j in b["succ"]:
                preds[j].append(i)
        outs = [dict() for _ in blocks]
        while changed:
            changed = False
            for i, b in enumerate(blocks):
                inn = {}
                for pj in preds[i]:
                    for r, ds in outs[pj].items():
                        inn.setdefault(r, set()).update(ds)
                b["in"] = inn
                out = {r: set(ds) for r, ds in inn.items()}
                for r, did in b["gen"].items():
                    out[r] = {did}
                if out != outs[i]:
                    outs[i] = out
                    changed = True
        # check uses
        for b in blocks:
            reach = {r: set(ds) for r, ds in b["in"].items()}
            for k_inst, ((ln, op, ops, text), (sd, did)) in enumerate(zip(b["insts"], b["d"])):
                check = []
                if op.startswith(("buffer_", "tbuffer_")):
                    for k, o in enumerate(ops):
                        m = re.match(r"s\[(\d+):(\d+)\]", o)
                        if m and int(m.group(2)) - int(m.group(1)) == 3:
                            check += [("descriptor word %d" % (q - int(m.group(1))), q)
                                      for q in range(int(m.group(1)), int(m.group(2)) + 1)]
                            if k + 1 < len(ops):
                                m2 = re.match(r"s(\d+)\b", ops[k + 1])
                                if m2:
                                    check.append(("scalar offset", int(m2.group(1))))
                            break
                elif op.startswith(("s_load_", "s_buffer_load_", "s_store_")) and len(ops) >= 2:
                    m = re.match(r"s\[(\d+):(\d+)\]", ops[1])
                    if m:
                        check += [("scalar base", q) for q in range(int(m.group(1)), int(m.group(2)) + 1)]
                elif op.startswith(("global_", "scratch_")):
                    for o in ops[1:]:
                        m = re.match(r"s\[(\d+):(\d+)\]", o)
                        if m and int(m.group(2)) - int(m.group(1)) == 1:
                            check += [("saddr", q) for q in range(int(m.group(1)), int(m.group(2)) + 1)]
                for what, r in check:
                    for d in reach.get(("s", r), ()):
                        if defs[d][2]:
                            found.append((path, name, "clobber", defs[d][0], defs[d][1], ln,
                                          text + "    <- " + what + " s%d" % r, -1))
                # a use reached only by constant moves that disagree: a rematerialised loop-invariant
                # constant whose register was reused without the restore on some path
                for r in (b["u"][k_inst] if CONSTMIX else ()):
                    ds = reach.get(r, ())
                    if len(ds) >= 2:
                        cs = {defs[d][3] for d in ds}
                        if None not in cs and len(cs) >= 2:
                            found.append((path, name, "constmix", min(defs[d][0] for d in ds),
                                          " | ".join(sorted(defs[d][1] for d in ds)), ln, text, -1))
                for r in sd:
                    reach[r] = {did}
    return found


WAITRE = re.compile(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)")


def build_cfg(path):
    """-> [(name, blocks)], blocks as in clobber_scan (label, insts, succ)"""
    funcs = []
    cur = None
    with open(path) as fh:
        for ln, line in enumerate(fh, 1):
            s = line.split(";")[0].strip()
            if s.endswith(":") and not s.startswith("//"):
                lab = s[:-1]
                if lab.startswith(".L"):
                    if cur is not None:
                        cur[1].append((ln, lab, None))
                elif not lab.startswith("."):
                    cur = (lab, [])
                    funcs.append(cur)
                continue
            if s.startswith(".amdhsa_kernel") or s.startswith(".section"):
                cur = None
                continue
            if cur is None:
                continue
            p = parse(line)
            if p is not None:
                cur[1].append((ln, None, p))
    out = []
    for name, items in funcs:
        blocks = []
        blk = {"label": None, "insts": [], "succ": []}
        for ln, lab, p in items:
            if lab is not None:
                if blk["insts"] or blk["label"] is not None:
                    blocks.append(blk)
                blk = {"label": lab, "insts": [], "succ": []}
                continue
            op, ops, text = p
            blk["insts"].append((ln, op, ops, text))
            if op.startswith(TERM):
                blocks.append(blk)
                blk = {"label": None, "insts": [], "succ": []}
        if blk["insts"] or blk["label"] is not None:
            blocks.append(blk)
        index = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
        for i, b in enumerate(blocks):
            last = b["insts"][-1] if b["insts"] else None
            fall = True
            if last is not None:
                op, ops = last[1], last[2]
                if op == "s_branch":
                    fall = False
                    if ops and ops[0] in index:
                        b["succ"].append(index[ops[0]])
                elif op.startswith("s_cbranch_"):
                    if ops and ops[-1] in index:
                        b["succ"].append(index[ops[-1]])
                elif op.startswith(("s_endpgm", "s_setpc")):
                    fall = False
            if fall and i + 1 < len(blocks):
                b["succ"].append(i + 1)
        if blocks:
            out.append((name, blocks))
    return out


def waitcnt_scan(path):
    found = []
    for name, blocks in build_cfg(path):
        preds = [[] for _ in blocks]
        for i, b in enumerate(blocks):
            for j in b["succ"]:
                preds[j].append(i)

        def step(state, inst, report):
            """state = (vm: {reg: (after, ln, text)}, lg: {reg: (after, ln, text)}, smem_pending: bool)"""
            vm, lg, smem = state
            ln, op, ops, text = inst
            d, u, fl = classify(op, ops)
            if op == "s_waitcnt":
                for m in WAITRE.finditer(" ".join(ops)):
                    n = int(m.group(2))
                    if m.group(1) == "vmcnt":
                        vm = {r: v for r, v in vm.items() if v[0] < n}
                    elif m.group(1) == "lgkmcnt":
                        if n == 0:
                            lg = {}
                            smem = False
                        elif not smem:
                            lg = {r: v for r, v in lg.items() if v[0] < n}
                return vm, lg, smem
            touched = {r for r in (d | u) if r[0] in ("v", "s")}
            if "vmem" in fl:
                touched = {r for r in u if r[0] in ("v", "s")}    # load over load: in-order return, no hazard
            if report is not None:
                for r in touched:
                    if r in vm:
                        report.append((path, name, "waitcnt", vm[r][1], vm[r][2], ln, text + "    <- vmcnt, %s%d" % r, -1))
                    if r in lg:
                        report.append((path, name, "waitcnt", lg[r][1], lg[r][2], ln, text + "    <- lgkmcnt, %s%d" % r, -1))
            is_vm = "vmem" in fl
            is_lds = op.startswith("ds_")
            is_smem = op.startswith(("s_load_", "s_buffer_load_", "s_store_", "s_memtime", "s_memrealtime", "s_dcache", "s_atc"))
            if is_vm:
                # (vmcnt is a 6-bit counter: the 64th outstanding operation cannot issue before the oldest
                # has returned, and loads return in order -- a load with 63 younger operations behind it
                # has completed, which is what the compiler's own wait insertion relies on)
                vm = {r: (v[0] + 1, v[1], v[2]) for r, v in vm.items() if v[0] + 1 < 63}
                if "lds" not in " ".join(ops).split():
                    for r in d:
                        if r[0] == "v":
                            vm[r] = (0, ln, text)
            if is_lds or is_smem:
                lg = {r: (v[0] + 1, v[1], v[2]) for r, v in lg.items()}
                for r in d:
                    if r[0] in ("v", "s"):
                        lg[r] = (0, ln, text)
                if is_smem:
                    smem = True
            return vm, lg, smem

        def merge(a, b):
            if a is None:
                return ({r: v for r, v in b[0].items()}, {r: v for r, v in b[1].items()}, b[2])
            vm, lg, smem = a
            for src, dst in ((b[0], vm), (b[1], lg)):
                for r, v in src.items():
                    if r not in dst or v[0] < dst[r][0]:
                        dst[r] = v
            return vm, lg, smem or b[2]

        outs = [None] * len(blocks)
        ins = [None] * len(blocks)
        ins[0] = ({}, {}, False)
        work = list(range(len(blocks)))
        it = 0
        while work and it < 40 * len(blocks):
            it += 1
            i = work.pop(0)
            st = None
            if i == 0:
                st = ({}, {}, False)
            for pj in preds[i]:
                if outs[pj] is not None:
                    st = merge(st, outs[pj])
            if st is None:
                continue
            ins[i] = ({r: v for r, v in st[0].items()}, {r: v for r, v in st[1].items()}, st[2])
            for inst in blocks[i]["insts"]:
                st = step(st, inst, None)
            key = (sorted((r, v[0]) for r, v in st[0].items()), sorted((r, v[0]) for r, v in st[1].items()), st[2])
            old = outs[i]
            oldkey = None if old is None else (sorted((r, v[0]) for r, v in old[0].items()),
                                               sorted((r, v[0]) for r, v in old[1].items()), old[2])
            if key != oldkey:
                outs[i] = st
                for j in blocks[i]["succ"]:
                    if j not in work:
                        work.append(j)
        seen = set()
        for i, b in enumerate(blocks):
            if ins[i] is None:
                continue
            st = ins[i]
            for inst in b["insts"]:
                rep = []
                st = step(st, inst, rep)
                for x in rep:
                    k = (x[3], x[5])
                    if k not in seen:
                        seen.add(k)
                        found.append(x)
    return found


# end of an if (`s_or_b64 exec, exec, saved`) and the two forms of an else entry
EXECWIDEN = re.compile(r"^(s_or_b64\s+exec,\s*exec,|s_or_saveexec_b64\b|s_andn2_saveexec_b64\b)")


def execprologue_scan(path):
    found = []
    for name, blocks in build_cfg(path):
        preds = [[] for _ in blocks]
        for i, b in enumerate(blocks):
            for j in b["succ"]:
                preds[j].append(i)

        def narrows_at_end(pb):
            """the block hands over an EXEC it has just narrowed (the head of an if / else / loop)"""
            for ln, op, ops, text in pb["insts"][-4:]:
                if op.startswith("s_") and ("saveexec" in op or re.match(r"s_\w+\s+exec\b", text)) \
                        and not re.match(r"s_or_b64\s+exec,\s*exec,", text):
                    return True
                if op.startswith("v_cmpx"):
                    return True
            return False

        for bi, b in enumerate(blocks):
            # a block all of whose predecessors have just narrowed EXEC is the body of that branch,
            # and a widening at its end is the (tail-duplicated) end of the if: not this pattern
            if not preds[bi] or all(narrows_at_end(blocks[pj]) for pj in preds[bi]):
                continue
            vec = None
            for ln, op, ops, text in b["insts"]:
                if EXECWIDEN.match(text):
                    if vec is not None:
                        found.append((path, name, "execprologue", vec[0], vec[1], ln, text, -1))
                    break
                if op.startswith("s_") and (re.search(r"\bexec\b", text) or "saveexec" in op) \
                        or op.startswith("v_cmpx"):
                    break                       # EXEC narrowed inside this block first: a region of its own
                if vec is None and op.startswith(("v_", "ds_", "buffer_", "global_", "scratch_", "flat_")) \
                        and not op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
                    vec = (ln, text)
    return found


def split_functions(path, outdir):
    """one file per function (label that is not a local .L label ... its .amdhsa_kernel / .section)"""
    parts = []
    cur = None
    with open(path) as fh:
        for line in fh:
            s = line.split(";")[0].strip()
            if s.endswith(":") and not s.startswith(".") and not s.startswith("//"):
                if cur is not None:
                    cur[1].close()
                name = os.path.join(outdir, "%s.%d.s" % (os.path.basename(path), len(parts)))
                cur = (name, open(name, "w"))
                parts.append(name)
            elif s.startswith(".amdhsa_kernel") or s.startswith(".section"):
                if cur is not None:
                    cur[1].close()
                    cur = None
            if cur is not None:
                cur[1].write(line)
    if cur is not None:
        cur[1].close()
    return parts


def check_one(arg):
    orig, part = arg
    f = scan(part) + clobber_scan(part) + waitcnt_scan(part) + execprologue_scan(part)
    return [(orig,) + x[1:] for x in f]


def main():
    import multiprocessing
    import tempfile
    global CONSTMIX
    if "--constmix" in sys.argv:
        CONSTMIX = True
        sys.argv.remove("--constmix")
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        jobs = []
        for path in sys.argv[1:]:
            jobs += [(path, part) for part in split_functions(path, td)]
        # (line numbers in the report are relative to the start of the function)
        jobs.sort(key=lambda j: -os.path.getsize(j[1]))
        nproc = max(1, min(len(jobs), len(os.sched_getaffinity(0))))
        if nproc > 1:
            with multiprocessing.Pool(nproc) as pool:
                results = pool.map(check_one, jobs, chunksize=1)
        else:
            results = [check_one(j) for j in jobs]
    by_path = {path: [] for path in sys.argv[1:]}
    for r in results:
        for x in r:
            by_path[x[0]].append(x)
    for path in sys.argv[1:]:
        f = by_path[path]
        bad += len(f)
        byfn = {}
        for x in f:
            byfn.setdefault(x[1], []).append(x)
        print(f"{path}: {len(f)} hazard(s) in {len(byfn)} function(s)")
        for fn, xs in byfn.items():
            print(f"  {fn}: {len(xs)}")
            for _, _, kind, pl, pt, ln, t, ws in xs[:12]:
                print(f"    [{kind}]" + (f" wait states {ws}" if ws >= 0 else "") + f"\n       +{pl}: {pt}\n       +{ln}: {t}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
