#!/usr/bin/env python3
"""ISA lint for the inline-asm MFMAs: hipcc's hazard recognizer does not look inside asm statements, so an
accumulator copy it inserts (v_accvgpr_read / v_accvgpr_mov / any non-MFMA reader of an AGPR) could overtake
the matrix core.  This scans the gfx950 code objects of the built translation units and fails if such a
reader follows an MFMA that wrote the same AGPR (or, in the VGPR-accumulator translation units (-DTOA_ACC_VGPR: ba_schur), any
instruction that touches a VGPR an MFMA wrote) with fewer than MIN_WAIT wait states in between
(every instruction = 1, `s_nop N` = N + 1, a later MFMA = its passes - 1; straight-line approximation, conservative).

usage: python tools/isa_lint.py [objects...]    (default: tinyopt_amd/csrc/_obj/*.o)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MIN_WAIT = 19  # XDL write VGPR -> VALU read, 16-pass op (CDNA3/4 ISA guide, software wait states)
MIN_WAIT_MEM = 18  # XDL write VGPR -> LDS / VMEM / FLAT read of that register, 16-pass op: one less than for a VALU reader
                   # (what hipcc itself leaves after a builtin DGEMM MFMA whose result it stores straight from the AGPRs)
MEM_READER = re.compile(r"^(ds_|buffer_|global_|flat_|scratch_)")

AREG = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
VMEM = re.compile(r"^(buffer|global|flat|scratch)_(load|store|atomic)")
COMPILER_COUNTED_LOADS = ("ts_data_pass",)   # large_fused.hip: __builtin_amdgcn_raw_buffer_load_* only


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def touched_vregs(op, text):
    """VGPRs an instruction reads or writes.  Packed-f32 ops name 64-bit pairs but a source whose op_sel and
    op_sel_hi pick the same half reads only that one register (the other may legally belong to a pending load)."""
    if not op.startswith("v_pk_") or "f32" not in op:
        return vregs(text)
    ops_txt = re.split(r"\s+op_sel", text)[0]
    fields = [f.strip() for f in ops_txt.split(",")]
    nsrc = len(fields) - 1
    sel = [0] * nsrc
    sel_hi = [1] * nsrc
    m = re.search(r"op_sel:\[([\d,]+)\]", text)
    if m:
        sel = [int(x) for x in m.group(1).split(",")][:nsrc] + [0] * max(0, nsrc - len(m.group(1).split(",")))
    m = re.search(r"op_sel_hi:\[([\d,]+)\]", text)
    if m:
        sel_hi = [int(x) for x in m.group(1).split(",")][:nsrc] + [1] * max(0, nsrc - len(m.group(1).split(",")))
    out = set(vregs(fields[0]))
    for k, f in enumerate(fields[1:]):
        regs = sorted(vregs(f))
        if len(regs) == 2:
            out.add(regs[sel[k]])
            out.add(regs[sel_hi[k]])
        else:
            out.update(regs)
    return out


def lint_inflight(func, insts):
    """Second rule: the load pipeline keeps buffer loads in flight across arithmetic (asm statements whose outputs
    are only valid after a later `s_waitcnt vmcnt`).  No instruction may touch the destination VGPRs of a load
    that is still outstanding - e.g. a register copy the compiler inserts at a loop back-edge would read stale
    data.  insts: [(offset, op, rest)].  Dataflow over the control-flow graph: the state is the in-order list of
    outstanding vector-memory accesses; `s_waitcnt vmcnt(n)` leaves the youngest n."""
    problems = set()
    index_of = {off: i for i, (off, _, _) in enumerate(insts)}
    n_inst = len(insts)

    def target(rest):
        m = re.search(r"\+0x([0-9a-f]+)>", rest)
        return index_of.get(int(m.group(1), 16)) if m else None

    seen = set()
    work = [(0, ())]
    while work:
        i, state = work.pop()
        while True:
            if i >= n_inst or (i, state) in seen:
                break
            seen.add((i, state))
            off, op, rest = insts[i]
            if op == "s_endpgm":
                break
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", rest)
                if m:
                    n = int(m.group(1))
                    state = state[len(state) - n:] if n < len(state) else state
                    if n == 0:
                        state = ()
                    while state and not state[0][0]:
                        state = state[1:]
            else:
                touched = touched_vregs(op, rest.split("<")[0])
                sink_load = op == "buffer_load_ubyte"   # L2 prefetches into a register nobody reads (dense_row.hpp q_prefetch)
                for dst, loff in state:
                    hit = touched & set(dst)
                    if hit and sink_load and insts[index_of[loff]][1] == "buffer_load_ubyte" and min(hit) == min(dst):
                        continue   # one prefetch overwriting the sink of another: write after write of a value never read
                    if hit:
                        problems.add(f"{func[:90]}: `{op} {rest.split('<')[0].strip()}` (+{off:#x}) touches v{min(hit)} while the load issued at +{loff:#x} is in flight")
                if VMEM.match(op):
                    # only the hand-written pipeline's loads (`buffer_load ... offen` from the asm statements) are
                    # protected; every other vector-memory access just occupies a vmcnt slot behind them
                    # (an LDS-DMA load, `... offen lds`, has no register destination: its first operand is the address)
                    prot = op.startswith("buffer_load") and "offen" in rest and not re.search(r"\blds\b", rest)
                    dst = tuple(sorted(vregs(rest.split(",")[0]))) if prot else ()
                    state = state + ((dst, off if prot else 0),)
                    while state and not state[0][0]:   # unprotected accesses older than every protected one retire first
                        state = state[1:]
                    if len(state) > 63:
                        state = state[-63:]
            if op == "s_branch":
                t = target(rest)
                if t is None:
                    break
                i = t
                continue
            if op.startswith("s_cbranch"):
                t = target(rest)
                if t is not None:
                    work.append((t, state))
            i += 1
    return sorted(problems)


def aregs(text):
    out = set()
    for m in AREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint_object(obj):
    problems = []
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, os.path.basename(obj))
        with open(obj, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        if obj.endswith(".co"):   # a bare gfx950 code object (e.g. one taken out of the run-time compiler's cache, tools/jit_lint.py)
            cos = [local]
        else:
            subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=td, check=True, capture_output=True)
            cos = [p for p in glob.glob(local + ".*") if "amdgcn" in p]
        for co in cos:
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
            func = "?"
            func_addr = 0
            insts = []
            last_write = {}   # agpr -> wait states since the MFMA that wrote it
            for line in dis.splitlines() + ["0 <end>:"]:
                m = re.match(r"^([0-9a-f]+) <(.+)>:", line)
                if m:
                    # (functions whose buffer loads are all COMPILER-visible builtins — hipcc counts those itself and waits before
                    #  any use — are exempt from the in-flight rule, which exists for the hand-issued asm loads)
                    if insts and not any(tag in func for tag in COMPILER_COUNTED_LOADS):
                        problems += [f"{os.path.basename(obj)}: {q}" for q in lint_inflight(func, insts)]
                    func, last_write, insts = m.group(2), {}, []
                    func_addr = int(m.group(1), 16)
                    continue
                ins = line.split("//")[0].strip()
                if not ins or ins.startswith("."):
                    continue
                op, _, rest = ins.partition(" ")
                am = re.search(r"//\s*([0-9A-Fa-f]+):", line)
                tm = re.search(r"<[^>]*\+0x[0-9a-f]+>", line)
                if am:
                    insts.append((int(am.group(1), 16) - func_addr, op, rest + (" " + tm.group(0) if tm else "")))
                step = 1
                if op == "s_nop":
                    step = int(rest.strip()) + 1
                if op.startswith("v_mfma"):
                    dst = rest.split(",")[0]
                    # the matrix core runs one MFMA at a time: this one could only issue once the previous one was
                    # (passes - 1) issue slots into its execution (8-pass f32 16x16x4, 16-pass f64 16x16x4)
                    step = 15 if "f64" in op else 7
                    # VGPR-form accumulators (-DTOA_ACC_VGPR translation units): an MFMA whose A / B operand is a register a
                    # recent MFMA wrote would need the XDL-write -> XDL-read-SrcA/B wait states (SrcC == vDst accumulation does not)
                    ab = rest.split(",")[1:3]
                    for r in vregs(",".join(ab)):
                        if ("v", r) in last_write and last_write[("v", r)] < MIN_WAIT:
                            problems.append(f"{os.path.basename(obj)}: {func[:90]}: `{ins}` reads v{r} as an A/B operand {last_write[('v', r)]} wait states after an MFMA wrote it")
                    for r in last_write:
                        last_write[r] += step
                    for r in aregs(dst):
                        last_write[r] = 0
                    if dst.strip().startswith("v"):
                        for r in vregs(dst):
                            last_write[("v", r)] = 0
                    continue
                # readers: anything that names an AGPR as a source
                srcs = rest.split(",", 1)[1] if "," in rest else ""
                if op.startswith("v_accvgpr_write"):
                    srcs = ""
                need = MIN_WAIT_MEM if MEM_READER.match(op) else MIN_WAIT
                for r in aregs(srcs):
                    if r in last_write and last_write[r] < need:
                        problems.append(f"{os.path.basename(obj)}: {func[:90]}: `{ins}` reads a{r} {last_write[r]} wait states after an MFMA wrote it")
                # VGPR-form accumulators: any non-MFMA instruction that reads OR writes (WAW) a VGPR an MFMA wrote too recently
                if any(isinstance(k, tuple) for k in last_write) and not op.startswith("s_"):
                    for r in vregs(rest.split("<")[0]):
                        k = ("v", r)
                        if k in last_write:
                            if last_write[k] < need:
                                problems.append(f"{os.path.basename(obj)}: {func[:90]}: `{ins}` touches v{r} {last_write[k]} wait states after an MFMA wrote it")
                            last_write.pop(k, None)
                # any write to an AGPR by a non-MFMA instruction ends the tracking of that register
                dst = rest.split(",")[0]
                if op.startswith("v_accvgpr_write") or op.startswith("v_accvgpr_mov"):
                    for r in aregs(dst):
                        last_write.pop(r, None)
                for r in list(last_write):
                    last_write[r] += step
                    if last_write[r] > 64:
                        del last_write[r]
    return problems


def main(argv):
    objs = argv or sorted(glob.glob(os.path.join(ROOT, "tinyopt_amd", "csrc", "_obj", "*.o")))
    bad = []
    for o in objs:
        bad += lint_object(o)
    for b in bad:
        print("HAZARD", b)
    print(f"isa_lint: {len(objs)} objects, {len(bad)} hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
