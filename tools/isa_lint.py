#!/usr/bin/env python3
"""ISA lint for the inline-asm MFMAs: hipcc's hazard recognizer does not look inside asm statements, so an
accumulator copy it inserts (v_accvgpr_read / v_accvgpr_mov / any non-MFMA reader of an AGPR) could overtake
the matrix core.  This scans the gfx950 code objects of the built translation units and fails if such a
reader follows an MFMA that wrote the same AGPR with fewer than MIN_WAIT wait states in between
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

AREG = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")


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
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=td, check=True, capture_output=True)
        cos = [p for p in glob.glob(local + ".*") if "amdgcn" in p]
        for co in cos:
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
            func = "?"
            last_write = {}   # agpr -> wait states since the MFMA that wrote it
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    func, last_write = m.group(1), {}
                    continue
                ins = line.split("//")[0].strip()
                if not ins or ins.startswith("."):
                    continue
                op, _, rest = ins.partition(" ")
                step = 1
                if op == "s_nop":
                    step = int(rest.strip()) + 1
                if op.startswith("v_mfma"):
                    dst = rest.split(",")[0]
                    # the matrix core runs one MFMA at a time: this one could only issue once the previous one was
                    # (passes - 1) issue slots into its execution (8-pass f32 16x16x4, 16-pass f64 16x16x4)
                    step = 15 if "f64" in op else 7
                    for r in last_write:
                        last_write[r] += step
                    for r in aregs(dst):
                        last_write[r] = 0
                    continue
                # readers: anything that names an AGPR as a source
                srcs = rest.split(",", 1)[1] if "," in rest else ""
                if op.startswith("v_accvgpr_write"):
                    srcs = ""
                for r in aregs(srcs):
                    if r in last_write and last_write[r] < MIN_WAIT:
                        problems.append(f"{os.path.basename(obj)}: {func[:90]}: `{ins}` reads a{r} {last_write[r]} wait states after an MFMA wrote it")
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
