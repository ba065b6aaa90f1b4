#!/usr/bin/env python3
"""Scan the gfx950 code objects for the pattern that cost large_chol_solve_kernel two thirds of its trailing update (round 5):
an `s_waitcnt vmcnt(0)` within a few instructions IN FRONT OF a global store, many times per kernel — hipcc emits it when the stored
value (or its address) is first read after a branch join while loads are pending, and every such store then also waits for the
acknowledgement of the store before it.  Prints, per kernel, stores / stores with such a wait / scratch (spill) accesses.
usage: python tools/store_wait_scan.py tinyopt_amd/csrc/_obj/*.o"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main(objs):
    rows = []
    for obj in objs:
        with tempfile.TemporaryDirectory() as td:
            local = os.path.join(td, os.path.basename(obj))
            subprocess.run(["cp", obj, local], check=True)
            subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=td, check=True, capture_output=True)
            for f in os.listdir(td):
                if "gfx950" not in f:
                    continue
                dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", os.path.join(td, f)], check=True, capture_output=True, text=True).stdout
                name, body = None, []
                def flush():
                    if name is None:
                        return
                    stores = waits = scratch = 0
                    for i, ins in enumerate(body):
                        if ins.startswith("scratch_"):
                            scratch += 1
                        if ins.startswith("global_store") or ins.startswith("flat_store"):
                            stores += 1
                            if any(b.startswith("s_waitcnt vmcnt(0)") for b in body[max(0, i - 6):i]):
                                waits += 1
                    if stores:
                        rows.append((waits, stores, scratch, os.path.basename(obj), name))
                for line in dis.splitlines():
                    m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
                    if m:
                        flush()
                        name, body = m.group(1), []
                    elif line.startswith("\t"):
                        body.append(line.strip().split("//")[0].strip())
                flush()
    rows.sort(reverse=True)
    for waits, stores, scratch, obj, name in rows[:60]:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print(f"{waits:4d} of {stores:4d} stores behind a vmcnt(0)   scratch {scratch:4d}   {obj:22s} {dem[:150]}")


if __name__ == "__main__":
    main(sys.argv[1:])
