#!/usr/bin/env python3
"""Print VGPR / AGPR / SGPR-spill / occupancy figures of the kernels in a hipcc object (gfx950 code object notes).
usage: python tools/kernel_regs.py <object.o> [regex]"""
import glob, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main(obj, pat=None):
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, os.path.basename(obj))
        with open(obj, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=td, check=True, capture_output=True)
        for co in [p for p in glob.glob(local + ".*") if "amdgcn" in p]:
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                f = dict(re.findall(r"\.(\w+):\s+(\S+)", ".agpr_count:" + blk))
                name = f.get("name", "?")
                if pat and not re.search(pat, name):
                    continue
                total = int(f.get("vgpr_count", 0))
                waves = min(8, 512 // max(8, (total + 7) // 8 * 8))
                print(f"{name[:100]}: vgpr+agpr={total} agpr={f.get('agpr_count')} sgpr_spills={f.get('sgpr_spill_count')} "
                      f"vgpr_spills={f.get('vgpr_spill_count')} scratch={f.get('private_segment_fixed_size')} -> {waves} waves/SIMD")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
