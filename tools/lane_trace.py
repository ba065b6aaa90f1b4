"""Kernel trace of the n > 128 pipeline (rocprofv3 --kernel-trace CSV): per kernel start / end relative to the first, by
stream, for the LAST solve in the trace — who overlaps whom in the two-lane form.  usage: lane_trace.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [r for r in rows if "large_" in r["Kernel_Name"]]
ks.sort(key=lambda r: int(r["Start_Timestamp"]))
# last solve: from the last large_init_kernel on
i0 = max(i for i, r in enumerate(ks) if "large_init" in r["Kernel_Name"])
ks = ks[i0:]
t0 = int(ks[0]["Start_Timestamp"])
import re
def short(nm):
    m = re.search(r"large_(\w+?)_kernel", nm)
    return m.group(1) if m else nm[:20]
qcol = "Queue_Id" if "Queue_Id" in ks[0] else None
scol = "Stream_Id" if "Stream_Id" in ks[0] else None
for r in ks[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{short(r['Kernel_Name']):12s} q={r.get(qcol, '?'):>3s} s={r.get(scol, '?'):>3s} {s/1e3:9.1f} .. {e/1e3:9.1f} us  ({(e-s)/1e3:7.1f})  grid={r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')}")
print("total", (int(ks[-1]["End_Timestamp"]) - t0) / 1e3, "us")
