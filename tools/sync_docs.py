#!/usr/bin/env python3
"""Re-writes the headline numbers quoted in DESIGN.md / README.md / profiles/README.md from profiles/r01_*.json so the
prose cannot drift from the committed measurements.   usage: python tools/sync_docs.py [round tag, default r01]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
L = lambda f: json.load(open(os.path.join(ROOT, "profiles", f)))
b, c3, u, pm = L(f"{tag}_bench_c4.json"), L(f"{tag}_bench_c3.json"), L(f"{tag}_bench_under_rocprof.json"), L(f"{tag}_pmc.json")
rf, r3 = b["roofline"], c3["roofline"]


def sub(path, pat, rep, count=1):
    s = open(path).read()
    s2, k = re.subn(pat, rep, s, count=count, flags=re.S)
    if k == 0:
        print(f"[sync_docs] pattern not found in {os.path.basename(path)}: {pat[:60]}...")
    open(path, "w").write(s2)


D = os.path.join(ROOT, "DESIGN.md")
sub(D, r"\| the sum of its phases \| [0-9.]+ ms / launch = [0-9.]+ M LM it/s, [0-9.]+ TB/s algorithmic = \*\*[0-9.]+ % of the 8 TB/s HBM peak\*\* \([0-9]+ % of the [0-9.]+ TB/s this GPU streams in a read-only probe\) \|",
    f"| the sum of its phases | {b['ms_per_step']:.2f} ms / launch = {b['value']/1e6:.1f} M LM it/s, {rf['achieved']/1e3:.2f} TB/s algorithmic = **{100*rf['frac']:.1f} % of the 8 TB/s HBM peak** ({100*rf['frac_of_measured_ceiling']:.0f} % of the {rf['measured_read_ceiling_GBps']/1e3:.2f} TB/s this GPU streams in a read-only probe) |")
sub(D, r"C4 shard [0-9.]+ M LM it/s \([0-9.]+ ms/step\),\n[0-9.]+ TB/s = [0-9.]+ % of HBM peak on the fused kernel \(measured read ceiling [0-9.]+ TB/s\); CPU oracle [0-9]+ it/s on one\nEPYC 9575F core \(×[0-9]+;",
    f"C4 shard {b['value']/1e6:.2f} M LM it/s ({b['ms_per_step']:.2f} ms/step),\n{rf['achieved']/1e3:.2f} TB/s = {100*rf['frac']:.1f} % of HBM peak on the fused kernel (measured read ceiling {rf['measured_read_ceiling_GBps']/1e3:.2f} TB/s); CPU oracle {b['cpu_baseline']['value']:.0f} it/s on one\nEPYC 9575F core (×{b['config']['speedup_vs_cpu_1thread']:.0f};")
sub(D, r"C3 \(fp64 n=12\): [0-9.]+ M LM it/s, [0-9.]+ ms/launch \([0-9.]+ % of HBM peak;",
    f"C3 (fp64 n=12): {c3['value']/1e6:.1f} M LM it/s, {c3['ms_per_step']:.2f} ms/launch ({100*r3['frac']:.1f} % of HBM peak;")
sub(D, r"HBM bytes per launch = [0-9.]+ GB vs [0-9.]+ GB algorithmic \(ratio [0-9.]+,",
    f"HBM bytes per launch = {pm['hbm_bytes_per_launch']/1e9:.2f} GB vs {pm['algorithmic_bytes_per_launch']/1e9:.2f} GB algorithmic (ratio {pm['traffic_over_algorithmic']:.3f},")
sub(D, r"CPU [0-9.]+ k it/s \(×[0-9]+\)\.", f"CPU {c3['cpu_baseline']['value']/1e3:.1f} k it/s (×{c3['config']['speedup_vs_cpu_1thread']:.0f}).")
R = os.path.join(ROOT, "README.md")
sub(R, r"\*\*[0-9.]+ M LM\niterations/s\*\*, fused kernel at [0-9.]+ TB/s algorithmic = \*\*[0-9.]+ % of the 8 TB/s HBM peak\*\* \(HBM traffic measured\n= [0-9.]+ × algorithmic; a read-only probe streams [0-9.]+ TB/s on the same part\), vs [0-9.]+ k it/s",
    f"**{b['value']/1e6:.1f} M LM\niterations/s**, fused kernel at {rf['achieved']/1e3:.2f} TB/s algorithmic = **{100*rf['frac']:.1f} % of the 8 TB/s HBM peak** (HBM traffic measured\n= {pm['traffic_over_algorithmic']:.3f} × algorithmic; a read-only probe streams {rf['measured_read_ceiling_GBps']/1e3:.2f} TB/s on the same part), vs {b['cpu_baseline']['value']/1e3:.1f} k it/s")
sub(R, r"C3 \(10 000 × n=12 × m=500, fp64\): [0-9.]+ M LM iterations/s(?: \([0-9]+ % of HBM peak\))*", f"C3 (10 000 × n=12 × m=500, fp64): {c3['value']/1e6:.0f} M LM iterations/s ({100*r3['frac']:.0f} % of HBM peak)")
PR = os.path.join(ROOT, "profiles", "README.md")
s = open(PR).read()
a = s.index("Headline (round 1, final)")
s = s[:a] + f"""Headline (round 1, final): `lm_fused_kernel<DenseRowModel<float,3,3>>` {u['roofline']['kernel_ms_avg']:.2f} ms average per launch by the bench's own HIP
events in the profiled run (rocprofv3 kernel-trace durations of the same launches agree within 1 %; see `{tag}_kernel_stats.csv`,
whose average also contains the slower first launches), {b['ms_per_step']:.2f} ms un-profiled; {rf['passes_per_launch']:.0f} data passes per
launch x 408 000 B = {pm['algorithmic_bytes_per_launch']/1e9:.2f} GB algorithmic, {pm['hbm_bytes_per_launch']/1e9:.2f} GB measured HBM traffic (ratio {pm['traffic_over_algorithmic']:.4f})
-> {rf['achieved']/1e3:.2f} TB/s = {100*rf['frac']:.1f} % of 8 TB/s ({b['value']/1e6:.2f} M LM iterations/s); matrix-core flops issued by the accumulate passes:
{rf['mfma_secondary']['achieved']:.0f} TFLOP/s = {100*rf['mfma_secondary']['frac']:.0f} % of the 157.3 TF f32 MFMA peak.  C3: {c3['value']/1e6:.1f} M it/s, {100*r3['frac']:.1f} %.
Box-to-box spread of the same binary is about +-3 %.  SQ counters of the same kernel (`{tag}_pmc.json`): ~1.9 GHz
(GRBM_GUI_ACTIVE / 8 XCDs / duration), MFMA busy ~45 % + VALU active ~43 % of the SIMD cycles (they do not overlap for f32 MFMA).
`{tag}_large_n_kernel_stats.csv`: kernel split of the n > 63 path (tools/prof_large_n.py; DESIGN.md §4b).
"""
lp = os.path.join(ROOT, "profiles", f"{tag}_bench_large128.json")
if os.path.exists(lp):
    d = json.load(open(lp))
    s += (f"`{tag}_bench_large128.json`: the same bench line for the n > 63 path (`python bench.py --workload large128`): {d['value']/1e3:.0f} k LM it/s,\n"
          f"{d['ms_per_step']:.1f} ms per batched solve, {d['roofline']['achieved']:.1f} TFLOP/s of GEMM work = {100*d['roofline']['frac']:.0f} % of the f32 MFMA peak over the whole pass, "
          f"x{d['config']['speedup_vs_cpu_1thread']:.0f} the CPU oracle.\n")
open(PR, "w").write(s)
print("[sync_docs] done:", f"C4 {b['value']/1e6:.2f} M it/s {100*rf['frac']:.1f} %; C3 {c3['value']/1e6:.1f} M it/s {100*r3['frac']:.1f} %")
