#!/usr/bin/env python3
"""Re-writes the round-3 measurement table of DESIGN.md §5 (between the <!-- r03-table --> markers) from profiles/r03_*.json, so
that the table cannot drift from the committed bench lines.   usage: python tools/design_table.py"""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = lambda n: json.loads(open(os.path.join(ROOT, "profiles", f"r03_{n}.json")).read())
c4, c40, c400, c3 = L("bench_c4"), L("bench_c4_coop0"), L("bench_c4_memo0_coop0"), L("bench_c3")
c2, c5, c1, l128, l256, ba, bl = L("bench_c2"), L("bench_c5"), L("bench_c1"), L("bench_large128"), L("bench_large256"), L("bench_ba"), L("bench_balists")
l256r, blr = L("bench_large256_rocsolver"), L("bench_balists_rocsolver")
pm, pm3, pml = L("pmc"), L("pmc_c3"), L("pmc_large128")
r = c4["roofline"]; pi = r["per_iteration"]
rows = []
rows.append(f"| c4 | **{c4['value']/1e6:.2f} M LM it/s** (`value_at_oracle_iters` {c4['config']['value_at_oracle_iters']/1e6:.2f} M) | {c4['ms_per_step']:.2f} ms (kernel {r['kernel_ms_avg']:.2f}) | "
            f"per iteration (SURVEY §8d: 408 000 B per problem-iteration, 19.6 M it/s = 100 %): **{100*pi['frac']:.1f} %**; by the bytes the launch really streams "
            f"({r['mfma_secondary']['accumulate_passes_per_launch']:.0f} accumulate + {r['passes_per_launch']-r['mfma_secondary']['accumulate_passes_per_launch']:.0f} evaluate passes; "
            f"{r['builds_from_memo_per_launch']:.0f} Builds come from the memo): {r['achieved']/1e3:.2f} TB/s = {100*r['frac']:.1f} % of peak, {100*r['frac_of_measured_ceiling']:.0f} % of the "
            f"{r['measured_read_ceiling_GBps']/1e3:.2f} TB/s read ceiling; PMC traffic {pm['hbm_bytes_per_launch']/1e9:.2f} GB vs {pm['algorithmic_bytes_per_launch']/1e9:.2f} GB streamed-algorithmic "
            f"({pm['traffic_over_algorithmic']:.3f}: the memo slots and x) | {c4['cpu_baseline']['value']:.0f} it/s (× {c4['value']/c4['cpu_baseline']['value']:.0f}) |")
rows.append(f"| c4, `coop_off` / `memo_off + coop_off` (same call) | {c40['value']/1e6:.2f} M / {c400['value']/1e6:.2f} M | {c40['ms_per_step']:.2f} / {c400['ms_per_step']:.2f} ms | "
            f"{100*c40['roofline']['per_iteration']['frac']:.1f} % / {100*c400['roofline']['per_iteration']['frac']:.1f} % per iteration | — |")
r3 = c3["roofline"]
rows.append(f"| c3 | **{c3['value']/1e6:.1f} M LM it/s** | {c3['ms_per_step']:.3f} ms (kernel {r3['kernel_ms_avg']:.3f}) | {r3['achieved']/1e3:.2f} TB/s = **{100*r3['frac']:.1f} %**; PMC "
            f"{pm3['hbm_bytes_per_launch']/1e9:.3f} GB vs {pm3['algorithmic_bytes_per_launch']/1e9:.3f} GB ({pm3['traffic_over_algorithmic']:.3f}) | {c3['cpu_baseline']['value']/1e3:.1f} k it/s (× {c3['value']/c3['cpu_baseline']['value']:.0f}) |")
rows.append(f"| c2 / c5 / c1 | {c2['value']/1e3:.1f} k / {c5['value']/1e3:.1f} k / {c1['value']/1e3:.0f} k it/s | {c2['ms_per_step']*1e3:.1f} / {c5['ms_per_step']*1e3:.1f} / {c1['ms_per_step']*1e3:.1f} µs | latency | "
            f"{c2['cpu_baseline']['value']/1e3:.1f} k / {c5['cpu_baseline']['value']:.0f} / {c1['cpu_baseline']['value']/1e6:.2f} M it/s |")
rl = l128["roofline"]
rows.append(f"| large128 | **{l128['value']/1e6:.2f} M it/s** (round 2: 0.90) | {l128['ms_per_step']:.2f} ms | MFMA {100*rl['frac']:.1f} % (algorithmic), {100*rl['frac_issued']:.1f} % issued; PMC reads "
            f"{pml['traffic_over_algorithmic']:.3f}× (round 2: 1.067×) | {l128['cpu_baseline']['value']:.0f} it/s |")
r2 = l256["roofline"]
rows.append(f"| large256 (128 × n = 256 × m = 8192 fp32) | **{l256['value']/1e3:.1f} k it/s** (round 2: 39.3 k; `large_library_solver`, same call: {l256r['value']/1e3:.1f} k) | {l256['ms_per_step']:.1f} ms | "
            f"MFMA {100*r2['frac']:.1f} % of peak over the WHOLE batched solve in algorithmic flops (the Gram kernel alone: 104 TFLOP/s issued) | {l256['cpu_baseline']['value']:.0f} it/s |")
rows.append(f"| ba | {ba['value']/1e6:.2f} M it/s | {ba['ms_per_step']:.2f} ms | as round 2 | {ba['cpu_baseline']['value']:.0f} it/s |")
rows.append(f"| balists (4 scenes × 64 cameras × 5000 points × 6 observations per point, fp64) | **{bl['value']:.0f} it/s** (library solver on side streams, same call: {blr['value']:.0f}) | {bl['ms_per_step']:.2f} ms | "
            f"latency; PMC traffic {bl['roofline']['traffic']/1e9 if bl['roofline'].get('traffic') else float('nan'):.2f} GB per solve (calibrated) | {bl['cpu_baseline']['value']:.0f} it/s (dense oracle on a bounded sample) |")
table = "| workload | value | per step | roofline | CPU oracle, 1 core |\n|---|---|---|---|---|\n" + "\n".join(rows) + "\n"
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a, b = "<!-- r03-table -->\n", "<!-- /r03-table -->\n"
i0, i1 = s.index(a) + len(a), s.index(b)
open(p, "w").write(s[:i0] + table + s[i1:])
print(table)
