#!/usr/bin/env python3
"""Turns gpurun_out/refresh (tools/refresh_profiles.sh) into the committed summaries under profiles/.
usage: python tools/digest_profiles.py <round-tag>   e.g. r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def counters(dirpath, kernel_substr):
    cc = glob.glob(os.path.join(dirpath, "runc", "*_counter_collection.csv"))[0]
    kt = glob.glob(os.path.join(dirpath, "runc", "*_kernel_trace.csv"))[0]
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(cc)):
        if kernel_substr in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt))
            if kernel_substr in r["Kernel_Name"]]
    # the first launch of a process pays lazy code loading: use the later ones
    return {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in vals.items()}, durs


def main():
    os.makedirs(DST, exist_ok=True)
    shutil.copy(glob.glob(os.path.join(SRC, "stats", "runc", "*_kernel_stats.csv"))[0], os.path.join(DST, f"{tag}_kernel_stats.csv"))
    for wl in ("c3", "c2", "c5", "large128", "large256", "ba", "balists", "c4_text", "c4_ad"):
        hits = glob.glob(os.path.join(SRC, f"stats_{wl}", "runc", "*_kernel_stats.csv"))
        if hits:
            shutil.copy(hits[0], os.path.join(DST, f"{tag}_kernel_stats_{wl}.csv"))
    # warm-only launch durations of the dominant kernel from the same rocprofv3 run (the *_kernel_stats.csv average includes
    # the warm-up launches, the first of which pays lazy code loading): the timed launches are the LAST `steps` of the run
    for sub, wl_tag in (("stats", "c4"), ("stats_c3", "c3"), ("stats_large128", "large128")):
        tr = glob.glob(os.path.join(SRC, sub, "runc", "*_kernel_trace.csv"))
        bj = os.path.join(SRC, "bench_under_rocprof.json" if wl_tag == "c4" else f"bench_under_rocprof_{wl_tag}.json")
        if not tr or not os.path.exists(bj):
            continue
        line = json.loads([l for l in open(bj).read().splitlines() if l.startswith("{")][-1])
        ksub = "large_fused" if wl_tag == "large128" else "lm_fused_kernel"
        rows = sorted(((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
                       for r in csv.DictReader(open(tr[0])) if ksub in r["Kernel_Name"]))
        durs = [d for _, d in rows]
        warm = durs[-line["steps"]:]
        with open(os.path.join(DST, f"{tag}_kernel_warm_{wl_tag}.json"), "w") as f:
            json.dump({"round": tag, "workload": wl_tag, "kernel": ksub, "source": "rocprofv3 --kernel-trace of `bench.py --no-cpu` (the run behind "
                       f"{tag}_kernel_stats{'' if wl_tag == 'c4' else '_' + wl_tag}.csv)", "launches_in_run": len(durs), "all_ms": durs,
                       "timed_launches": len(warm), "warm_avg_ms": sum(warm) / len(warm),
                       "bench_line_kernel_ms_avg_hip_events": line["roofline"].get("kernel_ms_avg"),
                       "ratio_rocprof_over_hip_events": (sum(warm) / len(warm)) / line["roofline"]["kernel_ms_avg"] if line["roofline"].get("kernel_ms_avg") else None,
                       "bench_line_value": line["value"], "bench_line_frac": line["roofline"]["frac"]}, f, indent=1)
    for txt in ("row_model_bench", "ad_ratio", "large_n_bench", "throughput_map", "lf_balance", "se3_probe", "robust_probe", "k3_crossover", "probe_phases", "coop_sweep", "llc_probe", "pytest_gpu"):
        if os.path.exists(os.path.join(SRC, txt + ".txt")):
            shutil.copy(os.path.join(SRC, txt + ".txt"), os.path.join(DST, f"{tag}_{txt}.txt"))
    if os.path.isdir(os.path.join(SRC, "pmc_large128")):   # counters of the n = 128 workgroup-per-problem kernel
        lf = {}
        for d in sorted(glob.glob(os.path.join(SRC, "pmc_large128", "*"))):
            try:
                c, durs = counters(d, "large_fused")
            except IndexError:
                continue
            lf.update(c)
            lf.setdefault("_kernel_ms", {})[os.path.basename(d)] = durs
        globals()["_large128"] = lf   # finished below, once the FETCH_SIZE calibration is known
    for name in ("bench_c4", "bench_c3", "bench_c2", "bench_c5", "bench_c1", "bench_under_rocprof", "bench_under_rocprof_c3",
                 "bench_under_rocprof_c2", "bench_under_rocprof_c5", "bench_under_rocprof_large128", "bench_large128", "bench_ba", "bench_under_rocprof_ba",
                 "bench_balists", "bench_under_rocprof_balists", "bench_large256", "bench_under_rocprof_large256", "bench_c4_coop0", "bench_c4_memo0_coop0",
                 "bench_large256_rocsolver", "bench_balists_rocsolver", "bench_large128_rowsplit", "bench_large256_one_lane", "bench_large256_two_lanes", "bench_large256_plain_deal",
                 "bench_large128_memo0", "bench_large256_memo0", "bench_c4_k6", "bench_c4_text", "bench_c4_ad", "bench_under_rocprof_c4_text", "bench_under_rocprof_c4_ad",
                 "bench_c4_huber", "bench_c4_huber_old_route", "bench_c3_huber", "bench_c3_huber_old_route", "bench_c2_batch_f32", "bench_c2_batch"):
        if not os.path.exists(os.path.join(SRC, name + ".json")):
            continue
        with open(os.path.join(SRC, name + ".json")) as f:
            line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
        with open(os.path.join(DST, f"{tag}_{name}.json"), "w") as f:
            f.write(line + "\n")
    bench = json.loads(open(os.path.join(DST, f"{tag}_bench_c4.json")).read())
    kern = "lm_fused_kernel"
    fused = {}
    for d in sorted(glob.glob(os.path.join(SRC, "pmc_fused", "*"))):
        c, durs = counters(d, kern)
        fused.update(c)
        fused.setdefault("_kernel_ms", {})[os.path.basename(d)] = durs
    ev, _ = counters(os.path.join(SRC, "pmc_eval", "FETCH_SIZE"), "accumulate_kernel")
    P, bpp = bench["config"]["problems_per_gpu"], bench["roofline"]["algorithmic_bytes_per_pass"]
    known = float(P) * bpp
    cal = known / (ev["FETCH_SIZE"] * 1024.0)      # bytes per reported byte (MI355X_MICROARCH.md: gfx950 reports ~1/2)
    hbm = fused["FETCH_SIZE"] * 1024.0 * cal + fused["WRITE_SIZE"] * 1024.0
    alg = bench["roofline"]["passes_per_launch"] * bpp
    out = {
        "round": tag, "workload": "c4", "problems": P,
        "kernel": "lm_fused_kernel<DenseRowModel<float,3,3,false,true>>",
        "FETCH_SIZE_KB_per_launch": fused["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": fused["WRITE_SIZE"],
        "fetch_calibration": {
            "kernel": "accumulate_kernel<DenseRowModel<float,3,3>> want_grad=0 (reads every packed byte exactly once)",
            "known_bytes": known, "FETCH_SIZE_KB": ev["FETCH_SIZE"], "bytes_per_reported_byte": cal,
            "note": "MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced stream; "
                    "calibrated here on this access pattern (12 B/lane buffer_load_dwordx3)"},
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
        "sq_counters_per_launch": {k: v for k, v in fused.items() if k.startswith("SQ_") or k.startswith("GRBM")},
        "kernel_ms_under_pmc": fused["_kernel_ms"],
    }
    for name in (f"{tag}_pmc.json", "pmc_latest.json"):
        with open(os.path.join(DST, name), "w") as f:
            json.dump(out, f, indent=1)
    # ---- the C4 shape as text (row models): the run-time build's fused kernel, FETCH_SIZE calibrated on its own cost-only seam
    if os.path.isdir(os.path.join(SRC, "pmc_fused_c4_text")) and os.path.exists(os.path.join(DST, f"{tag}_bench_c4_text.json")):
        bt = json.loads(open(os.path.join(DST, f"{tag}_bench_c4_text.json")).read())
        ft = {}
        for d in sorted(glob.glob(os.path.join(SRC, "pmc_fused_c4_text", "*"))):
            try:
                c, durs = counters(d, kern)
            except IndexError:
                continue
            ft.update(c)
            ft.setdefault("_kernel_ms", {})[os.path.basename(d)] = durs
        evt, _ = counters(os.path.join(SRC, "pmc_eval_c4_text", "FETCH_SIZE"), "accumulate_kernel")
        Pt, bppt = bt["config"]["problems_per_gpu"], bt["roofline"]["algorithmic_bytes_per_pass"]
        calt = float(Pt) * bppt / (evt["FETCH_SIZE"] * 1024.0)
        hbmt = ft["FETCH_SIZE"] * 1024.0 * calt + ft["WRITE_SIZE"] * 1024.0
        algt = bt["roofline"]["passes_per_launch"] * bppt
        outt = {"round": tag, "workload": "c4_text", "problems": Pt, "kernel": "lm_fused_kernel<RowModel<float, 3, 3, UserFunctor<float>>> (hiprtc build)",
                "FETCH_SIZE_KB_per_launch": ft["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": ft["WRITE_SIZE"],
                "fetch_calibration": {"kernel": "accumulate_kernel<RowModel<...>> want_grad=0 (LDS-DMA: reads every item byte exactly once)",
                                      "known_bytes": float(Pt) * bppt, "FETCH_SIZE_KB": evt["FETCH_SIZE"], "bytes_per_reported_byte": calt},
                "hbm_bytes_per_launch": hbmt, "algorithmic_bytes_per_launch": algt, "traffic_over_algorithmic": hbmt / algt,
                "sq_counters_per_launch": {k: v for k, v in ft.items() if k.startswith("SQ_") or k.startswith("GRBM")},
                "kernel_ms_under_pmc": ft["_kernel_ms"]}
        for name in (f"{tag}_pmc_c4_text.json", "pmc_latest_c4_text.json"):
            with open(os.path.join(DST, name), "w") as f:
                json.dump(outt, f, indent=1)
        print("c4_text", {k: outt[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic")})
    lf = globals().get("_large128")
    if lf and os.path.exists(os.path.join(DST, f"{tag}_bench_large128.json")):
        bl = json.loads(open(os.path.join(DST, f"{tag}_bench_large128.json")).read())
        hs = bl["roofline"]["hbm_secondary"]
        alg_l = hs["passes_per_launch"] * hs["algorithmic_bytes_per_pass"]
        hbm_l = lf["FETCH_SIZE"] * 1024.0 * cal if "FETCH_SIZE" in lf else None
        outl = {"round": tag, "workload": "large128", "problems": bl["config"]["problems_per_gpu"], "kernel": "large_fused_ts_kernel<float, 8> (tile-split data pass, two workgroups per CU)",
                "counters_per_launch": {k: v for k, v in lf.items() if not k.startswith("_")},
                "fetch_calibration_bytes_per_reported_byte": cal,
                "fetch_calibration_note": "the C4 calibration (wide coalesced streams report 1/2 on gfx950); reads only — the kernel's writes "
                                          "(partial Grams, H) stay in L2 and are not in this figure",
                "hbm_bytes_per_launch": hbm_l, "algorithmic_bytes_per_launch": alg_l,
                "traffic_over_algorithmic": (hbm_l / alg_l) if hbm_l else None,
                "kernel_ms_under_pmc": lf.get("_kernel_ms", {})}
        for name in (f"{tag}_pmc_large128.json", "pmc_latest_large128.json"):
            with open(os.path.join(DST, name), "w") as f:
                json.dump(outl, f, indent=1)
        print("large128", {k: outl[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic")})
    # ---- SQ counters of the hand-written Gram (n > 128) and of the C3 launch: raw per-launch averages
    for sub, kernel_sub, name in (("pmc_large256", "large_gram_kernel", "large256"), ("pmc_sq_c3", kern, "sq_c3")):
        if os.path.isdir(os.path.join(SRC, sub)):
            acc = {}
            for d in sorted(glob.glob(os.path.join(SRC, sub, "*"))):
                try:
                    c, durs = counters(d, kernel_sub)
                except IndexError:
                    continue
                acc.update(c)
                acc.setdefault("_kernel_ms", {})[os.path.basename(d)] = durs[:40]
            with open(os.path.join(DST, f"{tag}_pmc_{name}.json"), "w") as f:
                json.dump({"round": tag, "kernel": kernel_sub, "counters_per_launch_avg": {k: v for k, v in acc.items() if not k.startswith("_")},
                           "fetch_calibration_bytes_per_reported_byte": cal, "kernel_ms_under_pmc": acc.get("_kernel_ms", {})}, f, indent=1)
    # ---- HBM traffic of the BA-lists pipeline (tools/pmc_sum.sh): every kernel, per batched solve
    ps = os.path.join(SRC, "pmcsum_balists.json")
    if os.path.exists(ps) and os.path.exists(os.path.join(DST, f"{tag}_bench_balists.json")):
        d = json.load(open(ps))
        bb = json.loads(open(os.path.join(DST, f"{tag}_bench_balists.json")).read())
        alg_in = bb["roofline"]["algorithmic_bytes_per_pass"] * bb["roofline"]["passes_per_launch"]
        inter = 12.5e6 * bb["roofline"]["passes_per_launch"]
        raw = (d["FETCH_SIZE_KB_per_solve"] + d["WRITE_SIZE_KB_per_solve"]) * 1024.0
        calb = d["FETCH_SIZE_KB_per_solve"] * 1024.0 * cal + d["WRITE_SIZE_KB_per_solve"] * 1024.0
        outp = {"round": tag, "workload": "balists", "problems": bb["config"]["problems_per_gpu"],
                "kernels": "every kernel of the pipeline (bl_*, large_damp / finish, large_chol_solve_kernel), summed per batched solve: tools/pmc_sum.sh",
                "FETCH_SIZE_KB_per_launch": d["FETCH_SIZE_KB_per_solve"], "WRITE_SIZE_KB_per_launch": d["WRITE_SIZE_KB_per_solve"],
                "FETCH_SIZE_KB_top_kernels": d["FETCH_SIZE_KB_per_solve_top"], "WRITE_SIZE_KB_top_kernels": d["WRITE_SIZE_KB_per_solve_top"],
                "fetch_calibration_bytes_per_reported_byte": cal,
                "fetch_calibration_note": "the C4 calibration (a wide coalesced stream reports 1/2 on gfx950); this pipeline's reads are mostly gathers of 64-128 byte records, so the true figure lies between the raw and the calibrated one",
                "hbm_bytes_per_launch_raw": raw, "hbm_bytes_per_launch": calb, "algorithmic_bytes_per_launch": alg_in,
                "algorithmic_note": "the bench line's figure counts the INPUT only (one observation record per observation and pass); the intermediates the pipeline must write and read once (J_c, J_p, r per observation; the reduced system per scene) are ~12.5 MB per scene and pass",
                "traffic_over_algorithmic_input_only": calb / alg_in, "traffic_over_intermediates_estimate": calb / inter,
                "history": {"component-major observation arrays, library solver": {"FETCH_SIZE_KB": 814343.5, "WRITE_SIZE_KB": 218071.6},
                            "record-major observations, library solver on side streams": {"FETCH_SIZE_KB": 542263.0, "WRITE_SIZE_KB": 243112.5}}}
        for name in (f"{tag}_pmc_balists.json", "pmc_latest_balists.json"):
            with open(os.path.join(DST, name), "w") as f:
                json.dump(outp, f, indent=1)
    # ---- HBM traffic of the whole n > 128 pipeline (tools/pmc_sum.sh): every large_* kernel, per batched solve
    ps = os.path.join(SRC, "pmcsum_large256.json")
    if os.path.exists(ps) and os.path.exists(os.path.join(DST, f"{tag}_bench_large256.json")):
        d = json.load(open(ps))
        bb = json.loads(open(os.path.join(DST, f"{tag}_bench_large256.json")).read())
        rb = bb["roofline"].get("hbm_secondary", bb["roofline"])   # (the n > 63 lines are priced against the MFMA roof; bytes ride along)
        alg = rb["algorithmic_bytes_per_pass"] * rb["passes_per_launch"]
        raw = (d["FETCH_SIZE_KB_per_solve"] + d["WRITE_SIZE_KB_per_solve"]) * 1024.0
        calb = d["FETCH_SIZE_KB_per_solve"] * 1024.0 * cal + d["WRITE_SIZE_KB_per_solve"] * 1024.0
        outp = {"round": tag, "workload": "large256", "problems": bb["config"]["problems_per_gpu"],
                "kernels": "every kernel of the pipeline (large_rows_vec / gram / gram_reduce / pre / stage / chol_solve / post), summed per batched solve: tools/pmc_sum.sh",
                "FETCH_SIZE_KB_per_launch": d["FETCH_SIZE_KB_per_solve"], "WRITE_SIZE_KB_per_launch": d["WRITE_SIZE_KB_per_solve"],
                "FETCH_SIZE_KB_top_kernels": d["FETCH_SIZE_KB_per_solve_top"], "WRITE_SIZE_KB_top_kernels": d["WRITE_SIZE_KB_per_solve_top"],
                "fetch_calibration_bytes_per_reported_byte": cal,
                "hbm_bytes_per_launch_raw": raw, "hbm_bytes_per_launch": calb, "algorithmic_bytes_per_launch": alg,
                "algorithmic_note": "m (n + 1) sizeof(T) per pass that streamed the rows; an accumulate pass reads A TWICE by design (rows kernel: r and the row scales; Gram kernel: J^T J) and the solve works on n x n matrices in L2 / HBM",
                "traffic_over_algorithmic": calb / alg}
        for name in (f"{tag}_pmcsum_large256.json", "pmc_latest_large256.json"):
            with open(os.path.join(DST, name), "w") as f:
                json.dump(outp, f, indent=1)
        print("large256 pipeline", {k: outp[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic")})
    # ---- HBM traffic of the bundle-adjustment kernel (work arrays included: they do not fit the L2s)
    if os.path.isdir(os.path.join(SRC, "pmc_ba")) and os.path.exists(os.path.join(DST, f"{tag}_bench_ba.json")):
        bb = json.loads(open(os.path.join(DST, f"{tag}_bench_ba.json")).read())
        ba = {}
        for d in sorted(glob.glob(os.path.join(SRC, "pmc_ba", "*"))):
            c, durs = counters(d, "ba_schur_kernel")
            ba.update(c)
            ba.setdefault("_kernel_ms", {})[os.path.basename(d)] = durs
        rb = bb["roofline"]
        alg_b = rb["passes_per_launch"] * rb["algorithmic_bytes_per_pass"]
        raw = ba["FETCH_SIZE"] * 1024.0 + ba["WRITE_SIZE"] * 1024.0
        hbm_b = ba["FETCH_SIZE"] * 1024.0 * cal + ba["WRITE_SIZE"] * 1024.0
        ms = sum(ba["_kernel_ms"]["FETCH_SIZE"][1:]) / max(1, len(ba["_kernel_ms"]["FETCH_SIZE"]) - 1)
        outb = {"round": tag, "workload": "ba", "scenes": bb["config"]["problems_per_gpu"], "kernel": "ba_schur_kernel<double, 3, 1>",
                "FETCH_SIZE_KB_per_launch": ba["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": ba["WRITE_SIZE"],
                "fetch_calibration_bytes_per_reported_byte": cal,
                "fetch_calibration_note": "the C4 calibration (a wide coalesced stream reports 1/2 on gfx950); this kernel's reads are a mix "
                                          "of coalesced rows and gathers, so the true figure lies between the raw and the calibrated one",
                "hbm_bytes_per_launch_raw": raw, "hbm_bytes_per_launch": hbm_b, "algorithmic_bytes_per_launch": alg_b,
                "traffic_over_algorithmic": hbm_b / alg_b, "kernel_ms_under_pmc": ba["_kernel_ms"],
                "hbm_GBps_raw": raw / (ms * 1e-3) / 1e9, "hbm_GBps": hbm_b / (ms * 1e-3) / 1e9}
        for name in (f"{tag}_pmc_ba.json", "pmc_latest_ba.json"):
            with open(os.path.join(DST, name), "w") as f:
                json.dump(outb, f, indent=1)
        print("ba", {k: outb[k] for k in ("hbm_bytes_per_launch_raw", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "hbm_GBps_raw", "hbm_GBps")})
    # ---- the same HBM-traffic measurement for the C3 launch (fp64, n = 12)
    if os.path.isdir(os.path.join(SRC, "pmc_fused_c3")):
        b3 = json.loads(open(os.path.join(DST, f"{tag}_bench_c3.json")).read())
        f3 = {}
        for d in sorted(glob.glob(os.path.join(SRC, "pmc_fused_c3", "*"))):
            c, durs = counters(d, kern)
            f3.update(c)
            f3.setdefault("_kernel_ms", {})[os.path.basename(d)] = durs
        e3, _ = counters(os.path.join(SRC, "pmc_eval_c3", "FETCH_SIZE"), "accumulate_kernel")
        P3, bpp3 = b3["config"]["problems_per_gpu"], b3["roofline"]["algorithmic_bytes_per_pass"]
        cal3 = float(P3) * bpp3 / (e3["FETCH_SIZE"] * 1024.0)
        hbm3 = f3["FETCH_SIZE"] * 1024.0 * cal3 + f3["WRITE_SIZE"] * 1024.0
        alg3 = b3["roofline"]["passes_per_launch"] * bpp3
        out3 = {"round": tag, "workload": "c3", "problems": P3, "kernel": "lm_fused_kernel<DenseRowModel<double,1,0,false,true>>",
                "FETCH_SIZE_KB_per_launch": f3["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": f3["WRITE_SIZE"],
                "fetch_calibration": {"kernel": "accumulate_kernel<DenseRowModel<double,1,0>> want_grad=0", "known_bytes": float(P3) * bpp3,
                                      "FETCH_SIZE_KB": e3["FETCH_SIZE"], "bytes_per_reported_byte": cal3},
                "hbm_bytes_per_launch": hbm3, "algorithmic_bytes_per_launch": alg3, "traffic_over_algorithmic": hbm3 / alg3,
                "kernel_ms_under_pmc": f3["_kernel_ms"]}
        for name in (f"{tag}_pmc_c3.json", "pmc_latest_c3.json"):
            with open(os.path.join(DST, name), "w") as f:
                json.dump(out3, f, indent=1)
        print("c3", json.dumps({k: out3[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic")}))
    print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic")}))
    print("calibration", cal, "bench value", bench["value"], "frac", bench["roofline"]["frac"])


if __name__ == "__main__":
    main()
