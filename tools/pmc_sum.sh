#!/bin/bash
# usage (on the GPU box): bash tools/pmc_sum.sh <tag> "<kernel regex>" <solves in the run> <bench args...>
# FETCH_SIZE and WRITE_SIZE (two separate --pmc passes) summed over EVERY kernel of a multi-kernel pipeline that matches the
# regex, per solve -> gpurun_out/pmcsum_<tag>.json
R=${GRAFT_REPO_ROOT:-$PWD}; tag=$1; rx=$2; solves=$3; shift 3
O=$R/gpurun_out/pmcsum_$tag; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -- python $R/bench.py "$@" --no-cpu > $O/bench_$C.json 2> $O/err_$C.txt
done
python - <<PY
import csv, glob, json, re
out = {"kernel_regex": "$rx", "solves_in_run": $solves}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    cc = glob.glob("$O/%s/runc/*_counter_collection.csv" % C)[0]
    tot, per = 0.0, {}
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] == C and re.search(r"$rx", r["Kernel_Name"]):
            v = float(r["Counter_Value"]); tot += v
            k = r["Kernel_Name"][:60]; per[k] = per.get(k, 0.0) + v
    out[C + "_KB_per_solve"] = tot / $solves
    out[C + "_KB_per_solve_top"] = {k: v / $solves for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:8]}
json.dump(out, open("$R/gpurun_out/pmcsum_$tag.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.endswith("_top")}))
PY
