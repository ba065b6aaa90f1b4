#!/bin/bash
# usage (on the GPU box): bash tools/prof_one.sh <tag> <bench args...>   ->  gpurun_out/prof_<tag>/ + top kernels on stdout
R=${GRAFT_REPO_ROOT:-$PWD}; tag=$1; shift
O=$R/gpurun_out/prof_$tag; rm -rf $O; mkdir -p $O   # (delete the LOCAL copy too before calling gpurun: results are merged)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py "$@" --no-cpu > $O/bench.json 2> $O/err.txt
python - <<PY
import csv, glob
f = glob.glob("$O/runc/*_kernel_stats.csv")[0]
for i, r in enumerate(csv.DictReader(open(f))):
    if i < 14: print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
PY
tail -1 $O/bench.json | cut -c1-200
