#!/bin/bash
# usage (on the GPU box): bash tools/pmc_one.sh <tag> <kernel substring> "<counters>" <bench args...>
# one rocprofv3 --pmc pass (no other trace domains besides --kernel-trace), per-kernel counter averages on stdout
R=${GRAFT_REPO_ROOT:-$PWD}; tag=$1; kern=$2; ctr=$3; shift 3
O=$R/gpurun_out/pmc_$tag; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O -- python $R/bench.py "$@" --no-cpu > $O/bench.json 2> $O/err.txt
python - <<PY
import csv, glob, collections
cc = glob.glob("$O/runc/*_counter_collection.csv")[0]
v = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if "$kern" in r["Kernel_Name"]:
        v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, x in v.items():
    print(k, "n", len(x), "max", max(x), "first5", [round(t) for t in x[:5]])
PY
