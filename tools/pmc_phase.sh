#!/bin/bash
# usage (on the GPU box): bash tools/pmc_phase.sh <tag> <lib or -> <phase acc|eval|fused> <kernel substring> "<ctr group>" ["<ctr group>" ...]
# one rocprofv3 --pmc pass per counter group over tools/prof_phase.py; prints the per-launch maximum of every counter
R=${GRAFT_REPO_ROOT:-$PWD}; tag=$1; lib=$2; phase=$3; kern=$4; shift 4
[ "$lib" != "-" ] && export TINYOPT_AMD_LIB=$lib
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "$@"; do
  O=$R/gpurun_out/pmcph_${tag}_$i; rm -rf $O; mkdir -p $O
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O -- python $R/tools/prof_phase.py $phase c4 > $O/out.txt 2> $O/err.txt
  python - <<PY
import csv, glob, collections
cc = glob.glob("$O/*/*_counter_collection.csv")
v = collections.defaultdict(list)
for f in cc:
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, x in v.items():
    print("$tag", k, "n", len(x), "last", x[-1])
PY
  i=$((i+1))
done
