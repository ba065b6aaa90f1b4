#!/bin/bash
# HBM traffic of the C3 launch (fp64, n = 12, 10 000 problems) from PMC counters, same recipe as refresh_profiles.sh:
# separate --pmc passes, FETCH_SIZE calibrated on the evaluate seam of the same shape.  Output: gpurun_out/pmc_c3/.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc_c3; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/fused/$C -- python $R/tools/prof_phase.py fused c3 > /dev/null 2>&1
done
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/eval/FETCH_SIZE -- python $R/tools/prof_phase.py eval c3 > /dev/null 2>&1
find $O -name "*.csv" | wc -l
