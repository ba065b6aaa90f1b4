#!/bin/bash
# Partial refresh (runs on the GPU box): only the BA-lists lines, kernel statistics and the pipeline's PMC sum, into the SAME
# gpurun_out/refresh/ tree tools/refresh_profiles.sh fills — for a change late in a round that touches nothing else.
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
python bench.py --workload balists > $O/bench_balists.json 2> $O/bench_balists.err
python bench.py --workload balists --no-cpu --tuning large_library_solver=1 > $O/bench_balists_rocsolver.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $O/stats_balists
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_balists -- python $R/bench.py --workload balists --steps 20 --warmup 3 --no-cpu > $O/bench_under_rocprof_balists.json 2> $O/stats_balists.err
cd $R; bash tools/pmc_sum.sh balists "bl_|rocsolver|rocblas|Cijk|large_" 6 --workload balists --steps 3 --warmup 2 > $O/pmcsum_balists.txt 2>&1; cp gpurun_out/pmcsum_balists.json $O/
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
