#!/bin/bash
# Partial refresh (runs on the GPU box): only the lines, kernel statistics and PMC passes of the n > 128 pipeline, into the SAME
# gpurun_out/refresh/ tree tools/refresh_profiles.sh fills — for a kernel change late in a round that touches nothing else.
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
python bench.py --workload large256 --steps 5 --warmup 2 > $O/bench_large256.json 2> $O/bench_large256.err
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning large_library_solver=1 > $O/bench_large256_rocsolver.json 2>/dev/null
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning large_one_lane=1 > $O/bench_large256_one_lane.json 2>/dev/null
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning large_one_lane=2 > $O/bench_large256_two_lanes.json 2>/dev/null
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning large_gram_plain_deal=1 > $O/bench_large256_plain_deal.json 2>/dev/null
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning memo_off=1 > $O/bench_large256_memo0.json 2>/dev/null
python tools/large_n_bench.py > $O/large_n_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/stats_large256 $O/pmc_large256
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_large256 -- python $R/bench.py --workload large256 --steps 20 --warmup 3 --no-cpu > $O/bench_under_rocprof_large256.json 2> $O/stats_large256.err
for C in "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" FETCH_SIZE; do
  tag=$(echo $C | tr " " "_" | cut -c1-48)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_large256/$tag -- python $R/bench.py --workload large256 --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
done
cd $R; bash tools/pmc_sum.sh large256 "large_" 4 --workload large256 --steps 2 --warmup 1 > $O/pmcsum_large256.txt 2>&1; cp gpurun_out/pmcsum_large256.json $O/
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
