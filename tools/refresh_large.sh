#!/bin/bash
# targeted refresh of the evidence the n > 128 / BA-lists solver touches (run through gpurun): same outputs, same names as
# tools/refresh_profiles.sh leaves under gpurun_out/refresh, so that tools/digest_profiles.py picks them up
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/refresh; mkdir -p $O; cd $R
for wl in large256 balists; do python bench.py --workload $wl $( [ $wl = large256 ] && echo "--steps 5 --warmup 2" ) > $O/bench_$wl.json 2> $O/bench_$wl.err; done
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning large_library_solver=1 > $O/bench_large256_rocsolver.json 2>/dev/null
python bench.py --workload balists --no-cpu --tuning large_library_solver=1 > $O/bench_balists_rocsolver.json 2>/dev/null
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning large_one_lane=1 > $O/bench_large256_one_lane.json 2>/dev/null
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning large_one_lane=2 > $O/bench_large256_two_lanes.json 2>/dev/null
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning large_gram_plain_deal=1 > $O/bench_large256_plain_deal.json 2>/dev/null
python tools/large_n_bench.py > $O/large_n_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for wl in large256 balists; do
  rm -rf $O/stats_$wl
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python $R/bench.py --workload $wl --steps 20 --warmup 3 --no-cpu > $O/bench_under_rocprof_$wl.json 2> $O/stats_$wl.err
done
cd $R; bash tools/pmc_sum.sh balists "bl_|rocsolver|rocblas|Cijk|large_" 6 --workload balists --steps 3 --warmup 2 > $O/pmcsum_balists.txt 2>&1; cp gpurun_out/pmcsum_balists.json $O/
bash tools/pmc_sum.sh large256 "large_" 4 --workload large256 --steps 2 --warmup 1 > $O/pmcsum_large256.txt 2>&1; cp gpurun_out/pmcsum_large256.json $O/
