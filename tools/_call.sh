set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/stats_large128 $O/pmc_large128
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_large128 -- python $R/bench.py --workload large128 --steps 20 --warmup 3 --no-cpu > $O/bench_under_rocprof_large128.json 2> $O/stats_large128.err
for C in "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" FETCH_SIZE; do
  tag=$(echo $C | tr " " "_" | cut -c1-48)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_large128/$tag -- python $R/bench.py --workload large128 --steps 5 --warmup 1 --no-cpu > /dev/null 2>&1
done
ls $O/pmc_large128
