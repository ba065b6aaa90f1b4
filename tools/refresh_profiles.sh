#!/bin/bash
# Runs on the GPU box (via gpurun): tests, bench lines, rocprofv3 kernel stats and the separate --pmc passes
# that profiles/ is built from.  Everything lands under gpurun_out/refresh/.
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O   # NOTE: also delete the LOCAL gpurun_out/refresh before calling gpurun (results are merged, not mirrored)
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python bench.py > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --workload c3 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --workload large128 --steps 5 --warmup 2 > $O/bench_large128.json 2> $O/bench_large128.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > $O/bench_under_rocprof.json 2> $O/stats.err
for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_MFMA"; do
  tag=$(echo $C | tr " " "_" | cut -c1-48)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_fused/$tag -- python $R/tools/prof_phase.py fused > /dev/null 2>&1
done
# FETCH_SIZE calibration on the evaluate seam (reads every packed byte exactly once)
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_eval/FETCH_SIZE -- python $R/tools/prof_phase.py eval > /dev/null 2>&1
find $O -name "*.csv" | head -50 > $O/files.txt
du -sh $O
