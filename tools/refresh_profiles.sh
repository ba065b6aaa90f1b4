#!/bin/bash
# Runs on the GPU box (via gpurun): tests, every bench line, rocprofv3 kernel stats and the separate --pmc passes that
# profiles/ is built from.  Everything lands under gpurun_out/refresh/; tools/digest_profiles.py <tag> turns it into profiles/.
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O   # NOTE: also delete the LOCAL gpurun_out/refresh before calling gpurun (results are merged, not mirrored)
cd $R
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/pytest_gpu.txt
for wl in c4 c3 c2 c5 c1 ba balists c4_text c4_ad c2_batch_f32 c2_batch; do python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err; done
python tools/row_model_bench.py 12500 > $O/row_model_bench.txt 2>&1
python bench.py --workload c4 --loss huber:0.5 > $O/bench_c4_huber.json 2> $O/bench_c4_huber.err
python bench.py --workload c4 --loss huber:0.5 --no-cpu --tuning narrow_mfma_pass=1 > $O/bench_c4_huber_old_route.json 2>/dev/null
python bench.py --workload c3 --loss huber:0.5 --no-cpu > $O/bench_c3_huber.json 2>/dev/null
python bench.py --workload c3 --loss huber:0.5 --no-cpu --tuning narrow_mfma_pass=1 > $O/bench_c3_huber_old_route.json 2>/dev/null
python bench.py --workload large128 --steps 5 --warmup 2 > $O/bench_large128.json 2> $O/bench_large128.err
python bench.py --workload large256 --steps 5 --warmup 2 > $O/bench_large256.json 2> $O/bench_large256.err
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning large_library_solver=1 > $O/bench_large256_rocsolver.json 2>/dev/null
python bench.py --workload balists --no-cpu --tuning large_library_solver=1 > $O/bench_balists_rocsolver.json 2>/dev/null
python bench.py --workload large128 --steps 5 --warmup 2 --no-cpu --tuning large_row_split=1 > $O/bench_large128_rowsplit.json 2>/dev/null
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning large_one_lane=1 > $O/bench_large256_one_lane.json 2>/dev/null
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning large_one_lane=2 > $O/bench_large256_two_lanes.json 2>/dev/null
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning large_gram_plain_deal=1 > $O/bench_large256_plain_deal.json 2>/dev/null
python bench.py --workload large128 --steps 5 --warmup 2 --no-cpu --tuning memo_off=1 > $O/bench_large128_memo0.json 2>/dev/null
python bench.py --workload large256 --steps 5 --warmup 2 --no-cpu --tuning memo_off=1 > $O/bench_large256_memo0.json 2>/dev/null
python bench.py --workload c4 --no-cpu --tuning coop_chunks=6 > $O/bench_c4_k6.json 2>/dev/null
tools/ubench/llc_probe 20 > $O/llc_probe.txt 2>&1
python bench.py --workload c4 --no-cpu --tuning coop_off=1 > $O/bench_c4_coop0.json 2> $O/bench_c4_coop0.err
python bench.py --workload c4 --no-cpu --tuning memo_off=1,coop_off=1 > $O/bench_c4_memo0_coop0.json 2> $O/bench_c4_memo0_coop0.err
python tools/ad_ratio.py > $O/ad_ratio.txt 2>&1
python tools/large_n_bench.py > $O/large_n_bench.txt 2>&1
python tools/throughput_map.py > $O/throughput_map.txt 2>&1
python tools/lf_balance.py > $O/lf_balance.txt 2>&1
(python tools/se3_batch_probe.py 20000 400 f64; python tools/se3_batch_probe.py 40000 400 f32; python tools/jit_c5.py) > $O/se3_probe.txt 2>&1
(python tools/robust_probe.py; python tools/robust_probe.py 40000 12 500) > $O/robust_probe.txt 2>&1
python tools/k3_crossover.py > $O/k3_crossover.txt 2>&1
(python tools/probe.py c4; python tools/probe.py c3) > $O/probe_phases.txt 2>&1
bash tools/coop_sweep.sh > $O/coop_sweep.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_under_rocprof.json 2> $O/stats.err
for wl in c3 c2 c5 large128 large256 ba balists c4_text c4_ad; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python $R/bench.py --workload $wl --steps 20 --warmup 3 --no-cpu > $O/bench_under_rocprof_$wl.json 2> $O/stats_$wl.err
done
for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_MFMA"; do
  tag=$(echo $C | tr " " "_" | cut -c1-48)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_fused/$tag -- python $R/tools/prof_phase.py fused > /dev/null 2>&1
done
# the C4 shape as text (row models, round 6): traffic and issue mix of the run-time build's fused kernel, calibrated on ITS cost-only seam
for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_MFMA"; do
  tag=$(echo $C | tr " " "_" | cut -c1-48)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_fused_c4_text/$tag -- python $R/tools/prof_phase.py fused c4_text > /dev/null 2>&1
done
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_eval_c4_text/FETCH_SIZE -- python $R/tools/prof_phase.py eval c4_text > /dev/null 2>&1
# FETCH_SIZE calibration on the evaluate seam (reads every packed byte exactly once)
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_eval/FETCH_SIZE -- python $R/tools/prof_phase.py eval > /dev/null 2>&1
# the same for the C3 launch (fp64, n = 12)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_fused_c3/$C -- python $R/tools/prof_phase.py fused c3 > /dev/null 2>&1
done
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_eval_c3/FETCH_SIZE -- python $R/tools/prof_phase.py eval c3 > /dev/null 2>&1
# matrix-core occupancy of the n = 128 kernel
for C in "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" FETCH_SIZE; do
  tag=$(echo $C | tr " " "_" | cut -c1-48)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_large128/$tag -- python $R/bench.py --workload large128 --steps 5 --warmup 1 --no-cpu > /dev/null 2>&1
done
# the hand-written Gram of the n > 128 pipeline
for C in "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" FETCH_SIZE; do
  tag=$(echo $C | tr " " "_" | cut -c1-48)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_large256/$tag -- python $R/bench.py --workload large256 --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
done
# issue mix of the C3 launch
for C in "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA"; do
  tag=$(echo $C | tr " " "_" | cut -c1-48)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_sq_c3/$tag -- python $R/tools/prof_phase.py fused c3 > /dev/null 2>&1
done
# HBM traffic of the bundle-adjustment kernel (its per-scene work arrays do not fit the L2s)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_ba/$C -- python $R/bench.py --workload ba --steps 5 --warmup 1 --no-cpu > /dev/null 2>&1
done
# HBM traffic of the whole BA-lists pipeline (every kernel, per batched solve)
cd $R; bash tools/pmc_sum.sh balists "bl_|rocsolver|rocblas|Cijk|large_" 6 --workload balists --steps 3 --warmup 2 > $O/pmcsum_balists.txt 2>&1; cp gpurun_out/pmcsum_balists.json $O/
# ... and of the whole n > 128 pipeline (rows, Gram, reduce, pre, stage, factorisation, post), per batched solve
bash tools/pmc_sum.sh large256 "large_" 4 --workload large256 --steps 2 --warmup 1 > $O/pmcsum_large256.txt 2>&1; cp gpurun_out/pmcsum_large256.json $O/
find $O -name "*.csv" | wc -l
du -sh $O
