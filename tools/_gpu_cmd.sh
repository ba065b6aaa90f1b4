cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2j
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r2j/pytest_gpu.txt
python tools/probe.py c4 2>&1 | tail -2 > gpurun_out/r2j/probe_c4.txt
for wl in c4 c3; do python bench.py --workload $wl --no-cpu > gpurun_out/r2j/bench_$wl.json 2> gpurun_out/r2j/bench_$wl.err; done
tail -15 gpurun_out/r2j/pytest_gpu.txt; cat gpurun_out/r2j/probe_c4.txt
for wl in c4 c3; do python -c "
import json; d=json.load(open('gpurun_out/r2j/bench_$wl.json')); print('$wl', round(d['value']/1e6,3),'M it/s', round(d['ms_per_step'],4),'ms/step kernel', round(d['roofline']['kernel_ms_avg'],4), 'frac', round(d['roofline']['frac'],4))"; done
