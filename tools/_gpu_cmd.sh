cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2f
python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r2f/pytest_gpu.txt
for wl in c4 c3; do python bench.py --workload $wl --no-cpu > gpurun_out/r2f/bench_$wl.json 2> gpurun_out/r2f/bench_$wl.err; done
tail -30 gpurun_out/r2f/pytest_gpu.txt
for wl in c4 c3; do python -c "
import json; d=json.load(open('gpurun_out/r2f/bench_$wl.json')); print('$wl', round(d['value']/1e6,3),'M it/s', round(d['ms_per_step'],4),'ms/step kernel', round(d['roofline']['kernel_ms_avg'],4), 'frac', round(d['roofline']['frac'],4))"; done
