set -x
python -m pytest tests/test_gpu_large_n.py -x -q -k "seam or narrow or loss" 2>&1 | tail -8
python bench.py --workload large128 --steps 5 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r.get('one_problem_per_slot'), r.get('balanced_batch'))"
