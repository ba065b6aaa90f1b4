#!/bin/bash
# C3: resident workgroups against the tail of the launch (toa_tuning::max_workgroups caps the grid), same call
for rep in 1 2; do
  for g in ${GS:-256 320 384 448 512 576 640 768}; do
    python bench.py --workload c3 --no-cpu --tuning max_workgroups=$g | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); r=d['roofline']; print('rep $rep grid=$g', round(d['value']/1e6,2), 'M it/s  kernel', round(r['kernel_ms_avg'],4), 'ms  frac', round(r['frac'],4))"
  done
done
