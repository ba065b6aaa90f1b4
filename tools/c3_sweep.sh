#!/bin/bash
# same-box sweep of the cooperative tail at C3 (fp64, n = 12, m = 500): off and chunk counts, interleaved
for rep in ${REPS:-1 2}; do
  for k in ${KS:-0 2 3 4}; do
    if [ $k = 0 ]; then T="coop_off=1"; else T="coop_chunks=$k"; fi
    python bench.py --workload c3 --no-cpu --tuning $T | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('rep $rep K=$k', round(d['value']/1e6,2), 'M it/s  kernel', round(d['roofline']['kernel_ms_avg'],4), 'ms frac', round(d['roofline']['frac'],4))"
  done
done
