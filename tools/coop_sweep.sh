#!/bin/bash
# same-box sweep of the cooperative-tail chunk count at C4 (two interleaved rounds); prints M it/s and kernel ms
for rep in ${REPS:-1 2}; do
  for k in ${KS:-0 2 4 8}; do
    if [ $k = 0 ]; then T="coop_off=1"; else T="coop_chunks=$k"; fi
    python bench.py --no-cpu --tuning $T | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('rep $rep K=$k', round(d['value']/1e6,3), 'M it/s  kernel', round(d['roofline']['kernel_ms_avg'],3), 'ms')"
  done
done
