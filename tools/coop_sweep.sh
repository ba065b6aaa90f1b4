#!/bin/bash
# same-box sweep of the cooperative-tail chunk count at C4 (two interleaved rounds); prints M it/s and kernel ms
for rep in ${REPS:-1 2}; do
  for k in ${KS:-0 2 4 8}; do
    if [ $k = 0 ]; then export TOA_COOP=0; unset TOA_COOP_K; else export TOA_COOP=1 TOA_COOP_K=$k; fi
    python bench.py --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('rep $rep K=$k', round(d['value']/1e6,3), 'M it/s  kernel', round(d['roofline']['kernel_ms_avg'],3), 'ms')"
  done
done
