#!/bin/bash
# price the quantisation of the C3 launch — 10 000 problems on 3 072 resident waves are 3.26 rounds, every problem
# takes exactly 4 iterations, so the last round runs at 26 % occupancy.  Same call, interleaved: P = 3 072 k and the BASELINE's 10 000.
for rep in ${REPS:-1 2}; do
  for P in ${PS:-3072 6144 9216 10000 12288 15360 18432 24576 40960}; do
    python bench.py --workload c3 --no-cpu --problems $P | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); r=d['roofline']; print('rep $rep P=$P', round(d['value']/1e6,2), 'M it/s  kernel', round(r['kernel_ms_avg'],4), 'ms  frac', round(r['frac'],4), ' ns/problem', round(r['kernel_ms_avg']*1e6/$P,2))"
  done
done
