#!/bin/bash
# Same-box A/B of the load-ring depth (TOA_DEPTH) on the C3 shape; variants built by tools/variant_build.sh d2|d3|d4.
for rep in 1 2 3; do
  for v in d2 d3 d4; do TINYOPT_AMD_LIB=$PWD/tinyopt_amd/_variants/lib_$v.so python bench.py --workload c3 --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', '$v', round(d['value']/1e6,3), round(d['ms_per_step'],4))"; done
done
for v in d2 d3 d4; do TINYOPT_AMD_LIB=$PWD/tinyopt_amd/_variants/lib_$v.so python bench.py --workload c3 --problems 40960 --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 P=40960', '$v', round(d['value']/1e6,3), round(d['ms_per_step'],4))"; done
