#!/bin/bash
# Same-box interleaved A/B of variant libraries built by tools/variant_build.sh:  tools/ab.sh <workload c4|c3> <tag> <tag> ...
wl=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do TINYOPT_AMD_LIB=$PWD/tinyopt_amd/_variants/lib_$v.so python bench.py --workload $wl --steps 12 --warmup 4 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', '$v', round(d['value']/1e6,3), round(d['ms_per_step'],4))"; done
done
