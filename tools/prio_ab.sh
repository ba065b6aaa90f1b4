for rep in 1 2 3; do
  for v in p0 p1 p2; do TINYOPT_AMD_LIB=$PWD/tinyopt_amd/_variants/lib_$v.so python bench.py --steps 10 --warmup 4 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4', '$v', round(d['value']/1e6,3), round(d['ms_per_step'],3))"; done
done
for rep in 1 2 3; do
  for v in q0 q1 q2; do TINYOPT_AMD_LIB=$PWD/tinyopt_amd/_variants/lib_$v.so python bench.py --workload c3 --steps 10 --warmup 4 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', '$v', round(d['value']/1e6,3), round(d['ms_per_step'],3))"; done
done
for v in p0 p1 p2; do TINYOPT_AMD_LIB=$PWD/tinyopt_amd/_variants/lib_$v.so python bench.py --problems 49152 --steps 4 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 P=49152', '$v', round(d['value']/1e6,3), round(d['ms_per_step'],3))"; done
