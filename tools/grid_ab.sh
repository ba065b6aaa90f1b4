# A/B of the fused kernel's grid size on one box, interleaved: every slot the chip has (TOA_X=1: no cap) vs TOA_MAX_WGS caps
for rep in 1 2 3; do
  for cfg in "TOA_X=1" "TOA_MAX_WGS=625" "TOA_MAX_WGS=576" "TOA_MAX_WGS=512"; do
    env $cfg python bench.py --steps 10 --warmup 4 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4', '$cfg', round(d['value']/1e6,3), round(d['ms_per_step'],3))"
  done
done
for rep in 1 2; do
  for cfg in "TOA_X=1" "TOA_MAX_WGS=834" "TOA_MAX_WGS=625" "TOA_MAX_WGS=512"; do
    env $cfg python bench.py --workload c3 --steps 10 --warmup 4 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', '$cfg', round(d['value']/1e6,3), round(d['ms_per_step'],3))"
  done
done
