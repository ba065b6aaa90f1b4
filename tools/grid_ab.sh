# A/B of the fused kernel's grid size on one box, interleaved: every slot the chip has (max_workgroups=0: no cap) vs caps
for rep in 1 2 3; do
  for cfg in "max_workgroups=0" "max_workgroups=625" "max_workgroups=576" "max_workgroups=512"; do
    python bench.py --tuning $cfg --steps 10 --warmup 4 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4', '$cfg', round(d['value']/1e6,3), round(d['ms_per_step'],3))"
  done
done
for rep in 1 2; do
  for cfg in "max_workgroups=0" "max_workgroups=834" "max_workgroups=625" "max_workgroups=512"; do
    python bench.py --tuning $cfg --workload c3 --steps 10 --warmup 4 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', '$cfg', round(d['value']/1e6,3), round(d['ms_per_step'],3))"
  done
done
