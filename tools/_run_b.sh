mkdir -p gpurun_out; R=$PWD; O=$R/gpurun_out/chol_ab.txt; : > $O
for rep in 1 2; do for v in 0 1 2 3; do
export TINYOPT_AMD_LIB=$R/tinyopt_amd/_variants/lib_v$v.so
python bench.py --workload large256 --steps 8 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v$v large256', round(d['ms_per_step'],3), round(d['value']))" >> $O
python bench.py --workload balists --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v$v balists', round(d['ms_per_step'],3), round(d['value']))" >> $O
done; done
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3; do for wl in large256 balists; do
export TINYOPT_AMD_LIB=$R/tinyopt_amd/_variants/lib_v$v.so
rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu > /dev/null 2>&1
f=$(find /tmp/st -name "*kernel_stats.csv" | head -1)
python - "$f" "v$v $wl" >> $O <<'PY'
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:]:
    if 'chol_solve' in r[0]: print(sys.argv[2], 'chol kernel avg us', round(float(r[3])/1e3,1), 'calls', r[1])
PY
done; done
cat $O
