#!/bin/bash
# GPU-box check used during round 2: parity tests + every bench line.  Output under gpurun_out/r2a/.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > $O/pytest_gpu.txt
for wl in c4 c3 c2 c5 c1; do
  python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err
done
tail -5 $O/pytest_gpu.txt
for wl in c4 c3 c2 c5 c1; do python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$wl.json").read())
    c = d["config"]
    print("$wl", round(d["value"]), "it/s", round(d["ms_per_step"], 4), "ms/step frac", round(d["roofline"]["frac"], 4),
          "its/prob", c.get("iters_per_problem", c.get("lm_iterations_per_solve")), "cpu", d.get("cpu_baseline", {}).get("value"),
          "cpu its/prob", d.get("cpu_baseline", {}).get("iters_per_problem"), "us/solve", c.get("us_per_solve_device"))
except Exception as e:
    print("$wl", "FAILED", e, open("$O/bench_$wl.err").read()[-1500:])
PY
done
