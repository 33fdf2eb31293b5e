#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03_d
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
for r in 0 1 2 3; do BA_RIDERS=$r python bench.py --no-cpu-baseline > $O/bench_c3_riders$r.json 2> $O/bench_c3_riders$r.err; python - <<PY
import json
d=json.loads(open("$O/bench_c3_riders$r.json").read().strip().splitlines()[-1])
print("riders $r", d["value"], d["value_min"], d["value_max"], d["kernel_profile_us"])
PY
done
for r in 0 3; do BA_RIDERS=$r python bench.py --config C5 --camera bal --no-cpu-baseline > $O/bench_c5_bal_riders$r.json 2> $O/bench_c5_bal_riders$r.err; python - <<PY
import json
d=json.loads(open("$O/bench_c5_bal_riders$r.json").read().strip().splitlines()[-1])
print("C5 bal riders $r", d["value"], d["config"]["pcg_iterations_per_lm"], d["kernel_profile_us"])
PY
done
