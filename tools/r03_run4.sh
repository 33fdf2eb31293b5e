#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03_f
mkdir -p $O
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -12 $O/tests.log
show() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["value_min"], d["value_max"], d["config"]["pcg_iterations_per_lm"], d["config"]["final_rmse_px"], d["kernel_profile_us"])
PY
}
python bench.py --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; show $O/bench_c3.json "C3"
python bench.py --no-cpu-baseline --pcg-model-tol 0 > $O/bench_c3_nomodel.json 2> $O/bench_c3_nomodel.err; show $O/bench_c3_nomodel.json "C3 model test off"
python bench.py --config C5 --camera bal --no-cpu-baseline > $O/bench_c5_bal.json 2> $O/bench_c5_bal.err; show $O/bench_c5_bal.json "C5 bal"
python bench.py --config C5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; show $O/bench_c5.json "C5 pinhole"
python tools/bal_solve_times.py > $O/bal_solve_times.txt 2>&1; cat $O/bal_solve_times.txt
python tools/solve_times.py C2 C3 C5 > $O/solve_times.txt 2>&1; cat $O/solve_times.txt
python tools/window_latency.py > $O/window_latency.txt 2>&1; cat $O/window_latency.txt
