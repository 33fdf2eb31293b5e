#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03_e
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
show() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["value_min"], d["value_max"], d["config"]["pcg_iterations_per_lm"], d["kernel_profile_us"])
PY
}
for r in 0 3; do BA_RIDERS=$r python bench.py --no-cpu-baseline > $O/bench_c3_riders$r.json 2> $O/bench_c3_riders$r.err; show $O/bench_c3_riders$r.json "C3 riders $r"; done
BA_HIP_LIB=$PWD/bundle_adjustment_amd/libba_hip_vc8.so python bench.py --no-cpu-baseline > $O/bench_c3_vc8.json 2> $O/bench_c3_vc8.err; show $O/bench_c3_vc8.json "C3 VEC_CAMS=8"
python bench.py --config C5 --camera bal --no-cpu-baseline > $O/bench_c5_bal.json 2> $O/bench_c5_bal.err; show $O/bench_c5_bal.json "C5 bal"
BA_HIP_LIB=$PWD/bundle_adjustment_amd/libba_hip_vc8.so python bench.py --config C5 --camera bal --no-cpu-baseline > $O/bench_c5_bal_vc8.json 2> $O/bench_c5_bal_vc8.err; show $O/bench_c5_bal_vc8.json "C5 bal VEC_CAMS=8"
for sg in 16 32; do BA_CAM_SEGL=$sg python bench.py --config C5 --camera bal --no-cpu-baseline > $O/bench_c5_bal_segl$sg.json 2> $O/bench_c5_bal_segl$sg.err; show $O/bench_c5_bal_segl$sg.json "C5 bal SEGL $sg"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -o run -- python3 tools/solve_times.py C5 > $O/solve_times_c5.txt 2> $O/trace_c5.err
cat $O/solve_times_c5.txt
python tools/trace_summary.py $O/trace_c5/run_kernel_stats.csv 40 > $O/c5_kernel_summary.txt 2>&1; grep -i "coarse\|fill" $O/c5_kernel_summary.txt
python tools/window_latency.py > $O/window_latency.txt 2>&1; cat $O/window_latency.txt
