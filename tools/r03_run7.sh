#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03_i
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout -k 10 120 python bench.py --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-230 $O/bench_c3.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --repeats 3 > $O/bench_trace.json 2> $O/trace.err
python3 tools/trace_summary.py $O/trace/run_kernel_stats.csv 8
rm -f $O/trace/run_kernel_trace.csv
timeout -k 10 120 python bench.py --config C5 --camera bal --no-cpu-baseline > $O/bench_c5_bal.json 2> $O/bench_c5_bal.err; cut -c1-230 $O/bench_c5_bal.json
