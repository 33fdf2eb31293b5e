#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03_h
mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -5 $O/tests.log
python tools/window_latency.py > $O/window_latency.txt 2>&1; cat $O/window_latency.txt
python tools/run_end_to_end.py > $O/end_to_end.txt 2>&1; tail -12 $O/end_to_end.txt
python bench.py --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-260 $O/bench_c3.json
