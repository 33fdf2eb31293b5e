#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03_g
mkdir -p $O
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -8 $O/tests.log
bash tools/profile_round.sh r03_mid > $O/profile.log 2>&1; tail -40 $O/profile.log
python tools/bal_solve_times.py > $O/bal_solve_times.txt 2>&1; cat $O/bal_solve_times.txt
python tools/window_latency.py > $O/window_latency.txt 2>&1; cat $O/window_latency.txt
