#!/bin/bash
# round-3 GPU session 1: full GPU suite, benches (C3 with / without riders, C5 BAL), BAL kernel stats, x* export
export TMPDIR=/tmp
O=gpurun_out/r03_c
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
python bench.py --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; cat $O/bench_c3.json | cut -c1-400
BA_NO_RIDERS=1 python bench.py --no-cpu-baseline > $O/bench_c3_noriders.json 2> $O/bench_c3_noriders.err; cut -c1-200 $O/bench_c3_noriders.json
python bench.py --config C5 --camera bal --no-cpu-baseline > $O/bench_c5_bal.json 2> $O/bench_c5_bal.err; cut -c1-600 $O/bench_c5_bal.json
python bench.py --config C5 --camera bal --jacobian f32 --no-cpu-baseline > $O/bench_c5_bal_f32.json 2> $O/bench_c5_bal_f32.err; cut -c1-300 $O/bench_c5_bal_f32.json
python tools/bal_solve_times.py > $O/bal_solve_times.txt 2>&1; cat $O/bal_solve_times.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bal -o run -- python3 tools/bal_solve_times.py > /dev/null 2> $O/trace_bal.err
python tools/trace_summary.py $O/trace_bal/run_kernel_stats.csv 24 > $O/bal_kernel_summary.txt 2>&1; cat $O/bal_kernel_summary.txt
python tools/export_converged.py C2 C3 > $O/export.txt 2>&1; cat $O/export.txt
