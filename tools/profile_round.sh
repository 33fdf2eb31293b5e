#!/bin/bash
# Profiles of the headline workload on the GPU box, summarised into profiles/<tag>_*:
#   tools/profile_round.sh r03_final        (run through gpurun from the repository root)
# Separate rocprofv3 runs, as MI355X_MICROARCH.md prescribes: --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE;
# two SQ-counter passes.  Every pass runs the program itself behind "--" (python3 bench.py ...).
# Then the same --kernel-trace --stats for BASELINE config 5 as stated (bench.py --config C5 --camera bal).
set -e
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT profiles
export TMPDIR=/tmp
ARGS="bench.py --steps 20 --warmup 3 --no-cpu-baseline --repeats 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 $ARGS > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 $ARGS > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq1 -o run -- python3 $ARGS > /dev/null 2> $OUT/sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -o run -- python3 $ARGS > /dev/null 2> $OUT/sq2.err
cp $OUT/trace/run_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
python3 tools/pmc_summary.py $OUT/fetch/run_counter_collection.csv $OUT/write/run_counter_collection.csv $TAG
python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err       # (reads the traffic.json just written)
python3 tools/sq_summary.py $OUT/sq1/run_counter_collection.csv $OUT/sq2/run_counter_collection.csv > profiles/${TAG}_c3_sq_counters.txt
cp $OUT/bench.json profiles/${TAG}_bench.json
python3 tools/trace_summary.py profiles/${TAG}_kernel_stats.csv 16
cat profiles/traffic.json
# ---- config 5 as stated: BAL camera
BARGS="bench.py --config C5 --camera bal --steps 20 --warmup 3 --no-cpu-baseline --repeats 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bal -o run -- python3 $BARGS > $OUT/bench_bal_trace.json 2> $OUT/trace_bal.err
cp $OUT/trace_bal/run_kernel_stats.csv profiles/${TAG}_c5_bal_kernel_stats.csv
python3 bench.py --config C5 --camera bal --steps 20 --warmup 3 > profiles/${TAG}_c5_bal_bench.json 2> $OUT/bench_bal.err
python3 bench.py --config C5 --camera bal --jacobian f32 --steps 20 --warmup 3 --no-cpu-baseline > profiles/${TAG}_c5_bal_f32_bench.json 2> $OUT/bench_bal_f32.err
python3 tools/trace_summary.py profiles/${TAG}_c5_bal_kernel_stats.csv 12
# what was written under profiles/ on the box travels back inside gpurun_out/ (only that directory is merged)
mkdir -p $OUT/profiles_out && cp profiles/${TAG}_* profiles/traffic.json $OUT/profiles_out/
# the traces themselves stay on the box (kernel_trace.csv is tens of MB)
rm -f $OUT/*/run_kernel_trace.csv $OUT/*/run_counter_collection.csv
