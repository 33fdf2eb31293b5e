#!/bin/bash
# round-3 final measurement session (GPU box): everything DESIGN.md section 6 quotes.  bash tools/r03_final.sh
export TMPDIR=/tmp
TAG=r03_final
O=gpurun_out/$TAG
mkdir -p $O
bash tools/profile_round.sh $TAG > $O/profile.log 2>&1; tail -30 $O/profile.log
P=$O/profiles_out
python3 bench.py --config C5 --camera bal --pcg-model-tol 0.5 --no-cpu-baseline > $P/${TAG}_c5_bal_modeltest_bench.json 2>> $O/misc.err
python3 bench.py --config C5 --no-cpu-baseline > $P/${TAG}_c5_bench.json 2>> $O/misc.err
python3 bench.py --config C2 --no-cpu-baseline > $P/${TAG}_c2_bench.json 2>> $O/misc.err
python3 bench.py --config C1 --no-cpu-baseline > $P/${TAG}_c1_bench.json 2>> $O/misc.err
python3 bench.py --config C3x10 --no-cpu-baseline --repeats 3 > $P/${TAG}_c3x10_bench.json 2>> $O/misc.err
BA_RIDERS=0 python3 bench.py --no-cpu-baseline > $P/${TAG}_no_riders_bench.json 2>> $O/misc.err
BA_COMM_FORCE=1 python3 bench.py --no-cpu-baseline > $P/${TAG}_rccl_one_rank_bench.json 2>> $O/misc.err
python3 bench.py --jacobian f32 --no-cpu-baseline > $P/${TAG}_f32_bench.json 2>> $O/misc.err
python3 tools/solve_times.py C2 C3 C5 > $P/${TAG}_solve_times.txt 2>> $O/misc.err
python3 tools/bal_solve_times.py > $P/${TAG}_bal_solve_times.txt 2>> $O/misc.err
python3 tools/window_latency.py > $P/${TAG}_window_latency.txt 2>> $O/misc.err
python3 tools/shard_times.py 1 2 4 8 > $P/${TAG}_shard_times.txt 2>> $O/misc.err
python3 tools/bal_like_times.py 1000 100000 450000 > $P/${TAG}_bal_like_times.txt 2>> $O/misc.err
python3 tools/run_end_to_end.py C3 > $P/${TAG}_end_to_end_c3.txt 2>> $O/misc.err
( for a in "5 500 4" "5 1500 4" "2 100 2" "8 1000 5"; do BA_SMALL_STAMPS=1 python3 tools/small_phases.py $a; done; echo "--- BA_SMALL_MW=0 (one workgroup, k_small_lm)"; for a in "5 500 4" "5 1500 4" "2 100 2" "8 1000 5"; do BA_SMALL_MW=0 BA_SMALL_STAMPS=1 python3 tools/small_phases.py $a; done ) > $P/${TAG}_small_phases.txt 2>&1
for f in $P/${TAG}_*bench.json; do python3 - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["value"], "LM it/s", d["config"]["pcg_iterations_per_lm"], "PCG/LM rmse", d["config"]["final_rmse_px"],
      "|", r["kernel"], r["mean_launch_us"], "us frac", r["frac"], "iter frac", r["lm_iteration"]["frac"])
PY
done
cat $P/${TAG}_shard_times.txt $P/${TAG}_bal_like_times.txt $P/${TAG}_solve_times.txt $P/${TAG}_bal_solve_times.txt $P/${TAG}_window_latency.txt; tail -8 $P/${TAG}_end_to_end_c3.txt
