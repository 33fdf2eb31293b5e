#!/bin/bash
# round-4 measurement session (GPU box): everything DESIGN.md sections 6 / 10 quote.  bash tools/r04_final.sh [tag]
export TMPDIR=/tmp
TAG=${1:-r04_final}
O=gpurun_out/$TAG
mkdir -p $O
bash tools/profile_round.sh $TAG > $O/profile.log 2>&1; tail -30 $O/profile.log
P=$O/profiles_out
python3 bench.py --config C5 --camera bal --pcg-model-tol 0.5 --no-cpu-baseline > $P/${TAG}_c5_bal_modeltest_bench.json 2>> $O/misc.err
python3 bench.py --config C5 --no-cpu-baseline > $P/${TAG}_c5_bench.json 2>> $O/misc.err
python3 bench.py --config C2 --no-cpu-baseline > $P/${TAG}_c2_bench.json 2>> $O/misc.err
python3 bench.py --config C1 --no-cpu-baseline > $P/${TAG}_c1_bench.json 2>> $O/misc.err
python3 bench.py --config C3x10 --no-cpu-baseline --repeats 3 > $P/${TAG}_c3x10_bench.json 2>> $O/misc.err
BA_RIDERS=3 python3 bench.py --no-cpu-baseline > $P/${TAG}_no_fused_probe_bench.json 2>> $O/misc.err
python3 bench.py --no-cpu-baseline --precond-lag 0 > $P/${TAG}_no_lag_bench.json 2>> $O/misc.err
BA_RIDERS=3 python3 bench.py --no-cpu-baseline --precond-lag 0 > $P/${TAG}_round3_structure_bench.json 2>> $O/misc.err
BA_COMM_FORCE=1 python3 bench.py --no-cpu-baseline > $P/${TAG}_rccl_one_rank_bench.json 2>> $O/misc.err
BA_COMM_FORCE=1 BA_IPC=1 python3 bench.py --no-cpu-baseline > $P/${TAG}_rccl_one_rank_ipc_bench.json 2>> $O/misc.err
BA_COMM=shm python3 bench.py --gpus 2 --no-cpu-baseline --repeats 5 > $P/${TAG}_two_ranks_one_gpu_shm_bench.json 2>> $O/misc.err
BA_COMM=shm BA_IPC=1 python3 bench.py --gpus 2 --no-cpu-baseline --repeats 5 > $P/${TAG}_two_ranks_one_gpu_shm_ipc_bench.json 2>> $O/misc.err
python3 bench.py --jacobian f32 --no-cpu-baseline > $P/${TAG}_f32_bench.json 2>> $O/misc.err
python3 tools/solve_times.py C2 C3 C5 > $P/${TAG}_solve_times.txt 2>> $O/misc.err
python3 tools/bal_solve_times.py > $P/${TAG}_bal_solve_times.txt 2>> $O/misc.err
python3 tools/c5_policy_matrix.py > $P/${TAG}_c5_policy_matrix.txt 2>> $O/misc.err
python3 tools/window_latency.py > $P/${TAG}_window_latency.txt 2>> $O/misc.err
python3 tools/shard_times.py 1 2 4 8 > $P/${TAG}_shard_times.txt 2>> $O/misc.err
python3 tools/run_end_to_end.py C3 > $P/${TAG}_end_to_end_c3.txt 2>> $O/misc.err
( for m in device host; do echo "--- BA_SETUP=$m"; BA_SETUP=$m BA_TIME_SETUP=1 python3 -c "
from bundle_adjustment_amd import hip_backend
from bundle_adjustment_amd.synthetic import make_config
import time
p = make_config('C3', seed=0)
with hip_backend.Solver(0) as s:
    for i in range(3):
        t = time.perf_counter(); s.set_problem(p, with_params=False); print('ba_set_problem C3: %.3f ms' % ((time.perf_counter() - t) * 1e3))
"; done ) > $P/${TAG}_set_problem_c3.txt 2>&1
for f in $P/${TAG}_*bench.json; do python3 - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["value"], "LM it/s", d["config"]["pcg_iterations_per_lm"], "PCG/LM rmse", d["config"]["final_rmse_px"],
      "|", r["kernel"], r["mean_launch_us"], "us frac", r["frac"], "iter frac", r["lm_iteration"]["frac"])
PY
done
cat $P/${TAG}_shard_times.txt $P/${TAG}_solve_times.txt $P/${TAG}_bal_solve_times.txt $P/${TAG}_window_latency.txt; tail -8 $P/${TAG}_end_to_end_c3.txt; tail -12 $P/${TAG}_set_problem_c3.txt
