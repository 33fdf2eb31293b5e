#!/bin/bash
O=gpurun_out/r03_m; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_bal.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() { # label, env...
  lab=$1; shift
  for cam in pinhole bal; do
    env "$@" BA_TIME_SETUP=1 python3 bench.py --config C5 --camera $cam --no-cpu-baseline --repeats 5 > $O/$lab.$cam.json 2> $O/$lab.$cam.err
    python3 - $O/$lab.$cam.json "$lab $cam" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], d['value'], 'LM it/s', r['kernel'], r['mean_launch_us'], 'us')
PY
  grep "point passes" $O/$lab.$cam.err | head -1
  done
}
run auto A=1
run old BA_LONG_SLOTS=1
