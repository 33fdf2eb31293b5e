#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03_j; mkdir -p $O
true

for v in vc8 vc16; do
  lib=bundle_adjustment_amd/libba_hip.so; [ $v = vc16 ] && lib=bundle_adjustment_amd/libba_hip_vc16.so
  export BA_HIP_LIB=$PWD/$lib
  python3 bench.py --config C5 --camera bal --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$v -o t -- python3 bench.py --config C5 --camera bal --no-cpu-baseline --repeats 3 > $O/tr_$v.json 2> $O/tr_$v.err
  f=$(find $O/trace_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:9]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>7s} avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Percentage"]}%')
PY
  find $O/trace_$v -name '*.csv' ! -name '*kernel_stats.csv' -delete; find $O/trace_$v -name '*.db' -delete
  python3 -c "import json;d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]);print(d['value'],d['config'].get('pcg_iterations_per_lm'),d['config'].get('final_rmse_px'))"
done
