#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer run of the host-side native code (csrc/mapwalk.c) -- CPU only (GPU
# sanitizers are not available on this pool).  Builds an instrumented copy of the extension into a scratch directory
# and runs the CPU tests that exercise it (window walk, cache, write-back) with the instrumented module first on the path.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(mktemp -d /tmp/mapwalk_asan.XXXX)
mkdir -p $OUT/bundle_adjustment_amd
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
INC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
gcc -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fPIC -shared -Wall -I "$INC" \
    -o $OUT/_mapwalk$EXT $ROOT/bundle_adjustment_amd/csrc/mapwalk.c
trap 'rm -rf '"$OUT" EXIT
cat > $OUT/run.py <<PY
import importlib.util, sys, os
sys.path.insert(0, "$ROOT")
spec = importlib.util.spec_from_file_location("bundle_adjustment_amd._mapwalk", "$OUT/_mapwalk$EXT")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
import bundle_adjustment_amd
sys.modules["bundle_adjustment_amd._mapwalk"] = mod
bundle_adjustment_amd._mapwalk = mod
assert "asan" in open("/proc/self/maps").read()          # the instrumented runtime is the one loaded
import pytest
sys.exit(pytest.main(["-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                      "$ROOT/tests/test_window_reuse.py", "$ROOT/tests/test_bundle_adjuster_host.py"]))
PY
ASAN=$(gcc -print-file-name=libasan.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python3 $OUT/run.py
echo "mapwalk.c: AddressSanitizer + UBSan clean"
