#!/bin/bash
# GPU box: parity tests, then the bench line (launcher breakdown on stderr-free stdout).
set -o pipefail
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
r = json.load(open("gpurun_out/bench.json"))
print("img/s %.0f  ms/step %.4f  roofline %s" % (r["value"], r["ms_per_step"], json.dumps(r["roofline"])))
print(" ".join("%s=%.1f" % kv for kv in r["launcher_us"].items()))
PY
