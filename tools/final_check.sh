#!/bin/bash
# GPU box: the whole GPU test suite, the default bench line, and the kernel trace of the same command.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/final
rm -rf $out && mkdir -p $out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/tests.log 2>&1
rc=$?; tail -3 $out/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -5 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
timeout -k 10 300 python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 bench.py --no-cpu-baseline --no-extra-configs --steps 200 --warmup 20 > $out/trace.log 2>&1 || exit 1
python3 -c "
import json; d=json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
