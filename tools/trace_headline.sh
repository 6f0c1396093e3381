#!/bin/bash
# GPU box: kernel trace of the headline step (batch 1024, 60 steps): per-kernel averages into gpurun_out/trace_headline_summary.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/trace_headline
rm -rf $out && mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out -o t -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs "$@" > $out.log 2>&1 || exit 1
python3 tools/prof_summary.py $out 280 > gpurun_out/trace_headline_summary.txt
rm -rf $out
