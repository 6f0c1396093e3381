#!/bin/bash
# GPU box: kernel trace of the general-shape step at one configuration.  tools/trace_shape.sh <shape> <batch> [bench flags]
# Output: gpurun_out/trace_<shape>/ (rocprofv3 database) and gpurun_out/trace_<shape>.txt (kernels of one step in order).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
sh=$1; b=$2; shift 2
out=gpurun_out/trace_$sh
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o t -- python3 bench.py --shape $sh --batch $b --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out.log 2>&1 || exit 1
db=$(find $out -name '*.db' | head -1)
python3 tools/step_trace.py $db 15 > gpurun_out/trace_$sh.txt && python3 tools/prof_summary.py $out 4 > gpurun_out/trace_${sh}_summary.txt
