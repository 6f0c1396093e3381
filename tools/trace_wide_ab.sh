#!/bin/bash
# GPU box: A/B of one environment switch on the wide step's kernel trace, same box, one process each.
#   tools/trace_wide_ab.sh VAR valueA valueB [shape batch]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
var=$1; sh=${4:-wide}; b=${5:-16}
for v in $2 $3 $2 $3; do
  out=gpurun_out/ab_$v
  rm -rf $out && mkdir -p $out
  env $var=$v true
  export $var=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o t -- python3 bench.py --shape $sh --batch $b --steps 3 --warmup 1 --no-cpu-baseline > $out.log 2>&1 || exit 1
  db=$(find $out -name '*.db' | head -1)
  echo "== $var=$v" >> gpurun_out/ab.txt
  python3 tools/step_trace.py $db 100 | grep -E "panel|sum of" >> gpurun_out/ab.txt
  rm -rf $out
done
