#!/bin/bash
# GPU box: the wide step's panel launches at several rows-per-workgroup settings (VT_PANEL_ROWS)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for r in 128 256; do
  out=gpurun_out/trace_wide_r$r
  rm -rf $out && mkdir -p $out
  VT_PANEL_ROWS=$r timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out -o t -- python3 bench.py --shape wide --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > $out.log 2>&1 || exit 1
  db=$(find $out -name '*.db' | head -1)
  python3 tools/step_trace.py $db 15 | grep -E "panel|sum of" > gpurun_out/trace_wide_r$r.txt
  rm -rf $out
done
