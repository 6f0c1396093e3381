#!/bin/bash
# GPU box: kernel trace and the two HBM-traffic counter passes of the general-shape step at the two BASELINE
# configurations (scaled batch 1024, wide batch 16).  Outputs under gpurun_out/shapes/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/shapes
rm -rf $out && mkdir -p $out
# a line a minute while the counter passes run (they print nothing until they end)
( while sleep 50; do date >> $out/heartbeat.txt; done ) &
hb=$!
trap 'kill $hb 2>/dev/null' EXIT
run() {  # name, shape, batch
  timeout -k 10 300 python3 bench.py --shape $2 --batch $3 --steps 5 --warmup 2 --no-cpu-baseline > $out/$1_bench.json 2> $out/$1_bench.err || return 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/$1_trace -o t -- python3 bench.py --shape $2 --batch $3 --steps 3 --warmup 1 --no-cpu-baseline > $out/$1_trace.log 2>&1 || return 1
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$1_fetch -o f -- python3 bench.py --shape $2 --batch $3 --steps 1 --warmup 1 --no-cpu-baseline > $out/$1_fetch.log 2>&1 || return 1
  timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$1_write -o w -- python3 bench.py --shape $2 --batch $3 --steps 1 --warmup 1 --no-cpu-baseline > $out/$1_write.log 2>&1 || return 1
  echo "$1 done"
}
run scaled scaled 1024 && run wide wide 16
