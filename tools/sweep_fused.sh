for b in 32 128 256 512 1024; do for f in 0 1; do for g in "" "--graph"; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --batch $b --steps 300 --fused-reductions $f $g > gpurun_out/sw.json 2>> gpurun_out/sw.err || exit 1
  python -c "
import json
r=json.load(open('gpurun_out/sw.json'))
print('B=$b fused=$f graph=%s  %.4f ms  %.0f img/s' % ('$g' != '', r['ms_per_step'], r['value']))" >> gpurun_out/sweep.txt
done; done; done
