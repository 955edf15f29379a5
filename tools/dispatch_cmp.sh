#!/bin/bash
# The queue (default) beside the hardware's dispatcher (WBCQP_FLAG_HW_DISPATCH) on the bench's tick stream, three full cycles of its 128 ticks per figure:
#   tools/dispatch_cmp.sh [flags of the second leg, default 2] > gpurun_out/dispatch_cmp.txt
HW=${1:-2}
for B in 512 768 1024 1536 2048 4096 8192; do
  for F in 0 $HW; do
    python bench.py --headline-only --no-cpu-baseline --batch $B --steps 384 --warmup 20 --flags $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('B %5d flags %d: %.3f M QP/s, %.4f ms per step' % ($B, $F, d['value']/1e6, d['ms_per_step']))"
  done
done
