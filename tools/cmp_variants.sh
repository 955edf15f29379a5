#!/bin/bash
# usage: tools/cmp_variants.sh tag1 tag2 ...   (libraries inria_wbc_amd/lib/libwbcqp_<tag>.so built by tools/variants.sh)
# one line per variant: the straggler and the median QP of the bench tick alone (us), the per-iteration fit, and tools/throughput_time.py's three regimes
for t in "$@"; do
  lib=inria_wbc_amd/lib/libwbcqp_$t.so
  a=$(python tools/straggler_time.py --lib $lib 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('straggler %.1f us (%d it) median %.1f fit %.3f us/it' % (d['longest'][0]['us_alone'], d['longest'][0]['iters'], d['median']['us_alone'], d['fit_us']['per_iteration']))")
  b=$(python tools/throughput_time.py --lib $lib 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('B8192 %.2f M  cfg2 %.2f M  B1024 %.2f M' % (d['stream_b8192']['qps']/1e6, d['config2_replayed_b1024']['qps']/1e6, d['stream_b1024']['qps']/1e6))")
  echo "$t: $a | $b"
done
