#!/bin/bash
# usage: run_variants.sh tag1 tag2 ... ; prints one line per variant
for t in "$@"; do
  python tools/straggler_time.py --lib inria_wbc_amd/lib/libwbcqp_$t.so 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['lib'], 'longest', d['longest'][0]['us_alone'], [x['us_alone'] for x in d['longest'][1:]], 'median', d['median']['us_alone'], 'fit', d['fit_us'])
"
done
