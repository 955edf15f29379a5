#!/bin/bash
# Builds libwbcqp.so variants side by side (inria_wbc_amd/lib/libwbcqp_<tag>.so) from -D switches, four compiles at a time:
#   tools/variants.sh base "" stop3 "-DWBCQP_X_STOP=3" ...
# tools/run_variants.sh <tag> ... (tools/straggler_time.py --lib <path>) then measures them in one GPU call.
cd "$(dirname "$0")/.."
mkdir -p inria_wbc_amd/lib
n=0
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=on -Rpass-analysis=kernel-resource-usage $flags \
      inria_wbc_amd/csrc/wbcqp_api.hip -o inria_wbc_amd/lib/libwbcqp_$tag.so -ldl > /tmp/variant_$tag.log 2>&1
    python3 - "$tag" <<'PY'
import sys
sys.path.insert(0, ".")
from inria_wbc_amd import build
u = build._resource_usage(open("/tmp/variant_%s.log" % sys.argv[1]).read())
for k, v in u.items():
    if "solve_queue_kernelIdLb1" in k or "solve_kernelIdLb1" in k:
        print(sys.argv[1], k[10:40], "VGPR", v.get("VGPRs"), "AGPR", v.get("AGPRs"), "scratch", v.get("ScratchSize [bytes/lane]"), "occ", v.get("Occupancy [waves/SIMD]"))
PY
  ) &
  n=$((n+1))
  if [ $((n % 4)) -eq 0 ]; then wait; fi
done
wait
