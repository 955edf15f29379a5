#!/bin/bash
# Per-tick wall time of the C++ facade's closed loop (PosTracker + humanoid::move_com on the Talos-like model) for several
# batch sizes: what a caller of behavior->update() sees, host staging included.  Usage (GPU box): tools/facade_latency.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python3 tools/emit_configs.py
for B in 1 8 64 512; do
  sed "s/batch: 8/batch: $B/; s#model: talos_like.model.yaml#model: $ROOT/configs/talos/talos_like.model.yaml#; s#frames: frames.yaml#frames: $ROOT/configs/talos/frames.yaml#; s#tasks: tasks.yaml#tasks: $ROOT/configs/talos/tasks.yaml#" configs/talos/pos_tracker_model.yaml > /tmp/pt_$B.yaml
  echo "batch $B: $(inria_wbc_amd/lib/qp_timer_test /tmp/pt_$B.yaml configs/talos/squat.yaml - 400 2>&1 | grep '^t:' | tail -200 | awk -F'solver:' '{split($2,a,"ms"); s+=a[1]; n++} END {printf "%.3f ms per tick (mean of the last %d)", s/n, n}')"
done
