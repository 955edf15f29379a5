#!/bin/bash
# sclk / power read by rocm-smi while the solve kernel runs back to back (B = 8192, 8 s): is the chip at its rated clock under this load?
cd $GRAFT_REPO_ROOT
python - <<'PY' &
import time, numpy as np, torch, sys
sys.path.insert(0,'.')
from inria_wbc_amd import capi, structure, synth
st=structure.talos_structure(); dev=torch.device('cuda',0)
inp=synth.generate(st,1024,synth.SEED_BASE['talos'])
d={k: torch.from_numpy(np.ascontiguousarray(np.tile(v,(8,1)))).to(dev) for k,v in inp.items() if v.size}
B=8192
o=dict(x=torch.zeros(B,st.n,dtype=torch.float64,device=dev),tau=torch.zeros(B,st.na,dtype=torch.float64,device=dev),status=torch.zeros(B,dtype=torch.int32,device=dev),iters=torch.zeros(B,dtype=torch.int32,device=dev))
h=capi.Handle(0,capi.F64); h.set_structure(0,st)
sp=torch.cuda.current_stream().cuda_stream
t0=time.time()
n=0
while time.time()-t0<8:
    for _ in range(50): h.solve_batch(0,B,d,o,stream=sp)
    torch.cuda.synchronize(); n+=50
print("launches",n,"ms each",(time.time()-t0)/n*1e3)
PY
sleep 4
for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -3; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -2; sleep 1; done
wait
rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2
