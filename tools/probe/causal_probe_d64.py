"""Dev probe that found the G = 1 row -> token bug of the 256-row extend kernels: q picks score 0 for every key, V[t] = t, so
row t of a causal extend must be mean(0..t); RX_EXT_D256_AT64 selects the kernel.  python tools/probe/causal_probe_d64.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_amd import ops
dev="cuda"; d=64; E=256
g=torch.Generator().manual_seed(1)
kb=torch.zeros(8,1,d).to(torch.bfloat16); vb=torch.zeros(8,1,d).to(torch.bfloat16)
q=torch.zeros(E,1,d); q[:,0,0]=1.0; q=q.to(torch.bfloat16)          # scores = k[:,0]
ke=torch.zeros(E,1,d); ke=ke.to(torch.bfloat16)                      # all scores 0 -> uniform softmax
ve=torch.zeros(E,1,d); ve[:,0,:]=torch.arange(E)[:,None].float(); ve=ve.to(torch.bfloat16)   # V[t] = t in every column
kvp=torch.tensor([0,0],dtype=torch.int32); qo=torch.tensor([0,E],dtype=torch.int64); kvi=torch.zeros(0,dtype=torch.int64)
os.environ["RX_EXT_D256_AT64"]="1"
o=torch.full((E,1,d),float("nan"),dtype=torch.bfloat16,device=dev)
ops.extend_attention_fwd(q.to(dev),ke.to(dev),ve.to(dev),o,kb.to(dev),vb.to(dev),qo.to(dev),kvp.to(dev),kvi.to(dev),None,True,None,E,1.0,1.0,sm_scale=1.0,page_size=1)
torch.cuda.synchronize()
o=o.float().cpu().numpy()[:,0,:]
want=np.array([np.arange(t+1).mean() for t in range(E)])
for t in (0,1,2,3,15,16,17,31,32,63,64,65,127,128,255):
    print(t, "want", want[t], "got cols 0..7", o[t,:8], "col 16,32,48:", o[t,16],o[t,32],o[t,48])
