import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_amd import ops
DEV="cuda"
d=128; hq=4; hkv=1; T=64
g=torch.Generator(device=DEV).manual_seed(0)
q=torch.randn(T,hq,d,device=DEV,generator=g).bfloat16(); ke=torch.randn(T,hkv,d,device=DEV,generator=g).bfloat16(); ve=torch.randn(T,hkv,d,device=DEV,generator=g).bfloat16()
kb=torch.zeros(16,hkv,d,device=DEV).bfloat16(); vb=kb.clone()
qo=torch.tensor([0,T],dtype=torch.int64,device=DEV); kvp=torch.zeros(2,dtype=torch.int32,device=DEV); kvi=torch.zeros(0,dtype=torch.int64,device=DEV)
res={}
for mode in ("0","2"):
    os.environ["RX_EXT_PW"]=mode
    o=torch.zeros(T,hq,d,dtype=torch.bfloat16,device=DEV)
    lse=torch.zeros(T,hq,dtype=torch.float32,device=DEV)
    ops.extend_attention_fwd(q,ke,ve,o,kb,vb,qo,kvp,kvi,None,True,None,T,1.0,1.0,sm_scale=d**-0.5,page_size=1,lse_extend=lse)
    torch.cuda.synchronize(); res[mode]=(o.float(),lse.clone())
r=(res["2"][0]/res["0"][0])
print("ratio o_pw/o_old per row (head 0): median over d")
print(r[:,0].median(-1).values.cpu().numpy().round(3))
print("lse old", res["0"][1][:8,0].cpu().numpy().round(3)); print("lse pw ", res["2"][1][:8,0].cpu().numpy().round(3))
print("exp(lse_old - lse_pw) rows:", torch.exp(res["0"][1][:,0]-res["2"][1][:,0]).cpu().numpy().round(3))
