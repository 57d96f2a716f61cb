export DIMS=128x128
for i in 1 2; do
echo -n "auto "; python3 tools/extend_dims.py 2>/dev/null | tail -1
echo -n "autopack off "; RX_EXT32_AUTOPACK=0 python3 tools/extend_dims.py 2>/dev/null | tail -1
done
python - <<'PY'
import os, torch, sys
sys.path.insert(0, os.getcwd())
from sglang_amd import ops
dev="cuda"; HQ,HKV,D,ps=32,8,128,16
g=torch.Generator(device=dev).manual_seed(3)
for P,E,chunk in ((3584,512,4),(700,300,3),(0,1024,2),(5000,257,2)):
    n_pages=(P+ps-1)//ps+chunk*((E+ps-1)//ps)+1
    kb=torch.randn(n_pages,HKV,ps,D,device=dev,generator=g).to(torch.bfloat16); vb=torch.randn(n_pages,HKV,ps,D,device=dev,generator=g).to(torch.bfloat16)
    lay=ops.kv_layout_hnd(kb,vb); T=chunk*E
    q=torch.randn(T,HQ,D,device=dev,generator=g).to(torch.bfloat16); ke=torch.randn(T,HKV,D,device=dev,generator=g).to(torch.bfloat16); ve=torch.randn(T,HKV,D,device=dev,generator=g).to(torch.bfloat16)
    pages=torch.randperm(n_pages-1,device=dev,generator=g)[:(P+ps-1)//ps]+1
    slots=(pages[:,None]*ps+torch.arange(ps,device=dev)[None,:]).reshape(-1)[:P].to(torch.int64)
    kvi=slots.repeat(chunk); kvp=(torch.arange(chunk+1,device=dev)*P).to(torch.int32); qo=(torch.arange(chunk+1,device=dev)*E).to(torch.int64)
    outs=[]
    for mode in ("1","0"):
        os.environ["RX_EXT32_AUTOPACK"]=mode
        o=torch.empty(T,HQ,D,device=dev,dtype=torch.bfloat16); lse=torch.zeros(T,HQ,device=dev)
        ops.extend_attention_fwd(q,ke,ve,o,kb,vb,qo,kvp,kvi,None,True,None,E,1.0,1.0,sm_scale=D**-0.5,page_size=ps,kv_layout=lay,lse_extend=lse)
        torch.cuda.synchronize(); outs.append((o.clone(),lse.clone()))
    print(P,E,chunk,"bit-identical o:",torch.equal(outs[0][0].view(torch.int16),outs[1][0].view(torch.int16)),"lse:",torch.equal(outs[0][1],outs[1][1]))
PY
