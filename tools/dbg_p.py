import numpy as np, torch, sys
sys.path.insert(0, '.')
from star_amd import ops
torch.manual_seed(0)
DEV='cuda:0'
B,h,L,K,hd=64,8,114,116,128
q=torch.randn(B,h,L,hd,device=DEV).bfloat16()
k=torch.randn(B,K,h,hd,device=DEV).bfloat16().permute(0,2,1,3)
bias=torch.randn(B,K,device=DEV)-2.5
cnt=torch.randint(4,K+1,(B,),device=DEV,dtype=torch.int32)
o,p=ops.xattn(q,k,k,bias=bias,key_len=cnt,need_probs=True)
dev=(p.sum(-1)-1).abs()
print(dev.max().item(), (dev>1e-4).sum().item())
w=(dev>1e-4).nonzero()[:10]
print(w, cnt[w[:,0]] if len(w) else None)
