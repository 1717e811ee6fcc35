import os, sys
sys.path.insert(0, "/root/repo/ring-flash-attention_amd")
import torch
from ring_flash_attn.backend import get_backend
be = get_backend(); dev = torch.device("cuda:0")
S,H,HK,D = 8192,32,8,128
q = torch.randn(1,S,H,D,device=dev,dtype=torch.bfloat16); k = torch.randn(1,S,HK,D,device=dev,dtype=torch.bfloat16); v = torch.randn_like(k)
out = torch.empty_like(q); lse = torch.empty((1,H,S),dtype=torch.float32,device=dev)
oa = torch.zeros((1,S,H,D),dtype=torch.float32,device=dev); la = torch.zeros((1,H,S),dtype=torch.float32,device=dev)
def t(fn,n=30):
    fn(); fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
sc=D**-0.5; h=S//2
for rep in range(3):
    print("plain causal        %.3f"%t(lambda: be.fwd(q,k,v,softmax_scale=sc,causal=True,out=out,lse=lse)))
    print("acc_init causal     %.3f"%t(lambda: be.fwd(q,k,v,softmax_scale=sc,causal=True,out_acc=oa,lse_acc=la,acc_init=True)))
    print("merge causal        %.3f"%t(lambda: be.fwd(q,k,v,softmax_scale=sc,causal=True,out_acc=oa,lse_acc=la)))
    print("plain front (q x k/2) %.3f"%t(lambda: be.fwd(q,k[:,:h],v[:,:h],softmax_scale=sc,causal=False,out=out,lse=lse)))
    print("merge front          %.3f"%t(lambda: be.fwd(q,k[:,:h],v[:,:h],softmax_scale=sc,causal=False,out_acc=oa,lse_acc=la)))
    print("plain back (q/2 x k) %.3f"%t(lambda: be.fwd(q[:,h:],k,v,softmax_scale=sc,causal=False,out=out[:,h:],lse=lse[:,:,h:])))
    print("merge back           %.3f"%t(lambda: be.fwd(q[:,h:],k,v,softmax_scale=sc,causal=False,out_acc=oa[:,h:],lse_acc=la[:,:,h:])))
    print("cast fp32->bf16      %.3f"%t(lambda: be.cast(oa, torch.bfloat16)))
