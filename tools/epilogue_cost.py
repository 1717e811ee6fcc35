#!/usr/bin/env python3
"""Fixed (prologue + epilogue) cost of the attention kernels: Sq = 8192 queries against ONE 64-key
tile, so the K loop is a single iteration and the time is Q/dO loading + result stores."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ring-flash-attention_amd"))
import torch
from ring_flash_attn import _C
from ring_flash_attn.backend import get_backend
be = get_backend(); dev = torch.device("cuda:0")
S, H, HK, D = 8192, 32, 8, 128
q = torch.randn(1, S, H, D, device=dev, dtype=torch.bfloat16)
k = torch.randn(1, S, HK, D, device=dev, dtype=torch.bfloat16); v = torch.randn_like(k)
do = torch.randn_like(q)
out = torch.empty_like(q); lse = torch.empty(1, H, S, device=dev); delta = torch.zeros_like(lse)
oacc = torch.empty(1, S, H, D, device=dev); lacc = torch.empty_like(lse)
dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(k)
sc = D ** -0.5
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n
ks, vs = k[:, :64], v[:, :64]
print("fwd  plain  (bf16 out)      : %.3f ms" % t(lambda: be.fwd(q, ks, vs, softmax_scale=sc, causal=False, out=out, lse=lse)))
print("fwd  acc init (fp32 write)  : %.3f ms" % t(lambda: be.fwd(q, ks, vs, softmax_scale=sc, causal=False, out_acc=oacc, lse_acc=lacc, acc_init=True)))
print("fwd  acc merge (fp32 r+w)   : %.3f ms" % t(lambda: be.fwd(q, ks, vs, softmax_scale=sc, causal=False, out_acc=oacc, lse_acc=lacc)))
print("dq   plain                  : %.3f ms" % t(lambda: be.bwd(do, q, ks, vs, lse, delta, softmax_scale=sc, causal=False, dq=dq, dk=dk[:, :64], dv=dv[:, :64], phases=_C.BWD_COMPUTE | _C.BWD_SKIP_DKDV)))
# dkdv fixed cost: all 8192 keys against ONE 64-row q tile
qs, dos = q[:, :64], do[:, :64]
print("dkdv plain (64 q rows)      : %.3f ms" % t(lambda: be.bwd(dos, qs, k, v, lse[:, :, :64], delta[:, :, :64], softmax_scale=sc, causal=False, dq=dq[:, :64], dk=dk, dv=dv, phases=_C.BWD_COMPUTE | _C.BWD_SKIP_DQ)))
