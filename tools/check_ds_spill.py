"""diagnostic: compare the spilled dS blocks of the dK/dV kernel with a torch restatement, element by element"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ring-flash-attention_amd"))
from ring_flash_attn.backend import get_backend
from ring_flash_attn import _C

def run(S, causal):
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    H = 1
    q, k, v, do = (torch.randn(1, S, H, 128, device=dev, dtype=torch.bfloat16) for _ in range(4))
    be = get_backend(); scale = 128 ** -0.5
    out = torch.empty_like(q); lse = torch.empty(1, H, S, device=dev, dtype=torch.float32)
    be.fwd(q, k, v, softmax_scale=scale, causal=causal, out=out, lse=lse)
    delta = torch.empty_like(lse); be.bwd_preprocess(do, out, delta)
    nb = (S + 31) // 32
    scratch = torch.zeros(nb * nb * 2048, dtype=torch.uint8, device=dev)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    be.bwd(do, q, k, v, lse, delta, softmax_scale=scale, causal=causal, dq=dq, dk=dk, dv=dv, ds_scratch=scratch)
    torch.cuda.synchronize()
    qf, kf, vf, dof = (t[0, :, 0].float() for t in (q, k, v, do))
    s = qf @ kf.T * scale
    p = torch.exp(s - lse[0, 0][:, None])
    if causal:
        p = p * (torch.arange(S, device=dev)[None, :] <= torch.arange(S, device=dev)[:, None])
    ds = p * (dof @ vf.T - delta[0, 0][:, None])
    blocks = scratch.view(torch.bfloat16).view(nb, nb, 128, 8).float()   # [qb][kb][slot][e]
    got = torch.zeros(S, S, device=dev)
    for key in range(32):
        for g in range(2):
            for i in range(2):
                slot = 16 * (key >> 2) + 8 * i + 4 * g + (key & 3)
                for e in range(8):
                    r = 8 * i + e
                    row = (r & 3) + 8 * (r >> 2) + 4 * g
                    got[row::32, key::32] = blocks[:, :, slot, e]
    err = (got - ds).abs()
    if causal:  # blocks entirely above the diagonal are never written
        qb = torch.arange(S, device=dev)[:, None] // 32; kb = torch.arange(S, device=dev)[None, :] // 32
        err = err * (kb <= qb)
    bad = err > 0.02 + 0.02 * ds.abs().max()
    print(f"S={S} causal={causal}: dS max|err| {err.max():.3e} bad {int(bad.sum())} of {S*S}")
    if bad.any():
        idx = bad.nonzero()
        import collections
        print("  by (q tile64, key blk32):", sorted(collections.Counter((int(a) // 64, int(b) // 32) for a, b in idx.tolist()).items())[:40])
        print("  by q row in 32-block:", sorted(collections.Counter(int(a) % 32 for a, b in idx.tolist()).items()))
        print("  by sub-tile t:", sorted(collections.Counter((int(a) % 64) // 32 for a, b in idx.tolist()).items()))
        print("  by key in block:", sorted(collections.Counter(int(b) % 32 for a, b in idx.tolist()).items()))
        for a, b in idx[:6].tolist():
            print("   ", a, b, got[a, b].item(), ds[a, b].item(), "p", p[a, b].item())

for S, c in ((64, False), (128, False), (256, False), (256, True)):
    run(S, c)
