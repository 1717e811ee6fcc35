#!/usr/bin/env python3
"""Compute-only cost of ONE rank's zigzag schedule at world size W on a single GPU: the exact kernel
sequence of zigzag_ring_flash_attn_{forward,backward} (fused merge epilogues, fp32 accumulators,
two-phase backward) with the ring exchange replaced by pre-filled local buffers.  Compared with
W x the world-size-1 time, it isolates what the multi-step form costs before any xGMI traffic.
usage: python tools/virtual_ring_bench.py [W] [rank] [kv_heads] [ring|gather]   (default: gather = the
default exchange of the dense zigzag path: single-phase backward into per-chunk fp32 slots)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ring-flash-attention_amd"))
import torch
from ring_flash_attn import _C
from ring_flash_attn.backend import get_backend

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
RANK = int(sys.argv[2]) if len(sys.argv) > 2 else 3
HK = int(sys.argv[3]) if len(sys.argv) > 3 else 8
MODE = sys.argv[4] if len(sys.argv) > 4 else "gather"
S, H, D = 8192, 32, 128
dev = torch.device("cuda:0")
be = get_backend()
torch.manual_seed(0)
q = torch.randn(1, S, H, D, device=dev, dtype=torch.bfloat16)
ks = [torch.randn(1, S, HK, D, device=dev, dtype=torch.bfloat16) for _ in range(W)]
vs = [torch.randn(1, S, HK, D, device=dev, dtype=torch.bfloat16) for _ in range(W)]
dout = torch.randn(1, S, H, D, device=dev, dtype=torch.bfloat16)
scale = D ** -0.5
half = S // 2


def fwd():
    out_acc = torch.empty((1, S, H, D), dtype=torch.float32, device=dev)
    lse_acc = torch.empty((1, H, S), dtype=torch.float32, device=dev)
    for step in range(W):
        k, v = ks[step], vs[step]
        if step == 0:
            be.fwd(q, k, v, softmax_scale=scale, causal=True, out_acc=out_acc, lse_acc=lse_acc, acc_init=True)
        elif step <= RANK:
            be.fwd(q, k[:, :half], v[:, :half], softmax_scale=scale, causal=False, out_acc=out_acc, lse_acc=lse_acc)
        else:
            be.fwd(q[:, half:], k, v, softmax_scale=scale, causal=False, out_acc=out_acc[:, half:], lse_acc=lse_acc[:, :, half:])
    return be.cast(out_acc, torch.bfloat16), lse_acc


def bwd_gather(out, lse):
    delta = torch.empty((1, H, S), dtype=torch.float32, device=dev)
    be.bwd_preprocess(dout, out, delta)
    dq = torch.empty((1, S, H, D), dtype=torch.float32, device=dev)
    dk_all = torch.zeros((W, 1, S, HK, D), dtype=torch.float32, device=dev)
    dv_all = torch.zeros_like(dk_all)
    for step in range(W):
        k, v = ks[step], vs[step]
        if step == 0:
            be.bwd(dout, q, k, v, lse, delta, softmax_scale=scale, causal=True, dq_acc=dq, dk_acc=dk_all[0], dv_acc=dv_all[0], acc_init=True)
        elif step <= RANK:
            be.bwd(dout, q, k[:, :half], v[:, :half], lse, delta, softmax_scale=scale, causal=False, dq_acc=dq,
                   dk_acc=dk_all[step][:, :half], dv_acc=dv_all[step][:, :half], phases=_C.BWD_KV_OVERWRITE)
        else:
            be.bwd(dout[:, half:], q[:, half:], k, v, lse[:, :, half:], delta[:, :, half:], softmax_scale=scale, causal=False,
                   dq_acc=dq[:, half:], dk_acc=dk_all[step], dv_acc=dv_all[step], phases=_C.BWD_KV_OVERWRITE)
    # (the reduce-scatter would run here; its local part is one pass over dk_all / dv_all)
    return be.cast(dq, torch.bfloat16), be.cast(dk_all[0], torch.bfloat16), be.cast(dv_all[0], torch.bfloat16)


def bwd(out, lse):
    if MODE == "gather":
        return bwd_gather(out, lse)
    delta = torch.empty((1, H, S), dtype=torch.float32, device=dev)
    be.bwd_preprocess(dout, out, delta)
    dq = torch.empty((1, S, H, D), dtype=torch.float32, device=dev)
    dk = torch.empty((1, S, HK, D), dtype=torch.float32, device=dev)
    dv = torch.empty_like(dk)
    for step in range(W):
        k, v = ks[step], vs[step]
        if step == 0:
            be.bwd(dout, q, k, v, lse, delta, softmax_scale=scale, causal=True, dq_acc=dq, dk_acc=dk, dv_acc=dv, acc_init=True)
            continue
        front = step <= RANK
        if front:
            args = (dout, q, k[:, :half], v[:, :half], lse, delta)
            dqv, dkv, dvv = dq, dk[:, :half], dv[:, :half]
        else:
            args = (dout[:, half:], q[:, half:], k, v, lse[:, :, half:], delta[:, :, half:])
            dqv, dkv, dvv = dq[:, half:], dk, dv
        be.bwd(*args, softmax_scale=scale, causal=False, dq_acc=dqv, dk_acc=dkv, dv_acc=dvv, phases=_C.BWD_COMPUTE)
        be.bwd(*args, softmax_scale=scale, causal=False, dq_acc=dqv, dk_acc=dkv, dv_acc=dvv, phases=_C.BWD_REDUCE)
    return be.cast(dq, torch.bfloat16), be.cast(dk, torch.bfloat16), be.cast(dv, torch.bfloat16)


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


tf, (out, lse) = timeit(fwd)
tb, _ = timeit(lambda: bwd(out, lse))
# world-size-1 reference on the same GPU
o1 = torch.empty_like(q); l1 = torch.empty((1, H, S), dtype=torch.float32, device=dev)
t1f, _ = timeit(lambda: be.fwd(q, ks[0], vs[0], softmax_scale=scale, causal=True, out=o1, lse=l1), 20)
d1 = torch.empty_like(l1); be.bwd_preprocess(dout, o1, d1)
g = [torch.empty_like(q), torch.empty_like(ks[0]), torch.empty_like(vs[0])]
t1b, _ = timeit(lambda: be.bwd(dout, q, ks[0], vs[0], l1, d1, softmax_scale=scale, causal=True, dq=g[0], dk=g[1], dv=g[2]), 20)
print(f"W={W} rank={RANK} Hk={HK} {MODE}: fwd {tf:.3f} ms (= {tf / W:.3f}/step, W=1 kernel {t1f:.3f})  "
      f"bwd {tb:.3f} ms (= {tb / W:.3f}/step, W=1 {t1b:.3f})  fwd+bwd {tf + tb:.2f} ms -> {1e3 / (tf + tb):.1f} it/s compute-only; "
      f"ideal {1e3 / (W * (t1f + t1b)):.1f} it/s; efficiency {(W * (t1f + t1b)) / (tf + tb):.3f}")
