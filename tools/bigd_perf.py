#!/usr/bin/env python3
"""Device time of the head-dim > 128 kernels (csrc/rfa_bigd.hip) at shapes with the headline's FLOP count:
S = 8192 causal, H * D = 4096 (H = 16 / Hk = 4 at D = 256, ...).  Prints ms and algorithmic TFLOP/s per call
(forward = 4 B H S^2 D / 2; backward = 2.5 x forward, split dQ 1.0 (0.5 + nothing credited for the recomputation),
dK/dV 2.0 as in bench.py).   usage: python tools/bigd_perf.py [D ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    sys.path.insert(0, p)
import torch

from ring_flash_attn import _C
from ring_flash_attn.backend import get_backend

dev = torch.device("cuda:0")
be = get_backend()


def timed(fn, n=40):
    # (a long warm-up: from idle the chip needs tens of milliseconds of work to reach its sustained clock — 13 launches of
    #  a 0.8 ms kernel read 20 % slow)
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dims = [int(x) for x in sys.argv[1:]] or [256, 192, 128]
    S, B = 8192, 1
    for D in dims:
        H, Hk = 4096 // D, max(1, 1024 // D)
        H = H // Hk * Hk
        q = torch.randn(B, S, H, D, device=dev).to(torch.bfloat16)
        k, v = (torch.randn(B, S, Hk, D, device=dev).to(torch.bfloat16) for _ in range(2))
        do = torch.randn_like(q)
        out, lse = torch.empty_like(q), torch.empty(B, H, S, dtype=torch.float32, device=dev)
        sc = D ** -0.5
        f = 4.0 * B * H * S * S * D / 2
        t_f = timed(lambda: be.fwd(q, k, v, softmax_scale=sc, causal=True, out=out, lse=lse))
        delta = torch.empty_like(lse)
        be.bwd_preprocess(do, out, delta)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        spill = os.environ.get("RFA_BWD_DS_SPILL", "1")   # (D = 128 and 256 have a 5-GEMM form; RFA_BWD_DS_SPILL=0: 7-GEMM everywhere)
        kw = dict(softmax_scale=sc, causal=True, dq=dq, dk=dk, dv=dv)
        t_b = timed(lambda: be.bwd(do, q, k, v, lse, delta, **kw))
        t_dq = timed(lambda: be.bwd(do, q, k, v, lse, delta, phases=_C.BWD_SKIP_DKDV, **kw))
        t_kv = timed(lambda: be.bwd(do, q, k, v, lse, delta, phases=_C.BWD_SKIP_DQ, **kw))
        print(f"D={D:3d} H={H}/{Hk}: fwd {t_f:.3f} ms {f / t_f / 1e9:6.0f} TF | bwd {t_b:.3f} ms {2.5 * f / t_b / 1e9:6.0f} TF "
              f"(dq {t_dq:.3f} ms {0.5 * f / t_dq / 1e9:5.0f} TF credited, dk/dv {t_kv:.3f} ms {2.0 * f / t_kv / 1e9:5.0f} TF)", flush=True)


if __name__ == "__main__":
    main()
