#!/usr/bin/env python3
"""Device time of the backward's kernels at the headline shape (q (1, 8192, 32, 128), 8 kv heads, bf16, causal) for the
library RFA_LIB_PATH points to (tools/ab_variants.py builds tuning variants): dK/dV kernel, dQ kernel, reduction —
rfa_bwd_args.prof_events, 30 backward calls after a spin-up.  usage: [RFA_LIB_PATH=...] python tools/bwd_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    sys.path.insert(0, p)
import torch

import bench
from ring_flash_attn.backend import get_backend


def main():
    be, hip, dev = get_backend(), bench._Hip(), torch.device("cuda:0")
    S, H, Hk, D = 8192, 32, 8, 128
    q = torch.randn(1, S, H, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(1, S, Hk, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(1, S, Hk, D, device=dev, dtype=torch.bfloat16)
    do = torch.randn_like(q)
    out, lse = torch.empty_like(q), torch.empty(1, H, S, device=dev, dtype=torch.float32)
    sc = D ** -0.5
    be.fwd(q, k, v, softmax_scale=sc, causal=True, out=out, lse=lse)
    delta = torch.empty_like(lse)
    be.bwd_preprocess(do, out, delta)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    evs = []
    for it in range(230):
        ev = (hip.C.c_void_p * 4)(*[hip.event() for _ in range(4)]) if it >= 200 else None
        be.bwd(do, q, k, v, lse, delta, softmax_scale=sc, causal=True, dq=dq, dk=dk, dv=dv, prof_events=ev)
        if ev is not None:
            evs.append(ev)
    torch.cuda.synchronize()
    t = [sum(hip.ms(e[i], e[i + 1]) for e in evs) / len(evs) for i in range(3)]
    f = 4.0 * H * S * S * D / 2
    print(f"dkdv {t[0]:.4f} ms ({2 * f / t[0] / 1e9:.0f} TFLOP/s)  dq {t[1]:.4f} ms  reduce {t[2]:.4f} ms  total {sum(t):.4f} ms")


if __name__ == "__main__":
    main()
