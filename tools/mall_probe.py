#!/usr/bin/env python3
"""Does the dS hand-off (dK/dV kernel -> dQ kernel) get faster when it fits the 256 MiB Infinity Cache?
Times the two kernels of one backward (rfa_bwd_args.prof_events) over shapes whose causal dS scratch is
34 MB ... 2.2 GB and prints the dQ kernel's rate per dS byte.  usage: python tools/mall_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    sys.path.insert(0, p)
import torch

import bench
from ring_flash_attn.backend import get_backend


def main():
    be = get_backend()
    hip = bench._Hip()
    dev = torch.device("cuda:0")
    D = 128
    print(f"{'B':>3s} {'S':>6s} {'H':>3s} {'Hk':>3s} {'dS MB':>8s} {'dkdv ms':>8s} {'dq_ds ms':>8s} {'dq_ds GB/s (dS)':>16s} {'dq TFLOP/s':>10s}")
    for B, S, H, Hk in [(1, 1024, 32, 8), (1, 2048, 32, 8), (4, 2048, 32, 8), (1, 4096, 32, 8), (1, 8192, 8, 2),
                        (1, 8192, 32, 8), (16, 2048, 32, 8)]:
        q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
        k = torch.randn(B, S, Hk, D, device=dev, dtype=torch.bfloat16)
        v = torch.randn(B, S, Hk, D, device=dev, dtype=torch.bfloat16)
        do = torch.randn_like(q)
        out, lse = torch.empty_like(q), torch.empty(B, H, S, device=dev, dtype=torch.float32)
        sc = D ** -0.5
        be.fwd(q, k, v, softmax_scale=sc, causal=True, out=out, lse=lse)
        delta = torch.empty_like(lse)
        be.bwd_preprocess(do, out, delta)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        evs = []
        for it in range(12):
            ev = (hip.C.c_void_p * 4)(*[hip.event() for _ in range(4)])
            be.bwd(do, q, k, v, lse, delta, softmax_scale=sc, causal=True, dq=dq, dk=dk, dv=dv, prof_events=ev)
            if it >= 2:
                evs.append(ev)
        torch.cuda.synchronize()
        t1 = sum(hip.ms(e[0], e[1]) for e in evs) / len(evs)
        t2 = sum(hip.ms(e[1], e[2]) for e in evs) / len(evs)
        nb = S // 32
        ds = B * H * nb * (nb + 1) // 2 * 2048
        fl = 0.5 * 4 * B * H * S * S * D / 2
        print(f"{B:3d} {S:6d} {H:3d} {Hk:3d} {ds / 1e6:8.1f} {t1:8.4f} {t2:8.4f} {ds / t2 / 1e6:16.1f} {fl / t2 / 1e9:10.1f}")


if __name__ == "__main__":
    main()
