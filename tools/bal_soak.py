#!/usr/bin/env python3
"""Soak of the balanced dK/dV schedule's in-kernel pair exchange: the same backward many times, every result compared bit for
bit with the first (a + b == b + a: whichever workgroup of a pair arrives last, the block must come out the same; a stale
read of the partner's slot or of the flag would show as a difference or a non-finite value).  Shapes with pairs on one XCD
and on different XCDs, under load (back-to-back launches, no host sync between them except the comparison).
usage: python tools/bal_soak.py [iterations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ring-flash-attention_amd"))
import torch
from ring_flash_attn import config
from ring_flash_attn.backend import get_backend

dev = torch.device("cuda:0")
be = get_backend()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
bad = 0
for B, S, H, Hk, D in ((1, 8192, 32, 8, 128), (4, 2048, 32, 8, 128), (1, 1024, 4, 1, 128), (2, 2048, 8, 2, 128), (3, 1536, 6, 3, 128), (2, 1024, 8, 4, 64)):
    torch.manual_seed(S + H)
    q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, S, Hk, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(B, S, Hk, D, device=dev, dtype=torch.bfloat16)
    do = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
    out = torch.empty_like(q)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    delta = torch.empty_like(lse)
    be.fwd(q, k, v, softmax_scale=D ** -0.5, causal=True, out=out, lse=lse)
    be.bwd_preprocess(do, out, delta)
    n = iters if S < 8192 else max(50, iters // 4)
    with config.override(dkdv_wide=2):
        ref = None
        diffs = 0
        for i in range(n):
            dq, dk, dv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
            be.bwd(do, q, k, v, lse, delta, softmax_scale=D ** -0.5, causal=True, dq=dq, dk=dk, dv=dv)
            if ref is None:
                ref = (dq, dk, dv)
                assert all(torch.isfinite(t).all().item() for t in ref)
            elif not (torch.equal(dk, ref[1]) and torch.equal(dv, ref[2]) and torch.equal(dq, ref[0])):
                diffs += 1
    bad += diffs
    print(f"B {B} S {S} H {H}/{Hk} D {D}: {n} backwards, {diffs} differing from the first", flush=True)
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
