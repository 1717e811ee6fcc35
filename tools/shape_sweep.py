#!/usr/bin/env python3
"""Single-GPU kernel throughput over shapes (world size 1 path: rfa_fwd / rfa_bwd through the C ABI),
to see how the headline-tuned kernels behave elsewhere.  Algorithmic FLOPs: fwd 4*B*H*Sq*Sk*D (/2 causal),
bwd 2.5x.   usage: python tools/shape_sweep.py [B,S,H,Hk,D,causal ...]   (no arguments: the standard table)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ring-flash-attention_amd"))
import torch
from ring_flash_attn.backend import get_backend

dev = torch.device("cuda:0")
be = get_backend()


def timeit(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def run(B, S, H, Hk, D, causal, dtype=torch.bfloat16):
    torch.manual_seed(0)
    q = torch.randn(B, S, H, D, device=dev, dtype=dtype)
    k = torch.randn(B, S, Hk, D, device=dev, dtype=dtype)
    v = torch.randn(B, S, Hk, D, device=dev, dtype=dtype)
    do = torch.randn(B, S, H, D, device=dev, dtype=dtype)
    out = torch.empty_like(q)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    delta = torch.empty_like(lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    scale = D ** -0.5
    n = max(3, min(50, int(2e13 / (B * H * S * S * D))))
    tf = timeit(lambda: be.fwd(q, k, v, softmax_scale=scale, causal=causal, out=out, lse=lse), n)

    def bwd():
        be.bwd_preprocess(do, out, delta)
        be.bwd(do, q, k, v, lse, delta, softmax_scale=scale, causal=causal, dq=dq, dk=dk, dv=dv)

    tb = timeit(bwd, n)
    fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
    print(f"| {B} | {S} | {H}/{Hk} | {D} | {'causal' if causal else 'full'}{'' if dtype == torch.bfloat16 else ' fp16'} | {tf:.3f} | {fl / tf / 1e9:.0f} | {tb:.3f} | "
          f"{2.5 * fl / tb / 1e9:.0f} |", flush=True)


print("| B | S | H/Hk | D | mask | fwd ms | fwd TFLOP/s | bwd ms | bwd TFLOP/s |")
print("|---|---|---|---|---|---|---|---|---|")
if len(sys.argv) > 1:
    for spec in sys.argv[1:]:
        B, S, H, Hk, D, causal = (int(x) for x in spec.split(","))
        run(B, S, H, Hk, D, bool(causal))
    sys.exit(0)
for S in (1024, 2048, 4096, 8192, 16384, 32768):
    run(max(1, 8192 // S), S, 32, 8, 128, True)
run(1, 8192, 32, 32, 128, True)
run(1, 8192, 32, 8, 128, False)
run(1, 8192, 32, 8, 64, True)
run(1, 8192, 32, 8, 96, True)
run(1, 8192, 16, 4, 256, True)           # head dims > 128: csrc/rfa_bigd.hip
run(1, 8192, 16, 4, 256, False)
run(1, 8192, 20, 5, 192, True)
run(2, 4096, 8, 8, 256, True)
run(2, 4096, 16, 16, 128, True)          # BASELINE cfg 2 block shape
run(1, 8192, 32, 8, 128, True, torch.float16)
