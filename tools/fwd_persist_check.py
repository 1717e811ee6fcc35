#!/usr/bin/env python3
"""The persistent 256-row forward (config.fwd_form = "p8x32" -> RFA_FWD_P8x32) against the plain forms: results must be
bit-identical to the 8 x 32 form (same arithmetic in the same order per query row), and time.
usage: python tools/fwd_persist_check.py [B,S,H,Hk,causal ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ring-flash-attention_amd"))
import torch
from ring_flash_attn import config
from ring_flash_attn.backend import get_backend

dev = torch.device("cuda:0")
be = get_backend()


def timeit(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best


def run(B, S, H, Hk, causal, Sk=None, dtype=torch.bfloat16):
    D = 128
    Sk = Sk or S
    torch.manual_seed(0)
    q = torch.randn(B, S, H, D, device=dev, dtype=dtype)
    k = torch.randn(B, Sk, Hk, D, device=dev, dtype=dtype)
    v = torch.randn(B, Sk, Hk, D, device=dev, dtype=dtype)
    scale = D ** -0.5
    res, times = {}, {}
    n = max(3, min(50, int(2e13 / (B * H * S * Sk * D))))
    for form in ("auto", "8x32", "p8x32", "4x32"):
        with config.override(fwd_form=form):
            out = torch.full_like(q, float("nan"))
            lse = torch.full((B, H, S), float("nan"), dtype=torch.float32, device=dev)
            be.fwd(q, k, v, softmax_scale=scale, causal=bool(causal), out=out, lse=lse)
            torch.cuda.synchronize()
            res[form] = (out.clone(), lse.clone())
            times[form] = timeit(lambda: be.fwd(q, k, v, softmax_scale=scale, causal=bool(causal), out=out, lse=lse), n)
    same = torch.equal(res["p8x32"][0], res["8x32"][0]) and torch.equal(res["p8x32"][1], res["8x32"][1])
    fin = torch.isfinite(res["p8x32"][0]).all().item()
    fl = 4.0 * B * H * S * Sk * D * (0.5 if causal and Sk == S else 1.0)
    print(f"| {B} | {S}x{Sk} | {H}/{Hk} | {'causal' if causal else 'full'} | {times['auto']:.4f} | {times['8x32']:.4f} | {times['p8x32']:.4f} | "
          f"{times['4x32']:.4f} | {fl / times['p8x32'] / 1e9:.0f} | {'bit-identical' if same else 'DIFFERS'} finite {fin} |", flush=True)


print("| B | SqxSk | H/Hk | mask | auto ms | 8x32 ms | p8x32 ms | 4x32 ms | p8x32 TFLOP/s | vs 8x32 |")
print("|---|---|---|---|---|---|---|---|---|---|")
specs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [
    (1, 1000, 4, 2, 1), (2, 777, 8, 2, 0), (1, 8192, 32, 8, 1), (1, 8192, 32, 8, 0), (2, 4096, 32, 8, 1), (4, 2048, 32, 8, 1),
    (8, 1024, 32, 8, 1), (16, 512, 32, 8, 1), (1, 16384, 32, 8, 1), (1, 8192, 32, 32, 1), (3, 1536, 24, 8, 1), (1, 4096, 32, 8, 1),
    (4, 4096, 32, 8, 1), (2, 8192, 32, 8, 1)]
for sp in specs:
    run(*sp)
run(1, 4096, 32, 8, 0, Sk=8192)       # ring "back" step shape
run(1, 4096, 32, 8, 1, Sk=8192)       # causal, bottom-right aligned
run(1, 8192, 32, 8, 1, dtype=torch.float16)
