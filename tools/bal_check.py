#!/usr/bin/env python3
"""The balanced causal dK/dV schedule (config.dkdv_wide = 2 -> RFA_DKDV_BAL) against the shared-range plans:
results (dq must be bit-identical: the dS hand-off does not depend on the schedule; dk / dv differ in the order of
their fp32 sums only), run-to-run determinism, and time.   usage: python tools/bal_check.py [B,S,H,Hk ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ring-flash-attention_amd"))
import torch
from ring_flash_attn import _C, config
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


def run(B, S, H, Hk, D=128):
    torch.manual_seed(0)
    q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, S, Hk, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(B, S, Hk, D, device=dev, dtype=torch.bfloat16)
    do = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
    out = torch.empty_like(q)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    delta = torch.empty_like(lse)
    scale = D ** -0.5
    be.fwd(q, k, v, softmax_scale=scale, causal=True, out=out, lse=lse)
    be.bwd_preprocess(do, out, delta)

    def bwd(res):
        dq, dk, dv = res
        be.bwd(do, q, k, v, lse, delta, softmax_scale=scale, causal=True, dq=dq, dk=dk, dv=dv)

    def fresh():
        return torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))

    res = {}
    times = {}
    n = max(3, min(40, int(1e13 / (B * H * S * S * D))))
    for name, ov in (("auto", {}), ("bal", dict(dkdv_wide=2)), ("ns1", dict(dkdv_wide=1, dkdv_nsplit=1)),
                     ("ns2", dict(dkdv_wide=1, dkdv_nsplit=2)), ("ns4", dict(dkdv_wide=1, dkdv_nsplit=4))):
        with config.override(**ov):
            r = fresh()
            bwd(r)
            torch.cuda.synchronize()
            r2 = fresh()
            bwd(r2)
            torch.cuda.synchronize()
            det = all(torch.equal(a, b) for a, b in zip(r, r2))
            res[name] = r
            times[name] = timeit(lambda: bwd(r2), n)
            if not det:
                print(f"  !! {name}: not deterministic run to run")
    ref = res["ns2"]
    line = f"| {B} | {S} | {H}/{Hk} |"
    for name in ("auto", "bal", "ns1", "ns2", "ns4"):
        line += f" {times[name]:.4f} |"
    fl = 2.5 * 4.0 * B * H * S * S * D * 0.5
    line += f" {fl / times['bal'] / 1e9:.0f} | {fl / min(times['ns1'], times['ns2'], times['ns4']) / 1e9:.0f} |"
    dq_same = torch.equal(res["bal"][0], ref[0])
    dk_d = (res["bal"][1].float() - ref[1].float()).abs().max().item()
    dv_d = (res["bal"][2].float() - ref[2].float()).abs().max().item()
    kmax = ref[1].float().abs().max().item()
    fin = all(torch.isfinite(t).all().item() for t in res["bal"])
    line += f" dq {'same' if dq_same else 'DIFF'} dk {dk_d:.2e} dv {dv_d:.2e} (max|dk| {kmax:.2f}) finite {fin} |"
    print(line, flush=True)


print("| B | S | H/Hk | auto ms | bal ms | ns1 ms | ns2 ms | ns4 ms | bal TFLOP/s | best shared TFLOP/s | parity |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
specs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [
    (1, 512, 4, 2), (1, 1024, 8, 2), (2, 2048, 8, 4), (16, 512, 32, 8), (8, 1024, 32, 8), (4, 2048, 32, 8),
    (2, 4096, 32, 8), (1, 8192, 32, 8), (1, 8192, 32, 32), (1, 16384, 32, 8), (1, 8192, 8, 2), (3, 1536, 32, 8)]
for sp in specs:
    run(*sp)
