#!/usr/bin/env python3
"""Forward kernel forms side by side (RFA_FWD_FORM = 8x32 | 4x64): numerics against the CPU oracle and against each
other on ragged / bottom-right / packed / accumulate-mode cases, then device time at the headline shape.
usage: python tools/fwd64_check.py [--perf-only]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    sys.path.insert(0, p)
import torch

from ring_flash_attn.backend import get_backend

BF = torch.bfloat16
dev = torch.device("cuda:0")
be = get_backend()


def run(form, q, k, v, causal, **kw):
    os.environ["RFA_FWD_FORM"] = form
    __import__("ring_flash_attn").config.reload()
    B, Sq, H, D = q.shape if q.dim() == 4 else (1,) + tuple(q.shape)
    out = torch.empty_like(q)
    lse = torch.empty((q.shape[0], q.shape[2], q.shape[1]) if q.dim() == 4 else (q.shape[1], q.shape[0]), dtype=torch.float32, device=dev)
    be.fwd(q, k, v, softmax_scale=q.shape[-1] ** -0.5, causal=causal, out=out, lse=lse, **kw)
    return out, lse


def numerics():
    from oracle import flash_attn_ref as O

    bad = 0
    g = torch.Generator().manual_seed(0)
    for (B, Sq, Sk, H, Hk, causal, dt) in [(1, 1000, 1000, 4, 2, True, BF), (2, 300, 777, 2, 2, True, BF), (1, 513, 513, 2, 1, False, BF),
                                           (1, 64, 64, 1, 1, True, BF), (1, 3824, 3824, 5, 5, True, BF), (1, 900, 260, 2, 2, True, torch.float16)]:
        q = torch.randn(B, Sq, H, 128, generator=g).to(dt)
        k = torch.randn(B, Sk, Hk, 128, generator=g).to(dt)
        v = torch.randn(B, Sk, Hk, 128, generator=g).to(dt)
        ro, rl, _, _ = O._flash_attn_forward(q, k, v, 0.0, 128 ** -0.5, causal)
        qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
        o8, l8 = run("8x32", qd, kd, vd, causal)
        o4, l4 = run("4x64", qd, kd, vd, causal)
        fin = torch.isfinite(rl)
        e4 = (o4.cpu().float() - ro.float()).abs().max().item()
        e8 = (o8.cpu().float() - ro.float()).abs().max().item()
        el = (l4.cpu() - rl)[fin].abs().max().item() if fin.any() else 0.0
        pat = torch.equal(torch.isfinite(l4.cpu()), fin)
        ok = e4 <= max(2 * e8, 8e-3) and el < 1e-4 and pat
        bad += not ok
        print(f"dense B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} causal={causal} {dt}: out err 4x64 {e4:.2e} (8x32 {e8:.2e}) lse {el:.1e} pattern {pat} {'ok' if ok else 'FAIL'}")
    # spike keys: force the deferred-rescale branch in the middle of the loop
    q = torch.randn(1, 512, 2, 128, generator=g).to(BF)
    k = torch.randn(1, 512, 2, 128, generator=g).to(BF)
    v = torch.randn(1, 512, 2, 128, generator=g).to(BF)
    k[0, 300] = q[0, 400] * 3.0
    k[0, 100, 1] = q[0, 200, 1] * 5.0
    ro, rl, _, _ = O._flash_attn_forward(q, k, v, 0.0, 128 ** -0.5, True)
    o4, l4 = run("4x64", q.to(dev), k.to(dev), v.to(dev), True)
    e4, el = (o4.cpu().float() - ro.float()).abs().max().item(), (l4.cpu() - rl).abs().max().item()
    ok = e4 < 2e-2 and el < 1e-3
    bad += not ok
    print(f"spike keys (rescale branch): out err {e4:.2e} lse {el:.1e} {'ok' if ok else 'FAIL'}")
    # packed sequences + halves + accumulate mode: against the 8x32 form
    cu = torch.tensor([0, 128, 1248, 2240], dtype=torch.int32, device=dev)
    T = 2240
    q = torch.randn(T, 4, 128, generator=g).to(BF).to(dev)
    k = torch.randn(T, 2, 128, generator=g).to(BF).to(dev)
    v = torch.randn(T, 2, 128, generator=g).to(BF).to(dev)
    vl = dict(cu_seqlens_q=cu, cu_seqlens_k=cu, max_seqlen_q=1120, max_seqlen_k=1120)
    res = {}
    for form in ("8x32", "4x64"):
        os.environ["RFA_FWD_FORM"] = form
        __import__("ring_flash_attn").config.reload()
        oa = torch.empty(T, 4, 128, dtype=torch.float32, device=dev)
        la = torch.empty(4, T, dtype=torch.float32, device=dev)
        sc = 128 ** -0.5
        be.fwd(q, k, v, softmax_scale=sc, causal=True, out_acc=oa, lse_acc=la, acc_init=True, **vl)
        be.fwd(q, k, v, softmax_scale=sc, causal=False, out_acc=oa, lse_acc=la, k_half=1, **vl)
        be.fwd(q, k, v, softmax_scale=sc, causal=False, out_acc=oa, lse_acc=la, q_half=2, **vl)
        res[form] = (oa.cpu(), la.cpu())
    e, el = (res["4x64"][0] - res["8x32"][0]).abs().max().item(), (res["4x64"][1] - res["8x32"][1]).abs().max().item()
    ok = e < 5e-3 and el < 1e-4
    bad += not ok
    print(f"packed + halves + accumulate vs 8x32: out {e:.2e} lse {el:.1e} {'ok' if ok else 'FAIL'}")
    return bad


def perf():
    import bench

    hip = bench._Hip()
    S, H, Hk, D = 8192, 32, 8, 128
    q = torch.randn(1, S, H, D, device=dev, dtype=BF)
    k = torch.randn(1, S, Hk, D, device=dev, dtype=BF)
    v = torch.randn(1, S, Hk, D, device=dev, dtype=BF)
    f = 4.0 * H * S * S * D / 2
    for _ in range(200):                     # clocks up
        run("8x32", q, k, v, True)
    for rep in range(2):
        for form in ("8x32", "4x64"):
            for causal, fl in ((True, f), (False, 2 * f)):
                for _ in range(3):
                    run(form, q, k, v, causal)
                e0, e1 = hip.event(), hip.event()
                hip.record(e0)
                for _ in range(20):
                    run(form, q, k, v, causal)
                hip.record(e1)
                ms = hip.ms(e0, e1) / 20
                print(f"perf {form} causal={causal}: {ms:.4f} ms  {fl / ms / 1e9:.0f} TFLOP/s")


if __name__ == "__main__":
    bad = 0 if "--perf-only" in sys.argv else numerics()
    perf()
    print("FWD64 CHECK", "FAILED" if bad else "PASSED")
    sys.exit(1 if bad else 0)
