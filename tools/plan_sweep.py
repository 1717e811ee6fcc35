#!/usr/bin/env python3
"""Do the launch-plan rules pick (nearly) the best kernel form around their own boundaries?  (VERDICT r5 weak #7 / next #8.)

csrc/rfa_api.cpp chooses the forward form (256- or 128-row workgroups, split-KV shares) and the dK/dV plan (128- or 256-key
workgroups, shares of the query range) from constants tuned on 256-CU boxes: `wgs8 >= 112 && wgs8 <= 192 && tiles >= 128`,
`wgs128 * (ns + 1) <= 640`, `sq <= 1024 ? wgs < 160 : wgs * ns < 320`, ...  This sweep walks the shapes on both sides of
those boundaries, times the form the library CHOOSES against every form that can be FORCED (config.fwd_form /
fwd_kv_nsplit / dkdv_wide / dkdv_nsplit — the switches the rules' tuning used) and reports chosen / best.  A neighbouring
shape that falls off a cliff shows as a ratio well above 1.

    python tools/plan_sweep.py [--quick]        prints one markdown table per direction; exit status 1 if any ratio > --tol
tests/test_gpu_plan_rules.py runs `sweep(quick=True)` in the extended GPU tier."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _time(fn, iters, reps=3):
    import torch

    fn()
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best


def fwd_points(quick):
    """(label, B, Sq, Sk, H, Hk, causal): q blocks against long key chains (llama3 head groups: the split rules), and short
    self-attention sequences (the 256- vs 128-row rule)"""
    pts = []
    for H in ((12, 16, 24, 28) if quick else (12, 14, 16, 20, 24, 28)):            # wgs8 = 8 H at Sq = 2048: 96 .. 224
        for Sk in ((6144, 8192, 10240) if not quick else (6144, 8192)):             # 96 / 128 / 160 key tiles
            pts.append((f"cross Sq2048 Sk{Sk} H{H}", 1, 2048, Sk, H, max(1, H // 4) if H % 4 == 0 else H // 2, False))
    for S in (768, 1024, 1280):
        pts.append((f"self S{S} B{8192 // S} H32", 8192 // S, S, S, 32, 8, True))
    if not quick:
        # llama3 head groups (BASELINE config 5: 2048 tokens per rank against the gathered keys of ranks 0 .. r, causal,
        # bottom-right aligned): one K/V head with its 2 query heads, half and all of the Qwen3-0.6B layer's 16 / 8 heads
        for Sk in (8192, 16384):
            for H, Hk in ((2, 1), (4, 2), (8, 4), (16, 8)):
                pts.append((f"llama3 Sq2048 Sk{Sk} H{H}/{Hk}", 1, 2048, Sk, H, Hk, True))
        for S, B in ((2048, 4), (4096, 2), (4096, 1), (8192, 1)):
            pts.append((f"self S{S} B{B} H32", B, S, S, 32, 8, True))
            pts.append((f"self S{S} B{B} H8", B, S, S, 8, 2, True))
        # shapes the round-6 plan estimate was NOT fitted on (validation)
        pts.append(("cross Sq4096 Sk12288 H10", 1, 4096, 12288, 10, 5, False))
        pts.append(("cross Sq1536 Sk9216 H24", 1, 1536, 9216, 24, 6, False))
        pts.append(("llama3 Sq4096 Sk32768 H8/4", 1, 4096, 32768, 8, 4, True))
        pts.append(("llama3 Sq1024 Sk8192 H16/8", 1, 1024, 8192, 16, 8, True))
        pts.append(("self S3072 B2 H16", 2, 3072, 3072, 16, 4, True))
        pts.append(("self S6144 B1 H12", 1, 6144, 6144, 12, 4, True))
        pts.append(("self S2048 B1 H16", 1, 2048, 2048, 16, 8, True))
    return pts


def bwd_points(quick):
    """(label, B, Sq, Sk, H, Hk, causal)"""
    pts = []
    for S in ((768, 1024, 1280) if quick else (768, 1024, 1280, 2048, 4096)):
        for Hk in ((1, 2, 8) if quick else (1, 2, 4, 8)):
            pts.append((f"S{S} B{max(1, 8192 // S)} Hk{Hk}", max(1, 8192 // S), S, S, 4 * Hk, Hk, True))
    # the two remote steps of the zigzag ring at the headline shape (W = 8: all queries x the first half of the keys, the
    # second half of the queries x all keys, no mask) and a llama3 head-group backward (rank 3 / 7 of 8)
    pts.append(("zigzag front step Sq8192 Sk4096 Hk8", 1, 8192, 4096, 32, 8, False))
    pts.append(("zigzag back step Sq4096 Sk8192 Hk8", 1, 4096, 8192, 32, 8, False))
    pts.append(("full S8192 B1 Hk8", 1, 8192, 8192, 32, 8, False))
    if not quick:
        for S, B, Hk in ((2048, 1, 8), (2048, 2, 4), (4096, 1, 2), (8192, 1, 1), (8192, 1, 2), (8192, 1, 4)):
            pts.append((f"S{S} B{B} Hk{Hk}", B, S, S, 4 * Hk, Hk, True))
        for S, B, Hk in ((512, 16, 8), (1024, 16, 8), (1024, 12, 8), (2048, 8, 8), (2048, 3, 8), (4096, 4, 8), (8192, 2, 8), (1536, 3, 8),
                         (8192, 1, 16), (8192, 1, 32), (5120, 1, 8), (1024, 4, 8), (2048, 2, 8)):              # multi-batch / partly filled (round 6, second session)
            pts.append((f"S{S} B{B} Hk{Hk}", B, S, S, 4 * Hk if Hk < 32 else Hk, Hk, True))
        for S, B, Hk in ((3072, 2, 2), (6144, 1, 8), (1536, 4, 4), (8192, 1, 8), (16384, 1, 2)):      # validation (not fitted on)
            pts.append((f"S{S} B{B} Hk{Hk}", B, S, S, 4 * Hk, Hk, True))
        pts.append(("llama3 Sq2048 Sk8192 H16/8", 1, 2048, 8192, 16, 8, True))
        pts.append(("llama3 Sq2048 Sk16384 H16/8", 1, 2048, 16384, 16, 8, True))
        pts.append(("llama3 Sq2048 Sk16384 H2/1", 1, 2048, 16384, 2, 1, True))
        # wide head dims (rfa_bigd.hip: one form, the share count is the plan)
        pts.append(("D256 S8192 H16/4", 1, 8192, 8192, 16, 4, True, 256))
        pts.append(("D192 S8192 H20/5", 1, 8192, 8192, 20, 5, True, 192))
        pts.append(("D256 S4096 B2 H8/8", 2, 4096, 4096, 8, 8, True, 256))
        pts.append(("D256 S2048 H16/4", 1, 2048, 2048, 16, 4, True, 256))
    return pts


def sweep(quick=False, log=print, dump=None):
    import torch

    from ring_flash_attn import config
    from ring_flash_attn.backend import get_backend

    be, dev = get_backend(), torch.device("cuda:0")
    D = 128
    torch.manual_seed(0)
    rows = []
    log("| forward shape | chosen ms | best forced ms | best form | chosen / best |")
    log("|---|---|---|---|---|")
    for label, B, Sq, Sk, H, Hk, causal in fwd_points(quick):
        q = torch.randn(B, Sq, H, D, device=dev, dtype=torch.bfloat16)
        k = torch.randn(B, Sk, Hk, D, device=dev, dtype=torch.bfloat16)
        v = torch.randn(B, Sk, Hk, D, device=dev, dtype=torch.bfloat16)
        out, lse = torch.empty_like(q), torch.empty(B, H, Sq, device=dev, dtype=torch.float32)

        def run():
            be.fwd(q, k, v, softmax_scale=D ** -0.5, causal=causal, out=out, lse=lse)

        iters = 20
        chosen = _time(run, iters)
        forced = {}
        for form in ("8x32", "4x32"):
            for ns in ((1, 2, 3, 4, 6, 8) if Sk >= 4096 else (1,)):
                with config.override(fwd_form=form, fwd_kv_nsplit=ns):
                    forced[f"{form} ns{ns}"] = _time(run, iters)
        chosen = min(chosen, _time(run, iters))                   # (re-timed behind the forced forms: drift shows as a ratio < 1)
        bname = min(forced, key=forced.get)
        if dump is not None:
            dump.append(dict(dir="fwd", label=label, B=B, Sq=Sq, Sk=Sk, H=H, Hk=Hk, causal=causal, chosen=chosen, forced=forced))
        rows.append(("fwd", label, chosen, forced[bname], bname))
        log(f"| {label} | {chosen:.4f} | {forced[bname]:.4f} | {bname} | {chosen / forced[bname]:.3f} |")
    log("")
    log("| backward shape | chosen ms | best forced ms | best plan | chosen / best |")
    log("|---|---|---|---|---|")
    for pt in bwd_points(quick):
        label, B, S, Sk, H, Hk, causal = pt[:7]
        D = pt[7] if len(pt) > 7 else 128
        q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
        k = torch.randn(B, Sk, Hk, D, device=dev, dtype=torch.bfloat16)
        v = torch.randn(B, Sk, Hk, D, device=dev, dtype=torch.bfloat16)
        do = torch.randn_like(q)
        out, lse = torch.empty_like(q), torch.empty(B, H, S, device=dev, dtype=torch.float32)
        delta = torch.empty_like(lse)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        be.fwd(q, k, v, softmax_scale=D ** -0.5, causal=causal, out=out, lse=lse)
        be.bwd_preprocess(do, out, delta)

        def run():
            be.bwd(do, q, k, v, lse, delta, softmax_scale=D ** -0.5, causal=causal, dq=dq, dk=dk, dv=dv)

        iters = 15
        chosen = _time(run, iters)
        forced = {}
        if D <= 128:
            with config.override(dkdv_wide=0):
                forced["128-key"] = _time(run, iters)
        for ns in (1, 2) + ((3, 4, 6, 8) if S >= 1024 else ()):
            with config.override(dkdv_wide=1, dkdv_nsplit=ns):
                forced[f"{'256' if D <= 128 else '128'}-key ns{ns}"] = _time(run, iters)
        if D == 128 and causal and S == Sk and S % 512 == 0:             # the balanced causal schedule (round 6), where eligible
            with config.override(dkdv_wide=2):
                forced["balanced"] = _time(run, iters)
        chosen = min(chosen, _time(run, iters))
        bname = min(forced, key=forced.get)
        if dump is not None:
            dump.append(dict(dir="bwd", label=label, B=B, S=S, Sk=Sk, H=H, Hk=Hk, D=D, causal=causal, chosen=chosen, forced=forced))
        rows.append(("bwd", label, chosen, forced[bname], bname))
        log(f"| {label} | {chosen:.4f} | {forced[bname]:.4f} | {bname} | {chosen / forced[bname]:.3f} |")
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--tol", type=float, default=1.05)
    ap.add_argument("--dump", default=None, help="write every forced timing as JSON (calibration of the plan rules)")
    args = ap.parse_args()
    dump = [] if args.dump else None
    rows = sweep(args.quick, dump=dump)
    if args.dump:
        import json

        with open(args.dump, "w") as fh:
            json.dump(dump, fh, indent=1)
    bad = [(d, l, c / b) for d, l, c, b, _ in rows if c > args.tol * b + 0.003]
    print(f"\n{len(rows)} points, {len(bad)} with chosen / best > {args.tol} (+ 3 us)")
    for d, l, r in bad:
        print(f"  {d} {l}: {r:.3f}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
