#!/usr/bin/env python3
"""Is a CACHE-RESIDENT dS hand-off cheaper — in time or in joules?  

The 5-GEMM backward moves 2.16 GB of dS through HBM (dkdv_kernel stores it, dq_ds_kernel streams it back).  If the
hand-off is cut into chunks of n query heads (67.4 MB of dS per head at S = 8192, causal) and every chunk re-uses ONE
scratch of n x 67.4 MB, the reader's loads can hit the 256 MiB Infinity Cache instead of HBM.  Smaller chunks also mean
smaller grids, so the residency effect is isolated by an A/B of the SAME launches:

    reuse    every chunk hands over through the same n x 67.4 MB buffer          (cache-resident candidate)
    rotate   chunk c hands over through its own region of the full 2.16 GB       (same grids, HBM round trip)

and, so that small chunks do not starve the chip, with the chunks dealt round-robin to `streams` HIP streams (each stream
its own scratch: live dS = streams x n x 67.4 MB).  Per row: ms per backward (all 32 heads), board watts, joules.
The cache policy of the hand-off is a compile-time choice (nt stores + nt loads in the product): run the script again with
RFA_LIB_PATH=build/variants/defpol/librfa_hip.so (tools/ab_variants.py defpol:-DRFA_SPILL_AUX=0,-DRFA_DQS_NT=0).

    python tools/handoff_residency.py [--seconds 2.5] [--rows 32x1,4x1,4x2,2x1,2x2,2x4,1x1,1x2,1x4]      (rows: heads x streams)
Sub-head chunks (n < G) write their dK / dV shares over each other (timing only; energy-equivalent)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--rows", default="32x1,8x1,4x1,4x2,2x1,2x2,2x4,1x1,1x2,1x4")
    args = ap.parse_args()
    import torch

    from power_probe import Sampler
    from ring_flash_attn.backend import get_backend

    be, dev = get_backend(), torch.device("cuda:0")
    S, H, Hk, D = 8192, 32, 8, 128
    G = H // Hk
    torch.manual_seed(0)
    q, k, v, do = (torch.randn(1, S, h_, D, device=dev, dtype=torch.bfloat16) for h_ in (H, Hk, Hk, H))
    out, lse = torch.empty_like(q), torch.empty(1, H, S, device=dev, dtype=torch.float32)
    delta = torch.empty_like(lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    sc = D ** -0.5
    be.fwd(q, k, v, softmax_scale=sc, causal=True, out=out, lse=lse)
    be.bwd_preprocess(do, out, delta)
    nb = S // 32
    per_head = nb * (nb + 1) // 2 * 2048
    full = torch.empty(H * per_head, dtype=torch.uint8, device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    lib = os.environ.get("RFA_LIB_PATH", "in-tree (nt stores, nt loads)")
    print(f"# library: {lib}; dS per query head {per_head / 1e6:.1f} MB; idle floor etc.: profiles/r06_power_limiters.md")
    print("| query heads per chunk | streams | live dS MB | scratch | ms per backward | TFLOP/s | W | J per backward |")
    print("|---|---|---|---|---|---|---|---|")
    smp = Sampler()
    smp.start()
    f = 2.5 * 4.0 * H * S * S * D / 2

    def chunks(n):
        if n >= G:
            return [(h0 // G, n // G, h0, n) for h0 in range(0, H, n)]                  # (first kv head, kv heads, first q head, q heads)
        return [(h0 // G, 1, h0, n) for h0 in range(0, H, n)]

    for row in args.rows.split(","):
        n, ns = (int(x) for x in row.split("x"))
        cl = chunks(n)
        for mode in ("reuse", "rotate"):
            if n == H and mode == "rotate":
                continue

            def backward():
                cur = torch.cuda.current_stream(dev)
                for s_ in streams[:ns]:
                    s_.wait_stream(cur)
                for i, (hk0, nhk, h0, nh) in enumerate(cl):
                    st = streams[i % ns] if ns > 1 else cur
                    if mode == "reuse":
                        scratch = full[(i % ns) * n * per_head:(i % ns + 1) * n * per_head]
                    else:
                        scratch = full[h0 * per_head:(h0 + nh) * per_head]
                    with torch.cuda.stream(st):
                        be.bwd(do[:, :, h0:h0 + nh], q[:, :, h0:h0 + nh], k[:, :, hk0:hk0 + nhk], v[:, :, hk0:hk0 + nhk],
                               lse[:, h0:h0 + nh], delta[:, h0:h0 + nh], softmax_scale=sc, causal=True,
                               dq=dq[:, :, h0:h0 + nh], dk=dk[:, :, hk0:hk0 + nhk], dv=dv[:, :, hk0:hk0 + nhk], ds_scratch=scratch)
                for s_ in streams[:ns]:
                    cur.wait_stream(s_)

            for _ in range(5):
                backward()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                backward()
            torch.cuda.synchronize()
            per = (time.perf_counter() - t0) / 5
            it = max(10, int(args.seconds / per))
            t0 = time.perf_counter()
            for _ in range(it):
                backward()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            pw, _ = smp.window(t0 + 0.3 * (t1 - t0), t1)
            ms = (t1 - t0) / it * 1e3
            w = pw["avg_w"]
            print(f"| {n} | {ns} | {ns * n * per_head / 1e6:.0f} | {mode} | {ms:.4f} | {f / ms / 1e9:.0f} | {w:.0f} | {w * ms * 1e-3:.3f} |"
                  if w else f"| {n} | {ns} | {ns * n * per_head / 1e6:.0f} | {mode} | {ms:.4f} | {f / ms / 1e9:.0f} | - | - |", flush=True)
            time.sleep(0.3)
    smp.stop()


if __name__ == "__main__":
    main()
