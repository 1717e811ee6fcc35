#!/usr/bin/env python3
"""Does `utils.Agreement.resolve()` stall the host?  (VERDICT r5 weak #12 / next #9.)

The zigzag gather form agrees, per forward, whether EVERY rank kept its gathered K/V (one 4-byte all-reduce under the side
stream, copied to pinned memory behind an event) and reads the answer at the START of that layer's backward with
`event.synchronize()`.  In a deep model the first backward layer runs right after the last forward layer: if that forward's
flag has not retired yet, the host waits.  This tool runs an L-layer stack of `zigzag_ring_flash_attn_func` (gather form,
K/V kept) forward and backward on a ONE-rank RCCL group forced onto the multi-step path (ring_flash_attn._testing
.force_steps: the real side stream, all-reduce, pinned copy and event; one GPU) and reports, per step, the host time spent
inside resolve() against the step's wall time, per layer position (the first backward layer is the one that can stall).

    python tools/agreement_stall.py [--layers 28] [--tokens 2048] [--heads 16] [--kv-heads 8] [--steps 20]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--tokens", type=int, default=2048)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--kv-heads", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29617")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import ring_flash_attn as R
    from ring_flash_attn import _testing, config, utils

    _testing.force_steps(True)
    config.set(zigzag_exchange="gather")
    spent = []                       # (seconds inside resolve(), had to wait?) per call
    orig = utils.Agreement.resolve

    def timed_resolve(self):
        pending = self._value is None and self._event is not None and not self._event.query()
        t0 = time.perf_counter()
        v = orig(self)
        spent.append((time.perf_counter() - t0, pending))
        return v

    utils.Agreement.resolve = timed_resolve
    L, T, H, Hk, D = args.layers, args.tokens, args.heads, args.kv_heads, 128
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(1, T, H, D, device=dev, dtype=torch.bfloat16, generator=g).requires_grad_(True)
    ks = [torch.randn(1, T, Hk, D, device=dev, dtype=torch.bfloat16, generator=g).requires_grad_(True) for _ in range(L)]
    vs = [torch.randn(1, T, Hk, D, device=dev, dtype=torch.bfloat16, generator=g).requires_grad_(True) for _ in range(L)]

    def step():
        h = x
        for i in range(L):
            h = h + R.zigzag_ring_flash_attn_func(h, ks[i], vs[i], causal=True)
        h.float().sum().backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    del spent[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps
    n = len(spent) // args.steps
    per_pos = [[spent[s * n + i] for s in range(args.steps)] for i in range(n)]
    tot = sum(t for t, _ in spent) / args.steps
    print(f"{L} layers of zigzag_ring_flash_attn_func, q (1, {T}, {H}, {D}), {Hk} K/V heads, gather form, K/V kept; one-rank RCCL group, forced multi-step path")
    print(f"step wall time {wall * 1e3:.3f} ms; resolve() calls per step {n}; host time inside resolve() {tot * 1e6:.1f} us per step = {100 * tot / wall:.3f} % of the step")
    print("| backward layer (0 = first to run = last forward layer) | mean us in resolve() | max us | calls that found the flag still pending |")
    print("|---|---|---|---|")
    for i in (0, 1, 2, n // 2, n - 1):
        if i < n:
            ts = [t for t, _ in per_pos[i]]
            print(f"| {i} | {1e6 * sum(ts) / len(ts):.1f} | {1e6 * max(ts):.1f} | {sum(1 for _, p_ in per_pos[i] if p_)} of {len(ts)} |")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
