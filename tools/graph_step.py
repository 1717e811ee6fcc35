#!/usr/bin/env python3
"""Launch-bound steps: eager launches vs ONE captured HIP graph (VERDICT r4 item 5 iii).

    python tools/graph_step.py

For each shape: forward + backward of zigzag_ring_flash_attn_kvpacked_func (single rank) timed as (a) eager calls from the
host, (b) torch.cuda.CUDAGraph replays of the captured step, (c) the kernels alone (the backend calls of one step looped
without autograd / allocation: the device-side floor).  Short steps (B x S of a few thousand tokens) are host-bound in
eager mode — about five 10-100 us kernels plus autograd's own allocations and launches per step — and the graph removes
exactly that; at the headline shape (2 ms of kernels) nothing changes.  Prints a markdown table."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29588")
    dist.init_process_group("gloo", rank=0, world_size=1)
    import ring_flash_attn as R
    from ring_flash_attn.backend import get_backend

    dev = torch.device("cuda:0")
    be = get_backend()
    H, Hk, D = 32, 8, 128
    shapes = [(16, 512), (8, 1024), (4, 2048), (2, 4096), (1, 8192)]
    print("| B x S | kernels only ms | eager ms | graph replay ms | eager / graph | fwd+bwd TFLOP/s eager -> graph |")
    print("|---|---|---|---|---|---|")
    for B, S in shapes:
        torch.manual_seed(0)
        q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
        kv = torch.randn(B, S, 2, Hk, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
        do = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
        flops = 3.5 * 4.0 * B * H * S * S * D / 2

        def step():
            q.grad = None
            kv.grad = None
            R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True).backward(do)

        def timed(fn, n):
            for _ in range(max(10, n // 5)):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        n = max(50, int(400 / max(0.1, (B * S * S) / (8192.0 * 8192.0) * 2.0)))
        n = min(n, 2000)
        t_eager = timed(step, n)
        # kernels only
        k, v = kv.detach()[:, :, 0], kv.detach()[:, :, 1]
        qd = q.detach()
        out, lse = torch.empty_like(qd), torch.empty(B, H, S, device=dev, dtype=torch.float32)
        delta = torch.empty_like(lse)
        dq, dkv = torch.empty_like(qd), torch.empty_like(kv.detach())
        sc = D ** -0.5

        def kernels():
            be.fwd(qd, k, v, softmax_scale=sc, causal=True, out=out, lse=lse)
            be.bwd_preprocess(do, out, delta)
            be.bwd(do, qd, k, v, lse, delta, softmax_scale=sc, causal=True, dq=dq, dk=dkv[:, :, 0], dv=dkv[:, :, 1])

        t_kern = timed(kernels, n)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        q.grad = None
        kv.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True).backward(do)
        t_graph = timed(graph.replay, n)
        print(f"| {B} x {S} | {t_kern:.4f} | {t_eager:.4f} | {t_graph:.4f} | {t_eager / t_graph:.2f} | "
              f"{flops / t_eager / 1e9:.0f} -> {flops / t_graph / 1e9:.0f} |", flush=True)
        del graph
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
