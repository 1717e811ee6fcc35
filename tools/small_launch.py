#!/usr/bin/env python3
"""The short-launch regime of BASELINE.json configs[4] (llama3_flash_attn_varlen_func through the HF adapter, Qwen3-0.6B
layer shape: 16 query / 8 kv heads, head dim 128) on ONE GPU: rank `--rank` of a `--world`-rank job with the exchange
looped back to local buffers (ring_flash_attn.utils.set_loopback — the exact kernel sequence of that rank, no
communication), `--tokens` packed tokens per rank in one sequence, heads_k_stride as the HF adapter sets it (1) and as
the reference's benchmark does (4), forward and forward + backward.  Prints ms per call and the fraction of the bf16 MFMA
peak (algorithmic FLOPs of that rank's rows: 4 H D (T r T + T^2 / 2) forward, x 3.5 with the backward).
usage: python tools/small_launch.py [--tokens 2048] [--world 8] [--rank 7] [--heads 16] [--kv-heads 8]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=2048)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=7)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--kv-heads", type=int, default=8)
    args = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29588")
    dist.init_process_group("gloo", rank=0, world_size=1)
    import ring_flash_attn as R
    from ring_flash_attn import utils
    from ring_flash_attn import _testing

    dev = torch.device("cuda:0")
    T, W, r, H, Hk, D = args.tokens, args.world, args.rank, args.heads, args.kv_heads, 128
    q = torch.randn(T, H, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
    kv = torch.randn(T, 2, Hk, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn(T, H, D, device=dev, dtype=torch.bfloat16)
    cu = torch.tensor([0, T * W], dtype=torch.int32)
    cq, ck, mq, mk, sl = R.llama3_flash_attn_prepare_cu_seqlens(cu, True, r, W)
    cq, ck = cq.to(dev), ck.to(dev)
    fwd_flops = 4.0 * H * D * (T * r * T + T * T / 2.0)
    _testing.set_loopback((r, W))
    print(f"llama3, rank {r} of {W} (loopback), {T} tokens per rank (one sequence of {T * W}), {H}/{Hk} heads, head dim {D}")
    print("| heads_k_stride | pass | ms | TFLOP/s | of 2.5 PF |")
    print("|---|---|---|---|---|")
    for stride in (1, 4, Hk):
        def call():
            return R.llama3_flash_attn_varlen_kvpacked_func(q, kv, cq, ck, mq, mk, heads_k_stride=stride, local_k_slice=sl, causal=True)

        for name, fl in (("fwd", fwd_flops), ("fwd+bwd", 3.5 * fwd_flops)):
            def step():
                if name == "fwd":
                    with torch.no_grad():
                        call()
                else:
                    q.grad = kv.grad = None
                    call().backward(do)

            for _ in range(10):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 50
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            print(f"| {stride} | {name} | {ms:.3f} | {fl / ms / 1e9:.0f} | {fl / ms / 1e9 / 2500:.3f} |", flush=True)
    _testing.set_loopback(None)


if __name__ == "__main__":
    main()
