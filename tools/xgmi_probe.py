#!/usr/bin/env python3
"""xGMI / RCCL probe for an N-GPU MI355X node: what the zigzag schedules can expect from the links.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/xgmi_probe.py

Rank 0 prints ONE JSON object:
  p2p            uni-directional point-to-point GB/s from rank 0 to every peer (one link each) per message size
  collectives    all-gather / all-to-all / neighbour-hop GB/s per rank (ring_flash_attn.tuning.comm_probe) at the
                 message sizes of the headline schedule (K+V of one rank: 32 MiB at Hk = 8, 128 MiB at Hk = 32)
  contention     the headline forward kernel (256-CU grid) alone, the K/V all-gather alone, and both at once on
                 the compute / side stream: how much each slows the other (RCCL kernels take CUs — SURVEY H3)
  autotune       ring vs gather fwd+bwd of the headline schedule on this node (tuning.autotune_zigzag_exchange)
Runs on one GPU too (N = 1: a one-rank RCCL group; the numbers are then loopback rates).
"""
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    sys.path.insert(0, p)

import torch
import torch.distributed as dist


def main():
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    saved = os.dup(1)
    os.dup2(2, 1)                       # RCCL banners (C stdio, flushed at exit) go to stderr for the whole run;
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # the one JSON line is written to `saved`
    from ring_flash_attn import tuning
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn.utils import comm_stream

    res = {"world": world, "device": torch.cuda.get_device_name(dev)}

    def sync_all():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    # ---- point to point, rank 0 -> peer
    p2p = {}
    for mb in (1, 16, 64, 256):
        buf = torch.zeros(mb << 20, dtype=torch.uint8, device=dev)
        per = {}
        for peer in range(1, world):
            sync_all()
            t0 = time.perf_counter()
            for _ in range(5):
                if rank == 0:
                    dist.send(buf, peer)
                elif rank == peer:
                    dist.recv(buf, 0)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            t = torch.tensor([dt if rank in (0, peer) else 0.0], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            per[str(peer)] = (mb << 20) / t.item() / 1e9
        p2p[f"{mb}MiB"] = per
    res["p2p_GBps_from_rank0"] = p2p

    # ---- the schedule's collectives
    res["collectives"] = {f"{nb >> 20}MiB": tuning.comm_probe(None, dev, nb) for nb in (32 << 20, 128 << 20)}

    # ---- contention between a 256-CU attention grid and an all-gather
    be = get_backend()
    S, H, Hk, D = 8192, 32, 8, 128
    q = torch.randn(1, S, H, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(1, S, Hk, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn_like(k)
    out, lse = torch.empty_like(q), torch.empty(1, H, S, device=dev, dtype=torch.float32)
    kv = torch.stack([k, v], dim=2).contiguous()
    gathered = torch.empty((world,) + tuple(kv.shape), dtype=kv.dtype, device=dev)
    side = comm_stream(dev)

    def attn(n):
        for _ in range(n):
            be.fwd(q, k, v, softmax_scale=D ** -0.5, causal=True, out=out, lse=lse)

    def gather(n):
        with torch.cuda.stream(side):
            for _ in range(n):
                dist.all_gather_into_tensor(gathered, kv)

    def timed(fn):
        fn()
        sync_all()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item() * 1e3

    n = 20
    t_attn, t_gather = timed(lambda: attn(n)) / n, timed(lambda: gather(n)) / n
    t_both = timed(lambda: (gather(n), attn(n))) / n
    res["contention"] = {"attention_fwd_ms": t_attn, "all_gather_32MiB_ms": t_gather, "both_concurrent_ms_per_pair": t_both,
                         "ideal_overlap_ms": max(t_attn, t_gather), "serial_ms": t_attn + t_gather}

    # ---- ring vs gather on the real schedule
    if world > 1:
        res["autotune"] = tuning.autotune_zigzag_exchange(None, q, k, v, iters=3, warm=2)
    if rank == 0:
        os.write(saved, (json.dumps(res) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
