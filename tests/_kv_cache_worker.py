"""Worker of test_schedules_cpu.py::test_kv_cache_follows_the_autograd_state: two gloo ranks, oracle backend"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(rank, W, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RFA_ZIGZAG_EXCHANGE="gather")
    dist.init_process_group("gloo", rank=rank, world_size=W)
    import ring_flash_attn as R
    from ring_flash_attn import backend, zigzag_ring_flash_attn as Z
    from oracle.oracle_backend import OracleBackend

    backend.set_backend(OracleBackend())
    torch.manual_seed(rank)
    q = torch.randn(1, 64, 2, 32).bfloat16().requires_grad_(True)
    kv = torch.randn(1, 64, 2, 2, 32).bfloat16().requires_grad_(True)
    counts = []
    out = R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True)
    counts.append(len(Z._KV_CACHE))                 # kept for the backward
    out.sum().backward()
    counts.append(len(Z._KV_CACHE))                 # consumed
    with torch.no_grad():
        R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True)
    counts.append(len(Z._KV_CACHE))                 # inference keeps nothing
    R.zigzag_ring_flash_attn_kvpacked_func(q.detach(), kv.detach(), causal=True)
    counts.append(len(Z._KV_CACHE))                 # no input needs a gradient
    ret[rank] = counts
    dist.barrier()
    dist.destroy_process_group()
