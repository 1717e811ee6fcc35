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
    from ring_flash_attn import _testing
    from oracle.oracle_backend import OracleBackend

    _testing.set_backend(OracleBackend())
    torch.manual_seed(rank)
    q = torch.randn(1, 64, 2, 32).bfloat16().requires_grad_(True)
    kv = torch.randn(1, 64, 2, 2, 32).bfloat16().requires_grad_(True)
    import gc
    import weakref

    def n_saved(out):
        return None if out.grad_fn is None else len(out.grad_fn.saved_tensors)

    counts = []
    out = R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True)
    counts.append(n_saved(out))                     # q, k, v, out, lse + the ONE gathered (packed) kv buffer
    gathered = out.grad_fn.saved_tensors[5]
    assert gathered.shape[0] == W * kv.shape[0]
    ref = weakref.ref(gathered)
    del gathered
    out.sum().backward()
    g_keep = kv.grad.clone()
    del out
    gc.collect()
    counts.append(ref() is None)                    # freed with the graph: nothing lives on after the backward
    with torch.no_grad():
        o2 = R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True)
    counts.append(n_saved(o2))                      # inference: no graph, nothing kept
    o3 = R.zigzag_ring_flash_attn_kvpacked_func(q.detach(), kv.detach(), causal=True)
    counts.append(n_saved(o3))                      # no input needs a gradient
    # over the per-call limit (or config.kv_keep = False): nothing is kept, the backward gathers again, same gradients
    from ring_flash_attn import config

    for field, val in (("kv_keep_bytes", 16), ("kv_keep", False)):
        with config.override(**{field: val}):
            kv.grad = None
            o4 = R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True)
            counts.append(n_saved(o4))
            o4.sum().backward()
            counts.append(bool(torch.equal(kv.grad, g_keep)) or float((kv.grad.float() - g_keep.float()).abs().max()) < 2e-2)
    # the budget over ALL pending backwards of the process (config.kv_keep_total_bytes; ADVICE r3: an L-layer model holds
    # L kept buffers): with room for one buffer, the first forward keeps, a second one — while the first is still
    # pending — does not; the reservation returns when the first backward has run
    assert config.kept_budget.live == 0
    one = W * kv.numel() * kv.element_size()
    with config.override(kv_keep_total_bytes=one + one // 2):
        oa = R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True)
        ob = R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True)
        counts += [n_saved(oa), n_saved(ob), config.kept_budget.live == one]
        oa.sum().backward()
        counts.append(config.kept_budget.live == 0)
        oc = R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True)
        counts.append(n_saved(oc))                  # room again
        del oa, ob, oc
        gc.collect()
        counts.append(config.kept_budget.live == 0)  # a graph that is dropped without a backward returns its share too
    ret[rank] = counts
    dist.barrier()
    dist.destroy_process_group()
