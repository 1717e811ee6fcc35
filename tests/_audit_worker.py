"""Worker of test_schedules_cpu.py::test_exchange_check_names_a_corrupted_receive: W gloo ranks, oracle backend,
config.exchange_check on.  Clean calls pass the audit in every exchange form; then ONE receive buffer of ONE rank is
overwritten after it landed (ring_flash_attn._testing.corrupt_receive — what a wrongly recycled receive buffer looks
like) and that rank's call must raise, naming itself, the step and the buffer, while its peers finish (the audit's
all-gather is the last collective of the call: nobody is left waiting)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(rank, W, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=W)
    import ring_flash_attn as R
    from ring_flash_attn import config, _testing
    from oracle.oracle_backend import OracleBackend

    _testing.set_backend(OracleBackend())
    torch.manual_seed(rank)
    q = torch.randn(1, 64, 4, 32).bfloat16().requires_grad_(True)
    kv = torch.randn(1, 64, 2, 2, 32).bfloat16().requires_grad_(True)
    do = torch.randn(1, 64, 4, 32).bfloat16()

    def fwd_bwd(fn=R.zigzag_ring_flash_attn_kvpacked_func):
        q.grad = kv.grad = None
        out = fn(q, kv, causal=True)
        out.backward(do)
        return out.detach().clone(), q.grad.clone(), kv.grad.clone()

    res = {}
    ref = {}
    for form in ("ring", "gather", "gather_ps"):
        with config.override(zigzag_exchange=form, exchange_check=False):
            ref[form] = fwd_bwd()
        with config.override(zigzag_exchange=form, exchange_check=True):
            got = fwd_bwd()
        res[f"clean_{form}"] = all(torch.equal(a, b) for a, b in zip(got, ref[form]))
    with config.override(exchange_check=True):
        res["clean_ring_func"] = fwd_bwd(R.ring_flash_attn_kvpacked_func) is not None
        res["clean_stripe_func"] = fwd_bwd(R.stripe_flash_attn_kvpacked_func) is not None
    # corruption: the victim's first audited receive is overwritten after it landed.  Forward only: the failing rank leaves
    # the call at the audit (the LAST collective of the forward, which every rank completes); in a real job the raise ends
    # the job, here the peers simply return from their forward
    victim = W - 1
    for form in ("ring", "gather_ps"):
        with config.override(zigzag_exchange=form, exchange_check=True), torch.no_grad():
            if rank == victim:
                _testing.corrupt_receive(0)
            try:
                R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True)
                res[f"corrupt_{form}"] = "no error"
            except RuntimeError as e:
                res[f"corrupt_{form}"] = str(e)
            _testing.corrupt_receive(None)
        dist.barrier()
        # the ranks are in step again: a clean call passes
        with config.override(zigzag_exchange=form, exchange_check=True):
            res[f"after_{form}"] = all(torch.equal(a, b) for a, b in zip(fwd_bwd(), ref[form]))
    ret[rank] = res
    dist.barrier()
    dist.destroy_process_group()
