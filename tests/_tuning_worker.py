"""Worker of test_schedules_cpu.py::test_exchange_autotune_*: gloo ranks, oracle backend"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(rank, W, port, ret, break_gather):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("RFA_ZIGZAG_EXCHANGE", None)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from ring_flash_attn import backend, tuning, zigzag_ring_flash_attn as Z
    from ring_flash_attn import _testing
    from oracle.oracle_backend import OracleBackend

    _testing.set_backend(OracleBackend())
    torch.manual_seed(rank)
    q = torch.randn(1, 64, 4, 32).bfloat16()
    k = torch.randn(1, 64, 2, 32).bfloat16()
    v = torch.randn(1, 64, 2, 32).bfloat16()
    if break_gather:
        # the gather form cannot run on this "node" (its all-to-all raises, on every rank — an unsupported collective;
        # a failure on SOME ranks only would leave the others inside the collective, which no caller can repair):
        # every rank must reach the same decision — ring
        def broken(*a, **kw):
            raise RuntimeError("simulated all_to_all failure")

        Z.all_to_all_async = broken
    before = Z.exchange_mode(k, W, q)
    rep = tuning.autotune_zigzag_exchange(None, q, k, v, iters=2, warm=1)
    after = Z.exchange_mode(k, W, q)
    from ring_flash_attn import config

    with config.override(zigzag_exchange="ring"):
        forced = Z.exchange_mode(k, W, q)
    probe = None if break_gather else tuning.comm_probe(None, torch.device("cpu"), 1 << 16, iters=2, warm=1)
    # the in-call path (config.autotune, opt-in): same measurement, never raises — with EVERY form broken (on all ranks
    # alike) it returns None, leaves the shape rule in charge and does not try again
    tuning.clear()
    in_call = tuning.measure_in_call(None, q, k, v)
    in_call_none = None
    if break_gather:
        def broken_hop(self, *a, **kw):
            raise RuntimeError("simulated isend failure")

        tuning.clear()
        saved = Z.RingComm.send_recv_kv
        Z.RingComm.send_recv_kv = broken_hop
        try:
            in_call_none = [tuning.measure_in_call(None, q, k, v), tuning.measure_in_call(None, q, k, v),
                            Z.exchange_mode(k, W, q)]
        finally:
            Z.RingComm.send_recv_kv = saved
    ret[rank] = dict(before=before, chosen=rep["chosen"], after=after, forced=forced, ms=rep["ms"], failed=rep["failed"],
                     probe=probe, other_shape=Z.exchange_mode(k[:, :32], W, q[:, :32]), in_call=in_call,
                     in_call_none=in_call_none)
    dist.barrier()
    dist.destroy_process_group()
