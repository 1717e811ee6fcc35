"""Worker of test_schedules_cpu.py::test_collective_sequence_survives_rank_local_state: gloo ranks, oracle backend.

The sequence of collectives a schedule posts must be a function of group-consistent state only:
the zigzag gather form's backward skips its all-gather when the forward KEPT the gathered K/V — a decision that reads a
process-local byte budget — and `exchange_mode` picks the exchange form from a process-local tuning record.  Each
scenario below makes the ranks' local state DISAGREE; the call must still finish (no rank waits in a collective its
peers never join — the test's timeout is the detector) and produce the gradients of the symmetric run."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(rank, W, port, ret, form="gather"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("RFA_ZIGZAG_EXCHANGE", None)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    import ring_flash_attn as R
    from ring_flash_attn import backend, config, tuning, zigzag_ring_flash_attn as Z
    from ring_flash_attn import _testing
    from oracle.oracle_backend import OracleBackend

    _testing.set_backend(OracleBackend())
    torch.manual_seed(rank)
    q = torch.randn(1, 64, 4, 32).bfloat16().requires_grad_(True)
    kv = torch.randn(1, 64, 2, 2, 32).bfloat16().requires_grad_(True)
    do = torch.randn(1, 64, 4, 32).bfloat16()
    posted = []                                  # the collectives this rank posted, in order
    orig_gather, orig_a2a = Z._gather_kv, Z.all_to_all_async
    orig_ring = Z.RingComm.send_recv_kv

    def gather_kv(*a, **kw):
        posted.append("all_gather")
        return orig_gather(*a, **kw)

    def a2a(*a, **kw):
        posted.append("all_to_all")
        return orig_a2a(*a, **kw)

    def ring_hop(self, *a, **kw):
        posted.append("hop")
        return orig_ring(self, *a, **kw)

    Z._gather_kv, Z.all_to_all_async, Z.RingComm.send_recv_kv = gather_kv, a2a, ring_hop

    def fwd_bwd(q_=q, kv_=kv):
        q_.grad = kv_.grad = None
        del posted[:]
        out = R.zigzag_ring_flash_attn_kvpacked_func(q_, kv_, causal=True)
        n_saved = len(out.grad_fn.saved_tensors)
        out.backward(do)
        return dict(out=out.detach().clone(), dq=q_.grad.clone(), dkv=kv_.grad.clone(), posted=list(posted), n_saved=n_saved)

    res = {}
    with config.override(zigzag_exchange=form):          # gather (one collective) or gather_ps (per-source arrival): the same
        both_keep = fwd_bwd()                             # K/V hand-over logic and collective sequence (`_gather_kv` is counted)
        with config.override(kv_keep=False):
            none_keep = fwd_bwd()
        # (1) ONE rank's budget of kept bytes is exhausted (an output held on that rank only, say): it cannot keep,
        #     its peer can.  Every rank must gather again in the backward.
        for loser in range(W):
            with config.override(kv_keep_total_bytes=0 if rank == loser else config.get().kv_keep_total_bytes):
                res[f"budget_rank{loser}"] = fwd_bwd()
        # (2) the budget is exhausted by a live graph on one rank only — the realistic form of (1): rank 0 keeps an
        #     earlier output alive (its kept K/V with it), rank 1 dropped it
        one = W * kv.numel() * kv.element_size()
        with config.override(kv_keep_total_bytes=one + one // 2):
            held = R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True)
            if rank != 0:
                del held
            res["held_graph"] = fwd_bwd()
            held = None
    assert config.kept_budget.live == 0, config.kept_budget.live
    # (3) a tuning record on some ranks only / different records: ignored by every rank (shape rule: gather); the
    #     same record on every rank: used
    qs, ks = tuple(q.shape), (1, 64, 2, 32)
    if rank == 0:
        tuning.record(qs, ks, q.dtype, W, "ring")
    res["record_rank0_only"] = fwd_bwd()
    res["record_rank0_only"]["mine_after"] = tuning.lookup(qs, ks, q.dtype, W)
    tuning.clear()
    tuning.record(qs, ks, q.dtype, W, "ring" if rank == 0 else "gather")
    res["records_differ"] = fwd_bwd()
    tuning.clear()
    tuning.record(qs, ks, q.dtype, W, "ring")
    res["record_everywhere"] = fwd_bwd()
    # (4) ADVICE r5: the group HAS agreed on the key (the call above); now ONE rank installs a record late (a tuning file
    #     it loads lazily).  That rank-local act must not change which collectives the rank posts: the agreed record
    #     keeps deciding (no agreement all-reduce that only rank 0 would join — the timeout is the detector) ...
    if rank == 0:
        tuning.record(qs, ks, q.dtype, W, "gather")
    res["late_record_rank0"] = fwd_bwd()
    res["late_record_rank0"]["mine_after"] = tuning.lookup(qs, ks, q.dtype, W)
    #     ... until every rank calls sync_records(): the parked record is installed, the group re-agrees at its next use,
    #     finds the ranks' records differ and falls back to the shape rule (gather) everywhere
    tuning.sync_records(None)
    res["after_sync_records"] = fwd_bwd()
    res["after_sync_records"]["mine_after"] = tuning.lookup(qs, ks, q.dtype, W)
    tuning.clear()
    # (5) ADVICE r5: a hand-installed record on ONE rank and then the collective measurement: no rank may return early
    #     (there is no report to return — it was a KeyError — and its peers would wait inside the measurement)
    if rank == 0:
        tuning.record(qs, ks, q.dtype, W, "ring")
    k_, v_ = kv.detach()[:, :, 0], kv.detach()[:, :, 1]
    rep = tuning.autotune_zigzag_exchange(None, q.detach(), k_, v_, iters=1, warm=0)
    rep2 = tuning.autotune_zigzag_exchange(None, q.detach(), k_, v_, iters=1, warm=0)      # now agreed + reported: returned as is
    res["autotune_after_local_record"] = fwd_bwd()
    res["autotune_after_local_record"]["mine_after"] = (rep["chosen"], rep2 is rep, tuning.lookup(qs, ks, q.dtype, W))
    tuning.clear()

    def same(a, b):
        return all(torch.equal(a[n], b[n]) for n in ("out", "dq", "dkv"))

    def close(a, b):
        return all(float((a[n].float() - b[n].float()).abs().max()) < 2e-2 for n in ("out", "dq", "dkv"))

    ret[rank] = dict(
        both_keep=(both_keep["posted"], both_keep["n_saved"]),
        none_keep=(none_keep["posted"], none_keep["n_saved"]),
        close_keep_vs_not=close(both_keep, none_keep),
        **{name: dict(posted=r["posted"], n_saved=r["n_saved"], same_as_no_keep=same(r, none_keep),
                      same_as_keep=same(r, both_keep), mine_after=r.get("mine_after", "-")) for name, r in res.items()},
    )
    dist.barrier()
    dist.destroy_process_group()
