"""Zigzag ring attention over packed (varlen) sequences.

Same public surface and step semantics as
/root/reference/ring_flash_attn/zigzag_ring_flash_attn_varlen.py (get_half_index :24-39,
get_half_lse :42-71, forward :74-191, backward :194-317, autograd :320-412, wrappers :415-505):
every packed sequence is split in 2W chunks, rank r holds chunks r and 2W-1-r of each, local
layout per sequence [front half | back half].

MI355X-first change: the reference selects "front half of every sequence" / "back half of every
sequence" with boolean-mask gathers and scatters on every step (built from a CPU mask, i.e. an
H2D copy + nonzero sync each time) and a TorchScript loop with .item() for the lse.  Here the
kernels take `q_half` / `k_half` selectors and do the offset arithmetic themselves
(csrc/rfa_common.hpp: resolve_span): no gathers, no copies, no host syncs; results land
directly in the right rows of the full-size accumulators.

Exchange forms (RFA_ZIGZAG_VARLEN_EXCHANGE = ring | gather, default ring): `ring` is the reference's hop-by-hop
protocol; `gather` is the mesh-aware form of the dense zigzag path (zigzag_ring_flash_attn.py) for packed
sequences — one all-gather of K/V beside the local block, every rank's dK/dV contribution for the rows of rank
c written into slot c (a "front" step fills only the front half of every sequence: the slot is zeroed first),
one all-to-all, fp32 sum of the W arrivals at the owner (rfa_sum_slots).  Same kernels, same step order.
"""
import torch

from . import _C, config
from .backend import get_backend, HALF_FRONT, HALF_BACK
from .utils import AllGatherComm, RingComm, all_to_all_async, single_rank
from ._common import dropout_arg
from ._api import make_autograd_function, make_varlen_api, _grad_buffers


def varlen_exchange_mode() -> str:
    """config.zigzag_varlen_exchange (RFA_ZIGZAG_VARLEN_EXCHANGE): ring | gather"""
    return config.get().zigzag_varlen_exchange


def _gather_kv(process_group, k, v, world):
    comm = AllGatherComm(process_group)
    k_cat = torch.empty((world * k.shape[0],) + tuple(k.shape[1:]), dtype=k.dtype, device=k.device)
    v_cat = torch.empty_like(k_cat)
    comm.all_gather(k_cat, k.contiguous())
    comm.all_gather(v_cat, v.contiguous())
    return comm, k_cat.view((world,) + tuple(k.shape)), v_cat.view((world,) + tuple(v.shape))


def get_half_index(cu_seqlens, *, front: bool):
    """API parity with the reference helper (zigzag_ring_flash_attn_varlen.py:24-39); unused by
    the kernels.  Returns a slice (single sequence) or a boolean row mask."""
    if len(cu_seqlens) == 2:
        if front:
            return slice(None, cu_seqlens[-1] // 2)
        else:
            return slice(cu_seqlens[-1] // 2, None)

    cu = [int(x) for x in cu_seqlens.tolist()]
    index = torch.zeros((cu[-1],), dtype=torch.bool)
    for i in range(len(cu) - 1):
        start, end = cu[i], cu[i + 1]
        if front:
            end = (start + end) // 2
        else:
            start = (start + end) // 2
        index[start:end] = True
    return index


def get_half_lse(lse, cu_seqlens, *, front: bool):
    """API parity with zigzag_ring_flash_attn_varlen.py:42-71 for the (nheads, total) layout."""
    cu = [int(x) for x in cu_seqlens.tolist()]
    new_lse = torch.empty((lse.shape[0], lse.shape[1] // 2), dtype=lse.dtype, device=lse.device)
    for i in range(len(cu) - 1):
        start, end = cu[i], cu[i + 1]
        new_start, new_end = start // 2, end // 2
        if front:
            end -= (end - start) // 2
        else:
            start += (end - start) // 2
        new_lse[:, new_start:new_end] = lse[:, start:end]
    return new_lse


def zigzag_ring_flash_attn_varlen_forward(
    process_group,
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    cu_seqlens,
    max_seqlen,
    softmax_scale,
    dropout_p=0,
    causal=True,
    window_size=(-1, -1),
    alibi_slopes=None,
    deterministic=False,
    dropout_seed=None,
):
    assert causal == True, "zigzag ring is meaningless for causal=False"
    be = get_backend()
    comm = RingComm(process_group)
    T, H, D = q.shape
    vl = dict(cu_seqlens_q=cu_seqlens, cu_seqlens_k=cu_seqlens, max_seqlen_q=max_seqlen, max_seqlen_k=max_seqlen)

    if single_rank(comm.world_size):
        out = torch.empty_like(q)
        lse = torch.empty((H, T), dtype=torch.float32, device=q.device)
        be.fwd(q, k, v, softmax_scale=softmax_scale, causal=True, out=out, lse=lse, window=window_size, dropout=dropout_arg(dropout_p, dropout_seed), **vl)
        return out, lse
    assert not dropout_p, "dropout over a multi-rank ring is not supported (as in the reference)"

    out_acc = torch.empty((T, H, D), dtype=torch.float32, device=q.device)
    lse_acc = torch.empty((H, T), dtype=torch.float32, device=q.device)

    if varlen_exchange_mode() == "gather":
        W, rank = comm.world_size, comm.rank
        gather, k_all, v_all = _gather_kv(process_group, k, v, W)
        be.fwd(q, k, v, softmax_scale=softmax_scale, causal=True,                  # beside the all-gather
               out_acc=out_acc, lse_acc=lse_acc, acc_init=True, **vl)
        gather.wait()
        for step in range(1, W):
            src = (rank - step) % W
            halves = dict(k_half=HALF_FRONT) if step <= rank else dict(q_half=HALF_BACK)
            be.fwd(q, k_all[src], v_all[src], softmax_scale=softmax_scale, causal=False,
                   out_acc=out_acc, lse_acc=lse_acc, **halves, **vl)
        return be.cast(out_acc, q.dtype), lse_acc

    next_k, next_v = None, None
    for step in range(comm.world_size):
        if step + 1 != comm.world_size:
            next_k, next_v = comm.send_recv_kv(k, v)

        if step == 0:
            be.fwd(q, k, v, softmax_scale=softmax_scale, causal=True,
                   out_acc=out_acc, lse_acc=lse_acc, acc_init=True, **vl)
        elif step <= comm.rank:
            be.fwd(q, k, v, softmax_scale=softmax_scale, causal=False, k_half=HALF_FRONT,
                   out_acc=out_acc, lse_acc=lse_acc, **vl)
        else:
            be.fwd(q, k, v, softmax_scale=softmax_scale, causal=False, q_half=HALF_BACK,
                   out_acc=out_acc, lse_acc=lse_acc, **vl)

        if step + 1 != comm.world_size:
            comm.wait()
            k, v = next_k, next_v

    return be.cast(out_acc, q.dtype), lse_acc


def zigzag_ring_flash_attn_varlen_backward(
    process_group,
    dout,
    q,
    k,
    v,
    out,
    softmax_lse,
    cu_seqlens,
    max_seqlen,
    softmax_scale,
    dropout_p=0,
    causal=True,
    window_size=(-1, -1),
    alibi_slopes=None,
    deterministic=False,
    dropout_seed=None,
    out_grads=None,
):
    assert causal == True, "zigzag ring is meaningless for causal=False"
    be = get_backend()
    kv_comm = RingComm(process_group)
    d_kv_comm = RingComm(process_group)
    T, H, D = q.shape
    vl = dict(cu_seqlens_q=cu_seqlens, cu_seqlens_k=cu_seqlens, max_seqlen_q=max_seqlen, max_seqlen_k=max_seqlen)
    if not softmax_lse.is_contiguous():
        softmax_lse = softmax_lse.contiguous()
    if dout.stride(-1) != 1:
        dout = dout.contiguous()

    delta = torch.empty((H, T), dtype=torch.float32, device=q.device)
    be.bwd_preprocess(dout, out, delta, cu_seqlens_q=cu_seqlens, max_seqlen_q=max_seqlen)

    if single_rank(kv_comm.world_size):
        dq, dk, dv = _grad_buffers(out_grads, q, k, v)
        be.bwd(dout, q, k, v, softmax_lse, delta, softmax_scale=softmax_scale, causal=True,
               dq=dq, dk=dk, dv=dv, deterministic=deterministic, window=window_size, dropout=dropout_arg(dropout_p, dropout_seed), **vl)
        return dq, dk, dv
    assert not dropout_p, "dropout over a multi-rank ring is not supported (as in the reference)"

    dq = torch.empty((T, H, D), dtype=torch.float32, device=q.device)
    dk = torch.empty(k.shape, dtype=torch.float32, device=q.device)
    dv = torch.empty(v.shape, dtype=torch.float32, device=q.device)

    if varlen_exchange_mode() == "gather":
        W, rank = kv_comm.world_size, kv_comm.rank
        gather, k_all, v_all = _gather_kv(process_group, k, v, W)
        # slot c: this rank's dK/dV contribution (io dtype) for the rows of rank c
        dk_all = torch.empty((W,) + tuple(k.shape), dtype=q.dtype, device=q.device)
        dv_all = torch.empty_like(dk_all)
        be.bwd(dout, q, k, v, softmax_lse, delta, softmax_scale=softmax_scale, causal=True,       # beside the all-gather
               dq_acc=dq, acc_init=True, dk=dk_all[rank], dv=dv_all[rank], deterministic=deterministic, **vl)
        gather.wait()
        for step in range(1, W):
            src = (rank - step) % W
            if step <= rank:
                halves = dict(k_half=HALF_FRONT)     # only the front half of every sequence receives a contribution
                dk_all[src].zero_()
                dv_all[src].zero_()
            else:
                halves = dict(q_half=HALF_BACK)
            be.bwd(dout, q, k_all[src], v_all[src], softmax_lse, delta, softmax_scale=softmax_scale, causal=False,
                   dq_acc=dq, acc_init=False, dk=dk_all[src], dv=dv_all[src], deterministic=deterministic,
                   **halves, **vl)
        dk_in, dv_in = torch.empty_like(dk_all), torch.empty_like(dv_all)
        works = [all_to_all_async(dk_in.view((-1,) + tuple(k.shape[1:])), dk_all.view((-1,) + tuple(k.shape[1:])),
                                  group=process_group),
                 all_to_all_async(dv_in.view((-1,) + tuple(v.shape[1:])), dv_all.view((-1,) + tuple(v.shape[1:])),
                                  group=process_group)]
        dq_out = be.cast(dq, q.dtype)                                                  # beside the exchange
        for w_ in works:
            w_.wait()
        _, dk_out, dv_out = _grad_buffers(out_grads, None, k, v)
        be.sum_slots(dk_in, dk_out)
        be.sum_slots(dv_in, dv_out)
        return dq_out, dk_out, dv_out

    next_dk, next_dv = None, None
    next_k, next_v = None, None
    dk_comm_buffer, dv_comm_buffer = None, None

    for step in range(kv_comm.world_size):
        if step + 1 != kv_comm.world_size:
            next_k, next_v = kv_comm.send_recv_kv(k, v)

        if step == 0:
            be.bwd(dout, q, k, v, softmax_lse, delta, softmax_scale=softmax_scale, causal=True,
                   dq_acc=dq, dk_acc=dk, dv_acc=dv, acc_init=True, deterministic=deterministic, **vl)
        else:
            if step <= kv_comm.rank:
                halves = dict(k_half=HALF_FRONT)
            else:
                halves = dict(q_half=HALF_BACK)
            common = dict(softmax_scale=softmax_scale, causal=False, deterministic=deterministic, **halves, **vl)
            part = be.bwd(dout, q, k, v, softmax_lse, delta, dq_acc=dq, dk_acc=dk, dv_acc=dv,
                          phases=_C.BWD_COMPUTE, **common)

            d_kv_comm.wait()
            dk_comm_buffer, dv_comm_buffer = dk, dv
            dk, dv = next_dk, next_dv

            be.bwd(dout, q, k, v, softmax_lse, delta, dq_acc=dq, dk_acc=dk, dv_acc=dv,
                   phases=_C.BWD_REDUCE, partials=part, **common)

        if step + 1 != kv_comm.world_size:
            kv_comm.wait()
            k, v = next_k, next_v

        next_dk, next_dv = d_kv_comm.send_recv_kv(dk, dv, dk_comm_buffer, dv_comm_buffer)

    d_kv_comm.wait()

    return be.cast(dq, q.dtype), be.cast(next_dk, q.dtype), be.cast(next_dv, q.dtype)


ZigZagRingFlashAttnVarlenFunc = make_autograd_function(
    "ZigZagRingFlashAttnVarlenFunc", zigzag_ring_flash_attn_varlen_forward,
    zigzag_ring_flash_attn_varlen_backward, 2)
(
    zigzag_ring_flash_attn_varlen_func,
    zigzag_ring_flash_attn_varlen_kvpacked_func,
    zigzag_ring_flash_attn_varlen_qkvpacked_func,
) = make_varlen_api(ZigZagRingFlashAttnVarlenFunc, "zigzag_ring_flash_attn_varlen", zigzag_ring_flash_attn_varlen_forward, zigzag_ring_flash_attn_varlen_backward)
