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
"""
import torch

from . import _C
from .backend import get_backend, HALF_FRONT, HALF_BACK
from .utils import RingComm, single_rank
from ._api import make_autograd_function, make_varlen_api, _grad_buffers


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
):
    assert causal == True, "zigzag ring is meaningless for causal=False"
    be = get_backend()
    comm = RingComm(process_group)
    T, H, D = q.shape
    vl = dict(cu_seqlens_q=cu_seqlens, cu_seqlens_k=cu_seqlens, max_seqlen_q=max_seqlen, max_seqlen_k=max_seqlen)

    if single_rank(comm.world_size):
        out = torch.empty_like(q)
        lse = torch.empty((H, T), dtype=torch.float32, device=q.device)
        be.fwd(q, k, v, softmax_scale=softmax_scale, causal=True, out=out, lse=lse, window=window_size, **vl)
        return out, lse

    out_acc = torch.empty((T, H, D), dtype=torch.float32, device=q.device)
    lse_acc = torch.empty((H, T), dtype=torch.float32, device=q.device)
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
               dq=dq, dk=dk, dv=dv, deterministic=deterministic, window=window_size, **vl)
        return dq, dk, dv

    dq = torch.empty((T, H, D), dtype=torch.float32, device=q.device)
    dk = torch.empty(k.shape, dtype=torch.float32, device=q.device)
    dv = torch.empty(v.shape, dtype=torch.float32, device=q.device)
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
