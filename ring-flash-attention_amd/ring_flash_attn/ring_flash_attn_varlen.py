"""Basic ring attention over packed (varlen) sequences.

Same public surface and step semantics as
/root/reference/ring_flash_attn/ring_flash_attn_varlen.py (forward :25-100, backward :103-192,
autograd :195-265, wrappers :268-358).  Every packed sequence is sharded contiguously over the
ring, so the local `cu_seqlens` are identical for q and k at every step.  lse is produced
natively as (nheads, total) — the layout of flash_attn >= 2.7 — so the reference's
flatten/unflatten shims (triton_utils.py) are never needed on this path.
"""
import torch

from . import _C
from .backend import get_backend
from .utils import RingComm, single_rank
from ._common import dropout_arg
from ._api import make_autograd_function, make_varlen_api, _grad_buffers


def ring_flash_attn_varlen_forward(
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
    be = get_backend()
    comm = RingComm(process_group)
    T, H, D = q.shape
    vl = dict(cu_seqlens_q=cu_seqlens, cu_seqlens_k=cu_seqlens, max_seqlen_q=max_seqlen, max_seqlen_k=max_seqlen)

    if single_rank(comm.world_size):
        out = torch.empty_like(q)
        lse = torch.empty((H, T), dtype=torch.float32, device=q.device)
        be.fwd(q, k, v, softmax_scale=softmax_scale, causal=causal, out=out, lse=lse, window=window_size, dropout=dropout_arg(dropout_p, dropout_seed), **vl)
        return out, lse
    assert not dropout_p, "dropout over a multi-rank ring is not supported (as in the reference)"

    out_acc = torch.empty((T, H, D), dtype=torch.float32, device=q.device)
    lse_acc = torch.empty((H, T), dtype=torch.float32, device=q.device)
    first = True
    next_k, next_v = None, None
    for step in range(comm.world_size):
        if step + 1 != comm.world_size:
            next_k, next_v = comm.send_recv_kv(k, v)
        if not causal or step <= comm.rank:
            be.fwd(q, k, v, softmax_scale=softmax_scale, causal=causal and step == 0,
                   out_acc=out_acc, lse_acc=lse_acc, acc_init=first, **vl)
            first = False
        if step + 1 != comm.world_size:
            comm.wait()
            k, v = next_k, next_v

    return be.cast(out_acc, q.dtype), lse_acc


def ring_flash_attn_varlen_backward(
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
        be.bwd(dout, q, k, v, softmax_lse, delta, softmax_scale=softmax_scale, causal=causal,
               dq=dq, dk=dk, dv=dv, deterministic=deterministic, window=window_size, dropout=dropout_arg(dropout_p, dropout_seed), **vl)
        return dq, dk, dv
    assert not dropout_p, "dropout over a multi-rank ring is not supported (as in the reference)"

    dq = None
    dk = torch.empty(k.shape, dtype=torch.float32, device=q.device)
    dv = torch.empty(v.shape, dtype=torch.float32, device=q.device)
    next_dk, next_dv = None, None
    next_k, next_v = None, None
    for step in range(kv_comm.world_size):
        if step + 1 != kv_comm.world_size:
            next_k, next_v = kv_comm.send_recv_kv(k, v)

        if step <= kv_comm.rank or not causal:
            bwd_causal = causal and step == 0
            common = dict(softmax_scale=softmax_scale, causal=bwd_causal, deterministic=deterministic, **vl)
            if dq is None:
                dq = torch.empty((T, H, D), dtype=torch.float32, device=q.device)
                be.bwd(dout, q, k, v, softmax_lse, delta, dq_acc=dq, dk_acc=dk, dv_acc=dv, acc_init=True, **common)
            else:
                part = be.bwd(dout, q, k, v, softmax_lse, delta, dq_acc=dq, dk_acc=dk, dv_acc=dv,
                              phases=_C.BWD_COMPUTE, **common)
                d_kv_comm.wait()
                dk, dv = next_dk, next_dv
                be.bwd(dout, q, k, v, softmax_lse, delta, dq_acc=dq, dk_acc=dk, dv_acc=dv,
                       phases=_C.BWD_REDUCE, partials=part, **common)
        elif step != 0:
            d_kv_comm.wait()
            dk, dv = next_dk, next_dv

        if step + 1 != kv_comm.world_size:
            kv_comm.wait()
            k, v = next_k, next_v

        next_dk, next_dv = d_kv_comm.send_recv_kv(dk, dv)

    d_kv_comm.wait()

    return be.cast(dq, q.dtype), be.cast(next_dk, q.dtype), be.cast(next_dv, q.dtype)


RingFlashAttnVarlenFunc = make_autograd_function(
    "RingFlashAttnVarlenFunc", ring_flash_attn_varlen_forward, ring_flash_attn_varlen_backward, 2)
(
    ring_flash_attn_varlen_func,
    ring_flash_attn_varlen_kvpacked_func,
    ring_flash_attn_varlen_qkvpacked_func,
) = make_varlen_api(RingFlashAttnVarlenFunc, "ring_flash_attn_varlen", ring_flash_attn_varlen_forward, ring_flash_attn_varlen_backward)
