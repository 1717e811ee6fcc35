"""Striped attention (token t lives on rank t mod W; Brandon et al. 2023).

Same public surface and step semantics as /root/reference/ring_flash_attn/stripe_flash_attn.py
(forward :7-101, backward :104-231, wrappers :300-378): always causal; at step s <= rank the block is
an ordinary causal block; at step s > rank the incoming keys are "one token ahead", so the block is
causal on the SHIFTED views q[:, 1:] x k[:, :-1] (:63-93), merged into rows [1:].
Built from the same kernels as the zigzag path: the shifted views are pointer offsets into the same
tensors (no copies), merged by the fused fp32 epilogue; dQ / dK / dV accumulate in fp32 in place.
"""
import torch

from . import _C
from .backend import get_backend
from .utils import RingComm, single_rank
from ._common import dropout_arg
from ._api import make_autograd_function, make_dense_api, _grad_buffers


def stripe_flash_attn_forward(
    process_group,
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    softmax_scale,
    dropout_p=0,
    causal=True,
    window_size=(-1, -1),
    alibi_slopes=None,
    deterministic=False,
    dropout_seed=None,
):
    assert (
        causal
    ), "stripe flash attn only supports causal attention, if not causal, use ring flash attn instead"
    be = get_backend()
    comm = RingComm(process_group)
    B, S, H, D = q.shape

    if single_rank(comm.world_size):
        out = torch.empty_like(q)
        lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
        be.fwd(q, k, v, softmax_scale=softmax_scale, causal=True, out=out, lse=lse, window=window_size, dropout=dropout_arg(dropout_p, dropout_seed))
        return out, lse
    assert not dropout_p, "dropout over a multi-rank ring is not supported (as in the reference)"

    out_acc = torch.empty((B, S, H, D), dtype=torch.float32, device=q.device)
    lse_acc = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
    next_k, next_v = None, None
    for step in range(comm.world_size):
        if step + 1 != comm.world_size:
            next_k, next_v = comm.send_recv_kv(k, v)

        if step <= comm.rank:
            be.fwd(q, k, v, softmax_scale=softmax_scale, causal=True,
                   out_acc=out_acc, lse_acc=lse_acc, acc_init=(step == 0))
        else:
            be.fwd(q[:, 1:], k[:, :-1], v[:, :-1], softmax_scale=softmax_scale, causal=True,
                   out_acc=out_acc[:, 1:], lse_acc=lse_acc[:, :, 1:])

        if step + 1 != comm.world_size:
            comm.wait()
            k, v = next_k, next_v

    return be.cast(out_acc, q.dtype), lse_acc


def stripe_flash_attn_backward(
    process_group,
    dout,
    q,
    k,
    v,
    out,
    softmax_lse,
    softmax_scale,
    dropout_p=0,
    causal=True,
    window_size=(-1, -1),
    alibi_slopes=None,
    deterministic=False,
    dropout_seed=None,
    out_grads=None,
):
    assert (
        causal
    ), "stripe flash attn only supports causal attention, if not causal, ring flash attn instead"
    be = get_backend()
    kv_comm = RingComm(process_group)
    d_kv_comm = RingComm(process_group)
    B, S, H, D = q.shape
    if not softmax_lse.is_contiguous():
        softmax_lse = softmax_lse.contiguous()
    if dout.stride(-1) != 1:
        dout = dout.contiguous()

    delta = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
    be.bwd_preprocess(dout, out, delta)

    if single_rank(kv_comm.world_size):
        dq, dk, dv = _grad_buffers(out_grads, q, k, v)
        be.bwd(dout, q, k, v, softmax_lse, delta, softmax_scale=softmax_scale, causal=True,
               dq=dq, dk=dk, dv=dv, deterministic=deterministic, window=window_size, dropout=dropout_arg(dropout_p, dropout_seed))
        return dq, dk, dv
    assert not dropout_p, "dropout over a multi-rank ring is not supported (as in the reference)"

    dq = torch.empty((B, S, H, D), dtype=torch.float32, device=q.device)
    dk = torch.empty(k.shape, dtype=torch.float32, device=q.device)
    dv = torch.empty(v.shape, dtype=torch.float32, device=q.device)
    next_dk, next_dv = None, None
    next_k, next_v = None, None
    dk_comm_buffer, dv_comm_buffer = None, None

    for step in range(kv_comm.world_size):
        if step + 1 != kv_comm.world_size:
            next_k, next_v = kv_comm.send_recv_kv(k, v)

        shift_causal = step > kv_comm.rank
        if shift_causal:
            args = (dout[:, 1:], q[:, 1:], k[:, :-1], v[:, :-1], softmax_lse[:, :, 1:], delta[:, :, 1:])
            dq_view = dq[:, 1:]
        else:
            args = (dout, q, k, v, softmax_lse, delta)
            dq_view = dq
        common = dict(softmax_scale=softmax_scale, causal=True, deterministic=deterministic)

        if step == 0:
            be.bwd(*args, dq_acc=dq, dk_acc=dk, dv_acc=dv, acc_init=True, **common)
        else:
            # dQ (+= fp32) and per-head dK/dV partials while the dk/dv accumulators are in flight
            part = be.bwd(*args, dq_acc=dq_view, dk_acc=dk, dv_acc=dv, phases=_C.BWD_COMPUTE, **common)
            d_kv_comm.wait()
            dk_comm_buffer, dv_comm_buffer = dk, dv
            dk, dv = next_dk, next_dv
            if shift_causal:
                be.bwd(*args, dq_acc=dq_view, dk_acc=dk[:, :-1], dv_acc=dv[:, :-1], phases=_C.BWD_REDUCE,
                       partials=part, **common)
            else:
                be.bwd(*args, dq_acc=dq_view, dk_acc=dk, dv_acc=dv, phases=_C.BWD_REDUCE, partials=part, **common)

        if step + 1 != kv_comm.world_size:
            kv_comm.wait()
            k, v = next_k, next_v

        next_dk, next_dv = d_kv_comm.send_recv_kv(dk, dv, dk_comm_buffer, dv_comm_buffer)

    d_kv_comm.wait()

    return be.cast(dq, q.dtype), be.cast(next_dk, q.dtype), be.cast(next_dv, q.dtype)


StripeFlashAttnFunc = make_autograd_function(
    "StripeFlashAttnFunc", stripe_flash_attn_forward, stripe_flash_attn_backward, 0)
(
    stripe_flash_attn_func,
    stripe_flash_attn_kvpacked_func,
    stripe_flash_attn_qkvpacked_func,
) = make_dense_api(StripeFlashAttnFunc, "stripe_flash_attn", stripe_flash_attn_forward, stripe_flash_attn_backward)
