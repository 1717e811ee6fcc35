"""Basic ring attention (contiguous sequence shard per rank).

Same public surface and step semantics as /root/reference/ring_flash_attn/ring_flash_attn.py
(forward :7-67, backward :70-154, autograd :157-220, wrappers :223-301): at step s rank r holds
the K/V of rank (r-s) mod W; with `causal` only steps s <= r compute and only step 0 is masked.
See zigzag_ring_flash_attn.py for the MI355X-first changes (fused fp32 merge / accumulate,
strided views, two-phase backward, world_size==1 short-circuit).  One deliberate fix: dq is
returned in q.dtype (the reference hard-codes bfloat16 at ring_flash_attn.py:154).
"""
import torch

from . import _C
from .backend import get_backend
from .utils import RingComm, single_rank
from ._common import dropout_arg
from ._api import make_autograd_function, make_dense_api, _grad_buffers


def ring_flash_attn_forward(
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
    be = get_backend()
    comm = RingComm(process_group)
    B, S, H, D = q.shape

    if single_rank(comm.world_size):
        out = torch.empty_like(q)
        lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
        be.fwd(q, k, v, softmax_scale=softmax_scale, causal=causal, out=out, lse=lse, window=window_size, dropout=dropout_arg(dropout_p, dropout_seed))
        return out, lse
    assert not dropout_p, "dropout over a multi-rank ring is not supported (as in the reference)"

    out_acc = torch.empty((B, S, H, D), dtype=torch.float32, device=q.device)
    lse_acc = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
    first = True
    next_k, next_v = None, None

    for step in range(comm.world_size):
        if step + 1 != comm.world_size:
            next_k, next_v = comm.send_recv_kv(k, v)

        if not causal or step <= comm.rank:
            be.fwd(q, k, v, softmax_scale=softmax_scale, causal=causal and step == 0,
                   out_acc=out_acc, lse_acc=lse_acc, acc_init=first)
            first = False

        if step + 1 != comm.world_size:
            comm.wait()
            k, v = next_k, next_v

    out = be.cast(out_acc, q.dtype)
    return out, lse_acc


def ring_flash_attn_backward(
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
        be.bwd(dout, q, k, v, softmax_lse, delta, softmax_scale=softmax_scale, causal=causal,
               dq=dq, dk=dk, dv=dv, deterministic=deterministic, window=window_size, dropout=dropout_arg(dropout_p, dropout_seed))
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
            if dq is None:
                dq = torch.empty((B, S, H, D), dtype=torch.float32, device=q.device)
                be.bwd(dout, q, k, v, softmax_lse, delta, softmax_scale=softmax_scale, causal=bwd_causal,
                       dq_acc=dq, dk_acc=dk, dv_acc=dv, acc_init=True, deterministic=deterministic)
            else:
                part = be.bwd(dout, q, k, v, softmax_lse, delta, softmax_scale=softmax_scale, causal=bwd_causal,
                              dq_acc=dq, dk_acc=dk, dv_acc=dv, deterministic=deterministic,
                              phases=_C.BWD_COMPUTE)
                d_kv_comm.wait()
                dk, dv = next_dk, next_dv
                be.bwd(dout, q, k, v, softmax_lse, delta, softmax_scale=softmax_scale, causal=bwd_causal,
                       dq_acc=dq, dk_acc=dk, dv_acc=dv, deterministic=deterministic,
                       phases=_C.BWD_REDUCE, partials=part)
        elif step != 0:
            d_kv_comm.wait()
            dk, dv = next_dk, next_dv

        if step + 1 != kv_comm.world_size:
            kv_comm.wait()
            k, v = next_k, next_v

        next_dk, next_dv = d_kv_comm.send_recv_kv(dk, dv)

    d_kv_comm.wait()

    return be.cast(dq, q.dtype), be.cast(next_dk, q.dtype), be.cast(next_dv, q.dtype)


RingFlashAttnFunc = make_autograd_function(
    "RingFlashAttnFunc", ring_flash_attn_forward, ring_flash_attn_backward, 0)
(
    ring_flash_attn_func,
    ring_flash_attn_kvpacked_func,
    ring_flash_attn_qkvpacked_func,
) = make_dense_api(RingFlashAttnFunc, "ring_flash_attn", ring_flash_attn_forward, ring_flash_attn_backward)
