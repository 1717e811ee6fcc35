"""`flash_attn.flash_attn_interface` on MI355X: the operator interface the reference imports.

The reference reaches its arithmetic through four PRIVATE functions of the CUDA-only `flash_attn`
package (/root/reference/ring_flash_attn/ring_flash_attn.py:3, zigzag_ring_flash_attn.py:3,
ring_flash_attn_varlen.py:3-6, zigzag_ring_flash_attn_varlen.py:2-5, llama3_flash_attn_varlen.py:3-6)
and its tests / benchmarks use the PUBLIC single-device functions as ground truth
(test/test_zigzag_ring_flash_attn_func.py:2, benchmark/benchmark_kvpacked_func.py:1).  This module
provides both sets with the flash_attn >= 2.7 signatures and return conventions, backed by the same
HIP kernels (librfa_hip.so through ring_flash_attn.backend) — so the UNMODIFIED reference schedules,
tests and benchmarks run on an MI355X by putting `ring-flash-attention_amd/shims/` on sys.path
(INTEGRATION.md route B; the directory is opt-in so that importing `ring_flash_attn` never shadows a real
flash_attn install).  There is no CPU path: CPU tensors raise.

Sliding windows (window_size_left / window_size_right) and dropout are supported by the kernels.  Dropout uses
this library's own counter-based mask (include/rfa.h: rfa_fwd_args.dropout_p — flash_attn's private Philox stream is
not reproducible without the package): a forward with dropout_p > 0 draws a seed from torch's default CPU generator
and returns it as `rng_state` = tensor([seed, 0]); the backward must be given that rng_state.  Unsupported features
raise instead of being silently ignored: dropout together with a window, softcap, alibi_slopes, paged KV
(block_table / leftpad_k / seqused_k), return_softmax / S_dmask.
"""
import math
from typing import Optional, Tuple

import torch



def _binding():
    """The kernels' Python binding (ring_flash_attn/_C.py + backend.py of THIS repo).  When the `ring_flash_attn`
    on sys.path is the reference's own package (route B: its schedules, this operator), the binding is loaded
    from its file location under the private package name `_rfa_binding` instead."""
    try:
        from ring_flash_attn.backend import get_backend as gb
        return gb
    except ImportError:
        import importlib
        import os
        import sys
        import types

        if "_rfa_binding" not in sys.modules:
            here = os.path.dirname(os.path.abspath(__file__))
            pkg = types.ModuleType("_rfa_binding")
            pkg.__path__ = [os.path.join(os.path.dirname(os.path.dirname(here)), "ring_flash_attn")]
            sys.modules["_rfa_binding"] = pkg
        return importlib.import_module("_rfa_binding.backend").get_backend


get_backend = _binding()

__all__ = [
    "_flash_attn_forward", "_flash_attn_backward", "_flash_attn_varlen_forward", "_flash_attn_varlen_backward",
    "flash_attn_func", "flash_attn_kvpacked_func", "flash_attn_qkvpacked_func",
    "flash_attn_varlen_func", "flash_attn_varlen_kvpacked_func", "flash_attn_varlen_qkvpacked_func",
]


def _reject(dropout_p, window_size_left, window_size_right, softcap, alibi_slopes, causal, **paged):
    if dropout_p and (window_size_left is not None and window_size_left >= 0 or
                      (window_size_right is not None and window_size_right >= 0 and not causal)):
        raise NotImplementedError("flash_attn (rfa): dropout together with a window is not supported")
    if softcap:
        raise NotImplementedError("flash_attn (rfa): softcap is not supported")
    if alibi_slopes is not None:
        raise NotImplementedError("flash_attn (rfa): alibi_slopes is not supported")
    for name, val in paged.items():
        if val is not None:
            raise NotImplementedError(f"flash_attn (rfa): {name} (paged / left-padded KV) is not supported")


def _win(window_size_left, window_size_right):
    wl = -1 if window_size_left is None else int(window_size_left)
    wr = -1 if window_size_right is None else int(window_size_right)
    return (wl, wr)


def _dropout(dropout_p, rng_state, forward):
    """(backend dropout argument, rng_state to return)"""
    if not dropout_p:
        return None, None
    if rng_state is None:
        if not forward:
            raise ValueError("flash_attn (rfa): the backward of a dropout call needs the forward's rng_state")
        rng_state = torch.tensor([int(torch.randint(0, 2 ** 62, (1,)).item()), 0], dtype=torch.int64)
    return (float(dropout_p), int(rng_state[0]), 0, 0, 0), rng_state


def _unit_last(t: torch.Tensor) -> torch.Tensor:
    return t if t.stride(-1) == 1 else t.contiguous()


# ------------------------------------------------------------------------------ private operator API
def _flash_attn_forward(q, k, v, dropout_p, softmax_scale, causal, window_size_left=-1, window_size_right=-1,
                        softcap=0.0, alibi_slopes=None, return_softmax=False, *, rng_state=None
                        ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
    """q (B,Sq,H,D), k/v (B,Sk,Hk,D) -> (out (B,Sq,H,D) q.dtype, softmax_lse (B,H,Sq) fp32, None, rng_state)."""
    _reject(dropout_p, window_size_left, window_size_right, softcap, alibi_slopes, causal)
    drop, rng_state = _dropout(dropout_p, rng_state, True)
    q, k, v = _unit_last(q), _unit_last(k), _unit_last(v)
    B, Sq, H, _ = q.shape
    out = torch.empty_like(q, memory_format=torch.contiguous_format)
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    get_backend().fwd(q, k, v, softmax_scale=softmax_scale, causal=causal, out=out, lse=lse,
                      window=_win(window_size_left, window_size_right), dropout=drop)
    return out, lse, None, rng_state


def _flash_attn_backward(dout, q, k, v, out, softmax_lse, dq, dk, dv, dropout_p, softmax_scale, causal,
                         window_size_left=-1, window_size_right=-1, softcap=0.0, alibi_slopes=None,
                         deterministic=False, rng_state=None) -> torch.Tensor:
    """Writes dq/dk/dv IN PLACE (caller views allowed, e.g. `dq_buffer[:, :seqlen_q]`,
    zigzag_ring_flash_attn.py:137-139); returns softmax_d = rowsum(dout*out), (B,H,Sq) fp32."""
    _reject(dropout_p, window_size_left, window_size_right, softcap, alibi_slopes, causal)
    drop = _dropout(dropout_p, rng_state, False)[0]
    be = get_backend()
    dout, q, k, v, out = (_unit_last(t) for t in (dout, q, k, v, out))
    B, Sq, H, _ = q.shape
    delta = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    be.bwd_preprocess(dout, out, delta)
    be.bwd(dout, q, k, v, softmax_lse, delta, softmax_scale=softmax_scale, causal=causal,
           dq=dq, dk=dk, dv=dv, deterministic=deterministic, window=_win(window_size_left, window_size_right),
           dropout=drop)
    return delta


def _flash_attn_varlen_forward(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p,
                               softmax_scale, causal, window_size_left=-1, window_size_right=-1, softcap=0.0,
                               alibi_slopes=None, return_softmax=False, block_table=None, leftpad_k=None,
                               seqused_k=None, zero_tensors=False, *, rng_state=None):
    """q (T,H,D), k/v (Tk,Hk,D), int32 cu_seqlens -> (out (T,H,D), softmax_lse (H,T) fp32, None, rng_state)."""
    _reject(dropout_p, window_size_left, window_size_right, softcap, alibi_slopes, causal,
            block_table=block_table, leftpad_k=leftpad_k, seqused_k=seqused_k)
    drop, rng_state = _dropout(dropout_p, rng_state, True)
    q, k, v = _unit_last(q), _unit_last(k), _unit_last(v)
    T, H, _ = q.shape
    alloc = torch.zeros if zero_tensors else torch.empty
    out = alloc(q.shape, dtype=q.dtype, device=q.device)
    lse = alloc((H, T), dtype=torch.float32, device=q.device)
    get_backend().fwd(q, k, v, softmax_scale=softmax_scale, causal=causal,
                      cu_seqlens_q=cu_seqlens_q.int(), cu_seqlens_k=cu_seqlens_k.int(),
                      max_seqlen_q=max_seqlen_q, max_seqlen_k=max_seqlen_k, out=out, lse=lse,
                      window=_win(window_size_left, window_size_right), dropout=drop)
    return out, lse, None, rng_state


def _flash_attn_varlen_backward(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k,
                                max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, causal,
                                window_size_left=-1, window_size_right=-1, softcap=0.0, alibi_slopes=None,
                                deterministic=False, rng_state=None, zero_tensors=False) -> torch.Tensor:
    """Varlen twin of _flash_attn_backward; softmax_lse and the returned softmax_d are (H,T) fp32."""
    _reject(dropout_p, window_size_left, window_size_right, softcap, alibi_slopes, causal)
    drop = _dropout(dropout_p, rng_state, False)[0]
    be = get_backend()
    dout, q, k, v, out = (_unit_last(t) for t in (dout, q, k, v, out))
    T, H, _ = q.shape
    cq, ck = cu_seqlens_q.int(), cu_seqlens_k.int()
    delta = torch.empty((H, T), dtype=torch.float32, device=q.device)
    be.bwd_preprocess(dout, out, delta, cu_seqlens_q=cq, max_seqlen_q=max_seqlen_q)
    if zero_tensors:
        dq.zero_(), dk.zero_(), dv.zero_()
    be.bwd(dout, q, k, v, softmax_lse, delta, softmax_scale=softmax_scale, causal=causal,
           cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=max_seqlen_q, max_seqlen_k=max_seqlen_k,
           dq=dq, dk=dk, dv=dv, deterministic=deterministic, window=_win(window_size_left, window_size_right),
           dropout=drop)
    return delta


# ------------------------------------------------------------------------------ public single-device API
class _FlashAttnFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, softmax_scale, causal, deterministic, return_lse, cu_q, cu_k, max_q, max_k, wl, wr,
                dropout_p=0.0):
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        if cu_q is None:
            out, lse, _, rng = _flash_attn_forward(q, k, v, dropout_p, softmax_scale, causal, wl, wr)
        else:
            out, lse, _, rng = _flash_attn_varlen_forward(q, k, v, cu_q, cu_k, max_q, max_k, dropout_p, softmax_scale,
                                                          causal, wl, wr)
        ctx.save_for_backward(q, k, v, out, lse, cu_q, cu_k)
        ctx.dropout = (dropout_p, rng)
        ctx.args = (softmax_scale, causal, deterministic, max_q, max_k)
        ctx.window = (wl, wr)
        ctx.mark_non_differentiable(lse)
        return out, lse

    @staticmethod
    def backward(ctx, dout, _dlse):
        q, k, v, out, lse, cu_q, cu_k = ctx.saved_tensors
        softmax_scale, causal, deterministic, max_q, max_k = ctx.args
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        p_drop, rng = ctx.dropout
        if cu_q is None:
            _flash_attn_backward(dout, q, k, v, out, lse, dq, dk, dv, p_drop, softmax_scale, causal, *ctx.window,
                                 deterministic=deterministic, rng_state=rng)
        else:
            _flash_attn_varlen_backward(dout, q, k, v, out, lse, dq, dk, dv, cu_q, cu_k, max_q, max_k, p_drop,
                                        softmax_scale, causal, *ctx.window, deterministic=deterministic, rng_state=rng)
        return (dq, dk, dv) + (None,) * 11


def _public(q, k, v, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic,
            return_attn_probs, cu_q=None, cu_k=None, max_q=None, max_k=None, **paged):
    wl, wr = (window_size if window_size is not None else (-1, -1))
    _reject(dropout_p, wl, wr, softcap, alibi_slopes, causal, **paged)
    out, lse = _FlashAttnFunc.apply(q, k, v, softmax_scale, causal, deterministic, return_attn_probs,
                                    cu_q, cu_k, max_q, max_k, *_win(wl, wr), float(dropout_p or 0.0))
    return (out, lse, None) if return_attn_probs else out


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                    alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """q (B,Sq,H,D), k/v (B,Sk,Hk,D) -> out, or (out, softmax_lse (B,H,Sq), None) with return_attn_probs."""
    return _public(q, k, v, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic,
                   return_attn_probs)


def flash_attn_kvpacked_func(q, kv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                             softcap=0.0, alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """kv (B,Sk,2,Hk,D)."""
    return _public(q, kv[:, :, 0], kv[:, :, 1], dropout_p, softmax_scale, causal, window_size, softcap,
                   alibi_slopes, deterministic, return_attn_probs)


def flash_attn_qkvpacked_func(qkv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                              softcap=0.0, alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """qkv (B,S,3,H,D)."""
    return _public(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], dropout_p, softmax_scale, causal, window_size,
                   softcap, alibi_slopes, deterministic, return_attn_probs)


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                           softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                           deterministic=False, return_attn_probs=False, block_table=None):
    """q (T,H,D), k/v (Tk,Hk,D) packed -> out (T,H,D), or (out, softmax_lse (H,T), None)."""
    return _public(q, k, v, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic,
                   return_attn_probs, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, block_table=block_table)


def flash_attn_varlen_kvpacked_func(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                                    softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                                    alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """kv (Tk,2,Hk,D)."""
    return _public(q, kv[:, 0], kv[:, 1], dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes,
                   deterministic, return_attn_probs, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k)


def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
                                     window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False,
                                     return_attn_probs=False):
    """qkv (T,3,H,D)."""
    return _public(qkv[:, 0], qkv[:, 1], qkv[:, 2], dropout_p, softmax_scale, causal, window_size, softcap,
                   alibi_slopes, deterministic, return_attn_probs, cu_seqlens, cu_seqlens, max_seqlen, max_seqlen)
