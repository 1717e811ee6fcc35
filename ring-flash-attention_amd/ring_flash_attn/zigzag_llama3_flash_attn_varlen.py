"""zigzag_llama3_flash_attn_varlen_func — the entry the reference lists as a TODO (/root/reference/README.md:131).

llama3-style context parallelism (all-gather of K/V, one attention per rank over everything it may see:
/root/reference/ring_flash_attn/llama3_flash_attn_varlen.py) gives rank r the r-th contiguous slice of the packed
token stream, so under a causal mask the last rank does about W times the work of the first.  This variant gives
rank r the TWO slices r and 2W-1-r of the stream cut into 2W equal slices (the zigzag layout of
zigzag_ring_flash_attn, applied to the packed stream instead of to every sequence): an early, cheap slice and a
late, expensive one — every rank does the same work up to sequence-boundary effects.

Layout contract (what the caller shards):
    the global packed stream of T tokens (T % 2W == 0, L = T / 2W) is cut into slices s_0 .. s_{2W-1};
    rank r holds  [s_r, s_{2W-1-r}]  concatenated: q (2L, H, D), k / v (2L, Hk, D); out / lse come back alike.
`cu_seqlens` is the GLOBAL cu_seqlens of the packed stream (as for llama3_flash_attn_prepare_cu_seqlens).

How it runs, with the pieces this package already has:
  * K and V are all-gathered (rank order) and re-ordered once into stream order (two strided copies);
  * each of the two local slices is ONE packed-sequence attention call whose (cu_seqlens_q, cu_seqlens_k, k slice)
    are exactly llama3_flash_attn_prepare_cu_seqlens(cu_seqlens, causal, slice index, 2W) — a slice of a 2W-way
    llama3 split — so forward, backward and masks are the kernels' ordinary varlen paths;
  * backward: the two calls add their dK/dV contributions into fp32 stream-order buffers (`dk_acc +=`), which are
    put back into rank order and reduce-scattered; dQ is written per slice.
Sliding windows work as in llama3 (one kernel sees all keys).  `heads_k_stride` is accepted for symmetry with
llama3_flash_attn_varlen_func; all heads are processed per call.
"""
import torch

from ._api import _check_unsupported, _opaque
from ._common import _as_cu
from .backend import get_backend
from .llama3_flash_attn_varlen import llama3_flash_attn_prepare_cu_seqlens
from .utils import AllGatherComm, group_rank_world, reduce_scatter_async, single_rank

__all__ = [
    "zigzag_llama3_flash_attn_prepare_cu_seqlens",
    "zigzag_llama3_flash_attn_varlen_func",
    "zigzag_llama3_flash_attn_varlen_kvpacked_func",
    "zigzag_llama3_flash_attn_varlen_qkvpacked_func",
]


def zigzag_llama3_flash_attn_prepare_cu_seqlens(cu_seqlens: torch.Tensor, causal: bool, rank: int, world_size: int):
    """The two parameter sets of rank `rank`: ((cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, k_slice) of
    its early slice `rank`, the same for its late slice `2*world_size - 1 - rank`), k_slice indexing the
    stream-ordered K/V."""
    n = 2 * world_size
    return tuple(llama3_flash_attn_prepare_cu_seqlens(cu_seqlens, causal, c, n) for c in (rank, n - 1 - rank))


def _to_stream_order(g: torch.Tensor, world: int) -> torch.Tensor:
    """(world * 2L, ...) in rank order [s_r, s_{2W-1-r}] per rank  ->  (2W * L, ...) in stream order"""
    rows = g.shape[0] // world
    half = rows // 2
    g = g.view((world, rows) + tuple(g.shape[1:]))
    return torch.cat([g[:, :half], g[:, half:].flip(0)], dim=0).reshape((world * rows,) + tuple(g.shape[2:]))


def _to_rank_order(s: torch.Tensor, world: int) -> torch.Tensor:
    """inverse of _to_stream_order"""
    half = s.shape[0] // (2 * world)
    s = s.view((2 * world, half) + tuple(s.shape[1:]))
    return torch.cat([s[:world], s[world:].flip(0)], dim=1).reshape((2 * world * half,) + tuple(s.shape[2:]))


def _gathered_stream_kv(process_group, k, v, world):
    if single_rank(world):
        return k, v                      # slices 0 and 1 of a 2-way split: already in stream order
    comm = AllGatherComm(process_group)
    kg = torch.empty((world * k.shape[0],) + tuple(k.shape[1:]), dtype=k.dtype, device=k.device)
    vg = torch.empty_like(kg)
    comm.all_gather(kg, k)
    comm.all_gather(vg, v)
    comm.wait()
    return _to_stream_order(kg, world), _to_stream_order(vg, world)


def zigzag_llama3_flash_attn_varlen_forward(process_group, q, k, v, params, softmax_scale, causal=True,
                                            window_size=(-1, -1)):
    be = get_backend()
    T, H, _ = q.shape
    L = T // 2
    world = group_rank_world(process_group)[1]
    kg, vg = _gathered_stream_kv(process_group, k, v, world)
    out = torch.empty_like(q)
    lse = torch.empty((H, T), dtype=torch.float32, device=q.device)
    for i, (cu_q, cu_k, mq, mk, ksl) in enumerate(params):
        rows = slice(i * L, (i + 1) * L)
        be.fwd(q[rows], kg[ksl], vg[ksl], softmax_scale=softmax_scale, causal=causal, out=out[rows],
               lse=lse[:, rows], window=window_size, cu_seqlens_q=cu_q, cu_seqlens_k=cu_k, max_seqlen_q=mq,
               max_seqlen_k=mk)
    return out, lse


def zigzag_llama3_flash_attn_varlen_backward(process_group, dout, q, k, v, out, softmax_lse, params, softmax_scale,
                                             causal=True, window_size=(-1, -1), deterministic=False):
    be = get_backend()
    T, H, _ = q.shape
    L = T // 2
    world = group_rank_world(process_group)[1]
    if dout.stride(-1) != 1:
        dout = dout.contiguous()
    kg, vg = _gathered_stream_kv(process_group, k, v, world)
    delta = torch.empty((H, T), dtype=torch.float32, device=q.device)
    dq = torch.empty_like(q)
    # this rank's dK/dV contributions for every token of the stream, fp32: the two slices' key ranges overlap
    dkg = torch.zeros(kg.shape, dtype=torch.float32, device=q.device)
    dvg = torch.zeros(vg.shape, dtype=torch.float32, device=q.device)
    for i, (cu_q, cu_k, mq, mk, ksl) in enumerate(params):
        rows = slice(i * L, (i + 1) * L)
        be.bwd_preprocess(dout[rows], out[rows], delta[:, rows], cu_seqlens_q=cu_q, max_seqlen_q=mq)
        be.bwd(dout[rows], q[rows], kg[ksl], vg[ksl], softmax_lse[:, rows], delta[:, rows],
               softmax_scale=softmax_scale, causal=causal, dq=dq[rows], dk_acc=dkg[ksl], dv_acc=dvg[ksl],
               acc_init=False, deterministic=deterministic, window=window_size, cu_seqlens_q=cu_q,
               cu_seqlens_k=cu_k, max_seqlen_q=mq, max_seqlen_k=mk)
    if single_rank(world):
        return dq, be.cast(dkg, k.dtype), be.cast(dvg, v.dtype)
    dk = torch.empty(k.shape, dtype=torch.float32, device=q.device)
    dv = torch.empty(v.shape, dtype=torch.float32, device=q.device)
    works = [reduce_scatter_async(dk, _to_rank_order(dkg, world), group=process_group),
             reduce_scatter_async(dv, _to_rank_order(dvg, world), group=process_group)]
    for w in works:
        w.wait()
    return dq, be.cast(dk, k.dtype), be.cast(dv, v.dtype)


class ZigZagLlama3FlashAttnVarlenFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens, heads_k_stride, dropout_p, softmax_scale, causal, window_size, alibi_slopes,
                deterministic, return_softmax, group):
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        _check_unsupported(dropout_p, window_size, alibi_slopes, windows_ok=True)   # K/V are gathered
        if dropout_p and dropout_p > 0:
            raise NotImplementedError("zigzag_llama3_flash_attn_varlen_func: dropout is not supported")
        if q.shape[0] % 2 != 0 or k.shape[0] != q.shape[0] or v.shape[0] != q.shape[0]:
            raise ValueError("zigzag_llama3: q, k, v hold the two stream slices of this rank (an even number of rows)")
        if q.stride(-1) != 1:
            q = q.contiguous()
        k = k.contiguous()      # all-gather source
        v = v.contiguous()
        rank, world = group_rank_world(group)
        host = torch.as_tensor(cu_seqlens).to("cpu", torch.int32)
        if int(host[-1]) != q.shape[0] * world:
            raise ValueError(f"zigzag_llama3: cu_seqlens ends at {int(host[-1])}, the ranks hold {q.shape[0] * world} tokens")
        params = tuple((_as_cu(cq, q.device), _as_cu(ck, q.device), mq, mk, sl)
                       for cq, ck, mq, mk, sl in zigzag_llama3_flash_attn_prepare_cu_seqlens(host, causal, rank, world))
        out, lse = zigzag_llama3_flash_attn_varlen_forward(group, q, k, v, params, softmax_scale, causal, window_size)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.params = params
        ctx.meta = (softmax_scale, causal, tuple(window_size), deterministic, group)
        return out if not return_softmax else (out, lse, None)

    @staticmethod
    def backward(ctx, dout, *args):
        q, k, v, out, lse = ctx.saved_tensors
        softmax_scale, causal, window_size, deterministic, group = ctx.meta
        dq, dk, dv = zigzag_llama3_flash_attn_varlen_backward(group, dout, q, k, v, out, lse, ctx.params, softmax_scale,
                                                              causal, window_size, deterministic)
        return (dq, dk, dv) + (None,) * 10


def _make_api():
    def func(q, k, v, cu_seqlens, heads_k_stride=1, dropout_p=0.0, softmax_scale=None, causal=False,
             window_size=(-1, -1), alibi_slopes=None, deterministic=False, return_attn_probs=False, group=None):
        return ZigZagLlama3FlashAttnVarlenFunc.apply(q, k, v, cu_seqlens, heads_k_stride, dropout_p, softmax_scale,
                                                     causal, window_size, alibi_slopes, deterministic,
                                                     return_attn_probs, group)

    def kvpacked_func(q, kv, cu_seqlens, heads_k_stride=1, **kw):
        return func(q, kv[:, 0], kv[:, 1], cu_seqlens, heads_k_stride, **kw)

    def qkvpacked_func(qkv, cu_seqlens, heads_k_stride=1, **kw):
        return func(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu_seqlens, heads_k_stride, **kw)

    func.__name__ = func.__qualname__ = "zigzag_llama3_flash_attn_varlen_func"
    kvpacked_func.__name__ = kvpacked_func.__qualname__ = "zigzag_llama3_flash_attn_varlen_kvpacked_func"
    qkvpacked_func.__name__ = qkvpacked_func.__qualname__ = "zigzag_llama3_flash_attn_varlen_qkvpacked_func"
    return _opaque(func), _opaque(kvpacked_func), _opaque(qkvpacked_func)


(
    zigzag_llama3_flash_attn_varlen_func,
    zigzag_llama3_flash_attn_varlen_kvpacked_func,
    zigzag_llama3_flash_attn_varlen_qkvpacked_func,
) = _make_api()
