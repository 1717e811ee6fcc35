"""Llama-3 style context parallelism: all-gather K/V per kv-head group, one varlen attention.

Same public surface and semantics as
/root/reference/ring_flash_attn/llama3_flash_attn_varlen.py (prepare_cu_seqlens :10-60,
forward :63-158, backward :161-299, autograd :302-387, wrappers :390-504).  The packed token
stream is cut into W contiguous slices; rank r attends its slice's queries against
`local_k_slice` of the gathered keys with a bottom-right aligned causal mask.

MI355X-first changes:
  * `heads_k_stride` is the reference's memory knob (one all-gather + one attention launch per group of
    that many kv heads; the reference benchmark uses 4, the HF adapter 1).  With 288 GB of HBM and a 256-CU
    chip a launch over 1 kv head is both unnecessary and starved (config 5: 2 q heads x 8 query blocks = 16
    workgroups), so consecutive groups are FUSED into super-groups for as long as the gathered K/V of a
    super-group (double buffered) stays below RFA_LLAMA3_GATHER_MAX_BYTES (default 1 GiB): fewer, larger
    collectives and launches over all fused heads.  Heads are independent, so the results do not change.
  * forward: the all-gather of super-group i+1 runs (side stream) beside the attention of super-group i;
  * backward: the dK/dV reduce-scatter of super-group i is posted asynchronously and runs beside the kernels
    of super-group i+1 (two contribution buffers); only the rows OUTSIDE `local_k_slice` are zero-filled
    (the kernel overwrites every row inside it) instead of the whole (2, T*W, g, D) buffer per group;
  * each group's attention writes straight into strided views of the final `out` / `lse` / `dq` tensors
    (no `torch.cat`, no per-group temporaries); delta is computed once; `prepare_cu_seqlens` does its
    integer work on one host copy of `cu_seqlens` instead of ~10 `.item()` syncs; world_size == 1 collapses
    to a single kernel for all heads.
"""
import torch

from . import config
from .backend import get_backend
from .utils import AllGatherComm as Comm, group_rank_world, reduce_scatter_async, single_rank
from ._api import _check_unsupported, _opaque
from ._common import _as_cu, dropout_arg, draw_dropout_seed, packed_pair


def fused_heads_k_stride(nheads_k: int, heads_k_stride: int, total_k: int, world: int, head_dim: int,
                         elt_bytes: int) -> int:
    """kv heads per super-group: the largest multiple of heads_k_stride that divides nheads_k, keeps the
    double-buffered gathered K/V below the budget and leaves at least config.llama3_min_groups super-groups (so that
    the exchange of one group can run beside the kernels of another) — never smaller than heads_k_stride itself."""
    cfg = config.get()
    budget = cfg.llama3_gather_max_bytes
    per_head = 2 * total_k * world * head_dim * elt_bytes          # K and V of one kv head, all ranks
    best = heads_k_stride
    for m in range(1, nheads_k // heads_k_stride + 1):
        hs = m * heads_k_stride
        if nheads_k % hs == 0 and 2 * hs * per_head <= budget and nheads_k // hs >= cfg.llama3_min_groups:
            best = hs
    return best


def llama3_flash_attn_prepare_cu_seqlens(cu_seqlens: torch.Tensor, causal: bool, rank: int, world_size: int):
    """Per-rank view of a packed token stream cut into `world_size` equal slices (integer work only; same results as
    /root/reference/ring_flash_attn/llama3_flash_attn_varlen.py:10-60, checked bit for bit against 72 vectors the
    reference produced: tests/test_abi.py::test_prepare_cu_seqlens_golden).

    cu_seqlens are the GLOBAL cumulative sequence lengths.  Returns
        cu_seqlens_q   boundaries of the sequence pieces inside this rank's query slice (starts at 0, ends at T/W)
        cu_seqlens_k   boundaries of the keys those pieces attend to, relative to the first needed key; with
                       `causal` the last piece's keys stop where the local queries stop, otherwise at the end of
                       its sequence — so the key range may be longer than T/W on both sides
        max_seqlen_q, max_seqlen_k
        local_k_slice  the slice of the GATHERED key rows that range covers
    The work is done on one host copy of cu_seqlens (a single device-to-host transfer)."""
    cu = [int(x) for x in cu_seqlens.tolist()]
    total = cu[-1]
    assert total % world_size == 0
    per_rank = total // world_size
    lo, hi = rank * per_rank, (rank + 1) * per_rank

    def first_at_least(val):                         # index of the first boundary >= val
        i = 0
        while i < len(cu) and cu[i] < val:
            i += 1
        return i

    left, right = first_at_least(lo), first_at_least(hi)
    if cu[left] != lo:                               # the slice starts inside a sequence: include its start
        left -= 1
    touched = cu[left:right + 1]                     # boundaries of every sequence with a token in [lo, hi)

    cu_q = [c - lo for c in touched]
    cu_q[0], cu_q[-1] = 0, per_rank                  # clip the first / last piece to the slice

    k_first = touched[0]
    k_last = hi if causal else cu[right]             # causal: nothing beyond the last local query is visible
    cu_k = [c - k_first for c in touched[:-1]] + [k_last - k_first]

    max_seqlen_q = max(b - a for a, b in zip(cu_q[:-1], cu_q[1:]))
    max_seqlen_k = max(b - a for a, b in zip(cu_k[:-1], cu_k[1:]))
    cu_seqlens_q = torch.tensor(cu_q, dtype=cu_seqlens.dtype, device=cu_seqlens.device)
    cu_seqlens_k = torch.tensor(cu_k, dtype=cu_seqlens.dtype, device=cu_seqlens.device)
    return cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, slice(k_first, k_last)


def llama3_flash_attn_varlen_forward(
    process_group,
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    cu_seqlens_q,
    cu_seqlens_k,
    max_seqlen_q,
    max_seqlen_k,
    heads_k_stride,
    local_k_slice,
    softmax_scale,
    dropout_p=0,
    causal=True,
    window_size=(-1, -1),
    alibi_slopes=None,
    deterministic=False,
    dropout_seed=None,
):
    be = get_backend()
    T, nheads, head_dim = q.shape
    total_k, nheads_k, _ = k.shape
    assert nheads_k % heads_k_stride == 0
    vl = dict(cu_seqlens_q=cu_seqlens_q, cu_seqlens_k=cu_seqlens_k, max_seqlen_q=max_seqlen_q,
              max_seqlen_k=max_seqlen_k)

    out = torch.empty_like(q)
    lse = torch.empty((nheads, T), dtype=torch.float32, device=q.device)
    rank, world_size = group_rank_world(process_group)
    # dropout positions are GLOBAL (stream position of a query = rank * T + local row; of a key = its row in the
    # gathered stream; head = its index among all query heads), so every rank — and every head group — draws the bits
    # an unsharded call with the same seed would (include/rfa.h)
    k_lo = local_k_slice.start or 0

    def drop(head0):
        return dropout_arg(dropout_p, dropout_seed, rank * T, k_lo, head0)

    if single_rank(world_size):
        be.fwd(q, k[local_k_slice], v[local_k_slice], softmax_scale=softmax_scale, causal=causal,
               out=out, lse=lse, window=window_size, dropout=drop(0), **vl)
        return out, lse

    hs = fused_heads_k_stride(nheads_k, heads_k_stride, total_k, world_size, head_dim, k.element_size())
    kvp = packed_pair(k, v) if hs == nheads_k else None
    if kvp is not None:
        # k and v are the two halves of ONE packed (T, 2, Hk, D) tensor (the kvpacked / qkvpacked entry points) and all
        # heads are one super-group: the packed tensor travels as it is — ONE all-gather instead of two, no contiguous
        # copies of the halves (round 5: those were 2 of the elementwise launches per pass of
        # profiles/history/r04_short_launch_kernel_trace.txt); the gathered K / V are strided views of the one buffer
        buf = torch.empty((total_k * world_size,) + tuple(kvp.shape[1:]), dtype=k.dtype, device=k.device)
        comm = Comm(process_group)
        comm.all_gather(buf, kvp)
        comm.wait()
        be.fwd(q, buf[local_k_slice, 0], buf[local_k_slice, 1], softmax_scale=softmax_scale, causal=causal,
               out=out, lse=lse, window=window_size, dropout=drop(0), **vl)
        return out, lse
    groups = list(range(0, nheads_k, hs))
    bufs = [torch.empty((2, total_k * world_size, hs, head_dim), dtype=k.dtype, device=k.device)
            for _ in range(min(2, len(groups)))]

    def post_gather(gi):
        g0 = groups[gi]
        comm = Comm(process_group)
        buf = bufs[gi % 2]
        comm.all_gather(buf[0], k[:, g0:g0 + hs].contiguous())
        comm.all_gather(buf[1], v[:, g0:g0 + hs].contiguous())
        return comm

    pending = post_gather(0)
    for gi, g0 in enumerate(groups):
        pending.wait()
        buf = bufs[gi % 2]
        if gi + 1 < len(groups):
            pending = post_gather(gi + 1)          # next super-group's K/V arrive beside this one's attention
        q_slice = slice(g0 * nheads // nheads_k, (g0 + hs) * nheads // nheads_k)
        be.fwd(q[:, q_slice], buf[0][local_k_slice], buf[1][local_k_slice],
               softmax_scale=softmax_scale, causal=causal, out=out[:, q_slice], lse=lse[q_slice], window=window_size,
               dropout=drop(q_slice.start), **vl)

    return out, lse


def llama3_flash_attn_varlen_backward(
    process_group,
    dout,
    q,
    k,
    v,
    out,
    softmax_lse,
    cu_seqlens_q,
    cu_seqlens_k,
    max_seqlen_q,
    max_seqlen_k,
    heads_k_stride,
    local_k_slice,
    softmax_scale,
    dropout_p=0,
    causal=True,
    window_size=(-1, -1),
    alibi_slopes=None,
    deterministic=False,
    dropout_seed=None,
    grads=None,
):
    """grads: optional (dq, dk, dv) buffers the gradients are written into — views of one packed gradient for the
    kv / qkv packed entry points (last stride 1), so that no per-tensor gradient is copied into it afterwards"""
    be = get_backend()
    T, nheads, head_dim = q.shape
    total_k, nheads_k, _ = k.shape
    assert nheads_k % heads_k_stride == 0
    vl = dict(cu_seqlens_q=cu_seqlens_q, cu_seqlens_k=cu_seqlens_k, max_seqlen_q=max_seqlen_q,
              max_seqlen_k=max_seqlen_k)
    if not softmax_lse.is_contiguous():
        softmax_lse = softmax_lse.contiguous()
    if dout.stride(-1) != 1:
        dout = dout.contiguous()

    delta = torch.empty((nheads, T), dtype=torch.float32, device=q.device)
    be.bwd_preprocess(dout, out, delta, cu_seqlens_q=cu_seqlens_q, max_seqlen_q=max_seqlen_q)

    if grads is not None:
        dq, dk, dv = grads
    else:
        dq = torch.empty_like(q)
        dk = torch.empty(k.shape, dtype=k.dtype, device=k.device)
        dv = torch.empty(v.shape, dtype=v.dtype, device=v.device)
    rank, world_size = group_rank_world(process_group)
    k_lo = local_k_slice.start or 0

    def drop(head0):                       # (the forward's positions)
        return dropout_arg(dropout_p, dropout_seed, rank * T, k_lo, head0)

    if single_rank(world_size):
        if local_k_slice.start != 0 or local_k_slice.stop != total_k:
            dk.zero_()
            dv.zero_()
        be.bwd(dout, q, k[local_k_slice], v[local_k_slice], softmax_lse, delta, softmax_scale=softmax_scale,
               causal=causal, dq=dq, dk=dk[local_k_slice], dv=dv[local_k_slice], deterministic=deterministic, window=window_size,
               dropout=drop(0), **vl)
        return dq, dk, dv

    hs = fused_heads_k_stride(nheads_k, heads_k_stride, total_k, world_size, head_dim, k.element_size())
    rows_all = total_k * world_size
    kvp = packed_pair(k, v) if hs == nheads_k else None
    if kvp is not None:
        # packed kv, one super-group (see the forward): ONE all-gather of the packed tensor, the dK/dV kernel writes this
        # rank's contributions into the two halves of ONE packed (rows, 2, Hk, D) buffer, and ONE reduce-scatter lands
        # them — in the packed gradient itself when the caller handed one over (the kvpacked entry points): no second
        # collective, no copy-back
        lo, hi = local_k_slice.start or 0, local_k_slice.stop if local_k_slice.stop is not None else rows_all
        buf = torch.empty((rows_all,) + tuple(kvp.shape[1:]), dtype=k.dtype, device=k.device)
        comm = Comm(process_group)
        comm.all_gather(buf, kvp)
        # this rank's contributions for EVERY rank's rows: the rows outside the local key slice stay zero in a buffer that is
        # kept and reused (backend.zero_outside: no fill launches per backward); the reduce-scatter below is waited for
        # before this function returns, so the next backward may write into it again
        dkvc = be.zero_outside(buf.shape, buf.dtype, buf.device, lo, hi)
        comm.wait()
        be.bwd(dout, q, buf[local_k_slice, 0], buf[local_k_slice, 1], softmax_lse, delta, softmax_scale=softmax_scale,
               causal=causal, dq=dq, dk=dkvc[local_k_slice, 0], dv=dkvc[local_k_slice, 1], deterministic=deterministic,
               window=window_size, dropout=drop(0), **vl)
        dst = packed_pair(dk, dv)
        land = dst if dst is not None else torch.empty((total_k,) + tuple(kvp.shape[1:]), dtype=k.dtype, device=k.device)
        reduce_scatter_async(land, dkvc, group=process_group).wait()
        if dst is None:
            dk.copy_(land[:, 0])
            dv.copy_(land[:, 1])
        return dq, dk, dv
    groups = list(range(0, nheads_k, hs))
    nbuf = min(2, len(groups))
    kv_bufs = [torch.empty((2, rows_all, hs, head_dim), dtype=k.dtype, device=k.device) for _ in range(nbuf)]
    lo, hi = local_k_slice.start or 0, local_k_slice.stop if local_k_slice.stop is not None else rows_all
    # this rank's dK/dV contributions for EVERY rank's rows (summed over ranks by the reduce-scatter): two kept buffers whose
    # rows outside the local key slice stay zero (backend.zero_outside) instead of two fills per head group
    dkv_bufs = [be.zero_outside((2, rows_all, hs, head_dim), k.dtype, k.device, lo, hi, slot=1 + i, dim=1) for i in range(nbuf)]
    whole = hs == nheads_k and dk.is_contiguous() and dv.is_contiguous()   # then the reduce-scatter lands in dk / dv directly
    if not whole:                                # its output must be contiguous
        rs_out = [torch.empty((2, total_k, hs, head_dim), dtype=k.dtype, device=k.device) for _ in range(nbuf)]

    def post_gather(gi):
        g0 = groups[gi]
        comm = Comm(process_group)
        buf = kv_bufs[gi % 2]
        comm.all_gather(buf[0], k[:, g0:g0 + hs].contiguous())
        comm.all_gather(buf[1], v[:, g0:g0 + hs].contiguous())
        return comm

    def finish(job):
        works, gi = job
        for w in works:
            w.wait()
        if not whole:
            g0 = groups[gi]
            dk[:, g0:g0 + hs] = rs_out[gi % 2][0]
            dv[:, g0:g0 + hs] = rs_out[gi % 2][1]

    pending = post_gather(0)
    job = None
    for gi, g0 in enumerate(groups):
        pending.wait()
        kv = kv_bufs[gi % 2]
        if gi + 1 < len(groups):
            pending = post_gather(gi + 1)
        dkv = dkv_bufs[gi % 2]                   # (its previous reduce-scatter, group gi-2, was finished at gi-1)
        q_slice = slice(g0 * nheads // nheads_k, (g0 + hs) * nheads // nheads_k)
        be.bwd(dout[:, q_slice], q[:, q_slice], kv[0][local_k_slice], kv[1][local_k_slice],
               softmax_lse[q_slice], delta[q_slice], softmax_scale=softmax_scale, causal=causal,
               dq=dq[:, q_slice], dk=dkv[0][local_k_slice], dv=dkv[1][local_k_slice],
               deterministic=deterministic, window=window_size, dropout=drop(q_slice.start), **vl)
        if job is not None:
            finish(job)                          # group gi-1's exchange ran beside the kernels just enqueued
        dst = (dk, dv) if whole else (rs_out[gi % 2][0], rs_out[gi % 2][1])
        job = ([reduce_scatter_async(dst[0], dkv[0], group=process_group),
                reduce_scatter_async(dst[1], dkv[1], group=process_group)], gi)
    finish(job)

    return dq, dk, dv


def _l3_forward(ctx, q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride, local_k_slice,
                dropout_p, softmax_scale, causal, window_size, alibi_slopes, deterministic, return_softmax, group):
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    _check_unsupported(dropout_p, window_size, alibi_slopes, windows_ok=True)   # K/V are gathered: one kernel sees them all
    # (strided views — the halves of a packed kv — are fine: the kernels take strides, and the all-gather sources are
    #  made contiguous per head group where they are posted)
    q, k, v = (t if t.stride(-1) == 1 else t.contiguous() for t in (q, k, v))
    cu_seqlens_q = _as_cu(cu_seqlens_q, q.device)
    cu_seqlens_k = _as_cu(cu_seqlens_k, q.device)
    ctx.dropout = (dropout_p, draw_dropout_seed()) if dropout_p and dropout_p > 0 else (0.0, None)
    out, softmax_lse = llama3_flash_attn_varlen_forward(
        group, q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride,
        local_k_slice, softmax_scale=softmax_scale, dropout_p=dropout_p, causal=causal,
        window_size=window_size, alibi_slopes=alibi_slopes, deterministic=False, dropout_seed=ctx.dropout[1],
    )
    ctx.save_for_backward(q, k, v, out, softmax_lse, cu_seqlens_q, cu_seqlens_k)
    ctx.static = (max_seqlen_q, max_seqlen_k, heads_k_stride, local_k_slice)
    ctx.softmax_scale = softmax_scale
    ctx.causal = causal
    ctx.deterministic = deterministic
    ctx.group = group
    ctx.window_size = tuple(window_size)
    return out if not return_softmax else (out, softmax_lse, None)


def _l3_backward(ctx, dout, grads=None):
    q, k, v, out, softmax_lse, cu_seqlens_q, cu_seqlens_k = ctx.saved_tensors
    return llama3_flash_attn_varlen_backward(
        ctx.group, dout, q, k, v, out, softmax_lse, cu_seqlens_q, cu_seqlens_k, *ctx.static,
        softmax_scale=ctx.softmax_scale, dropout_p=ctx.dropout[0], causal=ctx.causal, window_size=ctx.window_size,
        alibi_slopes=None, deterministic=ctx.deterministic, dropout_seed=ctx.dropout[1], grads=grads,
    )


class Llama3FlashAttnVarlenFunc(torch.autograd.Function):
    """autograd wrapper; argument order of reference llama3_flash_attn_varlen.py:302-323."""

    @staticmethod
    def forward(ctx, q, k, v, *rest):
        return _l3_forward(ctx, q, k, v, *rest)

    @staticmethod
    def backward(ctx, dout, *args):
        return _l3_backward(ctx, dout) + (None,) * 15


class Llama3FlashAttnVarlenKVPackedFunc(torch.autograd.Function):
    """(q, kv) entry: K and V are the two halves of `kv` (strided views, no copies) and their gradients are written
    straight into one packed gradient — autograd would otherwise build it from two per-tensor gradients with two
    fills, two strided copies and an add (measured: 7 extra launches, 4 % of a single-rank step)."""

    @staticmethod
    def forward(ctx, q, kv, *rest):
        return _l3_forward(ctx, q, kv[:, 0], kv[:, 1], *rest)

    @staticmethod
    def backward(ctx, dout, *args):
        k = ctx.saved_tensors[1]
        dkv = torch.empty((k.shape[0], 2) + tuple(k.shape[1:]), dtype=k.dtype, device=k.device)
        dq, _, _ = _l3_backward(ctx, dout, (torch.empty_like(ctx.saved_tensors[0]), dkv[:, 0], dkv[:, 1]))
        return (dq, dkv) + (None,) * 15


class Llama3FlashAttnVarlenQKVPackedFunc(torch.autograd.Function):
    """qkv entry: as above with the query gradient in the same packed buffer"""

    @staticmethod
    def forward(ctx, qkv, *rest):
        return _l3_forward(ctx, qkv[:, 0], qkv[:, 1], qkv[:, 2], *rest)

    @staticmethod
    def backward(ctx, dout, *args):
        q = ctx.saved_tensors[0]
        dqkv = torch.empty((q.shape[0], 3) + tuple(q.shape[1:]), dtype=q.dtype, device=q.device)
        _l3_backward(ctx, dout, (dqkv[:, 0], dqkv[:, 1], dqkv[:, 2]))
        return (dqkv,) + (None,) * 15


def _make_llama3_api():
    def func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride, local_k_slice,
             dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), alibi_slopes=None,
             deterministic=False, return_attn_probs=False, group=None):
        return Llama3FlashAttnVarlenFunc.apply(
            q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride, local_k_slice,
            dropout_p, softmax_scale, causal, window_size, alibi_slopes, deterministic, return_attn_probs, group)

    # (same positional tails as the reference wrappers, llama3_flash_attn_varlen.py:390-445)
    def kvpacked_func(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride, local_k_slice,
                      dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), alibi_slopes=None,
                      deterministic=False, return_attn_probs=False, group=None):
        return Llama3FlashAttnVarlenKVPackedFunc.apply(
            q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride,
            local_k_slice, dropout_p, softmax_scale, causal, window_size, alibi_slopes, deterministic,
            return_attn_probs, group)

    def qkvpacked_func(qkv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride, local_k_slice,
                       dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), alibi_slopes=None,
                       deterministic=False, return_attn_probs=False, group=None):
        return Llama3FlashAttnVarlenQKVPackedFunc.apply(
            qkv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride,
            local_k_slice, dropout_p, softmax_scale, causal, window_size, alibi_slopes, deterministic,
            return_attn_probs, group)

    func.__name__ = func.__qualname__ = "llama3_flash_attn_varlen_func"
    kvpacked_func.__name__ = kvpacked_func.__qualname__ = "llama3_flash_attn_varlen_kvpacked_func"
    qkvpacked_func.__name__ = qkvpacked_func.__qualname__ = "llama3_flash_attn_varlen_qkvpacked_func"

    # while dynamo TRACES a caller, a call without dropout is expressed with the registered operators rfa::llama3_fwd /
    # rfa::llama3_bwd (_ops.py: the whole schedule — head-group all-gathers, kernels, the reduce-scatter of dK/dV — as one
    # opaque node per direction, any world size), so the compiled graph has no break; eager calls keep the autograd
    # Functions above (packed gradients written into one buffer)
    def lower(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride, local_k_slice,
              dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), alibi_slopes=None,
              deterministic=False, return_attn_probs=False, group=None):
        from ._ops import llama3_attention

        _check_unsupported(dropout_p, window_size, alibi_slopes, windows_ok=True)
        return llama3_attention(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride,
                                local_k_slice, softmax_scale, causal, window_size, return_attn_probs, group)

    def lower_kv(q, kv, *a, **kw):
        return lower(q, kv[:, 0], kv[:, 1], *a, **kw)

    def lower_qkv(qkv, *a, **kw):
        return lower(qkv[:, 0], qkv[:, 1], qkv[:, 2], *a, **kw)

    def compilable(fn, low):
        import functools
        import inspect

        eager, sig = _opaque(fn), inspect.signature(fn)

        @functools.wraps(fn)
        def public(*args, **kwargs):
            if torch.compiler.is_compiling():
                bound = sig.bind(*args, **kwargs).arguments
                if not bound.get("dropout_p", 0.0) and isinstance(bound.get("local_k_slice"), slice):
                    return low(*args, **kwargs)
            return eager(*args, **kwargs)

        return public

    return compilable(func, lower), compilable(kvpacked_func, lower_kv), compilable(qkvpacked_func, lower_qkv)


(
    llama3_flash_attn_varlen_func,
    llama3_flash_attn_varlen_kvpacked_func,
    llama3_flash_attn_varlen_qkvpacked_func,
) = _make_llama3_api()
