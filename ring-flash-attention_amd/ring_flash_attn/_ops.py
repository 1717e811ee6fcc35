"""`torch.library` registration of the single-device attention operator.

The reference runs every test a second time under `torch.compile` (test/test.sh:23-25,
test/test_zigzag_ring_flash_attn_func.py:105-108).  The C-ABI calls of this package are ctypes
calls dynamo cannot trace, so they are registered as custom operators with fake (meta) kernels:

    rfa::attn_fwd(q, k, v, cu_seqlens?, max_seqlen, softmax_scale, causal, window_left, window_right) -> (out, lse)
    rfa::attn_bwd(dout, q, k, v, out, lse, cu_seqlens?, max_seqlen, softmax_scale, causal, window_left,
                  window_right, deterministic)
                                                                        -> (dq, dk, dv)

with an autograd formula linking them.  When a public function is traced by dynamo and the process
group has a single rank (every schedule then IS one attention call), it lowers to these operators and
the compiled graph contains them as opaque nodes — no graph break.

With more than one rank a schedule interleaves kernels with RCCL traffic, which dynamo cannot trace either; the WHOLE
schedule is then one operator (round 4):

    rfa::sched_fwd(schedule, q, k, v, cu_seqlens?, max_seqlen, softmax_scale, causal, group_name) -> (out, lse)
    rfa::sched_bwd(schedule, dout, q, k, v, out, lse, cu_seqlens?, max_seqlen, softmax_scale, causal, group_name)
                                                                        -> (dq, dk, dv)

`schedule` names a registered (forward, backward) pair — ring / zigzag / stripe and the two varlen forms — and
`group_name` the process group (ProcessGroup.group_name, resolved inside the operator), so a `torch.compile`d caller
captures a multi-rank call as one opaque node as well (`fullgraph=True` holds at world size 2, 4: tests/_ring_worker.py,
RFA_TEST_COMPILE).  The operator runs the eager schedule — same kernels, same exchange, same order — minus the
forward-to-backward hand-over of gathered K/V (an operator's backward sees only what it saved: it gathers again).
llama3 has its own pair (rfa::llama3_fwd / rfa::llama3_bwd, at the end of this file).  Dropout calls (a host-side seed
per call) stay behind `torch.compiler.disable`: a graph break around an eagerly executed schedule, identical results.  Eager calls never come through here: they keep the autograd Functions
of _api.py (packed gradients written into one buffer, kept K/V).
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from .backend import get_backend


def _audit_verify(group, where):
    from .utils import audit_verify

    audit_verify(group, where)


def _lse_shape(q: Tensor, varlen: bool):
    if varlen:
        return (q.shape[1], q.shape[0])                  # (H, T)
    return (q.shape[0], q.shape[2], q.shape[1])          # (B, H, S)


def _vl(cu_seqlens, max_seqlen):
    if cu_seqlens is None:
        return {}
    return dict(cu_seqlens_q=cu_seqlens, cu_seqlens_k=cu_seqlens, max_seqlen_q=max_seqlen, max_seqlen_k=max_seqlen)


@torch.library.custom_op("rfa::attn_fwd", mutates_args=())
def attn_fwd(q: Tensor, k: Tensor, v: Tensor, cu_seqlens: Optional[Tensor], max_seqlen: int,
             softmax_scale: float, causal: bool, window_left: int, window_right: int) -> Tuple[Tensor, Tensor]:
    varlen = cu_seqlens is not None
    out = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    lse = torch.empty(_lse_shape(q, varlen), dtype=torch.float32, device=q.device)
    get_backend().fwd(q, k, v, softmax_scale=softmax_scale, causal=causal, out=out, lse=lse,
                      window=(window_left, window_right), **_vl(cu_seqlens, max_seqlen))
    return out, lse


@attn_fwd.register_fake
def _attn_fwd_fake(q, k, v, cu_seqlens, max_seqlen, softmax_scale, causal, window_left, window_right):
    return (torch.empty(q.shape, dtype=q.dtype, device=q.device),
            torch.empty(_lse_shape(q, cu_seqlens is not None), dtype=torch.float32, device=q.device))


@torch.library.custom_op("rfa::attn_bwd", mutates_args=())
def attn_bwd(dout: Tensor, q: Tensor, k: Tensor, v: Tensor, out: Tensor, lse: Tensor,
             cu_seqlens: Optional[Tensor], max_seqlen: int, softmax_scale: float, causal: bool,
             window_left: int, window_right: int, deterministic: bool) -> Tuple[Tensor, Tensor, Tensor]:
    be = get_backend()
    varlen = cu_seqlens is not None
    if dout.stride(-1) != 1:
        dout = dout.contiguous()
    if not lse.is_contiguous():
        lse = lse.contiguous()
    delta = torch.empty_like(lse)
    if varlen:
        be.bwd_preprocess(dout, out, delta, cu_seqlens_q=cu_seqlens, max_seqlen_q=max_seqlen)
    else:
        be.bwd_preprocess(dout, out, delta)
    dq = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    dk = torch.empty(k.shape, dtype=k.dtype, device=k.device)
    dv = torch.empty(v.shape, dtype=v.dtype, device=v.device)
    be.bwd(dout, q, k, v, lse, delta, softmax_scale=softmax_scale, causal=causal, dq=dq, dk=dk, dv=dv,
           deterministic=deterministic, window=(window_left, window_right), **_vl(cu_seqlens, max_seqlen))
    return dq, dk, dv


@attn_bwd.register_fake
def _attn_bwd_fake(dout, q, k, v, out, lse, cu_seqlens, max_seqlen, softmax_scale, causal, window_left, window_right,
                   deterministic):
    return (torch.empty(q.shape, dtype=q.dtype, device=q.device),
            torch.empty(k.shape, dtype=k.dtype, device=k.device),
            torch.empty(v.shape, dtype=v.dtype, device=v.device))


# (`deterministic` is not threaded through: the kernels are always deterministic — no atomics — so the flag the eager
#  path records has no effect on the result either)
def _setup_context(ctx, inputs, output):
    q, k, v, cu_seqlens, max_seqlen, softmax_scale, causal, window_left, window_right = inputs
    ctx.window = (window_left, window_right)
    out, lse = output
    ctx.save_for_backward(q, k, v, out, lse, cu_seqlens)
    ctx.max_seqlen = max_seqlen
    ctx.softmax_scale = softmax_scale
    ctx.causal = causal


def _backward(ctx, dout, dlse):
    q, k, v, out, lse, cu_seqlens = ctx.saved_tensors
    dq, dk, dv = torch.ops.rfa.attn_bwd(dout, q, k, v, out, lse, cu_seqlens, ctx.max_seqlen, ctx.softmax_scale,
                                        ctx.causal, ctx.window[0], ctx.window[1], False)
    return dq, dk, dv, None, None, None, None, None, None


torch.library.register_autograd("rfa::attn_fwd", _backward, setup_context=_setup_context)


def single_device_attention(q, k, v, cu_seqlens, max_seqlen, softmax_scale, causal, return_attn_probs,
                            window_size=(-1, -1)):
    """what every schedule of this package reduces to on a single-rank group, as traceable operators.  Inputs are
    normalised exactly as the eager entry points do (_common._prep_qkv / _as_cu): unit head_dim stride, int32
    cu_seqlens on the compute device."""
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    if q.stride(-1) != 1:
        q = q.contiguous()
    if k.stride(-1) != 1:
        k = k.contiguous()
    if v.stride(-1) != 1:
        v = v.contiguous()
    if cu_seqlens is not None:
        if not torch.is_tensor(cu_seqlens):
            cu_seqlens = torch.tensor(cu_seqlens, dtype=torch.int32)
        cu_seqlens = cu_seqlens.to(device=q.device, dtype=torch.int32).contiguous()
    out, lse = torch.ops.rfa.attn_fwd(q, k, v, cu_seqlens, int(max_seqlen), float(softmax_scale), bool(causal),
                                      int(window_size[0]), int(window_size[1]))
    return (out, lse, None) if return_attn_probs else out


# ---------------------------------------------------------------------------------------------------------------
# multi-rank schedules as ONE operator each way
_SCHEDULES = {}


def register_schedule(name: str, forward_impl, backward_impl):
    """_api.make_dense_api / make_varlen_api register their (forward, backward) schedule under the public prefix"""
    _SCHEDULES[name] = (forward_impl, backward_impl)


def has_schedule(name: str) -> bool:
    return name in _SCHEDULES


def group_name_of(group) -> str:
    import torch.distributed as dist

    return (group if group is not None else dist.group.WORLD).group_name


def _group_of(name: str):
    from torch.distributed import distributed_c10d as c10d

    return c10d._resolve_process_group(name)


def _lead(cu_seqlens, max_seqlen):
    return () if cu_seqlens is None else (cu_seqlens, max_seqlen)


@torch.library.custom_op("rfa::sched_fwd", mutates_args=())
def sched_fwd(schedule: str, q: Tensor, k: Tensor, v: Tensor, cu_seqlens: Optional[Tensor], max_seqlen: int,
              softmax_scale: float, causal: bool, group_name: str) -> Tuple[Tensor, Tensor]:
    fwd, _ = _SCHEDULES[schedule]
    out, lse = fwd(_group_of(group_name), q, k, v, *_lead(cu_seqlens, max_seqlen), softmax_scale=softmax_scale,
                   dropout_p=0.0, causal=causal, window_size=(-1, -1), alibi_slopes=None, deterministic=False)
    _audit_verify(_group_of(group_name), f"{schedule} forward (compiled caller)")
    return out.contiguous(), lse.contiguous()


@sched_fwd.register_fake
def _sched_fwd_fake(schedule, q, k, v, cu_seqlens, max_seqlen, softmax_scale, causal, group_name):
    return (torch.empty(q.shape, dtype=q.dtype, device=q.device),
            torch.empty(_lse_shape(q, cu_seqlens is not None), dtype=torch.float32, device=q.device))


@torch.library.custom_op("rfa::sched_bwd", mutates_args=())
def sched_bwd(schedule: str, dout: Tensor, q: Tensor, k: Tensor, v: Tensor, out: Tensor, lse: Tensor,
              cu_seqlens: Optional[Tensor], max_seqlen: int, softmax_scale: float, causal: bool,
              group_name: str) -> Tuple[Tensor, Tensor, Tensor]:
    _, bwd = _SCHEDULES[schedule]
    if dout.stride(-1) != 1:
        dout = dout.contiguous()
    dq, dk, dv = bwd(_group_of(group_name), dout, q, k, v, out, lse, *_lead(cu_seqlens, max_seqlen),
                     softmax_scale=softmax_scale, dropout_p=0.0, causal=causal, window_size=(-1, -1),
                     alibi_slopes=None, deterministic=False)
    _audit_verify(_group_of(group_name), f"{schedule} backward (compiled caller)")
    return dq.contiguous(), dk.contiguous(), dv.contiguous()


@sched_bwd.register_fake
def _sched_bwd_fake(schedule, dout, q, k, v, out, lse, cu_seqlens, max_seqlen, softmax_scale, causal, group_name):
    return (torch.empty(q.shape, dtype=q.dtype, device=q.device),
            torch.empty(k.shape, dtype=k.dtype, device=k.device),
            torch.empty(v.shape, dtype=v.dtype, device=v.device))


def _sched_setup_context(ctx, inputs, output):
    schedule, q, k, v, cu_seqlens, max_seqlen, softmax_scale, causal, group_name = inputs
    out, lse = output
    ctx.save_for_backward(q, k, v, out, lse, cu_seqlens)
    ctx.meta = (schedule, max_seqlen, softmax_scale, causal, group_name)


def _sched_backward(ctx, dout, dlse):
    q, k, v, out, lse, cu_seqlens = ctx.saved_tensors
    schedule, max_seqlen, softmax_scale, causal, group_name = ctx.meta
    dq, dk, dv = torch.ops.rfa.sched_bwd(schedule, dout, q, k, v, out, lse, cu_seqlens, max_seqlen, softmax_scale,
                                         causal, group_name)
    return None, dq, dk, dv, None, None, None, None, None


torch.library.register_autograd("rfa::sched_fwd", _sched_backward, setup_context=_sched_setup_context)


def multi_rank_attention(schedule, q, k, v, cu_seqlens, max_seqlen, softmax_scale, causal, return_attn_probs, group):
    """a multi-rank call of a registered schedule as traceable operators; inputs normalised as the eager entry points do"""
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    q, k, v = (t if t.is_contiguous() else t.contiguous() for t in (q, k, v))
    if cu_seqlens is not None:
        if not torch.is_tensor(cu_seqlens):
            cu_seqlens = torch.tensor(cu_seqlens, dtype=torch.int32)
        cu_seqlens = cu_seqlens.to(device=q.device, dtype=torch.int32).contiguous()
    out, lse = torch.ops.rfa.sched_fwd(schedule, q, k, v, cu_seqlens, int(max_seqlen), float(softmax_scale), bool(causal),
                                       group_name_of(group))
    return (out, lse, None) if return_attn_probs else out


# ---------------------------------------------------------------------------------------------------------------
# llama3 all-gather context parallelism as one operator each way (any world size; windows included, dropout not: it
# draws a host-side seed per call).  local_k_slice travels as its (start, stop) pair.
@torch.library.custom_op("rfa::llama3_fwd", mutates_args=())
def llama3_fwd(q: Tensor, k: Tensor, v: Tensor, cu_seqlens_q: Tensor, cu_seqlens_k: Tensor, max_seqlen_q: int,
               max_seqlen_k: int, heads_k_stride: int, k_start: int, k_stop: int, softmax_scale: float, causal: bool,
               window_left: int, window_right: int, group_name: str) -> Tuple[Tensor, Tensor]:
    from .llama3_flash_attn_varlen import llama3_flash_attn_varlen_forward

    out, lse = llama3_flash_attn_varlen_forward(
        _group_of(group_name), q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride,
        slice(k_start, k_stop), softmax_scale=softmax_scale, dropout_p=0.0, causal=causal,
        window_size=(window_left, window_right), alibi_slopes=None, deterministic=False)
    return out.contiguous(), lse.contiguous()


@llama3_fwd.register_fake
def _llama3_fwd_fake(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride, k_start, k_stop,
                     softmax_scale, causal, window_left, window_right, group_name):
    return (torch.empty(q.shape, dtype=q.dtype, device=q.device),
            torch.empty((q.shape[1], q.shape[0]), dtype=torch.float32, device=q.device))


@torch.library.custom_op("rfa::llama3_bwd", mutates_args=())
def llama3_bwd(dout: Tensor, q: Tensor, k: Tensor, v: Tensor, out: Tensor, lse: Tensor, cu_seqlens_q: Tensor,
               cu_seqlens_k: Tensor, max_seqlen_q: int, max_seqlen_k: int, heads_k_stride: int, k_start: int, k_stop: int,
               softmax_scale: float, causal: bool, window_left: int, window_right: int,
               group_name: str) -> Tuple[Tensor, Tensor, Tensor]:
    from .llama3_flash_attn_varlen import llama3_flash_attn_varlen_backward

    if dout.stride(-1) != 1:
        dout = dout.contiguous()
    dq, dk, dv = llama3_flash_attn_varlen_backward(
        _group_of(group_name), dout, q, k, v, out, lse, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
        heads_k_stride, slice(k_start, k_stop), softmax_scale=softmax_scale, dropout_p=0.0, causal=causal,
        window_size=(window_left, window_right), alibi_slopes=None, deterministic=False)
    return dq.contiguous(), dk.contiguous(), dv.contiguous()


@llama3_bwd.register_fake
def _llama3_bwd_fake(dout, q, k, v, out, lse, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride,
                     k_start, k_stop, softmax_scale, causal, window_left, window_right, group_name):
    return (torch.empty(q.shape, dtype=q.dtype, device=q.device),
            torch.empty(k.shape, dtype=k.dtype, device=k.device),
            torch.empty(v.shape, dtype=v.dtype, device=v.device))


def _llama3_setup_context(ctx, inputs, output):
    (q, k, v, cu_q, cu_k, max_q, max_k, stride, k_start, k_stop, softmax_scale, causal, wl, wr, group_name) = inputs
    out, lse = output
    ctx.save_for_backward(q, k, v, out, lse, cu_q, cu_k)
    ctx.meta = (max_q, max_k, stride, k_start, k_stop, softmax_scale, causal, wl, wr, group_name)


def _llama3_backward(ctx, dout, dlse):
    q, k, v, out, lse, cu_q, cu_k = ctx.saved_tensors
    dq, dk, dv = torch.ops.rfa.llama3_bwd(dout, q, k, v, out, lse, cu_q, cu_k, *ctx.meta)
    return (dq, dk, dv) + (None,) * 12


torch.library.register_autograd("rfa::llama3_fwd", _llama3_backward, setup_context=_llama3_setup_context)


def llama3_attention(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, heads_k_stride, local_k_slice,
                     softmax_scale, causal, window_size, return_attn_probs, group):
    """llama3_flash_attn_varlen_func as traceable operators (inputs normalised as the eager entry point does)"""
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    q, k, v = (t if t.is_contiguous() else t.contiguous() for t in (q, k, v))

    def cu(c):
        if not torch.is_tensor(c):
            c = torch.tensor(c, dtype=torch.int32)
        return c.to(device=q.device, dtype=torch.int32).contiguous()

    if local_k_slice.step not in (None, 1):
        raise ValueError("local_k_slice must be a contiguous slice")
    k_start = int(local_k_slice.start or 0)
    k_stop = int(local_k_slice.stop) if local_k_slice.stop is not None else (1 << 62)      # (open end)
    out, lse = torch.ops.rfa.llama3_fwd(q, k, v, cu(cu_seqlens_q), cu(cu_seqlens_k), int(max_seqlen_q), int(max_seqlen_k),
                                        int(heads_k_stride), k_start, k_stop, float(softmax_scale), bool(causal),
                                        int(window_size[0]), int(window_size[1]), group_name_of(group))
    return (out, lse, None) if return_attn_probs else out
