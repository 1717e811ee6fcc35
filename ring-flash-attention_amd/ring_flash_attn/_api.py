"""Public-API factory: one autograd Function + the (func, kvpacked_func, qkvpacked_func)
trio per algorithm, generated from its `*_forward` / `*_backward` schedule.

The generated callables keep the exact keyword surface of the reference package
(/root/reference/ring_flash_attn/__init__.py:1-35 and SURVEY.md Appendix A):
    dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), alibi_slopes=None,
    deterministic=False, return_attn_probs=False, group=None
with the reference's semantics: softmax_scale None -> head_dim ** -0.5; alibi_slopes must be
None; dropout_p / window_size work wherever one kernel call sees all keys (single-rank groups, llama3) and raise
on a multi-rank ring (unsupported there in the reference too, README.md:158-159);
return_attn_probs=True -> (out, softmax_lse, None); `group=None` is the default process group;
inputs are the caller's LOCAL shard.
"""
import torch

from ._common import _prep_qkv, _as_cu, draw_dropout_seed
from .utils import audit_verify


def _opaque(fn):
    """The schedules are ctypes calls into librfa_hip.so interleaved with torch.distributed traffic, which dynamo
    cannot trace: behind `torch.compiler.disable` a compiled caller graph-breaks around them and runs them
    eagerly, with identical results (multi-rank groups; llama3)."""
    try:
        import functools

        wrapped = torch.compiler.disable(fn)
        functools.update_wrapper(wrapped, fn)
        return wrapped
    except Exception:        # very old torch without torch.compiler
        return fn


def _single_rank(group) -> bool:
    import torch.distributed as dist

    return dist.get_world_size(group) == 1


def _compilable(fn, lower, multi=None):
    """Public callable = `fn` (eager: the autograd Functions above, opaque to dynamo) except while dynamo is
    TRACING it on a single-rank group: then `lower(...)` expresses the call with the registered custom
    operators (_ops.py: rfa::attn_fwd / rfa::attn_bwd, fake kernels + autograd formula), so `torch.compile`
    captures the operator in its graph instead of breaking it (the reference runs its tests a second time under
    torch.compile: test/test.sh:23-25).  `multi(...)`, where the schedule has one: the same for a multi-rank group —
    the whole schedule as one operator per direction (_ops.py: rfa::sched_fwd / rfa::sched_bwd)."""
    import functools

    eager = _opaque(fn)

    import inspect

    sig = inspect.signature(fn)

    @functools.wraps(fn)
    def public(*args, **kwargs):
        if torch.compiler.is_compiling():
            # `group` may be passed positionally: resolve it the way the call itself would
            bound = sig.bind(*args, **kwargs).arguments
            # (dropout draws a host-side seed per call: such calls run eagerly behind a graph break)
            if not bound.get("dropout_p", 0.0):
                if _single_rank(bound.get("group", None)):
                    return lower(*args, **kwargs)
                # several ranks: the whole schedule as one registered operator (_ops.py: rfa::sched_fwd / sched_bwd),
                # where the schedule has one — windows raise over a multi-rank ring, llama3 keeps the graph break
                if multi is not None and not has_window(bound.get("window_size", (-1, -1))):
                    return multi(*args, **kwargs)
        return eager(*args, **kwargs)

    return public


def _check_unsupported(dropout_p, window_size, alibi_slopes, windows_ok=False):
    """Sliding windows and dropout are implemented in the kernels (flash_attn semantics; dropout: the counter-based
    mask of include/rfa.h) and usable wherever ONE kernel call sees all the keys a query may attend to: every function
    on a single-rank group, and llama3_flash_attn_varlen_func on any group (it gathers K/V) — the same coverage the
    reference gets from forwarding window_size / dropout_p to flash_attn (llama3_flash_attn_varlen.py:131-147).  The
    ring / zigzag / stripe schedules over several ranks would apply a window per block, which is wrong, and the
    reference declares dropout unsupported there (README.md:158-159): they raise.  Both together are not available."""
    assert alibi_slopes is None
    drop = bool(dropout_p) and dropout_p > 0
    if drop and not 0 < dropout_p < 1:
        raise ValueError("dropout_p must be in [0, 1)")
    if drop and not windows_ok:
        raise NotImplementedError("ring_flash_attn: dropout over a multi-rank ring is not supported (as in the "
                                  "reference); use llama3_flash_attn_varlen_func or a single-rank group")
    if not windows_ok and has_window(window_size):
        raise NotImplementedError("ring_flash_attn: sliding window over a multi-rank ring is not supported (as in the "
                                  "reference); use llama3_flash_attn_varlen_func or a single-rank group")
    if drop and has_window(window_size):
        raise NotImplementedError("ring_flash_attn: dropout together with a sliding window is not supported")


def has_window(window_size) -> bool:
    return window_size is not None and (window_size[0] >= 0 or window_size[1] >= 0)


def window_ok_for(group) -> bool:
    from .utils import group_rank_world

    return group_rank_world(group)[1] == 1


def _keep_list(ctx, forward_impl):
    """Schedules that can hand buffers from their forward to their backward (`forward_impl.keeps_for_backward`: the
    zigzag gather form keeps the K/V it gathered) get a list to append tensors to — only when an input needs a
    gradient.  The tensors are stored with ctx.save_for_backward, i.e. they are owned by the autograd graph: freed with
    it, never kept for torch.no_grad() calls (no graph), discarded and re-made under activation checkpointing, visible
    to saved-tensor hooks (offloading) — there is no process-global hand-off."""
    if getattr(forward_impl, "keeps_for_backward", False) and any(ctx.needs_input_grad[:3]):
        keep = _KeepList()
        return keep, {"keep": keep}
    return None, {}


class _KeepList(list):
    """the tensors a forward hands to its backward, the reservation they hold in the process-wide budget of kept bytes
    (config.kept_budget) and the group's agreement that EVERY rank kept (utils.Agreement; None: nothing was asked)"""
    token = None
    agreed = None


def _hold_kept(ctx, keep):
    ctx.kept_token = keep.token if keep is not None else None
    ctx.kept_agreed = keep.agreed if keep is not None else None


def _release_kept(ctx):
    """The reservation returns when the kept buffers are gone: with the backward when the graph is not retained, else
    with the graph (Token.__del__).  Whether this backward retains the graph is not visible from inside it, so the test
    runs as an engine callback at the END of the backward pass: by then a non-retained node has dropped its saved tensors
    (reading them raises), a retained one still has them and keeps its reservation."""
    tok = getattr(ctx, "kept_token", None)
    if tok is None:
        return

    def after_backward():
        try:
            ctx.saved_tensors
        except RuntimeError:
            tok.release()

    try:
        torch.autograd.Variable._execution_engine.queue_callback(after_backward)
    except Exception:                # (not inside an engine run: a direct call of the backward in a test)
        tok.release()


def _split_kept(ctx, more):
    n = getattr(ctx, "n_lead_t", len(more))
    tensors_lead, kept = more[:n], more[n:]
    agreed = getattr(ctx, "kept_agreed", None)
    if agreed is not None and not agreed.resolve():
        # some rank could not keep its gathered K/V: EVERY rank gathers again (a rank deciding alone would leave the
        # others in a collective it does not join); this rank's buffers are simply not used
        kept = ()
    return tensors_lead, ({"kept": tuple(kept)} if kept else {})


def make_autograd_function(name, forward_impl, backward_impl, n_lead):
    """n_lead: number of non-tensor positional arguments between (q,k,v) and the common tail
    (0 for the batch API, 2 = (cu_seqlens, max_seqlen) for varlen)."""

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, q, k, v, *rest):
            lead = rest[:n_lead]
            (dropout_p, softmax_scale, causal, window_size, alibi_slopes, deterministic,
             return_softmax, group) = rest[n_lead:]
            if softmax_scale is None:
                softmax_scale = q.shape[-1] ** (-0.5)
            _check_unsupported(dropout_p, window_size, alibi_slopes, windows_ok=window_ok_for(group))
            q, k, v = _prep_qkv(q, k, v, group)
            tensors_lead = ()
            if n_lead:
                cu = _as_cu(lead[0], q.device)
                lead = (cu,) + tuple(lead[1:])
                tensors_lead = (cu,)
            keep, extra = _keep_list(ctx, forward_impl)
            ctx.dropout = (dropout_p, draw_dropout_seed()) if dropout_p and dropout_p > 0 else (0.0, None)
            if ctx.dropout[1] is not None:
                extra["dropout_seed"] = ctx.dropout[1]
            out, softmax_lse = forward_impl(
                group, q, k, v, *lead, softmax_scale=softmax_scale, dropout_p=dropout_p, causal=causal,
                window_size=window_size, alibi_slopes=alibi_slopes, deterministic=False, **extra,
            )
            audit_verify(group, f"{name} forward")          # (config.exchange_check: a no-op otherwise)
            ctx.save_for_backward(q, k, v, out, softmax_lse, *tensors_lead, *(keep or ()))
            _hold_kept(ctx, keep)
            ctx.n_lead_t, ctx.n_keep = len(tensors_lead), len(keep or ())
            ctx.lead_rest = tuple(lead[1:]) if n_lead else ()
            ctx.softmax_scale = softmax_scale
            ctx.causal = causal
            ctx.deterministic = deterministic
            ctx.group = group
            ctx.window_size = tuple(window_size)
            return out if not return_softmax else (out, softmax_lse, None)

        @staticmethod
        def backward(ctx, dout, *args):
            q, k, v, out, softmax_lse, *more = ctx.saved_tensors
            tensors_lead, extra = _split_kept(ctx, more)
            if ctx.dropout[1] is not None:
                extra["dropout_seed"] = ctx.dropout[1]
            dq, dk, dv = backward_impl(
                ctx.group, dout, q, k, v, out, softmax_lse, *tensors_lead, *ctx.lead_rest,
                softmax_scale=ctx.softmax_scale, dropout_p=ctx.dropout[0], causal=ctx.causal,
                window_size=ctx.window_size, alibi_slopes=None, deterministic=ctx.deterministic, **extra,
            )
            _release_kept(ctx)
            audit_verify(ctx.group, f"{name} backward")
            return (dq, dk, dv) + (None,) * (n_lead + 8)

    _Fn.__name__ = _Fn.__qualname__ = name
    return _Fn


def make_packed_function(name, base_fn, forward_impl, backward_impl, n_lead, pack_dim, n_packed, packed_travel=False):
    """autograd Function for the packed entry points (`kv` = 2 tensors, `qkv` = 3 tensors stacked on
    `pack_dim`).  Same math as `base_fn`; the only difference is where the gradients land: ONE packed
    buffer whose slices are handed to the schedule as output views (`out_grads`), instead of letting
    autograd build the packed gradient from three slice-gradients (zero-fill + copy + add kernels
    per step).  Falls back to the generic path when the schedule returns fresh tensors (W > 1)."""

    class _PFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, *args):
            n_t = 1 if n_packed == 3 else 2                 # (qkv,) or (q, kv)
            tensors, rest = args[:n_t], args[n_t:]
            packed = tensors[-1]
            parts = [packed.select(pack_dim, i) for i in range(n_packed)]
            q, k, v = (parts if n_packed == 3 else [tensors[0]] + parts)
            lead = rest[:n_lead]
            (dropout_p, softmax_scale, causal, window_size, alibi_slopes, deterministic,
             return_softmax, group) = rest[n_lead:]
            if softmax_scale is None:
                softmax_scale = q.shape[-1] ** (-0.5)
            _check_unsupported(dropout_p, window_size, alibi_slopes, windows_ok=window_ok_for(group))
            q, k, v = _prep_qkv(q, k, v, group, packed_travel=packed_travel and n_packed == 2)
            tensors_lead = ()
            if n_lead:
                cu = _as_cu(lead[0], q.device)
                lead = (cu,) + tuple(lead[1:])
                tensors_lead = (cu,)
            keep, extra = _keep_list(ctx, forward_impl)
            ctx.dropout = (dropout_p, draw_dropout_seed()) if dropout_p and dropout_p > 0 else (0.0, None)
            if ctx.dropout[1] is not None:
                extra["dropout_seed"] = ctx.dropout[1]
            out, softmax_lse = forward_impl(
                group, q, k, v, *lead, softmax_scale=softmax_scale, dropout_p=dropout_p, causal=causal,
                window_size=window_size, alibi_slopes=alibi_slopes, deterministic=False, **extra,
            )
            audit_verify(group, f"{name} forward")          # (config.exchange_check: a no-op otherwise)
            ctx.save_for_backward(q, k, v, out, softmax_lse, *tensors_lead, *(keep or ()))
            _hold_kept(ctx, keep)
            ctx.n_lead_t, ctx.n_keep = len(tensors_lead), len(keep or ())
            ctx.lead_rest = tuple(lead[1:]) if n_lead else ()
            ctx.softmax_scale = softmax_scale
            ctx.causal = causal
            ctx.deterministic = deterministic
            ctx.group = group
            ctx.window_size = tuple(window_size)
            ctx.packed_meta = (packed.shape, packed.dtype, packed.device)
            return out if not return_softmax else (out, softmax_lse, None)

        @staticmethod
        def backward(ctx, dout, *args):
            q, k, v, out, softmax_lse, *more = ctx.saved_tensors
            tensors_lead, extra = _split_kept(ctx, more)
            shape, dtype, device = ctx.packed_meta
            dpacked = torch.empty(shape, dtype=dtype, device=device)
            views = [dpacked.select(pack_dim, i) for i in range(n_packed)]
            if n_packed == 3:
                out_grads = tuple(views)
            else:
                out_grads = (None, views[0], views[1])
            if ctx.dropout[1] is not None:
                extra["dropout_seed"] = ctx.dropout[1]
            dq, dk, dv = backward_impl(
                ctx.group, dout, q, k, v, out, softmax_lse, *tensors_lead, *ctx.lead_rest,
                softmax_scale=ctx.softmax_scale, dropout_p=ctx.dropout[0], causal=ctx.causal,
                window_size=ctx.window_size, alibi_slopes=None, deterministic=ctx.deterministic,
                out_grads=out_grads, **extra,
            )
            got = (dq, dk, dv)[3 - n_packed:]
            for view, g in zip(views, got):
                if g.data_ptr() != view.data_ptr():       # schedule returned its own tensor
                    view.copy_(g)
            grads = (dpacked,) if n_packed == 3 else (dq, dpacked)
            _release_kept(ctx)
            audit_verify(ctx.group, f"{name} backward")
            return grads + (None,) * (n_lead + 8)

    _PFn.__name__ = _PFn.__qualname__ = name
    return _PFn


def _grad_buffers(out_grads, q, k, v):
    """(dq, dk, dv) output tensors for the single-GPU path: caller-provided views or fresh."""
    og = out_grads or (None, None, None)
    return (og[0] if og[0] is not None else (torch.empty_like(q) if q is not None else None),
            og[1] if og[1] is not None else torch.empty_like(k),
            og[2] if og[2] is not None else torch.empty_like(v))


def make_dense_api(fn, prefix, forward_impl=None, backward_impl=None, packed_travel=False):
    """(B,S,H,D) API: returns (func, kvpacked_func, qkvpacked_func).  packed_travel: the schedule exchanges a
    packed `kv` as one buffer and writes dK/dV straight into the packed gradient (`out_grads`) at any world size."""
    kv_fn = qkv_fn = None
    if forward_impl is not None:
        kv_fn = make_packed_function(fn.__name__ + "KVPacked", fn, forward_impl, backward_impl, 0, 2, 2,
                                     packed_travel=packed_travel)
        qkv_fn = make_packed_function(fn.__name__ + "QKVPacked", fn, forward_impl, backward_impl, 0, 2, 3)

    def func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
             alibi_slopes=None, deterministic=False, return_attn_probs=False, group=None):
        return fn.apply(q, k, v, dropout_p, softmax_scale, causal, window_size, alibi_slopes,
                        deterministic, return_attn_probs, group)

    def kvpacked_func(q, kv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                      alibi_slopes=None, deterministic=False, return_attn_probs=False, group=None):
        if kv_fn is not None:
            return kv_fn.apply(q, kv, dropout_p, softmax_scale, causal, window_size, alibi_slopes,
                               deterministic, return_attn_probs, group)
        return fn.apply(q, kv[:, :, 0], kv[:, :, 1], dropout_p, softmax_scale, causal, window_size,
                        alibi_slopes, deterministic, return_attn_probs, group)

    def qkvpacked_func(qkv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                       alibi_slopes=None, deterministic=False, return_attn_probs=False, group=None):
        if qkv_fn is not None:
            return qkv_fn.apply(qkv, dropout_p, softmax_scale, causal, window_size, alibi_slopes,
                                deterministic, return_attn_probs, group)
        return fn.apply(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], dropout_p, softmax_scale, causal,
                        window_size, alibi_slopes, deterministic, return_attn_probs, group)

    must_be_causal = prefix in ("zigzag_ring_flash_attn", "stripe_flash_attn")

    def lower(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), alibi_slopes=None,
              deterministic=False, return_attn_probs=False, group=None):
        from ._ops import single_device_attention

        _check_unsupported(dropout_p, window_size, alibi_slopes, windows_ok=True)
        assert causal or not must_be_causal, f"{prefix} is meaningless for causal=False"
        return single_device_attention(q, k, v, None, 0, softmax_scale, causal, return_attn_probs, window_size)

    def lower_kv(q, kv, *a, **kw):
        return lower(q, kv[:, :, 0], kv[:, :, 1], *a, **kw)

    def lower_qkv(qkv, *a, **kw):
        return lower(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], *a, **kw)

    multi = multi_kv = multi_qkv = None
    if forward_impl is not None:
        from . import _ops

        _ops.register_schedule(prefix, forward_impl, backward_impl)

        def multi(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), alibi_slopes=None,
                  deterministic=False, return_attn_probs=False, group=None):
            _check_unsupported(dropout_p, window_size, alibi_slopes, windows_ok=False)
            assert causal or not must_be_causal, f"{prefix} is meaningless for causal=False"
            return _ops.multi_rank_attention(prefix, q, k, v, None, 0, softmax_scale, causal, return_attn_probs, group)

        def multi_kv(q, kv, *a, **kw):
            return multi(q, kv[:, :, 0], kv[:, :, 1], *a, **kw)

        def multi_qkv(qkv, *a, **kw):
            return multi(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], *a, **kw)

    for f, suffix in ((func, "func"), (kvpacked_func, "kvpacked_func"), (qkvpacked_func, "qkvpacked_func")):
        f.__name__ = f.__qualname__ = f"{prefix}_{suffix}"
    return (_compilable(func, lower, multi), _compilable(kvpacked_func, lower_kv, multi_kv),
            _compilable(qkvpacked_func, lower_qkv, multi_qkv))


def make_varlen_api(fn, prefix, forward_impl=None, backward_impl=None):
    """(T,H,D) + (cu_seqlens, max_seqlen) API.  With the schedule's forward / backward the packed entry points get
    their own autograd Function (gradients land in ONE packed buffer, see make_packed_function)."""
    kv_fn = qkv_fn = None
    if forward_impl is not None:
        kv_fn = make_packed_function(fn.__name__ + "KVPacked", fn, forward_impl, backward_impl, 2, 1, 2)
        qkv_fn = make_packed_function(fn.__name__ + "QKVPacked", fn, forward_impl, backward_impl, 2, 1, 3)

    def func(q, k, v, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
             window_size=(-1, -1), alibi_slopes=None, deterministic=False, return_attn_probs=False,
             group=None):
        return fn.apply(q, k, v, cu_seqlens, max_seqlen, dropout_p, softmax_scale, causal, window_size,
                        alibi_slopes, deterministic, return_attn_probs, group)

    def kvpacked_func(q, kv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
                      window_size=(-1, -1), alibi_slopes=None, deterministic=False,
                      return_attn_probs=False, group=None):
        if kv_fn is not None:
            return kv_fn.apply(q, kv, cu_seqlens, max_seqlen, dropout_p, softmax_scale, causal, window_size,
                               alibi_slopes, deterministic, return_attn_probs, group)
        return fn.apply(q, kv[:, 0], kv[:, 1], cu_seqlens, max_seqlen, dropout_p, softmax_scale, causal,
                        window_size, alibi_slopes, deterministic, return_attn_probs, group)

    def qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
                       window_size=(-1, -1), alibi_slopes=None, deterministic=False,
                       return_attn_probs=False, group=None):
        if qkv_fn is not None:
            return qkv_fn.apply(qkv, cu_seqlens, max_seqlen, dropout_p, softmax_scale, causal, window_size,
                                alibi_slopes, deterministic, return_attn_probs, group)
        return fn.apply(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu_seqlens, max_seqlen, dropout_p, softmax_scale,
                        causal, window_size, alibi_slopes, deterministic, return_attn_probs, group)

    must_be_causal = prefix.startswith("zigzag")

    def lower(q, k, v, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
              window_size=(-1, -1), alibi_slopes=None, deterministic=False, return_attn_probs=False, group=None):
        from ._ops import single_device_attention

        _check_unsupported(dropout_p, window_size, alibi_slopes, windows_ok=True)
        assert causal or not must_be_causal, f"{prefix} is meaningless for causal=False"
        return single_device_attention(q, k, v, cu_seqlens, max_seqlen, softmax_scale, causal, return_attn_probs,
                                       window_size)

    def lower_kv(q, kv, *a, **kw):
        return lower(q, kv[:, 0], kv[:, 1], *a, **kw)

    def lower_qkv(qkv, *a, **kw):
        return lower(qkv[:, 0], qkv[:, 1], qkv[:, 2], *a, **kw)

    multi = multi_kv = multi_qkv = None
    if forward_impl is not None:
        from . import _ops

        _ops.register_schedule(prefix, forward_impl, backward_impl)

        def multi(q, k, v, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
                  window_size=(-1, -1), alibi_slopes=None, deterministic=False, return_attn_probs=False, group=None):
            _check_unsupported(dropout_p, window_size, alibi_slopes, windows_ok=False)
            assert causal or not must_be_causal, f"{prefix} is meaningless for causal=False"
            return _ops.multi_rank_attention(prefix, q, k, v, cu_seqlens, max_seqlen, softmax_scale, causal,
                                             return_attn_probs, group)

        def multi_kv(q, kv, *a, **kw):
            return multi(q, kv[:, 0], kv[:, 1], *a, **kw)

        def multi_qkv(qkv, *a, **kw):
            return multi(qkv[:, 0], qkv[:, 1], qkv[:, 2], *a, **kw)

    for f, suffix in ((func, "func"), (kvpacked_func, "kvpacked_func"), (qkvpacked_func, "qkvpacked_func")):
        f.__name__ = f.__qualname__ = f"{prefix}_{suffix}"
    return (_compilable(func, lower, multi), _compilable(kvpacked_func, lower_kv, multi_kv),
            _compilable(qkvpacked_func, lower_qkv, multi_qkv))
