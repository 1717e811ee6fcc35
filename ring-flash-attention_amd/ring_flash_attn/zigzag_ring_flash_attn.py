"""Zigzag (causal-balanced) ring attention — the headline path.

Same public surface and step semantics as
/root/reference/ring_flash_attn/zigzag_ring_flash_attn.py (forward :7-88, backward :91-199,
autograd Function :202-265, wrappers :268-346): rank r holds sequence chunks r and 2W-1-r;
at step s it sees the K/V of rank (r-s) mod W and computes
    s == 0      : local causal block
    0 < s <= r  : all queries  x first-half keys   (no mask)
    s > r       : second-half queries x all keys   (no mask)
MI355X-first differences:
  * every block result is merged into fp32 (out, lse) accumulators INSIDE the attention
    kernel's epilogue (rfa_fwd accumulate mode) — no block_out round trip, no merge kernels,
    and half-slices are plain strided views (pointer offsets), never copies;
  * lse is kept contiguous (B,H,S) end to end (the reference returns a transposed view);
  * backward: delta = rowsum(dO*O) once per call; dQ accumulates in fp32 inside the dQ kernel;
    dK/dV partials are group-summed straight into the travelling fp32 accumulators, split in
    two phases so the kernels overlap the arrival of those accumulators;
  * world_size == 1 short-circuits to a single kernel writing q.dtype directly;
  * two exchange forms (config.zigzag_exchange / RFA_ZIGZAG_EXCHANGE = auto | gather | ring, default auto):
      ring    the reference's hop-by-hop protocol: per step one batched isend/irecv of K/V (and of the
              travelling fp32 dK/dV accumulators in the backward) to / from the ring neighbours, posted under
              the side stream before the step's kernels and waited after them.  O(S/W) memory per rank.
      gather  mesh-aware: ONE all-gather of K and V (overlapped with the local causal block) replaces the
              W-1 neighbour hops of the forward; in the backward one all-gather plus ONE exchange of the
              per-chunk dK/dV contributions (every rank returns chunk c's contribution to its owner c, the
              owner sums the W arrivals in fp32) replace 2(W-1)+W hops.  A neighbour ring drives 1 of a
              rank's 7 xGMI links and puts a transfer on the critical path of every step (33.5 MB bf16 K/V
              against a 0.5 ms forward step at the headline shape, 100 MB incl. fp32 dK/dV in the
              backward); the collectives use the whole mesh and put one transfer, not W, on it.
              Cost: scratch of W x (K,V) in the io dtype plus W x (dK,dV) contributions per rank — O(S_total)
              instead of O(S_total/W): 0.27 + 0.27 GB at W = 8, Hk = 8, S = 8192/rank (0.54 GB with fp32
              contributions), 4.3 + 4.3 GB at 128K tokens/rank.
      gather_ps  (round 6) the gather form with PER-SOURCE arrival: the W - 1 K/V exchanges are posted at once, in
              consumption order, each its own isend/irecv pair (utils.SourceArrivals), and step s waits for source
              (r - s) mod W only — the reference's "step s starts when hop s has landed"
              (zigzag_ring_flash_attn.py:60-84) without putting a hop on the critical path of every step.  Same kernels,
              same scratch, same backward return path (one all-to-all) as `gather`; bit-identical results.
      auto    the form the library MEASURED to be faster on this group for these shapes (first multi-rank call on an
              RCCL group: tuning.autotune_zigzag_exchange, a collective decision); without a measurement gather while
              that scratch stays below config.gather_max_bytes (default 4 GiB), ring beyond — long contexts keep ring
              attention's memory scaling.  Both rules give every rank the same answer.
    The per-step kernels, their arguments and the merge order are the same in both forms (the forward is
    bit-identical; dK/dV differ by summation order / rounding point only).
  * dK/dV contributions of the gather form travel in the io dtype by default (RFA_DKV_WIRE = io | fp32):
    every block's dK/dV is rounded to bf16 once and summed in fp32 at the owner — exactly the rounding
    points of the reference, whose flash_attn calls return bf16 block gradients that are then added into
    fp32 buffers (zigzag_ring_flash_attn.py:137-139,164-187) — at half the xGMI bytes of fp32.
"""
import torch

from . import _C, config
from .backend import get_backend
from .utils import Agreement, AllGatherComm, RingComm, SourceArrivals, all_to_all_async, reduce_scatter_async, single_rank
from ._api import make_autograd_function, make_dense_api, _grad_buffers
from ._common import packed_pair, dropout_arg


def gather_scratch_bytes(k: torch.Tensor, world: int, wire_fp32: bool) -> int:
    """per-rank scratch of the gather exchange: W x (K, V) io dtype + W x (dK, dV) contributions"""
    return world * 2 * k.numel() * (k.element_size() + (4 if wire_fp32 else k.element_size()))


def _wire_fp32() -> bool:
    return config.get().dkv_wire_fp32


def exchange_mode(k: torch.Tensor, world: int, q: torch.Tensor = None, group=None, v: torch.Tensor = None) -> str:
    """config.zigzag_exchange (RFA_ZIGZAG_EXCHANGE) = gather | ring forces a form.  auto (default): the form a
    MEASUREMENT on this group recorded for these shapes (tuning.autotune_zigzag_exchange — a collective: fwd + bwd a few
    times in each form on scratch tensors, max over ranks, so every rank records the same winner; bench.py runs it in its
    warm-up, a training script may call it once per shape; config.autotune / RFA_AUTOTUNE=1 lets the library take it
    inside the first multi-rank call of a (group, shapes) pair — opt-in since round 5: it costs 2 forms x 4 fwd + bwd at
    the caller's peak memory) — else, nothing measured or measurable, gather while its O(S_total) scratch stays below
    config.gather_max_bytes, ring beyond.  Every input of the decision is group-consistent: the configuration, the
    shapes, and a record only after tuning.agreed_lookup established that all ranks hold the same one."""
    cfg = config.get()
    mode = cfg.zigzag_exchange
    if mode == "auto":
        if q is not None:
            from . import tuning

            # (a record counts only once the group has established that every rank holds the same one: ranks that
            #  disagree about the form post different collectives)
            tuned = tuning.agreed_lookup(q.shape, k.shape, q.dtype, world, group, q.device)
            if tuned is None and v is not None and cfg.autotune and tuning.can_measure(group, q):
                tuned = tuning.measure_in_call(group, q, k, v)
            if tuned is not None:
                return tuned
        mode = "gather" if gather_scratch_bytes(k, world, cfg.dkv_wire_fp32) <= cfg.gather_max_bytes else "ring"
    return mode


# ---------------------------------------------------------------------------------------------
# The K/V gathered by a forward are kept for its backward (W x (K, V) io dtype: 0.27 GB at W = 8, Hk = 8,
# S = 8192/rank) instead of being gathered a second time.  With the K/V already present the backward can run its
# REMOTE steps first and the local causal block last, so that the one all-to-all of the dK/dV contributions is
# posted before the local block and runs beside it — no exchange is left on the critical path of the backward.
# The buffers are SAVED TENSORS of the autograd node (_api._keep_list): owned by the graph, freed with it, dropped
# and re-made under activation checkpointing, never created for a forward without a backward.  A forward whose
# gathered K/V exceed config.kv_keep_bytes (default 4 GiB per call; config.kv_keep = False: never), or whose K/V would
# push the kept buffers of ALL pending backwards of the process over config.kv_keep_total_bytes (default 4 GiB: an
# L-layer model without activation checkpointing holds L of them — 0.27 GB each at W = 8, Hk = 8, S = 8192 per rank —
# where the reference saves only the local k / v), keeps nothing: its backward gathers again and keeps the
# local-block-first order.  The reservation (config.kept_budget) is released when the backward has run or the graph
# is freed.
def _try_keep(keep, bufs, process_group):
    """Whether THIS rank has room is a rank-local fact (config.kept_budget counts the live kept bytes of the process, and
    graph lifetimes may differ between ranks: an output held for logging on rank 0 only, garbage-collection timing) — but
    the backward's collective sequence depends on it (kept: no second all-gather).  So every rank that could keep posts
    its flag into ONE tiny all-reduce (utils.Agreement, beside the all-gather on the side stream, no host stall) and the
    backward uses the kept buffers only when EVERY rank kept (_api._split_kept resolves the flag); otherwise all ranks
    gather again, the ones that did keep drop their buffers."""
    if keep is None:
        return
    if bufs[0].is_cuda and torch.cuda.is_current_stream_capturing():
        return                      # (the flag cannot be read back inside a capture; every rank captures alike)
    nbytes = sum(b_.numel() * b_.element_size() for b_ in bufs)
    if not config.kept_budget.may_keep(nbytes):
        # refused by the configuration and the call's size alone (kv_keep off, or this call over kv_keep_bytes): the same
        # answer on every rank, so nothing to agree on — no all-reduce, no pinned allocation, no event per layer;
        # `keep.agreed` stays None and the backward gathers again
        return
    token = config.kept_budget.try_reserve(nbytes)
    if token is not None:
        keep.extend(bufs)
        keep.token = token
    keep.agreed = Agreement(process_group, token is not None, bufs[0].device)


def _kv_views(bufs, k, world):
    """(k_all, v_all) views — indexed by source rank — of the gathered base buffers `bufs`"""
    if len(bufs) == 1:                                   # packed kv travelled as one buffer
        kv_all = bufs[0].view((world, k.shape[0], k.shape[1], 2) + tuple(k.shape[2:]))
        return kv_all.select(-3, 0), kv_all.select(-3, 1)
    return bufs[0].view((world,) + tuple(k.shape)), bufs[1].view((world,) + tuple(k.shape))


def _gather_kv(comm_group, k, v, world, per_source=False):
    """posts the all-gather of k and v; returns (handle, bufs, k_all, v_all): `bufs` the gathered base buffers,
    *_all[(src rank)] views of them.  k and v that are the two halves of one packed kv tensor (kvpacked entry points)
    travel as that one buffer: one collective of twice the size instead of two, no contiguous copies; the per-rank
    K / V are then strided views of the gathered buffer.
    per_source (the `gather_ps` form): W - 1 exchanges posted in consumption order instead of the one collective — the
    handle is a utils.SourceArrivals whose wait(step) waits for source (rank - step) mod W only; same buffers and views
    (the own slot stays unwritten: the schedules use the local k / v there)."""
    kv = packed_pair(k, v)
    if kv is not None:
        kv_cat = torch.empty((world * kv.shape[0],) + tuple(kv.shape[1:]), dtype=kv.dtype, device=kv.device)
        bufs, locals_ = [kv_cat], [kv]
    else:
        # (world*B, ...) for the collective (the concatenated form every backend accepts), (world, B, ...) to index
        k_cat = torch.empty((world * k.shape[0],) + tuple(k.shape[1:]), dtype=k.dtype, device=k.device)
        v_cat = torch.empty((world * v.shape[0],) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        bufs, locals_ = [k_cat, v_cat], [k.contiguous(), v.contiguous()]
    if per_source:
        gather = SourceArrivals(comm_group).post(locals_, [b_.view((world,) + tuple(t_.shape)) for b_, t_ in zip(bufs, locals_)])
    else:
        gather = AllGatherComm(comm_group)
        for b_, t_ in zip(bufs, locals_):
            gather.all_gather(b_, t_)
    return (gather, bufs) + _kv_views(bufs, k, world)


def zigzag_ring_flash_attn_forward(
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
    keep=None,
):
    """keep: a list (or None) the gather form appends its gathered K/V buffers to — the autograd Function saves them
    for the backward (`kept=`)"""
    assert causal == True, "zigzag ring is meaningless for causal=False"
    be = get_backend()
    comm = RingComm(process_group)
    B, S, H, D = q.shape
    half = S // 2

    if single_rank(comm.world_size):
        out = torch.empty_like(q)
        lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
        be.fwd(q, k, v, softmax_scale=softmax_scale, causal=True, out=out, lse=lse, window=window_size, dropout=dropout_arg(dropout_p, dropout_seed))
        return out, lse
    assert not dropout_p, "dropout over a multi-rank ring is not supported (as in the reference)"

    out_acc = torch.empty((B, S, H, D), dtype=torch.float32, device=q.device)
    lse_acc = torch.empty((B, H, S), dtype=torch.float32, device=q.device)

    mode = exchange_mode(k, comm.world_size, q, process_group, v)
    if mode in ("gather", "gather_ps"):
        gather, bufs, k_all, v_all = _gather_kv(process_group, k, v, comm.world_size, per_source=mode == "gather_ps")
        be.fwd(q, k, v, softmax_scale=softmax_scale, causal=True,          # runs beside the exchange
               out_acc=out_acc, lse_acc=lse_acc, acc_init=True)
        if mode == "gather":
            gather.wait()                                                  # one collective: everything or nothing
        _try_keep(keep, bufs, process_group)
        for step in range(1, comm.world_size):
            src = (comm.rank - step) % comm.world_size
            if mode == "gather_ps":
                gather.wait(step)                                          # source (r - step) has landed; later ones fly on
            ks, vs = k_all[src], v_all[src]
            if step <= comm.rank:
                be.fwd(q, ks[:, :half], vs[:, :half], softmax_scale=softmax_scale, causal=False,
                       out_acc=out_acc, lse_acc=lse_acc)
            else:
                be.fwd(q[:, half:], ks, vs, softmax_scale=softmax_scale, causal=False,
                       out_acc=out_acc[:, half:], lse_acc=lse_acc[:, :, half:])
        return be.cast(out_acc, q.dtype), lse_acc

    next_k, next_v = None, None

    for step in range(comm.world_size):
        if step + 1 != comm.world_size:
            next_k, next_v = comm.send_recv_kv(k, v)

        if step == 0:
            be.fwd(q, k, v, softmax_scale=softmax_scale, causal=True,
                   out_acc=out_acc, lse_acc=lse_acc, acc_init=True)
        elif step <= comm.rank:
            be.fwd(q, k[:, :half], v[:, :half], softmax_scale=softmax_scale, causal=False,
                   out_acc=out_acc, lse_acc=lse_acc)
        else:
            be.fwd(q[:, half:], k, v, softmax_scale=softmax_scale, causal=False,
                   out_acc=out_acc[:, half:], lse_acc=lse_acc[:, :, half:])

        if step + 1 != comm.world_size:
            comm.wait()
            k, v = next_k, next_v

    out = be.cast(out_acc, q.dtype)
    return out, lse_acc


def zigzag_ring_flash_attn_backward(
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
    kept=None,
):
    """kept: the gathered K/V buffers the forward handed over (`keep=`), or None"""
    assert causal == True, "zigzag ring is meaningless for causal=False"
    be = get_backend()
    kv_comm = RingComm(process_group)
    d_kv_comm = RingComm(process_group)
    B, S, H, D = q.shape
    half = S // 2
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

    mode = exchange_mode(k, kv_comm.world_size, q, process_group, v)
    if mode in ("gather", "gather_ps"):
        W, rank = kv_comm.world_size, kv_comm.rank
        wire32 = _wire_fp32()
        per_source = mode == "gather_ps"
        if kept:
            gather, (k_all, v_all) = None, _kv_views(kept, k, W)
        else:
            gather, _, k_all, v_all = _gather_kv(process_group, k, v, W, per_source=per_source)
        # K/V of every rank are here already: remote steps first, local block beside the all-to-all (the fp32
        # reduce-scatter wire keeps the local-first order)
        local_last = gather is None and not wire32
        # per-chunk contributions of THIS rank's queries: slot c holds this rank's dK/dV for the chunk owned by
        # rank c.  Every slot is written exactly once (a "front" step produces only the first half of its
        # chunk: the other half is zero-filled here, r half-chunks instead of the whole buffer), so the dK/dV
        # kernel stores straight into it: fp32 (BWD_KV_OVERWRITE) or the io dtype — no workspace, no reduction
        # pass.  The owner then sums the W arrivals of its chunk in fp32.
        wdt = torch.float32 if wire32 else q.dtype
        # k, v that are one packed kv tensor: the contributions are packed the same way and travel as ONE buffer
        packed = packed_pair(k, v) is not None
        if packed:
            pshape = (k.shape[0], k.shape[1], 2) + tuple(k.shape[2:])
            dkv_cat = torch.empty((W * pshape[0],) + pshape[1:], dtype=wdt, device=q.device)
            dkv_all = dkv_cat.view((W,) + pshape)
            dk_all, dv_all = dkv_all.select(-3, 0), dkv_all.select(-3, 1)
        else:
            dk_cat = torch.empty((W * k.shape[0],) + tuple(k.shape[1:]), dtype=wdt, device=q.device)
            dv_cat = torch.empty((W * v.shape[0],) + tuple(v.shape[1:]), dtype=wdt, device=q.device)
            dk_all, dv_all = dk_cat.view((W,) + tuple(k.shape)), dv_cat.view((W,) + tuple(v.shape))
        cats = [dkv_cat] if packed else [dk_cat, dv_cat]

        def slots(src, rows):
            if wire32:
                return dict(dk_acc=dk_all[src][:, rows], dv_acc=dv_all[src][:, rows])
            return dict(dk=dk_all[src][:, rows], dv=dv_all[src][:, rows])

        full = slice(None)
        if not local_last:
            be.bwd(dout, q, k, v, softmax_lse, delta, softmax_scale=softmax_scale, causal=True,
                   dq_acc=dq, acc_init=True, deterministic=deterministic, **slots(rank, full))   # beside the all-gather
            if gather is not None and not per_source:
                gather.wait()
        elif rank == 0:
            dq[:, :half].zero_()             # every remote step of rank 0 covers the second half of the queries only
        first = local_last                   # the first kernel that touches dq initialises the rows it covers
        for step in range(1, W):
            src = (rank - step) % W
            if gather is not None and per_source:
                gather.wait(step)            # (a backward that gathers again: the remote steps consume in arrival order too)
            ks, vs = k_all[src], v_all[src]
            init, first = first, False
            if step <= rank:
                if packed:
                    dkv_all[src][:, half:].zero_()
                else:
                    dk_all[src][:, half:].zero_()
                    dv_all[src][:, half:].zero_()
                be.bwd(dout, q, ks[:, :half], vs[:, :half], softmax_lse, delta, softmax_scale=softmax_scale,
                       causal=False, dq_acc=dq, acc_init=init, deterministic=deterministic,
                       phases=_C.BWD_KV_OVERWRITE, **slots(src, slice(0, half)))
            else:
                be.bwd(dout[:, half:], q[:, half:], ks, vs, softmax_lse[:, :, half:], delta[:, :, half:],
                       softmax_scale=softmax_scale, causal=False, dq_acc=dq[:, half:], acc_init=init,
                       deterministic=deterministic, phases=_C.BWD_KV_OVERWRITE, **slots(src, full))
        if wire32:
            sums = [torch.empty((c.shape[0] // W,) + tuple(c.shape[1:]), dtype=torch.float32, device=q.device) for c in cats]
            works = [reduce_scatter_async(s_, c, group=process_group) for s_, c in zip(sums, cats)]
            dq_out = be.cast(dq, q.dtype)                                  # runs beside the exchange
            for w_ in works:
                w_.wait()
            if packed:
                dkv = be.cast(sums[0], q.dtype)
                return dq_out, dkv.select(-3, 0), dkv.select(-3, 1)
            return dq_out, be.cast(sums[0], q.dtype), be.cast(sums[1], q.dtype)
        ins = [torch.empty_like(c) for c in cats]
        works = [all_to_all_async(i_, c, group=process_group) for i_, c in zip(ins, cats)]
        if local_last:
            # the local causal block runs beside the exchange; its contribution (this rank's own chunk) does not
            # travel: it is written next to the arrivals once they are in (the self chunk of the all-to-all
            # carried nothing)
            own = [torch.empty((c.shape[0] // W,) + tuple(c.shape[1:]), dtype=wdt, device=q.device) for c in cats]
            own_kv = (dict(dk=own[0].select(-3, 0), dv=own[0].select(-3, 1)) if packed else dict(dk=own[0], dv=own[1]))
            # (a one-rank group on the forced multi-step path has no remote step: the local block is then the first —
            #  and only — kernel that touches dq)
            be.bwd(dout, q, k, v, softmax_lse, delta, softmax_scale=softmax_scale, causal=True,
                   dq_acc=dq, acc_init=W == 1, deterministic=deterministic, **own_kv)
        dq_out = be.cast(dq, q.dtype)                                      # runs beside the exchange
        for w_ in works:
            w_.wait()
        if local_last:
            for i_, o_ in zip(ins, own):
                i_.view((W,) + tuple(o_.shape))[rank].copy_(o_)
        # owner side: sum the W arrivals of this rank's chunk in fp32, straight into the caller's gradient
        # buffers when it provided them (the packed kv gradient of the kvpacked entry points)
        _, dk, dv = _grad_buffers(out_grads, None, k, v)
        dst = packed_pair(dk, dv) if packed else None
        if dst is not None:
            be.sum_slots(ins[0].view((W,) + tuple(dst.shape)).flatten(-3, -2), dst.flatten(-3, -2))
        elif packed:
            arr = ins[0].view((W,) + pshape)
            be.sum_slots(arr.select(-3, 0), dk)
            be.sum_slots(arr.select(-3, 1), dv)
        else:
            be.sum_slots(ins[0].view((W,) + tuple(k.shape)), dk)
            be.sum_slots(ins[1].view((W,) + tuple(v.shape)), dv)
        return dq_out, dk, dv

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
                   dq_acc=dq, dk_acc=dk, dv_acc=dv, acc_init=True, deterministic=deterministic)
        else:
            front = step <= kv_comm.rank
            if front:
                args = (dout, q, k[:, :half], v[:, :half], softmax_lse, delta)
                dq_view = dq
            else:
                args = (dout[:, half:], q[:, half:], k, v, softmax_lse[:, :, half:], delta[:, :, half:])
                dq_view = dq[:, half:]
            # phase 1: dQ (+= in fp32) and per-head dK/dV partials — overlaps the dk/dv transfer
            part = be.bwd(*args, softmax_scale=softmax_scale, causal=False, dq_acc=dq_view,
                          dk_acc=dk, dv_acc=dv, deterministic=deterministic, phases=_C.BWD_COMPUTE)

            d_kv_comm.wait()
            dk_comm_buffer, dv_comm_buffer = dk, dv
            dk, dv = next_dk, next_dv

            # phase 2: add this step's dK/dV into the accumulators that just arrived
            if front:
                be.bwd(*args, softmax_scale=softmax_scale, causal=False, dq_acc=dq_view,
                       dk_acc=dk[:, :half], dv_acc=dv[:, :half], deterministic=deterministic,
                       phases=_C.BWD_REDUCE, partials=part)
            else:
                be.bwd(*args, softmax_scale=softmax_scale, causal=False, dq_acc=dq_view,
                       dk_acc=dk, dv_acc=dv, deterministic=deterministic, phases=_C.BWD_REDUCE,
                       partials=part)

        if step + 1 != kv_comm.world_size:
            kv_comm.wait()
            k, v = next_k, next_v

        next_dk, next_dv = d_kv_comm.send_recv_kv(dk, dv, dk_comm_buffer, dv_comm_buffer)

    d_kv_comm.wait()

    return be.cast(dq, q.dtype), be.cast(next_dk, q.dtype), be.cast(next_dv, q.dtype)


zigzag_ring_flash_attn_forward.keeps_for_backward = True

ZigZagRingFlashAttnFunc = make_autograd_function(
    "ZigZagRingFlashAttnFunc", zigzag_ring_flash_attn_forward, zigzag_ring_flash_attn_backward, 0)
(
    zigzag_ring_flash_attn_func,
    zigzag_ring_flash_attn_kvpacked_func,
    zigzag_ring_flash_attn_qkvpacked_func,
) = make_dense_api(ZigZagRingFlashAttnFunc, "zigzag_ring_flash_attn", zigzag_ring_flash_attn_forward,
                   zigzag_ring_flash_attn_backward, packed_travel=True)
