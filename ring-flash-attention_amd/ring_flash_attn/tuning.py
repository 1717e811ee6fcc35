"""Measured choice of the zigzag exchange form, and the communication probes behind it.

The dense zigzag schedule has two exchange forms (zigzag_ring_flash_attn.py: `ring` = the reference's neighbour
hops, /root/reference/ring_flash_attn/utils.py:98-151; `gather` = one all-gather + one all-to-all over the whole
xGMI mesh).  Which one is faster depends on the node (links per GPU pair, RCCL version, how the RCCL kernels share the
CUs with a 256-CU attention grid — SURVEY H3), so the default `auto` can be backed by a measurement:

    autotune_zigzag_exchange(group, q, k, v)     runs the real forward + backward of the schedule a few times in each
        form on scratch copies of the caller's shapes, takes the MAX over ranks of the per-iteration time (one
        all_reduce, so every rank reaches the same decision) and records the winner for (shapes, dtype, world size).
        `zigzag_ring_flash_attn.exchange_mode` consults that record before its shape rule.  A form that raises (a
        collective the installed RCCL does not support: the same error on every rank) is disqualified and the other
        one chosen instead of failing the job; a failure on some ranks only leaves the others inside the collective,
        which no caller can repair.
    comm_probe(group, device, nbytes)            achieved GB/s per rank of the three transfer kinds the schedules use
        (all-gather, all-to-all, one neighbour hop) at a given message size.

bench.py calls both in its warm-up at N > 1 and reports them in its `comm` block; tools/xgmi_probe.py is the
stand-alone version.  The record is a tuning cache (like a GEMM autotuner's): written once per shape under a lock,
read-only afterwards; it never holds tensors.
"""
import collections
import threading
import time
import weakref

import torch
import torch.distributed as dist

from . import config, utils

_LOCK = threading.Lock()
_TUNED = {}          # (group ranks, world, B, S, H, Hk, D, dtype) -> "gather" | "gather_ps" | "ring"
_REPORTS = {}        # same key -> the measurement (for bench.py / logs)
_PENDING = {}        # key -> form: a record installed by hand AFTER the group agreed on the key; waits for sync_records()
# group instance -> the keys on which that group has established that every rank holds the SAME record (or none).  What a
# rank finds here decides whether it POSTS the agreement all-reduce, so the set may only change in ways every rank of the
# group repeats: agreed_lookup / autotune_zigzag_exchange / sync_records (collectives) add and drop entries, and the
# per-group bound evicts in insertion order — the same order on every rank, which all make the same collective calls on a
# group in the same sequence.  A rank-local event (record(), a tuning file read by some ranks) never touches it.
_AGREED = {}
_AGREED_MAX = 1024   # keys per group (sequence lengths that vary from step to step must not grow the set for ever)
_CODES = {None: 0, "gather": 1, "ring": 2, "gather_ps": 3}


def _is_agreed(ginst, key) -> bool:
    return key in _AGREED.get(ginst, ())


def _set_agreed(ginst, key):
    d = _AGREED.setdefault(ginst, collections.OrderedDict())
    d[key] = True
    while len(d) > _AGREED_MAX:
        d.popitem(last=False)


_GROUP_SERIAL = {}   # (group name, id) -> (weak reference to the group object, serial)
_SERIAL = [0]


def _group_key(group):
    """identity of a process group: the sorted global ranks of its members.  A record measured on one group must not
    decide for another group of the same size (different links; and its peers may never have measured: ranks that
    disagree about the exchange form hang)"""
    try:
        if group is None:
            return "default"
        return tuple(sorted(dist.get_process_group_ranks(group)))
    except Exception:
        return ("group", id(group))


def _group_instance(group):
    """identity of this INSTANCE of the group: a group re-created after destroy_process_group() (a restart of some ranks,
    a test that re-initialises) gets the old one's name ('0' for WORLD) and possibly its `id()` — and must agree again.
    So the identity is a serial number handed to the group OBJECT, found again through a weak reference to it."""
    try:
        g = dist.group.WORLD if group is None else group
        k = (getattr(g, "group_name", None), id(g))
        ent = _GROUP_SERIAL.get(k)
        if ent is not None and ent[0]() is g:
            return ent[1]
        _SERIAL[0] += 1
        _GROUP_SERIAL[k] = (weakref.ref(g), _SERIAL[0])
        if len(_GROUP_SERIAL) > 64:
            for k_ in [k_ for k_, e_ in _GROUP_SERIAL.items() if e_[0]() is None]:
                _AGREED.pop(_GROUP_SERIAL.pop(k_)[1], None)
        return _SERIAL[0]
    except Exception:
        return ("unidentified", id(group))


def _key(world, q_shape, k_shape, dtype, group=None):
    B, S, H, D = q_shape
    return (_group_key(group), int(world), int(B), int(S), int(H), int(k_shape[2]), int(D), str(dtype))


def lookup(q_shape, k_shape, dtype, world, group=None):
    """the recorded exchange form for this problem on this group, or None — THIS rank's record; the schedules use
    `agreed_lookup`"""
    return _TUNED.get(_key(world, q_shape, k_shape, dtype, group))


def record(q_shape, k_shape, dtype, world, form, group=None):
    """install a record by hand (a tuning file, a test) — a RANK-LOCAL act, so it must not change which collectives this
    rank posts (dropping the agreement here made a rank that loaded a tuning file late post an all-reduce its
    peers never joined).  Before the group's first use of the key the record is simply stored: `agreed_lookup` then
    establishes, collectively, whether every rank holds the same one.  Once the group HAS agreed on the key, the new
    record is parked and the agreed state (record or none) keeps deciding until every rank calls `sync_records(group)`."""
    if form not in ("gather", "gather_ps", "ring"):
        raise ValueError(f"exchange form must be gather, gather_ps or ring, got {form!r}")
    key = _key(world, q_shape, k_shape, dtype, group)
    with _LOCK:
        if _is_agreed(_group_instance(group), key):
            _PENDING[key] = form
        else:
            _TUNED[key] = form


def sync_records(group=None):
    """COLLECTIVE (every rank of `group` calls it, at the same point of the program): records parked by `record()` are
    installed and the group's agreements are dropped, so that the next use of each key re-agrees — on every rank alike,
    because every rank dropped them here.  No communication happens in this call."""
    ginst, gkey = _group_instance(group), _group_key(group)
    with _LOCK:
        for key in [k_ for k_ in _PENDING if k_[0] == gkey]:
            _TUNED[key] = _PENDING.pop(key)
        _AGREED.pop(ginst, None)


def agreed_lookup(q_shape, k_shape, dtype, world, group, device):
    """The record every rank may act on.  A record is a rank-local fact (a tuning file read by some ranks, a process that
    joined later, a measurement that only part of the group took), and ranks that disagree about the exchange form post
    different collectives — so the FIRST use of a (group, shapes) pair in a process reduces the local record's code over
    the group (one MIN + MAX all-reduce of two integers, host-synchronous, once): all ranks hold the same record -> it is
    used; anything else (some without a record, different records) -> every rank forgets its own and the caller falls
    through to a measurement or the shape rule, which all ranks take alike.  Records written by
    `autotune_zigzag_exchange` are agreed by construction (it is a collective).  Not under stream capture (the result
    could not be read back): there only an already agreed record is used."""
    key = _key(world, q_shape, k_shape, dtype, group)
    ginst = _group_instance(group)
    if _is_agreed(ginst, key):
        return _TUNED.get(key)
    if utils._loopback() is not None or utils.single_rank(world):
        return _TUNED.get(key)             # (nobody to agree with; a one-rank group forced onto the multi-step path — the
                                           #  RCCL test on a one-GPU box — does run the reduction below)
    if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        return None
    code = _CODES[_TUNED.get(key)]
    on_host = utils.backend_of(group) == "gloo" or device.type != "cuda"
    t = torch.tensor([code, -code], dtype=torch.int32, device="cpu" if on_host else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hi, lo = int(t[0].item()), -int(t[1].item())
    with _LOCK:
        if hi != lo or hi == 0:
            if code and config.get().tuning_log:
                import sys

                sys.stderr.write(f"ring_flash_attn: exchange record {key} is not shared by every rank of the group: ignored\n")
            _TUNED.pop(key, None)
            _REPORTS.pop(key, None)
        _set_agreed(ginst, key)
    return _TUNED.get(key)


def report(q_shape, k_shape, dtype, world, group=None):
    return _REPORTS.get(_key(world, q_shape, k_shape, dtype, group))


def can_measure(group, q) -> bool:
    """the library measures by itself only where the measurement means something and cannot disturb the caller: device
    tensors on an RCCL group of several ranks (gloo groups are the CPU / shared-GPU test paths), no exchange loopback
    installed, outside stream capture, dynamo tracing and inference mode (the scratch tensors of the measurement would
    be inference tensors, which autograd refuses to save: ADVICE r4)"""
    if utils._loopback() is not None or not q.is_cuda:
        return False
    if utils.backend_of(group) == "gloo" or dist.get_world_size(group) < 2:
        return False
    if torch.compiler.is_compiling() or torch.cuda.is_current_stream_capturing() or torch.is_inference_mode_enabled():
        return False
    return True


_FAILED = set()      # keys whose in-call measurement failed (on every rank: the verdict is reduced): not tried again


def measure_in_call(group, q, k, v):
    """the implicit path of `exchange_mode` (config.autotune, opt-in): never raises — a measurement that fails (every form
    disqualified, on all ranks alike) leaves the shape rule in charge, and is not repeated"""
    key = _key(dist.get_world_size(group), q.shape, k.shape, q.dtype, group)
    if key in _FAILED:
        return None
    try:
        return autotune_zigzag_exchange(group, q, k, v)["chosen"]
    except RuntimeError:
        _FAILED.add(key)
        return None


def clear():
    with _LOCK:
        _TUNED.clear()
        _REPORTS.clear()
        _PENDING.clear()
        _AGREED.clear()
        _FAILED.clear()


def _sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)


def _max_over_ranks(value, group, dev):
    t = torch.tensor([value], dtype=torch.float64, device=dev if utils.backend_of(group) != "gloo" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.item()


def autotune_zigzag_exchange(group, q, k, v, iters=3, warm=1, modes=("gather", "gather_ps", "ring")):
    """Time fwd + bwd of zigzag_ring_flash_attn_func in every exchange form on tensors shaped like (q, k, v) and
    record the faster one.  Collective: every rank of `group` must call it with the same shapes.  Returns the
    report dict {"chosen", "ms": {form: ms per iteration, max over ranks}, "failed": {...}}."""
    from . import zigzag_ring_flash_attn as Z
    from .utils import group_rank_world

    world = group_rank_world(group)[1]
    key = _key(world, q.shape, k.shape, q.dtype, group)
    # Early return only on GROUP-CONSISTENT state: a report exists (only this collective writes one, on every rank) for a
    # key the group has agreed on.  A record installed by hand (`record()`: rank-local, no report) never lets a rank skip
    # the measurement its peers are inside — it is measured over.
    rep = _REPORTS.get(key)
    if rep is not None and _is_agreed(_group_instance(group), key):
        return rep
    dev = q.device
    # scratch data from a PRIVATE generator: the caller's default RNG stream must not depend on whether, or in which
    # order, shapes were measured
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED)

    def scratch(shape, dtype):
        return torch.randn(tuple(shape), device=dev, dtype=torch.float32, generator=gen).to(dtype)

    qs, ks, vs = (scratch(t.shape, t.dtype).requires_grad_(True) for t in (q, k, v))
    do = scratch(q.shape, q.dtype)
    ms, failed = {}, {}
    for mode in modes:
        with config.override(zigzag_exchange=mode), torch.enable_grad():
            ok = 1.0
            try:
                def one():
                    qs.grad = ks.grad = vs.grad = None
                    out = Z.zigzag_ring_flash_attn_func(qs, ks, vs, causal=True, group=group)
                    out.backward(do)

                for _ in range(warm):
                    one()
                _sync(dev)
                dist.barrier(group=group)
                _sync(dev)
                t0 = time.perf_counter()
                for _ in range(iters):
                    one()
                _sync(dev)
                el = (time.perf_counter() - t0) / iters * 1e3
            except Exception as e:          # a form that cannot run here loses; every rank learns it below
                ok, el = 0.0, float("inf")
                failed[mode] = f"{type(e).__name__}: {e}"
            # all ranks agree: a failure anywhere disqualifies the form everywhere
            bad = _max_over_ranks(1.0 - ok, group, dev)
            ms[mode] = float("inf") if bad > 0 else _max_over_ranks(el, group, dev)
    finite = {m: t for m, t in ms.items() if t != float("inf")}
    if not finite:
        raise RuntimeError(f"autotune_zigzag_exchange: every exchange form failed: {failed}")
    chosen = min(finite, key=finite.get)
    rep = {"chosen": chosen, "ms": {m: (None if t == float("inf") else t) for m, t in ms.items()}, "failed": failed,
           "iters": iters, "world": world}
    with _LOCK:
        _TUNED[key] = chosen
        _REPORTS[key] = rep
        _PENDING.pop(key, None)
        _set_agreed(_group_instance(group), key)         # a collective measurement: every rank recorded this winner
    if config.get().tuning_log and dist.get_rank(group) == 0:
        import sys

        sys.stderr.write(f"ring_flash_attn: zigzag exchange autotune {key}: {rep}\n")
    return rep


def comm_probe(group, device, nbytes, iters=5, warm=2):
    """GB/s per rank (bytes this rank SENDS / time, max time over ranks) of an all-gather of `nbytes` per rank, an
    all-to-all of `nbytes` per peer slot, and one neighbour hop of `nbytes` — the three transfers of the schedules.
    Collective over `group`."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    gloo = utils.backend_of(group) == "gloo"
    dev = torch.device("cpu") if gloo else device
    n = max(1, nbytes // 2)
    src = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    out = {}

    def timed(fn):
        for _ in range(warm):
            fn()
        _sync(dev)
        dist.barrier(group=group)
        _sync(dev)
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        _sync(dev)
        return _max_over_ranks((time.perf_counter() - t0) / iters, group, dev)

    gathered = torch.empty(world * n, dtype=torch.bfloat16, device=dev)
    t = timed(lambda: dist.all_gather_into_tensor(gathered, src, group=group))
    out["all_gather"] = {"bytes_sent": (world - 1) * n * 2, "ms": t * 1e3, "GBps": (world - 1) * n * 2 / t / 1e9 if world > 1 else None}
    a2a_in, a2a_out = torch.zeros(world * n, dtype=torch.bfloat16, device=dev), torch.empty(world * n, dtype=torch.bfloat16, device=dev)
    t = timed(lambda: dist.all_to_all_single(a2a_out, a2a_in, group=group))
    out["all_to_all"] = {"bytes_sent": (world - 1) * n * 2, "ms": t * 1e3, "GBps": (world - 1) * n * 2 / t / 1e9 if world > 1 else None}
    recv = torch.empty_like(src)
    nxt = dist.get_global_rank(group, (rank + 1) % world) if group is not None else (rank + 1) % world
    prv = dist.get_global_rank(group, (rank - 1) % world) if group is not None else (rank - 1) % world

    def hop():
        for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, nxt, group=group), dist.P2POp(dist.irecv, recv, prv, group=group)]):
            r.wait()

    t = timed(hop)
    out["neighbour_hop"] = {"bytes_sent": n * 2, "ms": t * 1e3, "GBps": n * 2 / t / 1e9}
    return out
