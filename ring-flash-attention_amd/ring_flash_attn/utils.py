"""Communication + merge helpers of the ring schedules.

Mirrors the public surface of /root/reference/ring_flash_attn/utils.py (`RingComm`,
`AllGatherComm`, `update_out_and_lse`, `flatten_varlen_lse`, `unflatten_varlen_lse`) with the
same method names and error behaviour, re-designed for RCCL over xGMI:

* `RingComm` posts ONE batched isend/irecv group per ring step under an explicit side HIP stream
  (`comm_stream`): the transfer is ordered after the compute stream at commit (send data ready,
  receive buffer free) and runs beside the attention kernel; `wait()` makes the compute stream
  wait for exactly that transfer — no host sync.  Collectives (`AllGatherComm`,
  `reduce_scatter_async`, `all_to_all_async`) are posted the same way.
* receive buffers are recycled across steps (two alternating sets) instead of a fresh
  `torch.empty_like` per step (reference utils.py:117).
* `update_out_and_lse` is one HIP kernel (csrc/rfa_aux.hip: merge_kernel) instead of ~6
  eager TorchScript ops; the schedules in this package do not even call it — they use the
  merge fused into the attention epilogue — it is kept for API parity.
"""
import weakref
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import config
from .backend import get_backend

__all__ = ["update_out_and_lse", "RingComm", "AllGatherComm", "flatten_varlen_lse", "unflatten_varlen_lse"]


def update_out_and_lse(
    out: Optional[torch.Tensor],
    lse: Optional[torch.Tensor],
    block_out: torch.Tensor,
    block_lse: torch.Tensor,
    slice_=None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Same contract as reference utils.py:53-73.  out fp32 (B,S,H,D); lse fp32 (B,S,H,1);
    block_out io dtype (B,Sq,H,D); block_lse fp32 (B,H,Sq).  Updates in place when possible."""
    be = get_backend()
    if out is None:
        if slice_ is not None:
            raise RuntimeError("first update_out_and_lse should not pass slice_ args")
        out = torch.empty(block_out.shape, dtype=torch.float32, device=block_out.device)
        B, S, H, _ = block_out.shape
        lse = torch.empty((B, S, H, 1), dtype=torch.float32, device=block_out.device)
        be.merge(out, lse.squeeze(-1).transpose(1, 2), block_out, block_lse, acc_init=True)
        return out, lse
    if slice_ is not None:
        o, l = out[slice_], lse[slice_]
    else:
        o, l = out, lse
    be.merge(o, l.squeeze(-1).transpose(1, 2), block_out, block_lse, acc_init=False)
    return out, lse


def flatten_varlen_lse(lse: torch.Tensor, cu_seqlens: torch.Tensor) -> torch.Tensor:
    """(batch, nheads, max_seqlen) -> (nheads, total)   (reference triton_utils.py:39-67)."""
    return get_backend().lse_flatten(lse, cu_seqlens)


def unflatten_varlen_lse(lse: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int) -> torch.Tensor:
    """(total, nheads, 1) -> (batch, nheads, max_seqlen)   (reference triton_utils.py:103-137)."""
    return get_backend().lse_unflatten(lse, cu_seqlens, max_seqlen)


# Test / measurement hooks (exchange loopback, forced multi-step path, host staging for gloo groups that share one GPU):
# their state and setters live in ring_flash_attn._testing, which nothing in the package imports; production carries
# this one `None`.
_TEST = None


def _loopback():
    return _TEST.loopback if _TEST is not None else None


_BACKEND_OF = {}


def backend_of(process_group) -> str:
    """the group's torch.distributed backend name, resolved ONCE per group OBJECT (not a `dist.get_backend()` per
    transfer).  The entry holds a weak reference to the group it was resolved for: after destroy_process_group() and a
    re-init in the same process (tests; a bench that switches gloo -> nccl) WORLD gets the name '0' again and `id()` values
    are recycled — an entry whose group object is gone, or is another object, is resolved anew."""
    g = dist.group.WORLD if process_group is None else process_group
    key = (getattr(g, "group_name", None), id(g))
    ent = _BACKEND_OF.get(key)
    if ent is not None and ent[0]() is g:
        return ent[1]
    b = str(dist.get_backend(process_group))
    try:
        _BACKEND_OF[key] = (weakref.ref(g), b)
    except TypeError:                      # (a group type that cannot be weakly referenced: resolved per call)
        pass
    if len(_BACKEND_OF) > 64:              # dead entries of destroyed groups
        for k_ in [k_ for k_, e_ in _BACKEND_OF.items() if e_[0]() is None]:
            del _BACKEND_OF[k_]
    return b


def _needs_host_staging(process_group, t: torch.Tensor) -> bool:
    """gloo cannot move device memory.  The product transport is RCCL; several TEST ranks sharing one MI355X stage
    through the host after ring_flash_attn._testing.allow_host_staging() — anything else fails loudly here."""
    if not (t.is_cuda and backend_of(process_group) == "gloo"):
        return False
    if _TEST is None or not _TEST.host_staging:
        raise RuntimeError("ring_flash_attn: device tensors on a gloo process group — the exchange runs on RCCL "
                           "(init_process_group('nccl')); gloo cannot move device memory")
    return True


# ---------------------------------------------------------------------------------------------
# Opt-in exchange audit (config.exchange_check / RFA_EXCHANGE_CHECK=1; round 6).  Every multi-rank GPU
# test of this repo moves data through gloo and the host; the RCCL orderings (side-stream commit / wait, the recycled
# receive buffers, posting W - 1 exchanges at once) run with more than one rank for the first time on the first 8-GPU job.
# With the audit on, every buffer handed to a transfer is checksummed by its SENDER (on the compute stream, before the
# transfer is posted) and by its RECEIVER (right after the wait that publishes it, before the kernel that consumes it);
# at the end of the schedule call the senders' checksums are all-gathered — ONE extra tiny collective per call, plus one
# host read-back: a debug mode — and every arrival is compared with what its sender sent.  A mismatch raises with the
# rank, the step and the buffer named; the first 8-GPU run then diagnoses itself instead of producing wrong gradients.
# (Sums are not audited: a reduce-scatter has no single sender.)
_AUDITS = {}
_AUDIT_MAX = 1024            # entries per call (fixed size: the table's all-gather must not depend on a rank's own path)


def _audit_on() -> bool:
    return config.get().exchange_check and _loopback() is None


def _cksum(t: torch.Tensor) -> torch.Tensor:
    """two 64-bit sums over the buffer's BIT PATTERNS: plain and position-weighted (a shifted or permuted buffer changes
    the second) — exact integer arithmetic, identical on sender and receiver whatever the device"""
    x = t.detach()
    if not x.is_contiguous():
        x = x.contiguous()
    w = {2: torch.int16, 4: torch.int32, 8: torch.int64}.get(x.element_size(), torch.uint8)
    v = x.view(w).flatten().to(torch.int64)
    pos = torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 8191 + 1
    return torch.stack([v.sum(), (v * pos).sum()])


class _Audit:
    def __init__(self, group):
        self.group, self.sent, self.recv = group, [], []

    def send(self, t) -> int:
        self.sent.append(_cksum(t))
        return len(self.sent) - 1

    def reserve(self, n) -> int:
        """n consecutive entries filled later (all-to-all: one per destination)"""
        base = len(self.sent)
        self.sent.extend([None] * n)
        return base

    def expect(self, idx, src, t, label):
        self.recv.append((idx, src, _cksum(t), label))


def _audit(group) -> "_Audit":
    g = dist.group.WORLD if group is None else group
    a = _AUDITS.get(id(g))
    if a is None:
        a = _AUDITS[id(g)] = _Audit(group)
    return a


def audit_verify(group, where: str):
    """compare everything this rank received during the call `where` with what the senders sent (collective when the
    audit is on: every rank of `group` calls it at the same point — the autograd Functions and the compiled-caller
    operators do, after each schedule forward / backward).  No-op when config.exchange_check is off."""
    g = dist.group.WORLD if group is None else group
    a = _AUDITS.pop(id(g), None)
    if a is None or not config.get().exchange_check:
        return
    if len(a.sent) > _AUDIT_MAX:
        raise RuntimeError(f"ring_flash_attn exchange check: {len(a.sent)} transfers in one call (> {_AUDIT_MAX})")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = next((c.device for c in a.sent if c is not None), torch.device("cpu"))
    on_host = backend_of(group) == "gloo"
    table = torch.zeros((_AUDIT_MAX + 1, 2), dtype=torch.int64, device=dev)
    table[0, 0] = len(a.sent)
    for i, c in enumerate(a.sent):
        if c is not None:
            table[i + 1] = c
    mine = table.cpu() if on_host else table
    all_ = torch.empty((world,) + tuple(mine.shape), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(all_.view(-1, 2), mine, group=group)
    all_ = all_.cpu()
    counts = [int(all_[r, 0, 0]) for r in range(world)]
    if len(set(counts)) != 1:
        raise RuntimeError(f"ring_flash_attn exchange check ({where}): the ranks posted different numbers of transfers "
                           f"{counts} — the collective sequence diverged")
    bad = []
    for idx, src, c, label in a.recv:
        got, want = c.cpu(), all_[src, idx + 1]
        if not torch.equal(got, want):
            bad.append(f"{label}: received checksum {got.tolist()} != sender rank {src}'s {want.tolist()}")
    if bad:
        raise RuntimeError(f"ring_flash_attn exchange check FAILED on rank {rank} in {where}: " + "; ".join(bad[:8])
                           + (f" (+ {len(bad) - 8} more)" if len(bad) > 8 else ""))


# ---------------------------------------------------------------------------------------------
# Side HIP stream for the exchange.  torch's RCCL process group orders a collective after the
# *current* stream at the moment it is posted and makes the *current* stream wait in Work.wait();
# by posting and waiting under an explicit per-device side stream, the dependencies between the
# attention kernels (compute stream) and the transfers are explicit:
#     commit():  side waits for `ready` (the compute stream up to here: the send data exists and the
#                receive buffer's last reader has been enqueued) -> the transfer is posted under `side`
#     wait():    side waits for the transfer; the compute stream then waits for the side stream
# so the compute stream never blocks on anything but the one transfer it is about to consume, and
# transfers posted later (the next step's) are not serialised behind this wait.
_SIDE_STREAMS = {}


def comm_stream(device: torch.device) -> "torch.cuda.Stream":
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(idx)
    if st is None:
        st = torch.cuda.Stream(device=torch.device("cuda", idx))
        _SIDE_STREAMS[idx] = st
    return st


def _use_side_stream(process_group, t: torch.Tensor) -> bool:
    return t.is_cuda and backend_of(process_group) != "gloo"


def single_rank(world_size: int) -> bool:
    """True when a schedule may collapse to its single-kernel form (a one-rank group; _testing.force_steps keeps the
    multi-step path even there: the RCCL calls on a one-GPU box)"""
    return world_size == 1 and not (_TEST is not None and _TEST.force_steps)


def group_rank_world(process_group):
    """(rank, world_size) of the group — the one place the schedules ask for it"""
    lb = _loopback()
    if lb is not None:
        return lb
    return dist.get_rank(process_group), dist.get_world_size(process_group)


class RingComm:
    """Neighbour exchange on a ring of the process group: send to rank+1, receive from rank-1.

    Usage pattern (identical to the reference, utils.py:98-151):
        nk, nv = comm.send_recv_kv(k, v)      # posts the transfer, returns receive buffers
        ... launch attention on the current k, v ...
        comm.wait(); k, v = nk, nv
    """

    def __init__(self, process_group: dist.ProcessGroup):
        self._process_group = process_group
        self._ops = []
        self.rank, self.world_size = group_rank_world(process_group)
        self._reqs = None
        self._staged = []          # (host_recv, device_recv) pairs for the gloo staging path
        self._local = []           # (src, dst) pairs of the loopback measurement hook
        self._pool = {}            # recycled receive buffers, keyed by (slot, shape, dtype, device)
        self._side = None          # side stream the pending transfer was posted under (RCCL path)
        self._audit = []           # (entry index, receive buffer) of the pending transfer (config.exchange_check)
        self._nhops = 0

        self.send_rank = (self.rank + 1) % self.world_size
        self.recv_rank = (self.rank - 1) % self.world_size

        if process_group is not None and _loopback() is None:
            self.send_rank = dist.get_global_rank(self._process_group, self.send_rank)
            self.recv_rank = dist.get_global_rank(self._process_group, self.recv_rank)

    def _recv_buffer(self, like: torch.Tensor) -> torch.Tensor:
        """Recycled receive buffer.  Per (position in the batch, shape, dtype) there are two
        buffers used alternately: the one handed out at step s was the *send* source of step
        s-1, whose transfer has been waited for and whose readers were all enqueued on the compute
        stream before this step's commit (which orders the transfer after that stream), so it is
        free again."""
        slot = len(self._ops) // 2 + len(self._local)
        key = (slot, tuple(like.shape), like.dtype, like.device)
        pair = self._pool.get(key)
        if pair is None:
            pair = [torch.empty_like(like, memory_format=torch.contiguous_format) for _ in range(2)]
            pair.append(0)
            self._pool[key] = pair
        idx = pair[2]
        if pair[idx].data_ptr() == like.data_ptr():
            idx ^= 1
        pair[2] = idx ^ 1
        return pair[idx]

    def send_recv(self, to_send: torch.Tensor, recv_tensor: Optional[torch.Tensor] = None) -> torch.Tensor:
        if recv_tensor is None:
            res = self._recv_buffer(to_send)
        else:
            res = recv_tensor
        if _loopback() is not None:
            self._local.append((to_send, res))
            return res
        if _audit_on():
            self._audit.append((_audit(self._process_group).send(to_send), res))
        if _needs_host_staging(self._process_group, to_send):
            host_send = to_send.detach().to("cpu")
            host_recv = torch.empty(res.shape, dtype=res.dtype, device="cpu")
            self._staged.append((host_recv, res))
            send_op = dist.P2POp(dist.isend, host_send, self.send_rank, group=self._process_group)
            recv_op = dist.P2POp(dist.irecv, host_recv, self.recv_rank, group=self._process_group)
        else:
            if not to_send.is_contiguous():
                to_send = to_send.contiguous()
            send_op = dist.P2POp(dist.isend, to_send, self.send_rank, group=self._process_group)
            recv_op = dist.P2POp(dist.irecv, res, self.recv_rank, group=self._process_group)
        self._ops.append(send_op)
        self._ops.append(recv_op)
        return res

    def commit(self):
        if self._reqs is not None:
            raise RuntimeError("commit called twice")
        if _loopback() is not None:
            for src, dst in self._local:
                dst.copy_(src)
            self._reqs = []
            return
        t = self._ops[0].tensor if self._ops else None
        if t is not None and _use_side_stream(self._process_group, t):
            side = comm_stream(t.device)
            side.wait_stream(torch.cuda.current_stream(t.device))     # `ready`
            with torch.cuda.stream(side):
                self._reqs = dist.batch_isend_irecv(self._ops)
            self._side = side
        else:
            self._reqs = dist.batch_isend_irecv(self._ops)

    def wait(self):
        if self._reqs is None:
            raise RuntimeError("wait called before commit")
        if self._side is not None:
            side = self._side
            with torch.cuda.stream(side):
                for req in self._reqs:
                    req.wait()                                         # the side stream waits for the transfer
            torch.cuda.current_stream(side.device).wait_stream(side)   # `done`
            self._side = None
        else:
            for req in self._reqs:
                req.wait()
        for host_recv, dev in self._staged:
            dev.copy_(host_recv)
        if _TEST is not None and _TEST.corrupt_recv is not None and self._audit:
            _TEST.corrupt(self._audit[0][1])          # (tests: a receive buffer that something overwrote after it landed)
        if self._audit:
            # the ring neighbour posts its transfers in the same order: entry i of ITS table is what arrived here
            src = (self.rank - 1) % self.world_size
            for j, (idx, buf) in enumerate(self._audit):
                _audit(self._process_group).expect(idx, src, buf, f"ring hop {self._nhops}, buffer {j} (from rank {src})")
            self._audit = []
        self._nhops += 1
        self._staged = []
        self._local = []
        self._reqs = None
        self._ops = []

    def send_recv_kv(
        self,
        k: torch.Tensor,
        v: torch.Tensor,
        k_buffer: Optional[torch.Tensor] = None,
        v_buffer: Optional[torch.Tensor] = None,
    ) -> Tuple[torch.Tensor, torch.Tensor]:
        next_k, next_v = self.send_recv(k, k_buffer), self.send_recv(v, v_buffer)
        self.commit()
        return next_k, next_v


class _Work:
    """a posted collective: wait() makes the compute (current) stream wait for it"""

    def __init__(self, handle, side=None, after=None):
        self.handle, self.side, self.after = handle, side, after

    def wait(self):
        if self.side is not None:
            with torch.cuda.stream(self.side):
                self.handle.wait()
            torch.cuda.current_stream(self.side.device).wait_stream(self.side)
        elif self.handle is not None:
            self.handle.wait()
        if self.after is not None:
            self.after()
            self.after = None


def _post(process_group, t: torch.Tensor, fn) -> _Work:
    """fn() posts an async collective and returns its Work; under the side stream when the group is RCCL and
    the data lives on the device"""
    if _use_side_stream(process_group, t):
        side = comm_stream(t.device)
        side.wait_stream(torch.cuda.current_stream(t.device))
        with torch.cuda.stream(side):
            return _Work(fn(), side)
    return _Work(fn())


class AllGatherComm:
    """Async all-gather handles (reference utils.py:154-168)."""

    def __init__(self, group=None) -> None:
        self.group = group
        self.handles = []
        self._audit = []           # (entry index, gathered output)  (config.exchange_check)

    def all_gather(self, output_tensor: torch.Tensor, input_tensor: torch.Tensor):
        if _loopback() is not None:
            world = _loopback()[1]
            output_tensor.view(world, -1).copy_(input_tensor.reshape(1, -1).expand(world, -1))
            return
        if _audit_on():
            self._audit.append((_audit(self.group).send(input_tensor), output_tensor))
        if _needs_host_staging(self.group, input_tensor):
            host_in = input_tensor.detach().to("cpu").contiguous()
            host_out = torch.empty(output_tensor.shape, dtype=output_tensor.dtype, device="cpu")
            handle = dist.all_gather_into_tensor(host_out, host_in, group=self.group, async_op=True)
            self.handles.append(_Work(handle, after=lambda: output_tensor.copy_(host_out)))
        else:
            self.handles.append(_post(self.group, input_tensor, lambda: dist.all_gather_into_tensor(
                output_tensor, input_tensor, group=self.group, async_op=True)))

    def wait(self):
        for handle in self.handles:
            handle.wait()
        self.handles = []
        if self._audit:
            rank, world = group_rank_world(self.group)
            for j, (idx, out) in enumerate(self._audit):
                parts = out.reshape(world, -1)
                for src in range(world):
                    if src != rank:
                        _audit(self.group).expect(idx, src, parts[src], f"all-gather {j}, slot of rank {src}")
            self._audit = []


class SourceArrivals:
    """The K/V of every other rank, RECEIVED IN CONSUMPTION ORDER (round 6: the `gather_ps` exchange form of the dense
    zigzag schedule).  One all-gather hands step 1 its K/V only when the LAST source has landed (7 x 32 MiB at W = 8 behind
    a 0.52 ms local block); the reference's hop protocol lets step s start when hop s is in
    (/root/reference/ring_flash_attn/zigzag_ring_flash_attn.py:60-84, utils.py:121-138).  Here all W - 1 exchanges are
    posted AT ONCE, in the order the steps consume them — exchange s: send the local K/V to rank r + s, receive those of
    rank r - s, a permutation in which every rank drives one direct xGMI link — each as its own batched isend/irecv group
    under the side stream, and `wait(s)` makes the compute stream wait for exchange s ONLY (the group's completion event):
    step s starts when source (r - s) mod W has landed, while the later sources are still on the wire.  RCCL runs the
    groups of one communicator in posting order, which IS the consumption order.

    `outs[i][src]` receives tensor i of rank src (the own slot is not written: the schedules use the local tensors there)."""

    def __init__(self, process_group):
        self._group = process_group
        self.rank, self.world_size = group_rank_world(process_group)
        self._reqs = {}            # step -> list of Work
        self._staged = {}          # step -> [(host_recv, device_dst)]  (gloo ranks sharing one GPU: tests)
        self._side = None
        self._audit = {}           # step -> [(entry index, receive buffer)]  (config.exchange_check)

    def _peer(self, r):
        r %= self.world_size
        return dist.get_global_rank(self._group, r) if self._group is not None else r

    def post(self, locals_, outs):
        W, rank = self.world_size, self.rank
        if _loopback() is not None:
            for t, out in zip(locals_, outs):
                for src in range(W):
                    if src != rank:
                        out[src].copy_(t)
            return self
        t0 = locals_[0]
        # (audit: ONE entry per tensor — every destination receives the same bytes)
        entries = [_audit(self._group).send(t) for t in locals_] if _audit_on() else None
        staging = _needs_host_staging(self._group, t0)
        sends = [t.detach().to("cpu") if staging else (t if t.is_contiguous() else t.contiguous()) for t in locals_]
        side = comm_stream(t0.device) if _use_side_stream(self._group, t0) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(t0.device))       # the send data exists, the receive buffers are free
        for s in range(1, W):
            dst, src = self._peer(rank + s), (rank - s) % W
            ops, staged = [], []
            for snd, out in zip(sends, outs):
                rcv = out[src]
                if staging:
                    host = torch.empty(rcv.shape, dtype=rcv.dtype, device="cpu")
                    staged.append((host, rcv))
                    rcv = host
                ops.append(dist.P2POp(dist.isend, snd, dst, group=self._group))
                ops.append(dist.P2POp(dist.irecv, rcv, self._peer(src), group=self._group))
            if side is not None:
                with torch.cuda.stream(side):
                    self._reqs[s] = dist.batch_isend_irecv(ops)
            else:
                self._reqs[s] = dist.batch_isend_irecv(ops)
            self._staged[s] = staged
            if entries is not None:
                self._audit[s] = [(e, out[src]) for e, out in zip(entries, outs)]
        self._side = side
        return self

    def wait(self, step: int):
        """the compute (current) stream waits for exchange `step` — and for nothing posted after it"""
        for req in self._reqs.pop(step, ()):
            req.wait()
        for host, dev in self._staged.pop(step, ()):
            dev.copy_(host)
        pend = self._audit.pop(step, ())
        if pend and _TEST is not None and _TEST.corrupt_recv is not None:
            _TEST.corrupt(pend[0][1])
        src = (self.rank - step) % self.world_size
        for j, (idx, buf) in enumerate(pend):
            _audit(self._group).expect(idx, src, buf, f"per-source exchange {step}, tensor {j} (from rank {src})")

    def wait_all(self):
        for s in sorted(self._reqs):
            self.wait(s)


def _sum_on_host(output, input_, group) -> _Work:
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    host = input_.detach().to("cpu", torch.float32)
    handle = dist.all_reduce(host, group=group, async_op=True)
    return _Work(handle, after=lambda: output.copy_(host.chunk(world, dim=0)[rank].reshape(output.shape).to(output.dtype)))


def reduce_scatter_async(output: torch.Tensor, input_: torch.Tensor, group=None) -> _Work:
    """output = this rank's dim-0 chunk of the sum over ranks of input_; the returned handle's wait() makes the
    compute stream wait for it.  gloo (tests: device memory shared by several ranks, or 16-bit CPU tensors,
    which gloo would add in their own precision): fp32 all_reduce on the host."""
    if _loopback() is not None:
        rank, world = _loopback()
        output.copy_(input_.chunk(world, dim=0)[rank].reshape(output.shape))
        return _Work(None)
    if backend_of(group) == "gloo" and (input_.dtype not in (torch.float32, torch.float64)
                                        or _needs_host_staging(group, input_)):
        return _sum_on_host(output, input_, group)
    return _post(group, input_, lambda: dist.reduce_scatter_tensor(output, input_, group=group, async_op=True))


class Agreement:
    """Group-wide AND of a rank-local boolean — the way a schedule turns a RANK-LOCAL fact (this rank's budget of kept
    bytes had room; this rank holds a tuning record) into a decision every rank takes alike.  The sequence of collectives
    a schedule posts must be a function of group-consistent state only: a rank that decides on its own whether to post
    an all-gather leaves its peers inside a collective it never joins.

    Posting does not stall the host: on an RCCL group the flag is reduced (MIN) under the side stream and copied to pinned
    host memory behind an event; `resolve()` — called where the decision is needed, for a forward's flag that is its
    backward, long after the transfer — waits for that event only.  gloo groups (CPU tensors; the shared-GPU test
    path) reduce on the host at once."""

    def __init__(self, process_group, flag: bool, device: torch.device):
        self._value, self._event, self._host, self._dev = None, None, None, None
        if _loopback() is not None:
            self._value = bool(flag)
            return
        if device.type == "cuda" and backend_of(process_group) != "gloo":
            side = comm_stream(device)
            with torch.cuda.stream(side):
                self._dev = torch.full((1,), int(bool(flag)), dtype=torch.int32, device=device)
                dist.all_reduce(self._dev, op=dist.ReduceOp.MIN, group=process_group, async_op=True).wait()
                self._host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                self._host.copy_(self._dev, non_blocking=True)
                self._event = torch.cuda.Event()
                self._event.record(side)
        else:
            t = torch.tensor([int(bool(flag))], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=process_group)
            self._value = bool(t.item())

    def resolve(self) -> bool:
        if self._value is None:
            # by the backward the flag has long retired (tools/agreement_stall.py: never pending over a 28-layer stack, 9 us of
            # host time per call — most of it the synchronize): ask first, block only if it really is still in flight
            if not self._event.query():
                self._event.synchronize()
            self._value = bool(self._host.item())
            self._event = self._host = self._dev = None
        return self._value


def reduce_scatter(output: torch.Tensor, input_: torch.Tensor, group=None):
    reduce_scatter_async(output, input_, group).wait()


def all_to_all_async(output: torch.Tensor, input_: torch.Tensor, group=None) -> _Work:
    """dim-0 chunk j of input_ goes to rank j; dim-0 chunk i of output comes from rank i"""
    if _loopback() is not None:
        output.copy_(input_)
        return _Work(None)
    check = None
    if _audit_on():
        rank, world = group_rank_world(group)
        a = _audit(group)
        base = a.reserve(world)                      # entry base + j: the chunk this rank sends to rank j
        for j, part in enumerate(input_.reshape(world, -1)):
            a.sent[base + j] = _cksum(part)

        def check():
            for src, part in enumerate(output.reshape(world, -1)):
                if src != rank:
                    a.expect(base + rank, src, part, f"all-to-all chunk from rank {src}")
    if _needs_host_staging(group, input_):
        host_in = input_.detach().to("cpu").contiguous()
        host_out = torch.empty(output.shape, dtype=output.dtype, device="cpu")
        handle = dist.all_to_all_single(host_out, host_in, group=group, async_op=True)

        def land():
            output.copy_(host_out)
            if check is not None:
                check()
        return _Work(handle, after=land)
    w = _post(group, input_, lambda: dist.all_to_all_single(output, input_, group=group, async_op=True))
    w.after = check
    return w
