"""Communication + merge helpers of the ring schedules.

Mirrors the public surface of /root/reference/ring_flash_attn/utils.py (`RingComm`,
`AllGatherComm`, `update_out_and_lse`, `flatten_varlen_lse`, `unflatten_varlen_lse`) with the
same method names and error behaviour, re-designed for RCCL over xGMI:

* `RingComm` posts ONE batched isend/irecv group per ring step (RCCL fuses it into one
  kernel on its own internal stream, so the transfer runs beside the attention kernel on the
  compute stream); `wait()` only makes the compute stream wait on that event — no host sync.
* receive buffers are recycled across steps (two alternating sets) instead of a fresh
  `torch.empty_like` per step (reference utils.py:117).
* `update_out_and_lse` is one HIP kernel (csrc/rfa_aux.hip: merge_kernel) instead of ~6
  eager TorchScript ops; the schedules in this package do not even call it — they use the
  merge fused into the attention epilogue — it is kept for API parity.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist

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


def _needs_host_staging(process_group, t: torch.Tensor) -> bool:
    # gloo cannot move device memory.  Only reached by the single-GPU multi-process parity test
    # (several ranks sharing one MI355X); production groups are nccl(=RCCL).
    return t.is_cuda and dist.get_backend(process_group) == "gloo"


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
        self.rank = dist.get_rank(self._process_group)
        self.world_size = dist.get_world_size(self._process_group)
        self._reqs = None
        self._staged = []          # (host_recv, device_recv) pairs for the gloo staging path
        self._pool = {}            # recycled receive buffers, keyed by (shape, dtype, device)

        self.send_rank = (self.rank + 1) % self.world_size
        self.recv_rank = (self.rank - 1) % self.world_size

        if process_group is not None:
            self.send_rank = dist.get_global_rank(self._process_group, self.send_rank)
            self.recv_rank = dist.get_global_rank(self._process_group, self.recv_rank)

    def _recv_buffer(self, like: torch.Tensor) -> torch.Tensor:
        """Recycled receive buffer.  Per (position in the batch, shape, dtype) there are two
        buffers used alternately: the one handed out at step s was the *send* source of step
        s-1, whose transfer and whose readers were all enqueued before this step's commit
        (RCCL orders its stream after the compute stream at commit), so it is free again."""
        slot = len(self._ops) // 2
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
        self._reqs = dist.batch_isend_irecv(self._ops)

    def wait(self):
        if self._reqs is None:
            raise RuntimeError("wait called before commit")
        for req in self._reqs:
            req.wait()
        for host_recv, dev in self._staged:
            dev.copy_(host_recv)
        self._staged = []
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


class AllGatherComm:
    """Async all-gather handles (reference utils.py:154-168)."""

    def __init__(self, group=None) -> None:
        self.group = group
        self.handles = []
        self._staged = []

    def all_gather(self, output_tensor: torch.Tensor, input_tensor: torch.Tensor):
        if _needs_host_staging(self.group, input_tensor):
            host_in = input_tensor.detach().to("cpu").contiguous()
            host_out = torch.empty(output_tensor.shape, dtype=output_tensor.dtype, device="cpu")
            handle = dist.all_gather_into_tensor(host_out, host_in, group=self.group, async_op=True)
            self._staged.append((host_out, output_tensor))
        else:
            handle = dist.all_gather_into_tensor(output_tensor, input_tensor, group=self.group, async_op=True)
        self.handles.append(handle)

    def wait(self):
        for handle in self.handles:
            handle.wait()
        for host_out, dev in self._staged:
            dev.copy_(host_out)
        self._staged = []
        self.handles = []


def reduce_scatter(output: torch.Tensor, input_: torch.Tensor, group=None):
    """dist.reduce_scatter_tensor with the same gloo host-staging escape hatch (gloo has no
    reduce_scatter_tensor for device memory; emulate with all_reduce on the host)."""
    if dist.get_backend(group) == "gloo":
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        host = input_.detach().to("cpu", torch.float32)
        dist.all_reduce(host, group=group)
        chunk = host.chunk(world, dim=0)[rank]
        output.copy_(chunk.to(output.dtype))
    else:
        dist.reduce_scatter_tensor(output, input_, group=group)
