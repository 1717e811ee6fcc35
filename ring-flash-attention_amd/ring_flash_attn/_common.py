"""Small helpers shared by the schedule modules."""

def _prep_qkv(q, k, v, group):
    """Kernels take strided views (last stride 1, 16-byte aligned rows).  K/V only have to be
    contiguous when they travel (world_size > 1: they are RCCL send buffers), so the packed
    `kv[:, :, 0]` views of the benchmark are not copied on a single GPU."""
    from .utils import group_rank_world, single_rank
    travels = not single_rank(group_rank_world(group)[1])
    if q.stride(-1) != 1:
        q = q.contiguous()
    if travels or k.stride(-1) != 1:
        k = k.contiguous()
    if travels or v.stride(-1) != 1:
        v = v.contiguous()
    return q, k, v


def _as_cu(cu_seqlens, device):
    """cu_seqlens as an int32 tensor on the compute device (the kernels read it on device)."""
    import torch
    if not torch.is_tensor(cu_seqlens):
        cu_seqlens = torch.tensor(cu_seqlens, dtype=torch.int32)
    if cu_seqlens.dtype != torch.int32:
        cu_seqlens = cu_seqlens.to(torch.int32)
    if cu_seqlens.device != device:
        cu_seqlens = cu_seqlens.to(device)
    return cu_seqlens.contiguous()
