"""Small helpers shared by the schedule modules."""

def packed_pair(k, v):
    """the contiguous (..., 2, Hk, D) tensor whose slices [..., 0, :, :] / [..., 1, :, :] k and v are (the `kv`
    argument of the kvpacked entry points and its gradient), or None"""
    import torch
    if (k.dim() < 3 or k.shape != v.shape or k.stride() != v.stride() or k.dtype != v.dtype or k.device != v.device
            or k.untyped_storage().data_ptr() != v.untyped_storage().data_ptr()):
        return None
    hk, d = k.shape[-2], k.shape[-1]
    lead = tuple(k.shape[:-2])
    want = [d, 1]
    run = 2 * hk * d
    for n in reversed(lead):
        want.insert(0, run)
        run *= n
    if tuple(k.stride()) != tuple(want) or v.storage_offset() - k.storage_offset() != hk * d:
        return None
    strides = tuple(want[:-2]) + (hk * d, d, 1)
    return torch.as_strided(k, lead + (2, hk, d), strides, k.storage_offset())


def _prep_qkv(q, k, v, group, packed_travel=False):
    """Kernels take strided views (last stride 1, 16-byte aligned rows).  K/V only have to be
    contiguous when they travel (world_size > 1: they are RCCL send buffers), so the packed
    `kv[:, :, 0]` views of the benchmark are not copied on a single GPU — nor on several when the schedule moves
    the packed tensor as ONE buffer (packed_travel: the zigzag gather form)."""
    from .utils import group_rank_world, single_rank
    travels = not single_rank(group_rank_world(group)[1])
    if travels and packed_travel and packed_pair(k, v) is not None:
        travels = False
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
