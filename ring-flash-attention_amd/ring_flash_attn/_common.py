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


def dropout_arg(dropout_p, dropout_seed, q_pos_offset=0, k_pos_offset=0, head_offset=0):
    """backend `dropout=` argument (p, seed, q_pos_offset, k_pos_offset, head_offset), or None when dropout is off.
    The seed comes from the autograd Function (one draw per forward, reused by its backward)."""
    if not dropout_p or not dropout_p > 0:
        return None
    if dropout_seed is None:
        raise ValueError("ring_flash_attn: dropout_p > 0 needs a dropout_seed (the public functions draw one)")
    return (float(dropout_p), int(dropout_seed), int(q_pos_offset), int(k_pos_offset), int(head_offset))


_DROPOUT_SOURCE = {"generator": None, "group": None, "sync": False}


def set_dropout_generator(generator=None, sync_group=None, sync=False):
    """Where the per-forward dropout seeds come from.  Default (generator=None): torch's default CPU generator, like
    any torch random op — reproducible under torch.manual_seed, but it advances the stream data loaders and
    initialisers share, and the documented "sharded mask == unsharded mask" property then needs every rank of a context-
    parallel group to have seeded alike.  With a dedicated `torch.Generator` the global stream is left alone; with
    `sync=True` every draw is additionally broadcast from rank 0 of `sync_group` (one 8-byte broadcast per forward), so
    the ranks agree whatever their seeds (frameworks that seed per rank).  INTEGRATION.md, "Dropout"."""
    _DROPOUT_SOURCE.update(generator=generator, group=sync_group, sync=bool(sync))


def draw_dropout_seed() -> int:
    """one 62-bit seed per forward: from torch's default CPU generator (reproducible under torch.manual_seed; ranks
    that seeded alike draw alike, which makes the mask of a sharded call equal to the unsharded one — include/rfa.h), or
    from the source installed with set_dropout_generator()"""
    import torch
    src = _DROPOUT_SOURCE
    seed = torch.randint(0, 2 ** 62, (1,), generator=src["generator"])
    if src["sync"]:
        import torch.distributed as dist

        if dist.get_backend(src["group"]) != "gloo":           # RCCL broadcasts device memory
            seed = seed.cuda()
        dist.broadcast(seed, dist.get_global_rank(src["group"], 0) if src["group"] is not None else 0, group=src["group"])
    return int(seed.item())
