"""MI355X-native ring flash-attention.

Drop-in for the public API of zhuzilin/ring-flash-attention
(/root/reference/ring_flash_attn/__init__.py:1-35): same 21 names, same signatures, same
sharding contracts — the arithmetic runs in hand-written gfx950 HIP kernels reached through
the C ABI of librfa_hip.so (include/rfa.h); communication is RCCL via torch.distributed.
"""
from .llama3_flash_attn_varlen import (
    llama3_flash_attn_prepare_cu_seqlens,
    llama3_flash_attn_varlen_func,
    llama3_flash_attn_varlen_kvpacked_func,
    llama3_flash_attn_varlen_qkvpacked_func,
)
from .ring_flash_attn import (
    ring_flash_attn_func,
    ring_flash_attn_kvpacked_func,
    ring_flash_attn_qkvpacked_func,
)
from .ring_flash_attn_varlen import (
    ring_flash_attn_varlen_func,
    ring_flash_attn_varlen_kvpacked_func,
    ring_flash_attn_varlen_qkvpacked_func,
)
from .zigzag_ring_flash_attn import (
    zigzag_ring_flash_attn_func,
    zigzag_ring_flash_attn_kvpacked_func,
    zigzag_ring_flash_attn_qkvpacked_func,
)
from .zigzag_ring_flash_attn_varlen import (
    zigzag_ring_flash_attn_varlen_func,
    zigzag_ring_flash_attn_varlen_kvpacked_func,
    zigzag_ring_flash_attn_varlen_qkvpacked_func,
)
from .stripe_flash_attn import (
    stripe_flash_attn_func,
    stripe_flash_attn_kvpacked_func,
    stripe_flash_attn_qkvpacked_func,
)
# beyond the reference (its README.md:131 lists this entry as a TODO): llama3-style context parallelism over a
# zigzag-balanced split of the packed token stream
from .zigzag_llama3_flash_attn_varlen import (
    zigzag_llama3_flash_attn_prepare_cu_seqlens,
    zigzag_llama3_flash_attn_varlen_func,
    zigzag_llama3_flash_attn_varlen_kvpacked_func,
    zigzag_llama3_flash_attn_varlen_qkvpacked_func,
)
from .adapters import (
    substitute_hf_flash_attn,
    update_ring_flash_attn_params,
)

__version__ = "0.1.0"

# this library's own knobs (not part of the reference surface): the configuration object (config.py: every RFA_* switch,
# resolved once) and the source of the per-forward dropout seeds
from . import config
from ._common import set_dropout_generator
