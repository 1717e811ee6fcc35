"""Minimal `flash_attn` package for MI355X backed by librfa_hip.so — exactly the names the reference's
schedules, tests and benchmarks import (see flash_attn_interface.py).  Version string follows the
flash_attn >= 2.7 calling convention this module implements."""
from .flash_attn_interface import (  # noqa: F401
    flash_attn_func,
    flash_attn_kvpacked_func,
    flash_attn_qkvpacked_func,
    flash_attn_varlen_func,
    flash_attn_varlen_kvpacked_func,
    flash_attn_varlen_qkvpacked_func,
)

__version__ = "2.7.4+rfa.gfx950"
