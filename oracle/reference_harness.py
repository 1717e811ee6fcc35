"""oracle/reference_harness.py — TEST INFRASTRUCTURE ONLY; usable only where /root/reference exists
(the build container), never on the GPU box.

Loads the UNMODIFIED reference modules from /root/reference/ring_flash_attn under the private
package name `ref_ring_flash_attn`:
  * a namespace stub replaces the package's __init__ (whose `adapters` import is broken against
    transformers 5.x: hf_adapter.py:9-19), so only the algorithm modules are imported;
  * `flash_attn.flash_attn_interface` — the CUDA-only dependency that is absent here — is
    provided by the CPU oracle (oracle/flash_attn_ref.py).
This is "the reference's CPU path" (BASELINE.md §3): reference schedule + merge + communication
code, executed under gloo, with the restated attention arithmetic underneath.  It is used to
generate the golden fixtures in tests/golden/ (tests/golden/make_golden.py).
Bytecode writing is disabled so nothing is dropped into the read-only reference tree.
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("RFA_REFERENCE_ROOT", "/root/reference")

_MODULES = [
    "utils",
    "ring_flash_attn",
    "zigzag_ring_flash_attn",
    "ring_flash_attn_varlen",
    "zigzag_ring_flash_attn_varlen",
    "llama3_flash_attn_varlen",
    "stripe_flash_attn",
]


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "ring_flash_attn"))


def load_reference():
    """returns dict name -> reference module."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True
    from . import flash_attn_ref as R

    if "flash_attn" not in sys.modules or not getattr(sys.modules["flash_attn"], "_rfa_oracle_stub", False):
        fa = types.ModuleType("flash_attn")
        fa._rfa_oracle_stub = True
        fai = types.ModuleType("flash_attn.flash_attn_interface")
        for name in R.__all__:
            setattr(fai, name, getattr(R, name))
        fa.flash_attn_interface = fai
        sys.modules["flash_attn"] = fa
        sys.modules["flash_attn.flash_attn_interface"] = fai

    if "ref_ring_flash_attn" not in sys.modules:
        pkg = types.ModuleType("ref_ring_flash_attn")
        pkg.__path__ = [os.path.join(REF_ROOT, "ring_flash_attn")]
        sys.modules["ref_ring_flash_attn"] = pkg
    return {m: importlib.import_module("ref_ring_flash_attn." + m) for m in _MODULES}
