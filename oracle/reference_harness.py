"""oracle/reference_harness.py — TEST INFRASTRUCTURE ONLY; usable only where /root/reference exists
(the build container), never on the GPU box.

Loads the UNMODIFIED reference modules from /root/reference/ring_flash_attn under the private
package name `ref_ring_flash_attn`:
  * a namespace stub replaces the package's __init__ (whose `adapters` import is broken against
    transformers 5.x: hf_adapter.py:9-19), so only the algorithm modules are imported;
  * `flash_attn.flash_attn_interface` — the CUDA-only dependency that is absent here — is
    provided by the CPU oracle (oracle/flash_attn_ref.py), or (provider="shim") by the repo's own
    `flash_attn` compatibility package.
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


def load_reference(provider="oracle"):
    """returns dict name -> reference module.

    provider = "oracle": `flash_attn` is the CPU restatement (oracle/flash_attn_ref.py) — the golden
    generator.  provider = "shim": `flash_attn` is THIS repo's compatibility package
    (ring-flash-attention_amd/shims/flash_attn, INTEGRATION.md route B) — used by the test that runs the
    unmodified reference schedules through the shipped operator interface (fresh process only: the
    reference modules bind `flash_attn` at import time)."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True
    if provider == "shim":
        pkg_root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ring-flash-attention_amd")
        for p in (pkg_root, os.path.join(pkg_root, "shims")):     # ring_flash_attn (backend) + the opt-in flash_attn shim
            if p not in sys.path:
                sys.path.insert(0, p)
        if getattr(sys.modules.get("flash_attn"), "_rfa_oracle_stub", False) or "ref_ring_flash_attn" in sys.modules:
            raise RuntimeError("load_reference(provider='shim') needs a process that has not loaded the oracle stub")
        import flash_attn  # noqa: F401  (the shipped compatibility package)
        import flash_attn.flash_attn_interface  # noqa: F401
    else:
        _install_oracle_stub()

    if "ref_ring_flash_attn" not in sys.modules:
        pkg = types.ModuleType("ref_ring_flash_attn")
        pkg.__path__ = [os.path.join(REF_ROOT, "ring_flash_attn")]
        sys.modules["ref_ring_flash_attn"] = pkg
    return {m: importlib.import_module("ref_ring_flash_attn." + m) for m in _MODULES}


def _install_oracle_stub():
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
