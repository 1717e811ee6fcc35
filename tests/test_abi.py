"""The C-ABI library loads, exports every symbol include/rfa.h declares, the ctypes mirror has
the C layout, argument errors are reported without touching a device, and the product path
fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rfa.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rfa_[a-z_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound(built):
    from ring_flash_attn import _C

    lib = C.CDLL(built.LIB)
    declared = _declared_symbols()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in rfa.h but not exported"
        assert name in _C.SYMBOLS, f"{name} not bound in _C.py"
    assert set(_C.SYMBOLS) == set(declared)


def test_library_exports_only_the_c_abi(built):
    """the dynamic symbol table of librfa_hip.so is exactly the C ABI of include/rfa.h: the library is built with
    -fvisibility=hidden and a linker version script (csrc/rfa_exports.map), so no C++ internal (rfa::launch_*), kernel
    stub or hipcc artefact is bindable"""
    out = subprocess.run(["nm", "-D", "--defined-only", built.LIB], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == _declared_symbols(), set(exported) ^ set(_declared_symbols())


def test_ctypes_structs_match_c_layout(built):
    from ring_flash_attn import _C

    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "rfa.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(rfa_fwd_args), sizeof(rfa_bwd_preprocess_args),
         sizeof(rfa_bwd_args), sizeof(rfa_merge_args), sizeof(rfa_strides));
  printf("%zu %zu %zu %zu\n", offsetof(rfa_fwd_args, dtype), offsetof(rfa_bwd_args, phases),
         offsetof(rfa_bwd_args, total_k), offsetof(rfa_merge_args, block_lse_row));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    sizes = [int(x) for x in out]
    assert sizes[:5] == [C.sizeof(_C.FwdArgs), C.sizeof(_C.BwdPreArgs), C.sizeof(_C.BwdArgs),
                         C.sizeof(_C.MergeArgs), C.sizeof(_C.Strides)]
    assert sizes[5:] == [_C.FwdArgs.dtype.offset, _C.BwdArgs.phases.offset, _C.BwdArgs.total_k.offset,
                         _C.MergeArgs.block_lse_row.offset]


def test_argument_errors_without_device(built):
    from ring_flash_attn import _C

    lib = _C.load()
    assert lib.rfa_abi_version() == _C.RFA_ABI_VERSION
    assert lib.rfa_fwd(None, None) == -1                      # RFA_ERR_NULL
    assert b"dynamic LDS" in lib.rfa_strerror(-9)             # RFA_ERR_ATTR (per-device kernel attribute)
    assert lib.rfa_strerror(-10) == b"unknown rfa status"
    a = _C.FwdArgs()
    a.B, a.H, a.Hk, a.D, a.Sq, a.Sk, a.dtype = 1, 4, 3, 64, 8, 8, 0
    assert lib.rfa_fwd(C.byref(a), None) == -4                # H % Hk
    a.Hk, a.D = 2, 264
    assert lib.rfa_fwd(C.byref(a), None) == -3                # head dim: above 256 ...
    a.D = 132
    assert lib.rfa_fwd(C.byref(a), None) == -3                # ... or not a multiple of 8
    a.D, a.dtype = 64, 7
    assert lib.rfa_fwd(C.byref(a), None) == -2                # dtype
    a.dtype = 0
    assert lib.rfa_fwd(C.byref(a), None) == -1                # q/k/v NULL
    a.Sq = 0
    assert lib.rfa_fwd(C.byref(a), None) == 0                 # empty problem is a no-op
    assert b"head_dim" in lib.rfa_strerror(-3)
    b = _C.BwdArgs()
    b.B, b.H, b.Hk, b.D, b.Sq, b.Sk, b.dtype = 1, 4, 2, 64, 8, 8, 0
    assert lib.rfa_bwd(C.byref(b), None) == -1
    b.total_k = 8
    assert lib.rfa_bwd_workspace_bytes(C.byref(b)) == 0       # GQA groups are summed inside the kernel
    b.phases = 1                                              # two-phase call: (rows, Hk, D) x 2 partials
    assert lib.rfa_bwd_workspace_bytes(C.byref(b)) == 2 * 8 * 2 * 64 * 2
    b.phases = 0


def test_product_path_has_no_cpu_fallback(built, single_rank_group):
    """CPU tensors must raise, not silently compute somewhere else."""
    import ring_flash_attn
    from ring_flash_attn import backend
    from ring_flash_attn import _testing

    _testing.set_backend(None)
    q = torch.randn(1, 16, 2, 32, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU"):
        ring_flash_attn.ring_flash_attn_func(q, q, q, causal=True)
    with pytest.raises(RuntimeError, match="no CPU"):
        ring_flash_attn.zigzag_ring_flash_attn_func(q, q, q, causal=True)


def test_missing_extension_fails_loudly(monkeypatch, built):
    from ring_flash_attn import _C

    monkeypatch.setattr(_C, "_lib", None)
    monkeypatch.setattr(_C, "LIB_PATH", "/nonexistent/librfa_hip.so")
    with pytest.raises(RuntimeError, match="not built"):
        _C.load()


def test_public_api_surface():
    """the 21 public names of reference ring_flash_attn/__init__.py:1-35, same signatures."""
    import inspect
    import ring_flash_attn as r

    tail = ["dropout_p", "softmax_scale", "causal", "window_size", "alibi_slopes", "deterministic",
            "return_attn_probs", "group"]
    for prefix, lead in (("ring_flash_attn", []), ("zigzag_ring_flash_attn", []), ("stripe_flash_attn", []),
                         ("ring_flash_attn_varlen", ["cu_seqlens", "max_seqlen"]),
                         ("zigzag_ring_flash_attn_varlen", ["cu_seqlens", "max_seqlen"])):
        for suffix, first in (("func", ["q", "k", "v"]), ("kvpacked_func", ["q", "kv"]), ("qkvpacked_func", ["qkv"])):
            fn = getattr(r, f"{prefix}_{suffix}")
            assert list(inspect.signature(fn).parameters) == first + lead + tail
    l3 = ["cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q", "max_seqlen_k", "heads_k_stride", "local_k_slice"]
    # (reference llama3_flash_attn_varlen.py:390-504: the packed wrappers take the same positional tail)
    for suffix, first in (("func", ["q", "k", "v"]), ("kvpacked_func", ["q", "kv"]), ("qkvpacked_func", ["qkv"])):
        sig = inspect.signature(getattr(r, f"llama3_flash_attn_varlen_{suffix}"))
        assert list(sig.parameters) == first + l3 + tail
        assert [sig.parameters[n].default for n in tail] == [0.0, None, False, (-1, -1), None, False, False, None]
    assert list(inspect.signature(r.llama3_flash_attn_prepare_cu_seqlens).parameters) == ["cu_seqlens", "causal", "rank", "world_size"]
    assert callable(r.substitute_hf_flash_attn) and callable(r.update_ring_flash_attn_params)
    # beyond the reference (its README TODO): zigzag_llama3 takes the GLOBAL cu_seqlens
    sig = list(inspect.signature(r.zigzag_llama3_flash_attn_varlen_func).parameters)
    assert sig == ["q", "k", "v", "cu_seqlens", "heads_k_stride"] + tail
    assert list(inspect.signature(r.zigzag_llama3_flash_attn_prepare_cu_seqlens).parameters) == ["cu_seqlens", "causal", "rank", "world_size"]


def test_prepare_cu_seqlens_golden(golden):
    """bit-exact integer parity with the reference's llama3_flash_attn_prepare_cu_seqlens
    (llama3_flash_attn_varlen.py:10-60) on its own test fixture [0,7,14,16] x 8 ranks and more."""
    from ring_flash_attn import llama3_flash_attn_prepare_cu_seqlens

    assert len(golden["prepare_cu_seqlens"]) > 50
    for g in golden["prepare_cu_seqlens"]:
        cu = torch.tensor(g["cu"], dtype=torch.int32)
        cq, ck, mq, mk, sl = llama3_flash_attn_prepare_cu_seqlens(cu, g["causal"], g["rank"], g["W"])
        assert cq.dtype == torch.int32 and cq.tolist() == g["cu_q"] and ck.tolist() == g["cu_k"]
        assert (mq, mk) == (g["max_q"], g["max_k"]) and (sl.start, sl.stop) == tuple(g["k_slice"])


def test_flash_attn_shim_surface(built):
    """The shipped `flash_attn` package exposes what the reference imports, in the form its
    `get_default_args` (utils.py:13-29) inspects: plain functions, flash_attn >= 2.7 parameter names
    (`window_size_left/right`, not `window_size`), `softcap` defaulting to 0.0."""
    import inspect
    import sys

    shims = os.path.join(ROOT, "ring-flash-attention_amd", "shims")     # opt-in root (INTEGRATION.md route B)
    assert "flash_attn" not in sys.modules or getattr(sys.modules["flash_attn"], "__file__", "").startswith(shims), \
        "importing ring_flash_attn must not put a `flash_attn` package on the path (it would shadow a real install)"
    if shims not in sys.path:
        sys.path.insert(0, shims)
    import flash_attn
    from flash_attn import flash_attn_interface as F

    for name in ("flash_attn_func", "flash_attn_kvpacked_func", "flash_attn_qkvpacked_func",
                 "flash_attn_varlen_func", "flash_attn_varlen_kvpacked_func", "flash_attn_varlen_qkvpacked_func"):
        assert callable(getattr(flash_attn, name))
    want = {
        "_flash_attn_forward": ["q", "k", "v", "dropout_p", "softmax_scale", "causal", "window_size_left",
                                "window_size_right", "softcap", "alibi_slopes", "return_softmax"],
        "_flash_attn_backward": ["dout", "q", "k", "v", "out", "softmax_lse", "dq", "dk", "dv", "dropout_p",
                                 "softmax_scale", "causal", "window_size_left", "window_size_right", "softcap",
                                 "alibi_slopes", "deterministic", "rng_state"],
    }
    for name, args in want.items():
        fn = getattr(F, name)
        assert inspect.isfunction(fn)
        spec = inspect.getfullargspec(fn)
        assert spec.args == args, (name, spec.args)
    for name in ("_flash_attn_varlen_forward", "_flash_attn_varlen_backward"):
        spec = inspect.getfullargspec(getattr(F, name))
        for a in ("cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q", "max_seqlen_k", "window_size_left", "softcap"):
            assert a in spec.args, (name, a)
    import torch

    q = torch.zeros(1, 8, 2, 64, dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):                       # dropout together with a window
        F._flash_attn_forward(q, q, q, 0.1, 0.125, True, window_size_left=4)
    with pytest.raises(NotImplementedError):
        F._flash_attn_forward(q, q, q, 0.0, 0.125, True, softcap=30.0)
    with pytest.raises(ValueError, match="rng_state"):            # a dropout backward without the forward's state
        F._flash_attn_backward(q, q, q, q, q, None, q, q, q, 0.1, 0.125, True)
    # (the forward's extra `rng_state` is keyword-only: flash_attn's positional signature is untouched)
    assert "rng_state" in inspect.getfullargspec(F._flash_attn_forward).kwonlyargs


def test_backward_plan_is_a_function_of_the_arguments(built, monkeypatch):
    """rfa_bwd_plan / rfa_bwd_workspace_bytes / rfa_bwd_ds_scratch_bytes (host code): which dK/dV kernel form a call
    runs and how much scratch it asks for — the numbers DESIGN.md quotes for the headline — and the rule that a
    BWD_COMPUTE / BWD_REDUCE pair and the sizing call agree because only the ARGUMENTS enter (ABI 4: the tuning
    overrides are fields of the call; the process environment is not consulted by the library)."""
    from ring_flash_attn import _C
    from ring_flash_attn import backend as BK
    from ring_flash_attn import _testing

    lib = _C.load()

    def args(B, Sq, Sk, H, Hk, D=128, varlen_total=None, acc=False, phases=0, window=None, halves=(0, 0),
             causal=False, form=0, nsplit=0):
        a = _C.BwdArgs()
        a.B, a.Sq, a.Sk, a.H, a.Hk, a.D, a.dtype = B, Sq, Sk, H, Hk, D, 0
        a.total_k = varlen_total if varlen_total is not None else B * Sk
        if varlen_total is not None:
            a.cu_seqlens_q = a.cu_seqlens_k = 1                  # non-null: host code never dereferences them
        if acc:
            a.dk_acc = a.dv_acc = 1
        a.phases = phases
        a.causal = 1 if causal else 0
        a.q_half, a.k_half = halves
        a.dkdv_form, a.dkdv_nsplit = form, nsplit
        if window:
            a.window, a.window_left, a.window_right = 1, window[0], window[1]
        return a

    unit = lambda rows, Hk, D=128: 2 * rows * Hk * D * 2        # one (dK, dV) partial set in the io dtype
    unit32 = lambda rows, Hk, D=128: 2 * rows * Hk * D * 4      # ... in fp32 (the partials of a split launch)
    ws = lambda a: lib.rfa_bwd_workspace_bytes(C.byref(a))
    ds = lambda a: lib.rfa_bwd_ds_scratch_bytes(C.byref(a))

    def plan(a):
        f, n, g = C.c_int32(), C.c_int32(), C.c_int32()
        assert lib.rfa_bwd_plan(C.byref(a), C.byref(f), C.byref(n), C.byref(g)) == 0
        return f.value, n.value

    # headline (GQA 32:8, S = 8192, dense causal self-attention): round 6's BALANCED schedule of the 256-key form — 256 equal
    # workgroups, the 16 lower key blocks of every (batch, K/V head) shared by two workgroups that add their fp32 partials
    # between themselves: one 256-key (dK, dV) pair slot + one flag word per shared block, no reduction pass
    pair_ws = lambda B, S, Hk, D=128: (B * Hk * (S // 512) * 4 + 255) // 256 * 256 + B * Hk * (S // 512) * 2 * 256 * D * 4
    assert plan(args(1, 8192, 8192, 32, 8, causal=True)) == (_C.DKDV_BAL, 1)
    assert ws(args(1, 8192, 8192, 32, 8, causal=True)) == pair_ws(1, 8192, 8) == 512 + unit32(8192, 8) // 2
    # ... named plans still run: the shared-range plan of rounds 2-5 (two fp32 partial sets + reduce_kernel) ...
    assert plan(args(1, 8192, 8192, 32, 8, causal=True, form=_C.DKDV_256, nsplit=2)) == (_C.DKDV_256, 2)
    assert ws(args(1, 8192, 8192, 32, 8, causal=True, form=_C.DKDV_256, nsplit=2)) == 2 * unit32(8192, 8)
    # ... and the balanced schedule by name where the estimate would not pick it (MHA: 1024 workgroups balance unshared)
    assert plan(args(1, 8192, 8192, 32, 32, causal=True, form=_C.DKDV_BAL)) == (_C.DKDV_BAL, 1)
    # not eligible -> the field reads as AUTO: no mask, sequences that are not whole pairs of 256-key blocks, Sq != Sk,
    # packed sequences, += into accumulators, two-phase calls, a window, another head dim
    assert plan(args(1, 8192, 8192, 32, 8, form=_C.DKDV_BAL)) == (_C.DKDV_256, 1)
    assert plan(args(1, 8192 - 256, 8192 - 256, 32, 8, causal=True, form=_C.DKDV_BAL))[0] == _C.DKDV_256
    assert plan(args(1, 4096, 8192, 32, 8, causal=True, form=_C.DKDV_BAL))[0] != _C.DKDV_BAL
    assert plan(args(3, 7392, 7392, 32, 8, varlen_total=8192, causal=True, form=_C.DKDV_BAL))[0] != _C.DKDV_BAL
    assert plan(args(1, 8192, 8192, 32, 8, causal=True, acc=True, form=_C.DKDV_BAL))[0] != _C.DKDV_BAL
    assert plan(args(1, 8192, 8192, 32, 8, causal=True, phases=_C.BWD_COMPUTE, form=_C.DKDV_BAL))[0] != _C.DKDV_BAL
    assert plan(args(1, 8192, 8192, 32, 8, causal=True, window=(512, 0), form=_C.DKDV_BAL))[0] != _C.DKDV_BAL
    assert plan(args(1, 8192, 8192, 32, 8, D=96, causal=True, form=_C.DKDV_BAL)) == (_C.DKDV_128, 1)
    # ... overwritten accumulators (the kernel stores fp32 itself) are
    a_ow = args(1, 8192, 8192, 32, 8, causal=True, acc=True)
    a_ow.acc_init = 1
    assert plan(a_ow) == (_C.DKDV_BAL, 1)
    # ... under-filled launches keep the shared-range plans (the balanced schedule has B * Hk * S / 256 workgroups)
    assert plan(args(1, 8192, 8192, 8, 2, causal=True))[0] == _C.DKDV_256
    assert plan(args(1, 4096, 4096, 32, 8, causal=True))[0] == _C.DKDV_256
    # dS scratch: rectangular rows when every block is visited, packed triangular rows for a dense causal call
    assert ds(args(1, 8192, 8192, 32, 8)) == 32 * 256 * 256 * 2048
    assert ds(args(1, 8192, 8192, 32, 8, causal=True)) == 32 * (256 * 257 // 2) * 2048
    # ... bottom-right aligned: 4096 queries x 8192 keys -> row qt holds 129 + qt blocks
    assert ds(args(1, 4096, 8192, 2, 2, causal=True)) == 2 * sum(129 + qt for qt in range(128)) * 2048
    # ... more queries than keys: the first (Sq - Sk) / 32 rows are empty
    assert ds(args(1, 8192, 4096, 2, 2, causal=True)) == 2 * sum(max(0, qt - 127) for qt in range(256)) * 2048
    # ... odd lengths
    assert ds(args(1, 1000, 1000, 1, 1, causal=True)) == sum(min(32, qt + 1) for qt in range(32)) * 2048
    # MHA: 1024 workgroups already, no split -> plain outputs need no workspace; accumulate / phased calls do
    assert plan(args(1, 8192, 8192, 32, 32, causal=True)) == (_C.DKDV_256, 1)
    assert ws(args(1, 8192, 8192, 32, 32)) == 0
    assert ws(args(1, 8192, 8192, 32, 32, acc=True)) == unit(8192, 32)
    assert ws(args(1, 8192, 8192, 32, 32, phases=_C.BWD_COMPUTE)) == ws(args(1, 8192, 8192, 32, 32, phases=_C.BWD_REDUCE))
    # ring "front" step of world size 8 (all queries x 4096 keys: 128 key-block workgroups): two shares each fill the
    # chip in ONE round (round 6: chosen 1.437 ms = the best forced plan, profiles/r06_plan_sweep_after.md; rounds 2-5 ran 4)
    assert plan(args(1, 8192, 4096, 32, 8)) == (_C.DKDV_256, 2)
    assert ws(args(1, 8192, 4096, 32, 8)) == 2 * unit32(4096, 8)
    # a launch with a handful of key blocks (4 x 2 K/V heads) is shared until the chip has work: round 6's sweep found such
    # launches up to 1.9 x behind the best plan under the old "shares of >= 2048 rows" rule
    assert plan(args(1, 1024, 1024, 4, 2)) == (_C.DKDV_256, 8)
    assert ws(args(1, 1024, 1024, 4, 2)) == 8 * unit32(1024, 2)
    # padded head dims and windows keep the 128-key form (no split, no workspace)
    assert ws(args(1, 8192, 8192, 32, 8, D=96)) == 0 and ds(args(1, 8192, 8192, 32, 8, D=96)) == 0
    assert plan(args(1, 8192, 8192, 32, 8, D=56, causal=True)) == (_C.DKDV_128, 1)
    # head dim 64 exactly (round 5): the 256-key form by the same shape rules, never a dS hand-off
    assert plan(args(1, 8192, 8192, 32, 8, D=64, causal=True)) == (_C.DKDV_BAL, 1)
    assert ws(args(1, 8192, 8192, 32, 8, D=64, causal=True)) == pair_ws(1, 8192, 8, 64) and ds(args(1, 8192, 8192, 32, 8, D=64)) == 0
    assert plan(args(1, 8192, 4096, 32, 8, D=64)) == (_C.DKDV_256, 2) and ws(args(1, 8192, 4096, 32, 8, D=64)) == 2 * unit32(4096, 8, 64)
    # (no mask: 256 equal workgroups are one balanced round of the chip — nothing to share, no workspace)
    assert plan(args(1, 8192, 8192, 32, 8)) == (_C.DKDV_256, 1) and ws(args(1, 8192, 8192, 32, 8, D=64)) == 0
    assert plan(args(1, 1024, 1024, 4, 2, D=64)) == (_C.DKDV_256, 8)
    assert ws(args(1, 8192, 8192, 32, 8, window=(512, -1))) == 0 and ds(args(1, 8192, 8192, 32, 8, window=(512, -1))) == 0
    # packed sequences: the form is chosen from the packed row count, and so is the scratch (ABI 5): total / 32 + B
    # query-block rows per head, each with the key blocks of the longest (half) sequence — 3.9 GB for the varlen
    # benchmark's (256, 7392, 544) pattern where B x the longest sequence would be 10.5 GB
    assert ws(args(3, 7392, 7392, 32, 8, varlen_total=8192)) == 3 * unit32(8192, 8)
    assert ds(args(3, 7392, 7392, 32, 8, varlen_total=8192, causal=True)) == 32 * (256 + 3) * 231 * 2048
    assert ds(args(3, 7392, 7392, 32, 8, varlen_total=8192, halves=(2, 1))) == 32 * (256 + 3) * 116 * 2048
    # the overrides are arguments ...
    assert ws(args(1, 8192, 8192, 32, 8, form=_C.DKDV_128)) == 0
    assert plan(args(1, 1024, 1024, 4, 2, form=_C.DKDV_256, nsplit=3)) == (_C.DKDV_256, 3)
    assert ws(args(1, 1024, 1024, 4, 2, form=_C.DKDV_256, nsplit=3)) == 3 * unit32(1024, 2)
    assert plan(args(1, 4096, 4096, 4, 2, form=_C.DKDV_256)) == (_C.DKDV_256, 8)    # (share count still by the estimate)
    assert plan(args(1, 1024, 1024, 4, 2, form=_C.DKDV_256)) == (_C.DKDV_256, 8)
    # short sequences with 8 K/V heads (profiles/history/r04_dkdv_plans_short_sequences.txt, re-measured in round 6): <= 1024 the
    # 256-key form unshared; 2048 and 4096 with 256 workgroups: two shares (the causal imbalance of ONE round)
    # round 6, second session (profiles/r06_balanced_schedule.md): ONE round of the chip (256 workgroups) runs the balanced
    # schedule from 2048 rows on; shorter sequences the 128-key form (twice the workgroups), launches of several rounds
    # (whose heaviest-first order — batch index fastest — balances them without any sharing) stay unshared
    assert plan(args(8, 1024, 1024, 32, 8, causal=True)) == (_C.DKDV_128, 1)       # (512 key-block workgroups, dealt heaviest first)
    assert plan(args(16, 512, 512, 32, 8, causal=True)) == (_C.DKDV_128, 1)
    assert plan(args(4, 2048, 2048, 32, 8, causal=True)) == (_C.DKDV_BAL, 1)
    assert plan(args(2, 4096, 4096, 32, 8, causal=True)) == (_C.DKDV_BAL, 1)
    assert plan(args(4, 4096, 4096, 32, 8, causal=True)) == (_C.DKDV_256, 1)
    assert plan(args(2, 8192, 8192, 32, 8, causal=True)) == (_C.DKDV_256, 1)
    # the plan is a pure function of the shapes: asking twice (memoised) gives the same answer
    assert plan(args(2, 4096, 4096, 32, 8, causal=True)) == (_C.DKDV_BAL, 1)
    # ... the process environment does not reach the library
    monkeypatch.setenv("RFA_DKDV_WIDE", "0")
    monkeypatch.setenv("RFA_DKDV_NSPLIT", "3")
    assert plan(args(1, 8192, 8192, 32, 8, causal=True)) == (_C.DKDV_BAL, 1)
    # ... it is translated once per backward by the Python backend (tests / tuning)
    assert BK._plan_overrides() == (_C.DKDV_128, 3)
    monkeypatch.setenv("RFA_DKDV_WIDE", "1")
    assert BK._plan_overrides() == (_C.DKDV_256, 3)
    monkeypatch.setenv("RFA_DKDV_WIDE", "2")                 # (a share count names the shared-range plan)
    assert BK._plan_overrides() == (_C.DKDV_256, 3)
    monkeypatch.delenv("RFA_DKDV_NSPLIT")
    assert BK._plan_overrides() == (_C.DKDV_BAL, 0)
    monkeypatch.delenv("RFA_DKDV_WIDE")
    assert BK._plan_overrides() == (_C.DKDV_AUTO, 0)
    # ---- ABI 5: the dS hand-off runs in head-group chunks over a scratch smaller than the whole hand-off
    def chunks(a, scratch_bytes):
        a.ds_scratch, a.ds_scratch_bytes = 16, scratch_bytes
        n, hc, gc, cb = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        assert lib.rfa_bwd_ds_chunks(C.byref(a), C.byref(n), C.byref(hc), C.byref(gc), C.byref(cb)) == 0
        return n.value, hc.value, gc.value, cb.value

    per = lambda S: (S // 32) * (S // 32 + 1) // 2 * 2048       # one query head of a dense causal call
    GiB = 1 << 30
    # the headline fits 2.5 GiB in one piece (2.0 GiB); 0 = "at least the whole hand-off"
    assert chunks(args(1, 8192, 8192, 32, 8, causal=True), 5 * GiB // 2) == (1, 8, 4, 32 * per(8192))
    assert chunks(args(1, 8192, 8192, 32, 8, causal=True), 0) == (1, 8, 4, 32 * per(8192))
    # S = 16384: 8.6 GB of dS -> 4 chunks of 2 K/V heads (8 query heads, 2.15 GB); the launch plan is that of a 2-head launch
    a16 = args(1, 16384, 16384, 32, 8, causal=True)
    assert ds(a16) == 32 * per(16384) and chunks(a16, 5 * GiB // 2) == (4, 2, 4, 8 * per(16384))
    assert plan(a16) == (_C.DKDV_256, 4) and ws(a16) == 4 * unit32(16384, 8)
    # S = 32768 (34 GB of dS, the long-context case that used to drop to the 7-GEMM form): one K/V head's 4 query heads
    # (4.3 GB) do not fit -> 16 chunks of 2 query heads of ONE K/V head, fp32 partials accumulated across the fractions
    a32 = args(1, 32768, 32768, 32, 8, causal=True)
    assert chunks(a32, 5 * GiB // 2) == (16, 1, 2, 2 * per(32768))
    g = C.c_int32()
    assert lib.rfa_bwd_plan(C.byref(a32), None, None, C.byref(g)) == 0 and g.value == 1          # still the 5-GEMM form
    assert ws(a32) == plan(a32)[1] * unit32(32768, 8)
    # MHA: chunks of whole heads; below one query head's share: the 7-GEMM form
    assert chunks(args(1, 32768, 32768, 32, 32, causal=True), 5 * GiB // 2) == (16, 2, 1, 2 * per(32768))
    assert chunks(args(1, 32768, 32768, 32, 8, causal=True), GiB // 2)[0] == 0
    assert lib.rfa_bwd_ds_scratch_min_bytes(C.byref(args(1, 32768, 32768, 32, 8, causal=True))) == per(32768)
    # two-phase (ring step) calls chunk by whole K/V heads only (their partials are not accumulated across launches)
    assert chunks(args(1, 32768, 32768, 32, 8, causal=True, phases=_C.BWD_COMPUTE), 5 * GiB // 2)[0] == 0
    assert chunks(args(1, 16384, 16384, 32, 8, causal=True, phases=_C.BWD_COMPUTE), 5 * GiB // 2) == (4, 2, 4, 8 * per(16384))
    # head dim 256 has no chunked form
    assert chunks(args(1, 8192, 8192, 16, 4, D=256, causal=True), GiB)[0] == 0
    # invalid plan fields are rejected
    bad = args(1, 64, 64, 1, 1, form=7)
    bad.dout = bad.q = bad.k = bad.v = bad.lse = bad.delta = bad.dq = bad.dk = bad.dv = 16
    assert lib.rfa_bwd(C.byref(bad), None) == -8


def test_forward_split_plan_is_a_function_of_the_arguments(built):
    """rfa_fwd_workspace_bytes reports the split-KV plan of a forward call (no device needed): few rows against many key
    tiles are split; round 6: between 49 and 256 workgroups of 256 rows the share count is the one with the smallest estimated
    makespan (rfa_api.cpp fwd_plan; tools/plan_sweep.py measures it against every forced form); a named form never splits by
    itself, a forced share count always does"""
    from ring_flash_attn import _C

    lib = _C.load()

    def plan(B, Sq, Sk, H, Hk, D=128, form=_C.FWD_AUTO, nsplit=0):
        a = _C.FwdArgs()
        a.B, a.Sq, a.Sk, a.H, a.Hk, a.D, a.dtype, a.causal = B, Sq, Sk, H, Hk, D, 0, 1
        a.fwd_form, a.kv_nsplit = form, nsplit
        n = C.c_int32()
        nbytes = lib.rfa_fwd_workspace_bytes(C.byref(a), C.byref(n))
        assert nbytes == (n.value * B * Sq * H * (D + 1) * 4 if n.value > 1 else 0)
        return n.value

    assert plan(1, 8192, 8192, 32, 8) == 1                       # the headline: 1024 workgroups
    assert plan(1, 256, 4096, 4, 2) == 2                         # few rows, 64 tiles
    assert plan(1, 2048, 16384, 16, 8) == 2 and plan(1, 2048, 8192, 16, 8) == 2      # llama3 head groups: 128 workgroups -> 256
    assert plan(1, 2048, 16384, 8, 4) == 4 and plan(1, 2048, 16384, 2, 1) == 8       # ... half a layer, one K/V head (128-row form)
    assert plan(1, 2048, 4096, 16, 8) == 2
    assert plan(1, 2048, 2048, 16, 8) == 1 and plan(1, 2048, 1536, 16, 8) == 1       # short key chains are not split
    assert plan(1, 2048, 8192, 24, 6) == 1 and plan(1, 2048, 8192, 20, 5) in (1, 3)  # 160 .. 224 workgroups: never 1.25 rounds
    assert plan(1, 2048, 16384, 16, 8, form=_C.FWD_8x32) == 1 and plan(1, 2048, 16384, 16, 8, form=_C.FWD_8x32, nsplit=3) == 3
    assert plan(1, 2048, 16384, 16, 8, nsplit=1) == 1
    assert plan(1, 2048, 16384, 16, 8, D=96) == 1                # (head dims without the split forms)


def test_bench_accounting_matches_the_survey():
    """bench.py's algorithmic FLOP and byte accounting (no GPU): SURVEY.md section 8(d) fixes the headline at
    3.5 * 2*B*H*D*8192^2 * W = 1.9242e12 * W FLOP per GPU and iteration (causal counted as half, bwd = 2.5 fwd)
    and 100 %-MFMA ceilings of 1299 / 650 / 325 / 162 it/s at W = 1 / 2 / 4 / 8"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for w, ceiling in ((1, 1299), (2, 650), (4, 325), (8, 162)):
        per_gpu = 3.5 * b.fwd_flops_per_gpu("zigzag", w)
        assert abs(per_gpu - 1.9242e12 * w) / (1.9242e12 * w) < 1e-3
        assert abs(b.MFMA_PEAK_TFLOPS * 1e12 / per_gpu - ceiling) < 1.0
    m = 2 * 8192 * 8 * 128 * 2                                   # K + V of one rank, bf16, Hk = 8 (32 MiB)
    assert m == 32 << 20
    # reference protocol: (W-1) M forward, (W-1) M + W * 2M (fp32 dK/dV) backward
    assert b.comm_bytes_per_iter("ring", False, 8, 8) == 7 * m + 7 * m + 8 * 2 * m
    # gather form: one all-gather (kept for the backward) + the bf16 contributions of the other W-1 chunks
    assert b.comm_bytes_per_iter("gather", False, 8, 8) == 7 * m + 7 * m
    assert b.comm_bytes_per_iter("gather", True, 8, 8) == 7 * m + 7 * 2 * m


def test_bench_cpu_baseline_times_a_whole_iteration(single_rank_group, monkeypatch):
    """bench.py's `cpu_baseline` leg (no GPU): by default ONE whole iteration is timed — every kv-head group, nothing
    extrapolated; the one-group sample is only the fallback for hosts predicted to need more than two minutes; where
    /root/reference exists the unmodified reference is attempted and, at world size 1 under gloo (it sends dK/dV to
    its own rank), replaced by the port instead of failing the bench"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    import sys
    import types

    monkeypatch.setattr(b, "SEQ", 512)
    saved = dict(sys.modules)            # the reference harness installs the oracle as `flash_attn`: undone below
    try:
        r = b.cpu_baseline(8, False)
        assert r["kind"] == "port" and r["extrapolated"] is False and r["sample"].startswith("all 8 kv-head groups")
        assert r["value"] > 0 and r["unit"] == "iters/sec" and r["cores"] >= 1
        # a host predicted to be too slow: the bounded one-group sample, labelled as such
        clock = iter([0.0, 100.0] + [200.0 + i for i in range(8)])      # (a stub for the module's `time`, not the global one)
        monkeypatch.setattr(b, "time", types.SimpleNamespace(perf_counter=lambda: next(clock)))
        r = b.cpu_baseline(8, False)
        assert r["extrapolated"] is True and "1 of 8 kv-head groups" in r["sample"]
    finally:
        for name in list(sys.modules):
            if name not in saved:
                del sys.modules[name]
        sys.modules.update(saved)


def test_wide_head_dim_kernels_fit_the_register_file(tmp_path):
    """csrc/rfa_bigd.hip runs one wave per SIMD so that a 256-wide row's fragments and accumulators fit the 512-entry
    register file; a kernel that starts spilling to scratch inside its tile loop would still be correct and several times
    slower (the one-launch dK + dV form did: DESIGN.md section 3.2).  Audit the generated code: forward and both dK/dV
    launches without scratch, the dQ kernel within its known 16 spilled registers; the one-launch dK + dV form of head dims
    <= 192 (round 6) with at most 4 spilled registers (outside its tile loop)."""
    import re
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "ring-flash-attention_amd", "csrc", "rfa_bigd.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", str(tmp_path / "x.o"),
                        "-Wno-inline-asm", "-save-temps=obj"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(tmp_path / [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")][0]).read()
    kernels = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)
    seen = {}
    for name, vgpr, spill in kernels:
        kind = next((k for k in ("fwd_big", "dq_big", "dkdv_big", "dkdv_fused_big") if k in name), None)
        if kind:
            seen.setdefault(kind, []).append((int(vgpr), int(spill)))
    # (two io dtypes x two widths: the three-quarter instances of head dims <= 192 and the full 256-wide ones)
    assert len(seen.get("fwd_big", [])) == 4 and len(seen.get("dq_big", [])) == 4 and len(seen.get("dkdv_big", [])) == 8, seen
    assert all(v <= 512 and s == 0 for v, s in seen["fwd_big"] + seen["dkdv_big"]), seen
    assert all(s <= 32 for _, s in seen["dq_big"]), seen
    # round 6: dK + dV in ONE launch for head dims <= 192 (V rows in LDS): two io dtypes x (plain, dropout), no scratch.  (The
    # 256-wide instance is not built: both accumulator sets fill the AGPR half and its tile loop goes to scratch — measured
    # 293 against 450 TFLOP/s; head dims > 192 keep one launch per tensor.)
    # (two registers of its prologue are spilled; the tile loop itself touches no scratch)
    assert len(seen.get("dkdv_fused_big", [])) == 4 and all(v <= 512 and s <= 4 for v, s in seen["dkdv_fused_big"]), seen


def test_tuned_kernels_run_two_waves_per_simd_without_scratch(tmp_path):
    """csrc/rfa_fwd.hip / rfa_bwd.hip / rfa_dqs.hip are built around TWO waves per SIMD: every instance (head dims 128, 96
    and 64, full and padded, windows, dropout, the 128-row forward, both dK/dV forms with and without the dS spill) must stay
    within 256 registers and off the scratch — the round-4 experiments that did not (the software-pipelined forward: Q
    fragments in scratch, 3x slower) were correct and useless.  Also pins the instance sets the launchers dispatch to:
    head dims 65 .. 96 have their own three-block instances of the three 7-GEMM kernels (bf16 and fp16)."""
    import re
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    found = {}
    for f in ("rfa_fwd.hip", "rfa_bwd.hip", "rfa_dqs.hip"):
        out = tmp_path / f
        out.mkdir()
        src = os.path.join(ROOT, "ring-flash-attention_amd", "csrc", f)
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", str(out / "x.o"),
                            "-Wno-inline-asm", "-Wno-unused-result", "-save-temps=obj"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        text = open(out / [g for g in os.listdir(out) if g.endswith("gfx950.s")][0]).read()
        for name, priv, vgpr, spill in re.findall(
                r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text):
            found[name] = (int(priv), int(vgpr), int(spill))
    kinds = {k: [n for n in found if k in n] for k in ("fwd_kernel", "dq_kernel", "dkdv_kernel", "dq_ds_kernel")}
    assert all(len(v) >= 4 for v in kinds.values()), {k: len(v) for k, v in kinds.items()}
    # (known exception, unchanged since round 2: the zero-PADDED 128-wide dK/dV instances with a window or dropout — head dims
    #  97 .. 127, or 65 .. 127 with a window / dropout — hold their staging registers on top of a full file and keep one or two
    #  values in scratch outside the MFMA blocks: 8 - 20 bytes)
    def padded_special(n):
        return "dkdv_kernel" in n and "Li128ELb0E" in n
    bad = {n: v for n, v in found.items() if any(k in n for k in kinds) and
           (v[1] > 256 or ((v[0] > 32 or v[2] > 4) if padded_special(n) else (v[0] != 0 or v[2] != 0)))}
    assert not bad, bad
    for k, per_dtype in (("fwd_kernel", 3), ("dq_kernel", 2), ("dkdv_kernel", 2)):      # (forward: + the 128-row form of D = 96)
        n96 = [n for n in kinds[k] if "Li96E" in n]
        assert len(n96) == 2 * per_dtype, (k, n96)


def test_zero_outside_rezeroes_only_the_rows_that_leave_the_range(built):
    """backend.zero_outside (round 6): the llama3 backward's dK/dV contribution buffer is kept and reused —
    its rows outside the local key slice must read as zero in the reduce-scatter, the rows inside are overwritten by the
    dK/dV kernel on every call.  Zeroed once when made; between calls only the rows that LEAVE the range are zeroed again
    (host logic: no device needed)."""
    import torch
    from ring_flash_attn.backend import HipBackend

    be, dev = HipBackend(), torch.device("cpu")
    b = be.zero_outside((8, 2), torch.float32, dev, 2, 6)
    assert b.shape == (8, 2) and not b.any()
    b[2:6] = 1.0                                             # (the kernel's stores)
    b2 = be.zero_outside((8, 2), torch.float32, dev, 3, 5)   # the range shrinks on both sides
    assert b2.data_ptr() == b.data_ptr()
    assert b2[:, 0].tolist() == [0, 0, 0, 1, 1, 0, 0, 0]
    b2[3:5] = 2.0
    b3 = be.zero_outside((8, 2), torch.float32, dev, 1, 7)   # grows: nothing to zero (the kernel overwrites [1, 7))
    assert b3[:, 0].tolist() == [0, 0, 0, 2, 2, 0, 0, 0]
    b3[1:7] = 3.0
    b4 = be.zero_outside((8, 2), torch.float32, dev, 6, 8)   # moves: [1, 6) left the range
    assert b4[:, 0].tolist() == [0, 0, 0, 0, 0, 0, 3, 0]
    # another slot / dim is another buffer; dim = 1: (2, rows, ...) layouts
    c = be.zero_outside((2, 8), torch.float32, dev, 2, 6, slot=1, dim=1)
    assert c.data_ptr() != b.data_ptr() and not c.any()
    c[:, 2:6] = 1.0
    c2 = be.zero_outside((2, 8), torch.float32, dev, 4, 6, slot=1, dim=1)
    assert c2[0].tolist() == [0, 0, 0, 0, 1, 1, 0, 0]
    be.release_scratch()
    assert not be.zero_outside((8, 2), torch.float32, dev, 0, 8).any()
