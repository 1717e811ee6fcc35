"""GPU parity of the shipped `flash_attn` compatibility package (ring-flash-attention_amd/shims/flash_attn, an opt-in sys.path root):
the four private operator functions the reference imports and the public single-device functions its
tests / benchmarks use as ground truth, against the CPU oracle's functions of the same names
(oracle/flash_attn_ref.py) on the same seeded inputs.  Tolerances as in test_gpu_kernels.py."""
import os
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.extended]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ring-flash-attention_amd"))
sys.path.insert(0, os.path.join(ROOT, "ring-flash-attention_amd", "shims"))     # INTEGRATION.md route B opt-in
BF = torch.bfloat16


def _check(name, got, ref, atol, rtol=0.0):
    """all criteria of tests/_tol.py; the kind follows from the historical (atol, rtol) pair of the call site:
    (2e-2, 0) out, (1e-3, 0) lse, (1e-2, 2e-2) gradients — single-rank bounds"""
    import _tol

    kind = {(2e-2, 0.0): "out", (1e-3, 0.0): "lse", (1e-2, 2e-2): "grad"}.get((atol, rtol))
    if kind is not None:
        return _tol.compare(name, got, ref, kind + "")
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{name}: {got.shape} vs {ref.shape}"
    diff = (got - ref).abs().max().item()
    lim = atol + rtol * ref.abs().max().item()
    assert diff <= lim, f"{name}: max|err| {diff:.3e} > {lim:.3e}"


@pytest.mark.parametrize("causal", [True, False])
def test_private_dense_functions_match_oracle(built, causal):
    from flash_attn import flash_attn_interface as F
    from oracle import flash_attn_ref as O

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    B, Sq, Sk, H, Hk, D = 2, 200, 264, 4, 2, 128
    q = torch.randn(B, Sq, H, D, generator=g).to(BF)
    k = torch.randn(B, Sk, Hk, D, generator=g).to(BF)
    v = torch.randn(B, Sk, Hk, D, generator=g).to(BF)
    do = torch.randn(B, Sq, H, D, generator=g).to(BF)
    scale = D ** -0.5
    ro, rl, _, _ = O._flash_attn_forward(q, k, v, 0.0, scale, causal)
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    rd = O._flash_attn_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, 0.0, scale, causal)

    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    res = F._flash_attn_forward(qd, kd, vd, 0.0, scale, causal, window_size_left=-1, window_size_right=-1,
                                softcap=0.0, alibi_slopes=None, return_softmax=False)
    assert len(res) == 4 and res[2] is None and res[3] is None          # the 4-tuple zigzag…:53-57 expects
    out, lse = res[0], res[1]
    assert out.dtype == BF and lse.dtype == torch.float32 and lse.shape == (B, H, Sq)
    # caller-provided, sliced gradient views (zigzag_ring_flash_attn.py:137-139)
    dq_buf = torch.full((B, Sq + 56, H, D), 7.0, dtype=BF, device=dev)
    dkv_buf = torch.full((2, B, Sk + 24, Hk, D), 7.0, dtype=BF, device=dev)
    dq, dk, dv = dq_buf[:, :Sq], dkv_buf[0][:, :Sk], dkv_buf[1][:, :Sk]
    d = F._flash_attn_backward(dod, qd, kd, vd, out, lse, dq, dk, dv, 0.0, scale, causal,
                               window_size_left=-1, window_size_right=-1, softcap=0.0, alibi_slopes=None,
                               deterministic=False, rng_state=None)
    _check("out", out, ro, 2e-2)
    _check("lse", lse, rl, 1e-3)
    _check("softmax_d", d, rd, 2e-2, 1e-2)
    for n, a, b in (("dq", dq, rdq), ("dk", dk, rdk), ("dv", dv, rdv)):
        _check(n, a, b, 1e-2, 2e-2)
    assert (dq_buf[:, Sq:] == 7).all() and (dkv_buf[:, :, Sk:] == 7).all(), "wrote outside the views"


def test_private_varlen_functions_match_oracle(built):
    from flash_attn import flash_attn_interface as F
    from oracle import flash_attn_ref as O

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(12)
    cu = torch.tensor([0, 120, 120, 1248, 1400], dtype=torch.int32)     # ragged, one empty sequence
    T, H, Hk, D = int(cu[-1]), 6, 2, 64
    mx = int((cu[1:] - cu[:-1]).max())
    q = torch.randn(T, H, D, generator=g).to(BF)
    k = torch.randn(T, Hk, D, generator=g).to(BF)
    v = torch.randn(T, Hk, D, generator=g).to(BF)
    do = torch.randn(T, H, D, generator=g).to(BF)
    scale = D ** -0.5
    ro, rl, _, _ = O._flash_attn_varlen_forward(q, k, v, cu, cu, mx, mx, 0.0, scale, True)
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_varlen_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, cu, cu, mx, mx, 0.0, scale, True)

    qd, kd, vd, dod, cud = (t.to(dev) for t in (q, k, v, do, cu))
    out, lse, _, _ = F._flash_attn_varlen_forward(qd, kd, vd, cud, cud, mx, mx, 0.0, scale, True)
    assert lse.shape == (H, T)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    F._flash_attn_varlen_backward(dod, qd, kd, vd, out, lse, dq, dk, dv, cud, cud, mx, mx, 0.0, scale, True)
    _check("out", out, ro, 2e-2)
    _check("lse", lse, rl, 1e-3)
    for n, a, b in (("dq", dq, rdq), ("dk", dk, rdk), ("dv", dv, rdv)):
        _check(n, a, b, 1e-2, 2e-2)


def test_public_functions_autograd(built):
    """The ground-truth functions of the reference's tests: (out, lse, None) with return_attn_probs,
    differentiable, packed variants are views of the unpacked ones."""
    import flash_attn as FA
    from oracle import flash_attn_ref as O

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(13)
    B, S, H, D = 1, 384, 4, 128
    qkv = torch.randn(B, S, 3, H, D, generator=g).to(BF)
    do = torch.randn(B, S, H, D, generator=g).to(BF)
    ref_in = qkv.double().requires_grad_(True)
    ro, rl = O.full_attention_fp64(ref_in[:, :, 0], ref_in[:, :, 1], ref_in[:, :, 2], True)
    ro.backward(do.double())

    x = qkv.to(dev).requires_grad_(True)
    out, lse, none = FA.flash_attn_qkvpacked_func(x, causal=True, return_attn_probs=True)
    assert none is None
    out.backward(do.to(dev))
    _check("out", out, ro, 2e-2)
    _check("lse", lse, rl, 1e-3)
    _check("dqkv", x.grad, ref_in.grad, 1e-2, 2e-2)

    # kvpacked + varlen public forms agree with the packed dense one
    q, kv = qkv[:, :, 0].to(dev), qkv[:, :, 1:].to(dev)
    o2 = FA.flash_attn_kvpacked_func(q, kv, causal=True)
    assert torch.equal(o2, out.detach())
    cu = torch.tensor([0, S], dtype=torch.int32, device=dev)
    o3, l3, _ = FA.flash_attn_varlen_qkvpacked_func(qkv[0].to(dev), cu, S, causal=True, return_attn_probs=True)
    assert torch.equal(o3, out.detach()[0]) and l3.shape == (H, S)
    # dropout through the public function (the seed is drawn from torch's CPU generator and kept on the autograd node):
    # against the oracle with the same seed — forward and backward
    from ring_flash_attn._common import draw_dropout_seed

    xd = qkv.to(dev).requires_grad_(True)
    torch.manual_seed(321)
    od = FA.flash_attn_func(xd[:, :, 0], xd[:, :, 1], xd[:, :, 2], dropout_p=0.1, causal=True)
    od.backward(do.to(dev))
    torch.manual_seed(321)
    rng = torch.tensor([draw_dropout_seed(), 0])
    rod, rld, _, _ = O._flash_attn_forward(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 0.1, D ** -0.5, True, rng_state=rng)
    gq, gk, gv = (torch.empty_like(qkv[:, :, 0]) for _ in range(3))
    O._flash_attn_backward(do, qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], rod, rld, gq, gk, gv, 0.1, D ** -0.5, True, rng_state=rng)
    _check("dropout out", od, rod, 2e-2)
    _check("dropout dqkv", xd.grad, torch.stack([gq, gk, gv], dim=2), 1e-2, 2e-2)
    assert (od.detach() - out.detach()).abs().max() > 0.05
    with pytest.raises(NotImplementedError):                       # dropout together with a window
        FA.flash_attn_func(q, kv[:, :, 0], kv[:, :, 1], dropout_p=0.1, window_size=(64, 0), causal=True)
    # sliding window through the public function: against the oracle with the same window
    xw = qkv.to(dev).requires_grad_(True)
    ow = FA.flash_attn_qkvpacked_func(xw, window_size=(128, 0), causal=True)
    ow.backward(do.to(dev))
    rw, rlw, _, _ = O._flash_attn_forward(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 0.0, D ** -0.5, True, 128, 0)
    _check("window out", ow, rw, 2e-2)
    assert (ow.detach() - out.detach()).abs().max() > 1e-2           # the window changed the result


@pytest.mark.parametrize("D", [128, 96])
def test_fp16_kernels_match_oracle(built, D):
    """The fp16 instances of every attention kernel (the reference accepts fp16 as well as bf16): D = 128
    takes the LDS-DMA path, D = 96 the zero-padded register path."""
    from flash_attn import flash_attn_interface as F
    from oracle import flash_attn_ref as O

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21)
    B, Sq, Sk, H, Hk = 1, 330, 330, 4, 2
    H16 = torch.float16
    q = torch.randn(B, Sq, H, D, generator=g).to(H16)
    k = torch.randn(B, Sk, Hk, D, generator=g).to(H16)
    v = torch.randn(B, Sk, Hk, D, generator=g).to(H16)
    do = torch.randn(B, Sq, H, D, generator=g).to(H16)
    scale = D ** -0.5
    ro, rl, _, _ = O._flash_attn_forward(q, k, v, 0.0, scale, True)
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, 0.0, scale, True)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    out, lse, _, _ = F._flash_attn_forward(qd, kd, vd, 0.0, scale, True)
    assert out.dtype == H16
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    F._flash_attn_backward(dod, qd, kd, vd, out, lse, dq, dk, dv, 0.0, scale, True)
    # fp16 has 3 more mantissa bits than bf16: tighter than the bf16 tolerances
    _check("out", out, ro, 4e-3)
    _check("lse", lse, rl, 1e-3)
    for n, a, b in (("dq", dq, rdq), ("dk", dk, rdk), ("dv", dv, rdv)):
        _check(n, a, b, 3e-3, 5e-3)
