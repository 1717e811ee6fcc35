"""Pins the oracle (oracle/flash_attn_ref.py, oracle/attn_ref.c).

The reference's tests contain no numeric golden vectors for the flash_attn boundary
(SURVEY.md §8c), so the oracle is pinned against an INDEPENDENT formulation: plain softmax
attention in fp64 with torch autograd — the same role `flash_attn_*_func` on the full sequence
plays in the reference tests (test/test_zigzag_ring_flash_attn_func.py:55-63).
Tolerances: the oracle computes in fp32 from bf16 inputs -> 2e-5 abs vs fp64 on fp32 outputs;
outputs rounded to bf16 -> half a bf16 ulp (2^-9 relative).
"""
import ctypes as C
import os

import pytest
import torch

from oracle import flash_attn_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [
    # B, Sq, Sk, H, Hk, D, causal
    (1, 64, 64, 2, 2, 32, True),
    (2, 100, 100, 4, 2, 64, True),
    (1, 48, 130, 4, 1, 32, True),     # bottom-right aligned
    (1, 130, 48, 2, 2, 32, True),     # rows without keys
    (1, 77, 93, 3, 3, 16, False),
]


def _rand(shape, seed, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype)


@pytest.mark.parametrize("B,Sq,Sk,H,Hk,D,causal", SHAPES)
def test_oracle_vs_fp64_autograd(B, Sq, Sk, H, Hk, D, causal):
    q, k, v = _rand((B, Sq, H, D), 1), _rand((B, Sk, Hk, D), 2), _rand((B, Sk, Hk, D), 3)
    do = _rand((B, Sq, H, D), 4)
    scale = D ** -0.5
    out, lse, _, _ = R._flash_attn_forward(q.float(), k.float(), v.float(), 0.0, scale, causal)
    qd, kd, vd = [t.double().requires_grad_(True) for t in (q, k, v)]
    ro, rl = R.full_attention_fp64(qd, kd, vd, causal, scale)
    empty = torch.isinf(rl) & (rl < 0)
    assert torch.equal(torch.isinf(lse) & (lse > 0), empty)       # flash_attn: +inf for empty rows
    assert (out.double() - ro).abs().max() < 2e-5
    assert (lse.double() - rl)[~empty].abs().max() < 2e-5
    ro.backward(do.double())
    dq, dk, dv = torch.empty_like(q, dtype=torch.float32), torch.empty_like(k, dtype=torch.float32), torch.empty_like(v, dtype=torch.float32)
    R._flash_attn_backward(do.float(), q.float(), k.float(), v.float(), out, lse, dq, dk, dv, 0.0, scale, causal)
    for got, ref in ((dq, qd.grad), (dk, kd.grad), (dv, vd.grad)):
        assert (got.double() - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max().item())


def test_oracle_rounds_like_flash_attn():
    q, k, v = _rand((1, 40, 2, 32), 1), _rand((1, 40, 2, 32), 2), _rand((1, 40, 2, 32), 3)
    out, lse, _, _ = R._flash_attn_forward(q, k, v, 0.0, 32 ** -0.5, True)
    assert out.dtype == torch.bfloat16 and lse.dtype == torch.float32 and lse.shape == (1, 2, 40)
    ro, _ = R.full_attention_fp64(q, k, v, True)
    assert (out.double() - ro).abs().max() <= 2 ** -8 * ro.abs().max()


def test_oracle_varlen_equals_per_sequence_dense():
    cu = torch.tensor([0, 15, 156, 200], dtype=torch.int32)
    cuk = torch.tensor([0, 40, 190, 300], dtype=torch.int32)
    H, Hk, D = 4, 2, 32
    q, k, v = _rand((200, H, D), 1), _rand((300, Hk, D), 2), _rand((300, Hk, D), 3)
    do = _rand((200, H, D), 4)
    out, lse, _, _ = R._flash_attn_varlen_forward(q, k, v, cu, cuk, 141, 150, 0.0, D ** -0.5, True)
    assert lse.shape == (H, 200)                                   # packed (nheads, total) layout
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    R._flash_attn_varlen_backward(do, q, k, v, out, lse, dq, dk, dv, cu, cuk, 141, 150, 0.0, D ** -0.5, True)
    for i in range(3):
        a, b, c, d = cu[i], cu[i + 1], cuk[i], cuk[i + 1]
        o1, l1, _, _ = R._flash_attn_forward(q[None, a:b], k[None, c:d], v[None, c:d], 0.0, D ** -0.5, True)
        assert torch.equal(o1[0], out[a:b]) and torch.equal(l1[0], lse[:, a:b])
        g = [torch.zeros_like(t) for t in (q[None, a:b], k[None, c:d], v[None, c:d])]
        R._flash_attn_backward(do[None, a:b], q[None, a:b], k[None, c:d], v[None, c:d], o1, l1, *g, 0.0, D ** -0.5, True)
        assert torch.equal(g[0][0], dq[a:b]) and torch.equal(g[1][0], dk[c:d]) and torch.equal(g[2][0], dv[c:d])


def _load_c_oracle(built):
    lib = C.CDLL(built.ORACLE_LIB)
    f = C.POINTER(C.c_float)
    i32 = C.POINTER(C.c_int32)
    lib.rfa_ref_fwd.argtypes = [f, f, f, f, f] + [C.c_int] * 6 + [i32, i32, C.c_int64, C.c_float, C.c_int]
    lib.rfa_ref_bwd.argtypes = [f] * 9 + [C.c_int] * 6 + [i32, i32, C.c_int64, C.c_int64, C.c_float, C.c_int]
    lib.rfa_ref_merge.argtypes = [f, f, f, f] + [C.c_int] * 4
    return lib


def _fp(t):
    return t.contiguous().data_ptr()


@pytest.mark.parametrize("B,Sq,Sk,H,Hk,D,causal", SHAPES[:4])
def test_c_oracle_matches_python_oracle(built, B, Sq, Sk, H, Hk, D, causal):
    lib = _load_c_oracle(built)
    q, k, v, do = [_rand(s, i).float() for i, s in enumerate([(B, Sq, H, D), (B, Sk, Hk, D), (B, Sk, Hk, D), (B, Sq, H, D)])]
    scale = D ** -0.5
    out, lse = torch.empty_like(q), torch.empty(B, H, Sq)
    P = lambda t: C.cast(_fp(t), C.POINTER(C.c_float))
    assert lib.rfa_ref_fwd(P(q), P(k), P(v), P(out), P(lse), B, H, Hk, D, Sq, Sk, None, None, 0, scale, int(causal)) == 0
    po, pl, _, _ = R._flash_attn_forward(q, k, v, 0.0, scale, causal)
    assert (out - po).abs().max() < 1e-5
    fin = ~torch.isinf(pl)
    assert torch.equal(torch.isinf(lse), torch.isinf(pl)) and (lse - pl)[fin].abs().max() < 1e-5
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    assert lib.rfa_ref_bwd(P(do), P(q), P(k), P(v), P(out), P(lse), P(dq), P(dk), P(dv), B, H, Hk, D, Sq, Sk,
                           None, None, 0, 0, scale, int(causal)) == 0
    gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    R._flash_attn_backward(do, q, k, v, po, pl, gq, gk, gv, 0.0, scale, causal)
    for a, b in ((dq, gq), (dk, gk), (dv, gv)):
        assert (a - b).abs().max() < 5e-5 * max(1.0, b.abs().max().item())


def test_c_oracle_merge_is_reference_formula(built):
    """rfa_ref_merge restates _update_out_and_lse (reference utils.py:40-48)."""
    import torch.nn.functional as F

    lib = _load_c_oracle(built)
    B, S, H, D = 2, 17, 3, 8
    g = torch.Generator().manual_seed(0)
    out, bo = torch.randn(B, S, H, D, generator=g), torch.randn(B, S, H, D, generator=g)
    lse, bl = torch.randn(B, H, S, generator=g) * 3, torch.randn(B, H, S, generator=g) * 3
    l4, b4 = lse.transpose(1, 2).unsqueeze(-1), bl.transpose(1, 2).unsqueeze(-1)
    ref_o = out - torch.sigmoid(b4 - l4) * (out - bo)
    ref_l = (l4 - F.logsigmoid(l4 - b4)).squeeze(-1).transpose(1, 2)
    o2, l2 = out.clone(), lse.clone()
    P = lambda t: C.cast(_fp(t), C.POINTER(C.c_float))
    lib.rfa_ref_merge(P(o2), P(l2), P(bo), P(bl.contiguous()), B, S, H, D)
    assert (o2 - ref_o).abs().max() < 1e-6 and (l2 - ref_l).abs().max() < 1e-6
    assert (l2 - torch.logaddexp(lse, bl)).abs().max() < 1e-6      # == logaddexp


@pytest.mark.parametrize("Sq,Sk,causal,window", [
    (70, 70, True, (16, 0)), (70, 70, False, (9, 5)), (40, 100, True, (30, -1)), (100, 40, False, (-1, 7)),
    (64, 64, False, (0, 0)),
])
def test_oracle_sliding_window_vs_explicit_mask(Sq, Sk, causal, window):
    """window semantics (flash_attn: key j visible to query i iff i + (Sk-Sq) - left <= j <= i + (Sk-Sq) + right,
    causal forces right = 0), pinned against an explicitly built boolean mask in fp64 with autograd"""
    H, Hk, D = 4, 2, 32
    q, k, v, do = _rand((1, Sq, H, D), 1), _rand((1, Sk, Hk, D), 2), _rand((1, Sk, Hk, D), 3), _rand((1, Sq, H, D), 4)
    scale = D ** -0.5
    wl, wr = window
    if causal:
        wr = 0
    qi = torch.arange(Sq).unsqueeze(1) + (Sk - Sq)
    kj = torch.arange(Sk).unsqueeze(0)
    vis = torch.ones(Sq, Sk, dtype=torch.bool)
    if wr >= 0:
        vis &= kj <= qi + wr
    if wl >= 0:
        vis &= kj >= qi - wl
    qd, kd, vd = [t.double().requires_grad_(True) for t in (q, k, v)]
    s = torch.einsum("bqhd,bkhd->bhqk", qd, kd.repeat_interleave(H // Hk, dim=2)) * scale
    s = s.masked_fill(~vis, float("-inf"))
    p = torch.nan_to_num(torch.softmax(s, -1), nan=0.0)
    ro = torch.einsum("bhqk,bkhd->bqhd", p, vd.repeat_interleave(H // Hk, dim=2))
    ro.backward(do.double())
    out, lse, _, _ = R._flash_attn_forward(q.float(), k.float(), v.float(), 0.0, scale, causal, window[0], window[1])
    assert (out.double() - ro).abs().max() < 2e-5
    dq, dk, dv = (torch.empty_like(t, dtype=torch.float32) for t in (q, k, v))
    R._flash_attn_backward(do.float(), q.float(), k.float(), v.float(), out, lse, dq, dk, dv, 0.0, scale, causal,
                           window[0], window[1])
    for got, ref in ((dq, qd.grad), (dk, kd.grad), (dv, vd.grad)):
        assert (got.double() - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max().item())


# ------------------------------------------------------------------------------------------------------------
# dropout: the mask is this library's own definition (include/rfa.h, csrc/rfa_common.hpp: drop_word), restated by
# oracle/flash_attn_ref.py: dropout_keep; the arithmetic around it is flash_attn's.
def _dropout_fp64(q, k, v, do, causal, p, keep):
    """independent formulation: explicit keep mask (B,H,Sq,Sk) applied to the softmax, fp64 autograd"""
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    qd, kd, vd = [t.double().requires_grad_(True) for t in (q, k, v)]
    s = torch.einsum("bqhd,bkhd->bhqk", qd, kd.repeat_interleave(H // Hk, dim=2)) * D ** -0.5
    if causal:
        i = torch.arange(Sq).view(-1, 1) + (Sk - Sq)
        s = s.masked_fill(torch.arange(Sk).view(1, -1) > i, float("-inf"))
    pr = torch.nan_to_num(torch.softmax(s, -1), nan=0.0)
    # kept probabilities are rescaled by 256 / round((1 - p) 256): the reciprocal of the rate the mask really keeps at
    out = torch.einsum("bhqk,bkhd->bqhd", torch.where(keep, pr * (256.0 / R.drop_threshold(p)), torch.zeros_like(pr)),
                       vd.repeat_interleave(H // Hk, dim=2))
    out.backward(do.double())
    return out.detach(), qd.grad, kd.grad, vd.grad


@pytest.mark.parametrize("B,Sq,Sk,H,Hk,D,causal", SHAPES[:3] + SHAPES[4:])
def test_oracle_dropout_vs_fp64_autograd(B, Sq, Sk, H, Hk, D, causal):
    q, k, v = _rand((B, Sq, H, D), 1), _rand((B, Sk, Hk, D), 2), _rand((B, Sk, Hk, D), 3)
    do = _rand((B, Sq, H, D), 4)
    p, scale = 0.17, D ** -0.5
    rng = torch.tensor([0x1234_5678_9ABC_DEF, 0])
    out, lse, _, rs = R._flash_attn_forward(q.float(), k.float(), v.float(), p, scale, causal, rng_state=rng)
    lse0 = R._flash_attn_forward(q.float(), k.float(), v.float(), 0.0, scale, causal)[1]
    assert torch.equal(rs, rng) and torch.equal(lse, lse0)        # lse is that of the undropped softmax
    keep = torch.stack([R.dropout_keep(int(rng[0]), p, b, range(H), range(Sq), range(Sk)) for b in range(B)])
    frac = keep.float().mean().item()
    assert abs(frac - R.drop_threshold(p) / 256) < 0.02           # keep probability = round((1-p) 256) / 256
    ro, rq, rk, rv = _dropout_fp64(q, k, v, do, causal, p, keep)
    assert (out.double() - ro).abs().max() < 2e-5
    dq, dk, dv = torch.empty_like(q, dtype=torch.float32), torch.empty_like(k, dtype=torch.float32), torch.empty_like(v, dtype=torch.float32)
    R._flash_attn_backward(do.float(), q.float(), k.float(), v.float(), out, lse, dq, dk, dv, p, scale, causal, rng_state=rng)
    for got, ref in ((dq, rq), (dk, rk), (dv, rv)):
        assert (got.double() - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max().item())


def test_dropout_rescale_is_the_reciprocal_of_the_quantised_keep_rate():
    """the library (and this oracle, changed in lockstep) rescale kept probabilities by 256 / keep,
    keep = round((1 - p) 256), NOT by flash_attn's 1 / (1 - p).  Pinned here against numbers written out by hand — not
    against the oracle's own helper — and against the property the choice exists for: E[dropout(P)] = P."""
    by_hand = {0.17: 212, 0.1: 230, 0.5: 128, 0.25: 192, 0.9: 26, 0.05: 243}          # round((1 - p) * 256)
    for p, keep in by_hand.items():
        assert R.drop_threshold(p) == keep
    q, k, v = _rand((1, 64, 2, 32), 11), _rand((1, 64, 2, 32), 12), _rand((1, 64, 2, 32), 13)
    v = torch.ones_like(v)                      # out = row sum of the dropped, rescaled probabilities: E = 1
    p = 0.17
    outs = []
    for seed in range(48):
        out = R._flash_attn_forward(q.float(), k.float(), v.float(), p, 32 ** -0.5, False,
                                    rng_state=torch.tensor([1000 + seed, 0]))[0]
        outs.append(out.double().mean().item())
    mean = sum(outs) / len(outs)
    # 48 x 8192 masked rows of 64 keys: the standard error of the mean is ~4e-4; flash_attn's factor 1 / (1 - p) on this
    # mask (kept at 212 / 256) would give 212 / 256 / 0.83 = 0.99774, i.e. 2.3e-3 low
    assert abs(mean - 1.0) < 1.2e-3, mean
    assert abs(212 / 256 / 0.83 - 1.0) > 2e-3


def test_dropout_mask_definition():
    """known-answer pins of the mask function (so that the C++ and the Python statement cannot drift together
    unnoticed), its position semantics, and its statistics"""
    # fmix32 is MurmurHash3's 32-bit finalizer: published test values
    import numpy as np

    assert [int(x) for x in R._fmix32(np.asarray([0, 1, 0xFFFFFFFF, 0x12345678], dtype=np.uint32))] == \
        [0, 0x514E28B7, 0x81F16F39, 0xE37CD1BC]
    a = R.dropout_keep(7, 0.5, 0, [3], range(10, 20), range(100, 164))
    # a mask is a function of GLOBAL positions: any window of a larger mask equals the mask of that window
    big = R.dropout_keep(7, 0.5, 0, [0, 3, 5], range(0, 32), range(64, 200))
    assert torch.equal(a[0], big[1, 10:20, 36:100])
    # different seed / head / batch -> different bits; same arguments -> same bits
    assert torch.equal(a, R.dropout_keep(7, 0.5, 0, [3], range(10, 20), range(100, 164)))
    for other in (R.dropout_keep(8, 0.5, 0, [3], range(10, 20), range(100, 164)),
                  R.dropout_keep(7, 0.5, 0, [4], range(10, 20), range(100, 164)),
                  R.dropout_keep(7, 0.5, 1, [3], range(10, 20), range(100, 164))):
        assert 0.3 < (other != a).float().mean() < 0.7
    # thresholds: p = 0 keeps everything, p -> 1 keeps (almost) nothing, keep rate = round((1 - p) 256) / 256
    assert R.drop_threshold(0.0) == 256 and R.drop_threshold(0.1) == 230 and R.drop_threshold(0.999) == 0
    m = R.dropout_keep(99, 0.1, 0, range(4), range(512), range(512)).float()
    assert abs(m.mean().item() - 230 / 256) < 2e-3
    # no visible structure along rows, keys or the 4-key word groups
    assert m.mean(dim=(0, 2)).std() < 0.02 and m.mean(dim=(0, 1)).std() < 0.02
    assert abs(m[..., 0::4].mean() - m[..., 3::4].mean()) < 5e-3


@pytest.mark.parametrize("B,Sq,Sk,H,Hk,D,causal", SHAPES)
def test_blocked_fp64_full_reference_is_the_oracle(B, Sq, Sk, H, Hk, D, causal):
    """tests/_fullref.py (the full-tensor fp64 reference of the full-size GPU parity tests, blocked over query rows so
    that it can run a whole kv-head group of the 65536-token configuration on the GPU box's device) against the CPU
    oracle: same outputs, same empty-row conventions, with a block size that cuts the rows unevenly"""
    import _fullref

    q, k, v = _rand((B, Sq, H, D), 11), _rand((B, Sk, Hk, D), 12), _rand((B, Sk, Hk, D), 13)
    do = _rand((B, Sq, H, D), 14)
    scale = D ** -0.5
    out, lse, _, _ = R._flash_attn_forward(q.float(), k.float(), v.float(), 0.0, scale, causal)
    dq, dk, dv = (torch.empty_like(t, dtype=torch.float32) for t in (q, k, v))
    R._flash_attn_backward(do.float(), q.float(), k.float(), v.float(), out, lse, dq, dk, dv, 0.0, scale, causal)
    for b in range(B):
        fo, fl, fdq, fdk, fdv = _fullref.attention_fwd_bwd_fp64(q[b], k[b], v[b], do[b], causal, rows_per_block=37)
        assert torch.equal(torch.isinf(fl), torch.isinf(lse[b]))
        fin = torch.isfinite(fl)
        assert (fl[fin] - lse[b].double()[fin]).abs().max() < 2e-5
        for got, ref in ((out[b], fo), (dq[b], fdq), (dk[b], fdk), (dv[b], fdv)):
            assert (got.double() - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max().item())
