"""GPU parity tests proper (`-m gpu`): the HIP path, called through the C ABI (ctypes ->
librfa_hip.so), against the CPU oracle on the same seeded inputs — on the reference's own fixture
shapes (3824/5/128 zigzag, 3816/5/128 ring, cu_seqlens [0,128,1248,4240] /
[0,120,1248,4232], llama3 D=8) — plus the side kernels.

Stated tolerances: tests/_tol.py (bf16 inputs N(0,1), fp32 accumulation; the oracle computes in fp32 and rounds
out/dq/dk/dv to bf16 like flash_attn) — per comparison max-abs, relative Frobenius norm, mean-abs and the
non-finite pattern; on the fixture shapes additionally flash_attn's own criterion, error vs an exact fp32
computation <= 2 x the error of a naive bf16 implementation.
"""
import os
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF = torch.bfloat16


def _dev():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch.device("cuda:0")


def _check(name, got, ref, atol, rtol=0.0, kind=None):
    """every criterion of tests/_tol.py for the comparison's kind (max-abs, relative Frobenius norm, mean-abs,
    non-finite pattern); comparisons with their own explicit bounds (exact side kernels) keep the max-abs form"""
    import _tol

    if kind is not None:
        return _tol.compare(name, got, ref, kind)
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{name}: {got.shape} vs {ref.shape}"
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(got), fin), f"{name}: non-finite pattern differs"
    diff = (got - ref)[fin].abs().max().item() if fin.any() else 0.0
    lim = atol + rtol * ref[fin].abs().max().item()
    assert diff <= lim, f"{name}: max|err| {diff:.3e} > {lim:.3e}"


def _oracle_dense(q, k, v, do, causal):
    from oracle import flash_attn_ref as O

    scale = q.shape[-1] ** -0.5
    out, lse, _, _ = O._flash_attn_forward(q, k, v, 0.0, scale, causal)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_backward(do, q, k, v, out, lse, dq, dk, dv, 0.0, scale, causal)
    return out, lse, dq, dk, dv


def _oracle_varlen(q, k, v, do, cu_q, cu_k, causal):
    from oracle import flash_attn_ref as O

    scale = q.shape[-1] ** -0.5
    out, lse, _, _ = O._flash_attn_varlen_forward(q, k, v, cu_q, cu_k, 0, 0, 0.0, scale, causal)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_varlen_backward(do, q, k, v, out, lse, dq, dk, dv, cu_q, cu_k, 0, 0, 0.0, scale, causal)
    return out, lse, dq, dk, dv


def _grads_ok(prefix, got, ref):
    for n, g, r in zip(("dq", "dk", "dv"), got, ref):
        _check(f"{prefix}.{n}", g, r, 0, kind="grad")


def test_native_selftest_binary(built):
    """torch-free C-ABI self test: layout probes + 25 parity groups (head dims 8 ... 256) against oracle/attn_ref.c"""
    exe = built.build_selftest()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    try:  # keep the full log where gpurun merges it back (post-mortem of any failure)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "selftest_pytest.log"), "w").write(r.stdout + "\n--- stderr ---\n" + r.stderr)
    except OSError:
        pass
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "SELFTEST PASSED" in r.stdout


@pytest.mark.parametrize("spill", ["1", "0"])
@pytest.mark.parametrize("api,seqlen,causal", [("zigzag", 3824, True), ("ring", 3816, True), ("ring", 1000, False)])
def test_dense_qkvpacked_reference_fixture(single_rank_group, monkeypatch, api, seqlen, causal, spill):
    """reference test/test_{zigzag_,}ring_flash_attn_func.py at world_size 1: B=1, H=5, D=128 bf16.
    spill = 1: the 5-GEMM backward (dS spilled by the dK/dV kernel, dQ by rfa_dqs.hip — the default for dense
    D = 128 calls); spill = 0: the 7-GEMM backward (dQ kernel recomputes S and dP)."""
    import ring_flash_attn as R

    monkeypatch.setenv("RFA_BWD_DS_SPILL", spill)

    dev = _dev()
    g = torch.Generator().manual_seed(42)
    qkv = torch.randn(1, seqlen, 3, 5, 128, generator=g).to(BF)
    do = torch.randn(1, seqlen, 5, 128, generator=g).to(BF)
    fn = R.zigzag_ring_flash_attn_qkvpacked_func if api == "zigzag" else R.ring_flash_attn_qkvpacked_func
    x = qkv.to(dev).requires_grad_(True)
    out, lse, _ = fn(x, dropout_p=0, causal=causal, window_size=(-1, -1), alibi_slopes=None, deterministic=False,
                     return_attn_probs=True)
    out.backward(do.to(dev))
    ro, rl, dq, dk, dv = _oracle_dense(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], do, causal)
    assert out.dtype == BF and lse.dtype == torch.float32 and lse.shape == (1, 5, seqlen) and lse.is_contiguous()
    _check("out", out, ro, 0, kind="out")
    _check("lse", lse, rl, 0, kind="lse")
    _grads_ok(api, (x.grad[:, :, 0], x.grad[:, :, 1], x.grad[:, :, 2]), (dq, dk, dv))
    # flash_attn's own criterion (and SURVEY section 8c): error against an exact fp32 computation at most twice the
    # error of a naive bf16 implementation of the same formula
    import _tol

    exact = _tol.exact_attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], do, causal)
    naive = _tol.naive_lowp_attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], do, causal)
    got = (out, x.grad[:, :, 0], x.grad[:, :, 1], x.grad[:, :, 2])
    for nm, g_, e_, n_ in zip(("out", "dq", "dk", "dv"), got, exact, naive):
        _tol.within_2x_naive(f"{api}{seqlen}.spill{spill}.{nm}", g_, e_, n_)


@pytest.mark.parametrize("spill", ["1", "0"])
@pytest.mark.parametrize("nsplit", [None, "2"])
@pytest.mark.parametrize("api,cu", [("zigzag", [0, 128, 1248, 4240]), ("ring", [0, 120, 1248, 4232])])
def test_varlen_reference_fixture(single_rank_group, monkeypatch, api, cu, nsplit, spill):
    """reference test/test_{zigzag_,}ring_flash_attn_varlen_func.py at world_size 1 (H=5, D=128); nsplit: the
    256-key dK/dV form forced onto this small shape (csrc/rfa_api.cpp bwd_dkdv_plan)."""
    import ring_flash_attn as R

    if nsplit:
        monkeypatch.setenv("RFA_DKDV_NSPLIT", nsplit)
    monkeypatch.setenv("RFA_BWD_DS_SPILL", spill)       # 5-GEMM (dS spill, packed layout) / 7-GEMM backward

    dev = _dev()
    g = torch.Generator().manual_seed(43)
    T = cu[-1]
    qkv = torch.randn(T, 3, 5, 128, generator=g).to(BF)
    do = torch.randn(T, 5, 128, generator=g).to(BF)
    cut = torch.tensor(cu, dtype=torch.int32)
    maxlen = int((cut[1:] - cut[:-1]).max())
    fn = R.zigzag_ring_flash_attn_varlen_qkvpacked_func if api == "zigzag" else R.ring_flash_attn_varlen_qkvpacked_func
    x = qkv.to(dev).requires_grad_(True)
    out, lse, _ = fn(x, cut.to(dev), maxlen, causal=True, return_attn_probs=True)
    out.backward(do.to(dev))
    ro, rl, dq, dk, dv = _oracle_varlen(qkv[:, 0], qkv[:, 1], qkv[:, 2], do, cut, cut, True)
    assert lse.shape == (5, T)
    _check("out", out, ro, 0, kind="out")
    _check("lse", lse, rl, 0, kind="lse")
    _grads_ok(api + "_varlen", (x.grad[:, 0], x.grad[:, 1], x.grad[:, 2]), (dq, dk, dv))


def test_llama3_reference_fixture(single_rank_group):
    """reference test/test_llama3_flash_attn_varlen_func.py at world_size 1: H=5, D=8, stride 1."""
    import ring_flash_attn as R

    dev = _dev()
    cu = torch.tensor([0, 120, 1248, 4232], dtype=torch.int32)
    g = torch.Generator().manual_seed(44)
    qkv = torch.randn(4232, 3, 5, 8, generator=g).to(BF)
    do = torch.randn(4232, 5, 8, generator=g).to(BF)
    cq, ck, mq, mk, sl = R.llama3_flash_attn_prepare_cu_seqlens(cu, causal=True, rank=0, world_size=1)
    x = qkv.to(dev).requires_grad_(True)
    out, lse, _ = R.llama3_flash_attn_varlen_qkvpacked_func(x, cq.to(dev), ck.to(dev), mq, mk, heads_k_stride=1,
                                                            local_k_slice=sl, causal=True, return_attn_probs=True)
    out.backward(do.to(dev))
    ro, rl, dq, dk, dv = _oracle_varlen(qkv[:, 0], qkv[:, 1], qkv[:, 2], do, cu, cu, True)
    _check("out", out, ro, 0, kind="out")
    _check("lse", lse, rl, 0, kind="lse")
    _grads_ok("llama3", (x.grad[:, 0], x.grad[:, 1], x.grad[:, 2]), (dq, dk, dv))


def test_gqa_kvpacked_strided_views_and_fp16(single_rank_group):
    """benchmark layout (kv packed, GQA 8:2) — K/V are strided views, not copied — and fp16."""
    import ring_flash_attn as R

    dev = _dev()
    for dtype in (BF, torch.float16):
        g = torch.Generator().manual_seed(45)
        q = torch.randn(2, 777, 8, 128, generator=g).to(dtype)
        kv = torch.randn(2, 777, 2, 2, 128, generator=g).to(dtype)
        do = torch.randn(2, 777, 8, 128, generator=g).to(dtype)
        qd, kvd = q.to(dev).requires_grad_(True), kv.to(dev).requires_grad_(True)
        out, lse, _ = R.zigzag_ring_flash_attn_kvpacked_func(qd, kvd, causal=True, return_attn_probs=True)
        out.backward(do.to(dev))
        ro, rl, dq, dk, dv = _oracle_dense(q, kv[:, :, 0], kv[:, :, 1], do, True)
        assert out.dtype == dtype
        _check("out", out, ro, 0, kind="out")
        _check("lse", lse, rl, 0, kind="lse")
        _grads_ok(str(dtype), (qd.grad, kvd.grad[:, :, 0], kvd.grad[:, :, 1]), (dq, dk, dv))


def test_merge_kernel_is_reference_update_out_and_lse():
    """update_out_and_lse (HIP merge kernel) == reference utils.py:40-48 formula, incl. slice_."""
    import torch.nn.functional as F
    from ring_flash_attn.utils import update_out_and_lse

    dev = _dev()
    g = torch.Generator().manual_seed(5)
    B, S, H, D = 2, 333, 5, 128
    b0, b1 = torch.randn(B, S, H, D, generator=g).to(BF), torch.randn(B, S, H, D, generator=g).to(BF)
    b2 = torch.randn(B, S - 100, H, D, generator=g).to(BF)
    l0, l1 = torch.randn(B, H, S, generator=g) * 4, torch.randn(B, H, S, generator=g) * 4
    l2 = torch.randn(B, H, S - 100, generator=g) * 4

    def ref_update(out, lse, bo, bl):
        bo = bo.float()
        bl = bl.transpose(-2, -1).unsqueeze(-1)
        return out - torch.sigmoid(bl - lse) * (out - bo), lse - F.logsigmoid(lse - bl)

    ro, rl = b0.float(), l0.transpose(-2, -1).unsqueeze(-1)
    ro, rl = ref_update(ro, rl, b1, l1)
    so, sl_ = ref_update(ro[:, 100:], rl[:, 100:], b2, l2)
    ro = ro.clone(); rl = rl.clone()
    ro[:, 100:], rl[:, 100:] = so, sl_

    out, lse = update_out_and_lse(None, None, b0.to(dev), l0.to(dev))
    assert out.dtype == torch.float32 and lse.shape == (B, S, H, 1)
    out, lse = update_out_and_lse(out, lse, b1.to(dev), l1.to(dev))
    out, lse = update_out_and_lse(out, lse, b2.to(dev), l2.to(dev), slice_=(slice(None), slice(100, None)))
    _check("merge.out", out, ro, 1e-5, 1e-5)
    _check("merge.lse", lse, rl, 1e-5, 1e-5)
    with pytest.raises(RuntimeError, match="first update_out_and_lse"):
        update_out_and_lse(None, None, b0.to(dev), l0.to(dev), slice_=(slice(None),))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_sum_slots_kernel_strided_destination(dtype):
    """rfa_sum_slots: W io-dtype contributions summed in fp32 into a slice of a packed gradient (dense) and into a
    packed-sequence tensor (T,H,D); bit-exact against the same fp32 sum rounded once."""
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be = get_backend()
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    W, B, S, Hk, D = 5, 2, 300, 3, 128
    slots = torch.randn(W, B, S, 2, Hk, D, generator=g).to(dtype).to(dev)
    packed = torch.full((B, S, 2, Hk, D), 7.0, dtype=dtype, device=dev)
    be.sum_slots(slots[:, :, :, 1], packed[:, :, 1])                       # strided source slices, strided destination
    ref = slots[:, :, :, 1].float().sum(0).to(dtype)
    assert torch.equal(packed[:, :, 1], ref) and bool((packed[:, :, 0] == 7.0).all())
    whole = torch.empty_like(packed)
    be.sum_slots(slots.flatten(-3, -2), whole.flatten(-3, -2))             # the packed tensor in one launch
    assert torch.equal(whole, slots.float().sum(0).to(dtype))
    tv = torch.randn(W, 777, 4, 64, generator=g).to(dtype).to(dev)         # (T,H,D), head dim 64
    dst = torch.empty(777, 4, 64, dtype=dtype, device=dev)
    be.sum_slots(tv, dst)
    assert torch.equal(dst, tv.float().sum(0).to(dtype))


def test_lse_flatten_unflatten_bit_exact():
    """reference test/test_triton_kernels.py: re-layout must be bit-identical (cu [0,15,156,529])."""
    from ring_flash_attn.utils import flatten_varlen_lse, unflatten_varlen_lse

    dev = _dev()
    cu = [0, 15, 156, 529]
    cut = torch.tensor(cu, dtype=torch.int32, device=dev)
    maxlen = max(b - a for a, b in zip(cu[:-1], cu[1:]))
    lse = torch.randn(3, 5, maxlen, device=dev)
    flat = flatten_varlen_lse(lse, cut)
    ref = torch.cat([lse[i, :, : cu[i + 1] - cu[i]] for i in range(3)], dim=1)
    assert flat.shape == (5, 529) and torch.equal(flat, ref)
    un = unflatten_varlen_lse(flat.transpose(-2, -1).unsqueeze(-1), cut, maxlen)
    for i in range(3):
        n = cu[i + 1] - cu[i]
        assert torch.equal(un[i, :, :n], lse[i, :, :n])


def test_smoke_entry(single_rank_group):
    import __graft_entry__ as ge

    ge.smoke()


def test_varlen_with_empty_sequences(single_rank_group):
    """packed batch containing zero-length sequences (repeated cu_seqlens entries) and a 1-token one"""
    import ring_flash_attn as R

    dev = _dev()
    cu = [0, 0, 70, 70, 71, 300, 300]
    cut = torch.tensor(cu, dtype=torch.int32)
    g = torch.Generator().manual_seed(46)
    q = torch.randn(300, 4, 128, generator=g).to(BF)
    k = torch.randn(300, 2, 128, generator=g).to(BF)
    v = torch.randn(300, 2, 128, generator=g).to(BF)
    do = torch.randn(300, 4, 128, generator=g).to(BF)
    qd, kd, vd = [t.to(dev).requires_grad_(True) for t in (q, k, v)]
    out, lse, _ = R.ring_flash_attn_varlen_func(qd, kd, vd, cut.to(dev), 229, causal=True, return_attn_probs=True)
    out.backward(do.to(dev))
    ro, rl, dq, dk, dv = _oracle_varlen(q, k, v, do, cut, cut, True)
    _check("out", out, ro, 0, kind="out")
    _check("lse", lse, rl, 0, kind="lse")
    _grads_ok("empty-seqs", (qd.grad, kd.grad, vd.grad), (dq, dk, dv))


def test_empty_problem_is_noop(single_rank_group):
    import ring_flash_attn as R

    dev = _dev()
    q = torch.empty(1, 0, 4, 128, device=dev, dtype=BF)
    out, lse, _ = R.ring_flash_attn_func(q, q, q, causal=True, return_attn_probs=True)
    assert out.shape == (1, 0, 4, 128) and lse.shape == (1, 4, 0)


@pytest.mark.parametrize("Sq,Sk,causal,B,H,Hk", [
    (700, 700, True, 2, 4, 2),        # ragged tails on both axes, GQA, batch
    (512, 1024, False, 1, 2, 2),      # ring "front" step shape: all queries x more keys, unmasked
    (1000, 488, True, 1, 4, 1),       # more queries than keys: bottom-right alignment leaves rows without keys
    (96, 4000, True, 1, 2, 2),        # one query block against 63 key tiles (deep dS ring)
])
@pytest.mark.extended
def test_ds_spill_backward_matches_oracle_and_recompute(Sq, Sk, causal, B, H, Hk):
    """The dS-spill backward through the backend (accumulate and plain outputs) against the CPU oracle and
    against the recompute backward on the same inputs (they share dK/dV bit for bit: same kernel, the spill
    only adds stores)."""
    from oracle import flash_attn_ref as O
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be = get_backend()
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    q = torch.randn(B, Sq, H, 128, generator=g).to(BF)
    k = torch.randn(B, Sk, Hk, 128, generator=g).to(BF)
    v = torch.randn(B, Sk, Hk, 128, generator=g).to(BF)
    do = torch.randn(B, Sq, H, 128, generator=g).to(BF)
    scale = 128 ** -0.5
    ro, rl, _, _ = O._flash_attn_forward(q, k, v, 0.0, scale, causal)
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, 0.0, scale, causal)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    out = torch.empty_like(qd)
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=dev)
    be.fwd(qd, kd, vd, softmax_scale=scale, causal=causal, out=out, lse=lse)
    delta = torch.empty_like(lse)
    be.bwd_preprocess(dod, out, delta)
    lse_b = torch.where(torch.isinf(lse), torch.zeros_like(lse), lse)       # rows without keys: P = 0 either way
    res = {}
    from ring_flash_attn import config

    for spill in (True, False):
        config.set(bwd_ds_spill=spill)
        dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq=dq, dk=dk, dv=dv)
        dqa = torch.full((B, Sq, H, 128), 3.0, dtype=torch.float32, device=dev)
        dka = torch.zeros((B, Sk, Hk, 128), dtype=torch.float32, device=dev)
        dva = torch.zeros_like(dka)
        be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq_acc=dqa, dk_acc=dka, dv_acc=dva)
        res[spill] = (dq, dk, dv, dqa)
    os.environ.pop("RFA_BWD_DS_SPILL", None)
    del lse_b
    for spill in (True, False):
        dq, dk, dv, dqa = res[spill]
        _grads_ok(f"spill={spill}", (dq, dk, dv), (rdq, rdk, rdv))
        _check(f"spill={spill}.dq_acc", dqa - 3.0, rdq, 0, kind="grad")
    assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][2], res[False][2])
    _check("dq spill vs recompute", res[True][0], res[False][0].float(), 0, kind="grad")


@pytest.mark.parametrize("B,Sq,Sk,H,Hk,causal,cu", [
    (1, 1024, 1024, 8, 2, True, None),          # GQA 4:1, dense causal (triangular scratch rows)
    (2, 640, 896, 6, 6, False, None),           # MHA, batch, rectangular rows, ragged tiles
    (3, 1100, 1100, 8, 4, True, [0, 300, 1400, 2100]),   # packed sequences
])
@pytest.mark.extended
def test_ds_handoff_in_head_group_chunks(B, Sq, Sk, H, Hk, causal, cu):
    """ABI 5: a dS scratch SMALLER than the whole hand-off does not switch the 5-GEMM backward off — the call runs it in
    head-group chunks over the one buffer (include/rfa.h: ds_scratch_bytes): chunks of whole K/V heads, then fractions
    of ONE K/V head's query heads whose dK/dV shares are accumulated in the fp32 partials.  Every chunking of a call
    must reproduce the oracle — plain outputs and `+=` accumulators — and the 7-GEMM fall-back below one query head."""
    import ctypes as C

    from ring_flash_attn import _C, config
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be = get_backend()
    dev = _dev()
    g = torch.Generator().manual_seed(B * Sq + H)
    D, scale = 128, 128 ** -0.5
    if cu is None:
        q, k, v, do = (torch.randn(B, s_, h_, D, generator=g).to(BF) for s_, h_ in ((Sq, H), (Sk, Hk), (Sk, Hk), (Sq, H)))
        ro, rl, rdq, rdk, rdv = _oracle_dense(q, k, v, do, causal)
        vl, pre = {}, {}
        lse_shape = (B, H, Sq)
    else:
        T = cu[-1]
        q, k, v, do = (torch.randn(T, h_, D, generator=g).to(BF) for h_ in (H, Hk, Hk, H))
        cut = torch.tensor(cu, dtype=torch.int32)
        ro, rl, rdq, rdk, rdv = _oracle_varlen(q, k, v, do, cut, cut, causal)
        mx = int((cut[1:] - cut[:-1]).max())
        vl = dict(cu_seqlens_q=cut.to(dev), cu_seqlens_k=cut.to(dev), max_seqlen_q=mx, max_seqlen_k=mx)
        pre = dict(cu_seqlens_q=cut.to(dev), max_seqlen_q=mx)
        lse_shape = (H, T)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    out, lse = torch.empty_like(qd), torch.empty(lse_shape, dtype=torch.float32, device=dev)
    be.fwd(qd, kd, vd, softmax_scale=scale, causal=causal, out=out, lse=lse, **vl)
    delta = torch.empty_like(lse)
    be.bwd_preprocess(dod, out, delta, **pre)

    def run(limit):
        be.release_scratch()
        with config.override(ds_spill_max_bytes=limit):
            dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
            be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq=dq, dk=dk, dv=dv, **vl)
            dqa = torch.full(qd.shape, 2.0, dtype=torch.float32, device=dev)
            dka = torch.full(kd.shape, -1.0, dtype=torch.float32, device=dev)
            dva = torch.full(kd.shape, 0.5, dtype=torch.float32, device=dev)
            be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq_acc=dqa, dk_acc=dka, dv_acc=dva, **vl)
            pool = list(be._ds_pool.values())
        be.release_scratch()
        return (dq, dk, dv, dqa - 2.0, dka + 1.0, dva - 0.5), (pool[0].numel() if pool else 0)

    # what the call asks for: the whole hand-off, one query head
    a = _C.BwdArgs()
    a.B, a.H, a.Hk, a.D, a.dtype, a.causal = (len(cu) - 1 if cu else B), H, Hk, D, 0, int(causal)
    a.Sq, a.Sk = (mx, mx) if cu else (Sq, Sk)
    a.total_k = kd.shape[0] if cu else B * Sk
    if cu:
        a.cu_seqlens_q = a.cu_seqlens_k = 1
    full, per = be.lib.rfa_bwd_ds_scratch_bytes(C.byref(a)), be.lib.rfa_bwd_ds_scratch_min_bytes(C.byref(a))
    assert full == H * per
    G = H // Hk
    limits = {"all heads": full, "whole K/V heads": per * G, "one query head": per, "below one head (7-GEMM)": per - 1}
    if G > 2:
        limits["half a group"] = per * (G // 2)
    for name, limit in limits.items():
        got, used = run(limit)
        assert used == (0 if limit < per else limit), (name, used, limit)       # the ONE scratch is never larger than the limit
        for nm, g_, r_ in zip(("dq", "dk", "dv", "dq_acc", "dk_acc", "dv_acc"), got, (rdq, rdk, rdv) * 2):
            _check(f"{name}.{nm}", g_, r_, 0, kind="grad")


@pytest.mark.parametrize("nsplit", ["1", "2", "3", "4"])
@pytest.mark.parametrize("Sq,Sk,causal,B,H,Hk", [
    (700, 700, True, 2, 4, 2),        # ragged tails, GQA, batch: key blocks of 256 with a 188-key tail
    (1000, 488, True, 1, 4, 1),       # more queries than keys; splits that receive no tile at all store zeros
    (512, 1024, False, 1, 2, 2),      # ring "front" step shape
    (96, 300, True, 1, 2, 2),         # fewer tiles (2) than splits
])
@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.extended
def test_dkdv_256_key_form_with_query_range_splits(monkeypatch, nsplit, Sq, Sk, causal, B, H, Hk, D):
    """The 256-key dK/dV kernel form (csrc/rfa_bwd.hip kWide; head dim 128 and — round 5 — 64), forced onto small shapes
    with every split count (RFA_DKDV_NSPLIT; production picks it from the shapes): plain io outputs, fp32 accumulate (+=),
    fp32 overwrite slots, two-phase COMPUTE / REDUCE — with and without the dS spill (head dim 64 has no hand-off: it
    runs its 7-GEMM form) — against the CPU oracle, and against the 128-key form on the same inputs."""
    from oracle import flash_attn_ref as O
    from ring_flash_attn import _C
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be = get_backend()
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, Sq, H, D, generator=g).to(BF)
    k = torch.randn(B, Sk, Hk, D, generator=g).to(BF)
    v = torch.randn(B, Sk, Hk, D, generator=g).to(BF)
    do = torch.randn(B, Sq, H, D, generator=g).to(BF)
    scale = D ** -0.5
    ro, rl, _, _ = O._flash_attn_forward(q, k, v, 0.0, scale, causal)
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, 0.0, scale, causal)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    out = torch.empty_like(qd)
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=dev)
    be.fwd(qd, kd, vd, softmax_scale=scale, causal=causal, out=out, lse=lse)
    delta = torch.empty_like(lse)
    be.bwd_preprocess(dod, out, delta)
    kw = dict(softmax_scale=scale, causal=causal)

    def run_all():
        res = {}
        dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        be.bwd(dod, qd, kd, vd, lse, delta, dq=dq, dk=dk, dv=dv, **kw)
        res["plain"] = (dq, dk, dv)
        dqa = torch.zeros((B, Sq, H, D), dtype=torch.float32, device=dev)
        dka = torch.full((B, Sk, Hk, D), 2.0, dtype=torch.float32, device=dev)
        dva = torch.full_like(dka, -1.0)
        be.bwd(dod, qd, kd, vd, lse, delta, dq_acc=dqa, dk_acc=dka, dv_acc=dva, **kw)          # += (workspace)
        res["acc"] = (dqa, dka - 2.0, dva + 1.0)
        dqa = torch.zeros_like(dqa)
        dka = torch.full_like(dka, 5.0)
        dva = torch.full_like(dka, 5.0)
        be.bwd(dod, qd, kd, vd, lse, delta, dq_acc=dqa, dk_acc=dka, dv_acc=dva, acc_init=True,
               phases=_C.BWD_KV_OVERWRITE, **kw)                                                 # fp32 slots
        res["overwrite"] = (dqa, dka, dva)
        dqa = torch.zeros_like(dqa)
        dka = torch.full_like(dka, 1.0)
        dva = torch.full_like(dka, 1.0)
        part = be.bwd(dod, qd, kd, vd, lse, delta, dq_acc=dqa, dk_acc=dka, dv_acc=dva, phases=_C.BWD_COMPUTE, **kw)
        be.bwd(dod, qd, kd, vd, lse, delta, dq_acc=dqa, dk_acc=dka, dv_acc=dva, phases=_C.BWD_REDUCE,
               partials=part, **kw)
        res["two_phase"] = (dqa, dka - 1.0, dva - 1.0)
        return res

    monkeypatch.setenv("RFA_DKDV_WIDE", "0")
    monkeypatch.delenv("RFA_DKDV_NSPLIT", raising=False)
    narrow = run_all()
    monkeypatch.setenv("RFA_DKDV_WIDE", "1")
    monkeypatch.setenv("RFA_DKDV_NSPLIT", nsplit)
    for spill in ("1", "0") if D > 64 else ("1",):
        monkeypatch.setenv("RFA_BWD_DS_SPILL", spill)
        wide = run_all()
        for mode, got in wide.items():
            _grads_ok(f"nsplit={nsplit} spill={spill} {mode}", got, (rdq, rdk, rdv))
            for n, a_, b_ in zip(("dk", "dv"), got[1:], narrow[mode][1:]):      # same math, other summation order
                _check(f"nsplit={nsplit} spill={spill} {mode}.{n} vs 128-key form", a_, b_.float(), 0, kind="grad")


@pytest.mark.parametrize("S,B,H,Hk,dtype", [
    (512, 2, 4, 2, torch.bfloat16),       # ONE pair of key blocks per (batch, K/V head): A_0 / B_0 only
    (1024, 1, 4, 1, torch.bfloat16),      # nkb = 4; one K/V head: the two workgroups of a pair sit on DIFFERENT XCDs (agent-scope hand-off)
    (1536, 1, 4, 2, torch.float16),       # nkb = 6 (odd half), fp16
    (2048, 2, 8, 2, torch.bfloat16),      # nkb = 8, G = 4, batch
    (1024, 2, 4, 2, 64),                  # head dim 64 (bf16): the 7-GEMM backward's dK/dV kernel in the same schedule
])
def test_dkdv_balanced_causal_schedule(monkeypatch, S, B, H, Hk, dtype):
    """Round 6: the balanced causal schedule of the 256-key dK/dV form (csrc/rfa_bwd.hip kBal, RFA_DKDV_BAL) — every workgroup
    of a dense causal self-attention block walks the same number of Q/dO tiles, a key block of the lower half is shared by
    two workgroups whose fp32 partials meet in the later one through an agent-scope flag — forced onto small shapes (production
    picks it for launches that fill the chip): plain io outputs and overwritten fp32 accumulators run it, with and without
    the dS hand-off; `+=` accumulators and two-phase calls are not eligible and must run what they ran before.  Against the
    CPU oracle, against the 128-key form (same math, other summation order), run to run (bit-identical: a + b == b + a), and
    with dQ bit-identical to the shared-range plan (the dS blocks do not depend on who computed them)."""
    from oracle import flash_attn_ref as O
    from ring_flash_attn import _C, config
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be = get_backend()
    dev = _dev()
    D = 128
    if dtype == 64:
        D, dtype = 64, torch.bfloat16
    g = torch.Generator().manual_seed(S + H)
    q = torch.randn(B, S, H, D, generator=g).to(dtype)
    k = torch.randn(B, S, Hk, D, generator=g).to(dtype)
    v = torch.randn(B, S, Hk, D, generator=g).to(dtype)
    do = torch.randn(B, S, H, D, generator=g).to(dtype)
    scale = D ** -0.5
    ro, rl, _, _ = O._flash_attn_forward(q, k, v, 0.0, scale, True)
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, 0.0, scale, True)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    out = torch.empty_like(qd)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    be.fwd(qd, kd, vd, softmax_scale=scale, causal=True, out=out, lse=lse)
    delta = torch.empty_like(lse)
    be.bwd_preprocess(dod, out, delta)
    kw = dict(softmax_scale=scale, causal=True)

    def plan_of(**extra):
        a = _C.BwdArgs()
        a.B, a.Sq, a.Sk, a.H, a.Hk, a.D, a.dtype, a.causal, a.total_k = B, S, S, H, Hk, D, 0, 1, B * S
        a.dkdv_form = _C.DKDV_BAL
        for n_, v_ in extra.items():
            setattr(a, n_, v_)
        return be.bwd_plan(a)[0]

    assert plan_of() == _C.DKDV_BAL and plan_of(dk_acc=1, dv_acc=1, acc_init=1) == _C.DKDV_BAL
    assert plan_of(dk_acc=1, dv_acc=1) != _C.DKDV_BAL and plan_of(phases=_C.BWD_COMPUTE) != _C.DKDV_BAL

    def run_all():
        res = {}
        dq, dk, dv = (torch.full_like(t, float("nan")) for t in (qd, kd, vd))
        be.bwd(dod, qd, kd, vd, lse, delta, dq=dq, dk=dk, dv=dv, **kw)
        res["plain"] = (dq, dk, dv)
        dqa = torch.zeros((B, S, H, D), dtype=torch.float32, device=dev)
        dka = torch.full((B, S, Hk, D), 5.0, dtype=torch.float32, device=dev)
        dva = torch.full_like(dka, 5.0)
        be.bwd(dod, qd, kd, vd, lse, delta, dq_acc=dqa, dk_acc=dka, dv_acc=dva, acc_init=True,
               phases=_C.BWD_KV_OVERWRITE, **kw)                                                 # fp32 slots, overwritten
        res["overwrite"] = (dqa, dka, dva)
        dqa = torch.zeros_like(dqa)
        dka = torch.full_like(dka, 2.0)
        dva = torch.full_like(dka, -1.0)
        be.bwd(dod, qd, kd, vd, lse, delta, dq_acc=dqa, dk_acc=dka, dv_acc=dva, **kw)          # += : not eligible
        res["acc"] = (dqa, dka - 2.0, dva + 1.0)
        dqa = torch.zeros_like(dqa)
        dka = torch.full_like(dka, 1.0)
        dva = torch.full_like(dka, 1.0)
        part = be.bwd(dod, qd, kd, vd, lse, delta, dq_acc=dqa, dk_acc=dka, dv_acc=dva, phases=_C.BWD_COMPUTE, **kw)
        be.bwd(dod, qd, kd, vd, lse, delta, dq_acc=dqa, dk_acc=dka, dv_acc=dva, phases=_C.BWD_REDUCE, partials=part, **kw)
        res["two_phase"] = (dqa, dka - 1.0, dva - 1.0)
        return res

    with config.override(dkdv_wide=0):
        narrow = run_all()
    with config.override(dkdv_wide=1, dkdv_nsplit=2):
        shared = run_all()
    for spill in (True, False) if D == 128 else (True,):
        with config.override(dkdv_wide=2, bwd_ds_spill=spill):
            bal = run_all()
            again = run_all()
        for mode, got in bal.items():
            _grads_ok(f"S={S} spill={spill} {mode}", got, (rdq, rdk, rdv))
            for n, a_, b_ in zip(("dk", "dv"), got[1:], narrow[mode][1:]):
                _check(f"S={S} spill={spill} {mode}.{n} vs 128-key form", a_, b_.float(), 0, kind="grad")
            for n, a_, b_ in zip(("dq", "dk", "dv"), got, again[mode]):
                assert torch.equal(a_, b_), f"S={S} spill={spill} {mode}.{n}: not bit-identical run to run"
        if spill:
            for mode in ("plain", "overwrite"):
                assert torch.equal(bal[mode][0], shared[mode][0]), f"S={S} {mode}: dq differs from the shared-range plan"


@pytest.mark.extended
def test_ds_scratch_capped_by_free_memory_is_kept(monkeypatch):
    """a dS scratch that the free-memory fraction CAPPED when it was taken is smaller than the hand-off for ever;
    it must be reused as it is (the hand-off then runs in head-group chunks over it) instead of being dropped, re-queried
    and re-made in every backward.  The allocator queries are replaced by fixed answers so that the cap is deterministic:
    34 MB of dS (8 heads, S = 2048, causal), 20 MB granted."""
    from ring_flash_attn import backend as BK
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be, dev = get_backend(), _dev()
    be.release_scratch()
    g = torch.Generator().manual_seed(41)
    B, S, H, Hk, D = 1, 2048, 8, 2, 128
    q, do = (torch.randn(B, S, H, D, generator=g).to(BF) for _ in range(2))
    k, v = (torch.randn(B, S, Hk, D, generator=g).to(BF) for _ in range(2))
    ro, rl, rdq, rdk, rdv = _oracle_dense(q, k, v, do, True)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    out, lse = torch.empty_like(qd), torch.empty((B, H, S), dtype=torch.float32, device=dev)
    be.fwd(qd, kd, vd, softmax_scale=D ** -0.5, causal=True, out=out, lse=lse)
    delta = torch.empty_like(lse)
    be.bwd_preprocess(dod, out, delta)
    queries = []
    monkeypatch.setattr(BK, "_SPILL_CHECK_ABOVE", 0)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a: (queries.append(1), (40 << 20, 288 << 30))[1])
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *a: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda *a: 0)
    bufs = []
    for it in range(4):
        dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=D ** -0.5, causal=True, dq=dq, dk=dk, dv=dv)
        _grads_ok(f"capped scratch, backward {it}", (dq, dk, dv), (rdq, rdk, rdv))
        (buf,) = be._ds_pool.values()
        bufs.append(buf)
    assert all(b_ is bufs[0] for b_ in bufs) and bufs[0].numel() == 20 << 20      # one buffer, 0.5 x the 40 MB "free"
    assert len(queries) == 1                                                        # the allocator was asked once, not per backward
    (left,) = be._ds_capped.values()
    assert left[0] == BK._SPILL_GROW_RETRY - 3
    be.release_scratch()
    assert not be._ds_pool and not be._ds_capped


@pytest.mark.extended
@pytest.mark.parametrize("causal", [True, False])
def test_ds_spill_packed_sequences_with_longer_keys(monkeypatch, causal):
    """dS-spill backward on packed sequences whose K/V are longer than Q (the llama3 shape: local queries against
    gathered keys, bottom-right aligned), with empty and 1-token sequences, against the oracle and against the
    recompute backward (dK/dV bit for bit)."""
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be = get_backend()
    dev = _dev()
    cu_q = torch.tensor([0, 100, 100, 101, 700, 1000], dtype=torch.int32)
    cu_k = torch.tensor([0, 300, 300, 333, 1500, 1800], dtype=torch.int32)
    Tq, Tk, H, Hk = 1000, 1800, 4, 2
    g = torch.Generator().manual_seed(21)
    q = torch.randn(Tq, H, 128, generator=g).to(BF)
    k = torch.randn(Tk, Hk, 128, generator=g).to(BF)
    v = torch.randn(Tk, Hk, 128, generator=g).to(BF)
    do = torch.randn(Tq, H, 128, generator=g).to(BF)
    ro, rl, rdq, rdk, rdv = _oracle_varlen(q, k, v, do, cu_q, cu_k, causal)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    kw = dict(softmax_scale=128 ** -0.5, causal=causal, cu_seqlens_q=cu_q.to(dev), cu_seqlens_k=cu_k.to(dev),
              max_seqlen_q=599, max_seqlen_k=1167)
    out = torch.empty_like(qd)
    lse = torch.empty((H, Tq), dtype=torch.float32, device=dev)
    be.fwd(qd, kd, vd, out=out, lse=lse, **kw)
    delta = torch.empty_like(lse)
    be.bwd_preprocess(dod, out, delta, cu_seqlens_q=kw["cu_seqlens_q"], max_seqlen_q=599)
    res = {}
    for spill in ("1", "0"):
        monkeypatch.setenv("RFA_BWD_DS_SPILL", spill)
        dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        be.bwd(dod, qd, kd, vd, lse, delta, dq=dq, dk=dk, dv=dv, **kw)
        _grads_ok(f"packed spill={spill}", (dq, dk, dv), (rdq, rdk, rdv))
        res[spill] = (dq, dk, dv)
    assert torch.equal(res["1"][1], res["0"][1]) and torch.equal(res["1"][2], res["0"][2])
    _check("dq spill vs recompute", res["1"][0], res["0"][0].float(), 0, kind="grad")


def test_torch_compile_fullgraph_on_gpu(single_rank_group):
    """the custom operators rfa::attn_fwd / rfa::attn_bwd run the HIP kernels under torch.compile(fullgraph=True)
    (no graph break) and reproduce the eager path bit for bit (test/test.sh:23-25 of the reference)."""
    import ring_flash_attn as R
    from ring_flash_attn import backend
    from ring_flash_attn import _testing

    _testing.set_backend(None)
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(1, 640, 3, 4, 128, generator=g).to(BF).to(dev)
    do = torch.randn(1, 640, 4, 128, generator=g).to(BF).to(dev)
    torch._dynamo.reset()
    xe = qkv.clone().requires_grad_(True)
    oe, le, _ = R.zigzag_ring_flash_attn_qkvpacked_func(xe, causal=True, return_attn_probs=True)
    oe.backward(do)
    xc = qkv.clone().requires_grad_(True)
    oc, lc, _ = torch.compile(R.zigzag_ring_flash_attn_qkvpacked_func, fullgraph=True)(xc, causal=True, return_attn_probs=True)
    oc.backward(do)
    assert torch.equal(oc, oe) and torch.equal(lc, le) and torch.equal(xc.grad, xe.grad)
    torch._dynamo.reset()


@pytest.mark.parametrize("shape", ["dense_5gemm", "dense_small_batch", "varlen", "dense_balanced"])
def test_step_under_hip_graph_capture(single_rank_group, shape):
    """a forward + backward of the single-rank step CAPTURED into a HIP graph
    (torch.cuda.graph: allocations from the graph's pool, every launch on the capturing stream) and replayed — the
    answer for launch-bound inner loops (B x S of a few thousand tokens: five 10-100 us kernels per step).  The library
    is capture-safe by construction: it launches only on the stream it is given, never synchronises, and its reusable
    dS scratch / workspaces are ordinary torch allocations of the capturing stream.  Replays on NEW data must reproduce
    the eager result bit for bit — the 5-GEMM backward with its dS hand-off, the split dK/dV plan with its reduction pass,
    packed sequences."""
    import ring_flash_attn as R

    dev = _dev()
    g = torch.Generator().manual_seed(77)
    if shape == "varlen":
        cu = torch.tensor([0, 300, 1324, 2048], dtype=torch.int32, device=dev)
        mk = lambda: (torch.randn(2048, 8, 128, generator=g).to(BF).to(dev), torch.randn(2048, 2, 2, 128, generator=g).to(BF).to(dev),
                      torch.randn(2048, 8, 128, generator=g).to(BF).to(dev))
        fn = lambda q, kv: R.zigzag_ring_flash_attn_varlen_kvpacked_func(q, kv, cu, 1024, causal=True)
    else:
        # dense_balanced (round 6): 4 x 2048 x 32 / 8 heads = 256 key-block workgroups — the balanced causal dK/dV schedule, whose
        # pair flags are zeroed by a memset NODE in front of the kernel and whose partials meet through agent-scope flags:
        # every replay must find the flags zero and reproduce the eager bits
        B, S, H, Hk = {"dense_5gemm": (1, 2048, 8, 2), "dense_small_batch": (8, 512, 8, 2), "dense_balanced": (4, 2048, 32, 8)}[shape]
        mk = lambda: (torch.randn(B, S, H, 128, generator=g).to(BF).to(dev), torch.randn(B, S, 2, Hk, 128, generator=g).to(BF).to(dev),
                      torch.randn(B, S, H, 128, generator=g).to(BF).to(dev))
        fn = lambda q, kv: R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True)
        if shape == "dense_balanced":
            from ring_flash_attn import _C
            from ring_flash_attn.backend import get_backend
            a = _C.BwdArgs()
            a.B, a.Sq, a.Sk, a.H, a.Hk, a.D, a.dtype, a.causal, a.total_k = B, S, S, H, Hk, 128, 0, 1, B * S
            assert get_backend().bwd_plan(a)[0] == _C.DKDV_BAL
    q0, kv0, do0 = mk()
    sq, skv, sdo = q0.clone().requires_grad_(True), kv0.clone().requires_grad_(True), do0.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up off the default stream, as torch's capture recipe asks
        for _ in range(3):
            sq.grad = skv.grad = None
            fn(sq, skv).backward(sdo)
    torch.cuda.current_stream().wait_stream(side)
    sq.grad = skv.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sout = fn(sq, skv)
        sout.backward(sdo)
    for trial in range(2):                             # replays on fresh inputs
        q1, kv1, do1 = mk()
        with torch.no_grad():
            sq.copy_(q1); skv.copy_(kv1); sdo.copy_(do1)
        graph.replay()
        torch.cuda.synchronize()
        eq, ekv = q1.clone().requires_grad_(True), kv1.clone().requires_grad_(True)
        eout = fn(eq, ekv)
        eout.backward(do1)
        assert torch.equal(sout, eout), f"{shape}: out differs on replay {trial}"
        assert torch.equal(sq.grad, eq.grad) and torch.equal(skv.grad, ekv.grad), f"{shape}: gradients differ on replay {trial}"


@pytest.mark.parametrize("Sq,Sk,D,causal,window", [
    (1000, 1000, 128, True, (200, 0)),      # causal sliding window (Mistral / Qwen2 style), multi-tile band
    (1000, 1000, 128, False, (130, 70)),    # two-sided local attention
    (333, 900, 128, True, (64, -1)),        # more keys than queries (bottom-right aligned band)
    (900, 333, 128, False, (-1, 50)),       # right-bounded only, rows without any visible key
    (640, 640, 64, True, (100, 0)),         # head dim 64 instances
    (512, 512, 128, False, (0, 0)),         # one-key band (diagonal)
])
@pytest.mark.extended
def test_sliding_window_kernels_match_oracle(Sq, Sk, D, causal, window):
    """window_left / window_right of rfa_fwd / rfa_bwd (flash_attn semantics; forwarded by the reference's llama3
    path and HF adapter) against the CPU oracle: forward, backward (plain and fp32-accumulate outputs)."""
    from oracle import flash_attn_ref as O
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be = get_backend()
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    B, H, Hk = 2, 4, 2
    q = torch.randn(B, Sq, H, D, generator=g).to(BF)
    k = torch.randn(B, Sk, Hk, D, generator=g).to(BF)
    v = torch.randn(B, Sk, Hk, D, generator=g).to(BF)
    do = torch.randn(B, Sq, H, D, generator=g).to(BF)
    scale = D ** -0.5
    ro, rl, _, _ = O._flash_attn_forward(q, k, v, 0.0, scale, causal, window[0], window[1])
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, 0.0, scale, causal, window[0], window[1])
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    out = torch.empty_like(qd)
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=dev)
    be.fwd(qd, kd, vd, softmax_scale=scale, causal=causal, out=out, lse=lse, window=window)
    _check("out", out, ro, 0, kind="out")
    _check("lse", lse, rl, 0, kind="lse")
    delta = torch.empty_like(lse)
    be.bwd_preprocess(dod, out, delta)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq=dq, dk=dk, dv=dv, window=window)
    _grads_ok("window", (dq, dk, dv), (rdq, rdk, rdv))


@pytest.mark.extended
def test_sliding_window_varlen_and_llama3_single_rank(single_rank_group):
    """packed sequences + window through the public API (llama3 entry point, world size 1)"""
    import ring_flash_attn as R
    from oracle import flash_attn_ref as O

    dev = _dev()
    cu = torch.tensor([0, 300, 301, 1100, 1500], dtype=torch.int32)
    g = torch.Generator().manual_seed(12)
    T, H, Hk, D = 1500, 4, 2, 128
    q, k, v, do = (torch.randn(T, h, D, generator=g).to(BF) for h in (H, Hk, Hk, H))
    win = (96, 0)
    ro, rl, _, _ = O._flash_attn_varlen_forward(q, k, v, cu, cu, 0, 0, 0.0, D ** -0.5, True, win[0], win[1])
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_varlen_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, cu, cu, 0, 0, 0.0, D ** -0.5, True, win[0], win[1])
    cq, ck, mq, mk, sl = R.llama3_flash_attn_prepare_cu_seqlens(cu, causal=True, rank=0, world_size=1)
    qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    out, lse, _ = R.llama3_flash_attn_varlen_func(qd, kd, vd, cq.to(dev), ck.to(dev), mq, mk, heads_k_stride=1,
                                                  local_k_slice=sl, causal=True, window_size=win, return_attn_probs=True)
    out.backward(do.to(dev))
    _check("out", out, ro, 0, kind="out")
    _check("lse", lse, rl, 0, kind="lse")
    _grads_ok("llama3 window", (qd.grad, kd.grad, vd.grad), (rdq, rdk, rdv))


# ------------------------------------------------------------------------------------------------------------
# dropout (include/rfa.h: rfa_fwd_args.dropout_p): the kernels' mask against the oracle's restatement of it
@pytest.mark.parametrize("D,H,Hk,Sq,Sk,causal,dtype", [
    (128, 4, 2, 777, 777, True, BF),          # GQA, odd length (masked tails)
    (128, 2, 2, 300, 555, True, BF),          # bottom-right aligned
    (64, 4, 1, 512, 512, False, BF),          # head-dim-64 instances, no mask
    (96, 2, 2, 260, 260, True, torch.float16),   # padded head dim (register staging), fp16
])
@pytest.mark.extended
def test_dropout_dense_matches_oracle(D, H, Hk, Sq, Sk, causal, dtype):
    from oracle import flash_attn_ref as O
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be, dev = get_backend(), _dev()
    g = torch.Generator().manual_seed(D + Sq)
    B, p, seed = 2, 0.15, 0x0123_4567_89AB_CDEF
    q, k, v = (torch.randn(B, s_, h_, D, generator=g).to(dtype) for s_, h_ in ((Sq, H), (Sk, Hk), (Sk, Hk)))
    do = torch.randn(B, Sq, H, D, generator=g).to(dtype)
    scale = D ** -0.5
    rng = torch.tensor([seed, 0])
    ro, rl, _, _ = O._flash_attn_forward(q, k, v, p, scale, causal, rng_state=rng)
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, p, scale, causal, rng_state=rng)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    out, lse = torch.empty_like(qd), torch.empty((B, H, Sq), dtype=torch.float32, device=dev)
    drop = (p, seed, 0, 0, 0)
    be.fwd(qd, kd, vd, softmax_scale=scale, causal=causal, out=out, lse=lse, dropout=drop)
    _check("drop.out", out, ro, 0, kind="out")
    _check("drop.lse", lse, rl, 0, kind="lse")
    plain = torch.empty_like(qd)
    be.fwd(qd, kd, vd, softmax_scale=scale, causal=causal, out=plain, lse=torch.empty_like(lse))
    assert (plain.float() - out.float()).abs().max().item() > 0.05          # the mask did something
    delta = torch.empty_like(lse)
    be.bwd_preprocess(dod, out, delta)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq=dq, dk=dk, dv=dv, dropout=drop)
    _grads_ok("drop", (dq, dk, dv), (rdq, rdk, rdv))
    # positions are global: rows [a, b) of the queries with q_pos_offset = a give rows [a, b) of the full result
    if not causal:
        a = 129
        part = torch.empty_like(qd[:, a:])
        be.fwd(qd[:, a:], kd, vd, softmax_scale=scale, causal=False, out=part, lse=torch.empty((B, H, Sq - a), dtype=torch.float32, device=dev),
               dropout=(p, seed, a, 0, 0))
        # (same mask bits; the arithmetic differs by a rounding: rows share their wave's rescale decisions with other rows)
        d = (part.float() - out[:, a:].float()).abs().max().item()
        assert d <= 1.6e-2 * out.float().abs().max().item(), f"offset call differs from the full call by {d:.3e}"


@pytest.mark.extended
@pytest.mark.parametrize("cu", [[0, 128, 1248, 2001], [0, 3, 70, 71, 600]])
def test_dropout_varlen_public_api_matches_oracle(single_rank_group, cu):
    """packed sequences starting at positions that are not multiples of 4 (the mask words then straddle the lanes' key
    groups), through the public function (seed drawn from torch's generator) — forward and backward"""
    import ring_flash_attn as R
    from ring_flash_attn._common import draw_dropout_seed
    from oracle import flash_attn_ref as O

    dev = _dev()
    g = torch.Generator().manual_seed(len(cu))
    T, H, Hk, D, p = cu[-1], 4, 2, 128, 0.1
    q, k, v = (torch.randn(T, h_, D, generator=g).to(BF) for h_ in (H, Hk, Hk))
    do = torch.randn(T, H, D, generator=g).to(BF)
    cu_t = torch.tensor(cu, dtype=torch.int32)
    mx = max(b - a for a, b in zip(cu[:-1], cu[1:]))
    torch.manual_seed(77)
    rng = torch.tensor([draw_dropout_seed(), 0])
    ro, rl, _, _ = O._flash_attn_varlen_forward(q, k, v, cu_t, cu_t, mx, mx, p, D ** -0.5, True, rng_state=rng)
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_varlen_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, cu_t, cu_t, mx, mx, p, D ** -0.5, True, rng_state=rng)
    qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    torch.manual_seed(77)
    out, lse, _ = R.zigzag_ring_flash_attn_varlen_func(qd, kd, vd, cu_t.to(dev), mx, dropout_p=p, causal=True,
                                                       return_attn_probs=True)
    out.backward(do.to(dev))
    _check("drop.varlen.out", out, ro, 0, kind="out")
    _check("drop.varlen.lse", lse, rl, 0, kind="lse")
    _grads_ok("drop.varlen", (qd.grad, kd.grad, vd.grad), (rdq, rdk, rdv))


@pytest.mark.extended
@pytest.mark.parametrize("W,stride", [(2, 1), (4, 2)])
def test_llama3_dropout_hip_matches_single_device_oracle(W, stride):
    """llama3 over W ranks sharing the GPU, dropout on: every rank / head group draws the bits of the unsharded call"""
    import torch.multiprocessing as mp
    import _dropout_worker as DW
    from conftest import free_port
    from oracle import flash_attn_ref as O
    from ring_flash_attn._common import draw_dropout_seed

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(DW.run, args=(W, free_port(), ret, True, stride), nprocs=W, join=True)
    q, k, v, do = DW.inputs()
    torch.manual_seed(DW.SEED)
    rng = torch.tensor([draw_dropout_seed(), 0])
    cu = torch.tensor(DW.CU, dtype=torch.int32)
    scale = DW.D ** -0.5
    ro, rl, _, _ = O._flash_attn_varlen_forward(q, k, v, cu, cu, 0, 0, DW.P_DROP, scale, True, rng_state=rng)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_varlen_backward(do, q, k, v, ro, rl, dq, dk, dv, cu, cu, 0, 0, DW.P_DROP, scale, True, rng_state=rng)
    T = DW.CU[-1] // W
    for r in range(W):
        got = ret[r]
        assert not isinstance(got, str), got
        sl = slice(r * T, (r + 1) * T)
        _check(f"r{r}.out", got["out"], ro[sl], 0, kind="out_ring")
        for name, ref in (("dq", dq), ("dk", dk), ("dv", dv)):
            _check(f"r{r}.{name}", got[name], ref[sl], 0, kind="grad_ring")


# ------------------------------------------------------------------------------------------------------------
# head dims 65 .. 96: kernel instances over the 128-wide LDS layout that run three of the four 32-column blocks of MFMA
# work (csrc/rfa_common.hpp: HeadGeo<96>) — D == 96 with LDS-DMA staging, 64 < D < 96 zero padded through registers
@pytest.mark.parametrize("B,Sq,Sk,H,Hk,D,causal,dtype", [
    (2, 700, 700, 4, 2, 96, True, BF),               # GQA, batch, ragged tails, the LDS-DMA instance
    (1, 512, 1300, 3, 3, 96, False, BF),             # rectangular, unmasked, odd head count (the last head's row ends the tensor)
    (1, 1000, 488, 2, 1, 96, True, torch.float16),   # more queries than keys (rows without keys), fp16 MFMAs
    (1, 640, 640, 2, 2, 80, True, BF),               # 64 < D < 96: register staging with zero fill, same instances
    (2, 300, 520, 4, 4, 72, False, BF),
    (1, 96, 4000, 2, 2, 96, True, BF),               # one query block against 63 key tiles
])
@pytest.mark.extended
def test_head_dims_65_to_96_match_oracle(B, Sq, Sk, H, Hk, D, causal, dtype):
    """forward (plain and merged into fp32 accumulators) and backward (plain and += outputs) of the three-block instances
    against the CPU oracle; packed (cu_seqlens) input through the same instances"""
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be, dev = get_backend(), _dev()
    g = torch.Generator().manual_seed(D + Sq)
    q = torch.randn(B, Sq, H, D, generator=g).to(dtype)
    k = torch.randn(B, Sk, Hk, D, generator=g).to(dtype)
    v = torch.randn(B, Sk, Hk, D, generator=g).to(dtype)
    do = torch.randn(B, Sq, H, D, generator=g).to(dtype)
    ro, rl, rdq, rdk, rdv = _oracle_dense(q, k, v, do, causal)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    scale = D ** -0.5
    out = torch.empty_like(qd)
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=dev)
    be.fwd(qd, kd, vd, softmax_scale=scale, causal=causal, out=out, lse=lse)
    _check("out", out, ro, 0, kind="out")
    _check("lse", lse, rl, 0, kind="lse")
    if not causal:                                   # the fused merge epilogue: two key halves into one accumulator
        acc = torch.zeros(B, Sq, H, D, dtype=torch.float32, device=dev)
        lacc = torch.empty(B, H, Sq, dtype=torch.float32, device=dev)
        half = Sk // 2
        be.fwd(qd, kd[:, :half], vd[:, :half], softmax_scale=scale, causal=False, out_acc=acc, lse_acc=lacc, acc_init=True)
        be.fwd(qd, kd[:, half:], vd[:, half:], softmax_scale=scale, causal=False, out_acc=acc, lse_acc=lacc)
        _check("out_acc", acc, ro, 0, kind="out")
        _check("lse_acc", lacc, rl, 0, kind="lse")
    delta = torch.empty_like(lse)
    be.bwd_preprocess(dod, out, delta)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq=dq, dk=dk, dv=dv)
    _grads_ok(f"D={D}", (dq, dk, dv), (rdq, rdk, rdv))
    dqa = torch.full((B, Sq, H, D), 3.0, dtype=torch.float32, device=dev)
    dka = torch.zeros((B, Sk, Hk, D), dtype=torch.float32, device=dev)
    dva = torch.zeros_like(dka)
    be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq_acc=dqa, dk_acc=dka, dv_acc=dva)
    _check("dq_acc", dqa - 3.0, rdq, 0, kind="grad")
    _check("dk_acc", dka, rdk, 0, kind="grad")
    _check("dv_acc", dva, rdv, 0, kind="grad")
    if B == 2 and Sq == Sk:                          # the same rows as two packed sequences of different lengths
        cu = torch.tensor([0, 300, 300 + Sq], dtype=torch.int32)
        T = int(cu[-1])
        qp, kp, vp, dop = (torch.cat([t[0, :300], t[1]], 0).contiguous() for t in (q, k, v, do))
        po, pl, pdq, pdk, pdv = _oracle_varlen(qp, kp, vp, dop, cu, cu, causal)
        qpd, kpd, vpd, dopd, cud = (t.to(dev) for t in (qp, kp, vp, dop, cu))
        o2 = torch.empty_like(qpd)
        l2 = torch.empty((H, T), dtype=torch.float32, device=dev)
        be.fwd(qpd, kpd, vpd, softmax_scale=scale, causal=causal, cu_seqlens_q=cud, cu_seqlens_k=cud, max_seqlen_q=Sq,
               max_seqlen_k=Sq, out=o2, lse=l2)
        _check("varlen.out", o2, po, 0, kind="out")
        d2 = torch.empty_like(l2)
        be.bwd_preprocess(dopd, o2, d2, cu_seqlens_q=cud, max_seqlen_q=Sq)
        g2 = [torch.empty_like(t) for t in (qpd, kpd, vpd)]
        be.bwd(dopd, qpd, kpd, vpd, l2, d2, softmax_scale=scale, causal=causal, cu_seqlens_q=cud, cu_seqlens_k=cud,
               max_seqlen_q=Sq, max_seqlen_k=Sq, dq=g2[0], dk=g2[1], dv=g2[2])
        _grads_ok(f"varlen D={D}", g2, (pdq, pdk, pdv))


# ------------------------------------------------------------------------------------------------------------
# the 128-row (4-wave) forward form: picked by the library for grids that would leave the chip under-filled
# (csrc/rfa_api.cpp: fewer than 384 workgroups of 256 rows); RFA_FWD_FORM=8x32 forces the 256-row form
@pytest.mark.parametrize("B,Sq,Sk,H,Hk,D,causal,dtype", [
    (1, 1000, 1000, 4, 2, 128, True, BF),            # odd tile count of the last workgroup, masked tails
    (2, 300, 777, 2, 2, 128, True, BF),              # bottom-right aligned, odd number of key tiles
    (1, 513, 513, 2, 1, 128, False, BF),             # one valid row in the last 128-row workgroup
    (1, 2048, 2048, 16, 8, 128, True, BF),           # the llama3 head-group regime the form exists for (128 workgroups of 256 rows)
    (1, 900, 260, 2, 2, 128, True, torch.float16),   # queries without any visible key (lse = +inf), fp16 MFMAs
    (1, 1500, 1500, 4, 4, 64, True, BF),             # head dim 64 instance
])
@pytest.mark.extended
def test_fwd_128_row_form_matches_oracle_and_the_256_row_form(monkeypatch, B, Sq, Sk, H, Hk, D, causal, dtype):
    """both forward forms on grids below the threshold: against the oracle, and against each other BIT FOR BIT (a wave's
    32 rows do not depend on how many waves share its workgroup) — plain outputs and the fused merge epilogue"""
    from oracle import flash_attn_ref as O
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be, dev = get_backend(), _dev()
    monkeypatch.setenv("RFA_FWD_KV_NSPLIT", "1")      # (split-KV launches regroup the sum over the keys: tested on their own)
    g = torch.Generator().manual_seed(Sq + Sk)
    q = torch.randn(B, Sq, H, D, generator=g).to(dtype)
    k = torch.randn(B, Sk, Hk, D, generator=g).to(dtype)
    v = torch.randn(B, Sk, Hk, D, generator=g).to(dtype)
    if Sq == 1000:                                   # spike keys: the deferred-rescale branch in the middle of the loop
        k[0, 300] = q[0, 400, 0:2] * 3.0
    ro, rl, _, _ = O._flash_attn_forward(q, k, v, 0.0, D ** -0.5, causal)
    res = {}
    for form in ("auto", "8x32"):
        monkeypatch.setenv("RFA_FWD_FORM", form)
        out = torch.empty(B, Sq, H, D, dtype=dtype, device=dev)
        lse = torch.empty(B, H, Sq, dtype=torch.float32, device=dev)
        be.fwd(q.to(dev), k.to(dev), v.to(dev), softmax_scale=D ** -0.5, causal=causal, out=out, lse=lse)
        acc = torch.zeros(B, Sq, H, D, dtype=torch.float32, device=dev)
        lacc = torch.empty(B, H, Sq, dtype=torch.float32, device=dev)
        half = Sk // 2
        be.fwd(q.to(dev), k.to(dev)[:, :half], v.to(dev)[:, :half], softmax_scale=D ** -0.5, causal=False, out_acc=acc, lse_acc=lacc, acc_init=True)
        be.fwd(q.to(dev), k.to(dev)[:, half:], v.to(dev)[:, half:], softmax_scale=D ** -0.5, causal=False, out_acc=acc, lse_acc=lacc)
        res[form] = (out, lse, acc, lacc)
        _check(f"{form}.out", out, ro, 0, kind="out")
        _check(f"{form}.lse", lse, rl, 0, kind="lse")
    for a_, b_ in zip(res["auto"], res["8x32"]):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("B,Sq,Sk,H,Hk,D,causal,nsplit", [
    (1, 256, 4096, 4, 2, 128, True, "0"),        # the regime it exists for: few rows, 64 key tiles -> 2 shares (the library's choice)
    (1, 2048, 8192, 16, 8, 128, True, "0"),      # round 5: a half-filled grid of 256-row workgroups with 128 key tiles -> TWO SHARES OF
                                                 # 256-ROW workgroups (rfa_api.cpp: fwd_split_256_rows; a llama3 head group at 2048 rows)
    (2, 200, 3000, 2, 2, 128, True, "3"),        # ragged tails, bottom-right alignment, odd tile count, forced 3 shares
    (1, 300, 1000, 2, 1, 64, False, "8"),        # head dim 64; more shares than tile pairs: empty shares
    (1, 700, 520, 2, 2, 128, True, "2"),         # more queries than keys: rows without any key (lse = +inf)
])
@pytest.mark.extended
def test_fwd_split_kv_matches_oracle(monkeypatch, B, Sq, Sk, H, Hk, D, causal, nsplit):
    """ABI 5 split-KV forward launches: the key tiles of a workgroup divided between several workgroups, normalised
    partials in a workspace, a combine pass — against the oracle and against the unsplit launch, plain outputs and the
    fused merge into fp32 accumulators (overwrite and +=)"""
    import ctypes as C

    from oracle import flash_attn_ref as O
    from ring_flash_attn import _C
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be, dev = get_backend(), _dev()
    g = torch.Generator().manual_seed(Sq * 7 + Sk)
    q = torch.randn(B, Sq, H, D, generator=g).to(BF)
    k = torch.randn(B, Sk, Hk, D, generator=g).to(BF)
    v = torch.randn(B, Sk, Hk, D, generator=g).to(BF)
    ro, rl, _, _ = O._flash_attn_forward(q, k, v, 0.0, D ** -0.5, causal)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    res = {}
    for mode in (nsplit, "1"):
        monkeypatch.setenv("RFA_FWD_KV_NSPLIT", mode)
        out = torch.empty(B, Sq, H, D, dtype=BF, device=dev)
        lse = torch.empty(B, H, Sq, dtype=torch.float32, device=dev)
        be.fwd(qd, kd, vd, softmax_scale=D ** -0.5, causal=causal, out=out, lse=lse)
        # accumulate mode: first half of the keys overwrites, second half merges in (non-causal halves: the ring steps' use)
        acc = torch.full((B, Sq, H, D), 7.0, dtype=torch.float32, device=dev)
        lacc = torch.full((B, H, Sq), 3.0, dtype=torch.float32, device=dev)
        half = Sk // 2
        be.fwd(qd, kd[:, :half], vd[:, :half], softmax_scale=D ** -0.5, causal=False, out_acc=acc, lse_acc=lacc, acc_init=True)
        be.fwd(qd, kd[:, half:], vd[:, half:], softmax_scale=D ** -0.5, causal=False, out_acc=acc, lse_acc=lacc)
        res[mode] = (out, lse, acc, lacc)
        _check(f"split{mode}.out", out, ro, 0, kind="out")
        _check(f"split{mode}.lse", lse, rl, 0, kind="lse")
    # the library's plan for the first case
    if nsplit == "0":
        a = _C.FwdArgs()
        a.B, a.Sq, a.Sk, a.H, a.Hk, a.D, a.dtype, a.causal = B, Sq, Sk, H, Hk, D, 0, 1
        n = C.c_int32()
        nbytes = be.lib.rfa_fwd_workspace_bytes(C.byref(a), C.byref(n))
        assert n.value == 2 and nbytes == 2 * B * Sq * H * (D + 1) * 4
    ra, rla, _, _ = O._flash_attn_forward(q, k, v, 0.0, D ** -0.5, False)
    _check("split.acc.out", res[nsplit][2], ra.float(), 0, kind="out")
    _check("split.acc.lse", res[nsplit][3], rla, 0, kind="lse")
    _check("split vs unsplit out", res[nsplit][0], res["1"][0].float(), 0, kind="out")
    _check("split vs unsplit acc", res[nsplit][2], res["1"][2], 0, kind="out")
    assert (res[nsplit][1] - res["1"][1])[torch.isfinite(res["1"][1])].abs().max().item() < 1e-5


@pytest.mark.parametrize("B,Sq,Sk,H,Hk,causal,dtype", [
    (1, 1000, 1000, 4, 2, True, torch.bfloat16),      # ragged last query block and key tile; 16 items on 256 CUs: one pass
    (3, 777, 1300, 8, 2, False, torch.float16),       # more keys than queries, fp16
    (2, 2048, 2048, 48, 8, True, torch.bfloat16),     # 768 items: three passes (the middle one dealt in reverse), GQA 6
    (1, 4096, 8192, 16, 4, True, torch.bfloat16),     # bottom-right aligned causal mask
])
def test_fwd_persistent_form_is_bit_identical(B, Sq, Sk, H, Hk, causal, dtype):
    """Round 6: the persistent 256-row forward (csrc/rfa_fwd.hip fwd_persist_kernel, RFA_FWD_P8x32): one workgroup per CU walks
    its share of the (batch, head, query block) items, the next item's first K/V tile and Q fragments fetched under the
    current item's last tile and epilogue.  Same arithmetic in the same order per query row as the 8 x 32 form: out and lse
    must be BIT-identical to it (which the reference fixtures and the oracle tests pin), in launches of one and of several
    passes."""
    from ring_flash_attn import config
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be = get_backend()
    dev = _dev()
    D = 128
    g = torch.Generator().manual_seed(Sq + H)
    q = torch.randn(B, Sq, H, D, generator=g).to(dtype).to(dev)
    k = torch.randn(B, Sk, Hk, D, generator=g).to(dtype).to(dev)
    v = torch.randn(B, Sk, Hk, D, generator=g).to(dtype).to(dev)
    res = {}
    for form in ("8x32", "p8x32"):
        with config.override(fwd_form=form):
            out = torch.full_like(q, float("nan"))
            lse = torch.full((B, H, Sq), float("nan"), dtype=torch.float32, device=dev)
            be.fwd(q, k, v, softmax_scale=D ** -0.5, causal=causal, out=out, lse=lse)
            res[form] = (out, lse)
    assert torch.isfinite(res["p8x32"][0]).all() and torch.isfinite(res["p8x32"][1]).all()
    assert torch.equal(res["p8x32"][0], res["8x32"][0]), "out differs from the 8 x 32 form"
    assert torch.equal(res["p8x32"][1], res["8x32"][1]), "lse differs from the 8 x 32 form"
    # a strided kv-packed view and an accumulate-mode call: the first runs the persistent form too, the second is not eligible
    kv = torch.stack([k, v], dim=2)
    with config.override(fwd_form="p8x32"):
        out2 = torch.empty_like(q)
        lse2 = torch.empty((B, H, Sq), dtype=torch.float32, device=dev)
        be.fwd(q, kv[:, :, 0], kv[:, :, 1], softmax_scale=D ** -0.5, causal=causal, out=out2, lse=lse2)
        assert torch.equal(out2, res["8x32"][0]) and torch.equal(lse2, res["8x32"][1])
        oacc = torch.zeros((B, Sq, H, D), dtype=torch.float32, device=dev)
        lacc = torch.empty((B, H, Sq), dtype=torch.float32, device=dev)
        be.fwd(q, k, v, softmax_scale=D ** -0.5, causal=causal, out_acc=oacc, lse_acc=lacc, acc_init=True)
        _check("acc-mode out", oacc, res["8x32"][0].float(), 1e-2, 2e-2)


@pytest.mark.extended
def test_fwd_split_kv_packed_sequences(single_rank_group, monkeypatch):
    """split-KV over packed sequences of very different lengths (shares that are empty for the short sequences) through the
    public varlen API, single-rank; and the llama3 path whose long gathered key ranges are what the form is for"""
    import ring_flash_attn as R

    dev = _dev()
    g = torch.Generator().manual_seed(99)
    cu = [0, 40, 2100, 2164, 4200]
    T = cu[-1]
    q = torch.randn(T, 4, 128, generator=g).to(BF)
    k = torch.randn(T, 2, 128, generator=g).to(BF)
    v = torch.randn(T, 2, 128, generator=g).to(BF)
    do = torch.randn(T, 4, 128, generator=g).to(BF)
    cut = torch.tensor(cu, dtype=torch.int32)
    ro, rl, rdq, rdk, rdv = _oracle_varlen(q, k, v, do, cut, cut, True)
    for mode in ("4", "1"):
        monkeypatch.setenv("RFA_FWD_KV_NSPLIT", mode)
        qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
        out, lse, _ = R.ring_flash_attn_varlen_func(qd, kd, vd, cut.to(dev), 2060, causal=True, return_attn_probs=True)
        out.backward(do.to(dev))
        _check(f"varlen.split{mode}.out", out, ro, 0, kind="out")
        _check(f"varlen.split{mode}.lse", lse, rl, 0, kind="lse")
        _grads_ok(f"varlen.split{mode}", (qd.grad, kd.grad, vd.grad), (rdq, rdk, rdv))


@pytest.mark.extended
def test_fwd_128_row_form_in_the_schedules(single_rank_group, monkeypatch):
    """packed sequences, half-sequence selectors and the fused fp32 merge epilogue: the zigzag varlen schedule forced
    onto its multi-step path (_testing.force_steps) with the library's choice of the forward form (128 rows on this
    grid) against the 256-row form: identical bits"""
    import ring_flash_attn as R

    dev = _dev()
    from ring_flash_attn import _testing

    _testing.force_steps(True)
    _testing.allow_host_staging(True)          # (the one-rank gloo group of the test process, device tensors)
    monkeypatch.setenv("RFA_FWD_KV_NSPLIT", "1")
    g = torch.Generator().manual_seed(64)
    cu = torch.tensor([0, 128, 1248, 2240], dtype=torch.int32, device=dev)
    q = torch.randn(2240, 4, 128, generator=g).to(BF).to(dev)
    k = torch.randn(2240, 2, 128, generator=g).to(BF).to(dev)
    v = torch.randn(2240, 2, 128, generator=g).to(BF).to(dev)
    res = {}
    for form in ("8x32", "auto"):
        monkeypatch.setenv("RFA_FWD_FORM", form)
        out, lse, _ = R.zigzag_ring_flash_attn_varlen_func(q, k, v, cu, 1120, causal=True, return_attn_probs=True)
        res[form] = (out.float().cpu(), lse.cpu())
    _testing.reset()
    assert torch.equal(res["auto"][0], res["8x32"][0]) and torch.equal(res["auto"][1], res["8x32"][1])
