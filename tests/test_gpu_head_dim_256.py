"""Head dims 129 … 256 (csrc/rfa_bigd.hip) against the CPU oracle (`-m gpu`).

flash_attn accepts head dims up to 256 and the reference only requires d % 8 == 0
(/root/reference/test/test_zigzag_ring_flash_attn_func.py:35).  Dims above 128 run their own three kernels (two
128-column chunks per row, one wave per SIMD); everything around them — parameter blocks, masks, merge / accumulate
epilogues, dropout mask, side kernels, schedules — is shared with the 128-wide path, so the cases below go through the
same entry points: the C ABI directly (plain and accumulate outputs, windows, dropout, packed halves) and the public
schedules (ring / zigzag / varlen / llama3 over several ranks sharing the GPU).
Tolerances: tests/_tol.py (same kinds as the 128-wide tests).
"""
import pytest
import torch

from test_gpu_kernels import BF, _check, _dev, _grads_ok

pytestmark = [pytest.mark.gpu, pytest.mark.extended]


def _oracle(q, k, v, do, causal, window=(-1, -1), cu=None, drop=None):
    from oracle import flash_attn_ref as O

    D = q.shape[-1]
    scale = D ** -0.5
    p = drop[0] if drop else 0.0
    kw = dict(rng_state=torch.tensor([drop[1], 0])) if drop else {}
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    if cu is None:
        out, lse, _, _ = O._flash_attn_forward(q, k, v, p, scale, causal, window[0], window[1], **kw)
        O._flash_attn_backward(do, q, k, v, out, lse, dq, dk, dv, p, scale, causal, window[0], window[1], **kw)
    else:
        out, lse, _, _ = O._flash_attn_varlen_forward(q, k, v, cu[0], cu[1], 0, 0, p, scale, causal, window[0], window[1], **kw)
        O._flash_attn_varlen_backward(do, q, k, v, out, lse, dq, dk, dv, cu[0], cu[1], 0, 0, p, scale, causal, window[0], window[1], **kw)
    return out, lse, dq, dk, dv


@pytest.mark.parametrize("D,B,Sq,Sk,H,Hk,causal,dtype", [
    (256, 2, 777, 777, 4, 2, True, BF),            # GQA, ragged tails
    (256, 1, 300, 901, 2, 2, True, BF),            # bottom-right aligned, more keys than queries
    (256, 1, 640, 640, 2, 1, False, BF),           # no mask
    (192, 2, 515, 515, 4, 4, True, BF),            # second chunk half filled (zero-padded DMA lanes)
    (160, 1, 1000, 1000, 2, 2, True, torch.float16),
    (136, 1, 260, 130, 2, 1, True, BF),            # one 16-byte chunk beyond 128; rows without a visible key
    (256, 1, 2500, 2500, 2, 1, True, BF),          # many tiles (both LDS stages, every wave role)
    (192, 1, 2500, 2500, 2, 1, True, BF),          # ... in the three-quarter instances (head dims <= 192 skip the padding quarter)
])
def test_dense_forward_backward(D, B, Sq, Sk, H, Hk, causal, dtype):
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be, dev = get_backend(), _dev()
    g = torch.Generator().manual_seed(D + Sq)
    q, k, v = (torch.randn(B, s_, h_, D, generator=g).to(dtype) for s_, h_ in ((Sq, H), (Sk, Hk), (Sk, Hk)))
    do = torch.randn(B, Sq, H, D, generator=g).to(dtype)
    ro, rl, rdq, rdk, rdv = _oracle(q, k, v, do, causal)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    scale = D ** -0.5
    out, lse = torch.empty_like(qd), torch.empty((B, H, Sq), dtype=torch.float32, device=dev)
    be.fwd(qd, kd, vd, softmax_scale=scale, causal=causal, out=out, lse=lse)
    _check("out", out, ro, 0, kind="out")
    _check("lse", lse, rl, 0, kind="lse")
    delta = torch.empty_like(lse)
    be.bwd_preprocess(dod, out, delta)
    ref_delta = (do.float() * ro.float()).sum(-1).transpose(1, 2)
    _check("delta", delta, ref_delta, 2e-2, 1e-2)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq=dq, dk=dk, dv=dv)
    _grads_ok(f"d{D}", (dq, dk, dv), (rdq, rdk, rdv))
    # accumulate outputs (what the ring steps use): fp32 accumulators, initialised by the first call, added to by a second
    oa, la = torch.empty(B, Sq, H, D, dtype=torch.float32, device=dev), torch.empty(B, H, Sq, dtype=torch.float32, device=dev)
    be.fwd(qd, kd, vd, softmax_scale=scale, causal=causal, out_acc=oa, lse_acc=la, acc_init=True)
    _check("out_acc", oa, ro, 0, kind="out")
    fin = torch.isfinite(rl)
    assert torch.allclose(la.cpu()[fin], rl[fin], atol=1e-4)
    dqa = torch.empty(B, Sq, H, D, dtype=torch.float32, device=dev)
    dka, dva = (torch.empty(B, Sk, Hk, D, dtype=torch.float32, device=dev) for _ in range(2))
    be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq_acc=dqa, dk_acc=dka, dv_acc=dva, acc_init=True)
    _grads_ok(f"d{D}.acc", (dqa, dka, dva), (rdq, rdk, rdv))
    be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq_acc=dqa, dk_acc=dka, dv_acc=dva, acc_init=False)
    _grads_ok(f"d{D}.acc2", (dqa / 2, dka / 2, dva / 2), (rdq, rdk, rdv))


@pytest.mark.parametrize("Sq,Sk,causal,B,H,Hk,D", [
    (1024, 1024, True, 1, 4, 2, 256),      # dense causal: triangular scratch, query range shared by 2 - 4 workgroups
    (300, 901, True, 2, 2, 2, 256),        # bottom-right aligned, ragged
    (640, 384, False, 1, 2, 1, 256),       # rectangular scratch
    (1024, 1024, True, 1, 4, 2, 192),      # second 128-column launch of rfa_dqs.hip with 64 columns (zero-filled K lanes)
    (300, 901, True, 2, 2, 2, 136),        # ... with 8 columns
    (640, 384, False, 1, 2, 1, 160),
])
def test_ds_spill_backward_at_head_dim_256(monkeypatch, Sq, Sk, causal, B, H, Hk, D):
    """the 5-GEMM backward at D = 136 .. 256 (the dK launch of rfa_bigd.hip stores dS, rfa_dqs.hip computes dQ from it in
    two 128-column launches, the second over the D - 128 columns that exist) against the oracle, against the 7-GEMM form
    (dK / dV bit-identical: the same kernel with and without the stores)"""
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be, dev = get_backend(), _dev()
    assert be.lib.rfa_bwd_ds_scratch_bytes is not None
    g = torch.Generator().manual_seed(Sq + Sk)
    q, k, v = (torch.randn(B, s_, h_, D, generator=g).to(BF) for s_, h_ in ((Sq, H), (Sk, Hk), (Sk, Hk)))
    do = torch.randn(B, Sq, H, D, generator=g).to(BF)
    ro, rl, rdq, rdk, rdv = _oracle(q, k, v, do, causal)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    scale = D ** -0.5
    out, lse = torch.empty_like(qd), torch.empty((B, H, Sq), dtype=torch.float32, device=dev)
    be.fwd(qd, kd, vd, softmax_scale=scale, causal=causal, out=out, lse=lse)
    delta = torch.empty_like(lse)
    be.bwd_preprocess(dod, out, delta)
    res = {}
    for spill in ("1", "0"):
        monkeypatch.setenv("RFA_BWD_DS_SPILL", spill)
        dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq=dq, dk=dk, dv=dv)
        _grads_ok(f"d{D}.spill{spill}", (dq, dk, dv), (rdq, rdk, rdv))
        res[spill] = (dq, dk, dv)
    assert torch.equal(res["1"][1], res["0"][1]) and torch.equal(res["1"][2], res["0"][2])
    monkeypatch.setenv("RFA_BWD_DS_SPILL", "1")
    dqa = torch.zeros(B, Sq, H, D, dtype=torch.float32, device=dev)
    dka, dva = (torch.empty(B, Sk, Hk, D, dtype=torch.float32, device=dev) for _ in range(2))
    be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq_acc=dqa, dk_acc=dka, dv_acc=dva, acc_init=True)
    _grads_ok(f"d{D}.spill.acc", (dqa, dka, dva), (rdq, rdk, rdv))


def test_forward_merge_of_two_key_halves_equals_one_call():
    """the fused online merge epilogue at D = 256: keys split in two calls (the second one merging into the first's
    accumulators) against one call over all keys"""
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be, dev = get_backend(), _dev()
    g = torch.Generator().manual_seed(5)
    B, S, H, Hk, D = 1, 1024, 2, 1, 256
    q, k, v = (torch.randn(B, S, h_, D, generator=g).to(BF).to(dev) for h_ in (H, Hk, Hk))
    scale = D ** -0.5
    out, lse = torch.empty_like(q), torch.empty(B, H, S, dtype=torch.float32, device=dev)
    be.fwd(q, k, v, softmax_scale=scale, causal=False, out=out, lse=lse)
    oa, la = torch.empty(B, S, H, D, dtype=torch.float32, device=dev), torch.empty(B, H, S, dtype=torch.float32, device=dev)
    be.fwd(q, k, v, softmax_scale=scale, causal=False, out_acc=oa, lse_acc=la, acc_init=True, k_half=1)
    be.fwd(q, k, v, softmax_scale=scale, causal=False, out_acc=oa, lse_acc=la, k_half=2)
    assert (oa - out.float()).abs().max().item() < 1e-2
    assert (la - lse).abs().max().item() < 1e-4


@pytest.mark.parametrize("Sq,Sk,D,causal,window", [
    (1000, 1000, 256, True, (200, 0)),
    (700, 700, 192, False, (130, 70)),
    (333, 900, 256, True, (64, -1)),
    (900, 333, 256, False, (-1, 50)),
])
def test_sliding_windows(Sq, Sk, D, causal, window):
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be, dev = get_backend(), _dev()
    g = torch.Generator().manual_seed(11)
    B, H, Hk = 2, 4, 2
    q, k, v = (torch.randn(B, s_, h_, D, generator=g).to(BF) for s_, h_ in ((Sq, H), (Sk, Hk), (Sk, Hk)))
    do = torch.randn(B, Sq, H, D, generator=g).to(BF)
    ro, rl, rdq, rdk, rdv = _oracle(q, k, v, do, causal, window)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    scale = D ** -0.5
    out, lse = torch.empty_like(qd), torch.empty((B, H, Sq), dtype=torch.float32, device=dev)
    be.fwd(qd, kd, vd, softmax_scale=scale, causal=causal, out=out, lse=lse, window=window)
    _check("out", out, ro, 0, kind="out")
    _check("lse", lse, rl, 0, kind="lse")
    delta = torch.empty_like(lse)
    be.bwd_preprocess(dod, out, delta)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq=dq, dk=dk, dv=dv, window=window)
    _grads_ok("window", (dq, dk, dv), (rdq, rdk, rdv))


@pytest.mark.parametrize("D,causal", [(256, True), (192, False)])
def test_dropout(D, causal):
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be, dev = get_backend(), _dev()
    g = torch.Generator().manual_seed(D)
    B, S, H, Hk, p, seed = 2, 555, 4, 2, 0.2, 0x0FED_CBA9_8765_4321
    q, k, v = (torch.randn(B, S, h_, D, generator=g).to(BF) for h_ in (H, Hk, Hk))
    do = torch.randn(B, S, H, D, generator=g).to(BF)
    ro, rl, rdq, rdk, rdv = _oracle(q, k, v, do, causal, drop=(p, seed))
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    scale = D ** -0.5
    out, lse = torch.empty_like(qd), torch.empty((B, H, S), dtype=torch.float32, device=dev)
    drop = (p, seed, 0, 0, 0)
    be.fwd(qd, kd, vd, softmax_scale=scale, causal=causal, out=out, lse=lse, dropout=drop)
    _check("drop.out", out, ro, 0, kind="out")
    _check("drop.lse", lse, rl, 0, kind="lse")
    delta = torch.empty_like(lse)
    be.bwd_preprocess(dod, out, delta)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    be.bwd(dod, qd, kd, vd, lse, delta, softmax_scale=scale, causal=causal, dq=dq, dk=dk, dv=dv, dropout=drop)
    _grads_ok("drop", (dq, dk, dv), (rdq, rdk, rdv))


@pytest.mark.parametrize("cu", [[0, 128, 1248, 2001], [0, 3, 70, 71, 600]])
def test_packed_sequences_public_api(single_rank_group, cu):
    """packed sequences (with empty and tiny ones) through zigzag_ring_flash_attn_varlen_func and the llama3 entry"""
    import ring_flash_attn as R

    dev = _dev()
    cu_t = torch.tensor(cu, dtype=torch.int32)
    g = torch.Generator().manual_seed(int(cu[-1]))
    T, H, Hk, D = int(cu[-1]), 4, 2, 256
    q, k, v, do = (torch.randn(T, h, D, generator=g).to(BF) for h in (H, Hk, Hk, H))
    ro, rl, rdq, rdk, rdv = _oracle(q, k, v, do, True, cu=(cu_t, cu_t))
    qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    max_len = int((cu_t[1:] - cu_t[:-1]).max())
    out, lse, _ = R.ring_flash_attn_varlen_func(qd, kd, vd, cu_t.to(dev), max_len, causal=True, return_attn_probs=True)
    out.backward(do.to(dev))
    _check("varlen.out", out, ro, 0, kind="out")
    _check("varlen.lse", lse, rl, 0, kind="lse")
    _grads_ok("varlen", (qd.grad, kd.grad, vd.grad), (rdq, rdk, rdv))
    cq, ck, mq, mk, sl = R.llama3_flash_attn_prepare_cu_seqlens(cu_t, causal=True, rank=0, world_size=1)
    qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    out = R.llama3_flash_attn_varlen_func(qd, kd, vd, cq.to(dev), ck.to(dev), mq, mk, heads_k_stride=1, local_k_slice=sl, causal=True)
    out.backward(do.to(dev))
    _check("llama3.out", out, ro, 0, kind="out")
    _grads_ok("llama3", (qd.grad, kd.grad, vd.grad), (rdq, rdk, rdv))


@pytest.mark.parametrize("cfg", [
    dict(kind="zigzag", W=4, B=1, S=2048, H=4, Hk=2, D=256, seed=223),
    dict(kind="zigzag", W=2, B=2, S=1024, H=2, Hk=2, D=192, seed=224),
    dict(kind="ring", W=2, B=1, S=2048, H=4, Hk=4, D=256, causal=True, seed=225),
    dict(kind="zigzag_varlen", W=2, cu=[0, 512, 2560, 4096], H=4, Hk=2, D=256, seed=226),
])
def test_schedules_over_several_ranks(monkeypatch, cfg):
    """the ring schedules at head dim 256 / 192: W processes share the GPU, every step kind (halves, packed halves,
    merge epilogue, two-phase fp32 accumulation of dK/dV, both zigzag exchange forms) on the wide kernels"""
    from test_gpu_configs import _run_and_compare

    for mode in ("gather", "ring") if cfg["kind"] == "zigzag" else (None,):
        if mode:
            monkeypatch.setenv("RFA_ZIGZAG_EXCHANGE", mode)
        _run_and_compare(cfg)


def test_head_dim_limits_of_the_c_abi():
    """D = 256 is accepted, 264 and 100 (not a multiple of 8) are RFA_ERR_HEAD_DIM — through the ctypes wrapper"""
    from ring_flash_attn.backend import get_backend
    from ring_flash_attn._testing import set_backend

    set_backend(None)
    be, dev = get_backend(), _dev()
    for D, ok in ((256, True), (264, False), (100, False)):
        q = torch.randn(1, 64, 1, D, device=dev).to(BF)
        out, lse = torch.empty_like(q), torch.empty(1, 1, 64, dtype=torch.float32, device=dev)
        if ok:
            be.fwd(q, q, q, softmax_scale=1.0, causal=True, out=out, lse=lse)
        else:
            with pytest.raises(Exception, match="head_dim"):
                be.fwd(q, q, q, softmax_scale=1.0, causal=True, out=out, lse=lse)
