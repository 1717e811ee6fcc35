"""Full-size checks at the BASELINE.json headline shape (B=1, S=8192, H=32, Hk=8, D=128, bf16,
causal) through size-independent properties, since the CPU oracle cannot finish the whole tensor
in seconds:
  * sampled query rows / key rows recomputed exactly on the host (fp64) — out, lse, dq, dk, dv
  * V = const  =>  out = const  (softmax rows sum to one over 8192 keys)
  * splitting the key range in two and merging with the fused accumulate epilogue == one pass
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

S, H, HK, D = 8192, 32, 8, 128


def _row(name, i, h, got, ref):
    """one sampled row (128 values) against its exact fp64 value: RELATIVE to the row itself — late rows of a long
    causal sequence have |out| ~ 0.02, where an absolute bound would hide a systematic error — plus the
    element-wise bound; the io dtype's rounding alone is 2^-9 ~ 2e-3 relative"""
    import os

    err = (got - ref)
    rel = (err.norm() / ref.norm().clamp_min(1e-30)).item()
    mx = err.abs().max().item()
    path = os.environ.get("RFA_TOL_LOG")
    if path:
        with open(path, "a") as f:
            f.write(f"row        rel-norm {rel:.3e} max_err {mx:.3e} max_ref {ref.abs().max().item():.3e}  headline.{name}[{i},{h}]\n")
    # (observed on MI355X: <= 2.7e-3; a row whose exact value is zero — dq of a query that sees one key — is held to
    #  the absolute floor instead)
    assert err.norm().item() <= 1e-2 * ref.norm().item() + 5e-4 * ref.numel() ** 0.5, \
        f"{name}[{i},{h}]: relative error of the row {rel:.3e} > 1e-2"
    assert mx <= 2e-3 + 1.6e-2 * ref.abs().max().item(), f"{name}[{i},{h}]: max|err| {mx:.3e}"


def _inputs(dev):
    g = torch.Generator().manual_seed(42)
    q = torch.randn(1, S, H, D, generator=g).to(torch.bfloat16)
    kv = torch.randn(1, S, 2, HK, D, generator=g).to(torch.bfloat16)
    do = torch.randn(1, S, H, D, generator=g).to(torch.bfloat16)
    return q, kv, do


def test_headline_sampled_rows(single_rank_group):
    import ring_flash_attn as R

    dev = torch.device("cuda:0")
    q, kv, do = _inputs(dev)
    qd, kvd = q.to(dev).requires_grad_(True), kv.to(dev).requires_grad_(True)
    out, lse, _ = R.zigzag_ring_flash_attn_kvpacked_func(qd, kvd, causal=True, return_attn_probs=True)
    out.backward(do.to(dev))
    out, lse = out.cpu().double(), lse.cpu().double()
    dq, dkv = qd.grad.cpu().double(), kvd.grad.cpu().double()
    qf, kf, vf, dof = q.double(), kv[:, :, 0].double(), kv[:, :, 1].double(), do.double()
    scale = 1.0 / math.sqrt(D)
    g = torch.Generator().manual_seed(1)
    rows = [0, 1, 255, 256, 4095, 4096, S - 1] + torch.randint(0, S, (9,), generator=g).tolist()
    for i in rows:
        h = int(torch.randint(0, H, (1,), generator=g))
        hk = h // (H // HK)
        s = (kf[0, : i + 1, hk] @ qf[0, i, h]) * scale
        l = torch.logsumexp(s, 0)
        p = torch.exp(s - l)
        o = p @ vf[0, : i + 1, hk]
        assert abs(l - lse[0, h, i]) < 1e-3
        _row("out", i, h, out[0, i, h], o)
        dp = vf[0, : i + 1, hk] @ dof[0, i, h]
        delta = (dof[0, i, h] * out[0, i, h]).sum()
        ds = p * (dp - delta) * scale
        ref_dq = ds @ kf[0, : i + 1, hk]
        _row("dq", i, h, dq[0, i, h], ref_dq)
    for j in [0, 1000, S - 1]:
        hk = int(torch.randint(0, HK, (1,), generator=g))
        dk, dv = torch.zeros(D, dtype=torch.float64), torch.zeros(D, dtype=torch.float64)
        for h in range(hk * (H // HK), (hk + 1) * (H // HK)):
            s = (qf[0, j:, h] @ kf[0, j, hk]) * scale
            p = torch.exp(s - lse[0, h, j:])
            dp = dof[0, j:, h] @ vf[0, j, hk]
            delta = (dof[0, j:, h] * out[0, j:, h]).sum(-1)
            ds = p * (dp - delta) * scale
            dk += ds @ qf[0, j:, h]
            dv += p @ dof[0, j:, h]
        _row("dk", j, hk, dkv[0, j, 0, hk], dk)
        _row("dv", j, hk, dkv[0, j, 1, hk], dv)


def test_headline_constant_v_and_split_merge(single_rank_group):
    from ring_flash_attn.backend import get_backend, set_backend

    set_backend(None)
    be = get_backend()
    dev = torch.device("cuda:0")
    q, kv, _ = _inputs(dev)
    q, k = q.to(dev), kv[:, :, 0].to(dev).contiguous()
    v = torch.full((1, S, HK, D), 0.75, dtype=torch.bfloat16, device=dev)
    scale = D ** -0.5
    out = torch.empty_like(q)
    lse = torch.empty((1, H, S), dtype=torch.float32, device=dev)
    be.fwd(q, k, v, softmax_scale=scale, causal=True, out=out, lse=lse)
    assert (out.float() - 0.75).abs().max().item() < 4e-3          # rows of P sum to 1

    # non-causal over [0,S) in one pass vs two half passes merged by the fused epilogue
    v = kv[:, :, 1].to(dev).contiguous()
    one = torch.empty((1, S, H, D), dtype=torch.float32, device=dev)
    l_one = torch.empty((1, H, S), dtype=torch.float32, device=dev)
    be.fwd(q, k, v, softmax_scale=scale, causal=False, out_acc=one, lse_acc=l_one, acc_init=True)
    two = torch.empty_like(one)
    l_two = torch.empty_like(l_one)
    be.fwd(q, k[:, : S // 2], v[:, : S // 2], softmax_scale=scale, causal=False, out_acc=two, lse_acc=l_two, acc_init=True)
    be.fwd(q, k[:, S // 2:], v[:, S // 2:], softmax_scale=scale, causal=False, out_acc=two, lse_acc=l_two)
    assert (l_one - l_two).abs().max().item() < 1e-4
    assert (one - two).abs().max().item() < 2e-3


def test_max_length_65536_single_gpu(single_rank_group):
    """The headline's TOTAL sequence (8192 x 8 = 65536) on one GPU: 64-bit addressing, 256 query blocks
    per head, 1024 KV tiles.  Size-independent checks: sampled rows of out/lse/dq in fp64 on the host,
    one key row of dk/dv, and constant-V => constant out."""
    import ring_flash_attn as R

    dev = torch.device("cuda:0")
    SL, HH, HKK = 65536, 4, 2
    g = torch.Generator().manual_seed(65)
    q = torch.randn(1, SL, HH, D, generator=g).to(torch.bfloat16)
    kv = torch.randn(1, SL, 2, HKK, D, generator=g).to(torch.bfloat16)
    do = torch.randn(1, SL, HH, D, generator=g).to(torch.bfloat16)
    qd, kvd = q.to(dev).requires_grad_(True), kv.to(dev).requires_grad_(True)
    out, lse, _ = R.zigzag_ring_flash_attn_kvpacked_func(qd, kvd, causal=True, return_attn_probs=True)
    out.backward(do.to(dev))
    out, lse = out.cpu().double(), lse.cpu().double()
    dq, dkv = qd.grad.cpu().double(), kvd.grad.cpu().double()
    qf, kf, vf, dof = q.double(), kv[:, :, 0].double(), kv[:, :, 1].double(), do.double()
    scale = 1.0 / math.sqrt(D)
    for i, h in [(0, 0), (255, 1), (256, 2), (32767, 3), (32768, 0), (65535, 1), (50001, 2)]:
        hk = h // (HH // HKK)
        s = (kf[0, : i + 1, hk] @ qf[0, i, h]) * scale
        l = torch.logsumexp(s, 0)
        p = torch.exp(s - l)
        o = p @ vf[0, : i + 1, hk]
        assert abs(l - lse[0, h, i]) < 1e-3
        _row("out", i, h, out[0, i, h], o)
        dp = vf[0, : i + 1, hk] @ dof[0, i, h]
        delta = (dof[0, i, h] * out[0, i, h]).sum()
        ref_dq = (p * (dp - delta) * scale) @ kf[0, : i + 1, hk]
        _row("dq", i, h, dq[0, i, h], ref_dq)
    j, hk = 60000, 1
    dk, dv = torch.zeros(D, dtype=torch.float64), torch.zeros(D, dtype=torch.float64)
    for h in range(hk * (HH // HKK), (hk + 1) * (HH // HKK)):
        s = (qf[0, j:, h] @ kf[0, j, hk]) * scale
        p = torch.exp(s - lse[0, h, j:])
        dp = dof[0, j:, h] @ vf[0, j, hk]
        delta = (dof[0, j:, h] * out[0, j:, h]).sum(-1)
        ds = p * (dp - delta) * scale
        dk += ds @ qf[0, j:, h]
        dv += p @ dof[0, j:, h]
    _row("dk", j, hk, dkv[0, j, 0, hk], dk)
    _row("dv", j, hk, dkv[0, j, 1, hk], dv)
