"""oracle/flash_attn_ref.py — TEST INFRASTRUCTURE ONLY (parity oracle; never the product path).

CPU restatement, in plain PyTorch, of the four private entry points of the third-party package
that the reference delegates all arithmetic to and which is absent from /root/reference:

    flash_attn.flash_attn_interface._flash_attn_forward          (used at zigzag_ring_flash_attn.py:52,
                                                                  ring_flash_attn.py:53)
    flash_attn.flash_attn_interface._flash_attn_backward         (zigzag_ring_flash_attn.py:156,
                                                                  ring_flash_attn.py:131)
    flash_attn.flash_attn_interface._flash_attn_varlen_forward   (ring_flash_attn_varlen.py:77,
                                                                  zigzag_ring_flash_attn_varlen.py:137,
                                                                  llama3_flash_attn_varlen.py:147)
    flash_attn.flash_attn_interface._flash_attn_varlen_backward  (ring_flash_attn_varlen.py:169,
                                                                  zigzag_ring_flash_attn_varlen.py:275,
                                                                  llama3_flash_attn_varlen.py:282)

Dependency: PyPI `flash-attn` (Dao-AILab/flash-attention).  The reference pins no version
(pyproject.toml:1-22); its call sites select behaviour by introspection (utils.py:13-29) and
support both the <2.7 and >=2.7 calling conventions.  This file restates the **>= 2.7**
convention: keyword `window_size_left/right`, 4-tuple forward return
`(out, softmax_lse, S_dmask, rng_state)`, packed `(nheads, total)` varlen LSE, bottom-right
aligned causal mask, rows without any visible key -> out 0 / lse +inf, in-place dq/dk/dv.

Published algorithm restated (FlashAttention-2, Dao 2023, Alg. 1/2), computed exactly (no
tiling — tiling does not change the mathematical result), fp32 math on the given inputs:
    S = scale * Q K^T (+mask)      lse = logsumexp(S)      O = softmax(S) V
    P = exp(S - lse)   D = rowsum(dO * O)   dP = dO V^T   dS = P * (dP - D)
    dQ = scale dS K    dK = scale dS^T Q    dV = P^T dO    (dK,dV summed over a GQA group)
Outputs are rounded to the input dtype exactly where flash_attn rounds (out, dq, dk, dv).

PARITY PINNING: the reference's own tests hold no numeric golden vectors for this boundary
(test/utils.py:15-38 only prints diffs), so it is "parity unpinned" against flash_attn itself.
It is pinned instead (tests/test_oracle.py) against an independent fp64 softmax-attention
with torch autograd, and against the C restatement oracle/attn_ref.c.
"""
import math
from typing import Optional

import torch

__all__ = [
    "_flash_attn_forward",
    "_flash_attn_backward",
    "_flash_attn_varlen_forward",
    "_flash_attn_varlen_backward",
]

_CHUNK = 1024  # query rows per score block (bounds memory, not a numerical choice)


def _window(causal, window):
    """(left, right) with -1 = unbounded; causal forces right = 0 (flash_attn >= 2.3 semantics)"""
    wl, wr = (-1, -1) if window is None else (int(window[0]), int(window[1]))
    wl = -1 if wl is None or wl < 0 else wl
    wr = -1 if wr is None or wr < 0 else wr
    if causal:
        wr = 0
    return wl, wr


def _mask(sq0: int, nq: int, lq: int, lk: int, causal: bool, device, window=None):
    """bool (nq, lk): True where key j is visible to query row sq0+i (bottom-right aligned):
    i + (lk - lq) - left <= j <= i + (lk - lq) + right   (each bound only when it is set)."""
    wl, wr = _window(causal, window)
    if wl < 0 and wr < 0:
        return None
    qi = torch.arange(sq0, sq0 + nq, device=device).unsqueeze(1) + (lk - lq)
    kj = torch.arange(lk, device=device).unsqueeze(0)
    m = torch.ones((nq, lk), dtype=torch.bool, device=device)
    if wr >= 0:
        m &= kj <= qi + wr
    if wl >= 0:
        m &= kj >= qi - wl
    return m


# --------------------------------------------------------------------------------------------
# Dropout.  flash_attn's own mask comes from its private Philox stream and cannot be reproduced without the
# package; what CAN be stated exactly is the mask THIS library defines (include/rfa.h: rfa_fwd_args.dropout_p,
# csrc/rfa_common.hpp: drop_word): a pure function of (seed, batch, head, query position, key position)
#     word(i, jq) = fmix32((i * 0x9E3779B1) ^ (jq * 0x85EBCA77) ^ head_key),  jq = j >> 2
#     keep(i, j)  = byte (j & 3) of word(i, j >> 2)  <  round((1 - p) * 256)
# restated here with numpy uint32 arithmetic.  Semantics are flash_attn's: probabilities entering P V are masked and
# rescaled, lse is that of the undropped softmax, dP = mask * (dO V^T) * rescale, dS = P (dP - delta).  The keep
# probability is quantised to keep / 256 (like flash_attn's uint8 threshold), so the rescale is 256 / keep — the
# reciprocal of the probability the mask REALLY keeps with — not 1 / (1 - p): E[dropout(P)] = P exactly for every p
# (with 1 / (1 - p) the expectation was off by up to 0.3 % when p is not a multiple of 1/256).
# --------------------------------------------------------------------------------------------
def _fmix32(x):
    import numpy as np

    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x85EBCA6B)).astype(np.uint32)
    x ^= x >> np.uint32(13)
    x = (x * np.uint32(0xC2B2AE35)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def drop_threshold(p: float) -> int:
    """keep threshold in 1/256 (rfa_api.cpp: drop_threshold)"""
    if not p > 0:
        return 256
    return max(0, min(255, int((1.0 - float(torch.tensor(p, dtype=torch.float32))) * 256.0 + 0.5)))


def drop_rescale(p: float) -> float:
    """256 / keep threshold (rfa_api.cpp: drop_rescale); 0 when nothing is kept"""
    k = drop_threshold(p)
    return 256.0 / k if k > 0 else 0.0


def dropout_keep(seed: int, p: float, batch: int, heads, i_pos, j_pos) -> torch.Tensor:
    """bool (len(heads), len(i_pos), len(j_pos)): True where the probability of (head, query position, key
    position) is kept.  heads / i_pos / j_pos: 1-D integer sequences of GLOBAL indices."""
    import numpy as np

    with np.errstate(over="ignore"):
        seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        h = np.asarray(list(heads), dtype=np.uint32)
        hk = _fmix32(np.uint32(seed & 0xFFFFFFFF) ^ _fmix32(np.uint32(seed >> 32) ^ _fmix32(
            (h * np.uint32(0x27D4EB2F)).astype(np.uint32) ^ _fmix32(np.asarray([(batch + 0x165667B1) & 0xFFFFFFFF], dtype=np.uint32)))))
        i = np.asarray(i_pos, dtype=np.int64).astype(np.uint32)
        j = np.asarray(j_pos, dtype=np.int64).astype(np.uint32)
        a = (i * np.uint32(0x9E3779B1)).astype(np.uint32)[None, :, None]
        b = ((j >> np.uint32(2)) * np.uint32(0x85EBCA77)).astype(np.uint32)[None, None, :]
        w = _fmix32(a ^ b ^ hk[:, None, None])
        byte = (w >> (np.uint32(8) * (j & np.uint32(3)))[None, None, :]) & np.uint32(0xFF)
        keep = byte < np.uint32(drop_threshold(p))
    return torch.from_numpy(keep)


def _drop_mask(drop, H, s0, n, Lk):
    """drop = dict(p, seed, batch, head0, q_pos0, k_pos0) -> keep (H, n, Lk) for query rows s0 .. s0+n-1"""
    return dropout_keep(drop["seed"], drop["p"], drop.get("batch", 0), range(drop.get("head0", 0), drop.get("head0", 0) + H),
                        range(drop.get("q_pos0", 0) + s0, drop.get("q_pos0", 0) + s0 + n),
                        range(drop.get("k_pos0", 0), drop.get("k_pos0", 0) + Lk))


def _fwd_one(q, k, v, scale, causal, window=None, drop=None):
    """q (Lq,H,D), k,v (Lk,Hk,D) -> out fp32 (Lq,H,D), lse fp32 (H,Lq).  drop: see _drop_mask."""
    Lq, H, D = q.shape
    Lk, Hk, _ = k.shape
    G = H // Hk
    qf = q.float().permute(1, 0, 2)                       # (H,Lq,D)
    kf = k.float().permute(1, 0, 2).repeat_interleave(G, dim=0)   # (H,Lk,D)
    vf = v.float().permute(1, 0, 2).repeat_interleave(G, dim=0)
    out = torch.zeros((H, Lq, D), dtype=torch.float32, device=q.device)
    lse = torch.full((H, Lq), float("inf"), dtype=torch.float32, device=q.device)
    if Lk == 0:
        return out.permute(1, 0, 2), lse
    for s0 in range(0, Lq, _CHUNK):
        n = min(_CHUNK, Lq - s0)
        s = torch.matmul(qf[:, s0:s0 + n], kf.transpose(1, 2)) * scale        # (H,n,Lk)
        m = _mask(s0, n, Lq, Lk, causal, q.device, window)
        if m is not None:
            s = s.masked_fill(~m, float("-inf"))
        l = torch.logsumexp(s, dim=-1)                                         # -inf for empty rows
        p = torch.exp(s - l.unsqueeze(-1))
        empty = torch.isinf(l) & (l < 0)
        p = torch.where(empty.unsqueeze(-1), torch.zeros_like(p), p)
        if drop is not None and drop["p"] > 0:
            p = torch.where(_drop_mask(drop, H, s0, n, Lk), p * drop_rescale(drop["p"]), torch.zeros_like(p))
        out[:, s0:s0 + n] = torch.matmul(p, vf)
        lse[:, s0:s0 + n] = torch.where(empty, torch.full_like(l, float("inf")), l)
    return out.permute(1, 0, 2), lse


def _bwd_one(dout, q, k, v, out, lse, scale, causal, delta=None, window=None, drop=None):
    """returns fp32 dq (Lq,H,D), dk, dv (Lk,Hk,D).  lse (H,Lq); delta (H,Lq) overrides rowsum(dO*O)."""
    Lq, H, D = q.shape
    Lk, Hk, _ = k.shape
    G = H // Hk
    qf = q.float().permute(1, 0, 2)
    dof = dout.float().permute(1, 0, 2)
    kf = k.float().permute(1, 0, 2).repeat_interleave(G, dim=0)
    vf = v.float().permute(1, 0, 2).repeat_interleave(G, dim=0)
    if delta is None:
        delta = (dof * out.float().permute(1, 0, 2)).sum(-1)                    # (H,Lq)
    dq = torch.zeros((H, Lq, D), dtype=torch.float32, device=q.device)
    dk = torch.zeros((H, Lk, D), dtype=torch.float32, device=q.device)
    dv = torch.zeros((H, Lk, D), dtype=torch.float32, device=q.device)
    if Lk > 0:
        for s0 in range(0, Lq, _CHUNK):
            n = min(_CHUNK, Lq - s0)
            s = torch.matmul(qf[:, s0:s0 + n], kf.transpose(1, 2)) * scale
            p = torch.exp(s - lse[:, s0:s0 + n].unsqueeze(-1))
            m = _mask(s0, n, Lq, Lk, causal, q.device, window)
            if m is not None:
                p = torch.where(m, p, torch.zeros_like(p))
            dp = torch.matmul(dof[:, s0:s0 + n], vf.transpose(1, 2))
            pd = p
            if drop is not None and drop["p"] > 0:
                keep = _drop_mask(drop, H, s0, n, Lk)
                rp = drop_rescale(drop["p"])
                dp = torch.where(keep, dp * rp, torch.zeros_like(dp))
                pd = torch.where(keep, p * rp, torch.zeros_like(p))
            ds = p * (dp - delta[:, s0:s0 + n].unsqueeze(-1)) * scale
            dq[:, s0:s0 + n] = torch.matmul(ds, kf)
            dk += torch.matmul(ds.transpose(1, 2), qf[:, s0:s0 + n])
            dv += torch.matmul(pd.transpose(1, 2), dof[:, s0:s0 + n])
    dk = dk.view(Hk, G, Lk, D).sum(1)
    dv = dv.view(Hk, G, Lk, D).sum(1)
    return dq.permute(1, 0, 2), dk.permute(1, 0, 2), dv.permute(1, 0, 2)


def _new_rng_state():
    """(seed, offset) like flash_attn's rng_state, drawn from torch's default CPU generator"""
    return torch.tensor([int(torch.randint(0, 2 ** 62, (1,)).item()), 0], dtype=torch.int64)


def _drop(dropout_p, rng_state, **pos):
    if not dropout_p:
        return None
    assert rng_state is not None, "oracle: the backward of a dropout call needs the forward's rng_state"
    return dict(p=float(dropout_p), seed=int(rng_state[0]), **pos)


def _check(dropout_p, window_size_left, window_size_right, softcap, alibi_slopes):
    if dropout_p:
        if not 0 <= dropout_p < 1:
            raise ValueError("dropout_p must be in [0, 1)")
        if window_size_left >= 0 or window_size_right >= 0:
            raise NotImplementedError("dropout together with a window is not restated (nor implemented by the kernels)")
    assert not softcap, "oracle: softcap not restated"
    assert alibi_slopes is None, "oracle: alibi not restated"


def _flash_attn_forward(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    dropout_p: float,
    softmax_scale: float,
    causal: bool,
    window_size_left: int = -1,
    window_size_right: int = -1,
    softcap: float = 0.0,
    alibi_slopes: Optional[torch.Tensor] = None,
    return_softmax: bool = False,
    *,
    rng_state: Optional[torch.Tensor] = None,
):
    """q (B,Sq,H,D), k/v (B,Sk,Hk,D) -> (out (B,Sq,H,D) q.dtype, lse (B,H,Sq) fp32, None, rng_state).
    rng_state (beyond flash_attn's signature): fix the dropout seed instead of drawing one."""
    _check(dropout_p, window_size_left, window_size_right, softcap, alibi_slopes)
    B = q.shape[0]
    outs, lses = [], []
    if dropout_p and rng_state is None:
        rng_state = _new_rng_state()
    for b in range(B):
        o, l = _fwd_one(q[b], k[b], v[b], softmax_scale, causal, (window_size_left, window_size_right),
                        drop=_drop(dropout_p, rng_state, batch=b))
        outs.append(o)
        lses.append(l)
    out = torch.stack(outs).to(q.dtype)
    lse = torch.stack(lses)
    return out, lse, None, (rng_state if dropout_p else None)


def _flash_attn_backward(
    dout: torch.Tensor,
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    out: torch.Tensor,
    softmax_lse: torch.Tensor,
    dq: Optional[torch.Tensor],
    dk: Optional[torch.Tensor],
    dv: Optional[torch.Tensor],
    dropout_p: float,
    softmax_scale: float,
    causal: bool,
    window_size_left: int = -1,
    window_size_right: int = -1,
    softcap: float = 0.0,
    alibi_slopes: Optional[torch.Tensor] = None,
    deterministic: bool = False,
    rng_state: Optional[torch.Tensor] = None,
):
    """Writes dq/dk/dv IN PLACE (views allowed, as the reference passes `dq_buffer[:, :seqlen_q]`,
    zigzag_ring_flash_attn.py:137-139).  Returns softmax_d (B,H,Sq)."""
    _check(dropout_p, window_size_left, window_size_right, softcap, alibi_slopes)
    B = q.shape[0]
    ds = []
    for b in range(B):
        gq, gk, gv = _bwd_one(dout[b], q[b], k[b], v[b], out[b], softmax_lse[b], softmax_scale, causal,
                              window=(window_size_left, window_size_right), drop=_drop(dropout_p, rng_state, batch=b))
        dq[b].copy_(gq.to(dq.dtype))
        dk[b].copy_(gk.to(dk.dtype))
        dv[b].copy_(gv.to(dv.dtype))
        ds.append((dout[b].float() * out[b].float()).sum(-1).transpose(0, 1))
    return torch.stack(ds)


def _flash_attn_varlen_forward(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    cu_seqlens_q: torch.Tensor,
    cu_seqlens_k: torch.Tensor,
    max_seqlen_q: int,
    max_seqlen_k: int,
    dropout_p: float,
    softmax_scale: float,
    causal: bool,
    window_size_left: int = -1,
    window_size_right: int = -1,
    softcap: float = 0.0,
    alibi_slopes: Optional[torch.Tensor] = None,
    return_softmax: bool = False,
    block_table: Optional[torch.Tensor] = None,
    leftpad_k: Optional[torch.Tensor] = None,
    seqused_k: Optional[torch.Tensor] = None,
    zero_tensors: bool = False,
    *,
    rng_state: Optional[torch.Tensor] = None,
):
    """q (Tq,H,D), k/v (Tk,Hk,D) -> (out (Tq,H,D), lse (H,Tq) fp32, None, rng_state).  Dropout positions of packed
    input are the absolute rows of the packed tensors (include/rfa.h)."""
    _check(dropout_p, window_size_left, window_size_right, softcap, alibi_slopes)
    assert block_table is None and leftpad_k is None and seqused_k is None
    Tq, H, D = q.shape
    out = torch.zeros((Tq, H, D), dtype=torch.float32, device=q.device)
    lse = torch.zeros((H, Tq), dtype=torch.float32, device=q.device)
    cq = [int(x) for x in cu_seqlens_q.tolist()]
    ck = [int(x) for x in cu_seqlens_k.tolist()]
    if dropout_p and rng_state is None:
        rng_state = _new_rng_state()
    for i in range(len(cq) - 1):
        o, l = _fwd_one(q[cq[i]:cq[i + 1]], k[ck[i]:ck[i + 1]], v[ck[i]:ck[i + 1]], softmax_scale, causal,
                        (window_size_left, window_size_right), drop=_drop(dropout_p, rng_state, q_pos0=cq[i], k_pos0=ck[i]))
        out[cq[i]:cq[i + 1]] = o
        lse[:, cq[i]:cq[i + 1]] = l
    return out.to(q.dtype), lse, None, (rng_state if dropout_p else None)


def _flash_attn_varlen_backward(
    dout: torch.Tensor,
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    out: torch.Tensor,
    softmax_lse: torch.Tensor,
    dq: Optional[torch.Tensor],
    dk: Optional[torch.Tensor],
    dv: Optional[torch.Tensor],
    cu_seqlens_q: torch.Tensor,
    cu_seqlens_k: torch.Tensor,
    max_seqlen_q: int,
    max_seqlen_k: int,
    dropout_p: float,
    softmax_scale: float,
    causal: bool,
    window_size_left: int = -1,
    window_size_right: int = -1,
    softcap: float = 0.0,
    alibi_slopes: Optional[torch.Tensor] = None,
    deterministic: bool = False,
    rng_state: Optional[torch.Tensor] = None,
    zero_tensors: bool = False,
):
    _check(dropout_p, window_size_left, window_size_right, softcap, alibi_slopes)
    cq = [int(x) for x in cu_seqlens_q.tolist()]
    ck = [int(x) for x in cu_seqlens_k.tolist()]
    for i in range(len(cq) - 1):
        a, b = cq[i], cq[i + 1]
        c, d = ck[i], ck[i + 1]
        gq, gk, gv = _bwd_one(dout[a:b], q[a:b], k[c:d], v[c:d], out[a:b], softmax_lse[:, a:b], softmax_scale, causal,
                              window=(window_size_left, window_size_right), drop=_drop(dropout_p, rng_state, q_pos0=a, k_pos0=c))
        dq[a:b] = gq.to(dq.dtype)
        dk[c:d] = gk.to(dk.dtype)
        dv[c:d] = gv.to(dv.dtype)
    return (dout.float() * out.float()).sum(-1).transpose(0, 1)


# --------------------------------------------------------------------------------------------
# Public single-device functions the reference TESTS use as ground truth
# (`from flash_attn import flash_attn_qkvpacked_func, ...`, test/test_zigzag_ring_flash_attn_func.py:2).
# Independent formulation: plain softmax attention in fp64 with torch autograd.
# --------------------------------------------------------------------------------------------
def full_attention_fp64(q, k, v, causal, softmax_scale=None, window=None):
    """q (B,Sq,H,D) k/v (B,Sk,Hk,D) any float dtype -> out fp64 (B,Sq,H,D), lse fp64 (B,H,Sq).
    Differentiable (autograd) — the independent reference the oracle itself is pinned to."""
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    qd = q.double().permute(0, 2, 1, 3)
    kd = k.double().permute(0, 2, 1, 3).repeat_interleave(H // Hk, dim=1)
    vd = v.double().permute(0, 2, 1, 3).repeat_interleave(H // Hk, dim=1)
    s = torch.matmul(qd, kd.transpose(-1, -2)) * scale
    m = _mask(0, Sq, Sq, Sk, causal, q.device, window)
    if m is not None:
        s = s.masked_fill(~m, float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)
    out = torch.matmul(p, vd).permute(0, 2, 1, 3)
    return out, lse
