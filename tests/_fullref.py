"""Full-tensor fp64 attention forward + backward in plain PyTorch, blocked over query rows — TEST INFRASTRUCTURE ONLY.

The CPU oracle (oracle/flash_attn_ref.py) finishes one K/V head group of the 8192-token headline in seconds, but not
of the 65536-token world-size-8 configuration.  This module is the same mathematics written once more, on whatever
device the caller's tensors live on (the GPU box's own device for the big case: rocBLAS fp64 GEMMs + torch's
exp / logsumexp — nothing of this repository's kernels), so that a WHOLE kv-head group — every query row, every
key row — of the full-size launches has a reference that does not consume anything the kernels produced
(its own lse, its own out, its own delta).  tests/test_oracle.py pins it to the CPU oracle on small shapes.

Math (flash_attn semantics, /root/reference call sites zigzag_ring_flash_attn.py:52,156): bottom-right aligned causal
mask, P = softmax(scale Q K^T), out = P V, delta = rowsum(dO ∘ out), dS = P ∘ (dO V^T − delta),
dQ = scale dS K, dK = scale dS^T Q, dV = P^T dO; GQA: dK / dV summed over the query heads of a K/V head.

One property of the ALGORITHM (flash_attn's and this library's alike) needs an allowance when dQ is compared with exact
mathematics: delta is computed from the saved output, which is rounded to the io dtype.  An error e_i in delta_i moves
dQ_i by  -e_i scale sum_j P_ij K_j  (exactly zero sensitivity for rows whose probability-weighted mean key is small —
all late rows of a long sequence — and O(1) for the first rows, which see a handful of keys).  `dq_delta_allowance`
bounds that term per row:  2^-8 ||dO_i ∘ out_i||_2  (about 6 sigma of the rounding noise of out_i: half an ulp, 2^-9
relative, uniformly distributed, per element)  times  scale ||sum_j P_ij K_j||_2."""
import torch


def attention_fwd_bwd_fp64(q, k, v, do, causal=True, rows_per_block=2048):
    """q, do: (Sq, H, D); k, v: (Sk, Hk, D), any float dtype, one device.  Returns fp64 (out (Sq,H,D), lse (H,Sq),
    dq (Sq,H,D), dk (Sk,Hk,D), dv (Sk,Hk,D)); rows without a visible key: out = 0, lse = +inf, no gradient.
    attention_fwd_bwd_fp64.dq_delta_allowance (Sq, H) of the LAST call: see the module docstring."""
    Sq, H, D = q.shape
    Sk, Hk, _ = k.shape
    g = H // Hk
    scale = D ** -0.5
    off = Sk - Sq
    qd, kd, vd, dod = (t.double() for t in (q, k, v, do))
    out = torch.zeros_like(qd)
    lse = torch.full((H, Sq), float("inf"), dtype=torch.float64, device=q.device)
    dq = torch.zeros_like(qd)
    dk, dv = torch.zeros_like(kd), torch.zeros_like(vd)
    allow = torch.zeros((Sq, H), dtype=torch.float64, device=q.device)
    kcol = torch.arange(Sk, device=q.device).view(1, -1)
    for hk in range(Hk):
        K, V = kd[:, hk], vd[:, hk]                                    # (Sk, D)
        for h in range(hk * g, (hk + 1) * g):
            for r0 in range(0, Sq, rows_per_block):
                r1 = min(Sq, r0 + rows_per_block)
                kend = min(Sk, r1 + off) if causal else Sk             # keys any row of the block can see
                if kend <= 0:
                    continue
                Kb, Vb = K[:kend], V[:kend]
                s = (qd[r0:r1, h] @ Kb.T) * scale                       # (rows, kend)
                if causal:
                    rows = torch.arange(r0, r1, device=q.device).view(-1, 1)
                    s = s.masked_fill(kcol[:, :kend] > rows + off, float("-inf"))
                l = torch.logsumexp(s, dim=1)                          # -inf for a row without a visible key
                seen = torch.isfinite(l)
                p = torch.exp(s - torch.where(seen, l, torch.zeros_like(l)).unsqueeze(1))
                p = torch.where(seen.unsqueeze(1), p, torch.zeros_like(p))
                o = p @ Vb
                out[r0:r1, h] = o
                lse[h, r0:r1] = torch.where(seen, l, torch.full_like(l, float("inf")))
                dob = dod[r0:r1, h]
                dp = dob @ Vb.T
                delta = (dob * o).sum(dim=1, keepdim=True)
                ds = p * (dp - delta) * scale
                dq[r0:r1, h] = ds @ Kb
                allow[r0:r1, h] = 2.0 ** -8 * (dob * o).norm(dim=1) * scale * (p @ Kb).norm(dim=1)
                dk[:kend, hk] += ds.T @ qd[r0:r1, h]
                dv[:kend, hk] += p.T @ dob
    attention_fwd_bwd_fp64.dq_delta_allowance = allow
    return out, lse, dq, dk, dv
