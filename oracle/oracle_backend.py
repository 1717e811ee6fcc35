"""oracle/oracle_backend.py — TEST INFRASTRUCTURE ONLY.

An object with the interface of `ring_flash_attn.backend.HipBackend`, implemented on CPU with
the oracle (oracle/flash_attn_ref.py) and the reference's merge / accumulate formulas
(/root/reference/ring_flash_attn/utils.py:40-48; zigzag_ring_flash_attn.py:164-187).  Tests inject
it with `ring_flash_attn._testing.set_backend(OracleBackend())` to run the *schedules* of the
package under gloo without a GPU.  It mimics the reference's rounding points (block results are
rounded to the io dtype before they are merged / accumulated in fp32) so that the schedules can
be compared tightly against golden fixtures produced by the unmodified reference code.
"""
import torch
import torch.nn.functional as F

from . import flash_attn_ref as R

HALF_FULL, HALF_FRONT, HALF_BACK = 0, 1, 2
BWD_ALL, BWD_COMPUTE, BWD_REDUCE = 0, 1, 2


def _span(start, length, half):
    if half == HALF_FRONT:
        return start, length // 2
    if half == HALF_BACK:
        mid = length // 2
        return start + mid, length - mid
    return start, length


def _seqs(t, cu, half):
    """yield (batch_index_or_None, row_start, row_len) for every sequence of t."""
    if cu is None:
        for b in range(t.shape[0]):
            s, l = _span(0, t.shape[1], half)
            yield b, s, l
    else:
        c = [int(x) for x in cu.tolist()]
        for i in range(len(c) - 1):
            s, l = _span(c[i], c[i + 1] - c[i], half)
            yield None, s, l


def _rows(t, b, s, l):
    return t[b, s:s + l] if b is not None else t[s:s + l]


def _lse_rows(t, b, s, l):          # (B,H,S) or (H,T) -> (H,l) view
    return t[b, :, s:s + l] if b is not None else t[:, s:s + l]


def _drop(dropout, bq, qs, ks):
    """HipBackend's dropout=(p, seed, q_pos_offset, k_pos_offset, head_offset) -> the oracle's position dict: dense
    sequences hash (batch, row inside the sequence), packed ones the absolute packed row (include/rfa.h)"""
    if dropout is None or not dropout[0] > 0:
        return None
    p, seed, q0, k0, h0 = dropout
    if bq is not None:
        return dict(p=float(p), seed=int(seed), batch=bq, head0=h0, q_pos0=q0 + qs, k_pos0=k0 + ks)
    return dict(p=float(p), seed=int(seed), batch=0, head0=h0, q_pos0=q0 + qs, k_pos0=k0 + ks)


class OracleBackend:
    name = "oracle"

    # ------------------------------------------------------------------ forward
    def fwd(self, q, k, v, *, softmax_scale, causal, cu_seqlens_q=None, cu_seqlens_k=None,
            max_seqlen_q=None, max_seqlen_k=None, q_half=0, k_half=0, out=None, lse=None,
            out_acc=None, lse_acc=None, acc_init=False, window=(-1, -1), dropout=None):
        for (bq, qs, ql), (bk, ks, kl) in zip(_seqs(q, cu_seqlens_q, q_half), _seqs(k, cu_seqlens_k, k_half)):
            o, l = R._fwd_one(_rows(q, bq, qs, ql), _rows(k, bk, ks, kl), _rows(v, bk, ks, kl), softmax_scale, causal,
                              window, drop=_drop(dropout, bq, qs, ks))
            o = o.to(q.dtype)                       # flash_attn returns out in the io dtype
            if out_acc is None:
                _rows(out, bq, qs, ql).copy_(o)
                _lse_rows(lse, bq, qs, ql).copy_(l)
                continue
            oa = _rows(out_acc, bq, qs, ql)                     # (l,H,D) fp32
            la = _lse_rows(lse_acc, bq, qs, ql)                 # (H,l)
            if acc_init:
                oa.copy_(o.float())
                la.copy_(torch.where(torch.isinf(l) & (l > 0), torch.full_like(l, float("-inf")), l))
                continue
            keep = torch.isinf(l) & (l > 0)                     # rows without keys: untouched
            bl = l.transpose(0, 1).unsqueeze(-1)                # (l,H,1)
            cur = la.transpose(0, 1).unsqueeze(-1)
            new_o = oa - torch.sigmoid(bl - cur) * (oa - o.float())
            new_l = cur - F.logsigmoid(cur - bl)
            k3 = keep.transpose(0, 1).unsqueeze(-1)
            oa.copy_(torch.where(k3, oa, new_o))
            la.copy_(torch.where(keep, la, new_l.squeeze(-1).transpose(0, 1)))

    # ------------------------------------------------------------------ backward
    def bwd_preprocess(self, dout, out, delta, *, cu_seqlens_q=None, max_seqlen_q=None, q_half=0):
        d = (dout.float() * out.float()).sum(-1)                # (B,S,H) / (T,H)
        if cu_seqlens_q is None:
            delta.copy_(d.transpose(1, 2))
        else:
            delta.copy_(d.transpose(0, 1))

    def bwd(self, dout, q, k, v, lse, delta, *, softmax_scale, causal, cu_seqlens_q=None,
            cu_seqlens_k=None, max_seqlen_q=None, max_seqlen_k=None, q_half=0, k_half=0,
            dq=None, dk=None, dv=None, dq_acc=None, dk_acc=None, dv_acc=None, acc_init=False,
            deterministic=False, phases=BWD_ALL, partials=None, ds_scratch=None, window=(-1, -1), dropout=None,
            prof_events=None):
        pairs = list(zip(_seqs(q, cu_seqlens_q, q_half), _seqs(k, cu_seqlens_k, k_half)))
        kv_init = acc_init or bool(phases & 16)          # RFA_BWD_KV_OVERWRITE (include/rfa.h)
        phases &= 3
        if phases in (BWD_ALL, BWD_COMPUTE):
            pend = []
            for (bq, qs, ql), (bk, ks, kl) in pairs:
                gq, gk, gv = R._bwd_one(_rows(dout, bq, qs, ql), _rows(q, bq, qs, ql), _rows(k, bk, ks, kl),
                                        _rows(v, bk, ks, kl), None, _lse_rows(lse, bq, qs, ql), softmax_scale,
                                        causal, delta=_lse_rows(delta, bq, qs, ql), window=window,
                                        drop=_drop(dropout, bq, qs, ks))
                gq, gk, gv = gq.to(q.dtype), gk.to(q.dtype), gv.to(q.dtype)   # flash_attn rounds here
                if dq_acc is not None:
                    t = _rows(dq_acc, bq, qs, ql)
                    t.copy_(gq.float() if acc_init else t + gq.float())
                else:
                    _rows(dq, bq, qs, ql).copy_(gq)
                pend.append((gk, gv))
            if phases == BWD_COMPUTE:
                return pend                      # the token the matching BWD_REDUCE call must be given
            partials = pend
        if phases in (BWD_ALL, BWD_REDUCE):
            if partials is None:
                raise RuntimeError("a BWD_REDUCE call needs the `partials` its BWD_COMPUTE call returned")
            for ((bq, qs, ql), (bk, ks, kl)), (gk, gv) in zip(pairs, partials):
                if dk_acc is not None:
                    tk, tv = _rows(dk_acc, bk, ks, kl), _rows(dv_acc, bk, ks, kl)
                    tk.copy_(gk.float() if kv_init else tk + gk.float())
                    tv.copy_(gv.float() if kv_init else tv + gv.float())
                else:
                    _rows(dk, bk, ks, kl).copy_(gk)
                    _rows(dv, bk, ks, kl).copy_(gv)
        return None

    # ------------------------------------------------------------------ side kernels
    def zero_outside(self, shape, dtype, device, lo, hi, slot=0, dim=0):
        """HipBackend.zero_outside: a buffer whose rows outside [lo, hi) along `dim` are zero (here: a fresh one per call)"""
        return torch.zeros(shape, dtype=dtype, device=device)

    def merge(self, out_acc, lse_acc, block_out, block_lse, *, acc_init=False):
        """lse_acc / block_lse: (B,H,S) views."""
        if acc_init:
            out_acc.copy_(block_out.float())
            lse_acc.copy_(block_lse)
            return
        bl = block_lse.transpose(1, 2).unsqueeze(-1)            # (B,S,H,1)
        cur = lse_acc.transpose(1, 2).unsqueeze(-1)
        new_o = out_acc - torch.sigmoid(bl - cur) * (out_acc - block_out.float())
        new_l = cur - F.logsigmoid(cur - bl)
        out_acc.copy_(new_o)
        lse_acc.copy_(new_l.squeeze(-1).transpose(1, 2))

    def sum_slots(self, src, dst):
        dst.copy_(src.float().sum(dim=0).to(dst.dtype))
        return dst

    def cast(self, src, dtype):
        return src.to(dtype)

    def lse_flatten(self, lse_padded, cu_seqlens):
        c = [int(x) for x in cu_seqlens.tolist()]
        return torch.cat([lse_padded[i, :, : c[i + 1] - c[i]] for i in range(len(c) - 1)], dim=1)

    def lse_unflatten(self, lse_packed, cu_seqlens, max_seqlen):
        if lse_packed.dim() == 3:
            lse_packed = lse_packed.squeeze(-1)
        c = [int(x) for x in cu_seqlens.tolist()]
        out = torch.zeros((len(c) - 1, lse_packed.shape[1], max_seqlen), dtype=lse_packed.dtype)
        for i in range(len(c) - 1):
            out[i, :, : c[i + 1] - c[i]] = lse_packed[c[i]:c[i + 1]].transpose(0, 1)
        return out
