"""Operator backend: the seam between the ring schedules (Python) and the attention kernels.

`HipBackend` is the product: it translates torch tensors into the plain-pointer C ABI of
librfa_hip.so (include/rfa.h) and launches on torch's current HIP stream.  It is the stand-in
for what the reference imports from the CUDA-only `flash_attn` package
(/root/reference/ring_flash_attn/zigzag_ring_flash_attn.py:3, ring_flash_attn_varlen.py:3-6).

There is deliberately NO CPU implementation here.  The test suite injects the CPU oracle
(oracle/flash_attn_ref.py) through ring_flash_attn._testing.set_backend to exercise the *schedules*
under gloo on a GPU-less machine; the product never selects anything but HipBackend, and HipBackend
raises if the extension is missing or a tensor is not on a HIP device.
"""
import ctypes as C
from typing import Optional

import torch

from . import _C, config

HALF_FULL, HALF_FRONT, HALF_BACK = _C.HALF_FULL, _C.HALF_FRONT, _C.HALF_BACK

_DTYPES = {torch.bfloat16: _C.RFA_BF16, torch.float16: _C.RFA_F16}


def _st3(t: torch.Tensor, varlen: bool) -> _C.Strides:
    """(batch,row,head) element strides of (B,S,H,D) or (T,H,D)."""
    if t.stride(-1) != 1:
        raise ValueError("last (head_dim) stride must be 1")
    if varlen:
        return _C.Strides(0, t.stride(0), t.stride(1))
    return _C.Strides(t.stride(0), t.stride(1), t.stride(2))


def _lse_st(t: torch.Tensor, varlen: bool):
    """(batch, head) strides of a (B,H,S) / (H,T) fp32 tensor; row stride must be 1."""
    if t.dtype != torch.float32:
        raise ValueError("lse/delta tensors must be float32")
    if t.stride(-1) != 1 and t.shape[-1] != 1:
        raise ValueError("lse row stride must be 1")
    if varlen:
        return 0, t.stride(0)
    return t.stride(0), t.stride(1)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class HipBackend:
    """MI355X kernels through the C ABI.  All methods enqueue on the current stream and return
    immediately; outputs are caller-allocated unless stated otherwise."""

    name = "hip"

    def __init__(self):
        self.lib = _C.load()
        self._ds_pool = {}            # (device index, stream handle) -> the ONE reusable dS scratch of that stream
        self._ds_capped = {}          # same key -> [backwards left until growth of a memory-capped scratch is retried]

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _check_dev(*ts):
        for t in ts:
            if t is not None and not t.is_cuda:
                raise RuntimeError(
                    "ring_flash_attn: tensors must live on a HIP (cuda) device — there is no CPU "
                    "compute path in this package"
                )

    @staticmethod
    def _dtype(t):
        try:
            return _DTYPES[t.dtype]
        except KeyError:
            raise TypeError(f"ring_flash_attn: unsupported dtype {t.dtype} (bf16/fp16 only)") from None

    # ------------------------------------------------------------------ forward
    def fwd(self, q, k, v, *, softmax_scale, causal, cu_seqlens_q=None, cu_seqlens_k=None,
            max_seqlen_q=None, max_seqlen_k=None, q_half=HALF_FULL, k_half=HALF_FULL,
            out=None, lse=None, out_acc=None, lse_acc=None, acc_init=False, window=(-1, -1), dropout=None):
        """Block attention.  Plain mode fills (out, lse); accumulate mode merges into the fp32
        (out_acc, lse_acc) pair (fused update_out_and_lse).  Dense: q (B,Sq,H,D); varlen: (T,H,D).
        dropout: (p, seed, q_pos_offset, k_pos_offset, head_offset) or None."""
        self._check_dev(q, k, v, out, lse, out_acc, lse_acc)
        varlen = cu_seqlens_q is not None
        a = _C.FwdArgs()
        a.q, a.k, a.v = _ptr(q), _ptr(k), _ptr(v)
        a.q_st, a.k_st, a.v_st = _st3(q, varlen), _st3(k, varlen), _st3(v, varlen)
        if out_acc is not None:
            a.out_acc, a.out_acc_st = _ptr(out_acc), _st3(out_acc, varlen)
            a.lse_acc = _ptr(lse_acc)
            a.lse_acc_batch, a.lse_acc_head = _lse_st(lse_acc, varlen)
            a.acc_init = 1 if acc_init else 0
        else:
            a.out, a.out_st = _ptr(out), _st3(out, varlen)
            a.lse = _ptr(lse)
            a.lse_batch, a.lse_head = _lse_st(lse, varlen)
        if varlen:
            a.cu_seqlens_q, a.cu_seqlens_k = _ptr(cu_seqlens_q), _ptr(cu_seqlens_k)
            a.B = cu_seqlens_q.numel() - 1
            a.H, a.D = q.shape[1], q.shape[2]
            a.Hk = k.shape[1]
            a.Sq, a.Sk = int(max_seqlen_q), int(max_seqlen_k)
            a.total_q = q.shape[0]
        else:
            a.B, a.Sq, a.H, a.D = q.shape
            a.Sk, a.Hk = k.shape[1], k.shape[2]
        a.q_half, a.k_half = q_half, k_half
        a.softmax_scale = float(softmax_scale)
        a.causal = 1 if causal else 0
        if window is not None and (window[0] >= 0 or window[1] >= 0):
            a.window, a.window_left, a.window_right = 1, int(window[0]), int(window[1])
        a.dtype = self._dtype(q)
        _set_dropout(a, dropout)
        a.fwd_form = _fwd_form()
        a.kv_nsplit = config.get().fwd_kv_nsplit
        # split-KV launches (few query rows against many keys on an under-filled grid: include/rfa.h) need a small
        # workspace for the partial (out, lse) pairs: a few tens of MB, only for the calls that split
        ws = None
        nbytes = self.lib.rfa_fwd_workspace_bytes(C.byref(a), None)
        if nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
            a.workspace = ws.data_ptr()
        _C.check(self.lib.rfa_fwd(C.byref(a), _stream(q)), "rfa_fwd")

    # ------------------------------------------------------------------ backward
    def bwd_preprocess(self, dout, out, delta, *, cu_seqlens_q=None, max_seqlen_q=None, q_half=HALF_FULL):
        """delta[b,h,i] = sum_d dout*out, fp32, laid out like lse."""
        self._check_dev(dout, out, delta)
        varlen = cu_seqlens_q is not None
        a = _C.BwdPreArgs()
        a.dout, a.out, a.delta = _ptr(dout), _ptr(out), _ptr(delta)
        a.dout_st, a.out_st = _st3(dout, varlen), _st3(out, varlen)
        a.delta_batch, a.delta_head = _lse_st(delta, varlen)
        if varlen:
            a.cu_seqlens_q = _ptr(cu_seqlens_q)
            a.B = cu_seqlens_q.numel() - 1
            a.H, a.D = dout.shape[1], dout.shape[2]
            a.Sq = int(max_seqlen_q)
        else:
            a.B, a.Sq, a.H, a.D = dout.shape
        a.q_half = q_half
        a.dtype = self._dtype(dout)
        _C.check(self.lib.rfa_bwd_preprocess(C.byref(a), _stream(dout)), "rfa_bwd_preprocess")

    def bwd(self, dout, q, k, v, lse, delta, *, softmax_scale, causal, cu_seqlens_q=None,
            cu_seqlens_k=None, max_seqlen_q=None, max_seqlen_k=None, q_half=HALF_FULL,
            k_half=HALF_FULL, dq=None, dk=None, dv=None, dq_acc=None, dk_acc=None, dv_acc=None,
            acc_init=False, deterministic=False, phases=_C.BWD_ALL, partials=None, ds_scratch=None,
            window=(-1, -1), prof_events=None, dropout=None):
        """dQ/dK/dV of one block.  Plain outputs (io dtype) or `+=` into fp32 accumulators.
        phases=BWD_COMPUTE / BWD_REDUCE splits the call so a ring step can overlap the kernels
        with the arrival of the dk/dv accumulators it adds into: the COMPUTE call RETURNS the buffer
        holding its dK/dV partials and the REDUCE call must be handed exactly that buffer (`partials=`),
        so interleaved backwards (other streams, re-entrant / checkpointed autograd, pipeline
        micro-batches) can never consume each other's partials.  Scratch comes from torch's caching
        allocator on the current stream, per call; the C library itself never allocates."""
        self._check_dev(dout, q, k, v, lse, delta, dq, dk, dv, dq_acc, dk_acc, dv_acc)
        varlen = cu_seqlens_q is not None
        a = _C.BwdArgs()
        a.dout, a.q, a.k, a.v = _ptr(dout), _ptr(q), _ptr(k), _ptr(v)
        a.dout_st, a.q_st = _st3(dout, varlen), _st3(q, varlen)
        a.k_st, a.v_st = _st3(k, varlen), _st3(v, varlen)
        a.lse = _ptr(lse)
        a.lse_batch, a.lse_head = _lse_st(lse, varlen)
        a.delta = _ptr(delta)
        a.delta_batch, a.delta_head = _lse_st(delta, varlen)
        if dq_acc is not None:
            a.dq_acc, a.dq_acc_st = _ptr(dq_acc), _st3(dq_acc, varlen)
        else:
            a.dq, a.dq_st = _ptr(dq), _st3(dq, varlen)
        if dk_acc is not None:
            a.dk_acc, a.dv_acc = _ptr(dk_acc), _ptr(dv_acc)
            a.dk_acc_st, a.dv_acc_st = _st3(dk_acc, varlen), _st3(dv_acc, varlen)
        else:
            a.dk, a.dv = _ptr(dk), _ptr(dv)
            a.dk_st, a.dv_st = _st3(dk, varlen), _st3(dv, varlen)
        a.acc_init = 1 if acc_init else 0
        if varlen:
            a.cu_seqlens_q, a.cu_seqlens_k = _ptr(cu_seqlens_q), _ptr(cu_seqlens_k)
            a.B = cu_seqlens_q.numel() - 1
            a.H, a.D = q.shape[1], q.shape[2]
            a.Hk = k.shape[1]
            a.Sq, a.Sk = int(max_seqlen_q), int(max_seqlen_k)
            a.total_k, a.total_q = k.shape[0], q.shape[0]
        else:
            a.B, a.Sq, a.H, a.D = q.shape
            a.Sk, a.Hk = k.shape[1], k.shape[2]
            a.total_k = a.B * a.Sk
        a.q_half, a.k_half = q_half, k_half
        a.softmax_scale = float(softmax_scale)
        a.causal = 1 if causal else 0
        a.deterministic = 1 if deterministic else 0
        if window is not None and (window[0] >= 0 or window[1] >= 0):
            a.window, a.window_left, a.window_right = 1, int(window[0]), int(window[1])
        a.dtype = self._dtype(q)
        a.phases = phases
        _set_dropout(a, dropout)
        if prof_events is not None:       # measurement (bench.py): a ctypes array of 4 hipEvent_t, see include/rfa.h
            a.prof_events = prof_events
        reduce_only = bool(phases & _C.BWD_REDUCE) and not (phases & _C.BWD_COMPUTE)
        # dK/dV launch plan: part of the call (ABI 4).  The tuning / test overrides (config.dkdv_wide, config.dkdv_nsplit)
        # are applied HERE, once per backward; a REDUCE call is given the plan its COMPUTE call RESOLVED (it travels with
        # the `partials` token), so the two phases cannot disagree whatever the configuration, or the chunking of the
        # dS hand-off, does in between.
        if reduce_only and partials is not None and hasattr(partials, "_rfa_plan"):
            a.dkdv_form, a.dkdv_nsplit = partials._rfa_plan
        else:
            a.dkdv_form, a.dkdv_nsplit = _plan_overrides()
        # 5-GEMM backward (csrc/rfa_dqs.hip): the dK/dV kernel hands dS to the dQ kernel through a scratch instead of dQ
        # recomputing S and dP, when the call is eligible (D == 128 or 256, whole sequences, dense or packed).  The
        # scratch is ONE reusable buffer per device and stream (bwd_ds_scratch below), at most config.ds_spill_max_bytes
        # large: a hand-off that does not fit runs in head-group chunks over it (include/rfa.h: ds_scratch_bytes).
        # config.bwd_ds_spill = False (RFA_BWD_DS_SPILL=0) keeps the 7-GEMM form.  Callers that split one backward over
        # several calls (measurement: BWD_SKIP_DQ / BWD_SKIP_DKDV) pass the same `ds_scratch` to both.
        if ds_scratch is None and not reduce_only and _spill_enabled():
            ds_scratch = self.bwd_ds_scratch(a, q.device)
        if ds_scratch is not None and not reduce_only:
            a.ds_scratch = ds_scratch.data_ptr()
            a.ds_scratch_bytes = ds_scratch.numel() * ds_scratch.element_size()
        nbytes = self.lib.rfa_bwd_workspace_bytes(C.byref(a))
        ws = None
        if reduce_only:
            if partials is None:
                raise RuntimeError("rfa_bwd: a BWD_REDUCE call needs the `partials` buffer its BWD_COMPUTE call returned")
            if partials.device != q.device or partials.numel() < nbytes:
                raise RuntimeError("rfa_bwd: `partials` does not belong to this call (device / size mismatch)")
            ws = partials
        elif nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
            form, ns, _ = self.bwd_plan(a)           # the RESOLVED plan (with this call's scratch): what a REDUCE call must repeat
            ws._rfa_plan = (form, ns if form == _C.DKDV_256 or a.D > 128 else 0)
        if ws is not None:
            a.workspace = ws.data_ptr()
        _C.check(self.lib.rfa_bwd(C.byref(a), _stream(q)), "rfa_bwd")
        return ws if (phases & _C.BWD_COMPUTE) else None

    def bwd_plan(self, a):
        """(form, nsplit, five_gemm) the call described by `a` (a filled BwdArgs) will run"""
        form, ns, five = C.c_int32(), C.c_int32(), C.c_int32()
        _C.check(self.lib.rfa_bwd_plan(C.byref(a), C.byref(form), C.byref(ns), C.byref(five)), "rfa_bwd_plan")
        return form.value, ns.value, five.value

    def bwd_ds_chunks(self, a):
        """(nchunks, kv_heads, q_heads_per_kv_head, chunk_bytes) of the dS hand-off of the call described by `a`
        (nchunks 0: the 7-GEMM form)"""
        n, hc, gc, cb = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        _C.check(self.lib.rfa_bwd_ds_chunks(C.byref(a), C.byref(n), C.byref(hc), C.byref(gc), C.byref(cb)), "rfa_bwd_ds_chunks")
        return n.value, hc.value, gc.value, cb.value

    def bwd_ds_scratch(self, a, device):
        """The dS scratch for the call described by `a` (a filled BwdArgs), or None: the call then runs the 7-GEMM form.

        ONE buffer per (device, stream), kept and reused by every backward on that stream (kernels of one stream run in
        order, so the next backward's dK/dV kernel cannot start before this one's dQ kernel has read the blocks) —
        no 2 GB allocation per backward, and peak memory independent of the head count and the sequence length: the
        buffer is at most config.ds_spill_max_bytes (default 4.5 GiB; the headline's hand-off is 2.0 GiB and takes just that) and,
        when it is first taken or has to grow, at most config.ds_spill_max_frac (0.5) of the memory free at that moment;
        a hand-off that does not fit runs in head-group chunks over it (include/rfa.h: ds_scratch_bytes; long contexts
        keep the 5-GEMM form — measured at S = 32768, 32 heads, 34 GB of dS: 942 TFLOP/s in 8 chunks of 4 query heads
        against 950 with the whole hand-off resident and 895 for the 7-GEMM form; chunks of 2 query heads — a 2.5 GiB
        limit — give the dQ kernel only one workgroup per CU and fall to 848, which is why the default is not smaller).  A hand-off whose smallest chunk does not fit, or an allocation failure, falls back to
        the 7-GEMM form instead of failing a backward that would have fitted without it (logged once).
        `release_scratch()` returns the buffers."""
        full = self.lib.rfa_bwd_ds_scratch_bytes(C.byref(a))
        if full <= 0:
            return None
        least = self.lib.rfa_bwd_ds_scratch_min_bytes(C.byref(a))
        want = min(full, max(_spill_limit(), 0))
        capped = False
        key = (device.index if device.index is not None else torch.cuda.current_device(),
               torch.cuda.current_stream(device).cuda_stream) if device.type == "cuda" else ("cpu", 0)
        buf = self._ds_pool.get(key)
        if buf is not None:
            if buf.numel() >= min(want, full):
                return buf                                  # (the steady state: no query, no allocation)
            # a buffer that was CAPPED by the free-memory fraction when it was taken is smaller than `want` for ever:
            # keep using it (the hand-off runs in chunks over it) instead of querying the allocator, dropping and
            # re-making it in every backward; growth is retried once in a while
            retry = self._ds_capped.get(key)
            if retry is not None and buf.numel() >= least:
                retry[0] -= 1
                if retry[0] > 0:
                    return buf
        if want < least:
            _log_once("spill-limit", f"ring_flash_attn: the dS hand-off needs at least {least / 2**30:.2f} GiB per chunk "
                                     f"(limit {want / 2**30:.2f} GiB): this backward runs the 7-GEMM form")
            return None
        if device.type == "cuda" and want > _SPILL_CHECK_ABOVE:
            free, _ = torch.cuda.mem_get_info(device)
            cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
            have = buf.numel() if buf is not None else 0
            cap = int(_spill_frac() * (free + cached + have))
            if cap < want:
                want = cap
                capped = True
        if buf is not None and buf.numel() >= least and want <= buf.numel():
            # a growth retry that would not GROW the pooled buffer (memory is as tight as before, or tighter): keep what
            # works — never trade a usable scratch for a smaller one or for the 7-GEMM form — and try again later
            self._ds_capped[key] = [_SPILL_GROW_RETRY]
            return buf
        if want < least:
            _log_once("spill-mem", f"ring_flash_attn: not enough free memory for a dS hand-off chunk ({least / 2**30:.2f} GiB): "
                                   "this backward runs the 7-GEMM form")
            return None
        self._ds_pool.pop(key, None)
        buf = None
        try:
            buf = torch.empty(want, dtype=torch.uint8, device=device)
        except torch.OutOfMemoryError:
            _log_once("spill-oom", f"ring_flash_attn: a dS hand-off scratch of {want / 2**30:.2f} GiB could not be allocated: "
                                   "this backward runs the 7-GEMM form")
            return None
        self._ds_pool[key] = buf
        if capped:
            self._ds_capped[key] = [_SPILL_GROW_RETRY]
        else:
            self._ds_capped.pop(key, None)
        return buf

    def zero_outside(self, shape, dtype, device, lo, hi, slot=0, dim=0):
        """A reusable buffer of `shape` whose rows OUTSIDE [lo, hi) along `dim` are zero — the dK/dV contribution buffer of
        the llama3 backward, whose rows inside the range the dK/dV kernel overwrites completely on every call (key blocks
        without a visible query store zeros) and whose rows outside must read as zero in the reduce-scatter.  Rounds 2-5
        allocated it per backward and zero-filled the outside rows with torch fills (7 us launches on a
        0.5 ms step); now ONE buffer per (device, stream, shape, slot) is zeroed when it is made and only the rows that LEAVE
        the range between two calls are zeroed again (a packed batch whose local key slice shrinks).  Safe to reuse: every
        caller waits for the collective that reads the buffer before it returns."""
        cuda = device.type == "cuda"
        key = ("zero", (device.index if device.index is not None else torch.cuda.current_device()) if cuda else -1,
               torch.cuda.current_stream(device).cuda_stream if cuda else 0, tuple(shape), dtype, slot, dim)
        ent = self._ds_pool.get(key)
        if ent is None:
            buf = torch.zeros(shape, dtype=dtype, device=device)
            self._ds_pool[key] = [buf, lo, hi]
            return buf
        buf, plo, phi = ent
        if lo > plo:                                   # rows [plo, lo) were inside, now outside
            buf.narrow(dim, plo, min(lo, phi) - plo).zero_()
        if hi < phi:
            start = max(hi, plo)
            buf.narrow(dim, start, phi - start).zero_()
        ent[1], ent[2] = lo, hi
        return buf

    def release_scratch(self):
        """drop the reusable dS scratch buffers (they are re-made on demand): up to config.ds_spill_max_bytes per device
        and stream that the pool otherwise holds for the life of the process — call it before a phase that needs the
        memory (evaluation with long sequences, checkpoint loading), or from an out-of-memory handler"""
        self._ds_pool.clear()
        self._ds_capped.clear()

    # ------------------------------------------------------------------ side kernels
    def merge(self, out_acc, lse_acc, block_out, block_lse, *, acc_init=False):
        """Stand-alone (out, lse) merge.  out_acc/block_out: (B,S,H,D) views; lse_acc/block_lse:
        (B,H,S) views of ANY strides (so the reference's (B,S,H,1) running lse works too)."""
        self._check_dev(out_acc, lse_acc, block_out, block_lse)
        if lse_acc.dtype != torch.float32 or block_lse.dtype != torch.float32:
            raise ValueError("lse tensors must be float32")
        a = _C.MergeArgs()
        a.out_acc, a.out_acc_st = _ptr(out_acc), _st3(out_acc, False)
        a.lse_acc = _ptr(lse_acc)
        a.lse_acc_batch, a.lse_acc_head, a.lse_acc_row = lse_acc.stride()
        a.block_out, a.block_out_st = _ptr(block_out), _st3(block_out, False)
        a.block_lse = _ptr(block_lse)
        a.block_lse_batch, a.block_lse_head, a.block_lse_row = block_lse.stride()
        a.B, a.S, a.H, a.D = block_out.shape
        a.acc_init = 1 if acc_init else 0
        a.dtype = self._dtype(block_out)
        _C.check(self.lib.rfa_merge(C.byref(a), _stream(block_out)), "rfa_merge")

    def sum_slots(self, src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
        """dst[...] = sum over dim 0 of src (io dtype, summed in fp32).  src: (W, B, S, H, D) or (W, T, H, D), each slot
        laid out like dst up to strides; dst may be a strided view (a slice of a packed gradient)."""
        self._check_dev(src, dst)
        varlen = dst.dim() == 3
        a = _C.SumSlotsArgs()
        a.src, a.dst = _ptr(src), _ptr(dst)
        a.nslots, a.slot_stride = src.shape[0], src.stride(0)
        a.src_st, a.dst_st = _st3(src[0], varlen), _st3(dst, varlen)
        if varlen:
            a.B, (a.S, a.H, a.D) = 1, dst.shape
        else:
            a.B, a.S, a.H, a.D = dst.shape
        a.dtype = self._dtype(dst)
        _C.check(self.lib.rfa_sum_slots(C.byref(a), _stream(dst)), "rfa_sum_slots")
        return dst

    def cast(self, src: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """fp32 accumulator -> io dtype (new contiguous tensor)."""
        self._check_dev(src)
        if not src.is_contiguous():
            src = src.contiguous()
        dst = torch.empty(src.shape, dtype=dtype, device=src.device)
        _C.check(self.lib.rfa_cast(dst.data_ptr(), src.data_ptr(), src.numel(), _DTYPES[dtype], _stream(src)),
                 "rfa_cast")
        return dst

    def lse_flatten(self, lse_padded, cu_seqlens):
        """(B,H,max_seqlen) -> (H,T)   (triton_utils.flatten_varlen_lse)."""
        self._check_dev(lse_padded, cu_seqlens)
        B, H, M = lse_padded.shape
        T = int(cu_seqlens[-1].item())
        src = lse_padded.contiguous()
        dst = torch.empty((H, T), dtype=torch.float32, device=src.device)
        _C.check(self.lib.rfa_lse_flatten(dst.data_ptr(), src.data_ptr(), cu_seqlens.data_ptr(), B, H, M,
                                          dst.stride(0), dst.stride(1), _stream(src)), "rfa_lse_flatten")
        return dst

    def lse_unflatten(self, lse_packed, cu_seqlens, max_seqlen):
        """(T,H,1)/(T,H) -> (B,H,max_seqlen)   (triton_utils.unflatten_varlen_lse)."""
        self._check_dev(lse_packed, cu_seqlens)
        if lse_packed.dim() == 3:
            lse_packed = lse_packed.squeeze(-1)
        T, H = lse_packed.shape
        B = cu_seqlens.numel() - 1
        dst = torch.empty((B, H, max_seqlen), dtype=torch.float32, device=lse_packed.device)
        _C.check(self.lib.rfa_lse_unflatten(dst.data_ptr(), lse_packed.data_ptr(), cu_seqlens.data_ptr(), B, H,
                                            int(max_seqlen), lse_packed.stride(1), lse_packed.stride(0),
                                            _stream(lse_packed)), "rfa_lse_unflatten")
        return dst


def _set_dropout(a, dropout):
    """dropout = (p, seed, q_pos_offset, k_pos_offset, head_offset) or None (include/rfa.h: rfa_fwd_args.dropout_p)"""
    if dropout is None or not dropout[0] > 0:
        return
    p, seed, q0, k0, h0 = dropout
    a.dropout_p, a.dropout_seed = float(p), int(seed) & 0xFFFFFFFFFFFFFFFF
    a.q_pos_offset, a.k_pos_offset, a.head_offset = int(q0), int(k0), int(h0)


_FWD_FORMS = {"auto": _C.FWD_AUTO, "8x32": _C.FWD_8x32, "4x32": _C.FWD_4x32, "p8x32": _C.FWD_P8x32}


def _fwd_form() -> int:
    """config.fwd_form (RFA_FWD_FORM = 8x32 | 4x32, tuning / tests); auto: the library's choice"""
    return _FWD_FORMS[config.get().fwd_form]


def _spill_enabled() -> bool:
    return config.get().bwd_ds_spill


def _spill_limit() -> int:
    return config.get().ds_spill_max_bytes


def _spill_frac() -> float:
    return config.get().ds_spill_max_frac


def _plan_overrides():
    """(dkdv_form, dkdv_nsplit) from the tuning / test switches config.dkdv_wide (RFA_DKDV_WIDE=0|1|2; 2 = the balanced causal schedule) and
    config.dkdv_nsplit (RFA_DKDV_NSPLIT=n; n > 0 also forces the 256-key form); (AUTO, 0) when unset"""
    c = config.get()
    form = _C.DKDV_AUTO
    if c.dkdv_wide == 0:
        form = _C.DKDV_128
    elif c.dkdv_wide == 2 and c.dkdv_nsplit <= 0:
        form = _C.DKDV_BAL               # (the balanced causal schedule where the call is eligible, else the library's choice)
    elif c.dkdv_nsplit > 0 or c.dkdv_wide >= 1:
        form = _C.DKDV_256               # (forced: the C plan takes form == 256 as given; nsplit 0 = its own choice)
    return form, c.dkdv_nsplit


_SPILL_CHECK_ABOVE = 256 << 20      # bytes: scratch requests up to this size are simply attempted
_SPILL_GROW_RETRY = 64               # backwards between two attempts to grow a scratch that free memory had capped

_LOGGED = set()


def _log_once(key, msg):
    if key not in _LOGGED:
        _LOGGED.add(key)
        import sys
        sys.stderr.write(msg + "\n")


_backend = None


def get_backend():
    """the operator backend of this process: the HIP library (raises if librfa_hip.so is missing — there is no CPU
    path).  Tests and bench.py's in-step timer replace it through ring_flash_attn._testing.set_backend."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend
