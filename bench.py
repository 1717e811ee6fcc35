#!/usr/bin/env python3
"""bench.py — headline benchmark: iters/sec of `zigzag_ring_flash_attn_kvpacked_func` forward+backward.

Workload (BASELINE.json metric; mirrors /root/reference/benchmark/benchmark_kvpacked_func.py:13-126):
per rank q = randn(1, 8192, 32, 128), kv = randn(1, 8192, 2, Hk, 128), dout like q, bf16, N(0,1),
seed 42+rank, causal; the data IS the local zigzag shard, total sequence = 8192 * world_size.
A "step" = one forward + one backward of that operator on every rank (grad reset each step).

    python bench.py --gpus 1 --steps K --warmup W                      # single GPU
    python bench.py --gpus N --steps K --warmup W                      # launches its own N ranks (torch.distributed.run)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W   # or started as ranks

Prints ONE JSON line (rank 0).  `value` = iterations/s of the whole job (max time over ranks).
At every N the line carries
  * `roofline`   the dominant kernel (largest device time per launch) against the bf16 MFMA peak, timed with
                 HIP events INSIDE the real step at N = 1 (`kernels_in_step`: events between the launches of the
                 public function's own call sequence; their sum must stay below `ms_per_step`), and on this
                 rank's step-0 (local causal) block on its own at N > 1 — the same launch at every world size;
  * `comm`       (N > 1) exchange form, bytes per rank per iteration, and `exposed_ms` = measured step minus
                 the same rank-local kernel sequence with the exchange looped back to local buffers
                 (ring_flash_attn._testing.set_loopback), max over ranks;
  * `cpu_baseline` the CPU path on a bounded sample of the same workload on this box's host cores: N = 1 the oracle's
                 forward / backward ("port"); N > 1 the zigzag schedule over N gloo CPU processes with the oracle
                 underneath (oracle/cpu_ring_baseline.py: the unmodified reference schedule where /root/reference
                 exists — "reference" — else this repository's ring-form schedule — "port"), at the largest total
                 sequence the host finishes within --cpu-baseline-budget-s, stated in `sample`.
Other rows of the reference's benchmark tables (README.md:82-98):
  --forward-only                      benchmark_kvpacked_func.py:85-96 (`forward_only`, under torch.no_grad())
  --workload ring | stripe            the other dense schedules of benchmark_kvpacked_func.py:142-147
  --workload ring_varlen | zigzag_varlen | llama3
                                      benchmark/benchmark_varlen_kvpacked_func.py:14-187: packed sequences, 8192 tokens
                                      per rank, 4 cu_seqlens patterns cycled, llama3 with heads_k_stride 4 —
                                      BASELINE.json configs[3]/[4] family.
"""
import argparse
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time

# the host driver only supports dmabuf IPC; RCCL / cross-process device memory need this (see task env)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

SEQ, HEADS, HEAD_DIM = 8192, 32, 128
MFMA_PEAK_TFLOPS = 2500.0          # dense bf16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
# local cu_seqlens patterns of the reference's varlen benchmark (benchmark_varlen_kvpacked_func.py:54-61)
VARLEN_PATTERNS = [[0, 8192], [0, 256, 7648, 8192], [0, 4096, 8192], [0, 3104, 6304, 7904, 8064, 8192]]
LLAMA3_HEADS_K_STRIDE = 4          # benchmark_varlen_kvpacked_func.py:132


def causal_fwd_flops(lengths):
    """flash-attention convention, causal = half: sum over sequences of 4*H*L^2*D/2"""
    return sum(4.0 * HEADS * float(L) * float(L) * HEAD_DIM / 2.0 for L in lengths)


DENSE = ("zigzag", "ring", "stripe")


def fwd_flops_per_gpu(workload, world):
    if workload in DENSE:            # (ring / stripe: the average over ranks — their per-rank work is not balanced)
        return causal_fwd_flops([SEQ * world]) / world
    tot = 0.0
    for cu in VARLEN_PATTERNS:      # global sequence lengths = local lengths * world (both varlen workloads)
        tot += causal_fwd_flops([(b - a) * world for a, b in zip(cu[:-1], cu[1:])]) / world
    return tot / len(VARLEN_PATTERNS)


def time_kernel(fn, iters=10, warm=2):
    """average device time of fn() in ms — events on the stream the kernels are launched on."""
    for _ in range(warm):
        fn()
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_breakdown(q, kv, dout, cu=None):
    """per-kernel device time of this rank's local causal block through the backend (the launches the operator
    makes at step 0 of every world size; dense (1,S,H,D) or packed (T,H,D) with cu_seqlens)."""
    from ring_flash_attn import _C
    from ring_flash_attn.backend import get_backend

    be = get_backend()
    varlen = cu is not None
    if varlen:
        k, v = kv[:, 0], kv[:, 1]
        T, H, D = q.shape
        lse = torch.empty((H, T), dtype=torch.float32, device=q.device)
        mx = int((cu[1:] - cu[:-1]).max().item())
        vl = dict(cu_seqlens_q=cu, cu_seqlens_k=cu, max_seqlen_q=mx, max_seqlen_k=mx)
        pre = dict(cu_seqlens_q=cu, max_seqlen_q=mx)
    else:
        k, v = kv[:, :, 0], kv[:, :, 1]
        B, S, H, D = q.shape
        lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
        vl, pre = {}, {}
    scale = D ** -0.5
    out = torch.empty_like(q)
    delta = torch.empty_like(lse)
    dq = torch.empty_like(q)
    dkc, dvc = torch.empty(k.shape, dtype=k.dtype, device=q.device), torch.empty(v.shape, dtype=v.dtype, device=q.device)
    t = {}
    t["fwd"] = time_kernel(lambda: be.fwd(q, k, v, softmax_scale=scale, causal=True, out=out, lse=lse, **vl))
    t["bwd_preprocess"] = time_kernel(lambda: be.bwd_preprocess(dout, out, delta, **pre))
    common = dict(softmax_scale=scale, causal=True, dq=dq, dk=dkc, dv=dvc, **vl)
    t["bwd_dq"] = time_kernel(lambda: be.bwd(dout, q, k, v, lse, delta, phases=_C.BWD_COMPUTE | _C.BWD_SKIP_DKDV, **common))
    t["bwd_dkdv"] = time_kernel(lambda: be.bwd(dout, q, k, v, lse, delta, phases=_C.BWD_COMPUTE | _C.BWD_SKIP_DQ, **common))
    return t


def library_digest():
    """rfa_build_id() of the librfa_hip.so this process loads (the digest of its sources, compiled into the binary):
    profiles/*_traffic.json carries the id of the library its counters were collected on — read from that library on
    the GPU box at collection time by profiles/collect_pmc.sh — and is only quoted for the same build"""
    from ring_flash_attn import _C

    return _C.load().rfa_build_id().decode()


def committed_traffic(kernel, hk):
    """HBM bytes per launch from the committed PMC pass (counters cannot be collected inside the timed process).
    Returns (entry or None, note)."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for n in sorted(os.listdir(pdir)):
        if n.endswith("_traffic.json"):
            best = n
    if best is None or hk not in (8, 32):
        return None, "no PMC traffic pass committed for this configuration"
    tr = json.load(open(os.path.join(pdir, best)))
    have = tr.get("library_build_id")
    if have != library_digest():
        sys.stderr.write(f"bench.py: profiles/{best} was collected on another build of librfa_hip.so "
                         f"({have} vs {library_digest()}): roofline.traffic withheld — re-run profiles/collect_pmc.sh\n")
        return None, f"profiles/{best} is stale for this librfa_hip.so (re-run profiles/collect_pmc.sh)"
    sect = tr if hk == 8 else tr.get("hk32", {})         # (top level: the Hk = 8 runs; "hk32": --kv-heads 32)
    if kernel not in sect:
        return None, f"profiles/{best} has no entry for {kernel} at kv_heads {hk}"
    return sect[kernel], f"profiles/{best} (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, separate passes, same binary)"


# Round 5 (profiles/history/r05_power_probe.json): the board sits at its 1400 W cap during the backward and the whole step, and a
# register-only MFMA loop sustains 2.47 PFLOP/s on zero operands but 1.82-1.90 PFLOP/s on random ones — the rate the chip can
# pay for depends on the data.  `roofline.peak` stays the datasheet's 2.5 PFLOP/s (MI355X_MICROARCH.md); the line additionally
# quotes the fraction of the MEASURED random-operand rate.  Round 6 (profiles/r06_power_limiters.md): the
# reference rate is the probe row with the kernels' operand RE-USE pattern (one operand held for 4 consecutive MFMAs, as in
# the P·V / dV / dK GEMMs: 1845 TFLOP/s; a fresh pair per MFMA 1821, one pair for all 1904) instead of round 5's 1830.
MFMA_RANDOM_DATA_TFLOPS = 1845.0


class PowerSampler:
    """board power beside the timed region (amdgpu hwmon power1_input / power1_average, microwatts; the maximum over the
    cards of the node = the one in use on a one-GPU box): a thread that reads the files every 10 ms.  The sensor is a
    moving average over about a second, so the figure is meaningful for timed regions of a few hundred ms and more;
    None where the files do not exist."""

    def __init__(self):
        import glob
        import threading

        self.files = []
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            for name in ("power1_input", "power1_average"):
                if os.path.exists(os.path.join(hw, name)):
                    self.files.append(os.path.join(hw, name))
                    break
        self.samples, self._on, self._quit = [], False, False
        self._th = threading.Thread(target=self._run, daemon=True)
        if self.files:
            self._th.start()

    def _run(self):
        while not self._quit:
            if self._on:
                best = None
                for f in self.files:
                    try:
                        w = int(open(f).read()) / 1e6
                        best = w if best is None or w > best else best
                    except Exception:
                        pass
                if best is not None:
                    self.samples.append(best)
            time.sleep(0.01)

    def start(self):
        self.samples, self._on = [], True

    def stop(self):
        self._on = False
        s = self.samples
        return (sum(s) / len(s), max(s), len(s)) if s else (None, None, 0)

    def close(self):
        self._quit = True


class _Hip:
    """the few HIP runtime calls the in-step timer needs (raw events: the C ABI takes hipEvent_t handles)"""

    def __init__(self):
        import ctypes as C

        self.C = C
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.lib.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.hipEventSynchronize.argtypes = [C.c_void_p]
        self.lib.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.lib.hipEventDestroy.argtypes = [C.c_void_p]

    def event(self):
        e = self.C.c_void_p()
        assert self.lib.hipEventCreate(self.C.byref(e)) == 0
        return e

    def record(self, e):
        assert self.lib.hipEventRecord(e, self.C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0

    def ms(self, e0, e1):
        self.lib.hipEventSynchronize(e1)
        t = self.C.c_float()
        assert self.lib.hipEventElapsedTime(self.C.byref(t), e0, e1) == 0
        return t.value


class InStepTimer:
    """Backend wrapper that times the launches INSIDE the real step: the public function runs unchanged — same
    launches, same stream, same order, same data.  To keep the measurement from perturbing what it measures (a HIP
    event between two kernels is a barrier packet of its own: with every launch of a step bracketed the intervals came
    out up to 8 % longer than rocprofv3's kernel trace of the same run), each instrumented step brackets ONE launch
    kind only, in rotation (fwd / bwd_preprocess / dK/dV / dQ / reduce / the side kernels): before and after the
    backend call, or — inside rfa_bwd — through rfa_bwd_args.prof_events, of which only the two needed entries are
    given.  Events are created before the steps.  Installed through ring_flash_attn._testing.set_backend for a few
    extra steps after the timed region; the timed region itself runs the plain backend."""

    KINDS = ("fwd", "bwd_preprocess", "bwd_first", "bwd_second", "bwd_reduce", "side", "empty")

    def __init__(self, be, nevents=4096):
        self.be, self.hip, self.calls = be, _Hip(), []
        self.pool = [self.hip.event() for _ in range(nevents)]
        self.kind = None          # what the current step brackets
        self.counts = {}          # launches per kind over ALL instrumented steps

    def _ev(self):
        return self.pool.pop() if self.pool else self.hip.event()      # (never fail a run over an empty pool)

    # ---- prefix mode (steps that make exactly one fwd, one bwd_preprocess and one bwd call: world size 1): ONE start
    # event in front of the step's first launch and ONE end event behind the forward (prefix 1) or the backward's
    # first / second kernel (3 / 4).  A long launch's time is the difference of two prefix medians, so nothing sits
    # between it and the launch in front of it (an event there is a packet with a system-scope release of its own:
    # bracketed directly, the forward and dK/dV kernels read 4 - 8 % longer than in rocprofv3's trace of the same
    # run); the two short launches (prefix kinds 2 and 5) are bracketed directly.
    prefix = 0

    def _prefix_call(self, name, fn, a, kw):
        if name == "fwd":
            self._p0 = self._ev()
            self.hip.record(self._p0)
            if self.prefix == 6:                       # the empty prefix: what the two event packets cost by themselves
                e1 = self._ev()
                self.hip.record(e1)
                self.calls.append((("prefix", 6), self._p0, e1))
        if name == "bwd":
            slot = self.prefix - 2                     # 1, 2 = after the first / second kernel; 3 = the reduction, bracketed
            if 1 <= slot <= 3:
                e1 = self._ev()
                ev = (self.hip.C.c_void_p * 4)()
                ev[slot] = e1
                if slot == 3:
                    ev[2] = self._ev()
                r = fn(*a, prof_events=ev, **kw)
                self.calls.append((("prefix", self.prefix), ev[2] if slot == 3 else self._p0, e1))
                return r
            return fn(*a, **kw)
        em = None
        if (name, self.prefix) == ("bwd_preprocess", 2):
            em = self._ev()
            self.hip.record(em)
        r = fn(*a, **kw)
        if (name, self.prefix) in (("fwd", 1), ("bwd_preprocess", 2)):
            e1 = self._ev()
            self.hip.record(e1)
            self.calls.append((("prefix", self.prefix), em or self._p0, e1))
        return r

    def prefix_totals(self, spill):
        """{launch name: ms} — medians over the instrumented steps of each kind.  The three long launches from the
        prefixes T1 (fwd), T3, T4 (the backward's two kernels; T2 = T1 + preprocess); the two short ones (preprocess,
        reduction: tens of microseconds, below the step-to-step spread of a prefix) from a direct bracket."""
        acc = {}
        for name, e0, e1 in self.calls:
            if isinstance(name, tuple):
                acc.setdefault(name[1], []).append(self.hip.ms(e0, e1))
        if os.environ.get("RFA_BENCH_DEBUG_PREFIX"):
            sys.stderr.write("prefix samples: " + json.dumps({p_: [round(x, 3) for x in v_] for p_, v_ in acc.items()}) + "\n")
        T = {p_: statistics.median(v_) for p_, v_ in acc.items()}
        self.overhead_ms = T.pop(6, 0.0)
        T = {p_: max(t - self.overhead_ms, 0.0) for p_, t in T.items()}
        first, second = ("bwd_dkdv", "bwd_dq") if spill else ("bwd_dq", "bwd_dkdv")
        out = {"fwd": T[1], "bwd_preprocess": T[2], first: T[3] - T[1] - T[2], second: T[4] - T[3]}
        if T[5] > 2e-3:
            out["bwd_reduce"] = T[5]
        return out, sum(out.values())

    def __getattr__(self, name):
        fn = getattr(self.be, name)
        if name not in ("fwd", "bwd_preprocess", "cast", "sum_slots", "merge"):
            return fn
        kind = name if name in ("fwd", "bwd_preprocess") else "side"

        def timed(*a, **kw):
            self.counts[name] = self.counts.get(name, 0) + 1
            if self.prefix and name in ("fwd", "bwd_preprocess"):
                return self._prefix_call(name, fn, a, kw)
            if self.kind == "empty" and name == "fwd" and len(self.pool) >= 2:
                # calibration: a bracket with nothing inside, at a kernel boundary of the real step — what the two event
                # packets themselves add to every bracketed interval
                e0, e1 = self._ev(), self._ev()
                self.hip.record(e0)
                self.hip.record(e1)
                self.calls.append(("empty", e0, e1))
            if self.kind != kind or not self.pool:
                return fn(*a, **kw)
            e0, e1 = self._ev(), self._ev()
            self.hip.record(e0)
            r = fn(*a, **kw)
            self.hip.record(e1)
            self.calls.append((name, e0, e1))
            return r

        return timed

    def bwd(self, *a, **kw):
        self.counts["bwd"] = self.counts.get("bwd", 0) + 1
        if self.prefix:
            return self._prefix_call("bwd", self.be.bwd, a, kw)
        slot = {"bwd_first": 0, "bwd_second": 1, "bwd_reduce": 2}.get(self.kind)
        if slot is None or len(self.pool) < 2:
            return self.be.bwd(*a, **kw)
        e0, e1 = self._ev(), self._ev()
        ev = (self.hip.C.c_void_p * 4)()
        ev[slot], ev[slot + 1] = e0, e1
        r = self.be.bwd(*a, prof_events=ev, **kw)
        self.calls.append((self.kind, e0, e1))
        return r

    def totals(self, spill):
        """{launch name: (ms summed over the bracketed launches, bracketed launches)}"""
        names = {"bwd_first": "bwd_dkdv" if spill else "bwd_dq", "bwd_second": "bwd_dq" if spill else "bwd_dkdv"}
        tot = {}
        empty = [self.hip.ms(e0, e1) for name, e0, e1 in self.calls if name == "empty"]
        self.overhead_ms = sum(empty) / len(empty) if empty else 0.0
        for name, e0, e1 in self.calls:
            if name == "empty":
                continue
            ms = self.hip.ms(e0, e1)
            if name == "bwd_reduce" and ms < 2e-3 + self.overhead_ms:
                continue                          # a call without a reduction pass
            ms = max(ms - self.overhead_ms, 0.0)  # the bracket's own event packets (calibrated above)
            t = tot.setdefault(names.get(name, name), [0.0, 0])
            t[0] += ms
            t[1] += 1
        return tot


def cpu_model():
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if line.startswith("Model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(hk, full, fwd_only=False):
    """The reference's CPU path, timed on this box's host cores.

    kind "port": the oracle's _flash_attn_forward/_backward called directly — at world size 1 the reference's zigzag
    schedule IS one such call plus a first-block merge (a copy) and two casts (zigzag_ring_flash_attn.py:28-88).
    kind "reference" (the UNMODIFIED reference schedule, zigzag_ring_flash_attn.py:7-199, with the oracle standing in
    for its `flash_attn` import, as BASELINE.md prescribes) is attempted where /root/reference exists — never on the
    GPU box — but its backward sends dK/dV to its own rank at world size 1, which gloo refuses, so a one-rank run
    falls back to the port and says so on stderr; the reference under gloo at W >= 2 is what tests/ use as the oracle
    of the schedules (oracle/reference_harness.py).
    Default: ONE whole iteration (all kv-head groups, full S = 8192 causal, fwd+bwd: about 37 s on the 128 threads of
    the pool's EPYC 9575F), after a warm-up pass on a quarter-length problem — nothing extrapolated.  Only when the
    second of two quarter-length warm-up passes predicts more than 120 s for it (a box with few host cores) the sample shrinks to ONE kv-head group scaled by
    the number of groups and is labelled extrapolated — that sample does not fill a many-core host and over-states the
    time (measured: 0.0106 vs 0.0274 it/s).  --cpu-baseline-full forces the whole iteration."""
    from oracle import flash_attn_ref as O

    g = HEADS // hk
    groups = hk
    gen = torch.Generator().manual_seed(42)
    q = torch.randn(1, SEQ, g * groups, HEAD_DIM, generator=gen).to(torch.bfloat16)
    k = torch.randn(1, SEQ, groups, HEAD_DIM, generator=gen).to(torch.bfloat16)
    v = torch.randn(1, SEQ, groups, HEAD_DIM, generator=gen).to(torch.bfloat16)
    do = torch.randn(1, SEQ, g * groups, HEAD_DIM, generator=gen).to(torch.bfloat16)
    scale = HEAD_DIM ** -0.5
    kind = "port"
    ref_fn = None
    try:
        from oracle import reference_harness

        if reference_harness.available():
            mods = reference_harness.load_reference(provider="oracle")
            for m in mods.values():
                ref_fn = getattr(m, "zigzag_ring_flash_attn_func", ref_fn)
            kind = "reference" if ref_fn is not None else "port"
    except Exception:
        ref_fn = None

    def once(n, ng, fn):             # the first ng kv-head groups, n rows; fn: the reference's public function or None
        qs, ks, vs, dos = q[:, :n, :g * ng], k[:, :n, :ng], v[:, :n, :ng], do[:, :n, :g * ng]
        if fn is not None:
            qq, kk, vv = (t.clone().requires_grad_(True) for t in (qs, ks, vs))
            if fwd_only:
                with torch.no_grad():
                    fn(qq, kk, vv, causal=True)
                return
            out = fn(qq, kk, vv, causal=True)
            out.backward(dos)
            return
        out, lse, _, _ = O._flash_attn_forward(qs, ks, vs, 0.0, scale, True)
        if fwd_only:
            return
        dq, dk, dv = torch.empty_like(qs), torch.empty_like(ks), torch.empty_like(vs)
        O._flash_attn_backward(dos, qs, ks, vs, out, lse, dq, dk, dv, 0.0, scale, True)

    try:
        once(SEQ // 8, hk, ref_fn)   # (thread pool, allocator, first touch)
    except Exception as e:           # the unmodified reference sends to itself at world size 1, which gloo refuses
        if ref_fn is None:
            raise
        sys.stderr.write(f"bench.py: reference-mode CPU baseline unavailable here ({type(e).__name__}); timing the oracle directly\n")
        ref_fn, kind = None, "port"
        once(SEQ // 8, hk, ref_fn)
    once(SEQ // 4, hk, ref_fn)       # warm-up at the predictor's size (the first large pass pays page faults)
    t0 = time.perf_counter()
    once(SEQ // 4, hk, ref_fn)       # predictor: a causal pass over S / 4 is 1 / 16 of the timed one
    predicted = 16.0 * (time.perf_counter() - t0)
    full = full or predicted <= 120.0
    groups = hk if full else 1
    t0 = time.perf_counter()
    once(SEQ, groups, ref_fn)
    dt = time.perf_counter() - t0
    total = dt * (hk // groups)
    return {
        "value": 1.0 / total,
        "unit": "iters/sec",
        "cores": torch.get_num_threads(),
        "host_cpus": os.cpu_count(),
        "cpu_model": cpu_model(),
        "kind": kind,
        "extrapolated": not full,
        "sample": (f"all {hk} kv-head groups, full S={SEQ} causal {'fwd' if fwd_only else 'fwd+bwd'}, one timed pass after a warm-up ({dt:.2f} s)"
                   if full else
                   f"1 of {hk} kv-head groups ({g} q heads), full S={SEQ} causal {'fwd' if fwd_only else 'fwd+bwd'}, one timed pass after a "
                   f"warm-up ({dt:.2f} s), scaled x{hk}"),
    }


def cpu_baseline_ring(world, hk, budget_s, fwd_only=False):
    """N > 1: the zigzag schedule on the host — `world` gloo CPU processes (cores / world threads each) run one
    forward + backward of zigzag_ring_flash_attn_func with the CPU oracle as their attention arithmetic
    (oracle/cpu_ring_baseline.py).  The full shape (8192 tokens per rank) costs world^2 times the
    one-rank iteration, so the timed sample is the largest TOTAL sequence out of {8192 world, 8192, 4096, 2048} that a
    2048-token probe predicts to finish within `budget_s` (causal attention: time ~ total^2); the sample's shape is
    stated, nothing is extrapolated, and `tflops` makes samples of different sizes comparable."""
    import socket

    cores = os.cpu_count() or 1
    threads = max(1, cores // world)

    def run(total):
        s_rank = total // world
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
            for k_ in ("RFA_ZIGZAG_EXCHANGE", "RFA_BENCH_FORCE_RCCL", "TORCHELASTIC_RUN_ID"):
                env.pop(k_, None)
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_ring_baseline.py"),
                                           str(s_rank), str(hk), str(threads), "fwd" if fwd_only else "fwdbwd"],
                                          env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE, text=True))
        outs = [p_.communicate(timeout=max(600.0, 20 * budget_s)) for p_ in procs]
        for p_, (so, se) in zip(procs, outs):
            if p_.returncode != 0:
                raise RuntimeError(f"cpu_ring_baseline rank failed ({p_.returncode}): {se[-800:]}")
        line = [l for l in outs[0][0].splitlines() if l.startswith("{")][-1]
        return json.loads(line)

    probe_total = 2048
    probe = run(probe_total)
    best, rep = probe_total, probe
    for total in (SEQ * world, SEQ, SEQ // 2):
        if total <= probe_total:
            break
        if probe["seconds"] * (total / probe_total) ** 2 <= budget_s:
            best, rep = total, run(total)
            break
    flops = (1.0 if fwd_only else 3.5) * causal_fwd_flops([best])
    full = best == SEQ * world
    return {
        "value": 1.0 / rep["seconds"],
        "unit": "iters/sec",
        "cores": threads * world,
        "host_cpus": cores,
        "cpu_model": cpu_model(),
        "kind": rep["kind"],
        "extrapolated": False,
        "full_shape": full,
        "tflops": flops / rep["seconds"] / 1e12,
        "sample": (f"zigzag_ring_flash_attn_func {'fwd' if fwd_only else 'fwd+bwd'} over {world} gloo CPU processes x {threads} threads, "
                   f"{best // world} tokens per rank (total {best}; the GPU line runs {SEQ} per rank = {SEQ * world}), "
                   f"h=32 hk={hk} d=128 bf16 causal, one timed iteration after a quarter-length warm-up "
                   f"({rep['seconds']:.2f} s)" + ("" if full else
                   f"; the full shape was predicted over the {budget_s:.0f} s budget from a {probe_total}-token probe "
                   f"({probe['seconds']:.2f} s)")),
    }


def comm_bytes_per_iter(mode, wire_fp32, world, hk):
    """xGMI bytes SENT per rank per iteration (fwd + bwd) of the dense zigzag exchange"""
    m = 2 * SEQ * hk * HEAD_DIM * 2                      # K + V of one rank, bf16
    if mode == "ring":                                   # (W-1) M fwd, (W-1) M + W 2M (fp32 dK/dV) bwd
        return (world - 1) * m + (world - 1) * m + world * 2 * m
    contrib = (2 * m) if wire_fp32 else m                # dK + dV contribution for one chunk
    # one all-gather (the backward reuses the K/V the forward gathered) + one all-to-all / reduce-scatter
    return (world - 1) * m + (world - 1) * contrib


def free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n):
    """re-run this script with the same arguments as `n` ranks of one node under torch.distributed.run; returns the
    launcher's exit status (non-zero when any rank failed).  RFA_BENCH_MASTER_PORT pins the rendezvous port."""
    port = os.environ.get("RFA_BENCH_MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (default 300: a 0.6 s timed region at N = 1 — long "
                                                             "enough for the board-power reading, a ~1 s moving average, to be valid)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--kv-heads", type=int, default=8, help="8 = the reference benchmark's GQA; 32 = MHA")
    ap.add_argument("--workload", default="zigzag", choices=["zigzag", "ring", "stripe", "ring_varlen", "zigzag_varlen", "llama3"],
                    help="zigzag = the BASELINE.json headline; ring / stripe: the other rows of benchmark_kvpacked_func.py; "
                         "the *_varlen / llama3 ones mirror benchmark_varlen_kvpacked_func.py")
    ap.add_argument("--forward-only", action="store_true",
                    help="time the forward alone under torch.no_grad() (benchmark_kvpacked_func.py:85-96)")
    ap.add_argument("--exchange", default=None, choices=["auto", "gather", "gather_ps", "ring"],
                    help="dense zigzag exchange form (default: RFA_ZIGZAG_EXCHANGE or auto)")
    ap.add_argument("--wire", default=None, choices=["io", "fp32"], help="dK/dV dtype on the wire (gather form)")
    ap.add_argument("--virtual-world", type=int, default=0,
                    help="N=1 only: additionally time rank --virtual-rank's kernel sequence of a job of this world size "
                         "with the exchange looped back to local buffers (compute-only cost of the multi-step path)")
    ap.add_argument("--virtual-rank", type=int, default=-1)
    ap.add_argument("--no-autotune", action="store_true",
                    help="N > 1, dense zigzag, exchange 'auto': skip the measured choice between the exchange forms "
                         "(ring_flash_attn.tuning.autotune_zigzag_exchange in the warm-up) and use the shape rule")
    ap.add_argument("--exchange-check-steps", type=int, default=0,
                    help="N > 1: run this many UNTIMED steps with ring_flash_attn.config.exchange_check on before the warm-up — "
                         "every K/V and dK/dV buffer a rank receives is checksummed against its sender (one extra tiny "
                         "all-gather per schedule call); a mismatch ends the run naming rank / step / buffer")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="always time one whole iteration (the default does unless the host is predicted to need more than 120 s)")
    ap.add_argument("--cpu-baseline-budget-s", type=float, default=60.0,
                    help="N > 1: seconds the timed CPU sample may be predicted to take (picks the total sequence length)")
    ap.add_argument("--no-breakdown", action="store_true")
    args = ap.parse_args()
    if args.exchange:
        os.environ["RFA_ZIGZAG_EXCHANGE"] = args.exchange
    if args.wire:
        os.environ["RFA_DKV_WIRE"] = args.wire

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N` (the driver's command line): launch the N ranks ourselves, the
        # way the reference's benchmark is started (`torchrun --nproc_per_node N`, /root/reference/README.md:135-147) —
        # one process per GPU on this node, rendezvous on 127.0.0.1 and a free port.  The children inherit this
        # process's stdout, so the contract's ONE JSON line (rank 0) still is the only thing on it.
        if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node has {torch.cuda.device_count()} visible GPU(s)")
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; start it as "
                         f"`python bench.py --gpus {args.gpus}` (it launches its own ranks) or under "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the operator has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if "MASTER_ADDR" not in os.environ:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ.setdefault("MASTER_PORT", "29541")
    # one process per GPU over RCCL ("nccl" on ROCm); a single process has nothing to exchange
    # stdout carries the ONE JSON line and nothing else: librccl / libgloo print version and connection banners on
    # fd 1 through C stdio buffers that are only flushed at exit (a one-rank RCCL run showed 5 banner lines BEHIND the
    # JSON line), so fd 1 is pointed at stderr for the whole run and the line is written to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # RFA_BENCH_FORCE_RCCL=1 (tests/test_gpu_rccl_world1.py): a ONE-rank RCCL group with the schedule forced onto its
    # multi-step path — this script's N > 1 branches (RCCL barrier / all_reduce, fixed-count spin-up, comm block)
    # on a one-GPU box.  The line it prints is marked and is not a measurement.
    forced = world == 1 and os.environ.get("RFA_BENCH_FORCE_RCCL") == "1"
    multi = world > 1 or forced
    if multi:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=0, world_size=1)

    import ring_flash_attn as R
    from ring_flash_attn import config as rfa_config
    from ring_flash_attn import _testing as rfa_testing

    if forced:
        rfa_testing.force_steps(True)
    from ring_flash_attn.zigzag_ring_flash_attn import exchange_mode, _wire_fp32

    hk = args.kv_heads
    wl = args.workload
    torch.manual_seed(42 + rank)
    if wl in DENSE:
        q = torch.randn(1, SEQ, HEADS, HEAD_DIM, device=dev, dtype=torch.bfloat16, requires_grad=True)
        kv = torch.randn(1, SEQ, 2, hk, HEAD_DIM, device=dev, dtype=torch.bfloat16, requires_grad=True)
        dout = torch.randn(1, SEQ, HEADS, HEAD_DIM, device=dev, dtype=torch.bfloat16)
        fn = {"zigzag": R.zigzag_ring_flash_attn_kvpacked_func, "ring": R.ring_flash_attn_kvpacked_func,
              "stripe": R.stripe_flash_attn_kvpacked_func}[wl]

        def call(i):
            return fn(q, kv, causal=True, window_size=(-1, -1), alibi_slopes=None, deterministic=False,
                      return_attn_probs=False)
    else:
        q = torch.randn(SEQ, HEADS, HEAD_DIM, device=dev, dtype=torch.bfloat16, requires_grad=True)
        kv = torch.randn(SEQ, 2, hk, HEAD_DIM, device=dev, dtype=torch.bfloat16, requires_grad=True)
        dout = torch.randn(SEQ, HEADS, HEAD_DIM, device=dev, dtype=torch.bfloat16)
        cus = [torch.tensor(c, device=dev, dtype=torch.int32) for c in VARLEN_PATTERNS]
        maxs = [max(b - a for a, b in zip(c[:-1], c[1:])) for c in VARLEN_PATTERNS]
        if wl in ("zigzag_varlen", "ring_varlen"):
            fn = R.zigzag_ring_flash_attn_varlen_kvpacked_func if wl == "zigzag_varlen" else R.ring_flash_attn_varlen_kvpacked_func

            def call(i):
                j = i % len(cus)
                return fn(q, kv, cus[j], maxs[j], causal=True, window_size=(-1, -1), alibi_slopes=None,
                          deterministic=False, return_attn_probs=False)
        else:
            fn = R.llama3_flash_attn_varlen_kvpacked_func
            prep = [R.llama3_flash_attn_prepare_cu_seqlens(torch.tensor(c, dtype=torch.int32) * world, True, rank, world)
                    for c in VARLEN_PATTERNS]
            prep = [(cq.to(dev), ck.to(dev), mq, mk, sl) for cq, ck, mq, mk, sl in prep]

            def call(i):
                cq, ck, mq, mk, sl = prep[i % len(prep)]
                return fn(q, kv, cq, ck, mq, mk, heads_k_stride=LLAMA3_HEADS_K_STRIDE, local_k_slice=sl, causal=True,
                          window_size=(-1, -1), alibi_slopes=None, deterministic=False, return_attn_probs=False)

    counter = [0]

    fwd_only = args.forward_only

    def step():
        if fwd_only:
            with torch.no_grad():
                call(counter[0])
            counter[0] += 1
            return
        q.grad = None
        kv.grad = None
        out = call(counter[0])
        counter[0] += 1
        out.backward(dout)

    def timed(n):
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        tmax = torch.tensor([el], dtype=torch.float64, device=dev if multi else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return tmax.item()

    # ---- N > 1: let the group MEASURE which exchange form is faster here (a few fwd+bwd in each form on scratch
    # tensors of the workload's shapes, max over ranks, same decision on every rank) instead of trusting a default
    # that has never run on this node; a form that fails is disqualified, not fatal.  Untimed, reported in `comm`.
    tune_rep, probe_rep = None, None
    if args.no_autotune:
        rfa_config.set(autotune=False)           # (the library would otherwise measure on its first multi-rank call)
    if multi and wl == "zigzag" and not args.no_autotune and rfa_config.get().zigzag_exchange == "auto":
        from ring_flash_attn import tuning

        kd, vd = kv.detach()[:, :, 0], kv.detach()[:, :, 1]
        try:
            tune_rep = tuning.autotune_zigzag_exchange(None, q.detach(), kd, vd, iters=3, warm=2)
        except Exception as e:           # (then the shape rule decides; the tuning step must never sink the benchmark)
            tune_rep = {"error": f"{type(e).__name__}: {e}"}
    if multi:
        from ring_flash_attn import tuning

        try:
            probe_rep = tuning.comm_probe(None, dev, 2 * SEQ * hk * HEAD_DIM * 2)   # K + V of one rank
        except Exception as e:           # the probe must never sink the benchmark
            probe_rep = {"error": f"{type(e).__name__}: {e}"}

    # ---- N > 1, opt-in: the exchange audit for a few untimed steps (the first 8-GPU run is also the first execution of
    # every RCCL ordering of the package: profiles/collect_scale.sh asks for 2).  A failure here is a wrong result, not a
    # reporting problem: it propagates.
    exchange_checked = 0
    if multi and args.exchange_check_steps > 0:
        with rfa_config.override(exchange_check=True):
            for _ in range(args.exchange_check_steps):
                step()
            torch.cuda.synchronize()
        exchange_checked = args.exchange_check_steps
        counter[0] = 0

    # device spin-up (not a measurement knob): the MI355X needs some tens of milliseconds of load to leave
    # its idle clocks; without it the W warm-up steps (W x ~2 ms) end while the clocks are still ramping and
    # the timed region measures the ramp, not the kernels.  Untimed, bounded, reported in the JSON line.
    spin_s = float(os.environ.get("RFA_BENCH_SPINUP_S", "0.3"))
    if not multi:
        t_spin = time.perf_counter()
        while spin_s > 0 and time.perf_counter() - t_spin < spin_s:
            step()
            torch.cuda.synchronize()
    elif spin_s > 0:
        # every rank must make the same number of collective calls: a fixed count instead of a wall-clock bound
        for _ in range(8):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    counter[0] = 0
    torch.cuda.reset_peak_memory_stats(dev)
    power = PowerSampler() if rank == 0 else None
    if power:
        power.start()
    elapsed = timed(args.steps)
    watts = power.stop() if power else (None, None, 0)
    if power:
        power.close()
    # peak device memory of the timed steps (the reference names memory as its known limitation, README.md:154; the
    # gather exchange form trades O(S_total) scratch for fewer transfers): allocator peak incl. the inputs, max over ranks
    peak = torch.tensor([float(torch.cuda.max_memory_allocated(dev))], dtype=torch.float64, device=dev if multi else "cpu")
    dist.all_reduce(peak, op=dist.ReduceOp.MAX)
    peak_gib = peak.item() / 2 ** 30

    ms = elapsed / args.steps * 1e3
    its = args.steps / elapsed
    per_gpu_flops = (1.0 if fwd_only else 3.5) * fwd_flops_per_gpu(wl, world)
    names = {"zigzag": "zigzag_ring_flash_attn_kvpacked_func", "ring": "ring_flash_attn_kvpacked_func",
             "stripe": "stripe_flash_attn_kvpacked_func", "ring_varlen": "ring_flash_attn_varlen_kvpacked_func",
             "zigzag_varlen": "zigzag_ring_flash_attn_varlen_kvpacked_func", "llama3": "llama3_flash_attn_varlen_kvpacked_func"}
    shape = (f"per-rank q=(1,{SEQ},{HEADS},{HEAD_DIM}) kv=(1,{SEQ},2,{hk},{HEAD_DIM})" if wl in DENSE else
             f"per-rank q=({SEQ},{HEADS},{HEAD_DIM}) kv=({SEQ},2,{hk},{HEAD_DIM}), 4 cu_seqlens patterns cycled"
             + (f", heads_k_stride {LLAMA3_HEADS_K_STRIDE}" if wl == "llama3" else ""))
    passes = "fwd" if fwd_only else "fwd+bwd"
    result = {
        "metric": f"iters/sec {passes} zigzag_ring, seq=8192*ws, h=32, d=128 bf16" if wl == "zigzag"
                  else f"iters/sec {passes} {wl}, 8192 tokens/rank, h=32, d=128 bf16",
        "value": its,
        "unit": "iters/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "spinup_s": spin_s,
        "ms_per_step": ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"{names[wl]} {passes}, {shape} bf16 causal, total seq {SEQ * world}",
            "kv_heads": hk,
            "world_size": world,
            "forward_only": fwd_only,
        },
        "peak_device_memory_gib": round(peak_gib, 3),
        # board power over the timed region (hwmon, a ~1 s moving average: meaningful from a few hundred ms of steps on)
        # and the energy of one step: at the 1400 W cap the step's time IS its energy / 1400 W
        # `valid`: the sensor is a ~1 s moving average — a timed region shorter than 0.5 s (the driver's --steps 20 is 0.04 s)
        # reads the tail of whatever ran before it (1293 W where 200-step runs read 1385-1400 W: VERDICT r5 weak #11)
        "power": ({"avg_w": round(watts[0], 1), "max_w": round(watts[1], 1), "samples": watts[2],
                   "joules_per_step": round(watts[0] * elapsed / args.steps, 4), "source": "amdgpu hwmon power1_input",
                   "timed_region_s": round(elapsed, 3), "valid": bool(elapsed >= 0.5)}
                  if watts[0] is not None else None),
        "algorithmic_tflops_per_gpu": per_gpu_flops * its / 1e12,
        "mfma_roofline_frac_end_to_end": per_gpu_flops * its / 1e12 / MFMA_PEAK_TFLOPS,
    }

    # ---- exchange accounting (N > 1): the same rank-local kernel sequence with the exchange looped back
    if forced:
        result["forced_rccl_world1"] = "N > 1 code path on a one-rank RCCL group (test only, not a measurement)"
    # (everything below is reporting: the measured line must survive a failure in it — it has only ever run on one GPU —
    #  so each block records its error under "errors" instead of ending the run; all ranks take the same path)
    errors = {}
    if multi:
        mode = exchange_mode(kv.detach()[:, :, 0], world, q.detach()) if wl == "zigzag" else {"llama3": "allgather+reduce_scatter"}.get(wl, "ring")
        comp = float("nan")
        rfa_testing.set_loopback((rank, world))
        try:
            for _ in range(2):
                step()
            counter[0] = 0
            comp = timed(args.steps) / args.steps * 1e3
        except Exception as e:
            errors["compute_only"] = f"{type(e).__name__}: {e}"
        finally:
            rfa_testing.set_loopback(None)
        result["comm"] = {
            "exchange": mode,
            "exchange_check_steps_passed": exchange_checked,      # (--exchange-check-steps: audited, untimed steps before the warm-up)
            "dkv_wire": ("fp32" if (_wire_fp32() or mode == "ring") else "bf16") if wl == "zigzag" else None,
            "backend": dist.get_backend(),
            "world_size_observed": dist.get_world_size(),
            "bytes_sent_per_rank_per_iter": comm_bytes_per_iter(mode, _wire_fp32(), world, hk) if wl == "zigzag" else None,
            "compute_only_ms": comp if comp == comp else None,
            "exposed_ms": (ms - comp) if comp == comp else None,
            "autotune": tune_rep,
            "probe": probe_rep,
            "note": "compute_only = this rank's exact kernel sequence with the exchange looped back to local buffers "
                    "(ring_flash_attn._testing.set_loopback), max over ranks; exposed = ms_per_step - compute_only",
        }

    if world == 1 and args.virtual_world > 1:
        vw = args.virtual_world
        vr = args.virtual_rank if args.virtual_rank >= 0 else vw // 2 - 1 + (vw > 2)
        rfa_testing.set_loopback((vr, vw))
        try:
            for _ in range(2):
                step()
            counter[0] = 0
            torch.cuda.reset_peak_memory_stats(dev)
            vms = timed(args.steps) / args.steps * 1e3
            vpeak = torch.cuda.max_memory_allocated(dev) / 2 ** 30
        finally:
            rfa_testing.set_loopback(None)
        result["virtual_ring"] = {
            "world": vw, "rank": vr, "ms_per_step": vms, "ideal_ms": vw * ms, "efficiency": vw * ms / vms,
            # allocator peak of one rank's steps at this world size in this exchange form (gather: O(S_total) scratch)
            "peak_device_memory_gib": round(vpeak, 3),
            "exchange": exchange_mode(kv.detach()[:, :, 0], vw) if wl == "zigzag" else None,
            "note": "one rank's exact kernel sequence at this world size, exchange looped back to local buffers; "
                    "ideal = world x the measured world-size-1 step",
        }

    # ---- kernel times INSIDE the step (every rank runs the instrumented steps: collectives must stay matched)
    instep = None
    if not args.no_breakdown:
        from ring_flash_attn import backend as rfa_backend

        timer = InStepTimer(rfa_backend.get_backend())
        rfa_testing.set_backend(timer)
        rounds = max(2, min(args.steps // 4, 16))           # instrumented steps = rounds x kinds, one kind per step
        span_ms = None
        try:
            for _ in range(4):                       # (the wrapper warm and the host ahead of the device again; nothing bracketed)
                step()
            # prefix timing needs a step that launches nothing but one fwd, one preprocess and one backward (the dense
            # zigzag call on one rank; the packed workloads run torch kernels in between — index/fill/add — and keep
            # the bracket form)
            single = wl in DENSE and timer.counts == {"fwd": 4, "bwd_preprocess": 4, "bwd": 4}
            timer.counts = {}
            counter[0] = 0
            nprof = 0
            for _ in range(rounds):
                for kind in (range(1, 7) if single else InStepTimer.KINDS):
                    if single:
                        timer.prefix = kind
                    else:
                        timer.kind = kind
                    step()
                    nprof += 1
            torch.cuda.synchronize()
        except Exception as e:
            errors["kernels_in_step"] = f"{type(e).__name__}: {e}"
            single = None
        finally:
            timer.kind, timer.prefix = None, 0
            rfa_testing.set_backend(None)
        spill = rfa_config.get().bwd_ds_spill
        instep = {}
        if single is None:
            instep = None
        elif single:
            per, span_ms = timer.prefix_totals(spill)
            for n, t in per.items():
                instep[n] = {"avg_launch_ms": t, "launches_per_step": 1.0, "ms_per_step": t}
        else:
            tot = timer.totals(spill)
            bwd_calls = timer.counts.get("bwd", 0) / nprof
            per_step = {"fwd": timer.counts.get("fwd", 0) / nprof, "bwd_preprocess": timer.counts.get("bwd_preprocess", 0) / nprof,
                        "bwd_dkdv": bwd_calls, "bwd_dq": bwd_calls, "bwd_reduce": bwd_calls}
            for n, t in tot.items():
                launches = per_step.get(n, timer.counts.get(n, 0) / nprof)
                if n == "bwd_reduce":                     # only the calls that made a reduction pass were counted
                    launches = bwd_calls * t[1] / max(1, sum(1 for c in timer.calls if c[0] == "bwd_reduce"))
                instep[n] = {"avg_launch_ms": t[0] / t[1], "launches_per_step": launches, "ms_per_step": t[0] / t[1] * launches}

    if rank == 0 and instep is not None:
        try:
            with torch.no_grad():
                if wl in DENSE:
                    iso = kernel_breakdown(q.detach(), kv.detach(), dout)
                    f = causal_fwd_flops([SEQ])
                else:
                    cu_b = torch.tensor(VARLEN_PATTERNS[1], device=dev, dtype=torch.int32)
                    iso = kernel_breakdown(q.detach(), kv.detach(), dout, cu_b)
                    f = causal_fwd_flops([b - a for a, b in zip(VARLEN_PATTERNS[1][:-1], VARLEN_PATTERNS[1][1:])])
            sum_ms = sum(v_["ms_per_step"] for v_ in instep.values())
            result["kernels_in_step"] = {
                "ms": {n: round(v_["ms_per_step"], 4) for n, v_ in instep.items()},
                "launches": {n: v_["launches_per_step"] for n, v_ in instep.items()},
                "sum_ms": round(sum_ms, 4),
                "ms_per_step": round(ms, 4),
                "other_ms": round(ms - sum_ms, 4),
                "bracket_overhead_ms": round(timer.overhead_ms, 5),
                # an instrumented step carries two event packets and is not overlapped with its neighbours' launches the
                # way the timed region's steps are: its launches may span up to 3 % more than the average timed step
                "consistent": bool(sum_ms <= ms * 1.03),
                "span_ms": round(span_ms, 4) if span_ms is not None else None,
                "how": (f"prefix timing inside the real step: one HIP event in front of the step's first launch, one behind "
                        f"the forward / the backward's first / second kernel, in rotation ({nprof} instrumented steps after the timed "
                        f"region; inside rfa_bwd through rfa_bwd_args.prof_events); a long launch's ms = difference of two "
                        f"prefix medians, the two short ones (preprocess, reduction) are bracketed directly, the events' own "
                        f"cost (an empty prefix) is subtracted; sum = span of a step's launches; other_ms = ms_per_step - sum (the host-side "
                        f"turn-around between two steps)") if span_ms is not None else
                       (f"HIP events around ONE launch kind per instrumented step, in rotation (InStepTimer: backend calls of "
                        f"the public function, rfa_bwd_args.prof_events inside the backward), {nprof} instrumented steps "
                        f"after the timed region; ms = (average bracketed interval - bracket_overhead_ms, the interval of an "
                        f"EMPTY bracket at a kernel boundary of the same steps) x launches per step; other_ms = ms_per_step "
                        f"- sum (host gaps, autograd, grad buffers)"),
            }
            # algorithmic GEMM work per launch (fwd = 4BHS^2D/2, bwd = 2.5 fwd, of which the
            # dK/dV kernel owns 4 of the 5 backward GEMMs and the dQ kernel the fifth; recomputation of S and dP
            # inside the 7-GEMM dQ kernel is NOT credited); a forward-only run has the forward launch alone
            algo = {"fwd": f} if fwd_only else {"fwd": f, "bwd_dkdv": 2.0 * f, "bwd_dq": 0.5 * f}
            in_step_line = world == 1 and wl in DENSE      # one launch of each kernel per step: its in-step time IS the launch's
            if in_step_line:
                t_in = {n: instep[n]["avg_launch_ms"] for n in algo if n in instep}
            else:
                # N > 1 / packed workloads: the step makes several launches of each kernel with different shapes; the
                # roofline line is this rank's local causal block (step 0 of every world size), timed on its own
                t_in = {n: iso[n] for n in algo}
            dom = max(t_in, key=lambda n: t_in[n])
            ach = algo[dom] / (t_in[dom] * 1e-3) / 1e12
            kn = {"fwd": "fwd_kernel", "bwd_dkdv": "dkdv_kernel", "bwd_dq": "dq_ds_kernel" if spill else "dq_kernel"}[dom]
            entry, note = committed_traffic(kn, hk) if wl in DENSE else (None, "collected for the dense workloads only")
            # traffic over what the algorithm has to move (inputs once, outputs once): the hand-off the 5-GEMM backward
            # creates between its two kernels shows up here (13.6x for dkdv_kernel, 26.9x for dq_ds_kernel)
            t_alg = entry.get("algorithmic_bytes") if entry else None
            result["roofline"] = {
                "kernel": kn,
                "launch": "this rank's local causal block (step 0 of every world size)",
                "bound": "mfma",
                "achieved": ach,
                "peak": MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": ach / MFMA_PEAK_TFLOPS,
                # against the rate a register-only MFMA loop sustains on RANDOM operands on this chip (measured:
                # tools/mfma_power_probe.hip `holdB4`, profiles/r06_power_limiters.md; 2.47 PFLOP/s on zero operands)
                "frac_of_random_operand_mfma_rate": ach / MFMA_RANDOM_DATA_TFLOPS,
                "random_operand_mfma_tflops": MFMA_RANDOM_DATA_TFLOPS,
                "traffic": entry["hbm_bytes_per_launch"] if entry else None,
                "traffic_algorithmic": entry.get("algorithmic_bytes") if entry else None,
                "traffic_handoff": entry.get("handoff_bytes") if entry else None,
                "traffic_over_algorithmic": (entry["hbm_bytes_per_launch"] / t_alg) if (entry and t_alg) else None,
                "traffic_source": note,
                "avg_launch_ms": t_in[dom],
                "timed": "in-step" if in_step_line else "isolated launches of the local block",
                # issue-side counters and the effective shader clock (GRBM_GUI_ACTIVE / kernel duration) of the PROFILED
                # passes of the same build (profiles/collect_pmc.sh): the fraction above is quoted against the 2.4 GHz
                # peak while the chip clocks to its power budget; wave cycles / busy fractions say whether a change saved
                # cycles or only moved the clock.  (The amdgpu sysfs / rocm-smi sclk nodes report a coarse DPM level —
                # 1403 MHz under this load, 95 MHz from rocm-smi — not the clock the kernels run at: measured, not used.)
                "profiled": ({k_: entry[k_] for k_ in ("effective_clock_ghz_profiled", "mfma_pipe_busy", "sq_wave_cycles",
                                                       "sq_active_inst_any", "sq_wait_any", "sq_wait_inst_any") if k_ in entry}
                             if entry else None),
            }
            result["kernels_ms"] = {k2: round(v2, 4) for k2, v2 in t_in.items()}
            result["kernels_ms_isolated"] = {k2: round(v2, 4) for k2, v2 in iso.items()}
            result["kernels_tflops"] = {n: algo[n] / (t_in[n] * 1e-3) / 1e12 for n in t_in}
            bwd_ms = sum(instep[n]["ms_per_step"] for n in instep if n.startswith("bwd"))
            if in_step_line and bwd_ms > 0:
                result["backward_tflops"] = 2.5 * f / (bwd_ms * 1e-3) / 1e12
        except Exception as e:
            errors["roofline"] = f"{type(e).__name__}: {e}"
    if rank == 0 and not args.no_cpu_baseline and wl in DENSE:
        # every N carries the CPU path next to the GPU number: N = 1 the oracle directly, N > 1 the
        # zigzag schedule over N gloo CPU processes (a one-rank forced-RCCL test line runs the two-rank form)
        try:
            if multi:
                result["cpu_baseline"] = cpu_baseline_ring(max(world, 2), hk, args.cpu_baseline_budget_s, fwd_only)
            else:
                result["cpu_baseline"] = cpu_baseline(hk, args.cpu_baseline_full, fwd_only)
        except Exception as e:
            errors["cpu_baseline"] = f"{type(e).__name__}: {e}"
    if rank == 0 and "roofline" not in result and not args.no_breakdown:
        # the contract's object, from the isolated launches of the local block when the in-step path failed
        try:
            with torch.no_grad():
                iso = kernel_breakdown(q.detach(), kv.detach(), dout) if wl in DENSE else None
            if iso:
                f = causal_fwd_flops([SEQ])
                ach = 2.0 * f / (iso["bwd_dkdv"] * 1e-3) / 1e12
                result["roofline"] = {"kernel": "dkdv_kernel", "bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS,
                                      "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS, "traffic": None,
                                      "avg_launch_ms": iso["bwd_dkdv"], "timed": "isolated launches of the local block (fallback)"}
        except Exception as e:
            errors["roofline_fallback"] = f"{type(e).__name__}: {e}"
    if errors:
        result["errors"] = errors

    if rank == 0:
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    os.close(real_stdout)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
