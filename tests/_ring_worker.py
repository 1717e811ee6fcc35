"""Multi-process worker shared by the schedule tests: runs the package's public functions on one
rank of a gloo ring and compares against the golden vectors produced by the unmodified reference
(tests/golden/make_golden.py).  Backend: the CPU oracle (not-gpu tests, schedules only) or the
real HIP kernels with every rank sharing cuda:0 (gpu tests; gloo stages through host memory)."""
import os
import sys
import traceback

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# tolerance of one comparison: |got - ref| <= atol + rtol * max|ref|
TOL_ORACLE = dict(out=(2e-2, 0.0), lse=(1e-5, 1e-5), grad=(3e-2, 1e-2))
# HIP kernels vs reference-with-oracle: both round block results to bf16 at different points
TOL_HIP = dict(out="out_ring", lse="lse_ring", grad="grad_ring")          # tests/_tol.py


def _cmp(name, got, ref, tol, errs):
    """tol: an (atol, rtol) pair (oracle backend: the schedules reproduce the reference's arithmetic up to the order
    of the fp32 merges) or the name of a tests/_tol.py kind (HIP kernels: max-abs, relative Frobenius norm, mean-abs)"""
    if isinstance(tol, str):
        import _tol

        errs += _tol.failures(name, got, ref, tol)
        return
    got, ref = got.float(), ref.float()
    if got.shape != ref.shape:
        errs.append(f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}")
        return
    atol, rtol = tol
    diff = (got - ref).abs().max().item()
    lim = atol + rtol * ref.abs().max().item()
    if not (diff <= lim):
        errs.append(f"{name}: max|diff| {diff:.3e} > {lim:.3e}")
    fro = ((got - ref).double().norm() / ref.double().norm().clamp_min(1e-30)).item()
    if not (fro <= 1e-2):
        errs.append(f"{name}: relative Frobenius error {fro:.3e} > 1e-2")


def run_rank(rank, W, port, names, use_hip, ret, via_reference=False):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=W)
        import make_golden as MG
        import ring_flash_attn as R
        from ring_flash_attn import backend
        from ring_flash_attn import _testing

        if via_reference:
            # INTEGRATION.md route B: the UNMODIFIED reference schedules on top of the shipped `flash_attn`
            # compatibility package (whose kernels are whatever backend is installed below)
            import types

            from oracle.reference_harness import load_reference

            ref_mods = load_reference(provider="shim")
            R = types.SimpleNamespace()
            for mod in ref_mods.values():
                for name in dir(mod):
                    if name.endswith("_func") or name == "llama3_flash_attn_prepare_cu_seqlens":
                        setattr(R, name, getattr(mod, name))

        golden = torch.load(os.path.join(ROOT, "tests", "golden", "ring_golden.pt"), weights_only=False)
        if use_hip:
            dev = torch.device("cuda:0")
            torch.cuda.set_device(dev)
            _testing.set_backend(None)
            _testing.allow_host_staging(True)       # several gloo ranks share this one GPU
            tol = TOL_HIP
        else:
            from oracle.oracle_backend import OracleBackend

            dev = torch.device("cpu")
            _testing.set_backend(OracleBackend())
            tol = TOL_ORACLE
        errs = []
        compiled = os.environ.get("RFA_TEST_COMPILE") == "1"
        if compiled:
            # the reference runs its tests a second time with the function under torch.compile at the full world
            # size (test/test.sh:23-25, test/test_zigzag_ring_flash_attn_func.py:105-108).  Every public function is
            # replaced by a compiled caller (traced tensor work on both sides of the call); the schedule itself is
            # captured as ONE registered operator per direction (ring_flash_attn/_ops.py) — no graph break — and must
            # give the same results as the plain call, bit for bit.  RFA_TEST_COMPILE_BACKEND: inductor — torch.compile's
            # default, what the reference's tests use — or aot_eager (dynamo + AOT autograd without code generation).
            import types
            from torch import _dynamo as dynamo

            dynamo.reset()
            # one compiled caller per public function serves every case of the list (shapes, dtypes, argument counts differ):
            # past dynamo's recompile limit a frame silently runs EAGERLY, i.e. the test would stop testing the compiled form
            for lim in ("recompile_limit", "cache_size_limit", "accumulated_recompile_limit", "accumulated_cache_size_limit"):
                if hasattr(dynamo.config, lim):
                    setattr(dynamo.config, lim, 4096)
            eager_R, R = R, types.SimpleNamespace()

            def compiled_caller(fn, fullgraph):
                def caller(*a, **kw_):
                    a = tuple((t * 1 if torch.is_tensor(t) and t.is_floating_point() else t) for t in a)
                    res = fn(*a, **kw_)
                    return tuple((t + 0 if torch.is_tensor(t) else t) for t in res) if isinstance(res, tuple) else res + 0
                return torch.compile(caller, backend=os.environ.get("RFA_TEST_COMPILE_BACKEND", "inductor"), fullgraph=fullgraph)

            # round 4: every schedule lowers to ONE registered operator per direction at any world size
            # (ring_flash_attn/_ops.py: rfa::sched_fwd / sched_bwd, rfa::llama3_fwd / llama3_bwd) —
            # `fullgraph=True` turns any graph break into an error
            for name in dir(eager_R):
                obj = getattr(eager_R, name)
                setattr(R, name, compiled_caller(obj, fullgraph=True) if name.endswith("_func") else obj)
        for n in names:
            c = MG.CASES[n]
            (q, k, v, do), extra = MG.shard(c, rank)
            if "sample" in c:
                assert MG.inputs_digest(c) == golden["cases"][n]["inputs_sha256"], "seeded inputs drifted"
            else:
                assert torch.equal(MG.make_inputs(c)[0], golden["cases"][n]["inputs"]["q"]), "seeded inputs drifted"
            ref = golden["cases"][n]["ranks"][rank]
            q, k, v, do = [t.to(dev) for t in (q, k, v, do)]
            q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
            kw = dict(dropout_p=0.0, window_size=(-1, -1), alibi_slopes=None, deterministic=False, return_attn_probs=True)
            kind = c["kind"]
            if kind == "zigzag" and os.environ.get("RFA_TEST_CHECKPOINT") == "1":
                # activation checkpointing: the first forward runs without grad (nothing is kept for a backward), the
                # recomputation forward is the one whose gathered K/V the backward reuses
                from torch.utils.checkpoint import checkpoint

                out, lse, _ = checkpoint(lambda a, b_, c_: R.zigzag_ring_flash_attn_func(a, b_, c_, causal=True, **kw),
                                         q, k, v, use_reentrant=False)
            elif kind == "zigzag":
                out, lse, _ = R.zigzag_ring_flash_attn_func(q, k, v, causal=True, **kw)
            elif kind == "ring":
                out, lse, _ = R.ring_flash_attn_func(q, k, v, causal=c["causal"], **kw)
            elif kind == "stripe":
                out, lse, _ = R.stripe_flash_attn_func(q, k, v, causal=True, **kw)
            elif kind == "zigzag_varlen":
                out, lse, _ = R.zigzag_ring_flash_attn_varlen_func(q, k, v, extra["cu_local"].to(dev), extra["max_local"], causal=True, **kw)
            elif kind == "ring_varlen":
                out, lse, _ = R.ring_flash_attn_varlen_func(q, k, v, extra["cu_local"].to(dev), extra["max_local"], causal=c["causal"], **kw)
            elif kind == "llama3":
                cu = torch.tensor(c["cu"], dtype=torch.int32)
                cq, ck, mq, mk, sl = R.llama3_flash_attn_prepare_cu_seqlens(cu, True, rank, W)
                if cq.tolist() != ref["cu_q"].tolist() or ck.tolist() != ref["cu_k"].tolist() or (sl.start, sl.stop) != tuple(ref["k_slice"]):
                    errs.append(f"{n}: prepare_cu_seqlens mismatch")
                l3_args, l3_kw = (cq.to(dev), ck.to(dev), mq, mk), dict(heads_k_stride=c["stride"], local_k_slice=sl, causal=True, **kw)
                out, lse, _ = R.llama3_flash_attn_varlen_func(q, k, v, *l3_args, **l3_kw)
            out.backward(do)
            if compiled and kind in ("zigzag", "ring", "zigzag_varlen", "llama3"):
                # same call without torch.compile: identical bits
                efn = {"zigzag": lambda: eager_R.zigzag_ring_flash_attn_func(qe, ke, ve, causal=True, **kw),
                       "ring": lambda: eager_R.ring_flash_attn_func(qe, ke, ve, causal=c["causal"], **kw),
                       "zigzag_varlen": lambda: eager_R.zigzag_ring_flash_attn_varlen_func(
                           qe, ke, ve, extra["cu_local"].to(dev), extra["max_local"], causal=True, **kw),
                       "llama3": lambda: eager_R.llama3_flash_attn_varlen_func(
                           qe, ke, ve, cq.to(dev), ck.to(dev), mq, mk, heads_k_stride=c["stride"], local_k_slice=sl,
                           causal=True, **kw)}[kind]
                qe, ke, ve = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
                # (the operator form of a multi-rank schedule has no forward-to-backward hand-over of gathered K/V: its
                #  backward gathers again and runs the local block FIRST — the eager call without the hand-over — where
                #  the eager default runs it last, beside the all-to-all: another fp32 summation order of dQ)
                from ring_flash_attn import config as _cfg
                import contextlib

                with (_cfg.override(kv_keep=False) if kind == "zigzag" else contextlib.nullcontext()):
                    oe, le, _ = efn()
                    oe.backward(do)
                for nm, a_, b_ in (("out", out, oe), ("lse", lse, le), ("dq", q.grad, qe.grad), ("dk", k.grad, ke.grad),
                                   ("dv", v.grad, ve.grad)):
                    if not torch.equal(a_.detach(), b_.detach()):
                        errs.append(f"{n}[r{rank}].{nm}: compiled caller differs from the eager call")

            def pick(t):
                # sampled cases store every `sample`-th row (+ first / last 8) of the row-indexed tensors
                if "sample" not in c:
                    return t
                dim = 0 if "cu" in c else 1
                return t.index_select(dim, MG.sample_rows(t.shape[dim], c["sample"]))

            _cmp(f"{n}[r{rank}].out", pick(out.detach().cpu()), ref["out"], tol["out"], errs)
            _cmp(f"{n}[r{rank}].lse", lse.detach().cpu(), ref["lse"], tol["lse"], errs)
            _cmp(f"{n}[r{rank}].dq", pick(q.grad.cpu()), ref["dq"], tol["grad"], errs)
            _cmp(f"{n}[r{rank}].dk", pick(k.grad.cpu()), ref["dk"], tol["grad"], errs)
            _cmp(f"{n}[r{rank}].dv", pick(v.grad.cpu()), ref["dv"], tol["grad"], errs)
            if out.dtype != torch.bfloat16 or q.grad.dtype != torch.bfloat16 or lse.dtype != torch.float32:
                errs.append(f"{n}: output dtypes {out.dtype} {q.grad.dtype} {lse.dtype}")
            if kind in ("zigzag_varlen", "ring_varlen"):
                # packed entry point of the varlen schedules (own autograd Function: packed gradient buffer)
                fnp = R.zigzag_ring_flash_attn_varlen_kvpacked_func if kind == "zigzag_varlen" else R.ring_flash_attn_varlen_kvpacked_func
                q2 = q.detach().clone().requires_grad_(True)
                kv = torch.stack([k.detach(), v.detach()], dim=1).requires_grad_(True)
                out2, lse2, _ = fnp(q2, kv, extra["cu_local"].to(dev), extra["max_local"],
                                    causal=True if kind == "zigzag_varlen" else c["causal"], **kw)
                out2.backward(do)
                _cmp(f"{n}[r{rank}].kvpacked.out", pick(out2.detach().cpu()), ref["out"], tol["out"], errs)
                _cmp(f"{n}[r{rank}].kvpacked.dq", pick(q2.grad.cpu()), ref["dq"], tol["grad"], errs)
                _cmp(f"{n}[r{rank}].kvpacked.dk", pick(kv.grad[:, 0].cpu()), ref["dk"], tol["grad"], errs)
                _cmp(f"{n}[r{rank}].kvpacked.dv", pick(kv.grad[:, 1].cpu()), ref["dv"], tol["grad"], errs)
            if kind == "llama3" and not via_reference:
                # packed entry points (own autograd Functions: K / V are views of the packed tensor, their gradients are
                # written into one packed gradient)
                q2 = q.detach().clone().requires_grad_(True)
                kv = torch.stack([k.detach(), v.detach()], dim=1).requires_grad_(True)
                out2, lse2, _ = R.llama3_flash_attn_varlen_kvpacked_func(q2, kv, *l3_args, **l3_kw)
                out2.backward(do)
                _cmp(f"{n}[r{rank}].kvpacked.out", pick(out2.detach().cpu()), ref["out"], tol["out"], errs)
                _cmp(f"{n}[r{rank}].kvpacked.dq", pick(q2.grad.cpu()), ref["dq"], tol["grad"], errs)
                _cmp(f"{n}[r{rank}].kvpacked.dk", pick(kv.grad[:, 0].cpu()), ref["dk"], tol["grad"], errs)
                _cmp(f"{n}[r{rank}].kvpacked.dv", pick(kv.grad[:, 1].cpu()), ref["dv"], tol["grad"], errs)
                if q.shape[1] == k.shape[1]:
                    qkv = torch.stack([q.detach(), k.detach(), v.detach()], dim=1).requires_grad_(True)
                    out3, _, _ = R.llama3_flash_attn_varlen_qkvpacked_func(qkv, *l3_args, **l3_kw)
                    out3.backward(do)
                    _cmp(f"{n}[r{rank}].qkvpacked.out", pick(out3.detach().cpu()), ref["out"], tol["out"], errs)
                    for i_, nm in enumerate(("dq", "dk", "dv")):
                        _cmp(f"{n}[r{rank}].qkvpacked.{nm}", pick(qkv.grad[:, i_].cpu()), ref[nm], tol["grad"], errs)
            if kind == "zigzag":
                # the kvpacked entry point: K/V travel as ONE packed buffer, dK/dV land in the packed gradient
                q2 = q.detach().clone().requires_grad_(True)
                kv = torch.stack([k.detach(), v.detach()], dim=2).requires_grad_(True)
                out2, lse2, _ = R.zigzag_ring_flash_attn_kvpacked_func(q2, kv, causal=True, **kw)
                out2.backward(do)
                _cmp(f"{n}[r{rank}].kvpacked.out", pick(out2.detach().cpu()), ref["out"], tol["out"], errs)
                _cmp(f"{n}[r{rank}].kvpacked.lse", lse2.detach().cpu(), ref["lse"], tol["lse"], errs)
                _cmp(f"{n}[r{rank}].kvpacked.dq", pick(q2.grad.cpu()), ref["dq"], tol["grad"], errs)
                _cmp(f"{n}[r{rank}].kvpacked.dk", pick(kv.grad[:, :, 0].cpu()), ref["dk"], tol["grad"], errs)
                _cmp(f"{n}[r{rank}].kvpacked.dv", pick(kv.grad[:, :, 1].cpu()), ref["dv"], tol["grad"], errs)
        ret[rank] = errs
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        ret[rank] = ["EXC: " + traceback.format_exc()]


def run_world(W, names, use_hip, port, via_reference=False):
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(run_rank, args=(W, port, names, use_hip, ret, via_reference), nprocs=W, join=True)
    errs = []
    for r in range(W):
        errs += list(ret.get(r, [f"rank {r} produced no result"]))
    return errs
