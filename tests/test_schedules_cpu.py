"""Host logic of the ring schedules under gloo (world_size 2 and 4) on CPU.

The package's schedules (zigzag / ring / varlen / llama3; forward AND backward, including the
two-phase dK/dV accumulate and the K/V + dK/dV ring rotation) run with the CPU oracle injected as
operator backend and must reproduce the golden vectors that the UNMODIFIED reference produced for
the same seeded inputs (tests/golden/ring_golden.pt).  This checks everything above the C ABI.
"""
import pytest

from conftest import free_port
import _ring_worker as RW
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as MG


@pytest.mark.parametrize("W", [2, 3, 4, 6, 8])
def test_schedules_match_reference_golden(W):
    names = [n for n, c in MG.CASES.items() if c["W"] == W]
    assert names
    errs = RW.run_world(W, names, use_hip=False, port=free_port())
    assert not errs, "\n".join(errs)


@pytest.mark.parametrize("W", [2, 3, 4, 8])
def test_zigzag_fp32_wire_matches_golden(W, monkeypatch):
    """gather exchange with fp32 dK/dV contributions + reduce-scatter (RFA_DKV_WIRE=fp32; the default sends the
    io dtype and sums at the owner, exercised by the test above)."""
    monkeypatch.setenv("RFA_ZIGZAG_EXCHANGE", "gather")
    monkeypatch.setenv("RFA_DKV_WIRE", "fp32")
    names = [n for n, c in MG.CASES.items() if c["W"] == W and c["kind"] == "zigzag"]
    errs = RW.run_world(W, names, use_hip=False, port=free_port())
    assert not errs, "\n".join(errs)


@pytest.mark.parametrize("W", [2, 4])
def test_zigzag_gather_without_the_kv_cache_matches_golden(W, monkeypatch):
    """RFA_ZIGZAG_KV_KEEP=0: the backward gathers K/V again and keeps the local-block-first order (the default —
    K/V kept from the forward, remote steps first, local block beside the all-to-all — is what the tests above
    run)."""
    monkeypatch.setenv("RFA_ZIGZAG_EXCHANGE", "gather")
    monkeypatch.setenv("RFA_ZIGZAG_KV_KEEP", "0")
    names = [n for n, c in MG.CASES.items() if c["W"] == W and c["kind"] == "zigzag"]
    errs = RW.run_world(W, names, use_hip=False, port=free_port())
    assert not errs, "\n".join(errs)


def test_zigzag_under_activation_checkpointing_matches_golden(monkeypatch):
    """torch.utils.checkpoint around the zigzag call (gather form): forward without grad, recomputation forward,
    backward that reuses the recomputation's gathered K/V"""
    monkeypatch.setenv("RFA_ZIGZAG_EXCHANGE", "gather")
    monkeypatch.setenv("RFA_TEST_CHECKPOINT", "1")
    names = [n for n, c in MG.CASES.items() if c["W"] == 4 and c["kind"] == "zigzag"]
    errs = RW.run_world(4, names, use_hip=False, port=free_port())
    assert not errs, "\n".join(errs)


def test_kept_kv_is_owned_by_the_autograd_graph():
    """the K/V gathered by the zigzag forward are SAVED TENSORS of its autograd node: present exactly when a backward
    can follow (not under torch.no_grad(), not for inputs without requires_grad), freed with the graph, not kept over
    config.kv_keep_bytes / with config.kv_keep = False / beyond the process-wide budget config.kv_keep_total_bytes of all
    pending backwards (the backward then gathers again, same gradients).  The only process-global state is that byte
    count."""
    import torch.multiprocessing as mp
    import _kv_cache_worker as KW
    from ring_flash_attn import zigzag_ring_flash_attn as Z, utils as U
    from ring_flash_attn import _testing

    assert not hasattr(Z, "_KV_CACHE") and not hasattr(U, "_BACKWARD_EXPECTED") and not hasattr(U, "_GRAD_MODE_AT_CALL")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(KW.run, args=(2, free_port(), ret), nprocs=2, join=True)
    want = [6, True, None, None, 5, True, 5, True, 6, 5, True, True, 6, True]
    assert ret[0] == want and ret[1] == want, dict(ret)


@pytest.mark.timeout(400)
@pytest.mark.parametrize("W,form", [(2, "gather"), (4, "gather"), (3, "gather_ps")])
def test_collective_sequence_survives_rank_local_state(W, form):
    """the keep / re-gather choice of the zigzag gather form reads a process-local byte budget, the
    exchange form a process-local tuning record — ranks whose local state disagrees must still post the SAME collectives.
    W gloo ranks; one rank's budget exhausted (by configuration — every rank in turn —, and by a graph only rank 0 keeps
    alive), a tuning record on one rank only, different records, the same record: every call finishes, the posted
    collectives are identical on all ranks, and the gradients are those of the symmetric run (bit for bit)."""
    import torch.multiprocessing as mp
    import _consistency_worker as CW

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(CW.run, args=(W, free_port(), ret, form), nprocs=W, join=True)
    res = [ret[r] for r in range(W)]
    kept_path = ["all_gather", "all_to_all"]                        # forward gathers; the backward only returns dK/dV
    regather_path = ["all_gather", "all_gather", "all_to_all"]      # ... and gathers K/V a second time
    for r in res:
        assert r["both_keep"] == (kept_path, 6) and r["none_keep"] == (regather_path, 5), r
        assert r["close_keep_vs_not"]
        for name in [f"budget_rank{x}" for x in range(W)] + ["held_graph"]:
            assert r[name]["posted"] == regather_path and r[name]["same_as_no_keep"], (name, r[name])
        for name in ("record_rank0_only", "records_differ"):
            assert r[name]["posted"] == kept_path and r[name]["same_as_keep"], (name, r[name])
            assert r[name]["mine_after"] in (None, "-")
        assert set(r["record_everywhere"]["posted"]) == {"hop"}, r["record_everywhere"]
        # a record installed on one rank AFTER the group agreed is parked (the agreed "ring" keeps deciding, on
        # every rank) until the collective sync_records(); then the ranks' records differ and the shape rule decides
        assert set(r["late_record_rank0"]["posted"]) == {"hop"} and r["late_record_rank0"]["mine_after"] == "ring", r["late_record_rank0"]
        assert r["after_sync_records"]["posted"] == kept_path and r["after_sync_records"]["mine_after"] is None, r["after_sync_records"]
        chosen, cached, mine = r["autotune_after_local_record"]["mine_after"]
        assert chosen in ("gather", "gather_ps", "ring") and cached and mine == chosen, r["autotune_after_local_record"]
        assert set(r["autotune_after_local_record"]["posted"]) == ({"hop"} if chosen == "ring" else set(kept_path))
    # the ranks that could keep did save their buffers (6 saved tensors), the loser not (5): the DECISION was the group's
    for loser in range(W):
        assert [res[r][f"budget_rank{loser}"]["n_saved"] for r in range(W)] == [5 if r == loser else 6 for r in range(W)]
    assert [res[r]["held_graph"]["n_saved"] for r in range(W)] == [5] + [6] * (W - 1)
    assert res[0]["record_rank0_only"]["mine_after"] is None
    assert len({r["autotune_after_local_record"]["mine_after"][0] for r in res}) == 1      # one winner for the whole group


@pytest.mark.parametrize("W", [2, 3])
def test_exchange_audit_passes_on_every_schedule(W, monkeypatch):
    """config.exchange_check on (RFA_EXCHANGE_CHECK=1): every schedule of the package — ring, zigzag in its three exchange
    forms, stripe, both varlen forms, llama3 (all-gather; its reduce-scatter has no single sender and is not audited),
    zigzag-llama3 — passes the audit and reproduces the golden vectors of the unmodified reference."""
    monkeypatch.setenv("RFA_EXCHANGE_CHECK", "1")
    names = [n for n, c in MG.CASES.items() if c["W"] == W and "sample" not in c]
    assert names
    for mode in ("gather", "ring", "gather_ps"):
        monkeypatch.setenv("RFA_ZIGZAG_EXCHANGE", mode)
        sel = names if mode == "gather" else [n for n in names if MG.CASES[n]["kind"] == "zigzag"]
        errs = RW.run_world(W, sel, use_hip=False, port=free_port())
        assert not errs, "\n".join(errs)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("W", [2, 3])
def test_exchange_check_names_a_corrupted_receive(W):
    """the opt-in debug mode (config.exchange_check / RFA_EXCHANGE_CHECK=1) checksums every K/V and dK/dV
    buffer a rank receives against its sender's checksum (utils.audit_verify: one extra tiny all-gather per schedule call).
    Clean calls pass — bit-identical to the unchecked call — in all three zigzag exchange forms and in the ring / stripe
    schedules; a receive buffer overwritten after it landed (ring_flash_attn._testing.corrupt_receive) makes THAT rank raise
    with its rank, the step and the buffer named, its peers finish, and the next call is clean again."""
    import torch.multiprocessing as mp
    import _audit_worker as AW

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(AW.run, args=(W, free_port(), ret), nprocs=W, join=True)
    for rank in range(W):
        r = ret[rank]
        for name in ("clean_ring", "clean_gather", "clean_gather_ps", "clean_ring_func", "clean_stripe_func", "after_ring", "after_gather_ps"):
            assert r[name] is True, (rank, name, r)
        if rank == W - 1:
            assert f"FAILED on rank {rank}" in r["corrupt_ring"] and "ring hop" in r["corrupt_ring"] and "sender rank" in r["corrupt_ring"], r["corrupt_ring"]
            assert f"FAILED on rank {rank}" in r["corrupt_gather_ps"] and "per-source exchange 1" in r["corrupt_gather_ps"], r["corrupt_gather_ps"]
        else:
            assert r["corrupt_ring"] == "no error" and r["corrupt_gather_ps"] == "no error", r


def test_bench_launches_its_own_ranks():
    """VERDICT r4 missing #2: the driver runs `python bench.py --gpus N`; without a launcher's WORLD_SIZE the script
    starts its own N ranks under torch.distributed.run (as the reference's benchmark is started with torchrun,
    /root/reference/README.md:135-147).  On a box without a GPU both ranks must get as far as the operator's
    'needs a GPU' exit — i.e. the launch itself works; with a mismatching WORLD_SIZE it says how to start it."""
    import subprocess
    import sys as _sys
    import torch

    if torch.cuda.is_available():
        pytest.skip("CPU-box check of the launcher; the GPU twin is tests/test_gpu_rccl_world1.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([_sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert p.stderr.count("bench.py needs a GPU") == 2, p.stderr[-2000:]
    env["WORLD_SIZE"] = "4"
    p = subprocess.run([_sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert p.returncode != 0 and "launches its own ranks" in p.stderr


def test_packed_pair_recognises_only_adjacent_halves_of_one_buffer():
    """_common.packed_pair: the kv argument of the kvpacked entry points (dense and varlen) and its gradient are
    moved / summed as ONE buffer; anything else (separate tensors, qkv-packed slices, swapped halves, a
    non-contiguous parent) is not"""
    import torch
    from ring_flash_attn._common import packed_pair

    kv = torch.randn(2, 16, 2, 3, 8)
    got = packed_pair(kv[:, :, 0], kv[:, :, 1])
    assert got is not None and got.shape == kv.shape and got.data_ptr() == kv.data_ptr() and torch.equal(got, kv)
    kvt = torch.randn(40, 2, 3, 8)                                    # packed sequences (T, 2, Hk, D)
    got = packed_pair(kvt[:, 0], kvt[:, 1])
    assert got is not None and torch.equal(got, kvt)
    off = torch.randn(3, 2, 16, 2, 3, 8)[1]                           # a contiguous sub-block with a storage offset
    got = packed_pair(off[:, :, 0], off[:, :, 1])
    assert got is not None and torch.equal(got, off)
    assert packed_pair(kv[:, :, 1], kv[:, :, 0]) is None              # swapped
    assert packed_pair(kv[:, :, 0].contiguous(), kv[:, :, 1].contiguous()) is None
    qkv = torch.randn(2, 16, 3, 3, 8)
    assert packed_pair(qkv[:, :, 1], qkv[:, :, 2]) is None            # row stride 3 Hk D: not a kv pair
    wide = torch.randn(2, 16, 2, 3, 16)[..., :8]                      # head_dim slice of a wider buffer
    assert packed_pair(wide[:, :, 0], wide[:, :, 1]) is None
    assert packed_pair(kv[:, ::2, 0], kv[:, ::2, 1]) is None          # strided rows


@pytest.mark.parametrize("W", [2, 3, 4])
def test_zigzag_varlen_gather_exchange_matches_golden(W, monkeypatch):
    """RFA_ZIGZAG_VARLEN_EXCHANGE=gather: the dense zigzag path's mesh-aware exchange for packed sequences"""
    monkeypatch.setenv("RFA_ZIGZAG_VARLEN_EXCHANGE", "gather")
    names = [n for n, c in MG.CASES.items() if c["W"] == W and c["kind"] == "zigzag_varlen"]
    assert names
    errs = RW.run_world(W, names, use_hip=False, port=free_port())
    assert not errs, "\n".join(errs)


@pytest.mark.parametrize("break_gather", [False, True])
def test_exchange_autotune_is_a_collective_decision(break_gather):
    """ring_flash_attn.tuning.autotune_zigzag_exchange: times fwd+bwd of the zigzag schedule in both exchange forms on
    the real group, every rank records the SAME winner (max over ranks), exchange_mode('auto') then follows the record
    for exactly those shapes, an explicit RFA_ZIGZAG_EXCHANGE still wins, and a form that raises (an unsupported
    collective) is disqualified instead of sinking the job (round-2 review: the default form had never run on the
    driver's 8-GPU node)."""
    import torch.multiprocessing as mp
    import _tuning_worker as TW

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(TW.run, args=(2, free_port(), ret, break_gather), nprocs=2, join=True)
    a, b = ret[0], ret[1]
    assert a["before"] == b["before"] == "gather"                 # the shape rule
    assert a["chosen"] == b["chosen"] == a["after"] == b["after"]
    assert a["forced"] == b["forced"] == "ring" and a["other_shape"] == "gather"
    assert a["ms"] == b["ms"]                                     # max over ranks: identical on both
    # the opt-in in-call measurement: a SECOND, independent measurement — both ranks take the same decision again (which of
    # three forms a few microseconds apart wins on CPU timings may differ between two measurements; between ranks never)
    assert a["in_call"] == b["in_call"] and a["in_call"] in ("gather", "gather_ps", "ring")
    if break_gather:
        assert a["in_call_none"] == b["in_call_none"] == [None, None, "gather"]      # every form broken: no raise, the shape rule
        assert a["chosen"] == a["in_call"] == "ring" and a["ms"]["gather"] is None and a["ms"]["gather_ps"] is None
        assert all(f_ in r_["failed"] for f_ in ("gather", "gather_ps") for r_ in (a, b))
    else:
        assert all(v_ is not None and v_ > 0 for v_ in a["ms"].values())
        for kind in ("all_gather", "all_to_all", "neighbour_hop"):
            assert a["probe"][kind]["GBps"] > 0 and a["probe"][kind]["ms"] == b["probe"][kind]["ms"]


@pytest.mark.parametrize("W,stride", [(2, 1), (4, 2)])
def test_llama3_dropout_is_consistent_across_ranks(W, stride):
    """dropout on the llama3 path (the one place the reference forwards dropout_p to flash_attn,
    llama3_flash_attn_varlen.py:131,266): the mask is a function of GLOBAL (head, query position, key position), so W
    ranks — each running its head groups one after the other — reproduce the single-device result with the same seed;
    forward and backward use the same mask (gradients match); the ring schedules raise over several ranks."""
    import torch
    import torch.multiprocessing as mp
    import _dropout_worker as DW
    from oracle import flash_attn_ref as O

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(DW.run, args=(W, free_port(), ret, False, stride), nprocs=W, join=True)
    q, k, v, do = DW.inputs()
    torch.manual_seed(DW.SEED)
    from ring_flash_attn._common import draw_dropout_seed

    rng = torch.tensor([draw_dropout_seed(), 0])
    cu = torch.tensor(DW.CU, dtype=torch.int32)
    scale = DW.D ** -0.5
    ro, rl, _, _ = O._flash_attn_varlen_forward(q, k, v, cu, cu, 0, 0, DW.P_DROP, scale, True, rng_state=rng)
    r0 = O._flash_attn_varlen_forward(q, k, v, cu, cu, 0, 0, 0.0, scale, True)[0]
    assert (ro.float() - r0.float()).abs().max() > 0.05            # dropout did something
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_varlen_backward(do, q, k, v, ro, rl, dq, dk, dv, cu, cu, 0, 0, DW.P_DROP, scale, True, rng_state=rng)
    T = DW.CU[-1] // W
    for r in range(W):
        got = ret[r]
        assert not isinstance(got, str), got
        sl = slice(r * T, (r + 1) * T)
        assert got["raised"] == [True, True, True]
        assert (got["out"].float() - ro[sl].float()).abs().max() <= 2e-2
        for name, ref in (("dq", dq), ("dk", dk), ("dv", dv)):
            d = (got[name].float() - ref[sl].float()).abs().max().item()
            assert d <= 3e-2 + 1e-2 * ref.float().abs().max().item(), f"W={W} r{r} {name}: {d:.3e}"


def test_dropout_argument_checks(single_rank_group):
    """dropout_p outside [0, 1), dropout together with a window, and a dropout call at world size 1 through every
    public function family (oracle backend)"""
    import torch
    import ring_flash_attn as R
    from ring_flash_attn import backend
    from ring_flash_attn import _testing
    from oracle.oracle_backend import OracleBackend

    _testing.set_backend(OracleBackend())
    try:
        g = torch.Generator().manual_seed(1)
        q = torch.randn(1, 64, 2, 32, generator=g).bfloat16().requires_grad_(True)
        with pytest.raises(ValueError):
            R.ring_flash_attn_func(q, q, q, dropout_p=1.0, causal=True)
        with pytest.raises(NotImplementedError):
            R.ring_flash_attn_func(q, q, q, dropout_p=0.1, causal=True, window_size=(8, 0))
        base = R.zigzag_ring_flash_attn_func(q, q, q, causal=True)
        for fn in (R.ring_flash_attn_func, R.zigzag_ring_flash_attn_func, R.stripe_flash_attn_func):
            torch.manual_seed(3)
            a = fn(q, q, q, dropout_p=0.3, causal=True)
            torch.manual_seed(3)
            b = fn(q, q, q, dropout_p=0.3, causal=True)
            c = fn(q, q, q, dropout_p=0.3, causal=True)          # next draw of the generator: another mask
            assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, base)
            a.sum().backward()
            assert torch.isfinite(q.grad).all()
        cu = torch.tensor([0, 24, 64], dtype=torch.int32)
        qv = torch.randn(64, 2, 32, generator=g).bfloat16()
        for fn in (R.ring_flash_attn_varlen_func, R.zigzag_ring_flash_attn_varlen_func):
            torch.manual_seed(5)
            a = fn(qv, qv, qv, cu, 40, dropout_p=0.3, causal=True)
            assert not torch.equal(a, fn(qv, qv, qv, cu, 40, causal=True))
    finally:
        _testing.set_backend(None)


def test_exchange_mode_auto_threshold(monkeypatch):
    """auto = gather while the O(S_total) scratch fits RFA_GATHER_MAX_BYTES, ring beyond"""
    import torch
    from ring_flash_attn.zigzag_ring_flash_attn import exchange_mode, gather_scratch_bytes

    k = torch.empty(1, 8192, 8, 128, dtype=torch.bfloat16, device="meta")
    monkeypatch.delenv("RFA_ZIGZAG_EXCHANGE", raising=False)
    monkeypatch.delenv("RFA_DKV_WIRE", raising=False)
    assert gather_scratch_bytes(k, 8, False) == 8 * 2 * k.numel() * 4          # 0.27 + 0.27 GB
    assert exchange_mode(k, 8) == "gather"
    big = torch.empty(1, 131072, 8, 128, dtype=torch.bfloat16, device="meta")   # 128K tokens per rank
    assert exchange_mode(big, 8) == "ring"
    monkeypatch.setenv("RFA_GATHER_MAX_BYTES", "1000")
    assert exchange_mode(k, 8) == "ring"
    monkeypatch.setenv("RFA_ZIGZAG_EXCHANGE", "gather")
    assert exchange_mode(big, 8) == "gather"


@pytest.mark.parametrize("W", [2, 3, 4, 6, 8])
def test_zigzag_ring_exchange_matches_golden(W, monkeypatch):
    """RFA_ZIGZAG_EXCHANGE=ring (the reference's hop-by-hop protocol; small shapes default to the mesh-aware
    gather form exercised by the test above) gives the same golden results."""
    monkeypatch.setenv("RFA_ZIGZAG_EXCHANGE", "ring")
    names = [n for n, c in MG.CASES.items() if c["W"] == W and c["kind"] == "zigzag"]
    assert names
    errs = RW.run_world(W, names, use_hip=False, port=free_port())
    assert not errs, "\n".join(errs)


@pytest.mark.parametrize("W", [2, 3, 4, 8])
def test_zigzag_per_source_arrival_matches_golden(W, monkeypatch):
    """RFA_ZIGZAG_EXCHANGE=gather_ps (round 6): the gather form with the W - 1 K/V exchanges posted at
    once in consumption order, step s waiting for source (r - s) mod W only (utils.SourceArrivals) — the reference's
    "step s starts when hop s has landed" (/root/reference/ring_flash_attn/zigzag_ring_flash_attn.py:60-84).  Same
    kernels, same merge order: the golden vectors of the unmodified reference, incl. an odd world size."""
    monkeypatch.setenv("RFA_ZIGZAG_EXCHANGE", "gather_ps")
    names = [n for n, c in MG.CASES.items() if c["W"] == W and c["kind"] == "zigzag"]
    assert names
    errs = RW.run_world(W, names, use_hip=False, port=free_port())
    assert not errs, "\n".join(errs)


@pytest.mark.parametrize("W", [2, 4])
def test_schedules_under_torch_compile_at_world_size_gt_1(W, monkeypatch):
    """the reference's second test pass (test/test.sh:23-25: every test again with the function compiled, at the full
    world size; default backend = inductor, as there).  A multi-rank schedule is captured as ONE registered operator per
    direction (`fullgraph=True`: no graph break); a compiled caller must reproduce the golden vectors AND the plain
    call bit for bit — for both exchange forms of the zigzag path."""
    monkeypatch.setenv("RFA_TEST_COMPILE", "1")
    names = [n for n, c in MG.CASES.items() if c["W"] == W and "sample" not in c]
    assert names
    for mode in ("gather", "ring", "gather_ps"):
        monkeypatch.setenv("RFA_ZIGZAG_EXCHANGE", mode)
        sel = names if mode == "gather" else [n for n in names if MG.CASES[n]["kind"] == "zigzag"]
        errs = RW.run_world(W, sel, use_hip=False, port=free_port())
        assert not errs, "\n".join(errs)


def test_unmodified_reference_runs_on_the_flash_attn_shim():
    """INTEGRATION.md route B: the reference's own schedule code (loaded unmodified from /root/reference)
    on top of the shipped `flash_attn` compatibility package reproduces the golden vectors.  Build
    container only (the reference tree does not travel); the kernels underneath are the CPU oracle
    here — the same shim over the HIP kernels is covered by tests/test_gpu_flash_attn_shim.py."""
    from oracle import reference_harness

    if not reference_harness.available():
        pytest.skip("/root/reference not present")
    names = [n for n, c in MG.CASES.items() if c["W"] == 2]
    errs = RW.run_world(2, names, use_hip=False, port=free_port(), via_reference=True)
    assert not errs, "\n".join(errs)


def test_ringcomm_guards(single_rank_group):
    """RingComm keeps the reference's state guards (utils.py:129-136)."""
    from ring_flash_attn.utils import RingComm

    c = RingComm(None)
    with pytest.raises(RuntimeError, match="wait called before commit"):
        c.wait()
    c._reqs = []
    with pytest.raises(RuntimeError, match="commit called twice"):
        c.commit()


def test_torch_compile_captures_custom_ops(single_rank_group):
    """the reference runs every test a second time under torch.compile (test/test.sh:23-25).  On a single-rank
    group the public functions lower to the registered custom operators rfa::attn_fwd / rfa::attn_bwd (fake
    kernels + autograd formula, ring_flash_attn/_ops.py): `fullgraph=True` proves there is NO graph break, the
    captured graph contains the operator, and results / gradients equal the eager path bit for bit."""
    import torch
    import ring_flash_attn as R
    from ring_flash_attn import backend
    from ring_flash_attn import _testing
    from oracle.oracle_backend import OracleBackend

    _testing.set_backend(OracleBackend())
    try:
        g = torch.Generator().manual_seed(3)
        qkv = torch.randn(1, 32, 3, 2, 16, generator=g).to(torch.bfloat16)
        do = torch.randn(1, 32, 2, 16, generator=g).to(torch.bfloat16)
        torch._dynamo.reset()
        torch._dynamo.config.capture_scalar_outputs = True
        graphs = []

        def spy(gm, example_inputs):
            graphs.append(gm)
            return gm.forward

        for fn in (R.zigzag_ring_flash_attn_qkvpacked_func, R.ring_flash_attn_qkvpacked_func,
                   R.stripe_flash_attn_qkvpacked_func):
            xe = qkv.clone().requires_grad_(True)
            eager, lse_e, _ = fn(xe, causal=True, return_attn_probs=True)
            eager.backward(do)
            xc = qkv.clone().requires_grad_(True)
            out, lse, none = torch.compile(fn, backend=spy, fullgraph=True)(xc, causal=True, return_attn_probs=True)
            out.backward(do)
            assert none is None and torch.equal(out, eager) and torch.equal(lse, lse_e)
            assert torch.equal(xc.grad, xe.grad)
        assert graphs and all(any("rfa.attn_fwd" in str(n.target) for n in gm.graph.nodes) for gm in graphs)

        # packed varlen form, dense kv-packed form with GQA, and a real inductor compile of a caller
        cu = torch.tensor([0, 12, 32], dtype=torch.int32)
        pv = qkv[0].clone().requires_grad_(True)
        ev = R.zigzag_ring_flash_attn_varlen_qkvpacked_func(pv, cu, 20, causal=True)
        cv = torch.compile(R.zigzag_ring_flash_attn_varlen_qkvpacked_func, backend=spy, fullgraph=True)(pv, cu, 20, causal=True)
        assert torch.equal(ev, cv)

        def model(q, kv):
            return (R.ring_flash_attn_kvpacked_func(q * 0.5, kv, causal=False) * 2.0).float().sum()

        q = torch.randn(1, 24, 4, 16, generator=g).to(torch.bfloat16)
        kv = torch.randn(1, 24, 2, 2, 16, generator=g).to(torch.bfloat16)
        qe, kve = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
        model(qe, kve).backward()
        qc, kvc = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
        torch.compile(model, fullgraph=True)(qc, kvc).backward()
        assert (qc.grad.float() - qe.grad.float()).abs().max() <= 1e-2 * qe.grad.float().abs().max()
        assert (kvc.grad.float() - kve.grad.float()).abs().max() <= 1e-2 * kve.grad.float().abs().max()
    finally:
        _testing.set_backend(None)
        torch._dynamo.reset()


def test_config1_ring_qkvpacked_w1_fp32_plumbing(single_rank_group):
    """BASELINE.json configs[0]: `ring_flash_attn_qkvpacked_func`, world_size 1, CPU / gloo, batch 1,
    seq 512, nheads 4, d 64, fp32, causal — the reference's own CPU-runnable plumbing case (fixture style:
    /root/reference/test/test_ring_flash_attn_func.py:17-36, qkv = randn(B, S, 3, H, D)).  The product
    has no CPU / fp32 compute path by design, so the operator backend is the oracle (test hook); what is
    checked is everything above the C ABI at this shape and dtype: packed-qkv autograd Function, W = 1
    short-circuit, dtype plumbing (fp32 in -> fp32 out / lse / one packed fp32 gradient), against an
    independent fp64 softmax attention with autograd."""
    import torch
    import ring_flash_attn as R
    from ring_flash_attn import backend
    from ring_flash_attn import _testing
    from oracle import flash_attn_ref as O
    from oracle.oracle_backend import OracleBackend

    _testing.set_backend(OracleBackend())
    try:
        g = torch.Generator().manual_seed(0)
        B, S, H, D = 1, 512, 4, 64
        qkv = torch.randn(B, S, 3, H, D, generator=g, dtype=torch.float32).requires_grad_(True)
        dout = torch.randn(B, S, H, D, generator=g, dtype=torch.float32)
        out, lse, _ = R.ring_flash_attn_qkvpacked_func(qkv, dropout_p=0.0, causal=True, window_size=(-1, -1),
                                                        alibi_slopes=None, deterministic=False,
                                                        return_attn_probs=True)
        out.backward(dout)
        assert out.dtype == torch.float32 and out.shape == (B, S, H, D)
        assert lse.dtype == torch.float32 and lse.shape == (B, H, S)
        assert qkv.grad.dtype == torch.float32 and qkv.grad.shape == qkv.shape

        ref = qkv.detach().double().requires_grad_(True)
        ro, rl = O.full_attention_fp64(ref[:, :, 0], ref[:, :, 1], ref[:, :, 2], True, D ** -0.5)
        ro.backward(dout.double())
        assert (out.double() - ro).abs().max() < 2e-5
        assert (lse.double() - rl).abs().max() < 2e-5
        assert (qkv.grad.double() - ref.grad).abs().max() < 5e-5 * max(1.0, ref.grad.abs().max().item())
    finally:
        _testing.set_backend(None)


@pytest.mark.parametrize("W", [2, 3, 4, 6])
def test_llama3_unfused_groups_match_golden(W, monkeypatch):
    """RFA_LLAMA3_GATHER_MAX_BYTES=0 disables the super-group fusion: one all-gather / launch / reduce-scatter
    per `heads_k_stride` group, double-buffered (the default fuses every group of these small cases into one)."""
    monkeypatch.setenv("RFA_LLAMA3_GATHER_MAX_BYTES", "0")
    names = [n for n, c in MG.CASES.items() if c["W"] == W and c["kind"] == "llama3"]
    assert names
    errs = RW.run_world(W, names, use_hip=False, port=free_port())
    assert not errs, "\n".join(errs)


def _llama3_groups_rank(rank, W, port, ret):
    import os
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    import ring_flash_attn as R
    from ring_flash_attn import backend, config
    from ring_flash_attn import _testing
    from oracle.oracle_backend import OracleBackend

    _testing.set_backend(OracleBackend())
    g = torch.Generator().manual_seed(77)
    T, H, Hk, D = 96, 8, 4, 16
    cu = torch.tensor([0, 20, 61, 96], dtype=torch.int32)
    q, k, v, do = (torch.randn(T, h, D, generator=g).to(torch.bfloat16) for h in (H, Hk, Hk, H))
    L = T // W
    sl = slice(rank * L, (rank + 1) * L)
    cq, ck, mq, mk, ks = R.llama3_flash_attn_prepare_cu_seqlens(cu, True, rank, W)
    res = {}
    for budget in ("0", str(1 << 30)):          # 4 groups of one kv head (buffers reused twice) vs one fused group
        config.set(llama3_gather_max_bytes=int(budget))
        ql, kl, vl = (t[sl].clone().requires_grad_(True) for t in (q, k, v))
        out, lse, _ = R.llama3_flash_attn_varlen_func(ql, kl, vl, cq, ck, mq, mk, heads_k_stride=1, local_k_slice=ks,
                                                      causal=True, return_attn_probs=True)
        out.backward(do[sl])
        res[budget] = (out.detach(), lse.detach(), ql.grad, kl.grad, vl.grad)
    a, b = res["0"], res[str(1 << 30)]
    bad = [i for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x, y)]
    # the kvpacked / qkvpacked entry points: the packed tensor travels as ONE buffer (one all-gather, one reduce-scatter
    # straight into the packed gradient) when all heads are one super-group — same bits again
    kvl = torch.stack([k[sl], v[sl]], dim=1).clone().requires_grad_(True)
    ql = q[sl].clone().requires_grad_(True)
    out, lse, _ = R.llama3_flash_attn_varlen_kvpacked_func(ql, kvl, cq, ck, mq, mk, heads_k_stride=1, local_k_slice=ks,
                                                           causal=True, return_attn_probs=True)
    out.backward(do[sl])
    c = (out.detach(), lse.detach(), ql.grad, kvl.grad[:, 0], kvl.grad[:, 1])
    bad += [10 + i for i, (x, y) in enumerate(zip(c, b)) if not torch.equal(x, y)]
    qkvl = torch.stack([q[sl][:, :Hk], k[sl], v[sl]], dim=1).clone().requires_grad_(True)      # (MHA-shaped qkv: H = Hk)
    o3 = R.llama3_flash_attn_varlen_qkvpacked_func(qkvl, cq, ck, mq, mk, heads_k_stride=Hk, local_k_slice=ks, causal=True)
    o3.backward(do[sl][:, :Hk])
    q3, k3, v3 = (t[sl].clone().requires_grad_(True) for t in (q[:, :Hk], k, v))
    o4 = R.llama3_flash_attn_varlen_func(q3, k3, v3, cq, ck, mq, mk, heads_k_stride=Hk, local_k_slice=ks, causal=True)
    o4.backward(do[sl][:, :Hk])
    d = (o3.detach(), qkvl.grad[:, 0], qkvl.grad[:, 1], qkvl.grad[:, 2])
    e = (o4.detach(), q3.grad, k3.grad, v3.grad)
    bad += [20 + i for i, (x, y) in enumerate(zip(d, e)) if not torch.equal(x, y)]
    ret[rank] = bad
    dist.barrier()
    dist.destroy_process_group()


def test_llama3_fused_equals_unfused_four_groups():
    """heads are independent: fusing the kv-head groups into one super-group must not change a single bit
    (out, lse, dq, dk, dv), including across the double-buffer reuse of a 4-group loop"""
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_llama3_groups_rank, args=(2, free_port(), ret), nprocs=2, join=True)
    assert ret[0] == [] and ret[1] == [], dict(ret)


def _llama3_window_rank(rank, W, port, ret):
    import os
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    import ring_flash_attn as R
    from ring_flash_attn import backend
    from ring_flash_attn import _testing
    from oracle import flash_attn_ref as O
    from oracle.oracle_backend import OracleBackend

    _testing.set_backend(OracleBackend())
    g = torch.Generator().manual_seed(78)
    T, H, Hk, D = 96, 4, 2, 16
    cu = torch.tensor([0, 20, 61, 96], dtype=torch.int32)
    q, k, v, do = (torch.randn(T, h, D, generator=g).to(torch.bfloat16) for h in (H, Hk, Hk, H))
    win = (7, 7)
    # single-process oracle on the whole packed batch
    ro, _, _, _ = O._flash_attn_varlen_forward(q, k, v, cu, cu, 0, 0, 0.0, D ** -0.5, True, win[0], win[1])
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    _, rl, _, _ = O._flash_attn_varlen_forward(q, k, v, cu, cu, 0, 0, 0.0, D ** -0.5, True, win[0], win[1])
    O._flash_attn_varlen_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, cu, cu, 0, 0, 0.0, D ** -0.5, True, win[0], win[1])
    L = T // W
    sl = slice(rank * L, (rank + 1) * L)
    cq, ck, mq, mk, ks = R.llama3_flash_attn_prepare_cu_seqlens(cu, True, rank, W)
    ql, kl, vl = (t[sl].clone().requires_grad_(True) for t in (q, k, v))
    out = R.llama3_flash_attn_varlen_func(ql, kl, vl, cq, ck, mq, mk, heads_k_stride=1, local_k_slice=ks, causal=True,
                                          window_size=win)
    out.backward(do[sl])
    errs = []
    for name, got, ref in (("out", out, ro[sl]), ("dq", ql.grad, rdq[sl]), ("dk", kl.grad, rdk[sl]), ("dv", vl.grad, rdv[sl])):
        d = (got.float() - ref.float()).abs().max().item()
        if d > 3e-2:
            errs.append(f"{name}: {d:.3e}")
    # the ring schedules must refuse a window on a multi-rank group (it would be applied per block)
    try:
        R.zigzag_ring_flash_attn_func(q[sl].unsqueeze(0), k[sl].unsqueeze(0), v[sl].unsqueeze(0), causal=True, window_size=win)
        errs.append("zigzag accepted a window on 2 ranks")
    except NotImplementedError:
        pass
    ret[rank] = errs
    dist.barrier()
    dist.destroy_process_group()


def test_llama3_sliding_window_two_ranks():
    """llama3 gathers K/V, so a sliding window is well defined on any group (the reference forwards window_size to
    flash_attn there: llama3_flash_attn_varlen.py:147,282); checked against the single-process oracle"""
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_llama3_window_rank, args=(2, free_port(), ret), nprocs=2, join=True)
    assert ret[0] == [] and ret[1] == [], dict(ret)


def test_single_rank_window_through_public_api(single_rank_group):
    import torch
    import ring_flash_attn as R
    from ring_flash_attn import backend
    from ring_flash_attn import _testing
    from oracle import flash_attn_ref as O
    from oracle.oracle_backend import OracleBackend

    _testing.set_backend(OracleBackend())
    try:
        g = torch.Generator().manual_seed(4)
        qkv = torch.randn(1, 48, 3, 2, 16, generator=g).to(torch.bfloat16).requires_grad_(True)
        out = R.zigzag_ring_flash_attn_qkvpacked_func(qkv, causal=True, window_size=(5, 0))
        out.sum().backward()
        ref, _ = O.full_attention_fp64(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], True, window=(5, 0))
        assert (out.double() - ref).abs().max() < 2e-2 and qkv.grad is not None
    finally:
        _testing.set_backend(None)


@pytest.mark.parametrize("W,case", [
    (1, dict(cu=[0, 40, 41, 128], H=4, Hk=2, D=32, causal=True, seed=61)),
    (2, dict(cu=[0, 22, 75, 128], H=4, Hk=2, D=32, causal=True, seed=62)),
    (3, dict(cu=[0, 50, 51, 132], H=2, Hk=2, D=32, causal=True, seed=63, packed=True)),
    (4, dict(cu=[0, 128], H=4, Hk=1, D=64, causal=True, seed=64)),
    (2, dict(cu=[0, 30, 100, 128], H=2, Hk=2, D=32, causal=False, seed=65)),
    (2, dict(cu=[0, 60, 128], H=2, Hk=1, D=32, causal=True, window=(16, 0), seed=66)),
])
def test_zigzag_llama3_matches_full_packed_attention(W, case):
    """zigzag_llama3_flash_attn_varlen_func (the reference's README TODO, built here): every rank holds slices r and
    2W-1-r of the packed stream; the result must be plain packed-sequence attention over the whole stream.  Sequence
    boundaries inside slices, 1-token sequences, one sequence spanning all slices, GQA, non-causal, a window."""
    import torch
    import _zz_llama3_worker as ZW
    from oracle import flash_attn_ref as O

    q, k, v, do = ZW.make_inputs(case)
    cu = torch.tensor(case["cu"], dtype=torch.int32)
    scale = case["D"] ** -0.5
    win = tuple(case.get("window", (-1, -1)))
    ro, rl, _, _ = O._flash_attn_varlen_forward(q, k, v, cu, cu, 0, 0, 0.0, scale, case["causal"],
                                                window_size_left=win[0], window_size_right=win[1])
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_varlen_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, cu, cu, 0, 0, 0.0, scale, case["causal"],
                                  window_size_left=win[0], window_size_right=win[1])
    res = ZW.run_world(W, case, use_hip=False, port=free_port())
    for r, got in enumerate(res):
        assert not isinstance(got, str), got
        for name, ref, tol in (("out", ro, (2e-2, 0.0)), ("dq", rdq, (3e-2, 1e-2)), ("dk", rdk, (3e-2, 1e-2)), ("dv", rdv, (3e-2, 1e-2))):
            want = ZW.shard(ref.float(), r, W)
            diff = (got[name] - want).abs().max().item()
            assert diff <= tol[0] + tol[1] * want.abs().max().item(), f"W={W} r{r} {name}: {diff:.3e}"
        want = ZW.shard(rl.transpose(0, 1).contiguous(), r, W).transpose(0, 1)
        assert (got["lse"] - want).abs().max().item() <= 1e-4, f"W={W} r{r} lse"

