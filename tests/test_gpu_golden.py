"""GPU parity against the committed golden fixtures: W ranks (processes) share the single MI355X
of the test box and exchange K/V and dK/dV through gloo (host-staged) — the full multi-rank HIP
path: fused merge epilogues, half-selection, two-phase dK/dV accumulation, all-gather /
reduce-scatter — compared with what the UNMODIFIED reference produced for the same inputs."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))


@pytest.mark.parametrize("W", [2, 4, 8])
def test_multirank_hip_matches_reference_golden(W):
    import _ring_worker as RW
    import make_golden as MG
    from conftest import free_port

    names = [n for n, c in MG.CASES.items() if c["W"] == W]
    errs = RW.run_world(W, names, use_hip=True, port=free_port())
    assert not errs, "\n".join(errs)


@pytest.mark.parametrize("W", [2, 4])
def test_zigzag_varlen_gather_exchange_hip_matches_golden(W, monkeypatch):
    """RFA_ZIGZAG_VARLEN_EXCHANGE=gather (the mesh-aware form for packed sequences; default: the ring protocol) on the
    HIP kernels: io-dtype dK/dV slots with half-sequence selection, all-to-all, rfa_sum_slots"""
    import _ring_worker as RW
    import make_golden as MG
    from conftest import free_port

    monkeypatch.setenv("RFA_ZIGZAG_VARLEN_EXCHANGE", "gather")
    names = [n for n, c in MG.CASES.items() if c["W"] == W and c["kind"] == "zigzag_varlen"]
    assert names
    errs = RW.run_world(W, names, use_hip=True, port=free_port())
    assert not errs, "\n".join(errs)


@pytest.mark.parametrize("W", [2, 4])
def test_schedules_under_torch_compile_at_world_size_gt_1_on_the_hip_kernels(W, monkeypatch):
    """The reference runs every test a second time with the function under torch.compile at the full world size
    (/root/reference/test/test.sh:23-25, test/test_zigzag_ring_flash_attn_func.py:105-108).  GPU twin of
    tests/test_schedules_cpu.py::test_schedules_under_torch_compile_at_world_size_gt_1: W processes share the GPU, every
    public function is called through a torch.compile'd caller — default (inductor) backend like the reference's tests,
    `fullgraph=True`: traced tensor work on both sides of the call, the multi-rank schedule captured as ONE registered
    operator per direction (rfa::sched_fwd / sched_bwd), no graph break — on the HIP kernels, and must reproduce the
    golden vectors of the unmodified reference: both exchange forms of the zigzag path, and the other schedules."""
    import _ring_worker as RW
    import make_golden as MG
    from conftest import free_port

    monkeypatch.setenv("RFA_TEST_COMPILE", "1")
    names = [n for n, c in MG.CASES.items() if c["W"] == W]        # (incl. the head-dim-128 multi-tile cases: the LDS-DMA instances)
    assert names
    for mode in ("gather", "ring", "gather_ps"):
        monkeypatch.setenv("RFA_ZIGZAG_EXCHANGE", mode)
        sel = names if mode == "gather" else [n for n in names if MG.CASES[n]["kind"] == "zigzag"]
        errs = RW.run_world(W, sel, use_hip=True, port=free_port())
        assert not errs, "\n".join(errs)


@pytest.mark.parametrize("W", [2, 4])
def test_exchange_audit_passes_on_the_hip_path(W, monkeypatch):
    """config.exchange_check (RFA_EXCHANGE_CHECK=1, round 6) with DEVICE buffers: every K/V and dK/dV buffer a rank receives
    is checksummed on the GPU against its sender's checksum (utils.audit_verify; the ranks share this GPU, the transfers are
    host-staged) — all schedules and every zigzag exchange form must pass the audit and still reproduce the golden vectors."""
    import _ring_worker as RW
    import make_golden as MG
    from conftest import free_port

    monkeypatch.setenv("RFA_EXCHANGE_CHECK", "1")
    names = [n for n, c in MG.CASES.items() if c["W"] == W and "sample" not in c]
    assert names
    for mode in ("gather", "ring", "gather_ps"):
        monkeypatch.setenv("RFA_ZIGZAG_EXCHANGE", mode)
        sel = names if mode == "gather" else [n for n in names if MG.CASES[n]["kind"] == "zigzag"]
        errs = RW.run_world(W, sel, use_hip=True, port=free_port())
        assert not errs, "\n".join(errs)
