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
