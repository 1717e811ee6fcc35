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


@pytest.mark.parametrize("W", [2, 4])
def test_schedules_match_reference_golden(W):
    names = [n for n, c in MG.CASES.items() if c["W"] == W]
    assert names
    errs = RW.run_world(W, names, use_hip=False, port=free_port())
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
