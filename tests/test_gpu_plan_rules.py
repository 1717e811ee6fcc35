"""The launch-plan rules of csrc/rfa_api.cpp, guarded around their own boundaries.

The forward form (256- / 128-row workgroups, split-KV shares) and the dK/dV plan (128- / 256-key workgroups, shares of the
query range) are chosen from constants tuned on 256-CU boxes at one clock / power state.  tools/plan_sweep.py walks shapes on
both sides of every boundary (forward: 96 .. 224 workgroups of 256 rows against 96 .. 160 key tiles, and sequences of 768 ..
1280 rows; backward: the same sequences at 1 .. 8 K/V heads — the shapes of
/root/reference/benchmark/benchmark_varlen_kvpacked_func.py:20-60 scaled to one GPU) and times the CHOSEN form against every
form that can be forced.  Bar: chosen <= 1.05 x the best forced form + 3 us (launch-to-launch noise on 0.1 ms kernels) —
a neighbouring shape must not fall off a cliff."""
import os
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.extended]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_chosen_plan_is_within_5_percent_of_the_best_forced_plan(single_rank_group):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import plan_sweep

    lines = []
    rows = plan_sweep.sweep(quick=True, log=lines.append)
    print("\n".join(lines))
    bad = [f"{d} {label}: chosen {c:.4f} ms vs {name} {b:.4f} ms ({c / b:.3f})" for d, label, c, b, name in rows
           if c > 1.05 * b + 0.003]
    assert not bad, "\n".join(bad)
