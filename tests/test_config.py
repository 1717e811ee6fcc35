"""ring_flash_attn.config: the package's switches are fields of ONE object, resolved from the environment once and
validated there (no os.environ lookups on the per-call paths, no bare KeyError /
ValueError out of a hot path for a mistyped variable)."""
import pytest


def test_defaults_and_environment(monkeypatch):
    from ring_flash_attn import config

    c = config.Config.from_env({})
    assert (c.zigzag_exchange, c.zigzag_varlen_exchange, c.dkv_wire_fp32, c.autotune, c.bwd_ds_spill) == ("auto", "ring", False, False, True)
    assert c.kv_keep and c.kv_keep_bytes == c.kv_keep_total_bytes == 4 << 30 and not hasattr(c, "force_steps")
    c = config.Config.from_env({"RFA_ZIGZAG_EXCHANGE": "Ring", "RFA_DKV_WIRE": "fp32", "RFA_ZIGZAG_KV_CACHE": "0",
                                "RFA_DKDV_NSPLIT": "3", "RFA_FWD_FORM": "4x32", "RFA_DS_SPILL_MAX_FRAC": "0.25",
                                "RFA_FWD_KV_NSPLIT": "1"})
    assert (c.zigzag_exchange, c.dkv_wire_fp32, c.kv_keep, c.dkdv_nsplit, c.fwd_form, c.ds_spill_max_frac, c.fwd_kv_nsplit) == \
        ("ring", True, False, 3, "4x32", 0.25, 1)
    monkeypatch.setenv("RFA_ZIGZAG_EXCHANGE", "gather")           # (tests/conftest.py re-resolves on RFA_* changes)
    assert config.get().zigzag_exchange == "gather"
    monkeypatch.delenv("RFA_ZIGZAG_EXCHANGE")
    assert config.get().zigzag_exchange == "auto"


@pytest.mark.parametrize("name,value", [("RFA_ZIGZAG_EXCHANGE", "mesh"), ("RFA_FWD_FORM", "16x16"), ("RFA_DKDV_NSPLIT", "two"),
                                        ("RFA_DKDV_NSPLIT", "-1"), ("RFA_DS_SPILL_MAX_FRAC", "1.5"), ("RFA_BWD_DS_SPILL", "maybe"),
                                        ("RFA_DKV_WIRE", "fp8"), ("RFA_GATHER_MAX_BYTES", "4G")])
def test_a_mistyped_switch_is_reported_by_name(name, value):
    from ring_flash_attn import config

    with pytest.raises(ValueError, match=name):
        config.Config.from_env({name: value})


def test_override_is_scoped_and_set_rejects_unknown_fields():
    from ring_flash_attn import config

    before = config.get().zigzag_exchange
    with config.override(zigzag_exchange="ring", dkv_wire_fp32=True) as c:
        assert c.zigzag_exchange == "ring" and config.get().dkv_wire_fp32
    assert config.get().zigzag_exchange == before and not config.get().dkv_wire_fp32
    with pytest.raises(AttributeError):
        config.set(no_such_switch=1)


def test_plan_overrides_follow_the_configuration():
    from ring_flash_attn import _C, backend, config

    with config.override(dkdv_wide=0, dkdv_nsplit=3):
        assert backend._plan_overrides() == (_C.DKDV_128, 3)
    with config.override(dkdv_wide=-1, dkdv_nsplit=2):
        assert backend._plan_overrides() == (_C.DKDV_256, 2)
    with config.override(dkdv_wide=-1, dkdv_nsplit=0):
        assert backend._plan_overrides() == (_C.DKDV_AUTO, 0)


def test_the_schedules_read_no_environment_variables():
    """the per-call paths (schedules, backend, comm helpers) contain no os.environ access: config.py and the library
    loader (_C.py: RFA_LIB_PATH, A/B tooling) are the only modules that look at the environment"""
    import os
    import re

    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ring-flash-attention_amd", "ring_flash_attn")
    offenders = []
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") and f not in ("config.py", "_C.py"):
                src = open(os.path.join(root, f)).read()
                if re.search(r"os\.environ|os\.getenv", src) and "FLASH_ATTENTION_DETERMINISTIC" not in src:
                    offenders.append(f)
    assert not offenders, offenders


def test_autotune_records_are_per_group():
    """a record measured on one process group must not decide for another group of the same size"""
    from ring_flash_attn import tuning

    class G:        # stand-ins: _group_key falls back to the object's identity when it is not a real group
        pass

    a, b = G(), G()
    ka = tuning._key(4, (1, 64, 8, 128), (1, 64, 2, 128), "bf16", a)
    kb = tuning._key(4, (1, 64, 8, 128), (1, 64, 2, 128), "bf16", b)
    assert ka != kb and ka == tuning._key(4, (1, 64, 8, 128), (1, 64, 2, 128), "bf16", a)
    tuning._TUNED[ka] = "ring"
    try:
        assert tuning.lookup((1, 64, 8, 128), (1, 64, 2, 128), "bf16", 4, a) == "ring"
        assert tuning.lookup((1, 64, 8, 128), (1, 64, 2, 128), "bf16", 4, b) is None
    finally:
        tuning._TUNED.pop(ka, None)


def test_llama3_min_groups_limits_the_head_fusion(monkeypatch):
    """config.llama3_min_groups (round 6): small models fuse all K/V heads into ONE super-group, which leaves nothing for the
    all-gather / reduce-scatter to overlap with; the knob keeps at least that many groups (results are unchanged: heads are
    independent — tests/test_schedules_cpu.py::test_llama3_fused_equals_unfused_four_groups)."""
    from ring_flash_attn import config
    from ring_flash_attn.llama3_flash_attn_varlen import fused_heads_k_stride as f

    assert f(8, 1, 2048, 8, 128, 2) == 8                      # Qwen3-0.6B at 2048 tokens per rank: everything fused
    with config.override(llama3_min_groups=2):
        assert f(8, 1, 2048, 8, 128, 2) == 4
    with config.override(llama3_min_groups=4):
        assert f(8, 1, 2048, 8, 128, 2) == 2 and f(8, 4, 2048, 8, 128, 2) == 4     # (never below heads_k_stride)
    monkeypatch.setenv("RFA_LLAMA3_MIN_GROUPS", "2")
    assert config.get().llama3_min_groups == 2
    import pytest

    with pytest.raises(ValueError, match="RFA_LLAMA3_MIN_GROUPS"):       # (tests/conftest.py re-resolves on every RFA_* change)
        monkeypatch.setenv("RFA_LLAMA3_MIN_GROUPS", "0")
    monkeypatch.delenv("RFA_LLAMA3_MIN_GROUPS")
