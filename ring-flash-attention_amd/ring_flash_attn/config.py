"""ring_flash_attn.config — the ONE place the package's switches live.

Every tuning / policy switch is a field of `Config`, resolved ONCE from the process environment (first use, or an
explicit `reload()`), validated there with a descriptive error, and read by the schedules as a plain attribute —
no `os.environ` lookup on any per-call path.  Three ways to set them:

    environment (before the first call, or followed by `config.reload()`)       RFA_ZIGZAG_EXCHANGE=ring python train.py
    code, process-wide                                                           config.set(zigzag_exchange="ring")
    code, scoped (tests, the exchange autotuner)                                 with config.override(zigzag_exchange="ring"): ...

| field                     | environment variable            | default | meaning |
|---------------------------|---------------------------------|---------|---------|
| zigzag_exchange           | RFA_ZIGZAG_EXCHANGE             | auto    | dense zigzag exchange form: auto / gather / gather_ps (per-source arrival) / ring (zigzag_ring_flash_attn.py) |
| zigzag_varlen_exchange    | RFA_ZIGZAG_VARLEN_EXCHANGE      | ring    | packed zigzag exchange form: ring / gather |
| dkv_wire_fp32             | RFA_DKV_WIRE                    | io      | gather form: dK/dV contributions travel in the io dtype (io) or fp32 |
| gather_max_bytes          | RFA_GATHER_MAX_BYTES            | 4 GiB   | auto without a measurement: gather while its O(S_total) scratch stays below |
| autotune                  | RFA_AUTOTUNE                    | 0       | auto: 1 = the first multi-rank call per (group, shapes) MEASURES both forms itself (tuning.py); default: only explicit tuning.autotune_zigzag_exchange calls (bench.py's warm-up) measure |
| kv_keep                   | RFA_ZIGZAG_KV_KEEP              | 1       | gather form: keep the gathered K/V of a forward for its backward |
| kv_keep_bytes             | RFA_ZIGZAG_KV_KEEP_BYTES        | 4 GiB   | ... unless one call's gathered K/V exceed this |
| kv_keep_total_bytes       | RFA_ZIGZAG_KV_KEEP_TOTAL_BYTES  | 4 GiB   | ... or all live kept buffers of the process together would (L layers x W x (K,V)) |
| llama3_gather_max_bytes   | RFA_LLAMA3_GATHER_MAX_BYTES     | 1 GiB   | llama3: head groups fused per collective while the gathered K/V stay below |
| llama3_min_groups         | RFA_LLAMA3_MIN_GROUPS           | 1       | llama3: never fuse the head groups into fewer than this many super-groups.  With ONE super-group (what small models get: the gathered K/V of Qwen3-0.6B's 8 K/V heads at 16384 tokens are 67 MB) nothing overlaps: all-gather, then attention.  With g super-groups the gather of group i+1 and the reduce-scatter of group i-1 run beside the kernels of group i — at most 1/g of the exchange is exposed — for about 7 % more kernel time at g = 2 (two half-size launches).  Whether that pays is a property of the node's xGMI all-gather rate; unmeasured (no multi-GPU box): 2 is the setting to try first on one |
| bwd_ds_spill              | RFA_BWD_DS_SPILL                | 1       | 5-GEMM backward (dS hand-off) where eligible; 0: always the 7-GEMM form |
| ds_spill_max_bytes        | RFA_DS_SPILL_MAX_BYTES          | 4.5 GiB | size of the ONE reusable dS scratch per device and stream; larger hand-offs run in head-group chunks |
| ds_spill_max_frac         | RFA_DS_SPILL_MAX_FRAC           | 0.5     | ... and never more than this fraction of the memory free when it is first taken |
| fwd_form                  | RFA_FWD_FORM                    | auto    | forward kernel form (tuning / tests): auto / 8x32 (256 rows) / 4x32 (128 rows) / p8x32 (persistent 256 rows) |
| dkdv_wide, dkdv_nsplit    | RFA_DKDV_WIDE, RFA_DKDV_NSPLIT  | unset   | dK/dV launch plan overrides (tuning / tests) |
| fwd_kv_nsplit             | RFA_FWD_KV_NSPLIT               | 0       | split-KV forward launches: 0 chosen from the shapes, 1 off, 2..8 forced (tuning / tests) |
| tuning_log                | RFA_TUNING_LOG                  | 0       | print autotune decisions on rank 0 |
| exchange_check            | RFA_EXCHANGE_CHECK              | 0       | DEBUG: checksum every K/V and dK/dV buffer a rank receives against its sender's checksum (one extra tiny all-gather and a host read-back per schedule call; a mismatch raises naming rank / step / buffer: utils.audit_verify) |
"""
import contextlib
import dataclasses
import os

_GiB = 1 << 30


def _bool(name, raw):
    v = raw.strip().lower()
    if v in ("1", "true", "on", "yes"):
        return True
    if v in ("0", "false", "off", "no"):
        return False
    raise ValueError(f"{name} must be 0 or 1, got {raw!r}")


def _int(name, raw, lo=0):
    try:
        v = int(raw.strip())
    except ValueError:
        raise ValueError(f"{name} must be an integer, got {raw!r}") from None
    if v < lo:
        raise ValueError(f"{name} must be >= {lo}, got {v}")
    return v


def _float(name, raw, lo, hi):
    try:
        v = float(raw.strip())
    except ValueError:
        raise ValueError(f"{name} must be a number, got {raw!r}") from None
    if not lo <= v <= hi:
        raise ValueError(f"{name} must be in [{lo}, {hi}], got {v}")
    return v


def _choice(name, raw, choices):
    v = raw.strip().lower()
    if v not in choices:
        raise ValueError(f"{name} must be one of {', '.join(choices)}; got {raw!r}")
    return v


@dataclasses.dataclass
class Config:
    zigzag_exchange: str = "auto"
    zigzag_varlen_exchange: str = "ring"
    dkv_wire_fp32: bool = False
    gather_max_bytes: int = 4 * _GiB
    autotune: bool = False
    kv_keep: bool = True
    kv_keep_bytes: int = 4 * _GiB
    kv_keep_total_bytes: int = 4 * _GiB
    llama3_gather_max_bytes: int = 1 * _GiB
    llama3_min_groups: int = 1
    bwd_ds_spill: bool = True
    ds_spill_max_bytes: int = 9 * _GiB // 2
    ds_spill_max_frac: float = 0.5
    fwd_form: str = "auto"
    dkdv_wide: int = -1          # -1 unset, 0 the 128-key form, 1 the 256-key form, 2 the balanced causal schedule (where eligible)
    dkdv_nsplit: int = 0         # 0 unset
    fwd_kv_nsplit: int = 0       # 0: chosen from the shapes; 1: never split; 2..8 forced
    tuning_log: bool = False
    exchange_check: bool = False

    @staticmethod
    def from_env(env=None) -> "Config":
        env = os.environ if env is None else env
        c = Config()

        def get(name):
            raw = env.get(name)
            return raw if raw is not None and raw.strip() != "" else None

        if (r := get("RFA_ZIGZAG_EXCHANGE")) is not None:
            c.zigzag_exchange = _choice("RFA_ZIGZAG_EXCHANGE", r, ("auto", "gather", "gather_ps", "ring"))
        if (r := get("RFA_ZIGZAG_VARLEN_EXCHANGE")) is not None:
            c.zigzag_varlen_exchange = _choice("RFA_ZIGZAG_VARLEN_EXCHANGE", r, ("ring", "gather"))
        if (r := get("RFA_DKV_WIRE")) is not None:
            c.dkv_wire_fp32 = _choice("RFA_DKV_WIRE", r, ("io", "bf16", "fp16", "fp32")) == "fp32"
        if (r := get("RFA_GATHER_MAX_BYTES")) is not None:
            c.gather_max_bytes = _int("RFA_GATHER_MAX_BYTES", r)
        if (r := get("RFA_AUTOTUNE")) is not None:
            c.autotune = _bool("RFA_AUTOTUNE", r)
        r = get("RFA_ZIGZAG_KV_KEEP")
        if r is None:
            r = get("RFA_ZIGZAG_KV_CACHE")            # (the switch's round-2 name)
        if r is not None:
            c.kv_keep = _bool("RFA_ZIGZAG_KV_KEEP", r)
        if (r := get("RFA_ZIGZAG_KV_KEEP_BYTES")) is not None:
            c.kv_keep_bytes = _int("RFA_ZIGZAG_KV_KEEP_BYTES", r)
        if (r := get("RFA_ZIGZAG_KV_KEEP_TOTAL_BYTES")) is not None:
            c.kv_keep_total_bytes = _int("RFA_ZIGZAG_KV_KEEP_TOTAL_BYTES", r)
        if (r := get("RFA_LLAMA3_GATHER_MAX_BYTES")) is not None:
            c.llama3_gather_max_bytes = _int("RFA_LLAMA3_GATHER_MAX_BYTES", r)
        if (r := get("RFA_BWD_DS_SPILL")) is not None:
            c.bwd_ds_spill = _bool("RFA_BWD_DS_SPILL", r)
        if (r := get("RFA_DS_SPILL_MAX_BYTES")) is not None:
            c.ds_spill_max_bytes = _int("RFA_DS_SPILL_MAX_BYTES", r)
        if (r := get("RFA_DS_SPILL_MAX_FRAC")) is not None:
            c.ds_spill_max_frac = _float("RFA_DS_SPILL_MAX_FRAC", r, 0.0, 1.0)
        if (r := get("RFA_FWD_FORM")) is not None:
            c.fwd_form = _choice("RFA_FWD_FORM", r, ("auto", "8x32", "4x32", "p8x32"))
        if (r := get("RFA_DKDV_WIDE")) is not None:
            c.dkdv_wide = 2 if r.strip() == "2" else (1 if _bool("RFA_DKDV_WIDE", r) else 0)
        if (r := get("RFA_DKDV_NSPLIT")) is not None:
            c.dkdv_nsplit = _int("RFA_DKDV_NSPLIT", r)
        if (r := get("RFA_FWD_KV_NSPLIT")) is not None:
            c.fwd_kv_nsplit = _int("RFA_FWD_KV_NSPLIT", r)
        if (r := get("RFA_LLAMA3_MIN_GROUPS")) is not None:
            c.llama3_min_groups = _int("RFA_LLAMA3_MIN_GROUPS", r, lo=1)
        if (r := get("RFA_TUNING_LOG")) is not None:
            c.tuning_log = _bool("RFA_TUNING_LOG", r)
        if (r := get("RFA_EXCHANGE_CHECK")) is not None:
            c.exchange_check = _bool("RFA_EXCHANGE_CHECK", r)
        return c


_cfg = None


def get() -> Config:
    """the process-wide configuration (resolved from the environment on first use)"""
    global _cfg
    if _cfg is None:
        _cfg = Config.from_env()
    return _cfg


def reload() -> Config:
    """re-resolve from the environment (after changing RFA_* variables in a running process)"""
    global _cfg
    _cfg = Config.from_env()
    return _cfg


def set(**fields) -> Config:
    """change fields process-wide; unknown names raise"""
    c = get()
    for k, v in fields.items():
        if not hasattr(c, k):
            raise AttributeError(f"ring_flash_attn.config: no field {k!r}")
        setattr(c, k, v)
    return c


@contextlib.contextmanager
def override(**fields):
    """fields changed for the duration of a `with` block (same thread of control: the configuration is process-wide)"""
    c = get()
    saved = {k: getattr(c, k) for k in fields}
    set(**fields)
    try:
        yield c
    finally:
        for k, v in saved.items():
            setattr(c, k, v)


# ---- budget of gathered K/V kept for backwards (Config.kv_keep_total_bytes) -------------------------------------
class _KeptBudget:
    """Counts the bytes of gathered K/V that forwards have handed to their backwards and that are still alive.  The
    reference saves only the local k / v per layer; the gather form of the zigzag schedule additionally keeps W x (K, V)
    per pending backward (0.27 GB at W = 8, Hk = 8, S = 8192 per rank) — without a bound an L-layer model without
    activation checkpointing would hold L of them.  A forward reserves before keeping; the reservation is released
    when its backward has run or its graph is freed, whichever comes first.  A forward that cannot reserve keeps
    nothing: its backward gathers again."""

    def __init__(self):
        self.live = 0

    class Token:
        def __init__(self, budget, nbytes):
            self._b, self.nbytes = budget, nbytes

        def release(self):
            if self.nbytes:
                self._b.live -= self.nbytes
                self.nbytes = 0

        __del__ = release

    def may_keep(self, nbytes) -> bool:
        """the part of the decision that depends on the configuration and the call's size only — the same on every rank
        of a group (nothing to agree on when it refuses); the live-bytes budget below is the rank-local part"""
        c = get()
        return bool(c.kv_keep) and nbytes <= c.kv_keep_bytes

    def try_reserve(self, nbytes):
        c = get()
        if not self.may_keep(nbytes) or self.live + nbytes > c.kv_keep_total_bytes:
            return None
        self.live += nbytes
        return _KeptBudget.Token(self, nbytes)


kept_budget = _KeptBudget()
