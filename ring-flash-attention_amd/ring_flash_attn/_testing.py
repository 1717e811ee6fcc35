"""ring_flash_attn._testing — every test / measurement hook of the package, in the ONE module production never imports.

Nothing under ring_flash_attn/ imports this module; tests/, bench.py, tools/, oracle/cpu_ring_baseline.py and
__graft_entry__.smoke() do.  Importing it installs nothing; calling one of the setters below creates the `Hooks` record
and hands it to the product modules, which otherwise carry a single `None` test on the paths concerned
(`utils._TEST is None`, `backend._backend`):

    set_backend(obj)          an object with HipBackend's interface instead of the HIP library — the CPU oracle
                              (oracle/oracle_backend.py) under the schedules, or bench.py's in-step timer around the real
                              backend.  None restores the HIP backend.
    set_loopback((rank, W))   the exchange helpers move data between LOCAL buffers: ONE process executes the exact kernel
                              sequence of rank `rank` of a W-rank job with no communication (bench.py `comm.exposed_ms`,
                              `--virtual-world`, tools/small_launch.py).  Results are meaningless.  None switches it off.
    force_steps(flag)         a one-rank group keeps the multi-step path of every schedule — exchange buffers,
                              collectives, side stream, fp32 accumulators — which is how the RCCL calls of the product
                              path get exercised on a one-GPU box (tests/test_gpu_rccl_world1.py, RFA_BENCH_FORCE_RCCL).
    corrupt_receive(n)        with config.exchange_check on: the n-th audited receive of this process (0-based, counted over
                              ring hops and per-source exchanges) is overwritten after it landed — what a wrongly recycled
                              receive buffer looks like; the audit must name it (tests/test_schedules_cpu.py).
    allow_host_staging(flag)  device tensors on a gloo group travel through host memory (several test ranks sharing one
                              MI355X: tests/_ring_worker.py).  Without it such a call RAISES: the product transport is
                              RCCL, and gloo cannot move device memory.
"""
from . import backend as _backend_mod
from . import utils as _utils


class Hooks:
    def __init__(self):
        self.loopback = None          # (rank, world) or None
        self.force_steps = False
        self.host_staging = False
        self.corrupt_recv = None      # countdown to the audited receive that gets corrupted, or None

    def corrupt(self, buf):
        if self.corrupt_recv == 0:
            flat = buf.view(-1)
            flat[flat.numel() // 2] += 1
            self.corrupt_recv = None
        elif self.corrupt_recv is not None:
            self.corrupt_recv -= 1


def _hooks() -> Hooks:
    if _utils._TEST is None:
        _utils._TEST = Hooks()
    return _utils._TEST


_saved_hip = None


def set_backend(obj):
    global _saved_hip
    cur = _backend_mod._backend
    if isinstance(cur, _backend_mod.HipBackend):
        _saved_hip = cur              # (keeps its scratch pools for when the HIP backend is restored)
    _backend_mod._backend = obj if obj is not None else _saved_hip


def set_loopback(rank_world=None):
    _hooks().loopback = rank_world


def force_steps(flag=True):
    _hooks().force_steps = bool(flag)


def allow_host_staging(flag=True):
    _hooks().host_staging = bool(flag)


def corrupt_receive(n=0):
    _hooks().corrupt_recv = None if n is None else int(n)


def reset():
    """production state: no hooks, the HIP backend"""
    _utils._TEST = None
    _utils._BACKEND_OF.clear()
    set_backend(None)
