#!/usr/bin/env python3
"""Power and clock beside the headline kernels ("collect power beside GRBM_GUI_ACTIVE / wall: at
1.80 GHz effective the kernel is power-limited, so report whether a variant saved cycles, watts or neither").

    python tools/power_probe.py [--seconds 3] [--lib PATH]

Loops each phase for `--seconds` while a sampler thread reads the GPU's power (amdgpu hwmon `power1_average` /
`power1_input`, microwatts; `amd-smi` / `rocm-smi` as fall-backs) and shader clock (`freq1_input`, `pp_dpm_sclk`) every
20 ms, and prints ONE JSON object: per phase ms per iteration, TFLOP/s, average / peak watts, joules per iteration and the
sampled clock.  Phases (headline shape, q (1, 8192, 32, 128), 8 kv heads, bf16, causal):
    idle        nothing (the board's floor)
    fwd         the forward kernel alone
    dkdv        the dK/dV kernel alone (5-GEMM form: with the dS spill)
    dq          the dQ kernel alone (dq_ds_kernel: the HBM-bound reader of the hand-off)
    bwd         the whole backward
    step        forward + preprocess + backward (bench.py's step)
    step_7gemm  the same with the 7-GEMM backward (no hand-off: more MFMA work, 4.3 GB less HBM traffic per step)
    copy        a device-to-device copy of 2 GiB (the HBM-bound reference point)
joules per iteration = average watts x seconds per iteration: the quantity a power-limited chip actually bounds."""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    sys.path.insert(0, p)


def _hwmon_files():
    cands = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for name in ("power1_average", "power1_input"):
            f = os.path.join(card, name)
            if os.path.exists(f):
                cands.append(f)
                break
    return cands


def _freq_files():
    return sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))


def _read_int(path):
    try:
        return int(open(path).read().strip())
    except Exception:
        return None


def _smi_power():
    """fall-back: one reading through a tool (slow: ~100 ms); watts or None"""
    for cmd in (["amd-smi", "metric", "-p", "--json"], ["rocm-smi", "--showpower", "--json"]):
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            txt = json.dumps(d)
            import re

            m = re.search(r'"(?:socket_power|average_socket_power|current_socket_power|Average Graphics Package Power \(W\)|'
                          r'Current Socket Graphics Package Power \(W\))"\s*:\s*(?:\{"value"\s*:\s*)?"?([0-9.]+)', txt)
            if m:
                return float(m.group(1))
        except Exception:
            continue
    return None


class Sampler(threading.Thread):
    def __init__(self, period=0.02):
        super().__init__(daemon=True)
        self.period = period
        self.pfiles, self.ffiles = _hwmon_files(), _freq_files()
        self.use_smi = not self.pfiles
        self.samples = []          # (t, watts, mhz)
        self._stop = threading.Event()

    def run(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            if self.use_smi:
                w = _smi_power()
            else:
                vals = [_read_int(f) for f in self.pfiles]
                vals = [v for v in vals if v is not None]
                w = max(vals) / 1e6 if vals else None
            fr = [_read_int(f) for f in self.ffiles]
            fr = [v for v in fr if v]
            self.samples.append((t, w, max(fr) / 1e9 if fr else None))
            time.sleep(self.period if not self.use_smi else 0.0)

    def stop(self):
        self._stop.set()

    def window(self, t0, t1):
        ws = [w for t, w, _ in self.samples if t0 <= t <= t1 and w is not None]
        fs = [f for t, _, f in self.samples if t0 <= t <= t1 and f is not None]
        return ({"avg_w": sum(ws) / len(ws), "max_w": max(ws), "n": len(ws)} if ws else {"avg_w": None, "max_w": None, "n": 0},
                (sum(fs) / len(fs) if fs else None))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--phases", default="idle,fwd,dkdv,dq,bwd,step,step_7gemm,copy")
    ap.add_argument("--data", default="randn", choices=["randn", "zeros", "ternary"],
                    help="operand data (diagnostic: the chip's power, and with it the rate it sustains, depends on how many "
                         "bits toggle): randn = the benchmark's N(0,1); zeros; ternary = values from {-1, 0, 1}")
    args = ap.parse_args()
    import torch

    from ring_flash_attn import _C, config
    from ring_flash_attn.backend import get_backend

    be, dev = get_backend(), torch.device("cuda:0")
    S, H, Hk, D = 8192, 32, 8, 128
    torch.manual_seed(0)
    def mk(*shape):
        if args.data == "zeros":
            return torch.zeros(*shape, device=dev, dtype=torch.bfloat16)
        if args.data == "ternary":
            return torch.randint(-1, 2, shape, device=dev).to(torch.bfloat16)
        return torch.randn(*shape, device=dev, dtype=torch.bfloat16)

    q, k, v, do = mk(1, S, H, D), mk(1, S, Hk, D), mk(1, S, Hk, D), mk(1, S, H, D)
    out, lse = torch.empty_like(q), torch.empty(1, H, S, device=dev, dtype=torch.float32)
    delta = torch.empty_like(lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    sc = D ** -0.5
    f = 4.0 * H * S * S * D / 2
    big_a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    big_b = torch.empty(1 << 30, dtype=torch.uint8, device=dev)

    def fwd():
        be.fwd(q, k, v, softmax_scale=sc, causal=True, out=out, lse=lse)

    def bwd(phases=0):
        be.bwd(do, q, k, v, lse, delta, softmax_scale=sc, causal=True, dq=dq, dk=dk, dv=dv, phases=phases)

    def step():
        fwd()
        be.bwd_preprocess(do, out, delta)
        bwd()

    fwd()
    be.bwd_preprocess(do, out, delta)
    bwd()
    torch.cuda.synchronize()
    phases = {
        "idle": (None, 0.0),
        "fwd": (fwd, f),
        "dkdv": (lambda: bwd(_C.BWD_COMPUTE | _C.BWD_SKIP_DQ), 2.0 * f),
        "dq": (lambda: bwd(_C.BWD_COMPUTE | _C.BWD_SKIP_DKDV), 0.5 * f),
        "bwd": (bwd, 2.5 * f),
        "step": (step, 3.5 * f),
        "step_7gemm": (step, 3.5 * f),
        "copy": (lambda: big_b.copy_(big_a), 0.0),
    }
    smp = Sampler()
    smp.start()
    res = {"power_source": smp.pfiles or "amd-smi / rocm-smi", "freq_source": smp.ffiles, "device": torch.cuda.get_device_name(dev),
           "data": args.data}
    for name in args.phases.split(","):
        fn, flops = phases[name]
        if name == "step_7gemm":
            config.set(bwd_ds_spill=False)
        try:
            if fn is None:
                t0 = time.perf_counter()
                time.sleep(min(args.seconds, 2.0))
                t1 = time.perf_counter()
                n = 0
            else:
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                # size the loop from a short measurement, then run it as ONE queue (no host sync inside)
                t0 = time.perf_counter()
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                per = (time.perf_counter() - t0) / 20
                n = max(20, int(args.seconds / per))
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
        finally:
            if name == "step_7gemm":
                config.set(bwd_ds_spill=True)
        # (skip the first 15 % of the window: the power reading is an average over some tens of ms)
        pw, mhz = smp.window(t0 + 0.15 * (t1 - t0), t1)
        ms = (t1 - t0) / n * 1e3 if n else None
        res[name] = {"ms_per_iter": ms, "tflops": (flops / (ms * 1e-3) / 1e12) if (ms and flops) else None,
                     "avg_w": pw["avg_w"], "max_w": pw["max_w"], "samples": pw["n"],
                     "joules_per_iter": (pw["avg_w"] * ms * 1e-3) if (ms and pw["avg_w"]) else None,
                     "sampled_sclk_ghz": mhz}
        time.sleep(0.5)
    smp.stop()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
