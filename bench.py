#!/usr/bin/env python3
"""bench.py — headline benchmark: iters/sec of `zigzag_ring_flash_attn_kvpacked_func` forward+backward.

Workload (BASELINE.json metric; mirrors /root/reference/benchmark/benchmark_kvpacked_func.py:13-126):
per rank q = randn(1, 8192, 32, 128), kv = randn(1, 8192, 2, Hk, 128), dout like q, bf16, N(0,1),
seed 42+rank, causal; the data IS the local zigzag shard, total sequence = 8192 * world_size.
A "step" = one forward + one backward of that operator on every rank (grad reset each step).

    python bench.py --gpus 1 --steps K --warmup W                      # single GPU
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `value` = iterations/s of the whole job (max time over ranks).
`roofline` = the dominant kernel (dK/dV backward) against the bf16 MFMA peak, timed with device
events on the launch stream; `cpu_baseline` = the CPU oracle ("port") on a bounded sample of the
same workload on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC; RCCL / cross-process device memory need this (see task env)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

SEQ, HEADS, HEAD_DIM = 8192, 32, 128
MFMA_PEAK_TFLOPS = 2500.0          # dense bf16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"


def fwd_flops(world):
    """flash-attention convention, causal = half: 4*B*H*S_tot^2*D/2, per GPU (/world)."""
    s_tot = SEQ * world
    return 4.0 * 1 * HEADS * s_tot * s_tot * HEAD_DIM / 2.0 / world


def time_kernel(fn, iters=10, warm=2):
    """average device time of fn() in ms — events on the stream the kernels are launched on."""
    for _ in range(warm):
        fn()
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_breakdown(q, kv, dout, hk):
    """per-kernel device time at N=1 through the backend (same launches the operator makes)."""
    from ring_flash_attn import _C
    from ring_flash_attn.backend import get_backend

    be = get_backend()
    k, v = kv[:, :, 0], kv[:, :, 1]
    B, S, H, D = q.shape
    scale = D ** -0.5
    out = torch.empty_like(q)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
    delta = torch.empty_like(lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    dkc, dvc = torch.empty(k.shape, dtype=k.dtype, device=q.device), torch.empty(v.shape, dtype=v.dtype, device=q.device)
    t = {}
    t["fwd"] = time_kernel(lambda: be.fwd(q, k, v, softmax_scale=scale, causal=True, out=out, lse=lse))
    t["bwd_preprocess"] = time_kernel(lambda: be.bwd_preprocess(dout, out, delta))
    common = dict(softmax_scale=scale, causal=True, dq=dq, dk=dkc, dv=dvc)
    t["bwd_dq"] = time_kernel(lambda: be.bwd(dout, q, k, v, lse, delta, phases=_C.BWD_COMPUTE | _C.BWD_SKIP_DKDV, **common))
    t["bwd_dkdv"] = time_kernel(lambda: be.bwd(dout, q, k, v, lse, delta, phases=_C.BWD_COMPUTE | _C.BWD_SKIP_DQ, **common))
    t["bwd_reduce"] = time_kernel(lambda: be.bwd(dout, q, k, v, lse, delta, phases=_C.BWD_REDUCE, **common))
    return t


def cpu_baseline(hk):
    """The CPU oracle (a port: oracle/flash_attn_ref.py, the restated flash_attn arithmetic the
    reference would run per block) on a bounded sample: ONE kv-head group (H/Hk q heads) at the full
    S=8192 causal shape, fwd+bwd once, scaled by the number of groups.  World size 1: the zigzag
    schedule degenerates to a single causal block, so no merge/communication is involved."""
    from oracle import flash_attn_ref as O

    g = HEADS // hk
    gen = torch.Generator().manual_seed(42)
    q = torch.randn(1, SEQ, g, HEAD_DIM, generator=gen).to(torch.bfloat16)
    k = torch.randn(1, SEQ, 1, HEAD_DIM, generator=gen).to(torch.bfloat16)
    v = torch.randn(1, SEQ, 1, HEAD_DIM, generator=gen).to(torch.bfloat16)
    do = torch.randn(1, SEQ, g, HEAD_DIM, generator=gen).to(torch.bfloat16)
    scale = HEAD_DIM ** -0.5
    t0 = time.perf_counter()
    out, lse, _, _ = O._flash_attn_forward(q, k, v, 0.0, scale, True)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_backward(do, q, k, v, out, lse, dq, dk, dv, 0.0, scale, True)
    dt = time.perf_counter() - t0
    full = dt * hk
    return {
        "value": 1.0 / full,
        "unit": "iters/sec",
        "cores": torch.get_num_threads(),
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"1 of {hk} kv-head groups ({g} q heads), full S={SEQ} causal fwd+bwd once "
                  f"({dt:.2f} s), scaled x{hk}",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--kv-heads", type=int, default=8, help="8 = the reference benchmark's GQA; 32 = MHA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the operator has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if "MASTER_ADDR" not in os.environ:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ.setdefault("MASTER_PORT", "29541")
    # one process per GPU over RCCL ("nccl" on ROCm); a single process has nothing to exchange
    # (libgloo / librccl print connection banners on fd 1: keep stdout for the ONE JSON line)
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        if world > 1:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=0, world_size=1)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from ring_flash_attn import zigzag_ring_flash_attn_kvpacked_func as fn

    hk = args.kv_heads
    torch.manual_seed(42 + rank)
    q = torch.randn(1, SEQ, HEADS, HEAD_DIM, device=dev, dtype=torch.bfloat16, requires_grad=True)
    kv = torch.randn(1, SEQ, 2, hk, HEAD_DIM, device=dev, dtype=torch.bfloat16, requires_grad=True)
    dout = torch.randn(1, SEQ, HEADS, HEAD_DIM, device=dev, dtype=torch.bfloat16)

    def step():
        q.grad = None
        kv.grad = None
        out = fn(q, kv, causal=True, window_size=(-1, -1), alibi_slopes=None, deterministic=False,
                 return_attn_probs=False)
        out.backward(dout)

    # device spin-up (not a measurement knob): the MI355X needs some tens of milliseconds of load to leave
    # its idle clocks; without it the W warm-up steps (W x ~2 ms) end while the clocks are still ramping and
    # the timed region measures the ramp, not the kernels.  Untimed, bounded, reported in the JSON line.
    spin_s = float(os.environ.get("RFA_BENCH_SPINUP_S", "0.3"))
    t_spin = time.perf_counter()
    while spin_s > 0 and time.perf_counter() - t_spin < spin_s:
        step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if world > 1 else "cpu")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = tmax.item()

    ms = elapsed / args.steps * 1e3
    its = args.steps / elapsed
    per_gpu_flops = 3.5 * fwd_flops(world)
    result = {
        "metric": "iters/sec fwd+bwd zigzag_ring, seq=8192*ws, h=32, d=128 bf16",
        "value": its,
        "unit": "iters/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "spinup_s": spin_s,
        "ms_per_step": ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"zigzag_ring_flash_attn_kvpacked_func fwd+bwd, per-rank q=(1,{SEQ},{HEADS},{HEAD_DIM}) "
                        f"kv=(1,{SEQ},2,{hk},{HEAD_DIM}) bf16 causal, total seq {SEQ * world}",
            "kv_heads": hk,
            "world_size": world,
        },
        "algorithmic_tflops_per_gpu": per_gpu_flops * its / 1e12,
        "mfma_roofline_frac_end_to_end": per_gpu_flops * its / 1e12 / MFMA_PEAK_TFLOPS,
    }

    if rank == 0 and world == 1 and not args.no_breakdown:
        with torch.no_grad():
            t = kernel_breakdown(q.detach(), kv.detach(), dout, hk)
        f = fwd_flops(1)
        # algorithmic GEMM work per launch (fwd = 4BHS^2D/2, bwd = 2.5 fwd, of which the
        # dK/dV kernel owns 4 of the 5 backward GEMMs and the dQ kernel the fifth; recomputation of
        # S and dP inside the dQ kernel is NOT credited)
        algo = {"fwd": f, "bwd_dkdv": 2.0 * f, "bwd_dq": 0.5 * f}
        dom = max(algo, key=lambda n: t[n])
        ach = algo[dom] / (t[dom] * 1e-3) / 1e12
        result["roofline"] = {
            "kernel": {"fwd": "fwd_kernel", "bwd_dkdv": "dkdv_kernel", "bwd_dq": "dq_kernel"}[dom],
            "bound": "mfma",
            "achieved": ach,
            "peak": MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": ach / MFMA_PEAK_TFLOPS,
            "traffic": None,
            "avg_launch_ms": t[dom],
        }
        # HBM bytes per launch come from the committed PMC pass (profiles/r01_traffic.json), not from
        # this run: counters cannot be collected inside the timed process
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            kn = result["roofline"]["kernel"]
            if hk == 8 and kn in tr:
                result["roofline"]["traffic"] = tr[kn]["hbm_bytes_per_launch"]
                result["roofline"]["traffic_source"] = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, separate passes)"
        except Exception:
            pass
        result["kernels_ms"] = {k2: round(v2, 4) for k2, v2 in t.items()}
        result["kernels_tflops"] = {n: algo[n] / (t[n] * 1e-3) / 1e12 for n in algo}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(hk)

    if rank == 0:
        print(json.dumps(result), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
