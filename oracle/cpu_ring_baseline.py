"""oracle/cpu_ring_baseline.py — TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's `cpu_baseline` leg at N > 1).

"The reference's CPU path" at world size W: W CPU processes under gloo run ONE forward +
backward of `zigzag_ring_flash_attn_func` on host cores, with the attention arithmetic of the CPU oracle
(oracle/flash_attn_ref.py) underneath —

  kind "reference": the UNMODIFIED reference schedule /root/reference/ring_flash_attn/zigzag_ring_flash_attn.py:7-199
                    (oracle/reference_harness.py, oracle as its `flash_attn`) — only where /root/reference exists,
                    i.e. in the build container, never on the GPU box;
  kind "port":      this repository's schedule of the same function in its `ring` exchange form — the reference's
                    hop-by-hop protocol (utils.py:98-151) — with oracle/oracle_backend.py as its kernel backend.

Nothing here is imported by the product; bench.py starts it as W sub-processes:
    RANK=r WORLD_SIZE=W MASTER_ADDR=127.0.0.1 MASTER_PORT=p python oracle/cpu_ring_baseline.py S_rank Hk threads [fwd|fwdbwd]
Rank 0 prints one JSON line {"seconds": ..., "kind": ..., ...} for ONE timed iteration after a quarter-length
warm-up iteration.  Inputs: per rank q (1, S_rank, 32, 128), k / v (1, S_rank, Hk, 128), bf16 N(0,1), seed 42 + rank,
causal — the benchmark's per-rank tensors (benchmark/benchmark_kvpacked_func.py:20-53) at the stated S_rank."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def main():
    s_rank, hk, threads = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    fwd_only = len(sys.argv) > 4 and sys.argv[4] == "fwd"
    os.environ["OMP_NUM_THREADS"] = str(threads)
    import torch
    import torch.distributed as dist

    torch.set_num_threads(threads)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, D = 32, 128
    kind, fn = "port", None
    try:
        from oracle import reference_harness

        if reference_harness.available():
            mods = reference_harness.load_reference(provider="oracle")
            fn = mods["zigzag_ring_flash_attn"].zigzag_ring_flash_attn_func
            kind = "reference"
    except Exception:
        fn = None
    if fn is None:
        os.environ["RFA_ZIGZAG_EXCHANGE"] = "ring"
        import ring_flash_attn as R
        from oracle.oracle_backend import OracleBackend
        from ring_flash_attn import backend
        from ring_flash_attn import _testing

        _testing.set_backend(OracleBackend())
        fn = R.zigzag_ring_flash_attn_func
    g = torch.Generator().manual_seed(42 + rank)
    q = torch.randn(1, s_rank, H, D, generator=g).to(torch.bfloat16)
    k = torch.randn(1, s_rank, hk, D, generator=g).to(torch.bfloat16)
    v = torch.randn(1, s_rank, hk, D, generator=g).to(torch.bfloat16)
    do = torch.randn(1, s_rank, H, D, generator=g).to(torch.bfloat16)

    def once(n):
        if fwd_only:
            with torch.no_grad():
                fn(q[:, :n], k[:, :n], v[:, :n], causal=True)
            return
        qq, kk, vv = (t[:, :n].clone().requires_grad_(True) for t in (q, k, v))
        out = fn(qq, kk, vv, causal=True)
        out.backward(do[:, :n])

    once(max(64, s_rank // 4))          # thread pools, allocator, gloo connections
    dist.barrier()
    t0 = time.perf_counter()
    once(s_rank)
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"seconds": dt.item(), "kind": kind, "world": world, "s_rank": s_rank, "threads_per_rank": threads}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
