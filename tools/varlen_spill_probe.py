"""per-pattern timing of the packed-sequence backward with / without the dS spill (bench.py's zigzag_varlen patterns)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ring-flash-attention_amd"))
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
dist.init_process_group("gloo", rank=0, world_size=1)
import ring_flash_attn as R
import bench
dev = torch.device("cuda:0")
q = torch.randn(8192, 32, 128, device=dev, dtype=torch.bfloat16, requires_grad=True)
kv = torch.randn(8192, 2, 8, 128, device=dev, dtype=torch.bfloat16, requires_grad=True)
do = torch.randn_like(q)
pats = bench.VARLEN_PATTERNS if len(sys.argv) < 2 else [bench.VARLEN_PATTERNS[int(sys.argv[1])]]
for cu in pats:
    cut = torch.tensor(cu, device=dev, dtype=torch.int32); mx = max(b - a for a, b in zip(cu[:-1], cu[1:]))
    for sp in (("1", "0") if len(sys.argv) < 3 else (sys.argv[2],)):
        os.environ["RFA_BWD_DS_SPILL"] = sp
        __import__("ring_flash_attn").config.reload()
        def step():
            q.grad = None; kv.grad = None
            R.zigzag_ring_flash_attn_varlen_kvpacked_func(q, kv, cut, mx, causal=True).backward(do)
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize()
        print(cu, "spill", sp, f"{(time.perf_counter() - t0) * 100:.3f} ms/iter", flush=True)
