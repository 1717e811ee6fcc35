"""torch.profiler table of ONE fwd+bwd iteration of a virtual ring rank (exchange looped back): which kernels run
beside the attention kernels (casts, zero fills, slot sums, autograd glue)?   usage: profile_virtual_step.py [world] [rank]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ring-flash-attention_amd"))
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("gloo", rank=0, world_size=1)
import ring_flash_attn as R
from ring_flash_attn import utils as U
from ring_flash_attn import _testing

W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
r = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
q = torch.randn(1, 8192, 32, 128, device=dev, dtype=torch.bfloat16, requires_grad=True)
kv = torch.randn(1, 8192, 2, 8, 128, device=dev, dtype=torch.bfloat16, requires_grad=True)
do = torch.randn_like(q)
_testing.set_loopback((r, W))
def step():
    q.grad = None; kv.grad = None
    out = R.zigzag_ring_flash_attn_kvpacked_func(q, kv, causal=True)
    out.backward(do)
for _ in range(3): step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
