"""Worker of the zigzag_llama3 tests: one rank of a gloo group (CPU oracle backend, or the HIP kernels with all ranks
on cuda:0) runs zigzag_llama3_flash_attn_varlen_func on its two slices of a seeded packed stream; the parent compares
with plain packed-sequence attention over the whole stream (oracle) sharded the same way."""
import os
import sys
import traceback

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

BF = torch.bfloat16


def make_inputs(c):
    g = torch.Generator().manual_seed(c["seed"])
    T = c["cu"][-1]
    q = torch.randn(T, c["H"], c["D"], generator=g).to(BF)
    k = torch.randn(T, c["Hk"], c["D"], generator=g).to(BF)
    v = torch.randn(T, c["Hk"], c["D"], generator=g).to(BF)
    do = torch.randn(T, c["H"], c["D"], generator=g).to(BF)
    return q, k, v, do


def shard(x, rank, W):
    ch = x.chunk(2 * W, dim=0)
    return torch.cat([ch[rank], ch[2 * W - 1 - rank]], dim=0).contiguous()


def run_rank(rank, W, port, c, use_hip, ret):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=W)
        import ring_flash_attn as R
        from ring_flash_attn import backend
        from ring_flash_attn import _testing

        if use_hip:
            dev = torch.device("cuda:0")
            torch.cuda.set_device(dev)
            _testing.set_backend(None)
            _testing.allow_host_staging(True)       # several gloo ranks share this one GPU
        else:
            from oracle.oracle_backend import OracleBackend

            dev = torch.device("cpu")
            _testing.set_backend(OracleBackend())
        q, k, v, do = [shard(t, rank, W).to(dev) for t in make_inputs(c)]
        cu = torch.tensor(c["cu"], dtype=torch.int32)
        kw = dict(causal=c["causal"], window_size=tuple(c.get("window", (-1, -1))), return_attn_probs=True)
        res = {}
        if c.get("packed"):
            kv = torch.stack([k, v], dim=1).requires_grad_(True)
            q.requires_grad_(True)
            out, lse, _ = R.zigzag_llama3_flash_attn_varlen_kvpacked_func(q, kv, cu, **kw)
            out.backward(do)
            res.update(dq=q.grad, dk=kv.grad[:, 0], dv=kv.grad[:, 1])
        else:
            q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
            out, lse, _ = R.zigzag_llama3_flash_attn_varlen_func(q, k, v, cu, **kw)
            out.backward(do)
            res.update(dq=q.grad, dk=k.grad, dv=v.grad)
        res.update(out=out.detach(), lse=lse.detach())
        ret[rank] = {n: t.float().cpu() for n, t in res.items()}
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        ret[rank] = "EXC: " + traceback.format_exc()


def run_world(W, c, use_hip, port):
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(run_rank, args=(W, port, c, use_hip, ret), nprocs=W, join=True)
    return [ret[r] for r in range(W)]
