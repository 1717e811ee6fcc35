"""Multi-process worker for the BASELINE.json configuration tests: every rank (process) shares
cuda:0, runs the package's public function on its shard through the HIP kernels, and stores its
outputs for the parent, which compares them with the CPU oracle on the FULL (unsharded) problem —
exactly the structure of the reference's own tests (test/test_ring_flash_attn_func.py:30-100)."""
import os
import sys
import traceback

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def global_inputs(c):
    g = torch.Generator().manual_seed(c["seed"])
    if "cu" in c:
        T = c["cu"][-1]
        shp_q, shp_k = (T, c["H"], c["D"]), (T, c["Hk"], c["D"])
    else:
        shp_q, shp_k = (c["B"], c["S"], c["H"], c["D"]), (c["B"], c["S"], c["Hk"], c["D"])
    q = torch.randn(*shp_q, generator=g).to(torch.bfloat16)
    k = torch.randn(*shp_k, generator=g).to(torch.bfloat16)
    v = torch.randn(*shp_k, generator=g).to(torch.bfloat16)
    do = torch.randn(*shp_q, generator=g).to(torch.bfloat16)
    return q, k, v, do


def shard(c, rank, tensors):
    import make_golden as MG

    W, kind = c["W"], c["kind"]
    if kind == "zigzag":
        return [MG.zigzag_extract(t, rank, W, 1) for t in tensors]
    if kind == "ring":
        return [t.chunk(W, dim=1)[rank].contiguous() for t in tensors]
    return [MG.varlen_extract(t, c["cu"], rank, W, kind == "zigzag_varlen") for t in tensors]


def run_rank(rank, W, port, c, outdir):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=W)
        import ring_flash_attn as R
        from ring_flash_attn import backend
        from ring_flash_attn import _testing

        _testing.set_backend(None)

        _testing.allow_host_staging(True)       # several gloo ranks share this one GPU
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        q, k, v, do = [t.to(dev) for t in shard(c, rank, global_inputs(c))]
        q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
        kw = dict(return_attn_probs=True)
        kind = c["kind"]
        if kind == "zigzag":
            out, lse, _ = R.zigzag_ring_flash_attn_func(q, k, v, causal=True, **kw)
        elif kind == "ring":
            out, lse, _ = R.ring_flash_attn_func(q, k, v, causal=c["causal"], **kw)
        else:
            cu = torch.tensor(c["cu"], dtype=torch.int32)
            cu_local = (cu // W).to(dev)
            max_local = int((cu[1:] - cu[:-1]).max()) // W
            fn = R.zigzag_ring_flash_attn_varlen_func if kind == "zigzag_varlen" else R.ring_flash_attn_varlen_func
            out, lse, _ = fn(q, k, v, cu_local, max_local, causal=c.get("causal", True), **kw)
        out.backward(do)
        torch.cuda.synchronize()
        torch.save(dict(out=out.detach().cpu(), lse=lse.detach().cpu(), dq=q.grad.cpu(), dk=k.grad.cpu(), dv=v.grad.cpu()),
                   os.path.join(outdir, f"rank{rank}.pt"))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        open(os.path.join(outdir, f"rank{rank}.err"), "w").write(traceback.format_exc())


def run_world(c, outdir, port):
    import torch.multiprocessing as mp

    mp.spawn(run_rank, args=(c["W"], port, c, outdir), nprocs=c["W"], join=True)
    res = []
    for r in range(c["W"]):
        err = os.path.join(outdir, f"rank{r}.err")
        if os.path.exists(err):
            raise RuntimeError(open(err).read())
        res.append(torch.load(os.path.join(outdir, f"rank{r}.pt")))
    return res
