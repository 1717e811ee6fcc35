"""Worker of the dropout schedule tests: gloo ranks; backend = CPU oracle (not-gpu) or the HIP kernels with all ranks
on cuda:0 (gpu).  Every rank seeds torch alike, so all ranks — and the single-device comparison in the parent — draw
the same dropout seed."""
import os
import sys
import traceback

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

CU = [0, 100, 612, 1024]          # global cu_seqlens (3 packed sequences, 1024 tokens)
H, HK, D, P_DROP, SEED = 4, 2, 64, 0.2, 4242


def inputs(dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(9)
    T = CU[-1]
    return (torch.randn(T, H, D, generator=g).to(dtype), torch.randn(T, HK, D, generator=g).to(dtype),
            torch.randn(T, HK, D, generator=g).to(dtype), torch.randn(T, H, D, generator=g).to(dtype))


def run(rank, W, port, ret, use_hip, stride):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=W)
        import ring_flash_attn as R
        from ring_flash_attn import backend
        from ring_flash_attn import _testing

        if use_hip:
            dev = torch.device("cuda:0")
            _testing.set_backend(None)
            _testing.allow_host_staging(True)       # several gloo ranks share this one GPU
        else:
            from oracle.oracle_backend import OracleBackend

            dev = torch.device("cpu")
            _testing.set_backend(OracleBackend())
        q, k, v, do = inputs()
        T = CU[-1] // W
        sl = slice(rank * T, (rank + 1) * T)
        ql, kl, vl = (t[sl].to(dev).requires_grad_(True) for t in (q, k, v))
        cq, ck, mq, mk, ks = R.llama3_flash_attn_prepare_cu_seqlens(torch.tensor(CU, dtype=torch.int32), True, rank, W)
        torch.manual_seed(SEED)
        out = R.llama3_flash_attn_varlen_func(ql, kl, vl, cq.to(dev), ck.to(dev), mq, mk, stride, ks, dropout_p=P_DROP,
                                              causal=True)
        out.backward(do[sl].to(dev))
        res = dict(out=out.detach().cpu(), dq=ql.grad.cpu(), dk=kl.grad.cpu(), dv=vl.grad.cpu())
        # the ring schedules declare dropout unsupported over several ranks (as the reference does)
        raised = []
        for fn, args in ((R.ring_flash_attn_func, ()), (R.zigzag_ring_flash_attn_func, ()), (R.stripe_flash_attn_func, ())):
            try:
                fn(ql.view(1, T, H, D), kl.view(1, T, HK, D), vl.view(1, T, HK, D), dropout_p=0.1, causal=True)
                raised.append(False)
            except NotImplementedError:
                raised.append(True)
        res["raised"] = raised
        ret[rank] = res
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        ret[rank] = "EXC: " + traceback.format_exc()
