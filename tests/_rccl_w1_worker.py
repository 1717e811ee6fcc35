"""Worker of tests/test_gpu_rccl_world1.py: ONE process, ONE GPU, process group backend "nccl" (= RCCL) with
world_size 1, ring_flash_attn._testing.force_steps() — every schedule then runs its multi-step code path (exchange buffers,
RCCL all_gather / all_to_all / reduce_scatter / batched isend+irecv to itself, the side stream, fp32
accumulators, final casts) instead of collapsing to one kernel.  At world size 1 the result must equal plain
attention over the local sequence, which the CPU oracle provides.  This is the only place a one-GPU box can
execute the RCCL calls of the product path (gloo, used by the multi-rank parity tests, stages through the host)."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ring-flash-attention_amd"))

import torch
import torch.distributed as dist

BF = torch.bfloat16


def check(name, got, ref, atol, rtol=0.0):
    """all criteria of tests/_tol.py (multi-step path: the *_ring bounds); kind from the call site's historical pair"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _tol

    kind = {(2e-2, 0.0): "out_ring", (1e-3, 0.0): "lse_ring", (1e-2, 2e-2): "grad_ring"}[(atol, rtol)]
    _tol.compare(name, got, ref, kind)


def main(port):
    from oracle import flash_attn_ref as O
    import ring_flash_attn as R
    from ring_flash_attn import _testing

    _testing.force_steps(True)

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    g = torch.Generator().manual_seed(5)
    H, Hk, D, S = 4, 2, 128, 1024
    scale = D ** -0.5

    def dense_case(fn, name, causal=True, B=2):
        q = torch.randn(B, S, H, D, generator=g).to(BF)
        k = torch.randn(B, S, Hk, D, generator=g).to(BF)
        v = torch.randn(B, S, Hk, D, generator=g).to(BF)
        do = torch.randn(B, S, H, D, generator=g).to(BF)
        ro, rl, _, _ = O._flash_attn_forward(q, k, v, 0.0, scale, causal)
        rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        O._flash_attn_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, 0.0, scale, causal)
        qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
        out, lse, _ = fn(qd, kd, vd, causal=causal, return_attn_probs=True)
        out.backward(do.to(dev))
        torch.cuda.synchronize()
        check(name + ".out", out, ro, 2e-2)
        check(name + ".lse", lse, rl, 1e-3)
        for n, a, b in (("dq", qd.grad, rdq), ("dk", kd.grad, rdk), ("dv", vd.grad, rdv)):
            check(f"{name}.{n}", a, b, 1e-2, 2e-2)
        print("ok", name, flush=True)

    from ring_flash_attn import config

    for mode in ("gather", "ring", "gather_ps"):
        for wire in ("io", "fp32"):
            with config.override(zigzag_exchange=mode, dkv_wire_fp32=wire == "fp32"):
                dense_case(R.zigzag_ring_flash_attn_func, f"zigzag[{mode},{wire}]")
    # the exchange audit on the RCCL group (device-resident checksum table, all-gather on the group, host read-back)
    with config.override(zigzag_exchange="ring", exchange_check=True):
        dense_case(R.zigzag_ring_flash_attn_func, "zigzag[ring, exchange_check]")
    with config.override(exchange_check=True):
        dense_case(R.ring_flash_attn_func, "ring.causal[exchange_check]")
    dense_case(R.ring_flash_attn_func, "ring.causal")
    dense_case(R.ring_flash_attn_func, "ring.full", causal=False)
    dense_case(R.stripe_flash_attn_func, "stripe")

    cu = torch.tensor([0, 128, 640, 1024], dtype=torch.int32)
    T = int(cu[-1])
    maxlen = int((cu[1:] - cu[:-1]).max())

    def varlen_case(fn, name, llama3=False):
        q = torch.randn(T, H, D, generator=g).to(BF)
        k = torch.randn(T, Hk, D, generator=g).to(BF)
        v = torch.randn(T, Hk, D, generator=g).to(BF)
        do = torch.randn(T, H, D, generator=g).to(BF)
        ro, rl, _, _ = O._flash_attn_varlen_forward(q, k, v, cu, cu, maxlen, maxlen, 0.0, scale, True)
        rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        O._flash_attn_varlen_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, cu, cu, maxlen, maxlen, 0.0, scale, True)
        qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
        if llama3:
            cq, ck, mq, mk, sl = R.llama3_flash_attn_prepare_cu_seqlens(cu, causal=True, rank=0, world_size=1)
            out, lse, _ = fn(qd, kd, vd, cq.to(dev), ck.to(dev), mq, mk, heads_k_stride=1, local_k_slice=sl,
                             causal=True, return_attn_probs=True)
        else:
            out, lse, _ = fn(qd, kd, vd, cu.to(dev), maxlen, causal=True, return_attn_probs=True)
        out.backward(do.to(dev))
        torch.cuda.synchronize()
        check(name + ".out", out, ro, 2e-2)
        check(name + ".lse", lse, rl, 1e-3)
        for n, a, b in (("dq", qd.grad, rdq), ("dk", kd.grad, rdk), ("dv", vd.grad, rdv)):
            check(f"{name}.{n}", a, b, 1e-2, 2e-2)
        print("ok", name, flush=True)

    varlen_case(R.zigzag_ring_flash_attn_varlen_func, "zigzag_varlen")
    varlen_case(R.ring_flash_attn_varlen_func, "ring_varlen")
    varlen_case(R.llama3_flash_attn_varlen_func, "llama3", llama3=True)
    config.set(llama3_gather_max_bytes=0)                    # one K/V head group per collective
    varlen_case(R.llama3_flash_attn_varlen_func, "llama3[unfused]", llama3=True)
    config.set(llama3_gather_max_bytes=1 << 30)
    # llama3 with a PACKED kv (round 5): one all-gather of the packed tensor, one reduce-scatter into the packed gradient
    q = torch.randn(T, H, D, generator=g).to(BF)
    kv = torch.randn(T, 2, Hk, D, generator=g).to(BF)
    do = torch.randn(T, H, D, generator=g).to(BF)
    ro, rl, _, _ = O._flash_attn_varlen_forward(q, kv[:, 0], kv[:, 1], cu, cu, maxlen, maxlen, 0.0, scale, True)
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(kv[:, 0]), torch.empty_like(kv[:, 1])
    O._flash_attn_varlen_backward(do, q, kv[:, 0], kv[:, 1], ro, rl, rdq, rdk, rdv, cu, cu, maxlen, maxlen, 0.0, scale, True)
    qd, kvd = q.to(dev).requires_grad_(True), kv.to(dev).requires_grad_(True)
    cq, ck, mq, mk, sl = R.llama3_flash_attn_prepare_cu_seqlens(cu, causal=True, rank=0, world_size=1)
    out, lse, _ = R.llama3_flash_attn_varlen_kvpacked_func(qd, kvd, cq.to(dev), ck.to(dev), mq, mk, heads_k_stride=1,
                                                           local_k_slice=sl, causal=True, return_attn_probs=True)
    out.backward(do.to(dev))
    torch.cuda.synchronize()
    check("llama3[packed kv].out", out, ro, 2e-2)
    check("llama3[packed kv].lse", lse, rl, 1e-3)
    for n, a, b in (("dq", qd.grad, rdq), ("dk", kvd.grad[:, 0], rdk), ("dv", kvd.grad[:, 1], rdv)):
        check(f"llama3[packed kv].{n}", a, b, 1e-2, 2e-2)
    print("ok llama3[packed kv]", flush=True)
    # zigzag_llama3 (all-gather + re-order to stream order + fp32 reduce-scatter)
    q = torch.randn(T, H, D, generator=g).to(BF)
    k = torch.randn(T, Hk, D, generator=g).to(BF)
    v = torch.randn(T, Hk, D, generator=g).to(BF)
    do = torch.randn(T, H, D, generator=g).to(BF)
    ro, rl, _, _ = O._flash_attn_varlen_forward(q, k, v, cu, cu, maxlen, maxlen, 0.0, scale, True)
    rdq, rdk, rdv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    O._flash_attn_varlen_backward(do, q, k, v, ro, rl, rdq, rdk, rdv, cu, cu, maxlen, maxlen, 0.0, scale, True)
    qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    out, lse, _ = R.zigzag_llama3_flash_attn_varlen_func(qd, kd, vd, cu, causal=True, return_attn_probs=True)
    out.backward(do.to(dev))
    torch.cuda.synchronize()
    check("zigzag_llama3.out", out, ro, 2e-2)
    check("zigzag_llama3.lse", lse, rl, 1e-3)
    for n, a, b in (("dq", qd.grad, rdq), ("dk", kd.grad, rdk), ("dv", vd.grad, rdv)):
        check(f"zigzag_llama3.{n}", a, b, 1e-2, 2e-2)
    print("ok zigzag_llama3", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("ALL OK", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]))
