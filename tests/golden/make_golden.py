#!/usr/bin/env python3
"""Generate tests/golden/ring_golden.pt by running the UNMODIFIED reference
(/root/reference/ring_flash_attn/*.py, loaded by oracle/reference_harness.py with the CPU oracle as
its `flash_attn`) under gloo, one process per rank, on seeded inputs.

    python tests/golden/make_golden.py           # only works where /root/reference exists

Each case stores the GLOBAL seeded inputs once and, per rank, the reference's outputs
(out, lse, dq, dk, dv) on that rank's shard — the golden vectors the parity tests compare against
(tests re-shard the inputs with the `shard()` function of this file).  Sharding follows the
reference tests (test/test_zigzag_ring_flash_attn_func.py:9-14, test_ring_flash_attn_func.py:36,
test_zigzag_ring_flash_attn_varlen_func.py:9-20, test_ring_flash_attn_varlen_func.py:9-15,
test_llama3_flash_attn_varlen_func.py:42-43, test_stripe_flash_attn_func.py:9-14).  Also stores golden outputs of
llama3_flash_attn_prepare_cu_seqlens (test/test_llama3_prepare_cu_seqlens.py fixture + extras).
"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "ring_golden.pt")

DT = torch.bfloat16

CASES = {
    "zigzag_dense_w4_gqa": dict(kind="zigzag", W=4, B=1, S=128, H=4, Hk=2, D=32, seed=11),
    "zigzag_dense_w2_mha_d128": dict(kind="zigzag", W=2, B=1, S=32, H=2, Hk=2, D=128, seed=12),
    "zigzag_dense_w8_gqa": dict(kind="zigzag", W=8, B=1, S=32, H=4, Hk=2, D=32, seed=23),   # the driver's largest N
    "ring_dense_w4_causal": dict(kind="ring", W=4, B=2, S=64, H=2, Hk=2, D=32, causal=True, seed=13),
    "ring_dense_w2_noncausal_gqa": dict(kind="ring", W=2, B=1, S=64, H=4, Hk=1, D=64, causal=False, seed=14),
    "zigzag_varlen_w2": dict(kind="zigzag_varlen", W=2, cu=[0, 16, 80, 128], H=2, Hk=2, D=32, seed=15),
    "zigzag_varlen_w4_gqa": dict(kind="zigzag_varlen", W=4, cu=[0, 32, 128], H=4, Hk=2, D=32, seed=16),
    "ring_varlen_w2_causal": dict(kind="ring_varlen", W=2, cu=[0, 20, 84, 128], H=2, Hk=2, D=32, causal=True, seed=17),
    "ring_varlen_w4_noncausal": dict(kind="ring_varlen", W=4, cu=[0, 32, 128], H=4, Hk=2, D=32, causal=False, seed=18),
    "stripe_w4": dict(kind="stripe", W=4, B=1, S=128, H=2, Hk=2, D=32, seed=21),
    "stripe_w2_gqa": dict(kind="stripe", W=2, B=2, S=48, H=4, Hk=2, D=32, seed=22),
    "llama3_w4": dict(kind="llama3", W=4, cu=[0, 22, 75, 128], H=4, Hk=2, D=32, stride=1, seed=19),
    "llama3_w2_stride2": dict(kind="llama3", W=2, cu=[0, 60, 62, 128], H=4, Hk=2, D=32, stride=2, seed=20),
}


def zigzag_extract(x, rank, W, dim):
    ch = x.chunk(2 * W, dim=dim)
    return torch.cat([ch[rank], ch[2 * W - 1 - rank]], dim=dim).contiguous()


def stripe_extract(x, rank, W, dim=1):
    """token t -> rank t mod W   (reference test/test_stripe_flash_attn_func.py:9-14)"""
    x = torch.stack(x.split(W, dim=dim), dim=dim).transpose(dim, dim + 1)
    slicer = [rank if i == dim else slice(None) for i in range(len(x.shape))]
    return x[slicer].contiguous()


def varlen_extract(x, cu, rank, W, zigzag):
    parts = []
    for i in range(len(cu) - 1):
        seq = x[cu[i]:cu[i + 1]]
        if zigzag:
            ch = seq.chunk(2 * W, dim=0)
            parts += [ch[rank], ch[2 * W - 1 - rank]]
        else:
            parts.append(seq.chunk(W, dim=0)[rank])
    return torch.cat(parts, dim=0).contiguous()


def make_inputs(c):
    g = torch.Generator().manual_seed(c["seed"])
    if "cu" in c:
        T = c["cu"][-1]
        q = torch.randn(T, c["H"], c["D"], generator=g).to(DT)
        k = torch.randn(T, c["Hk"], c["D"], generator=g).to(DT)
        v = torch.randn(T, c["Hk"], c["D"], generator=g).to(DT)
        do = torch.randn(T, c["H"], c["D"], generator=g).to(DT)
    else:
        q = torch.randn(c["B"], c["S"], c["H"], c["D"], generator=g).to(DT)
        k = torch.randn(c["B"], c["S"], c["Hk"], c["D"], generator=g).to(DT)
        v = torch.randn(c["B"], c["S"], c["Hk"], c["D"], generator=g).to(DT)
        do = torch.randn(c["B"], c["S"], c["H"], c["D"], generator=g).to(DT)
    return q, k, v, do


def shard(c, rank):
    """global tensors -> this rank's local q,k,v,dout (+ local cu / llama3 params)."""
    q, k, v, do = make_inputs(c)
    W, kind = c["W"], c["kind"]
    extra = {}
    if kind == "zigzag":
        loc = [zigzag_extract(t, rank, W, 1) for t in (q, k, v, do)]
    elif kind == "ring":
        loc = [t.chunk(W, dim=1)[rank].contiguous() for t in (q, k, v, do)]
    elif kind == "stripe":
        loc = [stripe_extract(t, rank, W) for t in (q, k, v, do)]
    elif kind in ("zigzag_varlen", "ring_varlen"):
        zz = kind == "zigzag_varlen"
        loc = [varlen_extract(t, c["cu"], rank, W, zz) for t in (q, k, v, do)]
        cu = torch.tensor(c["cu"], dtype=torch.int32)
        extra["cu_local"] = (cu // W).to(torch.int32)
        extra["max_local"] = int((cu[1:] - cu[:-1]).max().item()) // W
    elif kind == "llama3":
        L = c["cu"][-1] // W
        loc = [t[rank * L:(rank + 1) * L].contiguous() for t in (q, k, v, do)]
    return loc, extra


def run_case(name, c, rank, ref):
    (q, k, v, do), extra = shard(c, rank)
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    kind = c["kind"]
    kw = dict(dropout_p=0.0, window_size=(-1, -1), alibi_slopes=None, deterministic=False, return_attn_probs=True)
    if kind == "zigzag":
        out, lse, _ = ref["zigzag_ring_flash_attn"].zigzag_ring_flash_attn_func(q, k, v, causal=True, **kw)
    elif kind == "ring":
        out, lse, _ = ref["ring_flash_attn"].ring_flash_attn_func(q, k, v, causal=c["causal"], **kw)
    elif kind == "stripe":
        out, lse, _ = ref["stripe_flash_attn"].stripe_flash_attn_func(q, k, v, causal=True, **kw)
    elif kind == "zigzag_varlen":
        out, lse, _ = ref["zigzag_ring_flash_attn_varlen"].zigzag_ring_flash_attn_varlen_func(
            q, k, v, extra["cu_local"], extra["max_local"], causal=True, **kw)
    elif kind == "ring_varlen":
        out, lse, _ = ref["ring_flash_attn_varlen"].ring_flash_attn_varlen_func(
            q, k, v, extra["cu_local"], extra["max_local"], causal=c["causal"], **kw)
    elif kind == "llama3":
        m = ref["llama3_flash_attn_varlen"]
        cu = torch.tensor(c["cu"], dtype=torch.int32)
        cq, ck, mq, mk, sl = m.llama3_flash_attn_prepare_cu_seqlens(cu, True, rank, c["W"])
        out, lse, _ = m.llama3_flash_attn_varlen_func(q, k, v, cq, ck, mq, mk, heads_k_stride=c["stride"],
                                                      local_k_slice=sl, causal=True, **kw)
        extra.update(cu_q=cq, cu_k=ck, max_q=mq, max_k=mk, k_slice=(sl.start, sl.stop))
    out.backward(do)
    rec = dict(out=out.detach(), lse=lse.detach().contiguous().float(), dq=q.grad, dk=k.grad, dv=v.grad)
    rec.update(extra)
    return rec


def worker(rank, W, port, names, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from oracle.reference_harness import load_reference
    ref = load_reference()
    res = {}
    for n in names:
        res[n] = run_case(n, CASES[n], rank, ref)
    ret[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def prepare_cu_golden(ref):
    m = ref["llama3_flash_attn_varlen"]
    fixtures = [([0, 7, 14, 16], 8), ([0, 120, 1248, 4232], 8), ([0, 90, 300, 512], 4), ([0, 256], 4),
                ([0, 64, 128, 192, 256], 4), ([0, 3000, 9000, 16384], 8)]
    out = []
    for cu, W in fixtures:
        for causal in (True, False):
            for r in range(W):
                cq, ck, mq, mk, sl = m.llama3_flash_attn_prepare_cu_seqlens(torch.tensor(cu, dtype=torch.int32), causal, r, W)
                out.append(dict(cu=cu, W=W, causal=causal, rank=r, cu_q=cq.tolist(), cu_k=ck.tolist(), max_q=int(mq),
                                max_k=int(mk), k_slice=(int(sl.start), int(sl.stop))))
    return out


def main():
    golden = {"cases": {}, "meta": {n: dict(c) for n, c in CASES.items()}}
    by_w = {}
    for n, c in CASES.items():
        by_w.setdefault(c["W"], []).append(n)
    port = 29611
    for W, names in sorted(by_w.items()):
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(worker, args=(W, port, names, ret), nprocs=W, join=True)
        port += 1
        for n in names:
            gq, gk, gv, gdo = make_inputs(CASES[n])
            golden["cases"][n] = dict(inputs=dict(q=gq, k=gk, v=gv, dout=gdo), ranks=[ret[r][n] for r in range(W)])
    from oracle.reference_harness import load_reference
    golden["prepare_cu_seqlens"] = prepare_cu_golden(load_reference())
    golden["generator"] = "tests/golden/make_golden.py; reference @ /root/reference (2025-09-05); torch " + torch.__version__
    torch.save(golden, OUT)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
