#!/usr/bin/env python3
"""A/B runner for tuning variants of librfa_hip.so (built by tools/ab_variants.py into build/variants/<name>/).

    python tools/ab_run.py [--passes 2] [--what fwd,bwd,step] base v1 v2 ...

`base` is the in-tree library; every other name is build/variants/<name>/librfa_hip.so.  Each (variant, pass) is its own
process (the library is chosen at import: RFA_LIB_PATH); the passes walk the variants in order and again in order, so
that clock / thermal drift of the box shows up as a difference between the two rows of ONE variant instead of a
difference between variants.  Headline shape: q (1, 8192, 32, 128), 8 kv heads, bf16, causal.
  fwd   forward kernel looped on its own (hip events around 100 launches after 200 warm-up launches)
  bwd   the backward's kernels inside rfa_bwd (rfa_bwd_args.prof_events): dK/dV, dQ, reduce
  step  fwd + preprocess + bwd as one loop (what bench.py's step is made of), wall time per iteration
Worker mode (internal): python tools/ab_run.py --worker fwd|bwd|step"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(what):
    for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
        sys.path.insert(0, p)
    import torch

    import bench
    from ring_flash_attn.backend import get_backend

    be, hip, dev = get_backend(), bench._Hip(), torch.device("cuda:0")
    S, H, Hk, D = 8192, 32, 8, 128
    torch.manual_seed(0)
    q = torch.randn(1, S, H, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(1, S, Hk, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(1, S, Hk, D, device=dev, dtype=torch.bfloat16)
    do = torch.randn_like(q)
    out, lse = torch.empty_like(q), torch.empty(1, H, S, device=dev, dtype=torch.float32)
    delta = torch.empty_like(lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    sc = D ** -0.5
    f = 4.0 * H * S * S * D / 2

    def fwd():
        be.fwd(q, k, v, softmax_scale=sc, causal=True, out=out, lse=lse)

    def bwd(ev=None):
        be.bwd(do, q, k, v, lse, delta, softmax_scale=sc, causal=True, dq=dq, dk=dk, dv=dv, prof_events=ev)

    fwd()
    be.bwd_preprocess(do, out, delta)
    # checksums: variants of one kernel must agree bit for bit where the change is a re-ordering of exact operations
    if what == "fwd":
        for _ in range(200):
            fwd()
        e0, e1 = hip.event(), hip.event()
        hip.record(e0)
        for _ in range(100):
            fwd()
        hip.record(e1)
        t = hip.ms(e0, e1) / 100
        torch.cuda.synchronize()
        print(f"fwd {t:.4f} ms ({f / t / 1e9:.0f} TFLOP/s)  out_sum {out.float().sum().item():.6f} lse_sum {lse.sum().item():.6f}")
    elif what == "bwd":
        evs = []
        for it in range(230):
            ev = (hip.C.c_void_p * 4)(*[hip.event() for _ in range(4)]) if it >= 200 else None
            bwd(ev)
            if ev is not None:
                evs.append(ev)
        torch.cuda.synchronize()
        t = [sum(hip.ms(e[i], e[i + 1]) for e in evs) / len(evs) for i in range(3)]
        print(f"dkdv {t[0]:.4f} ms ({2 * f / t[0] / 1e9:.0f} TFLOP/s)  dq {t[1]:.4f}  reduce {t[2]:.4f}  total {sum(t):.4f} ms"
              f"  dq_sum {dq.float().sum().item():.4f} dk_sum {dk.float().sum().item():.4f} dv_sum {dv.float().sum().item():.4f}")
    else:
        def step():
            fwd()
            be.bwd_preprocess(do, out, delta)
            bwd()
        for _ in range(100):
            step()
        e0, e1 = hip.event(), hip.event()
        hip.record(e0)
        for _ in range(100):
            step()
        hip.record(e1)
        t = hip.ms(e0, e1) / 100
        print(f"step {t:.4f} ms ({1e3 / t:.1f} it/s, {3.5 * f / t / 1e9:.0f} TFLOP/s)")


def main():
    args = sys.argv[1:]
    if args[:1] == ["--worker"]:
        return worker(args[1])
    passes, what = 2, ["fwd", "bwd"]
    names = []
    i = 0
    while i < len(args):
        if args[i] == "--passes":
            passes = int(args[i + 1]); i += 2
        elif args[i] == "--what":
            what = args[i + 1].split(","); i += 2
        else:
            names.append(args[i]); i += 1
    for ps in range(passes):
        for n in names:
            env = dict(os.environ)
            if n != "base":
                env["RFA_LIB_PATH"] = os.path.join(ROOT, "build", "variants", n, "librfa_hip.so")
            for w in what:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", w], env=env,
                                   capture_output=True, text=True, timeout=300)
                line = (r.stdout.strip().splitlines() or ["(no output) " + r.stderr.strip()[-300:]])[-1]
                print(f"[pass {ps}] {n:>14s}  {line}", flush=True)


if __name__ == "__main__":
    main()
