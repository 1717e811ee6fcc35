#!/usr/bin/env python3
"""Does the HBM-bound dQ kernel (dq_ds_kernel: streams the dS hand-off back once) hide behind the MFMA-bound dK/dV kernel
of ANOTHER head group when the two run on different HIP streams?

Headline backward (q (1, 8192, 32, 128), 8 kv heads, bf16, causal) cut into C chunks of kv heads; chunk c runs as one
rfa_bwd call (dK/dV -> dQ -> reduce, in stream order) on stream c % 2, and its dK/dV launch is held back until the
dK/dV launch of chunk c - 1 has finished (hipStreamWaitEvent on the event rfa_bwd records behind its dK/dV launch:
rfa_bwd_args.prof_events[1]) — so that dQ(c - 1) and dK/dV(c) are in flight together instead of the two chains running in
lock-step.  Wall time per backward (hip events on the main stream around 50 iterations after 50), against the ONE-call
backward on one stream.  usage: python tools/bwd_pipeline_probe.py [chunks ...]   (default 2 4 8)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ring-flash-attention_amd")):
    sys.path.insert(0, p)
import torch

import bench
from ring_flash_attn.backend import get_backend


def main():
    chunk_counts = [int(x) for x in sys.argv[1:]] or [2, 4, 8]
    be, hip, dev = get_backend(), bench._Hip(), torch.device("cuda:0")
    C = hip.C
    hip.lib.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    S, H, Hk, D = 8192, 32, 8, 128
    G = H // Hk
    torch.manual_seed(0)
    q = torch.randn(1, S, H, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(1, S, Hk, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(1, S, Hk, D, device=dev, dtype=torch.bfloat16)
    do = torch.randn_like(q)
    out, lse = torch.empty_like(q), torch.empty(1, H, S, device=dev, dtype=torch.float32)
    sc = D ** -0.5
    be.fwd(q, k, v, softmax_scale=sc, causal=True, out=out, lse=lse)
    delta = torch.empty_like(lse)
    be.bwd_preprocess(do, out, delta)
    dq0, dk0, dv0 = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    main_s = torch.cuda.current_stream()

    def timed(fn, n_warm=50, n=50):
        for _ in range(n_warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = hip.event(), hip.event()
        hip.record(e0)
        for _ in range(n):
            fn()
        hip.record(e1)
        t = hip.ms(e0, e1) / n
        torch.cuda.synchronize()
        return t

    def whole():
        be.bwd(do, q, k, v, lse, delta, softmax_scale=sc, causal=True, dq=dq0, dk=dk0, dv=dv0)

    t_ref = timed(whole)
    print(f"one call, one stream           : {t_ref:.4f} ms")

    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    for nch in chunk_counts:
        hc = Hk // nch
        if hc < 1:
            continue
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        # events: per chunk the 4 prof events (only [1] = "dK/dV launch done" is used), plus fork / join events
        evs = [(C.c_void_p * 4)(None, hip.event(), None, None) for _ in range(nch)]
        fork, joins = hip.event(), [hip.event(), hip.event()]

        def run(skew=True):
            hip.lib.hipEventRecord(fork, C.c_void_p(main_s.cuda_stream))
            for s in side:
                hip.lib.hipStreamWaitEvent(C.c_void_p(s.cuda_stream), fork, 0)
            for c in range(nch):
                s = side[c % 2]
                hs, ks = slice(c * hc * G, (c + 1) * hc * G), slice(c * hc, (c + 1) * hc)
                with torch.cuda.stream(s):
                    if skew and c > 0:
                        hip.lib.hipStreamWaitEvent(C.c_void_p(s.cuda_stream), evs[c - 1][1], 0)
                    be.bwd(do[:, :, hs], q[:, :, hs], k[:, :, ks], v[:, :, ks], lse[:, hs], delta[:, hs], softmax_scale=sc,
                           causal=True, dq=dq[:, :, hs], dk=dk[:, :, ks], dv=dv[:, :, ks], prof_events=evs[c])
            for i, s in enumerate(side):
                hip.lib.hipEventRecord(joins[i], C.c_void_p(s.cuda_stream))
                hip.lib.hipStreamWaitEvent(C.c_void_p(main_s.cuda_stream), joins[i], 0)

        for skew in (True, False):
            t = timed(lambda: run(skew))
            ok = (torch.equal(dq, dq0), (dk.float() - dk0.float()).abs().max().item(), (dv.float() - dv0.float()).abs().max().item())
            print(f"{nch} chunks of {hc} kv heads, 2 streams, {'skewed ' if skew else 'lockstep'}: {t:.4f} ms ({t / t_ref:.3f} x)"
                  f"   dq identical {ok[0]}, max|ddk| {ok[1]:.2e}, max|ddv| {ok[2]:.2e}")

        # the same chunks one after the other on ONE stream (what the chunking alone costs)
        def serial():
            for c in range(nch):
                hs, ks = slice(c * hc * G, (c + 1) * hc * G), slice(c * hc, (c + 1) * hc)
                be.bwd(do[:, :, hs], q[:, :, hs], k[:, :, ks], v[:, :, ks], lse[:, hs], delta[:, hs], softmax_scale=sc,
                       causal=True, dq=dq[:, :, hs], dk=dk[:, :, ks], dv=dv[:, :, ks])
        t = timed(serial)
        print(f"{nch} chunks of {hc} kv heads, 1 stream             : {t:.4f} ms ({t / t_ref:.3f} x)")


if __name__ == "__main__":
    main()
