#!/usr/bin/env python3
"""<tag>_pmc_counters.txt (profiles/collect_pmc.sh) -> <tag>_traffic.json: HBM bytes per launch of every attention
kernel = FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB
x 1024 (every store of these kernels is 16 bytes wide), next to
  algorithmic_bytes  what the operation has to move (inputs once, outputs once) — the dS hand-off is NOT in it
  handoff_bytes      the dS blocks this implementation writes (dK/dV kernel) / reads back (dQ kernel): traffic the
                     5-GEMM design creates, reported separately so that the counter bytes can be read against both
and rfa_build_id() of the librfa_hip.so the counters were collected on (argument 3: read from that library by
collect_pmc.sh on the GPU box at collection time; bench.py only quotes `roofline.traffic` for the same build)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

S, H, HK, D = 8192, 32, 8, 128
QB, KB = S * H * D * 2, S * HK * D * 2            # bytes of one q-like / k-like bf16 tensor
LSE = S * H * 4
DS = H * (S // 32) * (S // 32 + 1) // 2 * 2048     # causal dS blocks (incl. the diagonal), 2 KiB each
ALGO = {
    "fwd_kernel": QB + 2 * KB + QB + LSE,                          # q, k, v -> out, lse
    "dq_kernel": 2 * QB + 2 * KB + 2 * LSE + QB,                   # dout, q, k, v, lse, delta -> dq
    "dkdv_kernel": 2 * QB + 2 * KB + 2 * LSE + 2 * KB,             # dout, q, k, v, lse, delta -> dk, dv
    "dkdv_kernel+spill": 2 * QB + 2 * KB + 2 * LSE + 2 * KB,       # the same operation
    "dq_ds_kernel": KB + QB,                                       # k -> dq (dS is the hand-off)
}
HANDOFF = {"dkdv_kernel+spill": DS, "dq_ds_kernel": DS}


def main():
    src, dst = sys.argv[1], sys.argv[2]
    vals = {}
    for line in open(src):
        m = re.match(r"\s*(\S+)\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)", line)
        if not m:
            continue
        name = m.group(1)
        for key in ("dq_ds_kernel", "dkdv_kernel", "dq_kernel", "fwd_kernel"):
            if key in name:
                spill = key == "dkdv_kernel" and "ELb1ELb1E" in name
                vals.setdefault(key + ("+spill" if spill else ""), {})[m.group(2)] = float(m.group(3))
                break
    out = {"_source": f"{os.path.basename(src)} (profiles/collect_pmc.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                      "over python bench.py, headline shape Hk=8, per launch); FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024",
           "library_build_id": sys.argv[3] if len(sys.argv) > 3 else None}
    for key, v in vals.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out[key] = {"fetch_kib": v["FETCH_SIZE"], "write_kib": v["WRITE_SIZE"],
                        "hbm_bytes_per_launch": int(v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024),
                        "algorithmic_bytes": ALGO.get(key), "handoff_bytes": HANDOFF.get(key, 0)}
    if "dkdv_kernel+spill" in out:
        out["dkdv_kernel"] = out["dkdv_kernel+spill"]       # the instance the product path launches
    json.dump(out, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main()
