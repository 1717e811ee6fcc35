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


KEEP = ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
        "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "EFFECTIVE_CLOCK_GHZ")


def parse(src):
    """{section ("hk8" | "hk32"): {kernel: {counter: per-dispatch average}}} — a line `== pass: <name>` opens a pass; pass
    names ending in _hk32 belong to the MHA (--kv-heads 32) runs"""
    vals, sec = {}, "hk8"
    for line in open(src):
        m = re.match(r"== pass: (\S+)", line)
        if m:
            sec = "hk32" if m.group(1).endswith("_hk32") else "hk8"
            continue
        m = re.match(r"\s*(\S+)\s+(" + "|".join(KEEP) + r")\s+([0-9.]+)", line)
        if not m:
            continue
        name = m.group(1)
        for key in ("dq_ds_kernel", "dkdv_kernel", "dq_kernel", "fwd_kernel"):
            if key in name:
                spill = key == "dkdv_kernel" and "ELb1ELb1E" in name
                vals.setdefault(sec, {}).setdefault(key + ("+spill" if spill else ""), {})[m.group(2)] = float(m.group(3))
                break
    return vals


def entries(vals, hk):
    kb = S * hk * D * 2
    algo = {
        "fwd_kernel": QB + 2 * kb + QB + LSE,
        "dq_kernel": 2 * QB + 2 * kb + 2 * LSE + QB,
        "dkdv_kernel": 2 * QB + 2 * kb + 2 * LSE + 2 * kb,
        "dkdv_kernel+spill": 2 * QB + 2 * kb + 2 * LSE + 2 * kb,
        "dq_ds_kernel": kb + QB,
    }
    out = {}
    for key, v in vals.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            e = {"fetch_kib": v["FETCH_SIZE"], "write_kib": v["WRITE_SIZE"],
                 "hbm_bytes_per_launch": int(v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024),
                 "algorithmic_bytes": algo.get(key), "handoff_bytes": HANDOFF.get(key, 0)}
            e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / e["algorithmic_bytes"] if e["algorithmic_bytes"] else None
            # issue-side counters of the same build (per dispatch, averaged over the counter's instances; SQ_* in
            # quad-cycles) and the clock the profiled pass ran at: a cycle saving must show here, not only in wall time
            for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
                      "GRBM_GUI_ACTIVE"):
                if c in v:
                    e[c.lower()] = v[c]
            if "EFFECTIVE_CLOCK_GHZ" in v:
                e["effective_clock_ghz_profiled"] = v["EFFECTIVE_CLOCK_GHZ"]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_WAVE_CYCLES" in v:
                e["mfma_pipe_busy"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (2.0 * v["SQ_WAVE_CYCLES"])
            out[key] = e
    if "dkdv_kernel+spill" in out:
        out["dkdv_kernel"] = out["dkdv_kernel+spill"]       # the instance the product path launches
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    parsed = parse(src)
    vals = parsed.get("hk8", {})
    out = {"_source": f"{os.path.basename(src)} (profiles/collect_pmc.sh: separate rocprofv3 --pmc passes over python bench.py, "
                      "headline shape, per launch); hbm bytes = FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024; the top-level "
                      "entries are the Hk = 8 (GQA) runs, `hk32` the --kv-heads 32 (MHA) runs",
           "library_build_id": sys.argv[3] if len(sys.argv) > 3 else None}
    out.update(entries(vals, HK))
    if "hk32" in parsed:
        out["hk32"] = entries(parsed["hk32"], 32)
    json.dump(out, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main()
