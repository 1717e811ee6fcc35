#!/usr/bin/env python3
"""Build several variants of librfa_hip.so with different -D tuning macros into build/variants/<name>/
(in-tree so they travel with gpurun) and print the shell line that benchmarks them all with the
native self test:   python tools/ab_variants.py name1:-DFOO=1 name2:-DFOO=2,-DBAR=3 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ring-flash-attention_amd", "csrc")
SRCS = ["rfa_fwd.hip", "rfa_bwd.hip", "rfa_bigd.hip", "rfa_dqs.hip", "rfa_aux.hip"]


def main():
    names = []
    procs = []
    for spec in sys.argv[1:]:
        name, _, flags = spec.partition(":")
        flags = [f for f in flags.split(",") if f]
        out = os.path.join(ROOT, "build", "variants", name)
        os.makedirs(out, exist_ok=True)
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
               "-Wno-unused-result", "-Wno-inline-asm"] + flags + [os.path.join(CSRC, s) for s in SRCS] + \
              ["-x", "hip", os.path.join(CSRC, "rfa_api.cpp"), "-o", os.path.join(out, "librfa_hip.so")]
        procs.append((name, subprocess.Popen(cmd)))
        names.append(name)
    for name, pr in procs:
        if pr.wait() != 0:
            raise SystemExit(f"variant {name} failed to build")
    loop = " ".join(names)
    print(f"for v in {loop}; do echo \"== $v\"; LD_LIBRARY_PATH=build/variants/$v ./tests/native/selftest --perf-only | grep -E 'fwd|bwd'; done")


if __name__ == "__main__":
    main()
