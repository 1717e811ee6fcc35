#!/usr/bin/env python3
"""Build recipe for the MI355X ring flash-attention library (gfx950 only).

Targets (all in-tree, so the artefacts travel with the repo snapshot to the GPU box):
  lib      ring-flash-attention_amd/ring_flash_attn/librfa_hip.so   hipcc, HIP kernels + C ABI
  oracle   oracle/libattn_ref.so                                     gcc -fopenmp, CPU checker
  selftest tests/native/selftest                                     hipcc, torch-free GPU test

`python ring-flash-attention_amd/build.py [lib] [oracle] [selftest] [--force]`
hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container.
"""
import hashlib
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ring-flash-attention_amd")
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "ring_flash_attn", "librfa_hip.so")
ORACLE_SRC = os.path.join(ROOT, "oracle", "attn_ref.c")
ORACLE_LIB = os.path.join(ROOT, "oracle", "libattn_ref.so")
SELFTEST_SRC = os.path.join(ROOT, "tests", "native", "selftest.cpp")
SELFTEST_BIN = os.path.join(ROOT, "tests", "native", "selftest")

HIP_SOURCES = ["rfa_fwd.hip", "rfa_bwd.hip", "rfa_bigd.hip", "rfa_dqs.hip", "rfa_aux.hip"]
API_SOURCE = "rfa_api.cpp"
HEADERS = ["rfa_common.hpp", "rfa_kernels.hpp", os.path.join(ROOT, "include", "rfa.h")]
EXPORTS_MAP = "rfa_exports.map"          # linker version script: only rfa_* is bindable


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build librfa_hip.so)")


def _digest(sources, extra=""):
    h = hashlib.sha256(extra.encode())
    for s in sources:
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, sources, extra=""):
    """Content-hash staleness (a `<target>.srchash` sidecar), NOT mtimes: repo snapshots copied to
    the GPU box do not preserve mtime order, and a spurious rebuild there would overwrite a shared
    library that the running test process has already mapped."""
    side = target + ".srchash"
    if not os.path.exists(target) or not os.path.exists(side):
        return True
    return open(side).read().strip() != _digest(sources, extra)


def _stamp(target, sources, extra=""):
    with open(target + ".srchash", "w") as f:
        f.write(_digest(sources, extra))


def _run(cmd, cwd=None):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=cwd, check=True)


def build_lib(force=False):
    hip_sources = HIP_SOURCES
    srcs = [os.path.join(CSRC, s) for s in hip_sources + [API_SOURCE]]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(CSRC, EXPORTS_MAP)]
    extra = ""
    if not force and not _stale(LIB, deps, extra):
        return LIB
    # the build id (rfa_build_id()): digest of exactly the sources that go into the binary
    build_id = _digest(deps, extra)[:16]
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fno-gpu-rdc", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wno-unused-result", "-Wno-inline-asm", f'-DRFA_BUILD_ID="{build_id}"']
    cmd += [os.path.join(CSRC, s) for s in hip_sources]
    cmd += ["-x", "hip", os.path.join(CSRC, API_SOURCE)]
    cmd += ["-Wl,--version-script=" + os.path.join(CSRC, EXPORTS_MAP), "-o", LIB]
    _run(cmd)
    _stamp(LIB, deps, extra)
    return LIB


def build_oracle(force=False):
    cmd = ["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-std=c11", ORACLE_SRC, "-o", ORACLE_LIB, "-lm"]
    if not force and not _stale(ORACLE_LIB, [ORACLE_SRC], " ".join(cmd)):
        return ORACLE_LIB
    _run(cmd)
    _stamp(ORACLE_LIB, [ORACLE_SRC], " ".join(cmd))
    return ORACLE_LIB


def build_selftest(force=False):
    build_lib(force)
    build_oracle(force)
    # the binary only depends on these sources + the two libraries' *interfaces*
    deps = [SELFTEST_SRC, os.path.join(CSRC, "rfa_common.hpp"), os.path.join(ROOT, "include", "rfa.h"), ORACLE_SRC]
    if not force and not _stale(SELFTEST_BIN, deps):
        return SELFTEST_BIN
    _run([_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value",
          SELFTEST_SRC, "-o", SELFTEST_BIN,
          "-L" + os.path.dirname(LIB), "-lrfa_hip", "-L" + os.path.dirname(ORACLE_LIB), "-lattn_ref",
          "-Wl,-rpath,$ORIGIN/../../ring-flash-attention_amd/ring_flash_attn",
          "-Wl,-rpath,$ORIGIN/../../oracle"])
    _stamp(SELFTEST_BIN, deps)
    return SELFTEST_BIN


def main(argv):
    force = "--force" in argv
    targets = [a for a in argv if not a.startswith("-")] or ["lib", "oracle"]
    for t in targets:
        if t == "lib":
            build_lib(force)
        else:
            {"oracle": build_oracle, "selftest": build_selftest}[t](force)


if __name__ == "__main__":
    main(sys.argv[1:])
