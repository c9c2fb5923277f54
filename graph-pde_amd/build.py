"""Build libgpde.so (hand-written HIP for gfx950 + the C ABI of include/gpde.h) in-tree.

    python graph-pde_amd/build.py            # incremental: recompiles sources newer than the .so
    python graph-pde_amd/build.py --force

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the GPU
box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(REPO, "include")
# Developer builds (ablations: -DGPDE_ABL_*, clock probes: -DGPDE_*_TIMING) carry a suffix and live OUTSIDE the package, under the
# git-ignored scripts/ubench/ - the package directory holds the one production library only
_SUFFIX = os.environ.get("GPDE_BUILD_SUFFIX", "")
DEVDIR = os.path.join(REPO, "scripts", "ubench", "lib")
LIB = os.path.join(DEVDIR, f"libgpde{_SUFFIX}.so") if _SUFFIX else os.path.join(PKG, "libgpde.so")
OBJDIR = os.path.join(DEVDIR, "build" + _SUFFIX) if _SUFFIX else os.path.join(PKG, "build")

SOURCES = ["gpde_api.hip", "gpde_csr.hip", "gpde_pack.hip", "gpde_fused.hip", "gpde_fused_f16v3.hip", "gpde_fused_f16v6.hip",
           "gpde_zagg.hip", "gpde_prep.hip", "gpde_gemm3.hip", "gpde_gemm.hip", "gpde_gemm_f16s.hip", "gpde_bwd.hip", "gpde_edge_bwd3.hip", "gpde_bwd_onepass.hip", "gpde_graph.hip", "gpde_weconv.hip", "gpde_cellgraph.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC,
         "-Wall", "-Wno-unused-function",
         "-fvisibility=hidden"]        # only the GPDE_API entry points of include/gpde.h are dynamic symbols


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(INCLUDE, "gpde.h"))
    return hdrs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


LAST_REPORT = {"compiled": [], "reused": [], "linked": False}      # what the last build() call did


def _compile(src, force, extra):
    obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if not (force or _stale(obj, [path] + _deps())):
        LAST_REPORT["reused"].append(src)
    else:
        LAST_REPORT["compiled"].append(src)
        cmd = [HIPCC] + FLAGS + list(extra) + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, extra_flags=()) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    LAST_REPORT.update(compiled=[], reused=[], linked=False)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, extra_flags), SOURCES))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        LAST_REPORT["linked"] = True
    return LIB


if __name__ == "__main__":
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    print(build(force="--force" in sys.argv, extra_flags=extra))
