#!/usr/bin/env python3
"""Build libwg_rasterizer.so (the C-ABI library of include/wg_rasterizer.h) for gfx950 with hipcc.

In-tree build: objects go to wild-gaussians_amd/build/, the shared library next to the Python package
(wild-gaussians_amd/diff_gaussian_rasterization/libwg_rasterizer.so) so that it travels with the tree.
hipcc cross-compiles gfx950 without a GPU.  Usage: python wild-gaussians_amd/build.py [--force] [--verbose]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "diff_gaussian_rasterization", "libwg_rasterizer.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

# A/B builds of kernel variants (scripts/ab_variants.sh): WG_BUILD_VARIANT=<name> WG_EXTRA_FLAGS="-DWG_...=0" puts objects and the
# library under build/<name>/; load it with WG_RASTERIZER_LIB=<path> (diff_gaussian_rasterization/_C.py).
VARIANT = os.environ.get("WG_BUILD_VARIANT", "")
EXTRA = os.environ.get("WG_EXTRA_FLAGS", "").split()
# per-file extra flags of a variant build: WG_FILE_FLAGS="render_fwd.hip:-mllvm -amdgpu-skip-threshold=24;render_bwd.hip:..."
FILE_EXTRA = {k: v.split() for k, v in (kv.split(":", 1) for kv in filter(None, os.environ.get("WG_FILE_FLAGS", "").split(";")))}
if VARIANT:
    OBJ = os.path.join(OBJ, VARIANT)
    OUT = os.path.join(OBJ, "libwg_rasterizer.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I" + INCLUDE, "-I" + CSRC]
# per-file extra flags: the preprocess kernel must not fuse multiply-adds (integer outputs bit-exact vs oracle)
SOURCES = {
    "preprocess.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "binning.hip": ["-fno-slp-vectorize"],
    "render_fwd.hip": ["-fno-slp-vectorize"],
    "render_bwd.hip": ["-munsafe-fp-atomics", "-fno-slp-vectorize"],
    "preprocess_bwd.hip": ["-fno-slp-vectorize"],
    "api.hip": [],
    "knn.hip": ["-ffp-contract=off"],  # simple_knn.distCUDA2 replacement (include/wg_knn.h)
    "ssim.hip": [],                    # fused SSIM map fwd/bwd (include/wg_ssim.h)
    "activations.hip": ["-ffp-contract=off"],   # fused activations + 3-D filter fwd/bwd (include/wg_activations.h); same flags as
                                               # preprocess.hip, whose raw-parameter mode runs the same device functions (wg_act.h): same bits
    "densify.hip": [],                 # fused densification statistics (include/wg_densify.h)
    "sh_eval.hip": [],                 # fused eval_sh fwd/bwd (include/wg_sh_eval.h)
    "adam.hip": ["-ffp-contract=off"],   # fused Adam step (include/wg_adam.h); no FMA contraction: torch's update, op for op
}
HEADERS = ["wg_common.h", "wg_alpha.h", "wg_sort.h", "wg_act.h", os.path.join(INCLUDE, "wg_rasterizer.h"), os.path.join(INCLUDE, "wg_knn.h"),
           os.path.join(INCLUDE, "wg_ssim.h"), os.path.join(INCLUDE, "wg_activations.h"), os.path.join(INCLUDE, "wg_densify.h"), os.path.join(INCLUDE, "wg_adam.h"), os.path.join(INCLUDE, "wg_sh_eval.h")]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([HIPCC] + COMMON + extra + EXTRA + FILE_EXTRA.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _newer(OUT, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"])
    return OUT


def build_driver(force: bool = False) -> str:
    """tests/native/c_abi_driver.cpp: a torch-free program that drives the C-ABI (sanitizer pass, boundary evidence).  Built next to
    the library it links (build/[<variant>/]c_abi_driver, rpath = the library's directory)."""
    src = os.path.join(os.path.dirname(HERE), "tests", "native", "c_abi_driver.cpp")
    out = os.path.join(OBJ, "c_abi_driver")
    lib = build(force)
    if force or _newer(out, [src, lib, os.path.join(INCLUDE, "wg_rasterizer.h")]):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-pthread", "-I" + INCLUDE] + EXTRA + [src, "-o", out, "-L" + os.path.dirname(lib),
               "-lwg_rasterizer", "-Wl,-rpath," + os.path.dirname(lib)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return out


def build_exp_probe(force: bool = False) -> str:
    """tests/native/exp_probe.hip -> tests/native/libexp_probe.so: the device unit test of wg_alpha.h's restated `exp` expansion against the
    compiler's own `exp(float)`.  Test infrastructure; -ffp-contract=off like the reference's no-contraction checker build."""
    src = os.path.join(os.path.dirname(HERE), "tests", "native", "exp_probe.hip")
    out = os.path.join(os.path.dirname(HERE), "tests", "native", "libexp_probe.so")
    if force or _newer(out, [src, os.path.join(CSRC, "wg_alpha.h"), os.path.abspath(__file__)]):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I" + CSRC, src, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return out


def build_probes(force: bool = False) -> list:
    """Stand-alone measurement / reproducer programs (no product code; test infrastructure): scripts/pmc_calibration.hip (PMC byte counters on
    known byte counts, round 6), tests/native/malloc_async_lost_stores.cpp (the stream-ordered allocator's lost stores, EXPERIMENTS.md R5.10 /
    R5.12), experiments/at_05a7d0c/nan_min_probe.hip.  Binaries go to build/ (they travel to the GPU box with the tree)."""
    root = os.path.dirname(HERE)
    out = []
    os.makedirs(OBJ, exist_ok=True)
    for src, extra in ((os.path.join(root, "scripts", "pmc_calibration.hip"), ["-O3"]),
                       (os.path.join(root, "tests", "native", "malloc_async_lost_stores.cpp"), ["-O2", "-pthread"]),
                       (os.path.join(root, "experiments", "at_05a7d0c", "nan_min_probe.hip"), ["-O3"])):
        exe = os.path.join(OBJ, os.path.splitext(os.path.basename(src))[0])
        if force or _newer(exe, [src]):
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-w"] + extra + [src, "-o", exe], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + src + "\n" + r.stdout + r.stderr)
        out.append(exe)
    return out


def build_torch_binding(force: bool = False) -> str:
    """csrc/torch_binding.cpp -> diff_gaussian_rasterization/_C_torch.so: the compiled torch binding of the C-ABI (INTEGRATION.md section 2 as
    a file; the ctypes binding stays the default).  Host C++ only: g++ against torch's headers, linked to libwg_rasterizer.so next to it
    (rpath $ORIGIN)."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    lib = build(force)
    src = os.path.join(CSRC, "torch_binding.cpp")
    out = os.path.join(os.path.dirname(lib), "_C_torch" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    if not (force or _newer(out, [src, lib, os.path.join(INCLUDE, "wg_rasterizer.h"), os.path.abspath(__file__)])):
        return out
    tlib = ce.library_paths()[0]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_C_torch",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations",
           "-I" + INCLUDE, "-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"]] + ["-I" + p for p in ce.include_paths()] + \
          [src, "-o", out, "-L" + tlib, "-L" + os.path.dirname(lib), "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
           "-lwg_rasterizer", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr[-4000:])
    return out


if __name__ == "__main__":
    if "--torch-binding" in sys.argv:
        print(build_torch_binding(force="--force" in sys.argv))
        sys.exit(0)
    if "--driver" in sys.argv:
        print(build_driver(force="--force" in sys.argv))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
