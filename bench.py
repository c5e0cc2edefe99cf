#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Gaussian-splat rasterizer.

Metric (BASELINE.json): train iters/s + forward Mpix/s at 1M Gaussians @ 1920x1080, plus dL/dtheta max-rel-err vs the
reference restatement (the CPU oracle).  A "step" is one pass of the hot path over one view: forward + backward of the
rasterization operator (through the drop-in GaussianRasterizer autograd surface -> C-ABI -> HIP kernels) with a fixed
cotangent, on synthetic data of SURVEY.md 8(d).  Inputs are resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1 is view-parallel (one camera per GPU, Gaussians replicated, all-reduce of the scalar loss only): weak scaling.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "wild-gaussians_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
# SURVEY 8(d)'s secondary (compute) view of the two render kernels, "report, not judged": fp32 vector peaks of the part,
# 256 CUs x 4 SIMDs x 16 lanes x 2 flop (FMA) x 2.4 GHz = 78.6 TFLOP/s with plain instructions, 157.3 with packed (v_pk_fma_f32) ones
FP32_PLAIN_PEAK_TFLOPS = 256 * 4 * 16 * 2 * 2.4 / 1e3
FP32_PACKED_PEAK_TFLOPS = 2 * FP32_PLAIN_PEAK_TFLOPS
# flops SURVEY 8(d) attributes to a pair: "~20 flop + 1 exp per evaluated (pixel, entry) pair forward, ~60 flop + 10 reduced adds per
# contributing pair backward"
FLOPS_PER_EVALUATED_PAIR_FWD = 21
FLOPS_PER_CONTRIBUTING_PAIR_BWD = 70


def algorithmic_bytes(stage: str, P: int, V: int, R: int, N: int, tiles: int, M: int, sh: bool) -> float:
    """ALGORITHMIC bytes per launch, SURVEY.md 8(d) formulas split per kernel (each compulsory datum counted once per
    kernel boundary).  Stated again in DESIGN.md."""
    c_in = 12 * M if sh else 12
    table = {
        "preprocess": P * (44 + c_in) + 8 * P + V * (4 + 48 + 8 + 24),  # attrs in; radii+tiles out; depth+record+rect+cov3D
        "scan": 8 * P,
        "duplicate_keys": 20 * P + 12 * R,
        "sort": 24 * R * ((32 + int(np.ceil(np.log2(max(tiles, 2)))) + 7) // 8),
        "tile_ranges": 8 * R + 8 * tiles,
        "render_forward": R * (4 + 36) + 28 * N + 16 * tiles,
        "render_backward": 28 * N + 12 * N + 40 * R + 40 * R,
        "preprocess_backward": V * (100 + 40) + V * 90 + P * (12 + 12 + 16 + 4) + (24 * M * V if sh else 0),
    }
    return float(table[stage])


def design_bytes(stage: str, P: int, V: int, R: int, N: int, tiles: int, M: int, sh: bool, walked: int) -> float:
    """Bytes THIS design's kernel of the stage has to move at least once: the 48-byte splat record, 4-byte bucket
    entries, only the instances the compositing walks actually reach (`walked` = sum of tile_last).  Every datum counted once per
    kernel, so this can never exceed what the HBM delivers; the PMC traffic is at or above it."""
    c_in = 12 * M if sh else 12
    hist = 512 * tiles * 4   # chunk x tile histograms of the LDS counting sort (BIN_CHUNKS = 512)
    table = {
        "preprocess": P * (44 + c_in) + 8 * P + V * (4 + 48 + 8 + 24),
        "scan": 8 * P + 3 * hist + 16 * tiles,            # rects in; histograms written, column-scanned in place, read by the tile scan
        "duplicate_keys": 8 * P + hist + 4 * R,           # rects + chunk bases in; one 4-byte bucket entry per instance out
        "sort": 4 * R + 4 * V + 4 * R,                    # bucket in, one depth per visible Gaussian, point_list out
        "tile_ranges": 8 * tiles,
        "render_forward": 4 * walked + 48 * V + 28 * N + 12 * tiles,
        "render_backward": 4 * walked + 48 * V + 28 * N + 40 * V,   # + one 10-float accumulator update per visible Gaussian
        "preprocess_backward": V * (100 + 40) + V * 90 + P * (12 + 12 + 16 + 4) + (24 * M * V if sh else 0),
    }
    return float(table[stage])


OPERATOR_SOURCES = ("api.hip", "binning.hip", "preprocess.hip", "preprocess_bwd.hip", "render_fwd.hip", "render_bwd.hip", "wg_act.h",
                    "wg_common.h", "wg_alpha.h", "wg_sort.h")


OPT_IN_SOURCES = ("knn.hip", "ssim.hip", "activations.hip", "densify.hip", "adam.hip", "sh_eval.hip")   # SURVEY 8f kernels: in no bench.py stage


def kernel_source_sha() -> str:
    """Hash of the operator's kernel sources (everything a bench.py stage runs) + the build script with its flags: stamps
    profiles/pmc_traffic.json to the code it was measured on.  The opt-in kernels either side of the path (knn, ssim, activations,
    densify, adam, sh_eval) are not part of any stage and not part of the stamp."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "wild-gaussians_amd", "csrc")
    for f in sorted(OPERATOR_SOURCES) + ["../build.py"]:
        if f.endswith((".hip", ".h", ".py")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


PRODUCT_LIB = os.path.join(ROOT, "wild-gaussians_amd", "diff_gaussian_rasterization", "libwg_rasterizer.so")


def _elf_sections(b: bytes) -> dict:
    """name -> [(offset, size)] of a little-endian ELF64 image (the library, or a gfx950 code object inside its fat binary)."""
    import struct
    shoff = struct.unpack_from("<Q", b, 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
    sh = lambda i: struct.unpack_from("<IIQQQQIIQQ", b, shoff + i * shentsize)   # noqa: E731
    so = sh(shstrndx)
    names = b[so[4]:so[4] + so[5]]
    out = {}
    for i in range(shnum):
        s_ = sh(i)
        out.setdefault(names[s_[0]:names.index(b"\0", s_[0])].decode(), []).append((s_[4], s_[5]))
    return out


_DEVICE_SHA = {}
RUNNING_LIB = PRODUCT_LIB   # main() sets it to the library the binding loaded (a variant build under WG_RASTERIZER_LIB is stamped as itself)


def device_code_sha(lib_path: str = PRODUCT_LIB) -> str:
    """Identity of the DEVICE code of a built library: SHA-256 over the `.text` and `.rodata` (instructions + kernel descriptors) of every
    gfx950 code object in its `.hip_fatbin` section, in link order.  Symbol names, notes and the per-file `__hip_cuid_<hash of the path>`
    symbols are not in it, so the same sources and flags give the same value from any build directory, and an edit of host code or of a
    comment leaves it alone.  The profiles under profiles/ (PMC traffic, pair counts) are measurements of device code: they carry this
    stamp beside `kernel_source_sha`, and either one matching the running tree makes them current."""
    import hashlib
    import struct
    if lib_path in _DEVICE_SHA:
        return _DEVICE_SHA[lib_path]
    with open(lib_path, "rb") as fh:
        b = fh.read()
    (off, size), = _elf_sections(b)[".hip_fatbin"]
    fb = b[off:off + size]
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    h = hashlib.sha256()
    n = 0
    pos = fb.find(magic)
    while pos >= 0:   # one bundle per translation unit
        cnt, = struct.unpack_from("<Q", fb, pos + 24)
        p = pos + 32
        for _ in range(cnt):
            o, sz, ts = struct.unpack_from("<QQQ", fb, p)
            triple = fb[p + 24:p + 24 + ts]
            p += 24 + ts
            if b"gfx950" in triple and sz:
                co = fb[pos + o:pos + o + sz]
                secs = _elf_sections(co)
                for name in (".text", ".rodata"):
                    for so_, ss in secs.get(name, []):
                        h.update(name.encode() + struct.pack("<Q", ss) + co[so_:so_ + ss])
                n += 1
        pos = fb.find(magic, pos + 1)
    if n == 0:
        raise RuntimeError(f"{lib_path}: no gfx950 code object found in .hip_fatbin")
    _DEVICE_SHA[lib_path] = h.hexdigest()[:16]
    return _DEVICE_SHA[lib_path]


def profile_stamps() -> dict:
    """The two stamps a profile file is matched on (and that pmc_traffic.py / count_pairs.py write): the operator's kernel sources + build
    script, and the device code of the product library as built."""
    out = {"kernel_source_sha": kernel_source_sha()}
    try:
        out["device_code_sha"] = device_code_sha(RUNNING_LIB)
    except (OSError, KeyError, ValueError, RuntimeError) as ex:
        out["device_code_sha"] = None
        out["device_code_sha_error"] = f"{type(ex).__name__}: {ex}"[:200]
    return out


def stamp_matches(rec: dict) -> bool:
    """True when a profile record was measured on what is running now: same kernel sources, or -- sources edited without touching device
    code (host code, comments, the build script's host parts) -- the same device code."""
    now = profile_stamps()
    if rec.get("kernel_source_sha") == now["kernel_source_sha"]:
        return True
    return bool(rec.get("device_code_sha")) and rec.get("device_code_sha") == now.get("device_code_sha")


def load_pmc(workload_key: str):
    """PMC HBM traffic per stage (separate rocprofv3 --pmc passes, scripts/profile_gpu.sh), only when it was collected on this very
    workload AND on this very device code (stamp_matches); otherwise {} and the reason."""
    import glob
    why = "no profiles/pmc_traffic*.json"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_traffic*.json"))):   # one file per profiled workload
        try:
            with open(path) as f:
                pt = json.load(f)
        except (OSError, ValueError):
            continue
        name = os.path.basename(path)
        if pt.get("workload") != workload_key:
            why = f"{name} is for another workload ({pt.get('workload')})" if why.startswith("no ") else why
            continue
        if not stamp_matches(pt):
            why = (f"{name} is stale: measured on kernel sources {pt.get('kernel_source_sha')} / device code {pt.get('device_code_sha')}, "
                   f"these are {profile_stamps()}")
            continue
        return pt.get("stages", {}), (f"{name}: pmc passes of {pt.get('collected', '?')} on kernel sources {pt.get('kernel_source_sha')}, "
                                      f"device code {pt.get('device_code_sha')}")
    return {}, why


def load_pair_counts(workload_key: str):
    """Per-launch pair counts of K8 / K9 from the counting variant build (tests/tools/count_pairs.py -> profiles/pair_counts*.json), only
    when collected on this workload and this device code (stamp_matches); else None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "pair_counts*.json"))):
        try:
            with open(path) as f:
                pc = json.load(f)
        except (OSError, ValueError):
            continue
        if pc.get("workload") == workload_key and stamp_matches(pc):
            pc["file"] = os.path.basename(path)
            return pc
    return None


def compute_view(dom: str, launch_ms: float, pmc_row, pairs, ref_pairs) -> dict:
    """SURVEY 8(d)'s secondary ceiling for a render kernel, from COUNTS: useful flops = the pairs the REFERENCE's walk evaluates (K8) or
    differentiates (K9) x the survey's flops per pair, over this run's launch time, against the part's fp32 vector peaks.  ref_pairs: the
    reference walk's counts (CPU oracle: this run's cpu_baseline leg, or the counting build's record); pairs: what THIS kernel evaluates
    (counting build); pmc_row: the VALU counters.  Every ratio in here is at most 1 by construction."""
    t = launch_ms * 1e-3
    out = {"kernel": dom, "peaks_TFLOPs": {"fp32_plain": round(FP32_PLAIN_PEAK_TFLOPS, 1), "fp32_packed": round(FP32_PACKED_PEAK_TFLOPS, 1)}}
    if ref_pairs:
        n, f = ((ref_pairs["pairs_evaluated"], FLOPS_PER_EVALUATED_PAIR_FWD) if dom == "render_forward" else
                (ref_pairs["pairs_blended"], FLOPS_PER_CONTRIBUTING_PAIR_BWD))
        tf = n * f / t / 1e12
        out.update({"algorithmic_pairs_per_launch": int(n), "flops_per_pair": f, "pairs_are": ("evaluated by the reference's walk" if dom == "render_forward"
                                                                                                else "contributing (blended) pairs"),
                    "useful_TFLOPs": round(tf, 2), "frac_of_fp32_plain_peak": round(tf / FP32_PLAIN_PEAK_TFLOPS, 4),
                    "frac_of_fp32_packed_peak": round(tf / FP32_PACKED_PEAK_TFLOPS, 4), "pairs_source": ref_pairs.get("source")})
    if pairs and pairs.get(dom):
        k = pairs[dom]
        useful = k.get("pairs_passing_both_skips", k.get("pairs_contributing"))
        out["this_kernel"] = {**k, "lane_pairs_useful_over_evaluated": round(useful / max(1, k["pairs_evaluated"]), 4), "source": pairs.get("file")}
    if pmc_row and pmc_row.get("SQ_INSTS_VALU"):
        v = {"wave_instructions_per_launch": pmc_row["SQ_INSTS_VALU"], "rate_G_wave_instr_per_s": round(pmc_row["SQ_INSTS_VALU"] / t / 1e9, 1)}
        if pmc_row.get("SQ_THREAD_CYCLES_VALU") and pmc_row.get("SQ_ACTIVE_INST_VALU"):
            v["thread_utilisation"] = round(pmc_row["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc_row["SQ_ACTIVE_INST_VALU"]), 4)   # rocprofiler's VALUUtilization / 100
        for c in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32"):
            if pmc_row.get(c) is not None:
                v[c] = pmc_row[c]
        if all(pmc_row.get(c) is not None for c in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32")):
            # executed float32 flops by the counters, at full lane width (an upper bound: masked lanes count) -- rocprofiler's VALU FLOPs expression
            ex = 64.0 * (2 * pmc_row["SQ_INSTS_VALU_FMA_F32"] + pmc_row["SQ_INSTS_VALU_MUL_F32"] + pmc_row["SQ_INSTS_VALU_ADD_F32"] + pmc_row["SQ_INSTS_VALU_TRANS_F32"])
            v["executed_f32_TFLOPs_at_full_lane_width"] = round(ex / t / 1e12, 2)
        v["note"] = ("a rate, not a fraction of a peak: SQ_INSTS_VALU counts every issued vector instruction, and one issued with no active lane (a strip whose "
                     "pixels have all stopped, the not-taken side of a short branch) retires in about a cycle instead of four -- which is why round 4's "
                     "'fraction of 614 G wave-instr/s' read 1.0 - 1.3 (dense frames, where most lanes have saturated, the highest)")
        out["valu"] = v
    return out


def governing_roofline(dom: str, d: dict, pmc_row, launch_ms: float, survey_bytes: float, design_b: float) -> dict:
    """The bench line's `roofline` object for the dominant kernel `dom`, as SURVEY 8(d) defines it: bound = HBM,
    achieved = ALGORITHMIC bytes per launch (8d's formula x this frame's P, V, R, N) / the kernel's average launch time (HIP events on the
    launch stream, this run), peak = 8 TB/s, frac = achieved / peak -- the number north_star's >= 0.60 target is judged on.
    d = the kernel's stage_rooflines row, pmc_row = its PMC record or None (no PMC pass stamped to the running sources)."""
    by_traffic = "hbm_traffic_GBps" in d
    s8d = d["reference_scheme_equiv_GBps"]
    r = {"bound": "hbm", "kernel": dom, "achieved": s8d, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(s8d / HBM_PEAK_GBS, 4),
         "frac_basis": "by_survey_8d_bytes", "bytes_per_launch": survey_bytes,
         "avg_launch_ms": round(launch_ms, 4), "traffic": pmc_row["hbm_bytes"] if by_traffic else None}
    if s8d > HBM_PEAK_GBS:
        # a binning stage: 8(d) counts the REFERENCE scheme's bytes (6 radix passes over 24-byte pairs), which this design does not move -- the
        # equivalent rate exceeds the peak and is not a fraction of it; the stage is then priced on the bytes its own kernels must move
        r.update({"achieved": d["design_GBps"], "frac": d["frac_of_peak_by_design_bytes"], "frac_basis": "by_design_bytes", "bytes_per_launch": design_b,
                  "survey_8d_equivalent_GBps": s8d})
    if by_traffic:
        r["traffic_over_algorithmic_bytes"] = round(pmc_row["hbm_bytes"] / survey_bytes, 3)
    r["other_byte_counts"] = {"by_design_bytes": {"bytes_per_launch": design_b, "achieved": d["design_GBps"], "frac": d["frac_of_peak_by_design_bytes"]},
                              "by_pmc_traffic": ({"bytes_per_launch": pmc_row["hbm_bytes"], "achieved": d["hbm_traffic_GBps"],
                                                  "frac": d["frac_of_peak_by_traffic"]} if by_traffic else None)}
    if dom in ("render_forward", "render_backward"):
        r["note"] = ("the render kernels are arithmetic-bound, not HBM-bound: their traffic is below their algorithmic bytes (a tile band's records stay in its "
                     "XCD's L2) and LDS bank conflicts are 0; north_star's >= 0.60 of 8 TB/s is met by the two streaming per-Gaussian kernels (stage_rooflines) "
                     "and NOT by these two -- `compute` carries SURVEY 8(d)'s secondary view from counted pairs")
    return r


def host_step_summary(stamps) -> dict:
    """Where the timed region's time went, step by step, from the host clock read after every call inside it (the host is in step with
    the GPU: a forward call returns behind its frame's scan).  The mean of these is ms_per_step up to the closing synchronize."""
    d = [1e3 * (b - a) for a, b in zip(stamps[:-1], stamps[1:])]
    if not d:
        return {}
    srt = sorted(d)
    out = {"p50": round(srt[len(srt) // 2], 4), "min": round(srt[0], 4), "max": round(srt[-1], 4), "argmax": int(d.index(srt[-1])),
           "mean": round(sum(d) / len(d), 4), "steps_over_1.25x_p50": int(sum(1 for x in d if x > 1.25 * srt[len(srt) // 2]))}
    out["slow_steps"] = [[i, round(x, 3)] for i, x in enumerate(d) if x > 1.15 * srt[len(srt) // 2]][:16]   # [index in the region, ms]
    if len(d) <= 64:
        out["series"] = [round(x, 3) for x in d]
    else:
        out["first_8"] = [round(x, 3) for x in d[:8]]
    return out


def self_launch(n: int):
    """`python bench.py --gpus N` without a launcher: replace this process by `torch.distributed.run` with N ranks of the same
    command line (one process per GPU, rendezvous on 127.0.0.1, a free port).  With fewer than N devices visible (the 1-GPU test
    box) the ranks share devices, which RCCL refuses: the collective transport falls back to gloo there and the JSON line says so."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # every rank gets its own slice of the host cores (wg_viewparallel.pin_rank) and a thread pool no larger than that slice
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, cores // max(n, 1)))))
    if torch.cuda.is_available() and torch.cuda.device_count() < n and env.get("WG_DIST_BACKEND") != "gloo":
        raise SystemExit(f"--gpus {n} but only {torch.cuda.device_count()} HIP device(s) visible: one process per GPU needs {n}.  "
                         "(WG_DIST_BACKEND=gloo lets ranks share devices over the host-side transport: a test of the launch flow, "
                         "not a scaling measurement.)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def config_legs(device, deg_unused, with_oracle: bool, clouds: dict) -> dict:
    """BASELINE.json configs 2, 3 and 5 at their QUOTED sizes, a short leg each, appended to the headline line as `configs` (VERDICT r5 item
    4: until round 6 only builder-run files held these numbers).  Per leg: the rate over a short timed region (barrier-free, one GPU:
    synchronize on both sides), `pins_ok` -- num_rendered and the SHA-256 of radii, n_contrib and final_T equal what the REFERENCE's own
    kernels (-ffp-contract=off build) gave for this frame on an MI355X (tests/golden/ref_hip_fullsize_sha256.json; no reference binary
    needed) --, and for the fwd+bwd legs `grad_worst`: max over the gradient tensors of max|g - g_oracle| / max|g_oracle| against the CPU
    oracle on the same inputs.  clouds: name -> a callable giving (cloud, cam, deg) (config 5's 10 M Gaussians take a while of single-threaded
    numpy: generated here, one leg at a time, never beside a timed region)."""
    import gc
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import fullsize_frames as FF
    import wg_scenes as S
    from diff_gaussian_rasterization import GaussianRasterizer, _C
    from tests.wg_testlib import make_settings, to_dev, compare_grads
    pins = json.load(open(FF.PINS))["frames"]
    legs = {}
    for key, name, fwd_only in (("2", "config2_500k_1080p_sh3", False), ("3", "config3_3M_1600x1200_precomp", False), ("5", "config5_10M_4K_sh3", True)):
        try:
            got = clouds[name]() if callable(clouds[name]) else clouds[name]
            cloud, cam, deg = got
            P, W, H, colours, _k = FF.FRAMES[name]
            rs = make_settings(cam, deg, device=device)
            rast = GaussianRasterizer(rs)
            t = {k: to_dev(v, device).requires_grad_(not fwd_only) for k, v in cloud.items()}
            means2D = torch.zeros((P, 3), device=device, requires_grad=not fwd_only)
            cot_np = S.make_cotangent(W, H)
            cot = to_dev(cot_np, device)

            def call():
                return rast(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t.get("shs"), colors_precomp=t.get("colors_precomp"),
                            scales=t["scales"], rotations=t["rotations"])

            def step():
                if fwd_only:
                    with torch.no_grad():
                        return call()[0]
                for v in t.values():
                    v.grad = None
                means2D.grad = None
                color = call()[0]
                color.backward(cot)
                return color
            # pins: the native module's image state of one forward pass
            with torch.no_grad():
                e = torch.Tensor([])
                R_, _c, radii, gb, bb, ib = _C.rasterize_gaussians(
                    rs.bg, t["means3D"], t["colors_precomp"] if "colors_precomp" in t else e, t["opacities"], t["scales"], t["rotations"], 1.0, e,
                    rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset, H, W, t["shs"] if "shs" in t else e, deg,
                    rs.campos, False, False)[:6]
                im = _C.view_image(ib, H, W)
                d = FF.digest(R_, radii.cpu().numpy(), im["n_contrib"].cpu().numpy(), im["final_T"].cpu().numpy())
                del _c, gb, bb, ib, im
            leg = {"workload": f"{P} Gaussians, {W}x{H}, {'SH deg 3' if colours == 'sh' else 'precomputed colours'}, {'forward only' if fwd_only else 'fwd+bwd'}",
                   "pins_ok": d == pins[name], "num_rendered": int(R_)}
            if d != pins[name]:
                leg["pins_mismatch"] = [k for k in d if d[k] != pins[name][k]]
            for _ in range(120 if fwd_only else 8):   # (config 5: the near / far split's adaptive aim settles within ~60 unsynchronised frames of a scene)
                step()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize(device)
            one = max(time.perf_counter() - t0, 1e-5)
            steps = int(max(20, min(600, 1.2 / one)))   # ~1.2 s of steps
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            leg["steps"] = steps
            leg["ms_per_step"] = round(1e3 * dt / steps, 4)
            leg["fps" if fwd_only else "iters_per_s"] = round(steps / dt, 2)
            # where the leg's time goes: 40 more steps with the library's per-stage events on (outside the timed region above)
            _C.profile_reset(); _C.profile_enable(True)
            for _ in range(40):
                step()
            torch.cuda.synchronize(device)
            st_ = _C.profile_read(); _C.profile_enable(False)
            leg["stages_ms"] = {k: round(v[0] / 40.0, 4) for k, v in st_.items() if v[1] > 0}   # per step (a stage may be several launches)
            if fwd_only:
                leg["near_far_split"] = {k: _C.get_option(k) for k in ("near_per_tile_now", "near_floor_now", "near_far_tiles_last", "near_split_backoff")}
            if not fwd_only and with_oracle:
                from oracle import oracle
                o = oracle.run_scene(cloud, cam, sh_degree=deg, cotangent=cot_np)
                step()
                torch.cuda.synchronize(device)
                grads = {"means3D": t["means3D"].grad, "means2D": means2D.grad, "opacities": t["opacities"].grad, "scales": t["scales"].grad,
                         "rotations": t["rotations"].grad, ("sh" if "shs" in t else "colors_precomp"): (t["shs"] if "shs" in t else t["colors_precomp"]).grad}
                errs = compare_grads({k: v.detach().cpu().numpy() for k, v in grads.items()}, o["grads"])
                leg["grad_worst"] = float(f"{max(errs.values()):.3e}")
                leg["grad_checker"] = "CPU oracle, same inputs: max over tensors of max|g - g_ref| / max|g_ref|"
                del o
            legs[key] = leg
            del t, means2D, cot, rast, cloud, got
            clouds[name] = None
        except Exception as ex:  # noqa: BLE001 -- an extra of the line, never a reason to lose it
            legs[key] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        gc.collect()
        torch.cuda.empty_cache()
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--colors", choices=["sh", "precomp"], default="sh")
    ap.add_argument("--scale-mult", type=float, default=1.0, help="multiply the Gaussian scales (denser per-tile lists; 1.0 = SURVEY 8d recipe)")
    ap.add_argument("--forward-only", action="store_true", help="stress mode: time only the forward pass (e.g. 10M Gaussians @ 4K)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle leg (parity + CPU timing)")
    ap.add_argument("--no-camera-sequence", action="store_true", help="skip the eight-camera sequence block (speculation misses over differing frames)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-stage HIP events in the timed region")
    ap.add_argument("--baseline-iters-per-s", type=float, default=None,
                    help="the 1-GPU value of this metric: the line then carries scaling_efficiency = value / (N * baseline)")
    ap.add_argument("--views", type=int, default=None,
                    help="views per step over ALL ranks (default: one per rank = --gpus, weak scaling).  BASELINE config 4 is `--views 8`: a fixed batch of the "
                         "eight cameras, rank r renders views r, r+N, ... of it every step (strong scaling; `--gpus 1 --views 8` is its N = 1 point)")
    ap.add_argument("--no-config-legs", action="store_true",
                    help="skip the short legs at BASELINE configs 2, 3 and 5's quoted sizes that the default (headline) run appends as `configs`")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams a rank deals its views of a step to, round robin (view k of the rank on stream k %% S): independent views overlap on one "
                         "GPU -- view k+1's projection / binning under view k's compositing.  1 = everything on one stream (the headline protocol)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="wg_set_option(NAME, VALUE) before the run (A/B of library options, e.g. grad_record=0); recorded in the JSON line")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)   # does not return

    import wg_scenes as S
    import wg_viewparallel as VP
    from diff_gaussian_rasterization import GaussianRasterizer, _C
    from tests.wg_testlib import make_settings, to_dev, compare_forward, compare_grads

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the rasterizer has no CPU path")
    # A bench step is ONE FULL pass of the hot path.  The binding's geometry reuse (consecutive calls over the same geometry tensor objects
    # with other precomputed colours skip projection and binning: WildGaussians' second call per step) would turn every step after the
    # first into a recolouring here, where the same tensors are rasterized again and again: off, whatever the caller's options say.
    _C.set_option("geometry_reuse", 0)
    global RUNNING_LIB
    RUNNING_LIB = _C._LIB_PATH
    for kv in args.option:
        k, v = kv.split("=", 1)
        if k == "geometry_reuse" and int(v) != 0:
            raise SystemExit("bench.py times full passes: geometry_reuse stays off (scripts/bench_wildgaussians_step.py measures it where it belongs)")
        _C.set_option(k, int(v))
    rank, local_rank, world = VP.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         f"or without a launcher (bench.py starts its own ranks)")
    device = torch.device("cuda", local_rank)

    W, H, P = args.width, args.height, args.gaussians
    N = W * H
    sh_degree = 3 if args.colors == "sh" else None
    # the default (headline) run also carries short legs at BASELINE configs 2, 3, 5 (config_legs)
    want_legs = (world == 1 and rank == 0 and not args.no_config_legs and not args.forward_only and args.views is None and args.scale_mult == 1.0 and
                 (P, W, H, args.colors) == (1_000_000, 1920, 1080, "sh") and args.streams == 1 and not args.option)
    # (their clouds are generated AFTER the headline's passes, one leg at a time: a generator thread beside the timed region was measured to
    #  cost it 4 % and single 3 ms steps -- the interpreter lock changes hands in 5 ms slices)
    cloud = S.make_cloud(P, W, H, sh_degree=sh_degree, seed=0, scale_mult=args.scale_mult)
    V_total = args.views if args.views is not None else world
    if V_total < world:
        raise SystemExit(f"--views {V_total} < --gpus {world}: a rank would have no view")
    my_views = VP.views_for_rank(V_total, rank, world)        # rank r renders views r, r + N, ...; default: view r alone
    cams = [VP.view_cameras(V_total, W, H)[k] for k in my_views]   # the base camera yawed by k * 5 degrees (config 4); view 0 = the base camera
    cam = cams[0]
    cot_np = S.make_cotangent(W, H)
    deg = 3 if args.colors == "sh" else 0
    M = 16 if args.colors == "sh" else 0

    rs = make_settings(cam, deg, device=device)
    rasts = [GaussianRasterizer(make_settings(c, deg, device=device)) for c in cams]
    rast = rasts[0]
    t = {k: to_dev(v, device).requires_grad_(not args.forward_only) for k, v in cloud.items()}
    means2D = torch.zeros((P, 3), device=device, requires_grad=not args.forward_only)
    cot = to_dev(cot_np, device)
    cot_flat = cot.reshape(-1)

    def call(r=None):
        return (r or rast)(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t.get("shs"),
                           colors_precomp=t.get("colors_precomp"), scales=t["scales"], rotations=t["rotations"])

    loss_stream = VP.LossStream(device)
    # --streams S: the rank's views of a step are independent (the operator has no cross-view state), so view k goes to stream
    # k % S.  Every timed region ends with a device-wide synchronize, which covers all of them; inputs are only read.  Each view's gradient
    # tensors are allocated (and freed, when the next view of that stream drops them) on the stream that computes them.
    n_streams = max(1, min(args.streams, len(rasts)))
    side_streams = [torch.cuda.Stream(device) for _ in range(n_streams)] if n_streams > 1 else []
    if side_streams:
        torch.cuda.synchronize(device)   # the uploads above are through before another stream reads them

    def one_view(r):
        for v in t.values():
            v.grad = None
        means2D.grad = None
        color, radii, acc = call(r)
        color.backward(cot)
        # the loss <image, cotangent> and the only collective (4 bytes, queued asynchronously: wg_viewparallel.LossStream); the timed
        # region's closing synchronize covers it
        return loss_stream.submit(color, cot_flat)

    def train_step():
        # one step = this rank's views of the batch, one full forward + backward pass each (one view per rank unless --views says otherwise)
        loss = None
        for k, r in enumerate(rasts):
            if side_streams:
                with torch.cuda.stream(side_streams[k % n_streams]):
                    loss = one_view(r)
            else:
                loss = one_view(r)
        return loss

    def fwd_step():
        with torch.no_grad():
            out = None
            for k, r in enumerate(rasts):
                if side_streams:
                    with torch.cuda.stream(side_streams[k % n_streams]):
                        out = call(r)[0]
                else:
                    out = call(r)[0]
            return out

    t_train_local = [0.0]

    def timed(fn, steps, keep=None, stamps=None, warmup=0):   # barrier + synchronize on both sides of exactly `steps` calls, max over ranks
        return VP.timed_region(fn, steps, device, keep, stamps, warmup)

    if args.forward_only:
        train_step = fwd_step
    # Order of the passes.  The per-stage pass (the same K steps with a pair of HIP events around every stage, on the stream the
    # kernels are launched on: wg_profile_* in the C-ABI library) runs FIRST, the W warm-up steps and the timed region after it.
    # Why: a fresh process starts on an idle GPU, and the device needs ~30 ms of load to reach its steady clocks -- with the driver's
    # `--steps 20 --warmup 5` the timed region WAS that ramp (round 3: mean 1.07 ms vs p50 1.00; round 4's host-clock series of the
    # region, `timed_region_host_ms`: 1.20 ms at step 3 falling monotonically to 1.07 at step 19, profiles/r4/driver_cmd_series.txt).
    # Nothing is dropped from the timed region: W untimed steps, then exactly K full steps between barrier + synchronize pairs.
    # The interpreter's cycle collector runs ONCE, here, and stays off until the forward-only passes are through: a collection takes
    # ~57 ms in this process (scripts/diag_idle_ramp.py), the GPU idles meanwhile, drops its clocks and needs 20 - 30 ms of load to regain
    # them (same script: the steps after a 10 ms pause run 1.21, 1.13, 1.13, 1.11 ... 1.05 ms against 1.04 without one).  Reference
    # counting still frees every step's tensors; no work of a step is skipped.
    import gc
    gc.collect()
    gc.disable()
    stages = {}
    roof_ctx = None
    ref_pairs = None   # the reference walk's pair counts of this frame, from the CPU oracle leg below (or the counting build's record)
    t_train_profiled = None
    if not args.no_profile:
        # (first calls of the process: lazy initialisations stay out of the stage times, and the device reaches its steady clocks before the
        #  stage events are read -- without the ramp the driver's 20-step run timed K9 at 0.44 ms where every longer run and rocprof's steady
        #  launches read 0.40: `roofline.achieved` is a statement about the kernel, not about the first 20 ms of a process)
        for _ in range(30):
            train_step()
        torch.cuda.synchronize(device)
        _C.profile_reset()
        _C.profile_enable(True)
        t_train_profiled = timed(train_step, args.steps)
        stages = _C.profile_read()
        _C.profile_enable(False)
    # W untimed warm-up steps, then the timed region: exactly K steps, no instrumentation inside (recording a HIP event costs ~15 us on
    # this stack, and a step would carry 16 of them).  The warm-up steps run inside timed_region, behind its collector pass and directly
    # in front of its opening barrier + synchronize: the GPU does not sit idle between its warm-up and its timed steps.
    host_stamps = []
    t_train = timed(train_step, args.steps, keep=t_train_local, stamps=host_stamps, warmup=args.warmup)

    # per-step distribution (median and p10/p90): one event pair per step, a third pass of the same K steps
    def step_quantiles(fn, steps):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        torch.cuda.synchronize(device)
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize(device)
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        q = lambda f: round(ms[min(len(ms) - 1, int(f * len(ms)))], 4)
        return {"p10": q(0.10), "p50": q(0.50), "p90": q(0.90)}

    step_q = step_quantiles(train_step, args.steps)

    t_fwd = timed(fwd_step, args.steps, warmup=max(1, args.warmup // 2))
    fwd_q = step_quantiles(fwd_step, args.steps)

    # workload statistics of this rank's view(s) (needed for the algorithmic byte counts)
    view_stats = []
    with torch.no_grad():
        e = torch.Tensor([])
        for r_ in rasts:
            rs_ = r_.raster_settings
            R_, _c, radii, gb, bb, ib = _C.rasterize_gaussians(
                rs_.bg, t["means3D"], t["colors_precomp"] if "colors_precomp" in t else e, t["opacities"], t["scales"], t["rotations"], 1.0, e,
                rs_.viewmatrix, rs_.projmatrix, rs_.tanfovx, rs_.tanfovy, rs_.kernel_size, rs_.subpixel_offset, H, W,
                t["shs"] if "shs" in t else e, deg, rs_.campos, False, False)
            view_stats.append({"R": int(R_), "V": int((radii > 0).sum().item()), "walked": int(_C.view_image(ib, H, W)["tile_last"].sum().item())})
        del _c, gb, bb, ib
    nv = len(view_stats)
    R, V, walked = view_stats[0]["R"], view_stats[0]["V"], view_stats[0]["walked"]   # the rank's first view (rank 0: the base camera)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)

    # who took part: an all-gather of (rank, device index, this rank's own ms/step, the instances of its views) over the job's collective backend
    my_ms = 1000.0 * t_train_local[0] / args.steps
    ranks_seen = VP.gather_over_ranks([float(rank), float(local_rank), my_ms, float(sum(v["R"] for v in view_stats)), float(nv)], device)

    job = VP.job_fields(world, args.steps, t_train, ranks_seen, args.baseline_iters_per_s, views_total=V_total)
    iters_per_s = job["value"]
    pl = f"{P // 1_000_000}M" if P % 1_000_000 == 0 and P >= 1_000_000 else (f"{P // 1000}k" if P % 1000 == 0 else str(P))
    size_label = f"{pl} Gaussians @{'1080p' if (W, H) == (1920, 1080) else '4K' if (W, H) == (3840, 2160) else f'{W}x{H}'}" + \
                 ("" if args.scale_mult == 1.0 else f", scales x{args.scale_mult:g}")
    fwd_fps = V_total * args.steps / t_fwd
    out = {
        "metric": (f"forward_fps (forward only, {size_label})" if args.forward_only else
                   f"train_iters_per_s (fwd+bwd of the rasterizer, {size_label})"),
        "value": round(iters_per_s, 3),
        "unit": "iter/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1000.0 * t_train / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak" if args.views is None else "strong",
        "collective_backend": VP.backend_name(),
        "loss_allreduced_last_step": None if args.forward_only else round(loss_stream.last(), 9),
        "loss_note": "loss = <image, cotangent> on the rasterizer's stream; its 4-byte all-reduce is queued asynchronously (wg_viewparallel.LossStream) "
                     "and covered by the device-wide synchronize that ends the timed region",
        "pass_order": ("per-stage event pass (K steps)" if not args.no_profile else "") + " -> W warm-up steps -> timed region (K steps) -> per-step quantile "
                      "pass -> forward-only passes: the GPU reaches its steady clocks before the timed region, see bench.py",
        "host_note": "the interpreter's cycle collector runs once, in front of the first pass, and is off until the last timed pass is through (a "
                     "collection is a ~57 ms GPU pause in this process, and an idle MI355X needs 20-30 ms of load to regain its clocks: "
                     "scripts/diag_idle_ramp.py); every step's work is unchanged",
        "rccl_ranks_seen": job["rccl_ranks_seen"],
        "per_rank_ms_per_step": job["per_rank_ms_per_step"],
        "per_rank_device": job["per_rank_device"],
        "vs_baseline": None,
        **({"scaling_efficiency": job["scaling_efficiency"]} if "scaling_efficiency" in job else {}),
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{P} Gaussians{'' if args.scale_mult == 1.0 else f' (scales x{args.scale_mult:g})'}, {W}x{H}, "
                               f"{'SH deg 3' if args.colors == 'sh' else 'precomputed colours'}, "
                               f"{'forward only' if args.forward_only else 'fwd+bwd'}, " +
                               ("one view per GPU (view-parallel, loss all-reduce only)" if args.views is None else
                                f"a fixed batch of {V_total} views per step dealt round-robin to {world} GPU(s) (view-parallel, loss all-reduce only; BASELINE config 4 "
                                "is --views 8)"),
                   "gaussians": P, "width": W, "height": H, "colors": args.colors, "views_per_step": V_total, "views_of_rank0": my_views,
                   "streams_per_rank": n_streams},
        "per_rank_num_rendered": job["per_rank_num_rendered"],
        "forward_fps": round(fwd_fps, 2),
        "forward_mpix_per_s": round(fwd_fps * N / 1e6, 1),
        "forward_ms": round(1000.0 * t_fwd / args.steps, 4),
        "step_ms_quantiles": step_q,
        "timed_region_host_ms": host_step_summary(host_stamps),
        "forward_ms_quantiles": fwd_q,
        "workload_stats": {"P": P, "V": V, "R": int(R), "N": N, "tiles": tiles, "instances_walked": walked,
                           **({"views_of_this_rank": view_stats} if nv > 1 else {})},
        "library": {"version": _C.version(), "path": os.path.relpath(_C._LIB_PATH, ROOT), "options": args.option,
                    **profile_stamps(), "geometry_reuse": _C.get_option("geometry_reuse"),
                    # speculative forward (rasterizer_impl.cu:284's rendezvous moved behind the call's last launch): how it fared in this process
                    "near_far_split": {k: _C.get_option(k) for k in ("near_adapt", "near_per_tile", "near_per_tile_now", "near_far_tiles_last")},
                    "forward_order": _C.get_option("forward_order"),
                    "speculative_forward": {k: _C.get_option(k) for k in ("speculative_forward", "spec_frames", "spec_misses", "forward_polls",
                                                                          "forward_polls_waited", "forward_wait_us_total")}},
    }

    if stages:
        # time of a stage per STEP: a stage may consist of several event-bracketed scopes per call (scan = threshold + count +
        # tile scan; render_backward = record clear + kernel; duplicate_keys = near + far scatter), so total / steps, not / scopes
        per_stage = {k: ms / args.steps / nv for k, (ms, n) in stages.items()}   # per view (= per launch of the stage's kernels)
        out["stages_ms"] = {k: round(v, 4) for k, v in per_stage.items()}
        out["stages_note"] = ("HIP-event pairs around each stage over a second timed pass of the same K steps "
                              f"({round(1000.0 * t_train_profiled / args.steps, 4)} ms/step with the events in)")
        dom = max(per_stage, key=lambda k: per_stage[k])
        is_sh = args.colors == "sh"
        # (a rank with several views: the byte counts are those of its MEAN view, over the stages' mean launch times)
        kw = dict(P=P, V=sum(v["V"] for v in view_stats) // nv, R=sum(v["R"] for v in view_stats) // nv, N=N, tiles=tiles, M=M, sh=is_sh)
        walked = sum(v["walked"] for v in view_stats) // nv
        pmc, pmc_note = load_pmc(f"{P} Gaussians, {W}x{H}, {args.colors}" + ("" if args.scale_mult == 1.0 else f", scales x{args.scale_mult:g}"))

        # Three byte counts per stage, each divided by the stage's HIP-event time of THIS run:
        #   hbm_traffic     PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950 correction), when stamped to these sources
        #   design          what this design's kernel must move at least once (walked instances, 48-B records): <= peak by construction
        #   survey_8d       SURVEY 8(d)'s formula of the REFERENCE scheme (R-based, 6 radix passes...): an equivalent rate, which
        #                   exceeds the peak wherever this design avoids the reference's traffic -- reported, never called a fraction
        def row_of(k):
            ms = per_stage[k]
            t = ms * 1e-3
            row = {"ms": round(ms, 4), "design_GBps": round(design_bytes(k, walked=walked, **kw) / t / 1e9, 1),
                   "reference_scheme_equiv_GBps": round(algorithmic_bytes(k, **kw) / t / 1e9, 1)}
            row["frac_of_peak_by_design_bytes"] = round(row["design_GBps"] / HBM_PEAK_GBS, 4)
            if k in pmc:
                row["hbm_traffic_GBps"] = round(pmc[k]["hbm_bytes"] / t / 1e9, 1)
                row["frac_of_peak_by_traffic"] = round(row["hbm_traffic_GBps"] / HBM_PEAK_GBS, 4)
                if pmc[k].get("SQ_INSTS_VALU"):
                    row["valu_G_wave_instr_per_s"] = round(pmc[k]["SQ_INSTS_VALU"] / t / 1e9, 1)   # a rate (see roofline.compute.valu.note), not a fraction
            return row
        rows = {k: row_of(k) for k, ms in per_stage.items() if ms > 0 and k != "render_fixup"}
        out["stage_rooflines"] = rows
        out["stage_rooflines_note"] = pmc_note

        out["roofline"] = governing_roofline(dom, rows[dom], pmc.get(dom), per_stage[dom], algorithmic_bytes(dom, **kw),
                                             design_bytes(dom, walked=walked, **kw))
        workload_key = f"{P} Gaussians, {W}x{H}, {args.colors}" + ("" if args.scale_mult == 1.0 else f", scales x{args.scale_mult:g}")
        roof_ctx = dict(dom=dom, per_stage=per_stage, pmc=pmc, pairs=load_pair_counts(workload_key))
        # whole forward / backward pipelines against the same roofline, for context
        fwd_names = ["preprocess", "scan", "duplicate_keys", "sort", "tile_ranges", "render_forward"]
        bwd_names = ["render_backward", "preprocess_backward"]
        tf = sum(per_stage[k] for k in fwd_names)
        tb = sum(per_stage[k] for k in bwd_names)
        pipe = {"forward_kernel_ms": round(tf, 4), "backward_kernel_ms": round(tb, 4)}
        for nm, names, tt in (("forward", fwd_names, tf), ("backward", bwd_names, tb)):
            if tt > 0:
                pipe[nm + "_design_GBps"] = round(sum(design_bytes(k, walked=walked, **kw) for k in names) / (tt * 1e-3) / 1e9, 1)
                pipe[nm + "_reference_scheme_equiv_GBps"] = round(sum(algorithmic_bytes(k, **kw) for k in names) / (tt * 1e-3) / 1e9, 1)
                if all(k in pmc for k in names if per_stage[k] > 0 and k != "tile_ranges"):
                    pipe[nm + "_hbm_traffic_GBps"] = round(sum(pmc[k]["hbm_bytes"] for k in names if k in pmc) / (tt * 1e-3) / 1e9, 1)
        out["pipeline_roofline"] = pipe

    if world == 1 and args.views is None and not args.forward_only and not args.no_camera_sequence:
        # BASELINE config 4's eight cameras (the base camera yawed by 0..35 degrees: R falls from 7.4 M to 3.5 M) cycled on ONE GPU:
        # what the speculative forward costs when consecutive frames differ.  The headline loop above renders one frame over and over,
        # where a prediction can never miss; here the thread's frame history starts empty, the first cycle has to learn the sizes
        # (a frame with more instances than predicted + margin re-issues its tail: a "miss"), later cycles run on the learnt maximum.
        seq_rasts = [GaussianRasterizer(make_settings(c, deg, device=device)) for c in VP.view_cameras(8, W, H)]

        def seq_step(i):
            for v in t.values():
                v.grad = None
            means2D.grad = None
            color = seq_rasts[i % 8](means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t.get("shs"),
                                 colors_precomp=t.get("colors_precomp"), scales=t["scales"], rotations=t["rotations"])[0]
            color.backward(cot)
        spec_mode = _C.get_option("speculative_forward")
        _C.set_option("speculative_forward", spec_mode)   # clears this thread's frame history and counters
        counter = [0]

        def seq_fn():
            seq_step(counter[0])
            counter[0] += 1
        t_cold = timed(seq_fn, 8)                          # first cycle: nothing learnt yet
        cold = {k: _C.get_option(k) for k in ("spec_frames", "spec_misses")}
        t_seq = timed(seq_fn, args.steps)
        warm = {k: _C.get_option(k) for k in ("spec_frames", "spec_misses")}
        _C.set_option("speculative_forward", 0)
        counter[0] = 0
        timed(seq_fn, 8)
        t_seq_classic = timed(seq_fn, args.steps)
        _C.set_option("speculative_forward", spec_mode)
        out["camera_sequence"] = {
            "what": "the eight config-4 cameras cycled (fwd+bwd, one frame per step) on this GPU; frame history cleared first",
            "first_cycle": {"steps": 8, "ms_per_step": round(1e3 * t_cold / 8, 4), **cold},
            "config4_N1_point": "`steady` IS BASELINE config 4's batch (the eight 1080p views over the 1 M cloud) on ONE GPU -- the N = 1 point of its 1/2/4/8 "
                                "curve, what `bench.py --gpus 1 --views 8` times as its headline value; `--gpus N --views 8` are the other points",
            "steady": {"steps": args.steps, "ms_per_step": round(1e3 * t_seq / args.steps, 4), "iters_per_s": round(args.steps / t_seq, 2),
                       "spec_frames": warm["spec_frames"] - cold["spec_frames"], "spec_misses": warm["spec_misses"] - cold["spec_misses"]},
            "steady_without_speculation": {"ms_per_step": round(1e3 * t_seq_classic / args.steps, 4)},
        }
        # ... and what ONE miss costs: the history taught on the smallest of the eight frames (camera 7, R 3.46 M), then the largest
        # (camera 0, R 7.44 M: more than the 25 % margin above it) -- the guarded kernels return at once and the host re-issues the tail
        # with the real sizes -- against the same frame once the history has learnt it.  Single steps between device synchronises.
        def one_step_ms(i):
            torch.cuda.synchronize(device)
            t0_ = time.perf_counter()
            seq_step(i)
            torch.cuda.synchronize(device)
            return 1e3 * (time.perf_counter() - t0_)
        miss_ms, hit_ms, misses = [], [], 0
        for _ in range(5):
            _C.set_option("speculative_forward", spec_mode)
            for _k in range(3):
                seq_step(7)
            m0 = _C.get_option("spec_misses")
            miss_ms.append(one_step_ms(0))
            misses += _C.get_option("spec_misses") - m0
            for _k in range(3):
                seq_step(0)
            hit_ms.append(one_step_ms(0))
        out["camera_sequence"]["one_miss"] = {
            "what": "frame history taught on camera 7 (smallest R), then camera 0 (largest): a frame that does not fit its predicted buffer; "
                    "single synchronised steps, median of 5", "misses_seen": misses,
            "step_with_miss_ms": round(sorted(miss_ms)[2], 4), "same_frame_predicted_ms": round(sorted(hit_ms)[2], 4)}
        with torch.no_grad():
            Rs = []
            for r8 in seq_rasts:
                rs8 = r8.raster_settings
                Rs.append(int(_C.rasterize_gaussians(rs8.bg, t["means3D"], t["colors_precomp"] if "colors_precomp" in t else e, t["opacities"], t["scales"],
                                                     t["rotations"], 1.0, e, rs8.viewmatrix, rs8.projmatrix, rs8.tanfovx, rs8.tanfovy, rs8.kernel_size,
                                                     rs8.subpixel_offset, H, W, t["shs"] if "shs" in t else e, deg, rs8.campos, False, False)[0]))
        out["camera_sequence"]["num_rendered_per_camera"] = Rs

    if side_streams and not args.forward_only:
        # Every view dealt to the streams must come out as it does alone on one stream: image, radii and accumulation bit for bit, and --
        # in the deterministic backward mode, whose sums do not depend on timing -- every gradient bit for bit.
        import hashlib

        def digest_views(streams_on):
            out_ = []
            with _C.call_options(deterministic_backward=1):
                for k, r in enumerate(rasts):
                    ctx_ = torch.cuda.stream(side_streams[k % n_streams]) if streams_on else contextlib.nullcontext()
                    with ctx_:
                        for v in t.values():
                            v.grad = None
                        means2D.grad = None
                        color, radii, acc = call(r)
                        color.backward(cot)
                        keep_ = [color.detach(), radii, acc.detach(), means2D.grad] + [v.grad for v in t.values()]
                    out_.append(keep_)
                    if not streams_on:
                        torch.cuda.synchronize(device)
            torch.cuda.synchronize(device)
            return [[hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest()[:16] for x in keep_] for keep_ in out_]
        import contextlib
        alone, dealt = digest_views(False), digest_views(True)
        out["streams"] = {"streams": n_streams, "views": len(rasts),
                          "every_view_bit_identical_to_its_run_alone": alone == dealt,
                          "compared": "image, radii, accumulation, and (deterministic backward mode) all gradients: SHA-256 per tensor and view",
                          "mismatching_views": [k for k in range(len(rasts)) if alone[k] != dealt[k]]}

    gc.enable()   # (off since the first pass: see above)
    # The same K steps once more as a training loop would run them: the interpreter's cycle collector ON inside the region (ADVICE r4:
    # round 3's protocol beside round 4's, so that a change of protocol and a change of the code can be told apart)
    if not args.forward_only:
        import time as _t
        torch.cuda.synchronize(device)
        VP.barrier()
        t0_ = _t.perf_counter()
        for _ in range(args.steps):
            train_step()
        torch.cuda.synchronize(device)
        t_gc = VP.max_over_ranks(_t.perf_counter() - t0_, device)
        out["region_with_cycle_collector_on"] = {"steps": args.steps, "ms_per_step": round(1e3 * t_gc / args.steps, 4),
                                                 "iters_per_s": round(V_total * args.steps / t_gc, 2),
                                                 "what": "K more full steps between barrier + synchronize pairs with gc enabled (rounds 1-3's protocol); "
                                                         "`value` is the region above, collector off"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.forward_only:
        # Forward-only workloads (BASELINE config 5: 10 M Gaussians @ 4K): the same two checkers, forward legs only.
        # (1) the CPU oracle, on a BOUNDED sample -- the first min(P, 2 M) Gaussians of the same cloud, same camera and frame (its
        #     single-threaded key sort of the full 211 M instances would take minutes) -- timed as cpu_baseline and used to check
        #     the product on that very sample;
        from oracle import oracle
        oracle.build()
        cores = os.cpu_count() or 1
        Ps = min(P, 2_000_000)
        sub = {k: np.ascontiguousarray(v[:Ps]) for k, v in cloud.items()}
        t0 = time.perf_counter()
        o = oracle.run_scene(sub, cam, sh_degree=deg)
        t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(1.0 / t_cpu, 4), "unit": "iter/s", "cores": cores, "kind": "port",
                               "sample": (f"1 forward pass over the first {Ps} of the {P} Gaussians, same camera and {W}x{H} frame "
                                          "(OpenMP over Gaussians/tiles; the (tile|depth) key sort is single-threaded)"),
                               "seconds": round(t_cpu, 2)}
        from tests.wg_testlib import run_hip_native
        hs = run_hip_native(sub, cam, sh_degree=deg, device=device)
        cf = compare_forward(hs["color"].cpu().numpy(), o)
        out["parity"] = {"checker": f"CPU oracle on the cpu_baseline sample ({Ps} Gaussians, {W}x{H}, forward)",
                         "fwd_max_abs_err_solid_pixels": float(f"{cf['max_err_solid']:.3e}"),
                         "fwd_fragile_pixels": cf["n_fragile"], "fwd_fragile_over_1e-4": cf["n_over_in_fragile"],
                         "radii_equal": bool((hs["radii"].cpu().numpy() == o["radii"]).all()),
                         "num_rendered_equal": bool(int(hs["num_rendered"]) == o["num_rendered"])}
        del hs, o, sub
        torch.cuda.empty_cache()
        # (2) the reference's own kernels (oracle/_ref, its CUDA sources built for gfx950 without fp contraction: the arithmetic the
        #     sources spell) on the FULL workload, beside the product: timing + full-size comparison of radii and image.
        ref_lib = os.path.join(ROOT, "oracle", "_ref", "libref_hip_rasterizer_nofma.so")
        if os.path.exists(ref_lib):
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_hip_bench.py"), "--gaussians", str(P), "--width", str(W),
                                    "--height", str(H), "--colors", args.colors, "--scale-mult", str(args.scale_mult), "--forward-only",
                                    "--variant", "nofma", "--steps", "5", "--warmup", "2"], capture_output=True, text=True, timeout=420)
                ref = json.loads(r.stdout.strip().splitlines()[-1])
                out["reference_on_this_gpu"] = ref
                out["speedup_vs_reference_on_this_gpu"] = {"forward": round(fwd_fps / ref["forward_fps"], 2)}
                pv = ref.get("product_vs_reference", {})
                out["parity"]["full_size_vs_reference_build"] = {
                    "radii_mismatch": pv.get("radii_mismatch"), "num_rendered_equal": bool(int(R) == int(ref["num_rendered"])),
                    "pixels_over_1e-4": pv.get("pixels_over_1e-4"), "pixels": pv.get("pixels"), "color_max_abs": pv.get("color_max_abs"),
                    "color_p9999_abs": pv.get("color_p9999_abs")}
            except Exception as ex:  # noqa: BLE001 -- a reported extra, never a reason to lose the bench line
                out["reference_on_this_gpu"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.forward_only:
        # CPU leg: the oracle (a CPU port of the reference's algorithm) on the same workload, timed on the host
        # cores, and used as the checker for the parity part of the metric.  Never on the measured path.
        from oracle import oracle
        oracle.build()
        cores = os.cpu_count() or 1
        t0 = time.perf_counter()
        o = oracle.run_scene(cloud, cam, sh_degree=deg, cotangent=cot_np)
        t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(1.0 / t_cpu, 4), "unit": "iter/s", "cores": cores, "kind": "port",
                               "sample": "1 full fwd+bwd step of the same workload (OpenMP over Gaussians/tiles)",
                               "seconds": round(t_cpu, 2)}
        if nv == 1:
            ref_pairs = {"pairs_evaluated": int(o["ctx"].get("n_evaluated").astype(np.int64).sum()),
                         "pairs_blended": int(o["ctx"].get("n_blended").astype(np.int64).sum()), "source": "the CPU oracle's walk of this very frame (cpu_baseline leg)"}
        train_step()
        torch.cuda.synchronize(device)
        grads = {"means3D": t["means3D"].grad, "means2D": means2D.grad, "opacities": t["opacities"].grad,
                 "scales": t["scales"].grad, "rotations": t["rotations"].grad}
        if "shs" in t:
            grads["sh"] = t["shs"].grad
        else:
            grads["colors_precomp"] = t["colors_precomp"].grad
        errs = compare_grads({k: v.detach().cpu().numpy() for k, v in grads.items()}, o["grads"])
        cf = compare_forward(fwd_step().cpu().numpy(), o)
        out["parity"] = {"grad_max_rel_err": {k: float(f"{v:.3e}") for k, v in errs.items()},
                         "grad_max_rel_err_worst": float(f"{max(errs.values()):.3e}"),
                         "fwd_max_abs_err_solid_pixels": float(f"{cf['max_err_solid']:.3e}"),
                         "fwd_fragile_pixels": cf["n_fragile"], "fwd_fragile_over_1e-4": cf["n_over_in_fragile"],
                         "radii_equal": bool((radii.cpu().numpy() == o["radii"]).all()),
                         "num_rendered_equal": bool(int(R) == o["num_rendered"])}
        out["speedup_vs_cpu_baseline"] = round(iters_per_s / (1.0 / t_cpu), 1)

        # Same leg, second baseline: the reference's OWN kernels on this GPU -- its CUDA sources compiled for gfx950 with hipcc
        # (oracle/ref_hip -> oracle/_ref/*.so, prebuilt; test infrastructure) -- timed on the same workload and compared with the
        # product's outputs at full size.  In a subprocess with a timeout: whatever happens there cannot touch the numbers above.
        ref_lib = os.path.join(ROOT, "oracle", "_ref", "libref_hip_rasterizer.so")
        if os.path.exists(ref_lib):
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_hip_bench.py"), "--gaussians", str(P), "--width", str(W),
                                    "--height", str(H), "--colors", args.colors, "--scale-mult", str(args.scale_mult), "--parity-variant", "nofma"],
                                   capture_output=True, text=True, timeout=240)
                ref = json.loads(r.stdout.strip().splitlines()[-1])
                out["reference_on_this_gpu"] = ref
                out["speedup_vs_reference_on_this_gpu"] = {"train": round(iters_per_s / ref["train_iters_per_s"], 2),
                                                           "forward": round(fwd_fps / ref["forward_fps"], 2)}
            except Exception as ex:  # noqa: BLE001 -- a reported extra, never a reason to lose the bench line
                out["reference_on_this_gpu"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    if roof_ctx is not None:
        # SURVEY 8(d)'s secondary (compute) view of the two render kernels, from counted pairs (see compute_view)
        pc = roof_ctx["pairs"]
        if ref_pairs is None and pc and pc.get("reference_walk") and pc["reference_walk"].get("gaussians") == P:
            ref_pairs = dict(pc["reference_walk"], source=f"profiles/{pc['file']} (CPU oracle's walk, recorded with the counting build)")
        views = {k: compute_view(k, roof_ctx["per_stage"][k], roof_ctx["pmc"].get(k), pc, ref_pairs)
                 for k in ("render_forward", "render_backward") if roof_ctx["per_stage"].get(k, 0) > 0}
        if roof_ctx["dom"] in views:
            out["roofline"]["compute"] = views.pop(roof_ctx["dom"])
        if views:
            out["render_compute"] = views
    if want_legs:
        t_legs = time.perf_counter()
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import fullsize_frames as FF
        del cloud, t, means2D
        gc.collect()
        torch.cuda.empty_cache()
        out["configs"] = config_legs(device, deg, not args.no_cpu_baseline,
                                     {nm: (lambda nm=nm: FF.frame_inputs(nm)) for nm in ("config2_500k_1080p_sh3", "config3_3M_1600x1200_precomp", "config5_10M_4K_sh3")})
        out["configs"]["note"] = ("BASELINE.json configs 2, 3 (operator level: one call, fwd+bwd) and 5 at their quoted sizes, a ~1.2 s timed leg each after the headline's "
                                  "passes; pins_ok = the frame's radii / n_contrib / final_T hash to the reference build's (tests/golden/ref_hip_fullsize_sha256.json); "
                                  f"seconds for the three legs, their clouds' generation included: {time.perf_counter() - t_legs:.1f}")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
