#!/usr/bin/env python3
"""Turn a rocprofv3 PMC summary (scripts/profile_gpu.sh -> pmc_summary.txt) into per-stage HBM traffic per launch.

FETCH_SIZE / WRITE_SIZE are in KiB.  Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled before comparing with a
byte count.  CALIBRATED in round 6 on the render kernels' own access patterns (scripts/pmc_calibration.hip ->
profiles/r6/pmc_calibration.json; kernels of known bytes, arrays past the Infinity Cache):
  * streaming 16-B reads: known / FETCH_SIZE = 2.000; streaming 16-B writes: known / WRITE_SIZE = 1.000;
  * 48-byte records gathered through a shuffled index list (the render kernels' staging): 2 x FETCH_SIZE = 163.4 B per record
    = the 128-byte lines a 48-byte record touches (1.25 lines = 160 B) + its 4-byte index: the DOUBLING HOLDS for gathers -- it gives
    the bytes MOVED (whole lines: 3.1 x the 52 useful bytes when nothing is reused);
  * one global_atomic_add_f32 instruction of ten lanes on a 48-byte record (render_bwd.hip's gradient-record update):
    WRITE_SIZE = 64.0 B and FETCH_SIZE = 32.0 B per instruction, whatever the 40 bytes of payload -- the backward kernel's
    WRITE_SIZE is exactly 64 B x its reduced instances.
So hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 are bytes moved over the fabric for every stage, render kernels included.
usage: pmc_traffic.py pmc_summary.txt "<workload string>" > profiles/pmc_traffic.json"""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STAGE_OF = {"preprocess_kernel": "preprocess", "tile_count_kernel": "scan", "chunk_scan_kernel": "scan", "tile_scan_kernel": "scan",
            "tile_scatter_kernel": "duplicate_keys", "tile_scatter_staged_kernel": "duplicate_keys", "tile_sort_kernel": "sort",
            "tile_front_sort_kernel": "sort", "render_fixup_kernel": "render_fixup", "tile_order_kernel": "tile_ranges",
            "split_hist_kernel": "scan", "split_pick_kernel": "scan", "tile_scatter_far_kernel": "duplicate_keys", "render_forward_kernel": "render_forward",
            "render_backward_kernel": "render_backward", "preprocess_backward_kernel": "preprocess_backward"}
vals = {}
for line in open(sys.argv[1]):
    m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_ACTIVE_INST_VALU|GRBM_GUI_ACTIVE|SQ_LDS_BANK_CONFLICT|"
                 r"SQ_INSTS_VALU_TRANS_F32|SQ_THREAD_CYCLES_VALU|SQ_INSTS_VALU_ADD_F32|SQ_INSTS_VALU_MUL_F32|SQ_INSTS_VALU_FMA_F32|SQ_INST_CYCLES_VALU|"
                 r"SQ_INSTS_VALU_INT32)\s+dispatches=\s*\d+\s+per_dispatch=\s*(\d+)", line)
    if not m:
        continue
    name = re.sub(r"^void ", "", m.group(1)).split("(")[0].replace("wg::", "").split("<")[0]
    st = STAGE_OF.get(name)
    if st:
        key = m.group(2) + "_KiB" if m.group(2).endswith("_SIZE") else m.group(2)  # SQ_INSTS_*: wave-instructions per launch
        d = vals.setdefault(st, {"FETCH_SIZE_KiB": 0, "WRITE_SIZE_KiB": 0})
        d[key] = d.get(key, 0) + int(m.group(3))
try:
    import bench
    st = bench.profile_stamps()   # kernel sources + the device code of the library as built (bench.device_code_sha)
except Exception:  # noqa: BLE001
    st = {"kernel_source_sha": None, "device_code_sha": None}
out = {"workload": sys.argv[2] if len(sys.argv) > 2 else "", "kernel_source_sha": st["kernel_source_sha"], "device_code_sha": st.get("device_code_sha"), "collected": time.strftime("%Y-%m-%d"), "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: bytes MOVED (whole 128-B lines); factors measured on known byte counts in round 6 "
                     "(profiles/r6/pmc_calibration.json): streaming reads 2.000, streaming writes 1.000, 48-byte record gathers 2 x FETCH = lines touched "
                     "(163.4 B per record for 52 useful), atomic record updates 64 B of WRITE_SIZE + 32 B of FETCH_SIZE per instruction",
       "correction_per_stage": {"render_forward": "gather-calibrated: 2 x FETCH_SIZE = lines moved", "render_backward": "gather-calibrated reads; WRITE_SIZE = 64 B per atomic instruction",
                                "other stages": "streaming-calibrated (2.000 / 1.000)"},
       "stages": {k: dict(v, hbm_bytes=(2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024) for k, v in vals.items()}}
# Lane utilisation of the vector ALU from counters alone (rocprofiler's VALUUtilization): SQ_THREAD_CYCLES_VALU counts thread-cycles of VALU
# execution, SQ_ACTIVE_INST_VALU the (quad-)cycles waves spent executing VALU instructions:
#   thread_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)            (<= 1: the mean fraction of a wave's 64 lanes that were active)
# (the two come from different passes of the same command: a few percent of run-to-run spread are in it).  Round 4 also derived a "busy
# fraction" 4 * SQ_ACTIVE_INST_VALU / (SIMDs * GRBM_GUI_ACTIVE / 8), which read 1.1 - 1.4 for the render kernels: SQ_ACTIVE_INST_VALU is summed
# over WAVES, and the execution windows of different waves' instructions on one SIMD overlap (a transcendental or a DPP instruction is still in
# the pipe when the next wave's instruction issues), so that sum is not bounded by the SIMD's cycles.  It is kept as
# `valu_wave_cycles_per_simd_cycle` -- an occupancy-like figure, NOT a fraction of anything.
for st in out["stages"].values():
    if st.get("SQ_ACTIVE_INST_VALU") and st.get("GRBM_GUI_ACTIVE"):
        st["valu_wave_cycles_per_simd_cycle"] = round(4.0 * st["SQ_ACTIVE_INST_VALU"] / (1024.0 * st["GRBM_GUI_ACTIVE"] / 8.0), 3)
    if st.get("SQ_ACTIVE_INST_VALU") and st.get("SQ_THREAD_CYCLES_VALU"):
        st["thread_utilisation"] = round(st["SQ_THREAD_CYCLES_VALU"] / (64.0 * st["SQ_ACTIVE_INST_VALU"]), 4)
out["valu_note"] = ("thread_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU): mean active lanes per executed vector instruction; "
                    "valu_wave_cycles_per_simd_cycle = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs): summed over waves, may exceed 1")
print(json.dumps(out, indent=1))
