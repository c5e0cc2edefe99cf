#!/usr/bin/env python3
"""Print bench.device_code_sha() of built libraries (default: the product library): SHA-256 over the .text and .rodata of every gfx950 code
object in the library's fat binary -- the stamp profiles/pmc_traffic*.json and profiles/pair_counts*.json are matched on beside
kernel_source_sha.  Independent of the build directory; unchanged by edits of host code or comments.
usage: scripts/device_code_sha.py [libwg_rasterizer.so ...]
To check a committed stamp: `git archive <commit> wild-gaussians_amd include | tar -x -C /tmp/x && python /tmp/x/wild-gaussians_amd/build.py`,
then this script on /tmp/x/wild-gaussians_amd/diff_gaussian_rasterization/libwg_rasterizer.so."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

for p in sys.argv[1:] or [bench.PRODUCT_LIB]:
    print(bench.device_code_sha(p), p)
