#!/bin/bash
# A/B builds of kernel variants (wild-gaussians_amd/build.py: WG_BUILD_VARIANT / WG_FILE_FLAGS), built HERE (hipcc cross-compiles),
# benchmarked on the GPU box by scripts/ab_run.sh.   usage: scripts/ab_variants.sh name1 "file.hip:-DFLAG=1 ..." [name2 "..."] ...
# (rejected render_bwd.hip variants live in experiments/r3_render_bwd_variants.patch: git apply it first)
set -e
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  WG_BUILD_VARIANT="$1" WG_FILE_FLAGS="$2" python wild-gaussians_amd/build.py > /dev/null
  echo "built wild-gaussians_amd/build/$1/libwg_rasterizer.so  ($2)"
  shift 2
done
