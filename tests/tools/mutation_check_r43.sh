#!/bin/bash
# Mutation check (VERDICT r4 "next" 1: "Done = the P mod 4 bug of R4.3, re-introduced on purpose, turns the suite red").
# Round 4's one real bug: the gradient-record clear of a two-colour backward stopped at the last whole float4 (commit ffd85c6 fixed
# launch_tile_order: `(clear_floats + 3) / 4`).  This script re-introduces it in a VARIANT build (build/mut_r43/, never the product
# library), and runs the reference-checked mode tests on it: they must FAIL.  Two steps, because the build needs the sources to be
# patched for a moment and the GPU box only receives the tree:
#   here (no GPU):  tests/tools/mutation_check_r43.sh build     -> wild-gaussians_amd/build/mut_r43/libwg_rasterizer.so
#   on the GPU box: tests/tools/mutation_check_r43.sh run       -> prints MUTATION_DETECTED (exit 0) or MUTATION_SURVIVED (exit 1)
set -u
cd "$(dirname "$0")/../.."
SRC=wild-gaussians_amd/csrc/binning.hip
LIB=wild-gaussians_amd/build/mut_r43/libwg_rasterizer.so
case "${1:-}" in
build)
    grep -q 'const size_t vec4 = clear ? (clear_floats + 3) / 4 : 0;' $SRC || { echo "anchor not found in $SRC"; exit 2; }
    cp $SRC /tmp/binning.hip.orig
    trap 'cp /tmp/binning.hip.orig '$SRC'; touch -r /tmp/binning.hip.orig '$SRC EXIT
    sed -i 's|const size_t vec4 = clear ? (clear_floats + 3) / 4 : 0;|const size_t vec4 = clear ? clear_floats / 4 : 0;|' $SRC
    WG_BUILD_VARIANT=mut_r43 python wild-gaussians_amd/build.py --force && echo "built $LIB"
    ;;
run)
    [ -f $LIB ] || { echo "no $LIB: run '$0 build' where hipcc and the sources are"; exit 2; }
    WG_RASTERIZER_LIB=$PWD/$LIB WG_BINDING=ctypes python -m pytest tests/test_reference_modes.py -q -m gpu -k "two_colour or two_tone" -p no:cacheprovider > gpurun_out/mutation_r43.log 2>&1
    rc=$?
    tail -5 gpurun_out/mutation_r43.log
    if [ $rc -ne 0 ] && grep -q "failed" gpurun_out/mutation_r43.log; then echo MUTATION_DETECTED; exit 0; fi
    echo MUTATION_SURVIVED; exit 1
    ;;
*) echo "usage: $0 build|run"; exit 2;;
esac
