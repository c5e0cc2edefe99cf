#!/usr/bin/env python3
"""Stage the reference's REAL caller -- wildgaussians/method.py and the three modules + one YAML it imports -- byte for byte
into tests/real_caller/_staged/ so that the GPU box (which has no /root/reference) can run `WildGaussians.train_iteration`
on this repo's `diff_gaussian_rasterization` / `simple_knn` (BASELINE config 3).

This is the Python counterpart of oracle/ref_hip/Makefile -> oracle/_ref/: the files are taken from where they lie under
/root/reference by this committed recipe, land in a git-ignored directory (never in history), and travel with the
gpurun snapshot.  TEST INFRASTRUCTURE ONLY: nothing under wild-gaussians_amd/ or bench.py's timed region imports them.
`manifest.json` (committed) holds the sha256 of every staged file; tests verify the staged copy against it, so "unchanged"
is checked, not assumed.  Run in the build container: python tests/real_caller/stage_reference_caller.py
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/wildgaussians"
DST = os.path.join(HERE, "_staged", "wildgaussians")
FILES = ["method.py", "config.py", "types.py", "dinov2.py", "configs/default.yml"]
MANIFEST = os.path.join(HERE, "manifest.json")


def sha256(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def staged_ok():
    """True when every staged file exists and has the committed hash."""
    if not os.path.exists(MANIFEST):
        return False
    man = json.load(open(MANIFEST))
    return all(os.path.exists(os.path.join(DST, f)) and sha256(os.path.join(DST, f)) == h for f, h in man["sha256"].items())


def stage(write_manifest=False):
    if not os.path.isdir(REF):
        return staged_ok()
    man = {}
    for f in FILES:
        d = os.path.join(DST, f)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        shutil.copyfile(os.path.join(REF, f), d)
        man[f] = sha256(d)
    if write_manifest or not os.path.exists(MANIFEST):
        json.dump({"source": REF, "sha256": man}, open(MANIFEST, "w"), indent=1)
    return staged_ok()


if __name__ == "__main__":
    ok = stage(write_manifest="--write-manifest" in sys.argv)
    print("staged:", ok, DST)
    sys.exit(0 if ok else 1)
