#!/usr/bin/env python3
"""Generates tests/golden/ref_hip_fullsize_sha256.json: the reference's own kernels (oracle/_ref, its CUDA sources built for gfx950 with
-ffp-contract=off by oracle/ref_hip/Makefile) run on every frame of tests/golden/fullsize_frames.py on an MI355X; per frame
num_rendered and the SHA-256 of radii, n_contrib, final_T.  Run on a GPU box (the output goes to gpurun_out/ as well, which is what comes
back from it):   python tests/golden/make_fullsize_ref_hashes.py [frame ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fullsize_frames as F  # noqa: E402


def main():
    import torch
    from oracle.ref_hip import ref_hip
    assert torch.cuda.is_available() and ref_hip.available("nofma"), "needs a HIP device and oracle/_ref (nofma)"
    names = sys.argv[1:] or list(F.FRAMES)
    out = {"generator": "tests/golden/make_fullsize_ref_hashes.py", "reference_build": "oracle/_ref/libref_hip_rasterizer_nofma.so (hipcc, -ffp-contract=off)",
           "device": torch.cuda.get_device_name(0), "date": time.strftime("%Y-%m-%d"), "frames": {}}
    if os.path.exists(F.PINS) and sys.argv[1:]:
        out["frames"] = json.load(open(F.PINS))["frames"]
    for name in names:
        cloud, cam, deg = F.frame_inputs(name)
        s = ref_hip.Session(cloud, cam, deg, variant="nofma")
        R = s.forward()
        torch.cuda.synchronize()
        out["frames"][name] = F.digest(R, s.radii.cpu().numpy(), s.n_contrib.cpu().numpy(), s.final_T.cpu().numpy())
        print(name, out["frames"][name], flush=True)
        del s
        torch.cuda.empty_cache()
    for path in (F.PINS, os.path.join(F.ROOT, "gpurun_out", "ref_hip_fullsize_sha256.json")):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
