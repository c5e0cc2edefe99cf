#!/usr/bin/env python3
"""Long sweeps of the call modes beyond the reference's surface against the REFERENCE'S OWN KERNELS (oracle/_ref, -ffp-contract=off): the
checkers of tests/ref_mode_checks.py -- the ones the driver-run suite uses in tests/test_reference_modes.py -- over any range of the
argument sweep's cases.  Replaces round 4's stress_sweep_round4_modes.py / stress_sweep_two_tone.py (same checks, now shared).
usage: python tests/tools/stress_sweep_modes.py [first] [count] [mode ...]      modes: two_colour two_tone one_tone raw (default: all)
-> one line per deviation and one summary line per mode"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_mode_checks as RC  # noqa: E402
from oracle.ref_hip import ref_hip  # noqa: E402

first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 500
modes = sys.argv[3:] or ["two_colour", "two_tone", "one_tone", "raw"]
for mode in modes:
    st = dict(cases=0, deviations=0, worst_img=0.0, worst_grad=0.0, worst_grad_name="")
    for i in range(first, first + count):
        pmod = i % 4
        if mode == "two_colour":
            rep = RC.check_two_colour(RC.mode_case(i, "precomp", pmod=pmod), ref_hip, seed=i)
        elif mode == "two_tone":
            rep = RC.check_two_tone(RC.mode_case(i, "sh", pmod=pmod, sh_degree=(i // 4) % 4), ref_hip, seed=i, second_plain=i % 4 == 1)
        elif mode == "one_tone":
            rep = RC.check_one_tone(RC.mode_case(i, "sh", pmod=pmod, sh_degree=(i // 4) % 4), ref_hip, seed=i)
        elif i % 6 == 5:
            continue   # raw-parameter mode needs a scale / rotation pair
        else:
            rep = RC.check_raw(RC.mode_case(i, "precomp", pmod=pmod, geometry="pair"), ref_hip, seed=i)
        st["cases"] += 1
        st["worst_img"] = max(st["worst_img"], *(v for k, v in rep.items() if k.endswith("_max")))
        w = rep["worst_grad"]
        if rep["grads"][w] > st["worst_grad"]:
            st["worst_grad"], st["worst_grad_name"] = rep["grads"][w], w
        if not rep["ok"]:
            st["deviations"] += 1
            print(mode, "deviation: case", i, {k: v for k, v in rep.items() if k != "grads"}, w, rep["grads"][w], flush=True)
    print(f"{mode}: cases {first}..{first + count - 1}:", st, flush=True)
