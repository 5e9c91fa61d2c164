#!/usr/bin/env python3
"""Sweep time against the number of pivots applied per sweep (is the sweep short of VALU or of
bandwidth?): python tools/r2_kscan.py"""
import json
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import lpr_381_group_v22_amd as pkg  # noqa: E402
from r2_probe import run  # noqa: E402

eng = pkg.Engine(0)
for variant in (0x3008, 0x4008):
    for block in (2, 4, 8, 12, 16):
        r = run(eng, 4096, 8192, variant, block=block, warm=4, steps=24,
                label=f"{variant:#x} block {block}")
        print(json.dumps(r), flush=True)
eng.close()
