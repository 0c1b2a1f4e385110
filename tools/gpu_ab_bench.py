#!/usr/bin/env python3
"""bench.py with the product library swapped for a variant built by tools/ab_build.sh (same-box A/B; dev aid).
usage: python tools/gpu_ab_bench.py <lib.so | -> [bench.py arguments ...]"""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
lib = sys.argv[1]
sys.argv = [os.path.join(REPO, "bench.py")] + sys.argv[2:]
if lib != "-":
    from gym_pomdp_amd import _native
    _native.LIB_PATH = os.path.abspath(lib)
runpy.run_path(sys.argv[0], run_name="__main__")
