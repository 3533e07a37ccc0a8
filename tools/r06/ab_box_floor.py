"""Round 6 A/B: the box-sum backward's geometry floor (fused._BOX_MIN_SRC_PIXELS: wide stages take the box kernels from this many source pixels per frame) in the
16-bit modes, whose box kernels changed this round (two chunks per window).  Runs bench.py with the floor patched.

    python tools/r06/ab_box_floor.py FLOOR [bench.py arguments ...]
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import stereospike_amd.fused as fused     # noqa: E402

fused._BOX_MIN_SRC_PIXELS = int(sys.argv[1])
print(f'[ab_box_floor] _BOX_MIN_SRC_PIXELS = {fused._BOX_MIN_SRC_PIXELS}', file=sys.stderr)
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
