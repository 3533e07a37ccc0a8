#!/usr/bin/env python3
"""A/B of the geometry floors of fused.stage_plan: does deconv3 / deconv4 gain from the box-sum backward / the sub-pixel forward in either mode?
usage: ab_stage_floors.py <box floor | -> <sub floor | -> [bench args...]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import stereospike_amd.fused as f
if sys.argv[1] != '-':
    f._BOX_MIN_SRC_PIXELS = int(sys.argv[1])
if sys.argv[2] != '-':
    f._SUB_MIN_SRC_PIXELS = int(sys.argv[2])
sys.argv = [os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline'] + sys.argv[3:]
runpy.run_path(sys.argv[0], run_name='__main__')
