#!/usr/bin/env python
"""Back-projection stage chain: stackless packed LBVH walk (default) vs the reference's stack walk (UTX_BVH_STACK_WALK=1), same process."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.texturetools.benchmarks import time_backprojection
for faces in (50000, 200000):
    for walk in (1, 0, 1, 0):
        _lib.set_option("UTX_BVH_STACK_WALK", walk)
        bp = time_backprojection(faces, 1024, 2048, iters=3, warmup=1)
        print("faces %6d %s | total %.2f ms | backproject %.2f ms | bvh_build %.2f ms | nodes/ray %s depth %s" % (
            bp["faces"], "stack " if walk else "packed", bp["total_ms"], bp["stages_ms"].get("backproject", -1), bp["stages_ms"].get("bvh_build", -1),
            bp.get("nodes_per_ray"), bp.get("bvh_depth")), flush=True)
_lib.set_option("UTX_BVH_STACK_WALK", 0)
