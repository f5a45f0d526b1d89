#!/usr/bin/env python
"""Back-projection stage chain, same process, interleaved: wave-wide packet walk over 8 x 8 texel tiles (round 4 default) vs one thread per ray over the
packed tree (round 2/3 default) vs the reference's stack walk."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd import _lib
from unitex_amd.texturetools.benchmarks import time_backprojection
for faces in (50000, 200000):
    for name, packet, stack in (("stack ", 0, 1), ("packed", 0, 0), ("packet", 1, 0)) * 2:
        _lib.set_option("UTX_BVH_PACKET", packet); _lib.set_option("UTX_BVH_STACK_WALK", stack)
        bp = time_backprojection(faces, 1024, 2048, iters=3, warmup=1)
        print("faces %6d %s | total %.2f ms | backproject %.2f ms | nn_fill %.2f ms | bvh_build %.2f ms | nodes/ray %s depth %s" % (
            bp["faces"], name, bp["total_ms"], bp["stages_ms"].get("backproject", -1), bp["stages_ms"].get("nn_fill", -1), bp["stages_ms"].get("bvh_build", -1),
            bp.get("nodes_per_ray"), bp.get("bvh_depth")), flush=True)
_lib.set_option("UTX_BVH_PACKET", 1); _lib.set_option("UTX_BVH_STACK_WALK", 0)
