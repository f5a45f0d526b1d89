#!/usr/bin/env python
"""Per-stage timing of the multi-view -> UV back-projection (HIP events), at the SURVEY 8d sizes.
usage: python tools/bench_backproject.py [--faces 50000 200000] [--view 1024] [--atlas 2048]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.texturetools.benchmarks import time_backprojection  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--faces", type=int, nargs="+", default=[50000, 200000])
ap.add_argument("--view", type=int, default=1024)
ap.add_argument("--atlas", type=int, default=2048)
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
for nf in a.faces:
    r = time_backprojection(nf, a.view, a.atlas, iters=a.iters)
    print(json.dumps(r))
    print("faces %d  atlas %d^2  views 6x%d^2 : total %.2f ms" % (r["faces"], a.atlas, a.view, r["total_ms"]))
    for k, v in r["stages_ms"].items():
        print("   %-18s %8.3f ms  %8.1f GB/s (algorithmic)" % (k, v, r["stages_gbps"].get(k, float("nan"))))
