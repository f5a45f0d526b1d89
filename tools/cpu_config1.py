#!/usr/bin/env python
"""BASELINE configs[0] (FLUX.1-dev 512^2 x 4 views, 4 denoise steps, CPU float32) run to COMPLETION with the oracle -- the one CPU number of
SURVEY 8d that is not extrapolated.  `bench.py --cpu-config1` runs the same function on the GPU box's host cores (~1 h of box time at the
0.36 TFLOP/s its 64 threads reach); this script runs it wherever it is started and writes a JSON record.
    python tools/cpu_config1.py <threads> <out.json>"""
import json, os, platform, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import cpu_config1, host_cores
threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
out = sys.argv[2] if len(sys.argv) > 2 else "cpu_config1.json"
t0 = time.time()
rec = cpu_config1(threads)
rec.update(host=host_cores(), machine=platform.processor() or platform.machine(), wall_seconds=time.time() - t0, torch=torch.__version__)
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
