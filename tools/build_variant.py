#!/usr/bin/env python
"""Build lib/libunitex_hip_<tag>.so: the product library with ONE source compiled differently (flags / defines), for A/B probes that load it through UTX_LIB
(tools/two_stream_probe.py).   python tools/build_variant.py <tag> <source> [flags ...]     flags replace the source's extra flags of csrc/build.py.
The source variants of round 5's two-stream investigation (-DUTX_QKV_RSTD_FENCE, -DUTX_QKV_QS_VGPR, -DUTX_QKV_PROD_FENCE=1|2 in csrc/dit_elementwise.hip) live in the
history (commit d9d2b5d and its two successors): the product source carries no investigation switches."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unitex_amd.csrc import build as b
tag, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
objdir = b.OBJDIR + "_var_" + tag
os.makedirs(objdir, exist_ok=True)
objs = []
for name, extra in b.SOURCES:
    if name == src:
        objs.append(b._compile((name, flags), objdir))
    else:
        objs.append(b._compile((name, extra)))          # the product objects
out = os.path.join(b.LIBDIR, "libunitex_hip_%s.so" % tag)
r = subprocess.run([b.HIPCC, "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-3000:]
print("built", out)
