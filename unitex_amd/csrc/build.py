"""Build libunitex_hip.so (gfx950) in-tree:  python unitex_amd/csrc/build.py

hipcc cross-compiles for gfx950 without a GPU.  Objects are cached under csrc/_obj keyed on source
mtime, so a rebuild after touching one kernel takes seconds.  The .so lands in unitex_amd/lib/ (git-
ignored, but it travels to the GPU box with the gpurun snapshot).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.normpath(os.path.join(HERE, "..", "lib"))
OBJDIR = os.path.join(HERE, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# -packed-fp32-ops (device pass; the host pass prints "not a recognized feature" and ignores it): no v_pk_mul / v_pk_add / v_pk_fma_f32 in this translation unit.
# Round 5 traced the two-stream corruption of utx_qkv_post to ONE instruction pattern hipcc builds from packed fp32: a packed multiply written in place and, within two
# instructions, a packed add whose LOW lane takes that result's HIGH half (op_sel): beside another queue's kernels the low lane saw 0 for it in lanes 50-63 of a wave
# (29 of 29 dissected events: the wrong Q element = a0 c q, the second product missing; 0 of 5000 once products and sums are separated -- DESIGN 9 b,
# tools/two_stream_dissect.py, profiles/r05_two_stream_*.log).  Every translation unit whose listing showed that pattern is built without packed fp32
# (tests/test_asm_hazards_cpu.py audits all listings for it); none of them is bound by VALU throughput.
NO_PK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]

# (source, extra flags)
SOURCES = [
    ("attention.hip", NO_PK),
    ("attention_glds.hip", ["-fno-slp-vectorize"]),
    ("attention_q64.hip", []),
    ("attention_fp8.hip", NO_PK),      # round 6: the key-split epilogue made hipcc form the cross-half packed pair (tests/test_asm_hazards_cpu.py): no packed fp32 in this TU either
    ("gemm.hip", []),
    ("gemm_pers.hip", []),
    ("gemm_w4.hip", []),
    ("dit_elementwise.hip", ["-ffp-contract=off"] + NO_PK),
    ("vae.hip", NO_PK),
    ("capi.cpp", []),
    ("meshproc.cpp", []),
    ("plan.cpp", []),
    ("dit_plan.cpp", []),
]
GEOM = [
    ("raster.hip", ["-ffp-contract=off"]),
    ("bvh.hip", ["-ffp-contract=off"] + NO_PK),
    ("backproject.hip", ["-ffp-contract=off"] + NO_PK),
    ("texture_post.hip", ["-ffp-contract=off"] + NO_PK),
    ("knn.hip", ["-ffp-contract=off"]),
    ("unwrap.hip", []),
]
for s in GEOM:
    if os.path.exists(os.path.join(HERE, s[0])):
        SOURCES.append(s)

# -fvisibility=hidden: the .so exports exactly what include/unitex_hip.h declares (the header pushes default visibility around its declarations);
# the utx_launch_* / *_impl launchers the objects share among themselves stay internal
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I" + HERE,
          "-I" + os.path.normpath(os.path.join(HERE, "..", "..", "include")), "-x", "hip"]


def _newer(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src] + [os.path.join(HERE, h) for h in os.listdir(HERE) if h.endswith(".h") or h.endswith(".inc")]
    deps.append(os.path.normpath(os.path.join(HERE, "..", "..", "include", "unitex_hip.h")))
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(item, objdir=None, defines=()):
    name, extra = item
    objdir = objdir or OBJDIR
    src = os.path.join(HERE, name)
    obj = os.path.join(objdir, name + ".o")
    if _newer(src, obj):
        cmd = [HIPCC] + COMMON + list(defines) + extra + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (name, r.stderr[-4000:]))
    return obj


def build(verbose=True, ablate=False):
    """ablate=True builds libunitex_hip_ablate.so instead: the same sources with -DUTX_ABLATION, i.e. with the timing-ablation
    switches (UTX_ATTN_VAR / UTX_ATTN_DEBUG / UTX_GEMM_DEBUG -- wrong results by design) compiled in.  Only tools/ load it
    (unitex_amd._lib.use_ablation_library()); the product, the tests and bench.py never do."""
    if ablate:      # the A/B arms of the generated 4 x 64 attention stream (attention_q64_asm_var.inc, git-ignored)
        gen = os.path.normpath(os.path.join(HERE, "..", "..", "tools", "gen_attn_q64.py"))
        r = subprocess.run([sys.executable, gen, "--variants", "--keep-default"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("gen_attn_q64.py failed:\n" + r.stderr[-2000:])
    objdir = OBJDIR + ("_ablate" if ablate else "")
    defines = ["-DUTX_ABLATION"] if ablate else []
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda it: _compile(it, objdir, defines), SOURCES))
    out = os.path.join(LIBDIR, "libunitex_hip_ablate.so" if ablate else "libunitex_hip.so")
    if (not os.path.exists(out)) or any(os.path.getmtime(o) > os.path.getmtime(out) for o in objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    if verbose:
        print("built", out)
    return out


if __name__ == "__main__":
    build(ablate="--ablate" in sys.argv)
    sys.exit(0)
