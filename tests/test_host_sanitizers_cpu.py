"""The host-side C++ of the library that touches no device -- the quadric-error-metric decimator (csrc/meshproc.cpp, utx_mesh_decimate_qem: replaces open3d's
simplify_quadric_decimation, TextureTools uv_atlas.py:155-163) -- built with gcc's AddressSanitizer + UndefinedBehaviorSanitizer and driven through its C entry point on
closed, open, degenerate and non-manifold meshes: a heap overrun, a stale index after a collapse or a signed overflow in the edge keys aborts the run.  (GPU sanitizers are
not available on the pool; this is the part of the product a CPU sanitizer can see.)"""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = textwrap.dedent(r"""
    #include <math.h>
    #include <stdio.h>
    #include <stdlib.h>
    #include <vector>
    #include "include/unitex_hip.h"

    static int check(const char* name, const std::vector<float>& v, const std::vector<int>& f, int target, int expect_rc_max) {
        const int V = (int)v.size() / 3, F = (int)f.size() / 3;
        std::vector<float> vo(v.size());
        std::vector<int> fo(f.size());
        int Vo = -1, Fo = -1;
        const int rc = utx_mesh_decimate_qem(v.data(), V, f.data(), F, target, 1.0, vo.data(), fo.data(), &Vo, &Fo);
        if (rc < 0 || rc > expect_rc_max) { printf("%s: rc %d\n", name, rc); return 1; }
        if (Vo <= 0 || Vo > V || Fo <= 0 || Fo > F) { printf("%s: counts %d %d\n", name, Vo, Fo); return 1; }
        if (rc == 0 && Fo > target) { printf("%s: %d faces above the target %d with rc 0\n", name, Fo, target); return 1; }
        for (int i = 0; i < 3 * Fo; ++i) if (fo[i] < 0 || fo[i] >= Vo) { printf("%s: index %d out of %d\n", name, fo[i], Vo); return 1; }
        for (int i = 0; i < Fo; ++i) if (fo[3 * i] == fo[3 * i + 1] || fo[3 * i + 1] == fo[3 * i + 2] || fo[3 * i] == fo[3 * i + 2]) { printf("%s: degenerate face %d\n", name, i); return 1; }
        for (int i = 0; i < 3 * Vo; ++i) if (!isfinite(vo[i])) { printf("%s: non-finite vertex\n", name); return 1; }
        printf("%s: V %d -> %d, F %d -> %d (target %d, rc %d)\n", name, V, Vo, F, Fo, target, rc);
        return 0;
    }

    static void sphere(int nu, int nv, std::vector<float>& v, std::vector<int>& f) {      // latitude / longitude sphere with shared poles: closed, manifold
        v.clear(); f.clear();
        v.insert(v.end(), {0.f, 0.f, 1.f});
        for (int i = 1; i < nv; ++i) for (int j = 0; j < nu; ++j) {
            const double t = M_PI * i / nv, p = 2 * M_PI * j / nu;
            v.insert(v.end(), {(float)(sin(t) * cos(p)), (float)(sin(t) * sin(p)), (float)cos(t)});
        }
        v.insert(v.end(), {0.f, 0.f, -1.f});
        const int south = 1 + (nv - 1) * nu;
        for (int j = 0; j < nu; ++j) f.insert(f.end(), {0, 1 + j, 1 + (j + 1) % nu});
        for (int i = 0; i < nv - 2; ++i) for (int j = 0; j < nu; ++j) {
            const int a = 1 + i * nu + j, b = 1 + i * nu + (j + 1) % nu, c = a + nu, d = b + nu;
            f.insert(f.end(), {a, c, b}); f.insert(f.end(), {b, c, d});
        }
        for (int j = 0; j < nu; ++j) f.insert(f.end(), {south, 1 + (nv - 2) * nu + (j + 1) % nu, 1 + (nv - 2) * nu + j});
    }

    static void grid(int n, std::vector<float>& v, std::vector<int>& f, float bump) {      // open n x n patch: boundary edges, constraint planes
        v.clear(); f.clear();
        for (int i = 0; i <= n; ++i) for (int j = 0; j <= n; ++j) v.insert(v.end(), {(float)i / n, (float)j / n, bump * (float)sin(0.7 * i) * (float)cos(0.9 * j)});
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
            const int a = i * (n + 1) + j, b = a + 1, c = a + n + 1, d = c + 1;
            f.insert(f.end(), {a, b, c}); f.insert(f.end(), {b, d, c});
        }
    }

    int main() {
        int bad = 0;
        std::vector<float> v; std::vector<int> f;
        sphere(96, 64, v, f);            bad += check("sphere 12k faces -> 2000", v, f, 2000, 0);
        sphere(24, 16, v, f);            bad += check("sphere -> 4 (as far as the link condition lets it)", v, f, 4, 1);
        grid(60, v, f, 0.05f);           bad += check("open bumpy grid -> 600", v, f, 600, 0);
        grid(40, v, f, 0.0f);            bad += check("flat grid (singular quadrics: placement falls back) -> 100", v, f, 100, 1);
        // duplicated vertices and zero-area faces: every position twice, faces pick either copy; plus slivers on one point
        grid(20, v, f, 0.02f);
        { const int V = (int)v.size() / 3; std::vector<float> v2(v); v2.insert(v2.end(), v.begin(), v.end());
          for (size_t i = 0; i < f.size(); i += 2) f[i] += V * (int)((i / 2) & 1);
          f.insert(f.end(), {0, 0 + V, 1}); f.insert(f.end(), {5, 5 + V, 5});
          bad += check("duplicated vertices + zero-area faces -> 200", v2, f, 200, 1); }
        // a non-manifold fan: three sheets on one edge
        { std::vector<float> vn = {0, 0, 0, 1, 0, 0, 0.5f, 1, 0, 0.5f, -1, 0, 0.5f, 0, 1, 0.5f, 0.3f, -1};
          std::vector<int> fn = {0, 1, 2, 1, 0, 3, 0, 1, 4, 1, 0, 5};
          bad += check("non-manifold fan -> 4", vn, fn, 4, 1); }
        // a target above the face count: nothing to do
        sphere(12, 8, v, f);             bad += check("target above F", v, f, 100000, 0);
        // invalid arguments are refused before anything is read
        { int Vo, Fo; float x[9] = {0}; int t[3] = {0, 1, 7};
          if (utx_mesh_decimate_qem(x, 3, t, 1, 4, 1.0, x, t, &Vo, &Fo) >= 0) { printf("index out of range accepted\n"); ++bad; }
          if (utx_mesh_decimate_qem(nullptr, 3, t, 1, 4, 1.0, x, t, &Vo, &Fo) >= 0) { printf("null accepted\n"); ++bad; } }
        printf(bad ? "FAILED\n" : "OK\n");
        return bad ? 1 : 0;
    }
""")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_qem_decimator_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    src = tmp_path / "harness.cpp"
    src.write_text(HARNESS)
    exe = tmp_path / "harness"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", ROOT,
           str(src), os.path.join(ROOT, "unitex_amd", "csrc", "meshproc.cpp"), "-o", str(exe)]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0 and r.stdout.rstrip().endswith("OK"), (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not installed")
def test_c_oracle_under_sanitizers_on_the_golden_fixtures():
    """The checker itself: oracle/geom_ref.c (rasteriser, LBVH build + walk, back-projection, composite chain, pull-push, lens blur) built with ASan + UBSan and run by the
    golden-fixture tests (G5, G6/7, G8, G9, G11 exercise it) in a child interpreter -- an out-of-bounds read in the checker could hide a kernel's error as easily as cause one."""
    b = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-3000:]
    so = os.path.join(ROOT, "oracle", "_build", "libgeom_ref_asan.so")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(asan) and os.path.exists(asan)):
        pytest.skip("libasan.so not found")
    # python itself is not instrumented: leak detection off (the interpreter's own allocations), everything else fatal
    env = dict(os.environ, UTX_ORACLE_SO=so, LD_PRELOAD=asan + ((":" + ubsan) if os.path.isabs(ubsan) and os.path.exists(ubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_golden_cpu.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    tail = (r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0 and " passed" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
