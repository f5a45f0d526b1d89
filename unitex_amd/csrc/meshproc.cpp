// Host-side mesh preparation (C++; no device code): quadric-error-metric edge-collapse decimation.
//
// Replaces open3d's simplify_quadric_decimation in the reference's preprocess_blank_mesh_o3d
// (/root/reference/TextureTools/texturetools/geometry/uv/uv_atlas.py:155-163; open3d / VTK are [3p], absent here): a UV-less input with more than
// max_faces triangles is reduced to max_faces before it is unwrapped.  The published algorithm is restated (Garland & Heckbert, "Surface
// Simplification Using Quadric Error Metrics", SIGGRAPH 97): per-vertex quadrics = area-weighted sum of the squared distances to the planes of the
// incident faces, boundary edges held by perpendicular constraint planes, every edge a collapse candidate placed at the minimiser of the summed
// quadric (solved 3 x 3; midpoint / end points when singular), cheapest first, a collapse that would flip a surviving face is skipped.
// Deterministic: ties in the heap break on the edge's vertex ids.  Runs on the host like the reference's (it is mesh PREPARATION, once per mesh,
// off the denoise / back-projection path); O(F log F).
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <queue>
#include <vector>
#include "../../include/unitex_hip.h"

namespace {
struct Quadric {           // symmetric 4 x 4: a2 ab ac ad | b2 bc bd | c2 cd | d2
    double m[10];
    Quadric() { memset(m, 0, sizeof(m)); }
    void add_plane(double a, double b, double c, double d, double w) {
        m[0] += w * a * a; m[1] += w * a * b; m[2] += w * a * c; m[3] += w * a * d;
        m[4] += w * b * b; m[5] += w * b * c; m[6] += w * b * d;
        m[7] += w * c * c; m[8] += w * c * d; m[9] += w * d * d;
    }
    Quadric operator+(const Quadric& o) const { Quadric r; for (int i = 0; i < 10; ++i) r.m[i] = m[i] + o.m[i]; return r; }
    double eval(const double p[3]) const {
        const double x = p[0], y = p[1], z = p[2];
        return m[0] * x * x + 2 * m[1] * x * y + 2 * m[2] * x * z + 2 * m[3] * x + m[4] * y * y + 2 * m[5] * y * z + 2 * m[6] * y + m[7] * z * z +
               2 * m[8] * z + m[9];
    }
    bool minimiser(double out[3]) const {      // solve A x = -b, A = upper-left 3 x 3, b = (m3, m6, m8)
        const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[7];
        const double det = a * (d * f - e * e) - b * (b * f - e * c) + c * (b * e - d * c);
        const double scale = fabs(a) + fabs(d) + fabs(f);
        if (fabs(det) < 1e-10 * scale * scale * scale || scale == 0.0) return false;
        const double r0 = -m[3], r1 = -m[6], r2 = -m[8];
        out[0] = (r0 * (d * f - e * e) - b * (r1 * f - e * r2) + c * (r1 * e - d * r2)) / det;
        out[1] = (a * (r1 * f - r2 * e) - r0 * (b * f - e * c) + c * (b * r2 - r1 * c)) / det;
        out[2] = (a * (d * r2 - e * r1) - b * (b * r2 - r1 * c) + r0 * (b * e - d * c)) / det;
        return true;
    }
};
struct Cand { double cost; int a, b; unsigned va, vb; double p[3]; };
struct CandLess { bool operator()(const Cand& x, const Cand& y) const {
    if (x.cost != y.cost) return x.cost > y.cost;
    if (x.a != y.a) return x.a > y.a;
    return x.b > y.b; } };

inline void cross3(const double* u, const double* v, double* o) { o[0] = u[1] * v[2] - u[2] * v[1]; o[1] = u[2] * v[0] - u[0] * v[2]; o[2] = u[0] * v[1] - u[1] * v[0]; }
}  // namespace

extern "C" int utx_mesh_decimate_qem(const float* verts_in, int V, const int* faces_in, int F, int target_faces, double boundary_weight,
                                     float* verts_out, int* faces_out, int* V_out, int* F_out) {
    if (!verts_in || !faces_in || !verts_out || !faces_out || !V_out || !F_out || V <= 0 || F <= 0 || target_faces < 4) return -2;
    std::vector<double> P(3 * (size_t)V);
    for (size_t i = 0; i < 3 * (size_t)V; ++i) P[i] = verts_in[i];
    std::vector<int> Fc(faces_in, faces_in + 3 * (size_t)F);
    for (int i = 0; i < 3 * F; ++i) if (Fc[i] < 0 || Fc[i] >= V) return -2;
    std::vector<char> falive(F, 1);
    std::vector<unsigned> ver(V, 0);
    std::vector<std::vector<int>> vf(V);
    for (int f = 0; f < F; ++f) for (int k = 0; k < 3; ++k) vf[Fc[3 * f + k]].push_back(f);
    std::vector<Quadric> Q(V);
    auto face_normal = [&](int a, int b, int c, double* n) {
        double u[3] = {P[3 * b] - P[3 * a], P[3 * b + 1] - P[3 * a + 1], P[3 * b + 2] - P[3 * a + 2]};
        double v[3] = {P[3 * c] - P[3 * a], P[3 * c + 1] - P[3 * a + 1], P[3 * c + 2] - P[3 * a + 2]};
        cross3(u, v, n);
    };
    // face quadrics (area-weighted) + boundary constraint planes
    std::vector<std::pair<long long, int>> edges;      // (key, face) to find boundary edges
    edges.reserve(3 * (size_t)F);
    for (int f = 0; f < F; ++f) {
        const int a = Fc[3 * f], b = Fc[3 * f + 1], c = Fc[3 * f + 2];
        double n[3]; face_normal(a, b, c, n);
        const double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (len > 0) {
            const double nx = n[0] / len, ny = n[1] / len, nz = n[2] / len, d = -(nx * P[3 * a] + ny * P[3 * a + 1] + nz * P[3 * a + 2]);
            for (int v : {a, b, c}) Q[v].add_plane(nx, ny, nz, d, 0.5 * len);
        }
        for (int k = 0; k < 3; ++k) {
            const int u = Fc[3 * f + k], w = Fc[3 * f + (k + 1) % 3];
            edges.push_back({(long long)std::min(u, w) * V + std::max(u, w), 3 * f + k});
        }
    }
    std::sort(edges.begin(), edges.end());
    for (size_t i = 0; i < edges.size();) {
        size_t j = i; while (j < edges.size() && edges[j].first == edges[i].first) ++j;
        if (j - i == 1 && boundary_weight > 0) {      // boundary edge: plane through it, perpendicular to its face
            const int f = edges[i].second / 3, k = edges[i].second % 3;
            const int u = Fc[3 * f + k], w = Fc[3 * f + (k + 1) % 3];
            double n[3]; face_normal(Fc[3 * f], Fc[3 * f + 1], Fc[3 * f + 2], n);
            double e[3] = {P[3 * w] - P[3 * u], P[3 * w + 1] - P[3 * u + 1], P[3 * w + 2] - P[3 * u + 2]}, pn[3];
            cross3(e, n, pn);
            const double len = sqrt(pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2]);
            if (len > 0) {
                const double el = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
                const double nx = pn[0] / len, ny = pn[1] / len, nz = pn[2] / len, d = -(nx * P[3 * u] + ny * P[3 * u + 1] + nz * P[3 * u + 2]);
                Q[u].add_plane(nx, ny, nz, d, boundary_weight * el * el); Q[w].add_plane(nx, ny, nz, d, boundary_weight * el * el);
            }
        }
        i = j;
    }
    std::priority_queue<Cand, std::vector<Cand>, CandLess> heap;
    auto push = [&](int a, int b) {
        if (a == b) return;
        if (a > b) std::swap(a, b);
        Cand c; c.a = a; c.b = b; c.va = ver[a]; c.vb = ver[b];
        const Quadric q = Q[a] + Q[b];
        double best = INFINITY;
        double cand[4][3]; int nc = 0;
        if (q.minimiser(cand[0])) nc = 1;
        for (int k = 0; k < 3; ++k) { cand[nc][k] = 0.5 * (P[3 * a + k] + P[3 * b + k]); cand[nc + 1][k] = P[3 * a + k]; cand[nc + 2][k] = P[3 * b + k]; }
        nc += 3;
        for (int i = 0; i < nc; ++i) { const double e = q.eval(cand[i]); if (e < best) { best = e; memcpy(c.p, cand[i], sizeof(c.p)); } }
        c.cost = best;
        heap.push(c);
    };
    for (size_t i = 0; i < edges.size();) {
        size_t j = i; while (j < edges.size() && edges[j].first == edges[i].first) ++j;
        push((int)(edges[i].first / V), (int)(edges[i].first % V));
        i = j;
    }
    int nf = F;
    std::vector<int> nb, na, nbv;
    while (nf > target_faces && !heap.empty()) {
        const Cand c = heap.top(); heap.pop();
        const int a = c.a, b = c.b;
        if (ver[a] != c.va || ver[b] != c.vb) continue;        // stale
        // flip check: every surviving face around a or b keeps the sign of its normal when its corner moves to c.p
        bool ok = true;
        int shared = 0;
        for (int pass = 0; pass < 2 && ok; ++pass) {
            const int v = pass ? b : a, o = pass ? a : b;
            for (int f : vf[v]) {
                if (!falive[f]) continue;
                const int* t = &Fc[3 * f];
                if (t[0] == o || t[1] == o || t[2] == o) { if (!pass) ++shared; continue; }      // collapses away
                double n0[3], n1[3], q[3][3];
                face_normal(t[0], t[1], t[2], n0);
                for (int k = 0; k < 3; ++k) for (int d = 0; d < 3; ++d) q[k][d] = (t[k] == v) ? c.p[d] : P[3 * t[k] + d];
                double u[3] = {q[1][0] - q[0][0], q[1][1] - q[0][1], q[1][2] - q[0][2]}, w[3] = {q[2][0] - q[0][0], q[2][1] - q[0][1], q[2][2] - q[0][2]};
                cross3(u, w, n1);
                const double dot = n0[0] * n1[0] + n0[1] * n1[1] + n0[2] * n1[2];
                const double l0 = n0[0] * n0[0] + n0[1] * n0[1] + n0[2] * n0[2], l1 = n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2];
                if (dot <= 0.04 * sqrt(l0 * l1)) { ok = false; break; }      // turned by more than ~78 degrees (or degenerate)
            }
        }
        if (!ok || shared > 2) continue;      // shared > 2: a non-manifold fan around the edge -- leave it alone
        // link condition: the vertices adjacent to BOTH ends must be exactly the apexes of the faces on the edge (one per shared face); a further common
        // neighbour means the collapse would pinch the surface (two sheets glued along an edge: duplicate / non-manifold faces)
        {
            na.clear(); nbv.clear();
            for (int f : vf[a]) if (falive[f]) for (int k = 0; k < 3; ++k) if (Fc[3 * f + k] != a) na.push_back(Fc[3 * f + k]);
            for (int f : vf[b]) if (falive[f]) for (int k = 0; k < 3; ++k) if (Fc[3 * f + k] != b) nbv.push_back(Fc[3 * f + k]);
            std::sort(na.begin(), na.end()); na.erase(std::unique(na.begin(), na.end()), na.end());
            std::sort(nbv.begin(), nbv.end()); nbv.erase(std::unique(nbv.begin(), nbv.end()), nbv.end());
            int common = 0;
            for (size_t i = 0, j = 0; i < na.size() && j < nbv.size();) {
                if (na[i] < nbv[j]) ++i; else if (na[i] > nbv[j]) ++j; else { ++common; ++i; ++j; }
            }
            if (common != shared) continue;
        }
        // collapse b into a
        for (int d = 0; d < 3; ++d) P[3 * a + d] = c.p[d];
        Q[a] = Q[a] + Q[b];
        ++ver[a]; ++ver[b];
        for (int f : vf[b]) {
            if (!falive[f]) continue;
            int* t = &Fc[3 * f];
            if (t[0] == a || t[1] == a || t[2] == a) { falive[f] = 0; --nf; continue; }
            for (int k = 0; k < 3; ++k) if (t[k] == b) t[k] = a;
            vf[a].push_back(f);
        }
        vf[b].clear();
        // compact a's face list and re-queue its edges
        auto& fa = vf[a];
        fa.erase(std::remove_if(fa.begin(), fa.end(), [&](int f) { return !falive[f]; }), fa.end());
        nb.clear();
        for (int f : fa) for (int k = 0; k < 3; ++k) if (Fc[3 * f + k] != a) nb.push_back(Fc[3 * f + k]);
        std::sort(nb.begin(), nb.end()); nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
        for (int v : nb) push(a, v);
    }
    // compact
    std::vector<int> remap(V, -1);
    int nv = 0, of = 0;
    for (int f = 0; f < F; ++f) {
        if (!falive[f]) continue;
        const int* t = &Fc[3 * f];
        if (t[0] == t[1] || t[1] == t[2] || t[0] == t[2]) continue;
        for (int k = 0; k < 3; ++k) {
            if (remap[t[k]] < 0) { remap[t[k]] = nv; for (int d = 0; d < 3; ++d) verts_out[3 * nv + d] = (float)P[3 * t[k] + d]; ++nv; }
            faces_out[3 * of + k] = remap[t[k]];
        }
        ++of;
    }
    *V_out = nv; *F_out = of;
    return of > target_faces ? 1 : 0;      // 1: every remaining collapse was rejected (flip / fan / link condition) before the target was reached -- the mesh is valid, but larger than asked
}
