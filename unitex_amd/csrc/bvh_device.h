// Device-side LBVH traversal shared by bvh.hip (utx_bvh_trace) and backproject.hip (fused visibility).
// Reproduces intersect_test2.slang:14-146 of the reference including its quirks; see bvh.hip / oracle header.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ bool aabb_hit(const float* ro, const float* rd, float tmin, float tmax, const float* bb) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float d = rd[i];
        if (d == 0.f) d = 0.000001f;
        const float inv = 1.0f / d;
        float t0 = (bb[i] - ro[i]) * inv;
        float t1 = (bb[3 + i] - ro[i]) * inv;
        if (inv < 0.0f) { const float t = t1; t1 = t0; t0 = t; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax < tmin) return false;
    }
    return true;
}
__device__ __forceinline__ float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ bool tri_hit(const float* ro, const float* rd, const float* v0, const float* v1, const float* v2, float& t_out) {
    const float eps = 1e-9f;
    const float E1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    const float E2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    float P[3]; cross3(rd, E2, P);
    const float det = dot3(E1, P);
    if (det > -eps && det < eps) return false;
    const float inv = 1.0f / det;
    const float T[3] = {ro[0] - v0[0], ro[1] - v0[1], ro[2] - v0[2]};
    const float u = dot3(T, P) * inv;
    if (u < 0 || u > 1) return false;
    float Q[3]; cross3(T, E1, Q);
    const float v = dot3(rd, Q) * inv;
    if (v < 0 || u + v > 1) return false;
    t_out = dot3(E2, Q) * inv;
    return true;
}

__device__ __forceinline__ int bvh_trace_one(const int* __restrict__ info, const float* __restrict__ aabb, const float* __restrict__ vert,
                             const int* __restrict__ faces, const float* ro, const float* rd_in) {
    const float nrm = sqrtf(dot3(rd_in, rd_in));
    const float rd[3] = {rd_in[0] / nrm, rd_in[1] / nrm, rd_in[2] / nrm};
    int stack[64];
    int count = 0;
    stack[count++] = 0;
    float closest = 1e9f;
    int hit_tid = -1;
    while (count > 0) {
        const int nd = stack[--count];
        float bb[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) bb[k] = aabb[6 * (long)nd + k];
        if (!aabb_hit(ro, rd, 0.f, closest, bb)) continue;
        const int L = info[3 * (long)nd], R = info[3 * (long)nd + 1];
        if (L != 0 && R != 0) {
            if (count + 2 <= 64) { stack[count++] = L; stack[count++] = R; }
        } else if (L == 0 && R == 0) {
            const int prim = info[3 * (long)nd + 2];
            const int* f = faces + 3 * (long)prim;
            float t;
            if (tri_hit(ro, rd, vert + 3 * (long)f[0], vert + 3 * (long)f[1], vert + 3 * (long)f[2], t)) {
                closest = t < closest ? t : closest;
                hit_tid = prim;
            }
        }
    }
    return hit_tid;
}


// ---- the same traversal over the PACKED tree (built by bvh_pack_kernel; used whenever the tree is shallow enough that the
// reference's 64-entry stack can never overflow, i.e. always in practice):
//   node  = two 16-byte words  {bb[0..3]} {bb[4], bb[5], link, esc}      (the reference reads 6 + 3 scalar dwords from two arrays)
//           link > 0: internal node, link = its SECOND child (the reference pushes first, second and pops the second first);
//           link < 0: leaf of primitive ~link.  esc = the node the reference would pop after this node's subtree is done:
//           the first child of the parent for a second child, the parent's esc for a first child, -1 at the root.
//   tri   = three 16-byte words {v0, 0} {v1, 0} {v2, 0}                  (the reference chases faces[] then three vertices)
// No stack, so no scratch memory and no dependent index load between "pop" and the next box test.  The order in which nodes and
// triangles are visited -- and with it the quirk that the LAST triangle hit wins, not the closest -- is the reference's exactly:
// miss -> esc; internal hit -> second child (whose esc is the first child); leaf -> test, esc.
__device__ __forceinline__ int bvh_trace_packed(const float4* __restrict__ nodes, const float4* __restrict__ tris, const float* ro,
                                                const float* rd_in, unsigned* visited) {
    const float nrm = sqrtf(dot3(rd_in, rd_in));
    const float rd[3] = {rd_in[0] / nrm, rd_in[1] / nrm, rd_in[2] / nrm};
    float closest = 1e9f;
    int hit_tid = -1;
    int nd = 0;
    unsigned nv = 0;
    while (nd >= 0) {
        const float4 a = nodes[2 * (long)nd], b = nodes[2 * (long)nd + 1];
        const float bb[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
        const int link = __float_as_int(b.z), esc = __float_as_int(b.w);
        ++nv;
        if (!aabb_hit(ro, rd, 0.f, closest, bb)) { nd = esc; continue; }
        if (link > 0) { nd = link; continue; }
        const int prim = ~link;
        const float4 t0 = tris[3 * (long)prim], t1 = tris[3 * (long)prim + 1], t2 = tris[3 * (long)prim + 2];
        const float v0[3] = {t0.x, t0.y, t0.z}, v1[3] = {t1.x, t1.y, t1.z}, v2[3] = {t2.x, t2.y, t2.z};
        float t;
        if (tri_hit(ro, rd, v0, v1, v2, t)) {
            closest = t < closest ? t : closest;
            hit_tid = prim;
        }
        nd = esc;
    }
    if (visited) *visited = nv;
    return hit_tid;
}

struct utx_bvh {
    int F;
    int* info;       // [2F-1][3]
    float* aabb;     // [2F-1][6]
    const float* verts;  // borrowed (caller keeps them alive while the bvh is used)
    const int* faces;
    // build temporaries (kept: small)
    float* ebox;     // [F][6]
    unsigned* codes; unsigned* codes_sorted; int* idx; int* idx_sorted;
    int* parent;     // [2F-1]
    int* counter;    // [F-1]
    unsigned* extent;  // 6 ordered-uint min/max
    void* sort_tmp; size_t sort_tmp_bytes;
    float4* nodes;   // [2F-1][2] packed tree (bvh_trace_packed)
    float4* tris;    // [F][3]
    int* depth_dev;  // max number of ancestors of a leaf
    int depth;       // host copy; the packed traversal is used when depth <= UTX_BVH_PACKED_MAX_DEPTH
};
#define UTX_BVH_PACKED_MAX_DEPTH 60   // the reference's stack holds 64 entries and the walk keeps at most depth + 1 of them
