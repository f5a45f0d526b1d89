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
};
