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

// ---- the same traversal for a COHERENT PACKET: the 64 rays of a wave walk the packed tree together (backproject.hip: an 8 x 8 texel tile of one view --
// parallel rays from neighbouring surface points).  One node per step for the whole wave, fetched through the SCALAR cache (a uniform 32-byte
// s_load instead of 64 divergent 2 x 16-byte vector loads per step: the thread-per-ray walk is bound by exactly those -- 50 dependent divergent
// fetches per ray); each lane tests the box with its OWN ray and its OWN `closest`; the packet descends where any lane hits, and a lane takes part
// in a node iff it hit every ancestor -- the 64-bit masks travel with the packet's DFS stack (<= depth entries of {node, mask}, kept lane-distributed
// in LDS, 12 bytes per entry and wave, written and read wave-uniformly).  (A first version kept the stack lane-distributed in VGPRs -- entry i in lane i,
// v_writelane / v_readlane -- and hung the GPU: entries parked in lanes that are INACTIVE in the calling branch are lost whenever the compiler copies the
// register, because its copies are EXEC-masked.)
// Per lane the sequence of nodes it takes part in, and of triangles it tests, is EXACTLY the sequence of its own stackless walk above (the packet
// visits second child before first child, as the reference pops them; a lane that misses a node skips the subtree as `esc` would): the same
// `closest` at every test, the same last-hit-wins result -- bit-identical to bvh_trace_packed, ray by ray.  `first child` is not stored in a node:
// it is the escape link of the second child, so the push of (first child, mask) happens when the second child's words arrive.
typedef float bvh_f32x8 __attribute__((ext_vector_type(8)));
typedef float bvh_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bvh_f32x8 bvh_sload8(const void* p) {
    bvh_f32x8 r;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return r;
}
__device__ __forceinline__ bvh_f32x4 bvh_sload4(const void* p) {
    bvh_f32x4 r;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return r;
}
__device__ __forceinline__ int bvh_trace_packet(const float4* __restrict__ nodes, const float4* __restrict__ tris, const float* ro, const float* rd_in,
                                                bool valid, int* stk /* this wave's 64 x 3 ints of LDS */, unsigned* visited) {
    const float nrm = sqrtf(dot3(rd_in, rd_in));
    const float rd[3] = {rd_in[0] / nrm, rd_in[1] / nrm, rd_in[2] / nrm};
    float closest = 1e9f;
    int hit_tid = -1;
    unsigned nv = 0;
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const unsigned long long lbit = 1ull << lane;
    int sp = 0;                                        // the packet's stack: stk[3 sp + {0, 1, 2}] = {node, mask lo, mask hi}
    int cur = 0;
    unsigned long long cm = __ballot(valid);
    bool pend = false;                                 // the first child of the node just descended from is still to be pushed (with mask pm)
    unsigned long long pm = 0;
    while (cm != 0) {
        const bvh_f32x8 n8 = bvh_sload8((const char*)nodes + (long)cur * 32);
        const float bb[6] = {n8[0], n8[1], n8[2], n8[3], n8[4], n8[5]};
        const int link = __float_as_int(n8[6]), esc = __float_as_int(n8[7]);
        if (pend) {
            stk[3 * sp] = esc; stk[3 * sp + 1] = (int)(unsigned)pm; stk[3 * sp + 2] = (int)(unsigned)(pm >> 32);      // every active lane writes the same words
            ++sp;
            pend = false;
        }
        const bool act = (cm & lbit) != 0;
        if (act) ++nv;
        const bool h = act && aabb_hit(ro, rd, 0.f, closest, bb);
        const unsigned long long hm = __ballot(h);
        if (link > 0) {
            if (hm != 0) { cur = link; cm = hm; pm = hm; pend = true; continue; }
        } else if (hm != 0) {
            const int prim = ~link;
            const char* tp = (const char*)tris + (long)prim * 48;
            const bvh_f32x8 ta = bvh_sload8(tp);
            const bvh_f32x4 tb = bvh_sload4(tp + 32);
            if (h) {
                const float v0[3] = {ta[0], ta[1], ta[2]}, v1[3] = {ta[4], ta[5], ta[6]}, v2[3] = {tb[0], tb[1], tb[2]};
                float t;
                if (tri_hit(ro, rd, v0, v1, v2, t)) {
                    closest = t < closest ? t : closest;
                    hit_tid = prim;
                }
            }
        }
        if (sp == 0) break;
        --sp;
        cur = __builtin_amdgcn_readfirstlane(stk[3 * sp]);
        cm = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(stk[3 * sp + 1]) |
             ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(stk[3 * sp + 2]) << 32);
    }
    if (visited) *visited = nv;
    return hit_tid;
}

#include <atomic>
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
    std::atomic<int> depth;   // host copy (-1: not read back yet -- utx_bvh_depth_impl waits for depth_ready once; atomic: a handle may see its first launches from two threads); the packed traversal is used when depth <= UTX_BVH_PACKED_MAX_DEPTH
    int* depth_host;           // pinned word the build copies depth_dev into
    hipEvent_t depth_ready;    // recorded behind that copy
    void* owned;               // the one hipMalloc of utx_bvh_build (null: the arrays live in the caller's workspace, utx_bvh_build_ws)
};
#define UTX_BVH_PACKED_MAX_DEPTH 60   // the reference's stack holds 64 entries and the walk keeps at most depth + 1 of them
