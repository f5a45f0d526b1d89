// Chart labelling for the UV unwrap of blank meshes (preprocess_blank_mesh; the reference calls UVAtlas through open3d,
// TextureTools/texturetools/geometry/uv/uv_atlas.py:171-175 [3p]).  A chart = connected component of the face adjacency graph
// restricted to faces of the same projection bucket.  Label propagation: chart[f] starts as f; every sweep takes the minimum over
// the face and its same-bucket neighbours, followed by pointer jumping (chart[f] = chart[chart[f]]); converges in O(log F) sweeps
// to the smallest face index of the component -- a result that does not depend on scheduling, so the labelling is deterministic.
#include "common.h"
#include "kernels.h"

__global__ __launch_bounds__(256) void chart_sweep_kernel(const int* __restrict__ adj, const int* __restrict__ bucket, int* chart, int F, int* changed) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int b = bucket[f];
    int best = chart[f];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int g = adj[3 * f + e];
        if (g >= 0 && bucket[g] == b) { const int c = chart[g]; best = c < best ? c : best; }
    }
    // pointer jumping: labels are face indices of the same component, following them only ever lowers the label
    int r = chart[best];
    best = r < best ? r : best;
    if (best < chart[f]) { atomicMin(&chart[f], best); *changed = 1; }
}

__global__ __launch_bounds__(256) void chart_init_kernel(int* chart, int F) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < F) chart[f] = f;
}

// adj [F][3] (neighbour face across edge e, -1 = border), bucket [F]; chart [F] out; flag: 1 device int.  Returns the number of
// sweeps (> 0) or a negative error.  Synchronises the stream once per 4 sweeps (reads the convergence flag).
extern "C" int utx_launch_chart_flood(const int* adj, const int* bucket, int F, int* chart, int* flag, hipStream_t stream) {
    if (F <= 0) return -2;
    const unsigned nb = (unsigned)((F + 255) / 256);
    hipLaunchKernelGGL(chart_init_kernel, dim3(nb), dim3(256), 0, stream, chart, F);
    int sweeps = 0;
    for (int round = 0; round < 4096; ++round) {
        if (hipMemsetAsync(flag, 0, sizeof(int), stream) != hipSuccess) return -5;
        for (int i = 0; i < 4; ++i) { hipLaunchKernelGGL(chart_sweep_kernel, dim3(nb), dim3(256), 0, stream, adj, bucket, chart, F, flag); ++sweeps; }
        int h = 0;
        if (hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess) return -5;
        if (hipStreamSynchronize(stream) != hipSuccess) return -5;
        if (!h) return sweeps;
    }
    return -6;
}
