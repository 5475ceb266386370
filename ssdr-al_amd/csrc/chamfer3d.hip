// chamfer_3D.forward for gfx950 — the Semantic3D variant's chamfer op
// (/root/reference/SSRD_AL_semantic3d/chamfer3D/chamfer3D.cu:12-152, bound in chamfer_cuda.cpp:30-33 and wrapped by
// dist_chamfer_3D.py:29-81; called per superpoint pair by fps_gcn_cuda.py:13-30).
// For every point of cloud A the squared distance to and the index of its nearest point in cloud B, and vice versa — both directions of every batch
// element in ONE launch.
//
// Arithmetic: fp32, d = (dx*dx + dy*dy) + dz*dz with dx = b - a, every product and sum rounded on its own (-ffp-contract=off; the parity test compares
// bit for bit against that expression); the lowest index wins exact ties (a strict `d < best` over ascending indices).  The CUDA build cannot run
// here, so agreement with nvcc's own contraction of that expression is unverified (PARITY UNPINNED, tests/test_chamfer3d.py).
//
// Shape of the work on a CDNA4 CU: the support cloud is staged through LDS in slabs and read back as wave-wide BROADCASTS — every lane the same
// address — which cost the LDS pipe the same whether one or four distances are evaluated per read; so a lane owns FOUR query points (q, q + 64, q + 128,
// q + 192 of its wave's 256) and one 12-byte read of a support point serves four (distance, compare, select) groups.  The superpoints this op is
// called on hold tens to a few hundred points: a wave's 256 queries cover most of them, and a workgroup's four waves share the staged slab.
#include "ssdr_internal.hpp"

namespace ssdr {
namespace {

constexpr int C3_SLAB = 1024;      // support points per staged slab (12 KiB)
constexpr int C3_QPL = 4;          // query points per lane
constexpr int C3_QPB = 256 * C3_QPL;

struct C3Dir { const float* q; const float* s; int nq, ns; float* dist; int* idx; int blocks; };

// blockIdx.x < d0.blocks: direction 0 (queries = cloud 1), else direction 1; blockIdx.y = batch element
__global__ __launch_bounds__(256) void c3_nearest(C3Dir d0, C3Dir d1) {
    __shared__ float slab[C3_SLAB * 3];
    const bool second = (int)blockIdx.x >= d0.blocks;
    const C3Dir d = second ? d1 : d0;
    const int blk = (int)blockIdx.x - (second ? d0.blocks : 0), b = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* Q = d.q + (size_t)b * d.nq * 3; const float* S = d.s + (size_t)b * d.ns * 3;
    float qx[C3_QPL], qy[C3_QPL], qz[C3_QPL], best[C3_QPL]; int bi[C3_QPL]; bool live[C3_QPL];
#pragma unroll
    for (int v = 0; v < C3_QPL; ++v) {
        const int j = blk * C3_QPB + w * 256 + 64 * v + lane;
        live[v] = j < d.nq;
        const int jj = live[v] ? j : 0;
        qx[v] = Q[3 * (size_t)jj]; qy[v] = Q[3 * (size_t)jj + 1]; qz[v] = Q[3 * (size_t)jj + 2];
        best[v] = 3.402823466e+38f; bi[v] = 0;
    }
    for (int s0 = 0; s0 < d.ns; s0 += C3_SLAB) {
        const int cnt = min(d.ns - s0, C3_SLAB);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * 3; t += 256) slab[t] = S[(size_t)s0 * 3 + t];
        __syncthreads();
        for (int k = 0; k < cnt; ++k) {
            const float sx = slab[3 * k], sy = slab[3 * k + 1], sz = slab[3 * k + 2];          // the same address in every lane
#pragma unroll
            for (int v = 0; v < C3_QPL; ++v) {
                const float dx = sx - qx[v], dy = sy - qy[v], dz = sz - qz[v];
                const float dd = (dx * dx + dy * dy) + dz * dz;
                const bool closer = dd < best[v] || (s0 + k == 0);                             // (the first support point always enters: a NaN distance too, like the reference's)
                best[v] = closer ? dd : best[v]; bi[v] = closer ? s0 + k : bi[v];
            }
        }
    }
#pragma unroll
    for (int v = 0; v < C3_QPL; ++v) {
        const int j = blk * C3_QPB + w * 256 + 64 * v + lane;
        if (live[v]) { d.dist[(size_t)b * d.nq + j] = best[v]; d.idx[(size_t)b * d.nq + j] = bi[v]; }
    }
}

}  // namespace
}  // namespace ssdr

using namespace ssdr;

extern "C" int ssdr_chamfer3d_forward_dev(const float* d_xyz1, const float* d_xyz2, size_t batch, size_t n, size_t m,
                                          float* d_dist1, float* d_dist2, int32_t* d_idx1, int32_t* d_idx2, void* stream) {
    if (!d_xyz1 || !d_xyz2 || !d_dist1 || !d_dist2 || !d_idx1 || !d_idx2 || batch == 0 || n == 0 || m == 0 || batch > 65535 || n > 0x3fffffff || m > 0x3fffffff) { set_error("chamfer3d_forward: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream);
    C3Dir d0{d_xyz1, d_xyz2, (int)n, (int)m, d_dist1, d_idx1, (int)((n + C3_QPB - 1) / C3_QPB)};
    C3Dir d1{d_xyz2, d_xyz1, (int)m, (int)n, d_dist2, d_idx2, (int)((m + C3_QPB - 1) / C3_QPB)};
    hipLaunchKernelGGL(c3_nearest, dim3((unsigned)(d0.blocks + d1.blocks), (unsigned)batch), dim3(256), 0, s, d0, d1);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}
