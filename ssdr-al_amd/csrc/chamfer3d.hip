// chamfer_3D.forward for gfx950 — the Semantic3D variant's CUDA op
// (/root/reference/SSRD_AL_semantic3d/chamfer3D/chamfer3D.cu:12-152, bound in chamfer_cuda.cpp:30-33 and wrapped by
// dist_chamfer_3D.py:29-81; called per superpoint pair by fps_gcn_cuda.py:13-30).
// For every point of cloud A the squared distance to and the index of its nearest point in cloud B, and vice versa.
// fp32, d = (dx*dx + dy*dy) + dz*dz with dx = b - a as in the reference kernel; the lowest index wins exact ties
// (the reference's strict `d < best` inside a chunk and strict `result > best` across chunks give the same rule).
// The CUDA build cannot run here, so bit-level agreement with nvcc's FMA contraction of that expression is unverified.
#include "ssdr_internal.hpp"

namespace ssdr {
namespace {

constexpr int CT = 512;   // support points staged per step, as the reference (chamfer3D.cu:13)

__global__ __launch_bounds__(256) void nm_distance(int n, const float* __restrict__ xyz, int m, const float* __restrict__ xyz2,
                                                   float* __restrict__ result, int* __restrict__ result_i) {
    __shared__ float buf[CT * 3];
    const int b = blockIdx.y;
    const float* A = xyz + (size_t)b * n * 3; const float* B = xyz2 + (size_t)b * m * 3;
    const int j = blockIdx.x * 256 + threadIdx.x;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (j < n) { x1 = A[3 * (size_t)j]; y1 = A[3 * (size_t)j + 1]; z1 = A[3 * (size_t)j + 2]; }
    float best = 0.f; int best_i = 0; bool have = false;
    for (int k2 = 0; k2 < m; k2 += CT) {
        const int end_k = min(m, k2 + CT) - k2;
        __syncthreads();
        for (int t = threadIdx.x; t < end_k * 3; t += 256) buf[t] = B[(size_t)k2 * 3 + t];
        __syncthreads();
        if (j < n) {
            for (int k = 0; k < end_k; ++k) {
                const float x2 = buf[3 * k] - x1, y2 = buf[3 * k + 1] - y1, z2 = buf[3 * k + 2] - z1;
                const float d = (x2 * x2 + y2 * y2) + z2 * z2;
                if (!have || d < best) { best = d; best_i = k + k2; have = true; }
            }
        }
    }
    if (j < n) { result[(size_t)b * n + j] = best; result_i[(size_t)b * n + j] = best_i; }
}

}  // namespace
}  // namespace ssdr

using namespace ssdr;

extern "C" int ssdr_chamfer3d_forward_dev(const float* d_xyz1, const float* d_xyz2, size_t batch, size_t n, size_t m,
                                          float* d_dist1, float* d_dist2, int32_t* d_idx1, int32_t* d_idx2, void* stream) {
    if (!d_xyz1 || !d_xyz2 || !d_dist1 || !d_dist2 || !d_idx1 || !d_idx2 || batch == 0 || n == 0 || m == 0 || batch > 65535) { set_error("chamfer3d_forward: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream);
    hipLaunchKernelGGL(nm_distance, dim3((unsigned)((n + 255) / 256), (unsigned)batch), dim3(256), 0, s, (int)n, d_xyz1, (int)m, d_xyz2, d_dist1, d_idx1);
    hipLaunchKernelGGL(nm_distance, dim3((unsigned)((m + 255) / 256), (unsigned)batch), dim3(256), 0, s, (int)m, d_xyz2, (int)n, d_xyz1, d_dist2, d_idx2);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}
