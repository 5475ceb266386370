// "Next" row N2 of the scope table: the steps right after inference in the reference's evaluation
// (S3/RandLANet.py:326-334, 353-411; S3/helper_tool.py:237-262): vote smoothing of per-point probabilities,
// arg-max, confusion matrix, IoU.  Small HBM-bound kernels.
#include "ssdr_internal.hpp"

namespace ssdr {
namespace {

// test_probs[p_idx] = s * test_probs[p_idx] + (1 - s) * probs   (RandLANet.py:333) with NumPy's fancy-assignment rule
// for repeated indices (padded tiles repeat points): every right-hand side uses the OLD row, the LAST occurrence wins.
__global__ __launch_bounds__(256) void vote_owner(const int* __restrict__ p_idx, int n, int* owner) {
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) atomicMax(&owner[p_idx[j]], j);
}
__global__ __launch_bounds__(256) void vote_apply(const int* __restrict__ p_idx, int n, int C, const float* __restrict__ probs, float smooth, float one_minus,
                                                  int* owner, float* test_probs) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)n * C; e += (long long)gridDim.x * 256) {
        const int j = (int)(e / C), c = (int)(e % C), p = p_idx[j];
        if (owner[p] == j) {
            float* t = test_probs + (size_t)p * C + c;
            *t = smooth * *t + one_minus * probs[e];
        }
    }
}
__global__ __launch_bounds__(256) void vote_reset(const int* __restrict__ p_idx, int n, int* owner) {
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) owner[p_idx[j]] = -1;
}

// preds = argmax(probs[proj_idx or i]) ; confusion[label][pred] += 1  (sklearn confusion_matrix with labels = 0..C-1)
__global__ __launch_bounds__(256) void confusion_kernel(const float* __restrict__ probs, int C, const int* __restrict__ proj_idx, const int* __restrict__ labels,
                                                        long long n, int* pred_out, unsigned long long* conf) {
    __shared__ unsigned int sc[32 * 32];
    for (int i = threadIdx.x; i < C * C; i += 256) sc[i] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float* p = probs + (size_t)(proj_idx ? proj_idx[i] : i) * C;
        int best = 0; float bv = p[0];
        for (int c = 1; c < C; ++c) if (p[c] > bv) { bv = p[c]; best = c; }          // np.argmax: first maximum
        if (pred_out) pred_out[i] = best;
        const int l = labels[i];
        if (l >= 0 && l < C) atomicAdd(&sc[l * C + best], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256) if (sc[i]) atomicAdd(&conf[i], (unsigned long long)sc[i]);
}

// DP.IoU_from_confusions (helper_tool.py:237-262), float64 like NumPy on an integer confusion matrix
__global__ void iou_kernel(const unsigned long long* __restrict__ conf, int C, double* iou) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double sum_iou = 0.0; int counts = 0;
    for (int c = 0; c < C; ++c) {
        double tp = (double)conf[c * C + c], fn = 0.0, fp = 0.0;
        for (int k = 0; k < C; ++k) { fn += (double)conf[c * C + k]; fp += (double)conf[k * C + c]; }
        iou[c] = tp / (fp + fn - tp + 1e-6);
        sum_iou += iou[c];
        if (!(fn < 1e-3)) ++counts;
    }
    const double miou = sum_iou / ((double)counts + 1e-6);
    for (int c = 0; c < C; ++c) {
        double fn = 0.0;
        for (int k = 0; k < C; ++k) fn += (double)conf[c * C + k];
        if (fn < 1e-3) iou[c] += miou;
    }
}

inline int grid_for(long long n) { return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, 4096)); }

}  // namespace
}  // namespace ssdr

using namespace ssdr;

extern "C" {

int ssdr_vote_smooth_dev(float* d_test_probs, const int32_t* d_point_idx, const float* d_probs, size_t n, int num_classes, double smooth,
                         int32_t* d_owner_scratch, void* stream) {
    if (!d_test_probs || !d_point_idx || !d_probs || !d_owner_scratch || num_classes < 1) { set_error("vote_smooth: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (n == 0) return SSDR_OK;
    hipStream_t s = pick_stream(stream);
    const int ni = (int)n;
    hipLaunchKernelGGL(vote_owner, dim3(grid_for(ni)), dim3(256), 0, s, d_point_idx, ni, d_owner_scratch);
    hipLaunchKernelGGL(vote_apply, dim3(grid_for((long long)ni * num_classes)), dim3(256), 0, s, d_point_idx, ni, num_classes, d_probs, (float)smooth, (float)(1.0 - smooth), d_owner_scratch, d_test_probs);   // the reference forms 1 - test_smooth in Python floats
    hipLaunchKernelGGL(vote_reset, dim3(grid_for(ni)), dim3(256), 0, s, d_point_idx, ni, d_owner_scratch);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int ssdr_confusion_dev(const float* d_probs, int num_classes, const int32_t* d_proj_idx, const int32_t* d_labels, size_t n, int32_t* d_pred,
                       uint64_t* d_confusion, double* d_iou, void* stream) {
    if (!d_probs || !d_labels || !d_confusion || num_classes < 1 || num_classes > 32) { set_error("confusion: bad arguments (num_classes <= 32)"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream);
    if (n) hipLaunchKernelGGL(confusion_kernel, dim3(grid_for((long long)n)), dim3(256), 0, s, d_probs, num_classes, d_proj_idx, d_labels, (long long)n, d_pred,
                              (unsigned long long*)d_confusion);
    if (d_iou) hipLaunchKernelGGL(iou_kernel, dim3(1), dim3(64), 0, s, (const unsigned long long*)d_confusion, num_classes, d_iou);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

}
