// Argument blocks of the RandLA-Net kernels (randla_kernels.hip) and their launchers.
#pragma once
#include "ssdr_internal.hpp"

namespace ssdr {

struct DenseArgs {        // y[M,N] = act( [x1 | x2(gathered)] [M,k1+k2] * W[k1+k2,N] + b )
    const float* x1; int k1;
    const float* x2; int k2;
    const int* idx2;      // optional: row of x2 inside its batch element for every output row
    int m_per_batch;      // output rows per batch element (with idx2)
    int x2_rows_per_batch;
    const float* W; const float* b;
    float* y; int M, N; int act;
};

struct LfaArgs {
    const float* xyz; size_t xyz_batch_stride;   // level points = prefix of the tile's points
    const int* neigh;                            // [B][n][16]
    const float* fin;                            // [B][n][D/2]  features to gather from the neighbours
    const float* w_l1; const float* b_l1;        // LFAmlp1 10 -> D/2 (BN folded)
    const float* w_l2; const float* b_l2;        // LFAmlp2 D/2 -> D/2 (second half only)
    const float* w_fc;                           // attention dense D -> D, no bias, [k][col]
    const float* w_fc_t;                         // the same transposed, [col][k]
    const float* g;                              // optional [B][n][D]: fin * w_fc[0:D/2] per point (d >= 64): the kernel then only multiplies the position half
    float* out;                                  // [B][n][D]   sum_k f * softmax_k(f W)
    int n;
};

int launch_dense(const DenseArgs& a, hipStream_t s);
int launch_lfa(int D, const LfaArgs& a, bool second, int B, hipStream_t s);
int launch_gather_max(const float* f, const int* idx, int n_in, int n_out, int idx_rows, int C, float* out, int B, hipStream_t s);
int launch_tail(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                int M, int C, float* feat32, float* probs, hipStream_t s);
int launch_head(const float* x, const float* W, const float* b, int M, int C, float* probs, hipStream_t s);

}  // namespace ssdr
