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
    int ldy;              // row stride of y in floats (0: N) — the thin level-0 layers write into the 16-float rows of the gather table [x y z 0 | f0..f7 | -]
    const float* xyz; size_t xyz_batch_stride; int xyz_rows_per_batch;      // optional: also write the row's coordinates to y[row * ldy - 4 .. - 1]
    // bf16 modes: the weights once more as bf16 pieces, transposed [N][kp] (k contiguous, kp = K rounded up to 64, zero padded)
    const uint16_t* wt_hi; const uint16_t* wt_lo; int kp;
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
    // bf16 modes: attention weights [col][k] (k over D, row stride D) and LFAmlp2 weights [col][k] (row stride kp2) as bf16 pieces
    const uint16_t* fc_hi; const uint16_t* fc_lo;
    const uint16_t* l2_hi; const uint16_t* l2_lo; int kp2;
};

struct Lfa32Args {        // randla_lfa32.hip: the same operation on 32 x 32 tiles, softmax inside the lane
    const float* xyz; size_t xyz_batch_stride;
    const int* neigh;                 // [B][n][16]
    const float* fin;                 // [B][n][H]  features gathered from the neighbours
    const float* g;                   // [B][n][D]  fin W[0:H] (x log2 e)
    float* out;                       // [B][n][D]
    int n;
    const uint16_t* w1p; const float* b1;                       // LocSE operand fragments [H][2][16] (7 reformulated inputs, hi | lo slots), bias
    const uint16_t *w2_hi, *w2_lo; const float* b2;             // LFAmlp2 [H out][H k, permuted inside 16-blocks]
    const uint16_t *fc_hi, *fc_lo;                              // attention, position half [D cols][H k, permuted], x log2 e
};

// arithmetic of the matrix products (ssdr_randla_set_precision)
constexpr int PREC_F32 = 0;       // exact f32-input MFMA (v_mfma_f32_16x16x4_f32)
constexpr int PREC_BF16X3 = 1;    // split bf16: hi*hi + lo*hi + hi*lo on v_mfma_f32_16x16x32_bf16, fp32 accumulate
constexpr int PREC_BF16 = 2;      // plain bf16 operands, fp32 accumulate

int launch_dense(const DenseArgs& a, hipStream_t s);
int launch_xyz_fill(const DenseArgs& a, hipStream_t s);
int launch_lfa(int D, const LfaArgs& a, bool second, int B, hipStream_t s);
struct TailArgs {         // fc1 + fc2 + fc + softmax on the bf16 cores
    const float* x; int M, C;
    const uint16_t *w1h, *w1l; int kp1; const float* b1;
    const uint16_t *w2h, *w2l; int kp2; const float* b2;
    const uint16_t *w3h, *w3l; int kp3; const float* b3;
    float* feat32; float* probs;
    // optional: the last decoder layer in front (x is then unused): x = lrelu([skip | up[idx]] Wd + bd), 32 + 32 -> 32
    const float* skip; const float* up; const int* idx; int m_per_batch, up_rows_per_batch;
    const uint16_t *wdh, *wdl; int kpd; const float* bd;
};
int launch_tail_bf16(const TailArgs& t, int prec, hipStream_t s);      // SSDR_ERR_UNSUPPORTED (no error text) when it has no instantiation
// randla_bf16.hip: the same two operations on the bf16 matrix cores (prec = PREC_BF16X3 / PREC_BF16)
int launch_dense_bf16(const DenseArgs& a, int prec, hipStream_t s);
int launch_dense_rows2(const DenseArgs& a, const DenseArgs& b, hipStream_t s);      // two thin layers (6 -> 8 -> 8) in one pass over the rows; SSDR_ERR_UNSUPPORTED (no error text) otherwise
int launch_dense_chain(const DenseArgs& l1, const DenseArgs& l2, int prec, hipStream_t s);      // y = l2([l1(x1) | x2]) in one launch; SSDR_ERR_UNSUPPORTED (no error text) for other shapes
int launch_lfa_bf16(int D, const LfaArgs& a, bool second, int B, int prec, hipStream_t s);
int launch_lfa32(int D, const Lfa32Args& a, bool second, int B, int prec, hipStream_t s);        // SSDR_ERR_UNSUPPORTED (no error text) for a D it has no instantiation for
int launch_gather_max(const float* f, const int* idx, int n_in, int n_out, int idx_rows, int C, float* out, int B, hipStream_t s);
int launch_tail(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                int M, int C, float* feat32, float* probs, hipStream_t s);
int launch_head(const float* x, const float* W, const float* b, int M, int C, float* probs, hipStream_t s);

}  // namespace ssdr
