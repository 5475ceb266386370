// RandLA-Net inference driver + C ABI (include/ssdr_al.h, "RandLA-Net" section).
//
// Layer table (BN already folded by the caller; W row-major [in,out], b [out]):
//   0                fc0        in_dim -> 8                                   (RandLANet.py:144-146)
//   1 + 8*i + 0..7   encoder i: mlp1 (d_in->h), LFAmlp1 (10->h), att1 fc (d->d, no bias), att1 mlp (d->h),
//                               LFAmlp2 (h->h), att2 fc (d->d, no bias), att2 mlp (d->d),
//                               residual = vstack(mlp2, shortcut) (d + d_in -> 2d)      (:505-527, :572-585)
//   1 + 8*L          decoder_0  (2*d_L -> 2*d_L)                                       (:159-161)
//   2 + 8*L + j      Decoder_layer_j: [skip | interp] -> skip                          (:165-172)
//   then fc1 (->64), fc2 (64->32 = last_second_features), fc (32->C)                   (:174-178)
#include "ssdr_internal.hpp"
#include "randla.hpp"
#include <cstring>

namespace ssdr {
namespace {

struct Layer {
    int in = 0, out = 0; bool has_b = true; DevBuf W, b, Wt; bool set = false;     // Wt: [out][in] copy for the attention layers
    DevBuf Wh, Wl; int kp = 0;        // bf16 pieces of W, transposed [out][kp] (kp = in rounded up to 64, zero padded): hi = bf16(w), lo = bf16(w - hi)
    DevBuf Ph, Pl, P1;                // operands of lfa32_kernel (randla_lfa32.hip): k-permuted pieces [out][K] / LocSE fragments [out][2][16]
};

struct Model {
    int L = 5, K = 16, C = 13, in_dim = 6;
    int prec = PREC_F32;
    int tiles32 = 1;             // lfa32 formulation in the bf16 modes (ssdr_randla_set_formulation)
    int d_out[8] = {16, 64, 128, 256, 512, 0, 0, 0};
    std::vector<Layer> layers;
    std::vector<DevBuf> ws;      // activation workspaces
};

int n_layers(const Model& m) { return 1 + 8 * m.L + 1 + m.L + 3; }

void shapes(Model& m) {
    m.layers.clear(); m.layers.resize(n_layers(m));
    auto set = [&](int i, int in, int out, bool b) { m.layers[i].in = in; m.layers[i].out = out; m.layers[i].has_b = b; };
    set(0, m.in_dim, 8, true);
    int d_in = 8;
    std::vector<int> enc_ch; enc_ch.push_back(2 * m.d_out[0]);
    for (int i = 0; i < m.L; ++i) {
        const int d = m.d_out[i], h = d / 2, base = 1 + 8 * i;
        set(base + 0, d_in, h, true); set(base + 1, 10, h, true); set(base + 2, d, d, false); set(base + 3, d, h, true);
        set(base + 4, h, h, true); set(base + 5, d, d, false); set(base + 6, d, d, true); set(base + 7, d + d_in, 2 * d, true);
        d_in = 2 * d; enc_ch.push_back(2 * d);
    }
    int idx = 1 + 8 * m.L;
    set(idx++, d_in, d_in, true);
    int feat = d_in;
    for (int j = 0; j < m.L; ++j) { const int skip = enc_ch[enc_ch.size() - j - 2]; set(idx++, skip + feat, skip, true); feat = skip; }
    set(idx++, feat, 64, true); set(idx++, 64, 32, true); set(idx++, 32, m.C, true);
}

DenseArgs dense(const float* x1, int k1, const Layer& ly, float* y, int M, int act) {
    DenseArgs a{}; a.x1 = x1; a.k1 = k1; a.x2 = nullptr; a.k2 = 0; a.idx2 = nullptr; a.m_per_batch = 1; a.x2_rows_per_batch = 0;
    a.W = ly.W.as<float>(); a.b = ly.has_b ? ly.b.as<float>() : nullptr; a.y = y; a.M = M; a.N = ly.out; a.act = act;
    a.wt_hi = ly.Wh.as<uint16_t>(); a.wt_lo = ly.Wl.as<uint16_t>(); a.kp = ly.kp;
    return a;
}

// round-to-nearest-even bf16 of a finite float
uint16_t bf16_rn(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
float bf16_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// slot (h, j) of a 16-wide k step that reuses an accumulator tile as an operand holds row 8 (j >> 2) + 4 h + (j & 3) of the tile
// (cdna_hip_programming.md section 3): position pos = 8 h + j of the stored block <- k offset
int kperm_src(int pos) { const int h = pos >> 3, j = pos & 7; return 8 * (j >> 2) + 4 * h + (j & 3); }

// rows k0 .. k0 + K of W [in][out] as bf16 pieces [out][K] in that k order, scaled
// (with_natural: followed by rows 0 .. K of W in natural k order, [out][K] again — the neighbour half of the attention matrix)
int permuted_pieces(Layer& ly, const float* W, int k0, int K, double scale, bool with_natural = false) {
    std::vector<uint16_t> hi((size_t)ly.out * K * (with_natural ? 2 : 1)), lo(hi.size());
    for (int x = 0; x < ly.out; ++x)
        for (int kb = 0; kb < K; kb += 16)
            for (int pos = 0; pos < 16; ++pos) {
                const float v = (float)((double)W[(size_t)(k0 + kb + kperm_src(pos)) * ly.out + x] * scale);
                const uint16_t h = bf16_rn(v);
                hi[(size_t)x * K + kb + pos] = h; lo[(size_t)x * K + kb + pos] = bf16_rn(v - bf16_f32(h));
            }
    if (with_natural)
        for (int x = 0; x < ly.out; ++x)
            for (int k = 0; k < K; ++k) {
                const float v = (float)((double)W[(size_t)k * ly.out + x] * scale);
                const uint16_t h = bf16_rn(v);
                hi[(size_t)(ly.out + x) * K + k] = h; lo[(size_t)(ly.out + x) * K + k] = bf16_rn(v - bf16_f32(h));
            }
    SSDR_TRY(ly.Ph.reserve(2 * hi.size())); SSDR_TRY(ly.Pl.reserve(2 * lo.size()));
    SSDR_HIP(hipMemcpy(ly.Ph.p, hi.data(), 2 * hi.size(), hipMemcpyHostToDevice));
    SSDR_HIP(hipMemcpy(ly.Pl.p, lo.data(), 2 * lo.size(), hipMemcpyHostToDevice));
    return SSDR_OK;
}

// LFAmlp1 (10 -> h) for lfa32_kernel: the position encoding [|d|, d, p, p_nbr] enters as the 7 inputs [|d|, d, p] (p_nbr = p - d), whose hi and
// lo pieces share the 16 k slots of one MFMA step [hi0..hi6, lo0 | lo1..lo6, 0, 0]; the weights repeat to match: fragment a carries
// the hi pieces of the 7 folded rows in the slots of BOTH input pieces, fragment b the lo pieces
int locse_fragments(Layer& ly, const float* W) {
    std::vector<uint16_t> f((size_t)ly.out * 32, 0);
    for (int c = 0; c < ly.out; ++c) {
        double w7[7];
        w7[0] = W[c];
        for (int a = 0; a < 3; ++a) { w7[1 + a] = (double)W[(size_t)(1 + a) * ly.out + c] - (double)W[(size_t)(7 + a) * ly.out + c]; w7[4 + a] = (double)W[(size_t)(4 + a) * ly.out + c] + (double)W[(size_t)(7 + a) * ly.out + c]; }
        for (int i = 0; i < 7; ++i) {
            const float v = (float)w7[i];
            const uint16_t h = bf16_rn(v), l = bf16_rn(v - bf16_f32(h));
            const int s0 = i, s1 = i == 0 ? 7 : 7 + i;       // slot of the input's hi piece, slot of its lo piece
            f[(size_t)c * 32 + s0] = h; f[(size_t)c * 32 + s1] = h;
            f[(size_t)c * 32 + 16 + s0] = l; f[(size_t)c * 32 + 16 + s1] = l;
        }
    }
    SSDR_TRY(ly.P1.reserve(2 * f.size()));
    SSDR_HIP(hipMemcpy(ly.P1.p, f.data(), 2 * f.size(), hipMemcpyHostToDevice));
    return SSDR_OK;
}

// Level 0 (d = 16, h = 8) of lfa32_kernel: two pairs of points share one 32 x 32 x 16 tile through block-diagonal weight operands, so the operand
// fragments are lane-dependent constants: table [fragment][64 lanes][8 bf16].  Lane l = (r = l & 31, half hh = l >> 5) holds k slots 8 hh + j, i.e.
// input j of point pair hh.  "T" fragments are the A operand of a transposed product (row r = (pair (r >> 2) & 1, channel (r & 3) + 4 (r >> 3)) for r < 16),
// "X" fragments the B operand of a plain one (column r = (pair r >> 4, channel r & 15); only the position channels 8..15 have weights).
template <class F> void level0_fragment(std::vector<uint16_t>& hi, std::vector<uint16_t>& lo, bool transposed, bool pos_only, F&& weight /* (input j, channel) */) {
    for (int l = 0; l < 64; ++l) {
        const int r = l & 31, hh = l >> 5;
        int pair, ch; bool used;
        if (transposed) { pair = (r >> 2) & 1; ch = (r & 3) + 4 * (r >> 3); used = r < 16; }
        else { pair = r >> 4; ch = r & 15; used = !pos_only || ch >= 8; if (pos_only) ch -= 8; }
        for (int j = 0; j < 8; ++j) {
            uint16_t h = 0, lw = 0;
            if (used && pair == hh) { const float v = weight(j, ch); h = bf16_rn(v); lw = bf16_rn(v - bf16_f32(h)); }
            hi.push_back(h); lo.push_back(lw);
        }
    }
}
int upload16(DevBuf& d, const std::vector<uint16_t>& v) {
    SSDR_TRY(d.reserve(2 * v.size()));
    SSDR_HIP(hipMemcpy(d.p, v.data(), 2 * v.size(), hipMemcpyHostToDevice));
    return SSDR_OK;
}
int level0_tables(Layer& ly, int role, const float* W, const float* b) {
    std::vector<uint16_t> hi, lo;
    if (role == 1) {            // LFAmlp1 10 -> 8: inputs [|d|, d, p, 1] (p_nbr = p - d folded into the rows, the bias rides on the constant 1)
        auto w8 = [&](int j, int c) -> float {
            if (j == 0) return W[c];
            if (j < 4) return (float)((double)W[(size_t)j * 8 + c] - (double)W[(size_t)(6 + j) * 8 + c]);
            if (j < 7) return (float)((double)W[(size_t)j * 8 + c] + (double)W[(size_t)(3 + j) * 8 + c]);
            return b[c];
        };
        std::vector<uint16_t> th, tl, xh, xl;
        level0_fragment(th, tl, true, false, w8); level0_fragment(xh, xl, false, true, w8);
        std::vector<uint16_t> all; all.insert(all.end(), th.begin(), th.end()); all.insert(all.end(), tl.begin(), tl.end());
        all.insert(all.end(), xh.begin(), xh.end()); all.insert(all.end(), xl.begin(), xl.end());
        return upload16(ly.P1, all);                    // fragments: T hi, T lo, X hi, X lo
    }
    if (role == 4) {            // LFAmlp2 8 -> 8
        auto w = [&](int j, int c) -> float { return W[(size_t)j * 8 + c]; };
        std::vector<uint16_t> xh, xl;
        level0_fragment(hi, lo, true, false, w); level0_fragment(xh, xl, false, true, w);
        hi.insert(hi.end(), xh.begin(), xh.end()); lo.insert(lo.end(), xl.begin(), xl.end());      // fragments: T, X
    } else {                    // attention 16 -> 16, x log2 e: fragment 0 = rows 8..15 (the position half), fragment 1 = rows 0..7 (the neighbour-feature half)
        for (int half = 1; half >= 0; --half) {
            auto w = [&](int j, int c) -> float { return (float)((double)W[(size_t)(8 * half + j) * 16 + c] * 1.4426950408889634); };
            level0_fragment(hi, lo, false, false, w);
        }
    }
    SSDR_TRY(upload16(ly.Ph, hi));
    return upload16(ly.Pl, lo);
}

int run_dense(const Model& m, const DenseArgs& a, hipStream_t s) { return m.prec == PREC_F32 ? launch_dense(a, s) : launch_dense_bf16(a, m.prec, s); }

}  // namespace
}  // namespace ssdr

using namespace ssdr;

extern "C" {

int ssdr_randla_create(int num_layers, const int32_t* d_out, int k_n, int num_classes, int in_dim, void** handle) {
    if (!handle || !d_out || num_layers < 1 || num_layers > 8) { set_error("randla_create: bad arguments"); return SSDR_ERR_INVALID; }
    if (k_n != 16) { set_error("randla: k_n=%d is not supported (16)", k_n); return SSDR_ERR_UNSUPPORTED; }
    if (num_classes < 1 || num_classes > 32) { set_error("randla: num_classes must be in [1,32]"); return SSDR_ERR_UNSUPPORTED; }
    for (int i = 0; i < num_layers; ++i)
        if (d_out[i] != 16 && d_out[i] != 64 && d_out[i] != 128 && d_out[i] != 256 && d_out[i] != 512) { set_error("randla: d_out[%d]=%d is not supported", i, d_out[i]); return SSDR_ERR_UNSUPPORTED; }
    Model* m = new Model();
    m->L = num_layers; m->K = k_n; m->C = num_classes; m->in_dim = in_dim;
    for (int i = 0; i < num_layers; ++i) m->d_out[i] = d_out[i];
    shapes(*m);
    *handle = m;
    return SSDR_OK;
}

int ssdr_randla_num_layers(void* handle) { return handle ? n_layers(*static_cast<Model*>(handle)) : 0; }

int ssdr_randla_layer_shape(void* handle, int layer, int* in, int* out, int* has_bias) {
    Model* m = static_cast<Model*>(handle);
    if (!m || layer < 0 || layer >= n_layers(*m)) { set_error("randla_layer_shape: bad layer"); return SSDR_ERR_INVALID; }
    if (in) *in = m->layers[layer].in;
    if (out) *out = m->layers[layer].out;
    if (has_bias) *has_bias = m->layers[layer].has_b;
    return SSDR_OK;
}

int ssdr_randla_set_layer(void* handle, int layer, const float* W, const float* b) {
    Model* m = static_cast<Model*>(handle);
    if (!m || layer < 0 || layer >= n_layers(*m) || !W) { set_error("randla_set_layer: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    Layer& ly = m->layers[layer];
    if (ly.has_b && !b) { set_error("randla_set_layer: layer %d needs a bias", layer); return SSDR_ERR_INVALID; }
    SSDR_TRY(ly.W.reserve(sizeof(float) * (size_t)ly.in * ly.out));
    SSDR_HIP(hipMemcpy(ly.W.p, W, sizeof(float) * (size_t)ly.in * ly.out, hipMemcpyHostToDevice));
    if (ly.has_b) { SSDR_TRY(ly.b.reserve(sizeof(float) * ly.out)); SSDR_HIP(hipMemcpy(ly.b.p, b, sizeof(float) * ly.out, hipMemcpyHostToDevice)); }
    if (!ly.has_b && ly.in == ly.out) {      // attention dense (d x d, no bias): the LFA kernel reads four consecutive k of one output column at once
        std::vector<float> t((size_t)ly.in * ly.out);
        for (int k = 0; k < ly.in; ++k) for (int c = 0; c < ly.out; ++c) t[(size_t)c * ly.in + k] = W[(size_t)k * ly.out + c];
        SSDR_TRY(ly.Wt.reserve(sizeof(float) * t.size()));
        SSDR_HIP(hipMemcpy(ly.Wt.p, t.data(), sizeof(float) * t.size(), hipMemcpyHostToDevice));
    }
    {   // bf16 pieces for the PREC_BF16X3 / PREC_BF16 modes
        ly.kp = (ly.in + 63) / 64 * 64;
        std::vector<uint16_t> hi((size_t)ly.out * ly.kp, 0), lo((size_t)ly.out * ly.kp, 0);
        // attention dense (no bias, square): the pieces carry W * log2(e), so the bf16 kernels' scores (and G = f W[0:h], computed from
        // the same pieces) come out in base-2 units and the softmax is a bare v_exp_f32
        const double scale = (!ly.has_b && ly.in == ly.out) ? 1.4426950408889634 : 1.0;
        for (int k = 0; k < ly.in; ++k)
            for (int c = 0; c < ly.out; ++c) {
                const float v = (float)((double)W[(size_t)k * ly.out + c] * scale);
                const uint16_t h = bf16_rn(v);
                hi[(size_t)c * ly.kp + k] = h; lo[(size_t)c * ly.kp + k] = bf16_rn(v - bf16_f32(h));
            }
        SSDR_TRY(ly.Wh.reserve(2 * hi.size())); SSDR_TRY(ly.Wl.reserve(2 * lo.size()));
        SSDR_HIP(hipMemcpy(ly.Wh.p, hi.data(), 2 * hi.size(), hipMemcpyHostToDevice));
        SSDR_HIP(hipMemcpy(ly.Wl.p, lo.data(), 2 * lo.size(), hipMemcpyHostToDevice));
    }
    if (layer >= 1 && layer < 1 + 8 * m->L) {       // operands of lfa32_kernel
        const int r = (layer - 1) % 8;
        if (m->d_out[(layer - 1) / 8] == 16 && (r == 1 || r == 2 || r == 4 || r == 5)) SSDR_TRY(level0_tables(ly, r == 5 ? 2 : r, W, b));
        else if (r == 1 && ly.in == 10) SSDR_TRY(locse_fragments(ly, W));
        else if (r == 4 && ly.in % 16 == 0) SSDR_TRY(permuted_pieces(ly, W, 0, ly.in, 1.0));
        else if ((r == 2 || r == 5) && ly.in % 32 == 0) SSDR_TRY(permuted_pieces(ly, W, ly.in / 2, ly.in / 2, 1.4426950408889634, true));
    }
    ly.set = true;
    return SSDR_OK;
}

int ssdr_randla_set_precision(void* handle, int mode) {
    Model* m = static_cast<Model*>(handle);
    if (!m || (mode != PREC_F32 && mode != PREC_BF16X3 && mode != PREC_BF16)) { set_error("randla_set_precision: mode must be 0 (f32), 1 (split bf16) or 2 (bf16)"); return SSDR_ERR_INVALID; }
    m->prec = mode;
    return SSDR_OK;
}

int ssdr_randla_set_formulation(void* handle, int tiles32) {
    Model* m = static_cast<Model*>(handle);
    if (!m) { set_error("randla_set_formulation: bad handle"); return SSDR_ERR_INVALID; }
    m->tiles32 = tiles32 ? 1 : 0;
    return SSDR_OK;
}

void ssdr_randla_destroy(void* handle) {
    Model* m = static_cast<Model*>(handle);
    if (!m) return;
    for (auto& l : m->layers) { l.W.release(); l.b.release(); l.Wt.release(); l.Wh.release(); l.Wl.release(); l.Ph.release(); l.Pl.release(); l.P1.release(); }
    for (auto& w : m->ws) w.release();
    delete m;
}

int ssdr_randla_infer_dev(void* handle, size_t B, size_t n0, const float* d_features, const float* d_xyz,
                          const int32_t* ratios, int32_t* const* d_neigh_idx, int32_t* const* d_interp_idx,
                          float* d_probs, float* d_feat32, void* stream) {
    Model* m = static_cast<Model*>(handle);
    if (!m || !d_features || !d_xyz || !ratios || !d_neigh_idx || !d_interp_idx || !d_probs || !d_feat32 || B == 0 || n0 == 0) { set_error("randla_infer: bad arguments"); return SSDR_ERR_INVALID; }
    for (auto& l : m->layers) if (!l.set) { set_error("randla_infer: not every layer has weights"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream);
    const int L = m->L, Bi = (int)B;
    std::vector<int> N(L + 1); N[0] = (int)n0;
    for (int i = 0; i < L; ++i) { if (ratios[i] <= 0) { set_error("ratio must be positive"); return SSDR_ERR_INVALID; } N[i + 1] = N[i] / ratios[i]; }
    if (N[L] <= 0) { set_error("randla_infer: tile too small for the pyramid"); return SSDR_ERR_INVALID; }

    // workspaces: per level f_pc, agg(d), agg-mlp, out(2d), sampled(2d); decoder ping-pong; fc1/fc2
    size_t need = 0;
    auto idxs = [&](int level, int which) { return level * 6 + which; };
    if (m->ws.size() < (size_t)(6 * L + 7)) m->ws.resize(6 * L + 7);
    auto buf = [&](int slot, size_t floats) -> float* { if (m->ws[slot].reserve(floats * 4) != SSDR_OK) return nullptr; (void)need; return m->ws[slot].as<float>(); };

    // fc0
    float* f0 = buf(6 * L + 0, B * N[0] * 8); if (!f0) return SSDR_ERR_HIP;
    const DenseArgs a_fc0 = dense(d_features, m->in_dim, m->layers[0], f0, Bi * N[0], 1);
    // (launched with level 0's mlp1: the thin 6 -> 8 -> 8 pair is one pass over the rows)
    if (L < 1) { set_error("randla_infer: no encoder level"); return SSDR_ERR_INVALID; }
    const float* f = f0; int d_in = 8;
    std::vector<const float*> enc; std::vector<int> enc_ch, enc_n;
    for (int i = 0; i < L; ++i) {
        const int d = m->d_out[i], h = d / 2, base = 1 + 8 * i, n = N[i];
        const size_t rows = B * (size_t)n;
        float* f_pc = buf(idxs(i, 0), rows * h); float* agg = buf(idxs(i, 1), rows * d); float* aggm = buf(idxs(i, 2), rows * d);
        float* out = buf(idxs(i, 3), rows * 2 * d); float* samp = buf(idxs(i, 4), B * (size_t)N[i + 1] * 2 * d);
        if (!f_pc || !agg || !aggm || !out || !samp) return SSDR_ERR_HIP;
        const bool use32 = m->prec != PREC_F32 && m->tiles32 != 0;    // 32 x 32-tile formulation (randla_lfa32.hip), every level
        // level 0 of the 32 x 32 formulation gathers from ONE table [x y z 0 | f0..f7 | -] per point (randla_lfa32.hip): mlp1 and the first attention
        // mlp write their 8 channels into its rows, the first one the coordinates as well
        float* tab0 = nullptr;
        if (use32 && d == 16) { tab0 = buf(6 * L + 5, rows * 16); if (!tab0) return SSDR_ERR_HIP; }
        {
            DenseArgs a1 = dense(f, d_in, m->layers[base + 0], tab0 ? tab0 + 4 : f_pc, (int)rows, 1);
            if (tab0) { a1.ldy = 16; a1.xyz = d_xyz; a1.xyz_batch_stride = n0 * 3; a1.xyz_rows_per_batch = n; }
            if (i == 0) {
                const int fused = launch_dense_rows2(a_fc0, a1, s);
                if (fused == SSDR_ERR_UNSUPPORTED) { SSDR_TRY(run_dense(*m, a_fc0, s)); SSDR_TRY(run_dense(*m, a1, s)); }
                else SSDR_TRY(fused);
            } else SSDR_TRY(run_dense(*m, a1, s));                                                           // mlp1
        }
        LfaArgs la{}; la.xyz = d_xyz; la.xyz_batch_stride = n0 * 3; la.neigh = d_neigh_idx[i]; la.n = n;
        la.w_l1 = m->layers[base + 1].W.as<float>(); la.b_l1 = m->layers[base + 1].b.as<float>();
        la.w_l2 = m->layers[base + 4].W.as<float>(); la.b_l2 = m->layers[base + 4].b.as<float>();
        la.fin = f_pc; la.w_fc = m->layers[base + 2].W.as<float>(); la.w_fc_t = m->layers[base + 2].Wt.as<float>(); la.out = agg;
        la.fc_hi = m->layers[base + 2].Wh.as<uint16_t>(); la.fc_lo = m->layers[base + 2].Wl.as<uint16_t>();
        la.l2_hi = m->layers[base + 4].Wh.as<uint16_t>(); la.l2_lo = m->layers[base + 4].Wl.as<uint16_t>(); la.kp2 = m->layers[base + 4].kp;
        const bool lfa16 = m->prec != PREC_F32 && d >= 64;         // the 16 x 16-tile bf16 kernels; their d = 16 level runs on the exact-f32 kernel in every mode
        float* gbuf = nullptr;
        if (d >= 64 && !(use32 && d <= 64)) {      // neighbour half of the attention product once per point (the 32 x 32 formulation multiplies it in the kernel up to d = 64): G = f_pc * W[0:h]  (rows 0..h-1 of the [d][d] weights)
            gbuf = buf(idxs(i, 5), rows * d); if (!gbuf) return SSDR_ERR_HIP;
            DenseArgs ga{}; ga.x1 = f_pc; ga.k1 = h; ga.W = la.w_fc; ga.b = nullptr; ga.y = gbuf; ga.M = (int)rows; ga.N = d; ga.act = 0; ga.m_per_batch = 1;
            ga.wt_hi = la.fc_hi; ga.wt_lo = la.fc_lo; ga.kp = m->layers[base + 2].kp;
            SSDR_TRY(run_dense(*m, ga, s));
        }
        la.g = gbuf;
        // 32 x 32-tile formulation (randla_lfa32.hip) where it has an instantiation; ssdr_randla_set_formulation(0) keeps the 16 x 16-tile kernels
        Lfa32Args l32{}; l32.xyz = d_xyz; l32.xyz_batch_stride = n0 * 3; l32.neigh = d_neigh_idx[i]; l32.n = n; l32.g = gbuf; l32.out = agg;
        l32.w1p = m->layers[base + 1].P1.as<uint16_t>(); l32.b1 = la.b_l1;
        l32.w2_hi = m->layers[base + 4].Ph.as<uint16_t>(); l32.w2_lo = m->layers[base + 4].Pl.as<uint16_t>(); l32.b2 = la.b_l2;
        auto run_lfa = [&](bool second, const float* fin, const Layer& fc) -> int {
            if (use32) {
                l32.fin = fin; l32.fc_hi = fc.Ph.as<uint16_t>(); l32.fc_lo = fc.Pl.as<uint16_t>();
                const int rc = launch_lfa32(d, l32, second, Bi, m->prec, s);
                if (rc != SSDR_ERR_UNSUPPORTED) return rc;
                // the buffers above were laid out for the 32 x 32 kernels (level 0: the features live in tab0 only; d <= 64: no G table): the
                // 16 x 16 kernels cannot run on them
                if (tab0 || !gbuf) { set_error("randla: the 32 x 32-tile attention kernel refused d = %d and the level's buffers were laid out for it", d); return SSDR_ERR_INTERNAL; }
            }
            return lfa16 ? launch_lfa_bf16(d, la, second, Bi, m->prec, s) : launch_lfa(d, la, second, Bi, s);
        };
        SSDR_TRY(run_lfa(false, tab0 ? tab0 : f_pc, m->layers[base + 2]));                                                 // LocSE + att pool 1
        {
            DenseArgs a3 = dense(agg, d, m->layers[base + 3], tab0 ? tab0 + 4 : aggm, (int)rows, 1);     // att1 mlp d->h
            if (tab0) a3.ldy = 16;
            SSDR_TRY(run_dense(*m, a3, s));
        }
        la.fin = aggm; la.w_fc = m->layers[base + 5].W.as<float>(); la.w_fc_t = m->layers[base + 5].Wt.as<float>(); la.out = agg;
        la.fc_hi = m->layers[base + 5].Wh.as<uint16_t>(); la.fc_lo = m->layers[base + 5].Wl.as<uint16_t>();
        if (gbuf) {
            DenseArgs ga{}; ga.x1 = aggm; ga.k1 = h; ga.W = la.w_fc; ga.b = nullptr; ga.y = gbuf; ga.M = (int)rows; ga.N = d; ga.act = 0; ga.m_per_batch = 1;
            ga.wt_hi = la.fc_hi; ga.wt_lo = la.fc_lo; ga.kp = m->layers[base + 5].kp;
            SSDR_TRY(run_dense(*m, ga, s));
        }
        SSDR_TRY(run_lfa(true, tab0 ? tab0 : aggm, m->layers[base + 5]));                                                  // LocSE2 + att pool 2
        DenseArgs a6 = dense(agg, d, m->layers[base + 6], aggm, (int)rows, 1);                               // att2 mlp d->d
        DenseArgs r = dense(aggm, d, m->layers[base + 7], out, (int)rows, 1);                               // lrelu(mlp2 + shortcut)
        r.x2 = f; r.k2 = d_in;
        // the two in one launch where the rows are many and narrow (d <= 64: the intermediate stays in LDS)
        int chained = m->prec != PREC_F32 ? launch_dense_chain(a6, r, m->prec, s) : SSDR_ERR_UNSUPPORTED;
        if (chained == SSDR_ERR_UNSUPPORTED) { SSDR_TRY(run_dense(*m, a6, s)); SSDR_TRY(run_dense(*m, r, s)); }
        else SSDR_TRY(chained);
        SSDR_TRY(launch_gather_max(out, d_neigh_idx[i], n, N[i + 1], n, 2 * d, samp, Bi, s));               // random_sample
        if (i == 0) { enc.push_back(out); enc_ch.push_back(2 * d); enc_n.push_back(n); }
        enc.push_back(samp); enc_ch.push_back(2 * d); enc_n.push_back(N[i + 1]);
        f = samp; d_in = 2 * d;
    }
    int li = 1 + 8 * L;
    float* dec = buf(6 * L + 1, B * (size_t)N[L] * d_in); if (!dec) return SSDR_ERR_HIP;
    SSDR_TRY(run_dense(*m, dense(enc.back(), d_in, m->layers[li++], dec, Bi * N[L], 1), s));                  // decoder_0
    const float* feat = dec; int feat_c = d_in, feat_n = N[L];
    // the last decoder layer (32 + 32 -> 32 in the reference's configurations) rides in front of the fused tail in the bf16 modes
    const Layer& l1 = m->layers[li + L]; const Layer& l2 = m->layers[li + L + 1]; const Layer& fc = m->layers[li + L + 2];
    const bool tail16 = m->prec != PREC_F32 && l1.in == 32 && l1.out == 64 && l2.out == 32 && l1.has_b && l2.has_b && fc.has_b && (m->C == 13 || m->C == 8);
    TailArgs t{};
    for (int j = 0; j < L; ++j) {
        const int e = (int)enc.size() - j - 2;
        const int skip_c = enc_ch[e], n = enc_n[e];
        const Layer& ld = m->layers[li];
        if (j == L - 1 && tail16 && skip_c == 32 && feat_c == 32 && ld.has_b) {
            t.skip = enc[e]; t.up = feat; t.idx = d_interp_idx[L - 1 - j]; t.m_per_batch = n; t.up_rows_per_batch = feat_n;
            t.wdh = ld.Wh.as<uint16_t>(); t.wdl = ld.Wl.as<uint16_t>(); t.kpd = ld.kp; t.bd = ld.b.as<float>();
            ++li; feat = nullptr; feat_c = skip_c; feat_n = n;
            break;
        }
        float* y = buf(6 * L + 2 + (j & 1), B * (size_t)n * skip_c); if (!y) return SSDR_ERR_HIP;
        DenseArgs a = dense(enc[e], skip_c, m->layers[li++], y, Bi * n, 1);
        a.x2 = feat; a.k2 = feat_c; a.idx2 = d_interp_idx[L - 1 - j]; a.m_per_batch = n; a.x2_rows_per_batch = feat_n;
        SSDR_TRY(run_dense(*m, a, s));
        feat = y; feat_c = skip_c; feat_n = n;
    }
    int fused = SSDR_ERR_UNSUPPORTED;
    if (tail16 && feat_c == 32) {     // fc1 + fc2 + fc + softmax (+ the decoder layer) on the bf16 matrix cores
        t.x = feat; t.M = Bi * N[0]; t.C = m->C;
        t.w1h = l1.Wh.as<uint16_t>(); t.w1l = l1.Wl.as<uint16_t>(); t.kp1 = l1.kp; t.b1 = l1.b.as<float>();
        t.w2h = l2.Wh.as<uint16_t>(); t.w2l = l2.Wl.as<uint16_t>(); t.kp2 = l2.kp; t.b2 = l2.b.as<float>();
        t.w3h = fc.Wh.as<uint16_t>(); t.w3l = fc.Wl.as<uint16_t>(); t.kp3 = fc.kp; t.b3 = fc.b.as<float>();
        t.feat32 = d_feat32; t.probs = d_probs;
        fused = launch_tail_bf16(t, m->prec, s);
        if (fused == SSDR_ERR_UNSUPPORTED && t.skip) { set_error("randla: the fused decoder + tail kernel refused its arguments"); return SSDR_ERR_INTERNAL; }
    }
    if (fused == SSDR_ERR_UNSUPPORTED && feat_c == 32 && l1.out == 64 && l2.out == 32 && l1.has_b && l2.has_b)          // fc1 + fc2 + fc + softmax in one pass
        fused = launch_tail(feat, l1.W.as<float>(), l1.b.as<float>(), l2.W.as<float>(), l2.b.as<float>(), fc.W.as<float>(), fc.b.as<float>(),
                            Bi * N[0], m->C, d_feat32, d_probs, s);
    if (fused == SSDR_ERR_UNSUPPORTED) {
        float* f1 = buf(6 * L + 4, B * (size_t)N[0] * 64); if (!f1) return SSDR_ERR_HIP;
        SSDR_TRY(run_dense(*m, dense(feat, feat_c, l1, f1, Bi * N[0], 1), s));                                 // fc1
        SSDR_TRY(run_dense(*m, dense(f1, 64, l2, d_feat32, Bi * N[0], 1), s));                                 // fc2 = last_second_features
        SSDR_TRY(launch_head(d_feat32, fc.W.as<float>(), fc.b.as<float>(), Bi * N[0], m->C, d_probs, s));     // fc + softmax
    } else SSDR_TRY(fused);
    return SSDR_OK;
}

}
