// Selection stage of one active-learning round for gfx950: per-point uncertainty, per-superpoint statistics,
// candidate features, chamfer / adjacency / feature propagation per cloud, farthest-point and k-center sampling.
//
// Reference (S3 = /root/reference/SSDR_AL_s3dis/): S3/sampler2.py:12-47,102-115,262-266,313-342,612-640;
// S3/fps_gcn_cpu.py:12-178; S3/kcenterGreedy.py:60-128.  The reference does all of this in NumPy / sklearn on
// the host, in float64 for everything after the network.  The kernels keep float64 there and, where a NumPy
// reduction decides a discrete outcome (the arg-max chain of FPS), reproduce NumPy's pairwise summation order
// (np_pairwise below) so that identical inputs give the identical index sequence.
//
// Superpoints are CSR: sp_off[S+1] into sp_pts[T] (point ids), the same information as the reference's
// pickled `components` object array (S3/partition/compute_superpoint.py:63-68).
#include <optional>
#include "ssdr_internal.hpp"
#include <map>
#include "block_prims.hpp"
#include "select_chamfer.hpp"

namespace ssdr {
namespace {

// NumPy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum) over get(i), i in [0,n)
template <class T, class Get>
__device__ T np_pairwise(Get get, int n) {
    // iterative version of the recursion: blocks are produced left to right; partial sums are combined exactly
    // like the call tree sum(a[:n2]) + sum(a[n2:]) with n2 = n/2 - (n/2)%8.
    struct Fr { int lo, n; int state; T left; };
    Fr st[24]; int sp = 0; T ret = T(0);
    st[0] = Fr{0, n, 0, T(0)};
    while (sp >= 0) {
        Fr& f = st[sp];
        if (f.n <= 128) {
            T res;
            if (f.n < 8) { res = T(0); for (int i = 0; i < f.n; ++i) res += get(f.lo + i); }
            else {
                T r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = get(f.lo + j);
                int i = 8;
                for (; i < f.n - (f.n % 8); i += 8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) r[j] += get(f.lo + i + j);
                }
                res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (; i < f.n; ++i) res += get(f.lo + i);
            }
            ret = res; --sp;
        } else if (f.state == 0) {
            int n2 = f.n / 2; n2 -= n2 % 8;
            f.state = 1; st[sp + 1] = Fr{f.lo, n2, 0, T(0)}; ++sp;
        } else if (f.state == 1) {
            int n2 = f.n / 2; n2 -= n2 % 8;
            f.left = ret; f.state = 2; st[sp + 1] = Fr{f.lo + n2, f.n - n2, 0, T(0)}; ++sp;
        } else { ret = f.left + ret; --sp; }
    }
    return ret;
}

// Same summation order for a compile-time length 8 <= D <= 128 that is a multiple of 8 (fully unrolled: the 32-d
// feature distance of farthest_features_sample).
template <int D, class Get>
__device__ __forceinline__ double np_pairwise_fixed(Get get) {
    static_assert(D >= 8 && D <= 128 && D % 8 == 0, "np_pairwise_fixed");
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = get(j);
#pragma unroll
    for (int i = 8; i < D; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += get(i + j);
    }
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

// The same order for one block of 8 <= n <= 128 terms with the eight accumulators on eight lanes of a wave (every lane of the wave calls it with uniform
// n; result on lane 0): lane k sums the terms k, k + 8, ... of the multiple-of-eight part, the accumulators are combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
// through lane exchanges (fp addition is commutative, so lane 0's a + b is the reference's), the leftover terms are added by lane 0 in order.
template <class T, class Get>
__device__ __forceinline__ T np_pairwise_w8(Get get, int n, int lane) {
    const int n8 = n - (n % 8);
    T r = T(0);
    if (lane < 8) { r = get(lane); for (int i = 8; i < n8; i += 8) r += get(i + lane); }
    auto xchg = [&](T v, int m) -> T {
        if constexpr (sizeof(T) == 8) {
            const long long b = __double_as_longlong((double)v);
            const unsigned lo = __shfl_xor((unsigned)b, m), hi = __shfl_xor((unsigned)(b >> 32), m);
            return (T)__longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        } else return (T)__shfl_xor((float)v, m);
    };
    r = r + xchg(r, 1); r = r + xchg(r, 2); r = r + xchg(r, 4);
    if (lane == 0) for (int i = n8; i < n; ++i) r += get(i);
    return r;
}

// The whole recursion with every block of at most 128 terms summed by np_pairwise_w8 (uniform n over the wave; result on lane 0): the call tree
// sum(a[:n2]) + sum(a[n2:]) is walked by all lanes alike, only its leaves use the eight lanes.  A superpoint of 185 points is two such blocks of ~12 dependent
// additions per lane instead of 185 on one lane.
template <class T, class Get>
__device__ T np_pairwise_wave(Get get, int n, int lane) {
    struct Fr { int lo, n; int state; T left; };
    Fr st[24]; int sp = 0; T ret = T(0);
    st[0] = Fr{0, n, 0, T(0)};
    while (sp >= 0) {
        Fr& f = st[sp];
        if (f.n <= 128) {
            T res = T(0);
            const int lo = f.lo;
            if (f.n < 8) { if (lane == 0) for (int i = 0; i < f.n; ++i) res += get(lo + i); }
            else res = np_pairwise_w8<T>([&](int i) { return get(lo + i); }, f.n, lane);
            ret = res; --sp;
        } else if (f.state == 0) {
            int n2 = f.n / 2; n2 -= n2 % 8;
            f.state = 1; st[sp + 1] = Fr{f.lo, n2, 0, T(0)}; ++sp;
        } else if (f.state == 1) {
            int n2 = f.n / 2; n2 -= n2 % 8;
            f.left = ret; f.state = 2; st[sp + 1] = Fr{f.lo + n2, f.n - n2, 0, T(0)}; ++sp;
        } else { ret = f.left + ret; --sp; }
    }
    return ret;
}

// The same recursion for MORE terms than a wave can stage at once: the call tree is walked down to nodes of at most `cap` terms; such a node's terms are staged
// by all lanes (stage(lo, n): terms lo .. lo + n - 1 into slots 0 .. n - 1) and summed by np_pairwise_wave over the staged values — exactly the sub-call
// sum(a[lo:lo+n]) of the reference's recursion.  Two sums over the same members (WetSU's) share the walk and the staging.  Results on lane 0.
template <class T, class Stage, class GetA, class GetB>
__device__ void np_pairwise_wave_chunked2(Stage stage, GetA get_a, GetB get_b, bool two, int n, int lane, int cap, T& out_a, T& out_b) {
    struct Fr { int lo, n; int state; T left_a, left_b; };
    Fr st[24]; int sp = 0; T ra = T(0), rb = T(0);
    st[0] = Fr{0, n, 0, T(0), T(0)};
    while (sp >= 0) {
        Fr& f = st[sp];
        if (f.n <= cap) {
            stage(f.lo, f.n);
            if (f.n >= 8) { ra = np_pairwise_wave<T>(get_a, f.n, lane); if (two) rb = np_pairwise_wave<T>(get_b, f.n, lane); }
            else { ra = np_pairwise<T>(get_a, f.n); if (two) rb = np_pairwise<T>(get_b, f.n); }
            --sp;
        } else if (f.state == 0) {
            int n2 = f.n / 2; n2 -= n2 % 8;
            f.state = 1; st[sp + 1] = Fr{f.lo, n2, 0, T(0), T(0)}; ++sp;
        } else if (f.state == 1) {
            int n2 = f.n / 2; n2 -= n2 % 8;
            f.left_a = ra; f.left_b = rb; f.state = 2; st[sp + 1] = Fr{f.lo + n2, f.n - n2, 0, T(0), T(0)}; ++sp;
        } else { ra = f.left_a + ra; rb = f.left_b + rb; --sp; }
    }
    out_a = ra; out_b = rb;
}

// ---- U1: compute_point_uncertainty (sampler2.py:28-47) + argmax class (:602) ------------------------------
// (round 4: a workgroup's 256 rows of C <= 32 probabilities are one contiguous run: loaded coalesced into LDS, read back one row per lane — a lane reading its
// own 52-byte row straight from memory made every load instruction touch 64 lines)
__global__ __launch_bounds__(256) void sel_point_unc(const float* __restrict__ prob, int n, int C, int mode, float* unc, int* cls) {
    __shared__ float s_p[256 * 32];
    for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {          // uniform over the workgroup
        const int rows = min(256, n - i0);
        const bool staged = C <= 32;
        if (staged) {
            __syncthreads();
            const float* src = prob + (size_t)i0 * C;
            for (int e = threadIdx.x; e < rows * C; e += 256) s_p[e] = src[e];
            __syncthreads();
        }
        const int i = i0 + (int)threadIdx.x;
        if (i >= n) continue;
        const float* p = staged ? s_p + (size_t)threadIdx.x * C : prob + (size_t)i * C;
        float best = p[0], second = -1.f; int bi = 0;
        for (int c = 1; c < C; ++c) {
            const float v = p[c];
            if (v > best) { second = best; best = v; bi = c; }
            else if (v > second) second = v;
        }
        float u;
        if (mode == 0) u = 1.0f - best;                              // lc
        else if (mode == 2) u = second / best;                       // sb: sorted[-2] / sorted[-1]
        else {                                                       // entropy, float32 like np.sum over 13 classes
            u = -1.f * np_pairwise<float>([&](int c) { const float v = p[c]; const float k = log2f(v); return v * (__builtin_isinf(k) ? 0.f : k); }, C);
        }
        unc[i] = u; cls[i] = bi;
    }
}

// ---- U2: per-superpoint loop of TSampler.prediction (sampler2.py:612-626) ---------------------------------
// mode 0 mean, 1 sum_weight, 2 WetSU.  One lane per superpoint (sums must follow NumPy's order).
__global__ __launch_bounds__(256) void sel_region_stats(const float* __restrict__ unc, const int* __restrict__ cls,
                                                        const int* __restrict__ sp_off, const int* __restrict__ sp_pts, int S, int C, int mode,
                                                        double* region_unc, int* dom, int* dom_cnt) {
    for (int s = blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256) {
        const int lo = sp_off[s], n = sp_off[s + 1] - lo;
        if (n <= 0) { region_unc[s] = 0.0; dom[s] = 0; dom_cnt[s] = 0; continue; }
        int h[32];
        for (int c = 0; c < 32; ++c) h[c] = 0;
        for (int j = 0; j < n; ++j) { const int c = cls[sp_pts[lo + j]]; if (c >= 0 && c < 32) h[c]++; }
        int d = 0;
        for (int c = 1; c < C; ++c) if (h[c] > h[d]) d = c;           // np.argmax: first maximum
        dom[s] = d; dom_cnt[s] = h[d];
        double r;
        if (mode == 0) {
            const float sum = np_pairwise<float>([&](int j) { return unc[sp_pts[lo + j]]; }, n);
            r = (double)(float)((double)sum / (double)n);
        } else if (mode == 1) {                                      // weights_percentage (:92-100) * uncertainty
            r = np_pairwise<double>([&](int j) { const int p = sp_pts[lo + j]; return ((double)h[cls[p]] / (double)n) * (double)unc[p]; }, n);
        } else {                                                     // WetSU (:19-26)
            const double a = np_pairwise<double>([&](int j) { const int p = sp_pts[lo + j]; return (double)unc[p] * (cls[p] == d ? 1.0 : 0.0); }, n);
            const double b = np_pairwise<double>([&](int j) { const int p = sp_pts[lo + j]; return (double)unc[p] * (1.0 - (cls[p] == d ? 1.0 : 0.0)); }, n);
            r = a - b;
        }
        region_unc[s] = r;
    }
}

// Wave-per-superpoint form of the same loop.  The one-lane form chased sp_pts -> (class, uncertainty) three times per member with 8 k lanes
// on the whole chip (0.37 ms at 1 % of the HBM rate).  Here the 64 lanes of a wave fetch the members' classes and uncertainties together
// into LDS (RS_CAP members per wave; a larger superpoint takes the one-lane routine), count the class histogram there, and lane 0 then runs
// the SAME summation routines (NumPy's pairwise order) over the staged values: same operations in the same order, same result.
constexpr int RS_CAP = 1024, NP_BUFSIZE = 8192;
__global__ __launch_bounds__(256) void sel_region_stats_w(const float* __restrict__ unc, const int* __restrict__ cls,
                                                          const int* __restrict__ sp_off, const int* __restrict__ sp_pts, int S, int C, int mode,
                                                          double* region_unc, int* dom, int* dom_cnt) {
    __shared__ float s_u[4][RS_CAP];
    __shared__ int s_c[4][RS_CAP];
    __shared__ int s_h[4][32];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int s = blockIdx.x * 4 + w; s < S; s += gridDim.x * 4) {
        const int lo = sp_off[s], n = sp_off[s + 1] - lo;
        if (n <= 0) { if (lane == 0) { region_unc[s] = 0.0; dom[s] = 0; dom_cnt[s] = 0; } continue; }
        if (lane < 32) s_h[w][lane] = 0;
        wave_sync();
        const bool staged = n <= RS_CAP;
        // (a wave runs in lockstep on the hardware; the block-level barrier is not available here because waves take different trip counts)
        for (int j = lane; j < n; j += 64) {
            const int p = sp_pts[lo + j];
            const int c = cls[p];
            if (staged) { s_c[w][j] = c; s_u[w][j] = unc[p]; }
            if (c >= 0 && c < 32) atomicAdd(&s_h[w][c], 1);
        }
        wave_sync();
        const int* h = s_h[w];
        int d = 0;
        for (int c = 1; c < C; ++c) if (h[c] > h[d]) d = c;           // np.argmax: first maximum (every lane: uniform)
        if (lane == 0) { dom[s] = d; dom_cnt[s] = h[d]; }
        auto U = [&](int j) { return staged ? s_u[w][j] : unc[sp_pts[lo + j]]; };
        auto K = [&](int j) { return staged ? s_c[w][j] : cls[sp_pts[lo + j]]; };
        const bool w8 = staged && n >= 8;                             // the blocks of NumPy's pairwise sum with their eight accumulators on eight lanes
        double r = 0.0;
        if (!staged) {
            // more members than the wave stages at once (a floor or a wall of a real partition: thousands of points; until round 6 lane 0 chased them one
            // dependent load at a time — 1.8 ms for a 3 600-point region, tools/sp_probe.py): the recursion's nodes of at most RS_CAP terms are staged one
            // after the other by all lanes and summed as above
            // ... and NumPy's reduction hands its inner loop at most NP_BUFSIZE = 8192 elements at a time (the ufunc buffer, whatever the dtype): a sum over more
            // terms is pairwise(first 8192) + pairwise(next 8192) + ..., accumulated left to right — not one recursion over everything (measured against np.sum
            // for 4 099 .. 40 000 terms: the chunked order 20 / 20, the single recursion 11-17 / 20; found when this path met a 8 193-point region)
            int base = 0;
            auto stage = [&](int l0, int cnt) {
                wave_sync();                                      // the previous node's reads are done
                for (int j = lane; j < cnt; j += 64) { const int p = sp_pts[lo + base + l0 + j]; s_c[w][j] = cls[p]; s_u[w][j] = unc[p]; }
                wave_sync();
            };
            float fa = 0.f; double da = 0.0, db = 0.0;
            for (; base < n; base += NP_BUFSIZE) {
                const int cn = min(NP_BUFSIZE, n - base);
                if (mode == 0) {
                    float sa = 0.f, sb = 0.f;
                    np_pairwise_wave_chunked2<float>(stage, [&](int j) { return s_u[w][j]; }, [&](int j) { return s_u[w][j]; }, false, cn, lane, RS_CAP, sa, sb);
                    fa = base == 0 ? sa : fa + sa;
                } else if (mode == 1) {
                    double sa = 0.0, sb = 0.0;
                    auto term = [&](int j) { return ((double)h[s_c[w][j]] / (double)n) * (double)s_u[w][j]; };
                    np_pairwise_wave_chunked2<double>(stage, term, term, false, cn, lane, RS_CAP, sa, sb);
                    da = base == 0 ? sa : da + sa;
                } else {
                    double sa = 0.0, sb = 0.0;
                    np_pairwise_wave_chunked2<double>(stage, [&](int j) { return (double)s_u[w][j] * (s_c[w][j] == d ? 1.0 : 0.0); },
                                                      [&](int j) { return (double)s_u[w][j] * (1.0 - (s_c[w][j] == d ? 1.0 : 0.0)); }, true, cn, lane, RS_CAP, sa, sb);
                    da = base == 0 ? sa : da + sa; db = base == 0 ? sb : db + sb;
                }
            }
            r = mode == 0 ? (double)(float)((double)fa / (double)n) : mode == 1 ? da : da - db;
        } else if (mode == 0) {
            float sum = 0.f;
            if (w8) sum = np_pairwise_wave<float>([&](int j) { return U(j); }, n, lane);
            else if (lane == 0) sum = np_pairwise<float>([&](int j) { return U(j); }, n);
            r = (double)(float)((double)sum / (double)n);
        } else if (mode == 1) {                                      // weights_percentage (:92-100) * uncertainty
            auto term = [&](int j) { return ((double)h[K(j)] / (double)n) * (double)U(j); };
            if (w8) r = np_pairwise_wave<double>(term, n, lane);
            else if (lane == 0) r = np_pairwise<double>(term, n);
        } else {                                                     // WetSU (:19-26)
            auto ta = [&](int j) { return (double)U(j) * (K(j) == d ? 1.0 : 0.0); };
            auto tb = [&](int j) { return (double)U(j) * (1.0 - (K(j) == d ? 1.0 : 0.0)); };
            double a = 0.0, b = 0.0;
            if (w8) { a = np_pairwise_wave<double>(ta, n, lane); b = np_pairwise_wave<double>(tb, n, lane); }
            else if (lane == 0) { a = np_pairwise<double>(ta, n); b = np_pairwise<double>(tb, n); }
            r = a - b;
        }
        if (lane == 0) region_unc[s] = r;
        wave_sync();
    }
}

// ---- D1: dominant ground-truth label + purity (sampler2.py:102-106 via oracle_labeling :127-144) ------------
// One WAVE per superpoint (a lane per superpoint walked its ~185 members one dependent load at a time with a 64-entry private histogram: 0.18 ms
// for the bench's 7000 regions, more than the whole scoring stage): members dealt to the lanes, histogram in LDS, first maximum like np.argmax.
__global__ __launch_bounds__(256) void sel_dominant_label(const int* __restrict__ labels, const int* __restrict__ sp_off, const int* __restrict__ sp_pts,
                                                          int S, int num_labels, int* out_label, double* out_purity, int* status) {
    __shared__ int s_h[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nl = min(num_labels, 64);
    for (int s = blockIdx.x * 4 + w; s < S; s += gridDim.x * 4) {
        const int lo = sp_off[s], n = sp_off[s + 1] - lo;
        s_h[w][lane] = 0;
        wave_sync();
        bool bad = false;
        for (int j0 = lane; j0 < n; j0 += 256) {          // four members' dependent loads (member, label) in flight per lane
            int p[4], c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = sp_pts[lo + min(j0 + 64 * u, n - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = labels[p[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (j0 + 64 * u < n) { if (c[u] >= 0 && c[u] < nl) atomicAdd(&s_h[w][c[u]], 1); else bad = true; }
        }
        if (bad) atomicOr(status, 1);
        wave_sync();
        // (count, lowest class first) as one key: the wave's maximum is np.argmax's first maximum
        int key = lane < nl ? (s_h[w][lane] << 6) | (63 - lane) : -1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) key = max(key, __shfl_xor(key, o));
        if (lane == 0) { const int d = 63 - (key & 63); out_label[s] = d; out_purity[s] = n > 0 ? (double)(key >> 6) / (double)n : 0.0; }
        wave_sync();
    }
}

// ---- clsbal (sampler2.py:262-266): u *= exp(-freq(dominant class among candidates + already selected)) -------
// `skip` != 0: the region is not in the population (labelled, or below min_size): prediction() appends only unlabelled regions to region_class
// (sampler2.py:612-627), so only those enter list_class
__global__ __launch_bounds__(256) void sel_class_hist(const int* __restrict__ region_class, int S, const unsigned char* __restrict__ skip, const int* __restrict__ extra, int n_extra, int* hist) {
    __shared__ int s_h[64];          // thousands of regions on a dozen classes: count in LDS, one global add per class and workgroup
    if (threadIdx.x < 64) s_h[threadIdx.x] = 0;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < S + n_extra; i += gridDim.x * 256) {
        if (i < S && skip && skip[i]) continue;
        const int c = i < S ? region_class[i] : extra[i - S];
        if (c >= 0 && c < 64) atomicAdd(&s_h[c], 1);
    }
    __syncthreads();
    if (threadIdx.x < 64 && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}
__global__ __launch_bounds__(256) void sel_clsbal(const int* __restrict__ region_class, int S, int total, const int* __restrict__ hist, double* region_unc) {
    if (total < 0) {                 // len(list_class) = what the histogram counted (a population behind a mask: the count is the device's)
        total = 0;
        for (int c = 0; c < 64; ++c) total += hist[c];
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < S; i += gridDim.x * 256) {
        const double w = (double)hist[region_class[i]] / (double)total;
        region_unc[i] = region_unc[i] * exp(-w);
    }
}

// ---- ranking: argsort(-u) (sampler2.py:640), ties by ascending index --------------------------------------------
__global__ __launch_bounds__(256) void sel_rank_keys(const double* __restrict__ u, int S, uint64_t* keys, uint32_t* vals) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < S; i += gridDim.x * 256) {
        unsigned long long b = (unsigned long long)__double_as_longlong(-u[i]);
        b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);            // order-preserving map of IEEE doubles to u64
        keys[i] = b; vals[i] = (uint32_t)i;
    }
}

// Ranking of up to RANK_SMALL regions by COUNTING, spread over the chip: the rank of region i is the number of regions whose (key, index) pair is
// smaller — descending uncertainty, equal values by ascending index: the order the stable radix sort of (key, index) pairs gives.  Every workgroup
// stages all keys in LDS and ranks 32 regions, eight lanes per region (S^2 / 8 comparisons per lane group: 51 M in all for the bench's 7149 regions).
// The segmented radix sorter takes ~14 launch-bound launches for a few thousand keys (0.17 ms in front of every selection), a bitonic network in one
// workgroup 0.1 ms (one CU's vector rate); this is one launch of ~10 us.
constexpr int RANK_SMALL = 8192;
__global__ __launch_bounds__(256) void sel_rank_count(const double* __restrict__ u, int S, int* __restrict__ sorted_inds) {
    __shared__ unsigned long long s_k[RANK_SMALL];
    const int tid = threadIdx.x;
    for (int i = tid; i < S; i += 256) {
        unsigned long long b = (unsigned long long)__double_as_longlong(-u[i]);
        s_k[i] = (b >> 63) ? ~b : (b | 0x8000000000000000ull);          // order-preserving map of IEEE doubles to u64 (as sel_rank_keys)
    }
    __syncthreads();
    const int i = blockIdx.x * 32 + (tid >> 3), part = tid & 7;
    const bool live = i < S;
    const unsigned long long ki = s_k[live ? i : 0];
    int cnt = 0;
    if (live) {
        // (kj, j) < (ki, i)  <=>  kj < ki + (j < i): one 64-bit add and compare per pair (ki + 1 does not wrap: only a NaN maps to all ones); eight
        // LDS reads in flight
        int j = part;
        for (; j + 56 < S; j += 64) {
            unsigned long long kj[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) kj[t] = s_k[j + 8 * t];
#pragma unroll
            for (int t = 0; t < 8; ++t) cnt += kj[t] < ki + (unsigned long long)(j + 8 * t < i);
        }
        for (; j < S; j += 8) cnt += s_k[j] < ki + (unsigned long long)(j < i);
    }
    cnt += __shfl_xor(cnt, 1); cnt += __shfl_xor(cnt, 2); cnt += __shfl_xor(cnt, 4);
    if (live && part == 0) sorted_inds[cnt] = i;
}

// ---- U3: compute_features (sampler2.py:333,339): float32 row-sequential mean over the dominant-class members ----
__global__ __launch_bounds__(256) void sel_segment_mean(const float* __restrict__ feat, int D, const int* __restrict__ cls_pred, const int* __restrict__ dom,
                                                        const int* __restrict__ sp_off, const int* __restrict__ sp_pts,
                                                        const int* __restrict__ sel, int nsel, float* out, const int* __restrict__ dn = nullptr,
                                                        double* y0 = nullptr, double* y1 = nullptr, const int* __restrict__ cls_lab = nullptr,
                                                        const int* __restrict__ dom_lab = nullptr, const int* __restrict__ d_nfirst = nullptr) {
    if (dn) nsel = min(nsel, *dn);               // the row count is the device's (candidate rule on the device)
    // rows >= *d_nfirst are the labelled regions: their dominant_point_ids come from the GROUND-TRUTH classes (sampler2.py:288-291), the candidates'
    // from the predicted ones (:625-626)
    const int nfirst = (cls_lab && d_nfirst) ? *d_nfirst : nsel;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nsel * D; e += gridDim.x * 256) {
        const int q = e / D, c = e % D, s = sel ? sel[q] : q;
        const int* __restrict__ cls = q < nfirst ? cls_pred : cls_lab;
        const int lo = sp_off[s], hi = sp_off[s + 1], d = (q < nfirst ? dom : dom_lab)[s];
        float sum = 0.f; int cnt = 0;
        // thirty-two members at a time: their three dependent loads (member, class, feature) are in flight together; the additions stay in
        // member order (sixteen: 40 us for the bench's 1184 regions of ~185 points — a chain of round trips, not bandwidth)
        for (int j0 = lo; j0 < hi; j0 += 32) {
            int p[32], k[32]; float v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) p[u] = sp_pts[min(j0 + u, hi - 1)];
#pragma unroll
            for (int u = 0; u < 32; ++u) { k[u] = cls[p[u]]; v[u] = feat[(size_t)p[u] * D + c]; }
#pragma unroll
            for (int u = 0; u < 32; ++u) if (j0 + u < hi && k[u] == d) { sum = sum + v[u]; ++cnt; }
        }
        const float mean = cnt ? sum / (float)cnt : 0.f;
        if (out) out[(size_t)q * D + c] = mean;
        if (y0) { y0[(size_t)q * D + c] = (double)mean; y1[(size_t)q * D + c] = (double)mean; }      // float32 -> float64 as np.concatenate / np.matmul promote it
    }
}

// ---- F1/F2: bbox centres, chamfer, adjacency, propagation for the superpoints `sel` of ONE cloud ----------------
// The lay-out of one cloud (one workgroup): slots of the cloud start at ITEM * (its first row), the per-superpoint tables at its first row.
__device__ void chamfer_plan_body(const int* __restrict__ sp_off, const int* __restrict__ sel, int n, ChamferPack P, int* counts, int* s_n) {
    const int tid = threadIdx.x;
    if (n > PACK_MAX) {
        for (int i = tid; i < n; i += 256) { P.big[i] = i; P.start[i] = -1; }
        if (tid == 0) { counts[0] = 0; counts[1] = n; }
        return;
    }
    for (int i = tid; i < n; i += 256) { const int sp = sel[i]; s_n[i] = sp_off[sp + 1] - sp_off[sp]; }
    __syncthreads();
    // The lay-out: whole superpoints into 256-slot items, in order.  Rounds 2-4 filled ONE item at a time (next fit: ~70 % full — and every padding slot is a
    // lane that runs the whole distance loop for nothing); round 5 keeps the last FOUR opened items open and takes the first of them with room (the fullest-
    // effort rule, first fit over ALL items by one wave with the free slots in registers, packed 3 % tighter and cost 0.1 ms more per step than it saved:
    // a lone wave spends ~1 us per superpoint on the ballot / readlane chain).  Which item a superpoint lands in changes nothing but the padding: its
    // roots are summed in an order that depends on its size alone.  One lane, the recurrence carried in scalars.
    __shared__ int s_start[PACK_MAX];
    if (tid == 0) {
        int base[4] = {0, 0, 0, 0}, used[4] = {ITEM, ITEM, ITEM, ITEM}, items = 0;
        for (int i = 0; i < n; ++i) {
            const int ni = s_n[i];
            if (ni == 0 || ni > ITEM) { s_start[i] = -1; continue; }
            int j = -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (j < 0 && used[k] + ni <= ITEM) j = k;
            if (j < 0) {          // nothing open takes it: the oldest open item is closed, a new one opened
#pragma unroll
                for (int k = 0; k < 3; ++k) { base[k] = base[k + 1]; used[k] = used[k + 1]; }
                j = 3; base[3] = items++ * ITEM; used[3] = 0;
            }
            int st = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k == j) { st = base[k] + used[k]; used[k] += ni; }
            s_start[i] = st;
        }
    }
    __syncthreads();
    // ranks of the item openers and of the pair-by-pair superpoints: counted per thread over a contiguous piece, then offset by the pieces before
    const int per = (n + 255) / 256, lo = min(n, tid * per), hi = min(n, lo + per);
    int ci = 0, cb = 0;
    for (int i = lo; i < hi; ++i) { const int st = s_start[i]; if (st < 0) ++cb; else if ((st & (ITEM - 1)) == 0) ++ci; }
    // exclusive prefix over the 256 pieces: inside a wave by shuffles, across the four waves through LDS (a serial pass by one lane took ~10 us)
    {
        const int lane = tid & 63, w = tid >> 6;
        int pi = ci, pb = cb;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int ui = __shfl_up(pi, o), ub = __shfl_up(pb, o); if (lane >= o) { pi += ui; pb += ub; } }
        __shared__ int s_wi[4], s_wb[4];
        if (lane == 63) { s_wi[w] = pi; s_wb[w] = pb; }
        __syncthreads();
        int oi = 0, ob = 0;
        for (int k = 0; k < w; ++k) { oi += s_wi[k]; ob += s_wb[k]; }
        if (tid == 255) { counts[0] = oi + pi; counts[1] = ob + pb; }
        ci = oi + pi - ci; cb = ob + pb - cb;          // exclusive
    }
    for (int i = lo; i < hi; ++i) {
        const int st = s_start[i];
        P.start[i] = st;
        if (st < 0) P.big[cb++] = i;
        else if ((st & (ITEM - 1)) == 0) P.item_slot[ci++] = st;
    }
}
// ... and its slots: one wave per superpoint (seg / cnt / r2item of the padding slots and unused items were preset by the launcher: -1 / 0 / 0).
// r2sp / r2item: the largest |p|^2 of the superpoint / of the item's superpoints, rounded up — what bounds the float32 screening's error (select_chamfer.hip)
__device__ void chamfer_fill_body(const float* __restrict__ xyz, const int* __restrict__ sp_off, const int* __restrict__ sp_pts, const int* __restrict__ sel, int n,
                                  double* __restrict__ centres, ChamferPack P) {
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        const int st = P.start[i];
        const int sp = sel[i], lo = sp_off[sp], ni = sp_off[sp + 1] - lo;
        // the superpoint's bounding-box centre (sel_centres' arithmetic; its launch is saved: the wave reads the points twice, the second time from L2)
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int a = lane; a < ni; a += 64) {
            const size_t q = sp_pts[lo + a];
#pragma unroll
            for (int d = 0; d < 3; ++d) { const float v = xyz[3 * q + d]; mn[d] = fminf(mn[d], v); mx[d] = fmaxf(mx[d], v); }
        }
        double cen[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) cen[d] = (double)(wave_min(mn[d]) + wave_max(mx[d])) / 2.0;      // float32 add, exact halving (fps_gcn_cpu.py:86-88)
        if (lane < 3) centres[3 * (size_t)i + lane] = lane == 0 ? cen[0] : lane == 1 ? cen[1] : cen[2];
        const double cx = cen[0], cy = cen[1], cz = cen[2];
        double r2 = 0.0;
        for (int a = lane; a < ni; a += 64) {
            const size_t q = sp_pts[lo + a];
            const double x = (double)xyz[3 * q] - cx, y = (double)xyz[3 * q + 1] - cy, z = (double)xyz[3 * q + 2] - cz;
            r2 = fmax(r2, (x * x + y * y) + z * z);
            if (st >= 0) {
                const int k = st + a;
                P.x[k] = x; P.y[k] = y; P.z[k] = z;
                P.seg[k] = i; P.cnt[k] = a == 0 ? ni : 0;
                P.src0[k] = source_operand(x, y, 0);
                const uint4 up = source_operand(z, 0.0, 1);
                P.src1[k] = (unsigned long long)up.x | ((unsigned long long)up.y << 32);
            }
        }
        const float r = wave_max((float)(r2 * 1.000001));
        if (lane == 0) {
            P.r2sp[i] = r;
            if (st >= 0) atomicMax((unsigned*)&P.r2item[st / ITEM], __float_as_uint(r));      // non-negative floats order like their bit patterns
        }
    }
}

__global__ __launch_bounds__(256) void sel_chamfer_plan(const int* __restrict__ sp_off, const int* __restrict__ sel, const int* __restrict__ coff, int nsingle, ChamferPack P) {
    __shared__ int s_n[PACK_MAX];
    const int c = blockIdx.x, lo = coff ? coff[c] : 0, n = coff ? coff[c + 1] - lo : nsingle;
    chamfer_plan_body(sp_off, sel + lo, n, pack_at(P, lo), P.counts + 2 * c, s_n);
}
__global__ __launch_bounds__(256) void sel_chamfer_fill(const float* __restrict__ xyz, const int* __restrict__ sp_off, const int* __restrict__ sp_pts,
                                                        const int* __restrict__ sel, const int* __restrict__ coff, int nsingle, double* __restrict__ centres, ChamferPack P) {
    const int c = blockIdx.y, lo = coff ? coff[c] : 0, n = coff ? coff[c + 1] - lo : nsingle;
    chamfer_fill_body(xyz, sp_off, sp_pts, sel + lo, n, centres + 3 * (size_t)lo, pack_at(P, lo));
}

// adj = exp(-(ED + CD)) - I (fps_gcn_cpu.py:102-104); rowsum (:106)
__device__ void adj_build_body(const double* __restrict__ centres, const double* __restrict__ dir, int n, double* adj, double* rowsum, double* part) {
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        double acc = 0.0;
        for (int j = threadIdx.x; j < n; j += 256) {
            const double dx = centres[3 * i] - centres[3 * j], dy = centres[3 * i + 1] - centres[3 * j + 1], dz = centres[3 * i + 2] - centres[3 * j + 2];
            const double ed = sqrt((dx * dx + dy * dy) + dz * dz);
            const double cd = dir[(size_t)i * n + j] + dir[(size_t)j * n + i];
            double v = exp(-(ed + cd));
            if (i == j) v = v - 1.0;
            adj[(size_t)i * n + j] = v; acc += v;
        }
        part[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) rowsum[i] = part[0];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void sel_adj_build(const double* __restrict__ centres, const double* __restrict__ dir, int n, double* adj, double* rowsum) {
    __shared__ double part[256];
    adj_build_body(centres, dir, n, adj, rowsum, part);
}
__global__ __launch_bounds__(256) void sel_adj_build_batch(const double* __restrict__ centres, const double* __restrict__ dir, const int* __restrict__ coff,
                                                           const long long* __restrict__ boff, double* adj, double* rowsum) {
    __shared__ double part[256];
    const int c = blockIdx.z, lo = coff[c];
    adj_build_body(centres + 3 * (size_t)lo, dir + boff[c], coff[c + 1] - lo, adj + boff[c], rowsum + lo, part);
}
// adj = adj * diag(1/rowsum) + I (:108-115): column j scaled by 1/rowsum[j], inf -> 0
__device__ void adj_norm_body(const double* __restrict__ rowsum, int n, double* adj) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)n * n; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / n), j = (int)(e % n);
        double dinv = 1.0 / rowsum[j];
        if (__builtin_isinf(dinv)) dinv = 0.0;
        adj[e] = adj[e] * dinv + (i == j ? 1.0 : 0.0);
    }
}
__global__ __launch_bounds__(256) void sel_adj_norm(const double* __restrict__ rowsum, int n, double* adj) { adj_norm_body(rowsum, n, adj); }
__global__ __launch_bounds__(256) void sel_adj_norm_batch(const double* __restrict__ rowsum, const int* __restrict__ coff, const long long* __restrict__ boff, double* adj) {
    const int c = blockIdx.z, lo = coff[c];
    adj_norm_body(rowsum + lo, coff[c + 1] - lo, adj + boff[c]);
}
// keep the gcn_top largest entries of every row (fps_gcn_cpu.py:153-160: mask[row, argsort(row)[-top:]] = 1); among equal entries the
// higher column index is kept (NumPy's quicksort leaves such ties unspecified).  One wave per row: the row is copied to LDS, every
// lane ranks its columns against the whole row (n^2 / 64 comparisons per lane) and zeroes the ones ranked >= top.
constexpr int TOPK_ROW = 2048;       // columns a wave keeps in LDS (4 waves x 16 KiB); longer rows rank against global memory
__device__ void adj_topk_body(double* adj, int n, int top, double* s_row /* [4][TOPK_ROW] */) {
    if (top <= 0 || top >= n) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double* mine = s_row + (size_t)wid * TOPK_ROW;
    const bool in_lds = n <= TOPK_ROW;
    for (int i0 = blockIdx.x * 4; i0 < n; i0 += gridDim.x * 4) {          // uniform over the workgroup: barriers inside
        const int i = i0 + wid;
        const bool live = i < n;
        double* row = adj + (size_t)(live ? i : 0) * n;
        if (live && in_lds) for (int k = lane; k < n; k += 64) mine[k] = row[k];
        __syncthreads();
        for (int base = 0; base < n; base += 64 * 64) {
            unsigned long long zero = 0ull;   // this lane's columns j = base + lane + 64 u to clear
            if (live)
                for (int u = 0; u < 64; ++u) {
                    const int j = base + lane + 64 * u;
                    if (j >= n) break;
                    // rows longer than the LDS copy are ranked in place: a column already decided is stored as -v - 4 (entries lie in
                    // [0, 2]) and decoded here, and is cleared after the whole row has been ranked
                    auto val = [&](int k) { const double v = in_lds ? mine[k] : row[k]; return v < -1.0 ? -(v + 4.0) : v; };
                    const double vj = val(j);
                    int larger = 0;
                    for (int k = 0; k < n; ++k) { const double vk = val(k); larger += (vk > vj) || (vk == vj && k > j); }
                    if (larger >= top) zero |= 1ull << u;
                }
            if (!in_lds) __syncthreads();     // every lane has read this pass's columns before anyone marks one
            for (int u = 0; u < 64; ++u) if ((zero >> u) & 1ull) row[base + lane + 64 * u] = in_lds ? 0.0 : -row[base + lane + 64 * u] - 4.0;
            if (!in_lds) __syncthreads();
        }
        if (live && !in_lds) for (int k = lane; k < n; k += 64) if (row[k] < -1.0) row[k] = 0.0;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void sel_adj_topk(double* adj, int n, int top) {
    __shared__ double s_row[4 * TOPK_ROW];
    adj_topk_body(adj, n, top, s_row);
}
__global__ __launch_bounds__(256) void sel_adj_topk_batch(double* adj, const int* __restrict__ coff, const long long* __restrict__ boff, int top) {
    __shared__ double s_row[4 * TOPK_ROW];
    const int c = blockIdx.z;
    adj_topk_body(adj + boff[c], coff[c + 1] - coff[c], top, s_row);
}
// Vout[rows[i]] = sum_j adj[i][j] * Vin[rows[j]]  (one hop of fps_gcn_cpu.py:164-165), comb[rows[i]] += Vout
__device__ void propagate_body(const double* __restrict__ adj, int n, const int* __restrict__ rows, const double* __restrict__ vin, int D,
                               double* vout, double* comb) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n * D; e += gridDim.x * 256) {
        const int i = e / D, c = e % D;
        double acc = 0.0;
        // eight terms' dependent loads (row index, then the feature) in flight; the additions stay in column order (np.matmul's row-times-column sum
        // is compared at 1e-12, and the selection downstream must not depend on the unrolling)
        for (int j0 = 0; j0 < n; j0 += 8) {
            int rj[8]; double a[8], v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) rj[t] = rows[min(j0 + t, n - 1)];
#pragma unroll
            for (int t = 0; t < 8; ++t) { a[t] = adj[(size_t)i * n + min(j0 + t, n - 1)]; v[t] = vin[(size_t)rj[t] * D + c]; }
#pragma unroll
            for (int t = 0; t < 8; ++t) if (j0 + t < n) acc += a[t] * v[t];
        }
        vout[(size_t)rows[i] * D + c] = acc;
        comb[(size_t)rows[i] * D + c] += acc;
    }
}
__global__ __launch_bounds__(256) void sel_propagate(const double* __restrict__ adj, int n, const int* __restrict__ rows, const double* __restrict__ vin, int D,
                                                     double* vout, double* comb) { propagate_body(adj, n, rows, vin, D, vout, comb); }
__global__ __launch_bounds__(256) void sel_propagate_batch(const double* __restrict__ adj, const int* __restrict__ coff, const long long* __restrict__ boff,
                                                           const int* __restrict__ rows, const double* __restrict__ vin, int D, double* vout, double* comb) {
    const int c = blockIdx.z, lo = coff[c];
    propagate_body(adj + boff[c], coff[c + 1] - lo, rows + lo, vin, D, vout, comb);
}

// ---- F4: farthest_features_sample (fps_gcn_cpu.py:119-147) / F5: kCenterGreedy (kcenterGreedy.py:84-128) --------
struct Part { double v; int i; int pad; };

__device__ __forceinline__ bool better(double v, int i, double bv, int bi) { return v > bv || (v == bv && i < bi); }   // np.argmax: first maximum

// Wavefront arg-max of (value, index) pairs, result in every lane.  The picks of FPS / k-center form a serial chain of
// ~600 such reductions: inside a row of 16 lanes the partners come through DPP (quad permutes, half-row and row mirrors),
// across rows through gfx950's v_permlane16_swap / v_permlane32_swap — no LDS crossbar round trips (ds_bpermute) at all.
// Two passes (round 4): the maximum of the VALUES alone (two moves and one v_max_f64 per step), then the smallest index among the lanes
// that hold it (one v_min_i32 per step) — 30 dependent instructions instead of 65 for the (value, index) pairs compared step by step,
// on a chain where a float64 instruction of a lone wave takes 12 cycles (tools/micro/valu_rate.hip).  Values are never NaN here.
#ifndef HIPEMU
__device__ __forceinline__ double max_f64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int CTRL> __device__ __forceinline__ unsigned dpp_u32(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ double dpp_max_f64(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = dpp_u32<CTRL>((unsigned)b), hi = dpp_u32<CTRL>((unsigned)(b >> 32));
    return max_f64(v, __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)));
}
template <int CTRL> __device__ __forceinline__ int dpp_min_i32(int t) { return min(t, (int)dpp_u32<CTRL>((unsigned)t)); }
// ... inside every row of 16 lanes (0xB1 quad_perm [1,0,3,2], 0x4E quad_perm [2,3,0,1], 0x141 row_half_mirror, 0x140 row_mirror)
__device__ __forceinline__ double row_max_f64(double v) { v = dpp_max_f64<0xB1>(v); v = dpp_max_f64<0x4E>(v); v = dpp_max_f64<0x141>(v); return dpp_max_f64<0x140>(v); }
__device__ __forceinline__ int row_min_i32(int t) { t = dpp_min_i32<0xB1>(t); t = dpp_min_i32<0x4E>(t); t = dpp_min_i32<0x141>(t); return dpp_min_i32<0x140>(t); }
#endif
__device__ __forceinline__ void wave_argmax(double& v, int& i) {
#ifndef HIPEMU
    double m = row_max_f64(v);
    // rows 0<->1, 2<->3, then the halves: after swap(a, a) the two results hold the even / odd row (half) of each pair
    {
        const long long b = __double_as_longlong(m);
        auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false), hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        m = max_f64(__longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])), __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1])));
    }
    {
        const long long b = __double_as_longlong(m);
        auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false), hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        m = max_f64(__longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])), __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1])));
    }
    int t = row_min_i32(v == m ? i : 0x7fffffff);
    { auto r = __builtin_amdgcn_permlane16_swap((unsigned)t, (unsigned)t, false, false); t = min((int)r[0], (int)r[1]); }
    { auto r = __builtin_amdgcn_permlane32_swap((unsigned)t, (unsigned)t, false, false); t = min((int)r[0], (int)r[1]); }
    v = m; i = t;
#else
    for (int o = 32; o > 0; o >>= 1) {
        const long long b = __double_as_longlong(v);
        const unsigned lo = __shfl_xor((unsigned)b, o), hi = __shfl_xor((unsigned)(b >> 32), o);
        const double ov = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        const int oi = __shfl_xor(i, o);
        if (better(ov, oi, v, i)) { v = ov; i = oi; }
    }
#endif
}
// the index of the best of NW <= 16 (value, index) pairs in LDS, in every lane of the calling wave: lanes 0..NW-1 take one pair each and reduce
// inside their row of 16 (the other rows reduce padding)
template <int NW> __device__ __forceinline__ int pairs_argmax_index(const double* sv, const int* si, int lane) {
    static_assert(NW <= 16, "one row of lanes");
#ifndef HIPEMU
    const double v = lane < NW ? sv[lane] : -2.0;
    const int i = lane < NW ? si[lane] : 0x7fffffff;
    const double m = row_max_f64(v);
    return __builtin_amdgcn_readfirstlane(row_min_i32(v == m ? i : 0x7fffffff));
#else
    (void)lane;
    double bv = sv[0]; int bi = si[0];
    for (int w = 1; w < NW; ++w) if (better(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
    return bi;
#endif
}

// One step: (1) every block reduces the previous step's partial maxima to learn the current centre,
// (2) updates the running min-distance of its points, (3) publishes its own partial maximum.
__global__ __launch_bounds__(256) void fps_step(const double* __restrict__ f, int n, int D, int from_partials, int start, int use_sqrt,
                                                const Part* __restrict__ pin, int npart, Part* pout, double* mind, int* out, const int* __restrict__ dn = nullptr) {
    __shared__ Part s_p[256];
    if (dn) n = min(n, *dn);
    __shared__ int s_c;
    const int tid = threadIdx.x;
    if (!from_partials) { if (tid == 0) s_c = start; }
    else {
        Part b; b.v = -1.0; b.i = 0x7fffffff;
        for (int k = tid; k < npart; k += 256) if (better(pin[k].v, pin[k].i, b.v, b.i)) b = pin[k];
        s_p[tid] = b;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o && better(s_p[tid + o].v, s_p[tid + o].i, s_p[tid].v, s_p[tid].i)) s_p[tid] = s_p[tid + o]; __syncthreads(); }
        if (tid == 0) s_c = s_p[0].i;
    }
    __syncthreads();
    const int c = s_c;
    if (blockIdx.x == 0 && tid == 0 && out) *out = c;
    __syncthreads();
    if (!pout) return;
    const double* fc = f + (size_t)c * D;
    Part b; b.v = -1.0; b.i = 0x7fffffff;
    for (int i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
        const double* fi = f + (size_t)i * D;
        double dist = np_pairwise<double>([&](int k) { const double d = fi[k] - fc[k]; return d * d; }, D);
        if (use_sqrt) dist = sqrt(dist);
        double m = mind[i];
        if (dist < m) { m = dist; mind[i] = m; }
        if (better(m, i, b.v, b.i)) { b.v = m; b.i = i; }
    }
    s_p[tid] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o && better(s_p[tid + o].v, s_p[tid + o].i, s_p[tid].v, s_p[tid].i)) s_p[tid] = s_p[tid + o]; __syncthreads(); }
    if (tid == 0) pout[blockIdx.x] = s_p[0];
}

// Whole FPS / k-center chain in ONE workgroup (no launch per iteration) for candidate sets that one CU can sweep
// per step.  DF > 0: feature length known at compile time; the first two points of every thread stay in registers
// (n <= 2048 -> no global feature traffic inside the loop).  mind[] lives in global memory; the arg-max is a wave
// shuffle + LDS reduction.
template <int DF>
__global__ __launch_bounds__(1024) void fps_block(const double* __restrict__ f, int n, int D, int from_partials, int start, int use_sqrt,
                                                  const Part* __restrict__ pin, int npart, double* mind, int count, int* out, const int* __restrict__ dn = nullptr) {
    if (dn) n = min(n, *dn);
    __shared__ double s_v[16];
    __shared__ int s_i[16];
    __shared__ int s_c;
    __shared__ double s_fc[DF > 0 ? DF : 1];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    auto block_argmax = [&](double v, int i) {
        wave_argmax(v, i);
        if (lane == 0) { s_v[wid] = v; s_i[wid] = i; }
        __syncthreads();
        if (tid == 0) {
            double bv = s_v[0]; int bi = s_i[0];
            for (int w = 1; w < 16; ++w) if (better(s_v[w], s_i[w], bv, bi)) { bv = s_v[w]; bi = s_i[w]; }
            s_c = bi;
        }
        __syncthreads();
    };
    constexpr int NR = 0;   // points per thread kept in registers: none (64 doubles per point spill at 1024 threads); L2 serves them
    double reg[NR > 0 ? NR : 1][DF > 0 ? DF : 1];
    double rmin[NR > 0 ? NR : 1];
    if (DF > 0) {
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int i = tid + q * 1024;
            rmin[q] = i < n ? mind[i] : -1.0;
#pragma unroll
            for (int k = 0; k < (DF > 0 ? DF : 1); ++k) reg[q][k] = i < n ? f[(size_t)i * D + k] : 0.0;
        }
    }
    if (!from_partials) { if (tid == 0) s_c = start; __syncthreads(); }
    else {
        double v = -1.0; int i = 0x7fffffff;
        for (int k = tid; k < npart; k += 1024) if (better(pin[k].v, pin[k].i, v, i)) { v = pin[k].v; i = pin[k].i; }
        block_argmax(v, i);
    }
    for (int it = 0; it < count; ++it) {
        const int c = s_c;
        if (tid == 0) out[it] = c;
        if (it + 1 == count) break;
        const double* fc = f + (size_t)c * D;
        if (DF > 0) { if (tid < DF) s_fc[tid] = fc[tid]; __syncthreads(); }
        double bv = -1.0; int bi = 0x7fffffff;
        if (DF > 0) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int i = tid + q * 1024;
                if (i < n) {
                    double dist = np_pairwise_fixed<(DF > 0 ? DF : 8)>([&](int k) { const double d = reg[q][k] - s_fc[k]; return d * d; });
                    if (use_sqrt) dist = sqrt(dist);
                    if (dist < rmin[q]) rmin[q] = dist;
                    if (better(rmin[q], i, bv, bi)) { bv = rmin[q]; bi = i; }
                }
            }
        }
        for (int i = tid + NR * 1024; i < n; i += 1024) {
            const double* fi = f + (size_t)i * D;
            double dist;
            if (DF > 0) dist = np_pairwise_fixed<(DF > 0 ? DF : 8)>([&](int k) { const double d = fi[k] - s_fc[k]; return d * d; });
            else dist = np_pairwise<double>([&](int k) { const double d = fi[k] - fc[k]; return d * d; }, D);
            if (use_sqrt) dist = sqrt(dist);
            double m = mind[i];
            if (dist < m) { m = dist; mind[i] = m; }
            if (better(m, i, bv, bi)) { bv = m; bi = i; }
        }
        __syncthreads();          // everyone has read s_c / s_fc
        block_argmax(bv, bi);
    }
}

// Register-resident variant for small candidate sets (n <= 512 * PPT): every thread owns PPT points whose features
// and running min-distance never leave its registers; the current centre's features are published through LDS by
// the owning thread, so the loop touches global memory only to store the selected index.
template <int DF, int PPT, int NT>
__global__ __launch_bounds__(NT) void fps_block_reg(const double* __restrict__ f, int n, int from_partials, int start, int use_sqrt,
                                                    const Part* __restrict__ pin, int npart, const double* __restrict__ mind, int count, int* out,
                                                    const int* __restrict__ dn = nullptr) {
    if (dn) n = min(n, *dn);
    constexpr int NW = NT / 64;
    __shared__ double s_v[2][NW];
    __shared__ int s_i[2][NW];
    __shared__ double s_fc[2][DF];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double reg[PPT][DF], rmin[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int i = tid + q * NT;
        rmin[q] = i < n ? mind[i] : -1.0;
#pragma unroll
        for (int k = 0; k < DF; ++k) reg[q][k] = i < n ? f[(size_t)i * DF + k] : 0.0;
    }
    // publishes this wave's best into s_v/s_i[par]; after the barrier every wave reduces the NW wave results itself
    auto block_argmax = [&](double v, int i, int par) -> int {
        wave_argmax(v, i);
        if (lane == 0) { s_v[par][wid] = v; s_i[par][wid] = i; }
        __syncthreads();
        return pairs_argmax_index<NW>(s_v[par], s_i[par], lane);
    };
    int c;
    if (!from_partials) c = start;
    else {
        double v = -1.0; int i = 0x7fffffff;
        for (int k = tid; k < npart; k += NT) if (better(pin[k].v, pin[k].i, v, i)) { v = pin[k].v; i = pin[k].i; }
        c = block_argmax(v, i, 1);
    }
    for (int it = 0; it < count; ++it) {
        const int par = it & 1;
        if (tid == 0) out[it] = c;
        if (it + 1 == count) break;
#pragma unroll
        for (int q = 0; q < PPT; ++q) if (c == tid + q * NT) {
#pragma unroll
            for (int k = 0; k < DF; ++k) s_fc[par][k] = reg[q][k];
        }
        __syncthreads();
        double bv = -1.0; int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int i = tid + q * NT;
            if (i < n) {
                double dist = np_pairwise_fixed<DF>([&](int k) { const double d = reg[q][k] - s_fc[par][k]; return d * d; });
                if (use_sqrt) dist = sqrt(dist);
                if (dist < rmin[q]) rmin[q] = dist;
                if (better(rmin[q], i, bv, bi)) { bv = rmin[q]; bi = i; }
            }
        }
        c = block_argmax(bv, bi, par);       // two barriers per iteration; parity double-buffering covers the reuse
    }
}

// FPS / k-center over more candidates than one workgroup sweeps per pick (n > 16384): ONE launch of G co-resident workgroups instead of a
// launch per pick (the reference's AL rounds pick 10 000 of ~2 x 10^4 regions, ssdr_main_S3DIS2.py:134: 10^4 dependent launches).  Every
// workgroup keeps the running min-distance of its points in registers; per pick it publishes its partial arg-max with write-through (sc1)
// stores, drains them, adds to a counter; all workgroups poll that counter and read the G partials with sc1 loads (the fence-free hand-off
// of MI355X_MICROARCH.md, "Valid forms": 9.3 us per pick with __threadfence on both sides, measured), and each reduces them itself.  Partials and counters alternate between two
// sets by the parity of the pick, so a workgroup that runs ahead never overwrites what a slower one still reads.
constexpr int FC_NT = 256, FC_PPT = 8;
struct FpsCoopArgs { const double* f; int n, D, from_partials, start, use_sqrt; const Part* pin; int npart; const double* mind; int count; int* out; Part* part; int* sync; int G; int* status; const int* dn; };
constexpr long FPS_COOP_SPINS = 1L << 22;      // ~0.3 s of polling: a pick among co-resident workgroups takes microseconds
// sync[0], sync[1]: arrival counters by pick parity; sync[2]: abort — a workgroup waited FPS_COOP_SPINS polls for one that never arrived (the launch was
// not co-resident).  It is checked at every pick by every workgroup, which then leaves (the picks from there on read -1), and the stream's selection
// status word takes bit 0: ssdr_select_status turns it into an error.  Without it a non-resident launch returned a wrong selection silently.
#ifndef HIPEMU
__global__ __launch_bounds__(FC_NT) void fps_coop(FpsCoopArgs a) {
    if (a.dn) a.n = min(a.n, *a.dn);
    __shared__ double s_v[FC_NT / 64]; __shared__ int s_i[FC_NT / 64]; __shared__ double s_fc[128];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = blockIdx.x, G = a.G;
    auto block_argmax = [&](double v, int i, double& ov, int& oi) {
        wave_argmax(v, i);
        if (lane == 0) { s_v[wid] = v; s_i[wid] = i; }
        __syncthreads();
        ov = s_v[0]; oi = s_i[0];
#pragma unroll
        for (int w = 1; w < FC_NT / 64; ++w) if (better(s_v[w], s_i[w], ov, oi)) { ov = s_v[w]; oi = s_i[w]; }
        __syncthreads();
    };
    // this workgroup's points: i = (k * G + g) * FC_NT + tid
    double rmin[FC_PPT];
#pragma unroll
    for (int k = 0; k < FC_PPT; ++k) { const long i = ((long)k * G + g) * FC_NT + tid; rmin[k] = i < a.n ? a.mind[i] : -1.0; }
    int c;
    if (!a.from_partials) c = a.start;
    else {
        double v = -1.0; int i = 0x7fffffff;
        for (int k = tid; k < a.npart; k += FC_NT) if (better(a.pin[k].v, a.pin[k].i, v, i)) { v = a.pin[k].v; i = a.pin[k].i; }
        double ov; block_argmax(v, i, ov, c);
    }
    __shared__ int s_abort;
    for (int it = 0; it < a.count; ++it) {
        if (tid == 0) s_abort = __hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_abort) { if (g == 0) for (int k = it + tid; k < a.count; k += FC_NT) a.out[k] = -1; return; }
        if (g == 0 && tid == 0) a.out[it] = c;
        if (it + 1 == a.count) break;
        const double* fc = a.f + (size_t)c * a.D;
        if (a.D <= 128) { if (tid < a.D) s_fc[tid] = fc[tid]; __syncthreads(); }
        double bv = -1.0; int bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < FC_PPT; ++k) {
            const long i = ((long)k * G + g) * FC_NT + tid;
            if (i < a.n) {
                const double* fi = a.f + (size_t)i * a.D;
                double dist;
                if (a.D == 32) dist = np_pairwise_fixed<32>([&](int q) { const double d = fi[q] - s_fc[q]; return d * d; });
                else if (a.D <= 128) dist = np_pairwise<double>([&](int q) { const double d = fi[q] - s_fc[q]; return d * d; }, a.D);
                else dist = np_pairwise<double>([&](int q) { const double d = fi[q] - fc[q]; return d * d; }, a.D);
                if (a.use_sqrt) dist = sqrt(dist);
                if (dist < rmin[k]) rmin[k] = dist;
                if (better(rmin[k], (int)i, bv, bi)) { bv = rmin[k]; bi = (int)i; }
            }
        }
        double wv; int wi;
        block_argmax(bv, bi, wv, wi);
        const int par = it & 1;
        Part* P = a.part + (size_t)par * G;
        if (tid == 0) {
            // write-through (sc1) stores of the partial, drained, then the arrival: no cache write-back / invalidate on either side
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(&P[g].v), (unsigned long long)__double_as_longlong(wv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&P[g].i, wi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&a.sync[par], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int want = G * (it / 2 + 1);
            long spins = 0;
            while (__hip_atomic_load(&a.sync[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > FPS_COOP_SPINS || __hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicOr(&a.sync[2], 1); atomicOr(a.status, 1); break; }
            }
        }
        __syncthreads();
        double v = -1.0; int i = 0x7fffffff;
        for (int k = tid; k < G; k += FC_NT) {
            const double pv = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&P[k].v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const int pi = __hip_atomic_load(&P[k].i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (better(pv, pi, v, i)) { v = pv; i = pi; }
        }
        double ov; block_argmax(v, i, ov, c);
    }
}
#endif

// The same chain for 32-d features with every workgroup's rows IN REGISTERS (512 per workgroup: one row per thread on eight waves since round 4 — two waves
// per SIMD issue a float64 instruction every 6.5 cycles, the lone wave of the 256-thread form with two rows per thread every 12: 3.21 / 3.64 / 4.51 -> 2.84 / 3.17 /
// 3.46 us per pick at 2368 / 4736 / 9472 rows on an idle GPU; 1024 rows per workgroup on 512 threads: 3.54 / 3.65 / 3.91) and partials that carry the candidate's
// features: a pick costs one publish (the owner's row through LDS, one 272-byte write-through store by 34 lanes, drained, counter) and ONE round of loads
// (every workgroup reads all G partials, features included, into LDS and finds the winner there) instead of three dependent rounds (partials, the winner's
// row from the feature table, every row's features from L2).  This is the replicated global FPS of the sharded run (2 / 4 / 8 ranks: 2368 / 4736 / 9472 rows).
#ifndef HIPEMU
constexpr int FR_NT = 512, FR_RPT = 1, FR_ROWS = FR_NT * FR_RPT, FR_REC = 34;        // record: v, (i, pad), f[32] as 34 doubles
__global__ __launch_bounds__(FR_NT) void fps_coop_reg(FpsCoopArgs a) {
    if (a.dn) a.n = min(a.n, *a.dn);
    extern __shared__ double s_all[];                      // [G][FR_REC]: the partials of a pick, as read
    __shared__ double s_v[FR_NT / 64]; __shared__ int s_i[FR_NT / 64]; __shared__ double s_fc[32]; __shared__ double s_pub[FR_REC];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = blockIdx.x, G = a.G;
    double reg[FR_RPT][32], rmin[FR_RPT];
#pragma unroll
    for (int q = 0; q < FR_RPT; ++q) {
        const long i = (long)g * FR_ROWS + q * FR_NT + tid;
        rmin[q] = i < a.n ? a.mind[i] : -1.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) reg[q][k] = i < a.n ? a.f[(size_t)i * 32 + k] : 0.0;
    }
    auto block_argmax = [&](double v, int i, double& ov, int& oi) {
        wave_argmax(v, i);
        if (lane == 0) { s_v[wid] = v; s_i[wid] = i; }
        __syncthreads();
        ov = s_v[0]; oi = s_i[0];
#pragma unroll
        for (int w = 1; w < FR_NT / 64; ++w) if (better(s_v[w], s_i[w], ov, oi)) { ov = s_v[w]; oi = s_i[w]; }
        __syncthreads();
    };
    int c;
    if (!a.from_partials) c = a.start;
    else {
        double v = -1.0; int i = 0x7fffffff;
        for (int k = tid; k < a.npart; k += FR_NT) if (better(a.pin[k].v, a.pin[k].i, v, i)) { v = a.pin[k].v; i = a.pin[k].i; }
        double ov; block_argmax(v, i, ov, c);
    }
    if (tid < 32) s_fc[tid] = a.f[(size_t)c * 32 + tid];    // the first centre's row comes from the table
    __syncthreads();
    double* recs = reinterpret_cast<double*>(a.part);      // [2][G][FR_REC]
    __shared__ int s_abort;
    for (int it = 0; it < a.count; ++it) {
        if (tid == 0) s_abort = __hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_abort) { if (g == 0) for (int k = it + tid; k < a.count; k += FR_NT) a.out[k] = -1; return; }
        if (g == 0 && tid == 0) a.out[it] = c;
        if (it + 1 == a.count) break;
        double bv = -1.0; int bi = 0x7fffffff, bq = 0;
#pragma unroll
        for (int q = 0; q < FR_RPT; ++q) {
            const int i = g * FR_ROWS + q * FR_NT + tid;
            if (i < a.n) {
                double dist = np_pairwise_fixed<32>([&](int k) { const double d = reg[q][k] - s_fc[k]; return d * d; });
                if (a.use_sqrt) dist = sqrt(dist);
                if (dist < rmin[q]) rmin[q] = dist;
                if (better(rmin[q], i, bv, bi)) { bv = rmin[q]; bi = i; bq = q; }
            }
        }
        double wv; int wi;
        block_argmax(bv, bi, wv, wi);
        // the owner of the workgroup's best row lays the record out in LDS ...
        if (wi == bi && bi != 0x7fffffff) {
            s_pub[0] = wv; s_pub[1] = __longlong_as_double((long long)(unsigned)wi);
#pragma unroll
            for (int k = 0; k < 32; ++k) s_pub[2 + k] = bq == 0 ? reg[0][k] : reg[FR_RPT - 1][k];
        } else if (wi == 0x7fffffff && tid == 0) { s_pub[0] = -1.0; s_pub[1] = __longlong_as_double(0x7fffffffll); }      // a workgroup of padding rows only
        __syncthreads();
        const int par = it & 1;
        double* mine = recs + ((size_t)par * G + g) * FR_REC;
        if (tid < 64) {
            // ... and one wave writes it through (sc1), drains, then arrives: no cache write-back / invalidate on either side
            if (tid < FR_REC) __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine + tid), (unsigned long long)__double_as_longlong(s_pub[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) {
                __hip_atomic_fetch_add(&a.sync[par], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int want = G * (it / 2 + 1);
                long spins = 0;
                while (__hip_atomic_load(&a.sync[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > FPS_COOP_SPINS || __hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicOr(&a.sync[2], 1); atomicOr(a.status, 1); break; }
                }
            }
        }
        __syncthreads();
        // every partial, features included, in ONE round of loads
        const double* all = recs + (size_t)par * G * FR_REC;
        for (int k = tid; k < G * FR_REC; k += FR_NT)
            s_all[k] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(all + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __syncthreads();
        double v = -1.0; int i = 0x7fffffff;
        for (int k = tid; k < G; k += FR_NT) {
            const double pv = s_all[(size_t)k * FR_REC]; const int pi = (int)(unsigned)__double_as_longlong(s_all[(size_t)k * FR_REC + 1]);
            if (better(pv, pi, v, i)) { v = pv; i = pi; }
        }
        double ov; block_argmax(v, i, ov, c);
        // the winner's row: the record of the workgroup that owns row c
        const int wg = c / FR_ROWS;
        if (tid < 32) s_fc[tid] = s_all[(size_t)wg * FR_REC + 2 + tid];
        __syncthreads();
    }
}

// The same chain with the hand-off in self-validating granules (MI355X_MICROARCH.md, price list: handoff-1to1 against handoff-flag): a record travels as 68
// 8-byte words {32 bits of data, pick number}, each written by ONE write-through store and polled directly by its reader — no drain, no counter, no second
// round trip.  Records alternate between two sets by the parity of the pick (a workgroup is at most one pick ahead of the slowest), the winner is found by
// every wave for itself out of the LDS copy (no barrier), and the distance loop reads the winner's features straight from that copy.  Four barriers per pick.
constexpr int FT_WORDS = 2 * FR_REC;         // 68 granules per record
__global__ __launch_bounds__(FR_NT) void fps_coop_tag(FpsCoopArgs a) {
    if (a.dn) a.n = min(a.n, *a.dn);
    extern __shared__ unsigned s_rec[];                    // [2][G][FT_WORDS]: the records of a pick, as read (data words)
    __shared__ double s_v[2][FR_NT / 64]; __shared__ int s_i[2][FR_NT / 64]; __shared__ unsigned s_pub[FT_WORDS]; __shared__ int s_abort, s_gave; __shared__ double s_f0[32];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = blockIdx.x, G = a.G;
    if (tid == 0) s_gave = 0;                              // (ordered before its first reader by the barrier behind s_f0 below)
    double reg[FR_RPT][32], rmin[FR_RPT];
#pragma unroll
    for (int q = 0; q < FR_RPT; ++q) {
        const long i = (long)g * FR_ROWS + q * FR_NT + tid;
        rmin[q] = i < a.n ? a.mind[i] : -1.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) reg[q][k] = i < a.n ? a.f[(size_t)i * 32 + k] : 0.0;
    }
    int c;
    if (!a.from_partials) c = a.start;
    else {
        double v = -1.0; int i = 0x7fffffff;
        for (int k = tid; k < a.npart; k += FR_NT) if (better(a.pin[k].v, a.pin[k].i, v, i)) { v = a.pin[k].v; i = a.pin[k].i; }
        wave_argmax(v, i);
        if (lane == 0) { s_v[0][wid] = v; s_i[0][wid] = i; }
        __syncthreads();
        v = s_v[0][0]; c = s_i[0][0];
        for (int w = 1; w < FR_NT / 64; ++w) if (better(s_v[0][w], s_i[0][w], v, c)) { v = s_v[0][w]; c = s_i[0][w]; }
        __syncthreads();
    }
    if (tid < 32) s_f0[tid] = a.f[(size_t)c * 32 + tid];    // the first centre's row comes from the table
    __syncthreads();
    const double* fc = s_f0;
    unsigned long long* recs = reinterpret_cast<unsigned long long*>(a.part);      // [2][G][FT_WORDS]
    for (int it = 0; it < a.count; ++it) {
        if (g == 0 && tid == 0) a.out[it] = c;
        if (it + 1 == a.count) break;
        const int par = it & 1;
        double bv = -1.0; int bi = 0x7fffffff, bq = 0;
#pragma unroll
        for (int q = 0; q < FR_RPT; ++q) {
            const int i = g * FR_ROWS + q * FR_NT + tid;
            if (i < a.n) {
                double dist = np_pairwise_fixed<32>([&](int k) { const double d = reg[q][k] - fc[k]; return d * d; });
                if (a.use_sqrt) dist = sqrt(dist);
                if (dist < rmin[q]) rmin[q] = dist;
                if (better(rmin[q], i, bv, bi)) { bv = rmin[q]; bi = i; bq = q; }
            }
        }
        double wv = bv; int wi = bi;
        wave_argmax(wv, wi);
        if (lane == 0) { s_v[par][wid] = wv; s_i[par][wid] = wi; }
        if (tid == 0) s_abort = __hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();                                   // (1)
        if (s_abort) { if (g == 0) for (int k = it + 1 + tid; k < a.count; k += FR_NT) a.out[k] = -1; return; }
        wv = s_v[par][0]; wi = s_i[par][0];
#pragma unroll
        for (int w = 1; w < FR_NT / 64; ++w) if (better(s_v[par][w], s_i[par][w], wv, wi)) { wv = s_v[par][w]; wi = s_i[par][w]; }
        // the owner of the workgroup's best row lays the record out in LDS ...
        if (wi == bi && bi != 0x7fffffff) {
            double* pub = reinterpret_cast<double*>(s_pub);
            pub[0] = wv; pub[1] = __longlong_as_double((long long)(unsigned)wi);
#pragma unroll
            for (int k = 0; k < 32; ++k) pub[2 + k] = bq == 0 ? reg[0][k] : reg[FR_RPT - 1][k];
        } else if (wi == 0x7fffffff && tid == 0) { double* pub = reinterpret_cast<double*>(s_pub); pub[0] = -1.0; pub[1] = __longlong_as_double(0x7fffffffll); }      // padding rows only
        __syncthreads();                                   // (2)
        // ... and 68 lanes write it through, one self-validating granule each
        const unsigned tag = (unsigned)it + 1u;
        if (tid < FT_WORDS) __hip_atomic_store(recs + ((size_t)par * G + g) * FT_WORDS + tid, ((unsigned long long)tag << 32) | s_pub[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // every record of the pick, polled granule by granule
        unsigned* mine = s_rec + (size_t)par * G * FT_WORDS;
        const unsigned long long* all = recs + (size_t)par * G * FT_WORDS;
        bool gave_up = false;
        for (int k = tid; k < G * FT_WORDS; k += FR_NT) {
            unsigned long long v; long spins = 0;
            while (((v = __hip_atomic_load(all + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != tag) {
                if (++spins > FPS_COOP_SPINS / 16 || (spins % 4096 == 0 && __hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { gave_up = true; break; }
            }
            if (gave_up) break;                            // (a stale granule is never stored, the remaining records are not waited for)
            mine[k] = (unsigned)v;
        }
        if (gave_up) { atomicOr(&a.sync[2], 1); atomicOr(a.status, 1); s_gave = 1; }      // the others see sync[2] in their own polls / at their next pick
        __syncthreads();                                   // (3)
        if (s_gave) { if (g == 0) for (int k = it + 1 + tid; k < a.count; k += FR_NT) a.out[k] = -1; return; }      // before the winner (an LDS pointer) is formed from garbage
        // the winner: every wave finds it for itself (G <= 128 records, two per lane)
        double v = -1.0; int i = 0x7fffffff;
        for (int k = lane; k < G; k += 64) {
            const double* r = reinterpret_cast<const double*>(mine + (size_t)k * FT_WORDS);
            const double pv = r[0]; const int pi = (int)(unsigned)__double_as_longlong(r[1]);
            if (better(pv, pi, v, i)) { v = pv; i = pi; }
        }
        wave_argmax(v, i);
        c = i;
        fc = reinterpret_cast<const double*>(mine + (size_t)(c / FR_ROWS) * FT_WORDS) + 2;      // the winner's row: the record of the workgroup that owns row c
    }
}

// Round 6: the chain at the reference's own scale (10 000 picks over 20 000 candidates, ssdr_main_S3DIS2.py:134) is all hand-off: 0.5 us of float64
// arithmetic per pick, the rest the all-to-all of the G partials.  Two changes against fps_coop_tag:
//  (1) the SWEEP: a record is 34 slots of 16 bytes, each slot two self-validating 8-byte granules {data, tag}; every thread issues ALL its slot loads
//      (16-byte `sc1` loads) before it looks at a tag and re-reads only the slots that were not there yet — one memory round trip per pass instead of one
//      per granule (the polled form walked ceil(68 G / 512) dependent round trips per pick: 6.3 us at G = 40).  The owner's features go to the record
//      straight from a per-wave LDS image written beside the wave arg-max: two barriers per pick instead of four.
//  (2) the TEAM (team = 1): only workgroups that find themselves on ONE XCD take part — HW_REG_XCC_ID is read, not assumed: the first workgroup to
//      arrive names the XCD, the others of that XCD take tickets for the G row blocks, everybody else leaves at once — so that the records travel
//      through that XCD's own L2 (plain stores keep the line there; `sc1` loads by-pass the reading CU's L1 only) instead of the fabric.  Launched with
//      8 (G + 2) workgroups: placement is the dispatcher's (round-robin over the XCDs as observed, promised nowhere); too few workgroups on the XCD is
//      the same bounded wait -> abort -> status as a launch that was not co-resident, never a wrong selection.
constexpr int FS_SLOTS = FR_REC;             // 16-byte slots per record: words (v lo, v hi), (i, 0), 32 x (f lo, f hi)
constexpr int FS_XCC_ID = ((4 - 1) << 11) | 20;      // s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4)
typedef unsigned fs_u4 __attribute__((ext_vector_type(4)));
// TIMED (development, SSDR_FPS_DBG=1): wave 0 of every workgroup accumulates s_memtime between the phases of a pick into dbg[g][8]
template <bool TIMED>
__global__ __launch_bounds__(FR_NT, 4) void fps_coop_sweep(FpsCoopArgs a, int team, int plain_store, long long* dbg) {
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto mark = [&](int k) { if (TIMED) { const long long t = (long long)__builtin_readcyclecounter(); tacc[k] += t - tprev; tprev = t; } };
    if (a.dn) a.n = min(a.n, *a.dn);
    extern __shared__ __attribute__((aligned(16))) unsigned s_rec[];      // [2][G][FT_WORDS] data words of the records of a pick
    __shared__ double s_v[2][FR_NT / 64]; __shared__ int s_i[2][FR_NT / 64]; __shared__ __attribute__((aligned(16))) unsigned s_pubw[FR_NT / 64][FT_WORDS];
    __shared__ int s_g, s_gave; __shared__ double s_f0[32];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, G = a.G;
    if (tid == 0) {
        s_gave = 0;
        int g = blockIdx.x;
        if (team) {
            const int xcc = (int)(__builtin_amdgcn_s_getreg(FS_XCC_ID) & 0xf);
            const int seen = atomicCAS(&a.sync[4], 0, xcc + 1);
            g = (seen == 0 || seen == xcc + 1) ? atomicAdd(&a.sync[5], 1) : G;
        }
        s_g = g;
    }
    __syncthreads();
    const int g = s_g;
    if (g >= G) return;                                     // another XCD's workgroup, or a spare of the team's
    const int row = g * FR_ROWS + tid;
    double reg[32], rmin = row < a.n ? a.mind[row] : -1.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) reg[k] = row < a.n ? a.f[(size_t)row * 32 + k] : 0.0;
    int c;
    if (!a.from_partials) c = a.start;
    else {
        double v = -1.0; int i = 0x7fffffff;
        for (int k = tid; k < a.npart; k += FR_NT) if (better(a.pin[k].v, a.pin[k].i, v, i)) { v = a.pin[k].v; i = a.pin[k].i; }
        wave_argmax(v, i);
        if (lane == 0) { s_v[0][wid] = v; s_i[0][wid] = i; }
        __syncthreads();
        v = s_v[0][0]; c = s_i[0][0];
        for (int w = 1; w < FR_NT / 64; ++w) if (better(s_v[0][w], s_i[0][w], v, c)) { v = s_v[0][w]; c = s_i[0][w]; }
        __syncthreads();
    }
    if (tid < 32) s_f0[tid] = a.f[(size_t)c * 32 + tid];    // the first centre's row comes from the table
    __syncthreads();
    const double* fc = s_f0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, 2 * G * FS_SLOTS * 16, 0x00020000);
    const int total = G * FS_SLOTS;
    for (int it = 0; it < a.count; ++it) {
        if (g == 0 && tid == 0) a.out[it] = c;
        if (it + 1 == a.count) break;
        const int par = it & 1;
        if (TIMED) tprev = (long long)__builtin_readcyclecounter();
        double wv = -1.0; int wi = 0x7fffffff;
        if (row < a.n) {
            double dist = np_pairwise_fixed<32>([&](int k) { const double d = reg[k] - fc[k]; return d * d; });
            if (a.use_sqrt) dist = sqrt(dist);
            if (dist < rmin) rmin = dist;
            wv = rmin; wi = row;
        }
        mark(0);
        wave_argmax(wv, wi);
        mark(1);
        // the wave's best row lays its record out (the workgroup's winner is read from the winning wave's image)
        if (wi == row) {
            double* pub = reinterpret_cast<double*>(s_pubw[wid]);
            pub[0] = wv; pub[1] = __longlong_as_double((long long)(unsigned)wi);
#pragma unroll
            for (int k = 0; k < 32; ++k) pub[2 + k] = reg[k];
        }
        if (lane == 0) { s_v[par][wid] = wv; s_i[par][wid] = wi; }
        mark(2);
        __syncthreads();                                   // (1)
        mark(3);
        const unsigned tag = (unsigned)it + 1u;
        const unsigned base = (unsigned)(par * G) * FS_SLOTS * 16u;
        if (tid < FS_SLOTS) {
            double bv = s_v[par][0]; int bi = s_i[par][0], bw = 0;
#pragma unroll
            for (int w = 1; w < FR_NT / 64; ++w) if (better(s_v[par][w], s_i[par][w], bv, bi)) { bv = s_v[par][w]; bi = s_i[par][w]; bw = w; }
            fs_u4 v;
            if (bi != 0x7fffffff) { v.x = s_pubw[bw][2 * tid]; v.z = s_pubw[bw][2 * tid + 1]; }
            else {          // padding rows only
                const unsigned long long neg1 = (unsigned long long)__double_as_longlong(-1.0);
                v.x = tid == 0 ? (unsigned)neg1 : tid == 1 ? 0x7fffffffu : 0u; v.z = tid == 0 ? (unsigned)(neg1 >> 32) : 0u;
            }
            v.y = tag; v.w = tag;
            const unsigned off = base + (unsigned)(g * FS_SLOTS + tid) * 16u;
            if (plain_store) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);                          // sc1: write-through
        }
        mark(4);
        // every record of the pick: all of a thread's slots in flight, the missing ones again
        unsigned* mine = s_rec + (size_t)par * G * FT_WORDS;
        bool gave_up = false; int passes = 0;
        for (int k0 = tid; k0 < total && !gave_up; k0 += 4 * FR_NT) {
            unsigned pend = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (k0 + j * FR_NT < total) pend |= 1u << j;
            long spins = 0;
            while (pend) {
                fs_u4 v[4];
                if (TIMED) ++passes;
#pragma unroll
                for (int j = 0; j < 4; ++j) if ((pend >> j) & 1u) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (unsigned)(k0 + j * FR_NT) * 16u, 0, 16);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (((pend >> j) & 1u) && v[j].y == tag && v[j].w == tag) {
                        *reinterpret_cast<uint2*>(mine + 2 * (size_t)(k0 + j * FR_NT)) = make_uint2(v[j].x, v[j].z);
                        pend &= ~(1u << j);
                    }
                if (pend && (++spins > FPS_COOP_SPINS / 16 || (spins % 1024 == 0 && __hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))) { gave_up = true; break; }
            }
        }
        if (gave_up) { atomicOr(&a.sync[2], 1); atomicOr(a.status, 1); s_gave = 1; }
        mark(5);
        if (TIMED) tacc[7] += passes;
        __syncthreads();                                   // (2)
        mark(6);
        if (s_gave) { if (g == 0) for (int k = it + 1 + tid; k < a.count; k += FR_NT) a.out[k] = -1; return; }
        // the winner: every wave finds it for itself (G <= 64 records, one per lane)
        double v = -1.0; int i = 0x7fffffff;
        if (lane < G) {
            const double* r = reinterpret_cast<const double*>(mine + (size_t)lane * FT_WORDS);
            v = r[0]; i = (int)(unsigned)__double_as_longlong(r[1]);
        }
        wave_argmax(v, i);
        c = i;
        fc = reinterpret_cast<const double*>(mine + (size_t)(c / FR_ROWS) * FT_WORDS) + 2;
        if (TIMED) { const long long t = (long long)__builtin_readcyclecounter(); tacc[6] += t - tprev; tprev = t; }      // (the last arg-max joins slot 6)
    }
    if (TIMED && tid == 0) for (int k = 0; k < 8; ++k) dbg[(size_t)g * 8 + k] = tacc[k];
}
#endif

#ifndef HIPEMU
// Round 6, second form (the phase clocks of fps_coop_sweep, tools/gpu_fps6_dbg.sh: of 4.3 us per pick at 20 000 rows 2.7 are INSIDE the workgroup — 0.73 the
// 95 dependent float64 instructions of a row's distance on a wave that issues one every ~12 cycles, 0.56 the wave arg-max and the owner's 272-byte LDS
// image, 0.43 the skew of two waves per SIMD at the barrier, 0.48 + 0.5 the two combines — and 1.5 the sweep of 40 x 544 bytes):
//  * a ROW IS SPLIT OVER LPR = 2 or 4 LANES: NumPy's eight pairwise accumulators are dealt to the lanes (lane q owns accumulators A q .. A q + A - 1, A = 8 / LPR,
//    i.e. features 8 k + A q + e), each lane adds its accumulators in the reference's order and the tree ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) closes through one
//    or two quad exchanges (a + b == b + a bit for bit): 29 / 50 dependent instructions instead of 95;
//  * a RECORD IS ONE 16-BYTE SLOT: two self-validating 8-byte granules {value half, 16-bit tag | index half} — one lane publishes, ONE wave sweeps all G
//    slots with every load in flight; the winner's ROW is then read from the feature table itself (it is immutable during the chain: plain, cacheable loads,
//    every lane straight into its registers) instead of travelling in every record.  G = n / (512 / LPR) workgroups: 79 / 157 at 20 000 rows.
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = dpp_u32<CTRL>((unsigned)b), hi = dpp_u32<CTRL>((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
constexpr int FQ_NT = 512;
template <int LPR, bool TIMED>
__global__ __launch_bounds__(FQ_NT) void fps_coop_split(FpsCoopArgs a, int slot_shift, int delay, long long* dbg) {
    static_assert(LPR == 2 || LPR == 4, "lanes per row");
    constexpr int A = 8 / LPR, F = 32 / LPR, ROWS = FQ_NT / LPR, NW = FQ_NT / 64, P = 4;      // 64 P >= G slots per sweeping lane
    if (a.dn) a.n = min(a.n, *a.dn);
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto mark = [&](int k) { if (TIMED) { const long long t = (long long)__builtin_readcyclecounter(); tacc[k] += t - tprev; tprev = t; } };
    __shared__ double s_v[2][NW]; __shared__ int s_i[2][NW]; __shared__ int s_c[2], s_gave;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = blockIdx.x, G = a.G, q = tid & (LPR - 1);
    if (tid == 0) s_gave = 0;
    const int row = g * ROWS + tid / LPR;
    // this lane's share of its row: x[k * A + e] = f[row][8 k + A q + e]
    double x[F], rmin = row < a.n ? a.mind[row] : -1.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < A; ++e) x[k * A + e] = row < a.n ? a.f[(size_t)row * 32 + 8 * k + A * q + e] : 0.0;
    int c;
    if (!a.from_partials) c = a.start;
    else {
        double v = -1.0; int i = 0x7fffffff;
        for (int k = tid; k < a.npart; k += FQ_NT) if (better(a.pin[k].v, a.pin[k].i, v, i)) { v = a.pin[k].v; i = a.pin[k].i; }
        wave_argmax(v, i);
        if (lane == 0) { s_v[0][wid] = v; s_i[0][wid] = i; }
        __syncthreads();
        v = s_v[0][0]; c = s_i[0][0];
        for (int w = 1; w < NW; ++w) if (better(s_v[0][w], s_i[0][w], v, c)) { v = s_v[0][w]; c = s_i[0][w]; }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, (2 * G) << slot_shift, 0x00020000);
    // The first polling pass WAITS `delay` units of 64 cycles: a pass that finds a record missing costs another round trip through the fabric (~2300 cycles), and right
    // behind the workgroup's own store the other workgroups' records are still on their way (1.6 passes per pick at 20 000 rows, 1.1 with 16 units).  Fixed delays
    // measured at seven row counts (tools/gpu_fps_delay.sh, profiles/r06_fps_poll_delay.txt): the best is 16 units up to ~70 workgroups and 20 above (20 000 rows 2.78
    // -> 2.60 us per pick, 9 472 rows 2.65 -> 2.37, 65 000 rows 3.29 -> 2.92); 24 is already slower everywhere.  A per-workgroup controller on the miss rate (two units
    // more after a missed pass, one less after sixteen clean picks) was built and drifts upwards — a few per cent of the picks have a straggler whatever the delay.
    const int dly = delay;
    for (int it = 0; it < a.count; ++it) {
        if (g == 0 && tid == 0) a.out[it] = c;
        if (it + 1 == a.count) break;
        const int par = it & 1;
        if (TIMED) tprev = (long long)__builtin_readcyclecounter();
        // the centre's row, this lane's share of it, from the table
        double fc[F];
        {
            const double* src = a.f + (size_t)c * 32 + A * q;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < A; ++e) fc[k * A + e] = src[8 * k + e];
        }
        mark(0);
        double wv = -1.0; int wi = 0x7fffffff;
        {
            double r[A];
#pragma unroll
            for (int e = 0; e < A; ++e) { const double d = x[e] - fc[e]; r[e] = d * d; }
#pragma unroll
            for (int k = 1; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < A; ++e) { const double d = x[k * A + e] - fc[k * A + e]; r[e] += d * d; }
            double p;
            if (LPR == 4) { p = r[0] + r[1]; p = p + dpp_f64<0xB1>(p); p = p + dpp_f64<0x4E>(p); }
            else { p = (r[0] + r[1]) + (r[A > 2 ? 2 : 0] + r[A > 2 ? 3 : 1]); p = p + dpp_f64<0xB1>(p); }
            if (a.use_sqrt) p = sqrt(p);
            if (row < a.n) { if (p < rmin) rmin = p; wv = rmin; wi = row; }
        }
        mark(1);
        wave_argmax(wv, wi);
        if (lane == 0) { s_v[par][wid] = wv; s_i[par][wid] = wi; }
        mark(2);
        __syncthreads();                                   // (1)
        mark(3);
        if (wid == 0) {
            // the eight waves' pairs: one per lane, reduced inside the row of 16 lanes (two DPP passes instead of a chain of eight dependent compares)
            const double pv0 = lane < NW ? s_v[par][lane] : -2.0;
            const int pi0 = lane < NW ? s_i[par][lane] : 0x7fffffff;
            const double bv = row_max_f64(pv0);
            const int bi = row_min_i32(pv0 == bv ? pi0 : 0x7fffffff);
            const unsigned tag = ((unsigned)it % 0xffffu + 1u) << 16;
            const unsigned base = (unsigned)(par * G) << slot_shift;
            if (lane == 0) {
                const unsigned long long b = (unsigned long long)__double_as_longlong(bv);
                fs_u4 v; v.x = (unsigned)b; v.y = tag | ((unsigned)bi >> 16); v.z = (unsigned)(b >> 32); v.w = tag | ((unsigned)bi & 0xffffu);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + ((unsigned)g << slot_shift), 0, 16);      // sc1: write-through
            }
            mark(4);
            // every record of the pick: all of a lane's slots in flight, the missing ones again
            double v = -1.0; int i = 0x7fffffff; bool gave_up = false; int passes = 0;
            for (int k0 = lane; k0 < G && !gave_up; k0 += P * 64) {
                unsigned pend = 0;
#pragma unroll
                for (int j = 0; j < P; ++j) if (k0 + j * 64 < G) pend |= 1u << j;
                long spins = 0;
                for (int z = 0; z < dly; ++z) __builtin_amdgcn_s_sleep(1);
                while (pend) {
                    fs_u4 u[P];
                    if (TIMED) ++passes;
#pragma unroll
                    for (int j = 0; j < P; ++j) if ((pend >> j) & 1u) u[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + ((unsigned)(k0 + j * 64) << slot_shift), 0, 16);
#pragma unroll
                    for (int j = 0; j < P; ++j)
                        if (((pend >> j) & 1u) && (u[j].y & 0xffff0000u) == tag && (u[j].w & 0xffff0000u) == tag) {
                            const double pv = __longlong_as_double((long long)(((unsigned long long)u[j].z << 32) | u[j].x));
                            const int pi = (int)(((u[j].y & 0xffffu) << 16) | (u[j].w & 0xffffu));
                            if (better(pv, pi, v, i)) { v = pv; i = pi; }
                            pend &= ~(1u << j);
                        }
                    if (pend && (++spins > FPS_COOP_SPINS / 16 || (spins % 1024 == 0 && __hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))) { gave_up = true; break; }
                }
            }
            if (gave_up) { atomicOr(&a.sync[2], 1); atomicOr(a.status, 1); s_gave = 1; }
            mark(5);
            if (TIMED) tacc[7] += passes;
            wave_argmax(v, i);
            if (lane == 0) s_c[par] = i;
        }
        __syncthreads();                                   // (2)
        mark(6);
        if (s_gave) { if (g == 0) for (int k = it + 1 + tid; k < a.count; k += FQ_NT) a.out[k] = -1; return; }
        c = s_c[par];
    }
    if (TIMED && tid == 0) for (int k = 0; k < 8; ++k) dbg[(size_t)g * 8 + k] = tacc[k];
}
#endif

#ifndef HIPEMU
// Third form (round 6; the phase clocks of fps_coop_split: of 3.05 us per pick 0.27 are the barrier in front of the workgroup's combine and 0.47 the combine
// itself — eight (value, index) pairs through LDS and a chain of dependent float64 compares by one wave): EVERY WAVE PUBLISHES ITS OWN 16-byte record, the
// eight records of a workgroup side by side in one 128-byte line, and ONE wave per workgroup sweeps all 8 G of them with every load in flight — the
// workgroup-level combine and its barrier are gone (a sweeper reads the same G lines as before); what is left per pick is the row fetch, 50 dependent
// float64 instructions, one wave arg-max, the hand-off (a write-through store becoming visible + ~1.2 loads' round trip through the fabric, ~1.5 us), one
// wave arg-max over the records and one barrier that hands the winner to the other seven waves.
constexpr int FW_NT = 512, FW_LPR = 2, FW_ROWS = FW_NT / FW_LPR, FW_NW = FW_NT / 64, FW_MAXP = 16;      // at most 64 * 16 / 8 = 128 workgroups = 32768 rows
template <bool TIMED>
__global__ __launch_bounds__(FW_NT) void fps_coop_wave(FpsCoopArgs a, long long* dbg) {
    constexpr int A = 8 / FW_LPR, F = 32 / FW_LPR;
    if (a.dn) a.n = min(a.n, *a.dn);
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto mark = [&](int k) { if (TIMED) { const long long t = (long long)__builtin_readcyclecounter(); tacc[k] += t - tprev; tprev = t; } };
    __shared__ double s_v[FW_NW]; __shared__ int s_i[FW_NW]; __shared__ int s_c[2], s_gave;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = blockIdx.x, G = a.G, q = tid & (FW_LPR - 1);
    const int row = g * FW_ROWS + tid / FW_LPR;
    if (tid == 0) s_gave = 0;
    double x[F], rmin = row < a.n ? a.mind[row] : -1.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < A; ++e) x[k * A + e] = row < a.n ? a.f[(size_t)row * 32 + 8 * k + A * q + e] : 0.0;
    int c;
    if (!a.from_partials) c = a.start;
    else {
        double v = -1.0; int i = 0x7fffffff;
        for (int k = tid; k < a.npart; k += FW_NT) if (better(a.pin[k].v, a.pin[k].i, v, i)) { v = a.pin[k].v; i = a.pin[k].i; }
        wave_argmax(v, i);
        if (lane == 0) { s_v[wid] = v; s_i[wid] = i; }
        __syncthreads();
        v = s_v[0]; c = s_i[0];
        for (int w = 1; w < FW_NW; ++w) if (better(s_v[w], s_i[w], v, c)) { v = s_v[w]; c = s_i[w]; }
    }
    __syncthreads();
    const int total = G * FW_NW;                            // records of a pick: [g][wave], 16 bytes each
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, 2 * total * 16, 0x00020000);
    for (int it = 0; it < a.count; ++it) {
        if (g == 0 && tid == 0) a.out[it] = c;
        if (it + 1 == a.count) break;
        const int par = it & 1;
        if (TIMED) tprev = (long long)__builtin_readcyclecounter();
        double fc[F];
        {
            const double* src = a.f + (size_t)c * 32 + A * q;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < A; ++e) fc[k * A + e] = src[8 * k + e];
        }
        double wv = -1.0; int wi = 0x7fffffff;
        {
            double r[A];
#pragma unroll
            for (int e = 0; e < A; ++e) { const double d = x[e] - fc[e]; r[e] = d * d; }
#pragma unroll
            for (int k = 1; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < A; ++e) { const double d = x[k * A + e] - fc[k * A + e]; r[e] += d * d; }
            double p = (r[0] + r[1]) + (r[2] + r[3]);
            p = p + dpp_f64<0xB1>(p);
            if (a.use_sqrt) p = sqrt(p);
            if (row < a.n) { if (p < rmin) rmin = p; wv = rmin; wi = row; }
        }
        wave_argmax(wv, wi);
        const unsigned tag = ((unsigned)it % 0xffffu + 1u) << 16;
        const unsigned base = (unsigned)(par * total) * 16u;
        if (lane == 0) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(wv);
            fs_u4 v; v.x = (unsigned)b; v.y = tag | ((unsigned)wi >> 16); v.z = (unsigned)(b >> 32); v.w = tag | ((unsigned)wi & 0xffffu);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + (unsigned)(g * FW_NW + wid) * 16u, 0, 16);      // sc1: write-through
        }
        mark(0);
        if (wid == 0) {
            // every record of the pick: all of a lane's slots in flight, the missing ones again
            double v = -1.0; int i = 0x7fffffff; bool gave_up = false; int passes = 0;
            unsigned pend = 0;
#pragma unroll
            for (int j = 0; j < FW_MAXP; ++j) if (lane + j * 64 < total) pend |= 1u << j;
            long spins = 0;
            while (pend) {
                fs_u4 u[FW_MAXP];
                if (TIMED) ++passes;
#pragma unroll
                for (int j = 0; j < FW_MAXP; ++j) if ((pend >> j) & 1u) u[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (unsigned)(lane + j * 64) * 16u, 0, 16);
#pragma unroll
                for (int j = 0; j < FW_MAXP; ++j)
                    if (((pend >> j) & 1u) && (u[j].y & 0xffff0000u) == tag && (u[j].w & 0xffff0000u) == tag) {
                        const double pv = __longlong_as_double((long long)(((unsigned long long)u[j].z << 32) | u[j].x));
                        const int pi = (int)(((u[j].y & 0xffffu) << 16) | (u[j].w & 0xffffu));
                        if (better(pv, pi, v, i)) { v = pv; i = pi; }
                        pend &= ~(1u << j);
                    }
                if (pend && (++spins > FPS_COOP_SPINS / 16 || (spins % 1024 == 0 && __hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))) { gave_up = true; break; }
            }
            if (gave_up) { atomicOr(&a.sync[2], 1); atomicOr(a.status, 1); s_gave = 1; }
            mark(1);
            if (TIMED) tacc[7] += passes;
            wave_argmax(v, i);
            if (lane == 0) s_c[par] = i;
            mark(2);
        }
        __syncthreads();
        mark(3);
        if (s_gave) { if (g == 0) for (int k = it + 1 + tid; k < a.count; k += FW_NT) a.out[k] = -1; return; }
        c = s_c[par];
    }
    if (TIMED && tid == 0) for (int k = 0; k < 8; ++k) dbg[(size_t)g * 8 + k] = tacc[k];
}
#endif

// farthest_superpoint_sample (sampler2.py:49-80, the "edcd" branch): FPS over one cloud's superpoints with the
// distance |centre_i - centre_c|^2 + CD(i, c), CD = dir + dir^T from sel_chamfer_dir.  One workgroup, n <= a few thousand.
__global__ __launch_bounds__(256) void fps_superpoint(const double* __restrict__ centres, const double* __restrict__ dir, int n, int start, int count, int* out) {
    __shared__ double s_v[256];
    __shared__ int s_i[256];
    SSDR_DYN_SHARED(double, mind);          // [n]
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) mind[i] = 1.0e10;
    int c = start;
    __syncthreads();
    for (int it = 0; it < count; ++it) {
        if (tid == 0) out[it] = c;
        if (it + 1 == count) break;
        double bv = -1.0; int bi = 0x7fffffff;
        for (int i = tid; i < n; i += 256) {
            const double dx = centres[3 * i] - centres[3 * c], dy = centres[3 * i + 1] - centres[3 * c + 1], dz = centres[3 * i + 2] - centres[3 * c + 2];
            const double ed = (dx * dx + dy * dy) + dz * dz;                                  // np.sum(.., axis=-1) over 3 terms
            const double cd = (i == c) ? 0.0 : dir[(size_t)c * n + i] + dir[(size_t)i * n + c];   // chamfer_distance(..)[i]: av_dist1 + av_dist2
            const double dist = ed + cd;
            double m = mind[i];
            if (dist < m) { m = dist; mind[i] = m; }
            if (better(m, i, bv, bi)) { bv = m; bi = i; }
        }
        s_v[tid] = bv; s_i[tid] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o && better(s_v[tid + o], s_i[tid + o], s_v[tid], s_i[tid])) { s_v[tid] = s_v[tid + o]; s_i[tid] = s_i[tid + o]; } __syncthreads(); }
        c = s_i[0];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void widen_f32_f64(const float* __restrict__ x, size_t n, double* y0, double* y1) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const double v = (double)x[i]; y0[i] = v; if (y1) y1[i] = v; }
}
__global__ __launch_bounds__(256) void fill_double(double* p, int n, double v) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = v;
}

// min_distances against the already-selected centres (kcenterGreedy.py:72-82); also the first partial maxima.  One wave per row, one lane
// per centre (each distance is summed in NumPy's pairwise order by its lane, the minimum over the centres is order-free): a thread per
// row walked the centres one after the other, 0.59 ms for 1400 rows x 240 centres.
__global__ __launch_bounds__(256) void kc_init(const double* __restrict__ f, int n, int D, const int* __restrict__ already, int na, double* mind, Part* pout,
                                               const int* __restrict__ dn = nullptr) {
    __shared__ Part s_p[4];
    if (dn) n = min(n, *dn);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    Part b; b.v = -1.0; b.i = 0x7fffffff;
    for (int i = blockIdx.x * 4 + wid; i < n; i += gridDim.x * 4) {
        const double* fi = f + (size_t)i * D;
        double m = 1.0e300;
        for (int a = lane; a < na; a += 64) {
            const double* fc = f + (size_t)already[a] * D;
            double dist = np_pairwise<double>([&](int k) { const double d = fi[k] - fc[k]; return d * d; }, D);
            m = fmin(m, sqrt(dist));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const long long bits = __double_as_longlong(m);
            const unsigned lo = __shfl_xor((unsigned)bits, o), hi = __shfl_xor((unsigned)(bits >> 32), o);
            m = fmin(m, __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)));
        }
        if (lane == 0) mind[i] = m;
        if (better(m, i, b.v, b.i)) { b.v = m; b.i = i; }
    }
    if (lane == 0) s_p[wid] = b;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) if (better(s_p[w].v, s_p[w].i, b.v, b.i)) b = s_p[w];
        pout[blockIdx.x] = b;
    }
}

// The same for the reference's own round (kcenterGreedy over 20 000 candidates + 4 000 labelled rows seeded with the 4 000: 96 M pairs): kc_init above reads the
// seed's row from L2 for every (row, seed) pair — 24.6 GB through the vector-memory path, ~10 ms.  Here a thread keeps ITS row in registers, the seeds of a slice
// pass through LDS in tiles of 32 (every lane reads the same address: a broadcast), and the slices of the seeds are spread over blockIdx.y and meet in an atomic
// minimum on the bit pattern (non-negative doubles order like their bits).  min over the seeds of sqrt(d) == sqrt(min d), bit for bit (sqrt is monotone and
// correctly rounded), so one root per row is taken afterwards (kc_finish, which also leaves the partial maxima the chain starts from).
constexpr int KT_SEEDS = 32;
__global__ __launch_bounds__(256) void kc_init_tiled(const double* __restrict__ f, int n, const int* __restrict__ already, int na, unsigned long long* mind2, const int* __restrict__ dn) {
    __shared__ double s_seed[KT_SEEDS][32];
    if (dn) n = min(n, *dn);
    const int tid = threadIdx.x, row = blockIdx.x * 256 + tid;
    const int per = (na + (int)gridDim.y - 1) / (int)gridDim.y, a0 = blockIdx.y * per, a1 = min(na, a0 + per);
    double reg[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) reg[k] = row < n ? f[(size_t)row * 32 + k] : 0.0;
    double m = 1.0e300;
    for (int t0 = a0; t0 < a1; t0 += KT_SEEDS) {
        const int cnt = min(KT_SEEDS, a1 - t0);
        __syncthreads();
        for (int e = tid; e < cnt * 32; e += 256) s_seed[e >> 5][e & 31] = f[(size_t)already[t0 + (e >> 5)] * 32 + (e & 31)];
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const double dist = np_pairwise_fixed<32>([&](int k) { const double d = reg[k] - s_seed[j][k]; return d * d; });
            m = fmin(m, dist);
        }
    }
    if (row < n && a1 > a0) atomicMin(&mind2[row], (unsigned long long)__double_as_longlong(m));
}
__global__ __launch_bounds__(256) void kc_finish(unsigned long long* mind2, int n, Part* pout, const int* __restrict__ dn) {
    __shared__ Part s_p[4];
    if (dn) n = min(n, *dn);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double* mind = reinterpret_cast<double*>(mind2);
    double v = -1.0; int i = 0x7fffffff;
    for (int r = blockIdx.x * 256 + tid; r < n; r += gridDim.x * 256) {
        const double mm = sqrt(__longlong_as_double((long long)mind2[r]));
        mind[r] = mm;
        if (better(mm, r, v, i)) { v = mm; i = r; }
    }
    wave_argmax(v, i);
    if (lane == 0) { s_p[wid].v = v; s_p[wid].i = i; }
    __syncthreads();
    if (tid == 0) {
        Part b = s_p[0];
        for (int w = 1; w < 4; ++w) if (better(s_p[w].v, s_p[w].i, b.v, b.i)) b = s_p[w];
        pout[blockIdx.x] = b;
    }
}

// ---- gcn.create_adj (gcn.py:116-191): the adjacency of the trained-GCN branch, torch float32 in the reference ---------------------------
// rows of V: features / max(|features|_2, 1e-12) (torch.nn.functional.normalize)
__global__ __launch_bounds__(256) void ca_normalize(const float* __restrict__ f, int n, int F, float* V) {
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        float ss = 0.f;
        for (int k = lane; k < F; k += 64) { const float v = f[(size_t)i * F + k]; ss += v * v; }
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        for (int k = lane; k < F; k += 64) V[(size_t)i * F + k] = f[(size_t)i * F + k] * inv;
    }
}
// one cloud's block: adj[rows[i]][rows[j]] = <V_i, V_j> * exp(-((float)ED + (float)CD)), minus 1 on the diagonal.  Entries between clouds are
// <V_i, V_j> * exp(-2e10) = +-0 in the reference: the matrix is cleared beforehand.
__global__ __launch_bounds__(256) void ca_block(const float* __restrict__ V, int F, const double* __restrict__ centres, const double* __restrict__ dir,
                                                const int* __restrict__ coff, const long long* __restrict__ boff, const int* __restrict__ rows, int N, float* adj) {
    const int c = blockIdx.z, r0 = coff[c], nc = coff[c + 1] - r0;
    const double* D = dir + boff[c];
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)nc * nc; e += (long)gridDim.x * 256) {
        const int i = (int)(e / nc), j = (int)(e % nc);
        const int gi = rows[r0 + i], gj = rows[r0 + j];
        const double dx = centres[3 * (size_t)(r0 + i)] - centres[3 * (size_t)(r0 + j)], dy = centres[3 * (size_t)(r0 + i) + 1] - centres[3 * (size_t)(r0 + j) + 1],
                     dz = centres[3 * (size_t)(r0 + i) + 2] - centres[3 * (size_t)(r0 + j) + 2];
        const double ed = sqrt((dx * dx + dy * dy) + dz * dz);
        const double cd = i == j ? 0.0 : D[(size_t)i * nc + j] + D[(size_t)j * nc + i];
        float lat = 0.f;
        for (int k = 0; k < F; ++k) lat += V[(size_t)gi * F + k] * V[(size_t)gj * F + k];
        adj[(size_t)gi * N + gj] = lat * expf(-((float)ed + (float)cd)) - (gi == gj ? 1.0f : 0.0f);
    }
}
// column sums (torch.sum(adj, dim=0)), then adj[:, j] *= 1 / sum_j, plus I
__global__ __launch_bounds__(256) void ca_colsum(const float* __restrict__ adj, int N, float* colsum) {
    for (int j = blockIdx.x * 256 + threadIdx.x; j < N; j += gridDim.x * 256) {
        float s = 0.f;
        for (int i = 0; i < N; ++i) s += adj[(size_t)i * N + j];
        colsum[j] = s;
    }
}
__global__ __launch_bounds__(256) void ca_scale(float* adj, int N, const float* __restrict__ colsum) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)N * N; e += (long)gridDim.x * 256) {
        const int i = (int)(e / N), j = (int)(e % N);
        adj[e] = adj[e] * (1.0f / colsum[j]) + (i == j ? 1.0f : 0.0f);
    }
}

// ---- the candidate rule on the device (sampler2.py:533-552 create_file_top_and_all, :745-753 GCN_FPS_sampling's caller) -----------------------
// order[] ranks the regions by descending uncertainty.  cand = the unlabelled ones in that order; the first min(batch, S) of them are "top";
// a cloud offers its first 2 x (its number of top regions) candidates.  The superpoints of cloud c are sp_base[c] .. sp_base[c+1]-1.
// Results: the candidates cloud by cloud (descending uncertainty inside a cloud) followed by the labelled regions (refs), the same rows grouped
// cloud by cloud (candidates then labelled of the cloud: the block structure of the graph), and the counts downstream kernels read instead of
// host-side sizes.  counts[]: 0 n_unl, 1 n_lab, 2 ntot, 3 nmax, 4 sampling_batch, 5 status (1: more rows, 2: more block elements than the
// caller's capacity), 6-7 the block elements as int64.
constexpr int CR_NT = 1024;
__global__ __launch_bounds__(CR_NT) void cand_rank(const int* __restrict__ order, int S, const unsigned char* __restrict__ labelled, int* rankpos, int* cploc, int* chunkcnt) {
    __shared__ int s_w[CR_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = blockIdx.x * CR_NT + tid;
    int sp = -1, v = 0;
    if (r < S) { sp = order[r]; v = labelled[sp] ? 0 : 1; }
    const unsigned long long m = __ballot(v);
    if (lane == 0) s_w[wid] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < CR_NT / 64; ++w) { const int c = s_w[w]; if (w < wid) base += c; tot += c; }
    if (sp >= 0) { rankpos[sp] = r; cploc[sp] = base + __popcll(m & ((1ull << lane) - 1ull)); }
    if (tid == 0) chunkcnt[blockIdx.x] = tot;
}
// exclusive prefix of n ints in place by one workgroup of 256: every thread owns a contiguous piece
__device__ void block_exscan_inplace(int* a, int n, int* s_part /* [257] */) {
    const int tid = threadIdx.x, per = (n + 255) / 256, lo = min(n, tid * per), hi = min(n, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += a[i];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int t = 0; t < 256; ++t) { const int v = s_part[t]; s_part[t] = run; run += v; } s_part[256] = run; }
    __syncthreads();
    int run = s_part[tid];
    for (int i = lo; i < hi; ++i) { const int v = a[i]; a[i] = run; run += v; }
    __syncthreads();
}
__global__ __launch_bounds__(256) void cand_chunkscan(int* chunkcnt, int nchunks) {
    __shared__ int s_part[257];
    block_exscan_inplace(chunkcnt, nchunks, s_part);
}
constexpr int CC_TILE = 2048;
__global__ __launch_bounds__(256) void cand_cloud(const int* __restrict__ rankpos, const int* __restrict__ cploc, const int* __restrict__ chunkoff,
                                                  const unsigned char* __restrict__ labelled, const int* __restrict__ sp_base, int S, int batch_size,
                                                  int* stage, int* ncand, int* ntop) {
    __shared__ __attribute__((aligned(16))) int s_tile[CC_TILE];
    __shared__ int s_red[8];
    const int tid = threadIdx.x, c = blockIdx.x, lo = sp_base[c], n = sp_base[c + 1] - lo;
    const int lim = min(batch_size, S);
    // blockIdx.y: a slice of the cloud's regions (every 256 * gridDim.y-th block of 256).  A region's place inside its cloud is found by counting, n^2 per cloud:
    // nothing at the ~450 regions of the stand-in's tiles, 5 ms on one workgroup at 8 300 (a partition of many small regions, tools/variety_probe.py)
    if ((int)blockIdx.y * 256 >= n && blockIdx.y > 0) return;
    int nv = 0, nt = 0;
    for (int j = tid; j < n; j += 256)
        if (!labelled[lo + j]) { ++nv; const int r = rankpos[lo + j]; nt += (chunkoff[r / CR_NT] + cploc[lo + j]) < lim; }
    block_sum2<256>(nv, nt, s_red);
    const int take = min(2 * nt, nv);
    if (tid == 0 && blockIdx.y == 0) { ncand[c] = take; ntop[c] = nt; }
    for (int j0 = (int)blockIdx.y * 256; j0 < n; j0 += 256 * (int)gridDim.y) {
        const int j = j0 + tid;
        const bool live = j < n && !labelled[lo + j];
        const int rj = live ? rankpos[lo + j] : 0x7fffffff;
        int pos = 0;
        for (int t0 = 0; t0 < n; t0 += CC_TILE) {
            const int m = min(CC_TILE, n - t0);
            __syncthreads();
            for (int k = tid; k < ((m + 3) & ~3); k += 256) s_tile[k] = (k >= m || labelled[lo + t0 + k]) ? 0x7fffffff : rankpos[lo + t0 + k];
            __syncthreads();
            if (live) {          // four rank positions per LDS read (the tile is padded with +infinity to a multiple of four)
                const int4* t4 = reinterpret_cast<const int4*>(s_tile);
                for (int k = 0; k < (m + 3) / 4; ++k) { const int4 v = t4[k]; pos += (v.x < rj) + (v.y < rj) + (v.z < rj) + (v.w < rj); }
            }
        }
        if (live && pos < take) stage[lo + pos] = lo + j;
    }
}
// slices of a cloud's regions for cand_cloud: one below ~1000 regions per cloud on average, up to 32 above (the host knows the totals, not a cloud's own count)
static unsigned cand_slices(size_t S, size_t B) { const size_t avg = S / std::max<size_t>(B, 1); return (unsigned)std::min<size_t>(32, std::max<size_t>(1, avg / 512)); }
__global__ __launch_bounds__(256) void cand_layout(const int* __restrict__ ncand, const int* __restrict__ ntop, const int* __restrict__ lab_off, int B,
                                                   long long cap_rows, long long cap_sq, int* uoff, int* coff, long long* boff, int* counts) {
    __shared__ long long s_p[3][257];
    __shared__ int s_mx[256];
    const int tid = threadIdx.x, per = (B + 255) / 256, lo = min(B, tid * per), hi = min(B, lo + per);
    long long su = 0, sc = 0, sb = 0; int mx = 0, st = 0;
    for (int c = lo; c < hi; ++c) { const long long a = ncand[c] + (lab_off[c + 1] - lab_off[c]); su += ncand[c]; sc += a; sb += a * a; mx = max(mx, (int)a); st += ntop[c]; }
    s_p[0][tid] = su; s_p[1][tid] = sc; s_p[2][tid] = sb; s_mx[tid] = mx;
    __syncthreads();
    // sampling_batch: the tops of all clouds
    __shared__ int s_st[256];
    s_st[tid] = st;
    __syncthreads();
    if (tid == 0) {
        long long r0 = 0, r1 = 0, r2 = 0; int m = 0, t = 0;
        for (int k = 0; k < 256; ++k) { const long long a = s_p[0][k], b = s_p[1][k], q = s_p[2][k]; s_p[0][k] = r0; s_p[1][k] = r1; s_p[2][k] = r2; r0 += a; r1 += b; r2 += q; m = max(m, s_mx[k]); t += s_st[k]; }
        int status = 0;
        if (r1 > cap_rows) status |= 1;
        if (r2 > cap_sq) status |= 2;
        counts[0] = status ? 0 : (int)r0; counts[1] = lab_off[B]; counts[2] = status ? 0 : (int)r1; counts[3] = m; counts[4] = t; counts[5] = status;
        counts[6] = (int)(r2 & 0xffffffffll); counts[7] = (int)(r2 >> 32);
        s_p[0][256] = status;
    }
    __syncthreads();
    const bool bad = s_p[0][256] != 0;           // over capacity: every block is empty, nothing downstream runs; the caller reads the status
    long long u = s_p[0][tid], k = s_p[1][tid], q = s_p[2][tid];
    for (int c = lo; c < hi; ++c) {
        const long long a = ncand[c] + (lab_off[c + 1] - lab_off[c]);
        uoff[c] = (int)u; coff[c] = bad ? 0 : (int)k; boff[c] = bad ? 0 : q;
        u += ncand[c]; k += a; q += a * a;
    }
    if (hi == B && lo < B) { uoff[B] = (int)u; coff[B] = bad ? 0 : (int)k; boff[B] = bad ? 0 : q; }
    if (B == 0 && tid == 0) { uoff[0] = 0; coff[0] = 0; boff[0] = 0; }
}
__global__ __launch_bounds__(256) void cand_fill(const int* __restrict__ stage, const int* __restrict__ sp_base, const int* __restrict__ ncand, const int* __restrict__ uoff,
                                                 const int* __restrict__ coff, const int* __restrict__ lab_off, const int* __restrict__ lab_sp, const int* __restrict__ counts,
                                                 int* sel, int* gsel, int* rows, int* already) {
    if (counts[5]) return;
    const int c = blockIdx.x, nc = ncand[c], u0 = uoff[c], g0 = coff[c], l0 = lab_off[c], nl = lab_off[c + 1] - l0, n_unl = counts[0], lo = sp_base[c];
    for (int k = threadIdx.x; k < nc; k += 256) { const int sp = stage[lo + k]; sel[u0 + k] = sp; gsel[g0 + k] = sp; rows[g0 + k] = u0 + k; }
    for (int k = threadIdx.x; k < nl; k += 256) { const int sp = lab_sp[l0 + k]; sel[n_unl + l0 + k] = sp; gsel[g0 + nc + k] = sp; rows[g0 + nc + k] = n_unl + l0 + k; already[l0 + k] = n_unl + l0 + k; }
}


// ---- the candidate rule of the SHARDED run on the device ------------------------------------------------------------------------------------------------
// Every rank holds the global ranking (regions of all ranks, global id = rank * Smax + local id, padding counted as labelled) and runs the rule over all
// clouds of all ranks (cloud rank * Bmax + b): what it keeps for its own graph are its own clouds; what it needs of the others are their candidate COUNTS
// (where each rank's rows sit in the padded all-gather of the propagated features) and, for the read-back, the global candidate list.
// plan[]: 0 n_unl (this rank), 1 n_lab, 2 rows, 3 largest block, 4 sampling_batch (all ranks), 5 status, 6-7 block elements, 8 n_unl of all ranks,
// 9 bit 0: a rank offers more than nu_max candidates; [16 .. 16 + W) candidates per rank; then [W * nu_max] the rows of the gathered array in
// candidate order, then [W * nu_max] the global candidate list (global region ids).
__global__ __launch_bounds__(256) void cand_global(const int* __restrict__ ncand, const int* __restrict__ ntop, int W, int Bmax, int nu_max, int* guoff, int* plan) {
    __shared__ int s_cnt[64], s_off[65], s_top[256], s_bad;
    const int tid = threadIdx.x, Bg = W * Bmax;
    int t = 0;
    for (int c = tid; c < Bg; c += 256) t += ntop[c];
    s_top[tid] = t;
    if (tid < W) { int n = 0; for (int b = 0; b < Bmax; ++b) n += ncand[tid * Bmax + b]; s_cnt[tid] = n; }
    __syncthreads();
    if (tid == 0) {
        int run = 0, bad = 0, tot = 0;
        for (int r = 0; r < W; ++r) { s_off[r] = run; run += s_cnt[r]; bad |= s_cnt[r] > nu_max; plan[16 + r] = s_cnt[r]; }
        s_off[W] = run;
        for (int k = 0; k < 256; ++k) tot += s_top[k];
        plan[4] = tot; plan[8] = bad ? 0 : run; plan[9] = bad; s_bad = bad;
    }
    __syncthreads();
    // candidate offsets of the global clouds, rank-major (a rank's clouds are consecutive)
    if (tid < W) { int run = s_off[tid]; for (int b = 0; b < Bmax; ++b) { guoff[tid * Bmax + b] = run; run += ncand[tid * Bmax + b]; } }
    if (tid == 0) guoff[Bg] = s_off[W];
    if (s_bad) return;          // a rank offers more than nu_max: the prefix sums run past the 2 * W * nu_max words behind the plan (plan[9] says so)
    int* src = plan + 16 + W;
    for (int r = 0; r < W; ++r) for (int p = tid; p < min(s_cnt[r], nu_max); p += 256) src[s_off[r] + p] = r * nu_max + p;
}
__global__ __launch_bounds__(256) void cand_fill_global(const int* __restrict__ stage, const int* __restrict__ gbase, const int* __restrict__ ncand, const int* __restrict__ guoff,
                                                        const int* __restrict__ plan, int* glist) {
    if (plan[9]) return;
    const int c = blockIdx.x, nc = ncand[c], u0 = guoff[c], lo = gbase[c];
    for (int k = threadIdx.x; k < nc; k += 256) glist[u0 + k] = stage[lo + k];
}
// cand_fill for one rank's clouds: stage holds global region ids, the local tables want local ones
__global__ __launch_bounds__(256) void cand_fill_local(const int* __restrict__ stage, const int* __restrict__ gbase, const int* __restrict__ ncand, const int* __restrict__ uoff,
                                                       const int* __restrict__ coff, const int* __restrict__ lab_off, const int* __restrict__ lab_sp, const int* __restrict__ counts,
                                                       int sub, int* sel, int* gsel, int* rows) {
    if (counts[5]) return;
    const int c = blockIdx.x, nc = ncand[c], u0 = uoff[c], g0 = coff[c], l0 = lab_off[c], nl = lab_off[c + 1] - l0, n_unl = counts[0], lo = gbase[c];
    for (int k = threadIdx.x; k < nc; k += 256) { const int sp = stage[lo + k] - sub; sel[u0 + k] = sp; gsel[g0 + k] = sp; rows[g0 + k] = u0 + k; }
    for (int k = threadIdx.x; k < nl; k += 256) { const int sp = lab_sp[l0 + k]; sel[n_unl + l0 + k] = sp; gsel[g0 + nc + k] = sp; rows[g0 + nc + k] = n_unl + l0 + k; }
}
// rows [0, n) of src -> dst, n on the device
// rows idx[k % n] of in -> row k of out for k < repeat * n (n on the device); nrep receives repeat * n
__global__ __launch_bounds__(256) void sel_gather_rows_rep(const uint32_t* __restrict__ in, const int* __restrict__ idx, int cap, int row_words, uint32_t* __restrict__ out,
                                                           const int* __restrict__ dn, int repeat, int* nrep) {
    const int n = *dn, tot = min(cap, n * repeat);
    if (blockIdx.x == 0 && threadIdx.x == 0) *nrep = tot;
    const long total = (long)tot * row_words;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int r = (int)(e / row_words), c = (int)(e % row_words);
        out[e] = in[(size_t)idx[r % n] * row_words + c];
    }
}
__global__ __launch_bounds__(256) void copy_rows_dn(const double* __restrict__ src, double* __restrict__ dst, int row_len, int cap, const int* __restrict__ dn) {
    const long total = (long)min(cap, *dn) * row_len;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) dst[e] = src[e];
}

// rows [*off, *off + n) of src -> rows [0, n) of dst (the labelled regions' propagated rows sit behind the candidates, whose count is the device's)
__global__ __launch_bounds__(256) void copy_rows_from(const double* __restrict__ src, double* __restrict__ dst, int row_len, int n, const int* __restrict__ off) {
    const long total = (long)n * row_len, o = (long)*off * row_len;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) dst[e] = src[o + e];
}
// The global array of the sharded k-center: [every rank's candidates, in candidate order | every rank's labelled regions, rank by rank] out of the all-gather
// of `per` = nu_max + nl_max rows per rank (candidates padded to nu_max, then the labelled ones); already[] = the labelled rows, *nrows = rows in all
__global__ __launch_bounds__(256) void sel_gather_kc(const uint32_t* __restrict__ in, const int* __restrict__ plan, int W, int nu_max, int per, const int* __restrict__ nlab_off,
                                                     int cap, int row_words, uint32_t* __restrict__ out, int* __restrict__ already, int* nrows) {
    const int n_unl = plan[9] ? 0 : plan[8], n_lab = nlab_off[W], tot = min(cap, n_unl + n_lab);
    const int* idx = plan + 16 + W;
    if (blockIdx.x == 0 && threadIdx.x == 0) *nrows = tot;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n_lab; j += gridDim.x * 256) already[j] = n_unl + j;
    const long total = (long)tot * row_words;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int r = (int)(e / row_words), c = (int)(e % row_words);
        int srow;
        if (r < n_unl) { const int g = idx[r]; srow = (g / nu_max) * per + g % nu_max; }
        else { const int j = r - n_unl; int q = 0; while (q + 1 < W && nlab_off[q + 1] <= j) ++q; srow = q * per + nu_max + (j - nlab_off[q]); }
        out[e] = in[(size_t)srow * row_words + c];
    }
}

struct SelState { RadixSorter sorter; DevBuf keys, vals, hist, mins, dir, rowsum, part, mind, vtmp, pack_xyz, pack_int, cand_i, cand_f, status; bool status_init = false;
                  const double* last_comb = nullptr; size_t last_cap = 0; };

// scratch of the chamfer packer for nrows superpoints in nclouds clouds
int chamfer_pack_buffers(SelState& Q, size_t nrows, size_t nclouds, ChamferPack& P) {
    const size_t slots = (size_t)ITEM * nrows;
    SSDR_TRY(Q.pack_xyz.reserve((3 * 8 + 16 + 8) * slots + 64)); SSDR_TRY(Q.pack_int.reserve(4 * (2 * slots + 5 * nrows + 2 * nclouds) + 64));
    P.x = Q.pack_xyz.as<double>(); P.y = P.x + slots; P.z = P.y + slots; P.src1 = (unsigned long long*)(P.z + slots); P.src0 = (uint4*)(P.src1 + slots);
    P.seg = Q.pack_int.as<int>(); P.cnt = P.seg + slots; P.r2item = (float*)(P.cnt + slots); P.r2sp = P.r2item + nrows;
    P.item_slot = (int*)(P.r2sp + nrows); P.big = P.item_slot + nrows; P.start = P.big + nrows; P.counts = P.start + nrows;
    return SSDR_OK;
}
// plan + fill of the packer for nclouds clouds (coff == nullptr: one cloud of nsingle superpoints); the fill also writes the superpoints' bounding-box
// centres (d_centres [nrows, 3]: what the adjacency kernels read) — rows the device-side count leaves unused are not touched
int chamfer_pack_launch(const ChamferPack& P, const float* d_xyz, const int* d_sp_off, const int* d_sp_pts, const int* d_sel, const int* d_coff, int nsingle,
                        size_t nrows, int n_max, unsigned nclouds, double* d_centres, hipStream_t s) {
    const size_t slots = (size_t)ITEM * nrows;
    SSDR_HIP(hipMemsetAsync(P.seg, 0xff, 4 * slots, s));          // padding slots: no superpoint ...
    SSDR_HIP(hipMemsetAsync(P.cnt, 0, 4 * (slots + nrows), s));   // ... and nothing to sum; r2item (behind cnt) = 0
    hipLaunchKernelGGL(sel_chamfer_plan, dim3(nclouds), dim3(256), 0, s, d_sp_off, d_sel, d_coff, nsingle, P);
    hipLaunchKernelGGL(sel_chamfer_fill, dim3(std::max(1, std::min((n_max + 3) / 4, 1024)), nclouds), dim3(256), 0, s, d_xyz, d_sp_off, d_sp_pts, d_sel, d_coff, nsingle, d_centres, P);
    return SSDR_OK;
}
// one scratch set per stream: calls on different streams may run concurrently (include/ssdr_al.h)
SelState& sst(hipStream_t st = nullptr) { return per_stream<SelState>(st); }

inline int grid_for(long n, int cap = 2048) { return (int)std::max<long>(1, std::min<long>((n + 255) / 256, cap)); }

}  // namespace
}  // namespace ssdr

using namespace ssdr;

extern "C" {

int ssdr_point_uncertainty_dev(const float* d_probs, size_t n, int num_classes, int mode, float* d_unc, int32_t* d_cls, void* stream) {
    if (!d_probs || !d_unc || !d_cls || num_classes < 2 || num_classes > 128 || mode < 0 || mode > 2) { set_error("point_uncertainty: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (n == 0) return SSDR_OK;
    ProfScope prof("sel_point_unc", pick_stream(stream), (4.0 * num_classes + 8.0) * (double)n);
    hipLaunchKernelGGL(sel_point_unc, dim3(grid_for((long)n)), dim3(256), 0, pick_stream(stream), d_probs, (int)n, num_classes, mode, d_unc, d_cls);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int ssdr_region_stats_dev(const float* d_unc, const int32_t* d_cls, const int32_t* d_sp_off, const int32_t* d_sp_pts, size_t S, int num_classes,
                          int mode, double* d_region_unc, int32_t* d_dom, int32_t* d_dom_cnt, void* stream) {
    if (!d_unc || !d_cls || !d_sp_off || !d_sp_pts || !d_region_unc || !d_dom || !d_dom_cnt || num_classes > 32 || mode < 0 || mode > 2) { set_error("region_stats: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (S == 0) return SSDR_OK;
    ProfScope prof("sel_region_stats", pick_stream(stream), 0.0);
    hipLaunchKernelGGL(sel_region_stats_w, dim3((unsigned)std::min<size_t>((S + 3) / 4, (size_t)ctx().num_cu * 16)), dim3(256), 0, pick_stream(stream), d_unc, d_cls, d_sp_off, d_sp_pts,
                       (int)S, num_classes, mode, d_region_unc, d_dom, d_dom_cnt);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int ssdr_dominant_label_dev(const int32_t* d_labels, const int32_t* d_sp_off, const int32_t* d_sp_pts, size_t S, int num_labels,
                            int32_t* d_label, double* d_purity, void* stream) {
    if (!d_labels || !d_sp_off || !d_sp_pts || !d_label || !d_purity || num_labels < 1 || num_labels > 64) { set_error("dominant_label: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (S == 0) return SSDR_OK;
    hipStream_t s = pick_stream(stream); SelState& Q = sst(s);
    SSDR_TRY(Q.hist.reserve(4 * 64)); SSDR_HIP(hipMemsetAsync(Q.hist.p, 0, 4, s));
    ProfScope prof("sel_dominant_label", s, 0.0);
    hipLaunchKernelGGL(sel_dominant_label, dim3((unsigned)std::min<size_t>((S + 3) / 4, (size_t)ctx().num_cu * 16)), dim3(256), 0, s, d_labels, d_sp_off, d_sp_pts, (int)S, num_labels, d_label, d_purity, Q.hist.as<int>());
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int ssdr_clsbal_dev(const int32_t* d_region_class, size_t S, const uint8_t* d_skip, const int32_t* d_selected_class_list, size_t n_selected, double* d_region_unc, void* stream) {
    if (!d_region_class || !d_region_unc || (n_selected && !d_selected_class_list)) { set_error("clsbal: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (S == 0) return SSDR_OK;
    hipStream_t s = pick_stream(stream); SelState& Q = sst(s);
    SSDR_TRY(Q.hist.reserve(4 * 64)); SSDR_HIP(hipMemsetAsync(Q.hist.p, 0, 4 * 64, s));
    ProfScope prof("sel_clsbal", s, 0.0);
    hipLaunchKernelGGL(sel_class_hist, dim3(grid_for((long)(S + n_selected))), dim3(256), 0, s, d_region_class, (int)S, d_skip, d_selected_class_list, (int)n_selected, Q.hist.as<int>());
    hipLaunchKernelGGL(sel_clsbal, dim3(grid_for((long)S)), dim3(256), 0, s, d_region_class, (int)S, d_skip ? -1 : (int)(S + n_selected), Q.hist.as<int>(), d_region_unc);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

/* multi-GPU flavour of add_clsbal: the class histogram is supplied (all-reduced by the caller) */
int ssdr_class_hist_dev(const int32_t* d_region_class, size_t S, const uint8_t* d_skip, const int32_t* d_selected_class_list, size_t n_selected, int32_t* d_hist64, void* stream) {
    if (!d_region_class || !d_hist64 || (n_selected && !d_selected_class_list)) { set_error("class_hist: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream);
    SSDR_HIP(hipMemsetAsync(d_hist64, 0, 4 * 64, s));
    if (S + n_selected == 0) return SSDR_OK;
    hipLaunchKernelGGL(sel_class_hist, dim3(grid_for((long)(S + n_selected))), dim3(256), 0, s, d_region_class, (int)S, d_skip, d_selected_class_list, (int)n_selected, d_hist64);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}
int ssdr_clsbal_hist_dev(const int32_t* d_region_class, size_t S, const int32_t* d_hist64, size_t total, double* d_region_unc, void* stream) {
    if (!d_region_class || !d_hist64 || !d_region_unc || total == 0) { set_error("clsbal_hist: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (S == 0) return SSDR_OK;
    hipLaunchKernelGGL(sel_clsbal, dim3(grid_for((long)S)), dim3(256), 0, pick_stream(stream), d_region_class, (int)S, (int)total, d_hist64, d_region_unc);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int ssdr_rank_regions_dev(const double* d_region_unc, size_t S, int32_t* d_sorted_inds, void* stream) {
    if (!d_region_unc || !d_sorted_inds) { set_error("rank_regions: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (S == 0) return SSDR_OK;
    hipStream_t s = pick_stream(stream); SelState& Q = sst(s);
    ProfScope prof("sel_rank", s, 0.0);
    if (S <= (size_t)RANK_SMALL) {
        hipLaunchKernelGGL(sel_rank_count, dim3((unsigned)((S + 31) / 32)), dim3(256), 0, s, d_region_unc, (int)S, d_sorted_inds);
        SSDR_HIP(hipGetLastError());
        return SSDR_OK;
    }
    SSDR_TRY(Q.keys.reserve(8 * S)); SSDR_TRY(Q.vals.reserve(4 * S));
    hipLaunchKernelGGL(sel_rank_keys, dim3(grid_for((long)S)), dim3(256), 0, s, d_region_unc, (int)S, Q.keys.as<uint64_t>(), Q.vals.as<uint32_t>());
    Q.sorter.wide_high = true;       // the keys' high words are the uncertainties' float bits: every pass up there runs
    SSDR_TRY(Q.sorter.sort(Q.keys.as<uint64_t>(), Q.vals.as<uint32_t>(), (int)S, nullptr, s));
    SSDR_HIP(hipMemcpyAsync(d_sorted_inds, Q.vals.p, 4 * S, hipMemcpyDeviceToDevice, s));
    return SSDR_OK;
}

int ssdr_segment_mean_features_dev(const float* d_feat, int feat_dim, const int32_t* d_cls, const int32_t* d_dom, const int32_t* d_sp_off,
                                   const int32_t* d_sp_pts, const int32_t* d_sel, size_t nsel, float* d_out, void* stream) {
    if (!d_feat || !d_cls || !d_dom || !d_sp_off || !d_sp_pts || !d_out || feat_dim < 1) { set_error("segment_mean_features: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (nsel == 0) return SSDR_OK;
    hipLaunchKernelGGL(sel_segment_mean, dim3(grid_for((long)nsel * feat_dim)), dim3(256), 0, pick_stream(stream), d_feat, feat_dim, d_cls, d_dom, d_sp_off, d_sp_pts, d_sel,
                       (int)nsel, d_out);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

// ---- sharded selection: device-side pieces of the exchanges (ssdr_al/distributed.py) -----------------------------------
__global__ __launch_bounds__(256) void sel_mask_regions(const double* __restrict__ u, const unsigned char* __restrict__ labelled, int S, int Spad, double* out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Spad; i += gridDim.x * 256)
        out[i] = (i < S && !labelled[i]) ? u[i] : __longlong_as_double((long long)0xfff0000000000000ULL);      // -inf: labelled regions and padding sort last
}
__global__ __launch_bounds__(256) void sel_gather_rows(const uint32_t* __restrict__ in, const int* __restrict__ idx, int n, int row_words, uint32_t* __restrict__ out,
                                                       const int* __restrict__ dn = nullptr) {
    if (dn) n = min(n, *dn);
    const long total = (long)n * row_words;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int r = (int)(e / row_words), c = (int)(e % row_words);
        out[e] = in[(size_t)idx[r] * row_words + c];
    }
}

int ssdr_mask_regions_dev(const double* d_region_unc, const uint8_t* d_labelled, size_t S, size_t S_padded, double* d_out, void* stream) {
    if (!d_region_unc || !d_labelled || !d_out || S_padded < S) { set_error("mask_regions: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (S_padded == 0) return SSDR_OK;
    hipLaunchKernelGGL(sel_mask_regions, dim3(grid_for((long)S_padded)), dim3(256), 0, pick_stream(stream), d_region_unc, d_labelled, (int)S, (int)S_padded, d_out);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int ssdr_gather_rows_dev(const void* d_in, const int32_t* d_idx, size_t n, size_t row_bytes, void* d_out, void* stream) {
    if (!d_in || !d_idx || !d_out || row_bytes % 4) { set_error("gather_rows: bad arguments (row_bytes must be a multiple of 4)"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (n == 0) return SSDR_OK;
    hipLaunchKernelGGL(sel_gather_rows, dim3(grid_for((long)n * (long)(row_bytes / 4))), dim3(256), 0, pick_stream(stream), (const uint32_t*)d_in, d_idx, (int)n,
                       (int)(row_bytes / 4), (uint32_t*)d_out);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int ssdr_widen_f32_f64_dev(const float* d_x, size_t n, double* d_y0, double* d_y1, void* stream) {
    if (!d_x || !d_y0) { set_error("widen_f32_f64: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (n == 0) return SSDR_OK;
    hipLaunchKernelGGL(widen_f32_f64, dim3(grid_for((long)n)), dim3(256), 0, pick_stream(stream), d_x, n, d_y0, d_y1);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int ssdr_cloud_graph_dev(const float* d_xyz, const int32_t* d_sp_off, const int32_t* d_sp_pts, const int32_t* d_sel, size_t nsel, size_t max_sp_size,
                         int gcn_top, double* d_centres, double* d_cd_dir, double* d_adj, void* stream) {
    if (!d_xyz || !d_sp_off || !d_sp_pts || !d_sel || !d_centres || !d_cd_dir || !d_adj || max_sp_size == 0) { set_error("cloud_graph: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (nsel == 0) return SSDR_OK;
    hipStream_t s = pick_stream(stream); SelState& Q = sst(s);
    const int n = (int)nsel;
    SSDR_TRY(Q.rowsum.reserve(8 * nsel));
    ChamferPack P; SSDR_TRY(chamfer_pack_buffers(Q, nsel, 1, P));
    SSDR_TRY(chamfer_pack_launch(P, d_xyz, d_sp_off, d_sp_pts, d_sel, nullptr, n, nsel, n, 1, d_centres, s));
    SSDR_TRY(chamfer_dir_launch(d_xyz, d_sp_off, d_sp_pts, d_sel, n, d_centres, d_cd_dir, P, s));
    hipLaunchKernelGGL(sel_adj_build, dim3(std::min(n, 2048)), dim3(256), 0, s, d_centres, d_cd_dir, n, d_adj, Q.rowsum.as<double>());
    hipLaunchKernelGGL(sel_adj_norm, dim3(grid_for((long)n * n)), dim3(256), 0, s, Q.rowsum.as<double>(), n, d_adj);
    if (gcn_top > 0) hipLaunchKernelGGL(sel_adj_topk, dim3(std::max(1, std::min((n + 3) / 4, 2048))), dim3(256), 0, s, d_adj, n, gcn_top);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int ssdr_cloud_graph_batch_dev(const float* d_xyz, const int32_t* d_sp_off, const int32_t* d_sp_pts, const int32_t* d_sel, const int32_t* d_coff,
                               const int64_t* d_boff, size_t num_clouds, size_t n_total, size_t n_max, int gcn_top,
                               double* d_centres, double* d_cd_dir, double* d_adj, void* stream) {
    if (!d_xyz || !d_sp_off || !d_sp_pts || !d_sel || !d_coff || !d_boff || !d_centres || !d_cd_dir || !d_adj || num_clouds > 65535) { set_error("cloud_graph_batch: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (num_clouds == 0 || n_total == 0 || n_max == 0) return SSDR_OK;
    hipStream_t s = pick_stream(stream); SelState& Q = sst(s);
    const int nm = (int)n_max; const unsigned nc = (unsigned)num_clouds;
    SSDR_TRY(Q.rowsum.reserve(8 * n_total));
    ChamferPack P; SSDR_TRY(chamfer_pack_buffers(Q, n_total, num_clouds, P));
    SSDR_TRY(chamfer_pack_launch(P, d_xyz, d_sp_off, d_sp_pts, d_sel, d_coff, 0, n_total, nm, nc, d_centres, s));
    SSDR_TRY(chamfer_dir_batch_launch(d_xyz, d_sp_off, d_sp_pts, d_sel, d_coff, (const long long*)d_boff, nm, nc, d_centres, d_cd_dir, P, s));
    hipLaunchKernelGGL(sel_adj_build_batch, dim3(std::min(nm, 1024), 1, nc), dim3(256), 0, s, d_centres, d_cd_dir, d_coff, (const long long*)d_boff, d_adj, Q.rowsum.as<double>());
    hipLaunchKernelGGL(sel_adj_norm_batch, dim3(grid_for((long)nm * nm, 256), 1, nc), dim3(256), 0, s, Q.rowsum.as<double>(), d_coff, (const long long*)d_boff, d_adj);
    if (gcn_top > 0) hipLaunchKernelGGL(sel_adj_topk_batch, dim3(std::max(1, std::min((nm + 3) / 4, 1024)), 1, nc), dim3(256), 0, s, d_adj, d_coff, (const long long*)d_boff, gcn_top);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int ssdr_propagate_batch_dev(const double* d_adj, const int32_t* d_coff, const int64_t* d_boff, size_t num_clouds, size_t n_max, const int32_t* d_rows,
                             const double* d_vin, int feat_dim, double* d_vout, double* d_comb, void* stream) {
    if (!d_adj || !d_coff || !d_boff || !d_rows || !d_vin || !d_vout || !d_comb || feat_dim < 1 || num_clouds > 65535) { set_error("propagate_batch: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (num_clouds == 0 || n_max == 0) return SSDR_OK;
    hipLaunchKernelGGL(sel_propagate_batch, dim3(grid_for((long)n_max * feat_dim, 256), 1, (unsigned)num_clouds), dim3(256), 0, pick_stream(stream), d_adj, d_coff,
                       (const long long*)d_boff, d_rows, d_vin, feat_dim, d_vout, d_comb);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int ssdr_propagate_dev(const double* d_adj, size_t n, const int32_t* d_rows, const double* d_vin, int feat_dim, double* d_vout, double* d_comb, void* stream) {
    if (!d_adj || !d_rows || !d_vin || !d_vout || !d_comb || feat_dim < 1) { set_error("propagate: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (n == 0) return SSDR_OK;
    hipLaunchKernelGGL(sel_propagate, dim3(grid_for((long)n * feat_dim)), dim3(256), 0, pick_stream(stream), d_adj, (int)n, d_rows, d_vin, feat_dim, d_vout, d_comb);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

#ifndef HIPEMU
// Co-operative chains of DIFFERENT streams share the chip: each is sized against the workgroups that can be resident together, so the sum of the
// grids in flight must stay inside that number too (three selection streams with `--select-lag 2` each launched "half the resident grid": not
// co-resident -> 0.3 s of polling -> abort).  Every cooperative launch leaves an event; a new one first drops the finished ones from the account
// and, while the sum would pass the budget, makes its stream wait for the oldest (device-side: the host never blocks).
struct CoopFlight { hipEvent_t ev; int g; };
static std::mutex g_coop_mu;
static std::vector<CoopFlight> g_coop_flights;
static std::vector<hipEvent_t> g_coop_pool;
// One admission: holds the account's lock from the admit to the recorded event of the launch it admitted (two threads can no longer both pass
// on the same sum), and never takes a chain out of the account before its event reports it finished: a chain that stream A was made to wait for
// still runs, and a stream C admitted right afterwards must see it in the sum (and waits for it as well) — the polling-abort case the account exists for.
struct CoopGuard {
    std::unique_lock<std::mutex> lk;
    int admit(hipStream_t s, int g, int budget) {
        lk = std::unique_lock<std::mutex>(g_coop_mu);
        static const int env_budget = [] { const char* e = getenv("SSDR_FPS_COOP_BUDGET"); return e ? atoi(e) : 0; }();      // tests: a budget that forces the serialisation
        if (env_budget > 0) budget = std::max(env_budget, g);
        size_t keep = 0; int sum = 0;
        for (size_t i = 0; i < g_coop_flights.size(); ++i) {
            if (hipEventQuery(g_coop_flights[i].ev) == hipSuccess) { g_coop_pool.push_back(g_coop_flights[i].ev); continue; }      // finished: its event may be re-recorded
            g_coop_flights[keep++] = g_coop_flights[i]; sum += g_coop_flights[i].g;
        }
        g_coop_flights.resize(keep);
        (void)hipGetLastError();          // (hipErrorNotReady of the queries is not an error)
        for (size_t k = 0; k < g_coop_flights.size() && sum + g > budget; ++k) {      // oldest first; the flights stay in the account
            SSDR_HIP(hipStreamWaitEvent(s, g_coop_flights[k].ev, 0));
            sum -= g_coop_flights[k].g;
        }
        return SSDR_OK;
    }
    int launched(hipStream_t s, int g) {
        if (!lk.owns_lock()) lk = std::unique_lock<std::mutex>(g_coop_mu);
        hipEvent_t ev;
        if (!g_coop_pool.empty()) { ev = g_coop_pool.back(); g_coop_pool.pop_back(); }
        else SSDR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        SSDR_HIP(hipEventRecord(ev, s));
        g_coop_flights.push_back({ev, g});
        lk.unlock();
        return SSDR_OK;
    }
};
#endif

// d_n (optional): the row count on the device; n is then the bound the launch shapes are chosen by
static int fps_like(const double* d_feat, size_t n, int D, const int32_t* d_already, size_t na, int start, size_t count, int use_sqrt, int32_t* d_out, hipStream_t s,
                    const int* d_n = nullptr) {
    SelState& Q = sst(s);
    // SURVEY 8d (F4 / F5): per pick n * D * 8 bytes of features + n * 8 of distances read and written (n = the capacity here: the count is the device's)
    ProfScope prof("fps_chain", s, (double)count * ((double)n * D * 8.0 + 16.0 * (double)n));
    int nb = grid_for((long)n, ctx().num_cu * 2);
    // seeded single-workgroup paths: kc_init takes a wave per row and its partial maxima are read once — as many workgroups as give every
    // wave a few rows (6 workgroups for 1400 rows left the kernel latency-bound at 0.57 ms)
    if (d_already && na && n <= 16384) nb = std::max(nb, (int)std::min<size_t>((n + 15) / 16, 2048));
    SSDR_TRY(Q.part.reserve(sizeof(Part) * 2 * (size_t)nb)); SSDR_TRY(Q.mind.reserve(8 * n));
    Part* p0 = Q.part.as<Part>(); Part* p1 = p0 + nb;
    static const bool kc_tiled = [] { const char* e = getenv("SSDR_KC_TILED"); return !e || e[0] != '0'; }();      // (A/B: 0 keeps kc_init at every size)
    if (kc_tiled && d_already && na && D == 32 && (double)n * (double)na > 4.0e6) {       // the reference's own round: rows in registers, seeds through LDS, seed slices over blockIdx.y
        const int rb = (int)((n + 255) / 256), ys = (int)std::max<size_t>(1, std::min<size_t>((na + KT_SEEDS - 1) / KT_SEEDS, (size_t)std::max(1, 2 * ctx().num_cu / rb)));
        SSDR_HIP(hipMemsetAsync(Q.mind.p, 0x7f, 8 * n, s));                    // 0x7f7f...: a positive double above every squared distance
        hipLaunchKernelGGL(kc_init_tiled, dim3(rb, ys), dim3(256), 0, s, d_feat, (int)n, d_already, (int)na, Q.mind.as<unsigned long long>(), d_n);
        hipLaunchKernelGGL(kc_finish, dim3(nb), dim3(256), 0, s, Q.mind.as<unsigned long long>(), (int)n, p1, d_n);
        SSDR_HIP(hipGetLastError());
    } else if (d_already && na) hipLaunchKernelGGL(kc_init, dim3(nb), dim3(256), 0, s, d_feat, (int)n, D, d_already, (int)na, Q.mind.as<double>(), p1, d_n);
    else hipLaunchKernelGGL(fill_double, dim3(grid_for((long)n)), dim3(256), 0, s, Q.mind.as<double>(), (int)n, 1.0e10);   // fps_gcn_cpu.py:135
    const bool seeded = d_already && na;
    if (D == 32 && n <= 1536) {   // register-resident single workgroup
        const int fp = seeded ? 1 : 0;
        // (1024 threads — four waves per SIMD issue a float64 instruction every 5.5 cycles, the two of this form every 6.5, tools/micro/valu_rate.hip — with row tid in
        // registers and rows 1024.. in LDS was built and measured: 2.42 against 2.39 ms for the selection stage; the barrier over sixteen waves takes the gain back)
        if (n <= 512) hipLaunchKernelGGL((fps_block_reg<32, 1, 512>), dim3(1), dim3(512), 0, s, d_feat, (int)n, fp, start, use_sqrt, p1, nb, Q.mind.as<double>(), (int)count, d_out, d_n);
        else if (n <= 1024) hipLaunchKernelGGL((fps_block_reg<32, 2, 512>), dim3(1), dim3(512), 0, s, d_feat, (int)n, fp, start, use_sqrt, p1, nb, Q.mind.as<double>(), (int)count, d_out, d_n);
        else hipLaunchKernelGGL((fps_block_reg<32, 3, 512>), dim3(1), dim3(512), 0, s, d_feat, (int)n, fp, start, use_sqrt, p1, nb, Q.mind.as<double>(), (int)count, d_out, d_n);
        SSDR_HIP(hipGetLastError());
        return SSDR_OK;
    }
#ifndef HIPEMU
    // cooperative kernels (G workgroups that meet at a counter per pick): only above the sizes one workgroup sweeps well (the 160 x 129 / 1000 x 129
    // k-center shapes keep the 1024-thread fps_block), and only with G workgroups the occupancy query says are resident together — checked, not assumed
    int coop_g = 0, coop_budget = 0; bool coop_reg = false;
    // The form of the 32-d chain (round 6; us per pick at 2368 / 4736 / 9472 / 20000 / 24000 rows on one box, profiles/r06_fps_*.txt): rounds 3-5 (polled granules up to 24
    // workgroups, drained record + counter above) 2.85 / 3.21 / 3.68 / 5.22 / 5.36; rows split over two lanes with 16-byte records (fps_coop_split<2>) 2.47 / 2.49 / 2.56 /
    // 2.61 / 2.73 — the default.  SSDR_FPS_COOP_SWEEP selects the others for A/B runs: 0 rounds 3-5, 1 swept 544-byte records, 2 / 3 the same among one XCD's workgroups
    // (plain / write-through stores), 5 rows over four lanes, 6 a record per wave.
    static const int sweep_env = [] { const char* e = getenv("SSDR_FPS_COOP_SWEEP"); return e ? atoi(e) : -1; }();
    const int sweep = sweep_env >= 0 ? sweep_env : 4;
    const int split_lpr = sweep == 4 ? 2 : sweep == 5 ? 4 : 0;
    if (D == 32 && n > 1536 && n <= (size_t)FR_ROWS * (size_t)(ctx().num_cu / 2)) { coop_g = (int)((n + FR_ROWS - 1) / FR_ROWS); coop_reg = true; }
    else if (n > 4096 && n <= (size_t)FC_NT * FC_PPT * (size_t)(ctx().num_cu / 2)) coop_g = (int)std::min<size_t>((size_t)ctx().num_cu / 2, (n + 2 * FC_NT - 1) / (2 * FC_NT));
    const bool split = coop_reg && split_lpr && (n + FQ_NT / split_lpr - 1) / (FQ_NT / split_lpr) <= 256;      // (a sweeping lane takes four slots)
    if (split) coop_g = (int)((n + FQ_NT / split_lpr - 1) / (FQ_NT / split_lpr));
    if (coop_g) {
        static const int force_g = [] { const char* e = getenv("SSDR_FPS_COOP_G"); return e ? atoi(e) : 0; }();      // tests: a grid above residency must be reported, not believed
        if (force_g > 0 && coop_g > 0 && !split) coop_g = force_g;
        else {
            int per_cu = 0;
            const hipError_t oe = split ? (split_lpr == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_coop_split<2, false>, FQ_NT, 0)
                                                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_coop_split<4, false>, FQ_NT, 0))
                                : coop_reg ? (coop_g > 24 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_coop_reg, FR_NT, 8 * (size_t)coop_g * FR_REC)
                                                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_coop_tag, FR_NT, 4 * 2 * (size_t)coop_g * FT_WORDS))
                                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_coop, FC_NT, 0);
            // half of what the query admits: the query is known to answer one block per CU high at some register counts (MI355X_MICROARCH.md), and
            // other chains / stage kernels share the CUs
            if (oe != hipSuccess || (long)per_cu * ctx().num_cu / 2 < coop_g) coop_g = 0;
            coop_budget = (int)((long)per_cu * ctx().num_cu / 2);
        }
    }
    const bool coop_ok = coop_g > 0;
    CoopGuard coop;
    if (coop_ok) SSDR_TRY(coop.admit(s, coop_g, std::max(coop_budget, coop_g)));
    if (coop_ok && !Q.status_init) { SSDR_TRY(Q.status.reserve(64)); SSDR_HIP(hipMemsetAsync(Q.status.p, 0, 64, s)); Q.status_init = true; }
#else
    const bool coop_ok = false;
#endif
    if (n <= 16384 && !coop_ok) {     // one CU sweeps the candidates faster than a launch per iteration costs
        if (D == 32) hipLaunchKernelGGL((fps_block<32>), dim3(1), dim3(1024), 0, s, d_feat, (int)n, D, seeded ? 1 : 0, start, use_sqrt, p1, nb, Q.mind.as<double>(), (int)count, d_out, d_n);
        else hipLaunchKernelGGL((fps_block<0>), dim3(1), dim3(1024), 0, s, d_feat, (int)n, D, seeded ? 1 : 0, start, use_sqrt, p1, nb, Q.mind.as<double>(), (int)count, d_out, d_n);
        SSDR_HIP(hipGetLastError());
        return SSDR_OK;
    }
#ifndef HIPEMU
    if (coop_ok && coop_reg) {       // rows in registers, partials that carry the candidate's features
        const int G = coop_g;
        SSDR_TRY(Q.vtmp.reserve(8 * 2 * (size_t)G * FR_REC * 2 + 64));      // (the granule form: 68 words of 8 bytes per record)
        Part* part = Q.vtmp.as<Part>(); int* sync = reinterpret_cast<int*>(Q.vtmp.as<char>() + 8 * 2 * (size_t)G * FR_REC * 2);
        SSDR_HIP(hipMemsetAsync(sync, 0, 16, s));
        static std::once_flag once;
        std::call_once(once, [] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_coop_reg), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * FR_REC * 128); });
        if (!split && 8 * (size_t)G * FR_REC > 8 * (size_t)FR_REC * 128) { set_error("fps: %d cooperative workgroups exceed the record table of the kernel (128)", G); return SSDR_ERR_INVALID; }
        FpsCoopArgs a{d_feat, (int)n, D, seeded ? 1 : 0, start, use_sqrt, p1, nb, Q.mind.as<double>(), (int)count, d_out, part, sync, G, Q.status.as<int>(), d_n};
        // two hand-off forms, measured (tools/gpu_fps.sh, us per pick at 2368 / 4736 / 9472 / 20000 rows): self-validating granules 3.46 / 3.92 / 4.71 / 6.30,
        // drained record + counter 3.90 / 4.14 / 4.77 / 5.74 — the granule form polls 68 words per record and loses from ~24 workgroups on
        static const int form_env = [] { const char* e = getenv("SSDR_FPS_COOP_COUNTER"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
        if (sweep == 6 && (n + FW_ROWS - 1) / FW_ROWS <= 64 * FW_MAXP / FW_NW) {          // a record per wave, one sweeping wave per workgroup
            const int G2 = (int)((n + FW_ROWS - 1) / FW_ROWS);
            const size_t recb = (size_t)2 * G2 * FW_NW * 16;
            SSDR_TRY(Q.vtmp.reserve(recb + 64));
            SSDR_HIP(hipMemsetAsync(Q.vtmp.p, 0, recb + 64, s));      // tags start at 0: no pick has that number; abort word
            a.part = Q.vtmp.as<Part>(); a.sync = reinterpret_cast<int*>(Q.vtmp.as<char>() + recb); a.G = G2;
            static const bool dbg_env = getenv("SSDR_FPS_DBG") != nullptr;
            if (dbg_env) {
                static DevBuf dbgbuf; SSDR_TRY(dbgbuf.reserve(8 * 8 * 512)); SSDR_HIP(hipMemsetAsync(dbgbuf.p, 0, 8 * 8 * 512, s));
                hipLaunchKernelGGL(fps_coop_wave<true>, dim3(G2), dim3(FW_NT), 0, s, a, dbgbuf.as<long long>());
                SSDR_HIP(hipStreamSynchronize(s));
                std::vector<long long> h(8 * 512); SSDR_HIP(hipMemcpy(h.data(), dbgbuf.p, 8 * 8 * 512, hipMemcpyDeviceToHost));
                const char* nm[8] = {"fetch+dist+argmax+store", "sweep", "argmax", "barrier", "-", "-", "-", "passes"};
                for (int k = 0; k < 8; ++k) {
                    long long mn = 1LL << 62, mx = 0, sum = 0;
                    for (int g2 = 0; g2 < G2; ++g2) { const long long v = h[(size_t)g2 * 8 + k]; mn = std::min(mn, v); mx = std::max(mx, v); sum += v; }
                    fprintf(stderr, "fps_coop_wave G=%d %-24s per pick: mean %.1f min %.1f max %.1f\n", G2, nm[k], (double)sum / G2 / count, (double)mn / count, (double)mx / count);
                }
                return coop.launched(s, G2);
            }
            hipLaunchKernelGGL(fps_coop_wave<false>, dim3(G2), dim3(FW_NT), 0, s, a, (long long*)nullptr);
            SSDR_HIP(hipGetLastError());
            return coop.launched(s, G2);
        }
        if (split) {          // rows split over 2 / 4 lanes, 16-byte records, the winner's row from the table
            const int lpr = split_lpr, G2 = G;
            static const int shift = [] { const char* e = getenv("SSDR_FPS_SLOT_SHIFT"); return e ? atoi(e) : 6; }();      // a record's slot: 16 bytes, or a line / several of its own
            static const int delay_env = [] { const char* e = getenv("SSDR_FPS_DELAY"); return e ? atoi(e) : -1; }();      // (development) s_sleep units in front of the first polling pass
            const int delay = delay_env >= 0 ? delay_env : (G2 >= 72 ? 20 : 16);
            const int launch_g = G2;
            const size_t recb = ((size_t)2 * G2) << shift;
            SSDR_TRY(Q.vtmp.reserve(recb + 64));
            SSDR_HIP(hipMemsetAsync(Q.vtmp.p, 0, recb + 64, s));      // tags start at 0: no pick has that number; abort word
            a.part = Q.vtmp.as<Part>(); a.sync = reinterpret_cast<int*>(Q.vtmp.as<char>() + recb); a.G = G2;
            static const bool dbg_env = getenv("SSDR_FPS_DBG") != nullptr;
            if (dbg_env) {
                static DevBuf dbgbuf; SSDR_TRY(dbgbuf.reserve(8 * 8 * 512)); SSDR_HIP(hipMemsetAsync(dbgbuf.p, 0, 8 * 8 * 512, s));
                if (lpr == 2) hipLaunchKernelGGL((fps_coop_split<2, true>), dim3(launch_g), dim3(FQ_NT), 0, s, a, shift, delay, dbgbuf.as<long long>());
                else hipLaunchKernelGGL((fps_coop_split<4, true>), dim3(launch_g), dim3(FQ_NT), 0, s, a, shift, delay, dbgbuf.as<long long>());
                SSDR_HIP(hipStreamSynchronize(s));
                std::vector<long long> h(8 * 512); SSDR_HIP(hipMemcpy(h.data(), dbgbuf.p, 8 * 8 * 512, hipMemcpyDeviceToHost));
                const char* nm[8] = {"row fetch", "dist", "wave_argmax", "barrier1", "combine+store", "sweep", "argmax+barrier2", "passes"};
                for (int k = 0; k < 8; ++k) {
                    long long mn = 1LL << 62, mx = 0, sum = 0;
                    for (int g2 = 0; g2 < G2 && g2 < 512; ++g2) { const long long v = h[(size_t)g2 * 8 + k]; mn = std::min(mn, v); mx = std::max(mx, v); sum += v; }
                    fprintf(stderr, "fps_coop_split<%d> G=%d %-16s per pick: mean %.1f min %.1f max %.1f\n", lpr, G2, nm[k], (double)sum / std::min(G2, 512) / count, (double)mn / count, (double)mx / count);
                }
                return coop.launched(s, G2);
            }
            if (lpr == 2) hipLaunchKernelGGL((fps_coop_split<2, false>), dim3(launch_g), dim3(FQ_NT), 0, s, a, shift, delay, (long long*)nullptr);
            else hipLaunchKernelGGL((fps_coop_split<4, false>), dim3(launch_g), dim3(FQ_NT), 0, s, a, shift, delay, (long long*)nullptr);
            SSDR_HIP(hipGetLastError());
            return coop.launched(s, G2);
        }
        if (sweep >= 1 && sweep <= 3 && G <= 64) {          // the swept 544-byte records (A/B runs)
            const bool team = sweep >= 2;
            SSDR_TRY(Q.vtmp.reserve(16 * 2 * (size_t)G * FS_SLOTS + 64));
            int* sync2 = reinterpret_cast<int*>(Q.vtmp.as<char>() + 16 * 2 * (size_t)G * FS_SLOTS);
            SSDR_HIP(hipMemsetAsync(Q.vtmp.p, 0, 16 * 2 * (size_t)G * FS_SLOTS + 64, s));      // tags start at 0: no pick has that number; abort word, team words
            static std::once_flag once3;
            std::call_once(once3, [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_coop_sweep<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * FT_WORDS * 64);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_coop_sweep<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * FT_WORDS * 64); });
            a.part = Q.vtmp.as<Part>(); a.sync = sync2;
            static const bool dbg_env = getenv("SSDR_FPS_DBG") != nullptr;
            if (dbg_env) {       // development: where a pick's time goes, per workgroup (wave 0's clock)
                static DevBuf dbgbuf; SSDR_TRY(dbgbuf.reserve(8 * 8 * 64)); SSDR_HIP(hipMemsetAsync(dbgbuf.p, 0, 8 * 8 * 64, s));
                hipLaunchKernelGGL(fps_coop_sweep<true>, dim3(team ? 8 * (G + 2) : G), dim3(FR_NT), 4 * 2 * (size_t)G * FT_WORDS, s, a, team ? 1 : 0, sweep == 2 ? 1 : 0, dbgbuf.as<long long>());
                SSDR_HIP(hipStreamSynchronize(s));
                std::vector<long long> h(8 * 64); SSDR_HIP(hipMemcpy(h.data(), dbgbuf.p, 8 * 8 * 64, hipMemcpyDeviceToHost));
                const char* nm[8] = {"dist", "wave_argmax", "pub", "barrier1", "store", "sweep", "barrier2+argmax", "passes"};
                for (int k = 0; k < 8; ++k) {
                    long long mn = 1LL << 62, mx = 0, sum = 0;
                    for (int g2 = 0; g2 < G; ++g2) { const long long v = h[(size_t)g2 * 8 + k]; mn = std::min(mn, v); mx = std::max(mx, v); sum += v; }
                    fprintf(stderr, "fps_coop_sweep G=%d %-16s per pick: mean %.1f min %.1f max %.1f (%s)\n", G, nm[k], (double)sum / G / count, (double)mn / count, (double)mx / count, k == 7 ? "passes" : "s_memtime ticks");
                }
                return coop.launched(s, G);
            }
            hipLaunchKernelGGL(fps_coop_sweep<false>, dim3(team ? 8 * (G + 2) : G), dim3(FR_NT), 4 * 2 * (size_t)G * FT_WORDS, s, a, team ? 1 : 0, sweep == 2 ? 1 : 0, (long long*)nullptr);
            SSDR_HIP(hipGetLastError());
            return coop.launched(s, G);
        }
        const bool counter_form = form_env >= 0 ? form_env == 1 : G > 24;
        if (!counter_form) {
            SSDR_HIP(hipMemsetAsync(part, 0, 8 * 2 * (size_t)G * FR_REC * 2, s));      // tags start at 0: no pick has that number
            static std::once_flag once2;
            std::call_once(once2, [] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_coop_tag), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * FT_WORDS * 128); });
            hipLaunchKernelGGL(fps_coop_tag, dim3(G), dim3(FR_NT), 4 * 2 * (size_t)G * FT_WORDS, s, a);
            SSDR_HIP(hipGetLastError());
            return coop.launched(s, G);
        }
        hipLaunchKernelGGL(fps_coop_reg, dim3(G), dim3(FR_NT), 8 * (size_t)G * FR_REC, s, a);
        SSDR_HIP(hipGetLastError());
        return coop.launched(s, G);
    }
    if (coop_ok) {          // one launch: co-resident workgroups meeting at a counter per pick
        const int G = coop_g;
        SSDR_TRY(Q.vtmp.reserve(sizeof(Part) * 2 * (size_t)G + 64));
        Part* part = Q.vtmp.as<Part>(); int* sync = reinterpret_cast<int*>(part + 2 * G);
        SSDR_HIP(hipMemsetAsync(sync, 0, 16, s));
        FpsCoopArgs a{d_feat, (int)n, D, seeded ? 1 : 0, start, use_sqrt, p1, nb, Q.mind.as<double>(), (int)count, d_out, part, sync, G, Q.status.as<int>(), d_n};
        hipLaunchKernelGGL(fps_coop, dim3(G), dim3(FC_NT), 0, s, a);
        SSDR_HIP(hipGetLastError());
        return coop.launched(s, G);
    }
#endif
    for (size_t it = 0; it < count; ++it) {
        Part* pin = (it & 1) ? p0 : p1; Part* pout = (it & 1) ? p1 : p0;
        const bool last = it + 1 == count;
        // k-center starts from the arg-max of the seeded distances; FPS from `start`
        hipLaunchKernelGGL(fps_step, dim3(last ? 1 : nb), dim3(256), 0, s, d_feat, (int)n, D, (seeded || it > 0) ? 1 : 0, start, use_sqrt, pin, nb,
                           last ? (Part*)nullptr : pout, Q.mind.as<double>(), d_out + it, d_n);
        (void)pin;
    }
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

/* gcn.create_adj (gcn.py:116-191): the adjacency the trained-GCN branch feeds its graph convolutions and the k-center step, torch float32 in the
 * reference.  d_feat [N,F] float32 = concatenate(unlabelled candidates, labelled regions) (gcn.py:199); the clouds' bbox centres and directed
 * chamfer means come from ssdr_cloud_graph_batch_dev (same d_coff / d_boff), d_rows [N] gives, cloud by cloud, the row of every member in the
 * result.  Outputs: d_out_v [N,F] (the normalised features the function returns) and d_out_adj [N,N] = (cos * exp(-(ED + CD)) - I) D^-1 + I. */
int ssdr_create_adj_dev(const float* d_feat, size_t N, int F, const double* d_centres, const double* d_cd_dir, const int32_t* d_coff, const int64_t* d_boff,
                        size_t num_clouds, size_t n_max, const int32_t* d_rows, float* d_out_v, float* d_out_adj, void* stream) {
    if (!d_feat || !d_centres || !d_cd_dir || !d_coff || !d_boff || !d_rows || !d_out_v || !d_out_adj || N == 0 || F < 1 || num_clouds == 0 || N > 0x7fff) { set_error("create_adj: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream); SelState& Q = sst(s);
    SSDR_TRY(Q.rowsum.reserve(8 * N));
    SSDR_HIP(hipMemsetAsync(d_out_adj, 0, 4 * N * N, s));
    hipLaunchKernelGGL(ca_normalize, dim3(grid_for((long)N * 64)), dim3(256), 0, s, d_feat, (int)N, F, d_out_v);
    hipLaunchKernelGGL(ca_block, dim3(grid_for((long)n_max * n_max, 256), 1, (unsigned)num_clouds), dim3(256), 0, s, d_out_v, F, d_centres, d_cd_dir, d_coff, (const long long*)d_boff, d_rows, (int)N, d_out_adj);
    hipLaunchKernelGGL(ca_colsum, dim3(grid_for((long)N)), dim3(256), 0, s, d_out_adj, (int)N, Q.rowsum.as<float>());
    hipLaunchKernelGGL(ca_scale, dim3(grid_for((long)N * N)), dim3(256), 0, s, d_out_adj, (int)N, Q.rowsum.as<float>());
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

/* GCN_FPS_sampling (sampler2.py:313-342, :736-781) behind the candidate rule of its caller (sampler2.py:533-552, :745-753), with no host decision in
 * between: ranking in, selected candidates out.  The candidate rule runs as four small kernels (cand_*); the row counts it finds stay on the device and
 * every kernel behind it (features, bbox centres, chamfer packer, chamfer, adjacency, keep-top mask, propagation hops, FPS) reads them there, its launch
 * shape chosen by the caller's capacities.  d_result: [0..7] counts (n_unl, n_lab, ntot, nmax, sampling_batch, status, block elements as int64),
 * [8 .. 8+max_select) the selected candidates (indices into the candidate list), [8+max_select .. +cap_rows) the candidate list followed by the labelled
 * regions (superpoint ids; the first n_unl are the candidates, cloud by cloud, descending uncertainty inside a cloud). */
int ssdr_gcn_fps_sampling_dev(const float* d_feat, int feat_dim, const int32_t* d_cls, const int32_t* d_dom, const int32_t* d_lab_cls, const int32_t* d_lab_dom,
                              const float* d_xyz, const int32_t* d_sp_off, const int32_t* d_sp_pts, const int32_t* d_order, size_t S, const uint8_t* d_labelled, const int32_t* d_sp_base, size_t num_clouds,
                              const int32_t* d_lab_off, const int32_t* d_lab_sp, size_t n_lab, size_t batch_size, int gcn_number, int gcn_top, int selector, int start,
                              size_t cap_rows, size_t cap_nmax, size_t cap_sq, size_t cap_unl, size_t max_select, int32_t* d_result, void* stream) {
    if (!d_feat || !d_cls || !d_dom || (!d_lab_cls != !d_lab_dom) || !d_xyz || !d_sp_off || !d_sp_pts || !d_order || !d_labelled || !d_sp_base || !d_lab_off || !d_result || feat_dim != 32 ||
        num_clouds == 0 || num_clouds > 65535 || S == 0 || S > 0x7ffffff0 || cap_rows == 0 || cap_nmax == 0 || cap_sq == 0 || cap_unl == 0 || cap_rows > (1u << 22) || gcn_number < 0 || start < 0 || selector < 0 || selector > 1 || (selector == 1 && n_lab == 0) || (n_lab && !d_lab_sp)) {
        set_error("gcn_fps_sampling: bad arguments (feat_dim == 32, at most 2^22 candidate + labelled rows, at most 65535 clouds, k-center needs labelled regions)"); return SSDR_ERR_INVALID;
    }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream); SelState& Q = sst(s);
    const int B = (int)num_clouds, nchunks = (int)((S + CR_NT - 1) / CR_NT), D = feat_dim;
    // ints: rankpos S, cploc S, stage S, chunk nchunks, ncand B, ntop B, uoff B+1, coff B+1, gsel cap, rows cap | int64: boff B+1
    const size_t ni = 3 * S + (size_t)nchunks + 4 * (size_t)B + 2 + 2 * cap_rows + n_lab + 16;
    SSDR_TRY(Q.cand_i.reserve(4 * ni + 8 * ((size_t)B + 2)));
    int* rankpos = Q.cand_i.as<int>(); int* cploc = rankpos + S; int* stage = cploc + S; int* chunk = stage + S; int* ncand = chunk + nchunks; int* ntop = ncand + B;
    int* uoff = ntop + B; int* coff = uoff + B + 1; int* gsel = coff + B + 1; int* rows = gsel + cap_rows; int* already = rows + cap_rows;
    long long* boff = reinterpret_cast<long long*>(Q.cand_i.as<char>() + ((4 * ni + 7) & ~(size_t)7));
    // doubles: V, comb, tmp0, tmp1 [cap_rows, D]; centres [cap_rows, 3]; dir, adj [cap_sq]
    SSDR_TRY(Q.cand_f.reserve(8 * (4 * cap_rows * D + 3 * cap_rows + 2 * cap_sq)));
    double* V = Q.cand_f.as<double>(); double* comb = V + cap_rows * D; double* tmp0 = comb + cap_rows * D; double* tmp1 = tmp0 + cap_rows * D;
    double* cen = tmp1 + cap_rows * D; double* dir = cen + 3 * cap_rows; double* adj = dir + cap_sq;
    Q.last_comb = comb; Q.last_cap = cap_rows;
    int* counts = d_result; int* out = d_result + 8; int* sel = out + max_select;
    std::optional<ProfScope> prof; prof.emplace("sel_candidate_rule", s, 0.0);
    hipLaunchKernelGGL(cand_rank, dim3(nchunks), dim3(CR_NT), 0, s, d_order, (int)S, d_labelled, rankpos, cploc, chunk);
    hipLaunchKernelGGL(cand_chunkscan, dim3(1), dim3(256), 0, s, chunk, nchunks);
    hipLaunchKernelGGL(cand_cloud, dim3(B, cand_slices((size_t)S, (size_t)B)), dim3(256), 0, s, rankpos, cploc, chunk, d_labelled, d_sp_base, (int)S, (int)std::min<size_t>(batch_size, 0x7fffffff), stage, ncand, ntop);
    hipLaunchKernelGGL(cand_layout, dim3(1), dim3(256), 0, s, ncand, ntop, d_lab_off, B, (long long)cap_rows, (long long)cap_sq, uoff, coff, boff, counts);
    hipLaunchKernelGGL(cand_fill, dim3(B), dim3(256), 0, s, stage, d_sp_base, ncand, uoff, coff, d_lab_off, d_lab_sp, counts, sel, gsel, rows, already);
    const int nt = (int)cap_rows, nm = (int)cap_nmax; const unsigned nc = (unsigned)B;
    prof.emplace("sel_features_pack", s, 0.0);
    // compute_features (sampler2.py:333,339) of the refs, widened; bbox centres of the grouped rows
    hipLaunchKernelGGL(sel_segment_mean, dim3(grid_for((long)nt * D)), dim3(256), 0, s, d_feat, D, d_cls, d_dom, d_sp_off, d_sp_pts, sel, nt, (float*)nullptr, counts + 2, V, comb,
                       d_lab_cls, d_lab_dom, counts);
    SSDR_TRY(Q.rowsum.reserve(8 * cap_rows));
    ChamferPack P; SSDR_TRY(chamfer_pack_buffers(Q, cap_rows, num_clouds, P));
    SSDR_TRY(chamfer_pack_launch(P, d_xyz, d_sp_off, d_sp_pts, gsel, coff, 0, cap_rows, nm, nc, cen, s));
    prof.emplace("sel_chamfer", s, 0.0);          // (pairs of points: the counts are the device's; bench.py derives the FLOPs from the result)
    SSDR_TRY(chamfer_dir_batch_launch(d_xyz, d_sp_off, d_sp_pts, gsel, coff, boff, nm, nc, cen, dir, P, s));
    prof.emplace("sel_adjacency_propagate", s, 0.0);
    hipLaunchKernelGGL(sel_adj_build_batch, dim3(std::min(nm, 1024), 1, nc), dim3(256), 0, s, cen, dir, coff, boff, adj, Q.rowsum.as<double>());
    hipLaunchKernelGGL(sel_adj_norm_batch, dim3(grid_for((long)nm * nm, 256), 1, nc), dim3(256), 0, s, Q.rowsum.as<double>(), coff, boff, adj);
    if (gcn_top > 0) hipLaunchKernelGGL(sel_adj_topk_batch, dim3(std::max(1, std::min((nm + 3) / 4, 1024)), 1, nc), dim3(256), 0, s, adj, coff, boff, gcn_top);
    const double* src = V;
    for (int hop = 0; hop < gcn_number; ++hop) {
        double* dst = (hop & 1) ? tmp1 : tmp0;
        hipLaunchKernelGGL(sel_propagate_batch, dim3(grid_for((long)nm * D, 256), 1, nc), dim3(256), 0, s, adj, coff, boff, rows, src, D, dst, comb);
        src = dst;
    }
    SSDR_HIP(hipGetLastError());
    prof.reset();
    if (max_select == 0) return SSDR_OK;
    // selector 1: kCenterGreedy over candidates + labelled rows, seeded with the labelled ones (kcenterGreedy.py:84-128; sampler2.py's "kcenter" branch)
    if (selector == 1) return fps_like(comb, cap_rows, D, already, n_lab, 0, max_select, 1, out, s, counts + 2);
    return fps_like(comb, cap_unl, D, nullptr, 0, start, max_select, 0, out, s, counts);
}

/* The propagated rows of the last ssdr_gcn_fps_sampling_dev call on `stream` (device pointer, [cap_rows][32] float64: the candidates first, then the labelled
 * regions — sum_i A^i V of fps_gcn_cpu.py:162-167, what its FPS / k-center ran over).  Valid until the next selection call on that stream. */
int ssdr_gcn_fps_sampling_rows(void* stream, const double** d_rows, size_t* cap_rows) {
    if (!d_rows) { set_error("gcn_fps_sampling_rows: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    SelState& Q = sst(pick_stream(stream));
    if (!Q.last_comb) { set_error("gcn_fps_sampling_rows: no ssdr_gcn_fps_sampling_dev call on this stream yet"); return SSDR_ERR_INVALID; }
    *d_rows = Q.last_comb; if (cap_rows) *cap_rows = Q.last_cap;
    return SSDR_OK;
}

/* What the enqueue-only selection calls on `stream` found and could not return: bit 0 = a cooperative FPS / k-center launch was not co-resident (a
 * workgroup waited for one that never arrived): its picks are invalid (-1 from the abort on).  Waits for the stream, clears the word. */
int ssdr_select_status(void* stream, int32_t* out_status) {
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream); SelState& Q = sst(s);
    int st = 0;
    if (Q.status_init) {
        SSDR_HIP(hipMemcpyAsync(&st, Q.status.p, 4, hipMemcpyDeviceToHost, s));
        SSDR_HIP(hipStreamSynchronize(s));
        if (st) SSDR_HIP(hipMemsetAsync(Q.status.p, 0, 4, s));
    } else SSDR_HIP(hipStreamSynchronize(s));
    if (out_status) *out_status = st;
    if (st & 1) { set_error("selection: a cooperative FPS / k-center launch was not co-resident (a workgroup never arrived); its picks are invalid"); return SSDR_ERR_INTERNAL; }
    return SSDR_OK;
}

/* The sharded run's selection without a host decision: two enqueue-only calls around the all-gather of the candidates' propagated features (exchange 3).
 * ssdr_gcn_fps_sharded_local_dev: the candidate rule over the GLOBAL ranking (d_gorder over Sg = world * Smax padded region ids, d_glabelled != 0 for
 * labelled regions and padding, global cloud c = rank * Bmax + b spans d_gbase[c] .. d_gbase[c+1]-1), then this rank's share of GCN_FPS_sampling (features,
 * chamfer graph, adjacency, propagation) for its own clouds.  d_comb_out [nu_max, 32]: its candidates' propagated features in candidate order (what the
 * all-gather sends); d_plan (int32, 16 + world + 2 * world * nu_max words): counts, candidates per rank, the rows of the gathered array in global candidate
 * order, the global candidate list.  ssdr_fps_gathered_dev: compacts the gathered array by the plan and runs the replicated global FPS from candidate `start`. */
int ssdr_gcn_fps_sharded_local_dev(const float* d_feat, int feat_dim, const int32_t* d_cls, const int32_t* d_dom, const int32_t* d_lab_cls, const int32_t* d_lab_dom,
                                   const float* d_xyz, const int32_t* d_sp_off, const int32_t* d_sp_pts, const int32_t* d_lab_off, const int32_t* d_lab_sp, size_t n_lab, size_t num_clouds,
                                   const int32_t* d_gorder, size_t Sg, const uint8_t* d_glabelled, const int32_t* d_gbase, int rank, int world, size_t Smax, size_t Bmax,
                                   size_t batch_size, int gcn_number, int gcn_top, size_t cap_rows, size_t cap_nmax, size_t cap_sq, size_t nu_max, size_t nl_max,
                                   double* d_comb_out, int32_t* d_plan, void* stream) {
    if (!d_feat || !d_cls || !d_dom || (!d_lab_cls != !d_lab_dom) || !d_xyz || !d_sp_off || !d_sp_pts || !d_lab_off || !d_gorder || !d_glabelled || !d_gbase || !d_comb_out || !d_plan || feat_dim != 32 ||
        world < 1 || world > 64 || rank < 0 || rank >= world || num_clouds == 0 || num_clouds > Bmax || Bmax * (size_t)world > 65535 || Sg != Smax * (size_t)world || Sg > 0x7ffffff0 ||
        cap_rows == 0 || cap_nmax == 0 || cap_sq == 0 || nu_max == 0 || gcn_number < 0 || (n_lab && !d_lab_sp) || (nl_max && n_lab > nl_max)) {
        set_error("gcn_fps_sharded_local: bad arguments (feat_dim == 32, world <= 64, Sg == world * Smax, n_lab <= nl_max)"); return SSDR_ERR_INVALID;
    }
    SSDR_TRY(ensure_init());
    hipStream_t s = pick_stream(stream); SelState& Q = sst(s);
    const int W = world, Bg = (int)(Bmax * (size_t)world), B = (int)num_clouds, nchunks = (int)((Sg + CR_NT - 1) / CR_NT), D = feat_dim;
    const size_t ni = 3 * Sg + (size_t)nchunks + 3 * (size_t)Bg + 2 * (size_t)B + 8 + 3 * cap_rows + 16;
    SSDR_TRY(Q.cand_i.reserve(4 * ni + 8 * ((size_t)B + 2)));
    int* rankpos = Q.cand_i.as<int>(); int* cploc = rankpos + Sg; int* stage = cploc + Sg; int* chunk = stage + Sg; int* ncand = chunk + nchunks; int* ntop = ncand + Bg;
    int* guoff = ntop + Bg; int* uoff = guoff + Bg + 1; int* coff = uoff + B + 1; int* gsel = coff + B + 1; int* rows = gsel + cap_rows; int* sel = rows + cap_rows;
    long long* boff = reinterpret_cast<long long*>(Q.cand_i.as<char>() + ((4 * ni + 7) & ~(size_t)7));
    SSDR_TRY(Q.cand_f.reserve(8 * (4 * cap_rows * D + 3 * cap_rows + 2 * cap_sq)));
    double* V = Q.cand_f.as<double>(); double* comb = V + cap_rows * D; double* tmp0 = comb + cap_rows * D; double* tmp1 = tmp0 + cap_rows * D;
    double* cen = tmp1 + cap_rows * D; double* dir = cen + 3 * cap_rows; double* adj = dir + cap_sq;
    int* plan = d_plan;
    std::optional<ProfScope> prof; prof.emplace("sel_candidate_rule", s, 0.0);
    hipLaunchKernelGGL(cand_rank, dim3(nchunks), dim3(CR_NT), 0, s, d_gorder, (int)Sg, d_glabelled, rankpos, cploc, chunk);
    hipLaunchKernelGGL(cand_chunkscan, dim3(1), dim3(256), 0, s, chunk, nchunks);
    hipLaunchKernelGGL(cand_cloud, dim3(Bg, cand_slices((size_t)Sg, (size_t)Bg)), dim3(256), 0, s, rankpos, cploc, chunk, d_glabelled, d_gbase, (int)Sg, (int)std::min<size_t>(batch_size, 0x7fffffff), stage, ncand, ntop);
    // this rank's clouds: the layout every kernel below reads (counts in plan[0..7]); then what the other ranks contribute
    hipLaunchKernelGGL(cand_layout, dim3(1), dim3(256), 0, s, ncand + (size_t)rank * Bmax, ntop + (size_t)rank * Bmax, d_lab_off, B, (long long)cap_rows, (long long)cap_sq, uoff, coff, boff, plan);
    hipLaunchKernelGGL(cand_global, dim3(1), dim3(256), 0, s, ncand, ntop, W, (int)Bmax, (int)nu_max, guoff, plan);
    hipLaunchKernelGGL(cand_fill_global, dim3(Bg), dim3(256), 0, s, stage, d_gbase, ncand, guoff, plan, plan + 16 + W + (size_t)W * nu_max);
    hipLaunchKernelGGL(cand_fill_local, dim3(B), dim3(256), 0, s, stage, d_gbase + (size_t)rank * Bmax, ncand + (size_t)rank * Bmax, uoff, coff, d_lab_off, d_lab_sp, plan,
                       (int)((size_t)rank * Smax), sel, gsel, rows);
    const int nt = (int)cap_rows, nm = (int)cap_nmax; const unsigned nc = (unsigned)B;
    prof.emplace("sel_features_pack", s, 0.0);
    hipLaunchKernelGGL(sel_segment_mean, dim3(grid_for((long)nt * D)), dim3(256), 0, s, d_feat, D, d_cls, d_dom, d_sp_off, d_sp_pts, sel, nt, (float*)nullptr, plan + 2, V, comb,
                       d_lab_cls, d_lab_dom, plan);
    SSDR_TRY(Q.rowsum.reserve(8 * cap_rows));
    ChamferPack P; SSDR_TRY(chamfer_pack_buffers(Q, cap_rows, num_clouds, P));
    SSDR_TRY(chamfer_pack_launch(P, d_xyz, d_sp_off, d_sp_pts, gsel, coff, 0, cap_rows, nm, nc, cen, s));
    prof.emplace("sel_chamfer", s, 0.0);          // (pairs of points: the counts are the device's; bench.py derives the FLOPs from the result)
    SSDR_TRY(chamfer_dir_batch_launch(d_xyz, d_sp_off, d_sp_pts, gsel, coff, boff, nm, nc, cen, dir, P, s));
    prof.emplace("sel_adjacency_propagate", s, 0.0);
    hipLaunchKernelGGL(sel_adj_build_batch, dim3(std::min(nm, 1024), 1, nc), dim3(256), 0, s, cen, dir, coff, boff, adj, Q.rowsum.as<double>());
    hipLaunchKernelGGL(sel_adj_norm_batch, dim3(grid_for((long)nm * nm, 256), 1, nc), dim3(256), 0, s, Q.rowsum.as<double>(), coff, boff, adj);
    if (gcn_top > 0) hipLaunchKernelGGL(sel_adj_topk_batch, dim3(std::max(1, std::min((nm + 3) / 4, 1024)), 1, nc), dim3(256), 0, s, adj, coff, boff, gcn_top);
    const double* src = V;
    for (int hop = 0; hop < gcn_number; ++hop) {
        double* dst = (hop & 1) ? tmp1 : tmp0;
        hipLaunchKernelGGL(sel_propagate_batch, dim3(grid_for((long)nm * D, 256), 1, nc), dim3(256), 0, s, adj, coff, boff, rows, src, D, dst, comb);
        src = dst;
    }
    prof.reset();
    // the candidates' rows (the first n_unl of comb) are what the exchange sends
    hipLaunchKernelGGL(copy_rows_dn, dim3(grid_for((long)nu_max * D)), dim3(256), 0, s, comb, d_comb_out, D, (int)nu_max, plan);
    // ... and, for the global k-center (nl_max > 0), this rank's labelled regions' rows behind them
    if (nl_max && n_lab) hipLaunchKernelGGL(copy_rows_from, dim3(grid_for((long)n_lab * D)), dim3(256), 0, s, comb, d_comb_out + nu_max * (size_t)D, D, (int)n_lab, plan);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

/* The replicated global k-center of the sharded run (BASELINE configuration 4; kcenterGreedy.py:84-128 over [candidates | labelled regions of ALL ranks], the labelled
 * ones already selected): d_gathered = the all-gather of every rank's nu_max + nl_max rows (ssdr_gcn_fps_sharded_local_dev with nl_max > 0), d_nlab_off[world + 1] =
 * prefix of the ranks' labelled counts (static: the host knows it), d_glob [cap_rows][32] and d_already [n_lab_total] scratch.  Nothing is read back. */
int ssdr_kcenter_gathered_dev(const double* d_gathered, const int32_t* d_plan, int world, size_t nu_max, size_t nl_max, const int32_t* d_nlab_off, size_t n_lab_total,
                              size_t cap_rows, size_t max_select, double* d_glob, int32_t* d_already, int32_t* d_out, void* stream) {
    if (!d_gathered || !d_plan || !d_glob || !d_out || !d_already || !d_nlab_off || world < 1 || world > 64 || nu_max == 0 || nl_max == 0 || n_lab_total == 0 || cap_rows <= n_lab_total ||
        n_lab_total > (size_t)world * nl_max) { set_error("kcenter_gathered: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (max_select == 0) return SSDR_OK;
    hipStream_t s = pick_stream(stream); SelState& Q = sst(s);
    SSDR_TRY(Q.hist.reserve(64));
    int* nrows = Q.hist.as<int>();
    hipLaunchKernelGGL(sel_gather_kc, dim3(grid_for((long)cap_rows * 64)), dim3(256), 0, s, (const uint32_t*)d_gathered, d_plan, world, (int)nu_max, (int)(nu_max + nl_max), d_nlab_off,
                       (int)cap_rows, 64, (uint32_t*)d_glob, d_already, nrows);
    SSDR_HIP(hipGetLastError());
    return fps_like(d_glob, cap_rows, 32, d_already, n_lab_total, 0, max_select, 1, d_out, s, nrows);
}

int ssdr_fps_gathered_dev(const double* d_gathered, const int32_t* d_plan, int world, size_t nu_max, size_t cap_rows, int repeat, int start, size_t max_select, double* d_glob,
                          int32_t* d_out, void* stream) {
    if (!d_gathered || !d_plan || !d_glob || !d_out || world < 1 || nu_max == 0 || cap_rows == 0 || repeat < 1 || start < 0) { set_error("fps_gathered: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (max_select == 0) return SSDR_OK;
    hipStream_t s = pick_stream(stream); SelState& Q = sst(s);
    SSDR_TRY(Q.hist.reserve(64));
    int* nrep = Q.hist.as<int>();
    hipLaunchKernelGGL(sel_gather_rows_rep, dim3(grid_for((long)cap_rows * 64)), dim3(256), 0, s, (const uint32_t*)d_gathered, d_plan + 16 + world, (int)cap_rows, 64, (uint32_t*)d_glob,
                       d_plan + 8, repeat, nrep);
    SSDR_HIP(hipGetLastError());
    return fps_like(d_glob, cap_rows, 32, nullptr, 0, start, max_select, 0, d_out, s, nrep);
}

int ssdr_fps_superpoint_dev(const double* d_centres, const double* d_cd_dir, size_t n, int start, size_t count, int32_t* d_out, void* stream) {
    if (!d_centres || !d_cd_dir || !d_out || start < 0 || (size_t)start >= n || count > n || n > 8192) { set_error("fps_superpoint: bad arguments (n <= 8192)"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (count == 0) return SSDR_OK;
    static bool attr_done = false;       // 8 n bytes of dynamic LDS next to the static arrays: beyond the 64 KiB default from n ~ 7800 on
    if (!attr_done) { SSDR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_superpoint), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192)); attr_done = true; }
    hipLaunchKernelGGL(fps_superpoint, dim3(1), dim3(256), 8 * n, pick_stream(stream), d_centres, d_cd_dir, (int)n, start, (int)count, d_out);
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

int ssdr_fps_dev(const double* d_feat, size_t n, int feat_dim, int start, size_t count, int32_t* d_out, void* stream) {
    if (!d_feat || !d_out || feat_dim < 1 || start < 0 || (size_t)start >= n || count > n) { set_error("fps: bad arguments"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (count == 0) return SSDR_OK;
    return fps_like(d_feat, n, feat_dim, nullptr, 0, start, count, 0, d_out, pick_stream(stream));
}

int ssdr_kcenter_dev(const double* d_feat, size_t n, int feat_dim, const int32_t* d_already_selected, size_t n_already, size_t count, int32_t* d_out, void* stream) {
    if (!d_feat || !d_out || feat_dim < 1 || !d_already_selected || n_already == 0) { set_error("kcenter: bad arguments (needs a non-empty already_selected)"); return SSDR_ERR_INVALID; }
    SSDR_TRY(ensure_init());
    if (count == 0) return SSDR_OK;
    return fps_like(d_feat, n, feat_dim, d_already_selected, n_already, 0, count, 1, d_out, pick_stream(stream));
}

}
