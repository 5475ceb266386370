// Row order of the reference's grid_subsampling output, reproduced on the device.
//
// grid_subsampling.cpp:85 emits voxels by iterating an std::unordered_map<size_t, SampledData>.  With GCC's
// libstdc++ (identity hash, max load factor 1) that order is a deterministic function of the sequence of
// first-seen voxel keys:
//   * bucket counts grow 13, 29, 59, 127, ... (table below); a rehash happens before the insert that would
//     make size exceed the bucket count;
//   * an insert into an empty bucket goes to the head of the global node list, an insert into a non-empty
//     bucket goes to the head of that bucket's run;
//   * a rehash re-inserts the nodes in list order under the same two rules.
// Closed form used here, one "phase" per bucket count B (a = nodes present before the phase, k = nodes
// inserted during it, p = position in the list before the phase, t = insertion rank):
//   rehash        list <- runs by bucket, runs ordered by descending first position, members by descending p
//   inserts       new-bucket runs first (latest created first), then the old runs; inside a run the new
//                 nodes (latest first) precede the old ones
// which is one stable sort by (major, minor) with
//   major = new bucket ? k-1-(t_first-a) : k + (a-1-p_first)      minor = new node ? k-1-(t-a) : k + (a-1-p)
// so every phase is: two atomicMin passes for p_first / t_first per bucket, one key pass, one radix sort.
// The oracle (oracle/oracle_subsample.c) walks the actual linked lists; tests compare both with the real
// reference build row by row.
#include "ssdr_internal.hpp"
#include "block_prims.hpp"
#include <map>

namespace ssdr {
namespace {

constexpr int NSCHED = 28;
__device__ const unsigned long long BKT_SCHED_D[NSCHED] = {
    13ull, 29ull, 59ull, 127ull, 257ull, 541ull, 1109ull, 2357ull, 5087ull, 10273ull, 20753ull, 42043ull, 85229ull, 172933ull,
    351061ull, 712697ull, 1447153ull, 2938679ull, 5967347ull, 12117689ull, 24607243ull, 49969847ull, 101473717ull,
    206062531ull, 418451333ull, 849749479ull, 1725587117ull, 3504151727ull};
const unsigned long long BKT_SCHED_H[NSCHED] = {
    13ull, 29ull, 59ull, 127ull, 257ull, 541ull, 1109ull, 2357ull, 5087ull, 10273ull, 20753ull, 42043ull, 85229ull, 172933ull,
    351061ull, 712697ull, 1447153ull, 2938679ull, 5967347ull, 12117689ull, 24607243ull, 49969847ull, 101473717ull,
    206062531ull, 418451333ull, 849749479ull, 1725587117ull, 3504151727ull};

struct OrdPtrs {
    const uint64_t* ks; const uint32_t* vs; const int* seg_start; const int* d_m;
    uint64_t* skey; uint32_t* sval;     // sort buffers
    uint64_t* kt;                       // voxel key by insertion rank t
    uint32_t* seq;                      // voxel (key-order index) by insertion rank t
    int* L;                             // current list: insertion ranks in iteration order
    int* pfirst; int* tfirst;           // per bucket
    int* cnt;                           // [NSCHED] live element count of each phase's sort (0 = phase not needed)
    int* row_of_voxel;
    uint32_t* major;                    // per insertion rank: run-order key of the current phase
};

__device__ __forceinline__ void phase_bounds(int r, int m, int& a, int& k, unsigned long long& B) {
    B = BKT_SCHED_D[r];
    const long long prev = r ? (long long)BKT_SCHED_D[r - 1] : 0;
    a = (int)min((long long)m, prev);
    k = (int)(min((long long)m, (long long)B) - a);
}

__global__ __launch_bounds__(BS) void ord_first_seen_keys(OrdPtrs o) {
    const int m = *o.d_m;
    for (int v = blockIdx.x * BS + threadIdx.x; v < m; v += gridDim.x * BS) { o.skey[v] = o.vs[o.seg_start[v]]; o.sval[v] = (uint32_t)v; }
    if (blockIdx.x == 0 && threadIdx.x < NSCHED) {
        int a, k; unsigned long long B; phase_bounds(threadIdx.x, m, a, k, B);
        o.cnt[threadIdx.x] = k > 0 ? a + k : 0;
    }
}

__global__ __launch_bounds__(BS) void ord_gather_keys(OrdPtrs o) {
    const int m = *o.d_m;
    for (int t = blockIdx.x * BS + threadIdx.x; t < m; t += gridDim.x * BS) { const uint32_t v = o.sval[t]; o.seq[t] = v; o.kt[t] = o.ks[o.seg_start[v]]; }
}

__global__ __launch_bounds__(BS) void ord_clear(OrdPtrs o, int r) {
    const int m = *o.d_m; int a, k; unsigned long long B; phase_bounds(r, m, a, k, B);
    if (k <= 0) return;
    for (unsigned long long b = blockIdx.x * (unsigned long long)BS + threadIdx.x; b < B; b += (unsigned long long)gridDim.x * BS) { o.pfirst[b] = 0x7fffffff; o.tfirst[b] = 0x7fffffff; }
}

__global__ __launch_bounds__(BS) void ord_firsts(OrdPtrs o, int r) {
    const int m = *o.d_m; int a, k; unsigned long long B; phase_bounds(r, m, a, k, B);
    if (k <= 0) return;
    for (int i = blockIdx.x * BS + threadIdx.x; i < a + k; i += gridDim.x * BS) {
        if (i < a) { const int t = o.L[i]; atomicMin(&o.pfirst[o.kt[t] % B], i); }
        else atomicMin(&o.tfirst[o.kt[i] % B], i);
    }
}

__global__ __launch_bounds__(BS) void ord_keys(OrdPtrs o, int r) {
    const int m = *o.d_m; int a, k; unsigned long long B; phase_bounds(r, m, a, k, B);
    if (k <= 0) return;
    for (int i = blockIdx.x * BS + threadIdx.x; i < a + k; i += gridDim.x * BS) {
        const int t = i < a ? o.L[i] : i;
        const unsigned long long b = o.kt[t] % B;
        const int pf = o.pfirst[b];
        const unsigned long long major = pf != 0x7fffffff ? (unsigned long long)k + (unsigned long long)(a - 1 - pf)
                                                          : (unsigned long long)(k - 1 - (o.tfirst[b] - a));
        const unsigned long long minor = i < a ? (unsigned long long)k + (unsigned long long)(a - 1 - i) : (unsigned long long)(k - 1 - (i - a));
        // two stable 32-bit sorts (minor first, then major) instead of one 64-bit sort: all passes stay wide
        o.skey[i] = minor; o.sval[i] = (uint32_t)t; o.major[t] = (uint32_t)major;
    }
}

__global__ __launch_bounds__(BS) void ord_major_keys(OrdPtrs o, int r) {
    const int m = *o.d_m; int a, k; unsigned long long B; phase_bounds(r, m, a, k, B);
    if (k <= 0) return;
    for (int i = blockIdx.x * BS + threadIdx.x; i < a + k; i += gridDim.x * BS) o.skey[i] = o.major[o.sval[i]];
}

__global__ __launch_bounds__(BS) void ord_store_list(OrdPtrs o, int r) {
    const int m = *o.d_m; int a, k; unsigned long long B; phase_bounds(r, m, a, k, B);
    if (k <= 0) return;
    const bool last = (a + k == m);
    for (int i = blockIdx.x * BS + threadIdx.x; i < a + k; i += gridDim.x * BS) {
        const int t = (int)o.sval[i];
        o.L[i] = t;
        if (last) o.row_of_voxel[o.seq[t]] = i;
    }
}

struct OrdState { RadixSorter sorter; DevBuf skey, sval, kt, seq, L, pfirst, tfirst, cnt, major; };
OrdState& ost(hipStream_t st) { return per_stream<OrdState>(st); }

}  // namespace

int subsample_order_reference(const uint64_t* d_ks, const uint32_t* d_vs, const int* d_seg_start, const int* d_m, int n_host,
                              int* d_row_of_voxel, hipStream_t s) {
    OrdState& S = ost(s);
    const size_t n = (size_t)n_host;
    int nph = 0;   // phases that can be needed for up to n_host voxels
    while (nph < NSCHED && (nph == 0 || BKT_SCHED_H[nph - 1] < (unsigned long long)n_host)) ++nph;
    if (nph == NSCHED && BKT_SCHED_H[NSCHED - 1] < (unsigned long long)n_host) { set_error("too many voxels for the order emulation"); return SSDR_ERR_UNSUPPORTED; }
    const size_t bmax = (size_t)BKT_SCHED_H[nph - 1];
    SSDR_TRY(S.skey.reserve(8 * n + 16)); SSDR_TRY(S.sval.reserve(4 * n + 16)); SSDR_TRY(S.kt.reserve(8 * n + 16));
    SSDR_TRY(S.seq.reserve(4 * n + 16)); SSDR_TRY(S.L.reserve(4 * n + 16)); SSDR_TRY(S.major.reserve(4 * n + 16));
    SSDR_TRY(S.pfirst.reserve(4 * bmax + 16)); SSDR_TRY(S.tfirst.reserve(4 * bmax + 16)); SSDR_TRY(S.cnt.reserve(4 * NSCHED));
    OrdPtrs o{d_ks, d_vs, d_seg_start, d_m, S.skey.as<uint64_t>(), S.sval.as<uint32_t>(), S.kt.as<uint64_t>(), S.seq.as<uint32_t>(),
              S.L.as<int>(), S.pfirst.as<int>(), S.tfirst.as<int>(), S.cnt.as<int>(), d_row_of_voxel, S.major.as<uint32_t>()};
    SSDR_TRY(S.sorter.reserve(n));
    const int g = std::max(1, std::min((n_host + BS - 1) / BS, ctx().num_cu * 8));
    hipLaunchKernelGGL(ord_first_seen_keys, dim3(g), dim3(BS), 0, s, o);
    SSDR_TRY(S.sorter.sort(o.skey, o.sval, n_host, d_m, s, 32));          // first-seen point indices
    hipLaunchKernelGGL(ord_gather_keys, dim3(g), dim3(BS), 0, s, o);
    for (int r = 0; r < nph; ++r) {
        const long long cap = (long long)std::min<unsigned long long>(BKT_SCHED_H[r], (unsigned long long)n_host);
        const int gr = std::max(1, std::min((int)((cap + BS - 1) / BS), ctx().num_cu * 8));
        const int gb = (int)std::max<long long>(1, std::min<long long>((long long)((BKT_SCHED_H[r] + BS - 1) / BS), (long long)ctx().num_cu * 8));
        hipLaunchKernelGGL(ord_clear, dim3(gb), dim3(BS), 0, s, o, r);
        hipLaunchKernelGGL(ord_firsts, dim3(gr), dim3(BS), 0, s, o, r);
        hipLaunchKernelGGL(ord_keys, dim3(gr), dim3(BS), 0, s, o, r);
        int bits = 1; while ((1ll << bits) <= 2 * cap) ++bits;           // major, minor < a + k <= cap
        SSDR_TRY(S.sorter.sort(o.skey, o.sval, (int)cap, o.cnt + r, s, bits));
        hipLaunchKernelGGL(ord_major_keys, dim3(gr), dim3(BS), 0, s, o, r);
        SSDR_TRY(S.sorter.sort(o.skey, o.sval, (int)cap, o.cnt + r, s, bits));
        hipLaunchKernelGGL(ord_store_list, dim3(gr), dim3(BS), 0, s, o, r);
    }
    SSDR_HIP(hipGetLastError());
    return SSDR_OK;
}

}  // namespace ssdr
