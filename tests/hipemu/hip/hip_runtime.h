// TEST INFRASTRUCTURE ONLY — a minimal CPU stand-in for <hip/hip_runtime.h>.
//
// The container the kernels are written in has no GPU, so the HIP sources under ssdr-al_amd/csrc are
// additionally compiled with g++ against this header into tests/hipemu/libssdr_al_emu.so and the
// kernel *logic* (indices, barriers, wave ballots / shuffles, atomics) is exercised by the
// `-m "not gpu"` tests.  It is never shipped, never loaded by the product loader, and proves nothing
// about performance or about the memory model; the `-m gpu` tests run the real gfx950 build.
//
// Execution model: each workgroup runs as `blockDim` ucontext fibers on one OS thread (workgroups are
// spread over OpenMP threads).  A fiber runs until it reaches __syncthreads() or a wave-level
// operation (__ballot, __shfl*); a wave operation completes once every unfinished lane of that
// 64-lane wave is blocked, with the lanes blocked at the operation as its active set; a barrier
// completes once every unfinished fiber of the workgroup waits at it.
#pragma once
#include <ucontext.h>
#include <time.h>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define HIPEMU 1

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint3e { unsigned x, y, z; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }
static inline int2 make_int2(int a, int b) { return int2{a, b}; }

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipDeviceAttributeMultiprocessorCount = 1 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

namespace hipemu {
enum { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
// Context switches: glibc's swapcontext saves and restores the signal mask — two system calls per switch, and a work-item yields at every wave operation and
// barrier: ~45 % of the CPU suite's time was spent in the kernel.  On x86-64 (outside the sanitizer build, whose runtime follows swapcontext only) a switch is
// the six callee-saved registers and the stack pointer.
#if defined(__x86_64__) && !defined(__SANITIZE_ADDRESS__) && !defined(HIPEMU_UCONTEXT)
#define HIPEMU_FAST_SWITCH 1
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(".text\n.weak hipemu_switch\n.type hipemu_switch,@function\nhipemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  subq $8, %rsp\n  stmxcsr (%rsp)\n  fnstcw 4(%rsp)\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  ldmxcsr (%rsp)\n  fldcw 4(%rsp)\n  addq $8, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size hipemu_switch,.-hipemu_switch\n");
#endif
struct Fiber {
#ifdef HIPEMU_FAST_SWITCH
    void* sp = nullptr;
#else
    ucontext_t ctx;
#endif
    char* stack = nullptr; int state = READY; uint3e tid; int lane, wave, op;
};
struct WaveX { uint64_t val[64], snap[64]; uint64_t snap_mask; };
struct BlockState {
    std::vector<Fiber> fibers; std::vector<WaveX> waves;
#ifdef HIPEMU_FAST_SWITCH
    void* sched = nullptr;
#else
    ucontext_t sched;
#endif
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
};
inline thread_local BlockState* g_bs = nullptr;
inline thread_local char* g_dynshared = nullptr;
constexpr size_t STACK = 256 * 1024;

#ifdef HIPEMU_FAST_SWITCH
inline void yield_to_sched() { BlockState* b = g_bs; hipemu_switch(&b->cur->sp, b->sched); }
inline void fiber_entry() { BlockState* b = g_bs; (*b->body)(); b->cur->state = DONE; hipemu_switch(&b->cur->sp, b->sched); __builtin_trap(); }
// a new fiber's stack as hipemu_switch expects to find it: control words, six zeroed registers, the entry point as the return address (the stack pointer
// is 8 modulo 16 when fiber_entry starts, as behind a call)
inline void* fiber_boot(char* stack, size_t size) {
    uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
    void** sp = (void**)(top - 8);                     // [top - 8]: a slot the entry point may take for its return address' place (never returns)
    *--sp = (void*)fiber_entry;                         // ret -> fiber_entry, rsp = top - 8 afterwards
    for (int i = 0; i < 6; ++i) *--sp = nullptr;        // rbp, rbx, r12-r15
    unsigned csr = 0, cw = 0;
    asm volatile("stmxcsr %0" : "=m"(csr)); asm volatile("fnstcw %0" : "=m"(*(unsigned short*)&cw));
    --sp; ((unsigned*)sp)[0] = csr; ((unsigned*)sp)[1] = cw;
    return sp;
}
#else
inline void yield_to_sched() { BlockState* b = g_bs; swapcontext(&b->cur->ctx, &b->sched); }
inline void fiber_entry() { BlockState* b = g_bs; (*b->body)(); b->cur->state = DONE; swapcontext(&b->cur->ctx, &b->sched); }
#endif
}  // namespace hipemu

inline thread_local uint3e threadIdx, blockIdx;
inline thread_local dim3 blockDim, gridDim;

namespace hipemu {
inline void run_block(BlockState& bs, unsigned nthreads, const std::function<void()>& body) {
    g_bs = &bs; bs.body = &body;
    if (bs.fibers.size() < nthreads) bs.fibers.resize(nthreads);
    bs.waves.resize((nthreads + 63) / 64);
    for (unsigned t = 0; t < nthreads; ++t) {
        Fiber& f = bs.fibers[t];
        if (!f.stack) f.stack = (char*)malloc(STACK);
        f.state = READY; f.lane = t & 63; f.wave = t >> 6;
        f.tid.x = t % blockDim.x; f.tid.y = (t / blockDim.x) % blockDim.y; f.tid.z = t / (blockDim.x * blockDim.y);
#ifdef HIPEMU_FAST_SWITCH
        f.sp = fiber_boot(f.stack, STACK);
#else
        getcontext(&f.ctx); f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = &bs.sched;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
#endif
    }
    for (;;) {
        bool any = false; unsigned done = 0;
        for (unsigned t = 0; t < nthreads; ++t) {
            Fiber& f = bs.fibers[t];
#ifdef HIPEMU_FAST_SWITCH
            if (f.state == READY) { bs.cur = &f; threadIdx = f.tid; hipemu_switch(&bs.sched, f.sp); any = true; }
#else
            if (f.state == READY) { bs.cur = &f; threadIdx = f.tid; swapcontext(&bs.sched, &f.ctx); any = true; }
#endif
            if (f.state == DONE) ++done;
        }
        if (done == nthreads) break;
        bool released = false;
        for (size_t w = 0; w < bs.waves.size(); ++w) {
            uint64_t mask = 0; int op = -1;
            unsigned lo = (unsigned)w * 64, hi = std::min(nthreads, lo + 64);
            for (unsigned t = lo; t < hi; ++t)
                if (bs.fibers[t].state == WAIT_WAVE) {
                    mask |= 1ull << (t - lo);
                    if (op >= 0 && op != bs.fibers[t].op) { fprintf(stderr, "hipemu: lanes of one wave wait at different wave ops\n"); abort(); }
                    op = bs.fibers[t].op;
                }
            if (!mask) continue;
            memcpy(bs.waves[w].snap, bs.waves[w].val, sizeof(bs.waves[w].val)); bs.waves[w].snap_mask = mask;
            for (unsigned t = lo; t < hi; ++t) if (bs.fibers[t].state == WAIT_WAVE) bs.fibers[t].state = READY;
            released = true;
        }
        if (released) continue;
        unsigned waiting = 0;
        for (unsigned t = 0; t < nthreads; ++t) if (bs.fibers[t].state == WAIT_BLOCK) ++waiting;
        if (waiting + done == nthreads && waiting) { for (unsigned t = 0; t < nthreads; ++t) if (bs.fibers[t].state == WAIT_BLOCK) bs.fibers[t].state = READY; continue; }
        if (!any) { fprintf(stderr, "hipemu: deadlock\n"); abort(); }
    }
}

inline void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    const long nblocks = (long)grid.x * grid.y * grid.z; const unsigned nthreads = block.x * block.y * block.z;
#pragma omp parallel
    {
        // the fibers' stacks stay with the OpenMP thread from launch to launch (a launch used to malloc and free nthreads x 256 KB per thread: a third of the
        // CPU suite's time was the kernel mapping and unmapping them)
        static thread_local BlockState bs; std::vector<char> dyn(shmem + 64);
        g_dynshared = dyn.data(); blockDim = block; gridDim = grid;
#pragma omp for schedule(dynamic, 1)
        for (long b = 0; b < nblocks; ++b) {
            blockIdx.x = (unsigned)(b % grid.x); blockIdx.y = (unsigned)((b / grid.x) % grid.y); blockIdx.z = (unsigned)(b / ((long)grid.x * grid.y));
            run_block(bs, nthreads, body);
        }
    }
}

inline uint64_t wave_op(uint64_t v, int op) {
    BlockState* b = g_bs; Fiber* f = b->cur;
    b->waves[f->wave].val[f->lane] = v; f->op = op; f->state = WAIT_WAVE; yield_to_sched();
    return 0;
}
template <class T> inline uint64_t to_bits(T v) { uint64_t u = 0; memcpy(&u, &v, sizeof(T)); return u; }
template <class T> inline T from_bits(uint64_t u) { T v; memcpy(&v, &u, sizeof(T)); return v; }
}  // namespace hipemu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })      /* launches complete before they return here: by reference (the states hold move-only buffers) */
#define SSDR_DYN_SHARED(type, name) type* name = reinterpret_cast<type*>(hipemu::g_dynshared)

static inline void __syncthreads() { hipemu::g_bs->cur->state = hipemu::WAIT_BLOCK; hipemu::yield_to_sched(); }
static inline unsigned long long __ballot(int p) {
    hipemu::wave_op(p ? 1 : 0, 1);
    auto& w = hipemu::g_bs->waves[hipemu::g_bs->cur->wave]; unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (((w.snap_mask >> l) & 1) && w.snap[l]) m |= 1ull << l;
    return m;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    hipemu::wave_op(hipemu::to_bits(v), 2);
    auto* f = hipemu::g_bs->cur; auto& w = hipemu::g_bs->waves[f->wave];
    int l = (f->lane & ~(width - 1)) + (src & (width - 1));
    return ((w.snap_mask >> l) & 1) ? hipemu::from_bits<T>(w.snap[l]) : v;
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64) { return __shfl(v, (hipemu::g_bs->cur->lane ^ m) & (width - 1), width); }
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = (hipemu::g_bs->cur->lane & (width - 1)) + (int)d; int self = hipemu::g_bs->cur->lane & (width - 1);
    return __shfl(v, l < width ? l : self, width);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int self = hipemu::g_bs->cur->lane & (width - 1); int l = self - (int)d;
    return __shfl(v, l >= 0 ? l : self, width);
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline double __longlong_as_double(long long i) { double f; memcpy(&f, &i, 8); return f; }
static inline long long __double_as_longlong(double f) { long long i; memcpy(&i, &f, 8); return i; }
#define __expf(x) expf(x)
static inline float __fdividef(float a, float b) { return a / b; }
using std::max; using std::min;

// workgroups run on several OS threads, so these must be real atomics
template <class T, class F> static inline T hipemu_rmw(T* p, F f) {
    T o; __atomic_load(p, &o, __ATOMIC_RELAXED);
    for (;;) { T n = f(o); if (__atomic_compare_exchange(p, &o, &n, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) return o; }
}
template <class T> static inline T atomicAdd(T* p, T v) { return hipemu_rmw(p, [v](T o) { return (T)(o + v); }); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicMax(T* p, T v) { return hipemu_rmw(p, [v](T o) { return v > o ? v : o; }); }
template <class T> static inline T atomicMin(T* p, T v) { return hipemu_rmw(p, [v](T o) { return v < o ? v : o; }); }
template <class T> static inline T atomicOr(T* p, T v) { return hipemu_rmw(p, [v](T o) { return (T)(o | v); }); }
template <class T> static inline T atomicAnd(T* p, T v) { return hipemu_rmw(p, [v](T o) { return (T)(o & v); }); }
template <class T> static inline T atomicExch(T* p, T v) { return hipemu_rmw(p, [v](T) { return v; }); }
template <class T> static inline T atomicCAS(T* p, T c, T v) { T o = c; __atomic_compare_exchange(p, &o, &v, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED); return o; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// f32-in MFMA 16x16x4 (v_mfma_f32_16x16x4_f32): lane l holds A[l&15][l>>4], B[l>>4][l&15]; C/D col = l&15,
// rows (l>>4)*4 + r.  Result is a k-ordered fmaf chain (cdna_hip_programming.md section 3).
typedef float hipemu_f32x4 __attribute__((vector_size(16)));
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c) {
    uint64_t packed = (uint64_t)__float_as_uint(a) | ((uint64_t)__float_as_uint(b) << 32);
    hipemu::wave_op(packed, 3);
    auto* f = hipemu::g_bs->cur; auto& w = hipemu::g_bs->waves[f->wave];
    const int col = f->lane & 15, rg = f->lane >> 4;
    hipemu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = rg * 4 + r; float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            const float av = __uint_as_float((unsigned)(w.snap[row + 16 * k] & 0xffffffffu));
            const float bv = __uint_as_float((unsigned)(w.snap[col + 16 * k] >> 32));
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}

// bf16 helpers and the 16x16x32 bf16 MFMA (v_mfma_f32_16x16x32_bf16): lane l holds A[row l&15][k = 8(l>>4)+j] and
// B[k = 8(l>>4)+j][col l&15], j = 0..7, two bf16 per dword (low half first); C/D as the f32 form.  Products of bf16 values are exact
// in fp32; the hardware's internal summation order is not specified, so the stand-in sums in double and rounds once.
typedef unsigned hipemu_u32x4 __attribute__((vector_size(16)));
typedef unsigned hipemu_u32x2 __attribute__((vector_size(8)));
static inline unsigned hipemu_bf16_rn(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7f800000u) == 0x7f800000u && (u & 0x7fffffu)) return (u >> 16) | 0x40u;      // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x32_bf16(hipemu_u32x4 a, hipemu_u32x4 b, hipemu_f32x4 c) {
    auto* f = hipemu::g_bs->cur; auto& w = hipemu::g_bs->waves[f->wave];
    const int col = f->lane & 15, rg = f->lane >> 4;
    unsigned A[4][4][4], Bv[4][4];          // A[r][g][dword], B[g][dword]
    for (int half = 0; half < 4; ++half) {  // a.xy, a.zw, b.xy, b.zw as four 64-bit exchanges
        const hipemu_u32x4& v = half < 2 ? a : b; const int d0 = (half & 1) * 2;
        hipemu::wave_op((uint64_t)v[d0] | ((uint64_t)v[d0 + 1] << 32), 4 + half);
        for (int g = 0; g < 4; ++g) {
            if (half < 2) for (int r = 0; r < 4; ++r) { const uint64_t s = w.snap[rg * 4 + r + 16 * g]; A[r][g][d0] = (unsigned)s; A[r][g][d0 + 1] = (unsigned)(s >> 32); }
            else { const uint64_t s = w.snap[col + 16 * g]; Bv[g][d0] = (unsigned)s; Bv[g][d0 + 1] = (unsigned)(s >> 32); }
        }
    }
    hipemu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        double acc = c[r];
        for (int g = 0; g < 4; ++g)
            for (int j = 0; j < 8; ++j) {
                const unsigned ad = A[r][g][j >> 1], bd = Bv[g][j >> 1];
                const float av = __uint_as_float((j & 1) ? (ad & 0xffff0000u) : (ad << 16));
                const float bv = __uint_as_float((j & 1) ? (bd & 0xffff0000u) : (bd << 16));
                acc += (double)av * (double)bv;
            }
        d[r] = (float)acc;
    }
    return d;
}

// 32x32x16 bf16 MFMA (v_mfma_f32_32x32x16_bf16): lane l (r = l&31, h = l>>5) holds A[row r][k = 8h+j] and B[k = 8h+j][col r],
// j = 0..7; C/D col = l&31, row = (reg&3) + 8(reg>>2) + 4(l>>5), reg = 0..15 (cdna_hip_programming.md section 3).
struct hipemu_f32x16 { float v[16]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x16_bf16(hipemu_u32x4 a, hipemu_u32x4 b, hipemu_f32x16 c) {
    auto* f = hipemu::g_bs->cur; auto& w = hipemu::g_bs->waves[f->wave];
    const int col = f->lane & 31, hh = f->lane >> 5;
    unsigned Al[16][2][4], Bv[2][4];          // A[reg's row][k half][dword], B[k half][dword]
    for (int half = 0; half < 4; ++half) {
        const hipemu_u32x4& v = half < 2 ? a : b; const int d0 = (half & 1) * 2;
        hipemu::wave_op((uint64_t)v[d0] | ((uint64_t)v[d0 + 1] << 32), 8 + half);
        for (int g = 0; g < 2; ++g) {
            if (half < 2) for (int q = 0; q < 16; ++q) { const uint64_t s = w.snap[(q & 3) + 8 * (q >> 2) + 4 * hh + 32 * g]; Al[q][g][d0] = (unsigned)s; Al[q][g][d0 + 1] = (unsigned)(s >> 32); }
            else { const uint64_t s = w.snap[col + 32 * g]; Bv[g][d0] = (unsigned)s; Bv[g][d0 + 1] = (unsigned)(s >> 32); }
        }
    }
    hipemu_f32x16 d = c;
    for (int q = 0; q < 16; ++q) {
        double acc = c[q];
        for (int g = 0; g < 2; ++g)
            for (int j = 0; j < 8; ++j) {
                const unsigned ad = Al[q][g][j >> 1], bd = Bv[g][j >> 1];
                const float av = __uint_as_float((j & 1) ? (ad & 0xffff0000u) : (ad << 16));
                const float bv = __uint_as_float((j & 1) ? (bd & 0xffff0000u) : (bd << 16));
                acc += (double)av * (double)bv;
            }
        d[q] = (float)acc;
    }
    return d;
}

// half precision as bit patterns (g++ 11 has no _Float16 in C++): round to nearest even, subnormals kept, overflow to infinity
static inline unsigned short hipemu_f32_to_f16(float f) {
    const unsigned u = __float_as_uint(f), sign = (u >> 16) & 0x8000u, mag = u & 0x7fffffffu;
    if (mag >= 0x7f800000u) return (unsigned short)(sign | 0x7c00u | (mag > 0x7f800000u ? 0x200u : 0u));      // inf / NaN
    if (mag >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);                                           // >= 65520: rounds to infinity
    if (mag < 0x38800000u) {                                                                                   // below 2^-14: subnormal (or zero)
        if (mag < 0x33000000u) return (unsigned short)sign;                                                    // below 2^-25: zero
        const int e = (int)(mag >> 23); const unsigned m = (mag & 0x7fffffu) | 0x800000u; const int sh = 126 - e;      // value = m * 2^(e-150); unit 2^-24 -> shift by 126 - e in [14, 24]
        const unsigned q = m >> sh, rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
        return (unsigned short)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
    }
    const unsigned v = mag - 0x38000000u;                                                                      // rebias the exponent by 112
    return (unsigned short)(sign | ((v + 0xfffu + ((v >> 13) & 1u)) >> 13));
}
static inline float hipemu_f16_to_f32(unsigned short h) {
    const unsigned sign = (unsigned)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    if (e == 31u) return __uint_as_float(sign | 0x7f800000u | (m << 13));
    if (e == 0u) { const float v = (float)m * 5.9604644775390625e-8f; return sign ? -v : v; }                    // m * 2^-24
    return __uint_as_float(sign | ((e + 112u) << 23) | (m << 13));
}
// 32x32x16 f16 MFMA (v_mfma_f32_32x32x16_f16): operand and result lay-out of the bf16 form; products of half-precision values are exact in
// fp32, the hardware's internal summation order is not specified: the stand-in sums in double and rounds once.
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x16_f16(hipemu_u32x4 a, hipemu_u32x4 b, hipemu_f32x16 c) {
    auto* f = hipemu::g_bs->cur; auto& w = hipemu::g_bs->waves[f->wave];
    const int col = f->lane & 31, hh = f->lane >> 5;
    unsigned Al[16][2][4], Bv[2][4];
    for (int half = 0; half < 4; ++half) {
        const hipemu_u32x4& v = half < 2 ? a : b; const int d0 = (half & 1) * 2;
        hipemu::wave_op((uint64_t)v[d0] | ((uint64_t)v[d0 + 1] << 32), 13 + half);
        for (int g = 0; g < 2; ++g) {
            if (half < 2) for (int q = 0; q < 16; ++q) { const uint64_t s = w.snap[(q & 3) + 8 * (q >> 2) + 4 * hh + 32 * g]; Al[q][g][d0] = (unsigned)s; Al[q][g][d0 + 1] = (unsigned)(s >> 32); }
            else { const uint64_t s = w.snap[col + 32 * g]; Bv[g][d0] = (unsigned)s; Bv[g][d0 + 1] = (unsigned)(s >> 32); }
        }
    }
    hipemu_f32x16 d = c;
    for (int q = 0; q < 16; ++q) {
        double acc = c[q];
        for (int g = 0; g < 2; ++g)
            for (int j = 0; j < 8; ++j) {
                const unsigned ad = Al[q][g][j >> 1], bd = Bv[g][j >> 1];
                const float av = hipemu_f16_to_f32((unsigned short)((j & 1) ? (ad >> 16) : (ad & 0xffffu)));
                const float bv = hipemu_f16_to_f32((unsigned short)((j & 1) ? (bd >> 16) : (bd & 0xffffu)));
                acc += (double)av * (double)bv;
            }
        d[q] = (float)acc;
    }
    return d;
}

// f32-in MFMA 32x32x2 (v_mfma_f32_32x32x2_f32): lane l (r = l&31, h = l>>5) holds A[row r][k = h] and B[k = h][col r]; C/D as the bf16 32x32 form.
// An exact fmaf chain in k order (MI355X_MICROARCH.md, "FP32-input MFMA").
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c) {
    uint64_t packed = (uint64_t)__float_as_uint(a) | ((uint64_t)__float_as_uint(b) << 32);
    hipemu::wave_op(packed, 12);
    auto* f = hipemu::g_bs->cur; auto& w = hipemu::g_bs->waves[f->wave];
    const int col = f->lane & 31, hh = f->lane >> 5;
    hipemu_f32x16 d = c;
    for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh; float acc = c[q];
        for (int k = 0; k < 2; ++k) {
            const float av = __uint_as_float((unsigned)(w.snap[row + 32 * k] & 0xffffffffu));
            const float bv = __uint_as_float((unsigned)(w.snap[col + 32 * k] >> 32));
            acc = fmaf(av, bv, acc);
        }
        d[q] = acc;
    }
    return d;
}

// ---- runtime API (host memory stands in for device memory) --------------------------------------
static inline const char* hipGetErrorString(hipError_t) { return "hipemu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 8; return hipSuccess; }
// every stream gets a handle of its own (the library keeps its scratch per stream); all of them execute synchronously
static inline hipError_t hipStreamCreate(hipStream_t* s) { static uintptr_t next = 0; *s = (void*)__atomic_add_fetch(&next, 16, __ATOMIC_SEQ_CST); return hipSuccess; }
#define hipStreamNonBlocking 1
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = malloc(sizeof(double)); return hipSuccess; }
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline double hipemu_now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { *(double*)e = hipemu_now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }      // kernels run synchronously here: everything recorded has finished
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(*(double*)b - *(double*)a); return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipMalloc((void**)p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < h; ++r) memmove((char*)d + r * dp, (const char*)s + r * sp, w);
    return hipSuccess;
}
