// Device-side helpers of the RandLA-Net kernels (gfx950): MFMA wrappers, bf16 splitting, cross-row lane exchanges.
//
// Split-bf16 arithmetic ("bf16x3"): an fp32 value x is carried as hi = bf16_rn(x), lo = bf16_rn(x - hi)
// (|x - hi - lo| <= 2^-16 |x|) and a product a*b is evaluated on the bf16 matrix cores as
// a_hi*b_hi + a_lo*b_hi + a_hi*b_lo with fp32 accumulation (v_mfma_f32_16x16x32_bf16, 16x the rate of the exact
// f32-input MFMA per instruction slot; three products => 5.3x).  Plain bf16 keeps only a_hi*b_hi.
#pragma once
#include "ssdr_internal.hpp"
#include "block_prims.hpp"

namespace ssdr {


#ifndef HIPEMU
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// 16x16x32 bf16: lane l holds A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15], j = 0..7 (two per dword, low half first);
// C/D col = l&15, row = 4(l>>4) + reg
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// (bf16_rn(b) << 16) | bf16_rn(a): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t)); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }    // v_rcp_f32, 1 ulp
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
// combine a value with the one held by lane ^ 16 / lane ^ 32 (op commutative): v_permlane16_swap / v_permlane32_swap hand every
// lane its own value and its partner's in the two results, no trip through the LDS crossbar
template <class Op> __device__ __forceinline__ float combine_xor16(float v, Op op) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
template <class Op> __device__ __forceinline__ float combine_xor32(float v, Op op) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
#define SSDR_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// 32x32x16 bf16: lane l (r = l&31, h = l>>5) holds A[row r][k = 8h+j], B[k = 8h+j][col r], j = 0..7; C/D col = l&31,
// row = (reg&3) + 8(reg>>2) + 4h: the 16 registers of a lane are 16 rows of ONE column (reductions over rows stay inside the lane)
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
#else
typedef hipemu_f32x4 f32x4;
typedef hipemu_u32x4 u32x4;
typedef hipemu_u32x2 u32x2;
static inline f32x4 mfma16(float a, float b, f32x4 c) { return hipemu_mfma_f32_16x16x4f32(a, b, c); }
static inline f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) { return hipemu_mfma_f32_16x16x32_bf16(a, b, c); }
typedef hipemu_f32x16 f32x16;
static inline f32x16 mfma32_bf16(u32x4 a, u32x4 b, f32x16 c) { return hipemu_mfma_f32_32x32x16_bf16(a, b, c); }
static inline unsigned pack_bf16(float a, float b) { return hipemu_bf16_rn(a) | (hipemu_bf16_rn(b) << 16); }
static inline float fast_rcp(float x) { return 1.0f / x; }
static inline float fast_exp2(float x) { return exp2f(x); }
template <class Op> static inline float combine_xor16(float v, Op op) { return op(v, __shfl_xor(v, 16)); }
template <class Op> static inline float combine_xor32(float v, Op op) { return op(v, __shfl_xor(v, 32)); }
#define SSDR_SCHED_FENCE() ((void)0)
#endif

// reductions over the four 16-lane rows of a wave (the 16 neighbours of a point live in 4 registers x 4 rows)
__device__ __forceinline__ float rows_max(float v) {
    auto mx = [](float a, float b) { return fmaxf(a, b); };
    return combine_xor32(combine_xor16(v, mx), mx);
}
__device__ __forceinline__ float rows_sum(float v) {
    auto ad = [](float a, float b) { return a + b; };
    return combine_xor32(combine_xor16(v, ad), ad);
}

// (fmaxf() on a value the compiler cannot prove canonical — every MFMA result — costs a second `v_max_f32 x, x, x` in front of it in IEEE mode; the attention
// kernels' translation unit is therefore compiled with -fno-honor-nans, csrc/Makefile.  Inline `v_max_f32` / `v_max3_f32` were tried first: identical tiles then
// gave different results in different batch slots — the hazard recogniser does not place the wait states a read of an MFMA result needs in front of inline
// assembly; `__builtin_amdgcn_fmed3f(a, b, inf)` is folded back into the canonicalising form.)
__device__ __forceinline__ float lrelu(float v) { return fmaxf(v, v * 0.2f); }     // = v > 0 ? v : 0.2 v, one instruction less

// x0, x1 -> packed hi pair and packed lo pair (lo = bf16_rn(x - hi))
__device__ __forceinline__ void split_bf16(float x0, float x1, unsigned& hi, unsigned& lo) {
    hi = pack_bf16(x0, x1);
    lo = pack_bf16(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ float bf16_lo_f32(unsigned pair) { return __uint_as_float(pair << 16); }
__device__ __forceinline__ float bf16_hi_f32(unsigned pair) { return __uint_as_float(pair & 0xffff0000u); }

}  // namespace ssdr
