// VALU issue-rate microbenchmark: cycles per wave-instruction per SIMD for a few opcodes at 1, 2, 4, 8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int OP> __global__ void k(float* out, int iters, long long* cyc) {
    extern __shared__ float pad_lds[];      // 100 KB of dynamic LDS: at most one workgroup per CU
    if (iters < 0) pad_lds[threadIdx.x] = 0.f;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float c = 1.0001f, d = 0.5f;
    double q0 = a0, q1 = a1, q2 = a2, q3 = a3; const double qc = 1.0001, qd = 0.5;
    d4 m0 = {q0, q1, q2, q3}, m1 = m0, m2 = m0, m3 = m0;
    f16v g0 = {}, g1 = {};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { // v_fma_f32 x8 independent
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if (OP == 1) { // v_exp_f32 x8
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 2) { // v_pk_fma_f32 x4 (8 floats)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(f2{c, c}), "v"(f2{d, d}));
        } else if (OP == 3) { // v_cvt_pk_bf16_f32 x8
            asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                         "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 4) { // v_pk_mul_f32 x4
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(f2{c, c}));
        } else if (OP == 5) { // v_max3_f32 x8
            asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                         "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if (OP == 6) { // v_pk_add_f32 x4
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(f2{c, c}));
        } else if (OP == 7) { // v_fma_f64 x4
            asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qc), "v"(qd));
        } else if (OP == 8) { // v_add_f64 x4
            asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qc));
        } else if (OP == 9) { // v_mul_f64 x4
            asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qc));
        } else if (OP == 10) { // v_min_f64 x4
            asm volatile("v_min_f64 %0, %0, %4\n v_min_f64 %1, %1, %4\n v_min_f64 %2, %2, %4\n v_min_f64 %3, %3, %4\n"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qc));
        } else if (OP == 11) { // v_mfma_f64_16x16x4_f64 x4 independent accumulators
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(q0, qc, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(q1, qc, m1, 0, 0, 0);
            m2 = __builtin_amdgcn_mfma_f64_16x16x4f64(q2, qc, m2, 0, 0, 0); m3 = __builtin_amdgcn_mfma_f64_16x16x4f64(q3, qc, m3, 0, 0, 0);
        } else if (OP == 12) { // one MFMA f64 + 16 v_min_f64 on other registers (do they overlap inside one wave?)
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(q0, qc, m0, 0, 0, 0);
            asm volatile("v_min_f64 %0, %0, %4\n v_min_f64 %1, %1, %4\n v_min_f64 %2, %2, %4\n v_min_f64 %3, %3, %4\n"
                         "v_min_f64 %0, %0, %4\n v_min_f64 %1, %1, %4\n v_min_f64 %2, %2, %4\n v_min_f64 %3, %3, %4\n"
                         "v_min_f64 %0, %0, %4\n v_min_f64 %1, %1, %4\n v_min_f64 %2, %2, %4\n v_min_f64 %3, %3, %4\n"
                         "v_min_f64 %0, %0, %4\n v_min_f64 %1, %1, %4\n v_min_f64 %2, %2, %4\n v_min_f64 %3, %3, %4\n"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qc));
        } else if (OP == 13) { // v_mfma_f32_32x32x2_f32 x2 independent accumulators
            g0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c, g0, 0, 0, 0); g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c, g1, 0, 0, 0);
        } else if (OP == 14) { // one f32-input MFMA + 16 v_max3_f32 on other registers: does the f32 matrix instruction share the vector pipe?
            g0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c, g0, 0, 0, 0);
            asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                         "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                         "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                         "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                         : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0.x) : "v"(c), "v"(d));
        } else if (OP == 15) { // the same beside the bf16 matrix instruction (32 cycles): control
            bf8 hb = {}; 
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hb, hb, g0, 0, 0, 0);
            asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                         "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                         "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                         "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                         : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0.x) : "v"(c), "v"(d));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(q0 + q1 + q2 + q3) + (float)(m0[0] + m1[1] + m2[2] + m3[3]) + g0[0] + g1[5];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int OP> void run(const char* name, int per_iter) {
    float* out; long long* cyc; hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8 * 16 * 1024); hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const int iters = 20000;
    for (int wps : {1, 2, 3, 4}) {       // waves per SIMD: workgroups of wps * 256 threads, one per CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipDeviceSynchronize(); hipEventRecord(e0, 0);
        hipMemset(cyc, 0, 8 * 16 * 1024);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(wps * 256), 100 * 1024, 0, out, iters, cyc);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> cs(16 * 1024); hipMemcpy(cs.data(), cyc, 8 * 16 * 1024, hipMemcpyDeviceToHost);
        long long c = 0; for (auto v : cs) c = v > c ? v : c;
        printf("   [wall %.1f us, ticks %lld => %.0f ticks/us] ", ms * 1e3, c, c / (ms * 1e3));
        // s_memtime counts at 100 MHz? report per-wave ticks per instruction and the SIMD-level instruction rate
        printf("%-18s waves/SIMD %d: %.2f ticks per wave-instruction (per wave); per SIMD %.2f ticks per instruction\n", name, wps, (double)c / ((double)iters * per_iter), (double)c / ((double)iters * per_iter) / wps);
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("v_fma_f32", 8); run<1>("v_exp_f32", 8); run<2>("v_pk_fma_f32", 4); run<3>("v_cvt_pk_bf16_f32", 8); run<4>("v_pk_mul_f32", 4); run<5>("v_max3_f32", 8); run<6>("v_pk_add_f32", 4);
    run<7>("v_fma_f64", 4); run<8>("v_add_f64", 4); run<9>("v_mul_f64", 4); run<10>("v_min_f64", 4); run<11>("v_mfma_f64_16x16x4", 4); run<12>("mfma_f64 + 16 min_f64", 17);
    run<13>("v_mfma_f32_32x32x2", 2); run<14>("mfma_f32x2 + 16 max3", 17); run<15>("mfma_bf16 + 16 max3", 17);
    return 0;
}
