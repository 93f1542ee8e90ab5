// mfma_valu_overlap.hip -- can a gfx950 SIMD run v_mfma_f32_32x32x16_bf16 and ordinary VALU instructions at the same time?
// The K-Means assign kernel (csrc/kmeans_cs.h) issues, per (point tile, centroid tile) pair, 9 dependent MFMAs into one
// accumulator set and then a 48-instruction min / med3 selection; its matrix pipe is ~40 % and its VALU ~46 % busy and every
// attempt to overlap the two (software pipelining in round 4 and 5, de-phasing the waves of a SIMD) left the time unchanged.
// This program measures the primitive: the same instruction counts, in one wave or in different waves of a SIMD, accumulators in
// VGPRs or in AGPRs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/mfma_valu_overlap && tools/ubench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VALU6(x, y, z)                                  \
    "v_med3_f32 %[" #x "], %[" #x "], %[" #y "], %[" #z "]\n" \
    "v_min_f32 %[" #y "], %[" #x "], %[" #y "]\n"          \
    "v_med3_f32 %[" #x "], %[" #x "], %[" #y "], %[" #z "]\n" \
    "v_min_f32 %[" #y "], %[" #x "], %[" #y "]\n"          \
    "v_med3_f32 %[" #x "], %[" #x "], %[" #y "], %[" #z "]\n" \
    "v_min_f32 %[" #y "], %[" #x "], %[" #y "]\n"

// MODE 0: 9 dependent MFMAs per iteration.  1: 54 VALU per iteration.  2: MFMA, 6 VALU, MFMA, ... in ONE wave, accumulator in VGPRs.
// 3: the same, accumulator in AGPRs.  4: even waves run mode 0, odd waves mode 1 (different waves of the same SIMD).
// 5: 9 MFMAs then 54 VALU, back to back in one wave (what the compiler emits for the kernel).
template <int MODE>
__global__ __launch_bounds__(1024) void overlap_kernel(int iters, float *out)
{
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x & 7)); b[i] = (__bf16)(0.002f * i); }
    float x = threadIdx.x * 1e-3f, y = 1.0f + x, z = 2.0f - x;
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || MODE == 2 || MODE == 3 || MODE == 5 || (MODE == 4 && (wave & 4) == 0);   // waves w, w+4, w+8, w+12 share a SIMD
    const bool do_valu = MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5 || (MODE == 4 && (wave & 4) != 0);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z) "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z)
                "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z) "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z)
                "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z) "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z)
                "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z) "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z)
                "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z)
                : [c] "+v"(acc), [x] "+v"(x), [y] "+v"(y) : [a] "v"(a), [b] "v"(b), [z] "v"(z));
        } else if (MODE == 3) {
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z) "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z)
                "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z) "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z)
                "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z) "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z)
                "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z) "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z)
                "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" VALU6(x, y, z)
                : [c] "+a"(acc), [x] "+v"(x), [y] "+v"(y) : [a] "v"(a), [b] "v"(b), [z] "v"(z));
        } else {
            if (do_mfma)
                asm volatile(
                    "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n"
                    "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n"
                    "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n" "v_mfma_f32_32x32x16_bf16 %[c], %[a], %[b], %[c]\n"
                    : [c] "+v"(acc) : [a] "v"(a), [b] "v"(b));
            if (do_valu)
                asm volatile(VALU6(x, y, z) VALU6(x, y, z) VALU6(x, y, z) VALU6(x, y, z) VALU6(x, y, z) VALU6(x, y, z) VALU6(x, y, z) VALU6(x, y, z) VALU6(x, y, z)
                             : [x] "+v"(x), [y] "+v"(y) : [z] "v"(z));
        }
    }
    float s = x + y;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static float run(int iters, float *out)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(overlap_kernel<MODE>, dim3(256), dim3(1024), 0, 0, 16, out);   // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(overlap_kernel<MODE>, dim3(256), dim3(1024), 0, 0, iters, out);   // 16 waves per CU = 4 per SIMD
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    float *out;
    CK(hipMalloc(&out, 64));
    const int iters = 20000;
    const double per = 1e-3 * 2.4e9 / iters;   // ms -> cycles per iteration at 2.4 GHz nominal
    const float t0 = run<0>(iters, out), t1 = run<1>(iters, out), t5 = run<5>(iters, out), t2 = run<2>(iters, out), t3 = run<3>(iters, out),
                t4 = run<4>(iters, out);
    printf("# 256 workgroups x 16 waves (4 per SIMD), %d iterations; one iteration = 9 dependent v_mfma_f32_32x32x16_bf16 and / or 54 v_med3/v_min_f32\n", iters);
    printf("# cycles = per iteration and wave-slot at 2.4 GHz nominal (4 waves share a SIMD: a SIMD sees 4x the work)\n");
    printf("mfma only (9 per wave-iteration)                           %8.3f ms  %7.1f cycles\n", t0, t0 * per);
    printf("valu only (54 per wave-iteration)                          %8.3f ms  %7.1f cycles\n", t1, t1 * per);
    printf("one wave: 9 mfma THEN 54 valu (the compiler's order)       %8.3f ms  %7.1f cycles   (sum of the two above: %.3f)\n", t5, t5 * per, t0 + t1);
    printf("one wave: mfma, 6 valu, mfma, ... accumulator in VGPRs     %8.3f ms  %7.1f cycles\n", t2, t2 * per);
    printf("one wave: mfma, 6 valu, mfma, ... accumulator in AGPRs     %8.3f ms  %7.1f cycles\n", t3, t3 * per);
    printf("two waves of a SIMD mfma only, the other two valu only     %8.3f ms  %7.1f cycles   (half the work of each kind per SIMD: sum %.3f, max %.3f)\n",
           t4, t4 * per, 0.5f * (t0 + t1), 0.5f * (t0 > t1 ? t0 : t1));
    return 0;
}
