// Layout check for v_mfma_f32_32x32x16_bf16 on gfx950 and for v_permlane32_swap, with asymmetric data.
// hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_layout.hip -o /tmp/mfma_layout && /tmp/mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned short f2bf(float f) { return (unsigned short)(__float_as_uint(f) >> 16); }  // exact for the test values

__global__ void k(const float *A /*32x16*/, const float *B /*16x32*/, float *C /*32x32*/, unsigned *swp)
{
    const int l = threadIdx.x;
    union { bf16x8 v; unsigned short s[8]; } a, b;
    for (int e = 0; e < 8; ++e) {
        const int kk = 8 * (l >> 5) + e;       // hypothesis: lane holds k = 8*(lane>>5) + e
        a.s[e] = f2bf(A[(l & 31) * 16 + kk]);  // A[row = lane&31][k]
        b.s[e] = f2bf(B[kk * 32 + (l & 31)]);  // B[k][col = lane&31]
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        C[row * 32 + col] = c[r];
    }
    unsigned x = 1000 + l, y = 2000 + l;
    auto r2 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    swp[l] = r2[0];
    swp[64 + l] = r2[1];
}

int main()
{
    std::vector<float> A(32 * 16), B(16 * 32), C(32 * 32), R(32 * 32, 0.f);
    for (int i = 0; i < 32; ++i) for (int k2 = 0; k2 < 16; ++k2) A[i * 16 + k2] = (float)((i * 3 + k2 * 5) % 17 - 8);
    for (int k2 = 0; k2 < 16; ++k2) for (int j = 0; j < 32; ++j) B[k2 * 32 + j] = (float)((k2 * 7 + j * 2) % 13 - 6) * 0.5f;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k2 = 0; k2 < 16; ++k2) R[i * 32 + j] += A[i * 16 + k2] * B[k2 * 32 + j];
    float *dA, *dB, *dC; unsigned *dS;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dS, 128 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dS);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    unsigned S[128]; hipMemcpy(S, dS, sizeof(S), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += C[i] != R[i];
    printf("mfma_f32_32x32x16_bf16 layout mismatches: %d of 1024\n", bad);
    printf("permlane32_swap(x=1000+l, y=2000+l): r0[0]=%u r0[31]=%u r0[32]=%u r0[63]=%u | r1[0]=%u r1[31]=%u r1[32]=%u r1[63]=%u\n",
           S[0], S[31], S[32], S[63], S[64], S[95], S[96], S[127]);
    return bad != 0;
}
