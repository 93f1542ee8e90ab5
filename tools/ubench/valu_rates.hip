// Micro-benchmark: issue rate of the VALU ops the KNN kernels are made of (gfx950).
// Each wave runs 8 independent dependency chains of one op; 20 waves per CU resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

#define KERNEL(NAME, TYPE, INIT, BODY)                                          \
    __global__ __launch_bounds__(256) void NAME(TYPE *out, TYPE seed)            \
    {                                                                            \
        TYPE a0 = INIT + seed, a1 = a0 + (TYPE)1, a2 = a0 + (TYPE)2, a3 = a0 + (TYPE)3, \
             a4 = a0 + (TYPE)4, a5 = a0 + (TYPE)5, a6 = a0 + (TYPE)6, a7 = a0 + (TYPE)7; \
        TYPE b = seed * (TYPE)0.5 + (TYPE)threadIdx.x;                           \
        for (int i = 0; i < ITERS; ++i) {                                        \
            BODY(a0) BODY(a1) BODY(a2) BODY(a3) BODY(a4) BODY(a5) BODY(a6) BODY(a7) \
        }                                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
    }

#define B_FMA32(x) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(b));
#define B_ADD32(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_MIN32(x) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_MED3(x)  asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define B_ADD64(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_MUL64(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_FMA64(x) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x) : "v"(b));
#define B_MIN64(x) asm volatile("v_min_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_MAX64(x) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_CVT64(x) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(x) : "v"(bf));
#define B_CMPADDC(x) asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x) : "v"(bf), "v"(cf) : "vcc");
#define B_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(bu) : );
#define B_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(bu));

KERNEL(k_fma32, float, 1.0f, B_FMA32)
KERNEL(k_add32, float, 1.0f, B_ADD32)
KERNEL(k_min32, float, 1.0f, B_MIN32)
KERNEL(k_med3, float, 1.0f, B_MED3)
KERNEL(k_add64, double, 1.0, B_ADD64)
KERNEL(k_mul64, double, 1.0, B_MUL64)
KERNEL(k_fma64, double, 1.0, B_FMA64)
KERNEL(k_min64, double, 1.0, B_MIN64)
KERNEL(k_max64, double, 1.0, B_MAX64)

typedef float float2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pkfma32(float2v *out, float seed)
{
    float2v a0 = {1.0f + seed, 2.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    float2v b = {seed * 0.5f + threadIdx.x, seed};
#define B_PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(b));
    for (int i = 0; i < ITERS; ++i) { B_PKFMA(a0) B_PKFMA(a1) B_PKFMA(a2) B_PKFMA(a3) B_PKFMA(a4) B_PKFMA(a5) B_PKFMA(a6) B_PKFMA(a7) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ __launch_bounds__(256) void k_pkadd32(float2v *out, float seed)
{
    float2v a0 = {1.0f + seed, 2.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    float2v b = {seed * 0.5f + threadIdx.x, seed};
#define B_PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
    for (int i = 0; i < ITERS; ++i) { B_PKADD(a0) B_PKADD(a1) B_PKADD(a2) B_PKADD(a3) B_PKADD(a4) B_PKADD(a5) B_PKADD(a6) B_PKADD(a7) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ __launch_bounds__(256) void k_cmpaddc(unsigned *out, float seed)
{
    unsigned a0 = 1, a1 = 2, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8;
    float bf = seed + threadIdx.x, cf = seed * 2.0f;
    for (int i = 0; i < ITERS; ++i) { B_CMPADDC(a0) B_CMPADDC(a1) B_CMPADDC(a2) B_CMPADDC(a3) B_CMPADDC(a4) B_CMPADDC(a5) B_CMPADDC(a6) B_CMPADDC(a7) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ __launch_bounds__(256) void k_cvt64(double *out, float seed)
{
    double a0 = 1, a1 = 2, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8;
    float bf = seed + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) { B_CVT64(a0) B_CVT64(a1) B_CVT64(a2) B_CVT64(a3) B_CVT64(a4) B_CVT64(a5) B_CVT64(a6) B_CVT64(a7) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ __launch_bounds__(256) void k_mov(unsigned *out, float seed)
{
    unsigned a0 = 1, a1 = 2, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8;
    unsigned bu = (unsigned)seed + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) { B_MOV(a0) B_MOV(a1) B_MOV(a2) B_MOV(a3) B_MOV(a4) B_MOV(a5) B_MOV(a6) B_MOV(a7) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <class F>
static int run(const char *name, F launch, int ops_per_instr)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    const double blocks = 256.0 * 5, waves = blocks * 4;
    const double instr = waves * (double)ITERS * 8 * ops_per_instr;  // wave-instructions
    const double simd_cycles = 1024.0 * (ms * 1e-3) * 2.4e9;         // at the 2.4 GHz nominal clock
    printf("%-12s %8.3f ms  %.2f cycles per wave-instruction per SIMD (2.4 GHz nominal)  %.1f G wave-instr/s\n", name, ms,
           simd_cycles / instr, instr / (ms * 1e-3) / 1e9);
    return 0;
}

int main()
{
    void *buf;
    CHECK(hipMalloc(&buf, 256 * 5 * 256 * 16));
    dim3 g(256 * 5), b(256);
#define RUN(K, T, OPS) run(#K, [&] { hipLaunchKernelGGL(K, g, b, 0, 0, (T *)buf, (decltype(+T()))1.5f); }, OPS)
    run("v_fma_f32", [&] { hipLaunchKernelGGL(k_fma32, g, b, 0, 0, (float *)buf, 1.5f); }, 1);
    run("v_add_f32", [&] { hipLaunchKernelGGL(k_add32, g, b, 0, 0, (float *)buf, 1.5f); }, 1);
    run("v_min_f32", [&] { hipLaunchKernelGGL(k_min32, g, b, 0, 0, (float *)buf, 1.5f); }, 1);
    run("v_med3_f32", [&] { hipLaunchKernelGGL(k_med3, g, b, 0, 0, (float *)buf, 1.5f); }, 1);
    run("v_pk_fma_f32", [&] { hipLaunchKernelGGL(k_pkfma32, g, b, 0, 0, (float2v *)buf, 1.5f); }, 1);
    run("v_pk_add_f32", [&] { hipLaunchKernelGGL(k_pkadd32, g, b, 0, 0, (float2v *)buf, 1.5f); }, 1);
    run("cmp+addc", [&] { hipLaunchKernelGGL(k_cmpaddc, g, b, 0, 0, (unsigned *)buf, 1.5f); }, 2);
    run("v_mov_b32", [&] { hipLaunchKernelGGL(k_mov, g, b, 0, 0, (unsigned *)buf, 1.5f); }, 1);
    run("v_add_f64", [&] { hipLaunchKernelGGL(k_add64, g, b, 0, 0, (double *)buf, 1.5); }, 1);
    run("v_mul_f64", [&] { hipLaunchKernelGGL(k_mul64, g, b, 0, 0, (double *)buf, 1.5); }, 1);
    run("v_fma_f64", [&] { hipLaunchKernelGGL(k_fma64, g, b, 0, 0, (double *)buf, 1.5); }, 1);
    run("v_min_f64", [&] { hipLaunchKernelGGL(k_min64, g, b, 0, 0, (double *)buf, 1.5); }, 1);
    run("v_max_f64", [&] { hipLaunchKernelGGL(k_max64, g, b, 0, 0, (double *)buf, 1.5); }, 1);
    run("v_cvt_f64_f32", [&] { hipLaunchKernelGGL(k_cvt64, g, b, 0, 0, (double *)buf, 1.5f); }, 1);
    return 0;
}
