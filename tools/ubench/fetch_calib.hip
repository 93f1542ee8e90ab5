// fetch_calib.hip -- known byte counts in the access patterns of the SOR kernels, for calibrating rocprofv3's FETCH_SIZE /
// WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, "HBM": "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming
// read ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE -d out -o pmc -- tools/ubench/fetch_calib      (WRITE_SIZE: a second pass)
// Every kernel runs ONCE over buffers far larger than L2 + Infinity Cache (1 GiB) unless the pattern itself is about reuse;
// the program prints the bytes each kernel reads / writes algorithmically.  Test infrastructure, not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// A: 16 B per lane, coalesced, streaming (what the guide calibrated)
__global__ void calib_read16_stream(const float4 *__restrict__ a, size_t n, float *__restrict__ out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = a[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
// B: 4 B per lane at a 12-byte stride, three columns (bbox / bucket_hist / bucket_scatter reading the (N,3) rows)
__global__ void calib_read4_rows3(const float *__restrict__ a, size_t n, float *__restrict__ out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += a[3 * i] + a[3 * i + 1] + a[3 * i + 2];
    if (acc == 12345.678f) out[0] = acc;
}
// C: knn_brick phase 1: a half-wave reads 32 CONSECUTIVE float4 (512 B), the wave two such runs at unrelated places
__global__ void calib_read16_runs32(const float4 *__restrict__ a, size_t n, const unsigned *__restrict__ starts, size_t nruns, float *__restrict__ out)
{
    float acc = 0.f;
    const size_t w = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 5, nw = ((size_t)gridDim.x * blockDim.x) >> 5;
    for (size_t r = w; r < nruns; r += nw) {
        const float4 v = a[(size_t)starts[r] + (threadIdx.x & 31)];
        acc += v.x + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
// D: knn_brick phase 2: every lane gathers ONE float4 at an unrelated index (a 160 MB array: 10M points -- inside the
//    Infinity Cache, as in the real kernel)
__global__ void calib_gather16(const float4 *__restrict__ a, const unsigned *__restrict__ idx, size_t n, float *__restrict__ out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = a[idx[i]];
        acc += v.x + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
// E: 16 B per lane coalesced streaming WRITE
__global__ void calib_write16_stream(float4 *__restrict__ a, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
// F: knn_brick's result: ONE 4-byte store per lane at an unrelated index (mean_out in original order), 40 MB array
__global__ void calib_scatter4(float *__restrict__ a, const unsigned *__restrict__ idx, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a[idx[i]] = (float)i;
}
// G: bucket_scatter's result: runs of 2-3 consecutive 16-byte records at unrelated places
__global__ void calib_scatter16_runs(float4 *__restrict__ a, const unsigned *__restrict__ starts, size_t nruns, int runlen)
{
    for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < nruns * (size_t)runlen; r += (size_t)gridDim.x * blockDim.x) {
        const size_t run = r / runlen;
        a[(size_t)starts[run] + (r - run * runlen)] = make_float4((float)r, 1.f, 2.f, 3.f);
    }
}

static unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }

int main()
{
    const size_t GiB = 1ull << 30;
    float4 *big;            // 1 GiB
    CK(hipMalloc(&big, GiB));
    CK(hipMemset(big, 0, GiB));
    float *out;
    CK(hipMalloc(&out, 64));
    const size_t n16 = GiB / 16, n10m = 10'000'000;
    // permutation of 0..10M-1 (every slot exactly once: the scatter writes every element of the 40 MB array once)
    unsigned *h = (unsigned *)malloc(sizeof(unsigned) * n10m);
    for (size_t i = 0; i < n10m; ++i) h[i] = (unsigned)i;
    unsigned s = 12345u;
    for (size_t i = n10m - 1; i > 0; --i) { const size_t j = lcg(s) % (i + 1); const unsigned t = h[i]; h[i] = h[j]; h[j] = t; }
    unsigned *perm;
    CK(hipMalloc(&perm, sizeof(unsigned) * n10m));
    CK(hipMemcpy(perm, h, sizeof(unsigned) * n10m, hipMemcpyHostToDevice));
    // run starts: multiples of 32 records in a 160 MB array (C) / of 4 records (G), shuffled
    const size_t nruns32 = n10m / 32;
    unsigned *h2 = (unsigned *)malloc(sizeof(unsigned) * n10m);
    for (size_t i = 0; i < nruns32; ++i) h2[i] = (unsigned)(32 * (h[i] % nruns32));
    unsigned *runs32;
    CK(hipMalloc(&runs32, sizeof(unsigned) * nruns32));
    CK(hipMemcpy(runs32, h2, sizeof(unsigned) * nruns32, hipMemcpyHostToDevice));
    const int runlen = 3;
    const size_t nruns3 = n10m / 4;            // slots of 4 records, 3 written
    size_t k = 0;
    for (size_t i = 0; i < n10m && k < nruns3; ++i) if (h[i] < nruns3) h2[k++] = 4u * h[i];
    unsigned *runs3;
    CK(hipMalloc(&runs3, sizeof(unsigned) * nruns3));
    CK(hipMemcpy(runs3, h2, sizeof(unsigned) * nruns3, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    const dim3 g(256 * 8), b(256);
    hipLaunchKernelGGL(calib_read16_stream, g, b, 0, 0, big, n16, out);
    hipLaunchKernelGGL(calib_read4_rows3, g, b, 0, 0, (const float *)big, GiB / 12, out);
    hipLaunchKernelGGL(calib_read16_runs32, g, b, 0, 0, big, n10m, runs32, nruns32, out);
    hipLaunchKernelGGL(calib_gather16, g, b, 0, 0, big, perm, n10m, out);
    hipLaunchKernelGGL(calib_write16_stream, g, b, 0, 0, big, n16);
    hipLaunchKernelGGL(calib_scatter4, g, b, 0, 0, (float *)big, perm, n10m);
    hipLaunchKernelGGL(calib_scatter16_runs, g, b, 0, 0, big, runs3, nruns3, runlen);
    CK(hipDeviceSynchronize());
    printf("# algorithmic bytes per kernel (one launch each)\n");
    printf("calib_read16_stream   read  %zu\n", GiB);
    printf("calib_read4_rows3     read  %zu\n", (GiB / 12) * 12);
    printf("calib_read16_runs32   read  %zu (+ %zu of run starts)\n", nruns32 * 32 * 16, nruns32 * 4);
    printf("calib_gather16        read  %zu (+ %zu of indices); distinct 128-B lines touched x 128 = %zu\n", n10m * 16, n10m * 4, (size_t)(n10m * 16));
    printf("calib_write16_stream  write %zu\n", GiB);
    printf("calib_scatter4        write %zu (every element of a 40 MB array once, in random order; + %zu of indices read)\n", n10m * 4, n10m * 4);
    printf("calib_scatter16_runs  write %zu (runs of %d records = %d B at random 64-B slots)\n", nruns3 * runlen * 16, runlen, runlen * 16);
    return 0;
}
