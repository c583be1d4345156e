// Known-bytes calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of libplfx
// (the sweep's PMC traffic came out BELOW its algorithmic bytes with the guide's "double FETCH_SIZE" rule,
// which is calibrated for 16 B/lane loads only).  Every kernel moves an exactly known number of bytes through arrays that
// are far larger than the 256 MiB Infinity Cache:
//   read8    33 SoA arrays of doubles, lane e reads a[k][e] (8 B/lane, the sweep's state loads)        33 * 8 * N bytes read
//   read16   the same bytes as 16 B/lane loads (double2)                                                33 * 8 * N bytes read
//   write8   13 SoA arrays written 8 B/lane (res_sig, res_depl, fyn)                                    13 * 8 * N bytes written
//   write16  the same bytes as 16 B/lane stores
//   first_touch: the same write kernel on freshly allocated memory, timed on its first and second launch (k_init_tangent)
// build: hipcc --offload-arch=gfx950 -O2 -o pmc_calib pmc_calib.hip ; run plain (timings) or under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... / rocprofv3 --pmc WRITE_SIZE ...  (tools/probes/pmc_calib.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int NR = 33, NW = 13;

__global__ void __launch_bounds__(256) read8(const double *__restrict__ a, size_t n, double *__restrict__ out)
{
    for (size_t e = blockIdx.x * 256ul + threadIdx.x; e < n; e += gridDim.x * 256ul) {
        double s = 0.;
#pragma unroll
        for (int k = 0; k < NR; k++) s += a[k * n + e];
        if (s == 1.2345e300) out[e] = s;  // never true: the loads cannot be dropped, nothing is written
    }
}

__global__ void __launch_bounds__(256) read16(const double2 *__restrict__ a, size_t n2, double *__restrict__ out)
{
    for (size_t e = blockIdx.x * 256ul + threadIdx.x; e < n2; e += gridDim.x * 256ul) {
        double s = 0.;
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const double2 v = a[k * n2 + e];
            s += v.x + v.y;
        }
        if (s == 1.2345e300) out[e] = s;
    }
}

__global__ void __launch_bounds__(256) write8(double *__restrict__ a, size_t n, double v)
{
    for (size_t e = blockIdx.x * 256ul + threadIdx.x; e < n; e += gridDim.x * 256ul) {
#pragma unroll
        for (int k = 0; k < NW; k++) a[k * n + e] = v + k;
    }
}

__global__ void __launch_bounds__(256) write16(double2 *__restrict__ a, size_t n2, double v)
{
    for (size_t e = blockIdx.x * 256ul + threadIdx.x; e < n2; e += gridDim.x * 256ul) {
#pragma unroll
        for (int k = 0; k < NW; k++) a[k * n2 + e] = make_double2(v + k, v - k);
    }
}

static float timed(hipStream_t s, void (*launch)(hipStream_t))
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a, s));
    launch(s);
    CHECK(hipEventRecord(b, s));
    CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

static double *g_a, *g_o, *g_fresh;
static size_t g_n;
int main(int argc, char **argv)
{
    g_n = (argc > 1 ? atol(argv[1]) : 4194304);  // 4 M "elements": 33 * 8 * 4 M = 1.1 GB read, 13 * 8 * 4 M = 436 MB written
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    CHECK(hipMalloc(&g_a, (size_t)NR * 8 * g_n));
    CHECK(hipMalloc(&g_o, 8 * g_n));
    CHECK(hipMemsetAsync(g_a, 0, (size_t)NR * 8 * g_n, s));
    CHECK(hipStreamSynchronize(s));
    const double rb = (double)NR * 8 * g_n, wb = (double)NW * 8 * g_n;
    for (int rep = 0; rep < 3; rep++) {
        float t1 = timed(s, [](hipStream_t st) { hipLaunchKernelGGL(read8, dim3(2048), dim3(256), 0, st, g_a, g_n, g_o); });
        float t2 = timed(s, [](hipStream_t st) { hipLaunchKernelGGL(read16, dim3(2048), dim3(256), 0, st, (const double2 *)g_a, g_n / 2, g_o); });
        float t3 = timed(s, [](hipStream_t st) { hipLaunchKernelGGL(write8, dim3(2048), dim3(256), 0, st, g_a, g_n, 1.0); });
        float t4 = timed(s, [](hipStream_t st) { hipLaunchKernelGGL(write16, dim3(2048), dim3(256), 0, st, (double2 *)g_a, g_n / 2, 2.0); });
        printf("rep %d: read8 %.1f MB in %.1f us = %.2f TB/s | read16 %.2f TB/s | write8 %.1f MB in %.1f us = %.2f TB/s | write16 %.2f TB/s\n", rep,
               rb / 1e6, t1 * 1e3, rb / t1 / 1e9, rb / t2 / 1e9, wb / 1e6, t3 * 1e3, wb / t3 / 1e9, wb / t4 / 1e9);
    }
    // first touch of freshly allocated memory (k_init_tangent: 27 arrays of nel doubles, 21 ms at 1024^2 on its first launch)
    for (int trial = 0; trial < 2; trial++) {
        CHECK(hipMalloc(&g_fresh, (size_t)NW * 8 * g_n));
        float f1 = timed(s, [](hipStream_t st) { hipLaunchKernelGGL(write8, dim3(2048), dim3(256), 0, st, g_fresh, g_n, 3.0); });
        float f2 = timed(s, [](hipStream_t st) { hipLaunchKernelGGL(write8, dim3(2048), dim3(256), 0, st, g_fresh, g_n, 4.0); });
        printf("first touch, trial %d: %.1f MB fresh hipMalloc: first launch %.1f us, second launch %.1f us\n", trial, wb / 1e6, f1 * 1e3, f2 * 1e3);
        CHECK(hipFree(g_fresh));
    }
    printf("expected per launch: read kernels %.0f bytes, write kernels %.0f bytes\n", rb, wb);
    return 0;
}
