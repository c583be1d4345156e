// FP64 v_fma_f64 issue rate on gfx950: NCHAIN independent dependent-chains per lane, WPS waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCHAIN>
__global__ void __launch_bounds__(256) k_fma(double *out, int iters, double a, double b)
{
    double x[NCHAIN];
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) x[c] = threadIdx.x * 1e-9 + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < NCHAIN; c++) x[c] = fma(x[c], a, b);
    }
    double s = 0.;
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) s += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NCHAIN>
void run(int wps)
{
    double *out; hipMalloc(&out, 8 << 20);
    const int iters = 20000, blocks = 256 * wps;  // 256 CUs x wps blocks of 4 waves -> wps waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_fma<NCHAIN>, dim3(blocks), dim3(256), 0, 0, out, 10, 0.999999, 1e-7);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_fma<NCHAIN>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999999, 1e-7);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fmas_per_simd = (double)iters * 16 * NCHAIN * wps;  // wave-instructions per SIMD
    const double tf = (double)iters * 16 * NCHAIN * 2. * 64 * 4 * blocks / (ms * 1e-3) / 1e12;
    printf("chains %d  waves/SIMD %d: %.3f ms  -> %.2f ns per wave-FMA per SIMD (%.1f clk @2.4GHz), %.1f TFLOP/s\n", NCHAIN, wps, ms,
           ms * 1e6 / fmas_per_simd, ms * 1e6 / fmas_per_simd * 2.4, tf);
    hipFree(out);
}
int main()
{
    for (int wps = 1; wps <= 4; wps *= 2) { run<1>(wps); run<2>(wps); run<4>(wps); run<8>(wps); }
    return 0;
}
