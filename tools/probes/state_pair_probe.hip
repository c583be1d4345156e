// Element-state access pattern of the sweep (read sig[6], epl[6], elstiff[21]; write res_sig[6], res_depl[6]) with the arrays
// as SoA of doubles (33 eight-byte loads per element) against SoA of double2 pairs (3 + 3 + 11 sixteen-byte loads).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/state_pair_probe tools/probes/state_pair_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PAIR>
__global__ void __launch_bounds__(256) k(int nel, const double *sig, const double *epl, const double *est, double *rs, double *rd)
{
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nel; e += gridDim.x * 256) {
        double s[6], p[6], D[22];
        if (PAIR) {
            const double2 *s2 = (const double2 *)sig, *p2 = (const double2 *)epl, *d2 = (const double2 *)est;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double2 a = s2[(size_t)k * nel + e], b = p2[(size_t)k * nel + e];
                s[2 * k] = a.x, s[2 * k + 1] = a.y, p[2 * k] = b.x, p[2 * k + 1] = b.y;
            }
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const double2 a = d2[(size_t)k * nel + e];
                D[2 * k] = a.x, D[2 * k + 1] = a.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) s[k] = sig[(size_t)k * nel + e], p[k] = epl[(size_t)k * nel + e];
#pragma unroll
            for (int k = 0; k < 21; k++) D[k] = est[(size_t)k * nel + e];
            D[21] = 0.;
        }
        double o[6], q[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            o[i] = s[i];
            q[i] = p[i];
#pragma unroll
            for (int j = 0; j < 21; j++) o[i] = fma(D[j], p[(i + j) % 6], o[i]), q[i] = fma(D[(j + 3) % 21], s[(i + j) % 6], q[i]);
        }
        if (PAIR) {
            double2 *r2 = (double2 *)rs, *q2 = (double2 *)rd;
#pragma unroll
            for (int k = 0; k < 3; k++) r2[(size_t)k * nel + e] = make_double2(o[2 * k], o[2 * k + 1]), q2[(size_t)k * nel + e] = make_double2(q[2 * k], q[2 * k + 1]);
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) rs[(size_t)k * nel + e] = o[k], rd[(size_t)k * nel + e] = q[k];
        }
    }
}
int main()
{
    const int nel = 1 << 20;
    double *sig, *epl, *est, *rs, *rd;
    hipMalloc(&sig, 6 * 8 * (size_t)nel); hipMalloc(&epl, 6 * 8 * (size_t)nel); hipMalloc(&est, 22 * 8 * (size_t)nel);
    hipMalloc(&rs, 6 * 8 * (size_t)nel); hipMalloc(&rd, 6 * 8 * (size_t)nel);
    hipMemset(sig, 0, 6 * 8 * (size_t)nel); hipMemset(epl, 0, 6 * 8 * (size_t)nel); hipMemset(est, 0, 22 * 8 * (size_t)nel);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1024, 4096}) for (int pair = 0; pair < 2; pair++) {
        float best = 1e9f;
        for (int rep = 0; rep < 30; rep++) {
            hipEventRecord(e0);
            if (pair) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, nel, sig, epl, est, rs, rd);
            else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, nel, sig, epl, est, rs, rd);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 5 && ms < best) best = ms;
        }
        const double bytes = (double)nel * (pair ? 34 + 12 : 33 + 12) * 8;
        printf("grid %5d %s: %.2f us  %.0f GB/s\n", grid, pair ? "pairs " : "SoA x1", best * 1e3, bytes / (best * 1e-3) / 1e9);
    }
    return 0;
}
