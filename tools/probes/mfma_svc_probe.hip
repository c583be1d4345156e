// Probe for BASELINE config 4's wording "SVC ML yield function ... as dense SV x GP MFMA kernel" (VERDICT r2 item 8):
// what would the FP64 matrix cores buy for the support-vector sums   f(x) = sum_k dual_k exp(-gamma |x - sv_k|^2) + b ?
// The only GEMM-shaped part is the cross term  x . sv_k  of  |x - sv_k|^2 = |x|^2 - 2 x.sv_k + |sv_k|^2 :
// [16 points x 6 features] . [6 x NSV]  ->  v_mfma_f64_16x16x4_f64 with K padded from 6 to 8 (two MFMAs per 16 x 16 tile).
// Measured here, per (point, support vector) pair, on all CUs:
//   A  rate of v_mfma_f64_16x16x4_f64 (independent accumulators)            -> FP64 matrix TFLOP/s
//   B  rate of v_fma_f64                                                     -> FP64 vector TFLOP/s
//   C  decision function, cross term on MFMA (16 points per wave), exp + dual-weighted sum on the VALU
//   D  decision function all on the VALU, one wave per point, support vectors split over the lanes (what libplfx does)
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_svc_probe mfma_svc_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int NSV = 1792;      // 1585 support vectors padded to 64 * 28 (dual = 0 on the padding), as in k_sweep_svc_wave
constexpr double GAMMA = 1.0;

__device__ __forceinline__ double exp2_poly(double y)
{   // 2^y for y <= 0: round-to-nearest split + degree-11 polynomial, the 17-instruction routine of plfx_device.hpp in spirit
    const double big = 6755399441055744.0;  // 1.5 * 2^52
    const double t = y + big;
    const double n = t - big;
    const double r = (y - n) * 0.6931471805599453;
    double p = 2.505210838544172e-08;
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.755731922398589e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 1.984126984126984e-04);
    p = fma(p, r, 1.388888888888889e-03);
    p = fma(p, r, 8.333333333333333e-03);
    p = fma(p, r, 4.166666666666666e-02);
    p = fma(p, r, 1.666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const long long bits = __double_as_longlong(p) + (((long long)__double_as_longlong(t)) << 52);
    return __longlong_as_double(bits);
}

__global__ void __launch_bounds__(256) k_mfma_rate(double *out, int iters)
{
    double4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

__global__ void __launch_bounds__(256) k_fma_rate(double *out, int iters)
{
    double acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = threadIdx.x * 1e-9 + k;
    const double a = 1.0 + 1e-12, b = 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = fma(acc[k], a, b);
    }
    double s = 0.;
#pragma unroll
    for (int k = 0; k < 8; k++) s += acc[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// C: one wave = 16 points.  MFMA 16x16x4 f64 operand layout (MI355X_MICROARCH.md): A[i][k] from lane 16*k + i, B[k][j] from lane
// 16*k + j, D[(lane>>4) + 4*r][lane&15] in register r.  sv in LDS as [8][NSV] (features 6, 7 zero), sq[NSV], dual[NSV].
__global__ void __launch_bounds__(256) k_decision_mfma(const double *__restrict__ svt, const double *__restrict__ sq,
                                                        const double *__restrict__ dual, const double *__restrict__ x, int npts,
                                                        double *__restrict__ f)
{
    extern __shared__ double lds[];
    double *s_sv = lds, *s_sq = lds + 8 * NSV, *s_du = s_sq + NSV;
    for (int i = threadIdx.x; i < 8 * NSV; i += 256) s_sv[i] = svt[i];
    for (int i = threadIdx.x; i < NSV; i += 256) { s_sq[i] = sq[i]; s_du[i] = dual[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int p0 = wave * 16;
    if (p0 >= npts) return;
    const int i = lane & 15, kq = lane >> 4;
    // A operands: features kq (first MFMA) and 4 + kq (second) of point p0 + i
    const double a0 = x[(size_t)(p0 + i) * 8 + kq], a1 = x[(size_t)(p0 + i) * 8 + 4 + kq];
    // |x|^2 of the four points whose rows this lane holds in D: rows kq + 4*r
    double xx[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double t = 0.;
        for (int c = 0; c < 6; c++) { const double v = x[(size_t)(p0 + kq + 4 * r) * 8 + c]; t = fma(v, v, t); }
        xx[r] = t;
    }
    const double g = -GAMMA * 1.4426950408889634;
    double acc[4] = {0., 0., 0., 0.};
    for (int j0 = 0; j0 < NSV; j0 += 16) {
        const double b0 = s_sv[kq * NSV + j0 + i], b1 = s_sv[(4 + kq) * NSV + j0 + i];
        double4_t d = {0, 0, 0, 0};
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, d, 0, 0, 0);
        const double sqj = s_sq[j0 + i], duj = s_du[j0 + i];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double hh = fma(-2., d[r], xx[r] + sqj);
            acc[r] = fma(duj, exp2_poly(g * hh), acc[r]);
        }
    }
    // sum over the 16 lanes that share kq (columns j): butterfly inside each group of 16
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double v = acc[r];
        for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (i == 0) f[p0 + kq + 4 * r] = v;
    }
}

// D: one wave = one point; lane L takes support vectors L, L + 64, ... (SoA tables in LDS, stride-1 across lanes)
__global__ void __launch_bounds__(256) k_decision_valu(const double *__restrict__ svt, const double *__restrict__ dual,
                                                        const double *__restrict__ x, int npts, double *__restrict__ f)
{
    extern __shared__ double lds[];
    double *s_sv = lds, *s_du = lds + 6 * NSV;
    for (int i = threadIdx.x; i < 6 * NSV; i += 256) s_sv[i] = svt[i];
    for (int i = threadIdx.x; i < NSV; i += 256) s_du[i] = dual[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    const double g = -GAMMA * 1.4426950408889634;
    for (int p = (blockIdx.x * 256 + threadIdx.x) >> 6; p < npts; p += nw) {
        double xv[6];
#pragma unroll
        for (int c = 0; c < 6; c++) xv[c] = x[(size_t)p * 8 + c];
        double a0 = 0., a1 = 0.;
        for (int k = lane; k < NSV; k += 128) {
            double h0 = 0., h1 = 0.;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const double d0 = xv[c] - s_sv[c * NSV + k], d1 = xv[c] - s_sv[c * NSV + k + 64];
                h0 = fma(d0, d0, h0);
                h1 = fma(d1, d1, h1);
            }
            a0 = fma(s_du[k], exp2_poly(g * h0), a0);
            a1 = fma(s_du[k + 64], exp2_poly(g * h1), a1);
        }
        double v = a0 + a1;
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) f[p] = v;
    }
}

static float run(hipStream_t s, void (*fn)(hipStream_t), int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    fn(s);
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(a, s));
    for (int i = 0; i < reps; i++) fn(s);
    CHECK(hipEventRecord(b, s));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipGetLastError());
    return ms / reps;
}

static double *d_out, *d_svt8, *d_svt6, *d_sq, *d_dual, *d_x, *d_f1, *d_f2;
static const int NPTS = 262144, ITERS = 4096, BLOCKS = 256 * 8;
int main()
{
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    std::vector<double> sv8(8 * NSV, 0.), sv6(6 * NSV), sq(NSV, 0.), du(NSV, 0.), x((size_t)NPTS * 8, 0.);
    srand(1);
    for (int k = 0; k < 1585; k++) {
        for (int c = 0; c < 6; c++) {
            const double v = (rand() / (double)RAND_MAX - 0.5) * 2.4;
            sv8[c * NSV + k] = v;
            sq[k] += v * v;
        }
        du[k] = (rand() / (double)RAND_MAX - 0.5) * 4.;
    }
    for (int c = 0; c < 6; c++) for (int k = 0; k < NSV; k++) sv6[c * NSV + k] = sv8[c * NSV + k];
    for (int p = 0; p < NPTS; p++) for (int c = 0; c < 6; c++) x[(size_t)p * 8 + c] = (rand() / (double)RAND_MAX - 0.5) * 2.;
    CHECK(hipMalloc(&d_out, 8ul * BLOCKS * 256)); CHECK(hipMalloc(&d_svt8, 8ul * 8 * NSV)); CHECK(hipMalloc(&d_svt6, 8ul * 6 * NSV));
    CHECK(hipMalloc(&d_sq, 8ul * NSV)); CHECK(hipMalloc(&d_dual, 8ul * NSV)); CHECK(hipMalloc(&d_x, 8ul * 8 * NPTS));
    CHECK(hipMalloc(&d_f1, 8ul * NPTS)); CHECK(hipMalloc(&d_f2, 8ul * NPTS));
    CHECK(hipMemcpy(d_svt8, sv8.data(), 8ul * 8 * NSV, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_svt6, sv6.data(), 8ul * 6 * NSV, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_sq, sq.data(), 8ul * NSV, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_dual, du.data(), 8ul * NSV, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_x, x.data(), 8ul * 8 * NPTS, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute((const void *)k_decision_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 10 * NSV));
    CHECK(hipFuncSetAttribute((const void *)k_decision_valu, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 7 * NSV));
    const float tA = run(s, [](hipStream_t st) { hipLaunchKernelGGL(k_mfma_rate, dim3(BLOCKS), dim3(256), 0, st, d_out, ITERS); }, 5);
    const float tB = run(s, [](hipStream_t st) { hipLaunchKernelGGL(k_fma_rate, dim3(BLOCKS), dim3(256), 0, st, d_out, ITERS); }, 5);
    const double flA = (double)BLOCKS * 4 * ITERS * 4 * 2048., flB = (double)BLOCKS * 256 * ITERS * 8 * 2.;
    printf("A v_mfma_f64_16x16x4_f64: %.3f ms -> %.1f TFLOP/s     B v_fma_f64: %.3f ms -> %.1f TFLOP/s   (guide: FP64 vector = matrix = 78.6)\n",
           tA, flA / tA / 1e9, tB, flB / tB / 1e9);
    const float tC = run(s, [](hipStream_t st) { hipLaunchKernelGGL(k_decision_mfma, dim3(NPTS / 64), dim3(256), 8 * 10 * NSV, st, d_svt8, d_sq, d_dual, d_x, NPTS, d_f1); }, 3);
    const float tD = run(s, [](hipStream_t st) { hipLaunchKernelGGL(k_decision_valu, dim3(2048), dim3(256), 8 * 7 * NSV, st, d_svt6, d_dual, d_x, NPTS, d_f2); }, 3);
    std::vector<double> f1(NPTS), f2(NPTS);
    CHECK(hipMemcpy(f1.data(), d_f1, 8ul * NPTS, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(f2.data(), d_f2, 8ul * NPTS, hipMemcpyDeviceToHost));
    double md = 0., mx = 0.;
    for (int p = 0; p < NPTS; p++) { md = fmax(md, fabs(f1[p] - f2[p])); mx = fmax(mx, fabs(f2[p])); }
    const double pairs = (double)NPTS * NSV;
    printf("C decision function, cross term on MFMA (16 points / wave): %.3f ms = %.3f ns per (point, SV) pair\n", tC, tC * 1e6 / pairs);
    printf("D decision function on the VALU (1 point / wave, SVs over lanes): %.3f ms = %.3f ns per pair   C/D = %.2f   max |C - D| = %.2e (max |f| %.2e)\n",
           tD, tD * 1e6 / pairs, tC / tD, md, mx);
    return 0;
}
