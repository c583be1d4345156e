// Round-trip cost of "kernel writes a flag -> host reads it -> host launches the next kernel":
//   A: hipMemcpyAsync D2H + hipStreamSynchronize        (what plfx_solve / plfx_sweep do today)
//   B: kernel writes to pinned host memory, host spins   (mailbox)
//   C: hipStreamSynchronize only (flag in pinned memory, no copy)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
struct Mbox { volatile uint64_t seq; int v; };
__global__ void k_work(double *x, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] * 1.0000001 + 1.; }
__global__ void k_flag(int *f, int it) { if (threadIdx.x == 0) *f = it; }
__global__ void k_flag_mbox(Mbox *m, uint64_t seq) { if (threadIdx.x == 0) { m->v = (int)seq; __threadfence_system(); m->seq = seq; } }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int n = 1 << 20; double *x; hipMalloc(&x, n * 8); hipMemset(x, 0, n * 8);
    int *f; hipMalloc(&f, 4); int hf = 0; int *hp; hipHostMalloc(&hp, 4);
    Mbox *mb; hipHostMalloc(&mb, sizeof(Mbox), hipHostMallocMapped | hipHostMallocCoherent); mb->seq = 0;
    const int N = 2000;
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int rep = 0; rep < 2; rep++) {
        auto t0 = now();
        for (int it = 1; it <= N; it++) {
            hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, x, n);
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, f, it);
            hipMemcpyAsync(hp, f, 4, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            hf += *hp;
        }
        double a = std::chrono::duration<double, std::micro>(now() - t0).count() / N;
        t0 = now();
        for (int it = 1; it <= N; it++) {
            hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, x, n);
            const uint64_t seq = (uint64_t)rep * N + it;
            hipLaunchKernelGGL(k_flag_mbox, dim3(1), dim3(64), 0, s, mb, seq);
            while (__atomic_load_n(&mb->seq, __ATOMIC_ACQUIRE) != seq) { __builtin_ia32_pause(); }
            hf += mb->v;
        }
        double b = std::chrono::duration<double, std::micro>(now() - t0).count() / N;
        t0 = now();
        for (int it = 1; it <= N; it++) {
            hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, x, n);
            const uint64_t seq = (uint64_t)(rep + 2) * N + it;
            hipLaunchKernelGGL(k_flag_mbox, dim3(1), dim3(64), 0, s, mb, seq);
            hipStreamSynchronize(s);
            hf += mb->v;
        }
        double cc = std::chrono::duration<double, std::micro>(now() - t0).count() / N;
        t0 = now();
        for (int it = 1; it <= N; it++) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, x, n);
        hipStreamSynchronize(s);
        double d = std::chrono::duration<double, std::micro>(now() - t0).count() / N;
        printf("rep %d: per round trip  A memcpy+sync %.1f us   B mailbox spin %.1f us   C sync only %.1f us   (back-to-back kernel alone %.1f us)\n", rep, a, b, cc, d);
    }
    return hf == 0;
}
