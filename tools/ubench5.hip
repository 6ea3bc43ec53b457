// ubench5.hip — v_mfma_f64_4x4x4_4b dependent-accumulator latency vs number of independent chains
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC>
__global__ __launch_bounds__(64) void k(double* out, int iters, long long* cyc)
{
    double acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = 0.0;
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[q], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += acc[q];
    if (s == 123.456) out[0] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int N> void run(double* d, long long* c)
{
    const int it = 20000;
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, d, it, c);
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, d, it, c);
    (void)hipDeviceSynchronize();
    long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%2d independent accumulators: %.1f cycles per MFMA (%.1f per round of %d)\n", N, (double)h / it / N, (double)h / it, N);
}
int main()
{
    double* d; long long* c;
    (void)hipMalloc(&d, 64); (void)hipMalloc(&c, 64);
    run<1>(d, c); run<2>(d, c); run<4>(d, c); run<8>(d, c); run<16>(d, c);
    return 0;
}
