// ubench9.hip — what does clock64() count?  Ratio of clock64 (s_memtime) to wall_clock64 (100 MHz s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long long* out)
{
    long long c0 = clock64(), w0 = wall_clock64();
    double x = threadIdx.x;
    for (int i = 0; i < 200000; ++i)
        x = fma(x, 1.0000001, 0.5);
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
int main()
{
    long long* d; long long h[3];
    hipMalloc(&d, 64);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipDeviceSynchronize(); }
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("clock64 ticks %lld, wall ticks (10 ns) %lld -> clock64 runs at %.1f MHz; 200000 dependent FMAs: %.2f clock64 ticks each\n", h[0], h[1],
           100.0 * h[0] / h[1], (double)h[0] / 200000);
    return 0;
}
