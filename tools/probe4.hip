#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out)
{
    int l = threadIdx.x;
    for (int lb = 0; lb < 64; ++lb) {
        double a = l + 1.0, b = (l == lb) ? 1.0 : 0.0;
        double r = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[lb * 64 + l] = r;
    }
}
int main()
{
    double* d;
    static double h[4096];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int lb = 0; lb < 64; ++lb)
        for (int l = 0; l < 64; ++l)
            if (h[lb * 64 + l] != 0.0) printf("T %d %d %d\n", (int)h[lb * 64 + l] - 1, lb, l); // la lb ld
    return 0;
}
