// ubench2.hip — fp64 matrix-core rate variants on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));

// variant 0: 16x16x4, distinct a/b regs per mfma; variant 1: 4x4x4 (4 blocks); variant 2: mfma + fma mixed
template <int VAR, int NACC>
__global__ __launch_bounds__(256) void k_var(double* out, int iters, long long* cyc)
{
    d4_t acc[NACC];
    double va[NACC], vb[NACC], f[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
        acc[q] = d4_t{0, 0, 0, 0};
        va[q] = 1.0 + threadIdx.x * 1e-3 + q;
        vb[q] = 1.0 - threadIdx.x * 1e-3 - q;
        f[q] = q;
    }
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            if (VAR == 0 || VAR == 2)
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[q], vb[q], acc[q], 0, 0, 0);
            if (VAR == 1)
                acc[q][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(va[q], vb[q], acc[q][0], 0, 0, 0);
            if (VAR == 2) {
                f[q] = fma(f[q], va[q], vb[q]);
                f[q] = fma(f[q], vb[q], va[q]);
                f[q] = fma(f[q], va[q], vb[q]);
                f[q] = fma(f[q], vb[q], va[q]);
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3] + f[q];
    if (s == 123.456) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename F>
void timeit(const char* name, F launch, double flops_total, int iters_per, long long* dcyc)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    long long c;
    (void)hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s %9.3f ms  cyc/iter %9.2f", name, best, (double)c / iters_per);
    if (flops_total > 0) printf("  %8.2f TFLOP/s", flops_total / (best * 1e-3) / 1e12);
    printf("\n");
}

int main()
{
    double* d;
    long long* c;
    (void)hipMalloc(&d, 64);
    (void)hipMalloc(&c, 64);
    const int it = 10000;
    for (int blocks : {256, 512, 1024, 2048}) {
        printf("-- %d blocks x 4 waves\n", blocks);
        timeit("16x16x4 distinct ab, 4 acc", [&] { hipLaunchKernelGGL((k_var<0, 4>), dim3(blocks), dim3(256), 0, 0, d, it, c); }, (double)blocks * 4 * it * 4 * 2048, it * 4, c);
        timeit("16x16x4 distinct ab, 8 acc", [&] { hipLaunchKernelGGL((k_var<0, 8>), dim3(blocks), dim3(256), 0, 0, d, it, c); }, (double)blocks * 4 * it * 8 * 2048, it * 8, c);
        timeit("4x4x4_4b, 8 acc", [&] { hipLaunchKernelGGL((k_var<1, 8>), dim3(blocks), dim3(256), 0, 0, d, it, c); }, (double)blocks * 4 * it * 8 * 512, it * 8, c);
        timeit("16x16x4 + 4 fma each, 8 acc (mfma flops)", [&] { hipLaunchKernelGGL((k_var<2, 8>), dim3(blocks), dim3(256), 0, 0, d, it, c); }, (double)blocks * 4 * it * 8 * 2048, it * 8, c);
    }
    return 0;
}
