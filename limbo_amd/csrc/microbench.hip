// microbench.hip — roofline denominators measured on the device itself.
//
// MI355X_MICROARCH.md has no fp64-MFMA row, so the peak used for `roofline.frac` of the
// trailing update is measured here: a register-only stream of independent
// v_mfma_f64_16x16x4_f64 (8 accumulators per wave, 1..2 waves per SIMD).
#include "dev.h"

__global__ __launch_bounds__(256) void k_mfma_peak(double* out, int iters)
{
    d4_t acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
        acc[q] = d4_t{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            acc[q] = mfma_f64(a, b, acc[q]);
    }
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
        s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if (s == 123.456)
        out[0] = s; // keep the chain live
}

double run_mfma_f64_peak(hipStream_t s)
{
    double* d = nullptr;
    if (hipMalloc(&d, 64) != hipSuccess)
        return -1.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4096;
    const int blocks = 256 * 2; // 2 workgroups of 4 waves per CU
    hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, s, d, 64); // warm-up
    double best = 0.0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, s, d, iters);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4.0 * iters * 8.0 * 2.0 * 16 * 16 * 4;
        double tf = flops / (ms * 1e-3) / 1e12;
        if (tf > best)
            best = tf;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d);
    return best;
}

__global__ __launch_bounds__(256) void k_stream_write(double2* __restrict__ p, int64_t n2, double v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n2; i += stride)
        p[i] = double2{v, v + 1.0};
}

double run_hbm_stream_peak(hipStream_t s)
{
    const int64_t bytes = (int64_t)1 << 30; // 1 GiB, well past the 256 MiB Infinity Cache
    double2* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess)
        return -1.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int64_t n2 = bytes / 16;
    hipLaunchKernelGGL(k_stream_write, dim3(2048), dim3(256), 0, s, d, n2, 1.0);
    double best = 0.0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_stream_write, dim3(2048), dim3(256), 0, s, d, n2, (double)rep);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        double gbs = (double)bytes / (ms * 1e-3) / 1e9;
        if (gbs > best)
            best = gbs;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d);
    return best;
}
