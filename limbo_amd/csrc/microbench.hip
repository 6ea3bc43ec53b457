// microbench.hip — roofline denominators measured on the device itself.
//
// MI355X_MICROARCH.md has no fp64-MFMA row.  The instruction the engine uses is
// v_mfma_f64_4x4x4_4b (gemm.hip): a register-only stream of independent ones (16 accumulators
// per wave, 4 waves per SIMD) measures 76-77 TFLOP/s, i.e. the 78.6 TFLOP/s datasheet peak;
// v_mfma_f64_16x16x4 tops out at 47-49 TFLOP/s on the same chip (tools/ubench2.hip).
#include "dev.h"

__global__ __launch_bounds__(256) void k_mfma_peak(double* out, int iters)
{
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
        acc[q] = 0.0;
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
            acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[q], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q)
        s += acc[q];
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
    const int blocks = 256 * 4; // 4 workgroups of 4 waves per CU
    hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, s, d, 64); // warm-up
    double best = 0.0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, s, d, iters);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4.0 * iters * 16.0 * 512.0; // 4x4x4_4b: 4 blocks x 2*4*4*4
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
