// ubench.hip — gfx950 instruction micro-benchmarks that size the potf2 / MFMA design choices.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, long long* cyc)
{
    d4_t acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = d4_t{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if (s == 123.456) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, long long* cyc)
{
    double acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = q;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9 * threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = fma(acc[q], a, b);
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += acc[q];
    if (s == 123.456) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// readlane + fma chain typical of a register-resident rank-1 update
__global__ __launch_bounds__(64) void k_readlane_fma(double* out, int iters, long long* cyc)
{
    double a[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) a[q] = q + threadIdx.x;
    double l = 1.0 + threadIdx.x * 1e-9;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            int lo = __builtin_amdgcn_readlane(__double2loint(l), q);
            int hi = __builtin_amdgcn_readlane(__double2hiint(l), q);
            double lk = __hiloint2double(hi, lo);
            a[q] = fma(-l, lk, a[q]);
        }
        l = l * 0.999999;
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += a[q];
    if (s == 123.456) out[0] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// dependent chain: pivot broadcast -> rsq + newton -> scale  (one potf2 column step's serial part)
__global__ __launch_bounds__(64) void k_pivot_chain(double* out, int iters, long long* cyc)
{
    double d = 2.0 + threadIdx.x * 1e-6;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        int lo = __builtin_amdgcn_readlane(__double2loint(d), 5);
        int hi = __builtin_amdgcn_readlane(__double2hiint(d), 5);
        double p = __hiloint2double(hi, lo);
        double y = __builtin_amdgcn_rsq(p);
        // one Newton step on y, then corrected sqrt
        double h = 0.5 * y;
        double g = p * y;
        double r = fma(-g, h, 0.5);
        g = fma(g, r, g);
        h = fma(h, r, h);
        double e = fma(-g, g, p);
        g = fma(e, h, g); // sqrt(p)
        double inv = 2.0 * h; // ~1/sqrt(p)
        d = d * inv + g * 1e-9;
    }
    long long t1 = clock64();
    if (d == 123.456) out[0] = d;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ __launch_bounds__(64) void k_sqrt_div_chain(double* out, int iters, long long* cyc)
{
    double d = 2.0 + threadIdx.x * 1e-6;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        double g = sqrt(d);
        double inv = 1.0 / g;
        d = d * inv + 1.0;
    }
    long long t1 = clock64();
    if (d == 123.456) out[0] = d;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// LDS write -> barrier -> broadcast read round trip with 4 waves
__global__ __launch_bounds__(256) void k_lds_bcast(double* out, int iters, long long* cyc)
{
    __shared__ double buf[2][64];
    double v = threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if ((threadIdx.x >> 6) == (it & 3)) buf[it & 1][threadIdx.x & 63] = v;
        __syncthreads();
        v = v * 0.5 + buf[it & 1][it & 63];
    }
    long long t1 = clock64();
    if (v == 123.456) out[0] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// mfma dependent latency
__global__ __launch_bounds__(64) void k_mfma_dep(double* out, int iters, long long* cyc)
{
    d4_t acc = {0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    long long t1 = clock64();
    if (acc[0] == 123.456) out[0] = acc[0];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename F>
void timeit(const char* name, F launch, double flops_total, int iters_per, long long* dcyc)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
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
    hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %9.3f ms  cyc/iter %9.2f  eff_clk %6.3f GHz", name, best, (double)c / iters_per, c / (best * 1e6));
    if (flops_total > 0) printf("  %8.2f TFLOP/s", flops_total / (best * 1e-3) / 1e12);
    printf("\n");
}

int main()
{
    double* d;
    long long* c;
    hipMalloc(&d, 64);
    hipMalloc(&c, 64);
    const int it = 20000;
    for (int blocks : {256, 512, 1024}) {
        printf("-- mfma f64 16x16x4, %d blocks x 4 waves\n", blocks);
        timeit("mfma 2 acc", [&] { hipLaunchKernelGGL(k_mfma<2>, dim3(blocks), dim3(256), 0, 0, d, it, c); }, (double)blocks * 4 * it * 2 * 2048, it * 2, c);
        timeit("mfma 4 acc", [&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, d, it, c); }, (double)blocks * 4 * it * 4 * 2048, it * 4, c);
        timeit("mfma 8 acc", [&] { hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(256), 0, 0, d, it, c); }, (double)blocks * 4 * it * 8 * 2048, it * 8, c);
        timeit("mfma 16 acc", [&] { hipLaunchKernelGGL(k_mfma<16>, dim3(blocks), dim3(256), 0, 0, d, it, c); }, (double)blocks * 4 * it * 16 * 2048, it * 16, c);
    }
    printf("-- mfma single wave on chip (1 block x 64)\n");
    timeit("mfma dep chain", [&] { hipLaunchKernelGGL(k_mfma_dep, dim3(1), dim3(64), 0, 0, d, it, c); }, 0, it, c);
    timeit("mfma 8acc 1blk", [&] { hipLaunchKernelGGL(k_mfma<8>, dim3(1), dim3(256), 0, 0, d, it, c); }, 0, it * 8, c);
    for (int blocks : {256, 1024}) {
        printf("-- v_fma_f64, %d blocks x 4 waves\n", blocks);
        timeit("fma 8 acc", [&] { hipLaunchKernelGGL(k_fma<8>, dim3(blocks), dim3(256), 0, 0, d, it, c); }, (double)blocks * 256 * it * 8 * 2, it * 8, c);
        timeit("fma 16 acc", [&] { hipLaunchKernelGGL(k_fma<16>, dim3(blocks), dim3(256), 0, 0, d, it, c); }, (double)blocks * 256 * it * 16 * 2, it * 16, c);
    }
    printf("-- single wave latency probes\n");
    timeit("fma dep chain(1acc)", [&] { hipLaunchKernelGGL(k_fma<1>, dim3(1), dim3(64), 0, 0, d, it, c); }, 0, it, c);
    timeit("fma 8acc 1 wave", [&] { hipLaunchKernelGGL(k_fma<8>, dim3(1), dim3(64), 0, 0, d, it, c); }, 0, it * 8, c);
    timeit("readlane x2 + fma (per)", [&] { hipLaunchKernelGGL(k_readlane_fma, dim3(1), dim3(64), 0, 0, d, it, c); }, 0, it * 32, c);
    timeit("pivot chain rsq+newton", [&] { hipLaunchKernelGGL(k_pivot_chain, dim3(1), dim3(64), 0, 0, d, it, c); }, 0, it, c);
    timeit("sqrt+div chain", [&] { hipLaunchKernelGGL(k_sqrt_div_chain, dim3(1), dim3(64), 0, 0, d, it, c); }, 0, it, c);
    timeit("lds wr->barrier->bcast rd", [&] { hipLaunchKernelGGL(k_lds_bcast, dim3(1), dim3(256), 0, 0, d, it, c); }, 0, it, c);
    return 0;
}
