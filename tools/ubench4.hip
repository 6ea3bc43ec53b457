// ubench4.hip — why does the GEMM inner loop (RA=4 x RB=16 4x4x4 MFMAs per k4-step) run at ~63 cycles/MFMA?
#include <hip/hip_runtime.h>
#include <cstdio>
static __device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// MODE 0: operands re-read from LDS every k4-step (as in gemm.hip). MODE 1: operands read once, reused.
// MODE 2: like 0 but RA=2,RB=8 (16 accs).  MODE 3: like 0 but n-outer order swapped (m outer)
template <int MODE, int RA, int RB>
__global__ __launch_bounds__(256) void k_loop(double* out, int iters)
{
    __shared__ double As[16 * 144], Bs[16 * 144];
    for (int e = threadIdx.x; e < 16 * 144; e += 256) { As[e] = 1.0 + e * 1e-6; Bs[e] = 1.0 - e * 1e-6; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    const int arow = wm + (lane & 15), bcol = wn + (lane & 3), kq = lane >> 4;
    double acc[RA][RB];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < RB; ++b) acc[a][b] = 0.0;
    double af[RA], bf[RB];
    if (MODE == 1) {
#pragma unroll
        for (int x = 0; x < RA; ++x) af[x] = As[kq * 144 + arow + 16 * x];
#pragma unroll
        for (int x = 0; x < RB; ++x) bf[x] = Bs[kq * 144 + bcol + 4 * x];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 16; ks += 4) {
            if (MODE != 1) {
#pragma unroll
                for (int x = 0; x < RA; ++x) af[x] = As[(ks + kq) * 144 + arow + 16 * x];
#pragma unroll
                for (int x = 0; x < RB; ++x) bf[x] = Bs[(ks + kq) * 144 + bcol + 4 * x];
            }
            if (MODE == 3) {
#pragma unroll
                for (int m = 0; m < RA; ++m)
#pragma unroll
                    for (int n = 0; n < RB; ++n) acc[m][n] = mfma4(af[m], bf[n], acc[m][n]);
            } else {
#pragma unroll
                for (int n = 0; n < RB; ++n)
#pragma unroll
                    for (int m = 0; m < RA; ++m) acc[m][n] = mfma4(af[m], bf[n], acc[m][n]);
            }
        }
        if (MODE != 1) __syncthreads();
    }
    double s = 0;
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < RB; ++b) s += acc[a][b];
    if (s == 123.456) out[0] = s;
}

template <typename F>
void run(const char* name, F launch, double mfmas)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.3f ms  %7.2f TFLOP/s  (%.1f ns per MFMA per wave)\n", name, ms, mfmas * 512 / (ms * 1e-3) / 1e12, ms * 1e6 / (mfmas / (256.0 * 4)) );
}

int main()
{
    double* d; (void)hipMalloc(&d, 64);
    const int it = 2000;
    for (int blocks : {256, 512}) {
        printf("-- %d blocks (x4 waves)\n", blocks);
        run("lds-fed 4x16 (gemm order)", [&] { hipLaunchKernelGGL((k_loop<0, 4, 16>), dim3(blocks), dim3(256), 0, 0, d, it); }, (double)blocks * 4 * it * 4 * 64);
        run("reg-resident operands 4x16", [&] { hipLaunchKernelGGL((k_loop<1, 4, 16>), dim3(blocks), dim3(256), 0, 0, d, it); }, (double)blocks * 4 * it * 4 * 64);
        run("lds-fed 4x16 (m outer)", [&] { hipLaunchKernelGGL((k_loop<3, 4, 16>), dim3(blocks), dim3(256), 0, 0, d, it); }, (double)blocks * 4 * it * 4 * 64);
        run("lds-fed 2x8", [&] { hipLaunchKernelGGL((k_loop<0, 2, 8>), dim3(blocks), dim3(256), 0, 0, d, it); }, (double)blocks * 4 * it * 4 * 16);
        run("reg-resident 2x8", [&] { hipLaunchKernelGGL((k_loop<1, 2, 8>), dim3(blocks), dim3(256), 0, 0, d, it); }, (double)blocks * 4 * it * 4 * 16);
    }
    return 0;
}
