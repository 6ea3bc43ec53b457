// lat.hip — instruction latencies that bound the pivot chain of the 64 x 64 diagonal block (diag_flow.h, wave P): one wave,
// dependent chains of 256 operations between two clock64() reads.  make -C tools lat && tools/lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
__global__ void k_lat(double* out, long long* cyc, double seed, int lane_sel)
{
    __shared__ double lds[1024];
    __shared__ int flag[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double v = seed + lane * 1e-9, w = seed * 0.5;
    long long t0, t1;
    if (threadIdx.x < 4)
        flag[threadIdx.x] = 0;
    __syncthreads();
    if (wave == 0) {
        // 0: dependent fma
        t0 = clock64();
#pragma unroll
        for (int i = 0; i < N; ++i)
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v) : "v"(w));
        t1 = clock64();
        if (lane == 0) cyc[0] = t1 - t0;
        // 1: dependent mul
        t0 = clock64();
#pragma unroll
        for (int i = 0; i < N; ++i)
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v) : "v"(w));
        t1 = clock64();
        if (lane == 0) cyc[1] = t1 - t0;
        v = seed + 1.0;
        // 2: dependent rsq
        t0 = clock64();
#pragma unroll
        for (int i = 0; i < N; ++i)
            asm volatile("v_rsq_f64 %0, %0" : "+v"(v));
        t1 = clock64();
        if (lane == 0) cyc[2] = t1 - t0;
        // 3: dependent rcp
        t0 = clock64();
#pragma unroll
        for (int i = 0; i < N; ++i)
            asm volatile("v_rcp_f64 %0, %0" : "+v"(v));
        t1 = clock64();
        if (lane == 0) cyc[3] = t1 - t0;
        // 4: fma -> readlane -> fma (the scalar feeds the next fma)
        double s = seed;
        t0 = clock64();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            v = fma(v, s, w);
            s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 5), __builtin_amdgcn_readlane(__double2loint(v), 5));
        }
        t1 = clock64();
        if (lane == 0) cyc[4] = t1 - t0;
        v += s;
        // 5: LDS write -> read of another lane's value (same wave)
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; ++i) {
            lds[lane] = v;
            v = lds[(lane + 1) & 63] + 1.0;
        }
        t1 = clock64();
        if (lane == 0) cyc[5] = t1 - t0;
        // 6: independent fma issue rate (8 chains)
        double a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            a[k] = v + k;
        t0 = clock64();
#pragma unroll
        for (int i = 0; i < N / 8; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[k]) : "v"(w));
        t1 = clock64();
        if (lane == 0) cyc[6] = t1 - t0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v += a[k];
        // 7: rsq + one Newton step (rsq_newton of diag_flow.h), dependent
        double p = seed + 2.0;
        t0 = clock64();
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const double y0 = __builtin_amdgcn_rsq(p);
            const double t = (0.5 * p) * y0;
            const double eh = fma(-t, y0, 0.5);
            p = fma(y0, eh, y0) + 1.0;
        }
        t1 = clock64();
        if (lane == 0) cyc[7] = t1 - t0; // N/4 x (rsq, mul, fma, fma, add)
        v += p;
        // 8: the same through a uniform (SGPR) value: readfirstlane in the chain
        // 9: independent rsq issue rate
        double b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            b[k] = seed + k + 1.0;
        t0 = clock64();
#pragma unroll
        for (int i = 0; i < N / 8; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_rsq_f64 %0, %0" : "+v"(b[k]));
        t1 = clock64();
        if (lane == 0) cyc[9] = t1 - t0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v += b[k];
        // 10: LDS ds_read latency alone (dependent address)
        int idx = lane;
        int* il = (int*)lds;
        il[lane] = (lane + 1) & 63;
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; ++i)
            idx = il[idx];
        t1 = clock64();
        if (lane == 0) cyc[10] = t1 - t0;
        v += idx;
    }
    // 11: ping-pong between wave 0 and wave 1 through LDS counters (lds_post / lds_await)
    __syncthreads();
    if (wave < 2 && blockDim.x >= 128) {
        volatile int* f = flag;
        t0 = clock64();
        for (int i = 1; i <= N; ++i) {
            int spins = 0;
            if (wave == 0) {
                if (lane == 0) f[0] = i;
                while (f[1] < i && ++spins < 100000) {}
            }
            else {
                while (f[0] < i && ++spins < 100000) {}
                if (lane == 0) f[1] = i;
            }
        }
        t1 = clock64();
        if (threadIdx.x == 0) cyc[11] = t1 - t0; // N round trips = 2 N one-way hand-overs
    }
    out[threadIdx.x] = v;
}
int main()
{
    double* out;
    long long* cyc;
    hipMalloc(&out, 8 * 512);
    hipMalloc(&cyc, 8 * 16);
    hipMemset(cyc, 0, 8 * 16);
    for (int threads : {64, 512}) {
        for (int r = 0; r < 2; ++r)
            hipLaunchKernelGGL(k_lat, dim3(1), dim3(threads), 0, 0, out, cyc, 1.000001, 5);
        hipDeviceSynchronize();
        long long h[16];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        const char* nm[] = {"dependent v_fma_f64", "dependent v_mul_f64", "dependent v_rsq_f64", "dependent v_rcp_f64", "fma -> v_readlane x2 -> fma",
                            "ds_write_b64 -> ds_read_b64 (+ add)", "independent v_fma_f64 (8 chains)", "rsq + Newton + add (5 ops)", "", "independent v_rsq_f64 (8 chains)",
                            "dependent ds_read_b32", "LDS counter ping-pong, one way"};
        printf("# %d threads in the workgroup (the other waves idle at a barrier)\n", threads);
        for (int i = 0; i < 12; ++i) {
            if (i == 8) continue;
            double per = (double)h[i] / N;
            if (i == 7) per = (double)h[i] / (N / 4);
            if (i == 11) per = (double)h[i] / (2 * N);
            printf("%-40s %8.1f cycles per %s\n", nm[i], per, i == 7 ? "step of 5" : "operation");
        }
    }
    return 0;
}
