// ubench7.hip — cost of the building blocks of one factorisation round of the 64 x 64 diagonal block
// (potrf.hip: DiagRound), one workgroup of 5 waves as in k_diag; cycles per iteration from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define IT 512

static __device__ __forceinline__ double bcast_lane(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__global__ void k_parts(double* out, long long* cyc)
{
    __shared__ double sh[5 * 64 * 4];
    double x = 1.0 + threadIdx.x * 1e-3, y = 0.5;
    long long t[8];
    // 0: s_barrier alone
    t[0] = clock64();
#pragma unroll 1
    for (int i = 0; i < IT; ++i)
        __syncthreads();
    // 1: LDS write -> barrier -> LDS read of a neighbour wave's value (dependent)
    t[1] = clock64();
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        sh[threadIdx.x] = x;
        __syncthreads();
        x += sh[(threadIdx.x + 64) % 320];
        __syncthreads();
    }
    // 2: dependent v_rsq_f64 + Newton chain as in RsqScale (rsq, mul, fma, mul, fma)
    t[2] = clock64();
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        const double y0 = __builtin_amdgcn_rsq(x);
        const double tt = (0.5 * x) * y0;
        const double eh = fma(-tt, y0, 0.5);
        const double l = x * y0;
        x = fma(l, eh, l) + 1.0;
    }
    // 3: readlane broadcast feeding a VALU op feeding the next readlane
    t[3] = clock64();
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        const double b = bcast_lane(x, i & 63);
        x = fma(b, 1e-9, x);
    }
    // 4: dependent ds_read_b128 x2 from a wave-uniform address + 4 dependent FMAs (one column of rank4_update)
    t[4] = clock64();
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        const double* lc = sh + ((i * 4) & 255);
        y = fma(-x, lc[0], y);
        y = fma(-x, lc[1], y);
        y = fma(-x, lc[2], y);
        y = fma(-x, lc[3], y);
        asm volatile("" : "+v"(y));
    }
    // 5: dependent fp64 FMA chain
    t[5] = clock64();
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        x = fma(x, 1.0000001, y);
        x = fma(x, 1.0000001, y);
        x = fma(x, 1.0000001, y);
        x = fma(x, 1.0000001, y);
    }
    t[6] = clock64();
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; ++k)
            cyc[k] = t[k + 1] - t[k];
    out[threadIdx.x] = x + y;
}

int main()
{
    double* out;
    long long* cyc;
    CHK(hipMalloc(&out, 8 * 1024));
    CHK(hipMalloc(&cyc, 64));
    long long h[6];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_parts, dim3(1), dim3(320), 0, 0, out, cyc);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h, cyc, 48, hipMemcpyDeviceToHost));
    }
    const char* nm[6] = {"s_barrier (5 waves)", "ds_write, barrier, ds_read, barrier", "rsq + Newton scale chain (5 dep. ops)",
                         "readlane x2 -> fma -> readlane", "uniform ds_read_b128 x2 + 4 dependent fma", "4 dependent fma"};
    for (int k = 0; k < 6; ++k)
        printf("%-44s %8.1f cycles per iteration\n", nm[k], (double)h[k] / IT);
    printf("ubench7 done\n");
    return 0;
}
