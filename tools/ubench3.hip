// ubench3.hip — can 4 x v_mfma_f64_4x4x4_4b (cbsz=2, abid=r) replace one v_mfma_f64_16x16x4?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4_t __attribute__((ext_vector_type(4)));

__global__ void k_cmp(const double* a, const double* b, double* out16, double* out4)
{
    int l = threadIdx.x;
    d4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 0);
    double r0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, 0, 0);
    double r1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, 1, 0);
    double r2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, 2, 0);
    double r3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, 3, 0);
    for (int r = 0; r < 4; ++r) out16[l * 4 + r] = c[r];
    out4[l * 4 + 0] = r0;
    out4[l * 4 + 1] = r1;
    out4[l * 4 + 2] = r2;
    out4[l * 4 + 3] = r3;
}

template <int NT>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters)
{
    double acc[NT][4];
    double va[NT], vb[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        va[q] = 1.0 + threadIdx.x * 1e-3 + q;
        vb[q] = 1.0 - threadIdx.x * 1e-3 - q;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] = 0;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            acc[q][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(va[q], vb[q], acc[q][0], 2, 0, 0);
            acc[q][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(va[q], vb[q], acc[q][1], 2, 1, 0);
            acc[q][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(va[q], vb[q], acc[q][2], 2, 2, 0);
            acc[q][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(va[q], vb[q], acc[q][3], 2, 3, 0);
        }
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < NT; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if (s == 123.456) out[0] = s;
}

int main()
{
    double ha[64], hb[64], h16[256], h4[256];
    for (int i = 0; i < 64; ++i) {
        ha[i] = sin(i * 1.7 + 0.3);
        hb[i] = cos(i * 0.9 + 0.1);
    }
    double *da, *db, *d16, *d4;
    (void)hipMalloc(&da, 512);
    (void)hipMalloc(&db, 512);
    (void)hipMalloc(&d16, 2048);
    (void)hipMalloc(&d4, 2048);
    (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cmp, dim3(1), dim3(64), 0, 0, da, db, d16, d4);
    (void)hipMemcpy(h16, d16, 2048, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h4, d4, 2048, hipMemcpyDeviceToHost);
    // reference from the documented 16x16x4 layout: A[i=l%16][k=l/16], B[k=l/16][j=l%16], D lane l reg r -> (i=4r+l/16, j=l%16)
    double maxd = 0, maxref = 0;
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            int i = 4 * r + l / 16, j = l % 16;
            double ref = 0;
            for (int k = 0; k < 4; ++k) ref += ha[i + 16 * k] * hb[j + 16 * k];
            maxref = fmax(maxref, fabs(h16[l * 4 + r] - ref));
            double d = fabs(h16[l * 4 + r] - h4[l * 4 + r]);
            if (d > 1e-15) ++bad;
            maxd = fmax(maxd, d);
        }
    printf("16x16x4 vs documented layout: max|diff| = %.3e\n", maxref);
    printf("4 x 4x4x4(cbsz=2,abid=r) vs 16x16x4: max|diff| = %.3e, mismatches = %d / 256\n", maxd, bad);
    if (bad) {
        for (int l = 0; l < 8; ++l) printf("lane %d: 16x16: %+.4f %+.4f %+.4f %+.4f | 4x4: %+.4f %+.4f %+.4f %+.4f\n", l, h16[l * 4], h16[l * 4 + 1], h16[l * 4 + 2], h16[l * 4 + 3], h4[l * 4], h4[l * 4 + 1], h4[l * 4 + 2], h4[l * 4 + 3]);
    }
    double* d;
    (void)hipMalloc(&d, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int it = 10000;
    for (int blocks : {256, 512, 1024}) {
        hipLaunchKernelGGL((k_rate<4>), dim3(blocks), dim3(256), 0, 0, d, it);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_rate<4>), dim3(blocks), dim3(256), 0, 0, d, it);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("4x(4x4x4 bcast) as 16x16x4, 4 tiles(16 acc), %4d blocks: %.3f ms  %.2f TFLOP/s\n", blocks, ms, (double)blocks * 4 * it * 4 * 2048 / (ms * 1e-3) / 1e12);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_rate<16>), dim3(blocks), dim3(256), 0, 0, d, it);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("4x(4x4x4 bcast) as 16x16x4, 16 tiles(64 acc), %4d blocks: %.3f ms  %.2f TFLOP/s\n", blocks, ms, (double)blocks * 4 * it * 16 * 2048 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
