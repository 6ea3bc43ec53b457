// ubench8.hip — LDS throughput of the access shapes used by the diagonal-block factorisation, 8 waves
// in one workgroup all issuing the same shape back to back: cycles of LDS pipe per wave instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define IT 256
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ void k_lds(double* out, long long* cyc)
{
    __shared__ __attribute__((aligned(16))) double sh[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x)
        sh[i] = i * 1e-3;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double acc = 0.0;
    long long t[8];
    // 0: wave-uniform ds_read_b128 (all lanes the same address), 8 independent per iteration
    t[0] = clock64();
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        const d2* p = (const d2*)(sh + ((i * 16 + w * 128) & 4095));
        d2 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3], v4 = p[4], v5 = p[5], v6 = p[6], v7 = p[7];
        acc += v0.x + v1.y + v2.x + v3.y + v4.x + v5.y + v6.x + v7.y;
    }
    // 1: per-lane ds_read_b128, 32-byte lane stride (the [row][4] panel)
    t[1] = clock64();
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        const d2* p = (const d2*)(sh + lane * 4 + ((i * 256) & 4095));
        d2 v0 = p[0], v1 = p[1], v2 = p[128], v3 = p[129], v4 = p[256], v5 = p[257], v6 = p[384], v7 = p[385];
        acc += v0.x + v1.y + v2.x + v3.y + v4.x + v5.y + v6.x + v7.y;
    }
    // 2: per-lane ds_read_b128, 48-byte lane stride
    t[2] = clock64();
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        const d2* p = (const d2*)(sh + lane * 6 + ((i * 384) & 2047));
        d2 v0 = p[0], v1 = p[1], v2 = p[192], v3 = p[193], v4 = p[384], v5 = p[385], v6 = p[576], v7 = p[577];
        acc += v0.x + v1.y + v2.x + v3.y + v4.x + v5.y + v6.x + v7.y;
    }
    // 3: per-lane ds_read_b64, consecutive lanes (ideal)
    t[3] = clock64();
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        const double* p = sh + lane + ((i * 64) & 4095);
        acc += p[0] + p[64] + p[128] + p[192] + p[256] + p[320] + p[384] + p[448];
    }
    // 4: wave-uniform ds_read_b64, 8 independent
    t[4] = clock64();
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        const double* p = sh + ((i * 16 + w * 128) & 4095);
        acc += p[0] + p[1] + p[2] + p[3] + p[4] + p[5] + p[6] + p[7];
    }
    // 5: ds_bpermute (shuffle) x8
    t[5] = clock64();
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {
        acc += __shfl(acc, (i + 1) & 63) + __shfl(acc, (i + 2) & 63) + __shfl(acc, (i + 3) & 63) + __shfl(acc, (i + 4) & 63);
    }
    t[6] = clock64();
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; ++k)
            cyc[k] = t[k + 1] - t[k];
    out[threadIdx.x] = acc;
}

int main()
{
    double* out;
    long long* cyc;
    CHK(hipMalloc(&out, 8 * 1024));
    CHK(hipMalloc(&cyc, 64));
    long long h[6];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_lds, dim3(1), dim3(512), 0, 0, out, cyc);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h, cyc, 48, hipMemcpyDeviceToHost));
    }
    const char* nm[6] = {"uniform ds_read_b128", "per-lane ds_read_b128, 32 B stride", "per-lane ds_read_b128, 48 B stride",
                         "per-lane ds_read_b64, consecutive", "uniform ds_read_b64", "shuffle of a double (2 ds_bpermute)"};
    const int per[6] = {8, 8, 8, 8, 8, 4};
    for (int k = 0; k < 6; ++k)
        printf("%-40s %7.1f cycles per wave instruction (8 waves issuing: %.1f LDS cycles each)\n", nm[k], (double)h[k] / IT / per[k],
               (double)h[k] / IT / per[k] / 8.0);
    printf("ubench8 done\n");
    return 0;
}
