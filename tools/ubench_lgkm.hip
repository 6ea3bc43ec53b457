// ubench_lgkm.hip — does "s_waitcnt lgkmcnt(n)" after 2n LDS reads guarantee the first n have landed?  (in-order return of
// ds_read_b128, also with two distinct addresses per wave.)  It does: 0 stale copies in every variant on MI355X — the wrong
// results that prompted this test were the compiler copying an in-flight register.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -w tools/ubench_lgkm.hip -o tools/ubench_lgkm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d_t __attribute__((ext_vector_type(2)));
template <int NPEND, int SPLIT>
__global__ void k(double* out, int* bad)
{
    __shared__ __attribute__((aligned(16))) double L[64 * 66];
    for (int e = threadIdx.x; e < 64 * 66; e += 64)
        L[e] = e;
    __syncthreads();
    const int h = SPLIT ? (threadIdx.x >> 5) : 0;
    const unsigned lbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) double*)(L + 2 * h);
    int nb = 0;
    for (int rep = 0; rep < 200; ++rep) {
        v2d_t q[7], p[7], c[7];
#pragma unroll
        for (int i = 0; i < 7; ++i)
            q[i] = p[i] = v2d_t{-1.0, -1.0};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 7; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(lbase), "n"((4 + i) * 66 * 8));
#pragma unroll
        for (int i = 0; i < 7; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(p[i]) : "v"(lbase), "n"((11 + i) * 66 * 8));
        asm volatile("s_waitcnt lgkmcnt(%7)"
                     : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6])
                     : "n"(NPEND));
#pragma unroll
        for (int i = 0; i < 7; ++i) { // use right after the partial wait
            c[i] = q[i] + v2d_t{0.0, 0.0};
            asm volatile("" : "+v"(c[i]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]));
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const double want = (4 + i) * 66 + 2 * h + 1;
            if (c[i][1] != want)
                ++nb;
        }
        out[threadIdx.x] = p[0][0];
    }
    atomicAdd(bad, nb);
}
int main()
{
    double* out; int* bad;
    hipMalloc(&out, 64 * 8); hipMalloc(&bad, 4);
    for (int v = 0; v < 4; ++v) {
        hipMemset(bad, 0, 4);
        if (v == 0) hipLaunchKernelGGL((k<7, 0>), dim3(1), dim3(64), 0, 0, out, bad);
        if (v == 1) hipLaunchKernelGGL((k<7, 1>), dim3(1), dim3(64), 0, 0, out, bad);
        if (v == 2) hipLaunchKernelGGL((k<0, 1>), dim3(1), dim3(64), 0, 0, out, bad);
        if (v == 3) hipLaunchKernelGGL((k<6, 1>), dim3(1), dim3(64), 0, 0, out, bad);
        int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("variant %d (pending allowed %s, split %d): stale copies = %d\n", v, v==0||v==1?"7":v==2?"0":"6", v != 0, hb);
    }
    return 0;
}
