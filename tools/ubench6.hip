// ubench6.hip — does straight-line code that is executed ONCE run slower than the same instructions in
// a loop?  (Instruction fetch: the unrolled diagonal-block factorisation is ~70 KB of code executed once.)
// Kernel A: NI dependent v_fma_f64 in straight-line code; kernel B: the same count as a loop of 64.
// One workgroup of 5 waves (as k_diag); cycles per instruction from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int N>
struct Rep {
    static __device__ __forceinline__ void run(double& x, double& y, double c)
    {
        Rep<N / 2>::run(x, y, c);
        Rep<N - N / 2>::run(x, y, c);
    }
};
template <>
struct Rep<1> {
    static __device__ __forceinline__ void run(double& x, double& y, double c)
    {
        asm volatile("v_fma_f64 %0, %0, %2, %1\n\tv_fma_f64 %1, %1, %2, %0" : "+v"(x), "+v"(y) : "v"(c));
    }
};

template <int NI>
__global__ void k_straight(double* out, long long* cyc, double c)
{
    double x = threadIdx.x, y = 1.0;
    long long t0 = clock64();
    Rep<NI / 2>::run(x, y, c);
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
    out[threadIdx.x] = x + y;
}
template <int NI>
__global__ void k_loop(double* out, long long* cyc, double c)
{
    double x = threadIdx.x, y = 1.0;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < NI / 128; ++i)
        Rep<64>::run(x, y, c);
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
    out[threadIdx.x] = x + y;
}

int main()
{
    double* out;
    long long* cyc;
    CHK(hipMalloc(&out, 8 * 1024));
    CHK(hipMalloc(&cyc, 64));
    long long h;
#define RUN(K, NI, name)                                                                   \
    for (int rep = 0; rep < 3; ++rep) {                                                    \
        hipLaunchKernelGGL((K<NI>), dim3(1), dim3(320), 0, 0, out, cyc, 1.0000001);        \
        CHK(hipDeviceSynchronize());                                                       \
        CHK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));                                 \
        printf("%-28s %6d instr (%3d KB of code): %8lld cycles = %.2f cycles/instr (launch %d)\n", name, NI, NI * 8 / 1024, h, \
               (double)h / NI, rep);                                                       \
    }
    RUN(k_loop, 8192, "loop of 128, x64");
    RUN(k_straight, 1024, "straight line");
    RUN(k_straight, 4096, "straight line");
    RUN(k_straight, 8192, "straight line");
    RUN(k_straight, 16384, "straight line");
    printf("ubench6 done\n");
    return 0;
}
