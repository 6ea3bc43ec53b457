// gapbench.hip — the gap between two dependent kernels of one stream (end of one -> start of the next, device wall clock), launched
// one by one and as a captured graph.   hipcc -O2 --offload-arch=gfx950 tools/gapbench.hip -o tools/gapbench && tools/gapbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_spin(long long* ts, int idx, long long ticks, double* sink, int nwrite)
{
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0)
        ts[2 * idx] = t0;
    // some dirty lines for the end-of-kernel write-back to find
    for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < nwrite; i += gridDim.x * blockDim.x)
        sink[i] = (double)i + (double)t0;
    while (wall_clock64() - t0 < ticks) {
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        ts[2 * idx + 1] = wall_clock64();
}
int main()
{
    const int K = 10;
    long long* ts;
    double* sink;
    CHK(hipMalloc(&ts, sizeof(long long) * 2 * K));
    const int nwrite = 4 << 20; // 32 MB
    CHK(hipMalloc(&sink, sizeof(double) * nwrite));
    hipStream_t s;
    CHK(hipStreamCreate(&s));
    std::vector<long long> h(2 * K);
    for (int grid : {256, 1024}) {
        for (int nw : {0, nwrite}) {
            for (int rep = 0; rep < 3; ++rep) {
                for (int i = 0; i < K; ++i)
                    hipLaunchKernelGGL(k_spin, dim3(grid), dim3(512), 0, s, ts, i, 3000LL, sink, nw); // 30 us each
                CHK(hipStreamSynchronize(s));
            }
            CHK(hipMemcpy(h.data(), ts, sizeof(long long) * 2 * K, hipMemcpyDeviceToHost));
            double g = 0;
            for (int i = 1; i < K; ++i)
                g += (h[2 * i] - h[2 * i - 1]) * 0.01;
            printf("grid %4d, %2d MB written per kernel: stream launches  mean gap %.2f us\n", grid, nw / (1 << 17), g / (K - 1));
            hipGraph_t graph;
            hipGraphExec_t ge;
            CHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < K; ++i)
                hipLaunchKernelGGL(k_spin, dim3(grid), dim3(512), 0, s, ts, i, 3000LL, sink, nw);
            CHK(hipStreamEndCapture(s, &graph));
            CHK(hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
            for (int rep = 0; rep < 3; ++rep) {
                CHK(hipGraphLaunch(ge, s));
                CHK(hipStreamSynchronize(s));
            }
            CHK(hipMemcpy(h.data(), ts, sizeof(long long) * 2 * K, hipMemcpyDeviceToHost));
            g = 0;
            for (int i = 1; i < K; ++i)
                g += (h[2 * i] - h[2 * i - 1]) * 0.01;
            printf("grid %4d, %2d MB written per kernel: graph launch     mean gap %.2f us\n", grid, nw / (1 << 17), g / (K - 1));
            hipGraphExecDestroy(ge);
            hipGraphDestroy(graph);
        }
    }
    return 0;
}
