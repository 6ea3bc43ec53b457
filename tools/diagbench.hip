// diagbench.hip — the 64 x 64 diagonal-block kernel (k_diag) alone: average launch time and the in-kernel round stamps.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDIAG_TIMING [-DDIAG8=0] tools/diagbench.hip -o tools/diagbench_[48]
#include "../limbo_amd/csrc/potrf.hip"
#include <vector>
thread_local BatchLaunch g_batch;
void launch_gemm_sub(hipStream_t, const GemmArgs&) {}
int main()
{
    const int n = 64, ld = 80;
    std::vector<double> K(ld * n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j)
            K[i + j * ld] = exp(-0.5 * (i - j) * (i - j) / 100.0) + (i == j ? 0.01 : 0.0);
    double *A, *A0, *Xt;
    int* info;
    hipMalloc(&A, sizeof(double) * ld * n);
    hipMalloc(&A0, sizeof(double) * ld * n);
    hipMalloc(&Xt, sizeof(double) * 4096);
    hipMalloc(&info, 64);
    hipMemset(info, 0, 64);
    hipMemcpy(A0, K.data(), sizeof(double) * ld * n, hipMemcpyHostToDevice);
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 200;
    float tot = 0.f;
    for (int r = 0; r < reps + 5; ++r) {
        hipMemcpyAsync(A, A0, sizeof(double) * ld * n, hipMemcpyDeviceToDevice, s);
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_diag, dim3(1), dim3(320), 0, s, A, (int64_t)ld, Xt, info, (int64_t)0);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r >= 5)
            tot += ms;
    }
    std::vector<double> L(ld * n);
    hipMemcpy(L.data(), A, sizeof(double) * ld * n, hipMemcpyDeviceToHost);
    double err = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double sum = 0.0;
            for (int k = 0; k <= j; ++k)
                sum += L[i + k * ld] * L[j + k * ld];
            err = fmax(err, fabs(sum - K[i + j * ld]));
        }
    long long h[32];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_diag_ts), sizeof(h));
    printf("DIAG8=%d: k_diag %.2f us per launch (events, incl. launch gap), max |L L^T - K| = %.2e\n", DIAG8, 1e3 * tot / reps, err);
    printf("  cycles (clock64 = s_memtime): load %lld | rounds total %lld | store %lld | per round:", (h[1] - h[0]),
           (h[2] - h[1]), (h[3] - h[2]));
    const int nr = DIAG8 != 0 ? 8 : 16;
    for (int g = 0; g < nr; ++g)
        printf(" %lld", (h[10 + g] - (g ? h[9 + g] : h[1])));
    printf("\n");
    long long arr[16][8];
    hipMemcpyFromSymbol(arr, HIP_SYMBOL(g_diag_arr), sizeof(arr));
    printf("  arrival at the closing barrier, cycles after the round began (waves 0-3 factor, 4 = inversion; * = owner):\n");
    for (int g = 0; g < nr; ++g) {
        const long long t0 = g ? h[9 + g] : h[1];
        printf("    round %2d:", g);
        for (int w = 0; w < 5; ++w)
            printf(" %5lld%s", arr[g][w] - t0, w == (g & 3) ? "*" : " ");
        printf("\n");
    }
    return 0;
}
