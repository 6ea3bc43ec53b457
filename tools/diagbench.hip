// diagbench.hip — the 64 x 64 diagonal-block kernel (k_diag) alone: average launch time and the in-kernel round stamps.
// build: make -C tools diagflow diagbench_0   (the data-flow form / the barrier rounds, -DDIAG_FLOW=0)
#include "../limbo_amd/csrc/potrf.hip"
#include "trace_stub.h"
#include <vector>
thread_local BatchLaunch g_batch;
void launch_gemm_sub(hipStream_t, const GemmArgs&) {}
int main()
{
    const int n = 64, ld = 80;
    std::vector<double> K(ld * n, nan("")); // everything outside the lower triangle is never meant to be read
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j)
            K[i + j * ld] = exp(-0.5 * (i - j) * (i - j) / 100.0) + (i == j ? 0.01 : 0.0);
    if (FILE* f = fopen("tools/tmp/K64.bin", "rb")) { // an optional 64 x 64 row-major test matrix
        std::vector<double> M(4096);
        if (fread(M.data(), 8, 4096, f) == 4096)
            for (int i = 0; i < n; ++i)
                for (int j = 0; j <= i; ++j)
                    K[i + j * ld] = M[i * 64 + j];
        fclose(f);
        printf("matrix from tools/tmp/K64.bin\n");
    }
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
        hipLaunchKernelGGL(k_diag, dim3(1), dim3(DIAG_THREADS), 0, s, A, (int64_t)ld, Xt, info, (int64_t)0);
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
    { // the half-block inverses: X11 L11 = I, X22 L22 = I  (Xt[col + 64 row] = X[row][col])
        std::vector<double> X(4096);
        hipMemcpy(X.data(), Xt, sizeof(double) * 4096, hipMemcpyDeviceToHost);
        double xe = 0.0;
        for (int hb = 0; hb < 2; ++hb)
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j <= i; ++j) {
                    double sum = 0.0;
                    for (int k = j; k <= i; ++k)
                        sum += X[(32 * hb + k) + 64 * (32 * hb + i)] * L[(32 * hb + k) + (32 * hb + j) * ld];
                    const double er = fabs(sum - (i == j ? 1.0 : 0.0));
                    if (er > 1e-10 && xe <= 1e-10)
                        printf("first bad (X L) entry: half %d row %d col %d err %.3e\n", hb, i, j, er);
                    xe = fmax(xe, er);
                }
        printf("max |X_hh L_hh - I| = %.2e\n", xe);
#if DIAG_FLOW
        double xf = 0.0; // the data-flow form also leaves the off-diagonal quarter: all of X L = I
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j <= i; ++j) {
                double sum = 0.0;
                for (int k = j; k <= i; ++k)
                    sum += X[k + 64 * i] * L[k + j * ld];
                xf = fmax(xf, fabs(sum - (i == j ? 1.0 : 0.0)));
            }
        printf("max |X L - I| (all 64 x 64) = %.2e\n", xf);
#endif
    }
#if DIAG_FLOW
    {
        long long f[8][20];
        hipMemcpyFromSymbol(f, HIP_SYMBOL(g_flow_ts), sizeof(f));
        printf("DIAG_FLOW: k_diag %.2f us per launch (events, incl. launch gap), max |L L^T - K| = %.2e\n", 1e3 * tot / reps, err);
        printf("  cycles: load %lld | P total %lld\n  round: end of the round in cycles after the start, per wave (P | U0 U1 U2 U3 | X | S) and P's round time\n", h[1] - h[0], f[0][15] - h[1]);
        for (int g = 0; g < 16; ++g) {
            printf("    %2d: %6lld |", g, f[0][g] - h[1]);
            for (int w = 1; w <= 4; ++w)
                printf(" %6lld", g < 14 ? f[w][g] - h[1] : 0LL);
            printf(" | %6lld | %6lld   P round %lld\n", f[5][g] - h[1], f[6][g] - h[1], f[0][g] - (g ? f[0][g - 1] : h[1]));
        }
        return 0;
    }
#endif
    printf("barrier rounds: k_diag %.2f us per launch (events, incl. launch gap), max |L L^T - K| = %.2e\n", 1e3 * tot / reps, err);
    printf("  cycles (clock64 = s_memtime): load %lld | rounds total %lld | store %lld | per round:", (h[1] - h[0]),
           (h[2] - h[1]), (h[3] - h[2]));
    const int nr = 16;
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
