// diagbench2.hip — the 128 x 128 pair block (diag_flow2.h) alone: launch time and the in-kernel round stamps of its waves.
// build: make -C tools diagflow2
#include "../limbo_amd/csrc/potrf.hip"
#include "trace_stub.h"
#include <vector>
thread_local BatchLaunch g_batch;
void launch_gemm_sub(hipStream_t, const GemmArgs&) {}
__global__ __launch_bounds__(512) void k_diag2(double* __restrict__ A, int64_t lda, double* __restrict__ Xt, int* __restrict__ info,
                                               double* __restrict__ LPtm, double* __restrict__ S)
{
    __shared__ __attribute__((aligned(16))) double lds[TAIL_LDS_DOUBLES];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* L1 = lds;
    double* L2 = lds + D2_OFF_L2;
    for (int e = threadIdx.x; e < 128 * 64; e += 512) {
        const int r = e & 127, c = e >> 7;
        L1[r * XS + c] = A[r + (int64_t)c * lda];
    }
    for (int e = threadIdx.x; e < 64 * 64; e += 512) {
        const int r = e & 63, c = e >> 6;
        L2[r * XS + c] = A[64 + r + (int64_t)(64 + c) * lda];
    }
    diag_flow_init(reinterpret_cast<DiagSync*>(lds + D2_OFF_SY));
    __syncthreads();
    Diag2Out o;
    o.lda = lda;
    o.Ad0 = A;
    o.Atm = A + 64;
    o.Ad1 = A + 64 + 64 * lda;
    o.LPtm = LPtm;
    o.SL21_0 = S + 1024;
    o.SL21_1 = S + 3072 + 1024;
    diag_flow2(lds, o, Xt, Xt + 4096, info, 0, wave, lane, S, S + 3072);
}
int main()
{
    const int n = 128, ld = 144;
    std::vector<double> K(ld * n, nan(""));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j)
            K[i + j * ld] = exp(-0.5 * (i - j) * (i - j) / 100.0) + (i == j ? 0.01 : 0.0);
    double *A, *A0, *Xt, *LP, *S;
    int* info;
    hipMalloc(&A, sizeof(double) * ld * n);
    hipMalloc(&A0, sizeof(double) * ld * n);
    hipMalloc(&Xt, sizeof(double) * 8192);
    hipMalloc(&LP, sizeof(double) * 4096);
    hipMalloc(&S, sizeof(double) * 6144);
    hipMalloc(&info, 64);
    hipMemset(info, 0, 64);
    hipMemcpy(A0, K.data(), sizeof(double) * ld * n, hipMemcpyHostToDevice);
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 100;
    float tot = 0.f;
    for (int r = 0; r < reps + 5; ++r) {
        hipMemcpyAsync(A, A0, sizeof(double) * ld * n, hipMemcpyDeviceToDevice, s);
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_diag2, dim3(1), dim3(512), 0, s, A, (int64_t)ld, Xt, info, LP, S);
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
    std::vector<double> X(8192);
    hipMemcpy(X.data(), Xt, sizeof(double) * 8192, hipMemcpyDeviceToHost);
    double xerr = 0.0;
    for (int b = 0; b < 2; ++b)
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) { // (X L)[i][j], X[i][k] = Xt[k + 64 i]
                double sum = 0.0;
                for (int k = j; k <= i; ++k)
                    sum += X[4096 * b + k + 64 * i] * L[(64 * b + k) + (64 * b + j) * ld];
                xerr = fmax(xerr, fabs(sum - (i == j ? 1.0 : 0.0)));
            }
    printf("k_diag2: %.2f us per launch (events, incl. launch gap), max |L L^T - K| = %.2e, max |X L - I| = %.2e\n", 1e3 * tot / reps, err, xerr);
    long long h[8][33];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_flow2_ts), sizeof(h));
    const long long t0 = h[0][32];
    printf("round: end of the round in cycles after P's start: P | U0 U1 U2 U3 | S   (P's round time)\n");
    for (int G = 0; G < 32; ++G)
        printf("  %2d: %6lld | %6lld %6lld %6lld %6lld | %6lld   P round %lld\n", G, h[0][G] - t0, G < 30 ? h[1][G] - t0 : 0, G < 30 ? h[2][G] - t0 : 0,
               G < 30 ? h[3][G] - t0 : 0, G < 30 ? h[4][G] - t0 : 0, h[6][G] - t0, h[0][G] - (G ? h[0][G - 1] : t0));
    printf("X wave done at %lld\n", h[5][31] - t0);
    return 0;
}
