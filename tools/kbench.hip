// kbench.hip — isolated launches of the engine's kernels at the N = 4096 shapes; run under
// `rocprofv3 --kernel-trace` and read the per-(kernel, grid) durations with tools/kstats.py.
// build: make -C tools kbench   (links the engine's object files)
#include "../limbo_amd/csrc/dev.h"
#include "trace_stub.h"
#include "../include/gpe.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

thread_local BatchLaunch g_batch; // dev.h: single-GP launches
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const int64_t N = 4096, ld = N + 32;
    const int D = 6, reps = argc > 1 ? atoi(argv[1]) : 10;
    hipStream_t s;
    CHK(hipStreamCreate(&s));
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0, 1);
    std::vector<double> hXt((size_t)ld * D);
    for (auto& v : hXt) v = U(rng);
    double *dXt, *A0, *A, *Xi, *w, *out;
    int* info;
    CHK(hipMalloc(&dXt, sizeof(double) * ld * D));
    CHK(hipMalloc(&A0, sizeof(double) * ld * N));
    CHK(hipMalloc(&A, sizeof(double) * ld * N));
    CHK(hipMalloc(&Xi, sizeof(double) * 64 * 4096));
    CHK(hipMemset(Xi, 0, sizeof(double) * 64 * 4096)); // (only block 0's inverse is ever computed here: the sweeps must not meet NaNs)
    CHK(hipMalloc(&w, sizeof(double) * ld));
    CHK(hipMalloc(&out, sizeof(double) * ld));
    CHK(hipMalloc(&info, 64));
    CHK(hipMemset(info, 0, 64));
    CHK(hipMemcpy(dXt, hXt.data(), sizeof(double) * ld * D, hipMemcpyHostToDevice));
    KParams kp{};
    kp.kind = GPE_KERNEL_SE_ARD;
    kp.D = D;
    kp.sf2 = 1.0;
    kp.inv_l = 1.0;
    kp.noise = 0.01;
    kp.diag_add = 0.01 + 1e-8;
    for (int d = 0; d < D; ++d) kp.inv_ell[d] = 1.0;
    launch_build_K(s, dXt, ld, N, kp, A0, ld);
    CHK(hipMemcpyAsync(A, A0, sizeof(double) * ld * N, hipMemcpyDeviceToDevice, s));
    CHK(hipMemsetAsync(w, 0, sizeof(double) * ld, s));
    for (int r = 0; r < reps; ++r) {
        // diagonal block
        launch_copy2d(s, A0, ld, A, ld, 64, 64);
        launch_diag(s, A, ld, 64, Xi, info, 0, 1);
        // fused panel step below the block just factored (3 more blocks in the panel)
        {
            static double* Hs = nullptr;
            if (!Hs) { CHK(hipMalloc(&Hs, sizeof(double) * 8 * 4096)); CHK(hipMemset(Hs, 0, sizeof(double) * 8 * 4096)); }
            launch_panel_step(s, A, ld, 0, N + 1, 3, Xi, Xi + 4096, 1, info, Hs, -1, -1, 0, nullptr, (gpe_epoch_t*)(Hs + 7 * 4096));
            launch_copy2d(s, A0, ld, A, ld, N, 320); // restore what the step consumed
        }
        // trsm shape: (4032 x 64) x (64 x 64), in place
        {
            GemmArgs g{};
            g.C = A + 64; g.ldc = ld; g.A = A + 64; g.lda = ld; g.B = Xi; g.ldb = 64; g.b_kmajor = 1;
            g.m = 4032; g.n = 64; g.k = 64; g.overwrite = 1; g.tile = 32;
            launch_gemm_sub(s, g);
        }
        // in-panel update: (4032 x 192), k = 64
        {
            GemmArgs g{};
            g.C = A + 64 + 64 * ld; g.ldc = ld; g.A = A + 64; g.lda = ld; g.B = A + 64; g.ldb = ld;
            g.m = 4032; g.n = 192; g.k = 64; g.tri = 1; g.grow0 = 64; g.gcol0 = 64;
            for (int t : {32, 64}) { g.tile = t; launch_gemm_sub(s, g); }
        }
        // outer updates, k = 256
        for (int64_t m : {3840, 2816, 1792, 768}) {
            GemmArgs g{};
            int64_t pe = N - m;
            g.C = A + pe + pe * ld; g.ldc = ld; g.A = A + pe; g.lda = ld; g.B = A + pe; g.ldb = ld;
            g.m = m; g.n = m; g.k = 256; g.tri = 1; g.grow0 = pe; g.gcol0 = pe;
            for (int t : {32, 64, 128}) { g.tile = t; launch_gemm_sub(s, g); }
        }
        // next-panel update (critical path): rows >= pe of the next 256 columns, k = 256
        for (int64_t pe : {256, 1536, 2816}) {
            GemmArgs g{};
            g.C = A + pe + pe * ld; g.ldc = ld; g.A = A + pe; g.lda = ld; g.B = A + pe; g.ldb = ld;
            g.m = N - pe; g.n = 256; g.k = 256; g.tri = 1; g.grow0 = pe; g.gcol0 = pe;
            for (int t : {32, 64, 128}) { g.tile = t; launch_gemm_sub(s, g); }
        }
        // backward step at the last block and in the middle
        launch_trsv_sweep(s, A, ld, N, Xi, w, out, ld, 1, 1);
    }
    CHK(hipStreamSynchronize(s));
#ifdef FLOW_TIMING
    {
        int* err;
        CHK(hipMalloc(&err, 8));
        CHK(hipMemset(err, 0, 8));
        double *a2, *yv;
        CHK(hipMalloc(&a2, sizeof(double) * ld));
        CHK(hipMalloc(&yv, sizeof(double) * ld));
        CHK(hipMemset(yv, 0, sizeof(double) * ld));
        for (int rep = 0; rep < 3; ++rep) { // the last (warm) repetition is the one reported
            launch_trsv_bwd_flow(s, A0, ld, N, Xi, yv, 1, ld, a2, ld, 1, err, 0, nullptr, 0, nullptr, 0);
            CHK(hipStreamSynchronize(s));
        }
        extern void dump_flow_timing(int);
        dump_flow_timing((int)(N / 64));
    }
#endif
#ifdef S2_TIMING
    { // the backward sweep whose hop is one matrix-vector product (sweep2.hip), in-kernel stamps of every block's workgroup
        int* err;
        CHK(hipMalloc(&err, 8));
        CHK(hipMemset(err, 0, 8));
        double *a2, *yv;
        CHK(hipMalloc(&a2, sizeof(double) * ld));
        CHK(hipMalloc(&yv, sizeof(double) * ld));
        CHK(hipMemset(yv, 0, sizeof(double) * ld));
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) { // the last (warm) repetition is the one reported
            CHK(hipMemsetAsync(a2, 0xFF, sizeof(double) * N, s));
            hipEventRecord(e0, s);
            launch_trsv_bwd_m(s, A0, ld, N, Xi, yv, 1, a2, err, 1, nullptr, nullptr);
            hipEventRecord(e1, s);
            CHK(hipStreamSynchronize(s));
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("k_trsv_bwd_m, N = %lld: %.2f us (events)\n", (long long)N, 1e3 * ms);
        }
        extern void dump_s2_timing(int);
        dump_s2_timing((int)(N / 64));
    }
#endif
#ifdef GEMM_TIMING
    extern void dump_gemm_timing();
    dump_gemm_timing();
#endif
#ifdef DIAG_TIMING
    extern void dump_diag_timing();
    dump_diag_timing();
    extern void dump_panel_timing();
    dump_panel_timing();
    { // the four steps of an outer panel one by one (nt = blocks of the panel still to come), full 64-row blocks only
        static double* Hs2 = nullptr;
        CHK(hipMalloc(&Hs2, sizeof(double) * 8 * 4096));
        CHK(hipMemset(Hs2, 0, sizeof(double) * 8 * 4096));
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int nt = 3; nt >= 0; --nt) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                launch_copy2d(s, A0, ld, A, ld, N, 320);
                launch_diag(s, A, ld, 64, Xi, info, 0, 1);
                CHK(hipStreamSynchronize(s));
                hipEventRecord(e0, s);
                launch_panel_step(s, A, ld, 0, N, nt, Xi, Xi + 4096, nt > 0, info, Hs2, -1, -1, 0, nullptr, (gpe_epoch_t*)(Hs2 + 7 * 4096));
                hipEventRecord(e1, s);
                CHK(hipStreamSynchronize(s));
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("step with nt = %d: %.2f us (events)\n", nt, 1e3 * best);
            dump_panel_timing();
        }
    }
#endif
#ifdef DIAG_TIMING
    { // all steps of the first outer panel in one launch (k_panel256)
        static double* Hs3 = nullptr;
        CHK(hipMalloc(&Hs3, sizeof(double) * 34 * 4096));
        CHK(hipMemset(Hs3, 0, sizeof(double) * 34 * 4096));
        CHK(hipMemset(Hs3 + 14 * 4096, 0xFF, sizeof(double) * 20 * 4096)); // the polled hand-over buffers (two of them)
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            launch_copy2d(s, A0, ld, A, ld, N, 320);
            launch_diag(s, A, ld, 64, Xi, info, 0, 1);
            CHK(hipStreamSynchronize(s));
            hipEventRecord(e0, s);
            launch_panel256(s, A, ld, 0, N, Xi, info, 256, Hs3 + 12 * 4096, Hs3 + (14 + 10 * (rep & 1)) * 4096,
                            Hs3 + (14 + 10 * ((rep + 1) & 1)) * 4096);
            hipEventRecord(e1, s);
            CHK(hipStreamSynchronize(s));
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("k_panel256, first panel of N = 4096: %.2f us (events)\n", 1e3 * best);
        extern void dump_p256_timing();
        dump_p256_timing();
    }
#endif
#ifdef DIAG_TIMING
    { // the first 1024 x 1024 block of K as one tiled data-flow launch (k_tail)
        const int64_t T = getenv("KB_TAIL_T") ? atoll(getenv("KB_TAIL_T")) : 1024; // (2816: the closing launch's shape)
        const int64_t need = tail_buf_doubles(T / 64, T / 64 + 1);
        double* tb;
        CHK(hipMalloc(&tb, sizeof(double) * 2 * need));
        CHK(hipMemset(tb, 0xFF, sizeof(double) * 2 * need));
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            launch_copy2d(s, A0, ld, A, ld, N, T);
            CHK(hipStreamSynchronize(s));
            hipEventRecord(e0, s);
            launch_tail(s, A, ld, 0, T, T, T + 1, Xi, info, tb + (rep & 1) * need, tb + ((rep + 1) & 1) * need);
            hipEventRecord(e1, s);
            CHK(hipStreamSynchronize(s));
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("k_tail, the leading %lld x %lld block of K (%lld tile columns + one right-hand-side row): %.2f us (events)\n", (long long)T, (long long)T, (long long)(T / 64), 1e3 * best);
        extern void dump_tail_timing(int);
        dump_tail_timing((int)(T / 64));
    }
#endif
    printf("kbench done\n");
    return 0;
}
