// updbench.hip — the 15 trailing updates of an N = 4096 factorisation (k = 256, lower triangle, one right-hand-side row
// under the matrix), each launch alone between two HIP events as bench.py's `roofline` times them: the tile-per-workgroup
// kernels of gemm.hip against the persistent stream-k kernel of gemm_sk.hip (variants / workgroup counts), with a
// correctness check of every variant against the first.   build: make -C tools updbench
#include "../limbo_amd/csrc/dev.h"
#include "trace_stub.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#include <string>

thread_local BatchLaunch g_batch;
bool gemm_sk_ok(const GemmArgs& g);
size_t gemm_sk_workspace_bytes();
void launch_gemm_sk_ex(hipStream_t s, const GemmArgs& g0, int rhs_rows, void* ws, int* err, int variant, int G_req, int smin_req, int wA, int wB);
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_null() {}
struct Cfg { const char* name; int kind; int variant, G, smin; int norhs = 0; int wA = 1, wB = 1; }; // kind 0: gemm.hip, 1: gemm_sk.hip

int main(int argc, char** argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 4096;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const int P = 1, nbo = 256;
    const int64_t ld = N + 32;
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<double> h((size_t)ld * N);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1, 1);
    for (auto& v : h) v = U(rng) / 16.0; // 256 x^2 ~ x: the chained updates neither blow up nor vanish
    double *A0, *A, *Aref;
    CHK(hipMalloc(&A0, sizeof(double) * ld * N));
    CHK(hipMalloc(&A, sizeof(double) * ld * N));
    CHK(hipMalloc(&Aref, sizeof(double) * ld * N));
    CHK(hipMemcpy(A0, h.data(), sizeof(double) * ld * N, hipMemcpyHostToDevice));
    void* ws;
    CHK(hipMalloc(&ws, gemm_sk_workspace_bytes()));
    CHK(hipMemset(ws, 0, gemm_sk_workspace_bytes()));
    int* err;
    CHK(hipHostMalloc(&err, 64, hipHostMallocMapped));
    memset(err, 0, 64);
    std::vector<Cfg> cfgs = {{"tiles (gemm.hip)", 0, 0, 0, 0}, {"tiles, rhs as FMAs", 2, 0, 0, 0}, {"sk v1 G256", 1, 1, 256, 0}, {"sk v2 G512 1:1", 1, 2, 512, 0}};
    const int nl = (int)(N / nbo) - 1;
    std::vector<std::vector<double>> us(cfgs.size(), std::vector<double>(nl, 1e30));
    std::vector<double> whole(cfgs.size(), 1e30);
    std::vector<hipEvent_t> ev(2 * nl + 2);
    for (auto& e : ev) CHK(hipEventCreate(&e));
    auto make = [&](int l, double* base) {
        const int64_t p0 = (int64_t)l * nbo, pe = p0 + nbo;
        GemmArgs g{};
        g.C = base + pe + pe * ld; g.ldc = ld;
        g.A = base + pe + p0 * ld; g.lda = ld;
        g.B = base + pe + p0 * ld; g.ldb = ld;
        g.m = N - pe + P; g.n = N - pe; g.k = nbo; g.tri = 1; g.grow0 = pe; g.gcol0 = pe;
        return g;
    };
    auto run = [&](const Cfg& c, const GemmArgs& g) {
        if (c.kind == 0) launch_gemm_sub(s, g);
        else if (c.kind == 2) { GemmArgs q = g; q.rhs_rows = P; launch_gemm_sub(s, q); }
        else if (c.norhs) { GemmArgs q = g; q.m -= P; launch_gemm_sk_ex(s, q, 0, ws, err, c.variant, c.G, c.smin, c.wA, c.wB); }
        else launch_gemm_sk_ex(s, g, P, ws, err, c.variant, c.G, c.smin, c.wA, c.wB);
    };
    // correctness: every variant against the first, launch by launch, each from the same start A0 (chaining the 15 updates
    // of random data overflows); elements on/below the diagonal of the updated block + the rhs row; and bitwise
    // reproducibility of every variant
    std::vector<double> href((size_t)ld * N), hv((size_t)ld * N), hv2((size_t)ld * N);
    std::vector<double> worst(cfgs.size(), 0.0);
    std::vector<size_t> ndiff(cfgs.size(), 0);
    double big = 0.0;
    for (int l = 0; l < nl; l += (l < 3 || l > nl - 5 ? 1 : 2)) {
        const int64_t pe = (int64_t)(l + 1) * nbo;
        for (size_t ci = 0; ci < cfgs.size(); ++ci) {
            for (int rep = 0; rep < (ci == 0 ? 1 : 2); ++rep) {
                CHK(hipMemcpyAsync(A, A0, sizeof(double) * ld * N, hipMemcpyDeviceToDevice, s));
                run(cfgs[ci], make(l, A));
                CHK(hipStreamSynchronize(s));
                CHK(hipMemcpy2D((ci == 0 ? href : rep == 0 ? hv : hv2).data() + pe * ld, sizeof(double) * ld, A + pe * ld, sizeof(double) * ld,
                                sizeof(double) * ld, N - pe, hipMemcpyDeviceToHost));
            }
            if (ci == 0)
                continue;
            for (int64_t j = pe; j < N; ++j)
                for (int64_t i = j; i < N + (cfgs[ci].norhs ? 0 : P); ++i) {
                    const double d = fabs(hv[i + j * ld] - href[i + j * ld]);
                    big = fmax(big, fabs(href[i + j * ld]));
                    worst[ci] = fmax(worst[ci], d);
                    ndiff[ci] += hv2[i + j * ld] != hv[i + j * ld];
                }
        }
    }
    for (size_t ci = 1; ci < cfgs.size(); ++ci)
        printf("check %-20s max |diff| vs tiles = %.3e (max |C| %.3e); a second run differs in %zu elements; err word %d\n", cfgs[ci].name,
               worst[ci], big, ndiff[ci], err[0]);
    // timing: every launch alone between two events; and the 15 launches back to back between two events
    for (int r = 0; r < reps + 1; ++r)
        for (size_t ci = 0; ci < cfgs.size(); ++ci) {
            for (int l = 0; l < nl; ++l) {
                CHK(hipEventRecord(ev[2 * l], s));
                run(cfgs[ci], make(l, A));
                CHK(hipEventRecord(ev[2 * l + 1], s));
            }
            CHK(hipStreamSynchronize(s));
            if (r > 0)
                for (int l = 0; l < nl; ++l) {
                    float ms; CHK(hipEventElapsedTime(&ms, ev[2 * l], ev[2 * l + 1]));
                    us[ci][l] = fmin(us[ci][l], 1e3 * ms);
                }
            CHK(hipEventRecord(ev[2 * nl], s));
            for (int l = 0; l < nl; ++l) run(cfgs[ci], make(l, A));
            CHK(hipEventRecord(ev[2 * nl + 1], s));
            CHK(hipStreamSynchronize(s));
            float ms; CHK(hipEventElapsedTime(&ms, ev[2 * nl], ev[2 * nl + 1]));
            if (r > 0) whole[ci] = fmin(whole[ci], 1e3 * ms);
        }
#ifdef SK_TIMING
    {
        extern void dump_sk_timing(int, const char*);
        extern void clear_sk_timing();
        for (int l : {0, 4})
            for (int v : {2, 3}) {
                clear_sk_timing();
                GemmArgs q = make(l, A);
                q.m -= P;
                launch_gemm_sk_ex(s, q, 0, ws, err, 2, 512, 0, v == 2 ? 4 : 3, v == 2 ? 3 : 2);
                CHK(hipStreamSynchronize(s));
                char nm[64];
                snprintf(nm, sizeof nm, "launch %d variant 2, weights %s (no rhs rows)", l + 1, v == 2 ? "4:3" : "3:2");
                dump_sk_timing(512, nm);
            }
    }
#endif
    // the floor of the measurement: an empty kernel between two events
    {
        float best = 1e9f;
        for (int r = 0; r < 10; ++r) {
            CHK(hipEventRecord(ev[0], s));
            hipLaunchKernelGGL(k_null, dim3(256), dim3(512), 0, s);
            CHK(hipEventRecord(ev[1], s));
            CHK(hipStreamSynchronize(s));
            float ms; CHK(hipEventElapsedTime(&ms, ev[0], ev[1]));
            best = ms < best ? ms : best;
        }
        printf("empty kernel (256 x 512 threads) between two events: %.1f us\n", 1e3 * best);
    }
    printf("\nN = %lld, k = %d, P = %d; us per launch (best of %d), each launch alone between two events\n%-4s %-6s", (long long)N, nbo, P, reps, "l", "n");
    for (auto& c : cfgs) printf(" %18s", c.name);
    printf("\n");
    double flops = 0.0;
    for (int l = 0; l < nl; ++l) {
        GemmArgs g = make(l, A);
        flops += gemm_flops(g);
        printf("%-4d %-6lld", l + 1, (long long)g.n);
        for (size_t ci = 0; ci < cfgs.size(); ++ci) printf(" %18.1f", us[ci][l]);
        printf("\n");
    }
    printf("%-11s", "sum us");
    for (size_t ci = 0; ci < cfgs.size(); ++ci) { double t = 0; for (double v : us[ci]) t += v; printf(" %18.1f", t); }
    printf("\n%-11s", "frac 78.6");
    for (size_t ci = 0; ci < cfgs.size(); ++ci) { double t = 0; for (double v : us[ci]) t += v; printf(" %18.3f", flops / (t * 1e-6) / 78.6e12); }
    printf("\n%-11s", "b2b us");
    for (size_t ci = 0; ci < cfgs.size(); ++ci) printf(" %18.1f", whole[ci]);
    printf("\nalgorithmic flops of the %d launches: %.4e; err word %d\n", nl, flops, err[0]);
    return 0;
}
