// cumask.hip — does a CU mask on a stream (hipExtStreamCreateWithCUMask) confine a launch to a set of XCDs on this box, and
// which mask bits belong to which XCD?  (two evaluation chains on disjoint halves of the chip.)
// Every workgroup records HW_REG_XCC_ID and HW_REG_HW_ID; the host prints, per mask pattern, workgroups per XCD and the number
// of distinct (XCD, SE, SH, CU) places seen.   make -C tools cumask && tools/cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>

__global__ void k_where(unsigned* out, int spin)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = xcc & 15;
        out[2 * blockIdx.x + 1] = hw;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { // keep the workgroup resident for a while so that the launch spreads
    }
}

static void run(const char* tag, const std::vector<uint32_t>& mask, bool use_mask)
{
    hipStream_t s;
    hipError_t e = use_mask ? hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) : hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
        printf("%-40s stream creation failed: %s\n", tag, hipGetErrorString(e));
        return;
    }
    const int G = 2048;
    unsigned* d;
    hipMalloc(&d, sizeof(unsigned) * 2 * G);
    hipMemset(d, 0xFF, sizeof(unsigned) * 2 * G);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a, s);
    hipLaunchKernelGGL(k_where, dim3(G), dim3(256), 0, s, d, 2000);
    hipEventRecord(b, s);
    hipStreamSynchronize(s);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> h(2 * G);
    hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * G, hipMemcpyDeviceToHost);
    int per[16] = {0};
    std::set<unsigned long long> places;
    for (int i = 0; i < G; ++i) {
        per[h[2 * i] & 15]++;
        const unsigned hw = h[2 * i + 1];
        places.insert(((unsigned long long)(h[2 * i] & 15) << 32) | (hw & 0xFF00)); // CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    }
    printf("%-40s %6.1f us  places %3zu  per XCD:", tag, ms * 1000.0f, places.size());
    for (int x = 0; x < 8; ++x)
        printf(" %4d", per[x]);
    printf("\n");
    fflush(stdout);
    hipFree(d);
    hipStreamDestroy(s);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device: %s, %d CUs\n", p.name, p.multiProcessorCount);
    const int words = (p.multiProcessorCount + 31) / 32;
    auto make = [&](auto pred) {
        std::vector<uint32_t> m(words, 0);
        for (int i = 0; i < p.multiProcessorCount; ++i)
            if (pred(i))
                m[i / 32] |= 1u << (i % 32);
        return m;
    };
    run("no mask", {}, false);
    run("all bits", make([](int) { return true; }), true);
    run("bits 0..127", make([](int i) { return i < 128; }), true);
    run("bits 128..255", make([](int i) { return i >= 128; }), true);
    run("i % 8 < 4", make([](int i) { return i % 8 < 4; }), true);
    run("i % 8 >= 4", make([](int i) { return i % 8 >= 4; }), true);
    run("even bits", make([](int i) { return i % 2 == 0; }), true);
    run("(i / 32) % 2 == 0", make([](int i) { return (i / 32) % 2 == 0; }), true);
    run("i % 8 == 0", make([](int i) { return i % 8 == 0; }), true);
    run("bits 0..31", make([](int i) { return i < 32; }), true);
    return 0;
}
