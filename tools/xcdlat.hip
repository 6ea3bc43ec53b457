// xcdlat.hip — one-way latency of a polled 8-byte hand-over between two workgroups, by where they run (same XCD: workgroup
// ids 0 and 8 of a 1-D grid; different XCDs: 0 and 1) and by the scope of the store and of the polling load.
// make -C tools xcdlat && tools/xcdlat
#include <hip/hip_runtime.h>
#include <cstdio>
#define ROUNDS 2000
template <int ST, int LD> // 0: workgroup scope (sc0), 1: agent scope (sc1), 2: system
__global__ void k_pp(unsigned long long* slots, long long* cyc, unsigned* xcc, int other)
{
    const int me = blockIdx.x;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[me] = id;
    }
    if (me != 0 && me != other)
        return;
    unsigned long long* mine = slots + (me == 0 ? 0 : 64);   // I write here
    unsigned long long* theirs = slots + (me == 0 ? 64 : 0); // and poll there
    const long long t0 = wall_clock64();
    bool dead = false;
    for (unsigned long long i = 1; i <= ROUNDS && !dead; ++i) {
        if (me == 0) {
            if (threadIdx.x == 0)
                __hip_atomic_store(mine, i, __ATOMIC_RELAXED, ST == 0 ? __HIP_MEMORY_SCOPE_WORKGROUP : ST == 1 ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_SYSTEM);
            long spins = 0;
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, LD == 0 ? __HIP_MEMORY_SCOPE_WORKGROUP : LD == 1 ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_SYSTEM) < i && ++spins < 100000) {
            }
        }
        else {
            long spins = 0;
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, LD == 0 ? __HIP_MEMORY_SCOPE_WORKGROUP : LD == 1 ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_SYSTEM) < i && ++spins < 100000) {
            }
            if (threadIdx.x == 0)
                __hip_atomic_store(mine, i, __ATOMIC_RELAXED, ST == 0 ? __HIP_MEMORY_SCOPE_WORKGROUP : ST == 1 ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0 && me == 0)
        cyc[0] = t1 - t0;
}
template <int ST, int LD>
static void run(const char* tag, unsigned long long* slots, long long* cyc, unsigned* xcc)
{
    for (int other : {1, 8, 16, 9}) {
        hipMemset(slots, 0, 8 * 128);
        hipLaunchKernelGGL((k_pp<ST, LD>), dim3(32), dim3(64), 0, 0, slots, cyc, xcc, other);
        hipDeviceSynchronize();
        long long h;
        unsigned x[32];
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        hipMemcpy(x, xcc, sizeof(x), hipMemcpyDeviceToHost);
        unsigned long long last[2];
        hipMemcpy(&last[0], slots, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&last[1], slots + 64, 8, hipMemcpyDeviceToHost);
        printf("%-34s workgroups 0 (XCC %u) <-> %2d (XCC %u): %7.1f ns one way%s\n", tag, x[0], other, x[other], h * 10.0 / (2.0 * ROUNDS),
               (last[0] == ROUNDS && last[1] == ROUNDS) ? "" : "   (TIMED OUT: a side never saw the other's store)");
        fflush(stdout);
    }
}
int main()
{
    unsigned long long* slots;
    long long* cyc;
    unsigned* xcc;
    hipMalloc(&slots, 8 * 128);
    hipMalloc(&cyc, 64);
    hipMalloc(&xcc, 4 * 32);
    run<1, 1>("store agent, load agent", slots, cyc, xcc);
    run<1, 0>("store agent, load workgroup", slots, cyc, xcc);
    run<0, 0>("store workgroup, load workgroup", slots, cyc, xcc);
    run<2, 2>("store system, load system", slots, cyc, xcc);
    return 0;
}
