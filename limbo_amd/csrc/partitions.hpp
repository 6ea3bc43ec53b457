// partitions.hpp — who may launch what when: the gate that orders data-flow launches on a device, the two CU-masked chain partitions, and the inter-process lock.
// A part of engine.hip's translation unit (included there, once, at the place its contents used to stand: they share the
// file-local types and helpers of the engine — gpe_ctx, PhaseScope, DevGuard ...); split out in round 6 for readability.
#pragma once

// ---- one data-flow launch at a time per device (dev.h: FlowGate) ----
namespace {
struct GateDev {
    std::recursive_mutex mu;
    int depth = 0, next = 0;
    hipEvent_t ring[64] = {};
    hipStream_t last_stream = nullptr; // where the device's last UNMASKED data-flow launch went
    // Round 5: two CU-masked stream pairs, half of every XCD's CUs each (mask bit i = XCD i % 8, CU i / 8 of it — measured,
    // profiles/r05_cumask_probe.log; a mask cannot leave an XCD empty, so "four XCDs each" is not to be had).  A data-flow
    // launch confined to its half always finds its lowest unfinished workgroup resident there — per XCD the dispatcher hands
    // a launch's workgroups out in order, and nothing else that WAITS can hold those CUs — so one chain per half runs
    // deadlock-free beside the other.  An unmasked data-flow launch can hold any CU: it waits for both halves to drain, and
    // the masked chains that follow wait for it.
    hipStream_t part[2] = {nullptr, nullptr}, part_aux[2] = {nullptr, nullptr};
    bool part_tried = false, part_dirty[2] = {false, false};
    unsigned rr = 0;
    unsigned char* d_owner = nullptr; // device: owner[xcd * 256 + place] = the half (0 / 1) that (XCD, CU) place belongs to, 255 unknown
    int* h_violation = nullptr;       // pinned: set by a workgroup of a masked chain that found itself in the OTHER half
    std::chrono::steady_clock::time_point last_busy{}; // when a chain last found another one in flight (ChainScope)
    bool ever_busy = false;
    // Round 6: other PROCESSES on the same GPU.  The gate above orders the data-flow launches of this process by stream events;
    // two processes have no events in common, and two data-flow launches resident together starve each other exactly as two
    // streams did (bounded polls, full re-runs: 1 evaluation/s).  Two files per GPU under /dev/shm, named by its PCI bus id:
    //   .users  every process that has a handle on the GPU write-locks ONE byte of it for its lifetime (POSIX record lock: the
    //           kernel drops it when the process ends, however it ends); F_GETLK over the whole range answers "is anybody
    //           else here?" in one system call (a process's own locks never conflict with itself);
    //   .lock   flock(LOCK_EX) around a data-flow launch (or an evaluation's whole chain) AND the host wait for it, taken only
    //           while somebody else is here: data-flow launches of different processes then never overlap on the device.
    // A process that is alone pays one fcntl per launch scope and never touches the lock.
    int xp_users = -1, xp_lock = -1, xp_byte = -1;
    bool xp_tried = false, xp_held = false;
    std::chrono::steady_clock::time_point xp_attach{};
    bool xp_crowded_at_attach = false;
};
GateDev g_gate[16];
std::atomic<int> g_live[16]; // live handles per physical device (gpe_create / gpe_destroy)
std::atomic<long long> g_xproc_waits{0}; // data-flow scopes that ran under the inter-process lock (gpe_xproc_waits)
bool gate_on()
{
    static const bool on = !(getenv("GPE_FLOW_GATE") && atoi(getenv("GPE_FLOW_GATE")) == 0);
    return on;
}
GateDev& gate_dev()
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    return g_gate[dev & 15];
}
} // namespace
namespace {
bool xproc_on()
{
    static const bool on = !(getenv("GPE_XPROC_LOCK") && atoi(getenv("GPE_XPROC_LOCK")) == 0);
    return on;
}
// is another process holding a byte of the users file?
bool xproc_others(GateDev& g)
{
    if (g.xp_users < 0)
        return false;
    struct flock fl {};
    fl.l_type = F_WRLCK;
    fl.l_whence = SEEK_SET;
    fl.l_start = 0;
    fl.l_len = 4096;
    return fcntl(g.xp_users, F_GETLK, &fl) == 0 && fl.l_type != F_UNLCK;
}
// once per process and device (under g.mu): open the two files, take a byte of the users file
void xproc_attach(GateDev& g)
{
    if (g.xp_tried)
        return;
    g.xp_tried = true;
    if (!xproc_on())
        return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), dev) != hipSuccess || !bus[0])
        return;
    for (char* p = bus; *p; ++p)
        if (*p == ':' || *p == '/')
            *p = '_';
    const char* dirs[2] = {"/dev/shm", "/tmp"};
    for (const char* d : dirs) {
        const std::string base = std::string(d) + "/limbo_amd.gpu-" + bus;
        const mode_t um = umask(0);
        const int fu = open((base + ".users").c_str(), O_RDWR | O_CREAT | O_CLOEXEC, 0666);
        const int fl = open((base + ".lock").c_str(), O_RDWR | O_CREAT | O_CLOEXEC, 0666);
        umask(um);
        if (fu >= 0 && fl >= 0) {
            g.xp_users = fu;
            g.xp_lock = fl;
            return;
        }
        if (fu >= 0)
            close(fu);
        if (fl >= 0)
            close(fl);
        g.xp_users = -1;
    }
}
// the process's first live handle on the device appears / its last one goes (gpe_create, gpe_destroy; under g.mu): a byte of
// the users file is held exactly while the process can have work on the GPU
void xproc_show(GateDev& g)
{
    xproc_attach(g);
    if (g.xp_users < 0 || g.xp_byte >= 0)
        return;
    g.xp_crowded_at_attach = xproc_others(g); // (before this process shows up in the file itself)
    for (int k = 0; k < 4096 && g.xp_byte < 0; ++k) { // a byte of my own, starting from my pid's
        struct flock fk {};
        fk.l_type = F_WRLCK;
        fk.l_whence = SEEK_SET;
        fk.l_start = (getpid() + k) % 4096;
        fk.l_len = 1;
        if (fcntl(g.xp_users, F_SETLK, &fk) == 0)
            g.xp_byte = (int)fk.l_start;
    }
    g.xp_attach = std::chrono::steady_clock::now();
}
void xproc_hide(GateDev& g)
{
    if (g.xp_users < 0 || g.xp_byte < 0)
        return;
    struct flock fk {};
    fk.l_type = F_UNLCK;
    fk.l_whence = SEEK_SET;
    fk.l_start = g.xp_byte;
    fk.l_len = 1;
    (void)fcntl(g.xp_users, F_SETLK, &fk);
    g.xp_byte = -1;
}
// outermost data-flow scope opens (under g.mu): take the inter-process lock while anybody else is on the GPU
void xproc_enter(GateDev& g)
{
    if (g.xp_lock < 0 || g.xp_byte < 0 || !xproc_others(g))
        return;
    // somebody who was here before me may have a launch in flight that it started believing it was alone: not before 5 ms
    // after I showed up in the users file (an evaluation is ~1 ms; it sees me from its next launch on)
    if (g.xp_crowded_at_attach) {
        const auto ready = g.xp_attach + std::chrono::milliseconds(5);
        if (std::chrono::steady_clock::now() < ready)
            std::this_thread::sleep_until(ready);
        g.xp_crowded_at_attach = false;
    }
    while (flock(g.xp_lock, LOCK_EX) != 0 && errno == EINTR) {
    }
    g.xp_held = true;
    g_xproc_waits.fetch_add(1, std::memory_order_relaxed);
    static std::atomic<bool> said{false};
    if (!said.exchange(true))
        fprintf(stderr, "limbo_amd: another process is using this GPU: data-flow launches take turns through %s (GPE_XPROC_LOCK=0 to disable)\n",
                "/dev/shm/limbo_amd.gpu-*.lock");
}
// ... closes: what was enqueued must be THROUGH on the device before the next process may start its own
void xproc_leave(GateDev& g, hipStream_t s, hipStream_t s2 = nullptr)
{
    if (!g.xp_held)
        return;
    (void)hipStreamSynchronize(s);
    if (s2)
        (void)hipStreamSynchronize(s2);
    g.xp_held = false;
    (void)flock(g.xp_lock, LOCK_UN);
}
} // namespace
// a stream is about to be destroyed: nobody may record on it afterwards
void flow_gate_forget(hipStream_t s)
{
    GateDev& g = gate_dev();
    std::lock_guard<std::recursive_mutex> lk(g.mu);
    if (g.last_stream == s)
        g.last_stream = nullptr; // (gpe_destroy synchronises the stream first: its launches are through)
}
// `s` waits for whatever is on `behind` now (an event at that stream's current end)
static void gate_order(GateDev& g, hipStream_t s, hipStream_t behind)
{
    hipEvent_t& e = g.ring[g.next];
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
        e = nullptr;
    if (e && hipEventRecord(e, behind) == hipSuccess) {
        (void)hipStreamWaitEvent(s, e, 0);
        g.next = (g.next + 1) % 64;
    }
}
// an unmasked data-flow launch on s: behind the device's previous one on another stream, and behind both masked halves
static void gate_unmasked(GateDev& g, hipStream_t s)
{
    // the previous data-flow launch of the device went to another stream: an event at that stream's current end (behind
    // that launch; nothing is recorded per launch — a batch of 64 members steps through ~60 gated launches on one stream)
    if (g.last_stream && g.last_stream != s)
        gate_order(g, s, g.last_stream);
}
// An unmasked data-flow launch can hold any CU: both masked halves must have drained before it.  A HOST wait (not an event the
// stream waits for: ChainScope's destructor says why) — and NOT under the gate's mutex (one thread's query beside
// threads running masked chains used to stall every other thread's enqueue for a whole chain): called with g.mu held ONCE by
// this thread (depth as it was before this scope), returns with it held again and both halves clean.
static void gate_drain_halves(GateDev& g)
{
    for (;;) {
        hipStream_t w[2];
        int nw = 0;
        for (int i = 0; i < 2; ++i)
            if (g.part_dirty[i]) {
                if (hipStreamQuery(g.part[i]) != hipErrorNotReady)
                    g.part_dirty[i] = false;
                else
                    w[nw++] = g.part[i];
            }
        if (nw == 0)
            return;
        g.mu.unlock();
        for (int k = 0; k < nw; ++k)
            (void)hipStreamSynchronize(w[k]);
        g.mu.lock(); // (others may have dirtied a half again meanwhile: look again)
    }
}
void flow_gate_enter(hipStream_t s)
{
    if (!gate_on())
        return;
    GateDev& g = gate_dev();
    g.mu.lock(); // (held until flow_gate_leave: the launch in between is a few microseconds of host time)
    if (g.depth == 0)
        gate_drain_halves(g);
    if (g.depth++ == 0) {
        xproc_enter(g);
        gate_unmasked(g, s);
    }
}
static bool partitions_on()
{
    static const bool on = !(getenv("GPE_FLOW_PARTITIONS") && atoi(getenv("GPE_FLOW_PARTITIONS")) == 0);
    return on && !g_partitions_broken.load(std::memory_order_relaxed);
}
// where a workgroup runs: XCD and (shader engine, array, CU) of it
__global__ void k_partition_probe(unsigned* __restrict__ out, int spin)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x] = ((xcc & 15u) << 16) | (hw & 0xFF00u); // CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { // stay resident for a moment so that the launch spreads over every CU it may use
    }
}
// do the two masked streams really confine their launches to disjoint halves of every XCD, here, in this process?
// (owner, optional: the half every place seen belongs to)
static bool partition_masks_hold(hipStream_t a, hipStream_t b, unsigned char* owner = nullptr)
{
    constexpr int G = 2048;
    unsigned* d = nullptr;
    if (hipMalloc(&d, sizeof(unsigned) * 2 * G) != hipSuccess)
        return false;
    hipLaunchKernelGGL(k_partition_probe, dim3(G), dim3(64), 0, a, d, 300);
    hipLaunchKernelGGL(k_partition_probe, dim3(G), dim3(64), 0, b, d + G, 300);
    std::vector<unsigned> h(2 * G);
    bool ok = hipStreamSynchronize(a) == hipSuccess && hipStreamSynchronize(b) == hipSuccess
        && hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * G, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok)
        return false;
    std::vector<unsigned> pa(h.begin(), h.begin() + G), pb(h.begin() + G, h.end());
    std::sort(pa.begin(), pa.end());
    pa.erase(std::unique(pa.begin(), pa.end()), pa.end());
    std::sort(pb.begin(), pb.end());
    pb.erase(std::unique(pb.begin(), pb.end()), pb.end());
    std::vector<unsigned> both;
    std::set_intersection(pa.begin(), pa.end(), pb.begin(), pb.end(), std::back_inserter(both));
    unsigned xa = 0, xb = 0; // XCDs seen
    for (unsigned v : pa)
        xa |= 1u << (v >> 16);
    for (unsigned v : pb)
        xb |= 1u << (v >> 16);
    if (owner) {
        for (unsigned v : pa)
            owner[(v >> 16) * 256 + ((v >> 8) & 255)] = 0;
        for (unsigned v : pb)
            owner[(v >> 16) * 256 + ((v >> 8) & 255)] = 1;
    }
    return both.empty() && !pa.empty() && !pb.empty() && pa.size() <= 128 && pb.size() <= 128 && xa == 0xFFu && xb == 0xFFu;
}
// at the head of every masked chain: 64 single-wave workgroups look where they are; one that sits in the other half's CUs says so
__global__ void k_partition_check(const unsigned char* __restrict__ owner, int half, int* __restrict__ violation)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const unsigned char o = owner[(xcc & 15u) * 256 + ((hw >> 8) & 255u)];
        if (o != 255 && o != (unsigned char)half)
            *violation = 1;
    }
}
static bool partition_streams(GateDev& g)
{
    if (!g.part_tried) {
        g.part_tried = true;
        int dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus == 256) {
            uint32_t lo[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0}; // CUs 0..15 of every XCD
            uint32_t hi[8] = {0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}; // CUs 16..31
            bool ok = hipExtStreamCreateWithCUMask(&g.part[0], 8, lo) == hipSuccess && hipExtStreamCreateWithCUMask(&g.part_aux[0], 8, lo) == hipSuccess
                && hipExtStreamCreateWithCUMask(&g.part[1], 8, hi) == hipSuccess && hipExtStreamCreateWithCUMask(&g.part_aux[1], 8, hi) == hipSuccess;
            auto drop_streams = [&g] { // (a partial or rejected set of masked streams is destroyed, not leaked)
                for (hipStream_t* st : {&g.part[0], &g.part_aux[0], &g.part[1], &g.part_aux[1]}) {
                    if (*st)
                        (void)hipStreamDestroy(*st);
                    *st = nullptr;
                }
            };
            if (!ok)
                drop_streams();
            else {
                // the runtime creates a stream's hardware queue at its FIRST launch (tens of milliseconds for a masked one):
                // here, not inside the first evaluation that meets another one
                void* word = nullptr;
                if (hipMalloc(&word, 64) == hipSuccess) {
                    for (hipStream_t st : {g.part[0], g.part_aux[0], g.part[1], g.part_aux[1]}) {
                        (void)hipMemsetAsync(word, 0, 64, st);
                        (void)hipStreamSynchronize(st);
                    }
                    (void)hipFree(word);
                }
                // ... and the assumption everything rests on is CHECKED, in this process, on these streams: launches on the two
                // halves land on disjoint sets of at most 128 (XCD, CU) places, all eight XCDs each.  If not: no partitions.
                std::vector<unsigned char> owner(16 * 256, 255);
                if (!partition_masks_hold(g.part[0], g.part[1], owner.data()) || !partition_masks_hold(g.part_aux[0], g.part_aux[1])
                    || hipMalloc(&g.d_owner, owner.size()) != hipSuccess
                    || hipMemcpy(g.d_owner, owner.data(), owner.size(), hipMemcpyHostToDevice) != hipSuccess
                    || hipHostMalloc(&g.h_violation, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
                    fprintf(stderr, "limbo_amd: the CU masks of the chain partitions are not honoured here: one chain at a time\n");
                    drop_streams();
                }
                else
                    *g.h_violation = 0;
            }
        }
    }
    return g.part[0] != nullptr;
}
ChainScope::ChainScope(gpe_ctx* c_, bool engage, bool may_partition) : c(c_), on(engage && gate_on())
{
    if (!on)
        return;
    GateDev& g = gate_dev();
    g.mu.lock(); // (held for the enqueue of the evaluation: ~50 us of host time)
    ++g.depth;   // the gates of the launches inside nest in this one
    if (g.depth == 1)
        xproc_enter(g); // (another PROCESS on the GPU: this chain runs under the inter-process lock, on the whole chip)
    if (g.h_violation && *g.h_violation) { // a masked chain saw one of its workgroups in the other half's CUs
        *g.h_violation = 0;
        g_masked_chains.fetch_add(1);
        partitions_give_up("a CU mask was not honoured");
    }
    if (g.depth == 1 && may_partition && !g.xp_held && partitions_on() && !c->prof && partition_streams(g)) {
        // is another chain in flight on the device?  (A query, not a guarantee: it picks the mode; ORDER comes from the
        // events below.)
        const bool full_busy = g.last_stream && g.last_stream != c->stream && hipStreamQuery(g.last_stream) == hipErrorNotReady;
        bool busy[2];
        for (int i = 0; i < 2; ++i)
            busy[i] = g.part_dirty[i] && hipStreamQuery(g.part[i]) == hipErrorNotReady;
        // Hysteresis: with R threads in flight a chain now and then finds the device idle for a moment (the others are between
        // evaluations on the host); were it to take the whole chip, both halves would have to drain for it and the next
        // masked chains to wait behind it — measured: 701 evaluations/s with four threads instead of 930.  So the device
        // stays in two halves for 3 ms after a chain last found another one in flight (a caller that alternates handles from
        // ONE thread never finds a chain in flight: always the whole chip; so does whoever comes 3 ms after the threads).
        const auto now = std::chrono::steady_clock::now();
        if (full_busy || busy[0] || busy[1]) {
            g.last_busy = now;
            g.ever_busy = true;
        }
        if (g.ever_busy && now - g.last_busy < std::chrono::milliseconds(3))
            part = !busy[0] ? 0 : (!busy[1] ? 1 : (int)(g.rr++ & 1));
    }
    if (part >= 0) {
        hipStream_t P = g.part[part];
        gate_order(g, P, c->stream); // behind the handle's own earlier work (uploads, the previous evaluation's readers)
        if (g.last_stream && g.last_stream != c->stream)
            gate_order(g, P, g.last_stream); // behind the device's last unmasked data-flow launch
        own = c->stream;
        own2 = c->stream2;
        c->stream = P;
        c->stream2 = g.part_aux[part];
        g.part_dirty[part] = true;
        g_masked_chains.fetch_add(1, std::memory_order_relaxed);
        static const bool fault = getenv("GPE_PARTITION_FAULT") && atoi(getenv("GPE_PARTITION_FAULT")) != 0; // (test hook: claims the other half)
        hipLaunchKernelGGL(k_partition_check, dim3(64), dim3(64), 0, P, g.d_owner, fault ? 1 - part : part, g.h_violation);
    }
    else if (g.depth == 1) {
        --g.depth; // (the mutex is released while the halves drain: the scope is not open yet)
        gate_drain_halves(g);
        ++g.depth;
        gate_unmasked(g, c->stream);
    }
}
ChainScope::~ChainScope()
{
    if (!on)
        return;
    GateDev& g = gate_dev();
    if (part >= 0) {
        hipStream_t P = c->stream;
        c->stream = own;
        c->stream2 = own2;
        // Whatever the handle does next comes behind the chain — by a HOST wait (wait_chain, in compute_finish), not by making
        // the handle's own stream wait for an event of the masked one: own streams are high-priority queues (create_main_stream),
        // masked ones cannot be (hipExtStreamCreateWithCUMask takes no priority), and a high-priority queue that sits on a
        // barrier packet keeps the scheduler from the normal-priority queue it is waiting for once the process has more
        // hardware queues than the chip maps at a time — measured with GPU_MAX_HW_QUEUES=8, torch in the process and eight
        // threads: masked chains got no service for seconds, their bounded polls fired (3 evaluations/s, re-runs); with the
        // own streams at the default priority, or with this, 950 evaluations/s.
        if (!c->chain_ev && hipEventCreateWithFlags(&c->chain_ev, hipEventDisableTiming) != hipSuccess)
            c->chain_ev = nullptr;
        if (c->chain_ev && hipEventRecord(c->chain_ev, P) == hipSuccess)
            c->chain_pending = true;
        else
            (void)hipStreamSynchronize(P);
    }
    else if (g.depth == 1)
        g.last_stream = c->stream;
    if (g.depth == 1)
        xproc_leave(g, c->stream, c->stream2);
    --g.depth;
    g.mu.unlock();
}
void flow_gate_leave(hipStream_t s)
{
    if (!gate_on())
        return;
    GateDev& g = gate_dev();
    if (--g.depth == 0) {
        g.last_stream = s;
        xproc_leave(g, s);
    }
    g.mu.unlock();
}
