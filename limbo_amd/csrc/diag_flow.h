// diag_flow.h — the 64 x 64 diagonal block as a data-flow of specialised waves (round 2; included by potrf.hip).
//
// The round-1 kernel (DiagRound, potrf.hip) rotates the "owner" role over four factor waves that all execute one barrier per
// four columns; its stamps (profiles/r02_diag_rounds.log) show three chains of similar length between consecutive barriers
// (the owner's pivot chain with two v_readlane hops per column, the previous owner's catch-up + regular update, the inversion
// wave), so that shortening one alone never moved a round.  Here every chain is shortened at once and nobody waits at a
// barrier:
//   wave 0      P  the PANEL wave, lane = row.  Per four columns: broadcast the 4 x 4 pivot block once (ten values), factor it
//                  redundantly in every lane (wave-uniform arithmetic: one v_rsq_f64 + 5 dependent operations per column, no
//                  cross-lane hop inside the group), solve the 64 x 4 column block against it per lane, publish the four
//                  columns into Ls and bump `prog`; then apply this panel to the NEXT group itself (16 FMAs) — the only
//                  update on the critical path.  It owns no other columns.
//   waves 1-4   U  update waves, one 16-column block column each, kept in the v_mfma_f64_4x4x4 accumulator layout: a rank-4
//                  update of a 16 x 16 block is four matrix-core instructions and five per-lane LDS reads.  After applying
//                  panel t a wave hands group t+2 (now updated through t) to P through the buffer H[(t+2)&1] + `hflag`.
//   wave 5      X  the inversion pipeline of round 1 (XPipe32: the two 32 x 32 half-block inverses by forward substitution, one
//                  round behind) on all 64 lanes, following `prog` instead of the barriers.
//   wave 6      S  stores the finished columns of L to global memory and checks the pivots — none of that is in P.
//   wave 7      W  W = L21 X11 once X11 is complete; the update waves then finish the inverse, X21 = -X22 W, when X22 is:
//                  all of X = L11^-1 leaves the kernel and the panel steps solve against L11 with one product.
// A lost hardware assumption would show as wrong numbers, not as a hang: every wait is for a counter another wave of the
// same workgroup is certain to advance (P waits only for the update waves' hand-overs, they only for P).
// Synchronisation is by counters in LDS: a publisher issues its data writes, then the counter write (LDS operations of one
// wave execute in order); a consumer reads the counter, then the data.  No s_barrier inside the rounds: P never stalls on a
// slower consumer, X and S may lag by several rounds and only have to finish shortly after P does.
#pragma once

struct DiagSync {
    int prog;     // number of 4-column panels P has published into Ls (and their inverse pivots into invd)
    int hflag[2]; // hflag[g & 1] == g: H[g & 1] holds group g with the panels 0..g-2 applied
    int xprog;    // rounds the inversion wave has finished (8: X11 complete in Xw, 16: X22 too)
    int wdone;    // W = L21 X11 is in Xw (diag_flow2.h: of block 0 -> 1, of block 1 -> 2)
    int pad[3];
};
#define DIAG_H_DOUBLES (2 * NB * 4)
// the inverse's LDS work area Xw: X11 | X22 | W = L21 X11, 32 x 32 each, row-major with stride XH
#define XH 34
#define DIAG_XW_DOUBLES (3 * 32 * XH)
// X leaves with device-scope (write-through) stores: k_panel256's other workgroups read it DURING the launch, possibly behind
// another XCD's L2 (potrf.hip); for the launches that hand X to the next launch it makes no difference
#define DIAG_XT_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// Optional early hand-over of the inverse to ANOTHER workgroup of the same launch (k_panel256's next factoring strip):
//   f_early <- epoch once X11 (Xt) and L21 (L21s[c + 32 k] = L[32 + c][k]) are out and acknowledged — half-way through the block;
//   f_x22   <- epoch once X22 (Xt) is — ~0.4 us before the update waves finish X21 and well before a barrier could say so.
// With these the consumer solves in the half-block form (Y1 = T1 X11^T, T2 -= Y1 L21^T early; Y2 = T2 X22^T at the end) and
// three quarters of its solve overlap with this factorisation.
typedef unsigned long long diag_epoch_t;
struct DiagEarly {
    bool mute; // test hook: nothing leaves
    // Polled copies, 1024 doubles each, [col + 32 row]: S[0 .. 1023] = X11, S[1024 ..] = L21 (row = row of L21 = row 32 + . of
    // the block), S[2048 ..] = X22.  They hold an all-ones pattern no arithmetic produces until the inversion wave (X11, X22:
    // four rows per round) and the store wave (L21: four columns per round) put the values there; consumers poll the values
    // themselves (an aligned 8-byte store is never torn and needs no ordering with anything else — the hand-off of the one-launch
    // sweeps, solve.hip).  No acknowledgement is awaited on this side and no flag travels behind the data: X11 and L21 are
    // visible one store latency after round 9 of 16, the last rows of X22 one store latency after they are computed.
    double* S;
};

static __device__ __forceinline__ int lds_peek(const int* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void lds_post(int* p, int v)
{
    asm volatile("" ::: "memory"); // the data writes stay in front of the counter write (the hardware keeps a wave's LDS operations in order)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void lds_await(const int* p, int want)
{
    while (lds_peek(p) < want) {
    }
    asm volatile("" ::: "memory"); // data reads stay behind the counter read
}

#ifdef DIAG_TIMING
__device__ long long g_flow_ts[8][20]; // [wave][round]: when the wave finished its round
#ifdef DIAG_NO_STAMPS
#define FTS(w, G) do { } while (0)
#else
#define FTS(w, G) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_flow_ts[w][G] = clock64(); } while (0)
#endif
#else
#define FTS(w, G) do { } while (0)
#endif

// wave-uniform 1/sqrt(p): v_rsq_f64 seed (~1e-8) + one Newton step folded in (relative error ~1.5e-16)
static __device__ __forceinline__ double rsq_newton(double p)
{
    const double y0 = __builtin_amdgcn_rsq(p);
    const double t = (0.5 * p) * y0;
    const double eh = fma(-t, y0, 0.5);
    return fma(y0, eh, y0);
}

#ifndef FLOWP_LATE_CHECK
#define FLOWP_LATE_CHECK 1 // 1: P waits for a missing hand-over at the END of its round, where it is needed, instead of half-way through (the hand-over loop takes ~600 cycles: 15.7 k -> 14.6 k cycles per block, tools/diagbench); 0: round 2
#endif
#ifndef FLOWP_LDSMUL
#define FLOWP_LDSMUL 1 // P's look-ahead multipliers of x0..x2 through LDS (1) or all by v_readlane (0)
#endif
// ---- P ---------------------------------------------------------------------------------------------------------------------
template <int G>
struct FlowP {
    // c: group G's four columns, complete (all earlier panels applied)
    static __device__ __forceinline__ void run(double (&c)[4], double (&n)[4], double* Ls, const double* H, double* invd,
                                               DiagSync* sy, int r)
    {
        FlowP<G - 1>::run(c, n, Ls, H, invd, sy, r);
        constexpr int c0 = 4 * G;
        constexpr bool fetch = G >= 1 && G < 15; // groups 0 and 1 were read from Ls at the start
        // the next group as the update waves left it, read ahead of time and checked further down: its counter first
        const double* h = H + ((G + 1) & 1) * (NB * 4) + r * 4;
        int seen = 0;
        if (fetch) {
            seen = lds_peek(&sy->hflag[(G + 1) & 1]);
            asm volatile("" ::: "memory");
            n[0] = h[0];
            n[1] = h[1];
            n[2] = h[2];
            n[3] = h[3];
        }
        // the 4 x 4 pivot block, once, for every lane
        const double b00 = bcast_lane(c[0], c0), b10 = bcast_lane(c[0], c0 + 1), b20 = bcast_lane(c[0], c0 + 2),
                     b30 = bcast_lane(c[0], c0 + 3);
        const double b11 = bcast_lane(c[1], c0 + 1), b21 = bcast_lane(c[1], c0 + 2), b31 = bcast_lane(c[1], c0 + 3);
        const double b22 = bcast_lane(c[2], c0 + 2), b32 = bcast_lane(c[2], c0 + 3);
        const double b33 = bcast_lane(c[3], c0 + 3);
        // its Cholesky factor in wave-uniform arithmetic; the column block follows lane by lane as the factors appear
        const double y0 = rsq_newton(b00);
        const double l10 = b10 * y0, l20 = b20 * y0, l30 = b30 * y0;
        const double x0 = c[0] * y0;
        const double y1 = rsq_newton(fma(-l10, l10, b11));
        const double l21 = fma(-l20, l10, b21) * y1, l31 = fma(-l30, l10, b31) * y1;
        const double x1 = fma(-x0, l10, c[1]) * y1;
        double* dst = Ls + r * XS + c0;
#if FLOWP_LDSMUL
        dst[0] = x0; // out as soon as they exist: the multipliers of the look-ahead update come back through LDS (below)
        dst[1] = x1;
#endif
#if !FLOWP_LATE_CHECK
        if (fetch && seen < G + 1) { // rare: the hand-over was not there yet
            lds_await(&sy->hflag[(G + 1) & 1], G + 1);
            n[0] = h[0];
            n[1] = h[1];
            n[2] = h[2];
            n[3] = h[3];
        }
#endif
        // from here on the update of the next group by this panel (multipliers = rows c0+4 .. c0+7 of the new columns)
        // fills the gaps of the pivot chain
        const double y2 = rsq_newton(fma(-l21, l21, fma(-l20, l20, b22)));
        const double l32 = fma(-l31, l21, fma(-l30, l20, b32)) * y2;
        const double x2 = fma(-x1, l21, fma(-x0, l20, c[2])) * y2;
#if FLOWP_LDSMUL
        // The look-ahead update needs x_j[row c0+4+e] in every lane.  For x0, x1, x2 those twelve values come back from LDS
        // (this wave's own writes, in order; wave-uniform reads: 8 instructions) while the chain computes y3 and x3 —
        // 24 v_readlane + their s_nop less in the wave that bounds the block; only x3's four are v_readlane broadcasts.
        dst[2] = x2;
        double m0[4], m1[4], m2[4];
        if (G < 15) {
            const double* mrow = Ls + (c0 + 4) * XS + c0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m0[e] = mrow[e * XS + 0];
                m1[e] = mrow[e * XS + 1];
                m2[e] = mrow[e * XS + 2];
            }
        }
#endif
        const double y3 = rsq_newton(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, b33))));
        const double x3 = fma(-x2, l32, fma(-x1, l31, fma(-x0, l30, c[3]))) * y3;
        // publish: columns into Ls (rows above the diagonal carry garbage nobody reads), inverse pivots, then the counter
#if FLOWP_LDSMUL
        dst[3] = x3;
#else
        dst[0] = x0;
        dst[1] = x1;
        dst[2] = x2;
        dst[3] = x3;
#endif
        if (r == 0) {
            invd[c0 + 0] = y0;
            invd[c0 + 1] = y1;
            invd[c0 + 2] = y2;
            invd[c0 + 3] = y3;
            lds_post(&sy->prog, G + 1);
        }
#if FLOWP_LATE_CHECK
        if (fetch && seen < G + 1) { // the hand-over was not there at the top of the round: wait for it where it is needed
            lds_await(&sy->hflag[(G + 1) & 1], G + 1);
            n[0] = h[0];
            n[1] = h[1];
            n[2] = h[2];
            n[3] = h[3];
        }
#endif
        if (G < 15) {
#if FLOWP_LDSMUL
#pragma unroll
            for (int e = 0; e < 4; ++e)
                n[e] = fma(-x0, m0[e], n[e]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                n[e] = fma(-x1, m1[e], n[e]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                n[e] = fma(-x2, m2[e], n[e]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                c[e] = fma(-x3, bcast_lane(x3, c0 + 4 + e), n[e]);
#else
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double v = n[e];
                v = fma(-x0, bcast_lane(x0, c0 + 4 + e), v);
                v = fma(-x1, bcast_lane(x1, c0 + 4 + e), v);
                v = fma(-x2, bcast_lane(x2, c0 + 4 + e), v);
                v = fma(-x3, bcast_lane(x3, c0 + 4 + e), v);
                c[e] = v;
            }
#endif
        }
        FTS(0, G);
    }
};
template <>
struct FlowP<-1> {
    static __device__ __forceinline__ void run(double (&)[4], double (&)[4], double*, const double*, double*, DiagSync*, int) {}
};

// ---- U ---------------------------------------------------------------------------------------------------------------------
// acc[bi][n]: element (row 16 bi + 4 ((lane >> 2) & 3) + (lane >> 4), column 16 BJ + 4 n + (lane & 3)) — st16's layout
template <int BJ, int T>
struct FlowU {
    static __device__ __forceinline__ void run(double (&acc)[4][4], const double* Ls, double* H,
                                               DiagSync* sy, int lane)
    {
        FlowU<BJ, T - 1>::run(acc, Ls, H, sy, lane);
        constexpr int c0 = 4 * T, g2 = T + 2; // g2: the first group that still needs this panel from the update waves
        if (4 * BJ + 3 < g2)
            return; // this block column is finished
        constexpr int bi_lo = (g2 >> 2) > BJ ? (g2 >> 2) : BJ;
        constexpr bool hand = (g2 >> 2) == BJ;
        constexpr int nh = g2 & 3;
        lds_await(&sy->prog, T + 1);
        const int kq = lane >> 4;
        double av[4], bv[4];
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
            const int nq = hand ? ((nh + nn) & 3) : nn; // the group to hand over first
            if (4 * BJ + nq >= g2)
                bv[nq] = Ls[(16 * BJ + 4 * nq + (lane & 3)) * XS + c0 + kq];
        }
#pragma unroll
        for (int bi = bi_lo; bi < 4; ++bi)
            av[bi] = -Ls[(16 * bi + (lane & 15)) * XS + c0 + kq];
        if (hand) {
#pragma unroll
            for (int bi = bi_lo; bi < 4; ++bi) {
                acc[bi][nh] = mfma4(av[bi], bv[nh], acc[bi][nh]);
                const int row = 16 * bi + 4 * ((lane >> 2) & 3) + (lane >> 4);
                H[(g2 & 1) * (NB * 4) + row * 4 + (lane & 3)] = acc[bi][nh];
            }
            if (lane == 0)
                lds_post(&sy->hflag[g2 & 1], g2);
        }
#pragma unroll
        for (int nq = 0; nq < 4; ++nq) {
            if (4 * BJ + nq < g2 || (hand && nq == nh))
                continue;
#pragma unroll
            for (int bi = bi_lo; bi < 4; ++bi)
                acc[bi][nq] = mfma4(av[bi], bv[nq], acc[bi][nq]);
        }
        FTS(1 + BJ, T);
    }
};
template <int BJ>
struct FlowU<BJ, -1> {
    static __device__ __forceinline__ void run(double (&)[4][4], const double*, double*, DiagSync*, int) {}
};
static __device__ __forceinline__ void flow_x21(DiagSync* sy, const double* Xw, double* __restrict__ Xt, int u, int lane, int poff);
template <int BJ>
static __device__ __forceinline__ void flow_u_wave(const double* Ls, double* H, DiagSync* sy, const double* Xw,
                                                   double* __restrict__ Xt, int lane)
{
    double acc[4][4];
#pragma unroll
    for (int bi = BJ; bi < 4; ++bi)
#pragma unroll
        for (int nq = 0; nq < 4; ++nq)
            acc[bi][nq] = Ls[(16 * bi + 4 * ((lane >> 2) & 3) + (lane >> 4)) * XS + 16 * BJ + 4 * nq + (lane & 3)];
    __syncthreads(); // every wave has its part of the block in registers: P may start overwriting Ls with L
    FlowU<BJ, 13>::run(acc, Ls, H, sy, lane);
    flow_x21(sy, Xw, Xt, BJ, lane, 0);
}

// ---- X ---------------------------------------------------------------------------------------------------------------------
// The inversion pipeline of round 1 (XPipe32: right-looking forward substitution on the identity, one round behind the
// factorisation) on all 64 lanes: lane = (column c of X, half h); half h folds the panel's columns 2h, 2h+1 into its own
// partial sums, the four rows that become final in a round are the sum over the two halves (v_permlane32_swap).
static __device__ __forceinline__ double sum_halves(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]); // the same bits in both halves
}
typedef double v2d_t __attribute__((ext_vector_type(2)));
// The fold's LDS reads are written out (the compiler either keeps three of them in flight, or lifts a whole round's reads to
// the top and spills them): CH reads per chunk, the next chunk's in flight under the current chunk's FMAs.  LDS operations
// return in order, so "at most n outstanding" means everything but the last n has arrived; the compiler's own waits only
// ever count fewer operations than are outstanding, i.e. they wait longer than they must, never shorter.
#define XFOLD_CH 7 // two chunks in flight = 14 operations (lgkmcnt is a 4-bit counter on gfx9)
template <int ROW0, int N>
static __device__ __forceinline__ void xfold_issue(v2d_t (&q)[XFOLD_CH], unsigned lbase)
{
#pragma unroll
    for (int i = 0; i < XFOLD_CH; ++i)
        if (i < N)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(lbase), "n"((ROW0 + i) * XS * 8));
}
template <int PENDING>
static __device__ __forceinline__ void xfold_wait(v2d_t (&q)[XFOLD_CH])
{
    static_assert(XFOLD_CH == 7, "operand list");
    asm volatile("s_waitcnt lgkmcnt(%7)"
                 : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6])
                 : "n"(PENDING));
}
template <int G, int CHK>
struct XFold { // rows i0 + 4 + XFOLD_CH CHK ..: q holds this chunk's multipliers, already requested
    static constexpr int i0 = 4 * (G & 7), nrow = 28 - i0, r0 = XFOLD_CH * CHK, left = nrow - r0;
    static __device__ __forceinline__ void run(double (&S)[32], v2d_t (&q)[XFOLD_CH], unsigned lbase, double xa, double xb)
    {
        if constexpr (left > 0) {
            constexpr int nnext = left - XFOLD_CH > XFOLD_CH ? XFOLD_CH : (left - XFOLD_CH > 0 ? left - XFOLD_CH : 0);
            v2d_t qn[XFOLD_CH];
#pragma unroll
            for (int i = 0; i < XFOLD_CH; ++i)
                qn[i] = v2d_t{xa, xb}; // (defined for the asm operands of a short tail chunk — NOT copied from q: the
                                       // compiler would read q's registers while their reads are still in flight)
            xfold_issue<i0 + 4 + r0 + XFOLD_CH, nnext>(qn, lbase);
            xfold_wait<nnext>(q);
#pragma unroll
            for (int i = 0; i < XFOLD_CH; ++i)
                if (i < left)
                    S[i0 + 4 + r0 + i] = fma(-q[i][0], xa, fma(-q[i][1], xb, S[i0 + 4 + r0 + i]));
#pragma unroll
            for (int i = 0; i < XFOLD_CH; ++i)
                if (i < left)
                    asm volatile("" : "+v"(S[i0 + 4 + r0 + i])); // the FMAs happen here, not after the next round's wait
            XFold<G, CHK + 1>::run(S, qn, lbase, xa, xb);
        }
    }
};
template <int G, int POFF = 0> // POFF: added to the panel / round counters (diag_flow2.h runs the pipeline a second time with 16)
struct FlowX {
    static __device__ __forceinline__ void run(double (&S)[32], const double* Ls, const double* invd, double* __restrict__ Xt,
                                               DiagSync* sy, double* Xw, int c, int h, double* __restrict__ S22)
    {
        FlowX<G - 1, POFF>::run(S, Ls, invd, Xt, sy, Xw, c, h, S22);
        constexpr int hb = G >> 3, i0 = 4 * (G & 7), base = 32 * hb, nrow = 28 - i0;
        if (G == 8) { // second half-block: start again from the identity
#pragma unroll
            for (int k = 0; k < 32; ++k)
                S[k] = (h == 0 && k == c) ? 1.0 : 0.0;
        }
        lds_await(&sy->prog, POFF + G + 1); // columns 4G..4G+3 of L and their inverse pivots are final
        const double* Lb = Ls + base * XS + base; // this half-block of L
        // my two columns (2h, 2h+1) of the panel, as an LDS byte address; the reads name rows relative to Lb's first
        const unsigned lbase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) double*)(Lb + i0 + 2 * h);
        const double s0 = sum_halves(S[i0]), s1 = sum_halves(S[i0 + 1]), s2 = sum_halves(S[i0 + 2]), s3 = sum_halves(S[i0 + 3]);
        const double x0 = s0 * invd[base + i0];
        const double x1 = fma(-Lb[(i0 + 1) * XS + i0], x0, s1) * invd[base + i0 + 1];
        double t2 = fma(-Lb[(i0 + 2) * XS + i0], x0, s2);
        t2 = fma(-Lb[(i0 + 2) * XS + i0 + 1], x1, t2);
        const double x2 = t2 * invd[base + i0 + 2];
        double t3 = fma(-Lb[(i0 + 3) * XS + i0], x0, s3);
        t3 = fma(-Lb[(i0 + 3) * XS + i0 + 1], x1, t3);
        t3 = fma(-Lb[(i0 + 3) * XS + i0 + 2], x2, t3);
        const double x3 = t3 * invd[base + i0 + 3];
        v2d_t q[XFOLD_CH];
#pragma unroll
        for (int i = 0; i < XFOLD_CH; ++i)
            q[i] = v2d_t{x0, x1}; // (defined for the asm operands)
        xfold_issue<i0 + 4, (nrow > XFOLD_CH ? XFOLD_CH : nrow)>(q, lbase);
        if (h == 0) { // rows base+i0 .. +3 of X are final: Xt[col + 64 row] = X[row][col]
            DIAG_XT_STORE(Xt + base + c + NB * (base + i0 + 0), x0);
            DIAG_XT_STORE(Xt + base + c + NB * (base + i0 + 1), x1);
            DIAG_XT_STORE(Xt + base + c + NB * (base + i0 + 2), x2);
            DIAG_XT_STORE(Xt + base + c + NB * (base + i0 + 3), x3);
            if (S22) { // ... and to the polled copy of this half-block (X11 at + 0, X22 at + 2048)
                double* Sp = S22 + 2048 * hb;
                DIAG_XT_STORE(Sp + c + 32 * (i0 + 0), x0);
                DIAG_XT_STORE(Sp + c + 32 * (i0 + 1), x1);
                DIAG_XT_STORE(Sp + c + 32 * (i0 + 2), x2);
                DIAG_XT_STORE(Sp + c + 32 * (i0 + 3), x3);
            }
            double* xh = Xw + hb * (32 * XH) + i0 * XH + c; // and into LDS, for the off-diagonal quarter (flow_x21)
            xh[0] = x0;
            xh[XH] = x1;
            xh[2 * XH] = x2;
            xh[3 * XH] = x3;
        }
        const double xa = h ? x2 : x0, xb = h ? x3 : x1;
        XFold<G, 0>::run(S, q, lbase, xa, xb);
        if ((G & 7) == 7 && c == 0 && h == 0)
            lds_post(&sy->xprog, POFF + G + 1); // a half-block of X is complete in Xw
        FTS(5, G);
    }
};
template <int POFF>
struct FlowX<-1, POFF> {
    static __device__ __forceinline__ void run(double (&)[32], const double*, const double*, double*, DiagSync*, double*, int, int, double*) {}
};

// ---- the off-diagonal quarter X21 = -X22 (L21 X11) -------------------------------------------------------------------------
// Round 1 left it to a kernel of its own after the factorisation (k_xinv_complete) because the panel steps can do with the
// two diagonal quarters (three small products, six barriers: trsm_tile_half).  With all of X the solve against L11 is ONE
// product (trsm_tile_full, potrf.hip).  Here: W = L21 X11 by the eighth wave while the second half of the block is still being
// factored, X21 = -X22 W by the four update waves once X22 is complete — 32 matrix-core instructions each, ~1 k cycles behind
// the inversion wave.  mfma4 layouts as in mm16 / st16 (potrf.hip).
static __device__ __forceinline__ void flow_w_wave(const double* Ls, DiagSync* sy, double* Xw, int lane, int poff = 0)
{
    lds_await(&sy->xprog, poff + 8);
    // ... and not before panel 10 is out: this wave shares its SIMD with update wave U2, which hands groups 8..11 to P in
    // rounds 6..9 — started at round 8 its matrix-core instructions delayed those hand-overs and P with them (round 10:
    // 1.6 k cycles instead of 1.0 k, r02_diag_flow_stamps.log of that build).  Three rounds are enough for W.
    lds_await(&sy->prog, poff + 11);
    const double* X11 = Xw;
    double* W = Xw + 2 * (32 * XH);
    const int kq = lane >> 4;
    double acc[2][8];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int n = 0; n < 8; ++n)
            acc[sl][n] = 0.0;
#pragma unroll
    for (int ks = 0; ks < 32; ks += 4) {
        double a[2];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
            a[sl] = Ls[(32 + 16 * sl + (lane & 15)) * XS + ks + kq];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (4 * n > ks + 3)
                continue; // X11 is lower triangular: its rows ks .. ks+3 are zero from column ks+4 on
            const double b = X11[(ks + kq) * XH + 4 * n + (lane & 3)];
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
                acc[sl][n] = mfma4(a[sl], b, acc[sl][n]);
        }
    }
    const int row = 4 * ((lane >> 2) & 3) + (lane >> 4), col = lane & 3;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int n = 0; n < 8; ++n)
            W[(16 * sl + row) * XH + 4 * n + col] = acc[sl][n];
    if (lane == 0)
        lds_post(&sy->wdone, (poff >> 4) + 1);
}
// update wave u: columns 8u .. 8u+7 of X21, all 32 rows
static __device__ __forceinline__ void flow_x21(DiagSync* sy, const double* Xw, double* __restrict__ Xt, int u, int lane, int poff = 0)
{
    lds_await(&sy->wdone, (poff >> 4) + 1);
    lds_await(&sy->xprog, poff + 16);
    const double* X22 = Xw + 32 * XH;
    const double* W = Xw + 2 * (32 * XH);
    const int kq = lane >> 4;
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
    for (int ks = 0; ks < 32; ks += 4) {
        double b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
            b[j] = W[(ks + kq) * XH + 4 * (2 * u + j) + (lane & 3)];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (16 * sl + 15 < ks)
                continue; // X22 is lower triangular
            const double a = -X22[(16 * sl + (lane & 15)) * XH + ks + kq];
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[sl][j] = mfma4(a, b[j], acc[sl][j]);
        }
    }
    const int row = 4 * ((lane >> 2) & 3) + (lane >> 4), col = lane & 3;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int j = 0; j < 2; ++j) // Xt[col + 64 row] = X[row][col]
            DIAG_XT_STORE(Xt + 4 * (2 * u + j) + col + NB * (32 + 16 * sl + row), acc[sl][j]);
}

// ---- S ---------------------------------------------------------------------------------------------------------------------
template <int G>
struct FlowS {
    static __device__ __forceinline__ void run(int& bad, const double* Ls, const double* invd,
                                               double* __restrict__ Ad, int64_t lda, const DiagSync* sy, int r,
                                               double* __restrict__ SL21)
    {
        FlowS<G - 1>::run(bad, Ls, invd, Ad, lda, sy, r, SL21);
        lds_await(&sy->prog, G + 1);
        constexpr int c0 = 4 * G;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double v = Ls[r * XS + c0 + e];
            if (c0 + e <= r)
                Ad[r + (int64_t)(c0 + e) * lda] = v;
            if (G < 8 && SL21 && r >= 32) // the polled copy of L21: [row - 32 + 32 col]
                DIAG_XT_STORE(SL21 + (r - 32) + 32 * (c0 + e), v);
            // first non-positive pivot (the reference never checks LLT::info(), gp.hpp:565): its inverse root is not a
            // positive finite number, and neither is any later one
            const double y = invd[c0 + e];
            if (bad == 0 && !(y > 0.0 && y < __builtin_huge_val()))
                bad = c0 + e + 1;
        }
        FTS(6, G);
    }
};
template <>
struct FlowS<-1> {
    static __device__ __forceinline__ void run(int&, const double*, const double*, double*, int64_t, const DiagSync*, int, double*) {}
};

// Factor the 64 x 64 block held in Ls (Ls[row * XS + col], lower triangle meaningful) and invert its two 32 x 32 diagonal
// blocks: L -> Ad (global, lower triangle) and Ls, X^T = L^-T -> Xt (all of it but the zero quarter).  Call with >= 8 waves,
// all of them, after Ls is complete and `sy` was cleared by diag_flow_init and a barrier; waves >= 8 just leave.  There is NO
// barrier at the end: a caller that re-uses the LDS afterwards has to synchronise itself.
static __device__ __forceinline__ void diag_flow_init(DiagSync* sy)
{
    if (threadIdx.x == 0) {
        sy->prog = 0;
        sy->hflag[0] = -1;
        sy->hflag[1] = -1;
        sy->xprog = 0;
        sy->wdone = 0;
    }
}
static __device__ __forceinline__ void diag_flow(double* Ls, double* H, double* invd,
                                                 DiagSync* sy, double* __restrict__ Ad, int64_t lda,
                                                 double* __restrict__ Xt, int* __restrict__ info, int64_t goff, int wave,
                                                 int lane, double* Xw, const DiagEarly* ea = nullptr)
{
    if (wave >= 8)
        return; // s_barrier only counts the waves that are still alive
    if (wave == 7) {
        __syncthreads();
        flow_w_wave(Ls, sy, Xw, lane);
        return;
    }
    if (wave >= 1 && wave <= 4) {
        switch (wave) {
        case 1: flow_u_wave<0>(Ls, H, sy, Xw, Xt, lane); break;
        case 2: flow_u_wave<1>(Ls, H, sy, Xw, Xt, lane); break;
        case 3: flow_u_wave<2>(Ls, H, sy, Xw, Xt, lane); break;
        default: flow_u_wave<3>(Ls, H, sy, Xw, Xt, lane); break;
        }
        return;
    }
    if (wave == 0) {
        double c[4], n[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c[e] = Ls[lane * XS + e];
            n[e] = Ls[lane * XS + 4 + e];
        }
        __syncthreads();
        FlowP<15>::run(c, n, Ls, H, invd, sy, lane);
        return;
    }
    __syncthreads();
    if (wave == 5) {
        double S[32];
#pragma unroll
        for (int k = 0; k < 32; ++k)
            S[k] = (lane < 32 && k == lane) ? 1.0 : 0.0;
        FlowX<15>::run(S, Ls, invd, Xt, sy, Xw, lane & 31, lane >> 5, (ea && !ea->mute) ? ea->S : nullptr);
        return;
    }
    int bad = 0;
    FlowS<15>::run(bad, Ls, invd, Ad, lda, sy, lane, (ea && !ea->mute) ? ea->S + 1024 : nullptr);
    if (lane == 0 && bad != 0 && *info == 0)
        *info = (int)(goff + bad);
}
