// diag_flow2.h — TWO consecutive 64 x 64 diagonal blocks and the tile between them as ONE data-flow of specialised waves
// (round 4; included by potrf.hip behind diag_flow.h, whose helpers and inversion pipeline it re-uses).
//
// A hop of the tiled factorisation (k_tail) is the panel wave's 64 dependent columns (~7 us) plus ~5.5 us between one
// diagonal block and the next: the last rows of the block inverse leave the factoring workgroup, cross the chip, the next
// diagonal workgroup finishes the solve of its left tile against them, squares it into its block, stages the block, starts.
// Here every second one of those crossings is gone: a workgroup factors the 128 x 128 block
//        [ A(c, c)                ]
//        [ A(c+1, c)  A(c+1, c+1) ]
// in one go — 32 column groups of four.  The panel wave holds two rows per lane (r and 64 + r) for the first 16 groups: the
// tile A(c+1, c) is solved BY the factorisation (its columns are scaled with the pivots like any other row below the
// diagonal), not against an inverse afterwards, and its square reaches A(c+1, c+1) through the update waves' rank-4 updates,
// round by round.  Measured on the panel wave with a dummy second row (tools/diagbench, -DFLOWP_DUMMY2): 19.1 k cycles for
// 16 groups of two rows against 15.9 k of one, i.e. 14.6 us for the pair where two single blocks and the crossing between
// them take 19.4.
//   wave 0      P  lane = row r (rows r and 64 + r while the columns are < 64).  As FlowP (diag_flow.h): 4 x 4 pivot block
//                  broadcast once, factored redundantly per lane, columns published into L1 / L2, `prog` bumped, the next
//                  group updated by this panel in the wave itself.
//   waves 1-3   U  the 16 x 16 blocks of the 128 x 128 lower triangle, block columns {0,5,6} / {1,4,7} / {2,3} per wave,
//                  in the v_mfma_f64_4x4x4 accumulator layout; hand group T + 2 to P through H + `hflag` after panel T.
//   wave 5      X  the inversion pipeline of diag_flow.h for block 0, behind panels 0..15 (wave 7 runs it for block 1 behind panels
//                  16..31; a work area each)
//   waves 6, 7  S  store L: wave 6 block 0, block 1 and the polled copies of both blocks' L21; wave 7 the tile (to the matrix
//                  AND to its polled hand-over slot, column by column: the tiles of column c + 1 read it from there), then X of block 1
//   wave 4      W  W = L21 X11 of both blocks (the panel wave's SIMD: the only role with next to no arithmetic); the update
//                  waves and this one finish the off-diagonal quarters X21 = -X22 W at the end
// LDS (doubles): L1 128 x XS | L2 64 x XS | H 2 x 128 x 4 | invd 128 | sync 8 | Xw 2 x 3 x 32 x XH  = 20,360 <= TAIL_LDS_DOUBLES
#pragma once

#define D2R (2 * NB)
#ifdef DIAG_TIMING
__device__ long long g_flow2_ts[8][33]; // [wave][round]: when the wave finished its round ([..][32]: start)
#define FTS2(w, G) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_flow2_ts[w][G] = clock64(); } while (0)
#else
#define FTS2(w, G) do { } while (0)
#endif
#define D2_OFF_L2 (D2R * XS)
#define D2_OFF_H (D2_OFF_L2 + NB * XS)
#define D2_OFF_INVD (D2_OFF_H + 2 * D2R * 4)
#define D2_OFF_SY (D2_OFF_INVD + D2R)
#define D2_OFF_XW (D2_OFF_SY + 8)
#define D2_LDS_DOUBLES (D2_OFF_XW + 2 * DIAG_XW_DOUBLES)

// ---- P ---------------------------------------------------------------------------------------------------------------------
template <int G>
struct FlowP2 {
    // ca / cb: group G's four columns for rows r / 64 + r, complete (ca is dead from group 16 on)
    static __device__ __forceinline__ void run(double (&ca)[4], double (&cb)[4], double (&na)[4], double (&nb)[4], double* L1,
                                               double* L2, const double* H, double* invd, DiagSync* sy, int r)
    {
        FlowP2<G - 1>::run(ca, cb, na, nb, L1, L2, H, invd, sy, r);
        constexpr int c0 = 4 * G, hf = G >> 4, lc0 = c0 & 63;
        constexpr bool two = hf == 0;            // rows r and 64 + r hold columns of this group
        constexpr bool two_next = G + 1 < 16;    // ... and of the next one
        constexpr bool fetch = G >= 1 && G < 31; // groups 0 and 1 were read from L1 at the start
        const double* h = H + ((G + 1) & 1) * (D2R * 4);
        int seen = 0;
        if (fetch) {
            seen = lds_peek(&sy->hflag[(G + 1) & 1]);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (two_next)
                    na[e] = h[r * 4 + e];
                nb[e] = h[(NB + r) * 4 + e];
            }
        }
        // the 4 x 4 pivot block: rows c0 .. c0+3 live in lanes lc0 .. lc0+3 of ca (first half) / cb (second half)
        double(&cp)[4] = *(hf == 0 ? &ca : &cb);
        const double b00 = bcast_lane(cp[0], lc0), b10 = bcast_lane(cp[0], lc0 + 1), b20 = bcast_lane(cp[0], lc0 + 2),
                     b30 = bcast_lane(cp[0], lc0 + 3);
        const double b11 = bcast_lane(cp[1], lc0 + 1), b21 = bcast_lane(cp[1], lc0 + 2), b31 = bcast_lane(cp[1], lc0 + 3);
        const double b22 = bcast_lane(cp[2], lc0 + 2), b32 = bcast_lane(cp[2], lc0 + 3);
        const double b33 = bcast_lane(cp[3], lc0 + 3);
        const double y0 = rsq_newton(b00);
        const double l10 = b10 * y0, l20 = b20 * y0, l30 = b30 * y0;
        double xa0 = 0.0, xa1 = 0.0, xa2 = 0.0, xa3 = 0.0;
        if (two)
            xa0 = ca[0] * y0;
        const double xb0 = cb[0] * y0;
        const double y1 = rsq_newton(fma(-l10, l10, b11));
        const double l21 = fma(-l20, l10, b21) * y1, l31 = fma(-l30, l10, b31) * y1;
        if (two)
            xa1 = fma(-xa0, l10, ca[1]) * y1;
        const double xb1 = fma(-xb0, l10, cb[1]) * y1;
        // where the columns go: row r of L1 (first half), row 64 + r of L1 / row r of L2
        double* dsta = L1 + r * XS + c0;
        double* dstb = hf == 0 ? L1 + (NB + r) * XS + c0 : L2 + r * XS + lc0;
        if (two) {
            dsta[0] = xa0;
            dsta[1] = xa1;
        }
        dstb[0] = xb0;
        dstb[1] = xb1;
        const double y2 = rsq_newton(fma(-l21, l21, fma(-l20, l20, b22)));
        const double l32 = fma(-l31, l21, fma(-l30, l20, b32)) * y2;
        if (two)
            xa2 = fma(-xa1, l21, fma(-xa0, l20, ca[2])) * y2;
        const double xb2 = fma(-xb1, l21, fma(-xb0, l20, cb[2])) * y2;
        if (two)
            dsta[2] = xa2;
        dstb[2] = xb2;
        // the look-ahead update's multipliers x_j[row c0+4+e], j = 0..2, come back through LDS (this wave's own writes, in order)
        double m0[4], m1[4], m2[4];
        if (G < 31) {
            const double* mrow = hf == 0 ? L1 + (c0 + 4) * XS + c0 : L2 + (lc0 + 4) * XS + lc0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m0[e] = mrow[e * XS + 0];
                m1[e] = mrow[e * XS + 1];
                m2[e] = mrow[e * XS + 2];
            }
        }
        const double y3 = rsq_newton(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, b33))));
        if (two)
            xa3 = fma(-xa2, l32, fma(-xa1, l31, fma(-xa0, l30, ca[3]))) * y3;
        const double xb3 = fma(-xb2, l32, fma(-xb1, l31, fma(-xb0, l30, cb[3]))) * y3;
        if (two)
            dsta[3] = xa3;
        dstb[3] = xb3;
        if (r == 0) {
            invd[c0 + 0] = y0;
            invd[c0 + 1] = y1;
            invd[c0 + 2] = y2;
            invd[c0 + 3] = y3;
            lds_post(&sy->prog, G + 1);
        }
        // The next group as the update waves left it was read at the top of the round; if its counter was not there yet, wait for it
        // HERE, where it is needed, and not earlier: the hand-over loop (P publishes panel G-1 -> an update wave sees it, applies it
        // to group G+1 and posts -> P sees that) takes ~600 cycles — checked half-way through the round P waited for most of them
        // in every round (1800 cycles a round, tools/diagbench2); at the end of the round they are hidden behind P's own work
        if (fetch && seen < G + 1) {
            lds_await(&sy->hflag[(G + 1) & 1], G + 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (two_next)
                    na[e] = h[r * 4 + e];
                nb[e] = h[(NB + r) * 4 + e];
            }
        }
        if (G < 31) {
            // x3's multipliers: rows c0+4 .. c0+7 sit in xa3 while they are < 64 (groups 0..14), in xb3 (lanes & 63) otherwise
            constexpr bool mul_in_a = hf == 0 && c0 + 4 < NB;
            const double xm = mul_in_a ? xa3 : xb3;
            double m3[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                m3[e] = bcast_lane(xm, (c0 + 4 + e) & 63);
            if (two_next) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    ca[e] = fma(-xa3, m3[e], fma(-xa2, m2[e], fma(-xa1, m1[e], fma(-xa0, m0[e], na[e]))));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                cb[e] = fma(-xb3, m3[e], fma(-xb2, m2[e], fma(-xb1, m1[e], fma(-xb0, m0[e], nb[e]))));
        }
        FTS2(0, G);
    }
};
template <>
struct FlowP2<-1> {
    static __device__ __forceinline__ void run(double (&)[4], double (&)[4], double (&)[4], double (&)[4], double*, double*,
                                               const double*, double*, DiagSync*, int) {}
};

// ---- U ---------------------------------------------------------------------------------------------------------------------
// element (row, col) of the 128 x 128 block as the initial data / the published columns hold it
static __device__ __forceinline__ const double* d2_elem(const double* L1, const double* L2, int row, int col)
{
    return col < NB ? L1 + row * XS + col : L2 + (row - NB) * XS + (col - NB);
}
// one block column BJ under panel T.  acc[bi][n]: element (row 16 bi + 4 ((lane >> 2) & 3) + (lane >> 4), column 16 BJ + 4 n + (lane & 3))
template <int BJ, int T, bool HANDPASS>
static __device__ __forceinline__ void flowu2_col(double (&acc)[8][4], const double (&av)[8], const double* L1, const double* L2,
                                                  double* H, DiagSync* sy, int lane)
{
    constexpr int c0 = 4 * T, g2 = T + 2, hf = T >> 4, lc0 = c0 & 63;
    if constexpr (4 * BJ + 3 >= g2) {
        constexpr int bi_lo = (g2 >> 2) > BJ ? (g2 >> 2) : BJ;
        constexpr bool hand = (g2 >> 2) == BJ;
        constexpr int nh = g2 & 3;
        const int kq = lane >> 4;
        if constexpr (HANDPASS) {
            if constexpr (hand) {
                const int brow = 16 * BJ + 4 * nh + (lane & 3);
                const double bv = hf == 0 ? L1[brow * XS + c0 + kq] : L2[(brow - NB) * XS + lc0 + kq];
#pragma unroll
                for (int bi = bi_lo; bi < 8; ++bi) {
                    acc[bi][nh] = mfma4(av[bi], bv, acc[bi][nh]);
                    const int row = 16 * bi + 4 * ((lane >> 2) & 3) + (lane >> 4);
                    H[(g2 & 1) * (D2R * 4) + row * 4 + (lane & 3)] = acc[bi][nh];
                }
                if (lane == 0)
                    lds_post(&sy->hflag[g2 & 1], g2);
            }
        }
        else {
#pragma unroll
            for (int nq = 0; nq < 4; ++nq) {
                if (4 * BJ + nq < g2 || (hand && nq == nh))
                    continue;
                const int brow = 16 * BJ + 4 * nq + (lane & 3);
                const double bv = hf == 0 ? L1[brow * XS + c0 + kq] : L2[(brow - NB) * XS + lc0 + kq];
#pragma unroll
                for (int bi = bi_lo; bi < 8; ++bi)
                    acc[bi][nq] = mfma4(av[bi], bv, acc[bi][nq]);
            }
        }
    }
}
// block columns of update wave U (three waves, one per SIMD the panel wave is not on): {0, 5, 6}, {1, 4, 7}, {2, 3} — 13, 12 and
// 11 of the 36 blocks in the first rounds, and still spread over the waves when only the late block columns are left
template <int U> struct U2Cols;
template <> struct U2Cols<0> { static constexpr int a = 0, b = 5, c = 6; };
template <> struct U2Cols<1> { static constexpr int a = 1, b = 4, c = 7; };
template <> struct U2Cols<2> { static constexpr int a = 2, b = 3, c = -1; };
template <int U, int T>
struct FlowU2 {
    static __device__ __forceinline__ void run(double (&accA)[8][4], double (&accB)[8][4], double (&accC)[8][4], const double* L1,
                                               const double* L2, double* H, DiagSync* sy, int lane)
    {
        FlowU2<U, T - 1>::run(accA, accB, accC, L1, L2, H, sy, lane);
        constexpr int BJa = U2Cols<U>::a, BJb = U2Cols<U>::b, BJc = U2Cols<U>::c, BJmax = BJc >= 0 ? BJc : BJb;
        constexpr int c0 = 4 * T, g2 = T + 2, hf = T >> 4, lc0 = c0 & 63;
        if constexpr (4 * BJmax + 3 >= g2) { // (the last of its block columns still needs this panel)
            lds_await(&sy->prog, T + 1);
            // the rows' operand, shared by the wave's block columns: row blocks from the first unfinished group on
            constexpr int lo = (g2 >> 2) > BJa ? (g2 >> 2) : BJa;
            const int kq = lane >> 4;
            double av[8];
#pragma unroll
            for (int bi = lo; bi < 8; ++bi) {
                const int arow = 16 * bi + (lane & 15);
                av[bi] = -(hf == 0 ? L1[arow * XS + c0 + kq] : L2[(arow - NB) * XS + lc0 + kq]);
            }
            // the group P is waiting for first, then everything else
            flowu2_col<BJa, T, true>(accA, av, L1, L2, H, sy, lane);
            flowu2_col<BJb, T, true>(accB, av, L1, L2, H, sy, lane);
            if constexpr (BJc >= 0)
                flowu2_col<BJc, T, true>(accC, av, L1, L2, H, sy, lane);
            flowu2_col<BJa, T, false>(accA, av, L1, L2, H, sy, lane);
            flowu2_col<BJb, T, false>(accB, av, L1, L2, H, sy, lane);
            if constexpr (BJc >= 0)
                flowu2_col<BJc, T, false>(accC, av, L1, L2, H, sy, lane);
            FTS2(1 + U, T);
        }
    }
};
template <int U>
struct FlowU2<U, -1> {
    static __device__ __forceinline__ void run(double (&)[8][4], double (&)[8][4], double (&)[8][4], const double*, const double*, double*,
                                               DiagSync*, int) {}
};
template <int U>
static __device__ __forceinline__ void flow_u2_wave(const double* L1, const double* L2, double* H, DiagSync* sy, int lane)
{
    constexpr int BJa = U2Cols<U>::a, BJb = U2Cols<U>::b, BJc = U2Cols<U>::c;
    double accA[8][4], accB[8][4], accC[8][4];
#pragma unroll
    for (int bi = 0; bi < 8; ++bi)
#pragma unroll
        for (int nq = 0; nq < 4; ++nq) {
            const int row = 16 * bi + 4 * ((lane >> 2) & 3) + (lane >> 4);
            accA[bi][nq] = bi >= BJa ? *d2_elem(L1, L2, row, 16 * BJa + 4 * nq + (lane & 3)) : 0.0;
            accB[bi][nq] = bi >= BJb ? *d2_elem(L1, L2, row, 16 * BJb + 4 * nq + (lane & 3)) : 0.0;
            accC[bi][nq] = (BJc >= 0 && bi >= BJc) ? *d2_elem(L1, L2, row, 16 * (BJc >= 0 ? BJc : 7) + 4 * nq + (lane & 3)) : 0.0;
        }
    __syncthreads(); // every wave has its part of the block in registers: P may start overwriting L1 with L
    FlowU2<U, 29>::run(accA, accB, accC, L1, L2, H, sy, lane);
}

// ---- S ---------------------------------------------------------------------------------------------------------------------
struct Diag2Out {
    double* Ad0;  // block (c, c) in the matrix
    double* Atm;  // tile (c+1, c) in the matrix ...
    double* LPtm; // ... and its polled hand-over slot ([row + 64 col]; nullptr: muted)
    double* Ad1;  // block (c+1, c+1)
    double* SL21_0; // polled copies of the two blocks' L21 ([row - 32 + 32 col]; nullptr: muted)
    double* SL21_1;
    int64_t lda;
};
// WHICH 0 (wave 6): block 0, the polled copies of L21, block 1, the pivot check; WHICH 1 (wave 7): the tile (c+1, c) — the
// matrix and its polled slot, 8 stores a round, what one wave doing everything fell 6 k cycles behind with
template <int G, int WHICH>
struct FlowS2 {
    static __device__ __forceinline__ void run(int& bad, const double* L1, const double* L2, const double* invd, const Diag2Out& o,
                                               const DiagSync* sy, int r)
    {
        FlowS2<G - 1, WHICH>::run(bad, L1, L2, invd, o, sy, r);
        constexpr int c0 = 4 * G, hf = G >> 4, lc0 = c0 & 63;
        if constexpr (WHICH == 1 && hf == 1)
            return;
        lds_await(&sy->prog, G + 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (WHICH == 1) {
                const double vb = L1[(NB + r) * XS + c0 + e];
                if (o.LPtm) // the tile's hand-over slot first: the tiles of column c + 1 are waiting for it
                    DIAG_XT_STORE(o.LPtm + r + NB * (c0 + e), vb);
                o.Atm[r + (int64_t)(c0 + e) * o.lda] = vb;
                continue;
            }
            if (hf == 0) {
                const double va = L1[r * XS + c0 + e];
                if (c0 + e <= r)
                    o.Ad0[r + (int64_t)(c0 + e) * o.lda] = va;
                if (G < 8 && o.SL21_0 && r >= 32)
                    DIAG_XT_STORE(o.SL21_0 + (r - 32) + 32 * (c0 + e), va);
            }
            else {
                const double vb = L2[r * XS + lc0 + e];
                if (lc0 + e <= r)
                    o.Ad1[r + (int64_t)(lc0 + e) * o.lda] = vb;
                if (G - 16 < 8 && o.SL21_1 && r >= 32)
                    DIAG_XT_STORE(o.SL21_1 + (r - 32) + 32 * (lc0 + e), vb);
            }
            const double y = invd[c0 + e];
            if (bad == 0 && !(y > 0.0 && y < __builtin_huge_val()))
                bad = c0 + e + 1;
        }
        FTS2(6 + WHICH, G);
    }
};
template <int WHICH>
struct FlowS2<-1, WHICH> {
    static __device__ __forceinline__ void run(int&, const double*, const double*, const double*, const Diag2Out&, const DiagSync*, int) {}
};

// Factor the 128 x 128 block held in L1 (rows 0..127, columns 0..63) and L2 (rows 64..127, columns 64..127) — lower triangle
// meaningful — and invert the two 64 x 64 diagonal blocks: X0^T -> Xt0, X1^T -> Xt1.  All 8 waves of the workgroup, after the
// block is complete in LDS and `sy` was cleared by diag_flow_init and a barrier.  No barrier at the end.
static __device__ __forceinline__ void diag_flow2(double* lds, const Diag2Out& out, double* __restrict__ Xt0,
                                                  double* __restrict__ Xt1, int* __restrict__ info, int64_t goff, int wave, int lane,
                                                  double* __restrict__ S0, double* __restrict__ S1)
{
    double* L1 = lds;
    double* L2 = lds + D2_OFF_L2;
    double* H = lds + D2_OFF_H;
    double* invd = lds + D2_OFF_INVD;
    DiagSync* sy = reinterpret_cast<DiagSync*>(lds + D2_OFF_SY);
    double* Xw0 = lds + D2_OFF_XW;
    double* Xw1 = Xw0 + DIAG_XW_DOUBLES;
    if (wave >= 8)
        return;
    // Wave i runs on SIMD i & 3.  The panel wave's SIMD gets no update wave for company: an update wave's blocks are 36+
    // fp64 matrix-core instructions a round, and they slowed the panel wave's fp64 arithmetic (1550 cycles a round against 1120
    // with the W wave — two short bursts in 32 rounds — on that SIMD; tools/diagbench2).  THREE update waves, on the other SIMDs.
    if (wave == 4) {
        __syncthreads();
        flow_w_wave(L1, sy, Xw0, lane, 0);
        flow_w_wave(L2, sy, Xw1, lane, 16);
        flow_x21(sy, Xw0, Xt0, 3, lane, 0); // (the fourth quarters of the X21: the update waves take the others)
        flow_x21(sy, Xw1, Xt1, 3, lane, 16);
        return;
    }
    if (wave >= 1 && wave <= 3) {
        switch (wave) {
        case 1: flow_u2_wave<0>(L1, L2, H, sy, lane); break;
        case 2: flow_u2_wave<1>(L1, L2, H, sy, lane); break;
        default: flow_u2_wave<2>(L1, L2, H, sy, lane); break;
        }
        // the off-diagonal quarters of the two inverses, X21 = -X22 W: a quarter of each per update wave, once they are through
        // (nothing of the chain waits for them: the next workgroup solves in the half-block form)
        flow_x21(sy, Xw0, Xt0, wave - 1, lane, 0);
        flow_x21(sy, Xw1, Xt1, wave - 1, lane, 16);
        return;
    }
    if (wave == 0) {
        double ca[4], cb[4], na[4], nb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ca[e] = L1[lane * XS + e];
            na[e] = L1[lane * XS + 4 + e];
            cb[e] = L1[(NB + lane) * XS + e];
            nb[e] = L1[(NB + lane) * XS + 4 + e];
        }
        __syncthreads();
        FTS2(0, 32);
        FlowP2<31>::run(ca, cb, na, nb, L1, L2, H, invd, sy, lane);
        return;
    }
    __syncthreads();
    if (wave == 5) { // the inverse of block 0: behind panels 0..15, sharing its SIMD with an update wave — it may lag, nothing of the
                     // chain waits for it (the tiles of column c do)
        double S[32];
#pragma unroll
        for (int k = 0; k < 32; ++k)
            S[k] = (lane < 32 && k == lane) ? 1.0 : 0.0;
        FlowX<15, 0>::run(S, L1, invd, Xt0, sy, Xw0, lane & 31, lane >> 5, S0);
        return;
    }
    int bad = 0;
    if (wave == 7) {
        // the tile's rows while there are any (panels 0..15), then the inverse of block 1 — the next chain workgroup waits for its
        // last rows: a wave of its own for it, on a SIMD whose update wave has little left to do by then (one wave doing both
        // inverses in turn finished 8.7 k cycles behind the panel wave)
        FlowS2<15, 1>::run(bad, L1, L2, invd, out, sy, lane);
        double S[32];
#pragma unroll
        for (int k = 0; k < 32; ++k)
            S[k] = (lane < 32 && k == lane) ? 1.0 : 0.0;
        FlowX<15, 16>::run(S, L2, invd + NB, Xt1, sy, Xw1, lane & 31, lane >> 5, S1);
        FTS2(5, 31);
        return;
    }
    FlowS2<31, 0>::run(bad, L1, L2, invd, out, sy, lane);
    if (lane == 0 && bad != 0 && *info == 0)
        *info = (int)(goff + bad);
}
