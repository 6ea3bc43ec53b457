// inv2.hip — K^-1 = L^-T L^-1 (GP::compute_inv_kernel, src/limbo/model/gp.hpp:254-264) by recursion on the factor, as a handful
// of launches whose products have k = 256 .. N/2 (round 5).  Host code: the plan and its execution; the kernels are
// gemm.hip: k_gemm_items / k_gemm_items64 (lists of tile products) and inv.hip: k_inv_panels.
//
// Rounds 1-4 built U = L^-T right-looking over 256-column panels: two launches of k = 256 per panel plus the rank-256
// updates of U U^T — 48 launches at N = 4096, each re-reading its C tiles and most of them under one round of the chip:
// 1.78 ms = 0.33 of the fp64 matrix-core peak for 2 N^3 / 3 flops.  Here:
//
//   leaves   X_p = inv(L_pp) of every 256 x 256 diagonal block (k_inv_panels, one launch): X_p^T into U, X_p into T
//   node (a | c) of the binary tree over the panels, bottom-up, all nodes of one height in the same launches:
//        W   = U_a B^T          B = L[c, a]; U_a upper triangular: tile row i has k >= i only
//        U_b = -W T_c^T         T_c = L_cc^-1 lower triangular: tile column j has k <= j only        -> U[a, c]
//     (T_c, the untransposed inverse of the right child, is what the second product needs as its B operand — both operands
//      of the direct-to-LDS kernel are contiguous along their NON-k index.  A node that is itself a right child therefore
//      leaves its inverse transposed as well: U_b^T by the fold of its tiles, its left child's triangle by tile transposes.)
//   K^-1     = U U^T, lower triangle: tile (i, j) has k >= i
//
// Every product is a list of tiles with its own depth.  A tile's k loop is serial on one CU (17 us per 128 of depth), so
// the k range of a tile is CUT into chunks where one launch would otherwise wait for its deepest tiles.  The chunks of a launch
// are dealt longest-first into `nbins` shares (two resident workgroups per CU), one persistent workgroup each.
// Round 6: the chunks of a tile are ADDED UP BY THE LAUNCH THAT FORMS THEM — chunk 0 goes to the tile, chunk c > 0 to partial
// buffer c, write-through; every chunk then counts in the tile's counter word, and the workgroup that counts LAST reads them
// all back device-wide, adds them in their order (D + P1 + P2 + P3, whoever adds: bitwise the sums of round 5's fold launches),
// stores the tile and, where a later level wants it, its transposed copy (gemm_glds64.h: store_item, fold_item).  Nobody waits
// for anybody.  Round 5 ran a fold launch behind every product launch for this: eight launches of N = 4096's seventeen.
// (Also measured in round 6: the chunks adding to the tile one after the other, each waiting for a flag from the one in front —
// no partial buffers, but the waits line up behind each other, U U^T 461 -> 603 us: profiles/r06_inv_fold_in_epilogue.log.)
// Algorithmic flops are the minimum, 2 N^3 / 3; nothing is zero-filled.
//
// Buffers (all ld x cap, same leading dimension): L (the factor), U (gpe_ctx::dLinv), K^-1 (dKinv), S = [T-forms and W |
// partial 1 | partial 2 | partial 3].  W of node (a | c) lies at [a, c] of S's first buffer, strictly above the diagonal
// blocks that hold the T-forms.
#include <algorithm>
#include <cstdlib>
#include <queue>
#include <vector>

#include "dev.h"

namespace {

constexpr int LEAF = 256, NPART = 3; // (NPART: a tile's k range is cut into at most 1 + NPART chunks)
// tile edge of a launch: 128 where the launch fills the chip with 128 x 128 tiles, 64 below that (the low levels of the tree)
constexpr int TILES128_MIN = 200;
enum Buf { B_L = 0, B_U = 1, B_K = 2, B_S0 = 3, B_P1 = 4, B_P2 = 5, B_P3 = 6, B_NONE = -1 };

struct SymRef {
    int buf = B_NONE;
    int64_t off = 0;
};
struct SymItem {
    SymRef a, b, c, t, d; // (d: the tile itself — c for chunk 0; t: its transposed copy, on every chunk of a cut range)
    int k = 0, neg = 0, mr = 0, nc = 0;
    int seq = 0, nch = 1, slot = 0; // chunk seq of nch of its tile's k range; the tile's counter word (nch > 1)
};
struct SymStep {
    int te = 128; // tile edge of the launch (128 or 64)
    int off = 0, count = 0, bins = 0, bin_off = 0;
    double flops = 0.0;
};
struct SymPlan {
    int prefix_steps = 0; // the product launches of the lowest tree level (inv2_run, part 1)
    int nslots = 0;       // counter words: one per tile whose k range is cut
    std::vector<SymItem> items;
    std::vector<int32_t> bin_start; // per step: bins + 1 entries (absolute item indices)
    std::vector<SymStep> steps;
};

// an unchunked tile product: depth klen from the operands' origins (a multiple of 64; the last tile of a ragged order may end
// short of a whole tile edge), destination d with mr x nc valid entries, optional transposed copy
struct Prod {
    SymRef a, b, d, t;
    int klen = 0, neg = 0, mr = 0, nc = 0;
};

struct Node {
    int lo = 0, hi = 0, mid = 0, height = 0, left = -1, right = -1;
    bool is_right = false;
    bool in_right = false; // this node or one above it is a right child: its block's inverse is wanted untransposed as well
};
int make_tree(std::vector<Node>& nodes, int lo, int hi, bool is_right, bool in_right = false)
{
    Node nd;
    nd.lo = lo;
    nd.hi = hi;
    nd.is_right = is_right;
    nd.in_right = in_right || is_right;
    const int me = (int)nodes.size();
    nodes.push_back(nd);
    if (hi - lo > 1) {
        const int mid = lo + (hi - lo + 1) / 2;
        const int l = make_tree(nodes, lo, mid, false, nd.in_right);
        const int r = make_tree(nodes, mid, hi, true, nd.in_right);
        nodes[me].mid = mid;
        nodes[me].left = l;
        nodes[me].right = r;
        nodes[me].height = 1 + std::max(nodes[l].height, nodes[r].height);
    }
    return me;
}

// one launch of products
void emit(SymPlan& pl, const std::vector<Prod>& prods, int64_t ld, int nbins, double load, int te, bool no_chunk = false)
{
    if (prods.empty())
        return;
    auto units_of = [&](const Prod& p) { return (p.klen + te - 1) / te; };
    int64_t total = 0;
    int lmax = 1;
    for (const Prod& p : prods) {
        total += units_of(p);
        lmax = std::max(lmax, units_of(p));
    }
    // chunk length (in units of one tile edge of depth): no chunk much longer than a share of the launch, at most 1 + NPART
    // chunks per tile, never below 256 of depth (a cut costs a second pass over the tile)
    const double avg = (double)total / (double)nbins;
    int ch = std::max(256 / te, (int)std::min(load * avg + 0.5, 1e9)); // (clamped before the cast: ADVICE r5)
    ch = std::max(ch, (lmax + NPART) / (NPART + 1));
    if (no_chunk) // a batch of >= 4 members: whole k ranges, a workgroup per tile product
        ch = std::max(ch, lmax);
    struct Chunk {
        SymItem it;
        int units, prod;
    };
    std::vector<Chunk> chunks;
    double flops = 0.0;
    for (size_t pi = 0; pi < prods.size(); ++pi) {
        const Prod& p = prods[pi];
        const int units = units_of(p);
        const int nch = (units + ch - 1) / ch;
        const int base = units / nch, rem = units % nch;
        const int slot = nch > 1 ? pl.nslots++ : 0;
        int u0 = 0;
        for (int c = 0; c < nch; ++c) {
            const int u = base + (c < rem ? 1 : 0);
            const int k0 = u0 * te, k1 = std::min((u0 + u) * te, p.klen);
            Chunk q;
            q.it.a = {p.a.buf, p.a.off + (int64_t)k0 * ld};
            q.it.b = {p.b.buf, p.b.off + (int64_t)k0 * ld};
            q.it.c = c == 0 ? p.d : SymRef{B_P1 + c - 1, p.d.off};
            q.it.d = p.d;
            q.it.t = p.t; // (whoever counts last writes it)
            q.it.k = k1 - k0;
            q.it.neg = p.neg;
            q.it.mr = p.mr;
            q.it.nc = p.nc;
            q.it.seq = c;
            q.it.nch = nch;
            q.it.slot = slot;
            q.units = u;
            q.prod = (int)pi;
            chunks.push_back(q);
            u0 += u;
        }
        flops += 2.0 * p.mr * p.nc * (double)p.klen;
    }
    // Shares: longest chunk first into the least loaded share (ties: the lower share, the earlier chunk — deterministic).
    // (Measured and dropped: dealing the tiles that read one operand panel to ONE XCD first, so that its 64 resident
    // workgroups fetch the panel into that XCD's L2 once — U U^T 481 us against 447, the other launches unchanged
    // (profiles/r05_inv2_ab.log): the products sit at the rate of the LDS-fed inner loop, not at the L2's.)
    std::stable_sort(chunks.begin(), chunks.end(), [](const Chunk& x, const Chunk& y) { return x.units > y.units; });
    const int nb = (int)std::min<size_t>((size_t)nbins, chunks.size());
    std::vector<std::vector<int>> bins((size_t)nb);
    typedef std::pair<double, int> Key;
    std::priority_queue<Key, std::vector<Key>, std::greater<Key>> heap;
    for (int b = 0; b < nb; ++b)
        heap.push(Key(0.0, b));
    for (size_t i = 0; i < chunks.size(); ++i) {
        Key k = heap.top();
        heap.pop();
        bins[(size_t)k.second].push_back((int)i);
        heap.push(Key(k.first + chunks[i].units + (te == 128 ? 0.35 : 1.5), k.second)); // (prologue + epilogue of a tile, in units)
    }
    SymStep st;
    st.te = te;
    st.off = (int)pl.items.size();
    st.bins = nb;
    st.bin_off = (int)pl.bin_start.size();
    st.flops = flops;
    for (int b = 0; b < nb; ++b) {
        pl.bin_start.push_back((int32_t)pl.items.size());
        for (int i : bins[(size_t)b])
            pl.items.push_back(chunks[(size_t)i].it);
    }
    pl.bin_start.push_back((int32_t)pl.items.size());
    st.count = (int)pl.items.size() - st.off;
    pl.steps.push_back(st);
}

void build(SymPlan& pl, int64_t N, int64_t ld, int nbins, double load, int members)
{
    // A batched sequence of at least four members fills the chip with whole tiles: no k range is cut and every tile product is a
    // workgroup of its own.
    const bool no_chunk = members >= 4; // (an explicit flag: ADVICE r5 — a magic load of 1e9 overflowed the chunk length's int from N ~ 32768 on)
    if (no_chunk)
        nbins = 1 << 20;
    const int npan = (int)((N + LEAF - 1) / LEAF);
    const int64_t N64 = (N + 63) / 64 * 64; // how far a k range may run: the buffers hold zeros between N and their capacity
    std::vector<Node> nodes;
    make_tree(nodes, 0, npan, false);
    int H = 0;
    for (const Node& n : nodes)
        H = std::max(H, n.height);
    auto at = [&](int64_t row, int64_t col) { return row + col * ld; };
    // A ragged order: the last panel is the only partial one and always sits in a RIGHT child, so every left block is whole
    // panels; tiles of the last row / column strip carry fewer valid rows / columns (mr, nc), k ranges over the right block end
    // at N rounded up to 64.
    for (int h = 1; h <= H; ++h) {
        int64_t tiles128 = 0;
        for (const Node& n : nodes)
            if (n.height == h)
                tiles128 += (int64_t)(n.mid - n.lo) * (n.hi - n.mid) * (LEAF / 128) * (LEAF / 128);
        const int TILE = tiles128 >= TILES128_MIN ? 128 : 64;
        std::vector<Prod> w_prods, u_prods;
        for (const Node& n : nodes) {
            if (n.height != h)
                continue;
            const int64_t a0 = (int64_t)n.lo * LEAF, am = (int64_t)n.mid * LEAF;
            const int64_t ce = std::min<int64_t>((int64_t)n.hi * LEAF, N), ce64 = std::min<int64_t>((int64_t)n.hi * LEAF, N64);
            const int ta = (int)((am - a0) / TILE), tc = (int)((ce - am + TILE - 1) / TILE);
            for (int i = 0; i < ta; ++i)
                for (int j = 0; j < tc; ++j) {
                    const int ncj = (int)std::min<int64_t>(TILE, ce - (am + (int64_t)j * TILE)); // valid columns of the strip
                    // W[i, j] = sum_{k >= i} U_a[i, k] B[j, k]
                    Prod w;
                    w.a = {B_U, at(a0 + (int64_t)i * TILE, a0 + (int64_t)i * TILE)};
                    w.b = {B_L, at(am + (int64_t)j * TILE, a0 + (int64_t)i * TILE)};
                    w.d = {B_S0, at(a0 + (int64_t)i * TILE, am + (int64_t)j * TILE)};
                    w.klen = (ta - i) * TILE;
                    w.mr = TILE;
                    w.nc = ncj;
                    w_prods.push_back(w);
                    // U_b[i, j] = -sum_{k <= j} W[i, k] T_c[j, k]
                    Prod u;
                    u.a = {B_S0, at(a0 + (int64_t)i * TILE, am)};
                    u.b = {B_S0, at(am + (int64_t)j * TILE, am)};
                    u.d = {B_U, at(a0 + (int64_t)i * TILE, am + (int64_t)j * TILE)};
                    u.klen = (int)std::min<int64_t>((int64_t)(j + 1) * TILE, ce64 - am);
                    u.neg = 1;
                    u.mr = TILE;
                    u.nc = ncj;
                    // A right child is its parent's T_c, untransposed, whole: U_b^T goes out with U_b — from this node and from every
                    // node inside a right child's block (the leaves have their T-forms from k_inv_panels).  (Round 5 transposed the
                    // inner ones later, tile by tile, in a fold launch behind the parent's products.)
                    if (n.in_right)
                        u.t = {B_S0, at(am + (int64_t)j * TILE, a0 + (int64_t)i * TILE)};
                    u_prods.push_back(u);
                }
        }
        emit(pl, w_prods, ld, nbins, load, TILE, no_chunk);
        emit(pl, u_prods, ld, nbins, load, TILE, no_chunk);
        if (h == 1) // (round 6: both launches of the lowest level — k_inv_panels in front of them went 73 -> 44 us, the second one fits
            // beside the sweep too: profiles/r06_inv_panels.log)
            pl.prefix_steps = (int)pl.steps.size();
    }
    // K^-1[i, j] = sum_{k >= i} U[i, k] U[j, k], i >= j
    std::vector<Prod> k_prods;
    constexpr int TILE = 128;
    const int nt = (int)((N + TILE - 1) / TILE);
    for (int i = 0; i < nt; ++i)
        for (int j = 0; j <= i; ++j) {
            Prod p;
            p.a = {B_U, at((int64_t)i * TILE, (int64_t)i * TILE)};
            p.b = {B_U, at((int64_t)j * TILE, (int64_t)i * TILE)};
            p.d = {B_K, at((int64_t)i * TILE, (int64_t)j * TILE)};
            p.klen = (int)(N64 - (int64_t)i * TILE);
            p.mr = (int)std::min<int64_t>(TILE, N - (int64_t)i * TILE);
            p.nc = (int)std::min<int64_t>(TILE, N - (int64_t)j * TILE);
            k_prods.push_back(p);
        }
    emit(pl, k_prods, ld, nbins, load, TILE, no_chunk);
}

// two resident workgroups per CU (74 KB of LDS each); chunks about as long as a share.  Measured around it
// (profiles/r05_inv2_ab.log): 256 shares 1.305 ms against 1.288 for K^-1 of N = 4096, chunks of half / 0.7 / twice a share
// 1.270 / 1.283 / 1.446.
constexpr int PLAN_BINS = 512;
constexpr double PLAN_LOAD = 1.0;

} // namespace

struct Inv2Plan {
    int64_t N = 0, ld = 0, pstride = 0;
    int members = 1;
    const double* L = nullptr;
    double *U = nullptr, *K = nullptr, *S = nullptr;
    std::vector<SymStep> steps;
    int prefix_steps = 0;
    char* dev = nullptr;
    GemmItem* dItems = nullptr;
    int32_t* dBins = nullptr;
    int* dFlags = nullptr; // nslots counter words per member: the chunks of a cut tile count in them (dev.h: GemmItem)
    int nslots = 0;
    double flops = 0.0;
};

bool inv2_supported(int64_t N)
{
    static const int on = getenv("GPE_INV2") ? atoi(getenv("GPE_INV2")) : 1;
    static const int64_t min_n = getenv("GPE_INV2_MIN_N") ? atoll(getenv("GPE_INV2_MIN_N")) : 1024;
    return on && N >= min_n;
}
int inv2_partials() { return NPART; }

void inv2_plan_free(Inv2Plan* p)
{
    if (!p)
        return;
    if (p->dev)
        (void)hipFree(p->dev);
    delete p;
}

Inv2Plan* inv2_plan_get(Inv2Plan* old, int64_t N, int64_t ld, const double* L, double* U, double* Kinv, double* S, int64_t pstride, int members,
                        bool* rebuilt)
{
    if (rebuilt)
        *rebuilt = false;
    if (old && old->N == N && old->ld == ld && old->L == L && old->U == U && old->K == Kinv && old->S == S && old->pstride == pstride
        && old->members == members) // (the members' flag words: one set each)
        return old;
    if (rebuilt)
        *rebuilt = true;
    inv2_plan_free(old);
    SymPlan sp;
    build(sp, N, ld, PLAN_BINS, PLAN_LOAD, members);
    Inv2Plan* p = new Inv2Plan;
    p->N = N;
    p->ld = ld;
    p->pstride = pstride;
    p->members = members;
    p->L = L;
    p->U = U;
    p->K = Kinv;
    p->S = S;
    p->steps = sp.steps;
    p->prefix_steps = sp.prefix_steps;
    double* base[7] = {const_cast<double*>(L), U, Kinv, S, S + pstride, S + 2 * pstride, S + 3 * pstride};
    auto res = [&](const SymRef& r) -> double* { return r.buf == B_NONE ? nullptr : base[r.buf] + r.off; };
    std::vector<GemmItem> gi(sp.items.size());
    for (size_t i = 0; i < gi.size(); ++i) {
        const SymItem& it = sp.items[i];
        gi[i].A = res(it.a);
        gi[i].B = res(it.b);
        gi[i].C = res(it.c);
        gi[i].T = res(it.t);
        gi[i].D = res(it.d);
        gi[i].P1 = it.nch > 1 ? base[B_P1] + it.d.off : nullptr;
        gi[i].k = it.k;
        gi[i].flags = (it.neg ? GEMM_ITEM_NEG : 0) | (it.mr << 8) | (it.nc << 16);
        gi[i].slot = it.slot;
        gi[i].nch = it.nch;
    }
    p->nslots = std::max(sp.nslots, 1);
    const size_t b0 = sizeof(GemmItem) * gi.size(), b1 = sizeof(int) * (size_t)p->nslots * (size_t)std::max(members, 1),
                 b2 = sizeof(int32_t) * sp.bin_start.size();
    const size_t o1 = (b0 + 255) / 256 * 256, o2 = o1 + (b1 + 255) / 256 * 256;
    if (hipMalloc(&p->dev, o2 + b2 + 256) != hipSuccess) {
        delete p;
        return nullptr;
    }
    p->dItems = (GemmItem*)p->dev;
    p->dFlags = (int*)(p->dev + o1);
    p->dBins = (int32_t*)(p->dev + o2);
    bool ok = hipMemcpy(p->dItems, gi.data(), b0, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(p->dFlags, 0, b1) == hipSuccess; // (a run adds exactly nch to the word of a tile cut into nch chunks)
    ok = ok && hipMemcpy(p->dBins, sp.bin_start.data(), b2, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        inv2_plan_free(p);
        return nullptr;
    }
    for (const SymStep& st : p->steps)
        p->flops += st.flops;
    return p;
}

double inv2_flops(const Inv2Plan* p) { return p ? p->flops : 0.0; }

// part 0: everything.  part 1: the leaves and W of the lowest tree level only — they depend on nothing but the factor, are
// bound by launch latency (128 workgroups, then 32 tile products) and occupy half / an eighth of the chip: gpe_hp_objective
// runs them on the second stream beside the backward sweep of alpha (110 us against its 116).  part 2: the rest.
void inv2_run(hipStream_t s, Inv2Plan* p, const double* Xt_all, int part)
{
    // X_p = inv(L_pp): as it is into the T-form buffer, transposed into U (zeros on the other side of either diagonal)
    if (part != 2)
        launch_inv_panels(s, p->L, p->ld, p->N, LEAF, Xt_all, p->S, p->ld, p->U, p->ld);
    const size_t first = part == 2 ? (size_t)p->prefix_steps : 0, last = part == 1 ? (size_t)p->prefix_steps : p->steps.size();
    for (size_t i = first; i < last; ++i) {
        const SymStep& st = p->steps[i];
        launch_gemm_items(s, p->dItems, p->dBins + st.bin_off, st.bins, p->ld, st.te, p->dFlags, p->nslots, p->pstride);
    }
}

// test hook (gpe_debug_inv_plan): the plan for order N, leading dimension ld; one row of 16 int64 per tile product, in launch
// order, share by share:
//   { step, 0 | 2 (128 x 128 | 64 x 64 tiles), A buf, A off, B buf, B off, C buf, C off, k, neg | share << 1, valid rows, valid
//     columns, chunk number | chunks of the tile << 8, counter word, T buf | -1, T off }   (C: the tile for chunk 0, partial
//     buffer c at the tile's offset for chunk c; T on every chunk of a cut range: the one that counts last writes it)
// buffers: 0 L, 1 U, 2 K^-1, 3 T-forms / W, 4..6 partials.  Returns the number of rows (also when out is too small).
// nbins < 0 asks for the plan of a batched sequence of -nbins members.
int inv2_debug_plan(int64_t N, int64_t ld, int nbins, int load_pct, int64_t* out, int64_t cap_rows)
{
    if (N <= 0 || ld < (N + 63) / 64 * 64)
        return -1;
    SymPlan sp;
    // nbins < 0: the plan of a BATCH of -nbins members (>= 4: no k range is cut, a workgroup per tile product)
    build(sp, N, ld, nbins > 0 ? nbins : PLAN_BINS, load_pct > 0 ? load_pct / 100.0 : PLAN_LOAD, nbins < 0 ? -nbins : 1);
    int64_t row = 0;
    for (size_t s = 0; s < sp.steps.size(); ++s) {
        const SymStep& st = sp.steps[s];
        for (int i = 0; i < st.count; ++i, ++row) {
            if (row >= cap_rows || !out)
                continue;
            int64_t* o = out + row * 16;
            const SymItem& it = sp.items[(size_t)(st.off + i)];
            int64_t bin = 0; // the share (= workgroup of the launch) this product belongs to
            while (bin + 1 < st.bins && sp.bin_start[(size_t)(st.bin_off + bin + 1)] <= st.off + i)
                ++bin;
            const int64_t v[16] = {(int64_t)s, st.te == 64 ? 2 : 0, it.a.buf, it.a.off, it.b.buf, it.b.off, it.c.buf, it.c.off, it.k,
                                   it.neg | (bin << 1), it.mr, it.nc, it.seq | ((int64_t)it.nch << 8), it.slot, it.t.buf, it.t.off};
            std::copy(v, v + 16, o);
        }
    }
    return (int)row;
}
