/*
 * gp_oracle.c — CPU restatement of limbo's GP posterior path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this
 * library.  The product (libgpengine.so) never links, loads or calls it.
 *
 * Parity status: the reference (resibots/limbo) cannot be compiled in this image (Eigen3
 * and Boost headers are absent, no network) and its test-suite holds no golden vectors for
 * this path.  The oracle is therefore pinned against
 *   (1) the closed-form known answers of src/tests/test_kernel.cpp:196-224,
 *   (2) an independent numpy/scipy(LAPACK) restatement (oracle/np_oracle.py), and
 *   (3) 50-digit mpmath ground truth at N <= 48 (oracle/make_golden.py -> tests/golden/),
 * and the property tests of src/tests/test_gp.cpp re-expressed in tests/.  Bitwise parity
 * with "Eigen" is undefined (un-pinned, swappable LLT backend) — parity is to tolerance.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Exported symbols mirror include/gpe.h with the prefix `orc_` so the same ctypes wrapper
 * drives both libraries.
 *
 * Arithmetic: plain IEEE double, like the reference (Eigen::MatrixXd), one long double
 * accumulator for logdet (gp.hpp:274).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

enum { K_SE_ARD = 0, K_MATERN52 = 1, K_MATERN32 = 2, K_EXP = 3, K_HOST_K = 4 };
#define MAX_THETA 64
#define MAX_D_LAM 64

typedef struct orc_ctx {
    int64_t N, cap;
    int D, P;
    double* X;        /* N x D row-major  (the std::vector<VectorXd> _samples, gp.hpp:520) */
    double* obs_mean; /* N x P col-major  (_obs_mean, gp.hpp:523) */
    double* K;        /* N x N col-major, ld = N (_kernel, gp.hpp:528) */
    double* L;        /* N x N col-major, upper zero (_matrixL, gp.hpp:530) */
    double* alpha;    /* N x P (_alpha, gp.hpp:525) */
    double* Kinv;     /* N x N (_inv_kernel) */
    int inv_ok;       /* _inv_kernel_updated, gp.hpp:533 */
    int have_L;
    int kind, n_theta;
    double theta[MAX_THETA];
    double noise;
    int host_K;
    char err[256];
} orc_ctx;
typedef orc_ctx* orc_handle;

/* ------------------------------------------------------------------------- */
/* kernel functors                                                            */
/* ------------------------------------------------------------------------- */

/* SquaredExpARD::kernel, k = 0 branch: src/limbo/kernel/squared_exp_ard.hpp:138-151;
 * parameters: set_params :96-105 (ell_d = exp(p_d), sf2 = exp(2 p_last)). */
static double k_se_ard(const double* x1, const double* x2, int D, const double* th)
{
    double z = 0.0;
    for (int d = 0; d < D; ++d) {
        double q = (x1[d] - x2[d]) / exp(th[d]); /* cwiseQuotient(_ell) :148 */
        z += q * q;                              /* squaredNorm() */
    }
    return exp(2.0 * th[D]) * exp(-0.5 * z); /* :150 */
}

/* SquaredExpARD with k > 0 columns of Lambda: squared_exp_ard.hpp:142-146.  Parameter vector (:96-105):
 * [log ell_1..log ell_D | Lambda(:,0) | .. | Lambda(:,k-1) | log sigma_f], Lambda NOT in log-space.
 * Restated literally: M = Lambda Lambda^T, M.diagonal() += ell^-2, z = d^T M d. */
static double se_ard_lam_z(const double* x1, const double* x2, int D, int k, const double* th)
{
    double M[MAX_D_LAM * MAX_D_LAM], d[MAX_D_LAM], Md[MAX_D_LAM];
    for (int a = 0; a < D; ++a)
        for (int b = 0; b < D; ++b) {
            double s = 0.0;
            for (int j = 0; j < k; ++j)
                s += th[(j + 1) * D + a] * th[(j + 1) * D + b]; /* _A(i, j) = p((j + 1) * dim + i)  :102 */
            M[a + b * D] = s;
        }
    for (int a = 0; a < D; ++a) {
        double ell = exp(th[a]);
        double inv = 1.0 / ell;
        M[a + a * D] += inv * inv; /* _ell.array().inverse().square()  :144 */
        d[a] = x1[a] - x2[a];
    }
    for (int b = 0; b < D; ++b) { /* (x1 - x2)^T * K : a row vector ... */
        double s = 0.0;
        for (int a = 0; a < D; ++a)
            s += d[a] * M[a + b * D];
        Md[b] = s;
    }
    double z = 0.0; /* ... times (x1 - x2) */
    for (int b = 0; b < D; ++b)
        z += Md[b] * d[b];
    return z;
}
static double k_se_ard_lam(const double* x1, const double* x2, int D, int k, const double* th)
{
    return exp(2.0 * th[D + D * k]) * exp(-0.5 * se_ard_lam_z(x1, x2, D, k, th)); /* :150 */
}

/* MaternFiveHalves::kernel: src/limbo/kernel/matern_five_halves.hpp:104-113; params :97-102 */
static double k_matern52(const double* x1, const double* x2, int D, const double* th)
{
    double l = exp(th[0]), sf2 = exp(2.0 * th[1]);
    double s = 0.0;
    for (int d = 0; d < D; ++d) {
        double q = x1[d] - x2[d];
        s += q * q;
    }
    double d_ = sqrt(s); /* (v1 - v2).norm() :106 */
    double d_sq = d_ * d_;
    double l_sq = l * l;
    double term1 = sqrt(5.0) * d_ / l;
    double term2 = 5.0 * d_sq / (3.0 * l_sq);
    return sf2 * (1 + term1 + term2) * exp(-term1);
}

/* MaternThreeHalves::kernel: src/limbo/kernel/matern_three_halves.hpp:101-107 */
static double k_matern32(const double* x1, const double* x2, int D, const double* th)
{
    double l = exp(th[0]), sf2 = exp(2.0 * th[1]);
    double s = 0.0;
    for (int d = 0; d < D; ++d) {
        double q = x1[d] - x2[d];
        s += q * q;
    }
    double term = sqrt(3.0) * sqrt(s) / l;
    return sf2 * (1 + term) * exp(-term);
}

/* Exp::kernel: src/limbo/kernel/exp.hpp:97-102 */
static double k_exp(const double* x1, const double* x2, int D, const double* th)
{
    double l = exp(th[0]), sf2 = exp(2.0 * th[1]);
    double s = 0.0;
    for (int d = 0; d < D; ++d) {
        double q = x1[d] - x2[d];
        s += q * q;
    }
    double r = s / (l * l);
    return sf2 * exp(-0.5 * r);
}

/* number of Lambda columns implied by the parameter count (params_size, squared_exp_ard.hpp:94) */
static int se_ard_k(int D, int nth) { return (D > 0 && nth > D + 1) ? (nth - 1) / D - 1 : 0; }

static double k_eval(int kind, const double* x1, const double* x2, int D, const double* th, int nth)
{
    switch (kind) {
    case K_SE_ARD:
        if (se_ard_k(D, nth) > 0)
            return k_se_ard_lam(x1, x2, D, se_ard_k(D, nth), th);
        return k_se_ard(x1, x2, D, th);
    case K_MATERN52: return k_matern52(x1, x2, D, th);
    case K_MATERN32: return k_matern32(x1, x2, D, th);
    default: return k_exp(x1, x2, D, th);
    }
}

/* BaseKernel::operator(): src/limbo/kernel/kernel.hpp:81-84 — noise + 1e-8 only when the
 * two sample INDICES are equal; query-time calls use the defaults i=-1, j=-2 (no noise). */
static double k_with_noise(const orc_ctx* c, const double* x1, const double* x2, int64_t i, int64_t j)
{
    return k_eval(c->kind, x1, x2, c->D, c->theta, c->n_theta) + ((i == j) ? c->noise + 1e-8 : 0.0);
}

/* Kernel::gradient wrt the log-hyper-parameters (without the noise entry):
 * SE-ARD   squared_exp_ard.hpp:127-135 (k = 0 branch)
 * Matern52 matern_five_halves.hpp:115-133
 * Matern32 matern_three_halves.hpp:109-121
 * Exp      exp.hpp:104-113 */
static void k_grad(int kind, const double* x1, const double* x2, int D, const double* th, int nth, double* g)
{
    if (kind == K_SE_ARD && se_ard_k(D, nth) > 0) { /* squared_exp_ard.hpp:109-126 */
        const int k = se_ard_k(D, nth);
        double z = se_ard_lam_z(x1, x2, D, k, th);
        double kv = exp(2.0 * th[D + D * k]) * exp(-0.5 * z);
        for (int d = 0; d < D; ++d) {
            double q = (x1[d] - x2[d]) / exp(th[d]);
            g[d] = q * q * kv; /* :116 */
        }
        for (int j = 0; j < k; ++j) {
            double proj = 0.0; /* (x1 - x2)^T _A.col(j) */
            for (int d = 0; d < D; ++d)
                proj += (x1[d] - x2[d]) * th[(j + 1) * D + d];
            for (int d = 0; d < D; ++d)
                g[(j + 1) * D + d] = -proj * (x1[d] - x2[d]) * kv; /* :119-120 */
        }
        g[D + D * k] = 2 * kv; /* :123 */
        return;
    }
    if (kind == K_SE_ARD) {
        double zs = 0.0;
        for (int d = 0; d < D; ++d) {
            double q = (x1[d] - x2[d]) / exp(th[d]);
            g[d] = q * q;
            zs += g[d];
        }
        double k = exp(2.0 * th[D]) * exp(-0.5 * zs);
        for (int d = 0; d < D; ++d)
            g[d] *= k;
        g[D] = 2 * k;
        return;
    }
    double l = exp(th[0]), sf2 = exp(2.0 * th[1]);
    double s = 0.0;
    for (int d = 0; d < D; ++d) {
        double q = x1[d] - x2[d];
        s += q * q;
    }
    if (kind == K_MATERN52) {
        double d_ = sqrt(s), d_sq = d_ * d_, l_sq = l * l;
        double term1 = sqrt(5.0) * d_ / l;
        double term2 = 5.0 * d_sq / (3.0 * l_sq);
        double r = exp(-term1);
        g[0] = sf2 * (r * term1 * (1 + term1 + term2) + (-term1 - 2. * term2) * r);
        g[1] = 2 * sf2 * (1 + term1 + term2) * r;
    }
    else if (kind == K_MATERN32) {
        double term = sqrt(3.0) * sqrt(s) / l;
        double r = exp(-term);
        g[0] = sf2 * (-term * r + (1 + term) * term * r);
        g[1] = 2 * sf2 * (1 + term) * r;
    }
    else {
        double r = s / (l * l);
        double k = sf2 * exp(-0.5 * r);
        g[0] = r * k;
        g[1] = 2 * k;
    }
}

/* ------------------------------------------------------------------------- */
/* dense linear algebra (what the reference delegates to Eigen)               */
/* ------------------------------------------------------------------------- */
#define A_(M, i, j, ld) (M)[(int64_t)(i) + (int64_t)(j) * (ld)]

/* unblocked lower Cholesky of the nb x nb block at A (in place); returns 0 or 1-based pivot */
static int64_t potf2(double* A, int64_t nb, int64_t ld)
{
    for (int64_t j = 0; j < nb; ++j) {
        double d = A_(A, j, j, ld);
        for (int64_t k = 0; k < j; ++k)
            d -= A_(A, j, k, ld) * A_(A, j, k, ld);
        if (!(d > 0.0))
            return j + 1;
        d = sqrt(d);
        A_(A, j, j, ld) = d;
        for (int64_t i = j + 1; i < nb; ++i) {
            double s = A_(A, i, j, ld);
            for (int64_t k = 0; k < j; ++k)
                s -= A_(A, i, k, ld) * A_(A, j, k, ld);
            A_(A, i, j, ld) = s / d;
        }
    }
    return 0;
}

/* Replaces Eigen::LLT<MatrixXd>(K).matrixL() at src/limbo/model/gp.hpp:565.  Blocked
 * right-looking (the structure of Eigen 3.3's LLT.h): diagonal block, panel solve,
 * symmetric rank-nb trailing update.  Lower triangle only is read. */
static int64_t cholesky_lower(double* A, int64_t n, int64_t ld)
{
    const int64_t NB = 96;
    for (int64_t k = 0; k < n; k += NB) {
        int64_t nb = (n - k < NB) ? n - k : NB;
        int64_t info = potf2(&A_(A, k, k, ld), nb, ld);
        if (info)
            return k + info;
        int64_t m = n - k - nb; /* rows below */
        if (m <= 0)
            break;
        double* A21 = &A_(A, k + nb, k, ld);
        /* A21 <- A21 * L11^-T : column by column, contiguous in the row index */
        for (int64_t j = 0; j < nb; ++j) {
            double* cj = A21 + j * ld;
            for (int64_t p = 0; p < j; ++p) {
                double l = A_(A, k + j, k + p, ld);
                const double* cp = A21 + p * ld;
                for (int64_t i = 0; i < m; ++i)
                    cj[i] -= l * cp[i];
            }
            double inv = A_(A, k + j, k + j, ld);
            for (int64_t i = 0; i < m; ++i)
                cj[i] /= inv;
        }
        /* A22 <- A22 - A21 A21^T (lower part) */
        for (int64_t j = 0; j < m; ++j) {
            double* cj = &A_(A, k + nb + j, k + nb + j, ld); /* column j, from the diagonal */
            for (int64_t p = 0; p < nb; ++p) {
                double l = A21[j + p * ld];
                const double* cp = A21 + p * ld + j;
                int64_t len = m - j;
                for (int64_t i = 0; i < len; ++i)
                    cj[i] -= l * cp[i];
            }
        }
    }
    return 0;
}

/* x <- L^-1 x, L lower (triangularView<Lower>().solve, gp.hpp:608-609, :620) */
static void trsv_lower(const double* L, int64_t n, int64_t ld, double* x)
{
    for (int64_t j = 0; j < n; ++j) {
        double xj = x[j] / A_(L, j, j, ld);
        x[j] = xj;
        const double* c = &A_(L, 0, j, ld);
        for (int64_t i = j + 1; i < n; ++i)
            x[i] -= xj * c[i];
    }
}
/* x <- L^-T x (triang.adjoint().solveInPlace, gp.hpp:610) */
static void trsv_lower_t(const double* L, int64_t n, int64_t ld, double* x)
{
    for (int64_t j = n - 1; j >= 0; --j) {
        const double* c = &A_(L, 0, j, ld);
        double s = x[j];
        for (int64_t i = j + 1; i < n; ++i)
            s -= c[i] * x[i];
        x[j] = s / c[j];
    }
}

/* ------------------------------------------------------------------------- */
/* handle plumbing                                                            */
/* ------------------------------------------------------------------------- */
static void free_mats(orc_ctx* c)
{
    free(c->K);
    free(c->L);
    free(c->alpha);
    free(c->Kinv);
    c->K = c->L = c->alpha = c->Kinv = NULL;
    c->inv_ok = 0;
    c->have_L = 0;
}

int orc_create(int device_id, orc_handle* out)
{
    (void)device_id;
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    if (!c)
        return -4;
    c->kind = K_SE_ARD;
    c->noise = 0.01; /* defaults::kernel::noise, kernel.hpp:57 */
    *out = c;
    return 0;
}

int orc_destroy(orc_handle c)
{
    if (!c)
        return -1;
    free_mats(c);
    free(c->X);
    free(c->obs_mean);
    free(c);
    return 0;
}

static double* dup_d(const double* p, int64_t n)
{
    if (!p || n <= 0)
        return NULL;
    double* q = (double*)malloc(sizeof(double) * (size_t)n);
    memcpy(q, p, sizeof(double) * (size_t)n);
    return q;
}

/* value semantics of limbo::model::GP (copy-constructed per objective evaluation,
 * src/limbo/model/gp/kernel_lf_opt.hpp:79) */
int orc_clone(orc_handle s, orc_handle* out)
{
    orc_ctx* c = (orc_ctx*)malloc(sizeof(orc_ctx));
    memcpy(c, s, sizeof(orc_ctx));
    c->X = dup_d(s->X, s->N * s->D);
    c->obs_mean = dup_d(s->obs_mean, s->N * s->P);
    c->K = dup_d(s->K, s->N * s->N);
    c->L = dup_d(s->L, s->N * s->N);
    c->alpha = dup_d(s->alpha, s->N * s->P);
    c->Kinv = dup_d(s->Kinv, s->N * s->N);
    *out = c;
    return 0;
}

/* placement has no meaning on the CPU: one "device" */
int orc_clone_to(orc_handle s, int device_id, orc_handle* out) { (void)device_id; return orc_clone(s, out); }
int orc_device_count(int* n) { *n = 1; return 0; }
int orc_get_device(orc_handle c, int* d) { (void)c; *d = 0; return 0; }

const char* orc_last_error(orc_handle c) { return c ? c->err : "null handle"; }
const char* orc_version(void) { return "oracle-1"; }

/* GP::compute up to (not including) the kernel: src/limbo/model/gp.hpp:88-111.  The mean
 * functor (gp.hpp:537-548) is evaluated by the caller; obs_mean = Y - m(X) arrives here. */
int orc_set_data(orc_handle c, const double* X, int64_t N, int D, const double* obs_mean, int P)
{
    if (!c || !X || !obs_mean || N <= 0 || D <= 0 || P <= 0)
        return -1;
    free_mats(c);
    free(c->X);
    free(c->obs_mean);
    c->N = N;
    c->D = D;
    c->P = P;
    c->X = dup_d(X, N * D);
    c->obs_mean = dup_d(obs_mean, N * P);
    c->host_K = 0;
    return 0;
}
int orc_set_data_device(orc_handle c, const double* X, int64_t N, int D, const double* om, int P)
{
    return orc_set_data(c, X, N, D, om, P);
}

/* BaseKernel::set_h_params: src/limbo/kernel/kernel.hpp:116-123 */
int orc_set_kernel(orc_handle c, int kind, const double* th, int n_theta, double noise)
{
    if (!c || n_theta > MAX_THETA || n_theta < 0)
        return -1;
    c->kind = kind;
    c->n_theta = n_theta;
    if (th)
        memcpy(c->theta, th, sizeof(double) * (size_t)n_theta);
    c->noise = noise;
    return 0;
}

int orc_set_K_host(orc_handle c, const double* K, int64_t ldk)
{
    if (!c || !K || c->N <= 0 || ldk < c->N)
        return -1;
    free(c->K);
    c->K = (double*)malloc(sizeof(double) * (size_t)(c->N * c->N));
    for (int64_t j = 0; j < c->N; ++j)
        memcpy(c->K + j * c->N, K + j * ldk, sizeof(double) * (size_t)c->N);
    c->host_K = 1;
    return 0;
}

/* GP::_compute_alpha: src/limbo/model/gp.hpp:605-611 */
static void compute_alpha(orc_ctx* c)
{
    int64_t n = c->N;
    free(c->alpha);
    c->alpha = dup_d(c->obs_mean, n * c->P);
    for (int p = 0; p < c->P; ++p) {
        trsv_lower(c->L, n, n, c->alpha + p * n);
        trsv_lower_t(c->L, n, n, c->alpha + p * n);
    }
}

static void build_K(orc_ctx* c)
{
    int64_t n = c->N;
    if (c->host_K)
        return;
    free(c->K);
    c->K = (double*)malloc(sizeof(double) * (size_t)(n * n));
    /* gp.hpp:556-558 lower triangle, :560-562 mirror */
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j <= i; ++j)
            A_(c->K, i, j, n) = k_with_noise(c, c->X + i * c->D, c->X + j * c->D, i, j);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < i; ++j)
            A_(c->K, j, i, n) = A_(c->K, i, j, n);
}

/* GP::_compute_full_kernel: src/limbo/model/gp.hpp:550-571 */
int orc_compute(orc_handle c)
{
    if (!c || c->N <= 0)
        return -2;
    int64_t n = c->N;
    build_K(c);
    free(c->L);
    c->L = dup_d(c->K, n * n);
    int64_t info = cholesky_lower(c->L, n, n); /* :565 (no .info() check in the reference) */
    for (int64_t j = 1; j < n; ++j)            /* matrixL() has a zero upper triangle */
        for (int64_t i = 0; i < j; ++i)
            A_(c->L, i, j, n) = 0.0;
    c->have_L = 1;
    compute_alpha(c); /* :567 */
    c->inv_ok = 0;    /* :570 */
    return (int)info;
}

/* GP::recompute(update_obs_mean, false): src/limbo/model/gp.hpp:241-252 */
int orc_update_alpha(orc_handle c, const double* obs_mean)
{
    if (!c || !c->have_L)
        return -2;
    if (obs_mean) {
        free(c->obs_mean);
        c->obs_mean = dup_d(obs_mean, c->N * c->P);
    }
    compute_alpha(c);
    return 0;
}

/* GP::add_sample + _compute_incremental_kernel: src/limbo/model/gp.hpp:126-152, :573-603 */
int orc_add_sample(orc_handle c, const double* x, int D_, const double* obs_mean, int P_)
{
    if (!c || !x || !obs_mean)
        return -1;
    if (c->N == 0) { /* :128-137 */
        c->D = D_;
        c->P = P_;
    }
    else if (D_ != c->D || P_ != c->P) /* :139-140 */
        return -1;
    if (c->N > 0 && !c->have_L)
        return -2;
    int64_t n0 = c->N, n = n0 + 1;
    int D = c->D, P = c->P;
    double* X = (double*)malloc(sizeof(double) * (size_t)(n * D));
    if (n0)
        memcpy(X, c->X, sizeof(double) * (size_t)(n0 * D));
    memcpy(X + n0 * D, x, sizeof(double) * (size_t)D);
    free(c->X);
    c->X = X;
    free(c->obs_mean);
    c->obs_mean = dup_d(obs_mean, n * P);
    /* conservativeResize of K and L (:581, :588) */
    double* K = (double*)calloc((size_t)(n * n), sizeof(double));
    double* L = (double*)calloc((size_t)(n * n), sizeof(double));
    for (int64_t j = 0; j < n0; ++j) {
        if (c->K)
            memcpy(K + j * n, c->K + j * n0, sizeof(double) * (size_t)n0);
        memcpy(L + j * n, c->L + j * n0, sizeof(double) * (size_t)n0);
    }
    free(c->K);
    free(c->L);
    c->K = K;
    c->L = L;
    c->N = n;
    for (int64_t i = 0; i < n; ++i) { /* :583-586 */
        A_(K, i, n - 1, n) = k_with_noise(c, c->X + i * D, c->X + (n - 1) * D, i, n - 1);
        A_(K, n - 1, i, n) = A_(K, i, n - 1, n);
    }
    double L_j;
    for (int64_t j = 0; j < n - 1; ++j) { /* :591-594 */
        double dot = 0.0;
        for (int64_t k = 0; k < j; ++k)
            dot += A_(L, j, k, n) * A_(L, n - 1, k, n);
        L_j = A_(K, n - 1, j, n) - dot;
        A_(L, n - 1, j, n) = L_j / A_(L, j, j, n);
    }
    double dot = 0.0;
    for (int64_t k = 0; k < n - 1; ++k)
        dot += A_(L, n - 1, k, n) * A_(L, n - 1, k, n);
    L_j = A_(K, n - 1, n - 1, n) - dot; /* :596 */
    A_(L, n - 1, n - 1, n) = sqrt(L_j); /* :597 */
    c->have_L = 1;
    compute_alpha(c); /* :599 */
    c->inv_ok = 0;    /* :602 */
    free(c->Kinv);
    c->Kinv = NULL;
    return 0;
}

/* GP::compute_log_lik: src/limbo/model/gp.hpp:267-282 (P-quirk: logdet and n log 2pi are
 * NOT multiplied by P) */
int orc_log_lik(orc_handle c, double* out)
{
    if (!c || !c->have_L || !out)
        return -2;
    int64_t n = c->N;
    double s = 0.0; /* Eigen's .array().log().sum() is a double reduction ... */
    for (int64_t i = 0; i < n; ++i)
        s += log(A_(c->L, i, i, n));
    long double logdet = 2 * s; /* ... widened afterwards (:274) */
    double a = 0.0;
    for (int p = 0; p < c->P; ++p)
        for (int64_t i = 0; i < n; ++i)
            a += c->obs_mean[i + p * n] * c->alpha[i + p * n]; /* trace(obs_mean^T alpha) :276 */
    *out = (double)(-0.5 * a - 0.5 * logdet - 0.5 * n * log(2 * M_PI));
    return 0;
}

/* GP::compute_inv_kernel: src/limbo/model/gp.hpp:254-264 (I -> L\I -> L^T\.) */
int orc_compute_inv_kernel(orc_handle c)
{
    if (!c || !c->have_L)
        return -2;
    int64_t n = c->N;
    free(c->Kinv);
    c->Kinv = (double*)calloc((size_t)(n * n), sizeof(double));
    for (int64_t j = 0; j < n; ++j) {
        double* col = c->Kinv + j * n;
        col[j] = 1.0;
        /* columns of L^-1 are zero above the diagonal: start the forward solve at j */
        trsv_lower(&A_(c->L, j, j, n), n - j, n, col + j);
        trsv_lower_t(c->L, n, n, col);
    }
    c->inv_ok = 1;
    return 0;
}

/* GP::compute_kernel_grad_log_lik: src/limbo/model/gp.hpp:285-311, with
 * BaseKernel::grad (kernel.hpp:86-96) appending 2*noise*delta_ij when optimize_noise. */
int orc_log_lik_grad(orc_handle c, double* grad, int n_grad, int optimize_noise)
{
    if (!c || !c->have_L || !grad)
        return -2;
    if (c->host_K)
        return -5;
    int64_t n = c->N;
    int nt = c->n_theta;
    if (n_grad != nt + (optimize_noise ? 1 : 0))
        return -1;
    if (!c->inv_ok)
        orc_compute_inv_kernel(c);
    for (int t = 0; t < n_grad; ++t)
        grad[t] = 0.0;
    double g[MAX_THETA + 1];
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t j = 0; j <= i; ++j) {
            double w = 0.0; /* w = alpha alpha^T - K^-1  (:293-296) */
            for (int p = 0; p < c->P; ++p)
                w += c->alpha[i + p * n] * c->alpha[j + p * n];
            w -= A_(c->Kinv, i, j, n);
            k_grad(c->kind, c->X + i * c->D, c->X + j * c->D, c->D, c->theta, c->n_theta, g);
            if (optimize_noise)
                g[nt] = (i == j) ? 2.0 * c->noise : 0.0;
            double f = (i == j) ? 0.5 : 1.0; /* :303-306 */
            for (int t = 0; t < n_grad; ++t)
                grad[t] += w * g[t] * f;
        }
    }
    return 0;
}

/* GP::compute_log_loo_cv: src/limbo/model/gp.hpp:339-351
 *   inv_diag_i = 1 / (K^-1)_ii ;  sum over (i, p) of  -1/2 alpha_ip^2 inv_diag_i - 1/2 log inv_diag_i - 1/2 log 2 pi */
int orc_log_loo_cv(orc_handle c, double* out)
{
    if (!c || !c->have_L || !out)
        return -2;
    if (!c->inv_ok)
        orc_compute_inv_kernel(c);
    int64_t n = c->N;
    double s = 0.0;
    for (int p = 0; p < c->P; ++p)
        for (int64_t i = 0; i < n; ++i) {
            double inv_d = 1.0 / A_(c->Kinv, i, i, n);
            double a = c->alpha[i + p * n];
            s += -0.5 * a * a * inv_d - 0.5 * log(inv_d) - 0.5 * log(2.0 * M_PI);
        }
    *out = s;
    return 0;
}

/* GP::compute_kernel_grad_log_loo_cv: src/limbo/model/gp.hpp:354-402, as written there: per
 * hyper-parameter j the dense dK/dtheta_j (:380-384), Zeta_j = K^-1 dK_j (:385), Zeta_j alpha (:386),
 * Zeta_j K^-1 (:387) and the column sums of :389.  O(T N^3): small N only. */
int orc_log_loo_cv_grad(orc_handle c, double* grad, int n_grad, int optimize_noise)
{
    if (!c || !c->have_L || !grad)
        return -2;
    if (c->host_K)
        return -5;
    int64_t n = c->N;
    int nt = c->n_theta, P = c->P;
    if (n_grad != nt + (optimize_noise ? 1 : 0))
        return -1;
    if (!c->inv_ok)
        orc_compute_inv_kernel(c);
    double* Ki = (double*)malloc(sizeof(double) * (size_t)(n * n)); /* full symmetric K^-1 */
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j <= i; ++j)
            Ki[i + j * n] = Ki[j + i * n] = A_(c->Kinv, i, j, n);
    double* dK = (double*)malloc(sizeof(double) * (size_t)(n * n * n_grad)); /* full_dk (:366-376) */
    double g[MAX_THETA + 1];
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j <= i; ++j) {
            k_grad(c->kind, c->X + i * c->D, c->X + j * c->D, c->D, c->theta, c->n_theta, g);
            if (optimize_noise)
                g[nt] = (i == j) ? 2.0 * c->noise : 0.0;
            for (int t = 0; t < n_grad; ++t)
                dK[(size_t)t * n * n + i + j * n] = dK[(size_t)t * n * n + j + i * n] = g[t];
        }
    double* Z = (double*)malloc(sizeof(double) * (size_t)(n * n));
    double* Za = (double*)malloc(sizeof(double) * (size_t)(n * P));
    double* Zd = (double*)malloc(sizeof(double) * (size_t)n);
    for (int t = 0; t < n_grad; ++t) {
        const double* G = dK + (size_t)t * n * n;
        for (int64_t i = 0; i < n; ++i)
            for (int64_t k = 0; k < n; ++k) { /* Zeta = K^-1 dK */
                double s = 0.0;
                for (int64_t m = 0; m < n; ++m)
                    s += Ki[i + m * n] * G[m + k * n];
                Z[i + k * n] = s;
            }
        for (int64_t i = 0; i < n; ++i) {
            for (int p = 0; p < P; ++p) { /* Zeta alpha */
                double s = 0.0;
                for (int64_t m = 0; m < n; ++m)
                    s += Z[i + m * n] * c->alpha[m + p * n];
                Za[i + p * n] = s;
            }
            double d = 0.0; /* diag(Zeta K^-1) */
            for (int64_t m = 0; m < n; ++m)
                d += Z[i + m * n] * Ki[m + i * n];
            Zd[i] = d;
        }
        double acc = 0.0; /* :389 then :395 (sum over outputs) */
        for (int p = 0; p < P; ++p)
            for (int64_t i = 0; i < n; ++i) {
                double inv_d = 1.0 / Ki[i + i * n];
                double a = c->alpha[i + p * n];
                acc += (a * Za[i + p * n] - 0.5 * (1.0 + a * a * inv_d) * Zd[i]) * inv_d;
            }
        grad[t] = acc;
    }
    free(Ki);
    free(dK);
    free(Z);
    free(Za);
    free(Zd);
    return 0;
}

/* Not in the reference: the weight matrix W with dLOO/dtheta_j = sum_ab W[a,b] dK_j[a,b], obtained by
 * collecting the terms of gp.hpp:385-389 that multiply dK_j (both are linear in it):
 *   W = sum_p sym(u_p alpha_p^T) - K^-1 diag(c) K^-1,  u_p = K^-1 (alpha_p / kappa),  kappa = diag K^-1,
 *   c_i = sum_p 1/2 (1 + alpha_ip^2 / kappa_i) / kappa_i.   Checker for gpe_get_loo_weights. */
int orc_get_loo_weights(orc_handle c, double* W, int64_t ld)
{
    if (!c || !c->have_L || !W || ld < c->N)
        return -2;
    if (!c->inv_ok)
        orc_compute_inv_kernel(c);
    int64_t n = c->N;
    int P = c->P;
    if (n <= 0 || P <= 0)
        return -2;
    double* Ki = (double*)malloc(sizeof(double) * (size_t)(n * n));
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j <= i; ++j)
            Ki[i + j * n] = Ki[j + i * n] = A_(c->Kinv, i, j, n);
    double* u = (double*)calloc((size_t)(n * P), sizeof(double));
    double* cc = (double*)calloc((size_t)n, sizeof(double));
    for (int p = 0; p < P; ++p)
        for (int64_t i = 0; i < n; ++i) {
            double a = c->alpha[i + p * n], kap = Ki[i + i * n];
            cc[i] += 0.5 * (1.0 + a * a / kap) / kap;
            for (int64_t m = 0; m < n; ++m)
                u[m + p * n] += Ki[m + i * n] * (a / kap);
        }
    for (int64_t a = 0; a < n; ++a)
        for (int64_t b = 0; b < n; ++b) {
            double w = 0.0;
            for (int p = 0; p < P; ++p)
                w += 0.5 * (u[a + p * n] * c->alpha[b + p * n] + c->alpha[a + p * n] * u[b + p * n]);
            double m = 0.0;
            for (int64_t i = 0; i < n; ++i)
                m += Ki[a + i * n] * cc[i] * Ki[i + b * n];
            W[a + b * ld] = w - m;
        }
    free(Ki);
    free(u);
    free(cc);
    return 0;
}

/* SparsifiedGP::_sparsify + _get_most_dense_point: src/limbo/model/sparsified_gp.hpp:124-183, in the
 * order of the serial build (tools::par::loop degenerates to a for loop without TBB,
 * tools/parallel.hpp:138-152): full distance matrix (:160-166); while more than max_points remain,
 * for every remaining i sort the distances to the others, add the D smallest in ascending order
 * (:136-143), remove the first strict minimum (:146-149, :174-178). */
static int cmp_double(const void* a, const void* b)
{
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}
int orc_sparsify(int device_id, const double* X, int64_t N, int D, int64_t max_points, int64_t* keep, int64_t* n_keep)
{
    (void)device_id;
    if (!X || !keep || !n_keep || N <= 0 || D <= 0 || max_points <= 0)
        return -1;
    int64_t n = N;
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)N);
    for (int64_t i = 0; i < N; ++i)
        idx[i] = i;
    if (N > max_points) {
        if (max_points <= D) {
            free(idx);
            return -5;
        }
        double* dist = (double*)malloc(sizeof(double) * (size_t)(N * N));
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < N; ++j) {
                double s = 0.0;
                for (int d = 0; d < D; ++d) {
                    double q = X[i * D + d] - X[j * D + d];
                    s += q * q;
                }
                dist[i + j * N] = sqrt(s); /* (samples[i] - samples[j]).norm() */
            }
        double* nb = (double*)malloc(sizeof(double) * (size_t)N);
        while (n > max_points) {
            double min_dist = DBL_MAX;
            int64_t denser = -1;
            for (int64_t a = 0; a < n; ++a) {
                int64_t m = 0;
                for (int64_t b = 0; b < n; ++b)
                    if (b != a) /* neighbors.erase(begin + i) */
                        nb[m++] = dist[idx[a] + idx[b] * N];
                qsort(nb, (size_t)m, sizeof(double), cmp_double); /* partial_sort: same first D values */
                double dsum = 0.0;
                for (int j = 0; j < D; ++j)
                    dsum += nb[j];
                if (dsum < min_dist) {
                    min_dist = dsum;
                    denser = a;
                }
            }
            if (denser < 0)
                break;
            for (int64_t a = denser; a + 1 < n; ++a) /* samp.erase / _remove_row / _remove_column */
                idx[a] = idx[a + 1];
            --n;
        }
        free(nb);
        free(dist);
    }
    for (int64_t i = 0; i < n; ++i)
        keep[i] = idx[i];
    *n_keep = n;
    free(idx);
    return 0;
}

/* KernelLFOptimization::operator(): src/limbo/model/gp/kernel_lf_opt.hpp:77-92 */
int orc_hp_objective(orc_handle c, int kind, const double* th, int n_theta, double noise,
    int optimize_noise, int want_grad, double* lik, double* grad)
{
    int rc = orc_set_kernel(c, kind, th, n_theta, noise);
    if (rc)
        return rc;
    int info = orc_compute(c); /* recompute(false) -> _compute_full_kernel */
    if (info < 0)
        return info;
    rc = orc_log_lik(c, lik);
    if (rc)
        return rc;
    if (want_grad) {
        rc = orc_log_lik_grad(c, grad, n_theta + (optimize_noise ? 1 : 0), optimize_noise);
        if (rc)
            return rc;
    }
    return info;
}

/* GP::_compute_k / _mu / _sigma for M points: src/limbo/model/gp.hpp:613-632.
 * Returns the raw pieces; mean, clamp (:623) and +noise (:166) are the caller's. */
int orc_query_batch(orc_handle c, const double* Xq, int64_t M, double* kta, double* var)
{
    if (!c || !c->have_L || !Xq)
        return -2;
    if (c->host_K)
        return -5;
    int64_t n = c->N;
    double* k = (double*)malloc(sizeof(double) * (size_t)n);
    for (int64_t m = 0; m < M; ++m) {
        const double* v = Xq + m * c->D;
        for (int64_t i = 0; i < n; ++i)
            k[i] = k_eval(c->kind, c->X + i * c->D, v, c->D, c->theta, c->n_theta); /* :626-632, no noise */
        if (kta)
            for (int p = 0; p < c->P; ++p) {
                double s = 0.0;
                for (int64_t i = 0; i < n; ++i)
                    s += k[i] * c->alpha[i + p * n]; /* :615 */
                kta[m + M * p] = s;
            }
        if (var) {
            trsv_lower(c->L, n, n, k); /* :620 */
            double zz = 0.0;
            for (int64_t i = 0; i < n; ++i)
                zz += k[i] * k[i];
            var[m] = k_eval(c->kind, v, v, c->D, c->theta, c->n_theta) - zz; /* :621 */
        }
    }
    free(k);
    return 0;
}

/* gp.hpp:537-548 result handed over again without touching X, K or L (recompute(true, .)) */
int orc_set_obs_mean(orc_handle c, const double* obs_mean)
{
    if (!c || !obs_mean || c->N <= 0)
        return -1;
    memcpy(c->obs_mean, obs_mean, sizeof(double) * (size_t)(c->N * c->P));
    return 0;
}

/* gp.hpp:613-624 with the cross kernel k* supplied by the caller (kernels without device code):
 * kta[m + M p] = k*_m^T alpha_p (:615), zz[m] = |L^-1 k*_m|^2 (:620-621) */
int orc_query_batch_cross(orc_handle c, const double* Ks, int64_t M, double* kta, double* zz)
{
    if (!c || !c->have_L || !Ks)
        return -2;
    int64_t n = c->N;
    double* k = (double*)malloc(sizeof(double) * (size_t)n);
    for (int64_t m = 0; m < M; ++m) {
        memcpy(k, Ks + m * n, sizeof(double) * (size_t)n);
        if (kta)
            for (int p = 0; p < c->P; ++p) {
                double s = 0.0;
                for (int64_t i = 0; i < n; ++i)
                    s += k[i] * c->alpha[i + p * n];
                kta[m + M * p] = s;
            }
        if (zz) {
            trsv_lower(c->L, n, n, k);
            double q = 0.0;
            for (int64_t i = 0; i < n; ++i)
                q += k[i] * k[i];
            zz[m] = q;
        }
    }
    free(k);
    return 0;
}

int orc_nb_samples(orc_handle c, int64_t* N)
{
    *N = c->N;
    return 0;
}
static int copy_out(const double* src, int64_t n, double* dst, int64_t ld)
{
    if (!src)
        return -2;
    for (int64_t j = 0; j < n; ++j)
        memcpy(dst + j * ld, src + j * n, sizeof(double) * (size_t)n);
    return 0;
}
int orc_get_L(orc_handle c, double* L, int64_t ld) { return copy_out(c->L, c->N, L, ld); }
int orc_get_K(orc_handle c, double* K, int64_t ld)
{
    build_K(c);
    return copy_out(c->K, c->N, K, ld);
}
int orc_get_Kinv(orc_handle c, double* Ki, int64_t ld)
{
    if (!c->inv_ok) {
        int rc = orc_compute_inv_kernel(c);
        if (rc)
            return rc;
    }
    return copy_out(c->Kinv, c->N, Ki, ld);
}
int orc_set_L(orc_handle c, const double* L, int64_t ld)
{
    int64_t n = c->N;
    free(c->L);
    c->L = (double*)calloc((size_t)(n * n), sizeof(double));
    for (int64_t j = 0; j < n; ++j)
        memcpy(c->L + j * n + j, L + j * ld + j, sizeof(double) * (size_t)(n - j));
    c->have_L = 1;
    c->inv_ok = 0;
    return 0;
}
int orc_get_alpha(orc_handle c, double* a)
{
    if (!c->alpha)
        return -2;
    memcpy(a, c->alpha, sizeof(double) * (size_t)(c->N * c->P));
    return 0;
}
int orc_set_alpha(orc_handle c, const double* a)
{
    free(c->alpha);
    c->alpha = dup_d(a, c->N * c->P);
    return 0;
}

/* MultiGP::compute (src/limbo/model/multi_gp.hpp:124-126): a loop over independent GPs */
int orc_batch_compute(orc_handle* hs, int G, int* status)
{
    for (int g = 0; g < G; ++g) {
        int rc = orc_compute(hs[g]);
        if (status)
            status[g] = rc;
    }
    return 0;
}
int orc_batch_log_lik(orc_handle* hs, int G, double* out)
{
    for (int g = 0; g < G; ++g) {
        int rc = orc_log_lik(hs[g], out + g);
        if (rc)
            return rc;
    }
    return 0;
}

/* KernelLFOptimization::operator() (src/limbo/model/gp/kernel_lf_opt.hpp:77-92) for G GPs: the reference runs the
 * restarts / outputs as independent tasks (opt/parallel_repeater.hpp:84-105, model/multi_gp/parallel_lf_opt.hpp:64-67):
 * here, one after the other */
int orc_batch_hp_objective(orc_handle* hs, int G, int kind, const double* th, int n_theta, const double* noise,
    int optimize_noise, int want_grad, double* lik, double* grad, int* status)
{
    int worst = 0;
    const int n_grad = n_theta + (optimize_noise ? 1 : 0);
    for (int g = 0; g < G; ++g) {
        int rc = orc_hp_objective(hs[g], kind, th + (size_t)g * n_theta, n_theta, noise[g], optimize_noise, want_grad,
            lik + g, want_grad ? grad + (size_t)g * n_grad : 0);
        if (status)
            status[g] = rc;
        if (rc < 0)
            worst = rc;
    }
    return worst;
}

int orc_synchronize(orc_handle c)
{
    (void)c;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Rprop (src/limbo/opt/rprop.hpp:84-144) over the KernelLFOpt objective        */
/* (src/limbo/model/gp/kernel_lf_opt.hpp:60-69).  Plain C so tests can compare  */
/* the host-side C++ optimiser driving the GPU against the same iteration.      */
/* ------------------------------------------------------------------------- */
static double signum(double x) { return (x > 0) - (x < 0); } /* tools/math.hpp */

int orc_kernel_lf_opt_rprop(orc_handle c, int optimize_noise, int iterations, double eps_stop,
    double* theta_out, double* best_lik_out, int* n_evals)
{
    int nt = c->n_theta;
    int dim = nt + (optimize_noise ? 1 : 0);
    double delta0 = 0.1, deltamin = 1e-6, deltamax = 50, etaminus = 0.5, etaplus = 1.2; /* :89-93 */
    double delta[MAX_THETA + 1], grad_old[MAX_THETA + 1], params[MAX_THETA + 1];
    double best_params[MAX_THETA + 1], grad[MAX_THETA + 1];
    for (int j = 0; j < nt; ++j)
        params[j] = c->theta[j];
    if (optimize_noise)
        params[nt] = log(sqrt(c->noise)); /* _noise_p, kernel.hpp:78 */
    for (int j = 0; j < dim; ++j) {
        delta[j] = delta0;
        grad_old[j] = 0.0;
        best_params[j] = params[j];
    }
    double best = -INFINITY; /* log(0) :109 */
    int evals = 0;
    for (int i = 0; i < iterations; ++i) {
        double lik;
        double noise = optimize_noise ? exp(2 * params[nt]) : c->noise; /* kernel.hpp:119-122 */
        int rc = orc_hp_objective(c, c->kind, params, nt, noise, optimize_noise, 1, &lik, grad);
        ++evals;
        if (rc < 0)
            return rc;
        if (lik > best) { /* :115-118 */
            best = lik;
            memcpy(best_params, params, sizeof(double) * (size_t)dim);
        }
        double nrm = 0.0;
        for (int j = 0; j < dim; ++j) {
            grad[j] = -grad[j];             /* :119 */
            grad_old[j] = grad_old[j] * grad[j]; /* :120 */
        }
        for (int j = 0; j < dim; ++j) { /* :122-136 */
            if (grad_old[j] > 0) {
                delta[j] = fmin(delta[j] * etaplus, deltamax);
            }
            else if (grad_old[j] < 0) {
                delta[j] = fmax(delta[j] * etaminus, deltamin);
                grad[j] = 0;
            }
            params[j] += -signum(grad[j]) * delta[j];
        }
        for (int j = 0; j < dim; ++j) {
            grad_old[j] = grad[j]; /* :138 */
            nrm += grad[j] * grad[j];
        }
        if (sqrt(nrm) < eps_stop) /* :139 */
            break;
    }
    /* KernelLFOpt::operator(): set best, recompute(false), compute_log_lik  (:66-68) */
    double noise = optimize_noise ? exp(2 * best_params[nt]) : c->noise;
    double lik;
    int rc = orc_hp_objective(c, c->kind, best_params, nt, noise, optimize_noise, 0, &lik, NULL);
    if (rc < 0)
        return rc;
    memcpy(theta_out, best_params, sizeof(double) * (size_t)dim);
    if (best_lik_out)
        *best_lik_out = lik;
    if (n_evals)
        *n_evals = evals;
    return 0;
}

/* scalar entry points for the kernel known-answer tests (test_kernel.cpp:196-224) */
double orc_kernel_eval(int kind, const double* x1, const double* x2, int D, const double* theta)
{
    return k_eval(kind, x1, x2, D, theta, kind == K_SE_ARD ? D + 1 : 2);
}
void orc_kernel_grad(int kind, const double* x1, const double* x2, int D, const double* theta, double* g)
{
    k_grad(kind, x1, x2, D, theta, kind == K_SE_ARD ? D + 1 : 2, g);
}
/* same with an explicit parameter count: SE-ARD with k = (n_theta - 1) / D - 1 columns of Lambda */
double orc_kernel_eval_n(int kind, const double* x1, const double* x2, int D, const double* theta, int n_theta)
{
    return k_eval(kind, x1, x2, D, theta, n_theta);
}
void orc_kernel_grad_n(int kind, const double* x1, const double* x2, int D, const double* theta, int n_theta, double* g)
{
    k_grad(kind, x1, x2, D, theta, n_theta, g);
}
