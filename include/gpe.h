/*
 * gpe.h — C-ABI of libgpengine.so, the MI355X (gfx950) GP posterior engine.
 *
 * This is the drop-in boundary for the hot path of resibots/limbo's
 * `limbo::model::GP<Params, Kernel, Mean, HPOpt>` (reference file
 * src/limbo/model/gp.hpp).  The reference has no FFI of its own (it is a
 * header-only C++ template over Eigen), so every entry point below names the
 * reference member function / Eigen call-site it replaces.  The C++ drop-in
 * template in include/limbo_amd/ is the only intended caller besides the test
 * and benchmark harnesses.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.
 *   - host pointers are caller-owned; the library copies (H2D/D2H) unless the
 *     function name ends in `_device`, in which case the pointer is a HIP
 *     device pointer on the handle's device.
 *   - matrices are COLUMN-MAJOR (Eigen::MatrixXd default) unless the name says
 *     `rowmajor`; sample matrices are row-major N x D (one sample per row,
 *     i.e. the concatenation of limbo's std::vector<Eigen::VectorXd>).
 *   - every function returns an int status: 0 = ok; >0 = 1-based index of the
 *     first non-positive Cholesky pivot (the reference never checks
 *     Eigen::LLT::info(), gp.hpp:565 — the C++ wrapper mirrors that and just
 *     records it); <0 = GPE_ERR_*.
 *   - hyper-parameters are in limbo's log-space (kernel/kernel.hpp:104-123).
 *   - one handle = one GP = one HIP stream; different handles may be driven
 *     from different host threads concurrently; const queries on one handle
 *     are serialised by an internal mutex.
 */
#ifndef GPE_H
#define GPE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpe_ctx* gpe_handle;

enum gpe_status {
    GPE_OK = 0,
    GPE_ERR_ARG = -1,      /* bad argument / shape (reference: assert) */
    GPE_ERR_STATE = -2,    /* call order (e.g. log_lik before compute) */
    GPE_ERR_HIP = -3,      /* HIP runtime error, see gpe_last_error()   */
    GPE_ERR_NOMEM = -4,
    GPE_ERR_UNSUPPORTED = -5
};

/* kernel functors with device code.
 * SE_ARD    kernel/squared_exp_ard.hpp:138-151               theta = [log l_1..log l_D, log sigma_f]
 *           with k > 0 columns of Lambda (:109-126, :142-146):  [log l_1..log l_D, Lambda(:,0), ..,
 *           Lambda(:,k-1), log sigma_f] (params_size :94; Lambda not in log-space, :100-102).  k follows
 *           from n_theta = D + D k + 1; limits: k <= D, n_theta <= 64.
 * MATERN52  kernel/matern_five_halves.hpp:104-113            theta = [log l, log sigma_f]
 * MATERN32  kernel/matern_three_halves.hpp                   theta = [log l, log sigma_f]
 * EXP       kernel/exp.hpp                                   theta = [log l, log sigma_f]
 * HOST_K    any user functor: K is built on the host by the C++ wrapper and
 *           uploaded with gpe_set_K_host(); factorisation/solves stay on device.
 */
enum gpe_kernel_kind {
    GPE_KERNEL_SE_ARD = 0,
    GPE_KERNEL_MATERN52 = 1,
    GPE_KERNEL_MATERN32 = 2,
    GPE_KERNEL_EXP = 3,
    GPE_KERNEL_HOST_K = 4
};

/* ---- lifetime ------------------------------------------------------------ */
int gpe_create(int device_id, gpe_handle* out);
/* deep copy (value semantics of limbo::model::GP, e.g. kernel_lf_opt.hpp:79) */
int gpe_clone(gpe_handle src, gpe_handle* out);
/* the same onto another device of the node (peer copy over xGMI): how tools::par::loop / par::max
 * (tools/parallel.hpp:138-191) spread the independent GPs of model::MultiGP (multi_gp.hpp:124-126) and the restarts of
 * opt::ParallelRepeater (parallel_repeater.hpp:86-105) over the 8 MI355X of a node — the C++ header deals the clones
 * round-robin.  gpe_device_count: devices visible to this process. */
int gpe_clone_to(gpe_handle src, int device_id, gpe_handle* out);
int gpe_device_count(int* n);
int gpe_get_device(gpe_handle h, int* device_id);
/* A counter that moves with every call that can change what a query answers (set_data, set_kernel, compute, add_sample,
 * update_alpha, set_L, set_alpha, the batched entry points ...): a clone made at epoch e answers like its source for as long as
 * the source's epoch is still e.  The C++ drop-in keeps one query replica per device on it (model/gp.hpp: query_batch over
 * the visible devices — the reference's parallel query, multi_gp.hpp:191-195 / tools/parallel.hpp:138-201). */
int gpe_epoch(gpe_handle h, uint64_t* epoch);
/* Several PROCESSES on one GPU (an ordinary way to run limbo experiments; the reference's gp.hpp:565 is re-entrant across
 * processes): the engine's data-flow launches wait for each other inside a launch and must not share the device with another
 * process's.  While another process of this library has a handle on the GPU (a byte of /dev/shm/limbo_amd.gpu-<bus id>.users
 * locked), every data-flow launch scope runs — launch AND host wait — under flock(.lock); alone, nothing is locked.
 * gpe_xproc_waits: how many scopes of this process ran under the lock so far. */
int gpe_xproc_waits(int64_t* n);
int gpe_destroy(gpe_handle h);
const char* gpe_last_error(gpe_handle h);
const char* gpe_version(void);

/* ---- data & hyper-parameters -------------------------------------------- */
/* gp.hpp:105-111 + :537-548.  X: N x D row-major.  obs_mean = Y - m(X), N x P
 * column-major; the mean functor is evaluated on the host by the caller
 * (it receives the GP itself, mean/data.hpp:59-63). */
int gpe_set_data(gpe_handle h, const double* X_rowmajor, int64_t N, int D,
                 const double* obs_mean, int P);
/* same, from device memory already resident on the handle's device */
int gpe_set_data_device(gpe_handle h, const double* dX_rowmajor, int64_t N, int D,
                        const double* d_obs_mean, int P);
/* gp.hpp:537-548 result again, X unchanged (recompute(true, .)): replaces obs_mean only */
int gpe_set_obs_mean(gpe_handle h, const double* obs_mean);
/* kernel.hpp:116-123 set_h_params + the per-kernel set_params.  log_theta has
 * n_theta entries (without the optional noise parameter); noise is sigma_n^2
 * (Params::kernel::noise(), or exp(2 p_noise) when optimize_noise). */
int gpe_set_kernel(gpe_handle h, int kind, const double* log_theta, int n_theta, double noise);
/* HOST_K fallback: full symmetric K (N x N, ld >= N), noise already on the diagonal */
int gpe_set_K_host(gpe_handle h, const double* K, int64_t ldk);

/* ---- the hot path -------------------------------------------------------- */
/* gp.hpp:550-571 `_compute_full_kernel`: K (kernel build, +noise+1e-8 on the
 * diagonal) -> L = chol(K) (replaces Eigen::LLT, gp.hpp:565) -> alpha
 * (gp.hpp:605-611).  Everything stays in HBM. */
int gpe_compute(gpe_handle h);
/* gp.hpp:241-252 recompute(update_obs_mean, false): new obs_mean, same L */
int gpe_update_alpha(gpe_handle h, const double* obs_mean);
/* gp.hpp:126-152 + :573-603: append one sample; obs_mean is the FULL new
 * (N+1) x P matrix (mean::Data moves every row when a sample is added).  On an
 * empty handle D and P set the dimensions (gp.hpp:128-137); otherwise they
 * must match (gp.hpp:139-140 assert -> GPE_ERR_ARG). */
int gpe_add_sample(gpe_handle h, const double* x, int D, const double* obs_mean, int P);
/* gp.hpp:267-282 compute_log_lik */
int gpe_log_lik(gpe_handle h, double* out);
/* gp.hpp:254-264 compute_inv_kernel (K^-1 from L, cached until K changes) */
int gpe_compute_inv_kernel(gpe_handle h);
/* gp.hpp:285-311 compute_kernel_grad_log_lik (+ kernel.hpp:86-96 noise term).
 * grad has n_theta (+1 if optimize_noise) entries. */
int gpe_log_lik_grad(gpe_handle h, double* grad, int n_grad, int optimize_noise);
/* model/sparsified_gp.hpp:124-183 SparsifiedGP::_sparsify: while more than max_points samples remain,
 * drop the one whose D nearest remaining neighbours are closest in total.  X row-major N x D (host).
 * keep[0..*n_keep) = indices of the surviving samples, ascending.  Stand-alone (no handle): the
 * thinning happens before the GP exists.  D <= 64; max_points > D (the reference's partial_sort
 * is undefined otherwise). */
int gpe_sparsify(int device_id, const double* X_rowmajor, int64_t N, int D, int64_t max_points,
                 int64_t* keep, int64_t* n_keep);
/* gp.hpp:339-351 compute_log_loo_cv (leave-one-out log predictive probability) */
int gpe_log_loo_cv(gpe_handle h, double* out);
/* gp.hpp:354-402 compute_kernel_grad_log_loo_cv; same layout as gpe_log_lik_grad.
 * (The reference's 2 T dense N^3 products collapse to one: DESIGN.md.) */
int gpe_log_loo_cv_grad(gpe_handle h, double* grad, int n_grad, int optimize_noise);
/* the N x N weight matrix W (symmetric, col-major, ld) with d LOO / d theta_j = sum_ab W[a,b] dK_j[a,b]:
 * for kernels whose gradient only exists as a host functor (cf. gpe_set_K_host) */
int gpe_get_loo_weights(gpe_handle h, double* W, int64_t ld);
/* model/gp/kernel_lf_opt.hpp:77-92 KernelLFOptimization::operator() in one
 * call, without the reference's per-evaluation deep copy: set theta (and
 * noise), recompute(false), log-lik and (optionally) its gradient. */
int gpe_hp_objective(gpe_handle h, int kind, const double* log_theta, int n_theta,
                     double noise, int optimize_noise, int want_grad,
                     double* log_lik, double* grad);
/* gp.hpp:613-632 for a batch of M points (row-major M x D):
 *   kta[m + M*p] = k(X, v_m)^T alpha_p          (add m(v) on the host, :615)
 *   var[m]       = k(v_m, v_m) - ||L^-1 k*||^2  (clamp and +noise on the host, :166,:623)
 * either output may be NULL.  M <= 8 (the per-point gp.query(v) of an acquisition functor) runs the forward
 * substitution as one data-flow launch; larger batches the blocked matrix-core solve.  The two agree to
 * rounding (different summation order); each is bitwise reproducible from call to call. */
int gpe_query_batch(gpe_handle h, const double* Xq_rowmajor, int64_t M,
                    double* kta, double* var);

/* the same for kernels without device code: the caller supplies the cross kernel
 * Ks[i + N*m] = k(x_i, v_m) (N x M column-major, gp.hpp:626-632); zz[m] = ||L^-1 k*_m||^2 */
int gpe_query_batch_cross(gpe_handle h, const double* Ks, int64_t M, double* kta, double* zz);

/* ---- accessors (host mirrors for matrixL()/alpha()/save/load) ------------ */
int gpe_nb_samples(gpe_handle h, int64_t* N);
int gpe_get_L(gpe_handle h, double* L, int64_t ld);          /* lower, upper part zeroed (gp.hpp:411) */
int gpe_set_L(gpe_handle h, const double* L, int64_t ld);    /* load(..., recompute=false) gp.hpp:506-509 */
int gpe_get_alpha(gpe_handle h, double* alpha);              /* N x P */
int gpe_set_alpha(gpe_handle h, const double* alpha);
int gpe_get_Kinv(gpe_handle h, double* Kinv, int64_t ld);    /* full symmetric */
int gpe_get_K(gpe_handle h, double* K, int64_t ld);          /* rebuilds K; for tests */

/* ---- independent GPs: multi_gp.hpp:124-126, parallel_repeater.hpp:86-105 -- */
/* run gpe_compute on G handles (same device): kernels of different GPs are
 * interleaved on the device instead of being serialised. */
int gpe_batch_compute(gpe_handle* hs, int G, int* status);
int gpe_batch_log_lik(gpe_handle* hs, int G, double* out);
/* model/gp/kernel_lf_opt.hpp:77-92 KernelLFOptimization::operator() for G GPs at once: the restarts of
 * opt::ParallelRepeater (opt/parallel_repeater.hpp:84-105) stepped in lock-step, or the per-output fits of
 * multi_gp::ParallelLFOpt (model/multi_gp/parallel_lf_opt.hpp:64-67).  Member g: log_theta[g n_theta .. (g+1) n_theta),
 * noise[g] -> lik[g], grad[g n_grad .. (g+1) n_grad) with n_grad = n_theta + (optimize_noise ? 1 : 0), status[g]
 * (0 or the 1-based first non-positive pivot).  Handles of one shape on one device are stepped by ONE launch sequence
 * (kernel build, factorisation, alpha, K^-1 and the gradient pair sum, gridDim.z = member); others one by one. */
int gpe_batch_hp_objective(gpe_handle* hs, int G, int kind, const double* log_theta, int n_theta, const double* noise,
                           int optimize_noise, int want_grad, double* lik, double* grad, int* status);

/* ---- instrumentation ----------------------------------------------------- */
/* HIP stream the handle launches on (hipStream_t as void*) */
int gpe_get_stream(gpe_handle h, void** stream);
int gpe_synchronize(gpe_handle h);
/* when on, gpe_compute brackets every phase with HIP events on the handle's
 * stream; read back with gpe_get_phase_ms (sums since the last reset). */
enum gpe_phase {
    GPE_PH_KERNEL_BUILD = 0,
    GPE_PH_POTRF_PANEL = 1,
    GPE_PH_POTRF_UPDATE = 2,   /* trailing SYRK/GEMM (fp64 MFMA) */
    GPE_PH_SOLVE = 3,
    GPE_PH_LOGLIK = 4,
    GPE_PH_INV = 5,
    GPE_PH_GRAD = 6,
    GPE_PH_QUERY = 7,
    GPE_PH_POTRF_TALL = 8,     /* the tall data-flow launch in front of the closing one (k_tail, round 4) */
    GPE_PH_POTRF_TAIL = 9,     /* the closing data-flow launch (k_tail) */
    GPE_PH_COUNT = 10
};
int gpe_set_profiling(gpe_handle h, int on);
int gpe_get_phase_ms(gpe_handle h, double* ms, int64_t* launches, double* flops, int n);
int gpe_reset_phase_ms(gpe_handle h);
/* one-launch sweeps that had to be re-run block by block after a hand-off timeout (never expected; tests) */
int gpe_flow_retries(gpe_handle h, int64_t* n);
/* evaluations run a second time because a panel step's head-tile hand-over timed out (never expected; counted in
 * gpe_flow_retries as well).  The hand-over is switched off for the 16 evaluations that follow and re-armed after. */
int gpe_handover_reruns(gpe_handle h, int64_t* n);
/* calls (add_sample, point queries) served by the one-launch small-N path (csrc/small.hip; GPE_SMALL=0 disables it) */
int gpe_small_calls(gpe_handle h, int64_t* n);
/* Launch trace of the PRODUCTION schedule: while on, every kernel launch of the library carries its own start and stop
 * event (hipExtLaunchKernelGGL, the dispatch's own timestamps: no marker packets, no profiler) — rocprofv3's kernel trace
 * cannot show the fused next-panel update, it delays dispatches that carry a completion event by ~100 us.  gpe_trace(1/0)
 * switches it (and clears the records; GPE_TRACE=1 starts a process with it on); gpe_trace_dump writes one line per launch:
 * start us, end us, stream index, kernel, grid, block (times from the first recorded launch's start). */
int gpe_trace(int on);
int gpe_trace_dump(const char* path);
/* Test hook, host only (no device is touched): the dispatch table of a data-flow launch of `nt` tile columns x `nb` row strips
 * (csrc/potrf.hip: tail_order; lag = GPE_TAIL_LAG, pair = GPE_TAIL_PAIR) is built and checked — 1: a permutation of the launch's
 * tiles in which every workgroup waits for lower-numbered ones only (what makes the launch deadlock-free), 0: not, -1: bad
 * arguments.  The engine runs the same check before it uses a table. */
int gpe_debug_tail_order(int nt, int nb, int lag, int pair);
/* Test hook, host only: how the eight waves of k_tail's chain workgroup split its products (csrc/potrf.hip: syrk40, tri_solve32).
 * units10 = five { row block i of 16, column block j of 4 } pairs: the wave's units of the LOWER triangle of the 64 x 64 diagonal
 * block (40 in all, every one needed exactly once); *cols = the first of the wave's eight columns of a 32-column triangular product
 * (its k loop runs to cols + 8).  0: ok, -1: bad arguments. */
int gpe_debug_chain_split(int wave, int* units10, int* cols);
/* Test hook, host only: how the update of a ragged order's last block behind a data-flow launch deals its k range (csrc/potrf.hip:
 * launch_ragged_update): returns the number of workgroups (0: the general product runs instead — k < 256 or no room for two
 * 64 x 64 slots in scratch_doubles) and *kc = the k rows of each; -1: bad arguments. */
int gpe_debug_ragged_split(int64_t k, int64_t scratch_doubles, int* kc);
/* Test hook, host only: which workgroup takes which 128 x 128 tile of a triangular trailing update C[m x n] -= A B^T whose
 * element (0, 0) is element (grow0, gcol0) of the symmetric matrix (csrc/gemm.hip: tri_tile_map; the updates behind
 * gp.hpp:565's factorisation).  out[b] = ti | tj << 16 for workgroup b (XCD b mod 8), -1: none; returns the table's length
 * (a multiple of 8; at most cap entries are written), -1: bad arguments. */
int gpe_debug_tri_tile_map(int64_t m, int64_t n, int64_t grow0, int64_t gcol0, int* out, int cap);
/* ... and the schedule the engine picks for n samples, p outputs and a batched sequence of g members (g <= 1: one handle) under
 * the given widths (<= 0: the defaults GPE_TAIL_MAX / GPE_TALL / GPE_BATCH_TAIL_MAX): out8 = { t0 (first column of the closing
 * data-flow launch; -1: panels to the end), e0 (first column of the tall launch in front of it; -1: none), tile columns and row
 * strips of the closing launch, of the tall launch, n rounded down to 64, outer panel width }. */
int gpe_debug_tail_plan(int64_t n, int p, int g, int64_t tail_max, int64_t tall_max, int64_t batch_tail_max, int64_t* out8);
/* Test hook, host only: the launch plan of the recursive K^-1 (csrc/inv2.hip; replaces the dense solves of gp.hpp:254-264) for
 * order n and leading dimension ld (>= n rounded up to 64), dealt into nbins shares (<= 0: 512) with chunk-length factor
 * load_pct / 100 (<= 0: 1.0); nbins < 0: the plan of a batched sequence of -nbins members.  One row of 16 int64 per tile product, in
 * launch order, share by share:
 *   { step (= launch), 0 | 2 (128 x 128 | 64 x 64 tiles), A buf, A offset, B buf, B offset, C buf, C offset, k, negate | share << 1,
 *     valid rows, valid columns (a ragged order's last strip), chunk number | chunks of the tile's k range << 8, flag word of the
 *     tile, T buf | -1, T offset }
 *   share = the persistent workgroup of the launch that runs it (workgroup b sits on XCD b % 8); chunk 0 stores C, chunk c > 0 adds
 *   to it once chunk c - 1 has stored (it sits in no lower a share, behind it inside one); the last chunk also stores C^T at T.
 * buffers: 0 L, 1 U = L^-T, 2 K^-1, 3 T-forms / W; offsets in doubles.  Returns the number of rows of the
 * plan (out may be too small or null: nothing beyond cap_rows is written), -1 for bad arguments.  tests/test_inv_plan.py
 * executes the plan in numpy. */
int gpe_debug_inv_plan(int64_t n, int64_t ld, int nbins, int load_pct, int64_t* out, int64_t cap_rows);
/* fp64 MFMA peak micro-benchmark (v_mfma_f64_4x4x4_4b, the instruction the GEMM kernels issue), TFLOP/s */
int gpe_mfma_f64_peak(int device_id, double* tflops);
/* HBM write-stream micro-benchmark, GB/s */
int gpe_hbm_stream_peak(int device_id, double* gbs);

#ifdef __cplusplus
}
#endif
#endif /* GPE_H */
