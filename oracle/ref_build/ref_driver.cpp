// ref_driver.cpp — C-ABI around the UNMODIFIED reference: limbo::model::GP compiled from
// /root/reference/src (resibots/limbo) against the Eigen/Boost stand-ins of oracle/ref_build/shim.
// TEST INFRASTRUCTURE: builds oracle/_ref/libref.so, which tests/ use to pin oracle/gp_oracle.c (and
// through it the HIP engine) to the reference's own source.  Nothing under limbo_amd/ links or loads it.
//
// Every entry point below is a thin call into the reference's public members
// (src/limbo/model/gp.hpp:88-511); protected state (_kernel, _inv_kernel) is read through a
// subclass, exactly as the reference's own SparsifiedGP / test fixtures do.
#include <cstring>
#include <memory>
#include <tuple>

#include <limbo/acqui/ei.hpp>
#include <limbo/acqui/gp_ucb.hpp>
#include <limbo/acqui/ucb.hpp>
#include <limbo/kernel/exp.hpp>
#include <limbo/kernel/matern_five_halves.hpp>
#include <limbo/kernel/matern_three_halves.hpp>
#include <limbo/kernel/squared_exp_ard.hpp>
#include <limbo/mean/constant.hpp>
#include <limbo/mean/data.hpp>
#include <limbo/mean/null_function.hpp>
#include <limbo/model/gp.hpp>
#include <limbo/model/gp/kernel_lf_opt.hpp>
#include <limbo/model/gp/kernel_loo_opt.hpp>
#include <limbo/model/gp/kernel_mean_lf_opt.hpp>
#include <limbo/model/gp/mean_lf_opt.hpp>
#include <limbo/model/gp/no_lf_opt.hpp>
#include <limbo/opt/parallel_repeater.hpp>
#include <limbo/serialize/binary_archive.hpp>
#include <limbo/serialize/text_archive.hpp>

using namespace limbo;

// Params as a user of limbo writes them; the values the tests vary are BO_DYN_PARAMs (tools/macros.hpp:60-63)
struct Params {
    struct kernel {
        BO_DYN_PARAM(double, noise);
        BO_DYN_PARAM(bool, optimize_noise);
    };
    struct kernel_squared_exp_ard {
        BO_DYN_PARAM(int, k);
        BO_PARAM(double, sigma_sq, 1);
    };
    struct kernel_maternfivehalves : public defaults::kernel_maternfivehalves {
    };
    struct kernel_maternthreehalves : public defaults::kernel_maternthreehalves {
    };
    struct kernel_exp : public defaults::kernel_exp {
    };
    struct mean_constant {
        BO_DYN_PARAM(double, constant);
    };
    struct opt_rprop {
        BO_DYN_PARAM(int, iterations);
        BO_DYN_PARAM(double, eps_stop);
    };
    struct opt_parallelrepeater : public defaults::opt_parallelrepeater {
    };
    struct acqui_ucb : public defaults::acqui_ucb {
    };
    struct acqui_gpucb : public defaults::acqui_gpucb {
    };
    struct acqui_ei : public defaults::acqui_ei {
    };
};
BO_DECLARE_DYN_PARAM(double, Params::kernel, noise);
BO_DECLARE_DYN_PARAM(bool, Params::kernel, optimize_noise);
BO_DECLARE_DYN_PARAM(int, Params::kernel_squared_exp_ard, k);
BO_DECLARE_DYN_PARAM(double, Params::mean_constant, constant);
BO_DECLARE_DYN_PARAM(int, Params::opt_rprop, iterations);
BO_DECLARE_DYN_PARAM(double, Params::opt_rprop, eps_stop);

namespace {

    struct Statics {
        double noise = 0.01;
        bool optimize_noise = false;
        int k = 0;
        double constant = 1.0;
        void apply() const
        {
            Params::kernel::set_noise(noise);
            Params::kernel::set_optimize_noise(optimize_noise);
            Params::kernel_squared_exp_ard::set_k(k);
            Params::mean_constant::set_constant(constant);
        }
    };
    Statics g_next; // what the next ref_create() captures

    Eigen::VectorXd vec(const double* p, int n)
    {
        Eigen::VectorXd v(n);
        for (int i = 0; i < n; ++i)
            v(i) = p[i];
        return v;
    }

    struct IGP {
        Statics st;
        virtual ~IGP() {}
        virtual void compute(const double* X, const double* Y, int64_t N, int D, int P) = 0;
        virtual void add_sample(const double* x, int D, const double* y, int P) = 0;
        virtual void recompute(bool om, bool full) = 0;
        virtual void set_kernel_h(const double* p, int n) = 0;
        virtual int kernel_h_size() = 0;
        virtual void get_kernel_h(double* p) = 0;
        virtual void set_mean_h(const double* p, int n) = 0;
        virtual int mean_h_size() = 0;
        virtual void get_mean_h(double* p) = 0;
        virtual double noise() = 0;
        virtual void query(const double* x, int D, double* mu, double* s2) = 0;
        virtual void mu(const double* x, int D, double* mu) = 0;
        virtual double sigma(const double* x, int D) = 0;
        virtual double log_lik() = 0;
        virtual void kernel_grad_log_lik(double* g) = 0;
        virtual void mean_grad_log_lik(double* g) = 0;
        virtual double log_loo_cv() = 0;
        virtual void kernel_grad_log_loo_cv(double* g) = 0;
        virtual void compute_inv_kernel() = 0;
        virtual int inv_kernel_computed() = 0;
        virtual int64_t nb_samples() = 0;
        virtual int dim_out() = 0;
        virtual void get_matrix(int which, double* out) = 0; // 0 L, 1 alpha, 2 obs_mean, 3 K^-1, 4 K, 5 mean_vector
        virtual void optimize(int which) = 0;               // 0 KernelLFOpt, 1 KernelLooOpt, 2 MeanLFOpt, 3 KernelMeanLFOpt
        virtual double kernel_eval(const double* a, const double* b, int D, int i, int j) = 0;
        virtual void kernel_grad(const double* a, const double* b, int D, int i, int j, double* g) = 0;
        virtual void acqui(int which, int iteration, const double* Xq, int64_t M, int D, double* out) = 0; // acqui/*.hpp
        virtual void save(const char* dir, bool binary) = 0;                 // gp.hpp:439-460
        virtual void load(const char* dir, bool binary, bool recompute) = 0; // gp.hpp:462-511
    };

    template <class K, class M> struct W : IGP {
        using Base = model::GP<Params, K, M, model::gp::NoLFOpt<Params>>;
        struct G : Base {
            using Base::Base;
            const Eigen::MatrixXd& inv_kernel() const { return this->_inv_kernel; }
            const Eigen::MatrixXd& kernel_matrix() const { return this->_kernel; }
        };
        G gp;
        W(int D, int P) : gp(D, P) {}
        void compute(const double* X, const double* Y, int64_t N, int D, int P) override
        {
            std::vector<Eigen::VectorXd> xs, ys;
            for (int64_t i = 0; i < N; ++i) {
                xs.push_back(vec(X + i * D, D));
                ys.push_back(vec(Y + i * P, P));
            }
            gp.compute(xs, ys); // gp.hpp:88
        }
        void add_sample(const double* x, int D, const double* y, int P) override { gp.add_sample(vec(x, D), vec(y, P)); } // :126
        void recompute(bool om, bool full) override { gp.recompute(om, full); }                                           // :241
        void set_kernel_h(const double* p, int n) override { gp.kernel_function().set_h_params(vec(p, n)); }              // kernel.hpp:116
        int kernel_h_size() override { return (int)gp.kernel_function().h_params_size(); }
        void get_kernel_h(double* p) override
        {
            Eigen::VectorXd h = gp.kernel_function().h_params();
            for (int i = 0; i < h.size(); ++i)
                p[i] = h(i);
        }
        void set_mean_h(const double* p, int n) override { gp.mean_function().set_h_params(vec(p, n)); }
        int mean_h_size() override { return (int)gp.mean_function().h_params_size(); }
        void get_mean_h(double* p) override
        {
            Eigen::VectorXd h = gp.mean_function().h_params();
            for (int i = 0; i < h.size(); ++i)
                p[i] = h(i);
        }
        double noise() override { return gp.kernel_function().noise(); }
        void query(const double* x, int D, double* mu, double* s2) override
        {
            Eigen::VectorXd m;
            double s;
            std::tie(m, s) = gp.query(vec(x, D)); // :159
            for (int p = 0; p < m.size(); ++p)
                mu[p] = m(p);
            *s2 = s;
        }
        void mu(const double* x, int D, double* mu) override
        {
            Eigen::VectorXd m = gp.mu(vec(x, D)); // :174
            for (int p = 0; p < m.size(); ++p)
                mu[p] = m(p);
        }
        double sigma(const double* x, int D) override { return gp.sigma(vec(x, D)); } // :186
        double log_lik() override { return gp.compute_log_lik(); }                     // :267
        void kernel_grad_log_lik(double* g) override
        {
            Eigen::VectorXd v = gp.compute_kernel_grad_log_lik(); // :285
            for (int i = 0; i < v.size(); ++i)
                g[i] = v(i);
        }
        void mean_grad_log_lik(double* g) override
        {
            Eigen::VectorXd v = gp.compute_mean_grad_log_lik(); // :314
            for (int i = 0; i < v.size(); ++i)
                g[i] = v(i);
        }
        double log_loo_cv() override { return gp.compute_log_loo_cv(); } // :339
        void kernel_grad_log_loo_cv(double* g) override
        {
            Eigen::VectorXd v = gp.compute_kernel_grad_log_loo_cv(); // :354
            for (int i = 0; i < v.size(); ++i)
                g[i] = v(i);
        }
        void compute_inv_kernel() override { gp.compute_inv_kernel(); } // :254
        int inv_kernel_computed() override { return gp.inv_kernel_computed() ? 1 : 0; }
        int64_t nb_samples() override { return gp.nb_samples(); }
        int dim_out() override { return gp.dim_out(); }
        void get_matrix(int which, double* out) override
        {
            const Eigen::MatrixXd* m = nullptr;
            switch (which) {
            case 0: m = &gp.matrixL(); break;
            case 1: m = &gp.alpha(); break;
            case 2: m = &gp.obs_mean(); break;
            case 3: m = &gp.inv_kernel(); break;
            case 4: m = &gp.kernel_matrix(); break;
            default: m = &gp.mean_vector(); break;
            }
            for (Eigen::Index j = 0; j < m->cols(); ++j) // column-major, ld = rows
                for (Eigen::Index i = 0; i < m->rows(); ++i)
                    out[i + j * m->rows()] = (*m)(i, j);
        }
        void optimize(int which) override
        {
            switch (which) {
            case 0: { model::gp::KernelLFOpt<Params> o; o(gp); break; }     // kernel_lf_opt.hpp:60-69
            case 1: { model::gp::KernelLooOpt<Params> o; o(gp); break; }    // kernel_loo_opt.hpp
            case 2: { model::gp::MeanLFOpt<Params> o; o(gp); break; }       // mean_lf_opt.hpp
            default: { model::gp::KernelMeanLFOpt<Params> o; o(gp); break; } // kernel_mean_lf_opt.hpp
            }
        }
        double kernel_eval(const double* a, const double* b, int D, int i, int j) override
        {
            return gp.kernel_function()(vec(a, D), vec(b, D), i, j); // kernel.hpp:81-84
        }
        void kernel_grad(const double* a, const double* b, int D, int i, int j, double* g) override
        {
            Eigen::VectorXd v = gp.kernel_function().grad(vec(a, D), vec(b, D), i, j); // kernel.hpp:86-96
            for (int q = 0; q < v.size(); ++q)
                g[q] = v(q);
        }
        // the reference's own acquisition functors over its own model, first output as the aggregator (bo_base.hpp: FirstElem):
        // 0 UCB (acqui/ucb.hpp:83-90), 1 GP_UCB (acqui/gp_ucb.hpp:86-103), 2 EI (acqui/ei.hpp:85-116)
        void acqui(int which, int iteration, const double* Xq, int64_t npts, int D, double* out) override
        {
            auto first = [](const Eigen::VectorXd& v) { return v(0); };
            if (which == 0) {
                limbo::acqui::UCB<Params, G> a(gp, iteration);
                for (int64_t m = 0; m < npts; ++m)
                    out[m] = opt::fun(a(vec(Xq + m * D, D), first, false));
            }
            else if (which == 1) {
                limbo::acqui::GP_UCB<Params, G> a(gp, iteration);
                for (int64_t m = 0; m < npts; ++m)
                    out[m] = opt::fun(a(vec(Xq + m * D, D), first, false));
            }
            else {
                limbo::acqui::EI<Params, G> a(gp, iteration);
                for (int64_t m = 0; m < npts; ++m)
                    out[m] = opt::fun(a(vec(Xq + m * D, D), first, false));
            }
        }
        // the reference's own archives, writing / reading real files (serialize/text_archive.hpp, binary_archive.hpp)
        void save(const char* dir, bool binary) override
        {
            if (binary)
                gp.template save<serialize::BinaryArchive>(std::string(dir));
            else
                gp.template save<serialize::TextArchive>(std::string(dir));
        }
        void load(const char* dir, bool binary, bool recompute) override
        {
            if (binary)
                gp.template load<serialize::BinaryArchive>(std::string(dir), recompute);
            else
                gp.template load<serialize::TextArchive>(std::string(dir), recompute);
        }
    };

    template <class K> IGP* make_mean(int mean_kind, int D, int P)
    {
        switch (mean_kind) {
        case 0: return new W<K, mean::Data<Params>>(D, P);
        case 1: return new W<K, mean::NullFunction<Params>>(D, P);
        case 2: return new W<K, mean::Constant<Params>>(D, P);
        }
        return nullptr;
    }

    struct Scope {
        explicit Scope(IGP* g) { g->st.apply(); }
    };
} // namespace

extern "C" {

const char* ref_version(void) { return "resibots/limbo model::GP, unmodified headers, Eigen/Boost stand-in (oracle/ref_build)"; }

// statics read by the functors' constructors and by h_params()/grad(): captured by the next ref_create()
void ref_set_statics(double noise, int optimize_noise, int k_lambda, double mean_constant)
{
    g_next.noise = noise;
    g_next.optimize_noise = optimize_noise != 0;
    g_next.k = k_lambda;
    g_next.constant = mean_constant;
}
void ref_set_rprop(int iterations, double eps_stop)
{
    Params::opt_rprop::set_iterations(iterations);
    Params::opt_rprop::set_eps_stop(eps_stop);
}

// kernel_kind: 0 SquaredExpARD, 1 MaternFiveHalves, 2 MaternThreeHalves, 3 Exp (as include/gpe.h)
// mean_kind  : 0 Data, 1 NullFunction, 2 Constant
void* ref_create(int kernel_kind, int mean_kind, int D, int P)
{
    g_next.apply();
    IGP* g = nullptr;
    switch (kernel_kind) {
    case 0: g = make_mean<kernel::SquaredExpARD<Params>>(mean_kind, D, P); break;
    case 1: g = make_mean<kernel::MaternFiveHalves<Params>>(mean_kind, D, P); break;
    case 2: g = make_mean<kernel::MaternThreeHalves<Params>>(mean_kind, D, P); break;
    case 3: g = make_mean<kernel::Exp<Params>>(mean_kind, D, P); break;
    }
    if (g)
        g->st = g_next;
    return g;
}
void ref_destroy(void* h) { delete (IGP*)h; }

#define G ((IGP*)h)
void ref_compute(void* h, const double* X, const double* Y, int64_t N, int D, int P) { Scope s(G); G->compute(X, Y, N, D, P); }
void ref_add_sample(void* h, const double* x, int D, const double* y, int P) { Scope s(G); G->add_sample(x, D, y, P); }
void ref_recompute(void* h, int update_obs_mean, int update_full_kernel) { Scope s(G); G->recompute(update_obs_mean != 0, update_full_kernel != 0); }
void ref_set_kernel_h_params(void* h, const double* p, int n) { Scope s(G); G->set_kernel_h(p, n); }
int ref_kernel_h_params_size(void* h) { Scope s(G); return G->kernel_h_size(); }
void ref_get_kernel_h_params(void* h, double* p) { Scope s(G); G->get_kernel_h(p); }
void ref_set_mean_h_params(void* h, const double* p, int n) { Scope s(G); G->set_mean_h(p, n); }
int ref_mean_h_params_size(void* h) { Scope s(G); return G->mean_h_size(); }
void ref_get_mean_h_params(void* h, double* p) { Scope s(G); G->get_mean_h(p); }
double ref_noise(void* h) { Scope s(G); return G->noise(); }
// mu: M x P row-major, sigma: M   (gp.hpp:159-167 per point)
void ref_query(void* h, const double* Xq, int64_t M, int D, double* mu, double* sigma)
{
    Scope s(G);
    const int P = G->dim_out();
    for (int64_t m = 0; m < M; ++m)
        G->query(Xq + m * D, D, mu + m * P, sigma + m);
}
void ref_mu(void* h, const double* x, int D, double* mu) { Scope s(G); G->mu(x, D, mu); }
double ref_sigma(void* h, const double* x, int D) { Scope s(G); return G->sigma(x, D); }
double ref_log_lik(void* h) { Scope s(G); return G->log_lik(); }
void ref_kernel_grad_log_lik(void* h, double* g) { Scope s(G); G->kernel_grad_log_lik(g); }
void ref_mean_grad_log_lik(void* h, double* g) { Scope s(G); G->mean_grad_log_lik(g); }
double ref_log_loo_cv(void* h) { Scope s(G); return G->log_loo_cv(); }
void ref_kernel_grad_log_loo_cv(void* h, double* g) { Scope s(G); G->kernel_grad_log_loo_cv(g); }
void ref_compute_inv_kernel(void* h) { Scope s(G); G->compute_inv_kernel(); }
int ref_inv_kernel_computed(void* h) { return G->inv_kernel_computed(); }
int64_t ref_nb_samples(void* h) { return G->nb_samples(); }
// which: 0 matrixL (N x N), 1 alpha (N x P), 2 obs_mean (N x P), 3 K^-1, 4 K, 5 mean_vector; column-major, ld = rows
void ref_get_matrix(void* h, int which, double* out) { Scope s(G); G->get_matrix(which, out); }
// which: 0 KernelLFOpt, 1 KernelLooOpt, 2 MeanLFOpt, 3 KernelMeanLFOpt  (all with opt::Rprop, ref_set_rprop)
void ref_optimize_hyperparams(void* h, int which) { Scope s(G); G->optimize(which); }
double ref_kernel_eval(void* h, const double* a, const double* b, int D, int i, int j) { Scope s(G); return G->kernel_eval(a, b, D, i, j); }
void ref_kernel_grad(void* h, const double* a, const double* b, int D, int i, int j, double* g) { Scope s(G); G->kernel_grad(a, b, D, i, j, g); }
// acquisition values at M points (which: 0 UCB, 1 GP_UCB, 2 EI; aggregator = first output)
void ref_acqui(void* h, int which, int iteration, const double* Xq, int64_t M, int D, double* out) { Scope s(G); G->acqui(which, iteration, Xq, M, D, out); }
// GP::save<TextArchive / BinaryArchive>(directory) and GP::load<...>(directory, recompute) of the reference (gp.hpp:439-511)
void ref_save(void* h, const char* dir, int binary) { Scope s(G); G->save(dir, binary != 0); }
void ref_load(void* h, const char* dir, int binary, int recompute) { Scope s(G); G->load(dir, binary != 0, recompute != 0); }
#undef G

} // extern "C"
