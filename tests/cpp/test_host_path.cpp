// Driver for the host path of the drop-in model::GP (limbo/model/gp/host_small.hpp): a model below
// Params::gpu::min_n_for_gpu() samples never touches the device, so this runs on a box WITHOUT a GPU
// (tests/test_host_path.py compares what it prints with the reference itself, oracle/_ref, and with the C oracle).
//   test_host_path <input file> : kind mean P D n0 n1 M, then n1 rows of X (D) and Y (P), then M query points
// prints: after compute() on the first n0 samples and after add_sample() up to n1: L, alpha, log_lik, mu, sigma^2
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include <limbo/kernel/exp.hpp>
#include <limbo/kernel/matern_five_halves.hpp>
#include <limbo/kernel/squared_exp_ard.hpp>
#include <limbo/mean/constant.hpp>
#include <limbo/mean/data.hpp>
#include <limbo/mean/null_function.hpp>
#include <limbo/model/gp.hpp>

struct Params {
    struct kernel : public limbo::defaults::kernel {
        BO_PARAM(double, noise, 0.01);
    };
    struct kernel_squared_exp_ard : public limbo::defaults::kernel_squared_exp_ard {
    };
    struct kernel_maternfivehalves : public limbo::defaults::kernel_maternfivehalves {
    };
    struct kernel_exp : public limbo::defaults::kernel_exp {
    };
    struct mean_constant {
        BO_PARAM(double, constant, 1.0);
    };
    struct opt_rprop : public limbo::defaults::opt_rprop {
    };
    struct gpu {
        BO_PARAM(int, device, 0);
        BO_PARAM(int, min_n_for_gpu, 1 << 20); // everything on the host: no device is ever asked for
    };
};

template <typename GP>
static void dump(const char* tag, GP& gp, const std::vector<Eigen::VectorXd>& Q)
{
    const Eigen::MatrixXd& L = gp.matrixL();
    const Eigen::MatrixXd& a = gp.alpha();
    std::printf("%s n %d\nL", tag, (int)gp.nb_samples());
    for (int j = 0; j < (int)L.cols(); ++j)
        for (int i = 0; i < (int)L.rows(); ++i)
            std::printf(" %.17g", L(i, j));
    std::printf("\nalpha");
    for (int p = 0; p < (int)a.cols(); ++p)
        for (int i = 0; i < (int)a.rows(); ++i)
            std::printf(" %.17g", a(i, p));
    std::printf("\nlog_lik %.17g\nmu", gp.compute_log_lik());
    for (const auto& q : Q) {
        Eigen::VectorXd m = gp.mu(q);
        for (int p = 0; p < (int)m.size(); ++p)
            std::printf(" %.17g", m(p));
    }
    std::printf("\nsigma");
    for (const auto& q : Q)
        std::printf(" %.17g", gp.sigma(q));
    std::printf("\nquery");
    for (const auto& q : Q) { // query() == (mu(), sigma()) bitwise (test_gp.cpp:502-510)
        auto r = gp.query(q);
        std::printf(" %.17g %.17g", std::get<0>(r)(0), std::get<1>(r));
    }
    std::printf("\nstatus %d\n", gp.last_status());
}

template <typename Kernel, typename Mean>
static int run(FILE* f, int P, int D, int n0, int n1, int M)
{
    std::vector<Eigen::VectorXd> X, Y, Q;
    for (int i = 0; i < n1; ++i) {
        Eigen::VectorXd x(D), y(P);
        for (int d = 0; d < D; ++d)
            if (std::fscanf(f, "%lf", &x(d)) != 1)
                return 2;
        for (int p = 0; p < P; ++p)
            if (std::fscanf(f, "%lf", &y(p)) != 1)
                return 2;
        X.push_back(x);
        Y.push_back(y);
    }
    for (int m = 0; m < M; ++m) {
        Eigen::VectorXd q(D);
        for (int d = 0; d < D; ++d)
            if (std::fscanf(f, "%lf", &q(d)) != 1)
                return 2;
        Q.push_back(q);
    }
    limbo::model::GP<Params, Kernel, Mean> gp(D, P);
    gp.compute(std::vector<Eigen::VectorXd>(X.begin(), X.begin() + n0), std::vector<Eigen::VectorXd>(Y.begin(), Y.begin() + n0));
    dump("full", gp, Q);
    for (int i = n0; i < n1; ++i)
        gp.add_sample(X[i], Y[i]);
    dump("incremental", gp, Q);
    limbo::model::GP<Params, Kernel, Mean> copy(gp); // value semantics: the copy is a host model too
    copy.recompute(true, true);
    dump("copy_recomputed", copy, Q);
    Eigen::MatrixXd mu;
    Eigen::VectorXd s2;
    gp.query_batch(std::vector<Eigen::VectorXd>(Q.begin(), Q.begin() + 2), mu, s2); // 2 x n < the batch crossover: host loop
    std::printf("batch2 %.17g %.17g %.17g %.17g\n", mu(0, 0), mu(1, 0), s2(0), s2(1));
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2)
        return 1;
    FILE* f = std::fopen(argv[1], "r");
    if (!f)
        return 1;
    int kind, mean, P, D, n0, n1, M;
    if (std::fscanf(f, "%d %d %d %d %d %d %d", &kind, &mean, &P, &D, &n0, &n1, &M) != 7)
        return 2;
    using namespace limbo;
    if (kind == 0 && mean == 0)
        return run<kernel::SquaredExpARD<Params>, mean::Data<Params>>(f, P, D, n0, n1, M);
    if (kind == 0 && mean == 1)
        return run<kernel::SquaredExpARD<Params>, mean::NullFunction<Params>>(f, P, D, n0, n1, M);
    if (kind == 1 && mean == 0)
        return run<kernel::MaternFiveHalves<Params>, mean::Data<Params>>(f, P, D, n0, n1, M);
    if (kind == 1 && mean == 2)
        return run<kernel::MaternFiveHalves<Params>, mean::Constant<Params>>(f, P, D, n0, n1, M);
    if (kind == 3 && mean == 0)
        return run<kernel::Exp<Params>, mean::Data<Params>>(f, P, D, n0, n1, M);
    return 3;
}
