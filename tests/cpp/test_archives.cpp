// Cross-implementation archive driver (VERDICT r4 "N3 for real"): the drop-in model::GP reads directories
// that STOCK limbo wrote (oracle/_ref/libref.so = /root/reference/src compiled unmodified, driven from
// tests/test_archives.py) and writes directories stock limbo reads back.  The format is limbo's
// (serialize/text_archive.hpp:63-151, binary_archive.hpp:66-161; what GP::save / GP::load put in a directory:
// gp.hpp:448-511).  Below Params::gpu::min_n_for_gpu() samples the model lives on the host, so the small cases run on a
// box without a GPU; LIMBO_AMD_MIN_N_FOR_GPU=0 sends the same binary through the device (-m gpu test).
//
//   test_archives load <kind> <mean> <dir_in> <text|bin> <recompute 0|1> <query file> <dir_out>
//       GP::load<A>(dir_in, recompute); print n, h_params, log_lik, mu / sigma^2 at the query points;
//       then GP::save<TextArchive>(dir_out/text) and GP::save<BinaryArchive>(dir_out/bin)
//   test_archives compute <kind> <mean> <data file> <dir_out>
//       data file: P D n M, n rows of X (D) Y (P), M query points; compute(), print as above, save both formats
//   test_archives acqui <kind> <mean> <dir_in> <text|bin> <query file> <iteration>
//       GP::load<A>(dir_in, false); print acqui::UCB / GP_UCB / EI ::batch() at the query points — the
//       batched acquisition layer against limbo's own functors over limbo's own model (tests/test_archives.py)
// kind: 0 SquaredExpARD, 1 MaternFiveHalves; mean: 0 Data, 2 Constant
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <limbo/acqui/ei.hpp>
#include <limbo/acqui/gp_ucb.hpp>
#include <limbo/acqui/ucb.hpp>
#include <limbo/kernel/matern_five_halves.hpp>
#include <limbo/kernel/squared_exp_ard.hpp>
#include <limbo/mean/constant.hpp>
#include <limbo/mean/data.hpp>
#include <limbo/model/gp.hpp>
#include <limbo/serialize/binary_archive.hpp>
#include <limbo/serialize/text_archive.hpp>

struct Params {
    struct kernel : public limbo::defaults::kernel {
        BO_PARAM(double, noise, 0.01);
    };
    struct kernel_squared_exp_ard : public limbo::defaults::kernel_squared_exp_ard {
    };
    struct kernel_maternfivehalves : public limbo::defaults::kernel_maternfivehalves {
    };
    struct mean_constant {
        BO_PARAM(double, constant, 1.0);
    };
    struct opt_rprop : public limbo::defaults::opt_rprop {
    };
    struct acqui_ucb : public limbo::defaults::acqui_ucb {
    };
    struct acqui_gpucb : public limbo::defaults::acqui_gpucb {
    };
    struct acqui_ei : public limbo::defaults::acqui_ei {
    };
};

static bool read_points(FILE* f, int M, int D, std::vector<Eigen::VectorXd>& Q)
{
    for (int m = 0; m < M; ++m) {
        Eigen::VectorXd q(D);
        for (int d = 0; d < D; ++d)
            if (std::fscanf(f, "%lf", &q(d)) != 1)
                return false;
        Q.push_back(q);
    }
    return true;
}

template <typename GP>
static void report_and_save(GP& gp, const std::vector<Eigen::VectorXd>& Q, const std::string& dir_out)
{
    std::printf("n %d dim_in %d dim_out %d\nh_params", (int)gp.nb_samples(), gp.dim_in(), gp.dim_out());
    Eigen::VectorXd h = gp.kernel_function().h_params();
    for (int i = 0; i < (int)h.size(); ++i)
        std::printf(" %.17g", h(i));
    std::printf("\nlog_lik %.17g\nmu", gp.compute_log_lik());
    for (const auto& q : Q) {
        Eigen::VectorXd m = gp.mu(q);
        for (int p = 0; p < (int)m.size(); ++p)
            std::printf(" %.17g", m(p));
    }
    std::printf("\nsigma");
    for (const auto& q : Q)
        std::printf(" %.17g", gp.sigma(q));
    std::printf("\n");
    gp.template save<limbo::serialize::TextArchive>(dir_out + "/text");
    gp.template save<limbo::serialize::BinaryArchive>(dir_out + "/bin");
}

template <typename Kernel, typename Mean>
static int run_load(char** a)
{
    const std::string dir_in = a[0];
    const bool binary = std::strcmp(a[1], "bin") == 0;
    const bool recompute = std::atoi(a[2]) != 0;
    limbo::model::GP<Params, Kernel, Mean> gp;
    if (binary)
        gp.template load<limbo::serialize::BinaryArchive>(dir_in, recompute);
    else
        gp.template load<limbo::serialize::TextArchive>(dir_in, recompute);
    FILE* f = std::fopen(a[3], "r");
    int M = 0;
    std::vector<Eigen::VectorXd> Q;
    if (!f || std::fscanf(f, "%d", &M) != 1 || !read_points(f, M, gp.dim_in(), Q))
        return 2;
    std::fclose(f);
    report_and_save(gp, Q, a[4]);
    return 0;
}

template <typename Kernel, typename Mean>
static int run_compute(char** a)
{
    FILE* f = std::fopen(a[0], "r");
    int P, D, n, M;
    if (!f || std::fscanf(f, "%d %d %d %d", &P, &D, &n, &M) != 4)
        return 2;
    std::vector<Eigen::VectorXd> X, Y, Q;
    for (int i = 0; i < n; ++i) {
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
    if (!read_points(f, M, D, Q))
        return 2;
    std::fclose(f);
    limbo::model::GP<Params, Kernel, Mean> gp(D, P);
    gp.compute(X, Y);
    report_and_save(gp, Q, a[1]);
    return 0;
}

template <typename Kernel, typename Mean>
static int run_acqui(char** a)
{
    using GP_t = limbo::model::GP<Params, Kernel, Mean>;
    GP_t gp;
    if (std::strcmp(a[1], "bin") == 0)
        gp.template load<limbo::serialize::BinaryArchive>(std::string(a[0]), false);
    else
        gp.template load<limbo::serialize::TextArchive>(std::string(a[0]), false);
    FILE* f = std::fopen(a[2], "r");
    int M = 0;
    std::vector<Eigen::VectorXd> Q;
    if (!f || std::fscanf(f, "%d", &M) != 1 || !read_points(f, M, gp.dim_in(), Q))
        return 2;
    std::fclose(f);
    const int iteration = std::atoi(a[3]);
    auto first = [](const Eigen::VectorXd& v) { return v(0); };
    limbo::acqui::UCB<Params, GP_t> ucb(gp, iteration);
    limbo::acqui::GP_UCB<Params, GP_t> gpucb(gp, iteration);
    limbo::acqui::EI<Params, GP_t> ei(gp, iteration);
    const std::vector<double> vals[3] = {ucb.batch(Q, first), gpucb.batch(Q, first), ei.batch(Q, first)};
    const char* names[3] = {"ucb", "gp_ucb", "ei"};
    for (int k = 0; k < 3; ++k) {
        std::printf("%s", names[k]);
        for (double v : vals[k])
            std::printf(" %.17g", v);
        std::printf("\n");
    }
    return 0;
}

template <typename Kernel, typename Mean>
static int dispatch(const char* cmd, char** rest)
{
    if (std::strcmp(cmd, "acqui") == 0)
        return run_acqui<Kernel, Mean>(rest);
    return std::strcmp(cmd, "load") == 0 ? run_load<Kernel, Mean>(rest) : run_compute<Kernel, Mean>(rest);
}

int main(int argc, char** argv)
{
    if (argc < 6)
        return 1;
    const char* cmd = argv[1];
    const int kind = std::atoi(argv[2]), mean = std::atoi(argv[3]);
    if ((std::strcmp(cmd, "load") == 0 && argc != 9) || (std::strcmp(cmd, "compute") == 0 && argc != 6) || (std::strcmp(cmd, "acqui") == 0 && argc != 8))
        return 1;
    using namespace limbo;
    if (kind == 0 && mean == 0)
        return dispatch<kernel::SquaredExpARD<Params>, mean::Data<Params>>(cmd, argv + 4);
    if (kind == 1 && mean == 2)
        return dispatch<kernel::MaternFiveHalves<Params>, mean::Constant<Params>>(cmd, argv + 4);
    if (kind == 1 && mean == 0)
        return dispatch<kernel::MaternFiveHalves<Params>, mean::Data<Params>>(cmd, argv + 4);
    return 3;
}
