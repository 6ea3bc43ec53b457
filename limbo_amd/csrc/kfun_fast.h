// kfun_fast.h — the kernel functors' value from z = sum_d ((x1_d - x2_d) / ell_d)^2 with the branch-free exp(-h) of the
// kernel-matrix build (kbuild.hip) — shared with the data-flow launches of potrf.hip, which generate their own tiles of K
// (round 4: K is never written for the columns those launches factor).  squared_exp_ard.hpp:148-150, exp.hpp:97-102,
// matern_five_halves.hpp:104-113, matern_three_halves.hpp:101-107.
#pragma once
static __device__ __forceinline__ double exp_nonpos(double x)
{
    const double n = __builtin_rint(x * 1.44269504088896338700e+00);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 1.60590438368216145994e-10; // 1/13!
    p = fma(p, r, 2.08767569878680989792e-09);
    p = fma(p, r, 2.50521083854417187751e-08);
    p = fma(p, r, 2.75573192239858906526e-07);
    p = fma(p, r, 2.75573192239858906526e-06);
    p = fma(p, r, 2.48015873015873015873e-05);
    p = fma(p, r, 1.98412698412698412698e-04);
    p = fma(p, r, 1.38888888888888888889e-03);
    p = fma(p, r, 8.33333333333333333333e-03);
    p = fma(p, r, 4.16666666666666666667e-02);
    p = fma(p, r, 1.66666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return x < -745.2 ? 0.0 : ldexp(p, (int)n);
}
template <int KIND>
static __device__ __forceinline__ double kfun_fast(double z, double sf2)
{
    if (KIND == 0 || KIND == 3)
        return sf2 * exp_nonpos(-0.5 * z);
    if (KIND == 1) {
        const double t1 = 2.23606797749978969641 * sqrt(z);
        return sf2 * (1.0 + t1 + (5.0 / 3.0) * z) * exp_nonpos(-t1);
    }
    const double t = 1.73205080756887729353 * sqrt(z);
    return sf2 * (1.0 + t) * exp_nonpos(-t);
}

// the same by run-time kind (gpe_kernel_kind)
static __device__ __forceinline__ double kfun_fast_rt(int kind, double z, double sf2)
{
    if (kind == 0 || kind == 3)
        return kfun_fast<0>(z, sf2);
    if (kind == 1)
        return kfun_fast<1>(z, sf2);
    return kfun_fast<2>(z, sf2);
}
