// Host-side (no GPU) SDE math of the C ABI: the R(t) table of the CLD SDE.
//
// The reference integrates dR/dt = F R + 1/2 G G^T R^-T with 10^5 / 10^6 sequential fixed steps inside a
// jax.lax.scan (cld_jax/sde_lib.py:93-118; RK4 body cld_jax/deis.py:5-17) and caches the table in a pickle.
// Here it is a plain float64 loop (a few hundred ms), so no cache file is needed.
#include <cmath>
#include <cstddef>

#include "../../include/gddim_amd.h"
#include "common.h"

namespace {

struct M2 {
    double a, b, c, d;  // [[a, b], [c, d]]
};

inline M2 mul(const M2 &x, const M2 &y) {
    return {x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}
inline M2 axpy(const M2 &x, double s, const M2 &y) { return {x.a + s * y.a, x.b + s * y.b, x.c + s * y.c, x.d + s * y.d}; }

struct Cld {
    double m_inv, Gamma, beta_0, beta_1;
    double beta(double t) const { return beta_0 + beta_1 * t; }
    M2 F(double t) const {
        const double b = beta(t);
        return {0.0, b * m_inv, -b, -Gamma * b * m_inv};
    }
    double g(double t) const { return std::sqrt(2.0 * Gamma * beta(t)); }
    // F x + 1/2 G G^T inv(x)^T with G = diag(0, g)
    M2 ode(const M2 &x, double t) const {
        const double det = x.a * x.d - x.b * x.c;
        // inv(x) = [[d, -b], [-c, a]] / det ; inv(x)^T = [[d, -c], [-b, a]] / det ; G G^T picks its second row
        const double gg = g(t) * g(t);
        M2 r = mul(F(t), x);
        r.c += 0.5 * gg * (-x.b / det);
        r.d += 0.5 * gg * (x.a / det);
        return r;
    }
};

}  // namespace

// ---- R(t) scan, in the arithmetic the configuration asks for ------------------------------------------------------------------
// The reference runs the scan in the dtype jax is configured with: float32 when config.model.x64 is False (its default,
// cld_jax/configs/default_cifar10_config.py; asserted at cld_jax/sde_lib.py:112), float64 otherwise.  A 10^6-step float32
// recurrence is only reproducible if the operations happen in the order the expressions are written, so the body below follows
//   _ode_fn        cld_jax/sde_lib.py:94-97    cur_F @ _x + 0.5 * cur_G @ cur_G.T @ inv_2x2(_x).T     (left to right)
//   runge_kutta    cld_jax/deis.py:5-17        x + grad * dt / 2 ; t + dt / 2 ; x + dt / 6 * (g1 + 2 g2 + 2 g3 + g4)
//   mid-point step cld_jax/sde_lib.py:103-107  carry + R_dt * (F_m @ carry + 0.5 * G_m @ G_m @ inv(carry).T)
//   inv_2x2        cld_jax/sde_lib.py:17-26    [[d, -b], [-c, a]] * (1 / (a d - b c))
// term by term in T, with no fused multiply-add (a 2-term dot product is commutative, so only contraction could change it).
#pragma clang fp contract(off)
namespace {

template <class T>
struct Mat2 {
    T a, b, c, d;  // [[a, b], [c, d]]
};

template <class T>
inline Mat2<T> matmul(const Mat2<T> &x, const Mat2<T> &y) {
    return {x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}

template <class T>
inline Mat2<T> inv_t(const Mat2<T> &m) {  // inv_2x2(m).T
    const T coef = (T)1 / (m.a * m.d - m.b * m.c);
    return {m.d * coef, -m.c * coef, -m.b * coef, m.a * coef};
}

template <class T>
struct CldT {
    T m_inv, Gamma, beta_0, beta_1;
    T beta(T t) const { return beta_0 + beta_1 * t; }
    Mat2<T> F(T t) const {
        const T b = beta(t);
        return {(T)0, b * m_inv, -b, -Gamma * b * m_inv};
    }
    T g(T t) const { return std::sqrt((T)2 * Gamma * beta(t)); }  // s_G = diag(0, sqrt(2 Gamma beta))
    Mat2<T> drift(const Mat2<T> &F, T g0, const Mat2<T> &x) const {
        const Mat2<T> halfGG{(T)0, (T)0, (T)0, ((T)0.5 * g0) * g0};  // (0.5 * G) @ G.T, G diagonal
        const Mat2<T> fx = matmul(F, x), gx = matmul(halfGG, inv_t(x));
        return {fx.a + gx.a, fx.b + gx.b, fx.c + gx.c, fx.d + gx.d};
    }
    Mat2<T> ode(const Mat2<T> &x, T t) const { return drift(F(t), g(t), x); }
};

template <class T>
inline Mat2<T> step_to(const Mat2<T> &x, const Mat2<T> &g, T h, T div) {  // x + g * h / div
    return {x.a + g.a * h / div, x.b + g.b * h / div, x.c + g.c * h / div, x.d + g.d * h / div};
}

template <class T>
int r_table(double m_inv, double beta_0, double beta_1, double vv_gamma, double numerical_eps, int is_rk, double R_dt, size_t n,
            T *out) {
    const CldT<T> s{(T)m_inv, (T)(2.0 / std::sqrt(m_inv)), (T)beta_0, (T)beta_1};  // Gamma = 2 / np.sqrt(m_inv): a Python float
    // R_0 (sde_lib.py:61-66): jnp.sqrt of Python floats -- the sums are formed in double, rounded to T, then rooted in T
    Mat2<T> x{std::sqrt((T)numerical_eps), (T)0, (T)0, std::sqrt((T)(vv_gamma / m_inv + numerical_eps))};
    // grid jnp.linspace(0, 1 + R_dt, n, endpoint=False) = start * (1 - step) + stop * step with step = iota / n (jax 0.2.8)
    const T stop = (T)(1.0 + R_dt), dt = (T)R_dt;
    for (size_t i = 0; i < n; ++i) {
        const T step = (T)i / (T)n;
        const T t = (T)0 * ((T)1 - step) + stop * step;
        out[4 * i + 0] = x.a; out[4 * i + 1] = x.b; out[4 * i + 2] = x.c; out[4 * i + 3] = x.d;
        if (is_rk) {
            const Mat2<T> g1 = s.ode(x, t);
            const Mat2<T> g2 = s.ode(step_to(x, g1, dt, (T)2), t + dt / (T)2);
            const Mat2<T> g3 = s.ode(step_to(x, g2, dt, (T)2), t + dt / (T)2);
            const Mat2<T> g4 = s.ode(step_to(x, g3, dt, (T)1), t + dt);
            const T w = dt / (T)6;
            x = {x.a + w * (g1.a + (T)2 * g2.a + (T)2 * g3.a + g4.a), x.b + w * (g1.b + (T)2 * g2.b + (T)2 * g3.b + g4.b),
                 x.c + w * (g1.c + (T)2 * g2.c + (T)2 * g3.c + g4.c), x.d + w * (g1.d + (T)2 * g2.d + (T)2 * g3.d + g4.d)};
        } else {
            const Mat2<T> F0 = s.F(t), F1 = s.F(t + dt);
            const Mat2<T> Fm{(F0.a + F1.a) / (T)2, (F0.b + F1.b) / (T)2, (F0.c + F1.c) / (T)2, (F0.d + F1.d) / (T)2};
            const T gm = (s.g(t) + s.g(t + dt)) / (T)2;
            const Mat2<T> r = s.drift(Fm, gm, x);
            x = {x.a + dt * r.a, x.b + dt * r.b, x.c + dt * r.c, x.d + dt * r.d};
        }
    }
    return 0;
}

}  // namespace

extern "C" int gd_cld_r_table(double m_inv, double beta_0, double beta_1, double vv_gamma, double numerical_eps, int is_rk,
                              double R_dt, size_t n, double *out_host) {
    GD_CHECK(out_host && n > 0 && R_dt > 0 && m_inv > 0, "gd_cld_r_table: bad argument");
    return r_table<double>(m_inv, beta_0, beta_1, vv_gamma, numerical_eps, is_rk, R_dt, n, out_host);
}

extern "C" int gd_cld_r_table_f32(double m_inv, double beta_0, double beta_1, double vv_gamma, double numerical_eps, int is_rk,
                                  double R_dt, size_t n, float *out_host) {
    GD_CHECK(out_host && n > 0 && R_dt > 0 && m_inv > 0, "gd_cld_r_table_f32: bad argument");
    return r_table<float>(m_inv, beta_0, beta_1, vv_gamma, numerical_eps, is_rk, R_dt, n, out_host);
}

// psi2 table of the rotated-frame sampler (cld_jax/sampling.py:272-285): d psi2/dt = psi1(t)^-1 F2(t) psi1(t) psi2.
extern "C" int gd_cld_psi2_table(double m_inv, double beta_0, double beta_1, double dt, size_t n, double *out_host) {
    GD_CHECK(out_host && n > 0 && dt > 0 && m_inv > 0, "gd_cld_psi2_table: bad argument");
    const double Gamma = 2.0 / std::sqrt(m_inv), ism = std::sqrt(m_inv), sm = std::sqrt(1.0 / m_inv);
    auto A = [&](double t) {  // psi1(t)^-1 F2(t) psi1(t)
        const double bi = beta_0 * t + 0.5 * beta_1 * t * t;  // beta_int(t) - beta_int(0)
        const double c = std::cos(bi * ism), sn = std::sin(bi * ism);
        const M2 p1{c, ism * sn, -sm * sn, c};      // s_f1_psi(0, t)
        const M2 ip1{c, -ism * sn, sm * sn, c};     // s_f1_psi(t, 0)
        const M2 F2{0.0, 0.0, 0.0, -Gamma * (beta_0 + beta_1 * t) * m_inv};
        return mul(ip1, mul(F2, p1));
    };
    M2 x{1.0, 0.0, 0.0, 1.0};
    double t = 0.0;
    for (size_t i = 0; i < n; ++i) {
        out_host[4 * i + 0] = x.a; out_host[4 * i + 1] = x.b; out_host[4 * i + 2] = x.c; out_host[4 * i + 3] = x.d;
        const M2 g1 = mul(A(t), x);
        const M2 g2 = mul(A(t + dt / 2), axpy(x, dt / 2, g1));
        const M2 g3 = mul(A(t + dt / 2), axpy(x, dt / 2, g2));
        const M2 g4 = mul(A(t + dt), axpy(x, dt, g3));
        x = {x.a + dt / 6 * (g1.a + 2 * g2.a + 2 * g3.a + g4.a), x.b + dt / 6 * (g1.b + 2 * g2.b + 2 * g3.b + g4.b),
             x.c + dt / 6 * (g1.c + 2 * g2.c + 2 * g3.c + g4.c), x.d + dt / 6 * (g1.d + 2 * g2.d + 2 * g3.d + g4.d)};
        t += dt;  // the reference accumulates cur_t + dt in the scan carry
    }
    return 0;
}

// ---- LambdaSDE tables (cld_jax/sde_lib.py:334-399) -------------------------------------------------------------------------
namespace {

// get_interp_fn (sde_lib.py:32-43) on a knot table of 2x2 matrices: searchsorted(side='right') clipped to [1, n-1], linear
struct Knots {
    const double *xp, *fp;
    size_t n;
    M2 at(double x) const {
        size_t lo = 0, hi = n;  // first index with xp[i] > x
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (xp[mid] <= x) lo = mid + 1; else hi = mid;
        }
        size_t i = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
        const double dx = xp[i] - xp[i - 1];
        const double *a = fp + 4 * (i - 1), *b = fp + 4 * i;
        if (dx == 0) return {b[0], b[1], b[2], b[3]};
        const double w = (x - xp[i - 1]) / dx;
        return {a[0] + w * (b[0] - a[0]), a[1] + w * (b[1] - a[1]), a[2] + w * (b[2] - a[2]), a[3] + w * (b[3] - a[3])};
    }
};

// \hat F(t) = F + (1 + lambda^2)/2 G G^T Sigma(t)^-1, Sigma = R R^T (sde_lib.py:351-355)
inline M2 hat_F(const Cld &s, const Knots &R, double lambda, double t) {
    const M2 r = R.at(t);
    const M2 cov{r.a * r.a + r.b * r.b, r.a * r.c + r.b * r.d, r.c * r.a + r.d * r.b, r.c * r.c + r.d * r.d};
    const double det = cov.a * cov.d - cov.b * cov.c;
    const double gg = s.g(t) * s.g(t) * 0.5 * (1.0 + lambda * lambda);
    M2 f = s.F(t);
    f.c += gg * (-cov.c / det);  // second row of G G^T inv(cov): gg * [ -c, a ] / det
    f.d += gg * (cov.a / det);
    return f;
}

inline M2 add4(const M2 &x, double h, const M2 &g1, const M2 &g2, const M2 &g3, const M2 &g4) {
    return {x.a + h / 6 * (g1.a + 2 * g2.a + 2 * g3.a + g4.a), x.b + h / 6 * (g1.b + 2 * g2.b + 2 * g3.b + g4.b),
            x.c + h / 6 * (g1.c + 2 * g2.c + 2 * g3.c + g4.c), x.d + h / 6 * (g1.d + 2 * g2.d + 2 * g3.d + g4.d)};
}

}  // namespace

extern "C" int gd_cld_lambda_hat_psi_table(double m_inv, double beta_0, double beta_1, double lambda, const double *r_xp,
                                           const double *r_fp, size_t r_n, double dt, size_t n, double *out_host) {
    GD_CHECK(out_host && r_xp && r_fp && r_n >= 2 && n > 0 && dt > 0, "gd_cld_lambda_hat_psi_table: bad argument");
    const Cld s{m_inv, 2.0 / std::sqrt(m_inv), beta_0, beta_1};
    const Knots R{r_xp, r_fp, r_n};
    const double h = (1.0 + dt) / (double)n;  // grid linspace(0, 1 + dt, n, endpoint=False); the RK4 step is dt
    M2 x{1.0, 0.0, 0.0, 1.0};
    for (size_t i = 0; i < n; ++i) {
        const double t = h * (double)i;
        out_host[4 * i + 0] = x.a; out_host[4 * i + 1] = x.b; out_host[4 * i + 2] = x.c; out_host[4 * i + 3] = x.d;
        const M2 g1 = mul(hat_F(s, R, lambda, t), x);
        const M2 g2 = mul(hat_F(s, R, lambda, t + dt / 2), axpy(x, dt / 2, g1));
        const M2 g3 = mul(hat_F(s, R, lambda, t + dt / 2), axpy(x, dt / 2, g2));
        const M2 g4 = mul(hat_F(s, R, lambda, t + dt), axpy(x, dt, g3));
        x = add4(x, dt, g1, g2, g3, g4);
    }
    return 0;
}

// cond_rev_cov (sde_lib.py:382-399) exactly as coded: dP/dtau = \hat F P + P \hat F (no transpose) + sign lambda^2 G G^T from
// P = 0, n_step RK4 steps of size (t - s)/n_step evaluated on the grid s + i (t - s)/(n_step + 1).
extern "C" int gd_cld_lambda_cond_rev_cov(double m_inv, double beta_0, double beta_1, double lambda, const double *r_xp,
                                          const double *r_fp, size_t r_n, double s_time, double t_time, int n_step,
                                          double *out4) {
    GD_CHECK(out4 && r_xp && r_fp && r_n >= 2 && n_step > 0, "gd_cld_lambda_cond_rev_cov: bad argument");
    const Cld s{m_inv, 2.0 / std::sqrt(m_inv), beta_0, beta_1};
    const Knots R{r_xp, r_fp, r_n};
    const double sign = t_time > s_time ? 1.0 : -1.0;
    const double dt = (t_time - s_time) / (double)n_step;
    const double hgrid = (t_time - s_time) / (double)(n_step + 1);
    auto ode = [&](const M2 &x, double t) {
        const M2 f = hat_F(s, R, lambda, t);
        M2 r = mul(f, x);
        const M2 xf = mul(x, f);
        r.a += xf.a; r.b += xf.b; r.c += xf.c; r.d += xf.d;
        r.d += sign * lambda * lambda * s.g(t) * s.g(t);
        return r;
    };
    M2 x{0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < n_step; ++i) {
        const double t = s_time + hgrid * (double)i;
        const M2 g1 = ode(x, t);
        const M2 g2 = ode(axpy(x, dt / 2, g1), t + dt / 2);
        const M2 g3 = ode(axpy(x, dt / 2, g2), t + dt / 2);
        const M2 g4 = ode(axpy(x, dt, g3), t + dt);
        x = add4(x, dt, g1, g2, g3, g4);
    }
    out4[0] = x.a; out4[1] = x.b; out4[2] = x.c; out4[3] = x.d;
    return 0;
}
