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

extern "C" int gd_cld_r_table(double m_inv, double beta_0, double beta_1, double vv_gamma, double numerical_eps, int is_rk,
                              double R_dt, size_t n, double *out_host) {
    GD_CHECK(out_host && n > 0 && R_dt > 0 && m_inv > 0, "gd_cld_r_table: bad argument");
    Cld s{m_inv, 2.0 / std::sqrt(m_inv), beta_0, beta_1};
    M2 x{std::sqrt(numerical_eps), 0.0, 0.0, std::sqrt(vv_gamma / m_inv + numerical_eps)};
    // grid of the scan: linspace(0, 1 + R_dt, n, endpoint=False)
    const double h = (1.0 + R_dt) / (double)n;
    for (size_t i = 0; i < n; ++i) {
        const double t = h * (double)i;
        out_host[4 * i + 0] = x.a; out_host[4 * i + 1] = x.b; out_host[4 * i + 2] = x.c; out_host[4 * i + 3] = x.d;
        if (is_rk) {
            const M2 g1 = s.ode(x, t);
            const M2 g2 = s.ode(axpy(x, R_dt / 2, g1), t + R_dt / 2);
            const M2 g3 = s.ode(axpy(x, R_dt / 2, g2), t + R_dt / 2);
            const M2 g4 = s.ode(axpy(x, R_dt, g3), t + R_dt);
            x = {x.a + R_dt / 6 * (g1.a + 2 * g2.a + 2 * g3.a + g4.a), x.b + R_dt / 6 * (g1.b + 2 * g2.b + 2 * g3.b + g4.b),
                 x.c + R_dt / 6 * (g1.c + 2 * g2.c + 2 * g3.c + g4.c), x.d + R_dt / 6 * (g1.d + 2 * g2.d + 2 * g3.d + g4.d)};
        } else {
            // "mid point" Euler of the reference: coefficients averaged over [t, t + R_dt]
            const M2 F0 = s.F(t), F1 = s.F(t + R_dt);
            const M2 Fm{(F0.a + F1.a) / 2, (F0.b + F1.b) / 2, (F0.c + F1.c) / 2, (F0.d + F1.d) / 2};
            const double gm = (s.g(t) + s.g(t + R_dt)) / 2;
            const double det = x.a * x.d - x.b * x.c;
            M2 r = mul(Fm, x);
            r.c += 0.5 * gm * gm * (-x.b / det);
            r.d += 0.5 * gm * gm * (x.a / det);
            x = axpy(x, R_dt, r);
        }
    }
    return 0;
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
