/*
 * ORACLE (test infrastructure only -- never imported by the product path).
 *
 * CPU restatement of the R(t) scan of the reference's CLD SDE:
 *   /root/reference/cld_jax/sde_lib.py:93-118   (_get_s_R_fn: _ode_fn, scan_body_fn, lax.scan)
 *   /root/reference/cld_jax/deis.py:5-17        (runge_kutta)
 *   /root/reference/cld_jax/sde_lib.py:17-26    (inv_2x2)
 *
 * dR/dt = F(t) R + 1/2 G(t) G(t)^T inv(R)^T, R(0) = R_0, integrated with a fixed step
 * R_dt over the grid ts = linspace(0, 1 + R_dt, n, endpoint=False); the scan emits the
 * carry BEFORE each step, so out[i] is the state after i steps.
 *
 * Two instantiations: float (the reference runs with x64=False) and double.
 * PARITY UNPINNED: the reference ships no golden R table and cannot run here (no jax).
 */
#include <stddef.h>

#define SQRT_float(x) __builtin_sqrtf(x)
#define SQRT_double(x) __builtin_sqrt(x)

#define DEFINE_SCAN(NAME, T)                                                                   \
    static void NAME##_inv2(const T m[4], T o[4]) {                                            \
        /* sde_lib.py:17-26: coef = 1/(a d - b c); [[d,-b],[-c,a]] * coef */                   \
        T coef = (T)1.0 / (m[0] * m[3] - m[1] * m[2]);                                         \
        o[0] = m[3] * coef; o[1] = -m[1] * coef; o[2] = -m[2] * coef; o[3] = m[0] * coef;      \
    }                                                                                          \
    static void NAME##_mm(const T a[4], const T b[4], T o[4]) {                                \
        o[0] = a[0] * b[0] + a[1] * b[2]; o[1] = a[0] * b[1] + a[1] * b[3];                    \
        o[2] = a[2] * b[0] + a[3] * b[2]; o[3] = a[2] * b[1] + a[3] * b[3];                    \
    }                                                                                          \
    static void NAME##_FG(T t, T m_inv, T Gamma, T beta0, T beta1, T F[4], T GG[4], int gt) {  \
        /* sde_lib.py:214-234: F = [[0, b m_inv],[-b, -Gamma b m_inv]]; G = diag(0, sqrt(2 Gamma b)) */ \
        T b = beta0 + beta1 * t;                                                               \
        F[0] = 0; F[1] = b * m_inv; F[2] = -b; F[3] = -Gamma * b * m_inv;                      \
        (void)gt;                                                                              \
        /* (0.5 * cur_G) @ cur_G.T as written at :96 (left to right; G diagonal): (0.5 g) g, g = sqrt(2 Gamma b) in T */ \
        { T g = (T)SQRT_##T((T)2.0 * Gamma * b); GG[0] = 0; GG[1] = 0; GG[2] = 0; GG[3] = ((T)0.5 * g) * g; } \
    }                                                                                          \
    static void NAME##_ode(const T x[4], T t, T m_inv, T Gamma, T b0, T b1, T g[4]) {          \
        /* sde_lib.py:94-97: F @ x + 0.5 * G @ G.T @ inv_2x2(x).T */                           \
        T F[4], GG[4], inv[4], invT[4], a[4], c[4];                                            \
        NAME##_FG(t, m_inv, Gamma, b0, b1, F, GG, 0);                                          \
        NAME##_inv2(x, inv);                                                                   \
        invT[0] = inv[0]; invT[1] = inv[2]; invT[2] = inv[1]; invT[3] = inv[3];                \
        NAME##_mm(F, x, a);                                                                    \
        NAME##_mm(GG, invT, c);                                                                \
        for (int i = 0; i < 4; ++i) g[i] = a[i] + c[i];                                        \
    }                                                                                          \
    /* out: n x 4 row-major 2x2 matrices. is_rk: RK4 (deis.py:5-17) else the "mid point"     */ \
    /* Euler variant (sde_lib.py:103-107).                                                    */ \
    void NAME(const T R0[4], T m_inv, T Gamma, T beta0, T beta1, T R_dt, const T *ts,          \
              size_t n, int is_rk, T *out) {                                                   \
        T x[4] = {R0[0], R0[1], R0[2], R0[3]};                                                 \
        for (size_t i = 0; i < n; ++i) {                                                       \
            T t = ts[i];                                                                       \
            for (int k = 0; k < 4; ++k) out[4 * i + k] = x[k];                                 \
            if (is_rk) {                                                                       \
                T g1[4], g2[4], g3[4], g4[4], x2[4], x3[4], x4[4];                             \
                NAME##_ode(x, t, m_inv, Gamma, beta0, beta1, g1);                              \
                for (int k = 0; k < 4; ++k) x2[k] = x[k] + g1[k] * R_dt / (T)2;                \
                NAME##_ode(x2, t + R_dt / (T)2, m_inv, Gamma, beta0, beta1, g2);               \
                for (int k = 0; k < 4; ++k) x3[k] = x[k] + g2[k] * R_dt / (T)2;                \
                NAME##_ode(x3, t + R_dt / (T)2, m_inv, Gamma, beta0, beta1, g3);               \
                for (int k = 0; k < 4; ++k) x4[k] = x[k] + g3[k] * R_dt;                       \
                NAME##_ode(x4, t + R_dt, m_inv, Gamma, beta0, beta1, g4);                      \
                for (int k = 0; k < 4; ++k)                                                    \
                    x[k] = x[k] + R_dt / (T)6 * (g1[k] + (T)2 * g2[k] + (T)2 * g3[k] + g4[k]); \
            } else {                                                                           \
                T F0[4], F1[4], G0[4], G1[4], F[4], GG[4], inv[4], invT[4], a[4], c[4];        \
                NAME##_FG(t, m_inv, Gamma, beta0, beta1, F0, G0, 0);                           \
                NAME##_FG(t + R_dt, m_inv, Gamma, beta0, beta1, F1, G1, 0);                    \
                for (int k = 0; k < 4; ++k) F[k] = (F0[k] + F1[k]) / (T)2;                     \
                /* cur_G = (G(t)+G(t+dt))/2 then cur_G @ cur_G: with beta_1 = 0 both equal */  \
                {                                                                              \
                    T b_a = beta0 + beta1 * t, b_b = beta0 + beta1 * (t + R_dt);               \
                    T ga = (T)SQRT_##T((T)2.0 * Gamma * b_a);                                  \
                    T gb = (T)SQRT_##T((T)2.0 * Gamma * b_b);                                  \
                    T g = (ga + gb) / (T)2;                                                    \
                    GG[0] = 0; GG[1] = 0; GG[2] = 0; GG[3] = ((T)0.5 * g) * g;                 \
                }                                                                              \
                NAME##_inv2(x, inv);                                                           \
                invT[0] = inv[0]; invT[1] = inv[2]; invT[2] = inv[1]; invT[3] = inv[3];        \
                NAME##_mm(F, x, a);                                                            \
                NAME##_mm(GG, invT, c);                                                        \
                for (int k = 0; k < 4; ++k) x[k] = x[k] + R_dt * (a[k] + c[k]);                \
            }                                                                                  \
        }                                                                                      \
    }

DEFINE_SCAN(oracle_rscan_f32, float)
DEFINE_SCAN(oracle_rscan_f64, double)
