/*
 * upkie_oracle_mpc.c -- CPU fp64 restatement of the MPC balancer path.
 * TEST INFRASTRUCTURE ONLY.
 *
 * Reference call sites: upkie/controllers/mpc_balancer.py:18-37 (target
 * states), :168-226 (problem construction), :237-312 (step). The arithmetic of
 * the condensed QP lives in third-party packages absent from /root/reference:
 * qpmpc 3.1.0/3.2.0 (pixi.lock:83,1821) -- systems.WheeledInvertedPendulum and
 * MPCQP -- and the solver proxsuite 0.7.3 (pixi.lock:58). Their published
 * algorithms are restated here: exact zero-order-hold discretisation of the
 * wheeled inverted pendulum, condensing x_k = Phi_k x0 + Psi_k U, and
 * P = w_u I + w_T Psi_N^T Psi_N + w_x sum_k Psi_k^T Psi_k,
 * q = w_T Psi_N^T (Phi_N x0 - x_goal) + w_x sum_k Psi_k^T (Phi_k x0 - x*_k).
 * The reference holds no numeric golden for this path
 * (tests/controllers/test_mpc_balancer.py:38-47 checks finiteness and the
 * velocity bound only): numeric parity with ProxQP is UNPINNED; this file pins
 * the QP to its exact solution instead (ProxQP stops at eps_abs = 1e-3,
 * mpc_balancer.py:76-77).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "upkie_oracle.h"

#define NX 4
#define MPC_GRAVITY 9.81 /* qpmpc GRAVITY constant */

/* A (4x4 row-major) and B (4) of the discretised pendulum. */
static void pendulum_discretisation(const UpkieMpcConfig* cfg, double A[16],
                                    double Bv[4]) {
  double T = cfg->sampling_period;
  double omega = sqrt(MPC_GRAVITY / cfg->leg_length);
  double ch = cosh(T * omega), sh = sinh(T * omega);
  double a[16] = {1, 0, T, 0, 0, ch, 0, sh / omega, 0, 0, 1, 0, 0, omega * sh, 0, ch};
  memcpy(A, a, sizeof(a));
  Bv[0] = T * T / 2.0;
  Bv[1] = -ch / MPC_GRAVITY + 1.0 / MPC_GRAVITY;
  Bv[2] = T;
  Bv[3] = -omega * sh / MPC_GRAVITY;
}

/* Phi[k] (4x4) and Psi[k] (4xN) for k = 0..N, as MPCQP's constructor loop. */
static void condense(const UpkieMpcConfig* cfg, double* Phi, double* Psi) {
  int N = cfg->nb_timesteps;
  double A[16], Bv[4];
  pendulum_discretisation(cfg, A, Bv);
  double* phi = (double*)calloc(16, sizeof(double));
  double* psi = (double*)calloc(4 * N, sizeof(double));
  for (int i = 0; i < 4; ++i) phi[5 * i] = 1.0;
  for (int k = 0; k <= N; ++k) {
    memcpy(Phi + 16 * k, phi, 16 * sizeof(double));
    memcpy(Psi + (size_t)4 * N * k, psi, (size_t)4 * N * sizeof(double));
    if (k == N) break;
    double nphi[16], *npsi = (double*)calloc(4 * N, sizeof(double));
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int l = 0; l < 4; ++l) s += A[4 * i + l] * phi[4 * l + j];
        nphi[4 * i + j] = s;
      }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < N; ++j) {
        double s = 0;
        for (int l = 0; l < 4; ++l) s += A[4 * i + l] * psi[N * l + j];
        npsi[N * i + j] = s;
      }
    for (int i = 0; i < 4; ++i) npsi[N * i + k] = Bv[i];
    memcpy(phi, nphi, sizeof(nphi));
    memcpy(psi, npsi, (size_t)4 * N * sizeof(double));
    free(npsi);
  }
  free(phi);
  free(psi);
}

void oracle_mpc_build(const UpkieMpcConfig* cfg, double* P, double* Kx,
                      double* kv) {
  int N = cfg->nb_timesteps;
  double T = cfg->sampling_period;
  double wu = cfg->stage_input_cost_weight, wx = cfg->stage_state_cost_weight,
         wT = cfg->terminal_cost_weight;
  double* Phi = (double*)malloc(sizeof(double) * 16 * (N + 1));
  double* Psi = (double*)malloc(sizeof(double) * 4 * N * (N + 1));
  condense(cfg, Phi, Psi);
  memset(P, 0, sizeof(double) * N * N);
  memset(Kx, 0, sizeof(double) * N * 4);
  memset(kv, 0, sizeof(double) * N);
  for (int i = 0; i < N; ++i) P[i * N + i] = wu;
  for (int k = 0; k <= N; ++k) {
    double w = (k == N) ? wT : wx;
    const double* psi = Psi + (size_t)4 * N * k;
    const double* phi = Phi + 16 * k;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        double s = 0;
        for (int l = 0; l < 4; ++l) s += psi[N * l + i] * psi[N * l + j];
        P[i * N + j] += w * s;
      }
    /* target_k = [p0 + k T v*, 0, v*, 0] (mpc_balancer.py:31-37):
     * Phi_k x0 - target_k = (Phi_k - e0 e0^T) x0 - (k T e0 + e2) v* */
    for (int i = 0; i < N; ++i) {
      for (int c = 0; c < 4; ++c) {
        double s = 0;
        for (int l = 0; l < 4; ++l) {
          double d = phi[4 * l + c] - ((l == 0 && c == 0) ? 1.0 : 0.0);
          s += psi[N * l + i] * d;
        }
        Kx[i * 4 + c] += w * s;
      }
      kv[i] += -w * (psi[N * 0 + i] * (k * T) + psi[N * 2 + i]);
    }
  }
  free(Phi);
  free(Psi);
}

void oracle_mpc_cost_vector(const UpkieMpcConfig* cfg, const double x0[4],
                            double v_target, double* q) {
  int N = cfg->nb_timesteps;
  double T = cfg->sampling_period;
  double* Phi = (double*)malloc(sizeof(double) * 16 * (N + 1));
  double* Psi = (double*)malloc(sizeof(double) * 4 * N * (N + 1));
  condense(cfg, Phi, Psi);
  /* get_target_states, mpc_balancer.py:28-37 */
  double* target = (double*)calloc((size_t)(N + 1) * NX, sizeof(double));
  for (int k = 0; k <= N; ++k) {
    target[k * NX] = x0[0] + (k * T) * v_target;
    target[k * NX + 2] = v_target;
  }
  memset(q, 0, sizeof(double) * N);
  for (int k = 0; k <= N; ++k) {
    double w = (k == N) ? cfg->terminal_cost_weight : cfg->stage_state_cost_weight;
    double c[4];
    for (int i = 0; i < 4; ++i) {
      double s = 0;
      for (int l = 0; l < 4; ++l) s += Phi[16 * k + 4 * i + l] * x0[l];
      c[i] = s - target[k * NX + i];
    }
    for (int j = 0; j < N; ++j) {
      double s = 0;
      for (int i = 0; i < 4; ++i) s += c[i] * Psi[(size_t)4 * N * k + N * i + j];
      q[j] += w * s;
    }
  }
  free(target);
  free(Phi);
  free(Psi);
}

static int chol(int n, const double* A, double* L) {
  memset(L, 0, sizeof(double) * n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
      if (i == j) {
        if (s <= 0) return -1;
        L[i * n + i] = sqrt(s);
      } else {
        L[i * n + j] = s / L[j * n + j];
      }
    }
  return 0;
}
static void chol_solve(int n, const double* L, const double* b, double* x) {
  double* y = (double*)malloc(sizeof(double) * n);
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k];
    y[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
  free(y);
}

static double qp_objective(int n, const double* P, const double* q, const double* u) {
  double f = 0;
  for (int i = 0; i < n; ++i) {
    double s = 0;
    for (int j = 0; j < n; ++j) s += P[i * n + j] * u[j];
    f += u[i] * (0.5 * s + q[i]);
  }
  return f;
}

/* Exact solution of min 1/2 u'Pu + q'u, |u_i| <= bound by projected Newton
 * with active-set identification (Bertsekas 1982). Returns iterations used,
 * or -1 if the KKT residual is not below 1e-10 at exit. */
int oracle_mpc_solve_exact(int n, const double* P, const double* q,
                           double bound, double* u) {
  double* g = (double*)malloc(sizeof(double) * n);
  double* d = (double*)malloc(sizeof(double) * n);
  double* trial = (double*)malloc(sizeof(double) * n);
  double* PF = (double*)malloc(sizeof(double) * n * n);
  double* LF = (double*)malloc(sizeof(double) * n * n);
  double* rhs = (double*)malloc(sizeof(double) * n);
  double* sol = (double*)malloc(sizeof(double) * n);
  int* free_idx = (int*)malloc(sizeof(int) * n);
  for (int i = 0; i < n; ++i) u[i] = 0.0;
  int it, status = -1;
  for (it = 0; it < 50 * n + 100; ++it) {
    for (int i = 0; i < n; ++i) {
      double s = q[i];
      for (int j = 0; j < n; ++j) s += P[i * n + j] * u[j];
      g[i] = s;
    }
    double kkt = 0;
    int nf = 0;
    for (int i = 0; i < n; ++i) {
      int at_lo = u[i] <= -bound, at_hi = u[i] >= bound;
      int active = (at_lo && g[i] > 0) || (at_hi && g[i] < 0);
      if (!active) {
        free_idx[nf++] = i;
        double r = fabs(g[i]);
        if (at_lo && g[i] > 0) r = 0;
        if (at_hi && g[i] < 0) r = 0;
        /* strictly interior or wrongly signed at bound */
        if (r > kkt) kkt = r;
      }
    }
    if (kkt < 1e-12) {
      status = it;
      break;
    }
    for (int a = 0; a < nf; ++a) {
      rhs[a] = -g[free_idx[a]];
      for (int b = 0; b < nf; ++b) PF[a * nf + b] = P[free_idx[a] * n + free_idx[b]];
    }
    if (chol(nf, PF, LF) != 0) break;
    chol_solve(nf, LF, rhs, sol);
    memset(d, 0, sizeof(double) * n);
    for (int a = 0; a < nf; ++a) d[free_idx[a]] = sol[a];
    double f0 = qp_objective(n, P, q, u), alpha = 1.0;
    int accepted = 0;
    for (int ls = 0; ls < 60; ++ls) {
      for (int i = 0; i < n; ++i) {
        double x = u[i] + alpha * d[i];
        trial[i] = x > bound ? bound : (x < -bound ? -bound : x);
      }
      if (qp_objective(n, P, q, trial) <= f0 - 1e-18) {
        accepted = 1;
        break;
      }
      alpha *= 0.5;
    }
    if (!accepted) {
      status = it; /* no further decrease possible at machine precision */
      break;
    }
    memcpy(u, trial, sizeof(double) * n);
  }
  free(g); free(d); free(trial); free(PF); free(LF); free(rhs); free(sol); free(free_idx);
  return status;
}

void oracle_mpc_minv(int n, const double* P, double rho, double* Minv) {
  double* A = (double*)malloc(sizeof(double) * n * n);
  double* L = (double*)malloc(sizeof(double) * n * n);
  double* e = (double*)calloc(n, sizeof(double));
  double* x = (double*)malloc(sizeof(double) * n);
  memcpy(A, P, sizeof(double) * n * n);
  for (int i = 0; i < n; ++i) A[i * n + i] += rho;
  chol(n, A, L);
  for (int c = 0; c < n; ++c) {
    memset(e, 0, sizeof(double) * n);
    e[c] = 1.0;
    chol_solve(n, L, e, x);
    for (int r = 0; r < n; ++r) Minv[r * n + c] = x[r];
  }
  free(A); free(L); free(e); free(x);
}

/* Same recurrences as the HIP kernel:
 *   U <- Minv (rho (z - y) - q);  z <- clip(U + y);  y <- y + U - z */
void oracle_mpc_admm_relaxed(int n, const double* Minv, const double* q, double rho, double alpha,
                             double bound, int iterations, double* z, double* y, double* u);
void oracle_mpc_admm(int n, const double* Minv, const double* q, double rho,
                     double bound, int iterations, double* z, double* y,
                     double* u) {
  oracle_mpc_admm_relaxed(n, Minv, q, rho, 1.0, bound, iterations, z, y, u);
}

/* ... with over-relaxation (UpkieMpcConfig.admm_relaxation; Boyd et al. 2011,
 * section 3.4.3): x^ = alpha U + (1 - alpha) z takes U's place in the z- and
 * y-updates; alpha = 1 is the plain iteration. */
void oracle_mpc_admm_relaxed(int n, const double* Minv, const double* q, double rho, double alpha,
                             double bound, int iterations, double* z, double* y, double* u) {
  double* r = (double*)malloc(sizeof(double) * n);
  for (int it = 0; it < iterations; ++it) {
    for (int i = 0; i < n; ++i) r[i] = rho * (z[i] - y[i]) - q[i];
    for (int i = 0; i < n; ++i) {
      double s = 0;
      for (int j = 0; j < n; ++j) s += Minv[i * n + j] * r[j];
      u[i] = s;
    }
    for (int i = 0; i < n; ++i) {
      const double relaxed = alpha * u[i] + (1.0 - alpha) * z[i];
      double x = relaxed + y[i];
      double zi = x > bound ? bound : (x < -bound ? -bound : x);
      y[i] = y[i] + relaxed - zi;
      z[i] = zi;
    }
  }
  free(r);
}

/* MPCBalancer.step for a batch, mpc_balancer.py:237-312. */
void oracle_mpc_step(const UpkieMpcConfig* cfg, double* workspace,
                     const double* x0, const double* v_target,
                     const uint8_t* contact, double dt, double* commanded,
                     double* first_input) {
  int N = cfg->nb_timesteps, B = cfg->num_envs;
  double* P = (double*)malloc(sizeof(double) * N * N);
  double* Kx = (double*)malloc(sizeof(double) * N * 4);
  double* kv = (double*)malloc(sizeof(double) * N);
  double* Minv = (double*)malloc(sizeof(double) * N * N);
  oracle_mpc_build(cfg, P, Kx, kv);
  oracle_mpc_minv(N, P, cfg->admm_rho, Minv);
#pragma omp parallel for schedule(static)
  for (int e = 0; e < B; ++e) {
    double q[128], z[128], y[128], u[128];
    const double* x = x0 + 4 * (size_t)e;
    for (int i = 0; i < N; ++i) {
      q[i] = Kx[4 * i] * x[0] + Kx[4 * i + 1] * x[1] + Kx[4 * i + 2] * x[2] +
             Kx[4 * i + 3] * x[3] + kv[i] * v_target[e];
      z[i] = workspace[(size_t)i * B + e];
      y[i] = workspace[(size_t)(N + i) * B + e];
    }
    oracle_mpc_admm_relaxed(N, Minv, q, cfg->admm_rho, cfg->admm_relaxation > 0.0 ? cfg->admm_relaxation : 1.0, cfg->max_ground_accel,
                            cfg->admm_iterations, z, y, u);
    for (int i = 0; i < N; ++i) {
      workspace[(size_t)i * B + e] = z[i];
      workspace[(size_t)(N + i) * B + e] = y[i];
    }
    double u0 = z[0];
    if (first_input) first_input[e] = u0;
    int fallen = fabs(x[1]) > cfg->fall_pitch; /* :260 */
    double v = commanded[e];
    if (fallen || !contact[e]) {
      v = v + (dt / 0.1) * (0.0 - v); /* :295-301, filters.py:77-80 */
    } else {
      v = v + u0 * dt / 2.0; /* :305-311 */
      if (v < -cfg->max_ground_velocity) v = -cfg->max_ground_velocity;
      if (v > cfg->max_ground_velocity) v = cfg->max_ground_velocity;
    }
    commanded[e] = v;
  }
  free(P); free(Kx); free(kv); free(Minv);
}
