/*
 * TEST INFRASTRUCTURE ONLY -- plain-C oracle / CPU baseline for the MINCO hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Build: make -C oracle   (gcc -O3 -march=native -shared -fPIC -> oracle/liboracle.so)
 *
 * What it restates
 *   oracle_minco_solve : the classic MINCO formulation -- ONE 2sN x 2sN banded collocation system
 *     in the monomial basis (boundary rows, per interior knot 1 waypoint row + 2s-1 continuity rows),
 *     solved by banded LU with partial pivoting, shared by the three axes.  This is the algorithm
 *     of upstream GCOPTER's minco.hpp (BandedSystem::factorizeLU/solve), which is NOT part of
 *     /root/reference: PARITY UNPINNED against the reference itself; pinned
 *     instead against the KKT minimiser of the reference-ASSEMBLED QP matrices
 *     (tests/golden/qp_<case>.npz, keys z_wp_c3 / z_wp_cs) and against oracle/minco_np.py.
 *     It is deliberately a different algorithm from the HIP kernels (Hermite block-tridiagonal).
 *   oracle_traj_cost   : Trajectory::getTrajCost, src/planner/include/gcopter/trajectory.hpp:354-427
 *   oracle_piece_eval  : Piece::getPos/getVel/getAcc/getJer, trajectory.hpp:75-133
 *
 * Layout: trajectory-major (reference flattening): coeffs[piece][axis][D], highest power first.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define MAXN 16
#define MAXDIM (8 * MAXN)

static double falling(int k, int j) {
  double r = 1.0;
  for (int i = 0; i < j; ++i) r *= (double)(k - i);
  return r;
}

/* j-th derivative of the ascending monomial basis at t */
static void drow(int D, double t, int j, double *row) {
  for (int k = 0; k < D; ++k) row[k] = 0.0;
  double tp = 1.0;
  for (int k = j; k < D; ++k) {
    row[k] = falling(k, j) * tp;
    tp *= t;
  }
}

/* Solve one trajectory.  head/tail: 3 x c (row = axis); wps: (N-1) x 3; T: N.
 * coeffs: N x 3 x D (highest power first), energy: int (p^(s))^2 over all axes (no 1/2). */
int oracle_minco_solve(int s, int c, int N, const double *head, const double *tail,
                       const double *wps, const double *T, double *coeffs, double *energy) {
  const int D = 2 * s, n = D * N;
  const int kl = 3 * s - 1, ku = 3 * s - 1;
  if (s < 2 || s > 4 || c < 1 || c > s || N < 1 || N > MAXN) return -1;
  static __thread double M[MAXDIM][MAXDIM];
  static __thread double R[MAXDIM][3];
  for (int i = 0; i < n; ++i) {
    memset(M[i], 0, sizeof(double) * n);
    R[i][0] = R[i][1] = R[i][2] = 0.0;
  }
  double row[8];
  int r = 0;
  for (int j = 0; j < s; ++j, ++r) {
    if (j < c) {
      drow(D, 0.0, j, row);
      for (int a = 0; a < 3; ++a) R[r][a] = head[a * c + j];
    } else {
      drow(D, 0.0, 2 * s - 1 - j, row); /* natural condition of the free derivative j */
    }
    memcpy(&M[r][0], row, sizeof(double) * D);
  }
  for (int i = 1; i < N; ++i) {
    const int cl = (i - 1) * D, cr = i * D;
    drow(D, T[i - 1], 0, row);
    memcpy(&M[r][cl], row, sizeof(double) * D);
    for (int a = 0; a < 3; ++a) R[r][a] = wps[(i - 1) * 3 + a];
    ++r;
    for (int j = 0; j < 2 * s - 1; ++j, ++r) {
      drow(D, T[i - 1], j, row);
      memcpy(&M[r][cl], row, sizeof(double) * D);
      drow(D, 0.0, j, row);
      for (int k = 0; k < D; ++k) M[r][cr + k] = -row[k];
    }
  }
  {
    const int cl = (N - 1) * D;
    for (int j = 0; j < s; ++j, ++r) {
      if (j < c) {
        drow(D, T[N - 1], j, row);
        for (int a = 0; a < 3; ++a) R[r][a] = tail[a * c + j];
      } else {
        drow(D, T[N - 1], 2 * s - 1 - j, row);
      }
      memcpy(&M[r][cl], row, sizeof(double) * D);
    }
  }
  /* banded LU with partial pivoting (fill-in bounded by kl + ku above the diagonal) */
  for (int j = 0; j < n; ++j) {
    int rmax = j + kl < n - 1 ? j + kl : n - 1;
    int cmax = j + kl + ku < n - 1 ? j + kl + ku : n - 1;
    int p = j;
    double best = fabs(M[j][j]);
    for (int i = j + 1; i <= rmax; ++i)
      if (fabs(M[i][j]) > best) { best = fabs(M[i][j]); p = i; }
    if (best == 0.0) return -2;
    if (p != j) {
      for (int k = j; k <= cmax; ++k) { double t = M[j][k]; M[j][k] = M[p][k]; M[p][k] = t; }
      for (int a = 0; a < 3; ++a) { double t = R[j][a]; R[j][a] = R[p][a]; R[p][a] = t; }
    }
    const double inv = 1.0 / M[j][j];
    for (int i = j + 1; i <= rmax; ++i) {
      const double f = M[i][j] * inv;
      if (f == 0.0) continue;
      for (int k = j + 1; k <= cmax; ++k) M[i][k] -= f * M[j][k];
      R[i][0] -= f * R[j][0]; R[i][1] -= f * R[j][1]; R[i][2] -= f * R[j][2];
    }
  }
  for (int j = n - 1; j >= 0; --j) {
    int cmax = j + kl + ku < n - 1 ? j + kl + ku : n - 1;
    for (int a = 0; a < 3; ++a) {
      double v = R[j][a];
      for (int k = j + 1; k <= cmax; ++k) v -= M[j][k] * R[k][a];
      R[j][a] = v / M[j][j];
    }
  }
  /* unpack: ascending -> highest power first; energy by exact integration of (p^(s))^2 */
  double e = 0.0;
  for (int i = 0; i < N; ++i) {
    double tp[8];
    tp[0] = 1.0;
    for (int k = 1; k < 8; ++k) tp[k] = tp[k - 1] * T[i];
    for (int a = 0; a < 3; ++a) {
      double *cm = coeffs ? coeffs + (i * 3 + a) * D : NULL;
      for (int k = 0; k < D; ++k)
        if (cm) cm[D - 1 - k] = R[i * D + k][a];
      for (int j = s; j < D; ++j)
        for (int k = s; k < D; ++k)
          e += falling(j, s) * falling(k, s) / (double)(j + k - 2 * s + 1) *
               tp[j + k - 2 * s + 1] * R[i * D + j][a] * R[i * D + k][a];
    }
  }
  if (energy) *energy = e;
  return 0;
}

/* ---- batch driver (trajectory-major arrays), optionally threaded: the CPU baseline ------------ */
typedef struct {
  int s, c, N;
  int64_t b0, b1;
  const double *head, *tail, *wps, *T;
  double *coeffs, *energy;
  int rc;
} job_t;

static void *job_run(void *arg) {
  job_t *j = (job_t *)arg;
  const int D = 2 * j->s;
  j->rc = 0;
  for (int64_t b = j->b0; b < j->b1; ++b) {
    int rc = oracle_minco_solve(j->s, j->c, j->N, j->head + b * 3 * j->c, j->tail + b * 3 * j->c,
                                j->wps + b * (j->N - 1) * 3, j->T + b * j->N,
                                j->coeffs ? j->coeffs + b * j->N * 3 * D : NULL,
                                j->energy ? j->energy + b : NULL);
    if (rc) j->rc = rc;
  }
  return NULL;
}

int oracle_minco_solve_batch(int s, int c, int N, int64_t B, const double *head, const double *tail,
                             const double *wps, const double *T, double *coeffs, double *energy,
                             int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  job_t jobs[256];
  pthread_t th[256];
  int64_t per = (B + nthreads - 1) / nthreads;
  int used = 0;
  for (int t = 0; t < nthreads; ++t) {
    int64_t b0 = t * per, b1 = b0 + per < B ? b0 + per : B;
    if (b0 >= b1) break;
    jobs[t] = (job_t){s, c, N, b0, b1, head, tail, wps, T, coeffs, energy, 0};
    ++used;
  }
  if (used == 1) {
    job_run(&jobs[0]);
  } else {
    for (int t = 0; t < used; ++t) pthread_create(&th[t], NULL, job_run, &jobs[t]);
    for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
  }
  for (int t = 0; t < used; ++t)
    if (jobs[t].rc) return jobs[t].rc;
  return 0;
}

/* ---- Trajectory restatement ------------------------------------------------------------------- */
/* Piece<D>::getPos/Vel/Acc/Jer (trajectory.hpp:75-133): cm is 3 x D (highest power first). */
void oracle_piece_eval(int D, const double *cm, double t, int d, double *out3) {
  const int deg = D - 1;
  out3[0] = out3[1] = out3[2] = 0.0;
  double tn = 1.0;
  for (int i = deg - d; i >= 0; --i) {
    const int k = deg - i;
    const double f = falling(k, d) * tn;
    for (int a = 0; a < 3; ++a) out3[a] += f * cm[a * D + i];
    tn *= t;
  }
}

/* Trajectory<D>::getTrajCost(order) (trajectory.hpp:354-427); m34 = 1400 is the reference. */
double oracle_traj_cost(int order, int N, const double *coeffs, const double *T, double m34) {
  const int D = 2 * order;
  double energy = 0.0;
  for (int i = 0; i < N; ++i) {
    const double t = T[i], t2 = t * t, t3 = t * t2, t4 = t2 * t2, t5 = t2 * t3;
    double Q[4][4];
    if (order == 4) {
      const double t6 = t3 * t3, t7 = t4 * t3;
      const double q[4][4] = {{100800 * t7, 50400 * t6, 20160 * t5, 5040 * t4},
                              {50400 * t6, 25920 * t5, 10800 * t4, 2880 * t3},
                              {20160 * t5, 10800 * t4, 4800 * t3, m34 * t2},
                              {5040 * t4, 2880 * t3, m34 * t2, 576 * t}};
      memcpy(Q, q, sizeof(q));
    } else {
      const double q[4][4] = {{720 * t5, 360 * t4, 120 * t3, 0}, {360 * t4, 192 * t3, 72 * t2, 0},
                              {120 * t3, 72 * t2, 36 * t, 0}, {0, 0, 0, 0}};
      memcpy(Q, q, sizeof(q));
    }
    for (int a = 0; a < 3; ++a) {
      const double *z = coeffs + (i * 3 + a) * D;
      double acc = 0.0;
      for (int j = 0; j < order; ++j)
        for (int k = 0; k < order; ++k) acc += z[j] * Q[j][k] * z[k];
      energy += 0.5 * acc;
    }
  }
  return energy;
}
