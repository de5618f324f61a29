// Trajectory<D> kernels: polynomial evaluation with the reference's piece location, getTrajCost and
// its derivative w.r.t. the durations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace anet {

// Trajectory<D>::getPos/Vel/Acc/Jer: one lane per trajectory, nq queries each.  The accumulation
// order is the reference's (ascending powers, tn *= t), trajectory.hpp:75-133.
struct EvalArgs {
  const double *coeffs, *T, *tq;
  double *out;
  int64_t B, ld;
  int N, nq, deriv;
};
template <int S>
__global__ void __launch_bounds__(256) k_traj_eval(EvalArgs a) {
  constexpr int D = 2 * S, DEG = D - 1;
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const int64_t ld = a.ld;
  const int N = a.N, d = a.deriv;
  for (int q = 0; q < a.nq; ++q) {
    double t = a.tq[(int64_t)q * ld + b];
    // locatePieceIdx (trajectory.hpp:496-514)
    int idx = 0;
    double dur = 0.0;
    for (; idx < N; ++idx) {
      dur = a.T[(int64_t)idx * ld + b];
      if (!(t > dur)) break;
      t -= dur;
    }
    if (idx == N) {
      --idx;
      t += a.T[(int64_t)idx * ld + b];
    }
    const double *cm = a.coeffs + (int64_t)(idx * 3 * D) * ld + b;
    double acc[3] = {0.0, 0.0, 0.0};
    double tn = 1.0;
    for (int i = DEG - d; i >= 0; --i) {
      const int k = DEG - i;  // power of column i
      double f = 1.0;
      for (int e = 0; e < d; ++e) f *= (double)(k - e);
      const double w = f * tn;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) acc[ax] += w * cm[(int64_t)(ax * D + i) * ld];
      tn *= t;
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) a.out[(int64_t)(q * 3 + ax) * ld + b] = acc[ax];
  }
}

// Piece<D>::normalizePosCoeffMat / normalizeVelCoeffMat / normalizeAccCoeffMat (trajectory.hpp:135-171): the coefficients of the
// piece's position / velocity / acceleration polynomial in NORMALISED time, column i of the result = (product of the d falling
// factors of its power) * coeffMat.col(i) * duration^power -- the running product `t *= duration` of the reference, from the
// constant column up.  One lane per piece; trajectory-major in and out (pieces of a batch are independent records):
// coeffs [P][3][D], T [P], out [P][3][D - d].
struct NormArgs {
  const double *coeffs, *T;
  double *out;
  int64_t P;
  int deriv;
};
template <int S>
__global__ void __launch_bounds__(256) k_piece_normalize(NormArgs a) {
  constexpr int D = 2 * S;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.P) return;
  const int d = a.deriv, W = D - d;
  const double dur = a.T[p];
  const double *cm = a.coeffs + p * 3 * D;
  double *o = a.out + p * 3 * W;
  double t = 1.0;
  for (int e = 0; e < d; ++e) t *= dur;          // velocity starts at duration, acceleration at duration^2
  for (int i = W - 1; i >= 0; --i) {             // column i of the result holds power (W - 1 - i) of the derivative
    const int k = (D - 1) - i;                   // ... which comes from power k of the position, column i of coeffMat
    double f = 1.0;
    for (int e = 0; e < d; ++e) f *= (double)(k - e);   // n (velocity), n * m (acceleration) of the reference's loops
    for (int ax = 0; ax < 3; ++ax) o[ax * W + i] = f * cm[ax * D + i] * t;
    t *= dur;
  }
}

// Trajectory<D>::getTrajCost (trajectory.hpp:354-427).
struct CostArgs {
  const double *coeffs, *T;
  double *cost;   // [B] or nullptr
  double *gradT;  // [N][ld] or nullptr: d cost / d T_i at fixed coefficients
  int64_t B, ld;
  int N;
  double m34;
};
template <int S>
__global__ void __launch_bounds__(256) k_traj_cost(CostArgs a) {
  constexpr int D = 2 * S;
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  const int64_t ld = a.ld;
  double energy = 0.0;
  for (int i = 0; i < a.N; ++i) {
    const double t = a.T[(int64_t)i * ld + b];
    const double t2 = t * t, t3 = t * t2, t4 = t2 * t2, t5 = t2 * t3;
    double Q[S][S], dQ[S][S];  // cost block and its derivative w.r.t. t
    if constexpr (S == 4) {
      const double t6 = t3 * t3, t7 = t4 * t3;
      Q[0][0] = 100800 * t7; Q[0][1] = 50400 * t6; Q[0][2] = 20160 * t5; Q[0][3] = 5040 * t4;
      Q[1][1] = 25920 * t5;  Q[1][2] = 10800 * t4; Q[1][3] = 2880 * t3;
      Q[2][2] = 4800 * t3;   Q[2][3] = a.m34 * t2;
      Q[3][3] = 576 * t;
      dQ[0][0] = 7 * 100800 * t6; dQ[0][1] = 6 * 50400 * t5; dQ[0][2] = 5 * 20160 * t4; dQ[0][3] = 4 * 5040 * t3;
      dQ[1][1] = 5 * 25920 * t4;  dQ[1][2] = 4 * 10800 * t3; dQ[1][3] = 3 * 2880 * t2;
      dQ[2][2] = 3 * 4800 * t2;   dQ[2][3] = 2 * a.m34 * t;
      dQ[3][3] = 576;
    } else if constexpr (S == 3) {
      Q[0][0] = 720 * t5; Q[0][1] = 360 * t4; Q[0][2] = 120 * t3;
      Q[1][1] = 192 * t3; Q[1][2] = 72 * t2;
      Q[2][2] = 36 * t;
      dQ[0][0] = 5 * 720 * t4; dQ[0][1] = 4 * 360 * t3; dQ[0][2] = 3 * 120 * t2;
      dQ[1][1] = 3 * 192 * t2; dQ[1][2] = 2 * 72 * t;
      dQ[2][2] = 36;
    } else {
      Q[0][0] = 12 * t3; Q[0][1] = 6 * t2;
      Q[1][1] = 4 * t;
      dQ[0][0] = 36 * t2; dQ[0][1] = 12 * t;
      dQ[1][1] = 4;
    }
#pragma unroll
    for (int j = 1; j < S; ++j)
#pragma unroll
      for (int k = 0; k < j; ++k) {
        Q[j][k] = Q[k][j];
        dQ[j][k] = dQ[k][j];
      }
    double gti = 0.0;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      double z[S];
#pragma unroll
      for (int j = 0; j < S; ++j) z[j] = a.coeffs[(int64_t)((i * 3 + ax) * D + j) * ld + b];
      double acc = 0.0, dacc = 0.0;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        double r = 0.0, dr = 0.0;
#pragma unroll
        for (int k = 0; k < S; ++k) {
          r += Q[j][k] * z[k];
          dr += dQ[j][k] * z[k];
        }
        acc += z[j] * r;
        dacc += z[j] * dr;
      }
      energy += 0.5 * acc;
      gti += 0.5 * dacc;
    }
    if (a.gradT) a.gradT[(int64_t)i * ld + b] = gti;
  }
  if (a.cost) a.cost[b] = energy;
}

}  // namespace anet
