// solve.hpp -- the tiny dense solves of one ICP iteration, shared by the host API and the
// single-lane device epilogue (so the fused loop never round-trips to the host).
//
// Replaces (all f64 here; the reference does them in f32 through Eigen3):
//   * Eigen::JacobiSVD<Matrix3f>         registration/transform_estimation.hpp:36-44
//                                        core/space_transformations.hpp:43-51 (rotation())
//   * Eigen::LDLT<Matrix6f>::solve       registration/transform_estimation.hpp:346
//   * AngleAxis / Translation update     registration/transform_estimation.hpp:349-357,361/365
//   * compose + delta norm               registration/icp_single_transform_combined_metric.hpp:207-216
//                                        registration/icp_single_transform_point_to_point_metric.hpp:56-64
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

namespace cilhip {

#define CILHIP_HD __host__ __device__ __forceinline__

// Layout of the raw f64 sums produced by the accumulation kernels (one slot per value).
//   Kabsch (point-to-point ICP class):  [0]=n, [1..3]=sum p, [4..6]=sum q, [7..15]=sum p q^T (row-major)
//   Gauss-Newton (combined metric class), plane part: [0]=n, [1..21]=upper triangle of sum e e^T
//     (row-major: 00 01 02 03 04 05 11 12 ... 55), [22..27]=sum (n.(d-s)) e
//   Gauss-Newton, point part (only when w_p2p>0): [28..30]=sum a, [31..36]=sum a a^T upper
//     (00 01 02 11 12 22), [37..39]=sum a x r, [40..42]=sum r          with a=d+s, r=d-s;  [43]=sum of the point-term
//     weights (streaming kernels only; equals n with unity evaluators)
constexpr int SUMS_KABSCH = 16;
constexpr int SUMS_PLANE = 28;
constexpr int SUMS_GN_FULL = 44;
constexpr int SUMS_MAX = 48;  // padded slot count per block partial

// ---- 3x3 two-sided Jacobi SVD (row-major), S >= 0 sorted descending --------------------------
CILHIP_HD void svd3(const double Ain[9], double U[9], double S[3], double V[9]) {
  double W[9];
  for (int i = 0; i < 9; ++i) { W[i] = Ain[i]; U[i] = V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  double scale = 0.0;
  for (int i = 0; i < 9; ++i) scale = fmax(scale, fabs(W[i]));
  if (scale == 0.0) { S[0] = S[1] = S[2] = 0.0; return; }
  for (int i = 0; i < 9; ++i) W[i] /= scale;
  const double precision = 2.0 * 2.220446049250313e-16;
  const double tiny = 2.2250738585072014e-308;
  double max_diag = fmax(fabs(W[0]), fmax(fabs(W[4]), fabs(W[8])));
  for (int sweep = 0; sweep < 64; ++sweep) {
    bool finished = true;
    for (int p = 1; p < 3; ++p) {
      for (int q = 0; q < p; ++q) {
        const double thr = fmax(tiny, precision * max_diag);
        if (fabs(W[p * 3 + q]) > thr || fabs(W[q * 3 + p]) > thr) {
          finished = false;
          const double a = W[q * 3 + q], b = W[q * 3 + p], c = W[p * 3 + q], d = W[p * 3 + p];
          double c1 = 1.0, s1 = 0.0;               // left rotation making the 2x2 block symmetric
          {
            const double t = a + d, dd = c - b;
            if (fabs(dd) > tiny) { const double h = sqrt(t * t + dd * dd); c1 = t / h; s1 = dd / h; }
          }
          const double x = c1 * a + s1 * c, y = c1 * b + s1 * d, z = -s1 * b + c1 * d;
          double cj = 1.0, sj = 0.0;               // symmetric Jacobi rotation
          if (fabs(y) > tiny) {
            const double tau = (z - x) / (2.0 * y);
            const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            cj = 1.0 / sqrt(1.0 + tt * tt);
            sj = tt * cj;
          }
          const double cl = cj * c1 + sj * s1, sl = cj * s1 - sj * c1;
          for (int k = 0; k < 3; ++k) {
            const double wq = W[q * 3 + k], wp = W[p * 3 + k];
            W[q * 3 + k] = cl * wq + sl * wp;
            W[p * 3 + k] = -sl * wq + cl * wp;
          }
          for (int k = 0; k < 3; ++k) {
            const double uq = U[k * 3 + q], up = U[k * 3 + p];
            U[k * 3 + q] = cl * uq + sl * up;
            U[k * 3 + p] = -sl * uq + cl * up;
          }
          for (int k = 0; k < 3; ++k) {
            const double wq = W[k * 3 + q], wp = W[k * 3 + p];
            W[k * 3 + q] = cj * wq - sj * wp;
            W[k * 3 + p] = sj * wq + cj * wp;
            const double vq = V[k * 3 + q], vp = V[k * 3 + p];
            V[k * 3 + q] = cj * vq - sj * vp;
            V[k * 3 + p] = sj * vq + cj * vp;
          }
          max_diag = fmax(max_diag, fmax(fabs(W[p * 3 + p]), fabs(W[q * 3 + q])));
        }
      }
    }
    if (finished) break;
  }
  for (int i = 0; i < 3; ++i) {
    double s = W[i * 3 + i];
    if (s < 0.0) { s = -s; for (int k = 0; k < 3; ++k) U[k * 3 + i] = -U[k * 3 + i]; }
    S[i] = s * scale;
  }
  for (int i = 0; i < 3; ++i) {
    int m = i;
    for (int j = i + 1; j < 3; ++j) if (S[j] > S[m]) m = j;
    if (m != i) {
      double ts = S[i]; S[i] = S[m]; S[m] = ts;
      for (int k = 0; k < 3; ++k) {
        double tu = U[k * 3 + i]; U[k * 3 + i] = U[k * 3 + m]; U[k * 3 + m] = tu;
        double tv = V[k * 3 + i]; V[k * 3 + i] = V[k * 3 + m]; V[k * 3 + m] = tv;
      }
    }
  }
}

CILHIP_HD double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// R = U V^T, negating column `flip_col` of U when det(U V) < 0
// (flip_col 2: transform_estimation.hpp:38-41; flip_col 0: space_transformations.hpp:45-48).
CILHIP_HD void uvt_fix(const double Uin[9], const double V[9], int flip_col, double R[9]) {
  double U[9], UV[9];
  for (int i = 0; i < 9; ++i) U[i] = Uin[i];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      UV[i * 3 + j] = U[i * 3] * V[j] + U[i * 3 + 1] * V[3 + j] + U[i * 3 + 2] * V[6 + j];
  if (det3(UV) < 0.0)
    for (int k = 0; k < 3; ++k) U[k * 3 + flip_col] = -U[k * 3 + flip_col];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      R[i * 3 + j] = U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + U[i * 3 + 2] * V[j * 3 + 2];
}

CILHIP_HD void nearest_rotation(const double L[9], double R[9]) {
  double U[9], S[3], V[9];
  svd3(L, U, S, V);
  uvt_fix(U, V, 0, R);
}

// The same result for the matrices the ICP loop actually hands over -- a rigid step, orthogonal up to round-off and the
// linearisation error -- without the SVD: U V^T is the orthogonal polar factor of L when det L > 0, and Newton's
// iteration X <- (X + X^-T) / 2 converges to it quadratically from X = L (Higham, "Computing the polar decomposition").
// A few cofactor inverses instead of Jacobi sweeps full of f64 square roots and divisions: the single-lane device
// epilogue spends ~1 us here instead of ~5.  Returns false (R untouched) when L is not close to a rotation or the
// iteration has not settled to 1e-15 in 6 steps; the caller then takes the SVD.
CILHIP_HD bool nearest_rotation_polar(const double L[9], double R[9]) {
  double X[9];
  for (int i = 0; i < 9; ++i) X[i] = L[i];
  // closeness gate: |L^T L - I|_max <= 0.25  (also rules out reflections and singular input)
  for (int r = 0; r < 3; ++r)
    for (int c = r; c < 3; ++c) {
      const double g = X[r] * X[c] + X[3 + r] * X[3 + c] + X[6 + r] * X[6 + c] - (r == c ? 1.0 : 0.0);
      if (!(fabs(g) <= 0.25)) return false;
    }
  for (int it = 0; it < 6; ++it) {
    // cofactors: C = det(X) * X^-T
    double C[9];
    C[0] = X[4] * X[8] - X[5] * X[7]; C[1] = X[5] * X[6] - X[3] * X[8]; C[2] = X[3] * X[7] - X[4] * X[6];
    C[3] = X[2] * X[7] - X[1] * X[8]; C[4] = X[0] * X[8] - X[2] * X[6]; C[5] = X[1] * X[6] - X[0] * X[7];
    C[6] = X[1] * X[5] - X[2] * X[4]; C[7] = X[2] * X[3] - X[0] * X[5]; C[8] = X[0] * X[4] - X[1] * X[3];
    const double det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
    if (!(det > 0.5)) return false;
    const double inv = 0.5 / det;
    double diff = 0.0;
    for (int i = 0; i < 9; ++i) {
      const double xn = 0.5 * X[i] + inv * C[i];
      diff = fmax(diff, fabs(xn - X[i]));
      X[i] = xn;
    }
    if (diff <= 1.0e-15) {
      for (int i = 0; i < 9; ++i) R[i] = X[i];
      return true;
    }
  }
  return false;
}

// ---- 6x6 LDL^T, diagonal pivoting, pseudo-inverse of D (Eigen LDLT::solve semantics) ----------
// (work arrays passed in: the pivoting indexes them dynamically, which would put function-local arrays into
// scratch = global memory on the device; the single-lane epilogue hands in LDS)
CILHIP_HD void ldlt6_solve_ws(const double Ain[36], const double bin[6], double x[6], double* A /*[36]*/, double* y /*[6]*/, int* perm /*[6]*/) {
  for (int i = 0; i < 36; ++i) A[i] = Ain[i];
  for (int i = 0; i < 6; ++i) perm[i] = i;
  const double tiny = 2.2250738585072014e-308;
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    double best = fabs(A[k * 6 + k]);
    for (int i = k + 1; i < 6; ++i)
      if (fabs(A[i * 6 + i]) > best) { best = fabs(A[i * 6 + i]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < 6; ++j) { double t = A[k * 6 + j]; A[k * 6 + j] = A[piv * 6 + j]; A[piv * 6 + j] = t; }
      for (int i = 0; i < 6; ++i) { double t = A[i * 6 + k]; A[i * 6 + k] = A[i * 6 + piv]; A[i * 6 + piv] = t; }
      int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    const double dk = A[k * 6 + k];
    if (fabs(dk) <= tiny) {
      for (int i = k + 1; i < 6; ++i) A[i * 6 + k] = 0.0;
      continue;
    }
    for (int i = k + 1; i < 6; ++i) A[i * 6 + k] /= dk;
    for (int i = k + 1; i < 6; ++i)
      for (int j = k + 1; j <= i; ++j) {
        A[i * 6 + j] -= A[i * 6 + k] * dk * A[j * 6 + k];
        A[j * 6 + i] = A[i * 6 + j];
      }
  }
  for (int i = 0; i < 6; ++i) y[i] = bin[perm[i]];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < i; ++j) y[i] -= A[i * 6 + j] * y[j];
  for (int i = 0; i < 6; ++i) {
    const double d = A[i * 6 + i];
    y[i] = (fabs(d) > tiny) ? y[i] / d : 0.0;
  }
  for (int i = 5; i >= 0; --i)
    for (int j = i + 1; j < 6; ++j) y[i] -= A[j * 6 + i] * y[j];
  for (int i = 0; i < 6; ++i) x[perm[i]] = y[i];
}

// Fast path of the same solve for a well-conditioned (positive definite) system: unpivoted LDL^T, fully unrolled with
// compile-time indices so that the matrix lives in registers -- the pivoted version above indexes its work arrays
// dynamically (LDS round trips on the device: ~4 us for one lane; this one ~1 us).  Returns false, x untouched, if a
// pivot is not safely positive (rank-deficient or indefinite normal equations): the caller then takes the pivoted solve,
// whose pseudo-inverse semantics matter exactly there.
CILHIP_HD bool ldlt6_solve_fast(const double Ain[36], const double bin[6], double x[6]) {
  double A[36], d[6], y[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) A[i] = Ain[i];
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double dk = A[k * 6 + k];
    // a pivot that kept less than 1e-11 of its diagonal entry through the elimination marks a (numerically) dependent
    // unknown -- a scale-invariant test: the unknowns of the ICP step have very different magnitudes
    ok = ok && (Ain[k * 6 + k] > 0.0) && (dk > 1.0e-11 * Ain[k * 6 + k]);
    d[k] = dk;
    const double inv = 1.0 / dk;
    double lcol[6];
#pragma unroll
    for (int i = k + 1; i < 6; ++i) lcol[i] = A[i * 6 + k] * inv;
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
#pragma unroll
      for (int j = k + 1; j <= i; ++j) A[i * 6 + j] -= lcol[i] * A[j * 6 + k];   // A[j][k] still holds l_jk * d_k
#pragma unroll
    for (int i = k + 1; i < 6; ++i) A[i * 6 + k] = lcol[i];
  }
  if (!ok) return false;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double v = bin[i];
#pragma unroll
    for (int j = 0; j < i; ++j) v -= A[i * 6 + j] * y[j];
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] /= d[i];
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double v = y[i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) v -= A[j * 6 + i] * y[j];
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = y[i];
  return true;
}

CILHIP_HD void ldlt6_solve(const double Ain[36], const double bin[6], double x[6]) {
  if (ldlt6_solve_fast(Ain, bin, x)) return;
  double A[36], y[6];
  int perm[6];
  ldlt6_solve_ws(Ain, bin, x, A, y, perm);
}

// tform = Ra * ta * Ra * tform  (transform_estimation.hpp:349-357); L row-major, in place.
CILHIP_HD void rigid_gn_update(const double dth[6], double L[9], double t[3]) {
  const double ax = dth[0], ay = dth[1], az = dth[2];
  const double na = sqrt(ax * ax + ay * ay + az * az);
  // theta = atan(|a|):  sin(theta) = |a| / sqrt(1+|a|^2), cos(theta) = 1 / sqrt(1+|a|^2)  (exact identities;
  // avoids three f64 libm calls in the single-lane device epilogue)
  double ux = 0.0, uy = 0.0, uz = 0.0;
  if (na > 0.0) { ux = ax / na; uy = ay / na; uz = az / na; }
  const double c = 1.0 / sqrt(1.0 + na * na), s = na * c;
  const double sx = s * ux, sy = s * uy, sz = s * uz;
  const double c1x = (1.0 - c) * ux, c1y = (1.0 - c) * uy, c1z = (1.0 - c) * uz;
  double Ra[9], tmp;
  tmp = c1x * uy; Ra[1] = tmp - sz; Ra[3] = tmp + sz;
  tmp = c1x * uz; Ra[2] = tmp + sy; Ra[6] = tmp - sy;
  tmp = c1y * uz; Ra[5] = tmp - sx; Ra[7] = tmp + sx;
  Ra[0] = c1x * ux + c; Ra[4] = c1y * uy + c; Ra[8] = c1z * uz + c;
  const double ta[3] = {c * dth[3], c * dth[4], c * dth[5]};
  double L1[9], t1[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      L1[i * 3 + j] = Ra[i * 3] * L[j] + Ra[i * 3 + 1] * L[3 + j] + Ra[i * 3 + 2] * L[6 + j];
    t1[i] = Ra[i * 3] * t[0] + Ra[i * 3 + 1] * t[1] + Ra[i * 3 + 2] * t[2] + ta[i];
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      L[i * 3 + j] = Ra[i * 3] * L1[j] + Ra[i * 3 + 1] * L1[3 + j] + Ra[i * 3 + 2] * L1[6 + j];
    t[i] = Ra[i * 3] * t1[0] + Ra[i * 3 + 1] * t1[1] + Ra[i * 3 + 2] * t1[2];
  }
}

// Closed-form rigid point-to-point estimate from raw moments (transform_estimation.hpp:11-48).
// The Kabsch rotation U V^T of a cross-covariance with POSITIVE determinant is its orthogonal polar factor: Newton's iteration with
// Higham's scaling, X <- (g X + X^-T / g) / 2, g = sqrt(|X^-1|_F / |X|_F), converges to it from X = sigma in a handful of steps for
// any non-singular sigma (the covariance of two 3-D clouds: condition number tens to thousands) -- cofactor inverses instead of
// Jacobi sweeps full of f64 square roots and divisions (the single-lane device epilogue: 14 -> ~11 us for the point-to-point metric).
// Returns false (R untouched) for det <= 0 (the reflection case of transform_estimation.hpp:38-41, planar / degenerate data: a
// singular sigma), when the iteration has not settled in 12 steps or the result is not orthogonal to 1e-13: the caller takes the SVD.
CILHIP_HD bool kabsch_rotation_polar(const double sig[9], double R[9]) {
  double X[9];
  double n2 = 0.0;
  for (int i = 0; i < 9; ++i) { X[i] = sig[i]; n2 += sig[i] * sig[i]; }
  if (!(n2 > 0.0) || !(n2 < 1.0e300)) return false;
  for (int it = 0; it < 12; ++it) {
    double C[9];      // cofactors: C = det(X) * X^-T
    C[0] = X[4] * X[8] - X[5] * X[7]; C[1] = X[5] * X[6] - X[3] * X[8]; C[2] = X[3] * X[7] - X[4] * X[6];
    C[3] = X[2] * X[7] - X[1] * X[8]; C[4] = X[0] * X[8] - X[2] * X[6]; C[5] = X[1] * X[6] - X[0] * X[7];
    C[6] = X[1] * X[5] - X[2] * X[4]; C[7] = X[2] * X[3] - X[0] * X[5]; C[8] = X[0] * X[4] - X[1] * X[3];
    const double det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
    double xn = 0.0, cn = 0.0;
    for (int i = 0; i < 9; ++i) { xn += X[i] * X[i]; cn += C[i] * C[i]; }
    // (relative to the matrix' own scale: det <= 1e-12 |X|^3 is singular for this purpose -- the reflection / planar cases)
    if (!(det > 1.0e-12 * xn * sqrt(xn))) return false;
    const double g2 = sqrt(cn) / (det * sqrt(xn));      // = |X^-1|_F / |X|_F = g^2
    const double g = sqrt(g2);
    const double a = 0.5 * g, b = 0.5 / (g * det);
    double diff = 0.0;
    for (int i = 0; i < 9; ++i) {
      const double v = a * X[i] + b * C[i];
      diff = fmax(diff, fabs(v - X[i]));
      X[i] = v;
    }
    if (it > 0 && diff <= 4.0e-16) {
      for (int r = 0; r < 3; ++r)
        for (int c = r; c < 3; ++c) {
          const double gq = X[r] * X[c] + X[3 + r] * X[3 + c] + X[6 + r] * X[6 + c] - (r == c ? 1.0 : 0.0);
          if (!(fabs(gq) <= 1.0e-13)) return false;
        }
      for (int i = 0; i < 9; ++i) R[i] = X[i];
      return true;
    }
  }
  return false;
}

// sums: SUMS_KABSCH layout. Identity when n == 0 (:20-23).
CILHIP_HD void kabsch_from_sums(const double* sums, double L[9], double t[3]) {
  for (int i = 0; i < 9; ++i) L[i] = (i % 4 == 0) ? 1.0 : 0.0;
  t[0] = t[1] = t[2] = 0.0;
  const double n = sums[0];
  if (!(n > 0.0)) return;
  double mud[3], mus[3], sig[9];
  for (int c = 0; c < 3; ++c) { mud[c] = sums[1 + c] / n; mus[c] = sums[4 + c] / n; }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) sig[r * 3 + c] = sums[7 + r * 3 + c] / n - mud[r] * mus[c];
  if (!kabsch_rotation_polar(sig, L)) {
    double U[9], S[3], V[9];
    svd3(sig, U, S, V);
    uvt_fix(U, V, 2, L);
  }
  for (int r = 0; r < 3; ++r) t[r] = mud[r] - (L[r * 3] * mus[0] + L[r * 3 + 1] * mus[1] + L[r * 3 + 2] * mus[2]);
}

// Normal equations of one Gauss-Newton step from the raw sums (transform_estimation.hpp:292-344).
// point_weighted: the sums carry per-pair weights; the point block's "n" is then the sum of the weights, slot 43.
CILHIP_HD void gn_normal_equations(const double* sums, double w_p2p, double w_p2pl, double AtA[36],
                                   double Atb[6], bool point_weighted = false) {
  // (every loop fully unrolled: the outputs stay in registers on the device)
#pragma unroll
  for (int i = 0; i < 36; ++i) AtA[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) Atb[i] = 0.0;
  if (w_p2pl > 0.0) {
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = a; b < 6; ++b) {
        const int k = 1 + a * 6 - (a * (a - 1)) / 2 + (b - a);      // slot of the (a, b) entry of the upper triangle, row by row
        const double v = w_p2pl * sums[k];
        AtA[a * 6 + b] += v;
        if (b != a) AtA[b * 6 + a] += v;
      }
#pragma unroll
    for (int a = 0; a < 6; ++a) Atb[a] += w_p2pl * sums[22 + a];
  }
  if (w_p2p > 0.0) {
    // E = [[a]x ; I3]  =>  E E^T = [[ (a.a) I - a a^T , [a]x ], [ [a]x^T , I ]],  E r = [a x r ; r]
    const double n = point_weighted ? sums[43] : sums[0];
    const double sa[3] = {sums[28], sums[29], sums[30]};
    const double aa00 = sums[31], aa01 = sums[32], aa02 = sums[33], aa11 = sums[34], aa12 = sums[35], aa22 = sums[36];
    const double tr = aa00 + aa11 + aa22;
    const double TL[9] = {tr - aa00, -aa01, -aa02, -aa01, tr - aa11, -aa12, -aa02, -aa12, tr - aa22};
    const double X[9] = {0.0, -sa[2], sa[1], sa[2], 0.0, -sa[0], -sa[1], sa[0], 0.0};  // sum [a]x
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        AtA[r * 6 + c] += w_p2p * TL[r * 3 + c];
        AtA[r * 6 + 3 + c] += w_p2p * X[r * 3 + c];
        AtA[(3 + r) * 6 + c] += w_p2p * X[c * 3 + r];
      }
#pragma unroll
    for (int r = 0; r < 3; ++r) AtA[(3 + r) * 6 + 3 + r] += w_p2p * n;
#pragma unroll
    for (int r = 0; r < 3; ++r) { Atb[r] += w_p2p * sums[37 + r]; Atb[3 + r] += w_p2p * sums[40 + r]; }
  }
}

// Instance-class tail: rotation() polish, transform_ = tform_iter * transform_, delta norm.
// T_cur/T_new col-major float 4x4 (Eigen Isometry storage).
CILHIP_HD float compose_update(const double Lin[9], const double t[3], const float T_cur[16], float T_new[16]) {
  double R[9];
  if (!nearest_rotation_polar(Lin, R)) nearest_rotation(Lin, R);
  double Lc[9], tc[3];
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Lc[r * 3 + c] = (double)T_cur[c * 4 + r]; tc[r] = (double)T_cur[12 + r]; }
  float out[16];
  for (int i = 0; i < 16; ++i) out[i] = 0.0f;
  out[15] = 1.0f;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c)
      out[c * 4 + r] = (float)(R[r * 3] * Lc[c] + R[r * 3 + 1] * Lc[3 + c] + R[r * 3 + 2] * Lc[6 + c]);
    out[12 + r] = (float)((R[r * 3] * tc[0] + R[r * 3 + 1] * tc[1] + R[r * 3 + 2] * tc[2]) + t[r]);
  }
  for (int i = 0; i < 16; ++i) T_new[i] = out[i];
  double dn = 0.0;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) { const double v = R[r * 3 + c] - ((r == c) ? 1.0 : 0.0); dn += v * v; }
  for (int r = 0; r < 3; ++r) dn += t[r] * t[r];
  return (float)sqrt(dn);
}

// The pinned f32 transform expression (DESIGN.md "Numeric contract"): q_r = (L_r0*x + (L_r1*y + L_r2*z)) + t_r,
// every operation individually rounded -- no FMA contraction.
// common_transformable_feature_adaptors.hpp:28-33.
#if defined(__HIP_DEVICE_COMPILE__)
#define CILHIP_MUL(a, b) __fmul_rn((a), (b))
#define CILHIP_ADD(a, b) __fadd_rn((a), (b))
#define CILHIP_SUB(a, b) __fsub_rn((a), (b))
#else
// host side is compiled with -ffp-contract=off (see build.py)
#define CILHIP_MUL(a, b) ((a) * (b))
#define CILHIP_ADD(a, b) ((a) + (b))
#define CILHIP_SUB(a, b) ((a) - (b))
#endif

// tform.linear().inverse().transpose() in f32, restating Eigen's fixed-size 3x3 inverse (cofactors, det along column 0 with the
// 3-term pairing x0 + (x1 + x2), multiplication by the reciprocal of det): what PointNormalFeaturesAdaptor applies to normals
// under a non-rigid transform (common_transformable_feature_adaptors.hpp:118-122).  M row-major.  (Eigen's own evaluation order
// is unpinnable here -- Eigen is absent; oracle and engine share this restatement.)
CILHIP_HD void linear_inverse_transpose_f32(const float T[16], float M[9]) {
  float m[3][3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) m[r][c] = T[c * 4 + r];
  float cof[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      cof[i][j] = CILHIP_SUB(CILHIP_MUL(m[i1][j1], m[i2][j2]), CILHIP_MUL(m[i1][j2], m[i2][j1]));
    }
  const float det = CILHIP_ADD(CILHIP_MUL(cof[0][0], m[0][0]), CILHIP_ADD(CILHIP_MUL(cof[1][0], m[1][0]), CILHIP_MUL(cof[2][0], m[2][0])));
  const float invdet = 1.0f / det;
  // inverse(i, j) = cofactor(j, i) * invdet  =>  (inverse^T)(i, j) = cofactor(i, j) * invdet
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i * 3 + j] = CILHIP_MUL(cof[i][j], invdet);
}

CILHIP_HD void transform_point(const float T[16], float x, float y, float z, float& qx, float& qy, float& qz) {
  qx = CILHIP_ADD(CILHIP_ADD(CILHIP_MUL(T[0], x), CILHIP_ADD(CILHIP_MUL(T[4], y), CILHIP_MUL(T[8], z))), T[12]);
  qy = CILHIP_ADD(CILHIP_ADD(CILHIP_MUL(T[1], x), CILHIP_ADD(CILHIP_MUL(T[5], y), CILHIP_MUL(T[9], z))), T[13]);
  qz = CILHIP_ADD(CILHIP_ADD(CILHIP_MUL(T[2], x), CILHIP_ADD(CILHIP_MUL(T[6], y), CILHIP_MUL(T[10], z))), T[14]);
}

// ---- symmetric 3x3 eigen-decomposition (cyclic Jacobi, f64), PCA convention ---------------------
// Replaces Eigen::SelfAdjointEigenSolver<Matrix3f> + the reordering of
// core/principal_component_analysis.hpp:76-84: V columns = eigenvectors by DESCENDING eigenvalue,
// last column negated when det(V) < 0.  A row-major, only read.
CILHIP_HD void sym_eig3(const double Ain[9], double w[3], double V[9]) {
  double A[9];
  for (int i = 0; i < 9; ++i) { A[i] = Ain[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 50; ++sweep) {
    const double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
    const double dia = fabs(A[0]) + fabs(A[4]) + fabs(A[8]);
    if (off <= 2.220446049250313e-16 * 0.125 * dia || off == 0.0) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq;
          A[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk;
          A[q * 3 + k] = s * apk + c * aqk;
        }
        A[p * 3 + q] = A[q * 3 + p] = 0.0;
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq;
          V[k * 3 + q] = s * vkp + c * vkq;
        }
      }
  }
  w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (w[j] > w[i]) {
        const double tw = w[i]; w[i] = w[j]; w[j] = tw;
        for (int k = 0; k < 3; ++k) { const double tv = V[k * 3 + i]; V[k * 3 + i] = V[k * 3 + j]; V[k * 3 + j] = tv; }
      }
  const double det = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
  if (det < 0.0) { V[2] = -V[2]; V[5] = -V[5]; V[8] = -V[8]; }
}

// ---- affine variants (transform_estimation.hpp:369-476 combined, :50-102 point-to-point class) --------------------
// 12 unknowns theta = (row-major 3x3 linear part, translation); eq_vec of a plane term = (n_0 s, n_1 s, n_2 s, n),
// the three eq_vecs of a point term = s in block j and e_j in the tail.  With st = (s, 1):
//   AtA[(j,a),(k,b)] = w_pl * sum n_j n_k st_a st_b  +  w_pt * delta_jk * sum st_a st_b,
//   Atb[(j,a)]       = w_pl * sum (n.d) n_j st_a     +  w_pt * sum st_a d_j,         index (j,a) = a < 3 ? 3j+a : 9+j.
// s0/s1/s2: the reduced sums of the IM_AFF0/1/2 passes (layout: internal.hpp, IterMetric).
inline int affine_pair_slot(int a, int b) {   // position of (a <= b) among the 10 pairs of 0..3, row by row
  if (a > b) { const int t = a; a = b; b = t; }
  const int base[4] = {0, 4, 7, 9};
  return base[a] + (b - a);
}
// pair_weighted: the sums carry per-pair weights (weight evaluators); the point block's "count" is then the sum of the point
// weights, slot 34 of s0
inline void affine_normal_equations(const double* s0, const double* s1, const double* s2, double w_pt, double w_pl,
                                    double AtA[144], double Atb[12], bool pair_weighted = false) {
  for (int i = 0; i < 144; ++i) AtA[i] = 0.0;
  for (int i = 0; i < 12; ++i) Atb[i] = 0.0;
  auto idx = [](int j, int a) { return a < 3 ? 3 * j + a : 9 + j; };
  auto S = [&](int a, int b) -> double {     // sum st_a st_b
    if (a > b) { const int t = a; a = b; b = t; }
    if (b < 3) { const int up[3] = {1, 4, 6}; return s0[up[a] + (b - a)]; }
    if (a < 3) return s0[7 + a];
    return pair_weighted ? s0[34] : s0[0];
  };
  auto M = [&](int j, int k, int a, int b) -> double {   // sum n_j n_k st_a st_b
    if (j > k) { const int t = j; j = k; k = t; }
    const int p = affine_pair_slot(a, b);
    if (j == 0) return s1[k * 10 + p];
    return s2[(j == 1 ? (k - 1) : 2) * 10 + p];
  };
  for (int j = 0; j < 3; ++j)
    for (int a = 0; a < 4; ++a) {
      const int r = idx(j, a);
      for (int k = 0; k < 3; ++k)
        for (int b = 0; b < 4; ++b) {
          double v = 0.0;
          if (w_pl > 0.0) v += w_pl * M(j, k, a, b);
          if (w_pt > 0.0 && j == k) v += w_pt * S(a, b);
          AtA[r * 12 + idx(k, b)] = v;
        }
      double bv = 0.0;
      if (w_pl > 0.0) bv += w_pl * s0[22 + 4 * j + a];
      if (w_pt > 0.0) bv += w_pt * (a < 3 ? s0[10 + 3 * a + j] : s0[19 + j]);
      Atb[r] = bv;
    }
}

// n x n LDL^T with diagonal pivoting and the pseudo-inverse of D (Eigen LDLT::solve semantics), n <= 12; host side.
inline void ldlt_solve_n(int n, const double* Ain, const double* bin, double* x) {
  double A[144], y[12];
  int perm[12];
  for (int i = 0; i < n * n; ++i) A[i] = Ain[i];
  for (int i = 0; i < n; ++i) perm[i] = i;
  const double tiny = 2.2250738585072014e-308;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (fabs(A[i * n + i]) > best) { best = fabs(A[i * n + i]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < n; ++j) { const double t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
      for (int i = 0; i < n; ++i) { const double t = A[i * n + k]; A[i * n + k] = A[i * n + piv]; A[i * n + piv] = t; }
      const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    const double dk = A[k * n + k];
    if (fabs(dk) <= tiny) {
      for (int i = k + 1; i < n; ++i) A[i * n + k] = 0.0;
      continue;
    }
    for (int i = k + 1; i < n; ++i) A[i * n + k] /= dk;
    for (int i = k + 1; i < n; ++i)
      for (int j = k + 1; j <= i; ++j) {
        A[i * n + j] -= A[i * n + k] * dk * A[j * n + k];
        A[j * n + i] = A[i * n + j];
      }
  }
  for (int i = 0; i < n; ++i) y[i] = bin[perm[i]];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j) y[i] -= A[i * n + j] * y[j];
  for (int i = 0; i < n; ++i) {
    const double d = A[i * n + i];
    y[i] = (fabs(d) > tiny) ? y[i] / d : 0.0;
  }
  for (int i = n - 1; i >= 0; --i)
    for (int j = i + 1; j < n; ++j) y[i] -= A[j * n + i] * y[j];
  for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
}

}  // namespace cilhip
