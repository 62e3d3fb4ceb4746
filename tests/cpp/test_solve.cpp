// Host-side check of the fast paths in cilantro_amd/csrc/solve.hpp (compiled with hipcc, runs without a GPU):
//   nearest_rotation_polar  vs the SVD-based nearest_rotation (U V^T, the reference's rotation() polish)
//   kabsch_rotation_polar   vs the SVD-based Kabsch rotation (U V^T with the reflection fix)
//   ldlt6_solve_fast        vs the pivoted LDL^T with pseudo-inverse semantics
#include "../../cilantro_amd/csrc/solve.hpp"

#include <cstdint>
#include <cstdio>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand() {   // xorshift64*, [0,1)
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return (double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}

int main() {
  using namespace cilhip;
  int failures = 0;
  // near-rotations: Rodrigues rotation times (I + small perturbation), as the Gauss-Newton step produces them
  double worst = 0.0;
  int fast_taken = 0;
  for (int trial = 0; trial < 20000; ++trial) {
    double ax[3] = {urand() - 0.5, urand() - 0.5, urand() - 0.5};
    const double nrm = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + 1e-30;
    for (double& v : ax) v /= nrm;
    const double ang = (trial % 5 == 0 ? 3.0 : 0.05) * urand(), c = std::cos(ang), s = std::sin(ang);
    double Rm[9] = {c + ax[0] * ax[0] * (1 - c), ax[0] * ax[1] * (1 - c) - ax[2] * s, ax[0] * ax[2] * (1 - c) + ax[1] * s,
                    ax[1] * ax[0] * (1 - c) + ax[2] * s, c + ax[1] * ax[1] * (1 - c), ax[1] * ax[2] * (1 - c) - ax[0] * s,
                    ax[2] * ax[0] * (1 - c) - ax[1] * s, ax[2] * ax[1] * (1 - c) + ax[0] * s, c + ax[2] * ax[2] * (1 - c)};
    const double eps = (trial % 3 == 0) ? 1e-2 : 1e-6;
    double L[9];
    for (int i = 0; i < 9; ++i) L[i] = Rm[i] + eps * (urand() - 0.5);
    double A[9], B[9];
    nearest_rotation(L, A);
    if (nearest_rotation_polar(L, B)) {
      ++fast_taken;
      for (int i = 0; i < 9; ++i) worst = std::fmax(worst, std::fabs(A[i] - B[i]));
    }
  }
  std::printf("polar vs SVD: fast path taken %d / 20000, max |dR| = %.3e\n", fast_taken, worst);
  if (!(worst <= 5e-15) || fast_taken < 19990) ++failures;
  // matrices the fast path must refuse: reflections, far from orthogonal, singular, NaN
  {
    const double refl[9] = {1, 0, 0, 0, 1, 0, 0, 0, -1}, far[9] = {2, 0, 0, 0, 1, 0, 0, 0, 1}, sing[9] = {1, 0, 0, 0, 1, 0, 0, 0, 0};
    double nanm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; nanm[4] = NAN;
    double R[9];
    if (nearest_rotation_polar(refl, R) || nearest_rotation_polar(far, R) || nearest_rotation_polar(sing, R) || nearest_rotation_polar(nanm, R)) {
      std::printf("polar fast path accepted a matrix it must refuse\n"); ++failures;
    }
  }
  // Kabsch rotation of a cross-covariance: scaled polar iteration vs the SVD (U V^T with the reference's reflection fix).  Covariances
  // of random clouds under random rotations, anisotropic up to 1 : 1e-3 per axis (condition numbers up to ~1e6), plus noise.
  {
    double worst_k = 0.0;
    int taken = 0, trials = 0;
    for (int trial = 0; trial < 20000; ++trial) {
      double ax[3] = {urand() - 0.5, urand() - 0.5, urand() - 0.5};
      const double nrm = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + 1e-30;
      for (double& v : ax) v /= nrm;
      const double ang = 3.1 * urand(), c = std::cos(ang), s = std::sin(ang);
      const double Rm[9] = {c + ax[0] * ax[0] * (1 - c), ax[0] * ax[1] * (1 - c) - ax[2] * s, ax[0] * ax[2] * (1 - c) + ax[1] * s,
                            ax[1] * ax[0] * (1 - c) + ax[2] * s, c + ax[1] * ax[1] * (1 - c), ax[1] * ax[2] * (1 - c) - ax[0] * s,
                            ax[2] * ax[0] * (1 - c) - ax[1] * s, ax[2] * ax[1] * (1 - c) + ax[0] * s, c + ax[2] * ax[2] * (1 - c)};
      const double sc[3] = {1.0, std::pow(10.0, -3.0 * urand()), std::pow(10.0, -3.0 * urand())};
      double sig[9] = {0};      // sum over points of (R p)(p)^T, p anisotropic
      for (int k = 0; k < 30; ++k) {
        double pt[3] = {sc[0] * (urand() - 0.5), sc[1] * (urand() - 0.5), sc[2] * (urand() - 0.5)}, q[3];
        for (int r = 0; r < 3; ++r) q[r] = Rm[r * 3] * pt[0] + Rm[r * 3 + 1] * pt[1] + Rm[r * 3 + 2] * pt[2] + 1e-4 * sc[2] * (urand() - 0.5);
        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) sig[r * 3 + cc] += q[r] * pt[cc];
      }
      double U[9], S[3], V[9], A[9], B[9];
      svd3(sig, U, S, V);
      uvt_fix(U, V, 2, A);
      ++trials;
      if (kabsch_rotation_polar(sig, B)) {
        ++taken;
        for (int i = 0; i < 9; ++i) worst_k = std::fmax(worst_k, std::fabs(A[i] - B[i]));
      }
    }
    std::printf("Kabsch polar vs SVD: fast path taken %d / %d, max |dR| = %.3e\n", taken, trials, worst_k);
    if (!(worst_k <= 1e-9) || taken < trials * 9 / 10) ++failures;
    // must refuse: reflection (det < 0), singular (planar cloud), zero, NaN
    const double refl[9] = {1, 0, 0, 0, 1, 0, 0, 0, -1}, sing[9] = {1, 0, 0, 0, 1, 0, 0, 0, 0}, zero[9] = {0};
    double nanm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; nanm[2] = NAN;
    double R[9];
    if (kabsch_rotation_polar(refl, R) || kabsch_rotation_polar(sing, R) || kabsch_rotation_polar(zero, R) || kabsch_rotation_polar(nanm, R)) {
      std::printf("Kabsch polar fast path accepted a matrix it must refuse\n"); ++failures;
    }
  }
  // LDL^T: random SPD normal equations (J^T J of 40 random rows + ridge), against the pivoted solve
  double worst_rel = 0.0;
  for (int trial = 0; trial < 5000; ++trial) {
    double A[36] = {0}, b[6], x1[6], x2[6];
    for (int r = 0; r < 40; ++r) {
      double row[6];
      for (double& v : row) v = urand() - 0.5;
      if (trial % 2) { row[3] *= 1e3; row[0] *= 1e-2; }   // badly scaled unknowns, as rotation vs translation terms are
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) A[i * 6 + j] += row[i] * row[j];
    }
    for (double& v : b) v = urand() - 0.5;
    double W[36], y[6]; int perm[6];
    ldlt6_solve_ws(A, b, x1, W, y, perm);
    if (!ldlt6_solve_fast(A, b, x2)) { std::printf("fast LDLT refused an SPD system\n"); ++failures; break; }
    double num = 0, den = 0;
    for (int i = 0; i < 6; ++i) { num += (x1[i] - x2[i]) * (x1[i] - x2[i]); den += x1[i] * x1[i]; }
    worst_rel = std::fmax(worst_rel, std::sqrt(num / (den + 1e-300)));
  }
  std::printf("LDLT fast vs pivoted: max relative difference %.3e\n", worst_rel);
  if (!(worst_rel <= 1e-9)) ++failures;
  // rank-deficient / indefinite / zero systems go to the pivoted solve
  {
    double A[36] = {0}, b[6] = {1, 2, 3, 4, 5, 6}, x[6];
    if (ldlt6_solve_fast(A, b, x)) { std::printf("fast LDLT accepted the zero matrix\n"); ++failures; }
    for (int i = 0; i < 5; ++i) A[i * 6 + i] = 1.0;       // rank 5
    if (ldlt6_solve_fast(A, b, x)) { std::printf("fast LDLT accepted a rank-deficient matrix\n"); ++failures; }
    A[35] = -1.0;                                          // indefinite
    if (ldlt6_solve_fast(A, b, x)) { std::printf("fast LDLT accepted an indefinite matrix\n"); ++failures; }
    ldlt6_solve(A, b, x);                                  // the wrapper still answers (pivoted)
    if (!(std::fabs(x[0] - 1.0) < 1e-12 && std::fabs(x[5] + 6.0) < 1e-12)) { std::printf("wrapper fallback wrong\n"); ++failures; }
  }
  // gn_normal_equations: weighted correspondences against a direct per-pair assembly (point terms E = [[a]x ; I], plane terms
  // e = [a x n ; n]); with per-pair weights the point block's translation part is the SUM OF THE WEIGHTS (slot 43), not the count
  {
    double sums[SUMS_MAX] = {0}, AtA_ref[36] = {0}, Atb_ref[6] = {0};
    const int N = 200;
    for (int k = 0; k < N; ++k) {
      double a[3], r[3], nrm[3];
      for (int c = 0; c < 3; ++c) { a[c] = urand() - 0.5; r[c] = 0.1 * (urand() - 0.5); nrm[c] = urand() - 0.5; }
      const double wq = 0.2 + urand(), wp = 0.2 + urand();
      double e[6] = {a[1] * nrm[2] - a[2] * nrm[1], a[2] * nrm[0] - a[0] * nrm[2], a[0] * nrm[1] - a[1] * nrm[0], nrm[0], nrm[1], nrm[2]};
      const double res = nrm[0] * r[0] + nrm[1] * r[1] + nrm[2] * r[2];
      double E[6][3] = {{0, -a[2], a[1]}, {a[2], 0, -a[0]}, {-a[1], a[0], 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
      for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) {
          AtA_ref[i * 6 + j] += wp * e[i] * e[j];
          for (int c = 0; c < 3; ++c) AtA_ref[i * 6 + j] += wq * E[i][c] * E[j][c];
        }
        Atb_ref[i] += wp * res * e[i];
        for (int c = 0; c < 3; ++c) Atb_ref[i] += wq * E[i][c] * r[c];
      }
      // the accumulation kernels' slot layout (solve.hpp), weights folded in
      sums[0] += 1.0;
      int s = 1;
      for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) sums[s++] += wp * e[i] * e[j];
      for (int i = 0; i < 6; ++i) sums[22 + i] += wp * res * e[i];
      for (int c = 0; c < 3; ++c) sums[28 + c] += wq * a[c];
      sums[31] += wq * a[0] * a[0]; sums[32] += wq * a[0] * a[1]; sums[33] += wq * a[0] * a[2];
      sums[34] += wq * a[1] * a[1]; sums[35] += wq * a[1] * a[2]; sums[36] += wq * a[2] * a[2];
      sums[37] += wq * (a[1] * r[2] - a[2] * r[1]); sums[38] += wq * (a[2] * r[0] - a[0] * r[2]); sums[39] += wq * (a[0] * r[1] - a[1] * r[0]);
      for (int c = 0; c < 3; ++c) sums[40 + c] += wq * r[c];
      sums[43] += wq;
    }
    double AtA[36], Atb[6], worst_w = 0.0;
    gn_normal_equations(sums, 1.0, 1.0, AtA, Atb, true);
    for (int i = 0; i < 36; ++i) worst_w = std::fmax(worst_w, std::fabs(AtA[i] - AtA_ref[i]));
    for (int i = 0; i < 6; ++i) worst_w = std::fmax(worst_w, std::fabs(Atb[i] - Atb_ref[i]));
    std::printf("weighted normal equations vs per-pair assembly: max |difference| = %.3e\n", worst_w);
    if (!(worst_w <= 1e-11)) ++failures;
    gn_normal_equations(sums, 1.0, 1.0, AtA, Atb, false);      // unweighted reading: the count instead of the weight sum
    if (std::fabs(AtA[3 * 6 + 3] - AtA_ref[3 * 6 + 3]) < 1e-6) { std::printf("the point block ignored the weight sum\n"); ++failures; }
  }
  std::printf(failures ? "FAILED (%d)\n" : "ALL OK\n", failures);
  return failures ? 1 : 0;
}
