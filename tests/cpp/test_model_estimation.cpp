// tests/cpp/test_model_estimation.cpp -- the C++ mirrors of PlaneRANSACEstimator3f and KMeans3f
// (include/cilantro_hip/model_estimation.hpp) used the way examples/ransac_plane_estimator.cpp:32-41 and
// examples/kmeans.cpp use cilantro, checked against the CPU oracle (linked as the CHECKER only).
//   test_model_estimation                      : needs a GPU; exit 0 iff every check passes
//   test_model_estimation --expect-no-device   : CPU box; exit 0 iff the calls fail loudly
#include <cilantro_hip/model_estimation.hpp>
#include <cilantro_hip/normal_estimation.hpp>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../oracle/icp_oracle.h"

static uint64_t splitmix(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static float u01(uint64_t seed, uint64_t i) { return (float)(splitmix(seed, i) >> 40) * (1.0f / 16777216.0f); }

int main(int argc, char** argv) {
  const bool expect_no_device = argc > 1 && !std::strcmp(argv[1], "--expect-no-device");
  using namespace cilantro_hip;
  const size_t n = 100003;
  std::vector<float> x(3 * n);
  for (size_t i = 0; i < n; ++i) {
    x[3 * i] = 2 * u01(1, 3 * i) - 1; x[3 * i + 1] = 2 * u01(1, 3 * i + 1) - 1; x[3 * i + 2] = 2 * u01(1, 3 * i + 2) - 1;
    if (i % 5 < 3) x[3 * i + 2] = 0.3f * x[3 * i] - 0.2f * x[3 * i + 1] + 0.1f + 0.008f * (u01(2, i) - 0.5f);   // 60% on a plane
  }
  int failures = 0;
  try {
    const size_t max_iter = 60;
    std::vector<uint32_t> samples(3 * max_iter);
    for (size_t i = 0; i < samples.size(); ++i) samples[i] = (uint32_t)(splitmix(3, i) % n);
    const ConstPointsView xv(x);
    PlaneRANSACEstimator3f pe(xv);
    pe.setMaxInlierResidual(0.01f).setTargetInlierCount((size_t)(0.59 * n)).setMaxNumberOfIterations(max_iter).setReEstimationStep(true);
    pe.setSamples(samples);
    const Hyperplane3f plane = pe.estimate().getModel();
    if (expect_no_device) { std::printf("FAIL: estimate() succeeded without a device\n"); return 1; }
    float po[4]; size_t ko = 0;
    std::vector<float> ro(n); std::vector<uint32_t> io(n);
    const size_t ito = orc_plane_ransac(x.data(), n, samples.data(), max_iter, 0.01f, (size_t)(0.59 * n), 1, ORC_MODE_MIXED, po, ro.data(), io.data(), &ko);
    const float sgn = (plane.coeffs()[0] * po[0] + plane.coeffs()[1] * po[1] + plane.coeffs()[2] * po[2]) > 0 ? 1.0f : -1.0f;
    float dmax = 0;
    for (int d = 0; d < 4; ++d) dmax = std::fmax(dmax, std::fabs(plane.coeffs()[d] - sgn * po[d]));
    std::vector<float> chk(n);
    orc_plane_residuals(x.data(), n, plane.coeffs(), chk.data());
    size_t bad = 0, k = 0;
    const auto& inl = pe.getModelInliers();
    for (size_t i = 0; i < n; ++i) {
      bad += chk[i] != pe.getModelResiduals()[i];
      if (chk[i] <= 0.01f) { bad += (k >= inl.size() || inl[k] != i); ++k; }
    }
    bad += k != inl.size();
    std::printf("plane RANSAC: iters gpu=%zu oracle=%zu, inliers gpu=%zu oracle=%zu, |plane diff|=%.2e, residual/inlier mismatches=%zu, absDistance(p0)=%.4f\n",
                pe.getNumberOfPerformedIterations(), ito, inl.size(), ko, dmax, bad, plane.absDistance(x.data()));
    if (pe.getNumberOfPerformedIterations() != ito || bad || !(dmax <= 2e-6f) || !pe.targetInlierCountAchieved()) ++failures;
    const Hyperplane3f all = pe.estimateModel();
    float pa[4];
    orc_plane_fit(x.data(), nullptr, n, ORC_MODE_MIXED, pa);
    const float sg2 = (all.coeffs()[0] * pa[0] + all.coeffs()[1] * pa[1] + all.coeffs()[2] * pa[2]) > 0 ? 1.0f : -1.0f;
    for (int d = 0; d < 4; ++d) if (!(std::fabs(all.coeffs()[d] - sg2 * pa[d]) <= 2e-6f)) ++failures;

    // KMeans3f with explicit initial centroids (examples/kmeans.cpp uses a cluster count; that draws at random)
    const size_t kc = 37;
    std::vector<float> c0(x.begin(), x.begin() + 3 * kc), co(c0);
    KMeans3f km(xv);
    km.cluster(ConstPointsView(c0.data(), kc), 8, 0.0f);
    std::vector<int64_t> lo(n, 0);
    const size_t itk = orc_kmeans(x.data(), n, co.data(), kc, 8, 0.0f, ORC_MODE_MIXED, lo.data());
    float cmax = 0; size_t ldiff = 0;
    for (size_t i = 0; i < 3 * kc; ++i) cmax = std::fmax(cmax, std::fabs(km.getClusterCentroids()[i] - co[i]));
    for (size_t i = 0; i < n; ++i) ldiff += km.getPointToClusterIndexMap()[i] != (size_t)lo[i];
    std::printf("KMeans3f: iters gpu=%zu oracle=%zu, max centroid diff=%.2e, label mismatches=%zu, clusters=%zu\n",
                km.getNumberOfPerformedIterations(), itk, cmax, ldiff, km.getClusterToPointIndicesMap().size());
    if (km.getNumberOfPerformedIterations() != itk || !(cmax <= 1e-6f) || ldiff > 2) ++failures;
    km.cluster(5, 3);
    if (km.getNumberOfClusters() != 5) ++failures;

    // KDTree3f::kNNSearch and NormalEstimation3f (examples/normal_estimation.cpp uses k-NN normals with a view point)
    const size_t nq = 2000, kk = 9;
    KDTree3f tree(xv);
    const NeighborhoodSet nns = tree.kNNSearch(ConstPointsView(x.data(), nq), kk);
    orc_kdtree* ot = orc_kdtree_build(x.data(), n, 10);
    std::vector<int64_t> oi(nq * kk); std::vector<float> od(nq * kk); std::vector<uint32_t> oc(nq);
    orc_knn_batch(ot, x.data(), nq, kk, INFINITY, oi.data(), od.data(), oc.data());
    size_t kbad = 0;
    for (size_t i = 0; i < nq; ++i) {
      kbad += nns[i].size() != oc[i];
      for (size_t j = 0; j < nns[i].size() && j < oc[i]; ++j) kbad += (nns[i][j].value != od[i * kk + j]) || ((int64_t)nns[i][j].index != oi[i * kk + j]);
    }
    orc_kdtree_free(ot);
    // KDTree3f::radiusSearch (list-returning, kd_tree.hpp:266-282) against the oracle's exhaustive search
    {
      const size_t nr = 300;
      const float r2 = 0.004f;
      const NeighborhoodSet rs = tree.radiusSearch(ConstPointsView(x.data(), nr), r2);
      std::vector<uint64_t> off(nr + 1);
      const size_t total = orc_radius_search(x.data(), n, x.data(), nr, r2, off.data(), nullptr, nullptr, 0);
      std::vector<int64_t> ri(total ? total : 1); std::vector<float> rd(total ? total : 1);
      orc_radius_search(x.data(), n, x.data(), nr, r2, off.data(), ri.data(), rd.data(), total);
      size_t rbad = 0;
      for (size_t i = 0; i < nr; ++i) {
        rbad += rs[i].size() != (size_t)(off[i + 1] - off[i]);
        for (size_t j = 0; j < rs[i].size() && j < (size_t)(off[i + 1] - off[i]); ++j)
          rbad += ((int64_t)rs[i][j].index != ri[off[i] + j]) || (rs[i][j].value != rd[off[i] + j]);
      }
      std::printf("radiusSearch: %zu neighbours over %zu queries, %zu mismatches vs oracle\n", total, nr, rbad);
      if (rbad || total == 0) ++failures;
    }
    NormalEstimation3f ne(xv);
    ne.setViewPoint(0.0f, 0.0f, 10.0f);
    std::vector<float> nrm, cur, onrm(3 * n), ocur(n);
    ne.getNormalsAndCurvatureKNN(nrm, cur, 10);
    const float vp[3] = {0.0f, 0.0f, 10.0f};
    orc_normals_knn(x.data(), n, 10, INFINITY, vp, ORC_MODE_MIXED, onrm.data(), ocur.data());
    size_t nbad = 0, nwell = 0;
    for (size_t i = 0; i < n; ++i) {
      if (!(cur[i] < 0.2f)) continue;   // nearly isotropic neighbourhoods: the normal itself is ill-conditioned
      ++nwell;
      const float dot = nrm[3 * i] * onrm[3 * i] + nrm[3 * i + 1] * onrm[3 * i + 1] + nrm[3 * i + 2] * onrm[3 * i + 2];
      nbad += !(dot > 1.0f - 1e-4f) || !(std::fabs(cur[i] - ocur[i]) < 1e-4f);
    }
    std::printf("k-NN: %zu mismatches vs oracle over %zu queries; normals: %zu of %zu well-defined ones differ\n", kbad, nq, nbad, nwell);
    if (kbad > 2 || nbad * 1000 > nwell) ++failures;
    {
      // RigidTransformRANSACEstimator3f (ransac_transform_estimator.hpp) through its index-list constructor: a known motion with
      // 30 % gross outliers; iteration count and model against the oracle's restatement with the same samples, the model's own
      // residuals / inliers bit for bit
      const size_t m = 40000;
      std::vector<float> d3(3 * m), s3(3 * m);
      const float ca = std::cos(0.3f), sa = std::sin(0.3f);
      for (size_t i = 0; i < m; ++i) {
        const float px = u01(77, 3 * i), py = u01(77, 3 * i + 1), pz = u01(77, 3 * i + 2);
        s3[3 * i] = px; s3[3 * i + 1] = py; s3[3 * i + 2] = pz;
        d3[3 * i] = ca * px - sa * py + 0.2f; d3[3 * i + 1] = sa * px + ca * py - 0.1f; d3[3 * i + 2] = pz + 0.05f;
        if (u01(78, i) < 0.3f) { d3[3 * i] += 1.0f + u01(79, i); d3[3 * i + 2] -= 0.7f; }
      }
      std::vector<size_t> ia(m), ib(m);
      for (size_t i = 0; i < m; ++i) ia[i] = ib[i] = i;
      std::vector<uint32_t> samples(3 * 60);
      for (size_t i = 0; i < samples.size(); ++i) samples[i] = (uint32_t)(u01(80, i) * m) % (uint32_t)m;
      RigidTransformRANSACEstimator3f te(ConstPointsView(d3.data(), m), ConstPointsView(s3.data(), m), ia, ib);
      te.setMaxInlierResidual(1e-3f).setTargetInlierCount(m / 2).setMaxNumberOfIterations(60).setSamples(samples);
      const RigidTransform3f Tm = te.estimate().getModel();
      float To[16]; std::vector<float> ores(m); std::vector<uint32_t> oinl(m); size_t ok = 0; int ohave = 0;
      const size_t oit = orc_transform_ransac(d3.data(), s3.data(), m, samples.data(), 60, 1e-3f, m / 2, 1, ORC_MODE_MIXED, To, ores.data(), oinl.data(), &ok, &ohave);
      double dmax = 0.0;
      for (int i = 0; i < 16; ++i) dmax = std::fmax(dmax, std::fabs((double)Tm.m[i] - (double)To[i]));
      std::vector<float> chk(m);
      orc_transform_residuals(d3.data(), s3.data(), m, Tm.m, chk.data());
      size_t tbad = (te.getNumberOfPerformedIterations() != oit) + !(dmax <= 5e-6) + (te.getModelResiduals().size() != m);
      size_t k = 0;
      for (size_t i = 0; i < m && te.getModelResiduals().size() == m; ++i) {
        tbad += te.getModelResiduals()[i] != chk[i];
        if (chk[i] <= 1e-3f) { tbad += (k >= te.getModelInliers().size()) || te.getModelInliers()[k] != i; ++k; }
      }
      tbad += k != te.getModelInliers().size();
      std::printf("transform RANSAC: %zu iterations (oracle %zu), %zu inliers (oracle %zu), |T - T_oracle|max %.2e, %zu mismatches\n",
                  te.getNumberOfPerformedIterations(), oit, te.getNumberOfInliers(), ok, dmax, tbad);
      if (tbad || te.getNumberOfInliers() < m / 2) ++failures;
    }
  } catch (const std::runtime_error& e) {
    if (expect_no_device) { std::printf("OK (failed loudly): %s\n", e.what()); return 0; }
    std::printf("FAIL: %s\n", e.what());
    return 1;
  }
  std::printf(failures ? "FAILED (%d)\n" : "ALL OK\n", failures);
  return failures ? 1 : 0;
}
