// tests/cpp/test_icp.cpp -- exercises the C++ host mirror (include/cilantro_hip/icp.hpp) the way
// examples/rigid_icp.cpp:116-133 uses cilantro, and checks the result against the CPU oracle
// (linked here as the CHECKER only).
//   test_icp                      : needs a GPU; exit 0 iff every parity check passes
//   test_icp --expect-no-device   : CPU box; exit 0 iff construction fails loudly (no CPU fallback)
#include <cilantro_hip/icp.hpp>

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

static double frob(const float* a, const float* b) {
  double s = 0;
  for (int i = 0; i < 16; ++i) s += ((double)a[i] - b[i]) * ((double)a[i] - b[i]);
  return std::sqrt(s);
}

int main(int argc, char** argv) {
  const bool expect_no_device = argc > 1 && !std::strcmp(argv[1], "--expect-no-device");
  const size_t n = 60000;
  const double h = std::pow((double)n, -1.0 / 3.0);
  std::vector<float> dst(3 * n), nrm(3 * n), src(3 * n);
  for (size_t i = 0; i < 3 * n; ++i) dst[i] = u01(42, i);
  for (size_t i = 0; i < n; ++i) {
    const double z = 2.0 * u01(43, 2 * i) - 1.0, phi = 6.283185307179586 * u01(43, 2 * i + 1), r = std::sqrt(1 - z * z);
    nrm[3 * i] = (float)(r * std::cos(phi)); nrm[3 * i + 1] = (float)(r * std::sin(phi)); nrm[3 * i + 2] = (float)z;
  }
  // src = dst + noise, moved by a small rigid motion (recipe shape of examples/rigid_icp.cpp:57-62)
  const double a = 0.4 * h, ca = std::cos(a), sa = std::sin(a);
  for (size_t i = 0; i < n; ++i) {
    const double x = dst[3 * i] + (2 * u01(44, 3 * i) - 1) * 0.05 * h, y = dst[3 * i + 1] + (2 * u01(44, 3 * i + 1) - 1) * 0.05 * h,
                 z = dst[3 * i + 2] + (2 * u01(44, 3 * i + 2) - 1) * 0.05 * h;
    src[3 * i] = (float)(ca * x - sa * y + 0.3 * h); src[3 * i + 1] = (float)(sa * x + ca * y - 0.2 * h); src[3 * i + 2] = (float)(z + 0.1 * h);
  }
  const float max_sq = (float)((2 * h) * (2 * h));
  using namespace cilantro_hip;
  int failures = 0;
  try {
    const ConstPointsView dst_v(dst), nrm_v(nrm), src_v(src);
    SimpleCombinedMetricRigidICP3f icp(dst_v, nrm_v, src_v);
    if (expect_no_device) { std::printf("FAIL: construction succeeded without a device\n"); return 1; }
    icp.setMaxNumberOfOptimizationStepIterations(1).setPointToPointMetricWeight(0.0f).setPointToPlaneMetricWeight(1.0f);
    icp.correspondenceSearchEngine().setMaxDistance(max_sq);
    icp.setConvergenceTolerance(1e-5f).setMaxNumberOfIterations(30);
    RigidTransform3f tf = icp.estimate().getTransform();
    {
      // after estimate() the engine holds the LAST iteration's correspondences (correspondence_search_kd_tree.hpp:231 keeps
      // them, icp_base.hpp:32-38 hands the engine out): as many as the last update was computed from, ascending source index,
      // and exactly what a search under the transform that iteration started from finds
      const auto last = icp.correspondenceSearchEngine().getCorrespondences();     // (a copy: the engine's set is replaced below)
      float Tl[16];
      int origin = -1;
      cilhip_get_last_matches_origin(icp.context(), &origin);
      size_t bad = (last.size() != icp.getNumberOfLastCorrespondences()) || last.empty() || cilhip_get_matches_transform(icp.context(), Tl) != CILHIP_OK;
      for (size_t k = 1; k < last.size(); ++k) bad += !(last[k - 1].indexInSecond < last[k].indexInSecond);
      RigidTransform3f tl;
      std::memcpy(tl.m, Tl, sizeof(Tl));
      const auto& again = icp.correspondenceSearchEngine().findCorrespondences(tl).getCorrespondences();
      bad += again.size() != last.size();
      for (size_t k = 0; k < last.size() && k < again.size(); ++k)
        bad += (last[k].indexInFirst != again[k].indexInFirst) || (last[k].indexInSecond != again[k].indexInSecond) || (last[k].value != again[k].value);
      std::printf("getCorrespondences() after estimate(): %zu correspondences (origin %d), %zu mismatches vs a search under that iteration's transform\n",
                  last.size(), origin, bad);
      if (bad) ++failures;
    }

    orc_icp_params p; std::memset(&p, 0, sizeof(p));
    p.metric = 1; p.w_p2p = 0; p.w_p2pl = 1; p.max_iter = 30; p.conv_tol = 1e-5f; p.max_opt_iter = 1; p.opt_conv_tol = 1e-5f;
    p.max_sq_dist = max_sq; p.mode = ORC_MODE_MIXED; p.num_threads = 0;
    orc_icp_result r;
    orc_icp_run(dst.data(), nrm.data(), n, src.data(), nullptr, n, nullptr, &p, nullptr, &r);
    const double e1 = frob(tf.m, r.T);
    std::printf("combined: iters gpu=%zu oracle=%zu converged=%d |T_gpu-T_oracle|_F=%.3e\n", icp.getNumberOfPerformedIterations(), r.iterations, (int)icp.hasConverged(), e1);
    if (!(e1 <= 1e-5) || !icp.hasConverged()) ++failures;

    // engine concept: findCorrespondences(tform) / getCorrespondences() vs the oracle's kd-tree
    icp.correspondenceSearchEngine().findCorrespondences(tf);
    const auto& corr = icp.correspondenceSearchEngine().getCorrespondences();
    std::vector<float> q(3 * n);
    orc_transform_points(tf.m, src.data(), n, q.data());
    orc_kdtree* tree = orc_kdtree_build(dst.data(), n, 10);
    std::vector<int64_t> di(n), si(n); std::vector<float> dv(n);
    const size_t nc = orc_find_correspondences(tree, q.data(), n, max_sq, di.data(), si.data(), dv.data(), 0);
    size_t bad = (nc != corr.size());
    for (size_t k = 0; k < nc && k < corr.size(); ++k)
      bad += (corr[k].indexInFirst != (size_t)di[k]) || (corr[k].indexInSecond != (size_t)si[k]) || (corr[k].value != dv[k]);
    std::printf("engine: %zu correspondences, %zu mismatches vs oracle\n", corr.size(), bad);
    if (bad) ++failures;
    orc_kdtree_free(tree);

    std::vector<float> res = icp.getResiduals();
    if (res.size() != n) ++failures;

    SimplePointToPointMetricRigidICP3f icp2(dst_v, src_v);
    icp2.correspondenceSearchEngine().setMaxDistance(max_sq);
    icp2.setMaxNumberOfIterations(30).setConvergenceTolerance(1e-5f).estimate();
    p.metric = 0;
    orc_icp_run(dst.data(), nullptr, n, src.data(), nullptr, n, nullptr, &p, nullptr, &r);
    const double e2 = frob(icp2.getTransform().m, r.T);
    std::printf("point-to-point: iters gpu=%zu oracle=%zu |T_gpu-T_oracle|_F=%.3e\n", icp2.getNumberOfPerformedIterations(), r.iterations, e2);
    if (!(e2 <= 1e-5)) ++failures;

    // search direction BOTH with reciprocity (correspondence_search_kd_tree_utilities.hpp:65-101) through the same classes
    SimpleCombinedMetricRigidICP3f icp3(dst_v, nrm_v, src_v);
    icp3.setMaxNumberOfOptimizationStepIterations(1).setPointToPointMetricWeight(0.0f).setPointToPlaneMetricWeight(1.0f);
    icp3.correspondenceSearchEngine().setMaxDistance(max_sq).setSearchDirection(CorrespondenceSearchDirection::BOTH).setRequireReciprocality(true);
    icp3.setConvergenceTolerance(1e-5f).setMaxNumberOfIterations(8).estimate();
    p.metric = 1; p.max_iter = 8; p.direction = 2; p.reciprocal = 1;
    orc_icp_run(dst.data(), nrm.data(), n, src.data(), nullptr, n, nullptr, &p, nullptr, &r);
    const double e3 = frob(icp3.getTransform().m, r.T);
    std::printf("BOTH+reciprocal: iters gpu=%zu oracle=%zu |T_gpu-T_oracle|_F=%.3e\n", icp3.getNumberOfPerformedIterations(), r.iterations, e3);
    if (!(e3 <= 1e-5) || icp3.getNumberOfPerformedIterations() != r.iterations) ++failures;

    // correspondence weight evaluators (icp_single_transform_combined_metric.hpp:95-101): RBF on the plane terms, Identity on the point terms
    SimpleCombinedMetricRigidICP3f icp6(dst_v, nrm_v, src_v);
    icp6.setPointToPointMetricWeight(0.2f).setPointToPlaneMetricWeight(1.0f);
    icp6.pointToPlaneCorrespondenceWeightEvaluator().setKind(CorrespondenceWeightEvaluator::RBFKernel).setSigma((float)h);
    icp6.pointToPointCorrespondenceWeightEvaluator().setKind(CorrespondenceWeightEvaluator::Identity);
    icp6.correspondenceSearchEngine().setMaxDistance(max_sq);
    icp6.setConvergenceTolerance(0.0f).setMaxNumberOfIterations(8).estimate();
    std::memset(&p, 0, sizeof(p));
    p.metric = 1; p.w_p2p = 0.2f; p.w_p2pl = 1; p.max_iter = 8; p.conv_tol = 0.0f; p.max_opt_iter = 1; p.opt_conv_tol = 1e-5f;
    p.max_sq_dist = max_sq; p.mode = ORC_MODE_MIXED;
    p.point_weight_kind = ORC_W_IDENTITY; p.plane_weight_kind = ORC_W_RBF; p.point_weight_sigma = 1.0f; p.plane_weight_sigma = (float)h;
    orc_icp_run(dst.data(), nrm.data(), n, src.data(), nullptr, n, nullptr, &p, nullptr, &r);
    const double e6 = frob(icp6.getTransform().m, r.T);
    std::printf("weight evaluators: iters gpu=%zu oracle=%zu |T_gpu-T_oracle|_F=%.3e\n", icp6.getNumberOfPerformedIterations(), r.iterations, e6);
    if (!(e6 <= 1e-5)) ++failures;

    // ... and the same two evaluators as FUNCTORS of the reference's call shape, evaluator(indexInFirst, indexInSecond, value)
    // (transform_estimation.hpp:303, :332): evaluated on the host through cilhip_set_pair_weight_callback, the loop step by step
    {
      SimpleCombinedMetricRigidICP3f icp7(dst_v, nrm_v, src_v);
      icp7.setPointToPointMetricWeight(0.2f).setPointToPlaneMetricWeight(1.0f);
      const float hf = (float)h;
      size_t calls = 0;
      icp7.pointToPlaneCorrespondenceWeightEvaluator().setFunctor([hf, &calls](size_t, size_t, float v) { ++calls; return std::exp((-0.5f / (hf * hf)) * v); });
      icp7.pointToPointCorrespondenceWeightEvaluator().setFunctor([](size_t, size_t, float v) { return v; });
      icp7.correspondenceSearchEngine().setMaxDistance(max_sq);
      icp7.setConvergenceTolerance(0.0f).setMaxNumberOfIterations(8).estimate();
      const double e7 = frob(icp7.getTransform().m, r.T), e76 = frob(icp7.getTransform().m, icp6.getTransform().m);
      std::printf("functor evaluators: iters gpu=%zu calls=%zu |T_gpu-T_oracle|_F=%.3e |T_functor-T_enum|_F=%.3e\n", icp7.getNumberOfPerformedIterations(), calls, e7, e76);
      if (!(e7 <= 1e-5) || !(e76 <= 1e-6) || calls == 0 || icp7.getNumberOfPerformedIterations() != icp6.getNumberOfPerformedIterations()) ++failures;
      // a functor that throws: caught at the C boundary, rethrown by estimate()
      icp7.pointToPointCorrespondenceWeightEvaluator().setFunctor([](size_t, size_t, float) -> float { throw std::runtime_error("evaluator failed"); });
      bool rethrown = false;
      try { icp7.estimate(); } catch (const std::runtime_error& ex) { rethrown = std::string(ex.what()) == "evaluator failed"; }
      std::printf("throwing functor: rethrown by estimate() = %d\n", (int)rethrown);
      if (!rethrown) ++failures;
    }

    // feature adaptors of the engine (common_transformable_feature_adaptors.hpp: point+normal :60-161, point+colour :164-252,
    // point+normal+colour :255-343) through the mirror: correspondence lists under tf against the oracle's exhaustive searches
    {
      const size_t m = 8000;
      const double hm = std::pow((double)m, -1.0 / 3.0);
      std::vector<float> d2p(dst.begin(), dst.begin() + 3 * m), d2n(nrm.begin(), nrm.begin() + 3 * m), s2p(3 * m), s2n(3 * m), dcol(3 * m), scol(3 * m);
      float Ti[16], Tf[16];
      std::memcpy(Tf, tf.m, sizeof(Tf));
      // inverse of the rigid tf (col-major): R^T, -R^T t
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Ti[c * 4 + r] = Tf[r * 4 + c];
      for (int r = 0; r < 3; ++r) Ti[12 + r] = -(Ti[r] * Tf[12] + Ti[4 + r] * Tf[13] + Ti[8 + r] * Tf[14]);
      Ti[3] = Ti[7] = Ti[11] = 0.f; Ti[15] = 1.f;
      std::vector<float> jp(3 * m);
      for (size_t i = 0; i < 3 * m; ++i) {
        jp[i] = d2p[i] + (2 * u01(51, i) - 1) * (float)(0.05 * hm);
        dcol[i] = u01(52, i);
        scol[i] = std::fmin(1.0f, std::fmax(0.0f, dcol[i] + (2 * u01(53, i) - 1) * 0.1f));
      }
      orc_transform_points(Ti, jp.data(), m, s2p.data());      // (ONE call: the oracle opens an OpenMP region per call)
      orc_transform_normals(Ti, d2n.data(), m, s2n.data());
      const ConstPointsView d2v(d2p), d2nv(d2n), s2v(s2p), s2nv(s2n), dcv(dcol), scv(scol);
      const float wn = (float)(0.6 * hm), wc = (float)(0.8 * hm), r2 = (float)((2.5 * hm) * (2.5 * hm));
      std::vector<float> dst6(6 * m), src6(6 * m), q6(6 * m), dst9(9 * m), src9(9 * m), q9(9 * m);
      std::vector<int64_t> di(2 * m + 1), si(2 * m + 1); std::vector<float> dv(2 * m + 1);
      auto compare = [&](const CorrespondenceSearchHIP::SearchResult& got, size_t nc, const char* what) {
        size_t bad = (nc != got.size()) || nc < m / 2;
        for (size_t k = 0; k < nc && k < got.size(); ++k)
          bad += (got[k].indexInFirst != (size_t)di[k]) || (got[k].indexInSecond != (size_t)si[k]) || (got[k].value != dv[k]);
        std::printf("feature adaptor %s: %zu correspondences, %zu mismatches vs oracle\n", what, got.size(), bad);
        if (bad) ++failures;
      };
      SimpleCombinedMetricRigidICP3f fi(d2v, d2nv, s2v);
      auto& eng = fi.correspondenceSearchEngine();
      eng.setMaxDistance(r2);
      // point + normal, BOTH directions
      eng.setSearchDirection(CorrespondenceSearchDirection::BOTH).setPointNormalFeatureAdaptors(s2nv, wn);
      orc_point_normal_features(d2p.data(), d2n.data(), m, wn, dst6.data());
      orc_point_normal_features(s2p.data(), s2n.data(), m, wn, src6.data());
      orc_transform_features6_mode(Tf, src6.data(), m, 0, q6.data());
      compare(eng.findCorrespondences(tf).getCorrespondences(),
              orc_find_correspondences_feat6_dir(dst6.data(), m, q6.data(), m, r2, 2, 0, di.data(), si.data(), dv.data(), 0), "point+normal (BOTH)");
      // point + colour
      eng.setSearchDirection(CorrespondenceSearchDirection::SECOND_TO_FIRST).setPointColorFeatureAdaptors(dcv, scv, wc);
      orc_point_normal_features(d2p.data(), dcol.data(), m, wc, dst6.data());
      orc_point_normal_features(s2p.data(), scol.data(), m, wc, src6.data());
      orc_transform_features6_mode(Tf, src6.data(), m, 2, q6.data());
      compare(eng.findCorrespondences(tf).getCorrespondences(),
              orc_find_correspondences_feat6_dir(dst6.data(), m, q6.data(), m, r2, 0, 0, di.data(), si.data(), dv.data(), 0), "point+colour");
      // point + normal + colour (9-D), FIRST_TO_SECOND
      eng.setSearchDirection(CorrespondenceSearchDirection::FIRST_TO_SECOND).setPointNormalColorFeatureAdaptors(s2nv, dcv, scv, wn, wc);
      orc_point_normal_color_features(d2p.data(), d2n.data(), dcol.data(), m, wn, wc, dst9.data());
      orc_point_normal_color_features(s2p.data(), s2n.data(), scol.data(), m, wn, wc, src9.data());
      orc_transform_features9_mode(Tf, src9.data(), m, 0, q9.data());
      compare(eng.findCorrespondences(tf).getCorrespondences(),
              orc_find_correspondences_feat9_dir(dst9.data(), m, q9.data(), m, r2, 1, 0, di.data(), si.data(), dv.data(), 0), "point+normal+colour (FIRST_TO_SECOND)");
    }

    // affine instances (icp_common_instances.hpp:255, :266) through the mirrors
    SimpleCombinedMetricAffineICP3f icp4(dst_v, nrm_v, src_v);
    icp4.setPointToPointMetricWeight(0.1f).setPointToPlaneMetricWeight(1.0f);
    icp4.correspondenceSearchEngine().setMaxDistance(max_sq);
    icp4.setConvergenceTolerance(1e-5f).setMaxNumberOfIterations(10).estimate();
    std::memset(&p, 0, sizeof(p));
    p.metric = 1; p.w_p2p = 0.1f; p.w_p2pl = 1; p.max_iter = 10; p.conv_tol = 1e-5f; p.max_opt_iter = 1; p.opt_conv_tol = 1e-5f;
    p.max_sq_dist = max_sq; p.mode = ORC_MODE_MIXED; p.transform_mode = 1;
    orc_icp_run(dst.data(), nrm.data(), n, src.data(), nullptr, n, nullptr, &p, nullptr, &r);
    const double e4 = frob(icp4.getTransform().m, r.T);
    std::printf("affine combined: iters gpu=%zu oracle=%zu |T_gpu-T_oracle|_F=%.3e\n", icp4.getNumberOfPerformedIterations(), r.iterations, e4);
    if (!(e4 <= 1e-4) || icp4.getNumberOfPerformedIterations() != r.iterations) ++failures;
    SimplePointToPointMetricAffineICP3f icp5(dst_v, src_v);
    icp5.correspondenceSearchEngine().setMaxDistance(max_sq);
    icp5.setMaxNumberOfIterations(10).setConvergenceTolerance(1e-5f).estimate();
    p.metric = 0;
    orc_icp_run(dst.data(), nullptr, n, src.data(), nullptr, n, nullptr, &p, nullptr, &r);
    const double e5 = frob(icp5.getTransform().m, r.T);
    std::printf("affine point-to-point: iters gpu=%zu oracle=%zu |T_gpu-T_oracle|_F=%.3e\n", icp5.getNumberOfPerformedIterations(), r.iterations, e5);
    if (!(e5 <= 1e-4) || icp5.getNumberOfPerformedIterations() != r.iterations) ++failures;
    {
      // CorrespondenceSearchCombinedMetricCombiner over two engines that own their contexts (correspondence_search_combined_metric_combiner.hpp):
      // the same options on both = the single-engine class; a tighter point engine changes the estimate
      CorrespondenceSearchHIPOwned e_pt(dst_v, nrm_v, src_v), e_pl(dst_v, nrm_v, src_v);
      e_pt.setMaxDistance(max_sq); e_pl.setMaxDistance(max_sq);
      CorrespondenceSearchCombinedMetricCombiner<CorrespondenceSearchHIPOwned, CorrespondenceSearchHIPOwned> comb(e_pt, e_pl);
      CombinedMetricRigidICP3f<decltype(comb)> icp7(comb);
      icp7.setPointToPointMetricWeight(0.2f).setPointToPlaneMetricWeight(1.0f).setMaxNumberOfIterations(5).setConvergenceTolerance(0.0f).estimate();
      SimpleCombinedMetricRigidICP3f icp8(dst_v, nrm_v, src_v);
      icp8.setPointToPointMetricWeight(0.2f).setPointToPlaneMetricWeight(1.0f);
      icp8.correspondenceSearchEngine().setMaxDistance(max_sq);
      icp8.setMaxNumberOfIterations(5).setConvergenceTolerance(0.0f).estimate();
      const double e7 = frob(icp7.getTransform().m, icp8.getTransform().m);
      comb.findCorrespondences(icp7.getTransform());
      const size_t n_pt = comb.getPointToPointCorrespondences().size(), n_pl = comb.getPointToPlaneCorrespondences().size();
      e_pt.setMaxDistance(max_sq * 0.0004f);      // (0.04 h: inside the +-0.05 h noise of the recipe)
      icp7.estimate();
      comb.findCorrespondences(icp7.getTransform());
      const size_t n_pt2 = comb.getPointToPointCorrespondences().size();
      std::printf("combiner: same engines vs single-engine class |dT|_F=%.3e, sets %zu / %zu, tighter point engine %zu, iterations %zu\n", e7, n_pt, n_pl, n_pt2,
                  icp7.getNumberOfPerformedIterations());
      if (!(e7 <= 2e-6) || n_pt != n_pl || !(n_pt2 < n_pt) || icp7.getNumberOfPerformedIterations() != 5) ++failures;
    }
  } catch (const std::runtime_error& e) {
    if (expect_no_device) { std::printf("OK (failed loudly): %s\n", e.what()); return 0; }
    std::printf("FAIL: %s\n", e.what());
    return 1;
  }
  std::printf(failures ? "FAILED (%d)\n" : "ALL OK\n", failures);
  return failures ? 1 : 0;
}
