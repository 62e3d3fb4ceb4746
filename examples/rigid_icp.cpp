// examples/rigid_icp.cpp -- the flow of cilantro's examples/rigid_icp.cpp (its lines 25-65 and 103-135) on the GPU engine:
// read a PLY with normals, make a distorted + moved copy, register it back with point-to-plane ICP, print the result.
//
//   g++ -O2 -std=c++17 -Iinclude examples/rigid_icp.cpp -o rigid_icp -Lcilantro_amd/lib -lcilantro_hip \
//       -Wl,-rpath,$PWD/cilantro_amd/lib -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
//   ./rigid_icp cloud.ply
//
// Differences from the reference example are only the ones the missing pieces force: no visualizer, no voxel
// down-sampling (not on the ICP path), a seeded uniform jitter instead of Eigen::Random.
#include <cilantro_hip/icp.hpp>
#include <cilantro_hip/point_cloud.hpp>

#include <chrono>
#include <cmath>
#include <cstdio>

using namespace cilantro_hip;

static float jitter(uint64_t& s) {   // uniform in [-1, 1)
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (2.0f / 16777216.0f) - 1.0f;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("Please provide path to PLY file.\n"); return 0; }
  PointCloud3f dst(argv[1]), src;
  if (!dst.hasNormals()) { std::printf("Input cloud is empty or does not have normals!\n"); return 0; }

  // distorted and transformed version of dst (reference example :29-62)
  src = dst;
  uint64_t seed = 44;
  for (float& v : src.points) v += 0.01f * jitter(seed);
  const float a = -0.1f, b = 0.1f, c = -0.1f;   // Rz(a) * Ry(b) * Rx(c)
  const float R[9] = {std::cos(a) * std::cos(b), std::cos(a) * std::sin(b) * std::sin(c) - std::sin(a) * std::cos(c), std::cos(a) * std::sin(b) * std::cos(c) + std::sin(a) * std::sin(c),
                      std::sin(a) * std::cos(b), std::sin(a) * std::sin(b) * std::sin(c) + std::cos(a) * std::cos(c), std::sin(a) * std::sin(b) * std::cos(c) - std::cos(a) * std::sin(c),
                      -std::sin(b), std::cos(b) * std::sin(c), std::cos(b) * std::cos(c)};
  const float t[3] = {-0.20f, -0.05f, 0.10f};
  for (size_t i = 0; i < src.size(); ++i) {
    const float x = src.points[3 * i], y = src.points[3 * i + 1], z = src.points[3 * i + 2];
    for (int r = 0; r < 3; ++r) src.points[3 * i + r] = R[3 * r] * x + R[3 * r + 1] * y + R[3 * r + 2] * z + t[r];
  }

  const auto t0 = std::chrono::steady_clock::now();
  const ConstPointsView dst_p(dst.points), dst_n(dst.normals), src_p(src.points);
  SimpleCombinedMetricRigidICP3f icp(dst_p, dst_n, src_p);                      // reference example :117
  icp.setMaxNumberOfOptimizationStepIterations(1).setPointToPointMetricWeight(0.0f).setPointToPlaneMetricWeight(1.0f);
  icp.correspondenceSearchEngine().setMaxDistance(0.1f * 0.1f);
  icp.setConvergenceTolerance(1e-4f).setMaxNumberOfIterations(30);
  const RigidTransform3f tf_est = icp.estimate().getTransform();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

  std::printf("Registration time: %.2fms (upload + index + loop)\n", ms);
  std::printf("Iterations performed: %zu\nHas converged: %d\n", icp.getNumberOfPerformedIterations(), (int)icp.hasConverged());
  std::printf("ESTIMATED transformation (should be the inverse of the applied motion):\n");
  for (int r = 0; r < 4; ++r) std::printf("  % .6f % .6f % .6f % .6f\n", tf_est.m[r], tf_est.m[4 + r], tf_est.m[8 + r], tf_est.m[12 + r]);
  // R_est * R should be the identity
  double err = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += (double)tf_est.m[4 * k + i] * R[3 * k + j];
      err += (s - (i == j)) * (s - (i == j));
    }
  std::printf("|R_est * R_applied - I|_F = %.3e\n", std::sqrt(err));
  const std::vector<float> residuals = icp.getResiduals();
  std::printf("Residuals: %zu values\n", residuals.size());
  return 0;
}
