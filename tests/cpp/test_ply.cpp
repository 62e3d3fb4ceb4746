// tests/cpp/test_ply.cpp -- PLY round trip through include/cilantro_hip/point_cloud.hpp (host only, no GPU):
//   test_ply write <out.ply> <binary 0|1>   : writes a seeded cloud with normals and colours
//   test_ply copy  <in.ply> <out.ply> <binary 0|1> : reads a PLY and writes it back
#include <cilantro_hip/point_cloud.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>

int main(int argc, char** argv) {
  using cilantro_hip::PointCloud3f;
  try {
    if (argc >= 4 && !std::strcmp(argv[1], "write")) {
      PointCloud3f c;
      const size_t n = 1000;
      c.points.resize(3 * n); c.normals.resize(3 * n); c.colors.resize(3 * n);
      for (size_t i = 0; i < 3 * n; ++i) {
        c.points[i] = 0.001f * (float)((i * 7919u) % 10007u) - 5.0f;
        c.normals[i] = (i % 3 == 2) ? 1.0f : 0.0f;
        c.colors[i] = (float)(i % 256) / 255.0f;
      }
      c.toPLYFile(argv[2], std::atoi(argv[3]) != 0);
      return 0;
    }
    if (argc >= 5 && !std::strcmp(argv[1], "copy")) {
      PointCloud3f c(argv[2]);
      std::printf("%zu points normals=%d colors=%d\n", c.size(), (int)c.hasNormals(), (int)c.hasColors());
      c.toPLYFile(argv[3], std::atoi(argv[4]) != 0);
      return 0;
    }
  } catch (const std::exception& e) {
    std::printf("FAIL: %s\n", e.what());
    return 1;
  }
  std::printf("usage: test_ply write <out> <binary> | copy <in> <out> <binary>\n");
  return 2;
}
