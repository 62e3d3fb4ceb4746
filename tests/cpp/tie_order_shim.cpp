// Test shim: csrc/tie_order.hpp (host-only) behind three C functions, so that the CPU suite can check the tree restatement
// against the reference's nanoflann (oracle/_ref) without a GPU.  Built by tests/cpp/build.sh into tests/cpp/bin/libtie_order_shim.so.
#include "../../cilantro_amd/csrc/tie_order.hpp"

struct Shim { std::vector<float> xyz; cilhip::TieOrderTree tree; };

extern "C" {
void* tie_shim_build(const float* xyz, uint32_t n) {
  Shim* s = new Shim();
  s->xyz.assign(xyz, xyz + 3 * (size_t)n);
  s->tree.build(s->xyz.data(), n);
  return s;
}
void tie_shim_free(void* h) { delete static_cast<Shim*>(h); }
// per query k: cand[k*stride .. +count[k]) -> out[k]
void tie_shim_first_met(void* h, const float* q, const uint32_t* cand, const int* count, int stride, uint32_t nq, uint32_t* out) {
  const Shim* s = static_cast<const Shim*>(h);
  for (uint32_t k = 0; k < nq; ++k) out[k] = s->tree.first_met(q + 3 * (size_t)k, cand + (size_t)k * stride, count[k]);
}
}
