// feat_warm.hip -- the feature adaptors' forward search WARM-STARTED from the previous iteration's matches (k_feat_warm; the loops of
// PointNormalFeaturesAdaptor / PointColorFeaturesAdaptor / PointNormalColorFeaturesAdaptor: common_transformable_feature_adaptors.hpp:60-343).
//
// The feature distance is d6 = |q - p|^2 + |f_q - f_p|^2 (+ the colour part) >= d3 = |q - p|^2.  For the target point p a query matched
// last time, any OTHER target point p' has d6(q, p') >= d3(q, p') >= (|p - p'| - |q - p|)^2 >= (nnd(p) - |q - p|)^2 with nnd(p) the distance
// from p to its nearest other target point, and |q - p| <= sqrt(d6(q, p)).  So 2 sqrt(d6(q, p)) < nnd(p) makes p the one nearest FEATURE: the
// margin test of the plain warm-started iteration's first round (k_warm<., 1>: DESIGN 6.2) with the feature distance in the place of the
// squared distance, against the same table (k_self_nn's safe2: a lower bound on nnd^2).  In pinned arithmetic: settled iff
// e = d6_pinned(q, p) < max_sq  and  4 e (1 + 2e-5) < safe2[p]  (strict: a second feature at the same distance fails it).  Every other query
// is listed per wave (ballot order) and searched in full by the wave's lanes, densely packed: nn_search_group<1, true> + tie_settle<true>,
// what k_search_feat6 runs -- the exact argmin with the reference's tie order either way.  nn_pos (and nn_d2 when somebody reads it) in,
// nn_pos out.  ACC != IM_NONE: the first Gauss-Newton step's sums in the same pass, on the matrix cores (rank_update.hpp: the terms of
// k_warm / the tiles; the three-cloud metric without per-pair weights) -- for the settled queries as they stream by, for the listed ones
// after their search; ACC == IM_NONE: search only, the sums are the streaming pass's (the symmetric metric, weight evaluators).
#include "search_device.hpp"
#include "rank_update.hpp"

namespace cilhip {

namespace {
constexpr int FW_ROUNDS = 16;                  // rounds between two searches of a wave's list
constexpr int FW_WCAP = FW_ROUNDS * 64;
constexpr int FW_WAVES = 4;

template <int ACC>
__global__ __launch_bounds__(256) void k_feat_warm(IterArgs a) {
  const IcpState* __restrict__ st = a.state;
  if (st->done) return;
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = st->T[k];
  const float smt[3] = {st->smt[0], st->smt[1], st->smt[2]};
  __shared__ uint32_t list[FW_WAVES * FW_WCAP];
  __shared__ __attribute__((aligned(16))) unsigned char raw[ACC != IM_NONE ? FW_WAVES * FUSED_WAVE_BYTES : 16];
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  uint32_t* const wl = list + wave * FW_WCAP;
  float* const zb = reinterpret_cast<float*>(raw) + (ACC != IM_NONE ? wave * (FUSED_WAVE_BYTES / 4) : 0);
  WaveRank<ACC == IM_NONE ? IM_KABSCH : ACC> rank;
  constexpr bool NRM = FusedZ<ACC>::needs_normal;
  const GridDev& g = a.grid;
  const uint32_t rounds = (a.ns + 255u) / 256u, last = a.ns - 1u;
  uint32_t wcnt = 0;      // (wave-uniform)
  uint32_t nlisted = 0;
  auto flush = [&]() {
    __builtin_amdgcn_wave_barrier();
    for (uint32_t k0 = 0; k0 < wcnt; k0 += 64u) {
      const bool active = k0 + (uint32_t)lane < wcnt;
      const uint32_t i = active ? wl[min(k0 + (uint32_t)lane, (uint32_t)(FW_WCAP - 1))] : 0u;      // (a slot beyond the list holds nothing meaningful)
      const float4 s4 = a.src[i];
      float qx, qy, qz;
      transform_point(T, s4.x, s4.y, s4.z, qx, qy, qz);
      uint32_t found = NONE_U32;
      if (active) {
        Feat6 f;
        query_features(a, T, i, true, f);
        NN best;
        nn_search_group<1, true>(g, qx, qy, qz, a.max_sq, 0, 1, best, &f);
        // (option "tie_rule": exactly equal feature distances take the pick of the reference's DIM = 6 / 9 tree)
        if (a.tie.mode != 0 && best.tie != 0u && best.pos != NONE_U32)
          best.pos = tie_settle<true>(g, a.tie, qx, qy, qz, best.pos, __uint_as_float((uint32_t)(best.key >> 32)), &f);
        a.nn_pos[i] = best.pos;
        if (a.nn_d2) a.nn_d2[i] = __uint_as_float((uint32_t)(best.key >> 32));
        found = best.pos;
      }
      if (ACC != IM_NONE) {      // (every lane of the wave: the rank update is a wave-wide operation)
        const bool hasf = found != NONE_U32;
        const float4 pf = g.pts[hasf ? found : 0u];
        float4 nvf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (NRM) nvf = g.nrm[hasf ? found : 0u];
        rank.update(zb, lane, hasf, qx, qy, qz, pf, nvf, a.dst_mean, smt);
      }
    }
    wcnt = 0;
    __builtin_amdgcn_wave_barrier();
  };
  uint32_t since = 0;
  for (uint32_t r = blockIdx.x; r < rounds; r += gridDim.x) {
    if (since == (uint32_t)FW_ROUNDS) { flush(); since = 0; }
    ++since;
    const uint32_t i = r * 256u + threadIdx.x;
    const bool valid = i < a.ns;
    const uint32_t ic = min(i, last);
    const float4 s4 = a.src[ic];
    const uint32_t w = a.nn_pos[ic];
    const bool has = valid && w != NONE_U32;
    const uint32_t wc = has ? w : 0u;
    const float4 p = g.pts[wc];
    const float sf = a.safe2[wc];
    float qx, qy, qz;
    transform_point(T, s4.x, s4.y, s4.z, qx, qy, qz);
    Feat6 f;
    query_features(a, T, ic, true, f);
    const float4 fn = f.nrm[wc];
    const float4 fc = f.att2 != nullptr ? f.att2[wc] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float e = d6_pinned(qx, qy, qz, f, p, fn, fc);
    const bool settled = has && e < a.max_sq && 4.0f * e * 1.00002f < sf;
    if (settled && a.nn_d2) a.nn_d2[i] = e;      // (the value the search compares: the same expression on the same operands)
    const bool todo = valid && !settled;
    const unsigned long long um = __ballot(todo);
    if (todo) wl[wcnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(um >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)um, 0u))] = i;
    wcnt += (uint32_t)__popcll(um);
    nlisted += (uint32_t)__popcll(um);
    if (ACC != IM_NONE) {
      float4 nvt = make_float4(0.f, 0.f, 0.f, 0.f);
      if (NRM) nvt = g.nrm[wc];      // (the target's NORMAL: the feature vector beside it may be a colour)
      rank.update(zb, lane, settled, qx, qy, qz, p, nvt, a.dst_mean, smt);
    }
  }
  flush();
  if (ACC != IM_NONE) rank.template write_row<FW_WAVES>(raw, wave, lane, a.partials + (size_t)blockIdx.x * SUMS_MAX);
  if (a.unproven_cnt && lane == 0 && nlisted != 0u) atomicAdd(a.unproven_cnt + 64u + ((blockIdx.x * FW_WAVES + (uint32_t)wave) & 63u), nlisted);      // listed queries: is the form paying?
}
}  // namespace

int feat_warm_blocks(uint32_t ns) {      // at least eight rounds of 256 queries per block
  long nb = ((long)ns + 8 * 256 - 1) / (8 * 256);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  return (int)nb;
}
void launch_feat_warm(const IterArgs& a, int acc_metric, hipStream_t s) {
  if (a.ns == 0) return;
  const dim3 gb(feat_warm_blocks(a.ns)), tb(256);
  switch (acc_metric) {
    case IM_KABSCH: hipLaunchKernelGGL((k_feat_warm<IM_KABSCH>), gb, tb, 0, s, a); break;
    case IM_PLANE: hipLaunchKernelGGL((k_feat_warm<IM_PLANE>), gb, tb, 0, s, a); break;
    case IM_POINT: hipLaunchKernelGGL((k_feat_warm<IM_POINT>), gb, tb, 0, s, a); break;
    case IM_BOTH: hipLaunchKernelGGL((k_feat_warm<IM_BOTH>), gb, tb, 0, s, a); break;
    default: hipLaunchKernelGGL((k_feat_warm<IM_NONE>), gb, tb, 0, s, a); break;
  }
}

}  // namespace cilhip
