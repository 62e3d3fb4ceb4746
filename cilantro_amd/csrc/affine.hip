// affine.hip -- the affine classes' moments over STORED matches in one streaming pass (k_acc_affine): what an iteration of their
// device-resident loop runs after a search-only kernel (cold iterations; the warm-started iterations accumulate inside k_warm<IM_AFFC /
// IM_AFFP>).  Terms and layout: affine_device.hpp; reduction + solve: epilogue.hip (launch_reduce_and_solve_affine).
#include "affine_device.hpp"

namespace cilhip {

constexpr int AFFACC_THREADS = 256;
constexpr int AFFACC_WAVES = AFFACC_THREADS / 64;

// One row of AFF_ROW sums per block over its contiguous share of the (sorted) source: per round a wave reads 64 source records and
// their stored matches, gathers the matched target points (and normals), leaves the 12-float records in LDS and feeds the matrix cores.
template <bool NRM>
__global__ __launch_bounds__(AFFACC_THREADS) void k_acc_affine(IterArgs a) {
  const IcpState* __restrict__ st = a.state;
  if (st->done) return;
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) T[i] = st->T[i];
  const bool raw_moments = a.no_centering != 0;
  const float smt[3] = {raw_moments ? 0.0f : st->smt[0], raw_moments ? 0.0f : st->smt[1], raw_moments ? 0.0f : st->smt[2]};
  const float dmn[3] = {raw_moments ? 0.0f : a.dst_mean[0], raw_moments ? 0.0f : a.dst_mean[1], raw_moments ? 0.0f : a.dst_mean[2]};
  __shared__ __attribute__((aligned(16))) unsigned char raw[AFFACC_WAVES * FUSED_WAVE_BYTES];
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  float* const zb = reinterpret_cast<float*>(raw) + wave * (FUSED_WAVE_BYTES / 4);
  const AffLane afl = aff_lane(lane);
  aff_double4 acc = {0.0, 0.0, 0.0, 0.0};
  const GridDev& g = a.grid;
  // rounds of 256 queries, dealt out evenly over the blocks (contiguous shares)
  const uint32_t rounds = (a.ns + AFFACC_THREADS - 1) / AFFACC_THREADS, nb = gridDim.x;
  const uint32_t rbase = rounds / nb, rrem = rounds % nb;
  const uint32_t r0 = blockIdx.x * rbase + min(blockIdx.x, rrem), r1 = r0 + rbase + (blockIdx.x < rrem ? 1u : 0u);
  const uint32_t last = a.ns ? a.ns - 1u : 0u;
  for (uint32_t r = r0; r < r1; ++r) {
    const uint32_t i = r * AFFACC_THREADS + threadIdx.x;
    const uint32_t ic = min(i, last);
    const float4 s4 = a.src[ic];
    const uint32_t pos = a.nn_pos[ic];
    const bool has = i < a.ns && pos != NONE_U32;
    const uint32_t pc = has ? pos : 0u;
    const float4 p = g.pts[pc];
    float4 nv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (NRM) nv = g.nrm[pc];
    float qx, qy, qz;
    transform_point(T, s4.x, s4.y, s4.z, qx, qy, qz);
    aff_record<NRM>(has, qx, qy, qz, p, nv, dmn, smt, zb + lane * AFF_REC);
    __builtin_amdgcn_wave_barrier();
    aff_mfma_round(zb, lane, afl, acc);
    __builtin_amdgcn_wave_barrier();
  }
  aff_write_row<AFFACC_WAVES>(raw, wave, lane, acc, a.partials + (size_t)blockIdx.x * AFF_ROW);
}

int affine_acc_blocks(uint32_t ns) {
  long nb = ((long)ns + 8 * AFFACC_THREADS - 1) / (8 * AFFACC_THREADS);      // at least eight rounds per block
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  return (int)nb;
}

void launch_acc_affine(const IterArgs& a, int metric, int nblocks, hipStream_t s) {
  if (metric == IM_AFFC) hipLaunchKernelGGL((k_acc_affine<true>), dim3(nblocks), dim3(AFFACC_THREADS), 0, s, a);
  else hipLaunchKernelGGL((k_acc_affine<false>), dim3(nblocks), dim3(AFFACC_THREADS), 0, s, a);
}

}  // namespace cilhip
