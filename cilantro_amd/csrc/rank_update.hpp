// rank_update.hpp -- the matrix-core accumulation of the rigid classes' normal-equation sums as a per-wave object (the streaming
// kernels outside warm.hip / kernels.hip: bidir.hip's k_reverse_warm).  Same terms (fused_z), same tile layouts and the same slot map
// (FusedZ<ACC>::slot_terms) as k_warm / k_search_tiled: Z += z z^T per correspondence on v_mfma_f64_16x16x4_f64, two groups of
// correspondences per tile when the term vector has at most eight components (DUAL).
#pragma once
#include "search_device.hpp"

namespace cilhip {

template <int ACC>
struct WaveRank {
  typedef double double4_t __attribute__((ext_vector_type(4)));
  static constexpr int NC = FusedZ<ACC>::NC;
  static constexpr bool DUAL = NC <= 8;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};

  // one round: the lane's correspondence (has: it has one) -> LDS -> the wave's tile.  zb: the wave's FUSED_WAVE_BYTES of scratch.
  __device__ __forceinline__ void update(float* zb, int lane, bool has, float qx, float qy, float qz, const float4 p, const float4 nv, const float* dmean, const float* smt) {
    float z[16];
    // (branch-free: the terms of every lane are formed, a lane without a correspondence then takes zeros by a select -- under a divergent
    //  branch around fused_z the gfx950 back end was seen to drop the PLANE vector's n_z component on one path: profiles/r06 notebook entry)
    fused_z<ACC>(true, qx, qy, qz, p, nv, dmean, smt, z);
#pragma unroll
    for (int k = 0; k < 16; ++k) z[k] = has ? z[k] : 0.0f;
    if (DUAL) {
      float4* w4 = reinterpret_cast<float4*>(zb + lane * 8 + (lane >= 32 ? 16 : 0));
      w4[0] = make_float4(z[0], z[1], z[2], z[3]);
      w4[1] = make_float4(z[4], z[5], z[6], z[7]);
    } else {
      float2* w2 = reinterpret_cast<float2*>(zb + lane * NC);
#pragma unroll
      for (int c = 0; c < NC / 2; ++c) w2[c] = make_float2(z[2 * c], z[2 * c + 1]);
    }
    __builtin_amdgcn_wave_barrier();
    if (DUAL) {
      const int comp = lane & 7, hf = (lane >> 3) & 1, k4 = lane >> 4;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int qi = hf * 32 + 4 * jj + k4;
        const double x = (double)zb[qi * 8 + hf * 16 + comp];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
      }
    } else {
      const int comp = lane & 15, k4 = lane >> 4;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const float f = zb[(4 * jj + k4) * NC + (comp < NC ? comp : 0)];
        const double x = comp < NC ? (double)f : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  // the block's row of SUMS_MAX sums: the waves' tiles added in wave order (fixed order: bitwise reproducible); raw: NW x FUSED_WAVE_BYTES
  // of LDS; every thread of the block calls this
  template <int NW>
  __device__ __forceinline__ void write_row(unsigned char* raw, int wave, int lane, double* row) {
    double* const db = reinterpret_cast<double*>(raw + wave * FUSED_WAVE_BYTES);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) db[r * 64 + lane] = acc[r];
    __syncthreads();
    if (threadIdx.x < SUMS_MAX) {
      int i1, j1, i2, j2;
      const bool used = FusedZ<ACC>::slot_terms((int)threadIdx.x, i1, j1, i2, j2);
      double v1 = 0.0, v2 = 0.0;
      if (used) {
        const int e1 = (i1 >> 2) * 64 + 16 * (i1 & 3) + j1, e1b = ((i1 + 8) >> 2) * 64 + 16 * ((i1 + 8) & 3) + j1 + 8;
        const int e2 = i2 >= 0 ? (i2 >> 2) * 64 + 16 * (i2 & 3) + j2 : 0, e2b = i2 >= 0 ? ((i2 + 8) >> 2) * 64 + 16 * ((i2 + 8) & 3) + j2 + 8 : 0;
        for (int w = 0; w < NW; ++w) {
          const double* dw = reinterpret_cast<const double*>(raw + w * FUSED_WAVE_BYTES);
          v1 += dw[e1];
          if (DUAL) v1 += dw[e1b];
          if (i2 >= 0) { v2 += dw[e2]; if (DUAL) v2 += dw[e2b]; }
        }
      }
      row[threadIdx.x] = v1 - v2;
    }
  }
};

}  // namespace cilhip
