// affine_device.hpp -- the moments of the affine estimators' 12-unknown normal equations in ONE pass on the matrix cores (device code
// shared by the warm-started iteration k_warm<IM_AFFC / IM_AFFP> and the streaming pass k_acc_affine).
//
// transform_estimation.hpp:369-476 (combined metric, affine) adds per plane correspondence eq_vec eq_vec^T and (n.d) eq_vec with
// eq_vec = (n_0 s, n_1 s, n_2 s, n) (:457-464), per point correspondence the blocks of (s, 1)(s, 1)^T and (s, 1) d_j (:436-442);
// s = q - src_mean', d = p - dst_mean and n.d formed in f32 as the reference forms them.  With s' = (s, 1) every entry of those sums is
// a product of one of the 13 numbers
//     a = (n_j n_k [j <= k: 6], (n.d) n_j [3], 1, d_j [3])        and one of the 10 numbers        b = (s'_a s'_b [a <= b])
// -- eq_vec eq_vec^T = (n n^T) (x) (s' s'^T) -- so ONE v_mfma_f64_16x16x4_f64 per four correspondences accumulates all 94 sums:
// D += a b^T, 13 x 10 of the 16 x 16 tile.  Per correspondence a wave leaves a 12-float record {n, s, 1, n.d, d, 0} in LDS (zeros for a
// query without a match); a lane of the rank update forms its entry of a and of b as the f64 product of two record entries (exact: two
// f32 factors), the matrix core rounds once per product and sum in f64 -- what accumulate_pair<IM_AFF0 / 1 / 2> computes in three
// passes with per-lane f64 accumulators (those stay for loops with per-pair weights).
#pragma once
#include "search_device.hpp"

namespace cilhip {

constexpr int AFF_REC = 12;      // floats per correspondence record: n [0..2], s [3..5], 1 [6], n.d [7], d [8..10], 0 [11]
static_assert(64 * AFF_REC * 4 <= FUSED_WAVE_BYTES, "a wave's 64 records fit the rank update's scratch");
typedef double aff_double4 __attribute__((ext_vector_type(4)));

// pair index of (a, b), a <= b < 4, row-major upper triangle
__host__ __device__ constexpr int aff_pair(int a, int b) { return a * 4 - a * (a - 1) / 2 + (b - a); }
// row-partial slots (internal.hpp AFF_ROW): M(jk, ab) = sum n_j n_k s'_a s'_b at jk * 10 + ab (jk over (0,0) (0,1) (0,2) (1,1) (1,2) (2,2));
// R(j, a) = sum (n.d) n_j s'_a at 60 + 4 j + a; S(ab) = sum s'_a s'_b at 72 + ab (ab = 9: the number of correspondences);
// Q(j, a) = sum d_j s'_a at 82 + 4 j + a
__host__ __device__ constexpr int aff_jk(int j, int k) { return j * 3 - j * (j - 1) / 2 + (k - j); }

// which record entries a lane multiplies: its entry of a (rows of the tile) and of b (columns)
struct AffLane { int a1, a2, b1, b2; };
__device__ __forceinline__ AffLane aff_lane(int lane) {
  const int m = lane & 15;
  AffLane L;
  // a_m
  if (m < 6) { const int j = m < 3 ? 0 : (m < 5 ? 1 : 2), k = m < 3 ? m : (m < 5 ? m - 2 : 2); L.a1 = j; L.a2 = k; }
  else if (m < 9) { L.a1 = 7; L.a2 = m - 6; }
  else if (m == 9) { L.a1 = 6; L.a2 = 6; }
  else if (m < 13) { L.a1 = 8 + (m - 10); L.a2 = 6; }
  else { L.a1 = 11; L.a2 = 11; }
  // b_m
  if (m < 10) {
    int aa = 0, k = m;
    while (k >= 4 - aa) { k -= 4 - aa; ++aa; }
    const int bb = aa + k;
    L.b1 = aa < 3 ? 3 + aa : 6; L.b2 = bb < 3 ? 3 + bb : 6;
  } else { L.b1 = 11; L.b2 = 11; }
  return L;
}

// the record of one correspondence (q = T s formed by the caller with the pinned expression); dmean / smt: zeros for the point-to-point
// class, whose moments are those of the raw coordinates (transform_estimation.hpp:50-102)
template <bool NRM>
__device__ __forceinline__ void aff_record(bool has, float qx, float qy, float qz, const float4 p, const float4 nv, const float* dmean, const float* smt, float* rec) {
  float4* r4 = reinterpret_cast<float4*>(rec);      // (48-byte records: 16-byte aligned)
  if (!has) { r4[0] = r4[1] = r4[2] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
  const float d0 = __fsub_rn(p.x, dmean[0]), d1 = __fsub_rn(p.y, dmean[1]), d2 = __fsub_rn(p.z, dmean[2]);
  const float s0 = __fsub_rn(qx, smt[0]), s1 = __fsub_rn(qy, smt[1]), s2 = __fsub_rn(qz, smt[2]);
  // n.dot(dst - dst_mean) (:464), f32 like the reference's dot product (the pairing of accumulate_pair<IM_AFF0>)
  const float res = NRM ? __fadd_rn(__fadd_rn(__fmul_rn(nv.x, d0), __fmul_rn(nv.y, d1)), __fmul_rn(nv.z, d2)) : 0.0f;
  r4[0] = make_float4(NRM ? nv.x : 0.0f, NRM ? nv.y : 0.0f, NRM ? nv.z : 0.0f, s0);
  r4[1] = make_float4(s1, s2, 1.0f, res);
  r4[2] = make_float4(d0, d1, d2, 0.0f);
}

// rank update with the wave's 64 records at zb (the caller's wave barriers around it)
__device__ __forceinline__ void aff_mfma_round(const float* zb, int lane, const AffLane& L, aff_double4& acc) {
  const int k4 = lane >> 4;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const float* r = zb + (4 * jj + k4) * AFF_REC;
    const double av = (double)r[L.a1] * (double)r[L.a2];
    const double bv = (double)r[L.b1] * (double)r[L.b2];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
}

// where slot s of a row sits in the tile: D[i][j] = sum a_i b_j
__device__ __forceinline__ bool aff_slot(int s, int& i, int& j) {
  i = j = 0;
  if (s < 60) { i = s / 10; j = s % 10; return true; }
  if (s < 72) { const int jj = (s - 60) >> 2, a = (s - 60) & 3; i = 6 + jj; j = aff_pair(a, 3); return true; }
  if (s < 82) { i = 9; j = s - 72; return true; }
  if (s < AFF_SUMS) { const int jj = (s - 82) >> 2, a = (s - 82) & 3; i = 10 + jj; j = aff_pair(a, 3); return true; }
  return false;
}
// position of entry (i, j) of a 16x16 f64 tile stored as db[reg * 64 + lane] (C/D: col = lane & 15, row = (lane >> 4) + 4 * reg)
__device__ __forceinline__ int aff_tile_elem(int i, int j) { return (i >> 2) * 64 + 16 * (i & 3) + j; }

// The block's row: the waves' tiles summed in wave order (fixed: bitwise reproducible).  raw: NW x FUSED_WAVE_BYTES of LDS (a tile
// is 2048 B); every thread of the block calls this.
template <int NW>
__device__ __forceinline__ void aff_write_row(unsigned char* raw, int wave, int lane, const aff_double4& acc, double* row) {
  double* const db = reinterpret_cast<double*>(raw + wave * FUSED_WAVE_BYTES);
  int i, j;
  const bool used = (int)threadIdx.x < AFF_ROW && aff_slot((int)threadIdx.x, i, j);
  const int e = aff_tile_elem(i, j);
  double v = 0.0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) db[r * 64 + lane] = acc[r];
  __syncthreads();
  if (used)
    for (int w = 0; w < NW; ++w) v += reinterpret_cast<const double*>(raw + w * FUSED_WAVE_BYTES)[e];
  if ((int)threadIdx.x < AFF_ROW) row[threadIdx.x] = v;
}

}  // namespace cilhip
