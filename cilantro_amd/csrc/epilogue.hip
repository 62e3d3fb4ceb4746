// epilogue.hip -- the epilogue of an iteration: fixed-order reduction of the block partials (k_reduce_stage1), the one-block solve + compose + loop state (k_solve, solve.hpp), k_init_state; split from kernels.hip.
#include "search_device.hpp"
#include "affine_device.hpp"

namespace cilhip {

// ---- epilogue: fixed-order reduction of block partials + solve + state update --------------------
__device__ void reduce_partials_block(const double* __restrict__ partials, int nblocks, double* sums /*shared*/) {
  __shared__ double sh[4][64];
  const int slot = threadIdx.x & 63, grp = threadIdx.x >> 6;
  double v = 0.0;
  if (slot < SUMS_MAX) {
    int b = grp;
    for (; b + 28 < nblocks; b += 32) {          // 8 independent loads in flight, added in ascending order
      double r[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = partials[(size_t)(b + 4 * k) * SUMS_MAX + slot];
#pragma unroll
      for (int k = 0; k < 8; ++k) v += r[k];
    }
    for (; b < nblocks; b += 4) v += partials[(size_t)b * SUMS_MAX + slot];
  }
  sh[grp][slot] = v;
  __syncthreads();
  if (threadIdx.x < SUMS_MAX)
    sums[threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
  __syncthreads();
}

// Stage 1 of the cross-block reduction: REDUCE_GROUPS blocks, each folding a contiguous slice of the
// per-block partials (fixed order => deterministic).  A single block reading all 2048 x 48 doubles
// is latency-bound (~170 us measured); 32 blocks do it in a few us.
constexpr int REDUCE_GROUPS_MAX = 128;
static inline int reduce_groups(int nblocks) { return nblocks > 4096 ? REDUCE_GROUPS_MAX : 32; }   // ~50 rows per group at most

__global__ __launch_bounds__(256) void k_reduce_stage1(const double* __restrict__ partials, int nblocks, double* __restrict__ stage) {
  __shared__ double sums[SUMS_MAX];
  const int per = (nblocks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int b0 = blockIdx.x * per;
  const int b1 = min(b0 + per, nblocks);
  reduce_partials_block(partials + (size_t)b0 * SUMS_MAX, max(b1 - b0, 0), sums);
  if (threadIdx.x < SUMS_MAX) stage[blockIdx.x * SUMS_MAX + threadIdx.x] = sums[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_reduce_partials(const double* partials, int nblocks, double* out) {
  __shared__ double sums[SUMS_MAX];
  reduce_partials_block(partials, nblocks, sums);
  if (threadIdx.x < SUMS_MAX) out[threadIdx.x] = sums[threadIdx.x];
}

// partials[nblocks][SUMS_MAX] -> out[SUMS_MAX]; `stage` is scratch of REDUCE_GROUPS*SUMS_MAX doubles.
void launch_reduce_partials(const double* partials, int nblocks, double* stage, double* out, hipStream_t s) {
  if (nblocks > 64) {
    const int G = reduce_groups(nblocks);
    hipLaunchKernelGGL(k_reduce_stage1, dim3(G), dim3(256), 0, s, partials, nblocks, stage);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, s, (const double*)stage, G, out);
  } else {
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, s, partials, nblocks, out);
  }
}

// Stage 1 with a number of groups the CALLER fixes (every group's row is written, empty groups as zeros)
void launch_reduce_stage1_groups(const double* partials, int nblocks, double* stage, int groups, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_stage1, dim3(groups), dim3(256), 0, s, partials, nblocks, stage);
}

// Stage 1 only (the epilogue kernel k_solve folds the REDUCE_GROUPS rows itself).  Returns the number
// of rows k_solve has to read from `stage`, or 0 if it should read `partials` directly.
int launch_reduce_stage1(const double* partials, int nblocks, double* stage, hipStream_t s) {
  if (nblocks <= 64) return 0;
  const int G = reduce_groups(nblocks);
  hipLaunchKernelGGL(k_reduce_stage1, dim3(G), dim3(256), 0, s, partials, nblocks, stage);
  return G;
}

// The warm-started iteration's margin test needs two numbers about the queries q = T s (s in the source's bounding box: centre c,
// half extents h, source coordinates):
//   * how far any query moves when T becomes T' :  |(T' - T)(s, 1)| = |A s + b| <= |A c + b| + sum_j |A e_j| h_j   (A = L' - L, b = t' - t;
//     an affine function of s, the norm of its linear part bounded column by column) -- f64 of f32 entries, then rounded UP;
//   * the rounding error of a computed query: three products and three sums per component, each within 2^-24 relative of
//     |L_r0 x| + |L_r1 y| + |L_r2 z| + |t_r|: at most 2^-22 of that sum per component, sqrt(3) 2^-22 < 2^-21 for the norm of the three.
__device__ __forceinline__ float motion_eps_of(const float* T, const float* c, const float* h) {
  float m = 0.0f;
  for (int r = 0; r < 3; ++r) {
    float v = fabsf(T[12 + r]);
    for (int j = 0; j < 3; ++j) v += fabsf(T[j * 4 + r]) * (fabsf(c[j]) + h[j]);
    m = fmaxf(m, v);
  }
  return m * 6.0e-7f;      // > 2^-21
}
__device__ __forceinline__ float motion_step_of(const float* Told, const float* Tnew, const float* c, const float* h) {
  double v2 = 0.0, spread = 0.0;
  for (int r = 0; r < 3; ++r) {
    double v = (double)Tnew[12 + r] - (double)Told[12 + r];
    for (int j = 0; j < 3; ++j) v += ((double)Tnew[j * 4 + r] - (double)Told[j * 4 + r]) * (double)c[j];
    v2 += v * v;
  }
  for (int j = 0; j < 3; ++j) {
    double col = 0.0;
    for (int r = 0; r < 3; ++r) { const double d = (double)Tnew[j * 4 + r] - (double)Told[j * 4 + r]; col += d * d; }
    spread += sqrt(col) * (double)h[j];
  }
  return (float)((sqrt(v2) + spread) * 1.000001);      // (the conversion rounds to nearest: 2^-24 relative, covered)
}

__device__ void reset_inner(IcpState* st) {
  for (int i = 0; i < 9; ++i) { st->dLd[i] = (i % 4 == 0) ? 1.0 : 0.0; st->innerL[i] = (i % 4 == 0) ? 1.0f : 0.0f; }
  for (int i = 0; i < 3; ++i) { st->dtd[i] = 0.0; st->innert[i] = 0.0f; }
  st->inner_done = 0;
  st->pad0 = 0;
}

// The tail of an outer iteration once the new transform is known (icp_base.hpp:75-86): motion clock of the warm-started form, Tprev / T,
// the transformed source mean, delta / iterations / ncorr / done, the slab guard.  Shared by the rigid and the affine epilogue.
__device__ __forceinline__ void finalize_state(IcpState* st, const SolveArgs& a, const float* Tn, const float delta) {
  {
    const float step = motion_step_of(st->T, Tn, a.src_center, a.src_half);
    const float prev = st->motion_step;
    st->motion_pred = (prev < INFINITY && prev > 0.0f) ? step * fminf(1.0f, step / prev) : 0.0f;
    st->motion_step = step;
    st->motion_acc = (float)(((double)st->motion_acc + (double)step) * 1.000001);
    st->motion_eps = motion_eps_of(Tn, a.src_center, a.src_half);
  }
  for (int i = 0; i < 16; ++i) { st->Tprev[i] = st->T[i]; st->T[i] = Tn[i]; }
  float mx, my, mz;
  transform_point(Tn, a.src_mean[0], a.src_mean[1], a.src_mean[2], mx, my, mz);
  st->smt[0] = mx; st->smt[1] = my; st->smt[2] = mz;
  st->prev_delta = st->delta;
  st->delta = delta;
  st->iterations += 1;
  st->ncorr = (unsigned long long)(st->sums[0] + 0.5);   // (the sums of the last accumulation that ran: a converged inner loop skips the later ones)
  st->done = (delta < a.conv_tol) ? 1 : 0;                            // icp_base.hpp:83
  reset_inner(st);
  if (a.guard_axis >= 0) {
    // |((T - T_part) p)_axis| over the source's bounding box: an affine function of p, extreme at a corner
    const int ax = a.guard_axis;
    float d = Tn[12 + ax] - a.guard_T[12 + ax], spread = 0.0f;
    for (int j = 0; j < 3; ++j) {
      const float dl = Tn[j * 4 + ax] - a.guard_T[j * 4 + ax];
      d += dl * a.guard_center[j];
      spread += fabsf(dl) * a.guard_half[j];
    }
    if (!(fabsf(d) + spread <= a.guard_slack) && st->slab_violation == 0) {
      st->slab_violation = 1;
      st->violation_iter = st->iterations; st->violation_delta = delta; st->violation_ncorr = st->ncorr;
      for (int i = 0; i < 16; ++i) st->violation_T[i] = Tn[i];
    }
  }
}

// what the host's paced loop and cilhip_get_last_run_trace read: the iteration's record, then -- after a system-scope fence -- `latest`
__device__ __forceinline__ void publish_state(const SolveArgs& a, const IcpState& lst) {
  if (a.trace != nullptr && a.gn_last_step && threadIdx.x == 0 && lst.iterations >= 1 && lst.iterations <= RUN_TRACE_CAP)
    a.trace[lst.iterations - 1] = make_uint4(lst.unproven, lst.listed, __float_as_uint(lst.motion_step), __float_as_uint(lst.delta));
  if (a.feedback != nullptr && a.gn_last_step && threadIdx.x == 0) {
    FeedbackSlot* sl = &a.feedback->slot[(unsigned int)lst.iterations & 3u];
    sl->unproven = lst.unproven;
    sl->listed = lst.listed;
    sl->delta = lst.delta;
    sl->prev_delta = lst.prev_delta;
    sl->step = lst.motion_step;
    sl->commit = ((unsigned long long)a.run_tag << 32) | (unsigned long long)(unsigned int)lst.iterations;
    __threadfence_system();
    a.feedback->latest = ((unsigned long long)a.run_tag << 32) | (lst.done ? 0x80000000ull : 0ull) | (unsigned long long)((unsigned int)lst.iterations & 0x7fffffffu);
  }
}

// The epilogue proper (one block of 256 threads): k_solve's body, also the tail of k_reduce_solve's last block.
__device__ __forceinline__ void solve_body(const SolveArgs& a) {
  __shared__ double sums[SUMS_MAX];
  __shared__ IcpState lst;   // the state is pulled into LDS in one coalesced pass, updated by one lane, written back in one pass:
                             // the serial epilogue then pays one global round trip instead of one per field it touches
  static_assert(sizeof(IcpState) % 4 == 0, "IcpState is copied as dwords");
  constexpr int ST_DWORDS = (int)(sizeof(IcpState) / 4);
  if (a.state->done) return;
  // (everything the kernel reads from global memory is REQUESTED before anything is waited for: the state -- one dword per thread --,
  //  the search kernels' counters and the partial rows travel together: one far round trip instead of three in a row)
  static_assert(ST_DWORDS <= 256, "one dword of the state per thread");
  const uint32_t sreg = (int)threadIdx.x < ST_DWORDS ? reinterpret_cast<const uint32_t*>(a.state)[threadIdx.x] : 0u;
  __shared__ unsigned int unproven_total;
  __shared__ unsigned int listed_total;
  const bool counters = a.unproven_cnt != nullptr && a.gn_last_step && threadIdx.x < 128;   // (once per iteration; wave 0: unproven, wave 1: listed)
  unsigned int cv = counters ? a.unproven_cnt[threadIdx.x] : 0u;
  if (a.nblocks > 0) {
    reduce_partials_block(a.partials, a.nblocks, sums);
  } else {
    if (threadIdx.x < SUMS_MAX) sums[threadIdx.x] = a.reduced[threadIdx.x];
  }
  if ((int)threadIdx.x < ST_DWORDS) reinterpret_cast<uint32_t*>(&lst)[threadIdx.x] = sreg;
  if (counters) {
    a.unproven_cnt[threadIdx.x] = 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cv += __shfl_down(cv, off, 64);
    if (threadIdx.x == 0) unproven_total = cv;
    if (threadIdx.x == 64) listed_total = cv;
  }
  __syncthreads();
  IcpState* st = &lst;
  if (threadIdx.x == 0) {
  if (a.unproven_cnt != nullptr && a.gn_last_step) { st->unproven = unproven_total; st->listed = listed_total; }

  const double n = sums[0];
  double L[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  bool finalize = false;
  if (a.metric == IM_KABSCH) {
    for (int i = 0; i < SUMS_MAX; ++i) st->sums[i] = sums[i];
    kabsch_from_sums(sums, L, t);
    finalize = true;
  } else {
    // transform_estimation.hpp:264-272: no usable terms => tform stays identity, return false
    const bool has_p2p = (n > 0.0) && (a.w_p2p > 0.0f);
    const bool has_p2pl = (n > 0.0) && (a.w_p2pl > 0.0f);
    if (!st->inner_done) {
      if ((!has_p2p && !has_p2pl) || (has_p2pl && !a.has_normals)) {
        st->pad0 = 1;  // identity step
        st->inner_done = 1;
        for (int i = 0; i < SUMS_MAX; ++i) st->sums[i] = sums[i];
      } else if (a.gn_zero_steps) {
        // no Gauss-Newton step at all: the inner transform stays the identity and is un-centred below
        st->inner_done = 1;
        for (int i = 0; i < SUMS_MAX; ++i) st->sums[i] = sums[i];
      } else {
        // (registers for the usual case -- every index below is a compile-time constant once the loops are unrolled; the pivoted
        //  solve indexes its arrays dynamically: LDS for that one, scratch would be global memory)
        double AtA[36], Atb[6], dth[6];
        gn_normal_equations(sums, has_p2p ? (double)a.w_p2p : 0.0, has_p2pl ? (double)a.w_p2pl : 0.0, AtA, Atb, a.point_weighted != 0);
        if (!ldlt6_solve_fast(AtA, Atb, dth)) {      // (pivoted: rank-deficient systems only)
          __shared__ double sA[36], sb[6], sx[6], wsA[36], wsy[6];
          __shared__ int wsperm[6];
#pragma unroll
          for (int i = 0; i < 36; ++i) sA[i] = AtA[i];
#pragma unroll
          for (int i = 0; i < 6; ++i) sb[i] = Atb[i];
          ldlt6_solve_ws(sA, sb, sx, wsA, wsy, wsperm);
#pragma unroll
          for (int i = 0; i < 6; ++i) dth[i] = sx[i];
        }
        rigid_gn_update(dth, st->dLd, st->dtd);
        for (int i = 0; i < 9; ++i) st->innerL[i] = (float)st->dLd[i];
        for (int i = 0; i < 3; ++i) st->innert[i] = (float)st->dtd[i];
        double nrm = 0.0;
        for (int i = 0; i < 6; ++i) nrm += dth[i] * dth[i];
        if (sqrt(nrm) < (double)a.opt_conv_tol) st->inner_done = 1;     // :360
        for (int i = 0; i < SUMS_MAX; ++i) st->sums[i] = sums[i];
      }
    }
    if (a.gn_last_step) {
      if (st->pad0 == 0) {
        for (int i = 0; i < 9; ++i) L[i] = st->dLd[i];
        // tform = t_dst * tform * t_src, t_src = Translation(-(transform_*src_mean_))   :361/:365
        for (int r = 0; r < 3; ++r)
          t[r] = st->dtd[r] - (L[r * 3] * (double)st->smt[0] + L[r * 3 + 1] * (double)st->smt[1] + L[r * 3 + 2] * (double)st->smt[2]) +
                 (double)a.dst_mean[r];
      }
      finalize = true;
    }
  }
  if (finalize) {
    float Tn[16];
    const float delta = compose_update(L, t, st->T, Tn);
    finalize_state(st, a, Tn, delta);
  }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < ST_DWORDS; k += 256) reinterpret_cast<uint32_t*>(a.state)[k] = reinterpret_cast<const uint32_t*>(&lst)[k];
  publish_state(a, lst);
}
__global__ __launch_bounds__(256) void k_solve(SolveArgs a) { solve_body(a); }

void launch_solve(const SolveArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_solve, dim3(1), dim3(256), 0, s, a);
}

// Stage 1 of the cross-block reduction AND the epilogue in one launch: the G blocks fold their slices as k_reduce_stage1 does (same
// rows, same order: the sums are bitwise the two-kernel path's), publish their row -- a device-scope release: the row leaves this
// XCD's L2 before the ticket is taken -- and the block that takes the LAST ticket (device-scope acquire: its caches are invalidated,
// the other XCDs' rows are read from memory) runs the epilogue over the G rows.  One launch and one kernel boundary less per
// iteration.  The same hand-over inside the 1024-block accumulation kernels was measured three times slower than they are
// (NOTEBOOK.md: every one of their blocks pays the write-back while the others are still streaming); here it is paid by 32 blocks
// whose only stores are their rows.
__global__ __launch_bounds__(256) void k_reduce_solve(const double* __restrict__ partials, int nblocks, double* __restrict__ stage, unsigned int* ticket, SolveArgs a) {
  if (a.state->done) return;      // (read by every block before any block can change it: the epilogue runs after the last ticket)
  {
    __shared__ double rsums[SUMS_MAX];
    const int per = (nblocks + (int)gridDim.x - 1) / (int)gridDim.x;
    const int b0 = blockIdx.x * per;
    const int b1 = min(b0 + per, nblocks);
    reduce_partials_block(partials + (size_t)b0 * SUMS_MAX, max(b1 - b0, 0), rsums);
    if (threadIdx.x < SUMS_MAX) stage[blockIdx.x * SUMS_MAX + threadIdx.x] = rsums[threadIdx.x];
  }
  __shared__ unsigned int last_block;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(ticket, 1u);
    last_block = (t == gridDim.x - 1u) ? 1u : 0u;
    if (t == gridDim.x - 1u) *ticket = 0u;      // (for the next launch: nobody else touches it any more)
  }
  __syncthreads();
  if (last_block == 0u) return;
  __threadfence();
  a.partials = stage;
  a.nblocks = (int)gridDim.x;
  a.reduced = nullptr;
  solve_body(a);
}
// partials[nblocks] -> epilogue.  Few rows: the epilogue folds them itself; many: one launch does both stages (above).
void launch_reduce_and_solve(const double* partials, int nblocks, double* stage, unsigned int* ticket, SolveArgs a, hipStream_t s) {
  if (nblocks <= 64 || ticket == nullptr) {
    const int rows = ticket == nullptr ? launch_reduce_stage1(partials, nblocks, stage, s) : 0;
    a.partials = rows ? stage : partials; a.nblocks = rows ? rows : nblocks; a.reduced = nullptr;
    hipLaunchKernelGGL(k_solve, dim3(1), dim3(256), 0, s, a);
    return;
  }
  hipLaunchKernelGGL(k_reduce_solve, dim3(reduce_groups(nblocks)), dim3(256), 0, s, partials, nblocks, stage, ticket, a);
}


// ---- the affine classes' epilogue (SimpleCombinedMetricAffineICP3f / SimplePointToPointMetricAffineICP3f) ----------------------------------
// Rows of AFF_ROW doubles (affine_device.hpp) -> fixed-order fold -> AtA (12x12), Atb -> AtA.ldlt().solve(Atb) (diagonal pivoting,
// pseudo-inverse of D: solve.hpp ldlt_solve_n operation for operation, its elimination steps dealt out to the block's threads) ->
// tform = t_dst * tform * t_src (transform_estimation.hpp:466-473) -> transform_ = tform_iter * transform_ in f32, delta = the
// Frobenius norm of tform_iter - I (icp_single_transform_combined_metric.hpp:207-216 without the rotation() polish of the rigid
// classes; icp_single_transform_point_to_point_metric.hpp:56-64) -> the loop state.  No host round trip per iteration.
constexpr int AFF_GROUPS = 32;
__device__ __forceinline__ double aff_fold_rows(const double* __restrict__ rows, int n, int slot) {
  double v = 0.0;
  int b = 0;
  for (; b + 8 <= n; b += 8) {          // 8 independent loads in flight, added in ascending order
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = rows[(size_t)(b + k) * AFF_ROW + slot];
#pragma unroll
    for (int k = 0; k < 8; ++k) v += r[k];
  }
  for (; b < n; ++b) v += rows[(size_t)b * AFF_ROW + slot];
  return v;
}
__global__ __launch_bounds__(AFF_ROW) void k_reduce_affine(const double* __restrict__ partials, int nrows, double* __restrict__ stage, const IcpState* st) {
  if (st->done) return;
  const int per = (nrows + (int)gridDim.x - 1) / (int)gridDim.x;
  const int b0 = min((int)blockIdx.x * per, nrows), b1 = min(b0 + per, nrows);
  stage[(size_t)blockIdx.x * AFF_ROW + threadIdx.x] = aff_fold_rows(partials + (size_t)b0 * AFF_ROW, b1 - b0, (int)threadIdx.x);
}

// (ONE wave: the barriers between the elimination's phases then cost nothing -- with two waves they were a third of the kernel)
constexpr int AFF_SOLVE_THREADS = 64;
__global__ __launch_bounds__(AFF_SOLVE_THREADS) void k_solve_affine(SolveArgs a) {
  __shared__ double sums[AFF_ROW];
  __shared__ IcpState lst;
  __shared__ double A[144], rhs[12], x[12];
  __shared__ int perm[12];
  __shared__ unsigned int unproven_total, listed_total;
  constexpr int ST_DWORDS = (int)(sizeof(IcpState) / 4);
  constexpr int NT = AFF_SOLVE_THREADS, N = 12;
  static_assert(AFF_ROW == 2 * NT, "two slots of a row per thread");
  const int t = (int)threadIdx.x;
  if (a.state->done) return;
#ifdef CILHIP_EXP_AFF_CLOCKS
  const unsigned long long c0_ = wall_clock64();
  unsigned long long c1_ = 0, c2_ = 0, c3_ = 0;
#define AFF_CLK(v) v = wall_clock64()
#else
#define AFF_CLK(v)
#endif
  uint32_t sreg[(ST_DWORDS + NT - 1) / NT];
#pragma unroll
  for (int k = 0; k < (ST_DWORDS + NT - 1) / NT; ++k) sreg[k] = t + k * NT < ST_DWORDS ? reinterpret_cast<const uint32_t*>(a.state)[t + k * NT] : 0u;
  const bool counters = a.unproven_cnt != nullptr;
  unsigned int cv = counters ? a.unproven_cnt[t] : 0u, cw = counters ? a.unproven_cnt[NT + t] : 0u;      // (the unproven counts / the listed ones)
  {      // both slots of the thread in one sweep over the rows: sixteen independent loads in flight, each slot added in ascending row order
    double v0 = 0.0, v1 = 0.0;
    int b = 0;
    for (; b + 8 <= a.nblocks; b += 8) {
      double r0[8], r1[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { r0[k] = a.partials[(size_t)(b + k) * AFF_ROW + t]; r1[k] = a.partials[(size_t)(b + k) * AFF_ROW + NT + t]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) { v0 += r0[k]; v1 += r1[k]; }
    }
    for (; b < a.nblocks; ++b) { v0 += a.partials[(size_t)b * AFF_ROW + t]; v1 += a.partials[(size_t)b * AFF_ROW + NT + t]; }
    sums[t] = v0; sums[NT + t] = v1;
  }
#pragma unroll
  for (int k = 0; k < (ST_DWORDS + NT - 1) / NT; ++k) if (t + k * NT < ST_DWORDS) reinterpret_cast<uint32_t*>(&lst)[t + k * NT] = sreg[k];
  if (counters) {
    a.unproven_cnt[t] = 0u; a.unproven_cnt[NT + t] = 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { cv += __shfl_down(cv, off, 64); cw += __shfl_down(cw, off, 64); }
    if (t == 0) { unproven_total = cv; listed_total = cw; }
  }
  __syncthreads();
  AFF_CLK(c1_);
  IcpState* st = &lst;
  const double n = sums[72 + aff_pair(3, 3)];
  const bool wp = a.w_p2p > 0.0f, wl = a.w_p2pl > 0.0f;
  // transform_estimation.hpp:400-409: no usable terms, or plane terms without normals -> identity (and n == 0: nothing to solve)
  const bool identity = (!wp && !wl) || (wl && !a.has_normals) || !(n > 0.0);
  if (!identity) {
    const double w_pt = wp ? (double)a.w_p2p : 0.0, w_pl = wl ? (double)a.w_p2pl : 0.0;
    // unknown (j, a): 3 j + a for the linear part's row j, 9 + j for the translation -- eq_vec's own order (:457-460)
    auto ja = [](int r, int& j, int& aa) { if (r < 9) { j = r / 3; aa = r % 3; } else { j = r - 9; aa = 3; } };
    for (int e = t; e < 144; e += NT) {
      const int r = e / 12, c = e % 12;
      int j, aa, k, bb;
      ja(r, j, aa); ja(c, k, bb);
      const int ab = aa < bb ? aff_pair(aa, bb) : aff_pair(bb, aa);
      double v = 0.0;
      if (w_pl > 0.0) v += w_pl * sums[(j < k ? aff_jk(j, k) : aff_jk(k, j)) * 10 + ab];      // sum n_j n_k s'_a s'_b
      if (w_pt > 0.0 && j == k) v += w_pt * sums[72 + ab];                                    // sum s'_a s'_b
      A[e] = v;
    }
    if (t < 12) {
      int j, aa;
      ja(t, j, aa);
      double bv = 0.0;
      if (w_pl > 0.0) bv += w_pl * sums[60 + 4 * j + aa];      // sum (n.d) n_j s'_a
      if (w_pt > 0.0) bv += w_pt * sums[82 + 4 * j + aa];      // sum d_j s'_a
      rhs[t] = bv;
      perm[t] = t;
    }
    __syncthreads();
    AFF_CLK(c2_);
    const double tiny = 2.2250738585072014e-308;
    // (what a lane updates in every elimination step: up to three entries (i, j) of the lower triangle, fixed)
    int ui[3], uj[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { const int e = t + q * NT; ui[q] = e < 144 ? e / 12 : 0; uj[q] = e < 144 ? e % 12 : 12; }
    for (int k = 0; k < N; ++k) {
      // the pivot: the largest remaining |diagonal|, the lowest index among equals (ldlt_solve_n's scan: a later entry replaces the best only
      // if strictly larger) -- lanes 0 .. 11 hold one diagonal entry each, a butterfly over 16 lanes finds it
      double pv = (t < N && t >= k) ? fabs(A[t * N + t]) : -1.0;
      int piv = t < N ? t : N;
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        const double ov = __shfl_xor(pv, off, 64);
        const int oi = __shfl_xor(piv, off, 64);
        if (ov > pv || (ov == pv && oi < piv)) { pv = ov; piv = oi; }
      }
      piv = __shfl(piv, 0, 64);
      if (!(piv >= k && piv < N)) piv = k;      // (nothing comparable on the diagonal: no swap)
      if (piv != k) {
        if (t < N) { const double v = A[k * N + t]; A[k * N + t] = A[piv * N + t]; A[piv * N + t] = v; }
        __syncthreads();
        if (t < N) { const double v = A[t * N + k]; A[t * N + k] = A[t * N + piv]; A[t * N + piv] = v; }
        if (t == 0) { const int v = perm[k]; perm[k] = perm[piv]; perm[piv] = v; }
        __syncthreads();
      }
      const double dk = A[k * N + k];
      if (fabs(dk) <= tiny) {
        if (t > k && t < N) A[t * N + k] = 0.0;
        __syncthreads();
        continue;
      }
      if (t > k && t < N) A[t * N + k] = A[t * N + k] / dk;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int i = ui[q], j = uj[q];
        if (j > k && j <= i) {
          const double v = __dsub_rn(A[i * N + j], __dmul_rn(__dmul_rn(A[i * N + k], dk), A[j * N + k]));
          A[i * N + j] = v;
          A[j * N + i] = v;
        }
      }
      __syncthreads();
    }
  }
  AFF_CLK(c3_);
  if (t == 0) {
    if (counters) { st->unproven = unproven_total; st->listed = listed_total; }
    double L[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tr[3] = {0, 0, 0};
    if (!identity) {
      // (the two triangular solves with y in REGISTERS -- every index a compile-time constant once unrolled: through an LDS array each of
      //  their 132 steps was a dependent LDS round trip, two thirds of the kernel)
      double yr[N];
#pragma unroll
      for (int i = 0; i < N; ++i) yr[i] = rhs[perm[i]];
#pragma unroll
      for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < i; ++j) yr[i] = __dsub_rn(yr[i], __dmul_rn(A[i * N + j], yr[j]));
      }
#pragma unroll
      for (int i = 0; i < N; ++i) { const double d = A[i * N + i]; yr[i] = (fabs(d) > 2.2250738585072014e-308) ? yr[i] / d : 0.0; }
#pragma unroll
      for (int i = N - 1; i >= 0; --i) {
#pragma unroll
        for (int j = i + 1; j < N; ++j) yr[i] = __dsub_rn(yr[i], __dmul_rn(A[j * N + i], yr[j]));
      }
#pragma unroll
      for (int i = 0; i < N; ++i) x[perm[i]] = yr[i];
      for (int i = 0; i < 9; ++i) L[i] = x[i];              // :470-472 row-major linear part, then the translation
      for (int i = 0; i < 3; ++i) tr[i] = x[9 + i];
      if (a.affine_centered)                                // :473 tform = t_dst * tform * t_src
        for (int r = 0; r < 3; ++r)
          tr[r] = tr[r] - (L[r * 3] * (double)st->smt[0] + L[r * 3 + 1] * (double)st->smt[1] + L[r * 3 + 2] * (double)st->smt[2]) + (double)a.dst_mean[r];
    }
    float dT[16];
    for (int i = 0; i < 16; ++i) dT[i] = 0.0f;
    dT[15] = 1.0f;
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) dT[c * 4 + r] = (float)L[r * 3 + c]; dT[12 + r] = (float)tr[r]; }
    // transform_ = tform_iter * transform_ (f32, Eigen's affine product) and |tform_iter - I|_F
    const float* T = st->T;
    float Tn[16];
    for (int i = 0; i < 16; ++i) Tn[i] = 0.0f;
    Tn[15] = 1.0f;
    for (int r = 0; r < 3; ++r) {
      for (int cc = 0; cc < 3; ++cc)
        Tn[cc * 4 + r] = __fadd_rn(__fadd_rn(__fmul_rn(dT[0 * 4 + r], T[cc * 4 + 0]), __fmul_rn(dT[1 * 4 + r], T[cc * 4 + 1])), __fmul_rn(dT[2 * 4 + r], T[cc * 4 + 2]));
      Tn[12 + r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(dT[0 * 4 + r], T[12]), __fmul_rn(dT[1 * 4 + r], T[13])), __fmul_rn(dT[2 * 4 + r], T[14])), dT[12 + r]);
    }
    float dn = 0.0f;
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) { const float v = __fsub_rn(dT[cc * 4 + r], r == cc ? 1.0f : 0.0f); dn = __fadd_rn(dn, __fmul_rn(v, v)); }
    for (int r = 0; r < 3; ++r) dn = __fadd_rn(dn, __fmul_rn(dT[12 + r], dT[12 + r]));
    const float delta = sqrtf(dn);
    for (int i = 0; i < SUMS_MAX; ++i) st->sums[i] = 0.0;
    st->sums[0] = n;
    finalize_state(st, a, Tn, delta);
#ifdef CILHIP_EXP_AFF_CLOCKS
    {      // (dev build: the trace's counters carry the phases in 10 ns ticks: fold | assembly, elimination | solves + compose + state)
      const unsigned long long c4_ = wall_clock64();
      st->unproven = (unsigned int)((c1_ - c0_) & 0xffffu) | (unsigned int)(((c2_ - c1_) & 0xffffu) << 16);
      st->listed = (unsigned int)((c3_ - c2_) & 0xffffu) | (unsigned int)(((c4_ - c3_) & 0xffffu) << 16);
    }
#endif
  }
  __syncthreads();
  for (int k = t; k < ST_DWORDS; k += NT) reinterpret_cast<uint32_t*>(a.state)[k] = reinterpret_cast<const uint32_t*>(&lst)[k];
  publish_state(a, lst);
}

// rows -> epilogue: more than 64 rows go through AFF_GROUPS partial folds first (same rows, same order every run)
void launch_reduce_and_solve_affine(const double* partials, int nrows, double* stage, const SolveArgs& a0, hipStream_t s) {
  SolveArgs a = a0;
  a.gn_last_step = 1;
  if (nrows > 64) {
    hipLaunchKernelGGL(k_reduce_affine, dim3(AFF_GROUPS), dim3(AFF_ROW), 0, s, partials, nrows, stage, (const IcpState*)a.state);
    a.partials = stage; a.nblocks = AFF_GROUPS;
  } else {
    a.partials = partials; a.nblocks = nrows;
  }
  a.reduced = nullptr;
  hipLaunchKernelGGL(k_solve_affine, dim3(1), dim3(AFF_SOLVE_THREADS), 0, s, a);
}

struct InitArgs { float T[16]; float src_mean[3]; Feedback* fb; unsigned int run_tag; float src_center[3], src_half[3]; unsigned int* tie_counters; };

__global__ void k_init_state(IcpState* st, InitArgs ia) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (ia.tie_counters != nullptr) { for (int k = 0; k < 4; ++k) ia.tie_counters[k] = 0u; }
  if (ia.fb != nullptr) {
    for (int k = 0; k < 4; ++k) { ia.fb->slot[k].unproven = 0u; ia.fb->slot[k].listed = 0u; ia.fb->slot[k].delta = 0.0f; ia.fb->slot[k].prev_delta = 0.0f; ia.fb->slot[k].step = 0.0f; ia.fb->slot[k].pad = 0.0f; ia.fb->slot[k].commit = 0ull; }
    __threadfence_system();
    ia.fb->latest = (unsigned long long)ia.run_tag << 32;
  }
  for (int i = 0; i < 16; ++i) st->T[i] = st->Tprev[i] = ia.T[i];
  float mx, my, mz;
  transform_point(ia.T, ia.src_mean[0], ia.src_mean[1], ia.src_mean[2], mx, my, mz);
  st->smt[0] = mx; st->smt[1] = my; st->smt[2] = mz;
  st->delta = INFINITY;
  st->prev_delta = INFINITY;
  st->iterations = 0;
  st->done = 0;
  st->ncorr = 0;
  for (int i = 0; i < SUMS_MAX; ++i) st->sums[i] = 0.0;
  st->slab_violation = 0; st->unproven = 0; st->listed = 0;
  st->violation_iter = 0; st->violation_delta = 0.0f; st->violation_ncorr = 0ull;
  for (int i = 0; i < 16; ++i) st->violation_T[i] = ia.T[i];
  st->motion_acc = 0.0f; st->motion_step = INFINITY; st->motion_pred = 0.0f;
  st->motion_eps = motion_eps_of(ia.T, ia.src_center, ia.src_half);
  reset_inner(st);
}

void launch_init_state(IcpState* st, const float T0[16], const float src_mean[3], hipStream_t s, Feedback* fb, unsigned int run_tag,
                       const float* src_center, const float* src_half, unsigned int* tie_counters) {
  InitArgs ia;
  ia.fb = fb; ia.run_tag = run_tag; ia.tie_counters = tie_counters;
  // (no bounding box given: a huge one -- the margin test then settles nothing)
  for (int i = 0; i < 3; ++i) { ia.src_center[i] = src_center ? src_center[i] : 0.0f; ia.src_half[i] = src_half ? src_half[i] : 1.0e30f; }
  for (int i = 0; i < 16; ++i) ia.T[i] = T0[i];
  for (int i = 0; i < 3; ++i) ia.src_mean[i] = src_mean[i];
  hipLaunchKernelGGL(k_init_state, dim3(1), dim3(64), 0, s, st, ia);
}

}  // namespace cilhip
