// warm.hip -- the warm-started iteration (k_warm<metric, rec>: margin proof, list search, matrix-core accumulation) and k_self_nn's nearest-other-point table; split from kernels.hip (overview there, DESIGN.md sections 5 and 6.2).
#include "search_device.hpp"
#include "affine_device.hpp"

namespace cilhip {

#ifdef CILHIP_EXP_PHASE_CLOCKS
// (instrumented build: per-block clocks of k_warm, dumped by c_api.hip through kernels.hip's debug_dump_phase_clocks)
__device__ unsigned long long g_warm_clk[16];
__device__ unsigned long long g_warm_stamp[4096][3];      // per block of the LAST k_warm launch: start / end (100 MHz wall clock)
#define WARM_CLK(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_warm_clk[k], now_ - tprev_); atomicAdd(&g_warm_clk[8 + (k)], 1ull); tprev_ = now_; } } while (0)
void debug_dump_warm_clocks() {
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_warm_clk), sizeof h) != hipSuccess) return;
  if (h[8] == 0) return;
  fprintf(stderr, "[warm clocks, 100 MHz ticks per block (thread 0: wave 0), %llu blocks] prologue=%.1f stream=%.1f levelA=%.1f (%.2f rounds) levelB=%.1f (%.2f) -=%.1f (%.2f) end=%.1f\n",
          h[8], (double)h[0] / h[8], (double)h[1] / h[8], (double)h[2] / h[8], (double)h[10] / h[8], (double)h[3] / h[8], (double)h[11] / h[8], (double)h[4] / h[8],
          (double)h[12] / h[8], (double)h[5] / h[8]);
  fprintf(stderr, "[warm list] entries at level A %llu, left open %llu (bound = radius: %llu, bound > cell: %llu)\n", h[6], h[7], h[14], h[15]);
  memset(h, 0, sizeof h);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_warm_clk), h, sizeof h);
  // the last launch, block by block: when it started / ended relative to the first start; by XCD (blockIdx & 7)
  static unsigned long long st[4096][3];
  if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_warm_stamp), sizeof st) != hipSuccess) return;
  unsigned long long t0 = ~0ull; int nb = 0;
  for (int b = 0; b < 4096; ++b) if (st[b][1]) { nb = b + 1; if (st[b][0] < t0) t0 = st[b][0]; }
  if (!nb) return;
  double smax = 0, emax = 0, emin = 1e30, dsum = 0, dmin = 1e30, dmax = 0;
  double xe[8] = {0}, xd[8] = {0}; int xn[8] = {0};
  for (int b = 0; b < nb; ++b) {
    const double s0 = (double)(st[b][0] - t0) / 100.0, e0 = (double)(st[b][1] - t0) / 100.0, d = e0 - s0;
    if (s0 > smax) smax = s0; if (e0 > emax) emax = e0; if (e0 < emin) emin = e0; dsum += d; if (d < dmin) dmin = d; if (d > dmax) dmax = d;
    if (e0 > xe[b & 7]) xe[b & 7] = e0; xd[b & 7] += d; ++xn[b & 7];
  }
  {
    int hist[24] = {0}; double lw = 0, lwo = 0; int nw = 0, nwo = 0; double pos[8] = {0}; int posn[8] = {0};
    for (int b = 0; b < nb; ++b) {
      const double d = (double)(st[b][1] - st[b][0]) / 100.0;
      int k = (int)(d / 5.0); if (k > 23) k = 23; ++hist[k];
      if (st[b][2]) { lw += d; ++nw; } else { lwo += d; ++nwo; }
      const int oct = ((b >> 3) * 8) / ((nb + 7) / 8); pos[oct < 8 ? oct : 7] += d; ++posn[oct < 8 ? oct : 7];
    }
    fprintf(stderr, "[warm stamps] lifetime histogram (5 us bins):");
    for (int k = 0; k < 24; ++k) fprintf(stderr, " %d", hist[k]);
    fprintf(stderr, "\n[warm stamps] blocks with listed queries: %d, avg life %.1f; without: %d, avg life %.1f; avg life by position of the chunk inside its XCD's share (eighths):", nw, nw ? lw / nw : 0.0, nwo, nwo ? lwo / nwo : 0.0);
    for (int k = 0; k < 8; ++k) fprintf(stderr, " %.1f", posn[k] ? pos[k] / posn[k] : 0.0);
    fprintf(stderr, "\n");
  }
  fprintf(stderr, "[warm stamps, last launch, %d blocks, us] last start=%.1f  first end=%.1f  last end=%.1f  block lifetime min/avg/max=%.1f/%.1f/%.1f  per XCD (avg life, last end):", nb, smax, emin, emax, dmin, dsum / nb, dmax);
  for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f,%.1f", xn[x] ? xd[x] / xn[x] : 0.0, xe[x]);
  fprintf(stderr, "\n");
  memset(st, 0, sizeof st);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_warm_stamp), st, sizeof st);
}
#else
#define WARM_CLK(k)
#endif

// ---- per target point: a lower bound on the squared distance to its nearest OTHER target point -------------------------
// What the warm-started iteration settles most queries with: if |q - p| < nnd(p) / 2 then p is the one nearest target
// point of q (any other p' has |q - p'| >= nnd(p) - |q - p| > |q - p|) -- no neighbour has to be looked at.  Computed once
// per target: minimum over the 3x3x3 block of cells around the point (itself excluded by position: a duplicate gives 0),
// capped by the distance to the faces of that block (whatever lies beyond is at least that far) -- a LOWER bound is all
// the test needs.
__global__ __launch_bounds__(256) void k_self_nn(GridDev g, float* __restrict__ safe2) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= g.n) return;
  const float4 p = g.pts[j];
  const int cx = min(max((int)floorf((p.x - g.ox) * g.inv_cell), 0), g.nx - 1), cy = min(max((int)floorf((p.y - g.oy) * g.inv_cell), 0), g.ny - 1),
            cz = min(max((int)floorf((p.z - g.oz) * g.inv_cell), 0), g.nz - 1);
  float best = INFINITY;
  const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
  for (int z = max(cz - 1, 0); z <= min(cz + 1, g.nz - 1); ++z)
    for (int y = max(cy - 1, 0); y <= min(cy + 1, g.ny - 1); ++y) {
      const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
      const uint32_t beg = g.cell_start[row + x0], end = g.cell_start[row + x1 + 1];
      for (uint32_t k = beg; k < end; ++k) {
        const float4 o = g.pts[k];
        const float e = d2_pinned(p.x, p.y, p.z, o.x, o.y, o.z);
        if (k != j) best = fminf(best, e);
      }
    }
  float b = INFINITY;      // faces of the block that still have cells beyond them
  if (cx - 1 > 0) b = fminf(b, p.x - (g.ox + (float)(cx - 1) * g.cell));
  if (cx + 2 < g.nx) b = fminf(b, (g.ox + (float)(cx + 2) * g.cell) - p.x);
  if (cy - 1 > 0) b = fminf(b, p.y - (g.oy + (float)(cy - 1) * g.cell));
  if (cy + 2 < g.ny) b = fminf(b, (g.oy + (float)(cy + 2) * g.cell) - p.y);
  if (cz - 1 > 0) b = fminf(b, p.z - (g.oz + (float)(cz - 1) * g.cell));
  if (cz + 2 < g.nz) b = fminf(b, (g.oz + (float)(cz + 2) * g.cell) - p.z);
  if (b != INFINITY) { b = fmaxf(b - g.margin, 0.0f); best = fminf(best, b * b * KSHRINK); }
  safe2[j] = best;
}
void launch_self_nn(const GridDev& g, float* safe2, hipStream_t s) {
  if (g.n == 0) return;
  hipLaunchKernelGGL(k_self_nn, dim3((g.n + 255u) / 256u), dim3(256), 0, s, g, safe2);
}

// ---- the WARM-STARTED iteration: search + accumulation from the previous iteration's matches -------------------------
// From the second iteration on every query has a match from the iteration before.  That match is a real target point, so
// its distance from the NEW q = T s bounds the search: anything nearer (or as near, with a lower index) lies in the ball
// of that radius around q.  Near alignment the radius is a small fraction of a cell and the ball stays inside q's octant
// block (the 2x2x2 cells q leans towards) -- usually inside q's own cell: per axis the neighbour is looked at only when
// the ball reaches its face.  No tile is staged: a lane reads its old match and the one to three cells its ball touches
// straight from memory (neighbouring lanes read neighbouring lines).  Two to three memory round trips per query:
// {old match, its normal, the run boundaries} -> {candidates, 4 per trip} -> done.  Queries without a usable bound (no old
// match, bound beyond the octant block, cell in the grid's outer layer) take the generic shell search -- exact as well --
// and are counted: the host goes back to the tiled kernels when they are many.  The matches and therefore the sums are
// the ones every other form finds; the accumulation is the tiles' rank update Z += z z^T on the matrix cores (per-wave
// 16x16 f64 tile kept in registers across the whole chunk, fixed order => bitwise reproducible run to run).
constexpr int WARM_THREADS = 256;
constexpr int WARM_WAVES = WARM_THREADS / 64;

// NR runs of the sorted target array, EVERY point evaluated (nothing culled: the caller wants a bound on all the points it did
// not choose), eight independent loads in flight per trip over the flattened index space of the runs; keeps the best key and the
// two smallest squared distances a1 <= b2 over the DISTINCT points met (the clamped re-reads past the end are not counted), and the
// best point's record (bp) so that the caller need not fetch it again.
template <int NR>
__device__ __forceinline__ void scan_runs_track2(const float4* __restrict__ pts, const uint32_t (&rb)[NR], const uint32_t (&re)[NR], float qx, float qy, float qz,
                                                 NN& best, float& a1, float& b2, float4& bp) {
  uint32_t pre[NR];      // inclusive prefix sums of the run lengths
  uint32_t total = 0;
#pragma unroll
  for (int r = 0; r < NR; ++r) { total += re[r] - rb[r]; pre[r] = total; }
  for (uint32_t t = 0; t < total; t += 8) {
    uint32_t j[8];
    float4 pc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t tt = min(t + (uint32_t)k, total - 1u);
      uint32_t jj = rb[0] + tt;
#pragma unroll
      for (int r = 1; r < NR; ++r) jj = tt >= pre[r - 1] ? rb[r] + (tt - pre[r - 1]) : jj;
      j[k] = jj;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) pc[k] = pts[j[k]];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float e = d2_pinned(qx, qy, qz, pc[k].x, pc[k].y, pc[k].z);
      const unsigned long long key = ((unsigned long long)__float_as_uint(e) << 32) | __float_as_uint(pc[k].w);
      if (key < best.key) { best.key = key; best.pos = j[k]; bp = pc[k]; }
      if (t + (uint32_t)k < total) { b2 = __builtin_amdgcn_fmed3f(a1, b2, e); a1 = fminf(a1, e); }
    }
  }
}
// REC: 0 = the old match, its normal and its table entry are gathered through warm_pos; 1 = the same, and every query's
// match record {point, table entry} {normal} (16 + 12 B, two arrays in query order) and a 12-byte copy of its source point
// are written; 2 = those are READ instead -- 40 B per query in three coalesced loads, no gather at all for the queries the
// table settles (nearly all of them); a query whose match changes rewrites its record.
// The queries the table does NOT settle (a percent or so) are not searched where they turn up -- nearly every wave holds
// one, and the whole wave would walk the search code for it: each wave lists them in LDS (ballot order: no atomics, the
// same list in every run) and searches the list afterwards, densely packed (the list holds all of the wave's queries if
// need be: a source far from alignment).
constexpr int WARM_QCAP = 256;                              // listed queries per wave (16 B each); a list that could not take another round is searched at once
#define Z4 make_float4(0.f, 0.f, 0.f, 0.f)
template <int ACC, int REC, bool SYM = false>
__global__ __launch_bounds__(WARM_THREADS, 4) void k_warm(IterArgs a) {
  const IcpState* __restrict__ st = a.state;
  if (st->done) return;
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = st->T[(i / 3) * 4 + (i % 3)];     // columns 0..3, rows 0..2
  // (loop state read HERE, into scalar registers: a load of it inside the streaming loop is a vector-memory load whose wait
  //  -- vmcnt counts in order -- also waits for the next round's prefetch, i.e. serialises memory latency and arithmetic)
  constexpr bool AFF = (ACC == IM_AFFC || ACC == IM_AFFP);      // the affine classes' moments (affine_device.hpp)
  const bool raw_moments = AFF && a.no_centering;                // (point-to-point class: no means subtracted)
  const float smt[3] = {raw_moments ? 0.0f : st->smt[0], raw_moments ? 0.0f : st->smt[1], raw_moments ? 0.0f : st->smt[2]};
  const float dmn[3] = {raw_moments ? 0.0f : a.dst_mean[0], raw_moments ? 0.0f : a.dst_mean[1], raw_moments ? 0.0f : a.dst_mean[2]};
  const MotionRef mref = {st->motion_acc, st->motion_eps};
  const float Dk = __fadd_rn(mref.acc, mref.eps) * 1.000001f;      // the motion clock now (rounded up): what a key is compared against
  const GridDev& g = a.grid;
  __shared__ __attribute__((aligned(16))) unsigned char raw[WARM_WAVES * FUSED_WAVE_BYTES];
  __shared__ float4 dq[WARM_WAVES][WARM_QCAP];             // listed queries: {q = T s, index}
  __shared__ float dr[WARM_WAVES][WARM_QCAP];              // ... and their bounds (squared)
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  float* const zb = reinterpret_cast<float*>(raw) + wave * (FUSED_WAVE_BYTES / 4);
  float4* const wq = dq[wave];
  float* const wr = dr[wave];
#ifdef CILHIP_EXP_PHASE_CLOCKS
  unsigned long long tprev_ = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x < 4096u) g_warm_stamp[blockIdx.x][0] = tprev_;
#endif
  typedef double double4_t __attribute__((ext_vector_type(4)));
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const AffLane afl = aff_lane((int)(threadIdx.x & 63u));
  constexpr int NC = FusedZ<ACC>::NC;
  constexpr bool DUAL = NC <= 8;
  constexpr bool NRM = FusedZ<ACC>::needs_normal;

  const uint32_t nb = gridDim.x;
  const uint32_t vb = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);      // XCD-aware (gridDim.x is a multiple of 8)
  // The source in ROUNDS of 256 queries, dealt out evenly: every block gets floor(R / nb) rounds, the first R mod nb blocks IN DISPATCH
  // ORDER (blockIdx: round-robin over the XCDs) one more -- no block without work (a chunk rounded up to whole rounds left the last
  // 2 % of the blocks idle at 10M), the heavier blocks spread over the XCDs.  vb's range starts after the rounds of the chunks before it.
  const uint32_t rounds_total = (a.ns + WARM_THREADS - 1) / WARM_THREADS, rbase = rounds_total / nb, rrem = rounds_total % nb;
  const uint32_t per_x = nb >> 3, xme = blockIdx.x & 7u, jme = blockIdx.x >> 3;
  uint32_t heavy_before = 0;      // heavier chunks among vb' < vb: chunk (x, j) is heavier iff its block index j * 8 + x < rrem
  for (uint32_t x = 0; x < xme; ++x) heavy_before += rrem > x ? min((rrem - x + 7u) >> 3, per_x) : 0u;
  heavy_before += rrem > xme ? min((rrem - xme + 7u) >> 3, jme) : 0u;
  const uint64_t beg64 = ((uint64_t)vb * rbase + heavy_before) * WARM_THREADS;
  const uint32_t my_rounds = rbase + (blockIdx.x < rrem ? 1u : 0u);
  const uint32_t beg = beg64 < a.ns ? (uint32_t)beg64 : a.ns;
  const uint32_t end = (beg64 + (uint64_t)my_rounds * WARM_THREADS < a.ns) ? (uint32_t)(beg64 + (uint64_t)my_rounds * WARM_THREADS) : a.ns;
  const int sy = g.nx, sz = g.nx * g.ny;
  uint32_t nfar = 0;

  auto transform = [&](const float4 s4, float& qx, float& qy, float& qz) {
    qx = __fadd_rn(__fadd_rn(__fmul_rn(T[0], s4.x), __fadd_rn(__fmul_rn(T[3], s4.y), __fmul_rn(T[6], s4.z))), T[9]);
    qy = __fadd_rn(__fadd_rn(__fmul_rn(T[1], s4.x), __fadd_rn(__fmul_rn(T[4], s4.y), __fmul_rn(T[7], s4.z))), T[10]);
    qz = __fadd_rn(__fadd_rn(__fmul_rn(T[2], s4.x), __fadd_rn(__fmul_rn(T[5], s4.y), __fmul_rn(T[8], s4.z))), T[11]);
  };

  // SYM -- the symmetric objective (transform_estimation.hpp:705-706; accumulate_pair's sym branch on the first Gauss-Newton step, where the inner
  // transform is the identity): the plane terms' normal is n_dst + R n_src, the source normal streamed with the query, every operation rounded
  // as the streaming pass rounds it.
  auto sym_normal = [&](const float4 nd, const float4 sn) -> float4 {
    const float t0 = __fadd_rn(__fmul_rn(T[0], sn.x), __fadd_rn(__fmul_rn(T[3], sn.y), __fmul_rn(T[6], sn.z)));
    const float t1 = __fadd_rn(__fmul_rn(T[1], sn.x), __fadd_rn(__fmul_rn(T[4], sn.y), __fmul_rn(T[7], sn.z)));
    const float t2 = __fadd_rn(__fmul_rn(T[2], sn.x), __fadd_rn(__fmul_rn(T[5], sn.y), __fmul_rn(T[8], sn.z)));
    return make_float4(__fadd_rn(nd.x, t0), __fadd_rn(nd.y, t1), __fadd_rn(nd.z, t2), 0.f);
  };
  // rank update of the wave's 16x16 tile with one round of (up to) 64 correspondences (k_search_tiled, step 5)
  // (two halves: the terms z of the wave's correspondences -> LDS; then LDS -> f64 operands -> the matrix cores.  Between them
  //  a round's streamed registers are dead, which is where the streaming loop requests the data of the round after next.)
  auto z_to_lds = [&](bool has, float qx, float qy, float qz, const float4 pm, const float4 nm) {
    if (AFF) { aff_record<NRM>(has, qx, qy, qz, pm, nm, dmn, smt, zb + lane * AFF_REC); return; }
    float z[16];
    fused_z<ACC>(has, qx, qy, qz, pm, nm, a.dst_mean, smt, z);
    if (DUAL) {
      float4* w4 = reinterpret_cast<float4*>(zb + lane * 8 + (lane >= 32 ? 16 : 0));
      w4[0] = make_float4(z[0], z[1], z[2], z[3]);
      w4[1] = make_float4(z[4], z[5], z[6], z[7]);
    } else {
      float2* w2 = reinterpret_cast<float2*>(zb + lane * NC);
#pragma unroll
      for (int c = 0; c < NC / 2; ++c) w2[c] = make_float2(z[2 * c], z[2 * c + 1]);
    }
  };
  auto lds_to_mfma = [&]() {
    __builtin_amdgcn_wave_barrier();
    if (AFF) {
      aff_mfma_round(zb, lane, afl, acc);
    } else if (DUAL) {
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
  };
  auto rank_update = [&](bool has, float qx, float qy, float qz, const float4 pm, const float4 nm) {
    z_to_lds(has, qx, qy, qz, pm, nm);
    lds_to_mfma();
  };

  // The search of the queries their margin did not settle.  A listed query comes with a BOUND: the squared distance R2 from its
  // new position to its old match (a real target point: the nearest one is no farther), or the radius without one.  Every target
  // point inside the ball of radius sqrt(R2) + extra around the query is evaluated (extra = a quarter of a cell) -- the cells the
  // ball does not reach are skipped -- keeping the best key, the best point and the two smallest squared distances: the best is
  // the exact match (it lies inside the ball), and every other target point is at least min(second smallest, sqrt(R2) + extra)
  // away: the query leaves with a fresh margin key of up to a quarter of a cell.  Two LEVELS, each run over the wave's list 64
  // entries at a time, what level A cannot take packed to the front of the list for level B:
  //   A: the ball lies inside the 3x3x3 block around the query's cell: nine rows, each clipped to the cells the ball reaches -- the
  //      18 run boundaries leave together, then one trip per eight candidates, one for the match's normal;
  //   B: the 5x5x5 block slab by slab, rows beyond sqrt(best so far) + extra skipped, then (rarely) the shell search with the
  //      same margin.
  // Stores the match, its record and the key.
  const float extra = a.warm_extra * g.cell;
  auto slow_finish = [&](bool v, uint32_t i, float qx, float qy, float qz, NN& best, float4& bp, float key, float4& pm, float4& nm) -> bool {
    const bool has = v && best.pos != NONE_U32;
    // (option "tie_rule": the two smallest distances of the search were equal -- the reference's pick among the points at that distance.
    //  The key stays: every point but the match is at least the match's own distance away, whichever of them the match is.)
    if (a.tie.mode != 0 && has && best.tie != 0u) {
      const uint32_t w = tie_settle(g, a.tie, qx, qy, qz, best.pos, __uint_as_float((uint32_t)(best.key >> 32)));
      if (w != best.pos) { best.pos = w; bp = g.pts[w]; }
    }
    nm = Z4;
    if (NRM) nm = g.nrm[has ? best.pos : 0u];      // (unconditional: one trip for the whole wave)
    pm = has ? make_float4(bp.x, bp.y, bp.z, 0.f) : Z4;
    if (!has) nm = Z4;
    if (v) {
      a.nn_pos[i] = best.pos;
      a.warm_rec[i] = make_float4(pm.x, pm.y, pm.z, key);
      if (NRM) a.warm_rec_n[i] = F3{nm.x, nm.y, nm.z};
    }
    return has;
  };
  struct SlowGeom { int cx, cy, cz; float ux, uy, uz; bool inner, inside; uint32_t cid; };
  auto slow_geom = [&](bool v, float qx, float qy, float qz) -> SlowGeom {
    SlowGeom s;
    const float BIG = 1.0e9f;
    const float fx = fminf(fmaxf((qx - g.ox) * g.inv_cell, -BIG), BIG), fy = fminf(fmaxf((qy - g.oy) * g.inv_cell, -BIG), BIG),
                fz = fminf(fmaxf((qz - g.oz) * g.inv_cell, -BIG), BIG);
    s.cx = (int)floorf(fx); s.cy = (int)floorf(fy); s.cz = (int)floorf(fz);
    s.inner = v & (s.cx >= 1) & (s.cx <= g.nx - 2) & (s.cy >= 1) & (s.cy <= g.ny - 2) & (s.cz >= 1) & (s.cz <= g.nz - 2);
    s.inside = v & (s.cx >= 0) & (s.cx < g.nx) & (s.cy >= 0) & (s.cy < g.ny) & (s.cz >= 0) & (s.cz < g.nz);
    s.ux = qx - (g.ox + (float)s.cx * g.cell); s.uy = qy - (g.oy + (float)s.cy * g.cell); s.uz = qz - (g.oz + (float)s.cz * g.cell);
    s.cid = ((uint32_t)s.cz * (uint32_t)g.ny + (uint32_t)s.cy) * (uint32_t)g.nx + (uint32_t)s.cx;
    return s;
  };
  // level A: returns whether it took the query (the ball fits the 3x3x3 block)
  auto slow_levelA = [&](bool v, float qx, float qy, float qz, float R2, NN& best, float4& bp, float& key) -> bool {
    const SlowGeom s = slow_geom(v, qx, qy, qz);
    const float Rr = __fsqrt_rn(R2) * 1.000001f + extra;       // the ball's radius, rounded up
    const float R2c = Rr * Rr * 1.000001f;
    float b = INFINITY;      // faces of the 3x3x3 block that still have cells beyond them
    if (s.cx - 1 > 0) b = fminf(b, s.ux);
    if (s.cx + 2 < g.nx) b = fminf(b, g.cell - s.ux);
    if (s.cy - 1 > 0) b = fminf(b, s.uy);
    if (s.cy + 2 < g.ny) b = fminf(b, g.cell - s.uy);
    if (s.cz - 1 > 0) b = fminf(b, s.uz);
    if (s.cz + 2 < g.nz) b = fminf(b, g.cell - s.uz);
    // (a query in the grid's OUTER layer is taken too: the rows and cells of its block that lie outside the grid do not exist -- every
    //  target point is inside the grid --, they are skipped; 2.7 % of the queries of a 220^3 grid, which used to go to the shells)
    const bool fits = s.inside && (b == INFINITY || Rr < (fmaxf(b, 0.0f) + g.cell - 2.0f * g.margin) * 0.999999f);
    // (addresses valid for every lane: a lane that is not taken reads the rows of cell (1,1,1) and is masked afterwards, so that
    //  the loads leave together instead of one exec-masked group after the other)
    const int c0 = fits ? (int)s.cid : sz + sy + 1;
    const float gm[3] = {fmaxf(s.uz - g.margin, 0.0f), 0.0f, fmaxf(g.cell - s.uz - g.margin, 0.0f)};
    const float gn[3] = {fmaxf(s.uy - g.margin, 0.0f), 0.0f, fmaxf(g.cell - s.uy - g.margin, 0.0f)};
    const float gxl = fmaxf(s.ux - g.margin, 0.0f), gxr = fmaxf(g.cell - s.ux - g.margin, 0.0f);
    uint32_t rb9[9], re9[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const float gyz2 = gm[r / 3] * gm[r / 3] + gn[r % 3] * gn[r % 3];
      const bool exists = (unsigned)(s.cz + r / 3 - 1) < (unsigned)g.nz && (unsigned)(s.cy + r % 3 - 1) < (unsigned)g.ny;
      const bool take = fits && exists && gyz2 * KSHRINK <= R2c;
      const bool left = take && s.cx > 0 && (gyz2 + gxl * gxl) * KSHRINK <= R2c, right = take && s.cx + 1 < g.nx && (gyz2 + gxr * gxr) * KSHRINK <= R2c;
      const int row = take ? c0 + (r / 3 - 1) * sz + (r % 3 - 1) * sy : sz + sy + 1;
      const uint32_t va = g.cell_start[row - (left ? 1 : 0)], vb2 = g.cell_start[row + 1 + (right ? 1 : 0)];
      rb9[r] = take ? va : 0u; re9[r] = take ? vb2 : 0u;
    }
    best.key = ((unsigned long long)__float_as_uint(a.max_sq) << 32);
    best.pos = NONE_U32;
    if (fits) {
      float a1 = INFINITY, b2 = INFINITY;
      scan_runs_track2<9>(g.pts, rb9, re9, qx, qy, qz, best, a1, b2, bp);
      best.tie = (best.pos != NONE_U32 && b2 == __uint_as_float((uint32_t)(best.key >> 32))) ? 1u : 0u;
      // every point that was not evaluated lies beyond the ball
      key = margin_key(best.pos != NONE_U32, best.pos != NONE_U32 ? b2 : a1, (__fsqrt_rn(R2) + extra) * 0.999999f, mref);
    }
    return fits;
  };
  // level B: settles every query it is given
  auto slow_levelB = [&](bool v, float qx, float qy, float qz, float R2, NN& best, float4& bp, float& key) {
    const SlowGeom s = slow_geom(v, qx, qy, qz);
    // (the bound enters as a key with a placeholder index that loses every tie: the old match itself lies inside what is scanned
    //  and is met again with its own)
    best.key = ((unsigned long long)__float_as_uint(fminf(R2, a.max_sq)) << 32) | 0xFFFFFFFFull;
    best.pos = NONE_U32;
    float a1 = INFINITY, b2 = INFINITY;
    bool proven = false;
    if (s.inner) {
      // the 5x5x5 block, one z-slab at a time from the middle outwards: five rows (runs of five x-adjacent cells, clipped to the
      // grid), a row skipped when its gap exceeds sqrt(best so far) + extra
      const int xa = max(s.cx - 2, 0), xb = min(s.cx + 2, g.nx - 1);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int dzz = (k == 0) ? 0 : (k & 1) ? (k + 1) / 2 : -(k / 2);      // 0, +1, -1, +2, -2
        const int z = s.cz + dzz;
        const bool zin = z >= 0 && z < g.nz;
        const float zl = g.oz + (float)z * g.cell;
        const float gz = axis_gap(qz, zl, zl + g.cell, g.margin);
        const float lim = (__fsqrt_rn(__uint_as_float((uint32_t)(best.key >> 32))) + extra) * 1.000001f;
        const float lim2 = lim * lim * 1.000001f;
        uint32_t rb5[5], re5[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const int y = s.cy + r - 2;
          const bool on = zin && y >= 0 && y < g.ny;
          const float yl = g.oy + (float)y * g.cell;
          const float gy = axis_gap(qy, yl, yl + g.cell, g.margin);
          const bool take = on && (gz * gz + gy * gy) * KSHRINK <= lim2;
          const uint32_t row = take ? ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx : 0u;
          const uint32_t va = g.cell_start[row + (uint32_t)xa], vb2 = g.cell_start[row + (uint32_t)xb + 1u];
          rb5[r] = take ? va : 0u; re5[r] = take ? vb2 : 0u;
        }
        scan_runs_track2<5>(g.pts, rb5, re5, qx, qy, qz, best, a1, b2, bp);
      }
      float b = INFINITY;      // faces of the 5x5x5 block that still have cells beyond them
      if (s.cx - 2 > 0) b = fminf(b, s.ux);
      if (s.cx + 3 < g.nx) b = fminf(b, g.cell - s.ux);
      if (s.cy - 2 > 0) b = fminf(b, s.uy);
      if (s.cy + 3 < g.ny) b = fminf(b, g.cell - s.uy);
      if (s.cz - 2 > 0) b = fminf(b, s.uz);
      if (s.cz + 3 < g.nz) b = fminf(b, g.cell - s.uz);
      if (b != INFINITY) b = fmaxf(b, 0.0f) + 2.0f * g.cell - 2.0f * g.margin;
      const float lim = (__fsqrt_rn(__uint_as_float((uint32_t)(best.key >> 32))) + extra) * 1.000001f;
      proven = b > 0.0f && lim * lim * 1.000001f < b * b * KSHRINK;      // everything outside the block lies beyond sqrt(best) + extra as well
    }
    bool skip = false;
    if (v && !proven) {
      // the shell search with the same margin (from the 5x5x5 block's result for an inner cell).  Counted: the host goes back to
      // the tiled kernels when these are many.
      ++nfar;
      const bool inside = (s.cx >= 0) & (s.cx < g.nx) & (s.cy >= 0) & (s.cy < g.ny) & (s.cz >= 0) & (s.cz < g.nz);
      int s0 = s.inner ? 3 : 0;
      if (!inside) {
        const float gx = axis_gap(qx, g.ox, g.ox + (float)g.nx * g.cell, g.margin), gy = axis_gap(qy, g.oy, g.oy + (float)g.ny * g.cell, g.margin),
                    gz = axis_gap(qz, g.oz, g.oz + (float)g.nz * g.cell, g.margin);
        skip = (gx * gx + gy * gy + gz * gz) * KSHRINK >= a.max_sq;      // farther than the radius from the whole grid
        s0 = max(0, max(max(-s.cx, s.cx - (g.nx - 1)), max(max(-s.cy, s.cy - (g.ny - 1)), max(-s.cz, s.cz - (g.nz - 1)))));
        if (skip) { best.pos = NONE_U32; key = margin_key(false, INFINITY, __fsqrt_rn((gx * gx + gy * gy + gz * gz) * KSHRINK) * 0.999999f, mref); }   // every target point lies inside the grid
      }
      if (!skip) nn_search_shells_margin(g, qx, qy, qz, s.cx, s.cy, s.cz, s0, best, a1, b2, bp, extra);
    }
    best.tie = (best.pos != NONE_U32 && b2 == __uint_as_float((uint32_t)(best.key >> 32))) ? 1u : 0u;
    if (!skip) {
      // every point that was not evaluated lies beyond sqrt(best) + extra (the radius + extra without a match)
      const float reach = (__fsqrt_rn(__uint_as_float((uint32_t)(best.key >> 32))) * 0.999999f + extra) * 0.999999f;
      key = margin_key(best.pos != NONE_U32, best.pos != NONE_U32 ? b2 : a1, reach, mref);
    }
  };

  uint32_t qcount = 0;      // (wave-uniform)
  // One round of the streaming loop for the query whose data has arrived: transform, the margin test, what a settled
  // query stores, the list entry of an unsettled one, the rank update.
  // The margin test: the record's key says that when the match p was established every OTHER target point was at
  // least |key| - (motion clock then) away from the query; the query has moved by at most (motion clock now) - (then) since, so
  // every other point is still at least mrg = |key| - Dk away -- if p is strictly nearer than that it is THE nearest target
  // point (ties excluded by the strictness), and nothing is looked at: not even the query's cell.  A negative key is the same
  // bound for a query WITHOUT a match, over all target points: if mrg still exceeds the radius there is still none.
  // lbv / s2 (REC 1): the key the search left in nn_lb (a.lb_valid) and the old match's nearest-other-point table entry -- any other
  // target point p' has |q - p'| >= nnd(p) - |q - p|: a second lower bound, the larger key wins.
  auto round = [&](uint32_t i, bool valid, const F3 s3c, uint32_t w, float4 pm, float4 nm, float keyv, float s2, const float4 sn = make_float4(0.f, 0.f, 0.f, 0.f)) {
    float qx, qy, qz;
    transform(make_float4(s3c.x, s3c.y, s3c.z, 0.f), qx, qy, qz);
    const float e_old = d2_pinned(qx, qy, qz, pm.x, pm.y, pm.z);
    if (REC != 2) {
      if (w != NONE_U32) {
        const float alt = margin_key(true, INFINITY, __fsub_rn(__fsqrt_rn(fmaxf(s2, 0.0f)) * 0.999999f, __fsqrt_rn(e_old) * 1.000001f), mref);
        keyv = fmaxf(a.lb_valid ? fmaxf(keyv, 0.0f) : 0.0f, alt);
      } else {
        keyv = a.lb_valid ? fminf(keyv, MARGIN_NONE_NO_MATCH) : MARGIN_NONE_NO_MATCH;
      }
    }
    const float mrg = __fsub_rn(fabsf(keyv), Dk);
    const float m2 = mrg * mrg * KSHRINK;
    const bool ok = valid && mrg > 0.0f;
    const bool shas = ok && keyv > 0.0f && e_old < m2 && e_old < a.max_sq;
    const bool settled = shas || (ok && keyv < 0.0f && m2 >= a.max_sq);
    if (settled) {
      if (REC != 2 && a.nn_pos != a.warm_pos) a.nn_pos[i] = w;
      if (REC == 1) { a.warm_rec[i] = make_float4(pm.x, pm.y, pm.z, keyv); if (NRM) a.warm_rec_n[i] = F3{nm.x, nm.y, nm.z}; }
    }
    const bool todo = valid && !settled;
    const unsigned long long um = __ballot(todo);
    if (todo) {
      // the list entry and its bound: the squared distance to the old match if it has one inside the radius, else the radius
      const uint32_t o = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(um >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)um, 0u));
      wq[o] = make_float4(qx, qy, qz, __uint_as_float(i));
      wr[o] = ((REC == 2 ? !(keyv < 0.0f) : w != NONE_U32) && e_old < a.max_sq) ? e_old : a.max_sq;
    }
    qcount += (uint32_t)__popcll(um);
    z_to_lds(shas, qx, qy, qz, pm, SYM ? sym_normal(nm, sn) : nm);
  };
  // What a round streams in.  REC 2: the 12-byte copy of the source point and the match record {point, margin key} {normal}
  // -- 40 B per query (28 without normals), all of it coalesced, TWO rounds in flight per wave (sets A and B; vmcnt retires in
  // order, so the wait for A leaves B's loads flying).  REC 1: the sorted source record, the stored position and the search's
  // margin key one round, the gathers through that position (old match, its normal, its table entry) the next: a three-stage
  // pipeline with one wait per round for loads that were issued a whole round earlier.
  uint32_t base = beg, qlisted = 0;
  F3 sA = F3{0.f, 0.f, 0.f}, nA = F3{0.f, 0.f, 0.f}, sB = F3{0.f, 0.f, 0.f}, nB = F3{0.f, 0.f, 0.f};
  float4 rA = Z4, rB = Z4;
  float4 snA = Z4, snB = Z4;      // (SYM) the queries' source normals
  float4 sn1 = Z4, sn2 = Z4;
  uint32_t iA = beg + threadIdx.x, iB = iA + WARM_THREADS;
  // (REC 0 / 1) stage 1 -> 2: source record + position of the round after next; stage 2 -> 3: what was gathered for the next round
  uint32_t w1 = NONE_U32, w2 = NONE_U32;
  float l1 = 0.0f, l2 = 0.0f;
  F3 s1 = F3{0.f, 0.f, 0.f}, s2_ = F3{0.f, 0.f, 0.f};
  float4 gp = Z4, gn = Z4;
  float gs = -1.0f;
  // (the three loads leave in THIS order everywhere -- scheduling barriers -- : the wait for a set is computed from the
  //  position of its loads in the in-order vmcnt queue, merged over all paths into the loop)
  auto load2 = [&](uint32_t k, F3& sv, float4& rv, F3& nv, float4& snv) {
    __builtin_amdgcn_sched_barrier(0);
    sv = a.warm_src3[k];
    __builtin_amdgcn_sched_barrier(0);
    rv = a.warm_rec[k];
    __builtin_amdgcn_sched_barrier(0);
    if (NRM) nv = a.warm_rec_n[k];
    __builtin_amdgcn_sched_barrier(0);
    if (SYM) { snv = a.src_nrm[k]; __builtin_amdgcn_sched_barrier(0); }
  };
  auto load1 = [&](uint32_t k, F3& sv, uint32_t& wv, float& lv, float4& snv) { const float4 t4 = a.src[k]; sv = F3{t4.x, t4.y, t4.z}; wv = a.warm_pos[k]; lv = a.nn_lb[k]; if (SYM) snv = a.src_nrm[k]; };
  // (every load of the streaming loop is UNCONDITIONAL, from an index clamped into the chunk / a position clamped into the
  //  target: a load under a divergent branch may or may not have been issued as far as the compiler's vmcnt bookkeeping
  //  is concerned, and the waits it then inserts drain the younger prefetches as well)
  const uint32_t last = end > beg ? end - 1u : 0u;
  auto gather = [&](uint32_t wv) {
    const uint32_t wc = wv != NONE_U32 ? wv : 0u;
    if (NRM && g.pn != nullptr) { gp = g.pn[2 * (size_t)wc]; gn = g.pn[2 * (size_t)wc + 1]; }      // (uniform) point and normal from one 32-byte record
    else { gp = g.pts[wc]; if (NRM) gn = g.nrm[wc]; }
    gs = a.safe2[wc];
    if (wv == NONE_U32) { gp = gn = Z4; gs = -1.0f; }
  };
  WARM_CLK(0);
  for (;;) {
  // (the pipeline is filled HERE, at every entry of the streaming loop -- also after a list that had to be searched early:
  //  registers with loads in flight must not live across that search, where they would be spilled and reloaded)
  // (unconditionally: an empty chunk reads element 0, which exists)
  if (REC == 2) {
    load2(min(iA, last), sA, rA, nA, snA);
    load2(min(iB, last), sB, rB, nB, snB);
  } else {
    load1(min(iA, last), s2_, w2, l2, sn2);         // next round: record, then (dependent) its gathers
    gather(w2);
    load1(min(iA + WARM_THREADS, last), s1, w1, l1, sn1);      // the round after: record
  }
  // stream rounds until the chunk is done -- or the wave's list could not take two more rounds' queries (a source far from
  // alignment lists most of them): then the list is searched first
  if (REC == 2) {
    for (; base < end && qcount <= (uint32_t)(WARM_QCAP - 128); base += 2 * WARM_THREADS) {
      // (a set's registers are consumed -- down to the terms in LDS -- BEFORE the set is requested again, so that the new
      //  loads can land in the same registers: no copy at the loop's end that would have to wait for them)
      round(iA, iA < end, sA, NONE_U32, make_float4(rA.x, rA.y, rA.z, 0.f), make_float4(nA.x, nA.y, nA.z, 0.f), rA.w, 0.0f, snA);
      __builtin_amdgcn_sched_barrier(0);
      iA += 2 * WARM_THREADS;
      load2(min(iA, last), sA, rA, nA, snA);
      __builtin_amdgcn_sched_barrier(0);
      lds_to_mfma();
      round(iB, iB < end, sB, NONE_U32, make_float4(rB.x, rB.y, rB.z, 0.f), make_float4(nB.x, nB.y, nB.z, 0.f), rB.w, 0.0f, snB);
      __builtin_amdgcn_sched_barrier(0);
      iB += 2 * WARM_THREADS;
      load2(min(iB, last), sB, rB, nB, snB);
      __builtin_amdgcn_sched_barrier(0);
      lds_to_mfma();
    }
  } else {
    for (; base < end && qcount <= (uint32_t)(WARM_QCAP - 64); base += WARM_THREADS) {
      const uint32_t i = iA;
      const F3 sc = s2_;
      const uint32_t wc = w2;
      const float4 pc = gp, nc = gn;
      const float gc = gs, lc = l2;
      const float4 snc = sn2;
      // next round: its record has arrived, its gathers leave now; the round after: its record leaves now
      iA += WARM_THREADS;
      s2_ = s1; w2 = w1; l2 = l1; sn2 = sn1;
      gather(w2);
      load1(min(iA + WARM_THREADS, last), s1, w1, l1, sn1);
      round(i, i < end, sc, wc, pc, nc, lc, gc, snc);
      lds_to_mfma();
    }
  }
  // the listed queries: two levels, 64 entries per round, what level A cannot take packed to the front of the list for level B
  __builtin_amdgcn_wave_barrier();
  WARM_CLK(1);
  qlisted += qcount;
  for (int level = 0; level < 2 && qcount != 0u; ++level) {
    uint32_t nopen = 0;
    for (uint32_t b0 = 0; b0 < qcount; b0 += 64u) {
      const bool v = b0 + (uint32_t)lane < qcount;
      const uint32_t e = min(b0 + (uint32_t)lane, (uint32_t)(WARM_QCAP - 1));      // (lanes beyond the list: masked by v)
      const float4 ent = wq[e];
      const float R2 = wr[e];
      NN best;
      float4 bp = Z4;
      float key = 0.0f;
      bool taken = true;
      if (level == 0) taken = slow_levelA(v, ent.x, ent.y, ent.z, R2, best, bp, key);
      else slow_levelB(v, ent.x, ent.y, ent.z, R2, best, bp, key);
      const bool open = v && !taken;
      const unsigned long long om = __ballot(open);
#ifdef CILHIP_EXP_PHASE_CLOCKS
      if (level == 0) {      // who is left open by level A: all / bound = the radius / bound beyond a cell
        if (v) atomicAdd(&g_warm_clk[6], 1ull);
        if (open) atomicAdd(&g_warm_clk[7], 1ull);
        if (open && R2 >= a.max_sq) atomicAdd(&g_warm_clk[14], 1ull);
        if (open && R2 < a.max_sq && R2 > g.cell * g.cell) atomicAdd(&g_warm_clk[15], 1ull);
      }
#endif
      // (this round's entries are in registers: the front of the list up to b0 + 64 is free)
      if (open) {
        const uint32_t o = nopen + __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u));
        wq[o] = ent; wr[o] = R2;
      }
      nopen += (uint32_t)__popcll(om);
      float4 pm = Z4, nm = Z4;
      const bool has = slow_finish(v && taken, __float_as_uint(ent.w), ent.x, ent.y, ent.z, best, bp, key, pm, nm);
      if (SYM) { const float4 snl = a.src_nrm[v ? __float_as_uint(ent.w) : 0u]; nm = sym_normal(nm, snl); }      // (after slow_finish stored the TARGET's normal in the record)
      rank_update(has, ent.x, ent.y, ent.z, pm, nm);
      WARM_CLK(2 + level);
    }
    qcount = nopen;
    __builtin_amdgcn_wave_barrier();
  }
  qcount = 0;
  __builtin_amdgcn_wave_barrier();
  if (base >= end) break;
  }
  if (a.unproven_cnt) {
    const double tot = wave_sum((double)nfar);
    if (lane == 0 && tot > 0.0) atomicAdd(a.unproven_cnt + (vb & 63u), (uint32_t)tot);
    if (lane == 0 && qlisted != 0u) atomicAdd(a.unproven_cnt + 64u + ((vb * WARM_WAVES + (uint32_t)wave) & 63u), qlisted);   // listed queries: is the form paying?
  }
  if (AFF) {
    aff_write_row<WARM_WAVES>(raw, wave, lane, acc, a.partials + (size_t)vb * AFF_ROW);
    return;
  }
  double* const db = reinterpret_cast<double*>(raw + wave * FUSED_WAVE_BYTES);
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
      for (int w = 0; w < WARM_WAVES; ++w) {
        const double* dw = reinterpret_cast<const double*>(raw + w * FUSED_WAVE_BYTES);
        v1 += dw[e1];
        if (DUAL) v1 += dw[e1b];
        if (i2 >= 0) { v2 += dw[e2]; if (DUAL) v2 += dw[e2b]; }
      }
    }
    a.partials[(size_t)vb * SUMS_MAX + threadIdx.x] = v1 - v2;
  }
  WARM_CLK(5);
#ifdef CILHIP_EXP_PHASE_CLOCKS
  if (threadIdx.x == 0 && blockIdx.x < 4096u) g_warm_stamp[blockIdx.x][1] = wall_clock64();
  if (lane == 0 && blockIdx.x < 4096u && qlisted) atomicAdd(&g_warm_stamp[blockIdx.x][2], (unsigned long long)qlisted);
#endif
}

#undef Z4
template <int ACC>
static void launch_warm_m(const IterArgs& a, int rec, int nblocks, hipStream_t s) {
  const dim3 g(nblocks), b(WARM_THREADS);
  const hipEvent_t ev_start = g_ev_start, ev_stop = g_ev_stop;      // (armed by set_launch_events: consumed here)
  g_ev_start = g_ev_stop = nullptr;
  // (source normals in the arguments = the symmetric objective: the plane-term forms stream them with the queries)
  if (FusedZ<ACC>::plane && ACC != IM_AFFC && a.src_nrm != nullptr) {
    if (rec == 2) launch_ev((k_warm<(FusedZ<ACC>::plane ? ACC : IM_PLANE), 2, true>), g, b, s, ev_start, ev_stop, a);
    else launch_ev((k_warm<(FusedZ<ACC>::plane ? ACC : IM_PLANE), 1, true>), g, b, s, ev_start, ev_stop, a);
    return;
  }
  if (rec == 2) launch_ev((k_warm<ACC, 2>), g, b, s, ev_start, ev_stop, a);
  else launch_ev((k_warm<ACC, 1>), g, b, s, ev_start, ev_stop, a);
}
int warm_num_blocks(uint32_t ns) {
  // ONE generation of blocks (4 resident per CU: registers, LDS): every wave searches its list once, at the end of its chunk --
  // with more, shorter blocks those latency-bound tails take wave slots from the streaming ones (measured: 2048 / 4096 / 8192
  // blocks 0.106 / 0.124 / 0.142 ms at 10M)
  static const long exp_nb = [] { const char* e = getenv("CILHIP_EXP_WARM_BLOCKS"); return e ? atol(e) : 0L; }();
  // Small clouds: at least eight rounds per wave (a block's fixed costs -- pipeline fill, list search, row -- against its share
  // of the stream), and 64 blocks are few enough for the epilogue to fold their rows itself, without the stage-1 kernel
  // (measured per step: 100k points 64 blocks 0.0317 ms, 392 blocks 0.0350; 1M points 512 blocks 0.0511, 1024 blocks 0.0530).
  if (exp_nb > 0) return (int)((exp_nb + 7) & ~7L);
  long nb = (long)ns / (8 * WARM_THREADS);
  if (nb > 1024) nb = 1024;
  if (nb < 64) nb = 64;
  return (int)((nb + 7) & ~7L);
}
// (squared distances are NOT written by this form -- a store inside the streaming loop shares the in-order vmcnt counter with
//  the prefetched loads; nobody reads them in the configurations that run warm-started (no post-filters, no weight
//  evaluators), and launch_fill_d2 recomputes them from the stored matches on demand)
void launch_warm(const IterArgs& a, int metric, int rec, int nblocks, hipStream_t s) {
  switch (metric) {
    case IM_KABSCH: launch_warm_m<IM_KABSCH>(a, rec, nblocks, s); break;
    case IM_PLANE: launch_warm_m<IM_PLANE>(a, rec, nblocks, s); break;
    case IM_POINT: launch_warm_m<IM_POINT>(a, rec, nblocks, s); break;
    case IM_AFFC: launch_warm_m<IM_AFFC>(a, rec, nblocks, s); break;
    case IM_AFFP: launch_warm_m<IM_AFFP>(a, rec, nblocks, s); break;
    default: launch_warm_m<IM_BOTH>(a, rec, nblocks, s); break;
  }
}

// the 12-byte copy of the sorted source the record-reading warm kernel streams (once per sort of a source)
__global__ void k_copy_src3(const float4* __restrict__ src, uint32_t ns, F3* __restrict__ out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) { const float4 v = src[i]; out[i] = F3{v.x, v.y, v.z}; }
}
__global__ void k_interleave_pn(const float4* __restrict__ pts, const float4* __restrict__ nrm, uint32_t n, float4* __restrict__ pn) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { pn[2 * (size_t)i] = pts[i]; pn[2 * (size_t)i + 1] = nrm[i]; }
}
void launch_interleave_pn(const float4* pts, const float4* nrm, uint32_t n, float4* pn, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_interleave_pn, dim3((n + 255u) / 256u < 8192u ? (n + 255u) / 256u : 8192u), dim3(256), 0, s, pts, nrm, n, pn);
}
void launch_copy_src3(const float4* src_sorted, uint32_t ns, F3* out, hipStream_t s) {
  if (ns) hipLaunchKernelGGL(k_copy_src3, dim3((int)((ns + 255u) / 256u < 4096u ? (ns + 255u) / 256u : 4096u)), dim3(256), 0, s, src_sorted, ns, out);
}

}  // namespace cilhip
