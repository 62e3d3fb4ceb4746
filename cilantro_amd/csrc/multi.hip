// multi.hip -- one process, several devices: the C entry of the sharded loops (cilhip_multi_*), split from c_api.hip.
#include "../../include/cilantro_hip/c_api.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "internal.hpp"
#include "rccl_api.hpp"

using namespace cilhip;

// =====================================================================================================================
// One process, several devices: the sharded protocols of DESIGN.md section 8 driven from C ("devices[]").
// One context + stream per device; per iteration every context enqueues its partial sums (cilhip_icp_partial_sums), the 48
// f64 are all-reduced ON THE DEVICES' STREAMS -- RCCL's ncclAllReduce (xGMI between the GPUs of a node), the library opened at
// run time so that libcilantro_hip.so itself does not depend on it -- and every context applies the same sums
// (cilhip_icp_apply_sums): identical transforms and convergence decisions everywhere, no host arithmetic in the loop.
// Partitions: 0 = the source in contiguous shards, the target on every device; 1 = spatial slabs of target (+ halo) and source
// with the device-side guard and re-partitioning.  Several shards on ONE device (devices[] repeating an
// ordinal: tests on a single GPU) reduce through a kernel instead of RCCL.

namespace {

// out[r][k] = sum over shards of in[s][k], the same order on every shard (all buffers on one device)
__global__ void k_sum_shards(double* const* bufs, int n) {
  const int k = threadIdx.x;
  if (k >= SUMS_MAX) return;      // (every k is independent: no barrier)
  double v = 0.0;
  for (int s = 0; s < n; ++s) v += bufs[s][k];
  for (int s = 0; s < n; ++s) bufs[s][k] = v;
}
// partitioning A on one device: out[r][i] = min over shards of in[s][i]  (what ncclAllReduce(ncclUint64, ncclMin) does between devices)
__global__ void k_min_shards(unsigned long long* const* bufs, int n, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long v = bufs[0][i];
    for (int s = 1; s < n; ++s) { const unsigned long long w = bufs[s][i]; v = w < v ? w : v; }
    for (int s = 0; s < n; ++s) bufs[s][i] = v;
  }
}
}  // namespace

struct cilhip_multi {
  int n = 0;
  std::vector<int> dev;
  std::vector<cilhip_ctx*> ctx;
  std::vector<double*> d_sums;
  bool distinct = true;            // all ordinals different: RCCL; otherwise the same-device reduction
  RcclApi rccl;
  std::vector<rccl_comm_t> comms;
  double** d_bufs = nullptr;       // (same-device reduction) the shards' sum buffers
  // partitioning A (index shards of the target): per shard the packed (d2, global index) keys and the traversal keys of one iteration
  std::vector<unsigned long long*> d_keys, d_okeys;
  unsigned long long** d_kbufs = nullptr;      // (same-device reduction) [2 n]: the shards' key buffers, then their traversal-key buffers
  std::vector<hipEvent_t> ev;
  std::string err;
  // the clouds (host copies: slabs are cut again when the guard fires)
  std::vector<float> dst, dstn, src;
  size_t nd = 0, ns = 0;
  float max_sq = 0.0f;
  int partition = 0;
  // slab partition
  int axis = 0;
  double halo = 0.0, slack = 0.0;
  std::vector<double> bounds;
  float T_part[16];
  float src_center[3] = {0, 0, 0}, src_half[3] = {0, 0, 0}, gdm[3] = {0, 0, 0}, gsm[3] = {0, 0, 0};
  int repartitions = 0;
  double slack_opt = -1.0;         // cilhip_multi_set_slab_slack (< 0: twice the search radius)
  std::vector<size_t> n_dst_local, n_src_local;
  // option "tie_rule" of the shards: the order among exactly equidistant nearest points is a property of the WHOLE target (the tree
  // the reference builds over it): built once from the host copy when some shard's search first meets a tie, every shard is handed
  // the entries of its own points (slabs: through the global index of each local point)
  cilhip_tie_order* order = nullptr;
  std::vector<std::vector<uint32_t>> gidx;      // slabs: per shard, global index of its target point i
  bool tie_pending = false;                     // some shard's counters showed ties met without tables before they were reset (a re-partition inside a run)
  // host time the shards' enqueue calls took in the last run (per iteration and shard, microseconds): with one host thread per shard
  // (multi_iterate) it is what bounds an iteration whose kernels take tens of microseconds, not its sum over the shards
  double host_us_per_iter_shard = 0.0;
  bool threads = true;                          // CILHIP_MULTI_THREADS=0: one host thread walks the shards (round 4)
};

static int mfail(cilhip_multi* m, int code, const std::string& msg) { if (m) m->err = msg; return code; }
static int multi_upload_fwd(cilhip_multi* m);
static void multi_free_keys(cilhip_multi* m) {
  for (size_t r = 0; r < m->d_keys.size(); ++r) {
    (void)hipSetDevice(m->dev[r]);
    if (m->d_keys[r]) (void)hipFree(m->d_keys[r]);
    if (m->d_okeys[r]) (void)hipFree(m->d_okeys[r]);
  }
  m->d_keys.clear(); m->d_okeys.clear();
  if (m->d_kbufs) { (void)hipSetDevice(m->dev[0]); (void)hipFree(m->d_kbufs); m->d_kbufs = nullptr; }
}
#define MCK(m, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return mfail((m), CILHIP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define MCTX(m, r, call) do { const int rc_ = (call); if (rc_ != CILHIP_OK) return mfail((m), rc_, std::string(#call) + ": " + cilhip_last_error((m)->ctx[r])); } while (0)

extern "C" {

int cilhip_multi_create(cilhip_multi** out, const int* devices, int ndev) {
  if (!out || !devices || ndev <= 0 || ndev > 64) return CILHIP_ERR_INVALID;
  *out = nullptr;
  cilhip_multi* m = new (std::nothrow) cilhip_multi();
  if (!m) return CILHIP_ERR_HIP;
  m->n = ndev;
  m->dev.assign(devices, devices + ndev);
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < i; ++j) if (devices[i] == devices[j]) m->distinct = false;
  { const char* e = getenv("CILHIP_MULTI_THREADS"); m->threads = !(e && e[0] == '0'); }
  m->ctx.assign(ndev, nullptr); m->d_sums.assign(ndev, nullptr);
  m->n_dst_local.assign(ndev, 0); m->n_src_local.assign(ndev, 0);
  int rc = CILHIP_OK;
  for (int r = 0; r < ndev && rc == CILHIP_OK; ++r) {
    rc = cilhip_create(&m->ctx[r], devices[r]);
    if (rc == CILHIP_OK && (hipSetDevice(devices[r]) != hipSuccess || hipMalloc(&m->d_sums[r], SUMS_MAX * sizeof(double)) != hipSuccess)) rc = CILHIP_ERR_HIP;
  }
  // (CILHIP_MULTI_FORCE_RCCL=1: a single shard goes through RCCL too -- a communicator of one rank: what a one-GPU box can check of that path)
  const bool force_rccl = ndev == 1 && getenv("CILHIP_MULTI_FORCE_RCCL") != nullptr && atoi(getenv("CILHIP_MULTI_FORCE_RCCL")) != 0;
  if (rc == CILHIP_OK && (ndev > 1 || force_rccl)) {
    if (m->distinct) {
      if (!m->rccl.load()) rc = CILHIP_ERR_UNSUPPORTED;      // several devices need RCCL (librccl.so.1)
      else {
        m->comms.assign(ndev, nullptr);
        if (m->rccl.CommInitAll(m->comms.data(), ndev, devices) != 0) rc = CILHIP_ERR_HIP;
      }
    } else {
      for (int r = 1; r < ndev; ++r) if (devices[r] != devices[0]) rc = CILHIP_ERR_UNSUPPORTED;   // (repeated ordinals: all shards on one device)
      if (rc == CILHIP_OK && (hipSetDevice(devices[0]) != hipSuccess || hipMalloc(&m->d_bufs, ndev * sizeof(double*)) != hipSuccess ||
                              hipMemcpy(m->d_bufs, m->d_sums.data(), ndev * sizeof(double*), hipMemcpyHostToDevice) != hipSuccess))
        rc = CILHIP_ERR_HIP;
      for (int r = 0; r < ndev && rc == CILHIP_OK; ++r) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = CILHIP_ERR_HIP; else m->ev.push_back(e); }
    }
  }
  if (rc != CILHIP_OK) { cilhip_multi_destroy(m); return rc; }
  *out = m;
  return CILHIP_OK;
}

void cilhip_multi_destroy(cilhip_multi* m) {
  if (!m) return;
  for (size_t r = 0; r < m->comms.size(); ++r) if (m->comms[r] && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(m->comms[r]);
  for (int r = 0; r < m->n; ++r) {
    if (m->d_sums[r]) { (void)hipSetDevice(m->dev[r]); (void)hipFree(m->d_sums[r]); }
    if (m->ctx[r]) cilhip_destroy(m->ctx[r]);
  }
  if (m->d_bufs) (void)hipFree(m->d_bufs);
  multi_free_keys(m);
  for (hipEvent_t e : m->ev) (void)hipEventDestroy(e);
  cilhip_tie_order_destroy(m->order);
  delete m;
}

const char* cilhip_multi_last_error(const cilhip_multi* m) { return m ? m->err.c_str() : "null handle"; }
cilhip_ctx* cilhip_multi_context(cilhip_multi* m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[rank] : nullptr; }
int cilhip_multi_repartitions(const cilhip_multi* m) { return m ? m->repartitions : 0; }
int cilhip_multi_set_slab_slack(cilhip_multi* m, float slack) {
  if (!m) return CILHIP_ERR_INVALID;
  m->slack_opt = slack;
  // (halos are sized when the clouds are cut: with clouds already set the slabs are cut again now)
  if (m->partition == 1 && (m->nd || m->ns)) {
    const double r = std::isfinite(m->max_sq) ? std::sqrt((double)m->max_sq) : 0.0;
    if (slack >= 0.0f && std::isfinite(m->max_sq)) { m->slack = slack; m->halo = r + m->slack; return multi_upload_fwd(m); }
  }
  return CILHIP_OK;
}
int cilhip_multi_shard_sizes(const cilhip_multi* m, int rank, size_t* n_target, size_t* n_source) {
  if (!m || rank < 0 || rank >= m->n) return CILHIP_ERR_INVALID;
  if (n_target) *n_target = m->n_dst_local[rank];
  if (n_source) *n_source = m->n_src_local[rank];
  return CILHIP_OK;
}

}  // extern "C"

// f64 mean rounded to f32: what the ICP classes hold as dst_mean_ / src_mean_ of the WHOLE clouds
static void global_mean(const std::vector<float>& xyz, size_t n, float out[3]) {
  double s[3] = {0, 0, 0};
  for (size_t i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) s[c] += (double)xyz[3 * i + c];
  for (int c = 0; c < 3; ++c) out[c] = n ? (float)(s[c] / (double)n) : 0.0f;
}

// uploads every shard's clouds under the current partition (slabs: cut under T_part)
static int multi_upload(cilhip_multi* m);
static int multi_upload_fwd(cilhip_multi* m) { return multi_upload(m); }
static int multi_upload(cilhip_multi* m) {
  const int n = m->n;
  if (m->partition != 2) multi_free_keys(m);
  if (m->partition == 0) {
    for (int r = 0; r < n; ++r) {
      MCTX(m, r, cilhip_set_shard_info(m->ctx[r], 0, nullptr, nullptr));      // (not an index shard, whatever the handle held before)
      const size_t base = m->ns / n, rem = m->ns % n;
      const size_t lo = r * base + std::min<size_t>(r, rem), hi = lo + base + ((size_t)r < rem ? 1 : 0);
      MCTX(m, r, cilhip_set_target(m->ctx[r], m->dst.data(), m->dstn.empty() ? nullptr : m->dstn.data(), m->nd, CILHIP_MEM_HOST));
      if (m->order) MCTX(m, r, cilhip_load_tie_order(m->ctx[r], m->order, nullptr));
      MCTX(m, r, cilhip_set_source(m->ctx[r], m->src.data() + 3 * lo, hi - lo, CILHIP_MEM_HOST));
      MCTX(m, r, cilhip_set_slab_guard(m->ctx[r], -1, 0.0f, nullptr, nullptr, nullptr));
      m->n_dst_local[r] = m->nd; m->n_src_local[r] = hi - lo;
    }
    return CILHIP_OK;
  }
  if (m->partition == 2) {
    // index shards of the TARGET (SURVEY 8(e) partitioning A; distributed.py TargetShardedRigidICP: the same cut): shard r holds the
    // target points [lo_r, hi_r) and ALL source points; per iteration MIN of the packed keys, then SUM of the partial sums
    multi_free_keys(m);
    m->d_keys.assign(n, nullptr); m->d_okeys.assign(n, nullptr);
    m->gidx.assign(n, std::vector<uint32_t>());
    static const float no_points[3] = {0.0f, 0.0f, 0.0f};
    for (int r = 0; r < n; ++r) {
      const size_t base = m->nd / n, rem = m->nd % n;
      const size_t lo = r * base + std::min<size_t>(r, rem), hi = lo + base + ((size_t)r < rem ? 1 : 0);
      MCTX(m, r, cilhip_set_target(m->ctx[r], hi > lo ? m->dst.data() + 3 * lo : no_points, m->dstn.empty() ? nullptr : (hi > lo ? m->dstn.data() + 3 * lo : no_points), hi - lo, CILHIP_MEM_HOST));
      m->gidx[r].resize(hi - lo);
      for (size_t i = lo; i < hi; ++i) m->gidx[r][i - lo] = (uint32_t)i;
      if (m->order) MCTX(m, r, cilhip_load_tie_order(m->ctx[r], m->order, m->gidx[r].empty() ? nullptr : m->gidx[r].data()));
      MCTX(m, r, cilhip_set_source(m->ctx[r], m->src.data(), m->ns, CILHIP_MEM_HOST));
      MCTX(m, r, cilhip_set_shard_info(m->ctx[r], lo, m->gdm, m->gsm));
      MCTX(m, r, cilhip_set_slab_guard(m->ctx[r], -1, 0.0f, nullptr, nullptr, nullptr));
      MCK(m, hipSetDevice(m->dev[r]));
      MCK(m, hipMalloc(&m->d_keys[r], (m->ns ? m->ns : 1) * sizeof(unsigned long long)));
      MCK(m, hipMalloc(&m->d_okeys[r], (m->ns ? m->ns : 1) * sizeof(unsigned long long)));
      m->n_dst_local[r] = hi - lo; m->n_src_local[r] = m->ns;
    }
    if (!m->distinct && n > 1) {
      std::vector<unsigned long long*> both(m->d_keys);
      both.insert(both.end(), m->d_okeys.begin(), m->d_okeys.end());
      MCK(m, hipSetDevice(m->dev[0]));
      MCK(m, hipMalloc(&m->d_kbufs, both.size() * sizeof(unsigned long long*)));
      MCK(m, hipMemcpy(m->d_kbufs, both.data(), both.size() * sizeof(unsigned long long*), hipMemcpyHostToDevice));
    }
    return CILHIP_OK;
  }
  // slabs along m->axis: rank r owns the source points whose image under T_part lies in [b_r, b_r+1) and holds the target points in
  // [b_r - halo, b_r+1 + halo)   (distributed.py SlabPartition: the same cut)
  const int ax = m->axis;
  std::vector<double> q(m->ns);
  for (size_t i = 0; i < m->ns; ++i)
    q[i] = (double)m->src[3 * i] * (double)m->T_part[0 * 4 + ax] + (double)m->src[3 * i + 1] * (double)m->T_part[1 * 4 + ax] +
           (double)m->src[3 * i + 2] * (double)m->T_part[2 * 4 + ax] + (double)m->T_part[12 + ax];
  m->bounds.assign(n + 1, 0.0);
  m->bounds[0] = -INFINITY; m->bounds[n] = INFINITY;
  if (n > 1 && m->ns) {      // boundaries at the source's quantiles: the queries are the work
    std::vector<double> qs(q);
    for (int r = 1; r < n; ++r) {
      const size_t k = std::min(m->ns - 1, (size_t)((double)m->ns * r / n));
      std::nth_element(qs.begin(), qs.begin() + k, qs.end());
      m->bounds[r] = qs[k];
    }
  }
  std::vector<float> d, dn, s;
  m->gidx.assign(n, std::vector<uint32_t>());
  // (whether the cloud HAS normals is a property of the whole cloud: a shard whose slab + halo holds no target point must still
  //  take the point-to-plane branch of the epilogue like every other shard -- the all-reduced sums are the same everywhere)
  static const float no_points[3] = {0.0f, 0.0f, 0.0f};
  for (int r = 0; r < n; ++r) {
    const double b0 = m->bounds[r], b1 = m->bounds[r + 1];
    d.clear(); dn.clear(); s.clear();
    std::vector<uint32_t>& gi = m->gidx[r];
    for (size_t i = 0; i < m->nd; ++i) {
      const double x = (double)m->dst[3 * i + ax];
      if (x >= b0 - m->halo && x < b1 + m->halo) {
        d.insert(d.end(), m->dst.begin() + 3 * i, m->dst.begin() + 3 * i + 3);
        if (!m->dstn.empty()) dn.insert(dn.end(), m->dstn.begin() + 3 * i, m->dstn.begin() + 3 * i + 3);
        gi.push_back((uint32_t)i);
      }
    }
    for (size_t i = 0; i < m->ns; ++i)
      if (q[i] >= b0 && q[i] < b1) s.insert(s.end(), m->src.begin() + 3 * i, m->src.begin() + 3 * i + 3);
    MCTX(m, r, cilhip_set_target(m->ctx[r], d.empty() ? no_points : d.data(), m->dstn.empty() ? nullptr : (dn.empty() ? no_points : dn.data()), d.size() / 3, CILHIP_MEM_HOST));
    if (m->order) MCTX(m, r, cilhip_load_tie_order(m->ctx[r], m->order, gi.empty() ? nullptr : gi.data()));
    MCTX(m, r, cilhip_set_source(m->ctx[r], s.data(), s.size() / 3, CILHIP_MEM_HOST));
    MCTX(m, r, cilhip_set_shard_info(m->ctx[r], 0, m->gdm, nullptr));
    MCTX(m, r, cilhip_set_slab_guard(m->ctx[r], ax, (float)m->slack, m->src_center, m->src_half, m->T_part));
    m->n_dst_local[r] = d.size() / 3; m->n_src_local[r] = s.size() / 3;
  }
  return CILHIP_OK;
}

extern "C" {

static int multi_set_clouds_impl(cilhip_multi* m, const float* dst_xyz, const float* dst_nrm, size_t nd, const float* src_xyz, size_t ns, float max_sq_dist,
                            int partition, const float* T_part);
int cilhip_multi_set_clouds(cilhip_multi* m, const float* dst_xyz, const float* dst_nrm, size_t nd, const float* src_xyz, size_t ns, float max_sq_dist,
                            int partition, const float* T_part) {
  if (!m) return CILHIP_ERR_INVALID;
  // (host copies of whole clouds: an allocation failure must not cross the C boundary)
  try { return multi_set_clouds_impl(m, dst_xyz, dst_nrm, nd, src_xyz, ns, max_sq_dist, partition, T_part); }
  catch (const std::bad_alloc&) { return mfail(m, CILHIP_ERR_HIP, "multi_set_clouds: out of host memory"); }
  catch (...) { return mfail(m, CILHIP_ERR_HIP, "multi_set_clouds: unexpected exception"); }
}
static int multi_set_clouds_impl(cilhip_multi* m, const float* dst_xyz, const float* dst_nrm, size_t nd, const float* src_xyz, size_t ns, float max_sq_dist,
                            int partition, const float* T_part) {
  if (!m || (nd && !dst_xyz) || (ns && !src_xyz) || partition < 0 || partition > 2) return CILHIP_ERR_INVALID;
  if (partition == 2 && nd > 0xFFFFFFF0ull) return CILHIP_ERR_INVALID;
  m->dst.assign(dst_xyz, dst_xyz + 3 * nd);
  if (dst_nrm) m->dstn.assign(dst_nrm, dst_nrm + 3 * nd); else m->dstn.clear();
  m->src.assign(src_xyz, src_xyz + 3 * ns);
  m->nd = nd; m->ns = ns; m->max_sq = max_sq_dist; m->partition = partition;
  memcpy(m->T_part, T_part ? T_part : kIdentity16, sizeof(m->T_part));
  global_mean(m->dst, nd, m->gdm);
  global_mean(m->src, ns, m->gsm);
  if (partition == 1) {
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (size_t i = 0; i < nd; ++i)
      for (int c = 0; c < 3; ++c) { const double v = m->dst[3 * i + c]; if (i == 0 || v < lo[c]) lo[c] = v; if (i == 0 || v > hi[c]) hi[c] = v; }
    m->axis = 0;
    for (int c = 1; c < 3; ++c) if (hi[c] - lo[c] > hi[m->axis] - lo[m->axis]) m->axis = c;
    const double r = std::isfinite(max_sq_dist) ? std::sqrt((double)max_sq_dist) : hi[m->axis] - lo[m->axis];
    m->slack = m->slack_opt >= 0.0 ? m->slack_opt : 2.0 * r; m->halo = r + m->slack;
    float slo[3] = {0, 0, 0}, shi[3] = {0, 0, 0};
    for (size_t i = 0; i < ns; ++i)
      for (int c = 0; c < 3; ++c) { const float v = m->src[3 * i + c]; if (i == 0 || v < slo[c]) slo[c] = v; if (i == 0 || v > shi[c]) shi[c] = v; }
    for (int c = 0; c < 3; ++c) { m->src_center[c] = 0.5f * (slo[c] + shi[c]); m->src_half[c] = std::max(shi[c] - m->src_center[c], m->src_center[c] - slo[c]) * 1.000001f; }
  }
  m->repartitions = 0;
  cilhip_tie_order_destroy(m->order); m->order = nullptr; m->tie_pending = false;      // (belongs to the previous target)
  return multi_upload(m);
}

}  // extern "C"

// the all-reduce of the shards' 48 partial sums, on the shards' streams
static int multi_allreduce(cilhip_multi* m) {
  if (m->n == 1 && m->comms.empty()) return CILHIP_OK;
  if (m->distinct) {
    if (m->rccl.GroupStart() != 0) return mfail(m, CILHIP_ERR_HIP, "ncclGroupStart");
    for (int r = 0; r < m->n; ++r)
      if (m->rccl.AllReduce(m->d_sums[r], m->d_sums[r], SUMS_MAX, RCCL_DOUBLE, RCCL_SUM, m->comms[r], cilhip::ctx_stream(m->ctx[r])) != 0) return mfail(m, CILHIP_ERR_HIP, "ncclAllReduce");
    if (m->rccl.GroupEnd() != 0) return mfail(m, CILHIP_ERR_HIP, "ncclGroupEnd");
    return CILHIP_OK;
  }
  // one device: shard 0's stream waits for the others' partial sums, sums all buffers in shard order, the others wait for it
  MCK(m, hipSetDevice(m->dev[0]));
  for (int r = 1; r < m->n; ++r) { MCK(m, hipEventRecord(m->ev[r], cilhip::ctx_stream(m->ctx[r]))); MCK(m, hipStreamWaitEvent(cilhip::ctx_stream(m->ctx[0]), m->ev[r], 0)); }
  hipLaunchKernelGGL(k_sum_shards, dim3(1), dim3(64), 0, cilhip::ctx_stream(m->ctx[0]), m->d_bufs, m->n);
  MCK(m, hipEventRecord(m->ev[0], cilhip::ctx_stream(m->ctx[0])));
  for (int r = 1; r < m->n; ++r) MCK(m, hipStreamWaitEvent(cilhip::ctx_stream(m->ctx[r]), m->ev[0], 0));
  return CILHIP_OK;
}

// partitioning A: the MIN all-reduce of one 64-bit key per source point, in place (which = 0: the packed (d2, global index) keys, 1: the
// traversal keys of the tie order)
static int multi_reduce_keys(cilhip_multi* m, int which) {
  if (m->n == 1 && m->comms.empty()) return CILHIP_OK;
  std::vector<unsigned long long*>& buf = which ? m->d_okeys : m->d_keys;
  if (m->ns == 0) return CILHIP_OK;
  if (m->distinct) {
    if (m->rccl.GroupStart() != 0) return mfail(m, CILHIP_ERR_HIP, "ncclGroupStart");
    for (int r = 0; r < m->n; ++r)
      if (m->rccl.AllReduce(buf[r], buf[r], m->ns, RCCL_UINT64, RCCL_MIN, m->comms[r], cilhip::ctx_stream(m->ctx[r])) != 0) return mfail(m, CILHIP_ERR_HIP, "ncclAllReduce(min)");
    if (m->rccl.GroupEnd() != 0) return mfail(m, CILHIP_ERR_HIP, "ncclGroupEnd");
    return CILHIP_OK;
  }
  MCK(m, hipSetDevice(m->dev[0]));
  for (int r = 1; r < m->n; ++r) { MCK(m, hipEventRecord(m->ev[r], cilhip::ctx_stream(m->ctx[r]))); MCK(m, hipStreamWaitEvent(cilhip::ctx_stream(m->ctx[0]), m->ev[r], 0)); }
  const int nb = (int)std::min<size_t>((m->ns + 255) / 256, 4096);
  hipLaunchKernelGGL(k_min_shards, dim3(nb), dim3(256), 0, cilhip::ctx_stream(m->ctx[0]), m->d_kbufs + (which ? m->n : 0), m->n, m->ns);
  MCK(m, hipEventRecord(m->ev[0], cilhip::ctx_stream(m->ctx[0])));
  for (int r = 1; r < m->n; ++r) MCK(m, hipStreamWaitEvent(cilhip::ctx_stream(m->ctx[r]), m->ev[0], 0));
  return CILHIP_OK;
}

// `iters` iterations of partitioning A: every shard's keys, their MIN; with the whole target's tie order loaded (some search met
// exactly equidistant nearest points) the traversal keys of the matches at the winning distance and THEIR MIN; every shard's sums over
// the pairs it won, the all-reduce of the 48 f64, every shard's epilogue.  One host thread per shard as in multi_iterate.
static int multi_iterate_keys(cilhip_multi* m, int iters) {
  const int n = m->n;
  const bool ordered = m->order != nullptr;
  auto shard_step = [&](int r, int step) -> int {
    switch (step) {
      case 0: return cilhip_icp_partial_keys(m->ctx[r], reinterpret_cast<uint64_t*>(m->d_keys[r]));
      case 1: return cilhip_icp_order_keys(m->ctx[r], reinterpret_cast<const uint64_t*>(m->d_keys[r]), reinterpret_cast<uint64_t*>(m->d_okeys[r]));
      case 2: return ordered ? cilhip_icp_sums_from_ordered_keys(m->ctx[r], reinterpret_cast<const uint64_t*>(m->d_keys[r]), reinterpret_cast<const uint64_t*>(m->d_okeys[r]), m->d_sums[r])
                             : cilhip_icp_sums_from_keys(m->ctx[r], reinterpret_cast<const uint64_t*>(m->d_keys[r]), m->d_sums[r]);
      default: return cilhip_icp_apply_sums(m->ctx[r], m->d_sums[r]);
    }
  };
  auto joint_step = [&](int step) -> int { return step == 0 ? multi_reduce_keys(m, 0) : (step == 1 ? multi_reduce_keys(m, 1) : (step == 2 ? multi_allreduce(m) : CILHIP_OK)); };
  m->host_us_per_iter_shard = 0.0;
  if (n == 1 || !m->threads) {
    for (int k = 0; k < iters; ++k)
      for (int step = 0; step < 4; ++step) {
        if (step == 1 && !ordered) continue;
        for (int r = 0; r < n; ++r) MCTX(m, r, shard_step(r, step));
        { const int rc = joint_step(step); if (rc) return rc; }
      }
    return CILHIP_OK;
  }
  std::atomic<int> arrived{0}, generation{0}, failed{0};
  std::vector<int> rcs(n, CILHIP_OK);
  int reduce_rc = CILHIP_OK;
  auto rendezvous = [&]() {
    const int gen = generation.load(std::memory_order_acquire);
    if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) { arrived.store(0, std::memory_order_relaxed); generation.fetch_add(1, std::memory_order_release); }
    else for (unsigned spins = 0; generation.load(std::memory_order_acquire) == gen; ++spins) cpu_relax(spins);
  };
  auto worker = [&](int r) {
    (void)hipSetDevice(m->dev[r]);
    for (int k = 0; k < iters; ++k)
      for (int step = 0; step < 4; ++step) {
        if (step == 1 && !ordered) continue;
        if (!failed.load(std::memory_order_relaxed)) { const int rc = shard_step(r, step); if (rc) { rcs[r] = rc; failed.store(1); } }
        if (step == 3) break;      // (the epilogue: the next iteration's search follows on the same stream)
        rendezvous();
        if (r == 0 && !failed.load()) { const int rc = joint_step(step); if (rc) { reduce_rc = rc; failed.store(1); } }
        rendezvous();
      }
  };
  std::vector<std::thread> th;
  for (int r = 1; r < n; ++r) th.emplace_back(worker, r);
  worker(0);
  for (auto& x : th) x.join();
  (void)hipSetDevice(m->dev[0]);
  for (int r = 0; r < n; ++r) if (rcs[r]) return mfail(m, rcs[r], std::string("shard ") + std::to_string(r) + ": " + cilhip_last_error(m->ctx[r]));
  return reduce_rc;
}

// `iters` iterations of {every shard's partial sums, the all-reduce of the 48 f64, every shard's epilogue}.  One host thread PER SHARD
// (each enqueues on its own device's stream: at 8 devices x ~4 launches x ~5 us one thread walking the shards would bound an
// iteration whose kernels take ~15 us), two rendezvous per iteration around the all-reduce, which one thread issues for all (RCCL
// group call over the distinct devices / the same-device kernel).  An error on any shard is carried to the end: nobody leaves a
// rendezvous early.
static int multi_iterate(cilhip_multi* m, int iters) {
  if (iters <= 0) return CILHIP_OK;
  if (m->partition == 2) return multi_iterate_keys(m, iters);
  const auto t_begin = std::chrono::steady_clock::now();
  if (m->n == 1 || !m->threads) {
    double host = 0.0;
    for (int k = 0; k < iters; ++k) {
      const auto t0 = std::chrono::steady_clock::now();
      double w0 = 0.0, w1 = 0.0;
      for (int r = 0; r < m->n; ++r) w0 += cilhip::ctx_wait_us(m->ctx[r]);
      for (int r = 0; r < m->n; ++r) MCTX(m, r, cilhip_icp_partial_sums(m->ctx[r], m->d_sums[r]));
      for (int r = 0; r < m->n; ++r) w1 += cilhip::ctx_wait_us(m->ctx[r]);
      host += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() - (w1 - w0);
      { const int rc = multi_allreduce(m); if (rc) return rc; }
      const auto t1 = std::chrono::steady_clock::now();
      for (int r = 0; r < m->n; ++r) MCTX(m, r, cilhip_icp_apply_sums(m->ctx[r], m->d_sums[r]));
      host += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
    }
    m->host_us_per_iter_shard = host / iters / m->n;
    return CILHIP_OK;
  }
  const int n = m->n;
  std::atomic<int> arrived{0}, generation{0}, failed{0};
  std::vector<int> rcs(n, CILHIP_OK);
  std::vector<double> host(n, 0.0);
  int reduce_rc = CILHIP_OK;
  auto rendezvous = [&]() {
    const int gen = generation.load(std::memory_order_acquire);
    if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) { arrived.store(0, std::memory_order_relaxed); generation.fetch_add(1, std::memory_order_release); }
    else for (unsigned spins = 0; generation.load(std::memory_order_acquire) == gen; ++spins) cpu_relax(spins);
  };
  auto worker = [&](int r) {
    (void)hipSetDevice(m->dev[r]);
    for (int k = 0; k < iters; ++k) {
      if (!failed.load(std::memory_order_relaxed)) {
        const auto t0 = std::chrono::steady_clock::now();
        const double w0 = cilhip::ctx_wait_us(m->ctx[r]);
        const int rc = cilhip_icp_partial_sums(m->ctx[r], m->d_sums[r]);
        host[r] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() - (cilhip::ctx_wait_us(m->ctx[r]) - w0);      // (waiting for the device's published state is not enqueue work)
        if (rc) { rcs[r] = rc; failed.store(1); }
      }
      rendezvous();
      if (r == 0 && !failed.load()) { reduce_rc = multi_allreduce(m); if (reduce_rc) failed.store(1); }
      rendezvous();
      if (!failed.load(std::memory_order_relaxed)) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = cilhip_icp_apply_sums(m->ctx[r], m->d_sums[r]);
        host[r] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (rc) { rcs[r] = rc; failed.store(1); }
      }
    }
  };
  std::vector<std::thread> th;
  for (int r = 1; r < n; ++r) th.emplace_back(worker, r);
  worker(0);
  for (auto& x : th) x.join();
  (void)hipSetDevice(m->dev[0]);
  for (int r = 0; r < n; ++r) if (rcs[r]) return mfail(m, rcs[r], std::string("shard ") + std::to_string(r) + ": " + cilhip_last_error(m->ctx[r]));
  if (reduce_rc) return reduce_rc;
  double mx = 0.0;
  for (int r = 0; r < n; ++r) mx = std::max(mx, host[r]);
  m->host_us_per_iter_shard = mx / iters;      // (the slowest shard's thread: what an iteration waits for on the host side)
  (void)t_begin;
  return CILHIP_OK;
}

extern "C" {

int cilhip_multi_last_host_time(const cilhip_multi* m, double* us_per_iteration_per_shard) {
  if (!m || !us_per_iteration_per_shard) return CILHIP_ERR_INVALID;
  *us_per_iteration_per_shard = m->host_us_per_iter_shard;
  return CILHIP_OK;
}

// IterativeClosestPointBase::estimate() (registration/icp_base.hpp:68-87) across the handle's devices.  check_every: how often the
// loop state is read back (convergence; the slab guard) -- 0: the default 5.
static int multi_icp_run_once(cilhip_multi* m, const cilhip_icp_params* p, const float* T0, int check_every, cilhip_icp_result* out);
// ties met by some shard's searches while no order tables were loaded (option "tie_rule" 2)?
static int multi_ties_pending(cilhip_multi* m, bool* pending) {
  *pending = m->tie_pending;
  for (int r = 0; r < m->n && !*pending; ++r) {
    cilhip_tie_order_info ti{};
    MCTX(m, r, cilhip_get_tie_order_info(m->ctx[r], &ti));
    if (!ti.loaded && ti.pending != 0) *pending = true;
  }
  return CILHIP_OK;
}
static int multi_icp_run_impl(cilhip_multi* m, const cilhip_icp_params* p, const float* T0, int check_every, cilhip_icp_result* out);
int cilhip_multi_icp_run(cilhip_multi* m, const cilhip_icp_params* p, const float* T0, int check_every, cilhip_icp_result* out) {
  if (!m) return CILHIP_ERR_INVALID;
  try { return multi_icp_run_impl(m, p, T0, check_every, out); }
  catch (const std::bad_alloc&) { return mfail(m, CILHIP_ERR_HIP, "multi_icp_run: out of host memory"); }
  catch (...) { return mfail(m, CILHIP_ERR_HIP, "multi_icp_run: unexpected exception"); }
}
static int multi_icp_run_impl(cilhip_multi* m, const cilhip_icp_params* p, const float* T0, int check_every, cilhip_icp_result* out) {
  if (!m || !p || !out) return CILHIP_ERR_INVALID;
  m->tie_pending = false;
  int rc = multi_icp_run_once(m, p, T0, check_every, out);
  if (rc || m->order) return rc;
  bool pending = false;
  rc = multi_ties_pending(m, &pending);
  if (rc || !pending) return rc;
  // the reference's order over the WHOLE target, once; every shard gets the entries of its points; the run is executed again from T0
  rc = cilhip_tie_order_create(m->dst.data(), m->nd, &m->order);
  if (rc) return mfail(m, rc, "tie_rule: building the order tables of the whole target failed");
  for (int r = 0; r < m->n; ++r)
    MCTX(m, r, cilhip_load_tie_order(m->ctx[r], m->order, (m->partition != 0 && !m->gidx[r].empty()) ? m->gidx[r].data() : nullptr));
  m->tie_pending = false;
  return multi_icp_run_once(m, p, T0, check_every, out);
}
static int multi_icp_run_once(cilhip_multi* m, const cilhip_icp_params* p, const float* T0, int check_every, cilhip_icp_result* out) {
  const int every0 = check_every > 0 ? check_every : 5;
  float T_ck[16];
  memcpy(T_ck, T0 ? T0 : kIdentity16, sizeof(T_ck));
  const size_t total = p->max_iter;
  size_t base = 0, since = 0, begin_base = 0;      // base: iterations up to the last checked state; begin_base: up to the last begin
  const float* gsm = m->n > 1 ? m->gsm : nullptr;  // (one shard: the context's own mean, as cilhip_icp_run)
  if (m->partition == 1) {
    gsm = m->gsm;
    // the slabs are exact for searches under the transform they were cut under (+- the slack the guard watches): start from T0's own cut
    if (memcmp(m->T_part, T_ck, sizeof(T_ck)) != 0) { memcpy(m->T_part, T_ck, sizeof(T_ck)); const int rc = multi_upload(m); if (rc) return rc; }
  }
  for (int r = 0; r < m->n; ++r) MCTX(m, r, cilhip_icp_begin(m->ctx[r], p, T_ck, gsm));
  bool fresh = true;                               // the partition was made under exactly T_ck
  int every = every0;
  cilhip_icp_result st{};
  memcpy(st.T, T_ck, sizeof(T_ck));
  while (base + since < total) {
    {      // up to the next look at the loop state, in one block (one host thread per shard inside it)
      const size_t to_check = (size_t)every - since % (size_t)every, left = total - base - since;
      const int blk = (int)std::min<size_t>(std::min(to_check, left), 1u << 20);
      const int rc = multi_iterate(m, blk);
      if (rc) return rc;
      since += (size_t)blk;
    }
    if (since % (size_t)every == 0 || base + since == total) {
      MCTX(m, 0, cilhip_icp_state(m->ctx[0], &st));          // (the same state on every shard: same sums, same epilogue)
      int bad = 0;
      cilhip_icp_result vs{};
      if (m->partition == 1) MCTX(m, 0, cilhip_get_slab_violation_state(m->ctx[0], &bad, &vs));
      // The flag is about the NEXT search: the update that raised it is still exact (its search ran inside the halos), so every
      // iteration up to and including it is kept (distributed.py SlabShardedRigidICP.estimate: the same bookkeeping).
      if (bad && vs.iterations > 0) {
        memcpy(T_ck, vs.T, sizeof(T_ck)); base = begin_base + vs.iterations; since = 0; fresh = false;
        if (vs.last_delta_norm < p->conv_tol || base >= total) { *out = vs; out->iterations = base; return CILHIP_OK; }
      } else if (!bad || (fresh && since == 1)) {
        memcpy(T_ck, st.T, sizeof(T_ck)); base += since; since = 0; fresh = false;
        if (st.last_delta_norm < p->conv_tol || base >= total) { *out = st; out->iterations = begin_base + st.iterations; return CILHIP_OK; }
        every = bad ? 1 : every0;
      }
      if (bad) {
        if (!m->order) { bool pend = false; const int prc = multi_ties_pending(m, &pend); if (prc) return prc; m->tie_pending = pend; }      // (the next begin resets the counters)
        memcpy(m->T_part, T_ck, sizeof(T_ck));
        { const int rc = multi_upload(m); if (rc) return rc; }
        ++m->repartitions;
        for (int r = 0; r < m->n; ++r) MCTX(m, r, cilhip_icp_begin(m->ctx[r], p, T_ck, m->gsm));
        since = 0; begin_base = base; fresh = true; every = 1;
      }
    }
  }
  MCTX(m, 0, cilhip_icp_state(m->ctx[0], &st));
  *out = st;
  out->iterations = begin_base + st.iterations;
  return CILHIP_OK;
}

}  // extern "C"
