// Dev tool (not product code): calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts,
// in the access patterns the ICP kernels use (VERDICT r1 item 2 / MI355X_MICROARCH.md "HBM": only 16 B/lane coalesced
// reads are calibrated there; everything else has to be calibrated in one's own pattern):
//   k_cal_stream16   16 B/lane coalesced read stream                        (source / match streams of the accumulation)
//   k_cal_stream4    4 B/lane coalesced read stream                         (nn_pos)
//   k_cal_rows       row staging of the tiled search: 16 lanes per row, rows of ROWLEN 16-B records every STRIDE records
//                    (each row = one contiguous 240 B run; lines are partly used)
//   k_cal_gather16   one 16 B record per lane at a random position of a large array (matched point / normal gathers)
//   k_cal_write4 / k_cal_write16   coalesced write streams                    (nn_pos / records)
// Every array is far larger than the 256 MiB Infinity Cache and touched once per launch.  Usage (on the GPU box):
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/fetch -- tools/bin/pmc_calibrate
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/write -- tools/bin/pmc_calibrate
// The program prints the known byte counts per kernel (useful bytes, and bytes at 64 B / 128 B line granularity);
// tools/pmc_calibration_table.py joins them with the counter CSVs into profiles/r02_calibration.txt.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr size_t NREC = 64ull << 20;           // 64 Mi records of 16 B = 1 GiB
constexpr int ROWLEN = 15, STRIDE = 220;       // the bench's grid: ~15-cell rows of a 220-cell-wide grid, 1 point per cell

__global__ void k_cal_stream16(const float4* __restrict__ a, size_t n, float* sink) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = a[i]; s += v.x + v.w; }
  if (s == 123.456f) *sink = s;
}
__global__ void k_cal_stream4(const uint32_t* __restrict__ a, size_t n, float* sink) {
  uint32_t s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 0x12345u) *sink = 1.f;
}
__global__ void k_cal_rows(const float4* __restrict__ a, size_t nrows, float* sink) {
  // 16 lanes per row: lane o reads record row * STRIDE + o for o < ROWLEN
  float s = 0.f;
  const int o = threadIdx.x & 15;
  for (size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; r < nrows; r += ((size_t)gridDim.x * blockDim.x) >> 4)
    if (o < ROWLEN) { const float4 v = a[r * STRIDE + o]; s += v.y; }
  if (s == 123.456f) *sink = s;
}
__global__ void k_cal_gather16(const float4* __restrict__ a, const uint32_t* __restrict__ idx, size_t n, float* sink) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = a[idx[i]]; s += v.z; }
  if (s == 123.456f) *sink = s;
}
__global__ void k_cal_write4(uint32_t* a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (uint32_t)i;
}
__global__ void k_cal_write16(float4* a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

int main() {
  float4* a = nullptr; uint32_t* idx = nullptr; uint32_t* b = nullptr; float* sink = nullptr;
  if (hipMalloc(&a, NREC * 16) != hipSuccess || hipMalloc(&b, NREC * 4) != hipSuccess || hipMalloc(&idx, (NREC / 4) * 4) != hipSuccess ||
      hipMalloc(&sink, 4) != hipSuccess) { std::printf("no device / out of memory\n"); return 2; }
  (void)hipMemset(a, 0, NREC * 16); (void)hipMemset(b, 0, NREC * 4);
  const size_t ng = NREC / 4;                                     // 16 Mi gathers over the 64 Mi records
  {
    std::vector<uint32_t> h(ng);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % NREC); }
    (void)hipMemcpy(idx, h.data(), ng * 4, hipMemcpyHostToDevice);
  }
  const size_t nrows = (NREC - ROWLEN) / STRIDE;
  const dim3 g(8192), t(256);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_cal_stream16, g, t, 0, 0, a, NREC, sink);
    hipLaunchKernelGGL(k_cal_stream4, g, t, 0, 0, b, NREC, sink);
    hipLaunchKernelGGL(k_cal_rows, g, t, 0, 0, a, nrows, sink);
    hipLaunchKernelGGL(k_cal_gather16, g, t, 0, 0, a, idx, ng, sink);
    hipLaunchKernelGGL(k_cal_write4, g, t, 0, 0, b, NREC);
    hipLaunchKernelGGL(k_cal_write16, g, t, 0, 0, a, NREC);
  }
  if (hipDeviceSynchronize() != hipSuccess) { std::printf("launch failed\n"); return 1; }
  // known bytes.  rows: a 240 B run at a 3520 B stride starts at 3520 r mod 64 (mod 128): count the lines it touches.
  size_t rows_l64 = 0, rows_l128 = 0;
  for (size_t r = 0; r < nrows; ++r) {
    const size_t b0 = r * STRIDE * 16, b1 = b0 + ROWLEN * 16 - 1;
    rows_l64 += b1 / 64 - b0 / 64 + 1; rows_l128 += b1 / 128 - b0 / 128 + 1;
  }
  std::printf("KNOWN k_cal_stream16 read useful=%zu lines64=%zu lines128=%zu\n", NREC * 16, NREC * 16, NREC * 16);
  std::printf("KNOWN k_cal_stream4 read useful=%zu lines64=%zu lines128=%zu\n", NREC * 4, NREC * 4, NREC * 4);
  std::printf("KNOWN k_cal_rows read useful=%zu lines64=%zu lines128=%zu\n", nrows * ROWLEN * 16, rows_l64 * 64, rows_l128 * 128);
  std::printf("KNOWN k_cal_gather16 read useful=%zu lines64=%zu lines128=%zu\n", ng * 16 + ng * 4, ng * 64 + ng * 4, ng * 128 + ng * 4);
  std::printf("KNOWN k_cal_write4 write useful=%zu lines64=%zu lines128=%zu\n", NREC * 4, NREC * 4, NREC * 4);
  std::printf("KNOWN k_cal_write16 write useful=%zu lines64=%zu lines128=%zu\n", NREC * 16, NREC * 16, NREC * 16);
  return 0;
}
