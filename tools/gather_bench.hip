// Ceiling of 128-byte row gathers on gfx950: groups of 8 lanes read random rows (16 B per lane)
// of a table that fits L2 / Infinity Cache / only HBM.  Prints achieved GB/s and bytes/clk/CU.
//   hipcc -O3 --offload-arch=gfx950 -o tools/gather_bench tools/gather_bench.hip && tools/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// MODE 0: uniformly random rows of the table.  MODE 1: rows within `window` rows of a base that
// slides with the group index (neighbouring groups share rows, like neighbouring queries).
template <int U, int LANES, int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ table, uint32_t rows,
                                                     uint32_t window, int iters, float* __restrict__ out) {
  const uint32_t gid = (blockIdx.x * 256 + threadIdx.x) / LANES;
  const uint32_t lane = threadIdx.x % LANES;
  constexpr int CH = 8 / LANES;  // 16-byte chunks per lane: a row is 8 chunks (128 B)
  float acc = 0.f;
  uint32_t state = mix(gid * 2654435761u + 12345u);
  const uint32_t base = MODE == 1 ? static_cast<uint32_t>((static_cast<uint64_t>(gid) * 7) % (rows - window)) : 0u;
  for (int it = 0; it < iters; ++it) {
    uint4 v[U][CH];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      state = state * 1664525u + 1013904223u;
      const uint32_t h = mix(state);
      const uint32_t r = MODE == 1 ? base + h % window : h % rows;
#pragma unroll
      for (int c = 0; c < CH; ++c) v[u][c] = table[static_cast<uint64_t>(r) * 8 + lane + c * LANES];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c)
        acc += __uint_as_float(v[u][c].x) + __uint_as_float(v[u][c].y) + __uint_as_float(v[u][c].z) + __uint_as_float(v[u][c].w);
  }
  if (acc == 123.456f) out[gid] = acc;
}

template <int U, int LANES, int MODE>
int run(const uint4* table, uint32_t rows, uint32_t window, float* out, const char* label) {
  const int blocks = 256 * 32, iters = 64;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((gather_kernel<U, LANES, MODE>), dim3(blocks), dim3(256), 0, 0, table, rows, window, iters, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((gather_kernel<U, LANES, MODE>), dim3(blocks), dim3(256), 0, 0, table, rows, window, iters, out);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double bytes = static_cast<double>(blocks) * 256 / LANES * iters * U * 128.0;
  const double gbs = bytes / (ms * 1e-3) / 1e9;
  printf("%-34s U=%d lanes/row=%d  table %8.1f MB  %7.3f ms  %8.1f GB/s  %5.1f B/clk/CU (at 2.4 GHz, 256 CUs)\n",
         label, U, LANES, rows * 128.0 / 1e6, ms, gbs, gbs * 1e9 / 2.4e9 / 256);
  return 0;
}

int main() {
  const size_t max_bytes = 2048ull << 20;
  uint4* table; float* out;
  CHECK(hipMalloc(&table, max_bytes));
  CHECK(hipMemset(table, 1, max_bytes));
  CHECK(hipMalloc(&out, 64 << 20));
  for (uint32_t mb : {2u, 32u, 2048u}) {
    const uint32_t rows = static_cast<uint32_t>((static_cast<size_t>(mb) << 20) / 128);
    if (run<4, 8, 0>(table, rows, 0, out, "random rows")) return 1;
    if (run<8, 8, 0>(table, rows, 0, out, "random rows")) return 1;
    if (run<16, 8, 0>(table, rows, 0, out, "random rows")) return 1;
    if (run<4, 4, 0>(table, rows, 0, out, "random rows, 2 chunks/lane")) return 1;
    if (run<8, 4, 0>(table, rows, 0, out, "random rows, 2 chunks/lane")) return 1;
  }
  const uint32_t rows = static_cast<uint32_t>((352ull << 20) / 128);  // the value tensor of the headline workload
  for (uint32_t window : {64u, 512u, 4096u}) {
    char label[64];
    snprintf(label, sizeof label, "sliding window of %u rows", window);
    if (run<8, 8, 1>(table, rows, window, out, label)) return 1;
    if (run<8, 4, 1>(table, rows, window, out, label)) return 1;
  }
  return 0;
}
