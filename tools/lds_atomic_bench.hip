// Microbenchmark: LDS atomic throughput on gfx950 (ds_add_f32 vs ds_add_u32 vs read-modify-write).
// Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o lds_atomic_bench lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kIters = 4096;
constexpr int kWords = 8192;  // 32 KB

template <int MODE>
__global__ __launch_bounds__(256) void bench(unsigned* out, int pattern, unsigned long long* cycles) {
  __shared__ float accf[kWords];
  unsigned* accu = reinterpret_cast<unsigned*>(accf);
  for (int i = threadIdx.x; i < kWords; i += 256) accf[i] = 0.f;
  __syncthreads();
  unsigned idx = threadIdx.x;
  unsigned step = 256;
  if (pattern == 1) { idx = (threadIdx.x * 2654435761u) >> 19; step = 977; }   // pseudo-random, conflicts
  if (pattern == 2) { idx = threadIdx.x / 8; step = 1; }                        // 8 lanes on one address
  const unsigned long long t0 = clock64();
#pragma unroll 8
  for (int it = 0; it < kIters; ++it) {
    const unsigned a = (idx + it * step) % kWords;
    if (MODE == 0) atomicAdd(&accf[a], 1.0f);
    if (MODE == 1) atomicAdd(&accu[a], 1u);
    if (MODE == 2) { float v = accf[a]; accf[a] = v + 1.0f; }
    if (MODE == 3) { unsigned r = atomicAdd(&accu[a], 1u); idx += r & 1; }
    if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long*>(accf) + (a >> 1), 1ull << 20);
    if (MODE == 5) atomicAdd(reinterpret_cast<double*>(accf) + (a >> 1), 1.0);
    if (MODE == 6) atomicMax(&accu[a], (unsigned)it);
  }
  __syncthreads();
  const unsigned long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  unsigned s = 0;
  for (int i = threadIdx.x; i < kWords; i += 256) s += accu[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int pattern, int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu;
  unsigned* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  bench<MODE><<<blocks, 256>>>(out, pattern, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  bench<MODE><<<blocks, 256>>>(out, pattern, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto c : h) avg += c; avg /= blocks;
  const double lane_ops = double(blocks) * 256 * kIters;
  printf("%-14s pattern=%d blocks/CU=%d: %.3f ms, %.1f G lane-ops/s chip, %.2f lane-ops/clk/CU (at 2.4GHz), block cycles %.0f -> %.2f cyc per wave-instr per block\n",
         name, pattern, blocks_per_cu, ms, lane_ops / ms / 1e6, lane_ops / (ms * 1e-3) / 256 / 2.4e9, avg, avg / kIters / 4);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int bpc : {1, 4}) {
    for (int p : {0, 1, 2}) {
      run<0>("ds_add_f32", p, bpc);
      run<1>("ds_add_u32", p, bpc);
      run<2>("read+write", p, bpc);
      run<3>("ds_add_rtn_u32", p, bpc);
      run<4>("ds_add_u64", p, bpc);
      run<5>("ds_add_f64", p, bpc);
      run<6>("ds_max_u32", p, bpc);
    }
  }
  return 0;
}
