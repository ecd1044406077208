// Microbenchmark: issue rate of the vector-ALU forms the kernels choose between on gfx950:
// scalar fp32 FMA, packed fp32 FMA, the bf16 dot product with fp32 accumulate, and the two
// 16-bit -> fp32 unpack operations (shift / mask).  Register-only (one store at the end).
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_bench valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kIters = 1 << 14;
constexpr int kAcc = 8;  // independent accumulators per lane

template <int MODE>
__global__ __launch_bounds__(256) void bench(float* out, float seed) {
  float a[2 * kAcc];
  for (int i = 0; i < 2 * kAcc; ++i) a[i] = seed * (threadIdx.x + i);
  const float w = seed + 1.0f;
  unsigned u = __float_as_uint(seed) + threadIdx.x;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < kAcc; ++i) {
      if constexpr (MODE == 0) {  // 2 scalar FMAs
        a[2 * i] = fmaf(w, a[2 * i], seed);
        a[2 * i + 1] = fmaf(w, a[2 * i + 1], seed);
      } else if constexpr (MODE == 1) {  // 1 packed FMA = the same 2 FMAs
        float2v x = {a[2 * i], a[2 * i + 1]};
        const float2v ww = {w, w}, ss = {seed, seed};
        x = __builtin_elementwise_fma(ww, x, ss);
        a[2 * i] = x.x;
        a[2 * i + 1] = x.y;
      } else if constexpr (MODE == 2) {  // 2 bf16 dot products (4 multiply-adds)
        const bf16x2 p = __builtin_bit_cast(bf16x2, u + i), q = __builtin_bit_cast(bf16x2, u ^ 0x3f803f80u);
        a[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(p, q, a[2 * i], false);
        a[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(q, p, a[2 * i + 1], false);
      } else {  // unpack a bf16 pair (shift, mask) and add: 4 ops
        const unsigned v = __float_as_uint(a[2 * i]) + u;
        a[2 * i] += __uint_as_float(v << 16);
        a[2 * i + 1] += __uint_as_float(v & 0xffff0000u);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2 * kAcc; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double ops_per_iter) {
  const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  float* out;
  if (hipMalloc(&out, blocks * 256 * sizeof(float)) != hipSuccess) return;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  bench<MODE><<<blocks, 256>>>(out, 1e-3f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  bench<MODE><<<blocks, 256>>>(out, 1e-3f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr = double(blocks) * 4 * kIters * ops_per_iter;
  printf("%-28s %.3f ms  %.2f wave-instr/clk/CU (2.4 GHz)  -> %.2f clk per wave-instr per SIMD\n", name,
         ms, wave_instr / (ms * 1e-3) / 256 / 2.4e9, 4.0 / (wave_instr / (ms * 1e-3) / 256 / 2.4e9));
  hipFree(out);
}

int main() {
  run<0>("v_fma_f32 x2", 2.0 * kAcc);
  run<1>("v_pk_fma_f32 x1", 1.0 * kAcc);
  run<2>("v_dot2_f32_bf16 x2", 2.0 * kAcc);
  run<3>("unpack+add (4 ops)", 4.0 * kAcc);
  return 0;
}
