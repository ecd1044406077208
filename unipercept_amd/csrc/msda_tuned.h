// Tuned MSDeformAttn kernels for gfx950: head dims that split into 16-byte lane chunks.
//
// Work decomposition (forward and the query-side half of the backward):
//   * a group of G lanes owns one (batch, query, head) triple; each lane holds CPL consecutive
//     channels = one 16-byte chunk (CPL = 4 for fp32, 8 for bf16/fp16), so D = G * CPL and one
//     bilinear corner of one head is a single contiguous D*e-byte segment fetched by G lanes
//     with one global_load_dwordx4 each;
//   * a 256-thread workgroup owns 256/G consecutive triples of one batch element.  Phase 1: all
//     256 threads read the group's sampling locations and attention weights with coalesced
//     loads, turn every point into {4 attention-scaled corner weights, 4 byte offsets} once,
//     and stage that in LDS.  Phase 2: each group walks its points, broadcasting the staged
//     record to its G lanes with two ds_read_b128, and gathers the corners.
//   Per-point scalar work (floor, bounds, weights) is therefore done once per point instead of
//   once per channel lane, and the value gather is issued as full 16-byte-per-lane loads.
#pragma once

#include <initializer_list>

#include "msda_common.h"

namespace msda {

constexpr int kNotTuned = INT32_MIN;

template <typename VT>
struct Chunk;  // 16 bytes of VT as CPL floats

template <>
struct Chunk<float> {
  static constexpr int kCPL = 4;
  __device__ __forceinline__ static void load(const void* p, float (&v)[4]) {
    const float4 t = *static_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ static void store(void* p, const float (&v)[4]) {
    *static_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

template <>
struct Chunk<bf16_t> {
  static constexpr int kCPL = 8;
  __device__ __forceinline__ static void load(const void* p, float (&v)[8]) {
    const uint4 t = *static_cast<const uint4*>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ static void store(void* p, const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = f32_to_bf16_bits(v[2 * i]) | (f32_to_bf16_bits(v[2 * i + 1]) << 16);
    *static_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

template <>
struct Chunk<f16_t> {
  static constexpr int kCPL = 8;
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  __device__ __forceinline__ static void load(const void* p, float (&v)[8]) {
    const half8 t = *static_cast<const half8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = static_cast<float>(t[i]);
  }
  __device__ __forceinline__ static void store(void* p, const float (&v)[8]) {
    half8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = static_cast<_Float16>(v[i]);
    *static_cast<half8*>(p) = t;
  }
};

// One staged sampling point occupies 2 (forward) or 3 (backward) 16-byte LDS slots:
//   slot 0: the four corner weights, multiplied by the attention weight, 0 for a corner outside
//           the level;
//   slot 1: the four corner byte offsets relative to the batch element's value slab (head offset
//           included, lane offset not); they are multiples of 16, and bit 0 is set when the
//           corner lies inside the level;
//   slot 2: (backward only) {lx, ly, attn * W, attn * H}, lx/ly being the fractional parts.
// LDS stride (in slots) between the point lists of consecutive groups is odd, so the 8 or 16
// groups of a wave read from disjoint banks.
__host__ __device__ constexpr int group_stride_slots(int LP, int slots) { return (slots * LP) | 1; }

template <typename LT, int SLOTS>
__device__ __forceinline__ void stage_points(const int64_t* __restrict__ shapes,
                                             const int64_t* __restrict__ lsi,
                                             const LT* __restrict__ loc_blk,
                                             const LT* __restrict__ attn_blk, uint4* lds,
                                             int groups_here, int first_group_m, int S, int M,
                                             int D, int L, int P, int esize) {
  // loc_blk / attn_blk point at the first point of the block's first group
  const int LP = L * P;
  const int npts = groups_here * LP;
  const int stride = group_stride_slots(LP, SLOTS);
  const uint32_t rowb = static_cast<uint32_t>(M) * D * esize;
  for (int t = threadIdx.x; t < npts; t += 256) {
    const int g = t / LP, i = t - g * LP, l = i / P;
    const int64_t h = shapes[2 * l], w = shapes[2 * l + 1], st = lsi[l];
    const bool ok = h > 0 && w > 0 && st >= 0 && h <= S && w <= S && h * w <= S && st + h * w <= S;
    const int H = ok ? static_cast<int>(h) : 0, W = ok ? static_cast<int>(w) : 0;
    const int start = ok ? static_cast<int>(st) : 0;
    const float x = Elem<LT>::load(loc_blk + 2 * t);
    const float y = Elem<LT>::load(loc_blk + 2 * t + 1);
    const float a = Elem<LT>::load(attn_blk + t);
    const Footprint<float> f = footprint<float>(x, y, H, W);
    const int m = (first_group_m + g) % M;
    const uint32_t headb = static_cast<uint32_t>(m) * D * esize;
    uint4 wrec, orec;
    wrec.x = __float_as_uint(a * f.w[0]);
    wrec.y = __float_as_uint(a * f.w[1]);
    wrec.z = __float_as_uint(a * f.w[2]);
    wrec.w = __float_as_uint(a * f.w[3]);
    orec.x = (static_cast<uint32_t>(start + f.pix[0]) * rowb + headb) | (f.valid[0] ? 1u : 0u);
    orec.y = (static_cast<uint32_t>(start + f.pix[1]) * rowb + headb) | (f.valid[1] ? 1u : 0u);
    orec.z = (static_cast<uint32_t>(start + f.pix[2]) * rowb + headb) | (f.valid[2] ? 1u : 0u);
    orec.w = (static_cast<uint32_t>(start + f.pix[3]) * rowb + headb) | (f.valid[3] ? 1u : 0u);
    uint4* dst = lds + g * stride + SLOTS * i;
    dst[0] = wrec;
    dst[1] = orec;
    if constexpr (SLOTS == 3) {
      const float xs = x * static_cast<float>(W) - 0.5f, ys = y * static_cast<float>(H) - 0.5f;
      const float lx = f.any ? xs - floorf(xs) : 0.f, ly = f.any ? ys - floorf(ys) : 0.f;
      uint4 frec;
      frec.x = __float_as_uint(lx);
      frec.y = __float_as_uint(ly);
      frec.z = __float_as_uint(a * static_cast<float>(W));
      frec.w = __float_as_uint(a * static_cast<float>(H));
      dst[2] = frec;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------
template <typename VT, typename LT, int G>
__global__ __launch_bounds__(256) void fwd_tuned_kernel(
    const VT* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const LT* __restrict__ loc, const LT* __restrict__ attn,
    VT* __restrict__ out, int S, int M, int Lq, int L, int P) {
  constexpr int CPL = Chunk<VT>::kCPL;
  constexpr int D = G * CPL;
  constexpr int GPB = 256 / G;  // groups per block
  extern __shared__ uint4 lds[];

  const int LP = L * P;
  const int b = blockIdx.y;
  const int groups_per_batch = Lq * M;
  const int g0 = blockIdx.x * GPB;  // first group of this block inside the batch element
  const int groups_here = min(GPB, groups_per_batch - g0);
  const int64_t gbase = static_cast<int64_t>(b) * groups_per_batch + g0;

  stage_points<LT, 2>(shapes, lsi, loc + gbase * LP * 2, attn + gbase * LP, lds, groups_here,
                      g0 % M, S, M, D, L, P, static_cast<int>(sizeof(VT)));
  __syncthreads();

  const int g = threadIdx.x / G, lane = threadIdx.x % G;
  if (g >= groups_here) return;
  const char* vb = reinterpret_cast<const char*>(value) +
                   static_cast<int64_t>(b) * S * M * D * static_cast<int64_t>(sizeof(VT));
  const uint32_t lane_b = lane * 16;
  const uint4* rec = lds + g * group_stride_slots(LP, 2);

  float acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = 0.f;

#pragma unroll 4
  for (int i = 0; i < LP; ++i) {
    const uint4 wr = rec[2 * i], orr = rec[2 * i + 1];
    const float w[4] = {__uint_as_float(wr.x), __uint_as_float(wr.y), __uint_as_float(wr.z),
                        __uint_as_float(wr.w)};
    const uint32_t o[4] = {orr.x & ~15u, orr.y & ~15u, orr.z & ~15u, orr.w & ~15u};
    float v[4][CPL];
#pragma unroll
    for (int k = 0; k < 4; ++k) Chunk<VT>::load(vb + (o[k] + lane_b), v[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < CPL; ++c) acc[c] = fmaf(w[k], v[k][c], acc[c]);
  }
  Chunk<VT>::store(reinterpret_cast<char*>(out) + ((gbase + g) * D + lane * CPL) * sizeof(VT), acc);
}

// ---------------------------------------------------------------------------------------------
// Backward, query side: grad_loc, grad_attn, and the grad_value scatter by fp32 atomics.
// ---------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  // butterfly inside groups of G <= 16 lanes on the DPP path (no LDS traffic)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  if constexpr (G >= 8)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  if constexpr (G >= 16)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

template <typename VT, typename LT, int G>
__global__ __launch_bounds__(256) void bwd_tuned_atomic_kernel(
    const VT* __restrict__ grad_out, const VT* __restrict__ value,
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const LT* __restrict__ loc, const LT* __restrict__ attn, float* __restrict__ grad_value_acc,
    LT* __restrict__ grad_loc, LT* __restrict__ grad_attn, int S, int M, int Lq, int L, int P) {
  constexpr int CPL = Chunk<VT>::kCPL;
  constexpr int D = G * CPL;
  constexpr int GPB = 256 / G;
  extern __shared__ uint4 lds[];

  const int LP = L * P;
  const int b = blockIdx.y;
  const int groups_per_batch = Lq * M;
  const int g0 = blockIdx.x * GPB;
  const int groups_here = min(GPB, groups_per_batch - g0);
  const int64_t gbase = static_cast<int64_t>(b) * groups_per_batch + g0;
  const LT* loc_blk = loc + gbase * LP * 2;
  const LT* attn_blk = attn + gbase * LP;

  stage_points<LT, 3>(shapes, lsi, loc_blk, attn_blk, lds, groups_here, g0 % M, S, M, D, L, P,
                      static_cast<int>(sizeof(VT)));
  __syncthreads();

  // lanes of groups past the end redo the last group (no writes) so that every DPP row is full
  const int g = min(static_cast<int>(threadIdx.x) / G, groups_here - 1);
  const bool live = static_cast<int>(threadIdx.x) / G < groups_here;
  const int lane = threadIdx.x % G;
  const int64_t slab = static_cast<int64_t>(b) * S * M * D;
  const char* vb = reinterpret_cast<const char*>(value) + slab * static_cast<int64_t>(sizeof(VT));
  float* gvb = grad_value_acc + slab;
  const uint32_t lane_b = lane * 16;
  const uint4* rec = lds + g * group_stride_slots(LP, 3);

  float go[CPL];
  Chunk<VT>::load(reinterpret_cast<const char*>(grad_out) +
                      ((gbase + g) * D + lane * CPL) * sizeof(VT),
                  go);

  for (int i = 0; i < LP; ++i) {
    const uint4 wr = rec[3 * i], orr = rec[3 * i + 1], fr = rec[3 * i + 2];
    const float aw[4] = {__uint_as_float(wr.x), __uint_as_float(wr.y), __uint_as_float(wr.z),
                         __uint_as_float(wr.w)};
    const uint32_t oraw[4] = {orr.x, orr.y, orr.z, orr.w};
    const float lx = __uint_as_float(fr.x), ly = __uint_as_float(fr.y);
    const float aW = __uint_as_float(fr.z), aH = __uint_as_float(fr.w);
    const float hx = 1.f - lx, hy = 1.f - ly;
    float v[4][CPL];
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[k] = (oraw[k] & ~15u) + lane_b;
      Chunk<VT>::load(vb + o[k], v[k]);
    }
    float dot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) d = fmaf(go[c], v[k][c], d);
      dot[k] = (oraw[k] & 1u) ? d : 0.f;  // a corner outside the level contributes nothing
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (live && aw[k] != 0.f) {
        float* dst = gvb + o[k] / sizeof(VT);
#pragma unroll
        for (int c = 0; c < CPL; ++c) atomicAdd(dst + c, aw[k] * go[c]);
      }
    }
    // with w = {hy*hx, hy*lx, ly*hx, ly*lx}, dw/dx = {-hy, hy, -ly, ly}, dw/dy = {-hx, -lx, hx, lx}
    float sv = hy * fmaf(hx, dot[0], lx * dot[1]) + ly * fmaf(hx, dot[2], lx * dot[3]);
    float sx = fmaf(hy, dot[1] - dot[0], ly * (dot[3] - dot[2]));
    float sy = fmaf(hx, dot[2] - dot[0], lx * (dot[3] - dot[1]));
    sv = group_sum<G>(sv);
    sx = group_sum<G>(sx);
    sy = group_sum<G>(sy);
    if (lane == 0 && live) {
      const int64_t gi = (gbase + g) * LP + i;
      Elem<LT>::store(grad_attn + gi, sv);
      Elem<LT>::store(grad_loc + 2 * gi, aW * sx);
      Elem<LT>::store(grad_loc + 2 * gi + 1, aH * sy);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Host-side dispatch
// ---------------------------------------------------------------------------------------------
template <typename VT>
__global__ __launch_bounds__(256) void convert_f32_image_kernel(const float* __restrict__ src,
                                                                 VT* __restrict__ dst,
                                                                 int64_t total8) {
  // 8 elements per thread: two float4 in, one 16-byte chunk out
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total8; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i];
    const float4 b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    Chunk<VT>::store(reinterpret_cast<char*>(dst) + i * 16, v);
  }
}

inline bool tuned_shape_ok(int N, int S, int M, int D, int Lq, int L, int P, int dtype_val,
                           int dtype_loc, int* G_out) {
  int cpl;
  if (dtype_val == 0 /*F32*/ && dtype_loc == 0) cpl = 4;
  else if (dtype_val == 2 /*BF16*/ && (dtype_loc == 2 || dtype_loc == 0)) cpl = 8;
  else if (dtype_val == 3 /*F16*/ && (dtype_loc == 3 || dtype_loc == 0)) cpl = 8;
  else return false;
  if (D % cpl) return false;
  const int G = D / cpl;
  if (G != 4 && G != 8 && G != 16) return false;
  const int esize = cpl == 4 ? 4 : 2;
  if (static_cast<int64_t>(S) * M * D * esize > static_cast<int64_t>(UINT32_MAX)) return false;
  if (N > 65535) return false;
  if (static_cast<int64_t>(Lq) * M > INT32_MAX) return false;
  // LDS: (256/G) groups * (2*L*P + 1) slots * 16 B
  if (static_cast<int64_t>(256 / G) * group_stride_slots(L * P, 3) * 16 > 64 * 1024) return false;
  *G_out = G;
  return true;
}

inline bool all_aligned16(std::initializer_list<const void*> ps) {
  for (const void* p : ps)
    if (reinterpret_cast<uintptr_t>(p) % 16) return false;
  return true;
}

template <typename VT, typename LT>
int tuned_forward_typed(const void* value, const int64_t* shapes, const int64_t* lsi,
                        const void* loc, const void* attn, void* out, int N, int S, int M, int Lq,
                        int L, int P, int G, hipStream_t stream) {
  const int GPB = 256 / G;
  const dim3 grid((Lq * M + GPB - 1) / GPB, N);
  const size_t lds = static_cast<size_t>(GPB) * group_stride_slots(L * P, 2) * 16;
#define MSDA_TF(GG)                                                                             \
  hipLaunchKernelGGL((fwd_tuned_kernel<VT, LT, GG>), grid, dim3(256), lds, stream,              \
                     static_cast<const VT*>(value), shapes, lsi, static_cast<const LT*>(loc),   \
                     static_cast<const LT*>(attn), static_cast<VT*>(out), S, M, Lq, L, P)
  switch (G) {
    case 4: MSDA_TF(4); break;
    case 8: MSDA_TF(8); break;
    default: MSDA_TF(16); break;
  }
#undef MSDA_TF
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : static_cast<int>(e);
}

inline int tuned_forward(const void* value, const int64_t* shapes, const int64_t* lsi,
                         const void* loc, const void* attn, void* out, int N, int S, int M, int D,
                         int Lq, int L, int P, int dtype_val, int dtype_loc, hipStream_t stream) {
  int G;
  if (!tuned_shape_ok(N, S, M, D, Lq, L, P, dtype_val, dtype_loc, &G)) return kNotTuned;
  if (!all_aligned16({value, out})) return kNotTuned;
#define MSDA_ARGS value, shapes, lsi, loc, attn, out, N, S, M, Lq, L, P, G, stream
  if (dtype_val == 0) return tuned_forward_typed<float, float>(MSDA_ARGS);
  if (dtype_val == 2)
    return dtype_loc == 0 ? tuned_forward_typed<bf16_t, float>(MSDA_ARGS)
                          : tuned_forward_typed<bf16_t, bf16_t>(MSDA_ARGS);
  return dtype_loc == 0 ? tuned_forward_typed<f16_t, float>(MSDA_ARGS)
                        : tuned_forward_typed<f16_t, f16_t>(MSDA_ARGS);
#undef MSDA_ARGS
}

inline int64_t tuned_backward_workspace_bytes(int N, int S, int M, int D, int Lq, int L, int P,
                                              int dtype_val, int dtype_loc) {
  int G;
  if (!tuned_shape_ok(N, S, M, D, Lq, L, P, dtype_val, dtype_loc, &G)) return -1;
  if (dtype_val == 0) return 0;
  return static_cast<int64_t>(N) * S * M * D * 4;  // fp32 accumulation image of grad_value
}

template <typename VT, typename LT>
int tuned_backward_typed(const void* grad_out, const void* value, const int64_t* shapes,
                         const int64_t* lsi, const void* loc, const void* attn, void* grad_value,
                         void* grad_loc, void* grad_attn, void* workspace, int N, int S, int M,
                         int D, int Lq, int L, int P, int G, hipStream_t stream) {
  const int GPB = 256 / G;
  const dim3 grid((Lq * M + GPB - 1) / GPB, N);
  const size_t lds = static_cast<size_t>(GPB) * group_stride_slots(L * P, 3) * 16;
  const int64_t total = static_cast<int64_t>(N) * S * M * D;
  constexpr bool kDirect = sizeof(VT) == 4;
  float* acc = kDirect ? static_cast<float*>(grad_value) : static_cast<float*>(workspace);
  hipError_t e = hipMemsetAsync(acc, 0, static_cast<size_t>(total) * 4, stream);
  if (e != hipSuccess) return static_cast<int>(e);
#define MSDA_TB(GG)                                                                              \
  hipLaunchKernelGGL((bwd_tuned_atomic_kernel<VT, LT, GG>), grid, dim3(256), lds, stream,        \
                     static_cast<const VT*>(grad_out), static_cast<const VT*>(value), shapes,    \
                     lsi, static_cast<const LT*>(loc), static_cast<const LT*>(attn), acc,        \
                     static_cast<LT*>(grad_loc), static_cast<LT*>(grad_attn), S, M, Lq, L, P)
  switch (G) {
    case 4: MSDA_TB(4); break;
    case 8: MSDA_TB(8); break;
    default: MSDA_TB(16); break;
  }
#undef MSDA_TB
  e = hipGetLastError();
  if (e != hipSuccess) return static_cast<int>(e);
  if constexpr (!kDirect) {
    const int64_t total8 = total / 8;  // D is a multiple of 8 on this path
    const int64_t cb = (total8 + 255) / 256;
    hipLaunchKernelGGL((convert_f32_image_kernel<VT>), dim3(static_cast<unsigned>(cb < 4096 ? cb : 4096)),
                       dim3(256), 0, stream, static_cast<const float*>(workspace),
                       static_cast<VT*>(grad_value), total8);
    e = hipGetLastError();
    if (e != hipSuccess) return static_cast<int>(e);
  }
  return 0;
}

inline int tuned_backward(const void* grad_out, const void* value, const int64_t* shapes,
                          const int64_t* lsi, const void* loc, const void* attn, void* grad_value,
                          void* grad_loc, void* grad_attn, void* workspace, int N, int S, int M,
                          int D, int Lq, int L, int P, int dtype_val, int dtype_loc,
                          hipStream_t stream) {
  int G;
  if (!tuned_shape_ok(N, S, M, D, Lq, L, P, dtype_val, dtype_loc, &G)) return kNotTuned;
  if (!all_aligned16({value, grad_out, grad_value, workspace})) return kNotTuned;
#define MSDA_ARGS                                                                               \
  grad_out, value, shapes, lsi, loc, attn, grad_value, grad_loc, grad_attn, workspace, N, S, M, \
      D, Lq, L, P, G, stream
  if (dtype_val == 0) return tuned_backward_typed<float, float>(MSDA_ARGS);
  if (dtype_val == 2)
    return dtype_loc == 0 ? tuned_backward_typed<bf16_t, float>(MSDA_ARGS)
                          : tuned_backward_typed<bf16_t, bf16_t>(MSDA_ARGS);
  return dtype_loc == 0 ? tuned_backward_typed<f16_t, float>(MSDA_ARGS)
                        : tuned_backward_typed<f16_t, f16_t>(MSDA_ARGS);
#undef MSDA_ARGS
}

}  // namespace msda
