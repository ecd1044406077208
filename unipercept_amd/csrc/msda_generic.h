// Generic MSDeformAttn kernels: any head dim D, any supported dtype, one lane per channel.
// A group of G lanes (G = 8/16/32/64, chosen from D) owns one (batch, query, head) triple and
// strides over the D channels.  These are the correctness baseline and the fallback for shapes
// the tuned kernels do not cover (D not a multiple of 4, D > 64, fp64, fp16).
#pragma once

#include "msda_common.h"

namespace msda {

constexpr int kMaxLevels = 16;

struct LevelTable {
  int H[kMaxLevels];
  int W[kMaxLevels];
  int start[kMaxLevels];
};

// Loads the level table from device memory; a level whose extent does not fit inside [0, S) is
// given H = W = 0, so that no address derived from it can leave the value tensor.
__device__ __forceinline__ void load_levels(const int64_t* __restrict__ shapes,
                                            const int64_t* __restrict__ lsi, int L, int S,
                                            LevelTable& t) {
  for (int l = 0; l < L; ++l) {
    const int64_t h = shapes[2 * l], w = shapes[2 * l + 1], st = lsi[l];
    const bool ok = h > 0 && w > 0 && st >= 0 && h <= S && w <= S && h * w <= S && st + h * w <= S;
    t.H[l] = ok ? static_cast<int>(h) : 0;
    t.W[l] = ok ? static_cast<int>(w) : 0;
    t.start[l] = ok ? static_cast<int>(st) : 0;
  }
}

template <typename VT, typename LT, int G>
__global__ __launch_bounds__(256) void fwd_generic_kernel(
    const VT* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const LT* __restrict__ loc, const LT* __restrict__ attn,
    VT* __restrict__ out, int64_t groups, int S, int M, int D, int Lq, int L, int P) {
  using CT = typename Elem<VT>::compute_t;
  const int64_t group = static_cast<int64_t>(blockIdx.x) * (256 / G) + threadIdx.x / G;
  const int lane = threadIdx.x % G;
  if (group >= groups) return;
  const int m = static_cast<int>(group % M);
  const int64_t b = group / M / Lq;
  const LT* loc_p = loc + group * L * P * 2;
  const LT* attn_p = attn + group * L * P;
  VT* out_p = out + group * D;
  const int64_t row = static_cast<int64_t>(M) * D;  // elements between two pixels of one head

  for (int c0 = 0; c0 < D; c0 += G) {
    const int c = c0 + lane;
    const bool act = c < D;
    CT acc = 0;
    for (int l = 0; l < L; ++l) {
      const int64_t h = shapes[2 * l], w = shapes[2 * l + 1], st = lsi[l];
      const bool ok = h > 0 && w > 0 && st >= 0 && h <= S && w <= S && h * w <= S && st + h * w <= S;
      const int H = ok ? static_cast<int>(h) : 0, W = ok ? static_cast<int>(w) : 0;
      const VT* base = value + (b * S + (ok ? st : 0)) * row + static_cast<int64_t>(m) * D + c;
      for (int p = 0; p < P; ++p) {
        const int i = l * P + p;
        const CT x = static_cast<CT>(Elem<LT>::load(loc_p + 2 * i));
        const CT y = static_cast<CT>(Elem<LT>::load(loc_p + 2 * i + 1));
        const CT a = static_cast<CT>(Elem<LT>::load(attn_p + i));
        const Footprint<CT> f = footprint<CT>(x, y, H, W);
        if (act) {
          // the four loads go out together: every corner address is clamped into the level; a
          // corner that does not take part is dropped afterwards, whatever was loaded
          CT v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = static_cast<CT>(Elem<VT>::load(base + f.pix[k] * row));
#pragma unroll
          for (int k = 0; k < 4; ++k) acc += f.valid[k] ? a * f.w[k] * v[k] : static_cast<CT>(0);
        }
      }
    }
    if (act) Elem<VT>::store(out_p + c, acc);
  }
}

// GT: accumulation type of the grad_value image (float, or double for fp64 inputs).
// SCATTER = false skips the grad_value atomics (deterministic mode computes it separately).
template <typename VT, typename LT, typename GT, int G, bool SCATTER>
__global__ __launch_bounds__(256) void bwd_generic_kernel(
    const VT* __restrict__ grad_out, const VT* __restrict__ value,
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const LT* __restrict__ loc, const LT* __restrict__ attn, GT* __restrict__ grad_value_acc,
    LT* __restrict__ grad_loc, LT* __restrict__ grad_attn, int64_t groups, int S, int M, int D,
    int Lq, int L, int P) {
  using CT = typename Elem<VT>::compute_t;
  int64_t group = static_cast<int64_t>(blockIdx.x) * (256 / G) + threadIdx.x / G;
  const int lane = threadIdx.x % G;
  // every lane stays in the loop so that the group-wide shuffles below are always fully populated
  const bool live = group < groups;
  if (!live) group = groups - 1;
  const int m = static_cast<int>(group % M);
  const int64_t b = group / M / Lq;
  const LT* loc_p = loc + group * L * P * 2;
  const LT* attn_p = attn + group * L * P;
  const VT* go_p = grad_out + group * D;
  const int64_t row = static_cast<int64_t>(M) * D;

  for (int l = 0; l < L; ++l) {
    const int64_t h = shapes[2 * l], w = shapes[2 * l + 1], st = lsi[l];
    const bool ok = h > 0 && w > 0 && st >= 0 && h <= S && w <= S && h * w <= S && st + h * w <= S;
    const int H = ok ? static_cast<int>(h) : 0, W = ok ? static_cast<int>(w) : 0;
    const int64_t base = (b * S + (ok ? st : 0)) * row + static_cast<int64_t>(m) * D;
    for (int p = 0; p < P; ++p) {
      const int i = l * P + p;
      const CT x = static_cast<CT>(Elem<LT>::load(loc_p + 2 * i));
      const CT y = static_cast<CT>(Elem<LT>::load(loc_p + 2 * i + 1));
      const CT a = static_cast<CT>(Elem<LT>::load(attn_p + i));
      const Footprint<CT> f = footprint<CT>(x, y, H, W);
      CT dot[4] = {0, 0, 0, 0};
      if (f.any) {
        for (int c = lane; c < D; c += G) {
          const CT g = static_cast<CT>(Elem<VT>::load(go_p + c));
          CT v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k)  // together; clamped addresses, see the forward
            v[k] = static_cast<CT>(Elem<VT>::load(value + base + f.pix[k] * row + c));
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (f.valid[k]) {
              const int64_t off = base + f.pix[k] * row + c;
              dot[k] += g * v[k];
              if (SCATTER && live) atomicAdd(grad_value_acc + off, static_cast<GT>(a * f.w[k] * g));
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int s = G / 2; s > 0; s >>= 1) dot[k] += __shfl_xor(dot[k], s, G);
      }
      if (lane == 0 && live) {
        CT sv = 0, sx = 0, sy = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          sv += f.w[k] * dot[k];
          sx += f.dwx[k] * dot[k];
          sy += f.dwy[k] * dot[k];
        }
        const int64_t gi = group * L * P + i;
        Elem<LT>::store(grad_attn + gi, sv);
        Elem<LT>::store(grad_loc + 2 * gi, static_cast<CT>(W) * a * sx);
        Elem<LT>::store(grad_loc + 2 * gi + 1, static_cast<CT>(H) * a * sy);
      }
    }
  }
}

// Deterministic grad_value: one thread per destination element (b, s, m, c) walks every point
// that can touch its pixel, in a fixed (q, p, corner) order.  O(Lq * P) per element.
template <typename VT, typename LT>
__global__ __launch_bounds__(256) void bwd_value_gather_kernel(
    const VT* __restrict__ grad_out, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const LT* __restrict__ loc, const LT* __restrict__ attn,
    VT* __restrict__ grad_value, int64_t total, int S, int M, int D, int Lq, int L, int P) {
  using CT = typename Elem<VT>::compute_t;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = static_cast<int>(idx % D);
  const int m = static_cast<int>(idx / D % M);
  const int s = static_cast<int>(idx / D / M % S);
  const int64_t b = idx / D / M / S;
  CT acc = 0;
  for (int l = 0; l < L; ++l) {
    const int64_t h = shapes[2 * l], w = shapes[2 * l + 1], st = lsi[l];
    const bool ok = h > 0 && w > 0 && st >= 0 && h <= S && w <= S && h * w <= S && st + h * w <= S;
    if (!ok || s < st || s >= st + h * w) continue;
    const int H = static_cast<int>(h), W = static_cast<int>(w);
    const int pix = s - static_cast<int>(st);
    for (int q = 0; q < Lq; ++q) {
      const int64_t group = (b * Lq + q) * M + m;
      const CT g = static_cast<CT>(Elem<VT>::load(grad_out + group * D + c));
      for (int p = 0; p < P; ++p) {
        const int64_t gi = group * L * P + l * P + p;
        const CT x = static_cast<CT>(Elem<LT>::load(loc + 2 * gi));
        const CT y = static_cast<CT>(Elem<LT>::load(loc + 2 * gi + 1));
        const CT a = static_cast<CT>(Elem<LT>::load(attn + gi));
        const Footprint<CT> f = footprint<CT>(x, y, H, W);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (f.pix[k] == pix && f.valid[k]) acc += a * f.w[k] * g;
      }
    }
  }
  Elem<VT>::store(grad_value + idx, acc);
}

// fp32 accumulation image -> 16-bit grad_value
template <typename VT>
__global__ __launch_bounds__(256) void convert_from_f32_kernel(const float* __restrict__ src,
                                                                VT* __restrict__ dst,
                                                                int64_t total) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += stride)
    Elem<VT>::store(dst + i, src[i]);
}

}  // namespace msda
