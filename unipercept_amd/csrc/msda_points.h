// Sampling-point epilogue for gfx950: turns the raw output of the (fused) sampling_offsets +
// attention_weights projection into the packed locations+weights tensor the attention kernels
// read, and back-propagates through it.
//
//   raw    (N, Lq, M, 3*L*P)  per (batch, query, head) row: L*P (dx, dy) offsets, then L*P logits
//   ref    (N, Lq, L, 2)      reference point of the query on every level, normalised to [0, 1]
//   packed (N, Lq, M, 3*L*P)  per row: L*P (x, y) = ref + (dx / W_l, dy / H_l), then
//                             softmax over the L*P logits
//
// A workgroup takes up to 256 consecutive rows (fewer when L*P is large).  Their elements are
// contiguous in memory, so they move between global memory and LDS with 16-byte lane accesses
// whatever L*P and the element type are; in LDS a row is padded to an odd number of words and
// one thread works on one row.
#pragma once

#include "msda_common.h"

namespace msda {

// rows per workgroup: 256 (one per thread) unless the LDS image would exceed ~64 KB
inline int point_rows_per_block(int L, int P, bool backward) {
  int rpb = 256;
  const size_t row_bytes = static_cast<size_t>((3 * L * P) | 1) * 4 * (backward ? 2 : 1);
  while (rpb > 16 && rpb * row_bytes > 64 * 1024) rpb /= 2;
  return rpb;
}

__host__ __device__ constexpr int point_row_stride(int width) { return width | 1; }  // words, odd

// LDS image of the block's rows as fp32 words, row r at r * stride
template <typename LT>
__device__ __forceinline__ void rows_to_lds(const LT* __restrict__ src, int64_t first_row,
                                            int n_rows, int width, float* rows) {
  const int stride = point_row_stride(width);
  const int64_t base = first_row * width;
  const int total = n_rows * width;
  constexpr int VE = 16 / sizeof(LT);  // elements per 16-byte access
  // the block's first element is 16-byte aligned: a multiple of 16 rows of 3*L*P elements
  const int nvec = total / VE;
  for (int v = threadIdx.x; v < nvec; v += 256) {
    LT tmp[VE];
    *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(src + base + v * VE);
    int r = (v * VE) / width, c = v * VE - r * width;  // one division per 16 bytes
#pragma unroll
    for (int k = 0; k < VE; ++k) {
      rows[r * stride + c] = Elem<LT>::load(&tmp[k]);
      if (++c == width) {
        c = 0;
        ++r;
      }
    }
  }
  for (int e = nvec * VE + threadIdx.x; e < total; e += 256) {
    const int r = e / width;
    rows[r * stride + (e - r * width)] = Elem<LT>::load(src + base + e);
  }
}

template <typename LT>
__device__ __forceinline__ void lds_to_rows(const float* rows, int64_t first_row, int n_rows,
                                            int width, LT* __restrict__ dst) {
  const int stride = point_row_stride(width);
  const int64_t base = first_row * width;
  const int total = n_rows * width;
  constexpr int VE = 16 / sizeof(LT);
  const int nvec = total / VE;
  for (int v = threadIdx.x; v < nvec; v += 256) {
    LT tmp[VE];
    int r = (v * VE) / width, c = v * VE - r * width;
#pragma unroll
    for (int k = 0; k < VE; ++k) {
      Elem<LT>::store(&tmp[k], rows[r * stride + c]);
      if (++c == width) {
        c = 0;
        ++r;
      }
    }
    *reinterpret_cast<uint4*>(dst + base + v * VE) = *reinterpret_cast<const uint4*>(tmp);
  }
  for (int e = nvec * VE + threadIdx.x; e < total; e += 256) {
    const int r = e / width;
    Elem<LT>::store(dst + base + e, rows[r * stride + (e - r * width)]);
  }
}

// forward: raw -> packed.  RT: element type of the projection output (and of its gradient), LT:
// element type of the reference points and of the packed tensor (fp32 next to a 16-bit RT keeps
// the sampling locations at full resolution: a bf16 location near 1.0 moves in steps of 1/256).
template <typename RT, typename LT>
__global__ __launch_bounds__(256) void points_forward_kernel(
    const RT* __restrict__ raw, const LT* __restrict__ ref, const int64_t* __restrict__ shapes,
    LT* __restrict__ packed, int64_t n_rows_total, int M, int L, int P, int rpb) {
  extern __shared__ float rows[];
  __shared__ float inv_wh[2 * 16];
  const int LP = L * P, width = 3 * LP, stride = point_row_stride(width);
  const int64_t first_row = static_cast<int64_t>(blockIdx.x) * rpb;
  const int n_rows = static_cast<int>(min<int64_t>(rpb, n_rows_total - first_row));
  if (threadIdx.x < L) {
    const float h = static_cast<float>(shapes[2 * threadIdx.x]);
    const float w = static_cast<float>(shapes[2 * threadIdx.x + 1]);
    inv_wh[2 * threadIdx.x] = 1.0f / w;
    inv_wh[2 * threadIdx.x + 1] = 1.0f / h;
  }
  rows_to_lds(raw, first_row, n_rows, width, rows);
  __syncthreads();
  if (static_cast<int>(threadIdx.x) < n_rows) {
    float* r = rows + threadIdx.x * stride;
    const int64_t bq = (first_row + threadIdx.x) / M;  // (batch, query) index of the row
    const LT* rp = ref + bq * L * 2;
    for (int l = 0; l < L; ++l) {
      const float rx = Elem<LT>::load(rp + 2 * l), ry = Elem<LT>::load(rp + 2 * l + 1);
      const float iw = inv_wh[2 * l], ih = inv_wh[2 * l + 1];
      for (int p = 0; p < P; ++p) {
        const int i = l * P + p;
        r[2 * i] = fmaf(r[2 * i], iw, rx);
        r[2 * i + 1] = fmaf(r[2 * i + 1], ih, ry);
      }
    }
    float* a = r + 2 * LP;
    float mx = a[0];
    for (int i = 1; i < LP; ++i) mx = fmaxf(mx, a[i]);
    float sum = 0.f;
    for (int i = 0; i < LP; ++i) {
      const float e = expf(a[i] - mx);
      a[i] = e;
      sum += e;
    }
    const float inv = 1.0f / sum;
    for (int i = 0; i < LP; ++i) a[i] *= inv;
  }
  __syncthreads();
  lds_to_rows(rows, first_row, n_rows, width, packed);
}

// backward: (grad_packed, packed) -> grad_raw
//   d dx = d x / W_l, d dy = d y / H_l;  d logit_i = w_i * (g_i - sum_k w_k g_k)
template <typename RT, typename LT>
__global__ __launch_bounds__(256) void points_backward_kernel(
    const LT* __restrict__ grad_packed, const LT* __restrict__ packed,
    const int64_t* __restrict__ shapes, RT* __restrict__ grad_raw, int64_t n_rows_total, int L,
    int P, int rpb) {
  extern __shared__ float rows[];  // gradient rows, then the weight rows
  __shared__ float inv_wh[2 * 16];
  const int LP = L * P, width = 3 * LP, stride = point_row_stride(width);
  float* wrows = rows + rpb * stride;
  const int64_t first_row = static_cast<int64_t>(blockIdx.x) * rpb;
  const int n_rows = static_cast<int>(min<int64_t>(rpb, n_rows_total - first_row));
  if (threadIdx.x < L) {
    const float h = static_cast<float>(shapes[2 * threadIdx.x]);
    const float w = static_cast<float>(shapes[2 * threadIdx.x + 1]);
    inv_wh[2 * threadIdx.x] = 1.0f / w;
    inv_wh[2 * threadIdx.x + 1] = 1.0f / h;
  }
  rows_to_lds(grad_packed, first_row, n_rows, width, rows);
  rows_to_lds(packed, first_row, n_rows, width, wrows);
  __syncthreads();
  if (static_cast<int>(threadIdx.x) < n_rows) {
    float* g = rows + threadIdx.x * stride;
    const float* w = wrows + threadIdx.x * stride + 2 * LP;
    for (int l = 0; l < L; ++l) {
      const float iw = inv_wh[2 * l], ih = inv_wh[2 * l + 1];
      for (int p = 0; p < P; ++p) {
        const int i = l * P + p;
        g[2 * i] *= iw;
        g[2 * i + 1] *= ih;
      }
    }
    float* ga = g + 2 * LP;
    float dot = 0.f;
    for (int i = 0; i < LP; ++i) dot = fmaf(w[i], ga[i], dot);
    for (int i = 0; i < LP; ++i) ga[i] = w[i] * (ga[i] - dot);
  }
  __syncthreads();
  lds_to_rows(rows, first_row, n_rows, width, grad_raw);
}

inline size_t points_lds_bytes(int L, int P, int rpb, bool backward) {
  return static_cast<size_t>(rpb) * point_row_stride(3 * L * P) * 4 * (backward ? 2 : 1);
}

}  // namespace msda
