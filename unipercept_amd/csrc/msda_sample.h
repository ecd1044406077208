// Point sampling with border padding for gfx950: out[n, c, p] = input[n, c] interpolated at
// point p of sample n.  This is the gather of the reference's `unipercept.vision.pointrend.sample`
// (sources/unipercept/vision/pointrend.py:25-76 of the reference):
//     grid_sample(input, 2 * point_coords - 1, mode, padding_mode="border", align_corners)
// with point_coords (N, P, 2) in [0, 1]^2, last dimension (x, y).
//
// Layout: input is (N, C, H, W) — every channel its own plane — so a point's four corners are
// four scalar loads per channel.  One thread owns one point of one sample and a slice of the
// channels: the footprint (weights, offsets, clamp state) is computed once per thread, the loads
// of neighbouring lanes are scattered over the plane (they sit in L2: a plane is at most a few
// hundred KB), and the stores to out[n, c, :] are contiguous over the lanes.
#pragma once

#include "msda_common.h"

namespace msda {

constexpr int kSampleBilinear = 0;
constexpr int kSampleNearest = 1;

// products and sums of the coordinate transform are kept uncontracted (no fma), so that a pixel
// coordinate comes out as the scalar formula of aten::grid_sampler_2d gives it
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }

// One sampled point on an H x W map.
template <typename CT>
struct BorderPoint {
  CT w[4];     // bilinear weights of (y0,x0), (y0,x0+1), (y0+1,x0), (y0+1,x0+1); 0 when skipped
  CT dwx[4];   // d w / d ix, d w / d iy (0 when skipped)
  CT dwy[4];
  int off[4];  // y * W + x of the corner, always inside the map
  unsigned valid;  // bit k: corner k takes part (it lies inside the map)
  CT gmx, gmy; // d ix / d coordinate x, d iy / d coordinate y (0 where the border clamp acts)
};

// pixel coordinate of normalised coordinate c in [0, 1] along an axis of `size` pixels, clamped
// to the border; `mult` receives d(pixel coordinate) / d c
template <typename CT>
__device__ __forceinline__ CT border_coordinate(CT c, int size, int align, CT& mult) {
  const CT g = add_rn(mul_rn(static_cast<CT>(2), c), static_cast<CT>(-1));
  CT x;
  if (align) {
    mult = static_cast<CT>(size - 1);  // 2 * (size - 1) / 2
    x = mul_rn(add_rn(g, static_cast<CT>(1)) / static_cast<CT>(2), static_cast<CT>(size - 1));
  } else {
    mult = static_cast<CT>(size);  // 2 * size / 2
    x = add_rn(mul_rn(add_rn(g, static_cast<CT>(1)), static_cast<CT>(size)), static_cast<CT>(-1)) /
        static_cast<CT>(2);
  }
  const CT mx = static_cast<CT>(size - 1);
  if (!(x > static_cast<CT>(0))) {  // also NaN: never an address outside the map
    mult = static_cast<CT>(0);
    return static_cast<CT>(0);
  }
  if (x >= mx) {
    mult = static_cast<CT>(0);
    return mx;
  }
  return x;
}

template <typename CT>
__device__ __forceinline__ BorderPoint<CT> border_point(CT cx, CT cy, int H, int W, int mode,
                                                       int align) {
  BorderPoint<CT> f;
  const CT ix = border_coordinate(cx, W, align, f.gmx);
  const CT iy = border_coordinate(cy, H, align, f.gmy);
  const CT z = static_cast<CT>(0), one = static_cast<CT>(1);
  if (mode == kSampleNearest) {
    // ties to even, as nearbyint under the default rounding mode
    const int xn = min(max(static_cast<int>(rint(ix)), 0), W - 1);
    const int yn = min(max(static_cast<int>(rint(iy)), 0), H - 1);
    f.gmx = f.gmy = z;
    f.valid = 1u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f.w[k] = k == 0 ? one : z;
      f.dwx[k] = f.dwy[k] = z;
      f.off[k] = yn * W + xn;
    }
    return f;
  }
  const CT xf = floor(ix), yf = floor(iy);
  const int x0 = static_cast<int>(xf), y0 = static_cast<int>(yf);
  const CT lx = ix - xf, ly = iy - yf, hx = one - lx, hy = one - ly;
  // ix lies in [0, W - 1]: x0 is inside, x0 + 1 is outside only when ix == W - 1 (weight 0 then)
  const bool vx1 = x0 + 1 < W, vy1 = y0 + 1 < H;
  const int x1 = vx1 ? x0 + 1 : x0, y1 = vy1 ? y0 + 1 : y0;
  const bool v[4] = {true, vx1, vy1, vx1 && vy1};
  const CT w[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
  const CT dx[4] = {-hy, hy, -ly, ly};
  const CT dy[4] = {-hx, -lx, hx, lx};
  f.off[0] = y0 * W + x0;
  f.off[1] = y0 * W + x1;
  f.off[2] = y1 * W + x0;
  f.off[3] = y1 * W + x1;
  f.valid = 1u | (vx1 ? 2u : 0u) | (vy1 ? 4u : 0u) | (vx1 && vy1 ? 8u : 0u);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f.w[k] = v[k] ? w[k] : z;
    f.dwx[k] = v[k] ? dx[k] : z;
    f.dwy[k] = v[k] ? dy[k] : z;
  }
  return f;
}

// Work units are (sample, channel slice) pairs, `per` workgroups each.  Workgroups are numbered so
// that all workgroups of a unit have the same blockIdx.x % 8, i.e. run on one XCD under the
// round-robin placement of workgroups: the unit's planes are then fetched into ONE L2 instead of
// all eight (speed only; any placement is correct).
struct SampleGrid {
  int per;       // workgroups per unit: blocks of 256 points, or bands of a plane
  int cslices;   // channel slices per sample
  int cper;      // channels per slice
  int64_t units; // samples * cslices
};

struct SampleBlock {
  int64_t n;  // sample
  int cs;     // channel slice
  int sub;    // workgroup of the unit
  bool live;
};

__device__ __forceinline__ SampleBlock sample_block(const SampleGrid& g) {
  const unsigned xcd = blockIdx.x % 8, j = blockIdx.x / 8;
  const int64_t unit = static_cast<int64_t>(j / g.per) * 8 + xcd;
  SampleBlock b;
  b.sub = static_cast<int>(j % g.per);
  b.live = unit < g.units;
  b.n = unit / g.cslices;
  b.cs = static_cast<int>(unit % g.cslices);
  return b;
}

inline int64_t sample_grid_blocks(const SampleGrid& g) { return (g.units + 7) / 8 * 8 * g.per; }

#ifndef MSDA_SAMPLE_PPT
#define MSDA_SAMPLE_PPT 2
#endif
constexpr int kSamplePPT = MSDA_SAMPLE_PPT;  // points per thread of the forward (loads of all in flight together)

template <typename T>
__global__ __launch_bounds__(256) void point_sample_fwd_kernel(
    const T* __restrict__ input, const T* __restrict__ coords, T* __restrict__ out, int C, int H,
    int W, int P, SampleGrid grid, int mode, int align) {
  using CT = typename Elem<T>::compute_t;
  const SampleBlock blk = sample_block(grid);
  const int64_t n = blk.n;
  if (!blk.live) return;
  int p[kSamplePPT];
  BorderPoint<CT> f[kSamplePPT];
#pragma unroll
  for (int u = 0; u < kSamplePPT; ++u) {
    p[u] = blk.sub * (256 * kSamplePPT) + u * 256 + threadIdx.x;
    const int pc = min(p[u], P - 1);  // a slot past the end redoes the last point (not stored)
    const T* xy = coords + (n * P + pc) * 2;
    f[u] = border_point<CT>(static_cast<CT>(Elem<T>::load(xy)),
                            static_cast<CT>(Elem<T>::load(xy + 1)), H, W, mode, align);
  }
  const int c0 = blk.cs * grid.cper, c1 = min(C, c0 + grid.cper);
  const int64_t plane = static_cast<int64_t>(H) * W;
  for (int c = c0; c < c1; ++c) {
    const T* src = input + (n * C + c) * plane;
    // all loads go out together (every offset lies inside the map); a corner that does not
    // take part is dropped afterwards, whatever its value
    CT v[kSamplePPT][4];
#pragma unroll
    for (int u = 0; u < kSamplePPT; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[u][k] = static_cast<CT>(Elem<T>::load(src + f[u].off[k]));
#pragma unroll
    for (int u = 0; u < kSamplePPT; ++u) {
      CT acc = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        acc += (f[u].valid >> k & 1u) ? v[u][k] * f[u].w[k] : static_cast<CT>(0);
      if (p[u] < P) Elem<T>::store(out + (n * C + c) * P + p[u], acc);
    }
  }
}

// Backward with global atomics (maps too large for the band kernel below).  SCATTER = false
// computes grad_coords only.  grad_input must be zero on entry when SCATTER; grad_coords must be
// zero on entry when grid.cslices > 1 (the slices of a point then add their parts), otherwise it
// is overwritten.
template <typename T, bool SCATTER>
__global__ __launch_bounds__(256) void point_sample_bwd_kernel(
    const T* __restrict__ grad_out, const T* __restrict__ input, const T* __restrict__ coords,
    T* __restrict__ grad_input, T* __restrict__ grad_coords, int C, int H, int W, int P,
    SampleGrid grid, int mode, int align) {
  const SampleBlock blk = sample_block(grid);
  const int64_t n = blk.n;
  const int p = blk.sub * 256 + threadIdx.x;
  if (!blk.live || p >= P) return;
  const T* xy = coords + (n * P + p) * 2;
  const BorderPoint<T> f = border_point<T>(xy[0], xy[1], H, W, mode, align);
  const int c0 = blk.cs * grid.cper, c1 = min(C, c0 + grid.cper);
  const int64_t plane = static_cast<int64_t>(H) * W;
  T sx = 0, sy = 0;
  if (SCATTER || mode == kSampleBilinear) {
    for (int c = c0; c < c1; ++c) {
      const T* src = input + (n * C + c) * plane;
      T* dst = grad_input + (n * C + c) * plane;
      const T g = grad_out[(n * C + c) * P + p];
      T v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = src[f.off[k]];  // together; offsets are always inside
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (f.valid >> k & 1u) {
          if constexpr (SCATTER) atomicAdd(dst + f.off[k], f.w[k] * g);
          sx += v[k] * f.dwx[k] * g;
          sy += v[k] * f.dwy[k] * g;
        }
      }
    }
  }
  T* gc = grad_coords + (n * P + p) * 2;
  if (grid.cslices > 1) {
    if (mode == kSampleBilinear) {
      atomicAdd(gc, f.gmx * sx);
      atomicAdd(gc + 1, f.gmy * sy);
    }
  } else {
    gc[0] = f.gmx * sx;
    gc[1] = f.gmy * sy;
  }
}

// Rows a point touches: ya (weight wa) and yb (weight wb); yb == ya when there is only one.
template <typename T>
__device__ __forceinline__ void point_rows(T cy, int H, int mode, int align, int& ya, int& yb,
                                           T& wa, T& wb) {
  T my;
  const T iy = border_coordinate<T>(cy, H, align, my);
  if (mode == kSampleNearest) {
    ya = yb = min(max(static_cast<int>(rint(iy)), 0), H - 1);
    wa = 1;
    wb = 0;
  } else {
    const T yf = floor(iy);
    ya = static_cast<int>(yf);
    yb = ya + 1;
    wb = iy - yf;
    wa = static_cast<T>(1) - wb;
    if (yb >= H) {  // only when iy == H - 1: the row below is outside the map
      yb = ya;
      wb = 0;
    }
  }
}

constexpr int kMaxBands = 64;

// Point lists per band of rows, one workgroup per sample: list[n][off[n][b] .. off[n][b + 1])
// holds the points of sample n that touch band b (a point whose two rows straddle a band
// boundary is listed in both).  `list` has 2 * P entries per sample, `off` kMaxBands + 1.
// The band kernel then walks P / bands points instead of all P.
template <typename T>
__global__ __launch_bounds__(1024) void point_band_lists_kernel(
    const T* __restrict__ coords, int* __restrict__ list, int* __restrict__ off, int H, int P,
    int band_rows, int n_bands, int mode, int align) {
  __shared__ int cnt[kMaxBands + 1];
  const int64_t n = blockIdx.x;
  const T* xy = coords + n * P * 2;
  int* my_list = list + n * 2 * static_cast<int64_t>(P);
  int* my_off = off + n * (kMaxBands + 1);
  if (threadIdx.x <= kMaxBands) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += 1024) {
    int ya, yb;
    T wa, wb;
    point_rows<T>(xy[2 * p + 1], H, mode, align, ya, yb, wa, wb);
    const int ba = ya / band_rows, bb = yb / band_rows;
    atomicAdd(&cnt[ba], 1);
    if (bb != ba) atomicAdd(&cnt[bb], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // exclusive scan of at most 64 counts; cnt becomes the cursor
    int run = 0;
    for (int b2 = 0; b2 < n_bands; ++b2) {
      const int c = cnt[b2];
      my_off[b2] = run;
      cnt[b2] = run;
      run += c;
    }
    my_off[n_bands] = run;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += 1024) {
    int ya, yb;
    T wa, wb;
    point_rows<T>(xy[2 * p + 1], H, mode, align, ya, yb, wa, wb);
    const int ba = ya / band_rows, bb = yb / band_rows;
    my_list[atomicAdd(&cnt[ba], 1)] = p;
    if (bb != ba) my_list[atomicAdd(&cnt[bb], 1)] = p;
  }
}

// grad_input without global atomics: one workgroup owns a band of `band_rows` rows of one plane,
// keeps it in LDS as doubles (ds_add_f64 runs ~10x faster than ds_add_f32 on gfx950 and the sum
// is rounded once), walks the points of its sample that touch the band (all points when no
// lists were built) and adds their corners, then stores the band with plain coalesced stores: no
// memset, every element written once.  The scattered global float atomics this replaces run at
// ~0.08 TB/s of added bytes when the 64 lanes of a wave hit 64 different lines, which is what
// random points do.
template <typename T>
__global__ __launch_bounds__(1024) void point_sample_bwd_band_kernel(
    const T* __restrict__ grad_out, const T* __restrict__ coords, T* __restrict__ grad_input,
    const int* __restrict__ list, const int* __restrict__ off, int C, int H, int W, int P,
    SampleGrid grid, int band_rows, int mode, int align) {
  extern __shared__ double band[];
  const SampleBlock blk = sample_block(grid);  // unit = plane (cper = 1), sub = band
  if (!blk.live) return;
  const int64_t n = blk.n;
  const int c = blk.cs;
  const int y_lo = blk.sub * band_rows, y_hi = min(H, y_lo + band_rows);  // [y_lo, y_hi)
  const int cells = (y_hi - y_lo) * W;
  for (int i = threadIdx.x; i < cells; i += 1024) band[i] = 0.0;
  __syncthreads();
  const T* go = grad_out + (n * C + c) * P;
  const T* xy = coords + n * P * 2;
  const int* my_list = list ? list + n * 2 * static_cast<int64_t>(P) : nullptr;
  const int e0 = list ? off[n * (kMaxBands + 1) + blk.sub] : 0;
  const int e1 = list ? off[n * (kMaxBands + 1) + blk.sub + 1] : P;
  for (int e = e0 + threadIdx.x; e < e1; e += 1024) {
    const int p = list ? my_list[e] : e;
    int ya, yb;
    T wa, wb;
    point_rows<T>(xy[2 * p + 1], H, mode, align, ya, yb, wa, wb);
    const bool in_a = ya >= y_lo && ya < y_hi, in_b = yb >= y_lo && yb < y_hi && yb != ya;
    if (!in_a && !in_b) continue;
    T mx;
    const T ix = border_coordinate<T>(xy[2 * p], W, align, mx);
    int xa, xb;
    T ua, ub;
    if (mode == kSampleNearest) {
      xa = xb = min(max(static_cast<int>(rint(ix)), 0), W - 1);
      ua = 1;
      ub = 0;
    } else {
      const T xf = floor(ix);
      xa = static_cast<int>(xf);
      xb = xa + 1;
      ub = ix - xf;
      ua = static_cast<T>(1) - ub;
      if (xb >= W) {
        xb = xa;
        ub = 0;
      }
    }
    const T g = go[p];
    const bool two_cols = xb != xa;
    if (in_a) {
      double* row = band + (ya - y_lo) * W;
      atomicAdd(row + xa, static_cast<double>(wa * ua * g));
      if (two_cols) atomicAdd(row + xb, static_cast<double>(wa * ub * g));
    }
    if (in_b) {
      double* row = band + (yb - y_lo) * W;
      atomicAdd(row + xa, static_cast<double>(wb * ua * g));
      if (two_cols) atomicAdd(row + xb, static_cast<double>(wb * ub * g));
    }
  }
  __syncthreads();
  T* dst = grad_input + ((n * C + c) * H + y_lo) * static_cast<int64_t>(W);
  for (int i = threadIdx.x; i < cells; i += 1024) dst[i] = static_cast<T>(band[i]);
}

// Channels per slice: enough slices to fill the chip when there are few points, whole samples per
// thread when there are many (the footprint is then computed once per point).
inline bool make_sample_grid(int N, int C, int P, SampleGrid* g, int64_t* blocks,
                             int points_per_block = 256) {
  const int pblocks = (P + points_per_block - 1) / points_per_block;
  int cper = C;
  // aim for >= 2048 workgroups; never fewer than 8 channels per slice unless C is smaller
  while (cper > 8 && static_cast<int64_t>(N) * pblocks * ((C + cper - 1) / cper) < 2048) cper = (cper + 1) / 2;
  const int cslices = (C + cper - 1) / cper;
  g->per = pblocks;
  g->cslices = cslices;
  g->cper = cper;
  g->units = static_cast<int64_t>(N) * cslices;
  *blocks = sample_grid_blocks(*g);
  return *blocks <= INT32_MAX;
}

// Band plan of the atomic-free grad_input kernel: rows per band so that a band of doubles fits
// `lds_bytes`, or 0 when the map is too wide or would need more than kMaxBands bands (every
// band's workgroup walks all points of its sample when no point lists were built).
inline int sample_band_rows(int H, int W, size_t lds_bytes) {
  const int64_t cells = static_cast<int64_t>(lds_bytes / sizeof(double));
  if (W > cells) return 0;
  int rows = static_cast<int>(cells / W);
  if (rows > H) rows = H;
  const int bands = (H + rows - 1) / rows;
  if (bands > kMaxBands) return 0;
  // even out the bands (the last one would otherwise be short)
  return (H + bands - 1) / bands;
}

// bytes of the band point lists of one backward call: per sample 2 * P point indices and
// kMaxBands + 1 offsets
inline size_t sample_lists_bytes(int N, int P) {
  return (static_cast<size_t>(N) * 2 * P + static_cast<size_t>(N) * (kMaxBands + 1)) * sizeof(int);
}

}  // namespace msda
