// Shared device helpers for the MSDeformAttn kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <tuple>
#include <utility>

namespace msda {

// ---- kernel launches ---------------------------------------------------------------------------
// Launches through hipLaunchKernel and keeps ITS return code: the status a call reports is that
// of its own launches, not whatever hipGetLastError() holds from earlier work of the thread
// (PyTorch's or anyone else's), and nobody else's pending error is cleared.  After a failed
// launch the later launches of the same call are skipped.
struct Launches {
  int rc = 0;  // 0 or the hipError_t of the first launch that failed

  template <typename... KArgs, typename... Args>
  void operator()(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds_bytes,
                  hipStream_t stream, Args... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count of the kernel");
    if (rc != 0) return;
    std::tuple<KArgs...> vals{static_cast<KArgs>(args)...};  // converted to the parameter types
    void* ptrs[sizeof...(KArgs) > 0 ? sizeof...(KArgs) : 1];
    fill(vals, ptrs, std::index_sequence_for<KArgs...>{});
    const hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs,
                                         lds_bytes, stream);
    rc = e == hipSuccess ? 0 : static_cast<int>(e);
  }

 private:
  template <typename Tuple, size_t... I>
  static void fill(Tuple& vals, void** out, std::index_sequence<I...>) {
    ((out[I] = const_cast<void*>(static_cast<const void*>(&std::get<I>(vals)))), ...);
  }
};

// ---- element types -------------------------------------------------------------------------
struct bf16_t {
  uint16_t bits;
};
using f16_t = _Float16;

template <typename T>
struct Elem;

template <>
struct Elem<float> {
  using compute_t = float;
  __device__ __forceinline__ static float load(const float* p) { return *p; }
  __device__ __forceinline__ static void store(float* p, float v) { *p = v; }
};
template <>
struct Elem<double> {
  using compute_t = double;
  __device__ __forceinline__ static double load(const double* p) { return *p; }
  __device__ __forceinline__ static void store(double* p, double v) { *p = v; }
};
template <>
struct Elem<bf16_t> {
  using compute_t = float;
  __device__ __forceinline__ static float load(const bf16_t* p) {
    return __uint_as_float(static_cast<uint32_t>(p->bits) << 16);
  }
  __device__ __forceinline__ static void store(bf16_t* p, float v) {
    // plain cast: v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even, NaN stays NaN)
    __bf16 h = static_cast<__bf16>(v);
    p->bits = __builtin_bit_cast(uint16_t, h);
  }
};
template <>
struct Elem<f16_t> {
  using compute_t = float;
  __device__ __forceinline__ static float load(const f16_t* p) { return static_cast<float>(*p); }
  __device__ __forceinline__ static void store(f16_t* p, float v) { *p = static_cast<f16_t>(v); }
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) {
  return __uint_as_float(bits16 << 16);
}
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float v) {
  __bf16 h = static_cast<__bf16>(v);
  return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, h));
}

// ---- one sampling point ----------------------------------------------------------------------
// Bilinear footprint of one (x, y) location on an H x W level, following grid_sample(bilinear,
// zeros, align_corners=False): pixel centres at integer + 0.5.  Corner k in {0,1,2,3} is
// (y0, x0), (y0, x0+1), (y0+1, x0), (y0+1, x0+1).
template <typename CT>
struct Footprint {
  CT w[4];      // bilinear weights (0 for a corner outside the level)
  CT dwx[4];    // d w / d x   (0 for a corner outside the level)
  CT dwy[4];    // d w / d y
  int pix[4];   // y*W + x of the corner, clamped into the level (valid address even when w == 0)
  int cy[2];    // clamped rows y0, y0+1    (corner k sits at row cy[k >> 1], column cx[k & 1])
  int cx[2];    // clamped columns x0, x0+1
  bool valid[4];  // corner lies inside the level (and the point inside the skirt)
  bool vy[2];   // row y0 / y0+1 lies inside the level: valid[k] = vy[k >> 1] && vx[k & 1]
  bool vx[2];   // column x0 / x0+1 lies inside the level
  CT lx, ly;    // fractional parts of the pixel coordinates (0 when !any)
  bool any;     // false when the point lies outside the (-1, H) x (-1, W) skirt: contributes nothing
};

// SMALL: H * W fits 24 bits (the caller's promise), so the row offsets come from the full-rate
// 24-bit multiply instead of the quarter-rate 32-bit one.
// pixel coordinate of a normalised location on an axis of n pixels (align_corners = False); one
// definition, so that every kernel rounds it the same way
template <typename CT>
__device__ __forceinline__ CT pixel_coord(CT loc, int n) {
  return loc * static_cast<CT>(n) - static_cast<CT>(0.5);
}

// The part of a footprint the origin tile kernel needs: whether the point lies inside the skirt,
// the footprint's origin + 1 (oy = y0 + 1 in 0 .. H, ox = x0 + 1 in 0 .. W) and the fractions —
// the same numbers footprint() derives from the same location.
__device__ __forceinline__ bool footprint_origin(float locx, float locy, int H, int W, int& oy,
                                                 int& ox, float& ly, float& lx) {
  const float x = pixel_coord(locx, W), y = pixel_coord(locy, H);
  const bool any = (y > -1.f) && (x > -1.f) && (y < static_cast<float>(H)) &&
                   (x < static_cast<float>(W)) && H > 0 && W > 0;
  const float xs = any ? x : 0.f, ys = any ? y : 0.f;
  const float xf = floorf(xs), yf = floorf(ys);
  lx = xs - xf;
  ly = ys - yf;
  ox = static_cast<int>(xf) + 1;
  oy = static_cast<int>(yf) + 1;
  return any;
}

template <typename CT, bool SMALL = false>
__device__ __forceinline__ Footprint<CT> footprint(CT locx, CT locy, int H, int W) {
  Footprint<CT> f;
  const CT x = pixel_coord(locx, W);
  const CT y = pixel_coord(locy, H);
  f.any = (y > static_cast<CT>(-1)) && (x > static_cast<CT>(-1)) && (y < static_cast<CT>(H)) &&
          (x < static_cast<CT>(W)) && H > 0 && W > 0;
  // clamp first so that the float->int conversion below is always defined (NaN/inf/huge inputs)
  const CT xs = f.any ? x : static_cast<CT>(0);
  const CT ys = f.any ? y : static_cast<CT>(0);
  const CT xf = floor(xs), yf = floor(ys);
  const int x0 = static_cast<int>(xf), y0 = static_cast<int>(yf);
  const CT lx = xs - xf, ly = ys - yf, hx = static_cast<CT>(1) - lx, hy = static_cast<CT>(1) - ly;
  const bool vx0 = f.any && x0 >= 0, vx1 = f.any && x0 + 1 < W;
  const bool vy0 = f.any && y0 >= 0, vy1 = f.any && y0 + 1 < H;
  const int xc0 = max(x0, 0), xc1 = max(min(x0 + 1, W - 1), 0);
  const int yc0 = max(y0, 0), yc1 = max(min(y0 + 1, H - 1), 0);
  const CT z = static_cast<CT>(0);
  const bool v[4] = {vy0 && vx0, vy0 && vx1, vy1 && vx0, vy1 && vx1};
  const CT w[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
  const CT dx[4] = {-hy, hy, -ly, ly};
  const CT dy[4] = {-hx, -lx, hx, lx};
  f.vy[0] = vy0;
  f.vy[1] = vy1;
  f.vx[0] = vx0;
  f.vx[1] = vx1;
  f.lx = lx;
  f.ly = ly;
  f.cy[0] = yc0;
  f.cy[1] = yc1;
  f.cx[0] = xc0;
  f.cx[1] = xc1;
  const int r0 = SMALL ? static_cast<int>(__umul24(yc0, W)) : yc0 * W;  // (both non-negative)
  const int r1 = SMALL ? static_cast<int>(__umul24(yc1, W)) : yc1 * W;
  f.pix[0] = r0 + xc0;
  f.pix[1] = r0 + xc1;
  f.pix[2] = r1 + xc0;
  f.pix[3] = r1 + xc1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f.valid[k] = v[k];
    f.w[k] = v[k] ? w[k] : z;
    f.dwx[k] = v[k] ? dx[k] : z;
    f.dwy[k] = v[k] ? dy[k] : z;
  }
  return f;
}

}  // namespace msda
