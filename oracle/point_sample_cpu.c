/*
 * TEST INFRASTRUCTURE — CPU oracle for PointRend point sampling.
 *
 * Scalar-loop restatement of `unipercept.vision.pointrend.sample(input, point_coords,
 * align_corners, mode=...)` (sources/unipercept/vision/pointrend.py:25-76 of the reference):
 *     grid_sample(input, 2 * point_coords - 1, mode, padding_mode="border", align_corners)
 * with point_coords (N, P, 2) in [0, 1]^2, last dimension (x, y), and the result (N, C, P).
 * The arithmetic lives in PyTorch's aten::grid_sampler_2d and its backward (third-party
 * dependency of the reference, torch >= 2.5.1, pyproject.toml:26 of the reference); what is
 * restated here is its published scalar algorithm:
 *     ix = ((gx + 1) * W - 1) / 2          (align_corners: (gx + 1) / 2 * (W - 1))
 *     border padding: ix = min(W - 1, max(ix, 0)), gradient 0 where the clamp acts
 *                     (ix <= 0 or ix >= W - 1)
 *     bilinear: corners floor(ix), floor(ix) + 1 (a corner outside the map is skipped)
 *     nearest:  nearbyint(ix), ties to even; no gradient for the coordinates
 * The factor 2 of `2 * point_coords - 1` is part of the coordinate gradient.
 *
 * Pinned by: the known-answer vector of the reference's own test
 * (tests/unipercept/vision/test_pointrend.py:6-13 of the reference) and the vectors under
 * tests/golden/pointrend_*.npz, generated in the build container by calling
 * torch.nn.functional.grid_sample exactly as the reference's function body does
 * (tests/golden/make_golden_pointrend.py); tests/test_pointrend_oracle.py checks this file
 * against both.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define PS_DEFINE(SUFFIX, T)                                                                      \
  static T ps_unnormalize_##SUFFIX(T g, int size, int align, T* mult) {                           \
    if (align) {                                                                                  \
      *mult = (T)(size - 1) / (T)2;                                                               \
      return ((g + (T)1) / (T)2) * (T)(size - 1);                                                 \
    }                                                                                             \
    *mult = (T)size / (T)2;                                                                       \
    return ((g + (T)1) * (T)size - (T)1) / (T)2;                                                  \
  }                                                                                               \
  static T ps_clip_##SUFFIX(T x, int size, T* mult) {                                             \
    if (!(x > (T)0)) { /* also NaN, as fmin/fmax in the original do */                            \
      *mult = (T)0;                                                                               \
      return (T)0;                                                                                \
    }                                                                                             \
    const T mx = (T)(size - 1);                                                                   \
    if (x >= mx) {                                                                                \
      *mult = (T)0;                                                                               \
      return mx;                                                                                  \
    }                                                                                             \
    return x;                                                                                     \
  }                                                                                               \
  void point_sample_cpu_forward_##SUFFIX(const T* input, const T* coords, T* out, int N, int C,   \
                                         int H, int W, int P, int mode, int align) {              \
    for (int n = 0; n < N; ++n)                                                                   \
      for (int p = 0; p < P; ++p) {                                                               \
        const T gx = (T)2 * coords[((int64_t)n * P + p) * 2] - (T)1;                              \
        const T gy = (T)2 * coords[((int64_t)n * P + p) * 2 + 1] - (T)1;                          \
        T mx, my, cx = (T)1, cy = (T)1;                                                           \
        T ix = ps_clip_##SUFFIX(ps_unnormalize_##SUFFIX(gx, W, align, &mx), W, &cx);              \
        T iy = ps_clip_##SUFFIX(ps_unnormalize_##SUFFIX(gy, H, align, &my), H, &cy);              \
        for (int c = 0; c < C; ++c) {                                                             \
          const T* plane = input + ((int64_t)n * C + c) * H * W;                                  \
          T* o = out + ((int64_t)n * C + c) * P + p;                                              \
          if (mode == 1) {                                                                        \
            const int xn = (int)nearbyint((double)ix), yn = (int)nearbyint((double)iy);           \
            *o = (xn >= 0 && xn < W && yn >= 0 && yn < H) ? plane[(int64_t)yn * W + xn] : (T)0;   \
            continue;                                                                             \
          }                                                                                       \
          const int x0 = (int)floor((double)ix), y0 = (int)floor((double)iy);                     \
          const T lx = ix - (T)x0, ly = iy - (T)y0, hx = (T)1 - lx, hy = (T)1 - ly;               \
          const T w[4] = {hy * hx, hy * lx, ly * hx, ly * lx};                                    \
          const int yy[4] = {y0, y0, y0 + 1, y0 + 1}, xx[4] = {x0, x0 + 1, x0, x0 + 1};           \
          T acc = (T)0;                                                                           \
          for (int k = 0; k < 4; ++k)                                                             \
            if (yy[k] >= 0 && yy[k] < H && xx[k] >= 0 && xx[k] < W)                               \
              acc += plane[(int64_t)yy[k] * W + xx[k]] * w[k];                                    \
          *o = acc;                                                                               \
        }                                                                                         \
      }                                                                                           \
  }                                                                                               \
  void point_sample_cpu_backward_##SUFFIX(const T* grad_out, const T* input, const T* coords,     \
                                          T* grad_input, T* grad_coords, int N, int C, int H,     \
                                          int W, int P, int mode, int align) {                    \
    memset(grad_input, 0, sizeof(T) * (size_t)N * C * H * W);                                     \
    for (int n = 0; n < N; ++n)                                                                   \
      for (int p = 0; p < P; ++p) {                                                               \
        const T gx = (T)2 * coords[((int64_t)n * P + p) * 2] - (T)1;                              \
        const T gy = (T)2 * coords[((int64_t)n * P + p) * 2 + 1] - (T)1;                          \
        T mx, my, cx = (T)1, cy = (T)1;                                                           \
        T ix = ps_clip_##SUFFIX(ps_unnormalize_##SUFFIX(gx, W, align, &mx), W, &cx);              \
        T iy = ps_clip_##SUFFIX(ps_unnormalize_##SUFFIX(gy, H, align, &my), H, &cy);              \
        T sx = (T)0, sy = (T)0;                                                                   \
        for (int c = 0; c < C; ++c) {                                                             \
          const T* plane = input + ((int64_t)n * C + c) * H * W;                                  \
          T* gplane = grad_input + ((int64_t)n * C + c) * H * W;                                  \
          const T g = grad_out[((int64_t)n * C + c) * P + p];                                     \
          if (mode == 1) {                                                                        \
            const int xn = (int)nearbyint((double)ix), yn = (int)nearbyint((double)iy);           \
            if (xn >= 0 && xn < W && yn >= 0 && yn < H) gplane[(int64_t)yn * W + xn] += g;        \
            continue;                                                                             \
          }                                                                                       \
          const int x0 = (int)floor((double)ix), y0 = (int)floor((double)iy);                     \
          const T lx = ix - (T)x0, ly = iy - (T)y0, hx = (T)1 - lx, hy = (T)1 - ly;               \
          const T w[4] = {hy * hx, hy * lx, ly * hx, ly * lx};                                    \
          const T dx[4] = {-hy, hy, -ly, ly}, dy[4] = {-hx, -lx, hx, lx};                         \
          const int yy[4] = {y0, y0, y0 + 1, y0 + 1}, xx[4] = {x0, x0 + 1, x0, x0 + 1};           \
          for (int k = 0; k < 4; ++k)                                                             \
            if (yy[k] >= 0 && yy[k] < H && xx[k] >= 0 && xx[k] < W) {                             \
              const int64_t at = (int64_t)yy[k] * W + xx[k];                                      \
              gplane[at] += w[k] * g;                                                             \
              sx += plane[at] * dx[k] * g;                                                        \
              sy += plane[at] * dy[k] * g;                                                        \
            }                                                                                     \
        }                                                                                         \
        /* d/d(point_coords) = 2 * d/d(grid) */                                                   \
        grad_coords[((int64_t)n * P + p) * 2] = (mode == 1) ? (T)0 : (T)2 * mx * cx * sx;         \
        grad_coords[((int64_t)n * P + p) * 2 + 1] = (mode == 1) ? (T)0 : (T)2 * my * cy * sy;     \
      }                                                                                           \
  }

PS_DEFINE(f32, float)
PS_DEFINE(f64, double)
