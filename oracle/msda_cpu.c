/*
 * TEST INFRASTRUCTURE — CPU oracle for multi-scale deformable attention.
 *
 * Scalar-loop restatement of the closed form of the reference's pure-PyTorch path
 * `reference.deform_attn(value, shapes, loc, attn)` (call site:
 * tests/unipercept/nn/layers/test_nn_layers_deform_attn.py:40-49 of the reference).  That path is
 * per-level `F.grid_sample(mode="bilinear", padding_mode="zeros", align_corners=False)` followed
 * by an attention-weighted sum; the arithmetic itself lives in PyTorch (aten::grid_sampler_2d and
 * its backward), which is a third-party dependency of the reference (torch >= 2.5.1,
 * pyproject.toml:26 of the reference).
 *
 * PARITY UNPINNED against the reference's own outputs: the reference holds no golden vectors for
 * this path and its implementation is absent from the snapshot.  The pin used instead is the set
 * of vectors under tests/golden/, generated in the build container from the canonical
 * grid_sample formulation (tests/golden/make_golden.py); tests/test_oracle.py checks this file
 * against them.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (unipercept_amd) never does.
 *
 * Conventions (all tensors contiguous, row-major):
 *   value  (N, S, M, D)          S = sum_l H_l*W_l
 *   shapes (L, 2) int64          (H_l, W_l)
 *   lsi    (L,)   int64          start offset of level l inside S
 *   loc    (N, Lq, M, L, P, 2)   last dim is (x, y), normalised to [0, 1]
 *   attn   (N, Lq, M, L, P)
 *   out    (N, Lq, M*D)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MSDA_CPU_DEFINE(SUFFIX, T)                                                                \
  void msda_cpu_forward_##SUFFIX(const T* value, const int64_t* shapes, const int64_t* lsi,       \
                                 const T* loc, const T* attn, T* out, int N, int S, int M, int D, \
                                 int Lq, int L, int P) {                                          \
    for (int b = 0; b < N; ++b)                                                                   \
      for (int q = 0; q < Lq; ++q)                                                                \
        for (int m = 0; m < M; ++m) {                                                             \
          T* o = out + (((int64_t)b * Lq + q) * M + m) * D;                                       \
          for (int c = 0; c < D; ++c) o[c] = (T)0;                                                \
          for (int l = 0; l < L; ++l) {                                                           \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                         \
            const T* vbase = value + ((int64_t)b * S + lsi[l]) * M * D + (int64_t)m * D;          \
            for (int p = 0; p < P; ++p) {                                                         \
              const int64_t pi = ((((int64_t)b * Lq + q) * M + m) * L + l) * P + p;               \
              const T x = loc[2 * pi] * (T)W - (T)0.5;                                            \
              const T y = loc[2 * pi + 1] * (T)H - (T)0.5;                                        \
              const T a = attn[pi];                                                               \
              if (!(y > (T)-1 && x > (T)-1 && y < (T)H && x < (T)W)) continue;                    \
              const int y0 = (int)floor((double)y), x0 = (int)floor((double)x);                   \
              const T ly = y - (T)y0, lx = x - (T)x0, hy = (T)1 - ly, hx = (T)1 - lx;             \
              const T w[4] = {hy * hx, hy * lx, ly * hx, ly * lx};                                \
              const int yy[4] = {y0, y0, y0 + 1, y0 + 1};                                         \
              const int xx[4] = {x0, x0 + 1, x0, x0 + 1};                                         \
              for (int k = 0; k < 4; ++k) {                                                       \
                if (yy[k] < 0 || yy[k] >= H || xx[k] < 0 || xx[k] >= W) continue;                 \
                const T* v = vbase + ((int64_t)yy[k] * W + xx[k]) * M * D;                        \
                const T aw = a * w[k];                                                            \
                for (int c = 0; c < D; ++c) o[c] += aw * v[c];                                    \
              }                                                                                   \
            }                                                                                     \
          }                                                                                       \
        }                                                                                         \
  }                                                                                               \
                                                                                                  \
  void msda_cpu_backward_##SUFFIX(const T* grad_out, const T* value, const int64_t* shapes,       \
                                  const int64_t* lsi, const T* loc, const T* attn, T* grad_value, \
                                  T* grad_loc, T* grad_attn, int N, int S, int M, int D, int Lq,  \
                                  int L, int P) {                                                 \
    memset(grad_value, 0, sizeof(T) * (size_t)N * S * M * D);                                     \
    for (int b = 0; b < N; ++b)                                                                   \
      for (int q = 0; q < Lq; ++q)                                                                \
        for (int m = 0; m < M; ++m) {                                                             \
          const T* g = grad_out + (((int64_t)b * Lq + q) * M + m) * D;                            \
          for (int l = 0; l < L; ++l) {                                                           \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                         \
            const int64_t base = ((int64_t)b * S + lsi[l]) * M * D + (int64_t)m * D;              \
            for (int p = 0; p < P; ++p) {                                                         \
              const int64_t pi = ((((int64_t)b * Lq + q) * M + m) * L + l) * P + p;               \
              grad_loc[2 * pi] = (T)0;                                                            \
              grad_loc[2 * pi + 1] = (T)0;                                                        \
              grad_attn[pi] = (T)0;                                                               \
              const T x = loc[2 * pi] * (T)W - (T)0.5;                                            \
              const T y = loc[2 * pi + 1] * (T)H - (T)0.5;                                        \
              const T a = attn[pi];                                                               \
              if (!(y > (T)-1 && x > (T)-1 && y < (T)H && x < (T)W)) continue;                    \
              const int y0 = (int)floor((double)y), x0 = (int)floor((double)x);                   \
              const T ly = y - (T)y0, lx = x - (T)x0, hy = (T)1 - ly, hx = (T)1 - lx;             \
              const T w[4] = {hy * hx, hy * lx, ly * hx, ly * lx};                                \
              const T dwx[4] = {-hy, hy, -ly, ly};                                                \
              const T dwy[4] = {-hx, -lx, hx, lx};                                                \
              const int yy[4] = {y0, y0, y0 + 1, y0 + 1};                                         \
              const int xx[4] = {x0, x0 + 1, x0, x0 + 1};                                         \
              T s_val = (T)0, s_dx = (T)0, s_dy = (T)0;                                           \
              for (int k = 0; k < 4; ++k) {                                                       \
                if (yy[k] < 0 || yy[k] >= H || xx[k] < 0 || xx[k] >= W) continue;                 \
                const int64_t off = base + ((int64_t)yy[k] * W + xx[k]) * M * D;                  \
                const T* v = value + off;                                                         \
                T* gv = grad_value + off;                                                         \
                const T aw = a * w[k];                                                            \
                T dot = (T)0;                                                                     \
                for (int c = 0; c < D; ++c) {                                                     \
                  dot += g[c] * v[c];                                                             \
                  gv[c] += aw * g[c];                                                             \
                }                                                                                 \
                s_val += w[k] * dot;                                                              \
                s_dx += dwx[k] * dot;                                                             \
                s_dy += dwy[k] * dot;                                                             \
              }                                                                                   \
              grad_attn[pi] = s_val;                                                              \
              grad_loc[2 * pi] = (T)W * a * s_dx;                                                 \
              grad_loc[2 * pi + 1] = (T)H * a * s_dy;                                             \
            }                                                                                     \
          }                                                                                       \
        }                                                                                         \
  }

MSDA_CPU_DEFINE(f32, float)
MSDA_CPU_DEFINE(f64, double)
