// libmsda: C ABI + launch logic of the MSDeformAttn kernels for gfx950 (see include/msda.h).
#include "../../include/msda.h"

#include <hip/hip_runtime.h>

#include <mutex>

#include "msda_generic.h"
#include "msda_points.h"
#include "msda_sample.h"
#include "msda_tuned.h"

namespace {

using namespace msda;

constexpr int kAbiVersion = 8;

inline size_t elem_size(int dtype) {
  switch (dtype) {
    case MSDA_F32: return 4;
    case MSDA_F64: return 8;
    case MSDA_BF16:
    case MSDA_F16: return 2;
    default: return 0;
  }
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

inline int check_common(int N, int S, int M, int D, int Lq, int L, int P, int dtype_val,
                        int dtype_loc) {
  if (N < 1 || S < 1 || M < 1 || D < 1 || Lq < 1 || L < 1 || P < 1) return MSDA_ESHAPE;
  if (L > kMaxLevels || static_cast<int64_t>(L) * P > 128) return MSDA_ELEVELS;
  const int64_t lim = INT32_MAX;
  if (static_cast<int64_t>(N) * Lq * M > lim || static_cast<int64_t>(S) * M * D > lim ||
      static_cast<int64_t>(M) * D > (1 << 24))
    return MSDA_ESHAPE;
  if (elem_size(dtype_val) == 0 || elem_size(dtype_loc) == 0) return MSDA_EDTYPE;
  const bool same = dtype_val == dtype_loc;
  const bool mixed = (dtype_val == MSDA_BF16 || dtype_val == MSDA_F16) && dtype_loc == MSDA_F32;
  if (!same && !mixed) return MSDA_EDTYPE;
  return MSDA_OK;
}

inline int group_lanes(int D) { return D <= 8 ? 8 : D <= 16 ? 16 : D <= 32 ? 32 : 64; }

struct Dims {
  int N, S, M, D, Lq, L, P;
};

// ---- generic forward ------------------------------------------------------------------------
template <typename VT, typename LT>
int fwd_generic(const void* value, const int64_t* shapes, const int64_t* lsi, const void* loc,
                const void* attn, void* out, const Dims& d, hipStream_t stream) {
  const int64_t groups = static_cast<int64_t>(d.N) * d.Lq * d.M;
  const int G = group_lanes(d.D);
  const int64_t blocks = (groups + (256 / G) - 1) / (256 / G);
  if (blocks > INT32_MAX) return MSDA_ESHAPE;
  Launches run;
#define MSDA_LAUNCH(GG)                                                                          \
  run((fwd_generic_kernel<VT, LT, GG>), dim3(static_cast<unsigned>(blocks)),      \
                     dim3(256), 0, stream, static_cast<const VT*>(value), shapes, lsi,           \
                     static_cast<const LT*>(loc), static_cast<const LT*>(attn),                  \
                     static_cast<VT*>(out), groups, d.S, d.M, d.D, d.Lq, d.L, d.P)
  switch (G) {
    case 8: MSDA_LAUNCH(8); break;
    case 16: MSDA_LAUNCH(16); break;
    case 32: MSDA_LAUNCH(32); break;
    default: MSDA_LAUNCH(64); break;
  }
#undef MSDA_LAUNCH
  return run.rc;
}

// ---- generic backward -----------------------------------------------------------------------
template <typename VT, typename LT, typename GT>
int bwd_generic(const void* grad_out, const void* value, const int64_t* shapes, const int64_t* lsi,
                const void* loc, const void* attn, void* grad_value, void* grad_loc,
                void* grad_attn, void* workspace, const Dims& d, bool deterministic,
                hipStream_t stream) {
  const int64_t groups = static_cast<int64_t>(d.N) * d.Lq * d.M;
  const int64_t total = static_cast<int64_t>(d.N) * d.S * d.M * d.D;
  const int G = group_lanes(d.D);
  const int64_t blocks = (groups + (256 / G) - 1) / (256 / G);
  if (blocks > INT32_MAX) return MSDA_ESHAPE;
  constexpr bool kDirect = sizeof(GT) == sizeof(VT);  // atomics go straight into grad_value
  GT* acc = kDirect ? static_cast<GT*>(grad_value) : static_cast<GT*>(workspace);
  if (!deterministic) {
    const hipError_t e = hipMemsetAsync(acc, 0, static_cast<size_t>(total) * sizeof(GT), stream);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  Launches run;
#define MSDA_LAUNCH(GG, SC)                                                                      \
  run((bwd_generic_kernel<VT, LT, GT, GG, SC>),                                   \
                     dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream,                  \
                     static_cast<const VT*>(grad_out), static_cast<const VT*>(value), shapes,    \
                     lsi, static_cast<const LT*>(loc), static_cast<const LT*>(attn), acc,        \
                     static_cast<LT*>(grad_loc), static_cast<LT*>(grad_attn), groups, d.S, d.M,  \
                     d.D, d.Lq, d.L, d.P)
#define MSDA_LAUNCH_G(SC)                 \
  switch (G) {                            \
    case 8: MSDA_LAUNCH(8, SC); break;    \
    case 16: MSDA_LAUNCH(16, SC); break;  \
    case 32: MSDA_LAUNCH(32, SC); break;  \
    default: MSDA_LAUNCH(64, SC); break;  \
  }
  if (deterministic) {
    MSDA_LAUNCH_G(false)
  } else {
    MSDA_LAUNCH_G(true)
  }
#undef MSDA_LAUNCH_G
#undef MSDA_LAUNCH
  if (deterministic) {
    const int64_t vb = (total + 255) / 256;
    if (vb > INT32_MAX) return MSDA_ESHAPE;
    run((bwd_value_gather_kernel<VT, LT>), dim3(static_cast<unsigned>(vb)),
                       dim3(256), 0, stream, static_cast<const VT*>(grad_out), shapes, lsi,
                       static_cast<const LT*>(loc), static_cast<const LT*>(attn),
                       static_cast<VT*>(grad_value), total, d.S, d.M, d.D, d.Lq, d.L, d.P);
    return run.rc;
  }
  if constexpr (!kDirect) {
    const int64_t cb = (total + 255) / 256;
    const unsigned grid = static_cast<unsigned>(cb < 8192 ? cb : 8192);
    run((convert_from_f32_kernel<VT>), dim3(grid), dim3(256), 0, stream,
                       static_cast<const float*>(workspace), static_cast<VT*>(grad_value), total);
  }
  return run.rc;
}

inline bool needs_f32_image(int dtype_val) { return dtype_val == MSDA_BF16 || dtype_val == MSDA_F16; }

int backward_impl(const void* grad_out, const void* value, const int64_t* shapes_dev,
                  const int64_t* lsi_dev, const void* loc, const void* attn,
                  const void* ref_points, void* grad_value, void* grad_loc, void* grad_attn,
                  void* workspace, size_t workspace_bytes, int N,
                  int S, int M, int D, int Lq, int L, int P, int dtype_val, int dtype_loc,
                  int flags, const int64_t* shapes_host, void* hip_stream);

// ---- msda_backward_noalloc: the backward's scratch as a static device array ------------------
// 128 MiB per GPU the library runs on, held until the process ends: enough for the backward of
// the CPU-reference configuration (512 x 1024 input, batch 1: 78 MiB) and anything smaller.
constexpr size_t kNoAllocScratchBytes = static_cast<size_t>(128) << 20;
__device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

// In host code `scratch` is only a handle for the symbol, not a device address.  Every GPU has
// its own copy of the array at its own address: it is looked up once per device, with that
// device current, and every later use goes through the pointer kept here.
constexpr int kMaxDevices = 64;
std::mutex scratch_mutex;
void* scratch_ptr[kMaxDevices] = {};

// The array that belongs to the GPU the launch will run on: the device of `stream`, or the
// current device for the default stream.
int noalloc_scratch(hipStream_t stream, void** out) {
  int cur = -1;
  hipError_t e = hipGetDevice(&cur);
  if (e != hipSuccess) return static_cast<int>(e);
  int dev = cur;
  if (stream != nullptr) {
    hipDevice_t sd;
    e = hipStreamGetDevice(stream, &sd);
    if (e != hipSuccess) return static_cast<int>(e);
    dev = static_cast<int>(sd);
  }
  if (dev < 0 || dev >= kMaxDevices) return MSDA_ENOALLOC;
  std::lock_guard<std::mutex> lock(scratch_mutex);
  if (scratch_ptr[dev] == nullptr) {
    // the symbol resolves to the copy of the device that is current during the lookup
    if (dev != cur) {
      e = hipSetDevice(dev);
      if (e != hipSuccess) return static_cast<int>(e);
    }
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (dev != cur) {
      const hipError_t back = hipSetDevice(cur);
      if (e == hipSuccess) e = back;
    }
    if (e != hipSuccess) return static_cast<int>(e);
    if (p == nullptr || reinterpret_cast<uintptr_t>(p) % 256 != 0) return MSDA_ENOALLOC;
    scratch_ptr[dev] = p;
  }
  *out = scratch_ptr[dev];
  return MSDA_OK;
}

}  // namespace

extern "C" {

int msda_version(void) { return kAbiVersion; }

#if MSDA_STAMPS
// diagnostic builds only: copies the per-workgroup phase cycle sums of kernel `which` (0 query
// kernel, 1 tile kernel) to host memory; returns the number of workgroup rows
int msda_debug_read_stamps(int which, unsigned long long* host, int max_groups) {
  const int n = max_groups < msda::kStampGroups ? max_groups : msda::kStampGroups;
  hipDeviceSynchronize();
  const hipError_t e = hipMemcpyFromSymbol(
      host, HIP_SYMBOL(msda::stamp_buf), static_cast<size_t>(n) * msda::kStampSlots * 8,
      static_cast<size_t>(which) * msda::kStampGroups * msda::kStampSlots * 8, hipMemcpyDeviceToHost);
  return e == hipSuccess ? n : -static_cast<int>(e);
}
#endif

const char* msda_error_string(int code) {
  switch (code) {
    case MSDA_OK: return "success";
    case MSDA_ENULL: return "msda: a required pointer is NULL";
    case MSDA_ESHAPE: return "msda: invalid dimension (must be >= 1 and within 32-bit index limits)";
    case MSDA_EDTYPE: return "msda: unsupported dtype or dtype combination";
    case MSDA_EALIGN: return "msda: pointer not aligned to its element type";
    case MSDA_EWORKSPACE: return "msda: workspace missing, misaligned or too small";
    case MSDA_ESTEP: return "msda: im2col_step must be >= 1 and min(N, im2col_step) must divide N";
    case MSDA_ELEVELS: return "msda: too many levels/points (L <= 16, L*P <= 128)";
    case MSDA_ELAYOUT:
      return "msda: head-major value / packed point layout is not supported for this shape/path";
    case MSDA_ENOALLOC:
      return "msda: msda_backward_noalloc: the problem needs more scratch than the library's "
             "static 128 MiB array (see msda_backward_workspace_bytes), or the device ordinal is "
             "beyond 63";
    default:
      return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "msda: unknown error";
  }
}

int msda_check_im2col_step(int N, int im2col_step) {
  if (N < 1 || im2col_step < 1) return MSDA_ESTEP;
  const int step = N < im2col_step ? N : im2col_step;
  return (N % step == 0) ? MSDA_OK : MSDA_ESTEP;
}

int msda_forward(const void* value, const int64_t* shapes_dev, const int64_t* lsi_dev,
                 const void* loc, const void* attn, const void* ref_points, void* out, int N,
                 int S, int M, int D, int Lq, int L, int P, int dtype_val, int dtype_loc,
                 int flags, void* hip_stream) {
  const bool raw = (flags & MSDA_FLAG_RAW_POINTS) != 0;  // raw points come in the packed row form
  const bool packed = raw || (flags & MSDA_FLAG_PACKED_POINTS) != 0;  // then `attn` is not looked at
  if (!value || !shapes_dev || !lsi_dev || !loc || (!attn && !packed) || !out) return MSDA_ENULL;
  if (raw && !ref_points) return MSDA_ENULL;
  if (raw && !aligned(ref_points, 8)) return MSDA_EALIGN;
  const int rc = check_common(N, S, M, D, Lq, L, P, dtype_val, dtype_loc);
  if (rc != MSDA_OK) return rc;
  if (!aligned(value, elem_size(dtype_val)) || !aligned(out, elem_size(dtype_val)) ||
      !aligned(loc, elem_size(dtype_loc)) || (!packed && !aligned(attn, elem_size(dtype_loc))) ||
      !aligned(shapes_dev, 8) || !aligned(lsi_dev, 8))
    return MSDA_EALIGN;
  const Dims d{N, S, M, D, Lq, L, P};
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);

  if (!(flags & MSDA_FLAG_GENERIC)) {
    if (raw && !raw_points_supported(N, S, M, D, Lq, L, P, dtype_val, dtype_loc)) return MSDA_ELAYOUT;
    const int t = tuned_forward(value, shapes_dev, lsi_dev, loc, attn, out, N, S, M, D, Lq, L, P,
                                dtype_val, dtype_loc, (flags & MSDA_FLAG_VALUE_HEAD_MAJOR) != 0,
                                packed, raw ? static_cast<const float*>(ref_points) : nullptr,
                                stream);
    if (t != kNotTuned) return t;
  }
  if ((flags & MSDA_FLAG_VALUE_HEAD_MAJOR) || packed) return MSDA_ELAYOUT;
  switch (dtype_val) {
    case MSDA_F32:
      return fwd_generic<float, float>(value, shapes_dev, lsi_dev, loc, attn, out, d, stream);
    case MSDA_F64:
      return fwd_generic<double, double>(value, shapes_dev, lsi_dev, loc, attn, out, d, stream);
    case MSDA_BF16:
      return dtype_loc == MSDA_F32
                 ? fwd_generic<bf16_t, float>(value, shapes_dev, lsi_dev, loc, attn, out, d, stream)
                 : fwd_generic<bf16_t, bf16_t>(value, shapes_dev, lsi_dev, loc, attn, out, d, stream);
    case MSDA_F16:
      return dtype_loc == MSDA_F32
                 ? fwd_generic<f16_t, float>(value, shapes_dev, lsi_dev, loc, attn, out, d, stream)
                 : fwd_generic<f16_t, f16_t>(value, shapes_dev, lsi_dev, loc, attn, out, d, stream);
  }
  return MSDA_EDTYPE;
}

int msda_head_major_supported(int N, int S, int M, int D, int Lq, int L, int P, int dtype_val,
                              int dtype_loc) {
  if (check_common(N, S, M, D, Lq, L, P, dtype_val, dtype_loc) != MSDA_OK) return 0;
  return head_major_supported(N, S, M, D, Lq, L, P, dtype_val, dtype_loc) ? 1 : 0;
}

int msda_packed_points_supported(int N, int S, int M, int D, int Lq, int L, int P, int dtype_val,
                                 int dtype_loc) {
  if (check_common(N, S, M, D, Lq, L, P, dtype_val, dtype_loc) != MSDA_OK) return 0;
  int G;
  return tuned_shape_ok(N, S, M, D, Lq, L, P, dtype_val, dtype_loc, &G) && packed_shape_ok(L, P)
             ? 1
             : 0;
}

int msda_raw_points_supported(int N, int S, int M, int D, int Lq, int L, int P, int dtype_val,
                              int dtype_loc) {
  if (check_common(N, S, M, D, Lq, L, P, dtype_val, dtype_loc) != MSDA_OK) return 0;
  return raw_points_supported(N, S, M, D, Lq, L, P, dtype_val, dtype_loc) ? 1 : 0;
}

int msda_value_relayout(const void* src, void* dst, const unsigned char* padding_mask, int N,
                        int S, int M, int D, int dtype_val, int to_head_major, void* hip_stream) {
  if (!src || !dst) return MSDA_ENULL;
  if (N < 1 || S < 1 || M < 1 || D < 1) return MSDA_ESHAPE;
  const size_t e = elem_size(dtype_val);
  if (e == 0) return MSDA_EDTYPE;
  if ((static_cast<size_t>(D) * e) % 16 != 0) return MSDA_ELAYOUT;
  if (!aligned(src, 16) || !aligned(dst, 16)) return MSDA_EALIGN;
  const int chunks = static_cast<int>(static_cast<size_t>(D) * e / 16);
  const int64_t rows = static_cast<int64_t>(N) * S * M;
  const int64_t blocks = (rows * chunks + 255) / 256;
  const unsigned grid = static_cast<unsigned>(blocks < 16384 ? blocks : 16384);
  // to head-major: src (N,S,M,.) -> dst (N,M,S,.); back: src (N,M,S,.) -> dst (N,S,M,.)
  Launches run;
  run(relayout_rows_kernel, dim3(grid), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), static_cast<const uint4*>(src),
                     static_cast<uint4*>(dst), padding_mask, rows, to_head_major ? S : M,
                     to_head_major ? M : S, chunks, to_head_major ? 1 : 0);
  return run.rc;
}

size_t msda_backward_workspace_bytes(int N, int S, int M, int D, int Lq, int L, int P,
                                     int dtype_val, int dtype_loc, int flags,
                                     const int64_t* shapes_host) {
  if (check_common(N, S, M, D, Lq, L, P, dtype_val, dtype_loc) != MSDA_OK) return 0;
  const bool det = (flags & MSDA_FLAG_DETERMINISTIC) != 0;
  if (!(flags & MSDA_FLAG_GENERIC)) {
    const int64_t t = tuned_backward_workspace_bytes(
        N, S, M, D, Lq, L, P, dtype_val, dtype_loc, det || !(flags & MSDA_FLAG_ATOMIC_SCATTER),
        (flags & MSDA_FLAG_TILE_1D) ? nullptr : shapes_host, det);
    if (t >= 0) return static_cast<size_t>(t);
  }
  if (det) return 0;  // the per-destination gather needs none
  if (needs_f32_image(dtype_val)) return static_cast<size_t>(N) * S * M * D * sizeof(float);
  return 0;
}

int msda_backward(const void* grad_out, const void* value, const int64_t* shapes_dev,
                  const int64_t* lsi_dev, const void* loc, const void* attn,
                  const void* ref_points, void* grad_value, void* grad_loc, void* grad_attn,
                  void* workspace, size_t workspace_bytes, int N,
                  int S, int M, int D, int Lq, int L, int P, int dtype_val, int dtype_loc,
                  int flags, const int64_t* shapes_host, void* hip_stream) {
  return backward_impl(grad_out, value, shapes_dev, lsi_dev, loc, attn, ref_points, grad_value,
                       grad_loc, grad_attn, workspace, workspace_bytes, N, S, M, D, Lq, L, P,
                       dtype_val, dtype_loc, flags, shapes_host, hip_stream);
}

int msda_backward_noalloc(const void* grad_out, const void* value, const int64_t* shapes_dev,
                          const int64_t* lsi_dev, const void* loc, const void* attn,
                          const void* ref_points, void* grad_value, void* grad_loc,
                          void* grad_attn, int N, int S, int M,
                          int D, int Lq, int L, int P, int dtype_val, int dtype_loc, int flags,
                          const int64_t* shapes_host, void* hip_stream) {
  // size check first: a problem beyond the static array is refused before any HIP call
  const size_t need = msda_backward_workspace_bytes(N, S, M, D, Lq, L, P, dtype_val, dtype_loc,
                                                    flags, shapes_host);
  if (need > kNoAllocScratchBytes) return MSDA_ENOALLOC;
  void* ws = nullptr;
  if (need > 0) {
    const int rc = noalloc_scratch(static_cast<hipStream_t>(hip_stream), &ws);
    if (rc != MSDA_OK) return rc;
  }
  return backward_impl(grad_out, value, shapes_dev, lsi_dev, loc, attn, ref_points, grad_value,
                       grad_loc, grad_attn, ws, need > 0 ? kNoAllocScratchBytes : 0, N, S, M, D,
                       Lq, L, P, dtype_val, dtype_loc, flags, shapes_host, hip_stream);
}

}  // extern "C"

namespace {
int backward_impl(const void* grad_out, const void* value, const int64_t* shapes_dev,
                  const int64_t* lsi_dev, const void* loc, const void* attn,
                  const void* ref_points, void* grad_value, void* grad_loc, void* grad_attn,
                  void* workspace, size_t workspace_bytes, int N,
                  int S, int M, int D, int Lq, int L, int P, int dtype_val, int dtype_loc,
                  int flags, const int64_t* shapes_host, void* hip_stream) {
  const bool raw = (flags & MSDA_FLAG_RAW_POINTS) != 0;
  const bool packed = raw || (flags & MSDA_FLAG_PACKED_POINTS) != 0;  // attn / grad_attn not looked at
  if (!grad_out || !value || !shapes_dev || !lsi_dev || !loc || !grad_value || !grad_loc ||
      (!packed && (!attn || !grad_attn)) || (raw && !ref_points))
    return MSDA_ENULL;
  if (raw && !aligned(ref_points, 8)) return MSDA_EALIGN;
  if (raw && ((flags & (MSDA_FLAG_GENERIC | MSDA_FLAG_ATOMIC_SCATTER)) ||
              !raw_points_supported(N, S, M, D, Lq, L, P, dtype_val, dtype_loc)))
    return MSDA_ELAYOUT;
  const int rc = check_common(N, S, M, D, Lq, L, P, dtype_val, dtype_loc);
  if (rc != MSDA_OK) return rc;
  const size_t ev = elem_size(dtype_val), el = elem_size(dtype_loc);
  if (!aligned(grad_out, ev) || !aligned(value, ev) || !aligned(grad_value, ev) ||
      !aligned(loc, el) || !aligned(grad_loc, el) ||
      (!packed && (!aligned(attn, el) || !aligned(grad_attn, el))) || !aligned(shapes_dev, 8) ||
      !aligned(lsi_dev, 8))
    return MSDA_EALIGN;
  const size_t need =
      msda_backward_workspace_bytes(N, S, M, D, Lq, L, P, dtype_val, dtype_loc, flags,
                                    shapes_host);
  if (need > 0 && (!workspace || workspace_bytes < need || !aligned(workspace, 256)))
    return MSDA_EWORKSPACE;
  const Dims d{N, S, M, D, Lq, L, P};
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const bool det = (flags & MSDA_FLAG_DETERMINISTIC) != 0;
  // the two halves of the binned backward as separate calls
  const bool binned_given = (flags & MSDA_FLAG_BINNED) != 0, bin_only = (flags & MSDA_FLAG_BIN_ONLY) != 0;
  if (binned_given && bin_only) return MSDA_ELAYOUT;
  int phase = binned_given ? kPhaseGrad : bin_only ? kPhaseBin : (kPhaseBin | kPhaseGrad);
  if (phase != (kPhaseBin | kPhaseGrad) && (flags & (MSDA_FLAG_GENERIC | MSDA_FLAG_ATOMIC_SCATTER)))
    return MSDA_ELAYOUT;
#if MSDA_EXP_PHASES  // experiment builds: one of the two kernels of the binned backward alone
  if (flags & (1 << 20)) phase |= kPhaseNoTile;
  if (flags & (1 << 21)) phase |= kPhaseNoQuery;
#endif

  // MSDA_FLAG_DETERMINISTIC: the binned path with its reproducible tile kernel where the shape
  // has one (ignoring MSDA_FLAG_ATOMIC_SCATTER), else the generic per-destination gather
  if (!(flags & MSDA_FLAG_GENERIC)) {
    const int t = tuned_backward(grad_out, value, shapes_dev, lsi_dev, loc, attn, grad_value,
                                 grad_loc, grad_attn, workspace, N, S, M, D, Lq, L, P, dtype_val,
                                 dtype_loc, det || !(flags & MSDA_FLAG_ATOMIC_SCATTER),
                                 (flags & MSDA_FLAG_VALUE_HEAD_MAJOR) != 0, packed,
                                 (flags & MSDA_FLAG_TILE_1D) ? nullptr : shapes_host,
                                 raw ? static_cast<const float*>(ref_points) : nullptr,
                                 phase, det, stream);
    if (t != kNotTuned) return t;
  }
  if ((flags & MSDA_FLAG_VALUE_HEAD_MAJOR) || packed || phase != (kPhaseBin | kPhaseGrad))
    return MSDA_ELAYOUT;
#define MSDA_BWD(VT, LT, GT)                                                                  \
  bwd_generic<VT, LT, GT>(grad_out, value, shapes_dev, lsi_dev, loc, attn, grad_value,        \
                          grad_loc, grad_attn, workspace, d, det, stream)
  switch (dtype_val) {
    case MSDA_F32: return MSDA_BWD(float, float, float);
    case MSDA_F64: return MSDA_BWD(double, double, double);
    case MSDA_BF16:
      return dtype_loc == MSDA_F32 ? MSDA_BWD(bf16_t, float, float) : MSDA_BWD(bf16_t, bf16_t, float);
    case MSDA_F16:
      return dtype_loc == MSDA_F32 ? MSDA_BWD(f16_t, float, float) : MSDA_BWD(f16_t, f16_t, float);
  }
#undef MSDA_BWD
  return MSDA_EDTYPE;
}
}  // namespace

extern "C" {

namespace {
// raw / packed element types of the point epilogue: equal, or a 16-bit raw with fp32 packed
inline int check_point_dtypes(int dtype_raw, int dtype_packed) {
  const bool ok_raw = dtype_raw == MSDA_F32 || dtype_raw == MSDA_BF16 || dtype_raw == MSDA_F16;
  if (!ok_raw) return MSDA_EDTYPE;
  if (dtype_packed == dtype_raw) return MSDA_OK;
  return dtype_packed == MSDA_F32 ? MSDA_OK : MSDA_EDTYPE;
}
}  // namespace

int msda_points_forward(const void* raw, const void* ref_points, const int64_t* shapes_dev,
                        void* packed, int N, int Lq, int M, int L, int P, int dtype_raw,
                        int dtype_packed, void* hip_stream) {
  if (!raw || !ref_points || !shapes_dev || !packed) return MSDA_ENULL;
  if (N < 1 || Lq < 1 || M < 1) return MSDA_ESHAPE;
  if (L < 1 || P < 1 || L > 16 || L * P > 128) return MSDA_ELEVELS;
  if (check_point_dtypes(dtype_raw, dtype_packed) != MSDA_OK) return MSDA_EDTYPE;
  if (!aligned(raw, 16) || !aligned(packed, 16) || !aligned(ref_points, elem_size(dtype_packed)) ||
      !aligned(shapes_dev, 8))
    return MSDA_EALIGN;
  const int64_t rows = static_cast<int64_t>(N) * Lq * M;
  const int rpb = point_rows_per_block(L, P, false);
  const int64_t blocks = (rows + rpb - 1) / rpb;
  if (blocks > INT32_MAX) return MSDA_ESHAPE;
  const size_t lds = points_lds_bytes(L, P, rpb, false);
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  Launches run;
#define MSDA_PF(RT, LT)                                                                       \
  run((points_forward_kernel<RT, LT>), dim3(static_cast<unsigned>(blocks)),    \
                     dim3(256), lds, stream, static_cast<const RT*>(raw),                     \
                     static_cast<const LT*>(ref_points), shapes_dev, static_cast<LT*>(packed), \
                     rows, M, L, P, rpb)
  switch (dtype_raw) {
    case MSDA_F32: MSDA_PF(float, float); break;
    case MSDA_BF16:
      if (dtype_packed == MSDA_F32) MSDA_PF(bf16_t, float); else MSDA_PF(bf16_t, bf16_t);
      break;
    default:
      if (dtype_packed == MSDA_F32) MSDA_PF(f16_t, float); else MSDA_PF(f16_t, f16_t);
      break;
  }
#undef MSDA_PF
  return run.rc;
}

int msda_points_backward(const void* grad_packed, const void* packed, const int64_t* shapes_dev,
                         void* grad_raw, int N, int Lq, int M, int L, int P, int dtype_raw,
                         int dtype_packed, void* hip_stream) {
  if (!grad_packed || !packed || !shapes_dev || !grad_raw) return MSDA_ENULL;
  if (N < 1 || Lq < 1 || M < 1) return MSDA_ESHAPE;
  if (L < 1 || P < 1 || L > 16 || L * P > 128) return MSDA_ELEVELS;
  if (check_point_dtypes(dtype_raw, dtype_packed) != MSDA_OK) return MSDA_EDTYPE;
  if (!aligned(grad_packed, 16) || !aligned(packed, 16) || !aligned(grad_raw, 16) ||
      !aligned(shapes_dev, 8))
    return MSDA_EALIGN;
  const int64_t rows = static_cast<int64_t>(N) * Lq * M;
  const int rpb = point_rows_per_block(L, P, true);
  const int64_t blocks = (rows + rpb - 1) / rpb;
  if (blocks > INT32_MAX) return MSDA_ESHAPE;
  const size_t lds = points_lds_bytes(L, P, rpb, true);
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  Launches run;
#define MSDA_PB(RT, LT)                                                                        \
  run((points_backward_kernel<RT, LT>), dim3(static_cast<unsigned>(blocks)),    \
                     dim3(256), lds, stream, static_cast<const LT*>(grad_packed),              \
                     static_cast<const LT*>(packed), shapes_dev, static_cast<RT*>(grad_raw),   \
                     rows, L, P, rpb)
  switch (dtype_raw) {
    case MSDA_F32: MSDA_PB(float, float); break;
    case MSDA_BF16:
      if (dtype_packed == MSDA_F32) MSDA_PB(bf16_t, float); else MSDA_PB(bf16_t, bf16_t);
      break;
    default:
      if (dtype_packed == MSDA_F32) MSDA_PB(f16_t, float); else MSDA_PB(f16_t, f16_t);
      break;
  }
#undef MSDA_PB
  return run.rc;
}

int msda_point_sample_forward(const void* input, const void* coords, void* out, int N, int C,
                              int H, int W, int P, int dtype, int mode, int align_corners,
                              void* hip_stream) {
  if (!input || !coords || !out) return MSDA_ENULL;
  if (N < 1 || C < 1 || H < 1 || W < 1 || P < 1) return MSDA_ESHAPE;
  if (static_cast<int64_t>(H) * W > INT32_MAX) return MSDA_ESHAPE;
  const size_t e = elem_size(dtype);
  if (e == 0 || (mode != MSDA_SAMPLE_BILINEAR && mode != MSDA_SAMPLE_NEAREST)) return MSDA_EDTYPE;
  if (!aligned(input, e) || !aligned(coords, e) || !aligned(out, e)) return MSDA_EALIGN;
  SampleGrid g;
  int64_t blocks;
  if (!make_sample_grid(N, C, P, &g, &blocks, 256 * kSamplePPT)) return MSDA_ESHAPE;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  Launches run;
#define MSDA_SF(T)                                                                              \
  run((point_sample_fwd_kernel<T>), dim3(static_cast<unsigned>(blocks)),         \
                     dim3(256), 0, stream, static_cast<const T*>(input),                        \
                     static_cast<const T*>(coords), static_cast<T*>(out), C, H, W, P, g, mode,  \
                     align_corners ? 1 : 0)
  switch (dtype) {
    case MSDA_F32: MSDA_SF(float); break;
    case MSDA_F64: MSDA_SF(double); break;
    case MSDA_BF16: MSDA_SF(bf16_t); break;
    default: MSDA_SF(f16_t); break;
  }
#undef MSDA_SF
  return run.rc;
}

size_t msda_point_sample_workspace_bytes(int N, int C, int H, int W, int P, int dtype) {
  if (N < 1 || C < 1 || H < 1 || W < 1 || P < 1) return 0;
  if (dtype != MSDA_F32 && dtype != MSDA_F64) return 0;
  if (static_cast<int64_t>(N) * P > INT32_MAX / 2) return 0;
  return sample_band_rows(H, W, 64 * 1024) > 0 ? sample_lists_bytes(N, P) : 0;
}

int msda_point_sample_backward(const void* grad_out, const void* input, const void* coords,
                               void* grad_input, void* grad_coords, void* workspace,
                               size_t workspace_bytes, int N, int C, int H, int W, int P,
                               int dtype, int mode, int align_corners, void* hip_stream) {
  if (!grad_out || !input || !coords || !grad_input || !grad_coords) return MSDA_ENULL;
  if (N < 1 || C < 1 || H < 1 || W < 1 || P < 1) return MSDA_ESHAPE;
  if (static_cast<int64_t>(H) * W > INT32_MAX) return MSDA_ESHAPE;
  if ((dtype != MSDA_F32 && dtype != MSDA_F64) ||
      (mode != MSDA_SAMPLE_BILINEAR && mode != MSDA_SAMPLE_NEAREST))
    return MSDA_EDTYPE;
  const size_t e = elem_size(dtype);
  if (!aligned(grad_out, e) || !aligned(input, e) || !aligned(coords, e) ||
      !aligned(grad_input, e) || !aligned(grad_coords, e))
    return MSDA_EALIGN;
  SampleGrid g;
  int64_t blocks;
  if (!make_sample_grid(N, C, P, &g, &blocks)) return MSDA_ESHAPE;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  hipError_t err;
  // grad_input: by bands of the plane held in LDS when the map allows it, else by global atomics
  // 64 KB bands: two workgroups per CU overlap their phases (measured equal to 128 KB bands,
  // 32 KB bands 35 % slower: every band walks all points of its sample) and the kernel stays
  // within the default dynamic-LDS limit
  constexpr size_t kBandLds = 64 * 1024;
  const int band_rows = sample_band_rows(H, W, kBandLds);
  SampleGrid bg{};
  int64_t band_blocks = 0;
  if (band_rows > 0) {
    bg.per = (H + band_rows - 1) / band_rows;
    bg.cslices = C;  // one unit per plane
    bg.cper = 1;
    bg.units = static_cast<int64_t>(N) * C;
    band_blocks = sample_grid_blocks(bg);
  }
  const bool banded = band_rows > 0 && band_blocks <= INT32_MAX;
  // point lists per band, when the caller brought the workspace for them
  const size_t lists_need = msda_point_sample_workspace_bytes(N, C, H, W, P, dtype);
  const bool listed = banded && lists_need > 0 && workspace != nullptr &&
                      workspace_bytes >= lists_need && aligned(workspace, 4);
  int* list = listed ? static_cast<int*>(workspace) : nullptr;
  int* list_off = listed ? list + static_cast<size_t>(N) * 2 * P : nullptr;
  if (!banded) {
    err = hipMemsetAsync(grad_input, 0, static_cast<size_t>(N) * C * H * W * e, stream);
    if (err != hipSuccess) return static_cast<int>(err);
  }
  if (g.cslices > 1) {
    err = hipMemsetAsync(grad_coords, 0, static_cast<size_t>(N) * P * 2 * e, stream);
    if (err != hipSuccess) return static_cast<int>(err);
  }
  Launches run;
#define MSDA_SB(T, SCATTER)                                                                     \
  run((point_sample_bwd_kernel<T, SCATTER>), dim3(static_cast<unsigned>(blocks)), \
                     dim3(256), 0, stream, static_cast<const T*>(grad_out),                     \
                     static_cast<const T*>(input), static_cast<const T*>(coords),               \
                     static_cast<T*>(grad_input), static_cast<T*>(grad_coords), C, H, W, P, g,  \
                     mode, align_corners ? 1 : 0)
#define MSDA_SBAND(T)                                                                           \
  do {                                                                                          \
    const size_t lds = static_cast<size_t>(band_rows) * W * sizeof(double);                     \
    if (listed)                                                                                 \
      run((point_band_lists_kernel<T>), dim3(static_cast<unsigned>(N)),          \
                         dim3(1024), 0, stream, static_cast<const T*>(coords), list, list_off,  \
                         H, P, band_rows, bg.per, mode, align_corners ? 1 : 0);                 \
    run((point_sample_bwd_band_kernel<T>),                                       \
                       dim3(static_cast<unsigned>(band_blocks)), dim3(1024), lds, stream,       \
                       static_cast<const T*>(grad_out), static_cast<const T*>(coords),          \
                       static_cast<T*>(grad_input), list, list_off, C, H, W, P, bg, band_rows,  \
                       mode, align_corners ? 1 : 0);                                            \
  } while (0)
  if (dtype == MSDA_F32) {
    if (banded) {
      MSDA_SB(float, false);
      MSDA_SBAND(float);
    } else {
      MSDA_SB(float, true);
    }
  } else {
    if (banded) {
      MSDA_SB(double, false);
      MSDA_SBAND(double);
    } else {
      MSDA_SB(double, true);
    }
  }
#undef MSDA_SBAND
#undef MSDA_SB
  return run.rc;
}

}  // extern "C"
