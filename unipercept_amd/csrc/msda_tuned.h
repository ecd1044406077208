// Tuned MSDeformAttn kernels for gfx950: head dims that split into 16-byte lane chunks.
//
// Work decomposition shared by the forward and the query side of the backward:
//   * a group of G lanes owns one (batch, query, head) triple; each lane holds CPL consecutive
//     channels = one 16-byte chunk (CPL = 4 for fp32, 8 for bf16/fp16), so D = G * CPL and one
//     bilinear corner of one head is a single contiguous D*e-byte segment (one 128-byte line for
//     fp32 D = 32) fetched by G lanes with one global_load_dwordx4 each;
//   * a 256-thread workgroup owns consecutive queries of ONE head of one batch element, and the
//     head is blockIdx.x % M: workgroups that share an XCD (equal blockIdx % 8) then gather from
//     the same head's slab, which keeps the sliding window of that slab inside the XCD's L2;
//   * phase 1: all 256 threads read the chunk's sampling locations and attention weights, turn
//     every point into {4 attention-scaled corner weights, 4 byte offsets} once, and stage that
//     in LDS; phase 2: each group walks its points, broadcasting the staged record to its G
//     lanes with ds_read_b128, and gathers the corners.
//
// Backward grad_value (the scatter) avoids global float atomics, which run at the memory side
// at ~1.3 TB/s of added bytes on MI355X.  The query-side kernel appends one 16-byte record per
// (point, destination tile) to a per-workgroup region, grouped by tile, and publishes a
// directory; a second kernel owns one destination tile (64 positions of one head) per workgroup,
// sorts the tile's records by position in LDS and accumulates every position in the registers
// of the 4 lanes that own it (no float atomics there either), then stores the tile once.
#pragma once


#include <initializer_list>
#include <type_traits>

#include "msda_common.h"

namespace msda {

// Diagnostic builds only (-DMSDA_ABLATE=<bits>): phases to leave out so that their cost can be
// read off a timing difference; the results are then wrong.  1 query kernel: count + emit,
// 2 query kernel: grad_loc / grad_attn stores, 4 query kernel: value gather, 8 tile kernel: bucket
// walk, 32 tile kernel: bucket sort.
#ifndef MSDA_ABLATE
#define MSDA_ABLATE 0
#endif
constexpr int kAblate = MSDA_ABLATE;

// Diagnostic builds only (-DMSDA_STAMPS=1): thread 0 of every workgroup of the two backward
// kernels adds up the shader cycles it spends in each phase (s_memtime stamps behind the
// workgroup barriers) and leaves the sums in a device buffer of the library, read back through
// msda_debug_read_stamps().  Never defined in the shipped library.
#ifndef MSDA_STAMPS
#define MSDA_STAMPS 0
#endif
#if MSDA_STAMPS
constexpr int kStampSlots = 16;
constexpr int kStampGroups = 1 << 15;
__device__ unsigned long long stamp_buf[2][kStampGroups][kStampSlots];
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define MSDA_STAMP_DECL unsigned long long st_prev = stamp_now(), st_acc[kStampSlots] = {}
#define MSDA_STAMP(i)                              \
  do {                                             \
    const unsigned long long st_t = stamp_now();   \
    st_acc[i] += st_t - st_prev;                   \
    st_prev = st_t;                                \
  } while (0)
#define MSDA_STAMP_FLUSH(k)                                                               \
  do {                                                                                    \
    const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;                              \
    if (threadIdx.x == 0 && wg < kStampGroups)                                            \
      for (int st_i = 0; st_i < kStampSlots; ++st_i) stamp_buf[k][wg][st_i] = st_acc[st_i]; \
  } while (0)
#else
#define MSDA_STAMP_DECL
#define MSDA_STAMP(i)
#define MSDA_STAMP_FLUSH(k)
#endif

// The query kernel leaves a per-head copy (N, M, Lq, D) of grad_out in the workspace and the
// tile kernel gathers its rows from that copy: in the (N, Lq, M, D) tensor a head's rows are
// M*D*e bytes apart and fall on 2 of the 16 channels of the XCD's L2.
#ifndef MSDA_G_COPY
#define MSDA_G_COPY 1
#endif

// Experiment builds only (-DMSDA_EXP_MIN=1): compile the tuned kernels for fp32 rows of 8 chunks
// (the headline shapes) alone, which cuts the build time of an A/B variant to a fraction; every
// other type or head dim then reports "not tuned".  Never defined in the shipped library.
#ifndef MSDA_EXP_MIN
#define MSDA_EXP_MIN 0
#endif

constexpr int kNotTuned = INT32_MIN;
constexpr int MSDA_TUNED_ELAYOUT = -8;  // = MSDA_ELAYOUT of the public header
constexpr int kMaxLevelsTuned = 16;

template <typename VT>
struct Chunk;  // 16 bytes of VT as CPL floats

template <>
struct Chunk<float> {
  static constexpr int kCPL = 4;
  __device__ __forceinline__ static void load(const void* p, float (&v)[4]) {
    const float4 t = *static_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ static void unpack(const uint4& t, float (&v)[4]) {
    v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y);
    v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
  }
  // acc + <a, b> over the chunk's elements
  __device__ __forceinline__ static float dot(const uint4& a, const uint4& b, float acc) {
    acc = fmaf(__uint_as_float(a.x), __uint_as_float(b.x), acc);
    acc = fmaf(__uint_as_float(a.y), __uint_as_float(b.y), acc);
    acc = fmaf(__uint_as_float(a.z), __uint_as_float(b.z), acc);
    return fmaf(__uint_as_float(a.w), __uint_as_float(b.w), acc);
  }
  __device__ __forceinline__ static void store(void* p, const float (&v)[4]) {
    *static_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

template <>
struct Chunk<bf16_t> {
  static constexpr int kCPL = 8;
  __device__ __forceinline__ static void unpack(const uint4& t, float (&v)[8]) {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ static void load(const void* p, float (&v)[8]) {
    unpack(*static_cast<const uint4*>(p), v);
  }
  // acc + <a, b>: both operands stay packed, one v_dot2_f32_bf16 (two exact products, fp32
  // accumulation) per pair instead of two unpacks and an FMA per element
  __device__ __forceinline__ static float dot(const uint4& a, const uint4& b, float acc) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.x), __builtin_bit_cast(bf16x2, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.y), __builtin_bit_cast(bf16x2, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.z), __builtin_bit_cast(bf16x2, b.z), acc, false);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.w), __builtin_bit_cast(bf16x2, b.w), acc, false);
  }
  __device__ __forceinline__ static void store(void* p, const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = f32_to_bf16_bits(v[2 * i]) | (f32_to_bf16_bits(v[2 * i + 1]) << 16);
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
  __device__ __forceinline__ static void unpack(const uint4& r, float (&v)[8]) {
    const half8 t = __builtin_bit_cast(half8, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = static_cast<float>(t[i]);
  }
  __device__ __forceinline__ static float dot(const uint4& a, const uint4& b, float acc) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, a.x), __builtin_bit_cast(half2v, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, a.y), __builtin_bit_cast(half2v, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, a.z), __builtin_bit_cast(half2v, b.z), acc, false);
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, a.w), __builtin_bit_cast(half2v, b.w), acc, false);
  }
  __device__ __forceinline__ static void store(void* p, const float (&v)[8]) {
    half8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = static_cast<_Float16>(v[i]);
    *static_cast<half8*>(p) = t;
  }
};

struct LevelInfo {
  int H, W, start;
};


// A level whose extent does not fit inside [0, S) gets H = W = 0: no address derived from it can
// leave the value tensor, and its points contribute nothing.
__device__ __forceinline__ LevelInfo load_level(const int64_t* __restrict__ shapes,
                                                const int64_t* __restrict__ lsi, int l, int S) {
  const int64_t h = shapes[2 * l], w = shapes[2 * l + 1], st = lsi[l];
  const bool ok = h > 0 && w > 0 && st >= 0 && h <= S && w <= S && h * w <= S && st + h * w <= S;
  LevelInfo r;
  r.H = ok ? static_cast<int>(h) : 0;
  r.W = ok ? static_cast<int>(w) : 0;
  r.start = ok ? static_cast<int>(st) : 0;
  return r;
}

// Query order.  When the queries are the positions of the pyramid itself (the encoder: the levels
// tile [0, Lq) in order), a workgroup does not take consecutive queries — a piece of one image
// row, whose sampling points spread over a wide, flat strip of every level — but consecutive
// queries of a patch-by-patch walk over the levels: patches of kOrderTH x kOrderTW pixels in
// raster order, row-major inside (patches at the bottom and right edges are cut).  The points of
// a workgroup then fall into a compact region: the row gathers of the workgroups resident on a
// CU overlap more, and the scatter records of a query block land in fewer destination tiles, as
// longer runs.  Which workgroup takes which query changes nothing in the results.  Measured on
// MI355X (Cityscapes pyramid, fp32): forward 0.80 -> 0.78 ms, backward 2.05 -> 1.97 ms; patches
// of 8 x 8 to 64 x 8, 16 x 16 and column strips all within 1 % of one another.
#ifndef MSDA_ORDER_TH
#define MSDA_ORDER_TH 64
#endif
#ifndef MSDA_ORDER_TW
#define MSDA_ORDER_TW 8
#endif
constexpr int kOrderTH = MSDA_ORDER_TH, kOrderTW = MSDA_ORDER_TW;

// Query index of work item `w` in [0, Lq): the patch-order walk above when `enabled` and the
// levels tile [0, Lq) exactly, else w itself.  A bijection of [0, Lq) either way.  The level
// table is read from global memory here (uniform addresses: scalar loads), not from the
// workgroup's LDS copy, so that the callers need no barrier before it.
__device__ __forceinline__ int order_query(int w, const int64_t* __restrict__ shapes,
                                           const int64_t* __restrict__ lsi, int S, int L, int Lq,
                                           bool enabled) {
  if (!enabled) return w;
  int acc = 0, H = 0, W = 0, start = 0;
  bool tiled = true, found = false;
  for (int l = 0; l < L; ++l) {
    const LevelInfo v = load_level(shapes, lsi, l, S);
    const int n = v.H * v.W;
    tiled = tiled && n > 0 && v.start == acc;
    if (w >= acc && w < acc + n) {
      found = true;
      H = v.H;
      W = v.W;
      start = acc;
    }
    acc += n;
  }
  if (!tiled || acc != Lq || !found) return w;
  const uint32_t u = static_cast<uint32_t>(w - start);
  const uint32_t per_row = static_cast<uint32_t>(kOrderTH) * W;  // queries of a full row of patches
  const uint32_t pr = u / per_row;
  const uint32_t rem = u - pr * per_row;
  const uint32_t hr = min(static_cast<uint32_t>(kOrderTH), static_cast<uint32_t>(H) - pr * kOrderTH);
  const uint32_t per_patch = hr * kOrderTW;  // (the last patch of the row may be narrower)
  const uint32_t pc = rem / per_patch;
  const uint32_t rem2 = rem - pc * per_patch;
  const uint32_t wc = min(static_cast<uint32_t>(kOrderTW), static_cast<uint32_t>(W) - pc * kOrderTW);
  const uint32_t y = rem2 / wc;
  const uint32_t x = rem2 - y * wc;
  return start + static_cast<int>((pr * kOrderTH + y) * W + pc * kOrderTW + x);
}

// One staged sampling point occupies 2 or 3 16-byte LDS slots (see StageKind):
//   weights:   the four corner weights, multiplied by the attention weight, 0 for a corner
//              outside the level (not staged by the binned backward, which never uses them);
//   offsets:   the four corner byte offsets relative to the batch element's value slab (head
//              offset included, lane offset not); they are multiples of 16, and bit 0 is set
//              when the corner lies inside the level;
//   fractions: (backward only) {lx, ly, attn * W, attn * H}, lx/ly being the fractional parts.
// The LDS stride (in slots) between the point lists of consecutive groups is odd, so the groups
// of a wave read from disjoint banks.
__host__ __device__ constexpr int group_stride_slots(int LP, int slots) { return (slots * LP) | 1; }

// Lanes that share one query in the backward query kernel.  Rows of 8+ chunks are split two
// chunks per lane (lane, lane + lanes): the per-point work that does not depend on the channel
// count (LDS reads, address arithmetic, bilinear algebra, cross-lane sums) is then paid by half
// as many lanes, and 4 lanes x 16 bytes still make whole 64-byte L1 accesses.
#ifndef MSDA_K3_WIDE
#define MSDA_K3_WIDE 1
#endif
__host__ __device__ constexpr int query_lanes(int G) { return (MSDA_K3_WIDE && G >= 8) ? G / 2 : G; }

// Query kernel of the binned backward: locations and weights read once, up front, into registers
// (0: the count pass and every staging pass read them from memory).
#ifndef MSDA_K3_PRE
#define MSDA_K3_PRE 1
#endif
// Gather loop of the binned query kernel: 2 = two points (16 row loads) in flight per lane,
// 1 = one point at a time.
#ifndef MSDA_K3_DEPTH
#define MSDA_K3_DEPTH 2
#endif
// records of a workgroup leave through an LDS image of its region (1) or one store each (0)
#ifndef MSDA_K3_LDS_EMIT
#define MSDA_K3_LDS_EMIT 1
#endif
// diagnostic builds: unused LDS in the query kernel, to lower the number of resident workgroups
#ifndef MSDA_K3_LDS_PAD
#define MSDA_K3_LDS_PAD 0
#endif
// minimum waves per SIMD the query kernel's register allocation must allow
#ifndef MSDA_FWD_PPT2
#define MSDA_FWD_PPT2 1  // forward staging: two neighbouring points per thread
#endif
#ifndef MSDA_K3_WSTORE
#define MSDA_K3_WSTORE 1  // gradients of a lane's neighbouring points stored side by side
#endif
#ifndef MSDA_K3_PSEQ
#define MSDA_K3_PSEQ 1
#endif
#ifndef MSDA_K3_PRIO
// wave priority by phase.  1: the prologue (point loads, count, emit, park, stage) runs at high
// priority, the gather at normal priority, so a workgroup gets through its bookkeeping while
// its neighbours on the CU gather — fp32 rows only (headline backward 1.965 -> 1.954 ms, DVPS
// 4.53 -> 4.46, uniform locations 2.27 -> 2.24; with 16-bit rows it measured 0.3-1.2 % slower);
// 2: the gather high (no change); the same in the forward's staging phase measured slower, in
// the tile kernel neutral
#define MSDA_K3_PRIO 1
#endif
#ifndef MSDA_K3_MINWAVES
#define MSDA_K3_MINWAVES 1
#endif

// One scatter record (16 bytes): what the tile kernel needs to redo the footprint of a point.
//   x = query index | level << 28, y = attention weight, z/w = normalised location (x, y) as f32
constexpr uint32_t kRecInvalid = 0xffffffffu;

constexpr int kTileShift = 6;  // positions per 1-D destination tile: 64, one per 4-lane group

// 2-D tiles are square patches of 8 x 8 pixels, or 7 x 7 where the tile kernel works by footprint
// origin (bwd_tile_origin_kernel).  v / tw for those two widths without the integer-division
// sequence; 0 <= v < 13000 for tw = 7 (level extents are bounded on the host), and -1 -> -1.
__host__ __device__ __forceinline__ int tile_div(int v, int tw) {
  return tw == 8 ? v >> 3 : (v * 9363) >> 16;
}
constexpr int kMaxExtentTile7 = 8192;

// Destination tiles of the binned grad_value scatter.  tile2d = 0: a tile is TS consecutive
// positions of the flattened (level-concatenated) sequence.  tile2d = 1: a tile is an 8 x 8 pixel
// patch of one level — a bilinear footprint then straddles two tiles only near patch edges
// (1.27 records per point instead of about 2 on the Cityscapes pyramid); the tile count needs
// the level shapes on the host.
struct BinPlan {
  int ts_shift;     // log2 of the tile size in positions
  int n_tiles;      // tiles per (batch, head) slab
  int qb;           // queries per query-side workgroup
  int n_qblocks;    // ceil(Lq / qb)
  uint32_t region;  // records per workgroup region
  int tile2d;       // tile geometry (see above)
  int tw;           // tile2d: patches are tw x tw pixels (8, or 7 with the origin tile kernel)
  int in_order;     // always 1; a run-time value on purpose, see the value gather of bwd_query_kernel
  int phase;        // query kernel: bit 0 = bin the scatter (directory + records), bit 1 = gradients
  int q_per_block;  // (real) queries whose records one region holds: qb / split
  int split;        // virtual queries per query in the query kernel (rows of 17-32 points: 2)
  int patch_order;  // query kernel, register-resident form: query blocks in patch order (order_query)
};
constexpr int kPhaseBin = 1, kPhaseGrad = 2;
// experiment builds (MSDA_EXP_PHASES): launch only the query kernel / only the tile kernel
constexpr int kPhaseNoTile = 4, kPhaseNoQuery = 8;
#ifndef MSDA_EXP_PHASES
#define MSDA_EXP_PHASES 0
#endif

struct TileMap {  // tile2d only: first tile id and tiles per row of every level
  int tstart[kMaxLevelsTuned + 1];
  int tx[kMaxLevelsTuned];
};

__device__ __forceinline__ void build_tile_map(const LevelInfo* lv, int L, const BinPlan& plan,
                                               TileMap& tm) {
  int run = 0;
  for (int l = 0; l < L; ++l) {
    const int tx = tile_div(lv[l].W - 1, plan.tw) + 1;
    const int ty = tile_div(lv[l].H - 1, plan.tw) + 1;
    tm.tstart[l] = run;
    tm.tx[l] = tx;
    run += tx * ty;
  }
  tm.tstart[L] = run;
}
// (the same from the level table in global memory: no LDS copy, hence no barrier, needed first)
__device__ __forceinline__ void build_tile_map(const int64_t* __restrict__ shapes,
                                               const int64_t* __restrict__ lsi, int S, int L,
                                               const BinPlan& plan, TileMap& tm) {
  int run = 0;
  for (int l = 0; l < L; ++l) {
    const LevelInfo v = load_level(shapes, lsi, l, S);
    const int tx = tile_div(v.W - 1, plan.tw) + 1;
    const int ty = tile_div(v.H - 1, plan.tw) + 1;
    tm.tstart[l] = run;
    tm.tx[l] = tx;
    run += tx * ty;
  }
  tm.tstart[L] = run;
}

// tile2d: level and top-left pixel of a tile (level -1 when the id is past the last tile)
__device__ __forceinline__ void locate_tile(const TileMap& tm, int L, const BinPlan& plan, int tile,
                                            int& level, int& y0, int& x0) {
  int l = -1;
  for (int k = 0; k < L; ++k)
    if (tile >= tm.tstart[k] && tile < tm.tstart[k + 1]) l = k;
  level = l;
  y0 = x0 = 0;
  if (l >= 0) {
    const int r = tile - tm.tstart[l];
    y0 = (r / tm.tx[l]) * plan.tw;
    x0 = (r % tm.tx[l]) * plan.tw;
  }
}

// tile id and position inside the tile of the pixel (cy, cx) of level l
__device__ __forceinline__ void pixel_tile(const BinPlan& plan, const TileMap& tm,
                                           const LevelInfo& L_, int l, int cy, int cx,
                                           uint32_t& tile, uint32_t& local) {
  if (plan.tile2d) {
    const int tw = plan.tw, tyi = tile_div(cy, tw), txi = tile_div(cx, tw);
    tile = static_cast<uint32_t>(tm.tstart[l] + __umul24(tyi, tm.tx[l]) + txi);
    local = static_cast<uint32_t>(__mul24(cy - __mul24(tyi, tw), tw) + (cx - __mul24(txi, tw)));
  } else {
    const uint32_t pos = static_cast<uint32_t>(L_.start + __umul24(cy, L_.W) + cx);
    tile = pos >> plan.ts_shift;
    local = pos & ((1u << plan.ts_shift) - 1u);
  }
}

// Destination tiles of the four corners of a footprint: t[k] is the tile of corner k, keep[k]
// says whether a record goes to it (the corner counts and no earlier corner has the same tile).
// Tiles at or beyond plan.n_tiles (host and device level tables disagree) are dropped.
__device__ __forceinline__ void footprint_tiles(const Footprint<float>& f, const BinPlan& plan,
                                                const TileMap& tm, const LevelInfo& L_, int l,
                                                uint32_t (&t)[4], bool (&keep)[4]) {
  const uint32_t n_tiles = static_cast<uint32_t>(plan.n_tiles);
  if (plan.tile2d) {
    // a patch is a product of a row range and a column range, so rows and columns dedupe
    // separately: the second row (column) adds tiles only if it is inside the level and falls in
    // another patch row (column) than a first one that counts
    const int tw = plan.tw;
    const int ty0 = tile_div(f.cy[0], tw), ty1 = tile_div(f.cy[1], tw);
    const int tx0 = tile_div(f.cx[0], tw), tx1 = tile_div(f.cx[1], tw);
    const int txl = tm.tx[l];
    const uint32_t base0 = static_cast<uint32_t>(tm.tstart[l] + __umul24(ty0, txl));
    const uint32_t base1 = static_cast<uint32_t>(tm.tstart[l] + __umul24(ty1, txl));
    const bool r0 = f.vy[0], r1 = f.vy[1] && !(f.vy[0] && ty1 == ty0);
    const bool c0 = f.vx[0], c1 = f.vx[1] && !(f.vx[0] && tx1 == tx0);
    t[0] = base0 + tx0;
    t[1] = base0 + tx1;
    t[2] = base1 + tx0;
    t[3] = base1 + tx1;
    keep[0] = r0 && c0 && t[0] < n_tiles;
    keep[1] = r0 && c1 && t[1] < n_tiles;
    keep[2] = r1 && c0 && t[2] < n_tiles;
    keep[3] = r1 && c1 && t[3] < n_tiles;
  } else {
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      t[k] = static_cast<uint32_t>(L_.start + f.pix[k]) >> plan.ts_shift;
      ok[k] = f.valid[k] && t[k] < n_tiles;
    }
    keep[0] = ok[0];
    keep[1] = ok[1] && !(ok[0] && t[0] == t[1]);
    keep[2] = ok[2] && !(ok[0] && t[0] == t[2]) && !(ok[1] && t[1] == t[2]);
    keep[3] = ok[3] && !(ok[0] && t[0] == t[3]) && !(ok[1] && t[1] == t[3]) &&
              !(ok[2] && t[2] == t[3]);
  }
}

// t / d and t % d for small non-negative t (below ~2^20) and d >= 1 without the integer-division
// instruction sequence: a float estimate, corrected by at most one step.  `inv` = 1.0f / d.
__device__ __forceinline__ void small_divmod(int t, int d, float inv, int& q, int& r) {
  q = static_cast<int>((static_cast<float>(t) + 0.5f) * inv);
  r = t - __mul24(q, d);  // (both far below 2^23)
  if (r < 0) {
    --q;
    r += d;
  } else if (r >= d) {
    ++q;
    r -= d;
  }
}

// Elements between the (batch, query, head) rows of the location and weight arrays: 2*L*P and
// L*P for separate tensors, 3*L*P for both when they are the two parts of one packed tensor
// (row = L*P (x, y) pairs followed by L*P weights).  The location stride is even (the packed
// form is only taken for even L*P), so that an (x, y) pair moves with one 8-byte (4-byte for
// 16-bit types) access.
struct PointRows {
  int loc, attn;
};

// sampling location (x, y) and attention weight of point i of a row of the workgroup:
// `loc_wg` / `attn_wg` point at the first row the
// workgroup touches (a uniform 64-bit address), `row` counts rows from there, and the element
// offset comes from 24-bit multiplies (rows per workgroup and row lengths are small).
template <typename LT>
__device__ __forceinline__ void load_point_near(const LT* __restrict__ loc_wg,
                                                const LT* __restrict__ attn_wg, uint32_t row, int i,
                                                PointRows rs, float& x, float& y, float& a) {
  const LT* xy_p = loc_wg + (__umul24(row, static_cast<uint32_t>(rs.loc)) + 2u * i);
  if constexpr (sizeof(LT) == 4) {
    const float2 xy = *reinterpret_cast<const float2*>(xy_p);
    x = xy.x;
    y = xy.y;
  } else {
    LT pair[2];  // one 4-byte load for the two 16-bit coordinates
    __builtin_memcpy(pair, xy_p, 4);
    x = Elem<LT>::load(&pair[0]);
    y = Elem<LT>::load(&pair[1]);
  }
  a = Elem<LT>::load(attn_wg + (__umul24(row, static_cast<uint32_t>(rs.attn)) + i));
}

struct EmitCtx {
  uint4* records;      // this workgroup's record region
  uint32_t* tile_cur;  // LDS: per tile, where its next record goes inside the region
  BinPlan plan;        // by value: a pointer to the kernel argument would force it into scratch
  const TileMap* tm;
  uint4* lds_image;    // LDS image of the head of the region (null: records go out directly)
  uint32_t lds_cap;    // records the image holds
};

// What a staged point holds: forward {weights, offsets}; backward with atomic scatter {weights,
// offsets, fractions}; backward with binned scatter {offsets, fractions}.
enum StageKind { kStageFwd = 0, kStageBwdAtomic = 1, kStageBwdBinned = 2 };
// staged offset of a corner outside its level in the binned backward: beyond any value slab
// (slabs are at most 4 GB - 512 bytes there), and lane offsets (< 256) do not wrap it around
constexpr uint32_t kOffsetOutside = 0xFFFFFF00u;
__host__ __device__ constexpr int stage_slots(int kind) { return kind == kStageBwdAtomic ? 3 : 2; }

// Raw sampling points: locations and weights still as the layer's projection left them, i.e.
// per (batch, query, head) row L*P pixel offsets (dx, dy) followed by L*P attention logits, next
// to the query's reference point on every level (`ref`, (N, Lq, L, 2) fp32):
//     (x, y) = ref + (dx / W_l, dy / H_l),   weight = softmax over the row's L*P logits.
// The kernels do this arithmetic as they read the points, so the locations and weights never
// exist as a tensor.  softmax_stats: (max, 1 / sum of exp(logit - max)) of one row.
template <typename LT>
__device__ __forceinline__ float2 softmax_stats(const LT* __restrict__ logits, int LP) {
  float mx = Elem<LT>::load(logits);
  for (int i = 1; i < LP; ++i) mx = fmaxf(mx, Elem<LT>::load(logits + i));
  float sum = 0.f;
  for (int i = 0; i < LP; ++i) sum += expf(Elem<LT>::load(logits + i) - mx);
  return make_float2(mx, 1.0f / sum);
}

// (dx, dy, logit) in x / y / a -> (x, y, weight)
__device__ __forceinline__ void raw_to_point(const float* __restrict__ ref_ql, const LevelInfo& L_,
                                             float2 stats, float& x, float& y, float& a) {
  const float2 rp = *reinterpret_cast<const float2*>(ref_ql);
  x = fmaf(x, 1.0f / static_cast<float>(L_.W), rp.x);
  y = fmaf(y, 1.0f / static_cast<float>(L_.H), rp.y);
  a = expf(a - stats.x) * stats.y;
}

// Appends the scatter records of one sampling point (footprint f on level l, query index q) to
// the workgroup's record region: one record per destination tile the footprint touches.
__device__ __forceinline__ void emit_records(const Footprint<float>& f, const LevelInfo& L_, int l,
                                             float x, float y, float a, int q,
                                             const EmitCtx& emit) {
  if (f.any && a != 0.f && emit.records != nullptr) {
    uint32_t tiles[4];
    bool keep[4];
    footprint_tiles(f, emit.plan, *emit.tm, L_, l, tiles, keep);
    uint4 rec;
    rec.x = static_cast<uint32_t>(q) | (static_cast<uint32_t>(l) << 28);
    rec.y = __float_as_uint(a);
    rec.z = __float_as_uint(x);
    rec.w = __float_as_uint(y);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (keep[j]) {
        const uint32_t at = atomicAdd(&emit.tile_cur[tiles[j]], 1u);
        if (at < emit.lds_cap) {
          emit.lds_image[at] = rec;  // leaves the workgroup later, as part of one linear copy
        } else if (at < emit.plan.region) {  // always true: a bug stays in the region
          emit.records[at] = rec;
        }
      }
    }
  }
}

// Stages one sampling point (location (x, y), weight a, level l) at `dst` and, when EMIT, appends
// its scatter records (query index q) to the workgroup's record region.
template <int KIND, bool EMIT>
__device__ __forceinline__ void stage_point(const LevelInfo& L_, int l, float x, float y, float a,
                                            int q, uint4* dst, uint32_t rowb, uint32_t headb,
                                            const EmitCtx& emit, bool raw = false) {
  constexpr int SLOTS = stage_slots(KIND);
  const Footprint<float> f = footprint<float, true>(x, y, L_.H, L_.W);
  uint4 wrec, orec;
  wrec.x = __float_as_uint(a * f.w[0]);
  wrec.y = __float_as_uint(a * f.w[1]);
  wrec.z = __float_as_uint(a * f.w[2]);
  wrec.w = __float_as_uint(a * f.w[3]);
  orec.x = (__umul24(static_cast<uint32_t>(L_.start + f.pix[0]), rowb) + headb) | (f.valid[0] ? 1u : 0u);
  orec.y = (__umul24(static_cast<uint32_t>(L_.start + f.pix[1]), rowb) + headb) | (f.valid[1] ? 1u : 0u);
  orec.z = (__umul24(static_cast<uint32_t>(L_.start + f.pix[2]), rowb) + headb) | (f.valid[2] ? 1u : 0u);
  orec.w = (__umul24(static_cast<uint32_t>(L_.start + f.pix[3]), rowb) + headb) | (f.valid[3] ? 1u : 0u);
  if constexpr (KIND == kStageBwdBinned && MSDA_K3_DEPTH == 2) {
    // the binned query kernel gathers through a buffer descriptor of the value slab: a corner
    // outside the level gets an offset past the slab, which reads as zeros — no validity test
    // and no masking in the gather loop (offsets of corners inside stay plain multiples of 16)
    orec.x = f.valid[0] ? orec.x & ~15u : kOffsetOutside;
    orec.y = f.valid[1] ? orec.y & ~15u : kOffsetOutside;
    orec.z = f.valid[2] ? orec.z & ~15u : kOffsetOutside;
    orec.w = f.valid[3] ? orec.w & ~15u : kOffsetOutside;
  }
  if constexpr (KIND != kStageBwdBinned) dst[0] = wrec;
  dst[KIND == kStageBwdBinned ? 0 : 1] = orec;
  if constexpr (KIND != kStageFwd) {
    uint4 frec;
    frec.x = __float_as_uint(f.lx);
    frec.y = __float_as_uint(f.ly);
    // factor of the location gradient: d/d(x, y) carries (W, H); d/d(dx, dy) of raw points does
    // not (the offset is divided by them), and the weight itself is needed for the logit gradient
    frec.z = __float_as_uint(raw ? a : a * static_cast<float>(L_.W));
    frec.w = __float_as_uint(raw ? a : a * static_cast<float>(L_.H));
    dst[SLOTS - 1] = frec;
  }
  if constexpr (EMIT) emit_records(f, L_, l, x, y, a, q, emit);
}

// Stages the points of `nq` consecutive queries of head m (first query q_first) into LDS.
template <typename LT, int KIND, bool EMIT>
__device__ __forceinline__ void stage_chunk(const LevelInfo* lv, const LT* __restrict__ loc,
                                            const LT* __restrict__ attn, uint4* __restrict__ stage, int b,
                                            int q_first, int nq, int m, int M, int Lq, int LP,
                                            int P, int i_first, int nI, int LPB, uint32_t rowb,
                                            uint32_t headb, PointRows rs, const EmitCtx& emit,
                                            const float* __restrict__ ref = nullptr,
                                            const float2* __restrict__ row_stats = nullptr, int L = 0,
                                            const int* __restrict__ qidx = nullptr) {
  // points i_first .. i_first + nI - 1 of every query; LDS lists are laid out for LPB points
  // qidx (LDS, optional): the query index of each of the nq queries, when they are not
  // q_first, q_first + 1, ... (then Lq * M < 2^24, so rows count from the batch element's first)
  constexpr int SLOTS = stage_slots(KIND);
  const int npts = nq * nI;
  const int stride = group_stride_slots(LPB, SLOTS);
  const float inv_nI = 1.0f / static_cast<float>(nI), inv_P = 1.0f / static_cast<float>(P);
  const int q_base = qidx != nullptr ? 0 : q_first;
  const int64_t row0 = (static_cast<int64_t>(b) * Lq + q_base) * M + m;  // row the offsets count from
  const LT* loc_wg = loc + row0 * rs.loc;
  const LT* attn_wg = attn + row0 * rs.attn;
  const float* ref_wg = ref == nullptr ? nullptr : ref + (static_cast<int64_t>(b) * Lq + q_base) * L * 2;
  if constexpr (KIND == kStageFwd && !EMIT && MSDA_FWD_PPT2) {
    if ((nI & 1) == 0 && (i_first & 1) == 0) {  // (uniform)
      // two neighbouring points per thread: one 16-byte and one 8-byte load (8 and 4 bytes for
      // 16-bit types) instead of two of each — the vector-memory path is paid per instruction
      const int half = nI >> 1;
      const float inv_half = 1.0f / static_cast<float>(half);
      for (int t = threadIdx.x; t < nq * half; t += 256) {
        int g, jj;
        small_divmod(t, half, inv_half, g, jj);
        const int ii = 2 * jj, i = i_first + ii;
        const uint32_t qo = qidx != nullptr ? static_cast<uint32_t>(qidx[g]) : static_cast<uint32_t>(g);
        const uint32_t row = __umul24(qo, M);
        const LT* xy_p = loc_wg + (__umul24(row, static_cast<uint32_t>(rs.loc)) + 2u * i);
        const LT* a_p = attn_wg + (__umul24(row, static_cast<uint32_t>(rs.attn)) + i);
        LT xy[4], aw[2];
        __builtin_memcpy(xy, xy_p, sizeof(xy));
        __builtin_memcpy(aw, a_p, sizeof(aw));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          int l, p_unused;
          small_divmod(i + k, P, inv_P, l, p_unused);
          float x = Elem<LT>::load(&xy[2 * k]), y = Elem<LT>::load(&xy[2 * k + 1]);
          float a = Elem<LT>::load(&aw[k]);
          if (ref != nullptr)
            raw_to_point(ref_wg + (__umul24(qo, L) + l) * 2u, lv[l], row_stats[g], x, y, a);
          stage_point<KIND, EMIT>(lv[l], l, x, y, a, q_base + static_cast<int>(qo),
                                  stage + (__umul24(g, stride) + SLOTS * (ii + k)), rowb, headb, emit,
                                  ref != nullptr);
        }
      }
      return;
    }
  }
  for (int t = threadIdx.x; t < npts; t += 256) {
    int g, ii, l, p_unused;
    small_divmod(t, nI, inv_nI, g, ii);
    const int i = i_first + ii;
    small_divmod(i, P, inv_P, l, p_unused);
    const uint32_t qo = qidx != nullptr ? static_cast<uint32_t>(qidx[g]) : static_cast<uint32_t>(g);
    float x, y, a;
    load_point_near(loc_wg, attn_wg, __umul24(qo, M), i, rs, x, y, a);
    if (ref != nullptr)
      raw_to_point(ref_wg + (__umul24(qo, L) + l) * 2u, lv[l], row_stats[g], x, y, a);
    stage_point<KIND, EMIT>(lv[l], l, x, y, a, q_base + static_cast<int>(qo),
                            stage + (__umul24(g, stride) + SLOTS * ii), rowb, headb, emit, ref != nullptr);
  }
}

// Sums and maxima over the lanes of a group (4, 8 or 16 lanes of a DPP row)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                                           0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int G>
__device__ __forceinline__ float group_sum(float v) {
  // butterfly inside groups of G <= 16 lanes on the DPP path (no LDS traffic)
  v = dpp_add<0xB1>(v);                         // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);                         // quad_perm [2,3,0,1]
  if constexpr (G >= 8) v = dpp_add<0x141>(v);   // row_half_mirror
  if constexpr (G >= 16) v = dpp_add<0x140>(v);  // row_mirror
  return v;
}

// the same butterfly for the maximum (a lane's own value stands in for lanes DPP leaves out)
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
  const int iv = __builtin_bit_cast(int, v);
  return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, CTRL, 0xF, 0xF, false)));
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
  v = dpp_max<0xB1>(v);
  v = dpp_max<0x4E>(v);
  if constexpr (G >= 8) v = dpp_max<0x141>(v);
  if constexpr (G >= 16) v = dpp_max<0x140>(v);
  return v;
}

// softmax_stats by the G lanes of a group together (lane = index inside the group): every lane
// takes every G-th logit; all lanes return the row's (max, 1 / sum of exp(logit - max)).
template <typename LT, int G>
__device__ __forceinline__ float2 softmax_stats_group(const LT* __restrict__ logits, int LP, int lane) {
  float mx = -INFINITY;
  for (int i = lane; i < LP; i += G) mx = fmaxf(mx, Elem<LT>::load(logits + i));
  mx = group_max<G>(mx);
  float sum = 0.f;
  for (int i = lane; i < LP; i += G) sum += expf(Elem<LT>::load(logits + i) - mx);
  sum = group_sum<G>(sum);
  return make_float2(mx, 1.0f / sum);
}

// ---------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------
// Rows of 64 bytes (G = 4 chunks: fp32 D = 16, 16-bit D = 32): the two x-neighbouring corners of
// a footprint are then 128 contiguous bytes of the per-head copy.  Eight lanes share a query, the
// upper four take the x0 + 1 column: one load instruction fetches both corners of a footprint
// row as one 128-byte piece instead of two 64-byte halves of (mostly) the same line from two
// instructions, and the two columns are summed apart until the end.
#ifndef MSDA_FWD_PAIR
#define MSDA_FWD_PAIR 1
#endif
// Measured on bf16 D = 32: 0.62 -> 0.58 ms for 12 points per query, 0.85 -> 0.78 ms for 16, but
// 1.49 -> 1.56 ms for 32 points (4 levels x 8 points), so the pairs are used up to 16 points.
inline bool fwd_pairs(int G, int LP) { return MSDA_FWD_PAIR && G == 4 && LP <= 16; }

// points per register set of the gather loop (x 4 row loads, x 2 in the pair form)
#ifndef MSDA_FWD_U
#define MSDA_FWD_U 2
#endif
#ifndef MSDA_FWD_U_PAIR
#define MSDA_FWD_U_PAIR 2
#endif
template <typename VT, typename LT, int G, bool PAIR = false>
__global__ __launch_bounds__(256) void fwd_tuned_kernel(
    const VT* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const LT* __restrict__ loc, const LT* __restrict__ attn,
    VT* __restrict__ out, int S, int M, int Lq, int L, int P, int LPB, int head_major,
    PointRows rs, const float* __restrict__ ref, int patch_order) {
  constexpr int CPL = Chunk<VT>::kCPL;
  constexpr int D = G * CPL;
  // one chunk per lane here: two (as in the backward query kernel) measured 20 % slower, the
  // forward has nothing but the gather to amortise and wants the extra loads in flight.
  // 64-byte rows (G = 4) may be taken by 8 lanes per query (PAIR), see fwd_pairs().
  static_assert(!PAIR || G == 4, "pairs are for 64-byte rows");
  constexpr int GQ = PAIR ? 2 * G : G;  // lanes per query
  constexpr int NC = 1;          // 16-byte chunks of a row per lane
  constexpr int GPB = 256 / GQ;  // queries per block
  extern __shared__ uint4 lds[];
  __shared__ LevelInfo lv[kMaxLevelsTuned];

  const int LP = L * P;
  const int b = blockIdx.y;
  const int m = blockIdx.x % M;
  const int q0 = (blockIdx.x / M) * GPB;
  const int nq = min(GPB, Lq - q0);
  __shared__ float2 row_stats[GPB];  // raw points: softmax statistics of the block's rows
  __shared__ int qidx[GPB];          // the block's queries (see order_query)
  if (threadIdx.x < L) lv[threadIdx.x] = load_level(shapes, lsi, threadIdx.x, S);
  if (static_cast<int>(threadIdx.x) < nq)
    qidx[threadIdx.x] = order_query(q0 + threadIdx.x, shapes, lsi, S, L, Lq, patch_order != 0);
  __syncthreads();
  if (ref != nullptr) {  // (uniform) raw points: the statistics of a row by its lane group
    const int gs = min(static_cast<int>(threadIdx.x) / GQ, nq - 1);  // (DPP rows stay full)
    const float2 st = softmax_stats_group<LT, GQ>(
        attn + ((static_cast<int64_t>(b) * Lq + qidx[gs]) * M + m) * rs.attn, LP, threadIdx.x % GQ);
    if (threadIdx.x % GQ == 0 && static_cast<int>(threadIdx.x) / GQ < nq) row_stats[gs] = st;
    __syncthreads();
  }

  // value is (N,S,M,D), or (N,M,S,D) when head_major: bytes between positions / to head m
  const uint32_t rowb = static_cast<uint32_t>(head_major ? 1 : M) * D * sizeof(VT);
  const uint32_t headb = static_cast<uint32_t>(m) * (head_major ? S : 1) * D * sizeof(VT);
  const int g = threadIdx.x / GQ, lane = threadIdx.x % G;
  const bool right = PAIR && (threadIdx.x % GQ) >= G;  // this lane takes the x0 + 1 corners
  const bool live = g < nq;
  const char* vb = reinterpret_cast<const char*>(value) +
                   static_cast<int64_t>(b) * S * M * D * static_cast<int64_t>(sizeof(VT));
  const uint32_t lane_b = lane * 16;
  const uint4* rec = lds + __umul24(g, group_stride_slots(LPB, 2));

  float acc[NC][CPL];
#pragma unroll
  for (int j = 0; j < NC; ++j)
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[j][c] = 0.f;

  // the points of a query are staged LPB at a time (all of them at once for the usual L*P)
  for (int ib = 0; ib < LP; ib += LPB) {
    const int nI = min(LPB, LP - ib);
    if (ib > 0) __syncthreads();
    stage_chunk<LT, kStageFwd, false>(lv, loc, attn, lds, b, q0, nq, m, M, Lq, LP, P, ib, nI, LPB,
                                      rowb, headb, rs, EmitCtx{}, ref, row_stats, L,
                                      patch_order ? qidx : nullptr);
    __syncthreads();
    if (live) {
      // The row loads of U points are issued back to back and consumed while those of the next U
      // points are in flight (two register sets).  Left to the compiler's scheduler the number
      // of loads in flight is an accident of its register heuristics (from one load at a time
      // to eight, depending on the instantiation and on unrelated code).
      constexpr int NL = PAIR ? 2 : 4;  // row loads per point and lane
      constexpr int U = PAIR ? MSDA_FWD_U_PAIR : MSDA_FWD_U;  // points per register set
      struct Set {
        uint4 raw[U][NL];
        float w[U][NL];
      };
      const auto issue = [&](int i0, Set& st) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = min(i0 + u, nI - 1);  // a step past the end redoes the last point (unused)
          const uint4 wr = rec[2 * i], orr = rec[2 * i + 1];
          uint32_t o[NL];
          if constexpr (PAIR) {
            // the lane's column of the footprint: rows y0 and y0 + 1
            o[0] = right ? orr.y : orr.x;
            o[1] = right ? orr.w : orr.z;
            st.w[u][0] = __uint_as_float(right ? wr.y : wr.x);
            st.w[u][1] = __uint_as_float(right ? wr.w : wr.z);
          } else {
            o[0] = orr.x; o[1] = orr.y; o[2] = orr.z; o[3] = orr.w;
            st.w[u][0] = __uint_as_float(wr.x); st.w[u][1] = __uint_as_float(wr.y);
            st.w[u][2] = __uint_as_float(wr.z); st.w[u][3] = __uint_as_float(wr.w);
          }
#pragma unroll
          for (int k = 0; k < NL; ++k)
            st.raw[u][k] = *reinterpret_cast<const uint4*>(vb + ((o[k] & ~15u) + lane_b));
        }
      };
      const auto consume = [&](int i0, const Set& st) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (i0 + u < nI) {  // (uniform)
#pragma unroll
            for (int k = 0; k < NL; ++k) {
              float v[CPL];
              Chunk<VT>::unpack(st.raw[u][k], v);
#pragma unroll
              for (int c = 0; c < CPL; ++c) acc[0][c] = fmaf(st.w[u][k], v[c], acc[0][c]);
            }
          }
        }
      };
      Set sa, sb;
      issue(0, sa);
      for (int i0 = 0; i0 < nI; i0 += 2 * U) {
        if (i0 + U < nI) issue(i0 + U, sb);
        __builtin_amdgcn_sched_barrier(0);
        consume(i0, sa);
        __builtin_amdgcn_sched_barrier(0);
        if (i0 + 2 * U < nI) issue(i0 + 2 * U, sa);
        __builtin_amdgcn_sched_barrier(0);
        consume(i0 + U, sb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (!live) return;
  if constexpr (PAIR) {
    // left and right columns were summed apart: lane and lane + G hold the two halves of a sum
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[0][c] += __shfl_xor(acc[0][c], G);
    if (right) return;
  }
  const int64_t triple = (static_cast<int64_t>(b) * Lq + qidx[g]) * M + m;
#pragma unroll
  for (int j = 0; j < NC; ++j)
    Chunk<VT>::store(reinterpret_cast<char*>(out) + triple * D * sizeof(VT) + (lane + j * G) * 16,
                     acc[j]);
}

// ---------------------------------------------------------------------------------------------
// Backward, query side
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return static_cast<uint32_t>(
      __builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, ROW_MASK, 0xF, true));
}

// inclusive prefix sum over the 64 lanes of a wave, on the DPP path
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
  x += dpp_u32<0x111, 0xF>(x);  // row_shr:1
  x += dpp_u32<0x112, 0xF>(x);  // row_shr:2
  x += dpp_u32<0x114, 0xF>(x);  // row_shr:4
  x += dpp_u32<0x118, 0xF>(x);  // row_shr:8
  x += dpp_u32<0x142, 0xA>(x);  // row_bcast15 into rows 1 and 3
  x += dpp_u32<0x143, 0xC>(x);  // row_bcast31 into rows 2 and 3
  return x;
}

// Exclusive scan of data[0..n) in LDS by the whole 256-thread workgroup; returns the total.
// tmp: 4 words of LDS.  Starts with a barrier (data may have been written by any thread, tmp may
// still be read from a previous call) and ends with one.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t* data, int n, uint32_t* tmp) {
  const int per = (n + 255) / 256;
  const int lo = min(static_cast<int>(threadIdx.x) * per, n), hi = min(lo + per, n);
  __syncthreads();
  uint32_t sum = 0;
  for (int i = lo; i < hi; ++i) sum += data[i];
  const uint32_t incl = wave_inclusive_scan(sum);
  if ((threadIdx.x & 63) == 63) tmp[threadIdx.x >> 6] = incl;
  __syncthreads();
  const uint32_t t0 = tmp[0], t1 = tmp[1], t2 = tmp[2], t3 = tmp[3];
  const int wave = threadIdx.x >> 6;
  uint32_t run = incl - sum + (wave > 0 ? t0 : 0u) + (wave > 1 ? t1 : 0u) + (wave > 2 ? t2 : 0u);
  for (int i = lo; i < hi; ++i) {
    const uint32_t c = data[i];
    data[i] = run;
    run += c;
  }
  __syncthreads();
  return t0 + t1 + t2 + t3;
}


// MODE 0: grad_loc / grad_attn + grad_value by global fp32 atomics into `gv_acc`
// MODE 1: grad_loc / grad_attn + scatter records for the tile kernel
// KPC > 0 (MODE 1, 4 lanes per query, 3 chunks of 64 queries per workgroup, all L*P points of a
// query staged in one pass, L*P <= 4 * KPC): every thread reads the locations and weights of ALL
// its points of the workgroup — KPC per chunk — at the start of the kernel with independent loads
// and keeps them in registers.  The count pass and the three staging passes then work from
// registers: the tensors are read once instead of twice, and the nine dependent global round
// trips per thread of the count pass plus three per staging pass (each waits behind the gathers
// of the other workgroups of the CU) become one.
// SPLIT (KPC > 0): rows of more than 16 points are taken as SPLIT "virtual" queries of L*P / SPLIT
// points each — point i of virtual query v of real query q is point (v % SPLIT) * (L*P / SPLIT) + i
// of q — so that 32 points per query (4 levels x 8 points) run this form too: QB counts virtual
// queries, a workgroup owns QB / SPLIT real ones (plan.q_per_block), records carry the real
// query index, and the two halves of a row sit in neighbouring lane groups of one wave.
template <typename VT, typename LT, int G, int MODE, int KPC = 0, int QB = 192, int SPLIT = 1>
__global__ __launch_bounds__(256, MSDA_K3_MINWAVES) void bwd_query_kernel(
    const VT* __restrict__ grad_out, const VT* __restrict__ value,
    const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const LT* __restrict__ loc, const LT* __restrict__ attn, LT* __restrict__ grad_loc,
    LT* __restrict__ grad_attn, float* __restrict__ gv_acc, uint4* __restrict__ records,
    uint2* __restrict__ dir, VT* __restrict__ g_copy, int S, int M, int Lq, int L, int P, int LPB,
    int head_major, PointRows rs, BinPlan plan, const float* __restrict__ ref) {
  constexpr int CPL = Chunk<VT>::kCPL;
  constexpr int D = G * CPL;
  constexpr int GQ = query_lanes(G);  // lanes per query
  constexpr int NC = G / GQ;          // 16-byte chunks of a row per lane: lane, lane + GQ, ...
  constexpr int GPB = 256 / GQ;
  constexpr int KIND = MODE == 1 ? kStageBwdBinned : kStageBwdAtomic;
  constexpr int SL = stage_slots(KIND);
  constexpr int NCHUNK = QB / GPB;                         // KPC > 0: chunks of GPB queries per workgroup
  constexpr int PRE = KPC > 0 ? NCHUNK * KPC : 1;      // points a thread keeps in registers
  static_assert(KPC == 0 || MODE == 1, "register-resident points: binned mode only");
  static_assert(SPLIT == 1 || (KPC > 0 && (SPLIT == 2 || SPLIT == 4) && GQ * SPLIT <= 16),
                "virtual queries: register-resident form, halves or quarters of a row");
  extern __shared__ uint4 lds[];
  __shared__ LevelInfo lv[kMaxLevelsTuned];
  __shared__ TileMap tm;
  __shared__ uint32_t scan_tmp[4];

  const int LP = L * P;
  const int LPH = LP / SPLIT;  // points per (virtual) query
  const int b = blockIdx.y;
  const int m = blockIdx.x % M;
  const int qblock = blockIdx.x / M;
  const int qb0 = qblock * plan.q_per_block;              // first (real) query of the workgroup
  const int qb_n = min(plan.q_per_block, Lq - qb0);       // its real queries
  const int vq_n = qb_n * SPLIT;                          // its virtual queries
  MSDA_STAMP_DECL;
  if constexpr (MSDA_K3_PRIO == 1 && sizeof(VT) == 4) __builtin_amdgcn_s_setprio(3);
  // raw points (KPC > 0 only; the host refuses them elsewhere): softmax statistics of the rows
  __shared__ float2 row_stats[KPC > 0 ? QB / SPLIT : 1];
  // KPC > 0: the (real) queries of the workgroup — work items qb0 .. qb0 + qb_n - 1 of the query
  // order (order_query); everything below addresses a query through this table
  __shared__ int qmap[KPC > 0 ? QB / SPLIT : 1];
  const bool raw = KPC > 0 && ref != nullptr;
  if (threadIdx.x < L) lv[threadIdx.x] = load_level(shapes, lsi, threadIdx.x, S);
  if constexpr (KPC > 0) {
    if (static_cast<int>(threadIdx.x) < qb_n) {
      const int q = order_query(qb0 + threadIdx.x, shapes, lsi, S, L, Lq, plan.patch_order != 0);
      qmap[threadIdx.x] = q;
      if (raw)
        row_stats[threadIdx.x] =
            softmax_stats(attn + ((static_cast<int64_t>(b) * Lq + q) * M + m) * rs.attn, L * P);
    }
  }
  if (MODE == 1 && plan.tile2d && threadIdx.x == 255) build_tile_map(shapes, lsi, S, L, plan, tm);
  __syncthreads();
  MSDA_STAMP(0);  // level table + query table + tile map

  const uint32_t rowb = static_cast<uint32_t>(head_major ? 1 : M) * D * sizeof(VT);
  const uint32_t headb = static_cast<uint32_t>(m) * (head_major ? S : 1) * D * sizeof(VT);
  // LDS: [tile counters | tile offsets] (binned mode), staged points, parked points (KPC > 0)
  const int stage_total = GPB * group_stride_slots(LPB, SL);
  // one word per tile (+ 1): record count, then (scanned in place) offset inside the region,
  // then the cursor the records are placed with
  const int tile_slots = MODE == 1 ? (plan.n_tiles + 1 + 3) / 4 : 0;  // 16-byte slots
  uint32_t* tile_cnt = reinterpret_cast<uint32_t*>(lds);
  uint4* stage = lds + tile_slots;
  EmitCtx emit{};

  // KPC > 0: the thread's points.  Point j of a chunk is the chunk's point t = threadIdx.x +
  // 256 * j, i.e. point `i` (level `l`) of query `q` of the chunk — the same for every chunk;
  // px / py / pa hold location and weight of point j of chunk c at [c * KPC + j].
  float px[PRE], py[PRE], pa[PRE];
  uint32_t pdesc[PRE];  // per point: record mask and destination tiles, count pass -> emit pass
  const float inv_LPH = 1.0f / static_cast<float>(LPH), inv_P = 1.0f / static_cast<float>(P);
  // point j of a chunk: virtual query q of the chunk, slot ih of its LPH points, which is point i
  // (level l) of the real query
  const auto point_of = [&](int j, int& q, int& ih, int& i, int& l) {
    int p_unused;
    // (MSDA_K3_PSEQ: a thread takes KPC consecutive points of one query — LPH is a multiple of
    // KPC then — so that their loads are neighbours in memory)
    if (MSDA_K3_PSEQ && KPC > 0 && LPH % KPC == 0)
      small_divmod(static_cast<int>(threadIdx.x) * KPC + j, LPH, inv_LPH, q, ih);
    else
      small_divmod(static_cast<int>(threadIdx.x) + 256 * j, LPH, inv_LPH, q, ih);
    i = SPLIT == 1 ? ih : __mul24(q % SPLIT, LPH) + ih;
    small_divmod(i, P, inv_P, l, p_unused);
  };
  // pending chunks' points, parked in LDS while a chunk is gathered: [(c - 1) * KPC + j][x, y, a][thread]
  float* stash = reinterpret_cast<float*>(stage + stage_total);
  if constexpr (KPC > 0) {
    // rows of this workgroup: uniform 64-bit bases, small per-point offsets (from the block's
    // first row, or from the batch element's when the block's queries are not consecutive)
    const int q_base = plan.patch_order ? 0 : qb0;
    const int64_t row0 = (static_cast<int64_t>(b) * Lq + q_base) * M + m;
    const LT* loc_wg = loc + row0 * rs.loc;
    const LT* attn_wg = attn + row0 * rs.attn;
    const float* ref_wg = raw ? ref + (static_cast<int64_t>(b) * Lq + q_base) * L * 2 : nullptr;
    const bool pseq = MSDA_K3_PSEQ && LPH % KPC == 0;  // (uniform)
    if (pseq) {
      // the thread's KPC points of a chunk are neighbours in one row: address once, loads side
      // by side (the compiler merges them into wider ones)
      int q, ih0, i0, l0;
      point_of(0, q, ih0, i0, l0);
#pragma unroll
      for (int c = 0; c < NCHUNK; ++c) {
        const int qi = c * GPB + q;
        const bool have = q < GPB && qi < vq_n;
        const uint32_t qc = have ? static_cast<uint32_t>(qi) / SPLIT : 0;
        const uint32_t qo = static_cast<uint32_t>(qmap[qc] - q_base);
        const uint32_t row = __umul24(qo, M);
        const LT* xy_p = loc_wg + (__umul24(row, static_cast<uint32_t>(rs.loc)) + 2u * i0);
        const LT* a_p = attn_wg + (__umul24(row, static_cast<uint32_t>(rs.attn)) + i0);
        float xs[KPC], ys[KPC], as[KPC];
        if constexpr (sizeof(LT) == 4) {
          float2 xy[KPC];
          float aw[KPC];
          __builtin_memcpy(xy, xy_p, sizeof(xy));  // (8-byte aligned; KPC pairs side by side)
          __builtin_memcpy(aw, a_p, sizeof(aw));
#pragma unroll
          for (int j = 0; j < KPC; ++j) {
            xs[j] = xy[j].x;
            ys[j] = xy[j].y;
            as[j] = aw[j];
          }
        } else {
          LT pair[KPC][2], aw[KPC];
          __builtin_memcpy(pair, xy_p, sizeof(pair));
          __builtin_memcpy(aw, a_p, sizeof(aw));
#pragma unroll
          for (int j = 0; j < KPC; ++j) {
            xs[j] = Elem<LT>::load(&pair[j][0]);
            ys[j] = Elem<LT>::load(&pair[j][1]);
            as[j] = Elem<LT>::load(&aw[j]);
          }
        }
#pragma unroll
        for (int j = 0; j < KPC; ++j) {
          float x = xs[j], y = ys[j], a = as[j];
          if (raw) {
            int l, p_unused;
            small_divmod(i0 + j, P, inv_P, l, p_unused);
            raw_to_point(ref_wg + (__umul24(qo, L) + l) * 2u, lv[l], row_stats[qc], x, y, a);
          }
          px[c * KPC + j] = have ? x : -4.f;
          py[c * KPC + j] = have ? y : -4.f;
          pa[c * KPC + j] = have ? a : 0.f;
        }
      }
    } else
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
      for (int j = 0; j < KPC; ++j) {
        // every load is issued (from a clamped, valid row) and the result dropped afterwards when
        // the point does not exist: a branch around a load would make each one wait for the last
        int q, ih, i, l;
        point_of(j, q, ih, i, l);
        const int qi = c * GPB + q;
        const bool have = q < GPB && qi < vq_n;
        const uint32_t qc = have ? static_cast<uint32_t>(qi) / SPLIT : 0;  // real query of the block
        const uint32_t qo = static_cast<uint32_t>(qmap[qc] - q_base);
        float x, y, a;
        load_point_near(loc_wg, attn_wg, __umul24(qo, M), i, rs, x, y, a);
        if (raw) raw_to_point(ref_wg + (__umul24(qo, L) + l) * 2u, lv[l], row_stats[qc], x, y, a);
        px[c * KPC + j] = have ? x : -4.f;  // outside every level: no footprint
        py[c * KPC + j] = have ? y : -4.f;
        pa[c * KPC + j] = have ? a : 0.f;
      }
  }
  MSDA_STAMP(1);  // point loads issued

  // KPC > 0: the directory and the records may exist already (written by an earlier launch of
  // this kernel with kPhaseBin only, on the same points): then only the gradients are left
  const bool do_bin = KPC == 0 || (plan.phase & kPhaseBin) != 0;
  if (MODE == 1 && !(kAblate & 1)) {
   if (do_bin) {
    // ---- count the records of this workgroup per destination tile ----
    for (int t = threadIdx.x; t < plan.n_tiles; t += 256) tile_cnt[t] = 0;
    __syncthreads();
    MSDA_STAMP(2);  // counters zeroed
    if constexpr (KPC > 0) {
#pragma unroll
      for (int k = 0; k < PRE; ++k) {
        int q, ih, i, l;
        point_of(k % KPC, q, ih, i, l);
        const LevelInfo L_ = lv[l];
        const Footprint<float> f = footprint<float, true>(px[k], py[k], L_.H, L_.W);
        uint32_t desc = 0;
        if (f.any && pa[k] != 0.f) {
          uint32_t tiles[4];
          bool keep[4];
          footprint_tiles(f, plan, tm, L_, l, tiles, keep);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (keep[j]) {
              atomicAdd(&tile_cnt[tiles[j]], 1u);
              desc |= 1u << j;
            }
          // what the emit pass needs of this footprint: which corners send a record, and their
          // tiles — corner 1 / 3 sits in the tile of corner 0 / 2 or in the next one (tile
          // indices are below 4096)
          desc |= (tiles[0] & 0xfffu) << 4 | (tiles[2] & 0xfffu) << 16 |
                  ((tiles[1] - tiles[0]) & 1u) << 28 | ((tiles[3] - tiles[2]) & 1u) << 29;
        }
        pdesc[k] = desc;
      }
    } else {
      const int npts = qb_n * LP;
      const float inv_LP = 1.0f / static_cast<float>(LP), inv_P = 1.0f / static_cast<float>(P);
      const int64_t row0 = (static_cast<int64_t>(b) * Lq + qb0) * M + m;
      const LT* loc_wg = loc + row0 * rs.loc;
      const LT* attn_wg = attn + row0 * rs.attn;
      for (int t = threadIdx.x; t < npts; t += 256) {
        int g, i, l, p_unused;
        small_divmod(t, LP, inv_LP, g, i);
        small_divmod(i, P, inv_P, l, p_unused);
        const LevelInfo L_ = lv[l];
        float x, y, a;
        load_point_near(loc_wg, attn_wg, __umul24(g, M), i, rs, x, y, a);
        const Footprint<float> f = footprint<float, true>(x, y, L_.H, L_.W);
        if (f.any && a != 0.f) {
          uint32_t tiles[4];
          bool keep[4];
          footprint_tiles(f, plan, tm, L_, l, tiles, keep);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (keep[j]) atomicAdd(&tile_cnt[tiles[j]], 1u);
        }
      }
    }
    __syncthreads();
    MSDA_STAMP(3);  // count pass (waits for the point loads)
    // directory row of this workgroup: (offset, count) per tile; the offsets then serve as the
    // cursors of the emit pass
    const int64_t slab = static_cast<int64_t>(b) * M + m;
    uint2* dir_row = dir + (slab * plan.n_qblocks + qblock) * plan.n_tiles;
    const uint32_t n_records = block_exclusive_scan(tile_cnt, plan.n_tiles, scan_tmp);
    if (threadIdx.x == 0) tile_cnt[plan.n_tiles] = n_records;
    __syncthreads();
    for (int t = threadIdx.x; t < plan.n_tiles; t += 256)
      dir_row[t] = make_uint2(tile_cnt[t], tile_cnt[t + 1] - tile_cnt[t]);
    emit.records = records + (slab * plan.n_qblocks + qblock) * static_cast<int64_t>(plan.region);
    emit.tile_cur = tile_cnt;
    emit.plan = plan;
    emit.tm = &tm;
    emit.lds_image = nullptr;
    emit.lds_cap = 0;
    if constexpr (KPC > 0 && MSDA_K3_LDS_EMIT) {
      // the staging and parking areas are still free: they hold the head of the record region
      emit.lds_image = stage;
      emit.lds_cap = static_cast<uint32_t>(stage_total + (NCHUNK - 1) * KPC * 3 * 256 * 4 / 16);
    }
    __syncthreads();
    MSDA_STAMP(4);  // scan + directory row
    if constexpr (KPC > 0) {
      // all records of the workgroup now, while its points are in registers.  They are sorted by
      // tile as they are placed (offset of the tile + rank), first into an LDS image of the
      // region, which then goes out as one linear copy of whole 16-byte lanes: a record store
      // issued per point lands in 64 different lines per wave instruction.  Records beyond the
      // image (uniformly scattered locations produce more) are stored directly.
#pragma unroll
      for (int k = 0; k < PRE; ++k) {
        const uint32_t desc = pdesc[k];  // from the count pass: no second footprint
        if (desc & 15u) {
          int q, ih, i, l;
          point_of(k % KPC, q, ih, i, l);
          const uint32_t t0 = (desc >> 4) & 0xfffu, t2 = (desc >> 16) & 0xfffu;
          const uint32_t tiles[4] = {t0, t0 + ((desc >> 28) & 1u), t2, t2 + ((desc >> 29) & 1u)};
          uint4 rec;
          rec.x = static_cast<uint32_t>(qmap[((k / KPC) * GPB + q) / SPLIT]) |
                  (static_cast<uint32_t>(l) << 28);
          rec.y = __float_as_uint(pa[k]);
          rec.z = __float_as_uint(px[k]);
          rec.w = __float_as_uint(py[k]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if ((desc >> j) & 1u) {
              const uint32_t at = atomicAdd(&tile_cnt[tiles[j]], 1u);
              if (at < emit.lds_cap) {
                emit.lds_image[at] = rec;  // leaves the workgroup later, as part of one linear copy
              } else if (at < plan.region) {  // always true: a bug stays in the region
                emit.records[at] = rec;
              }
            }
          }
        }
      }
      if (emit.lds_cap > 0) {
        __syncthreads();
        const uint32_t n_copy = min(n_records, min(emit.lds_cap, plan.region));
        for (uint32_t t = threadIdx.x; t < n_copy; t += 256) emit.records[t] = emit.lds_image[t];
        __syncthreads();
      }
    }
   }  // do_bin
    if constexpr (KPC > 0) {
      if (!(plan.phase & kPhaseGrad)) return;  // binning only (uniform)
      // then the points of the second and third chunk are parked in LDS (the gather loop needs
      // the registers) and the first chunk is staged
#pragma unroll
      for (int k = KPC; k < PRE; ++k) {
        stash[((k - KPC) * 3 + 0) * 256 + threadIdx.x] = px[k];
        stash[((k - KPC) * 3 + 1) * 256 + threadIdx.x] = py[k];
        stash[((k - KPC) * 3 + 2) * 256 + threadIdx.x] = pa[k];
      }
      const int stride = group_stride_slots(LPB, SL);
      const int nq0 = min(GPB, vq_n);
#pragma unroll
      for (int j = 0; j < KPC; ++j) {
        int q, ih, i, l;
        point_of(j, q, ih, i, l);
        if (q < nq0)
          stage_point<KIND, false>(lv[l], l, px[j], py[j], pa[j], 0 /* no records here */,
                                   stage + (__umul24(q, stride) + SL * ih), rowb, headb, emit, raw);
      }
    }
  }

  if constexpr (MSDA_K3_PRIO == 1 && sizeof(VT) == 4) __builtin_amdgcn_s_setprio(0);
  if constexpr (MSDA_K3_PRIO == 2) __builtin_amdgcn_s_setprio(3);
  const int g = threadIdx.x / GQ, lane = threadIdx.x % GQ;
  const int64_t slab_elems = static_cast<int64_t>(b) * S * M * D;
  const char* vb = reinterpret_cast<const char*>(value) + slab_elems * static_cast<int64_t>(sizeof(VT));
  const uint32_t lane_b = lane * 16;

  // chunks of GPB queries; inside a chunk the points are staged LPB at a time (LPB is a multiple
  // of G, and all of L*P for the usual sizes)
  // (virtual queries, LPH points each: the real ones with all L*P points unless SPLIT > 1)
  for (int c0 = 0; c0 < vq_n; c0 += GPB)
  for (int ib = 0; ib < LPH; ib += LPB) {
    const int nq = min(GPB, vq_n - c0);
    const int nI = min(LPB, LPH - ib);
    if constexpr (KPC > 0) {
      // staged already: the first chunk by the prologue, the others behind the previous gather
    } else {
      stage_chunk<LT, KIND, MODE == 1>(lv, loc, attn, stage, b, qb0 + c0, nq, m, M, Lq, LP, P, ib,
                                       nI, LPB, rowb, headb, rs, emit);
    }
    MSDA_STAMP(5);  // staging + record emit (own share)
    __syncthreads();
    MSDA_STAMP(6);  // barrier behind the staging
    // lanes of groups past the end redo the last group (no writes) so that every DPP row is full
    const bool live = g < nq;
    const int gg = live ? g : nq - 1;
    const uint4* r = stage + __umul24(gg, group_stride_slots(LPB, SL));
    // the real query of this group
    const int rq = KPC > 0 ? qmap[(c0 + gg) / SPLIT] : qb0 + (c0 + gg) / SPLIT;
    const int half_i = ((c0 + gg) % SPLIT) * LPH;  // first point of the group's part of its row
    const int64_t triple = (static_cast<int64_t>(b) * Lq + rq) * M + m;
    float go[NC][CPL];
    uint4 go_raw[NC];  // the same chunks as loaded (16-bit types: still packed)
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      go_raw[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(grad_out) +
                                                  triple * D * sizeof(VT) + (lane + j * GQ) * 16);
      Chunk<VT>::unpack(go_raw[j], go[j]);
    }
    if (g_copy != nullptr && ib == 0 && live && half_i == 0) {
      // the row again, where the tile kernel will look for it: (batch, head, query) order
      const int64_t hm_row = (static_cast<int64_t>(b) * M + m) * Lq + rq;
#pragma unroll
      for (int j = 0; j < NC; ++j)
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(g_copy) + hm_row * D * sizeof(VT) +
                                  (lane + j * GQ) * 16) = go_raw[j];
    }
    MSDA_STAMP(7);  // grad_out row load (+ copy store issue)

    if constexpr (MODE == 1 && MSDA_K3_DEPTH == 2) {
      // Two points in flight: the eight row loads of point i + 1 are issued before the dot
      // products of point i are taken, so a lane always has 8-16 loads outstanding instead of
      // 0-8 (the kernel runs at 3 waves per SIMD: what hides the gather latency is the number of
      // loads each wave keeps in flight).  The rows come in through a buffer descriptor of the
      // batch element's value slab: one 32-bit offset register per load, and an offset outside
      // the slab reads zeros instead of faulting.
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const uint32_t slab_bytes = static_cast<uint32_t>(S) * M * D * sizeof(VT);
      const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(vb), 0, static_cast<int>(slab_bytes), 0x00020000);
      // (the fractions of a point are read from LDS when its rows are consumed, not held in
      // registers while the loads are in flight)
      struct PointRegs {
        int i;
        u32x4 raw[4][NC];
      };
      const auto issue = [&](int i, PointRegs& pr) {
        const uint4 orr = r[SL * i + SL - 2];
        pr.i = i;
        const uint32_t oraw[4] = {orr.x, orr.y, orr.z, orr.w};
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
          for (int j = 0; j < NC; ++j)
            pr.raw[c4][j] = __builtin_amdgcn_raw_buffer_load_b128(
                vres, static_cast<int>(oraw[c4] + lane_b + j * GQ * 16), 0, 0);
      };
      float keep_sv = 0.f, keep_sx = 0.f, keep_sy = 0.f, keep_a = 0.f;
      // raw points: weight and weight gradient of this lane's point of every pass, for the
      // softmax backward d logit_i = w_i * (g_i - sum_k w_k g_k) once all points are through
      float ka[4] = {0.f, 0.f, 0.f, 0.f}, kg[4] = {0.f, 0.f, 0.f, 0.f};
      const auto consume_to = [&](const PointRegs& pr, int k, float& o_sv, float& o_sx, float& o_sy,
                                  float& o_a) {
        const uint4 fr = r[SL * pr.i + SL - 1];
        const float lx = __uint_as_float(fr.x), ly = __uint_as_float(fr.y);
        const float aW = __uint_as_float(fr.z), aH = __uint_as_float(fr.w);
        const float hx = 1.f - lx, hy = 1.f - ly;
        float dot[4];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          float d = 0.f;
#pragma unroll
          for (int j = 0; j < NC; ++j)
            d = Chunk<VT>::dot(go_raw[j], __builtin_bit_cast(uint4, pr.raw[c4][j]), d);
          dot[c4] = d;  // (a corner outside the level read zeros)
        }
        // w = {hy*hx, hy*lx, ly*hx, ly*lx}; dw/dx = {-hy, hy, -ly, ly}; dw/dy = {-hx, -lx, hx, lx}
        float sv = hy * fmaf(hx, dot[0], lx * dot[1]) + ly * fmaf(hx, dot[2], lx * dot[3]);
        float sx = fmaf(hy, dot[1] - dot[0], ly * (dot[3] - dot[2]));
        float sy = fmaf(hx, dot[2] - dot[0], lx * (dot[3] - dot[1]));
        sv = group_sum<GQ>(sv);
        sx = group_sum<GQ>(sx);
        sy = group_sum<GQ>(sy);
        if (lane == k) {
          o_sv = sv;
          o_sx = aW * sx;
          o_sy = aH * sy;
          o_a = aW;  // raw points stage the weight itself here
        }
      };
      const auto consume = [&](const PointRegs& pr, int k) {
        consume_to(pr, k, keep_sv, keep_sx, keep_sy, keep_a);
      };
      // four points per pass whatever the group width: lane j < 4 of the group keeps the results
      // of point i0 + j (with 8 or 16 lanes per query a pass of that many points would be partly
      // empty for 12 points per query)
      constexpr int PP = 4;
      PointRegs pa_, pb_;
      // Whole rows in KPC passes (nI = 4 * KPC: every BASELINE shape): lane j of the group keeps
      // the results of the KPC NEIGHBOURING points j * KPC .. j * KPC + KPC - 1 — pass p takes
      // points {p, KPC + p, 2 KPC + p, 3 KPC + p} — and stores them once, side by side, after the
      // last pass: three store instructions per chunk instead of two per pass.
      constexpr int KP = KPC > 0 ? KPC : 1;
      bool stored = false;
      if constexpr (KPC > 0 && MSDA_K3_WSTORE) {
        if (nI == PP * KPC && ib == 0) {  // (uniform)
          stored = true;
          float r_sv[KP], r_sx[KP], r_sy[KP], r_a[KP];
#pragma unroll
          for (int p = 0; p < KP; ++p) r_sv[p] = r_sx[p] = r_sy[p] = r_a[p] = 0.f;
          issue(0, pa_);
#pragma unroll
          for (int s2 = 0; s2 < PP * KP; s2 += 2) {
            // point of step s: (s % PP) * KPC + s / PP
            issue(((s2 + 1) % PP) * KP + (s2 + 1) / PP, pb_);
            __builtin_amdgcn_sched_barrier(0);
            consume_to(pa_, s2 % PP, r_sv[s2 / PP], r_sx[s2 / PP], r_sy[s2 / PP], r_a[s2 / PP]);
            __builtin_amdgcn_sched_barrier(0);
            if (s2 + 2 < PP * KP) issue(((s2 + 2) % PP) * KP + (s2 + 2) / PP, pa_);
            __builtin_amdgcn_sched_barrier(0);
            consume_to(pb_, (s2 + 1) % PP, r_sv[(s2 + 1) / PP], r_sx[(s2 + 1) / PP],
                       r_sy[(s2 + 1) / PP], r_a[(s2 + 1) / PP]);
            __builtin_amdgcn_sched_barrier(0);
          }
          const bool mine = lane < PP && live && !(kAblate & 2);
          const int i_first = half_i + lane * KP;  // the lane's first point in the row
          if (raw) {
            float part = 0.f;
#pragma unroll
            for (int p = 0; p < KP; ++p) part = fmaf(lane < PP ? r_a[p] : 0.f, r_sv[p], part);
            // sum_k w_k g_k over the whole row: the lane groups of its SPLIT parts are neighbours
            const float dot_wg = group_sum<GQ * SPLIT>(part);
#pragma unroll
            for (int p = 0; p < KP; ++p) r_sv[p] = r_a[p] * (r_sv[p] - dot_wg);
          }
          if (mine) {
            LT o_attn[KP], o_loc[2 * KP];
#pragma unroll
            for (int p = 0; p < KP; ++p) {
              Elem<LT>::store(&o_attn[p], r_sv[p]);
              Elem<LT>::store(&o_loc[2 * p], r_sx[p]);
              Elem<LT>::store(&o_loc[2 * p + 1], r_sy[p]);
            }
            __builtin_memcpy(grad_attn + triple * rs.attn + i_first, o_attn, sizeof(o_attn));
            __builtin_memcpy(grad_loc + triple * rs.loc + 2 * i_first, o_loc, sizeof(o_loc));
          }
        }
      }
      if (!stored) {
      issue(0, pa_);
      for (int i0 = 0; i0 < nI; i0 += PP) {
        // a pass past the end redoes the last point (results unused)
#pragma unroll
        for (int k = 0; k < PP; k += 2) {
          issue(min(i0 + k + 1, nI - 1), pb_);
          __builtin_amdgcn_sched_barrier(0);
          consume(pa_, k);
          __builtin_amdgcn_sched_barrier(0);
          if (k + 2 < PP) {
            issue(min(i0 + k + 2, nI - 1), pa_);
          } else if (i0 + PP < nI) {
            issue(i0 + PP, pa_);  // first point of the next pass
          }
          __builtin_amdgcn_sched_barrier(0);
          consume(pb_, k + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        const int i = i0 + lane;
        if (raw) {
          const bool mine = lane < PP && i < nI;
#pragma unroll
          for (int ps = 0; ps < 4; ++ps)
            if (i0 == ps * PP) {
              ka[ps] = mine ? keep_a : 0.f;
              kg[ps] = mine ? keep_sv : 0.f;
            }
        }
        if (lane < PP && i < nI && live && !(kAblate & 2)) {
          if (!raw) Elem<LT>::store(grad_attn + triple * rs.attn + half_i + ib + i, keep_sv);
          LT* gl = grad_loc + triple * rs.loc + 2 * (half_i + ib + i);
          if constexpr (sizeof(LT) == 4) {
            *reinterpret_cast<float2*>(gl) = make_float2(keep_sx, keep_sy);
          } else {
            Elem<LT>::store(gl, keep_sx);
            Elem<LT>::store(gl + 1, keep_sy);
          }
        }
      }
      if (raw) {
        float part = 0.f;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) part = fmaf(ka[ps], kg[ps], part);
        // sum_k w_k g_k over the whole row: the lane groups of its SPLIT parts are neighbours
        const float dot_wg = group_sum<GQ * SPLIT>(part);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int i = ps * PP + lane;
          if (lane < PP && i < nI && live && !(kAblate & 2))
            Elem<LT>::store(grad_attn + triple * rs.attn + half_i + ib + i, ka[ps] * (kg[ps] - dot_wg));
        }
      }
      }  // !stored
    } else {
    // GQ points per pass: after the pass lane j of the group keeps the results of point i0 + j,
    // so the group writes GQ consecutive grad_attn / grad_loc elements with one store each
    for (int i0 = 0; i0 < nI; i0 += GQ) {
      float keep_sv = 0.f, keep_sx = 0.f, keep_sy = 0.f;
#pragma unroll
      for (int k = 0; k < GQ; ++k) {
        const int i = min(i0 + k, nI - 1);  // a pass past the end redoes the last point (unused)
        const uint4 orr = r[SL * i + SL - 2], fr = r[SL * i + SL - 1];
        const uint32_t oraw[4] = {orr.x, orr.y, orr.z, orr.w};
        const float lx = __uint_as_float(fr.x), ly = __uint_as_float(fr.y);
        const float aW = __uint_as_float(fr.z), aH = __uint_as_float(fr.w);
        const float hx = 1.f - lx, hy = 1.f - ly;
        float v[4][NC][CPL];
        uint32_t o[4];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          o[c4] = (oraw[c4] & ~15u) + lane_b;
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            // plan.in_order is always 1, but the compiler cannot know: every gather then sits
            // in a basic block of its own and is issued in program order, all loads of a point
            // back to back.  Left to itself the scheduler interleaves them with the arithmetic
            // to save registers, which keeps fewer rows in flight (measured: 3 % slower).
            if (plan.in_order && !(kAblate & 4)) {
              Chunk<VT>::load(vb + o[c4] + j * GQ * 16, v[c4][j]);
            } else {
#pragma unroll
              for (int c = 0; c < CPL; ++c) v[c4][j][c] = __uint_as_float(o[c4] + c);
            }
          }
        }
        float dot[4];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          float d = 0.f;
#pragma unroll
          for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int c = 0; c < CPL; ++c) d = fmaf(go[j][c], v[c4][j][c], d);
          dot[c4] = (oraw[c4] & 1u) ? d : 0.f;  // a corner outside the level contributes nothing
        }
        if constexpr (MODE == 0) {
          const uint4 wr = r[SL * i];
          const float aw[4] = {__uint_as_float(wr.x), __uint_as_float(wr.y),
                               __uint_as_float(wr.z), __uint_as_float(wr.w)};
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            if (live && aw[c4] != 0.f && i0 + k < nI) {
#pragma unroll
              for (int j = 0; j < NC; ++j) {
                float* dst = gv_acc + slab_elems + (o[c4] + j * GQ * 16) / sizeof(VT);
#pragma unroll
                for (int c = 0; c < CPL; ++c) atomicAdd(dst + c, aw[c4] * go[j][c]);
              }
            }
          }
        }
        // w = {hy*hx, hy*lx, ly*hx, ly*lx}; dw/dx = {-hy, hy, -ly, ly}; dw/dy = {-hx, -lx, hx, lx}
        float sv = hy * fmaf(hx, dot[0], lx * dot[1]) + ly * fmaf(hx, dot[2], lx * dot[3]);
        float sx = fmaf(hy, dot[1] - dot[0], ly * (dot[3] - dot[2]));
        float sy = fmaf(hx, dot[2] - dot[0], lx * (dot[3] - dot[1]));
        sv = group_sum<GQ>(sv);
        sx = group_sum<GQ>(sx);
        sy = group_sum<GQ>(sy);
        if (lane == k) {
          keep_sv = sv;
          keep_sx = aW * sx;
          keep_sy = aH * sy;
        }
      }
      const int i = i0 + lane;
      if (i < nI && live && !(kAblate & 2)) {
        // the gradients have the layout of their inputs (separate tensors or one packed row)
        Elem<LT>::store(grad_attn + triple * rs.attn + half_i + ib + i, keep_sv);
        LT* gl = grad_loc + triple * rs.loc + 2 * (half_i + ib + i);
        if constexpr (sizeof(LT) == 4) {
          *reinterpret_cast<float2*>(gl) = make_float2(keep_sx, keep_sy);
        } else {
          Elem<LT>::store(gl, keep_sx);
          Elem<LT>::store(gl + 1, keep_sy);
        }
      }
    }
    }
    MSDA_STAMP(8);  // gather + dot products + gradient stores
    __syncthreads();
    MSDA_STAMP(9);  // barrier behind the gather
    if constexpr (KPC > 0) {
      // the next chunk's points come back from where this thread parked them
      const int cn = c0 / GPB + 1;
      const int nqn = min(GPB, vq_n - cn * GPB);
      if (nqn > 0) {
        const int stride = group_stride_slots(LPB, SL);
#pragma unroll
        for (int j = 0; j < KPC; ++j) {
          int q, ih, i, l;
          point_of(j, q, ih, i, l);
          const float x = stash[(((cn - 1) * KPC + j) * 3 + 0) * 256 + threadIdx.x];
          const float y = stash[(((cn - 1) * KPC + j) * 3 + 1) * 256 + threadIdx.x];
          const float a = stash[(((cn - 1) * KPC + j) * 3 + 2) * 256 + threadIdx.x];
          if (q < nqn)
            stage_point<KIND, false>(lv[l], l, x, y, a, 0 /* no records here */,
                                     stage + (__umul24(q, stride) + SL * ih), rowb, headb, emit, raw);
        }
      }
    }
  }
  MSDA_STAMP_FLUSH(0);
}

// ---------------------------------------------------------------------------------------------
// Backward, destination side, gather walk: one workgroup per (batch, head, tile of 64 positions)
//
// The tile's records are taken in rounds of up to R.  A round bucket-sorts the in-tile corners of
// its records by destination position in LDS (count, one-wave DPP scan, fill; integer atomics
// only), which leaves per position a contiguous list of (query, weight) entries.  The 4 lanes
// that own a position for the whole life of the workgroup then walk its list, U entries at a
// time, fetch the grad_out rows of those entries straight from global memory (they sit in L2:
// the queries that hit one patch are neighbours) and accumulate in registers.  No LDS row stage,
// no float atomics, four barriers per round; the tile is stored from registers at the end.
// ---------------------------------------------------------------------------------------------
template <typename VT, int G, int RPT, int U>
__global__ __launch_bounds__(256) void bwd_tile_gather_kernel(
    const VT* __restrict__ grad_out, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const uint4* __restrict__ records,
    const uint2* __restrict__ dir, VT* __restrict__ grad_value, int S, int M, int Lq, int L,
    BinPlan plan, int g_head_major) {
  constexpr int CPL = Chunk<VT>::kCPL;
  constexpr int D = G * CPL;
  constexpr int PL = 4;                     // lanes per destination position
  constexpr int TS = 256 / PL;              // positions per tile: one per lane group, = wave size
  constexpr int NCH = D / (PL * CPL);       // 16-byte chunks of a row per lane
  constexpr int R = 256 * RPT;              // records per round
  constexpr int ROWB = D * sizeof(VT);      // bytes of one grad_out row of one head
  static_assert(NCH >= 1 && NCH * PL * CPL == D, "head dim must split into 4 lanes of chunks");
  static_assert(TS == 64, "the bucket scan is one wave wide");
  extern __shared__ uint4 lds[];
  __shared__ LevelInfo lv[kMaxLevelsTuned];
  __shared__ TileMap tm;
  __shared__ uint32_t n_runs_s;
  __shared__ uint32_t scan_tmp[4];
  __shared__ int tile_level_s, tile_y0_s, tile_x0_s;

  uint2* entries = reinterpret_cast<uint2*>(lds);                    // 4 * R
  uint32_t* cnt2 = reinterpret_cast<uint32_t*>(entries + 4 * R);     // 2 x TS: counts, then cursors
  uint32_t* start = cnt2 + 2 * TS;                                   // TS + 1
  uint32_t* run_first = start + TS + 1;                              // n_qblocks + 1
  uint32_t* run_base = run_first + plan.n_qblocks + 1;               // n_qblocks

  const int b = blockIdx.y;
  const int m = blockIdx.x % M;
  const int tile = plan.n_tiles - 1 - blockIdx.x / M;  // coarse (record-heavy) levels first
  const int64_t slab = static_cast<int64_t>(b) * M + m;
  MSDA_STAMP_DECL;
  if (threadIdx.x < L) lv[threadIdx.x] = load_level(shapes, lsi, threadIdx.x, S);
  if (threadIdx.x == 0) n_runs_s = 0;
  if (threadIdx.x < 2 * TS) cnt2[threadIdx.x] = 0;
  __syncthreads();
  if (plan.tile2d && threadIdx.x == 0) {
    build_tile_map(lv, L, plan, tm);
    locate_tile(tm, L, plan, tile, tile_level_s, tile_y0_s, tile_x0_s);
  }
  // the non-empty runs of this tile: (record count, first record index inside the slab)
  for (int qbk = threadIdx.x; qbk < plan.n_qblocks; qbk += 256) {
    const uint2 e = dir[(slab * plan.n_qblocks + qbk) * plan.n_tiles + tile];
    // (a run that does not fit its region can only come from a corrupted directory: skipped)
    if (e.y != 0 && e.x <= plan.region && e.y <= plan.region - e.x) {
      const uint32_t at = atomicAdd(&n_runs_s, 1u);
      run_first[at] = e.y;
      run_base[at] = static_cast<uint32_t>(qbk) * plan.region + e.x;
    }
  }
  __syncthreads();
  const int n_runs = static_cast<int>(n_runs_s);
  const uint32_t total = block_exclusive_scan(run_first, n_runs, scan_tmp);
  if (threadIdx.x == 0) run_first[n_runs] = total;
  __syncthreads();
  MSDA_STAMP(0);  // level table, directory column, run scan

  const uint4* slab_records = records + slab * plan.n_qblocks * static_cast<int64_t>(plan.region);
  // grad_out rows of this head: (N, Lq, M, D) as given, or the per-head copy (N, M, Lq, D)
  const char* go_slab = reinterpret_cast<const char*>(grad_out) +
                        (g_head_major ? slab * Lq : static_cast<int64_t>(b) * Lq * M + m) * ROWB;
  const uint32_t tile_u = static_cast<uint32_t>(tile);
  const int pg = threadIdx.x / PL, pl = threadIdx.x % PL;
  // bytes between the rows of consecutive queries; offsets inside a slab fit 32 bits and both
  // factors 24 bits (tuned_shape_ok)
  const uint32_t q_stride = static_cast<uint32_t>(g_head_major ? 1 : M) * ROWB;

  float sum[NCH][CPL];
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int c = 0; c < CPL; ++c) sum[j][c] = 0.f;

  // The two count arrays alternate between rounds: a round's counts are cleared during the scan
  // of the round before, so a wave that has finished its walk starts on the next round's records
  // (phase A: registers and the other count array only) while slower waves still walk, without a
  // barrier at the end of the round.
  int parity = 0;
  for (uint32_t base = 0; base < total; base += R, parity ^= 1) {
    uint32_t* cnt = cnt2 + parity * TS;
    // ---- A: this thread's records of the round: footprints, count the in-tile corners ----
    uint32_t rq[RPT];
    uint32_t rloc[RPT];  // 4 bytes: position inside the tile per corner, 0xff = not in this tile
    float rw[RPT][4];
    uint4 recs[RPT];
    // every record load is issued before the first is used (index clamped into the tile's list)
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const uint32_t i = min(base + j * 256 + threadIdx.x, total - 1);
      int lo = 0, hi = n_runs;  // run_first[lo] <= i < run_first[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (run_first[mid] <= i) lo = mid; else hi = mid;
      }
      recs[j] = slab_records[run_base[lo] + (i - run_first[lo])];
    }
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const uint32_t i = base + j * 256 + threadIdx.x;
      rloc[j] = 0xffffffffu;
      rq[j] = 0;
      // (a query index outside the tensor can only come from a corrupted record: dropped, so
      // that no row address is ever formed from it)
      if (i < total && (recs[j].x & 0x0fffffffu) < static_cast<uint32_t>(Lq) && !(kAblate & 32)) {
        const uint4 rec = recs[j];
        const int l = static_cast<int>(rec.x >> 28);
        const LevelInfo L_ = lv[l];
        const float a = __uint_as_float(rec.y);
        const Footprint<float> f =
            footprint<float, true>(__uint_as_float(rec.z), __uint_as_float(rec.w), L_.H, L_.W);
        rq[j] = rec.x & 0x0fffffffu;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint32_t t, local;
          pixel_tile(plan, tm, L_, l, f.cy[k >> 1], f.cx[k & 1], t, local);
          const bool in = f.valid[k] && t == tile_u;
          rw[j][k] = a * f.w[k];
          if (in) atomicAdd(&cnt[local], 1u);
          packed |= (in ? local : 0xffu) << (8 * k);
        }
        rloc[j] = packed;
      }
    }
    MSDA_STAMP(1);  // A: record loads, footprints, counts (own share)
    __syncthreads();
    MSDA_STAMP(2);  // barrier behind A
    // ---- B: exclusive scan of the 64 per-position counts by wave 0; cnt becomes the cursor ----
    if (threadIdx.x < TS) {
      const uint32_t c = cnt[threadIdx.x];
      const uint32_t incl = wave_inclusive_scan(c);
      start[threadIdx.x] = incl - c;
      cnt[threadIdx.x] = incl - c;
      if (threadIdx.x == TS - 1) start[TS] = incl;
      cnt2[(parity ^ 1) * TS + threadIdx.x] = 0;  // the next round's counts
    }
    __syncthreads();
    MSDA_STAMP(3);  // B: scan
    // ---- C: fill the buckets ----
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t local = (rloc[j] >> (8 * k)) & 0xffu;
        if (local != 0xffu) {
          const uint32_t at = atomicAdd(&cnt[local], 1u);
          entries[at] = make_uint2(rq[j], __float_as_uint(rw[j][k]));
        }
      }
    }
    MSDA_STAMP(4);  // C: fill (own share)
    __syncthreads();
    MSDA_STAMP(5);  // barrier behind C
    // ---- D: walk the bucket of this lane group's position ----
    if (!(kAblate & 8)) {
      const uint32_t e0 = start[pg], e1 = start[pg + 1];
      float w_cur[U], rv_cur[U][NCH][CPL];
      const auto fetch = [&](uint32_t e, float (&w)[U], float (&rv)[U][NCH][CPL]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint2 ent = entries[min(e + u, e1 - 1)];
          w[u] = e + u < e1 ? __uint_as_float(ent.y) : 0.f;  // weight 0: a repeated row adds nothing
          const char* row = go_slab + (__umul24(ent.x, q_stride) + pl * 16u);
#pragma unroll
          for (int j = 0; j < NCH; ++j) Chunk<VT>::load(row + j * PL * 16, rv[u][j]);
        }
      };
      const auto accumulate = [&](const float (&w)[U], const float (&rv)[U][NCH][CPL]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int c = 0; c < CPL; ++c) sum[j][c] = fmaf(w[u], rv[u][j][c], sum[j][c]);
      };
      for (uint32_t e = e0; e < e1; e += U) {
        fetch(e, w_cur, rv_cur);
        accumulate(w_cur, rv_cur);
      }
    }
    MSDA_STAMP(6);  // D: walk (own share)
    // no barrier here: the next round's phase A touches nothing this round's walk reads, and
    // the barrier behind it orders the next scan and fill after every wave's walk
  }

  MSDA_STAMP_FLUSH(1);
  // store the position of this lane group: lane pl holds chunks pl, pl + 4, ... of the row
  int64_t pos;
  if (plan.tile2d) {
    const int l = tile_level_s;
    if (l < 0) return;
    if (pg >= plan.tw * plan.tw) return;
    const int row = tile_div(pg, plan.tw);
    const int yy = tile_y0_s + row;
    const int xx = tile_x0_s + (pg - row * plan.tw);
    if (yy >= lv[l].H || xx >= lv[l].W) return;
    pos = lv[l].start + static_cast<int64_t>(yy) * lv[l].W + xx;
  } else {
    pos = static_cast<int64_t>(tile) * TS + pg;
  }
  if (pos >= S) return;
  char* dst = reinterpret_cast<char*>(grad_value) +
              (((static_cast<int64_t>(b) * S + pos) * M + m) * D) * sizeof(VT) + pl * 16;
#pragma unroll
  for (int j = 0; j < NCH; ++j) Chunk<VT>::store(dst + j * PL * 16, sum[j]);
}

// ---------------------------------------------------------------------------------------------
// Backward, destination side, by footprint origin: one workgroup per (batch, head, 7 x 7 patch)
//
// The position kernel above fetches a record's grad_out row once per corner that falls into the
// tile (3-4 times per record).  Here a record is fetched ONCE: the tile's records are bucketed by
// the ORIGIN of their footprint (its top-left pixel (y0, x0), y0 = -1 .. H - 1), and the 4 lanes
// that own an origin slot keep four running sums — one per corner of the 2 x 2 footprint — in
// registers.  The footprints that touch a 7 x 7 patch of output positions have their origins in
// the 8 x 8 block of pixels that starts one above and one left of it: 64 slots, one per lane
// group.  Patches of neighbouring workgroups therefore overlap in their origin blocks by one row
// / column (a footprint on a patch edge is walked by both, each keeping the corners inside its
// own patch: 1.31 records per sampled point instead of 1.27), and no sum ever leaves the
// workgroup.  At the end the four corner sums of every slot go through LDS once: output position
// (y, x) of the patch adds corner (dy, dx) of slot (y + 1 - dy, x + 1 - dx).
// Per round: one LDS atomic pair and one 20-byte entry per record (the position kernel: up to
// four of each), a walk of about a third of the rows.  Rows of up to 32 channels (32 accumulator
// registers per lane).
// ---------------------------------------------------------------------------------------------
// What the origin kernel knows about the pyramid, handed over as a kernel argument (the host has
// the level shapes whenever this kernel is chosen): no table load and no tile map to build at
// the start of every workgroup.  Patches of the fine levels, which receive few records each, are
// taken several per workgroup (`per`), so that the latencies of a patch's set-up — its directory
// column, its first records — are paid once per group instead of once per patch.
constexpr int kOriginMaxLevels = 8;
#ifndef MSDA_TO_T
#define MSDA_TO_T 2
#endif
#ifndef MSDA_TO_RPT0
#define MSDA_TO_RPT0 2
#endif
#ifndef MSDA_TO_W4
#define MSDA_TO_W4 1  // entries carry the four corner weights and the row offset (20 bytes)
#endif
#ifndef MSDA_TO_SEQ
#define MSDA_TO_SEQ 1  // a thread takes consecutive records of a round: one run search per thread
#endif
constexpr int kOriginMaxFuse = MSDA_TO_T;      // patches per workgroup, at most
constexpr int kOriginFuseRecords = MSDA_TO_RPT0;  // records per thread and patch of a fused first round
struct OriginLevels {
  int H[kOriginMaxLevels], W[kOriginMaxLevels], start[kOriginMaxLevels];
  int tstart[kOriginMaxLevels + 1];  // first patch id of every level
  int tx[kOriginMaxLevels];          // patches per row
  int gstart[kOriginMaxLevels + 1];  // first workgroup (group of patches) of every level
  int per[kOriginMaxLevels];         // patches per workgroup on the level
};

template <typename VT, int G, int RPT, int U>
__global__ __launch_bounds__(256) void bwd_tile_origin_kernel(
    const VT* __restrict__ grad_out, const uint4* __restrict__ records,
    const uint2* __restrict__ dir, VT* __restrict__ grad_value, int S, int M, int Lq, int L,
    BinPlan plan, int g_head_major, OriginLevels lvls) {
  constexpr int CPL = Chunk<VT>::kCPL;
  constexpr int D = G * CPL;
  constexpr int PL = 4;                     // lanes per origin slot
  constexpr int NS = 64;                    // origin slots: 8 x 8
  constexpr int NT = NS * PL;               // threads
  constexpr int NCH = D / (PL * CPL);       // 16-byte chunks of a row per lane
  constexpr int R = NT * RPT;               // records per round (one patch per workgroup)
  constexpr int T = kOriginMaxFuse;
  constexpr int RPT0 = kOriginFuseRecords;
  constexpr int R0 = NT * RPT0;             // records per round and patch (several per workgroup)
  constexpr int ROWB = D * sizeof(VT);
  constexpr int TW = 7;                     // output patch: TW x TW positions
  static_assert(NCH >= 1 && NCH * PL * CPL == D && NCH * CPL <= 8, "rows of up to 32 channels");
  static_assert(NS * D / 4 <= R0, "the corner exchange fits an entry region");
  static_assert(R <= T * R0, "a round of a single patch fits the entry regions");
  static_assert(T * NS <= NT && T <= NT / 64, "a thread per fused slot counter, a wave per patch");
  extern __shared__ uint4 lds[];
  __shared__ uint32_t n_runs_s[T];

  // LDS: T regions of R0 16-byte entries (a fifth word of each sits in w3s) — region k holds the
  // entries of patch k and then its corner exchange, a workgroup with a single patch uses them all as one round buffer; slot
  // counters / list starts of the fused first rounds and of the later rounds; run tables of the
  // T patches.
  uint4* region = lds;
  // (MSDA_TO_W4) entry T * R0 is the all-zero entry that list ends are padded with; w3s: the
  // fourth corner weight of every entry
  float* w3s = reinterpret_cast<float*>(region + T * R0 + (MSDA_TO_W4 ? 1 : 0));
  uint32_t* cntA = reinterpret_cast<uint32_t*>(w3s + (MSDA_TO_W4 ? T * R0 + 4 : 0));  // T x NS
  uint32_t* startA = cntA + T * NS;                                     // T x (NS + 1)
  uint32_t* cnt2 = startA + T * (NS + 1);                               // 2 x NS
  uint32_t* start = cnt2 + 2 * NS;                                      // NS + 1
  uint32_t* run_tab = start + NS + 1;                                   // per patch: n_qblocks + 1 starts, n_qblocks bases
  const int tab_words = 2 * plan.n_qblocks + 1;

  const int b = blockIdx.y;
  const int m = blockIdx.x % M;
  const int group = lvls.gstart[L] - 1 - static_cast<int>(blockIdx.x / M);  // coarse levels first
  const int64_t slab = static_cast<int64_t>(b) * M + m;
  MSDA_STAMP_DECL;
  if (group < 0) return;
  int tl = 0;
  for (int l = 1; l < L; ++l)
    if (group >= lvls.gstart[l]) tl = l;
  const int H = lvls.H[tl], W = lvls.W[tl], lstart = lvls.start[tl], txl = lvls.tx[tl];
  const int t_first = lvls.tstart[tl] + (group - lvls.gstart[tl]) * lvls.per[tl];
  const int nt = min(min(lvls.per[tl], T), lvls.tstart[tl + 1] - t_first);  // patches of this workgroup
  if (nt < 1) return;

  if (threadIdx.x < T) n_runs_s[threadIdx.x] = 0;
  if (MSDA_TO_W4 && threadIdx.x == 0) {
    region[T * R0] = make_uint4(0u, 0u, 0u, 0u);
    w3s[T * R0] = 0.f;
  }
  if (threadIdx.x < T * NS) cntA[threadIdx.x] = 0;
  if (threadIdx.x < 2 * NS) cnt2[threadIdx.x] = 0;
  __syncthreads();
  // the directory columns of the patches: their non-empty runs (record count, first record)
#pragma unroll
  for (int k = 0; k < T; ++k) {
    if (k < nt) {
      uint32_t* rf = run_tab + k * tab_words;
      uint32_t* rb = rf + plan.n_qblocks + 1;
      for (int qbk = threadIdx.x; qbk < plan.n_qblocks; qbk += NT) {
        const uint2 e = dir[(slab * plan.n_qblocks + qbk) * plan.n_tiles + t_first + k];
        // (a run that does not fit its region can only come from a corrupted directory: skipped)
        if (e.y != 0 && e.x <= plan.region && e.y <= plan.region - e.x) {
          const uint32_t at = atomicAdd(&n_runs_s[k], 1u);
          rf[at] = e.y;
          rb[at] = static_cast<uint32_t>(qbk) * plan.region + e.x;
        }
      }
    }
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < nt) {  // wave k turns the run lengths of patch k into run starts
    uint32_t* rf = run_tab + wave * tab_words;
    const int n = static_cast<int>(n_runs_s[wave]);
    uint32_t carry = 0;
    for (int b0 = 0; b0 < n; b0 += 64) {
      const int i = b0 + lane;
      const uint32_t x = i < n ? rf[i] : 0u;
      const uint32_t incl = wave_inclusive_scan(x);
      if (i < n) rf[i] = carry + incl - x;
      carry += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
    }
    if (lane == 0) rf[n] = carry;
  }
  __syncthreads();
  MSDA_STAMP(0);  // directory columns, run scans

  const uint4* slab_records = records + slab * plan.n_qblocks * static_cast<int64_t>(plan.region);
  const char* go_slab = reinterpret_cast<const char*>(grad_out) +
                        (g_head_major ? slab * Lq : static_cast<int64_t>(b) * Lq * M + m) * ROWB;
  const int pg = threadIdx.x / PL, pl = threadIdx.x % PL;
  const uint32_t q_stride = static_cast<uint32_t>(g_head_major ? 1 : M) * ROWB;
  // the rows of this (batch, head) come in through a buffer descriptor: one 32-bit offset per
  // load instead of a 64-bit address, and an offset past the rows reads zeros
  const uint32_t go_bytes = static_cast<uint32_t>(Lq - 1) * q_stride + ROWB;
  const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(go_slab), 0, static_cast<int>(go_bytes), 0x00020000);
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

  float acc[4][NCH][CPL];  // [corner (y0,x0) (y0,x0+1) (y0+1,x0) (y0+1,x0+1)]
  const auto clear_acc = [&]() {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
      for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c4][j][c] = 0.f;
  };
  // record i of a patch's list (runs `rf` / `rb`, n of them)
  const auto load_record = [&](const uint32_t* rf, const uint32_t* rb, int n, uint32_t i) {
    int lo = 0, hi = n;  // rf[lo] <= i < rf[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (rf[mid] <= i) lo = mid; else hi = mid;
    }
    return slab_records[rb[lo] + (i - rf[lo])];
  };
  // records i0, i0 + 1, ... of a patch's list (clamped to the last one): one search, then the run
  // table is followed
  const auto load_records_seq = [&](const uint32_t* rf, const uint32_t* rb, int n, uint32_t i0,
                                    uint32_t total, auto count_c, uint4* out) {
    constexpr int CNT = decltype(count_c)::value;
    const uint32_t first = min(i0, total - 1);
    int lo = 0, hi = n;  // rf[lo] <= first < rf[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (rf[mid] <= first) lo = mid; else hi = mid;
    }
    uint32_t cur = rf[lo], nxt = rf[lo + 1], rbase = rb[lo];
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
      const uint32_t i = min(i0 + j, total - 1);
      while (i >= nxt) {  // (rf[n] = total > i)
        ++lo;
        cur = nxt;
        nxt = rf[lo + 1];
        rbase = rb[lo];
      }
      out[j] = slab_records[rbase + (i - cur)];
    }
  };
  // origin slot of a record in the patch whose first output position is (ty0, tx0); NS when the
  // record does not belong here (a query index outside the tensor or another level's record can
  // only come from a corrupted workspace: dropped before any address is formed from it)
  const auto slot_of = [&](const uint4& rec, int ty0, int tx0, uint4& ent, float& w3) -> uint32_t {
    const uint32_t q = rec.x & 0x0fffffffu;
    if (q >= static_cast<uint32_t>(Lq) || static_cast<int>(rec.x >> 28) != tl) return NS;
    // origin + 1 (0 .. H, 0 .. W): row / column 0 holds the footprints that hang over the edge
    int oy, ox;
    float ly, lx;
    const bool any = footprint_origin(__uint_as_float(rec.z), __uint_as_float(rec.w), H, W, oy, ox, ly, lx);
    const int sy = oy - ty0, sx = ox - tx0;
    if (!any || sy < 0 || sy > TW || sx < 0 || sx > TW) return NS;
    if (MSDA_TO_W4) {
      const float a = __uint_as_float(rec.y);
      const float hx = 1.f - lx, hy = 1.f - ly;
      ent = make_uint4(__umul24(q, q_stride), __float_as_uint(a * (hy * hx)),
                       __float_as_uint(a * (hy * lx)), __float_as_uint(a * (ly * hx)));
      w3 = a * (ly * lx);
    } else {
      ent = make_uint4(q, rec.y, __float_as_uint(lx), __float_as_uint(ly));
    }
    return static_cast<uint32_t>(sy * (TW + 1) + sx);
  };
  // the lane group's slot list [e0, e1) of `entries`: one grad_out row per entry, four corner sums
  // (`zero`: index of the all-zero entry relative to `entries`)
  const auto walk = [&](const uint4* entries, const float* w3e, uint32_t zero, uint32_t e0, uint32_t e1) {
    for (uint32_t e = e0; e < e1; e += U) {
      float w[U][4], rv[U][NCH][CPL];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t row;
        if (MSDA_TO_W4) {
          const uint32_t idx = (u == 0 || e + u < e1) ? e + u : zero;  // past the end: weights 0
          const uint4 en = entries[idx];
          w[u][0] = __uint_as_float(en.y);
          w[u][1] = __uint_as_float(en.z);
          w[u][2] = __uint_as_float(en.w);
          w[u][3] = w3e[idx];
          row = en.x + pl * 16u;
        } else {
          const uint4 en = entries[min(e + u, e1 - 1)];
          const float a = e + u < e1 ? __uint_as_float(en.y) : 0.f;  // a repeated row adds nothing
          const float lx = __uint_as_float(en.z), ly = __uint_as_float(en.w);
          const float hx = 1.f - lx, hy = 1.f - ly;
          w[u][0] = a * (hy * hx);
          w[u][1] = a * (hy * lx);
          w[u][2] = a * (ly * hx);
          w[u][3] = a * (ly * lx);
          row = __umul24(en.x, q_stride) + pl * 16u;
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j)
          Chunk<VT>::unpack(
              __builtin_bit_cast(uint4, static_cast<u32x4>(__builtin_amdgcn_raw_buffer_load_b128(
                                            gres, static_cast<int>(row + j * PL * 16), 0, 0))),
              rv[u][j]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
          for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c4][j][c] = fmaf(w[u][c4], rv[u][j][c], acc[c4][j][c]);
    }
  };
  // Rounds of RR * NT records of one patch, from record `base` on: count per slot, scan, fill,
  // walk.  The two count arrays alternate between rounds and the next round's counts are
  // cleared during the scan, so a wave that has finished its walk starts on the next round's
  // records while slower waves still walk (no barrier at the end of a round).  cnt2 is all zero
  // on entry; on return one of its halves is not.
  const auto rounds = [&](auto rr_c, uint32_t base, uint32_t total, uint4* entries, float* w3e,
                          uint32_t zero, const uint32_t* rf, const uint32_t* rb, int n_runs,
                          int ty0, int tx0) {
    constexpr int RR = decltype(rr_c)::value;
    int parity = 0;
    for (; base < total; base += RR * NT, parity ^= 1) {
      uint32_t* cnt = cnt2 + parity * NS;
      uint4 ent[RR];       // {query, weight, lx, ly} / {row offset, three corner weights}
      float w3[RR];        // (MSDA_TO_W4) the fourth corner weight
      uint32_t slot[RR];   // NS = not in this patch
      uint4 recs[RR];
      if (MSDA_TO_SEQ) {
        load_records_seq(rf, rb, n_runs, base + threadIdx.x * RR, total, rr_c, recs);
      } else {
#pragma unroll
        for (int j = 0; j < RR; ++j)
          recs[j] = load_record(rf, rb, n_runs, min(base + j * NT + threadIdx.x, total - 1));
      }
#pragma unroll
      for (int j = 0; j < RR; ++j) {
        const uint32_t i = MSDA_TO_SEQ ? base + threadIdx.x * RR + j : base + j * NT + threadIdx.x;
        slot[j] = i < total ? slot_of(recs[j], ty0, tx0, ent[j], w3[j]) : NS;
        if (slot[j] < NS) atomicAdd(&cnt[slot[j]], 1u);
      }
      MSDA_STAMP(1);
      __syncthreads();
      MSDA_STAMP(2);
      if (threadIdx.x < NS) {  // exclusive scan of the slot counts by wave 0; cnt becomes the cursor
        const uint32_t c = cnt[threadIdx.x];
        const uint32_t incl = wave_inclusive_scan(c);
        start[threadIdx.x] = incl - c;
        cnt[threadIdx.x] = incl - c;
        if (threadIdx.x == NS - 1) start[NS] = incl;
        cnt2[(parity ^ 1) * NS + threadIdx.x] = 0;  // the next round's counts
      }
      __syncthreads();
      MSDA_STAMP(3);
#pragma unroll
      for (int j = 0; j < RR; ++j)
        if (slot[j] < NS) {
          const uint32_t at = atomicAdd(&cnt[slot[j]], 1u);
          entries[at] = ent[j];
          if (MSDA_TO_W4) w3e[at] = w3[j];
        }
      MSDA_STAMP(4);
      __syncthreads();
      MSDA_STAMP(5);
      walk(entries, w3e, zero, start[pg], start[pg + 1]);
      MSDA_STAMP(6);
    }
  };
  // The four corner sums of every slot meet at their output positions: lane group pg also owns
  // output position (pg / 8, pg % 8) of the patch (when both are < 7), which adds corner (dy, dx)
  // of slot (y + 1 - dy, x + 1 - dx).  `comb`: NS rows of D floats.  Starts with a barrier.
  const auto combine_store = [&](float4* comb, int ty0, int tx0) {
    constexpr int F4 = D / 4;            // float4 per row
    constexpr int F4L = NCH * CPL / 4;   // float4 per lane and row: the lane's channels, in order
    const int py = pg >> 3, px = pg & 7;
    float out[NCH][CPL];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int c = 0; c < CPL; ++c) out[j][c] = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      __syncthreads();  // the last walk (c4 = 0) / the reads of the corner before are done
#pragma unroll
      for (int k = 0; k < F4L; ++k) {
        const int j = (k * 4) / CPL, c = (k * 4) % CPL;
        comb[pg * F4 + pl * F4L + k] =
            make_float4(acc[c4][j][c], acc[c4][j][c + 1], acc[c4][j][c + 2], acc[c4][j][c + 3]);
      }
      __syncthreads();
      const int sy = py + 1 - (c4 >> 1), sx = px + 1 - (c4 & 1);  // 0 .. 8; 8 only for unused groups
      if (py < TW && px < TW) {
#pragma unroll
        for (int k = 0; k < F4L; ++k) {
          const float4 v = comb[(sy * (TW + 1) + sx) * F4 + pl * F4L + k];
          const int j = (k * 4) / CPL, c = (k * 4) % CPL;
          out[j][c] += v.x;
          out[j][c + 1] += v.y;
          out[j][c + 2] += v.z;
          out[j][c + 3] += v.w;
        }
      }
    }
    const int yy = ty0 + py, xx = tx0 + px;
    if (py >= TW || px >= TW || yy >= H || xx >= W) return;
    const int64_t pos = lstart + static_cast<int64_t>(yy) * W + xx;
    if (pos >= S) return;
    // the lane's channels are chunks pl, pl + 4, ... of the row, as everywhere in these kernels
    char* dst = reinterpret_cast<char*>(grad_value) +
                (((static_cast<int64_t>(b) * S + pos) * M + m) * D) * sizeof(VT) + pl * 16;
#pragma unroll
    for (int j = 0; j < NCH; ++j) Chunk<VT>::store(dst + j * PL * 16, out[j]);
  };
  const auto patch_origin = [&](int k, int& ty0, int& tx0) {
    const int r = t_first + k - lvls.tstart[tl];
    const int row = r / txl;
    ty0 = row * TW;
    tx0 = (r - row * txl) * TW;
  };

  if (nt == 1) {
    // one (record-heavy) patch: rounds of R records through all the regions
    int ty0, tx0;
    patch_origin(0, ty0, tx0);
    const uint32_t* rf = run_tab;
    const int n_runs = static_cast<int>(n_runs_s[0]);
    clear_acc();
    rounds(std::integral_constant<int, RPT>{}, 0u, rf[n_runs], region, w3s,
           static_cast<uint32_t>(T * R0), rf, rf + plan.n_qblocks + 1, n_runs, ty0, tx0);
    combine_store(reinterpret_cast<float4*>(region), ty0, tx0);
    MSDA_STAMP(7);
    MSDA_STAMP_FLUSH(1);
    return;
  }

  // ---- several patches: the first R0 records of each are taken together ----
  {
    uint4 recs[T][RPT0], ent[T][RPT0];
    float w3[T][RPT0];
    uint32_t slot[T][RPT0];
#pragma unroll
    for (int k = 0; k < T; ++k) {
      const uint32_t* rf = run_tab + k * tab_words;
      const int n = k < nt ? static_cast<int>(n_runs_s[k]) : 0;
      const uint32_t total = n > 0 ? rf[n] : 0u;
#pragma unroll
      for (int j = 0; j < RPT0; ++j) slot[k][j] = total;  // (kept until the loads have been issued)
      if (MSDA_TO_SEQ) {
        if (total > 0)
          load_records_seq(rf, rf + plan.n_qblocks + 1, n, threadIdx.x * RPT0, total,
                           std::integral_constant<int, RPT0>{}, recs[k]);
      } else {
#pragma unroll
        for (int j = 0; j < RPT0; ++j)
          if (total > 0)
            recs[k][j] = load_record(rf, rf + plan.n_qblocks + 1, n, min(j * NT + threadIdx.x, total - 1));
      }
    }
#pragma unroll
    for (int k = 0; k < T; ++k) {
      int ty0, tx0;
      patch_origin(k, ty0, tx0);
#pragma unroll
      for (int j = 0; j < RPT0; ++j) {
        const uint32_t total = slot[k][j];
        slot[k][j] = NS;
        if ((MSDA_TO_SEQ ? threadIdx.x * RPT0 + j : j * NT + threadIdx.x) < total) {
          slot[k][j] = slot_of(recs[k][j], ty0, tx0, ent[k][j], w3[k][j]);
          if (slot[k][j] < NS) atomicAdd(&cntA[k * NS + slot[k][j]], 1u);
        }
      }
    }
    __syncthreads();
    if (wave < nt) {  // wave k: exclusive scan of patch k's slot counts; cntA becomes the cursor
      const uint32_t c = cntA[wave * NS + lane];
      const uint32_t incl = wave_inclusive_scan(c);
      startA[wave * (NS + 1) + lane] = incl - c;
      cntA[wave * NS + lane] = incl - c;
      if (lane == NS - 1) startA[wave * (NS + 1) + NS] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < T; ++k)
#pragma unroll
      for (int j = 0; j < RPT0; ++j)
        if (slot[k][j] < NS) {
          const uint32_t at = k * R0 + atomicAdd(&cntA[k * NS + slot[k][j]], 1u);
          region[at] = ent[k][j];
          if (MSDA_TO_W4) w3s[at] = w3[k][j];
        }
    __syncthreads();
  }
  MSDA_STAMP(1);
  for (int k = 0; k < nt; ++k) {
    int ty0, tx0;
    patch_origin(k, ty0, tx0);
    const uint32_t* rf = run_tab + k * tab_words;
    const int n_runs = static_cast<int>(n_runs_s[k]);
    const uint32_t total = n_runs > 0 ? rf[n_runs] : 0u;
    clear_acc();
    walk(region + k * R0, w3s + k * R0, static_cast<uint32_t>((T - k) * R0),
         startA[k * (NS + 1) + pg], startA[k * (NS + 1) + pg + 1]);
    if (total > R0) {  // (uniform) the rest of a patch with more records, R0 per round, in its own region
      __syncthreads();  // every wave has walked the first round: the region is free
      rounds(std::integral_constant<int, RPT0>{}, static_cast<uint32_t>(R0), total, region + k * R0,
             w3s + k * R0, static_cast<uint32_t>((T - k) * R0), rf, rf + plan.n_qblocks + 1, n_runs,
             ty0, tx0);
      __syncthreads();  // all walks done before the counters are cleared for the next user
      if (threadIdx.x < 2 * NS) cnt2[threadIdx.x] = 0;
    }
    MSDA_STAMP(6);
    combine_store(reinterpret_cast<float4*>(region + k * R0), ty0, tx0);
    MSDA_STAMP(7);
  }
  MSDA_STAMP_FLUSH(1);
}

// ---------------------------------------------------------------------------------------------
// Backward, destination side, reproducible form (MSDA_FLAG_DETERMINISTIC on the binned path).
//
// The fast tile kernel adds a position's contributions in whatever order LDS atomics ranked them:
// the order of the records inside a (query block, tile) run, the order of the runs, and the
// order of a position's entries inside a round all change from launch to launch, and with them
// the last bits of the fp32 sums.  Here every one of those orders is fixed by the data:
//   * the runs of the tile are taken in query-block order (empty ones included);
//   * a round is a set of WHOLE runs (as many consecutive ones as fit R records), so which
//     records share a round does not depend on the order inside a run; a run longer than R is
//     taken in passes over sub-ranges of its query block (R / (L*P) queries each: a query sends at
//     most L*P records to one tile), each pass scanning the whole run;
//   * inside a round, every position's entry list is sorted by (query, weight bits) before it is
//     walked — entries that compare equal contribute identical products, so their order is free.
// Two launches on the same inputs therefore give bit-equal grad_value.  About twice the fast
// kernel's time (records are read and their footprints computed twice, plus the sort).
// ---------------------------------------------------------------------------------------------
template <typename VT, int G>
__global__ __launch_bounds__(256) void bwd_tile_gather_det_kernel(
    const VT* __restrict__ grad_out, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const uint4* __restrict__ records,
    const uint2* __restrict__ dir, VT* __restrict__ grad_value, int S, int M, int Lq, int L,
    int LP, BinPlan plan, int g_head_major) {
  constexpr int CPL = Chunk<VT>::kCPL;
  constexpr int D = G * CPL;
  constexpr int PL = 4;
  constexpr int TS = 256 / PL;
  constexpr int NCH = D / (PL * CPL);
  constexpr int R = 512;  // records per round (4 * R entries)
  constexpr int U = 3;
  constexpr int ROWB = D * sizeof(VT);
  static_assert(TS == 64, "the bucket scan is one wave wide");
  extern __shared__ uint4 lds[];
  __shared__ LevelInfo lv[kMaxLevelsTuned];
  __shared__ TileMap tm;
  __shared__ uint32_t scan_tmp[4];
  __shared__ int tile_level_s, tile_y0_s, tile_x0_s;

  uint2* entries = reinterpret_cast<uint2*>(lds);                  // 4 * R, in fill order
  uint2* sorted = entries + 4 * R;                                 // 4 * R, in walk order
  uint32_t* cnt = reinterpret_cast<uint32_t*>(sorted + 4 * R);     // TS: counts, then cursors
  uint32_t* start = cnt + TS;                                      // TS + 1
  uint32_t* run_first = start + TS + 1;                            // n_qblocks + 1
  uint32_t* run_base = run_first + plan.n_qblocks + 1;             // n_qblocks

  const int b = blockIdx.y;
  const int m = blockIdx.x % M;
  const int tile = plan.n_tiles - 1 - blockIdx.x / M;
  const int64_t slab = static_cast<int64_t>(b) * M + m;
  if (threadIdx.x < L) lv[threadIdx.x] = load_level(shapes, lsi, threadIdx.x, S);
  __syncthreads();
  if (plan.tile2d && threadIdx.x == 0) {
    build_tile_map(lv, L, plan, tm);
    locate_tile(tm, L, plan, tile, tile_level_s, tile_y0_s, tile_x0_s);
  }
  // every query block's run of this tile, in query-block order (length 0 when it has none)
  const int n_runs = plan.n_qblocks;
  for (int qbk = threadIdx.x; qbk < n_runs; qbk += 256) {
    const uint2 e = dir[(slab * plan.n_qblocks + qbk) * plan.n_tiles + tile];
    const bool ok = e.y != 0 && e.x <= plan.region && e.y <= plan.region - e.x;
    run_first[qbk] = ok ? e.y : 0u;
    run_base[qbk] = static_cast<uint32_t>(qbk) * plan.region + (ok ? e.x : 0u);
  }
  const uint32_t total = block_exclusive_scan(run_first, n_runs, scan_tmp);
  if (threadIdx.x == 0) run_first[n_runs] = total;
  __syncthreads();

  const uint4* slab_records = records + slab * plan.n_qblocks * static_cast<int64_t>(plan.region);
  const char* go_slab = reinterpret_cast<const char*>(grad_out) +
                        (g_head_major ? slab * Lq : static_cast<int64_t>(b) * Lq * M + m) * ROWB;
  const uint32_t tile_u = static_cast<uint32_t>(tile);
  const int pg = threadIdx.x / PL, pl = threadIdx.x % PL;
  const uint32_t q_stride = static_cast<uint32_t>(g_head_major ? 1 : M) * ROWB;
  // largest r in [0, n_runs] with run_first[r] <= x
  const auto last_start_le = [&](uint32_t x) {
    int lo = 0, hi = n_runs + 1;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (run_first[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
  };

  float sum[NCH][CPL];
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int c = 0; c < CPL; ++c) sum[j][c] = 0.f;

  const uint32_t q_sub = static_cast<uint32_t>(max(1, R / max(LP, 1)));  // queries per pass of a long run
  uint32_t base = 0, sub = 0;
  while (base < total) {  // (uniform)
    // ---- the round: records [lo, hi) of the tile's list, queries [q_lo, q_hi) ----
    uint32_t lo = base, hi, q_lo = 0u, q_hi = 0xffffffffu, next_base, next_sub = 0u;
    const int r_end = last_start_le(base + R);  // run_first[r_end] <= base + R
    if (run_first[r_end] > base) {
      hi = run_first[r_end];  // whole runs
      next_base = hi;
    } else {
      const int r0 = last_start_le(base);  // the (non-empty) run that starts at base: longer than R
      hi = run_first[r0 + 1];
      q_lo = static_cast<uint32_t>(r0) * plan.q_per_block + sub * q_sub;
      q_hi = q_lo + q_sub;
      const bool last = (sub + 1) * q_sub >= static_cast<uint32_t>(plan.q_per_block);
      next_base = last ? hi : base;
      next_sub = last ? 0u : sub + 1;
    }
    if (threadIdx.x < TS) cnt[threadIdx.x] = 0;
    __syncthreads();
    // the records are visited twice with the same arithmetic: to count, then to fill
    const auto visit = [&](auto&& corner) {
      for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
        const int r = last_start_le(i);
        const uint4 rec = slab_records[run_base[r] + (i - run_first[r])];
        const uint32_t q = rec.x & 0x0fffffffu;
        if (q >= static_cast<uint32_t>(Lq) || q < q_lo || q >= q_hi) continue;
        const int l = static_cast<int>(rec.x >> 28);
        const LevelInfo L_ = lv[l];
        const float a = __uint_as_float(rec.y);
        const Footprint<float> f =
            footprint<float, true>(__uint_as_float(rec.z), __uint_as_float(rec.w), L_.H, L_.W);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint32_t t, local;
          pixel_tile(plan, tm, L_, l, f.cy[k >> 1], f.cx[k & 1], t, local);
          if (f.valid[k] && t == tile_u) corner(local, q, a * f.w[k]);
        }
      }
    };
    visit([&](uint32_t local, uint32_t, float) { atomicAdd(&cnt[local], 1u); });
    __syncthreads();
    if (threadIdx.x < TS) {
      const uint32_t c = cnt[threadIdx.x];
      const uint32_t incl = wave_inclusive_scan(c);
      start[threadIdx.x] = incl - c;
      cnt[threadIdx.x] = incl - c;
      if (threadIdx.x == TS - 1) start[TS] = incl;
    }
    __syncthreads();
    visit([&](uint32_t local, uint32_t q, float w) {
      const uint32_t at = atomicAdd(&cnt[local], 1u);
      if (at < 4u * R) entries[at] = make_uint2(q, __float_as_uint(w));
    });
    __syncthreads();
    // ---- sort the list of this lane group's position by (query, weight bits): rank sort ----
    const uint32_t e0 = min(start[pg], 4u * R), e1 = min(start[pg + 1], 4u * R);
    for (uint32_t i = e0 + pl; i < e1; i += PL) {
      const uint2 me = entries[i];
      uint32_t rank = 0;
      for (uint32_t j = e0; j < e1; ++j) {
        const uint2 o = entries[j];
        const bool before = o.x < me.x || (o.x == me.x && (o.y < me.y || (o.y == me.y && j < i)));
        rank += before ? 1u : 0u;
      }
      sorted[e0 + rank] = me;
    }
    __syncthreads();
    // ---- walk, in list order ----
    for (uint32_t e = e0; e < e1; e += U) {
      float w[U], rv[U][NCH][CPL];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint2 ent = sorted[min(e + u, e1 - 1)];
        w[u] = e + u < e1 ? __uint_as_float(ent.y) : 0.f;
        const char* row = go_slab + (__umul24(ent.x, q_stride) + pl * 16u);
#pragma unroll
        for (int j = 0; j < NCH; ++j) Chunk<VT>::load(row + j * PL * 16, rv[u][j]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
          for (int c = 0; c < CPL; ++c) sum[j][c] = fmaf(w[u], rv[u][j][c], sum[j][c]);
    }
    __syncthreads();  // the next round reuses cnt, entries and sorted
    base = next_base;
    sub = next_sub;
  }

  int64_t pos;
  if (plan.tile2d) {
    const int l = tile_level_s;
    if (l < 0) return;
    if (pg >= plan.tw * plan.tw) return;
    const int row = tile_div(pg, plan.tw);
    const int yy = tile_y0_s + row;
    const int xx = tile_x0_s + (pg - row * plan.tw);
    if (yy >= lv[l].H || xx >= lv[l].W) return;
    pos = lv[l].start + static_cast<int64_t>(yy) * lv[l].W + xx;
  } else {
    pos = static_cast<int64_t>(tile) * TS + pg;
  }
  if (pos >= S) return;
  char* dst = reinterpret_cast<char*>(grad_value) +
              (((static_cast<int64_t>(b) * S + pos) * M + m) * D) * sizeof(VT) + pl * 16;
#pragma unroll
  for (int j = 0; j < NCH; ++j) Chunk<VT>::store(dst + j * PL * 16, sum[j]);
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
  if (L > kMaxLevelsTuned) return false;
  const int esize = cpl == 4 ? 4 : 2;
  // (512 bytes short of 4 GB: see kOffsetOutside)
  if (static_cast<int64_t>(S) * M * D * esize > static_cast<int64_t>(UINT32_MAX) - 512) return false;
  // the kernels form pixel, row and query offsets with 24-bit multiplies and 32-bit sums
  if (S >= (1 << 24) || Lq >= (1 << 24) || static_cast<int64_t>(M) * D * esize >= (1 << 24) ||
      static_cast<int64_t>(M) * 3 * L * P >= (1 << 24) ||
      static_cast<int64_t>(Lq) * M * D * esize > static_cast<int64_t>(UINT32_MAX))
    return false;
  if (N > 65535) return false;
  // rows of a query block (up to 192 queries x M heads) and their element offsets are formed
  // with 24-bit multiplies and 32-bit sums as well
  if (M > 65535) return false;
  const int GPB = 256 / (2 * G);  // fewest queries per workgroup of any kernel
  if ((static_cast<int64_t>(Lq) + GPB - 1) / GPB * M > INT32_MAX) return false;
  *G_out = G;
  return true;
}

// The packed location/weight form needs rows of an even number of elements that start on a pair
// boundary (8 bytes for fp32, 4 for 16-bit types).
inline bool pair_aligned(const void* p, int dtype_loc) {
  return reinterpret_cast<uintptr_t>(p) % (dtype_loc == 0 ? 8 : 4) == 0;
}
inline bool packed_shape_ok(int L, int P) { return (L * P) % 2 == 0; }

// Points staged per pass: the largest multiple of G (lanes per query) whose LDS lists (`slots`
// 16-byte slots per point, 256/G groups) stay within ~26 KB, or all L*P points when they fit.
inline int stage_points_per_pass(int G, int LP, int slots) {
  const int GPB = 256 / G;
  int lpb = G;
  while (lpb + G <= 4096 &&
         static_cast<int64_t>(GPB) * group_stride_slots(lpb + G, slots) * 16 <= 26 * 1024)
    lpb += G;
  return LP <= lpb ? LP : lpb;
}

inline bool all_aligned16(std::initializer_list<const void*> ps) {
  for (const void* p : ps)
    if (reinterpret_cast<uintptr_t>(p) % 16) return false;
  return true;
}

// Binned grad_value plan; returns false when the shape is outside what the binned path handles
// (very long sequences), in which case the fp32-atomic query kernel is used.  `levels_host`
// (L pairs (H, W), host memory, may be null) selects the 2-D tile geometry when the level table
// is a plain pyramid (positive extents that sum to S) and its tile count is within the limits;
// otherwise tiles are runs of the flattened sequence.
// Queries per query-side workgroup: 192 (three passes of 64 at 4 lanes per query) measured 2 %
// faster than 256 on the Cityscapes shapes (14336 workgroups fill the chip's 1024 resident ones
// 14 times exactly; 128 and 320 sit between, 64 and 512 are slower than 256).  The
// register-resident form of the query kernel parks the points of all but the first pass in LDS;
// when that leaves room for only two workgroups per CU (16 points per query: 61 KB) and 128
// queries per workgroup make it three, 128 it is (COCO 4 levels x 4 points, bf16: backward
// 2.44 -> 2.27 ms; with 12 points per query both sizes run three workgroups and 128 is 2 %
// slower).  G = 16-byte chunks per row, 0 when not known.
// Rows of 17-32 points (an even number) run the register-resident query kernel as two virtual
// queries of half a row each.
inline int row_split(int LP) { return MSDA_K3_PRE && LP > 16 && LP <= 32 && LP % 2 == 0 ? 2 : 1; }

// LP: points per virtual query
#ifndef MSDA_K3_QB
#define MSDA_K3_QB 0  // experiment builds: force 128 or 192 queries per workgroup (0: choose)
#endif
inline int choose_qb(int G, int LP, int64_t n_tiles) {
  if (!MSDA_K3_PRE || G == 0 || LP > 16) return 192;
  if (MSDA_K3_QB) return MSDA_K3_QB;
  const int GQ = query_lanes(G), GPB = 256 / GQ;
  const int kpc = GQ == 4 ? (LP <= 12 ? 3 : 4) : 2;
  const auto resident = [&](int qb) {
    const int64_t lds = static_cast<int64_t>(GPB) * group_stride_slots(LP, stage_slots(kStageBwdBinned)) * 16 +
                        (n_tiles + 4) / 4 * 16 + static_cast<int64_t>(qb / GPB - 1) * kpc * 3 * 1024;
    return (160 * 1024) / (lds + 2560);  // (+ static arrays: 1.9 KB, and allocation granularity)
  };
  return resident(192) < 3 && resident(128) > resident(192) ? 128 : 192;
}

// Tile kernel by footprint origin (bwd_tile_origin_kernel, 7 x 7 patches) instead of by
// destination position (8 x 8): -DMSDA_TILE_ORIGIN=0 builds the library without it, for A/B runs.
#ifndef MSDA_TILE_ORIGIN
#define MSDA_TILE_ORIGIN 1
#endif
// records per thread and round / entries per walk step of the origin kernel
#ifndef MSDA_TO_RPT
#define MSDA_TO_RPT 3
#endif
#ifndef MSDA_TO_U
#define MSDA_TO_U 2
#endif
#ifndef MSDA_TO_U16
#define MSDA_TO_U16 3  // entries per walk step, 16-bit rows
#endif
#ifndef MSDA_TO_RPT16
#define MSDA_TO_RPT16 MSDA_TO_RPT
#endif

// Patch order of the queries (order_query): only self-attention over the pyramid qualifies, and
// the rows of a batch element must be addressable from its first row with 24-bit multiplies
// and 32-bit element offsets (rows of at most 3 * L * P elements).
#ifndef MSDA_PATCH_ORDER
#define MSDA_PATCH_ORDER 1
#endif
inline bool patch_order_ok(int S, int M, int Lq, int L, int P) {
  const int64_t rows = static_cast<int64_t>(Lq) * M;
  return MSDA_PATCH_ORDER && Lq == S && rows < (1 << 24) &&
         rows * 3 * L * P <= static_cast<int64_t>(UINT32_MAX);
}

// `det`: the plan of the reproducible backward (MSDA_FLAG_DETERMINISTIC), whose tile kernel works
// by destination position; the geometry (and with it the workspace size) depends on it.
inline bool make_bin_plan(int N, int S, int M, int D, int Lq, int L, int P, int G,
                          const int64_t* levels_host, BinPlan* plan, size_t* dir_bytes,
                          size_t* rec_bytes, bool det = false) {
  if (D > 64) return false;                  // accumulators of a position: D / 4 registers per lane
  const int ts_shift = kTileShift;
  const int TS = 1 << ts_shift;
  const int64_t slabs = static_cast<int64_t>(N) * M;
  int64_t n_tiles = (static_cast<int64_t>(S) + TS - 1) / TS;
  // tile counts of the 2-D geometries (7 x 7 and 8 x 8 patches) when the level table allows them
  int64_t nt2d[2] = {-1, -1};
  if (levels_host != nullptr) {
    int64_t total = 0, nt[2] = {0, 0};
    bool ok = true, small = true;
    for (int l = 0; l < L; ++l) {
      const int64_t h = levels_host[2 * l], w = levels_host[2 * l + 1];
      if (h < 1 || w < 1 || h > S || w > S) { ok = false; break; }
      total += h * w;
      nt[0] += ((h + 6) / 7) * ((w + 6) / 7);
      nt[1] += ((h + 7) / 8) * ((w + 7) / 8);
      small = small && h <= kMaxExtentTile7 && w <= kMaxExtentTile7;
      if (total > S) { ok = false; break; }
    }
    if (ok && total == S) {
      nt2d[1] = nt[1];
      // the origin kernel keeps four corner sums of a row in registers: rows of up to 32 channels
      if (MSDA_TILE_ORIGIN && !det && small && D <= 32 && L <= kOriginMaxLevels) nt2d[0] = nt[0];
    }
  }
  // virtual queries per workgroup (split rows: always 128, one kernel variant fewer) and the
  // real queries they stand for
  const int split = G == 0 ? 1 : row_split(L * P);
  const int64_t nt_guess = nt2d[0] >= 0 ? nt2d[0] : nt2d[1] >= 0 ? nt2d[1] : n_tiles;
  const int qb = split > 1 ? 128 : choose_qb(G, L * P, nt_guess);
  const int qpb = qb / split;
  const int64_t n_qblocks = (static_cast<int64_t>(Lq) + qpb - 1) / qpb;
  const auto fits = [&](int64_t nt) {
    return nt <= 4096 && slabs * n_qblocks * nt * 8 <= (static_cast<int64_t>(1) << 30) &&
           nt * M <= INT32_MAX;
  };
  int tile2d = 0, tw = 0;
  if (nt2d[0] >= 0 && fits(nt2d[0])) {
    n_tiles = nt2d[0];
    tile2d = 1;
    tw = 7;
  } else if (nt2d[1] >= 0 && fits(nt2d[1])) {
    n_tiles = nt2d[1];
    tile2d = 1;
    tw = 8;
  }
  if (!fits(n_tiles) || n_qblocks > 2048 || Lq >= (1 << 28)) return false;
  const int64_t dir_b = slabs * n_qblocks * n_tiles * 8;
  // a point lands in at most 4 tiles: regions can never overflow
  const int64_t region = static_cast<int64_t>(qpb) * L * P * 4;
  const int64_t rec_b = slabs * n_qblocks * region * 16;
  if (n_qblocks * M > INT32_MAX) return false;
  plan->ts_shift = ts_shift;
  plan->n_tiles = static_cast<int>(n_tiles);
  plan->qb = qb;
  plan->n_qblocks = static_cast<int>(n_qblocks);
  plan->region = static_cast<uint32_t>(region);
  plan->tile2d = tile2d;
  plan->tw = tw;
  plan->in_order = 1;
  plan->phase = kPhaseBin | kPhaseGrad;
  plan->q_per_block = qpb;
  plan->split = split;
  // (the reproducible tile kernel takes long runs in passes by query range: consecutive queries)
  plan->patch_order = (!det && patch_order_ok(S, M, Lq, L, P)) ? 1 : 0;
  *dir_bytes = (static_cast<size_t>(dir_b) + 255) & ~static_cast<size_t>(255);
  *rec_bytes = static_cast<size_t>(rec_b);
  return true;
}

// True when both the forward and the backward of this shape run on kernels that accept a
// head-major (N,M,S,D) value tensor.
inline bool head_major_supported(int N, int S, int M, int D, int Lq, int L, int P, int dtype_val,
                                 int dtype_loc) {
  int G;
  BinPlan plan;
  size_t a, b;
  return tuned_shape_ok(N, S, M, D, Lq, L, P, dtype_val, dtype_loc, &G) &&
         make_bin_plan(N, S, M, D, Lq, L, P, G, nullptr, &plan, &a, &b);
}

// Whether forward and backward take raw points (offsets + logits + reference points): tuned
// kernels, the packed row form, the binned backward in its register-resident form.
inline bool raw_points_supported(int N, int S, int M, int D, int Lq, int L, int P, int dtype_val,
                                 int dtype_loc) {
  int G;
  BinPlan plan;
  size_t a, b;
  return MSDA_K3_PRE && MSDA_K3_DEPTH == 2 &&
         tuned_shape_ok(N, S, M, D, Lq, L, P, dtype_val, dtype_loc, &G) && packed_shape_ok(L, P) &&
         make_bin_plan(N, S, M, D, Lq, L, P, G, nullptr, &plan, &a, &b) &&
         L * P / plan.split <= 16 && (plan.qb == 192 || plan.qb == 128);
}

// (N,S,M,D) <-> (N,M,S,D): rows of D*e bytes move whole, 16 bytes per lane.  `mask` (n, S) bytes,
// optional: rows of a position whose byte is non-zero are written as zeros (the padding mask of
// the layer, applied in the same pass); the position is index a of the source when `pos_is_a`.
__global__ __launch_bounds__(256) void relayout_rows_kernel(const uint4* __restrict__ src,
                                                           uint4* __restrict__ dst,
                                                           const unsigned char* __restrict__ mask,
                                                           int64_t rows, int A, int B, int chunks,
                                                           int pos_is_a) {
  // src is (n, A, B, chunks), dst is (n, B, A, chunks)
  const int64_t total = rows * chunks;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += stride) {
    const int c = static_cast<int>(i % chunks);
    const int64_t row = i / chunks;  // destination row index: (n, b, a)
    const int a = static_cast<int>(row % A);
    const int bb = static_cast<int>(row / A % B);
    const int64_t n = row / A / B;
    uint4 v = src[((n * A + a) * B + bb) * chunks + c];
    if (mask != nullptr && mask[pos_is_a ? n * A + a : n * B + bb]) v = make_uint4(0u, 0u, 0u, 0u);
    dst[i] = v;
  }
}

template <typename VT, typename LT>
int tuned_forward_typed(const void* value, const int64_t* shapes, const int64_t* lsi,
                        const void* loc, const void* attn, void* out, int N, int S, int M, int Lq,
                        int L, int P, int G, bool head_major, bool packed, const float* ref,
                        hipStream_t stream) {
  const PointRows rs = packed ? PointRows{3 * L * P, 3 * L * P} : PointRows{2 * L * P, L * P};
  if (packed) attn = static_cast<const LT*>(loc) + 2 * L * P;  // the weights follow the pairs
  const bool pair = fwd_pairs(G, L * P);
  const int GQ = pair ? 2 * G : G;
  const int GPB = 256 / GQ;
  const dim3 grid(static_cast<unsigned>((Lq + GPB - 1) / GPB) * M, N);
  const int LPB = stage_points_per_pass(GQ, L * P, stage_slots(kStageFwd));
  const size_t lds = static_cast<size_t>(GPB) * group_stride_slots(LPB, 2) * 16;
  Launches run;
#define MSDA_TF(GG, PP)                                                                         \
  run((fwd_tuned_kernel<VT, LT, GG, PP>), grid, dim3(256), lds, stream,          \
                     static_cast<const VT*>(value), shapes, lsi, static_cast<const LT*>(loc),   \
                     static_cast<const LT*>(attn), static_cast<VT*>(out), S, M, Lq, L, P, LPB,   \
                     head_major ? 1 : 0, rs, ref, patch_order_ok(S, M, Lq, L, P) ? 1 : 0)
  switch (G) {
#if !MSDA_EXP_MIN
    case 4:
      if (pair) {
        MSDA_TF(4, true);
      } else {
        MSDA_TF(4, false);
      }
      break;
    case 16: MSDA_TF(16, false); break;
#endif
    default: MSDA_TF(8, false); break;
  }
#undef MSDA_TF
  return run.rc;
}

inline int tuned_forward(const void* value, const int64_t* shapes, const int64_t* lsi,
                         const void* loc, const void* attn, void* out, int N, int S, int M, int D,
                         int Lq, int L, int P, int dtype_val, int dtype_loc, bool head_major,
                         bool packed, const float* ref, hipStream_t stream) {
  int G;
  if (!tuned_shape_ok(N, S, M, D, Lq, L, P, dtype_val, dtype_loc, &G)) return kNotTuned;
#if MSDA_EXP_MIN
  if (dtype_val != 0 || G != 8) return kNotTuned;
  return tuned_forward_typed<float, float>(value, shapes, lsi, loc, attn, out, N, S, M, Lq, L, P, G,
                                           head_major, packed, ref, stream);
#else
  if (!all_aligned16({value, out})) return kNotTuned;
  if (!pair_aligned(loc, dtype_loc) || (packed && !packed_shape_ok(L, P))) return kNotTuned;
#define MSDA_ARGS \
  value, shapes, lsi, loc, attn, out, N, S, M, Lq, L, P, G, head_major, packed, ref, stream
  if (dtype_val == 0) return tuned_forward_typed<float, float>(MSDA_ARGS);
  if (dtype_val == 2)
    return dtype_loc == 0 ? tuned_forward_typed<bf16_t, float>(MSDA_ARGS)
                          : tuned_forward_typed<bf16_t, bf16_t>(MSDA_ARGS);
  return dtype_loc == 0 ? tuned_forward_typed<f16_t, float>(MSDA_ARGS)
                        : tuned_forward_typed<f16_t, f16_t>(MSDA_ARGS);
#undef MSDA_ARGS
#endif
}

// bytes of the per-head grad_out copy at the end of the binned backward's workspace
// (the position-based tile kernels only: they fetch a row once per corner, 3-4 times per
// record, and the copy pays for itself — tile kernel 1.29 -> 1.08 ms; the origin kernel fetches a
// row once per record and runs 0.03 ms faster, with 352 MB less workspace, from grad_out as given)
inline size_t g_copy_bytes(int N, int M, int D, int Lq, int dtype_val, const BinPlan& plan) {
  if (!MSDA_G_COPY || (plan.tile2d && plan.tw == 7)) return 0;
  return static_cast<size_t>(N) * M * Lq * D * (dtype_val == 0 ? 4 : 2);
}

inline int64_t tuned_backward_workspace_bytes(int N, int S, int M, int D, int Lq, int L, int P,
                                              int dtype_val, int dtype_loc, bool allow_binned,
                                              const int64_t* levels_host, bool det = false) {
  int G;
  if (!tuned_shape_ok(N, S, M, D, Lq, L, P, dtype_val, dtype_loc, &G)) return -1;
  BinPlan plan;
  size_t dir_b, rec_b;
  if (allow_binned && make_bin_plan(N, S, M, D, Lq, L, P, G, levels_host, &plan, &dir_b, &rec_b, det))
    return static_cast<int64_t>(dir_b + rec_b + g_copy_bytes(N, M, D, Lq, dtype_val, plan));
  if (det) return -1;  // reproducible form: binned path only, else the caller's generic gather
  if (dtype_val == 0) return 0;
  return static_cast<int64_t>(N) * S * M * D * 4;  // fp32 accumulation image of grad_value
}

// (the plan picks 7 x 7 patches only for rows the origin kernel is built for)
#ifndef MSDA_TO_LDS_PAD
#define MSDA_TO_LDS_PAD 0  // diagnostic builds: unused LDS, to lower the number of resident workgroups
#endif
// patches per workgroup on the fine levels (1: never more than one)
#ifndef MSDA_TO_FUSE
#define MSDA_TO_FUSE kOriginMaxFuse
#endif

// Levels and patch groups of the origin kernel's launch.  A level's patches are taken
// kOriginMaxFuse per workgroup when a patch is expected to hold no more records than the fused
// first round takes (every level receives about the same number of records — Lq * P * 1.31 per
// head — so the patches of a fine level hold few), and when the run tables of the patches stay
// within 8 KB of LDS; else one per workgroup.
inline OriginLevels make_origin_levels(const int64_t* levels_host, int L, int Lq, int P,
                                       const BinPlan& plan) {
  OriginLevels o{};
  int run = 0, pos = 0, least = INT32_MAX;
  for (int l = 0; l < L; ++l) {
    o.H[l] = static_cast<int>(levels_host[2 * l]);
    o.W[l] = static_cast<int>(levels_host[2 * l + 1]);
    o.start[l] = pos;
    pos += o.H[l] * o.W[l];
    o.tx[l] = (o.W[l] + 6) / 7;
    o.tstart[l] = run;
    const int n = o.tx[l] * ((o.H[l] + 6) / 7);
    run += n;
    least = n < least ? n : least;
  }
  o.tstart[L] = run;
  int cap = 2048 / (2 * plan.n_qblocks + 1);
  cap = cap < 1 ? 1 : cap > MSDA_TO_FUSE ? MSDA_TO_FUSE : cap;
  (void)least;
  const double per_level = static_cast<double>(Lq) * P * 1.31;  // records of a level, per head
  int groups = 0;
  for (int l = 0; l < L; ++l) {
    const int n = o.tstart[l + 1] - o.tstart[l];
    const int per = per_level / n <= 0.8 * 256 * kOriginFuseRecords ? cap : 1;
    o.per[l] = per;
    o.gstart[l] = groups;
    groups += (n + per - 1) / per;
  }
  o.gstart[L] = groups;
  return o;
}

// (the plan picks 7 x 7 patches only for rows the origin kernel is built for)
template <typename VT, int GG, int RR, int UU, typename... Args>
inline void launch_tile_origin(Launches& run, dim3 grid, size_t lds_bytes, hipStream_t stream,
                               Args... args) {
  if constexpr (GG * Chunk<VT>::kCPL <= 32) {
    // 16-bit rows are one 16-byte chunk per lane: a third entry per walk step fits the registers
    constexpr int U_ = sizeof(VT) == 2 ? MSDA_TO_U16 : UU;
    constexpr int R_ = sizeof(VT) == 2 ? MSDA_TO_RPT16 : RR;
    run((bwd_tile_origin_kernel<VT, GG, R_, U_>), grid, dim3(256), lds_bytes, stream, args...);
  } else {
    run.rc = static_cast<int>(hipErrorInvalidValue);
  }
}

template <typename VT, typename LT>
int tuned_backward_typed(const void* grad_out, const void* value, const int64_t* shapes,
                         const int64_t* lsi, const void* loc, const void* attn, void* grad_value,
                         void* grad_loc, void* grad_attn, void* workspace, int N, int S, int M,
                         int D, int Lq, int L, int P, int G, bool allow_binned, bool head_major,
                         bool packed, const int64_t* levels_host, const float* ref, int phase,
                         bool det, hipStream_t stream) {
  const PointRows rs = packed ? PointRows{3 * L * P, 3 * L * P} : PointRows{2 * L * P, L * P};
  if (packed) {  // the weights (and their gradient) follow the pairs inside each packed row
    attn = static_cast<const LT*>(loc) + 2 * L * P;
    grad_attn = static_cast<LT*>(grad_loc) + 2 * L * P;
  }
  const int GQ = query_lanes(G);
  const int GPB = 256 / GQ;
  const int LP = L * P;
  BinPlan plan{};
  size_t dir_b = 0, rec_b = 0;
  const bool binned =
      allow_binned && make_bin_plan(N, S, M, D, Lq, L, P, G, levels_host, &plan, &dir_b, &rec_b, det);
  const int slots = stage_slots(binned ? kStageBwdBinned : kStageBwdAtomic);
  int LPB = stage_points_per_pass(GQ, LP, slots);
  // Register-resident points (see bwd_query_kernel): 4 lanes per query, three chunks of 64
  // queries per workgroup, up to 16 points per query staged in one pass (34 KB of LDS at 16).
  int kpc = 0;
  const int LPH = binned ? LP / plan.split : LP;  // points per (virtual) query
  if (MSDA_K3_PRE && binned && (plan.qb == 192 || plan.qb == 128) && LPH <= 16) {
    kpc = GQ == 4 ? (LPH <= 12 ? 3 : 4) : 2;  // ceil(GPB * LPH / 256), rounded up to a built variant
    LPB = LPH;
  }
  Launches run;
  // raw points are handled by the register-resident form of the query kernel only
  if (ref != nullptr && (kpc == 0 || MSDA_K3_DEPTH != 2)) return MSDA_TUNED_ELAYOUT;
  // so is the split of binning and gradients into two launches
  if ((phase & 3) != (kPhaseBin | kPhaseGrad) && (kpc == 0 || !binned)) return MSDA_TUNED_ELAYOUT;
  plan.phase = phase & 3;
  // the reproducible form exists for the binned path only (the caller falls back otherwise)
  if (det && !binned) return kNotTuned;

  if (binned) {
    uint2* dir = static_cast<uint2*>(workspace);
    uint4* records = reinterpret_cast<uint4*>(static_cast<char*>(workspace) + dir_b);
    VT* g_copy = g_copy_bytes(N, M, D, Lq, sizeof(VT) == 4 ? 0 : 2, plan) > 0
                     ? reinterpret_cast<VT*>(static_cast<char*>(workspace) + dir_b + rec_b)
                             : static_cast<VT*>(nullptr);
    const dim3 qgrid(static_cast<unsigned>(plan.n_qblocks) * M, N);
    const size_t qlds = static_cast<size_t>(GPB) * group_stride_slots(LPB, slots) * 16 +
                        (static_cast<size_t>(plan.n_tiles + 4) / 4) * 16 +
                        (kpc ? static_cast<size_t>((plan.qb / GPB - 1) * kpc) * 3 * 256 * 4 : 0) +  // parked points
                        MSDA_K3_LDS_PAD;
    const dim3 tgrid(static_cast<unsigned>(plan.n_tiles) * M, N);
    const size_t TSz = static_cast<size_t>(1) << plan.ts_shift;
    // Rounds of 512 records (2 per thread) and 3 entries per walk step (2 for rows of 16 chunks):
    // 64-74 registers and 19 KB of LDS, so 6-8 workgroups per CU whose phases overlap.  Measured
    // against rounds of 1024 records with 8 row loads per step (94 registers, 5 workgroups per
    // CU): headline shapes 1.11 -> 1.01 ms, and no slower with 4 levels x 8 points or in bf16;
    // rounds of 768 or 256 records and 2 / 4 / 6 / 8 entries per step are all slower.
    constexpr int RPT = 2;
    const size_t tlds = 4 * 256 * static_cast<size_t>(RPT) * 8 + (3 * TSz + 1) * 4 +
                        (2 * static_cast<size_t>(plan.n_qblocks) + 1) * 4;
    // by origin: entry regions, slot counters and list starts, run tables of the patches of a
    // workgroup (see the kernel)
    OriginLevels olv{};
    dim3 ogrid = tgrid;
    size_t olds = 0;
    if (plan.tile2d && plan.tw == 7) {
      olv = make_origin_levels(levels_host, L, Lq, P, plan);
      int fuse = 1;
      for (int l = 0; l < L; ++l) fuse = olv.per[l] > fuse ? olv.per[l] : fuse;
      ogrid = dim3(static_cast<unsigned>(olv.gstart[L]) * M, N);
      olds = static_cast<size_t>(kOriginMaxFuse) * 256 * kOriginFuseRecords * 16 +
             (MSDA_TO_W4 ? 16 + (static_cast<size_t>(kOriginMaxFuse) * 256 * kOriginFuseRecords + 4) * 4 : 0) +
             static_cast<size_t>(kOriginMaxFuse) * (64 + 65) * 4 + (2 * 64 + 65) * 4 +
             static_cast<size_t>(fuse) * (2 * static_cast<size_t>(plan.n_qblocks) + 1) * 4 + MSDA_TO_LDS_PAD;
    }
    // reproducible form: two entry arrays of 4 * 512, one count array
    const size_t dlds = 2 * 4 * 512 * 8 + (2 * TSz + 1) * 4 +
                        (2 * static_cast<size_t>(plan.n_qblocks) + 1) * 4;
    (void)dlds;  // (builds without the reproducible kernel)
#define MSDA_TILE(GG, RR, UU)                                                                    \
  run((bwd_tile_gather_kernel<VT, GG, RR, UU>), tgrid, dim3(256), tlds, stream,   \
                     g_copy ? static_cast<const VT*>(g_copy) : static_cast<const VT*>(grad_out), \
                     shapes, lsi, records, dir, static_cast<VT*>(grad_value), S, M, Lq, L, plan, \
                     g_copy ? 1 : 0)
#define MSDA_TILE_ORIGIN_K(GG, RR, UU)                                                           \
  launch_tile_origin<VT, GG, RR, UU>(                                                            \
      run, ogrid, olds, stream,                                                                  \
      g_copy ? static_cast<const VT*>(g_copy) : static_cast<const VT*>(grad_out),                \
      static_cast<const uint4*>(records), static_cast<const uint2*>(dir),                        \
      static_cast<VT*>(grad_value), S, M, Lq, L, plan, g_copy ? 1 : 0, olv)
#define MSDA_TILE_DET(GG)                                                                        \
  run((bwd_tile_gather_det_kernel<VT, GG>), tgrid, dim3(256), dlds, stream,                      \
      g_copy ? static_cast<const VT*>(g_copy) : static_cast<const VT*>(grad_out), shapes, lsi,   \
      records, dir, static_cast<VT*>(grad_value), S, M, Lq, L, LP, plan, g_copy ? 1 : 0)
#define MSDA_TQ(GG, KK)                                                                          \
  if ((KK) > 0 && plan.split == 2) {                                                             \
    MSDA_TQB(GG, KK, ((KK) > 0 ? 128 : 192), ((KK) > 0 ? 2 : 1));                                \
  } else if ((KK) > 0 && plan.qb == 128) {                                                       \
    MSDA_TQB(GG, KK, ((KK) > 0 ? 128 : 192), 1);                                                 \
  } else {                                                                                       \
    MSDA_TQB(GG, KK, 192, 1);                                                                    \
  }
#define MSDA_TQB(GG, KK, QQ, SS)                                                                 \
  run((bwd_query_kernel<VT, LT, GG, 1, KK, QQ, SS>), qgrid, dim3(256), qlds, stream,             \
                     static_cast<const VT*>(grad_out), static_cast<const VT*>(value), shapes,    \
                     lsi, static_cast<const LT*>(loc), static_cast<const LT*>(attn),             \
                     static_cast<LT*>(grad_loc), static_cast<LT*>(grad_attn),                    \
                     static_cast<float*>(nullptr), records, dir, g_copy, S, M, Lq, L, P, LPB,    \
                     head_major ? 1 : 0, rs, plan, ref)
#if MSDA_EXP_MIN
#define MSDA_TB(GG, UU)                                                                          \
  if (phase & kPhaseNoQuery) {                                                                   \
  } else if (MSDA_EXP_MIN == 2) {                                                                \
    if (kpc != 4 || plan.qb != 128 || plan.split != 2) return kNotTuned;                         \
    MSDA_TQB(GG, 4, 128, 2);                                                                     \
  } else {                                                                                       \
    if ((kpc != 3 && kpc != 2) || plan.split != 1) return kNotTuned;                             \
    if (plan.qb == 128) {                                                                        \
      MSDA_TQB(GG, (query_lanes(GG) == 4 ? 3 : 2), 128, 1);                                      \
    } else {                                                                                     \
      MSDA_TQB(GG, (query_lanes(GG) == 4 ? 3 : 2), 192, 1);                                      \
    }                                                                                            \
  }                                                                                              \
  if ((phase & kPhaseGrad) && !(phase & kPhaseNoTile)) {                                         \
    if (det) return kNotTuned;                                                                   \
    if (plan.tile2d && plan.tw == 7) {                                                           \
      MSDA_TILE_ORIGIN_K(GG, MSDA_TO_RPT, MSDA_TO_U);                                            \
    } else {                                                                                     \
      MSDA_TILE(GG, RPT, UU);                                                                    \
    }                                                                                            \
  }
#else
#define MSDA_TB(GG, UU)                                                                          \
  if (kpc == 0) {                                                                                \
    MSDA_TQ(GG, 0);                                                                              \
  } else if (query_lanes(GG) == 4) {                                                             \
    if (kpc == 3) {                                                                              \
      MSDA_TQ(GG, (query_lanes(GG) == 4 ? 3 : 2));                                               \
    } else {                                                                                     \
      MSDA_TQ(GG, (query_lanes(GG) == 4 ? 4 : 2));                                               \
    }                                                                                            \
  } else {                                                                                       \
    MSDA_TQ(GG, 2);                                                                              \
  }                                                                                              \
  if (phase & kPhaseGrad) {                                                                      \
    if (det) {                                                                                   \
      MSDA_TILE_DET(GG);                                                                         \
    } else if (plan.tile2d && plan.tw == 7) {                                                    \
      MSDA_TILE_ORIGIN_K(GG, MSDA_TO_RPT, MSDA_TO_U);                                            \
    } else {                                                                                     \
      MSDA_TILE(GG, RPT, UU);                                                                    \
    }                                                                                            \
  }
#endif
    switch (G) {
#if !MSDA_EXP_MIN
      case 4: MSDA_TB(4, 3); break;
      case 16: MSDA_TB(16, 2); break;
#endif
      default: MSDA_TB(8, 3); break;
    }
#undef MSDA_TB
#undef MSDA_TQ
#undef MSDA_TQB
#undef MSDA_TILE
#undef MSDA_TILE_DET
#undef MSDA_TILE_ORIGIN_K
    return run.rc;
  }

  // fp32-atomic fallback (very long sequences); it indexes grad_value with the value offsets,
  // so it only takes the position-major layout
  if (head_major) return MSDA_TUNED_ELAYOUT;
#if MSDA_EXP_MIN
  return kNotTuned;
#endif
  plan.qb = 256;
  plan.q_per_block = 256;
  plan.split = 1;
  plan.n_qblocks = (Lq + plan.qb - 1) / plan.qb;
  plan.n_tiles = 0;
  plan.tile2d = 0;
  plan.tw = 0;
  plan.in_order = 1;
  const dim3 grid(static_cast<unsigned>(plan.n_qblocks) * M, N);
  const size_t lds = static_cast<size_t>(GPB) * group_stride_slots(LPB, slots) * 16;
  const int64_t total = static_cast<int64_t>(N) * S * M * D;
  constexpr bool kDirect = sizeof(VT) == 4;
  float* acc = kDirect ? static_cast<float*>(grad_value) : static_cast<float*>(workspace);
  const hipError_t e = hipMemsetAsync(acc, 0, static_cast<size_t>(total) * 4, stream);
  if (e != hipSuccess) return static_cast<int>(e);
#define MSDA_TB(GG)                                                                              \
  run((bwd_query_kernel<VT, LT, GG, 0>), grid, dim3(256), lds, stream,            \
                     static_cast<const VT*>(grad_out), static_cast<const VT*>(value), shapes,    \
                     lsi, static_cast<const LT*>(loc), static_cast<const LT*>(attn),             \
                     static_cast<LT*>(grad_loc), static_cast<LT*>(grad_attn), acc,               \
                     static_cast<uint4*>(nullptr), static_cast<uint2*>(nullptr),                 \
                     static_cast<VT*>(nullptr), S, M, Lq, L, P, LPB, 0, rs, plan,                \
                     static_cast<const float*>(nullptr))
  switch (G) {
#if !MSDA_EXP_MIN
    case 4: MSDA_TB(4); break;
    case 16: MSDA_TB(16); break;
#endif
    default: MSDA_TB(8); break;
  }
#undef MSDA_TB
  if constexpr (!kDirect) {
    const int64_t total8 = total / 8;  // D is a multiple of 8 on this path
    const int64_t cb = (total8 + 255) / 256;
    run((convert_f32_image_kernel<VT>),
                       dim3(static_cast<unsigned>(cb < 4096 ? cb : 4096)), dim3(256), 0, stream,
                       static_cast<const float*>(workspace), static_cast<VT*>(grad_value), total8);
  }
  return run.rc;
}

inline int tuned_backward(const void* grad_out, const void* value, const int64_t* shapes,
                          const int64_t* lsi, const void* loc, const void* attn, void* grad_value,
                          void* grad_loc, void* grad_attn, void* workspace, int N, int S, int M,
                          int D, int Lq, int L, int P, int dtype_val, int dtype_loc,
                          bool allow_binned, bool head_major, bool packed,
                          const int64_t* levels_host, const float* ref, int phase, bool det,
                          hipStream_t stream) {
  int G;
  if (!tuned_shape_ok(N, S, M, D, Lq, L, P, dtype_val, dtype_loc, &G)) return kNotTuned;
  if (!all_aligned16({value, grad_out, grad_value, workspace})) return kNotTuned;
  if (!pair_aligned(loc, dtype_loc) || !pair_aligned(grad_loc, dtype_loc) ||
      (packed && !packed_shape_ok(L, P)))
    return kNotTuned;
#define MSDA_ARGS                                                                               \
  grad_out, value, shapes, lsi, loc, attn, grad_value, grad_loc, grad_attn, workspace, N, S, M, \
      D, Lq, L, P, G, allow_binned, head_major, packed, levels_host, ref, phase, det, stream
#if MSDA_EXP_MIN
  if (dtype_val != 0 || G != 8) return kNotTuned;
  return tuned_backward_typed<float, float>(MSDA_ARGS);
#else
  if (dtype_val == 0) return tuned_backward_typed<float, float>(MSDA_ARGS);
  if (dtype_val == 2)
    return dtype_loc == 0 ? tuned_backward_typed<bf16_t, float>(MSDA_ARGS)
                          : tuned_backward_typed<bf16_t, bf16_t>(MSDA_ARGS);
  return dtype_loc == 0 ? tuned_backward_typed<f16_t, float>(MSDA_ARGS)
                        : tuned_backward_typed<f16_t, f16_t>(MSDA_ARGS);
#endif
#undef MSDA_ARGS
}

}  // namespace msda
