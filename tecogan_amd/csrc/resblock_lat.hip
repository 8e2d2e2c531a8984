// One residual block of generator_F -- conv3x3 + ReLU + conv3x3 + skip, reference lib/frvsr.py:50-57 -- and the
// input-gradient chain of the same block under tf.gradients (lib/Teco.py:441-449) as ONE launch, for the LATENCY regime
// of the training recurrence ([B,32,32,64] frames: 4096 pixels, 19 frames x 16 blocks forward and backward per step).
//
// Why.  As two launches of conv3x3_tile<4,16> a block costs 2 x 4.5 us for 2 x 18 MFMAs per wave: each launch is a kernel
// boundary (1.45 us), a cold round trip to L2 / Infinity Cache for the halo tile and the weights, 0.12 us of matrix work
// and a store drain -- 1216 such launches are half of the TecoGAN step.  Here the intermediate tensor never leaves the
// CU, so one boundary and one memory round trip per block disappear:
//   * a workgroup (4 waves, one per SIMD) owns a 4x4 output tile: [4,32,32] -> 256 workgroups, one per CU;
//   * level 1 computes the first conv on the 6x6 halo region the second conv needs (from the 8x8 input region), as three
//     2x8 MFMA pixel tiles (columns 6, 7 are discarded padding), writes it to LDS as bf16 -- zero outside the image, that
//     is the second conv's SAME padding -- and stores the 4x4 interior to HBM (the weight gradients need it later);
//   * level 2 computes the second conv on the 4x4 tile from LDS, adds the skip from the staged input region and stores;
//   * a wave owns 16 output channels of BOTH convs: its 2 x 18 weight fragments (144 VGPRs) are requested in consumption
//     order before anything else, so the weight stream of the second conv flies during the first conv's matrix work; the
//     weights are what bounds a node (147 KB per CU through a 64 B/clk port) -- not the 72 MFMAs per wave;
//   * MFMA operands swapped (A = weights, B = pixels) exactly as in conv3x3_tile's chain tiles, taps and K-steps in the
//     same order: the result is BIT-IDENTICAL to the two-launch path (tests/test_kernels_gpu.py holds that);
//   * LDS: 160-byte pixel pitch; the input region with an 8-position row pitch and the intermediate with a 12-position
//     one are conflict-free under the gfx950 ds_read_b128 lane grouping for every tap (brute-force search over the
//     guide's bank table, tools/lds_layout_search.py); 22 KB of LDS, <= 256 registers: fits beside a VGG workgroup.
// Halo recompute: level 1 does 3 pixel tiles for 1 of output (2.25x the first conv's MACs) -- irrelevant at 0.3 us of
// matrix work per node; the kernel is NOT for the throughput regime (conv3x3_ws.hip has the 1080p convs).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct RbP {
  const void* x;        // [N,H,W,64] bf16  block input (forward) / gradient w.r.t. the block output (backward)
  const void* w1;       // [9][64][64] bf16 weights of the FIRST conv applied, [tap][out][in]
  const void* w2;       // ... of the second
  const float* b1;      // nullable
  const float* b2;      // nullable
  const void* aux1;     // nullable: level-1 result *= (aux1 > 0)   (backward: the saved relu(conv_1) output)
  const void* aux2;     // nullable: level-2 result *= (aux2 > 0)   (backward of block 1: the ReLU of the input stage)
  void* mid;            // [N,H,W,64] level-1 result, nullable (stateless forward)
  void* out;            // [N,H,W,64]
  int N, H, W;
  int flip;             // 1: taps mirrored (input-gradient form)
  float nslope1;        // level-1 activation max(v, v * nslope1): ReLU 0, none 1
  int tiles_y, tiles_x, ntiles;
  unsigned bytes;       // extent of every [N,H,W,64] tensor
  int prio;
  int wfrag;          // weights in fragment order (tg_pack_weights_frag) instead of [tap][out][in] rows
};

typedef unsigned int u32x4r __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2r __attribute__((ext_vector_type(2)));

namespace {
constexpr int RB_P = 160;                       // bytes per LDS position (64 bf16 + 32 pad)
constexpr int RB_XR = 8, RB_XPOS = 8 * 8 + 2;   // input region: 8x8 positions, row pitch 8 (+2: the padding columns of the last row read on)
constexpr int RB_HR = 12, RB_HPOS = 6 * 12;     // intermediate: 6 rows, row pitch 12
constexpr unsigned RB_OOB = 0x80000000u;
constexpr int RB_DIST = 14;                     // prefetch distance of the weight stream (fragments): 4 .. 18 measured within 8 %
                                                // (4.57 / 4.31 / 4.32 / 4.31 / 4.20 / 4.25 us per block at 4 / 6 / 8 / 10 / 14 / 18,
                                                // everything up front 4.91; profiles/r04c_ab.txt)
}  // namespace

// Cycle stamps (tools/trace_rb.py builds a private -DTG_RB_TRACE copy of the library; the product build has none of it).
#ifdef TG_RB_TRACE
__device__ unsigned long long tg_rb_trace_buf[4 * 16];
#define RB_STAMP(i)                                                                                          \
  do {                                                                                                       \
    if (blockIdx.x == gridDim.x / 2 && lane == 0) tg_rb_trace_buf[wave * 16 + (i)] = (unsigned long long)clock64(); \
  } while (0)
extern "C" int tg_debug_rb_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_rb_trace_buf), sizeof(unsigned long long) * 64);
}
#else
#define RB_STAMP(i) do { } while (0)
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void rb_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    rb_static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ u32x2r rb_pack4(const float (&v)[4]) {
  u32x2r o;
  o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
  o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
  return o;
}
__device__ __forceinline__ void rb_unpack4(const u32x2r& a, float (&f)[4]) {
  f[0] = __uint_as_float(a.x << 16);
  f[1] = __uint_as_float(a.x & 0xffff0000u);
  f[2] = __uint_as_float(a.y << 16);
  f[3] = __uint_as_float(a.y & 0xffff0000u);
}

// FRAG: weights in FRAGMENT order ([step][wave][lane][16 B], tg_pack_weights_frag: a wave-load is 1 KiB contiguous = 8 whole cache
//   lines) instead of the [tap][out][in] rows of tg_conv_forward's operand (16 HALF lines per wave-load).  This is the lever:
//   6.59 -> 4.42 us per block, weight stream of a node 3400 + 4300 -> 1900 + 1900 cycles (profiles/r04b_ab.txt): the CU's miss
//   path is bound by the number of lines in flight, and a half-line request holds a whole line's slot.  (nt loads: no gain.)
// DIST: prefetch distance of the weight stream in fragments.  The 36 fragments of both convs are ONE stream in consumption
//   order; DIST of them are requested before the input region is staged, and fragment i + DIST is requested right before the
//   MFMAs of fragment i.  A wave's vector-memory queue is shallow: "issue everything first" (round-4 session A) left the wave
//   stalled in load issue for 7900 of the node's 12700 cycles while landed fragments waited for their MFMAs.
template <bool HAS_AUX1, bool HAS_AUX2, bool FRAG, int DIST>
__global__ __launch_bounds__(256, 2) void resblock_lat_kernel(RbP p) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[RB_XPOS * RB_P];
  __shared__ __attribute__((aligned(16))) unsigned char hs[RB_HPOS * RB_P];
  static_assert(DIST >= 1 && DIST <= 36, "prefetch distance in fragments");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fg = lane >> 4;
  if (p.prio) __builtin_amdgcn_s_setprio(3);
  RB_STAMP(0);

  // workgroup -> tile: consecutive blocks go to consecutive XCDs (block b runs on XCD b % 8, observed); give every XCD a
  // CONTIGUOUS range of tiles, so that the halo pixels a tile shares with its neighbours were written through the same L2
  int b = blockIdx.x;
  if ((p.ntiles & 7) == 0) b = (b & 7) * (p.ntiles >> 3) + (b >> 3);
  const int tx = b % p.tiles_x, t1 = b / p.tiles_x;
  const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
  const int y0 = ty * 4, x0 = tx * 4;

  const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.bytes, 0x00020000);
  const auto rsW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, 9 * 64 * 64 * 2, 0x00020000);
  const auto rsW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), 0, 9 * 64 * 64 * 2, 0x00020000);
  const auto rsA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_AUX1 ? p.aux1 : p.x), 0, (int)p.bytes, 0x00020000);
  const auto rsA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_AUX2 ? p.aux2 : p.x), 0, (int)p.bytes, 0x00020000);
  const int cbyte = (wave * 16 + fg * 4) * 2;       // byte offset of this lane's four output channels inside a pixel

  // ---- global loads (no branch around any of them: hipcc answers a load inside a branch with s_waitcnt vmcnt(0) at the join
  //      -- the whole weight stream -- and a null pointer is a zero-length buffer that reads zeros / drops stores) -------------
  const auto rsB1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b1), 0, p.b1 ? 256 : 0, 0x00020000);
  const auto rsB2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b2), 0, p.b2 ? 256 : 0, 0x00020000);
  const u32x4r bq1 = __builtin_amdgcn_raw_buffer_load_b128(rsB1, (wave * 16 + fg * 4) * 4, 0, 0);
  const u32x4r bq2 = __builtin_amdgcn_raw_buffer_load_b128(rsB2, (wave * 16 + fg * 4) * 4, 0, 0);
  // (1) the 8x8 input region: 512 16-byte items, two per thread; pixels outside the image read zeros (bounds check)
  u32x4r xr[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int item = tid + k * 256;
    const int pix = item >> 3, ch = item & 7;
    const int gy = y0 - 2 + (pix >> 3), gx = x0 - 2 + (pix & 7);
    const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(ok ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + ch * 16) : RB_OOB), 0, 0);
  }
  // (2) masks: four channels (8 bytes) of the pixel this lane finishes at each level
  u32x2r m1[HAS_AUX1 ? 3 : 1], m2;
  if constexpr (HAS_AUX1) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int ry = 2 * t + (frow >> 3), rx = frow & 7;
      const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
      const bool ok = rx < 6 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      m1[t] = __builtin_amdgcn_raw_buffer_load_b64(rsA1, (int)(ok ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + cbyte) : RB_OOB), 0, 0);
    }
  }
  const int oy = y0 + (frow >> 2), ox = x0 + (frow & 3);
  const bool out_ok = oy < p.H && ox < p.W;
  const int out_off = ((n * p.H + oy) * p.W + ox) * 128 + cbyte;
  if constexpr (HAS_AUX2) m2 = __builtin_amdgcn_raw_buffer_load_b64(rsA2, (int)(out_ok ? (unsigned)out_off : RB_OOB), 0, 0);
  // (3) the weight stream: fragment i < 18 is step i of the first conv, i >= 18 step i - 18 of the second; lane (frow, fg) of
  //     step s = (tap, K-half kk) holds w[tap][16 wave + frow][32 kk + 8 fg .. +8]
  u32x4r wA[18], wB[18];
  const int wlane = FRAG ? wave * 1024 + lane * 16 : ((wave * 16 + frow) * 64 + fg * 8) * 2;
  auto wload = [&](const auto& rs, int s) {
    const int tap = s >> 1, kk = s & 1;
    const int wtap = p.flip ? 8 - tap : tap;
    return __builtin_amdgcn_raw_buffer_load_b128(rs, wlane, FRAG ? (wtap * 2 + kk) * 4096 : wtap * 8192 + kk * 64, 0);
  };
#define RB_WISSUE(i)                                            \
  do {                                                          \
    if constexpr ((i) < 18) wA[(i) < 18 ? (i) : 0] = wload(rsW1, (i));          \
    else if constexpr ((i) < 36) wB[(i) < 36 && (i) >= 18 ? (i) - 18 : 0] = wload(rsW2, (i) - 18); \
  } while (0)
  rb_static_for<0, DIST>([&](auto i) { RB_WISSUE(decltype(i)::value); });
  const float bv1[4] = {__uint_as_float(bq1.x), __uint_as_float(bq1.y), __uint_as_float(bq1.z), __uint_as_float(bq1.w)};
  const float bv2[4] = {__uint_as_float(bq2.x), __uint_as_float(bq2.y), __uint_as_float(bq2.z), __uint_as_float(bq2.w)};
  __builtin_amdgcn_sched_barrier(0);          // keep the loads up here (hipcc sinks them to their uses otherwise)
  RB_STAMP(1);

  // ---- input region -> LDS ----------------------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int item = tid + k * 256;
    *reinterpret_cast<u32x4r*>(xs + (item >> 3) * RB_P + (item & 7) * 16) = xr[k];
  }
  RB_STAMP(2);
  __syncthreads();
  RB_STAMP(3);

  // ---- level 1: first conv on the 6x6 region, three 2x8 pixel tiles (lane: row 2t + frow/8, column frow%8) -------------
  const unsigned char* xb = xs + ((frow >> 3) * RB_XR + (frow & 7)) * RB_P + fg * 16;
  f32x4 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // software pipeline: the pixel fragments of step s + 1 are requested from LDS before the MFMAs of step s, so a step costs its
  // MFMAs, not an LDS round trip
  auto xfrag = [&](int s, int t) {
    const int tap = s >> 1, kk = s & 1;
    return *reinterpret_cast<const uint4*>(xb + ((2 * t + tap / 3) * RB_XR + tap % 3) * RB_P + kk * 64);
  };
  uint4 bf[3], nbf[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) bf[t] = xfrag(0, t);
  rb_static_for<0, 18>([&](auto sv) {
    constexpr int s = decltype(sv)::value;
    RB_WISSUE(s + DIST);                                       // (a queue slot has just been freed by fragment s)
    if constexpr (s < 17) {
#pragma unroll
      for (int t = 0; t < 3; ++t) nbf[t] = xfrag(s + 1, t);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wA[s]), *reinterpret_cast<bf16x8*>(&bf[t]),
                                                       acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);                         // (pinned: hoisted to the top these loads stall the issue again)
#pragma unroll
    for (int t = 0; t < 3; ++t) bf[t] = nbf[t];
  });
  RB_STAMP(4);
  // level-1 epilogue: bias, activation, mask; zero outside the image (= the second conv's SAME padding); bf16 -> LDS;
  // the tile's own 4x4 pixels also go to HBM
  // (columns 6, 7 of a pixel tile are padding: their results land in the unused positions 6, 7 of the 12-position rows)
  const auto rsM = __builtin_amdgcn_make_buffer_rsrc(p.mid, 0, p.mid ? (int)p.bytes : 0, 0x00020000);
  const auto rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.bytes, 0x00020000);
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int ry = 2 * t + (frow >> 3), rx = frow & 7;
    const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
    const bool inimg = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = acc[t][r] + bv1[r];
      v[r] = fmaxf(v[r], v[r] * p.nslope1);
    }
    if constexpr (HAS_AUX1) {
      float a[4];
      rb_unpack4(m1[t], a);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= a[r] > 0.f ? 1.f : 0.f;
    }
    u32x2r o = rb_pack4(v);
    if (!inimg) o = u32x2r{0u, 0u};
    *reinterpret_cast<u32x2r*>(hs + (ry * RB_HR + rx) * RB_P + cbyte) = o;
    const bool own = inimg && ry >= 1 && ry <= 4 && rx >= 1 && rx <= 4;
    __builtin_amdgcn_raw_buffer_store_b64(o, rsM, (int)(own ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + cbyte) : RB_OOB), 0, 0);
  }
  RB_STAMP(5);
  __syncthreads();
  RB_STAMP(6);

  // ---- level 2: second conv on the 4x4 tile (lane: row frow/4, column frow%4) ----------------------------------------
  const unsigned char* hb = hs + ((frow >> 2) * RB_HR + (frow & 3)) * RB_P + fg * 16;
  // one accumulator, 18 dependent MFMAs: all 18 pixel fragments are requested up front (the first conv's weight registers are
  // free by now), so the chain never waits for an LDS round trip -- only for its weight fragments
  f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 hf[18];
#pragma unroll
  for (int s = 0; s < 18; ++s)
    hf[s] = *reinterpret_cast<const uint4*>(hb + ((s / 6) * RB_HR + (s >> 1) % 3) * RB_P + (s & 1) * 64);
  __builtin_amdgcn_sched_barrier(0);
  rb_static_for<0, 18>([&](auto sv) {
    constexpr int s = decltype(sv)::value;
    RB_WISSUE(18 + s + DIST);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wB[s]), *reinterpret_cast<bf16x8*>(&hf[s]),
                                                   acc2, 0, 0, 0);
    if constexpr (18 + s + DIST < 36) __builtin_amdgcn_sched_barrier(0);
  });
#undef RB_WISSUE
  RB_STAMP(7);
  // level-2 epilogue: bias, skip (the centre of the staged input region), mask, store
  {
    const u32x2r sk = *reinterpret_cast<const u32x2r*>(xs + (((frow >> 2) + 2) * RB_XR + (frow & 3) + 2) * RB_P + cbyte);
    float s[4], v[4];
    rb_unpack4(sk, s);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = acc2[r] + bv2[r];
      v[r] += s[r];
    }
    if constexpr (HAS_AUX2) {
      float a[4];
      rb_unpack4(m2, a);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= a[r] > 0.f ? 1.f : 0.f;
    }
    __builtin_amdgcn_raw_buffer_store_b64(rb_pack4(v), rsO, (int)(out_ok ? (unsigned)out_off : RB_OOB), 0, 0);
  }
  RB_STAMP(8);
#ifdef TG_RB_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RB_STAMP(9);
#endif
}

extern "C" int tg_resblock(int mode, const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                           const void* aux1, const void* aux2, void* mid, void* out, int N, int H, int W, int C, int dtype,
                           int w_frag, void* stream) {
  TG_CHECK_ARG(mode == 0 || mode == 1, "mode must be 0 (forward) or 1 (input gradient)");
  TG_CHECK_ARG(dtype == TG_BF16 && C == 64, "bf16 tensors with 64 channels only (the fp32 parity mode runs the block as two tg_conv_forward launches)");
  TG_CHECK_ARG(x && w1 && w2 && out && N > 0 && H > 0 && W > 0, "null pointer / empty tensor");
  TG_CHECK_ARG((((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)out | (uintptr_t)mid | (uintptr_t)aux1 | (uintptr_t)aux2) & 15) == 0,
               "pointers must be 16-byte aligned");
  const int64_t bytes = (int64_t)N * H * W * 128;
  TG_CHECK_ARG(bytes < ((int64_t)1 << 31), "tensor too large for 32-bit buffer offsets");
  RbP p;
  p.x = x; p.w1 = w1; p.w2 = w2; p.b1 = b1; p.b2 = b2; p.aux1 = aux1; p.aux2 = aux2; p.mid = mid; p.out = out;
  p.N = N; p.H = H; p.W = W;
  p.flip = mode;
  p.nslope1 = mode == 0 ? 0.f : 1.f;            // forward: ReLU between the convs; backward: the mask does that job
  p.tiles_y = (H + 3) / 4; p.tiles_x = (W + 3) / 4;
  const int64_t nt = (int64_t)N * p.tiles_y * p.tiles_x;
  TG_CHECK_ARG(nt < ((int64_t)1 << 24), "too many tiles: this is the latency-regime kernel");
  p.ntiles = (int)nt;
  p.bytes = (unsigned)bytes;
  p.prio = 1;                                   // s_setprio 3 in the chain kernels (measured in round 2, see conv3x3.hip)
  p.wfrag = w_frag != 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double px = (double)N * H * W;
  const double fl = 2.0 * 2.0 * px * 64.0 * 576.0;
  const double by = px * 128.0 * (2 + (mid != nullptr) + (aux1 != nullptr) + (aux2 != nullptr)) + 2.0 * 73728.0;
  int dist = RB_DIST;
#ifdef TG_RB_TRACE
  if (getenv("TG_RB_DIST")) dist = atoi(getenv("TG_RB_DIST"));       // read per call: tools/trace_rb.py sweeps it
#endif
  auto go = [&](auto ftag, auto dtag) {
    constexpr bool F = decltype(ftag)::value;
    constexpr int D = decltype(dtag)::value;
    if (aux1 && aux2) TG_LAUNCH("resblock_lat<bwd,mask2>", fl, by, (resblock_lat_kernel<true, true, F, D>), dim3(p.ntiles), dim3(256), 0, st, p);
    else if (aux1) TG_LAUNCH("resblock_lat<bwd>", fl, by, (resblock_lat_kernel<true, false, F, D>), dim3(p.ntiles), dim3(256), 0, st, p);
    else if (aux2) TG_LAUNCH("resblock_lat<mask2>", fl, by, (resblock_lat_kernel<false, true, F, D>), dim3(p.ntiles), dim3(256), 0, st, p);
    else TG_LAUNCH("resblock_lat<fwd>", fl, by, (resblock_lat_kernel<false, false, F, D>), dim3(p.ntiles), dim3(256), 0, st, p);
  };
  using T = std::true_type;
  using Fa = std::false_type;
  if (!w_frag) go(Fa{}, std::integral_constant<int, RB_DIST>{});
#ifdef TG_RB_TRACE
  else if (dist == 4) go(T{}, std::integral_constant<int, 4>{});
  else if (dist == 6) go(T{}, std::integral_constant<int, 6>{});
  else if (dist == 8) go(T{}, std::integral_constant<int, 8>{});
  else if (dist == 10) go(T{}, std::integral_constant<int, 10>{});
  else if (dist == 18) go(T{}, std::integral_constant<int, 18>{});
  else if (dist == 36) go(T{}, std::integral_constant<int, 36>{});
#endif
  else go(T{}, std::integral_constant<int, RB_DIST>{});
  (void)dist;
  TG_CHECK_LAUNCH();
}
