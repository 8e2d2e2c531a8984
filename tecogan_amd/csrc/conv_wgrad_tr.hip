// Weight gradient of the 3x3 stride-1 SAME convolutions of the generator trunk (reference lib/frvsr.py:50-57 under
// tf.gradients, lib/Teco.py:438-449) with TRANSPOSE READS -- gfx950, bf16, 64 -> 64 channels, 32-pixel-wide images:
//
//     dW[kh][kw][ci][co] = sum over (n, y, x) of  X[n, y+kh-1, x+kw-1, ci] * dY[n, y, x, co]          (X zero outside the image)
//
// The GEMM's K dimension is the PIXEL index, the slow axis of both operands in memory (NHWC).  conv_wgrad_row3_bf16_kernel
// (conv_wgrad.hip) transposes both operands through registers into LDS and is bound by the issue rate of that staging code
// (160 TFLOP/s, 11 % MFMA busy, profiles/r02u_*).  Here the [pixel][channel] tiles stay in their natural layout:
//   * a stage = 8 image rows x 32 pixels: the X halo tile (10 x 34 pixels x 64 channels, 43 KB) and the dY tile (32 KB)
//     arrive by LDS-DMA into a double buffer (2 x 75 KB), the next stage's DMA issued between the MFMA groups;
//   * MFMA operands come from ds_read_b64_tr_b16 (builtin __builtin_amdgcn_ds_read_tr16_b64_v4i16): lane i of a 16-lane
//     group receives M[4j + i/4][i%4], j = 0..3, where M[L] are the 4 consecutive 16-bit elements lane L points at -- so
//     lane L = 4*row + chunk addresses (pixel k0 + L/4, channels c0 + 4*(L%4) ..) and lane i gets channel c0 + i of pixels
//     k0 .. k0+3; two reads = the 8 K-values a lane feeds v_mfma_f32_16x16x32_bf16 (A: X, rows = ci; B: dY, columns = co);
//   * a workgroup owns the full 64 x 64 block of ONE layer for all 9 taps: per 32-pixel chunk (one image row of the tile) a
//     wave (32 ci x 32 co quadrant) reads its dY fragments once (4 reads) and its X fragments per tap (36 reads) for 36
//     MFMAs; all 36 accumulators (144 VGPRs) stay in registers over the whole pixel range of the workgroup (split-K over
//     tiles), and are added to dW with fp32 atomics once at the end; the bias gradient is one extra MFMA per fragment
//     with an all-ones A operand in the waves of the first ci half.
// Grouped like the row kernel: `groups` layers of identical geometry in one launch.
// Images of any width that is a multiple of 32 (tiles of 8 x 32 pixels).  Second instantiation YC = 8 for the generator's
// OUTPUT conv (64 -> 3 channels, gradient tensor channel-padded to 8: reference lib/frvsr.py:80 under tf.gradients): the four
// waves split the 64 input channels, one 16-column MFMA tile holds the 3 (+5 zero, +8 don't-care) output channels; HBM-bound
// on the 64-channel HR activation (159 MB per TecoGAN step): 215 us with the row kernel (0.8 TB/s, profiles/r03d_*).
//
// Round 3: validated on MI355X (tools/probe_tr.hip confirmed the lane map, profiles/r03a_probe_tr.txt; parity test in
// tests/test_kernels_gpu.py) and default-on: the grouped trunk launch (32 layers x 76 images) 422.7 -> 353.1 us, 434 -> 520
// TFLOP/s (profiles/r03b_mb_wgrad.txt).  TG_WGRAD_TR=0 is the A/B switch.
#include "common.h"
#include <type_traits>
#include <mutex>
#include <stdlib.h>

#define TG_WTR_MAX_GROUPS 40
struct WgradTrP {
  const u16* xs[TG_WTR_MAX_GROUPS];
  const u16* ys[TG_WTR_MAX_GROUPS];
  float* dws[TG_WTR_MAX_GROUPS];
  float* dbs[TG_WTR_MAX_GROUPS];
  int groups, nsplit;
  int N, H, W;          // images, rows, columns (W a multiple of 32, H of 8); X has 64 channels
  int tiles_y, tiles_x;
  int ntiles;           // N * tiles_y * tiles_x
  int cout;             // real output channels (dW's innermost extent): 64, or <= 8 for the YC = 8 instantiation
  unsigned xbytes, ybytes;
  // Round 5: the LAST group may be a layer with FEWER input channels on the same geometry (the generator's input conv, 51 channels
  // padded to a 56-channel pixel, lib/frvsr.py:47-49): X pixels of narrow_pix bytes (the 16-byte chunks past them read zeros), dW
  // with narrow_rows rows per tap.  narrow_grp < 0: none.
  int narrow_grp, narrow_pix, narrow_rows;
};

typedef short s16x4t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) s16x4t lds_s16x4;

namespace {
constexpr int TR_W = 32, TR_TH = 8, TR_PIX = 128;                      // tile width; bytes per X pixel (64 bf16 channels)
constexpr int TR_XSLOTS = (TR_TH + 2) * (TR_W + 2) * 8;               // 2720 16-byte slots of the X halo tile
constexpr int TR_XINST = (TR_XSLOTS + 63) / 64;                       // 43 wave-wide DMA instructions
constexpr int TR_YOFF = TR_XINST * 1024;                              // 44032
constexpr int TR_NW = 8;                                              // waves per workgroup (two per SIMD)
constexpr int TR_XROUNDS = (TR_XINST + TR_NW - 1) / TR_NW;            // 6 DMA rounds of 8 waves
constexpr unsigned TR_OOB = 0x80000000u;
template <int YC> struct TrGeo {
  static constexpr int YPIX = YC * 2;                                 // bytes per dY pixel
  static constexpr int YROW_SLOTS = TR_W * YPIX / 16;                 // 16-byte slots per tile row: 256 / 32
  static constexpr int YINST = TR_TH * YROW_SLOTS / 64;               // 32 / 4
  static constexpr int YROUNDS = (YINST + TR_NW - 1) / TR_NW;         // 4 / 1
  static constexpr int STAGE = (TR_XINST + YINST) * 1024;             // 76800 / 48128
  static constexpr int ROUNDS = TR_XROUNDS + YROUNDS;                 // 10 / 7
};
}  // namespace

// LDS swizzle of the 128-byte pixels (round 4).  ds_read_b64_tr_b16 is served in two groups of 32 lanes, one LDS cycle each when
// the 32 eight-byte accesses fall into distinct banks (64 banks x 4 B, MI355X_MICROARCH.md LDS table).  A group reads 32 bytes of
// each of the pixels P .. P+3 and P+8 .. P+11; at a 128-byte pitch the even pixels all start at bank 0 and the odd ones at bank
// 32: a 4-WAY conflict, 8 LDS cycles per instruction instead of 2 -- 4 waves x 40 reads x 8 = 1280 cycles per 32-pixel chunk
// against 608 cycles of MFMA per SIMD.  Here the four 32-byte channel pairs of pixel q sit at pair position
// cp ^ tr_swz(q), tr_swz(q) = bit 1 of q | bit 3 of q << 1: for every base pixel the eight pixels of a group then cover all 64
// banks exactly once.  The permutation is applied on the GLOBAL side of the LDS-DMA (a lane picks which chunk it fetches);
// the readers hold 16 lane bases per accumulator row (the swizzle bits of P = K + lane pixel depend on K mod 16 only).
#ifdef TR_NO_SWZ                              // A/B build (tools/build_variant.py conv_wgrad_tr.hip -DTR_NO_SWZ): the round-3 layout
__device__ __forceinline__ int tr_swz(int) { return 0; }
#else
__device__ __forceinline__ int tr_swz(int q) { return ((q >> 1) & 1) | ((q >> 2) & 2); }
#endif

// 8 consecutive K-values (pixels) of one channel for this lane: two transpose reads, 4 pixels apart (each at its own swizzled
// address: pixel + 4 may differ from the pixel in swizzle bit 3)
__device__ __forceinline__ bf16x8 tr_frag2(const unsigned char* p, const unsigned char* p4) {
  const s16x4t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p4));
  typedef short s16x8t __attribute__((ext_vector_type(8)));
  const s16x8t v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf16x8, v);
}

// Round 4: EIGHT waves.  With one wave per SIMD a wave's fragment reads (40 ds_read_b64_tr_b16 per 32-pixel chunk), its 38 MFMAs
// and its share of the next stage's DMA issue ran one after the other (2700 cycles per chunk for 608 cycles of MFMA).  Waves w and
// w + 4 share a SIMD and a 32 x 32 quadrant and split the nine taps 5 + 4: each reads the dY fragments and its own taps' X
// fragments (24 / 20 reads for 20 / 16 MFMAs), so one wave's reads and DMA issue hide under the other's MFMAs; the accumulators
// (80 / 64 registers) stay private and every weight-gradient element is still added by exactly one wave.
template <int YC>
__global__ __launch_bounds__(512, 1) void conv_wgrad_tr_kernel(WgradTrP p) {
  using G = TrGeo<YC>;
  constexpr int YPIX = G::YPIX;
  constexpr int NI = YC == 64 ? 2 : 1, NJ = YC == 64 ? 2 : 1;           // 16 x 16 accumulator tiles per wave (ci x co)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 2 x STAGE (+ slack)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int quad = wave & 3, tset = wave >> 2;              // tset 0: taps 0..4, 1: taps 5..8
  // YC = 64: quadrants = (ci half, co half) of the 64 x 64 block; YC = 8: the four 16-channel ci tiles, one co tile
  const int ci0 = YC == 64 ? 32 * (quad >> 1) : 16 * quad;
  const int co0 = YC == 64 ? 32 * (quad & 1) : 0;
  const int frow = lane & 15, fg = lane >> 4;
  const int grp = blockIdx.x / p.nsplit, split = blockIdx.x - grp * p.nsplit;
  const u16* __restrict__ gx = p.xs[grp];
  const u16* __restrict__ gy = p.ys[grp];
  const auto rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(gx), 0, (int)p.xbytes, 0x00020000);
  const auto rsrcY = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(gy), 0, (int)p.ybytes, 0x00020000);
  const bool narrow = grp == p.narrow_grp;                 // workgroup-uniform
  const int xpix = narrow ? p.narrow_pix : TR_PIX;          // bytes per X pixel in memory (LDS pixels are always 128 bytes)
  const int arows = narrow ? p.narrow_rows : 64;            // rows of dW per tap

  // ---- DMA slot descriptors.  X: slot S = (wave + 8k)*64 + lane -> halo pixel S / 8 = (dy, dx), 16-byte channel chunk S % 8
  int xrel[TR_XROUNDS], xcode[TR_XROUNDS];
#pragma unroll
  for (int k = 0; k < TR_XROUNDS; ++k) {
    const int S = (wave + TR_NW * k) * 64 + lane;
    const int q = S >> 3, c = S & 7;
    const int dy = q / (TR_W + 2), dx = q - (TR_W + 2) * dy;
    const int cgl = (((c >> 1) ^ tr_swz(q)) << 1) | (c & 1);              // the 16-byte chunk of the pixel this slot receives
    xrel[k] = ((dy - 1) * p.W + dx - 1) * xpix + cgl * 16;
    xcode[k] = dy | (dx << 8) | ((S < TR_XSLOTS && cgl * 16 < xpix) ? (1 << 16) : 0);
  }
  //      dY: slot S -> tile row S / YROW_SLOTS, 16-byte unit S % YROW_SLOTS of that row's 32 pixels
  int yrel[G::YROUNDS];
#pragma unroll
  for (int k = 0; k < G::YROUNDS; ++k) {
    const int S = (wave + TR_NW * k) * 64 + lane;
    const int r = S / G::YROW_SLOTS, o = S - r * G::YROW_SLOTS;
    // YC = 64: unit o = pixel o / 8, chunk o % 8 of the row, swizzled like X (the tile's pixel index is 32 r + o / 8)
    const int og = YC == 64 ? (o & ~7) | ((((o & 7) >> 1) ^ tr_swz(o >> 3)) << 1) | (o & 1) : o;
    yrel[k] = (wave + TR_NW * k) < G::YINST ? r * p.W * YPIX + og * 16 : -1;
  }
  auto issue_dma = [&](int tile, int buf, int r0, int r1) {          // DMA rounds [r0, r1) of the stage (compile-time bounds)
    const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
    const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
    const int y0 = ty * TR_TH, x0 = tx * TR_W;
    const int pix0 = (n * p.H + y0) * p.W + x0;                             // wave-uniform: the tile's pixel (0,0)
    unsigned char* dst = smem + buf * G::STAGE;
#pragma unroll
    for (int r = r0; r < r1; ++r) {
      if (r < TR_XROUNDS) {
        const int inst = wave + TR_NW * r;
        if (r + 1 < TR_XROUNDS || inst < TR_XINST) {
          const int dy = xcode[r] & 255, dx = (xcode[r] >> 8) & 255;
          const bool ok = (xcode[r] >> 16) && (unsigned)(y0 + dy - 1) < (unsigned)p.H && (unsigned)(x0 + dx - 1) < (unsigned)p.W;
          const unsigned off = ok ? (unsigned)(pix0 * xpix + xrel[r]) : TR_OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void_t*)(dst + inst * 1024), 16, (int)off, 0, 0, 0);
        }
      } else if (r < G::ROUNDS) {
        const int k = r - TR_XROUNDS;
        const int inst = wave + TR_NW * k;
        if (k + 1 < G::YROUNDS || inst < G::YINST) {
          const unsigned off = yrel[k] >= 0 ? (unsigned)(pix0 * YPIX + yrel[k]) : TR_OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (lds_void_t*)(dst + TR_YOFF + inst * 1024), 16, (int)off, 0, 0, 0);
        }
      }
    }
  };

  int tile = split;
  if (tile >= p.ntiles) return;                             // (workgroup-uniform; its accumulators would be zero)
  issue_dma(tile, 0, 0, G::ROUNDS);

  // ---- per-lane fragment bases: lane L = frow of group fg points at pixel 8 fg + L/4 (+4 for the second read), channels
  //      c0 + 4 (L % 4); everything else (image row, tap shift, 16-channel tile) is a compile-time offset.  (YC = 8: the
  //      lanes with L % 4 >= 2 point 16 / 24 bytes into the NEXT pixel -- they feed output columns 8..15, which are discarded.)
  const int lp = 8 * fg + (frow >> 2), lc = 4 * (frow & 3);
  // halo (row 0, column lp) = image (y0 - 1, x0 + lp - 1): tap (0, 0) of row 0; the fragment of halo pixel offset K is read at
  // atab[i][K & 15] + K * TR_PIX (K compile time)
  int atab[NI][16], bbase[NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) atab[i][k] = lp * TR_PIX + (((ci0 >> 4) + i) ^ tr_swz(lp + k)) * 32 + lc * 2;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    bbase[j] = TR_YOFF + lp * YPIX + (YC == 64 ? (((co0 >> 4) + j) ^ tr_swz(lp)) * 32 + lc * 2 : (co0 + lc) * 2);

  f32x4 acc[5][NI][NJ];                                      // this wave's taps: T0 + t
  f32x4 accb[NJ];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NJ; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef short s16x8o __attribute__((ext_vector_type(8)));
  const s16x8o ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
  const bool do_bias = p.dbs[grp] != nullptr && ci0 == 0 && tset == 0;   // wave-uniform

  // the stage loop for the taps [T0, T0 + NTAP) (compile time: the two tap sets are two copies of the loop, chosen per wave).
  // BARRIER CONTRACT: the two copies are selected by a wave-divergent (workgroup-non-uniform) branch, so the waves of
  // a workgroup rendezvous at two different s_barrier instructions.  gfx9-family hardware counts barrier ARRIVALS per workgroup,
  // whatever the PC, so this is well defined on the target as long as both copies execute the SAME NUMBER of barriers: they do
  // by construction -- the trip count depends only on (tile, p.nsplit, p.ntiles), all workgroup-uniform, and the tap set enters
  // only the barrier-free body.  Do not add a barrier, an early exit or a tap-dependent trip count to one copy only.  (Round 5
  // built the alternative -- ONE loop and barrier site, the two bodies behind a wave-uniform branch inside it -- and measured it:
  // grouped trunk launch 238 -> 464 us, G = 20 / N = 40 88.7 -> 151.5 us, profiles/r05a_misc.txt (cause not investigated;
  // the step lost 0.2 ms with it).  Reverted.)
  auto run = [&](auto t0c, auto ntc) {
    constexpr int T0 = decltype(t0c)::value, NTAP = decltype(ntc)::value;
    int buf = 0;
    while (true) {
      const int ntile = tile + p.nsplit;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's slots of the stage
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                           // everybody's; nobody still reads the other buffer
      const bool has_next = ntile < p.ntiles;
      const unsigned char* sb = smem + buf * G::STAGE;
#pragma unroll
      for (int yy = 0; yy < TR_TH; ++yy) {                    // one image row of the tile = 32 pixels = one K step
        bf16x8 bq[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {      // (dY: the lane's pixel has bit 2 clear, so pixel + 4 keeps its swizzle bits)
          const unsigned char* q = sb + bbase[j] + yy * TR_W * YPIX;
          bq[j] = tr_frag2(q, q + 4 * YPIX);
        }
        if (do_bias) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bq[j], accb[j], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          const int kh = (T0 + t) / 3, kw = (T0 + t) % 3;
          bf16x8 aq[NI];
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const int K = (yy + kh) * (TR_W + 2) + kw;                                            // image (y0+yy+kh-1, x0+lp+kw-1)
            aq[i] = tr_frag2(sb + atab[i][K & 15] + K * TR_PIX, sb + atab[i][(K + 4) & 15] + (K + 4) * TR_PIX);
          }
#pragma unroll
          for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[i], bq[j], acc[t][i][j], 0, 0, 0);
        }
        // the next stage's DMA rounds spread over the rows (an LDS-DMA instruction costs 60-180 issue cycles)
        if (has_next) {
          __builtin_amdgcn_sched_barrier(0);
          if (yy * 2 < G::ROUNDS) issue_dma(ntile, buf ^ 1, yy * 2, yy * 2 + 2 < G::ROUNDS ? yy * 2 + 2 : G::ROUNDS);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!has_next) break;
      tile = ntile;
      buf ^= 1;
    }

    // ---- split-K reduction: D row 4 fg + r = input channel, column frow = output channel
    float* __restrict__ dw = p.dws[grp];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ci = ci0 + 16 * i + 4 * fg + r, co = co0 + 16 * j + frow;
            if ((YC == 64 || co < p.cout) && ci < arows) unsafeAtomicAdd(dw + ((T0 + t) * arows + ci) * p.cout + co, acc[t][i][j][r]);
          }
  };
  if (tset == 0) run(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
  else run(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{});
  if (do_bias && fg == 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int co = co0 + 16 * j + frow;
      if (YC == 64 || co < p.cout) unsafeAtomicAdd(p.dbs[grp] + co, accb[j][0]);
    }
  }
}

template <int YC>
static void wgrad_tr_go(const WgradTrP& p, double flops, double bytes, hipStream_t st) {
  constexpr int LDS = 2 * TrGeo<YC>::STAGE + 64;            // + slack: the YC = 8 fragment reads reach 16 bytes past the last pixel
  static std::once_flag attr_once;
  std::call_once(attr_once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_tr_kernel<YC>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  TG_LAUNCH(YC == 64 ? "conv_wgrad_tr" : "conv_wgrad_tr_out", flops, bytes, conv_wgrad_tr_kernel<YC>,
            dim3((unsigned)(p.groups * p.nsplit)), dim3(512), LDS, st, p);
}

// returns 1 if launched, 0 otherwise.  Geometries: 3x3 s1 SAME, 64 input channels (ldx = 64), images H % 8 == 0, W % 32 == 0;
// 64 -> 64 (ldy = 64: the generator trunk, grouped) or 64 -> <= 8 with ldy = 8 (the generator's output conv).
int tg_wgrad_tr_launch(const tg_conv_desc* d, int groups, const void* const* x, int ldx, const void* const* y, int ldy,
                       float* const* dw, float* const* dbias, hipStream_t st, const void* x_narrow, int ldx_narrow, int cin_narrow,
                       const void* y_narrow, float* dw_narrow, float* db_narrow) {
  static const bool enabled = getenv("TG_WGRAD_TR") == nullptr || atoi(getenv("TG_WGRAD_TR")) != 0;
  const int gtot = groups + (x_narrow ? 1 : 0);
  if (!enabled || tg_det() || groups < 1 || gtot > TG_WTR_MAX_GROUPS) return 0;
  if (x_narrow && !(d->Cout == 64 && ldy == 64 && ldx_narrow % 8 == 0 && ldx_narrow <= 64 && cin_narrow >= 1 && cin_narrow <= ldx_narrow &&
                    y_narrow && dw_narrow && ((((uintptr_t)x_narrow | (uintptr_t)y_narrow)) & 15) == 0)) return 0;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1 || d->mode != 0) return 0;
  if (d->Win % TR_W != 0 || d->Wout != d->Win || d->Hin != d->Hout || d->Hin % TR_TH != 0) return 0;
  const bool trunk = d->Cout == 64 && ldy == 64, outc = d->Cout <= 8 && ldy == 8;
  if (d->Cin != 64 || ldx != 64 || !(trunk || outc)) return 0;
  const int64_t px = (int64_t)d->N * d->Hin * d->Win;
  if (px * TR_PIX >= ((int64_t)1 << 31)) return 0;
  WgradTrP p;
  for (int g = 0; g < TG_WTR_MAX_GROUPS; ++g) {
    const int k = g < groups ? g : 0;
    p.xs[g] = (const u16*)x[k]; p.ys[g] = (const u16*)y[k]; p.dws[g] = dw[k]; p.dbs[g] = dbias ? dbias[k] : nullptr;
  }
  p.narrow_grp = -1; p.narrow_pix = TR_PIX; p.narrow_rows = 64;
  if (x_narrow) {
    p.xs[groups] = (const u16*)x_narrow; p.ys[groups] = (const u16*)y_narrow; p.dws[groups] = dw_narrow; p.dbs[groups] = db_narrow;
    p.narrow_grp = groups; p.narrow_pix = ldx_narrow * 2; p.narrow_rows = cin_narrow;
  }
  groups = gtot;
  p.groups = groups;
  p.N = d->N; p.H = d->Hin; p.W = d->Win;
  p.tiles_y = d->Hin / TR_TH; p.tiles_x = d->Win / TR_W;
  p.ntiles = d->N * p.tiles_y * p.tiles_x;
  p.cout = d->Cout;
  p.xbytes = (unsigned)(px * TR_PIX);
  p.ybytes = (unsigned)(px * ldy * 2);
  int nsplit = tg_num_cus() / groups;                              // one workgroup per CU (150 / 94 KB of LDS)
  if (nsplit < 1) nsplit = 1;
  if (nsplit > p.ntiles) nsplit = p.ntiles;
  p.nsplit = nsplit;
  const double M = (double)px;
  const double gfl = groups - (x_narrow ? 1.0 - cin_narrow / 64.0 : 0.0);      // the narrow group counts with its real channels
  const double flops = 2.0 * gfl * M * 9.0 * 64 * d->Cout, bytes = groups * (M * (TR_PIX + 2.0 * ldy) + 36.0 * 64 * d->Cout);
  if (trunk) wgrad_tr_go<64>(p, flops, bytes, st);
  else wgrad_tr_go<8>(p, flops, bytes, st);
  return 1;
}
