// One residual block of generator_F -- conv3x3 + ReLU + conv3x3 + skip, reference lib/frvsr.py:50-57 -- as ONE launch in the
// THROUGHPUT regime: the inference step at 1080p output ([1,270,480,64] per block, 16 blocks per frame, main.py:195-216).
//
// Why.  As two launches of conv3x3_ws<frag> a block costs 14.0 + 16.8 us for 19.1 GFLOP (MFMA floor 7.6 us): each launch pays its
// own kernel boundary, a prologue in which every workgroup waits for its first halo from HBM, and a store / reload of the 33 MB
// intermediate tensor -- 32 such launches are 0.53 of the 0.84 ms frame.  Round 3's first attempt at this node
// broke even (35.2 vs 35.3 us) on 2-way bank-conflicted fragment reads of its 144-byte pixel pitch and was deleted; this one is
// built from the parts that have been measured since:
//   * conv3x3_wr.hip's construction: a wave owns ALL pixels of the tile x 16 output channels of BOTH convs; the activations are
//     64-byte rows (one 32-channel chunk of a pixel), XOR-swizzled, read as ds_read_b128 fragments that feed three MFMAs each
//     (conflict-free under the gfx950 lane grouping; conv3x3_dma.hip); the input halo comes by LDS-DMA into a double buffer;
//   * weights RESIDENT in registers: 2 x 18 fragments (144 VGPRs) per wave from the fragment-order copies (tg_pack_weights_frag),
//     loaded once per workgroup -- the workgroups are persistent over tiles;
//   * tile = 6 x 14 output pixels: the first conv runs on the 8 x 16 region the second one needs (one MFMA pixel tile per row), its
//     ReLU output goes to LDS as bf16 in the SAME swizzled layout (zero outside the image = the second conv's SAME padding), the
//     second conv reads it as the first one read the halo; the skip comes from the staged halo; 1.33 x the MACs of two launches
//     (region recompute), none of their HBM round trip (33 MB written + read per block) and half their boundaries.
// Arithmetic: bf16 products, fp32 accumulation, the intermediate rounded to bf16 once (as the two-launch path stores it), the
// result rounded once: equal to the two-launch path up to the summation order inside each conv.
#include "common.h"
#include <type_traits>

struct RtP {
  const void* x;        // [N,H,W,64] bf16
  const void* w1;       // fragment order [18][4][64][8] (tg_pack_weights_frag, dst_t) of the first conv
  const void* w2;       // ... of the second
  const float* b1;      // nullable
  const float* b2;      // nullable
  void* out;            // [N,H,W,64] bf16
  int N, H, W;
  int tiles_y, tiles_x, ntiles;
  unsigned bytes;
};

typedef unsigned int u32x4t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;

namespace {
constexpr unsigned RT_OOB = 0x80000000u;
constexpr int RT_OR = 6, RT_OC = 14;                 // output rows / columns of a tile
constexpr int RT_MR = RT_OR + 2, RT_MW = 16;         // intermediate region: 8 rows x 16 columns
constexpr int RT_HR = RT_MR + 2, RT_HW = 18;         // input halo: 10 rows x 18 columns
constexpr int RT_HALO = RT_HR * RT_HW;               // 180 pixels per 32-channel chunk
constexpr int RT_CINST = (RT_HALO * 4 + 63) / 64;    // 12 DMA instructions (1 KB) per chunk
constexpr int RT_ROUNDS = 2 * RT_CINST / 4;          // 6 rounds of 4 waves for both chunks
constexpr int RT_CB = RT_CINST * 1024;               // 12288 bytes per chunk region
constexpr int RT_HB = 2 * RT_CB;                     // 24576 bytes per halo buffer
constexpr int RT_MPX = RT_MR * RT_MW + 8;            // 136 positions (the last fragments read 2 positions past the region)
constexpr int RT_MB = RT_MPX * 64;                   // 8704 bytes per chunk region of the intermediate
static_assert(RT_CB % 512 == 0 && RT_MB % 512 == 0, "regions start on a multiple of 8 rows (swizzle period)");
template <int I, int N, typename F>
__device__ __forceinline__ void rt_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    rt_static_for<I + 1, N>(f);
  }
}
}  // namespace

__global__ __launch_bounds__(256, 2) void resblock_thr_kernel(RtP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 x RT_HB (halo double buffer) + 2 x RT_MB (intermediate)
  unsigned char* const mid = smem + 2 * RT_HB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fg = lane >> 4;
  if ((int)blockIdx.x >= p.ntiles) return;

  const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.bytes, 0x00020000);
  const auto rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.bytes, 0x00020000);
  const auto rsW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, 9 * 64 * 64 * 2, 0x00020000);
  const auto rsW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), 0, 9 * 64 * 64 * 2, 0x00020000);
  const auto rsB1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b1), 0, p.b1 ? 256 : 0, 0x00020000);
  const auto rsB2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b2), 0, p.b2 ? 256 : 0, 0x00020000);

  // ---- halo DMA of one tile: instruction inst = wave + 4 k covers chunk inst / 12, slots (inst % 12) * 64 + lane; slot S = halo
  //      position q = S / 4, 16-byte group (S % 4) ^ 2 * ((q >> 2) & 1) of the pixel's 32-channel chunk
  int hq[RT_ROUNDS], hch[RT_ROUNDS];                       // per round: halo position and byte offset inside the pixel
#pragma unroll
  for (int k = 0; k < RT_ROUNDS; ++k) {
    const int inst = wave + 4 * k, chunk = inst / RT_CINST;
    const int S = (inst - chunk * RT_CINST) * 64 + lane;
    const int q = S >> 2;
    hq[k] = q < RT_HALO ? q : -1;
    hch[k] = chunk * 64 + (((S & 3) ^ (((S >> 4) & 1) << 1)) << 4);
  }
  auto tile_origin = [&](int tile, int& n, int& y0, int& x0) {
    const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
    n = t1 / p.tiles_y;
    y0 = (t1 % p.tiles_y) * RT_OR;
    x0 = tx * RT_OC;
  };
  unsigned hoff[RT_ROUNDS];
  auto halo_setup = [&](int tile) {
    int n, y0, x0;
    tile_origin(tile, n, y0, x0);
    const bool tok = tile < p.ntiles;
#pragma unroll
    for (int k = 0; k < RT_ROUNDS; ++k) {
      const int dy = hq[k] / RT_HW, dx = hq[k] - RT_HW * dy;
      const int gy = y0 - 2 + dy, gx = x0 - 2 + dx;
      const bool ok = tok && hq[k] >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      hoff[k] = ok ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + hch[k]) : RT_OOB;
    }
  };
  auto dma_round = [&](int k, int buf) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_t*)(smem + buf * RT_HB + (wave + 4 * k) * 1024), 16, (int)hoff[k], 0, 0, 0);
  };

  // ---- prologue: the first tile's halo, then the first conv's weight fragments of this wave (fragment s = 2 tap + kstep: bytes
  //      [(4 s + wave) * 1024, + 1024)), the biases, and LAST the second conv's fragments: the first tile starts on the counted wait
  //      below while those 18 loads are still in flight (the compiler waits for them at their first MFMA)
  int tile = blockIdx.x;
  halo_setup(tile);
#pragma unroll
  for (int k = 0; k < RT_ROUNDS; ++k) dma_round(k, 0);
  const int wlane = wave * 1024 + lane * 16;
  u32x4t w1[18], w2[18];
#pragma unroll
  for (int s = 0; s < 18; ++s) w1[s] = __builtin_amdgcn_raw_buffer_load_b128(rsW1, wlane, s * 4096, 0);
  const u32x4t bq1 = __builtin_amdgcn_raw_buffer_load_b128(rsB1, (wave * 16 + fg * 4) * 4, 0, 0);
  const u32x4t bq2 = __builtin_amdgcn_raw_buffer_load_b128(rsB2, (wave * 16 + fg * 4) * 4, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 18; ++s) w2[s] = __builtin_amdgcn_raw_buffer_load_b128(rsW2, wlane, s * 4096, 0);
  __builtin_amdgcn_sched_barrier(0);
  const float bv1[4] = {__uint_as_float(bq1.x), __uint_as_float(bq1.y), __uint_as_float(bq1.z), __uint_as_float(bq1.w)};
  const float bv2[4] = {__uint_as_float(bq2.x), __uint_as_float(bq2.y), __uint_as_float(bq2.z), __uint_as_float(bq2.w)};

  // fragment bases: position q = K + frow, swizzle bit (q >> 2) & 1 -> eight lane bases cover every compile-time K
  int abase[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) abase[d] = frow * 64 + ((fg ^ (((((frow & 7) + d) >> 2) & 1) << 1)) << 4);
  // this lane's 8 bytes (channels 16 wave + 4 fg .. + 4) inside a 64-byte row of chunk wave / 2: group (wave & 1) * 2 + fg / 2
  const int cgrp = (wave & 1) * 2 + (fg >> 1), chalf = (fg & 1) * 8, cchunk = wave >> 1;

  int buf = 0;
  bool first = true;
  while (true) {
    int n, y0, x0;
    tile_origin(tile, n, y0, x0);
    const int ntile = tile + gridDim.x;
    // this wave's halo slots have landed: the queue holds (oldest first) the halo DMA and the previous tile's 6 stores
    if (first) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");        // (all but the second conv's 18 weight fragments)
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // A: halo landed; nobody reads the other halo buffer or `mid` any more
    first = false;
    halo_setup(ntile);                                      // (out-of-range lanes past the last tile: zeros into the idle buffer)
    const unsigned char* hb = smem + buf * RT_HB;

    // ---- first conv on the 8 x 16 region: fragment (kstep, kw, halo row hr) feeds rows m = hr - kh
    f32x4 acc[RT_MR];
#pragma unroll
    for (int m = 0; m < RT_MR; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      constexpr int NS = 2 * 3 * RT_HR;
      auto rd = [&](int s) {
        const int ks = s / (3 * RT_HR), r = s - ks * 3 * RT_HR, kw = r / RT_HR, hr = r - kw * RT_HR;
        const int K = hr * RT_HW + kw;
        return *reinterpret_cast<const u32x4t*>(hb + ks * RT_CB + abase[K & 7] + K * 64);
      };
      u32x4t F[3];
      F[0] = rd(0);
      F[1] = rd(1);
      rt_static_for<0, NS>([&](auto sv) {
        constexpr int s = decltype(sv)::value;
        constexpr int ks = s / (3 * RT_HR), r = s - ks * 3 * RT_HR, kw = r / RT_HR, hr = r - kw * RT_HR;
        if constexpr (s + 2 < NS) F[(s + 2) % 3] = rd(s + 2);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int m = hr - kh;
          if (m >= 0 && m < RT_MR)
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1[(kh * 3 + kw) * 2 + ks]),
                                                             __builtin_bit_cast(bf16x8, F[s % 3]), acc[m], 0, 0, 0);
        }
        if constexpr (s < RT_ROUNDS) {                       // the next tile's halo, one DMA round per fragment step
          __builtin_amdgcn_sched_barrier(0);
          dma_round(s, buf ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    }
    // bias + ReLU, zero outside the image, bf16 -> `mid` (position m * 16 + frow of chunk wave / 2)
#pragma unroll
    for (int m = 0; m < RT_MR; ++m) {
      const int gy = y0 - 1 + m, gx = x0 - 1 + frow;
      const bool inimg = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = inimg ? fmaxf(acc[m][r] + bv1[r], 0.f) : 0.f;
      u32x2t o;
      o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
      o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
      const int q = m * RT_MW + frow;
      *reinterpret_cast<u32x2t*>(mid + cchunk * RT_MB + q * 64 + ((cgrp ^ (((q >> 2) & 1) << 1)) << 4) + chalf) = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // B: the intermediate region is complete

    // ---- second conv on the 6 x 16 output rows (columns 14, 15 are discarded): fragment (kstep, kw, region row mr) feeds rows i = mr - kh
    f32x4 acc2[RT_OR];
#pragma unroll
    for (int i = 0; i < RT_OR; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      constexpr int NS = 2 * 3 * RT_MR;
      auto rd = [&](int s) {
        const int ks = s / (3 * RT_MR), r = s - ks * 3 * RT_MR, kw = r / RT_MR, mr = r - kw * RT_MR;
        const int K = mr * RT_MW + kw;
        return *reinterpret_cast<const u32x4t*>(mid + ks * RT_MB + abase[K & 7] + K * 64);
      };
      u32x4t F[3];
      F[0] = rd(0);
      F[1] = rd(1);
      rt_static_for<0, NS>([&](auto sv) {
        constexpr int s = decltype(sv)::value;
        constexpr int ks = s / (3 * RT_MR), r = s - ks * 3 * RT_MR, kw = r / RT_MR, mr = r - kw * RT_MR;
        if constexpr (s + 2 < NS) F[(s + 2) % 3] = rd(s + 2);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int i = mr - kh;
          if (i >= 0 && i < RT_OR)
            acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2[(kh * 3 + kw) * 2 + ks]),
                                                              __builtin_bit_cast(bf16x8, F[s % 3]), acc2[i], 0, 0, 0);
        }
      });
    }
    // bias + skip (the block input at the output pixel = halo position (i + 2, frow + 2)) -> HBM
#pragma unroll
    for (int i = 0; i < RT_OR; ++i) {
      const int q = (i + 2) * RT_HW + frow + 2;
      const u32x2t sk = *reinterpret_cast<const u32x2t*>(hb + cchunk * RT_CB + q * 64 + ((cgrp ^ (((q >> 2) & 1) << 1)) << 4) + chalf);
      float v[4];
      v[0] = acc2[i][0] + bv2[0] + __uint_as_float(sk.x << 16);
      v[1] = acc2[i][1] + bv2[1] + __uint_as_float(sk.x & 0xffff0000u);
      v[2] = acc2[i][2] + bv2[2] + __uint_as_float(sk.y << 16);
      v[3] = acc2[i][3] + bv2[3] + __uint_as_float(sk.y & 0xffff0000u);
      u32x2t o;
      o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
      o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
      const int gy = y0 + i, gx = x0 + frow;
      const bool ok = frow < RT_OC && gy < p.H && gx < p.W;
      __builtin_amdgcn_raw_buffer_store_b64(o, rsO, (int)(ok ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + (wave * 16 + fg * 4) * 2) : RT_OOB), 0, 0);
    }
    if (ntile >= p.ntiles) break;
    tile = ntile;
    buf ^= 1;
  }
}

extern "C" int tg_resblock_c64_thr(const void* x, const void* w1_frag, const float* b1, const void* w2_frag, const float* b2, void* out,
                                   int N, int H, int W, void* stream) {
  TG_CHECK_ARG(x && w1_frag && w2_frag && out && N > 0 && H > 0 && W > 0, "null pointer / empty tensor");
  TG_CHECK_ARG((((uintptr_t)x | (uintptr_t)w1_frag | (uintptr_t)w2_frag | (uintptr_t)out) & 15) == 0, "pointers must be 16-byte aligned");
  TG_CHECK_ARG(x != out, "not in place: neighbouring tiles read the block input after this tile's output is written");
  const int64_t bytes = (int64_t)N * H * W * 128;
  TG_CHECK_ARG(bytes < ((int64_t)1 << 31), "tensor too large for 32-bit buffer offsets");
  RtP p;
  p.x = x; p.w1 = w1_frag; p.w2 = w2_frag; p.b1 = b1; p.b2 = b2; p.out = out;
  p.N = N; p.H = H; p.W = W;
  p.tiles_y = (H + RT_OR - 1) / RT_OR;
  p.tiles_x = (W + RT_OC - 1) / RT_OC;
  const int64_t nt = (int64_t)N * p.tiles_y * p.tiles_x;
  TG_CHECK_ARG(nt < ((int64_t)1 << 28), "too many tiles");
  p.ntiles = (int)nt;
  p.bytes = (unsigned)bytes;
  constexpr int LDS = 2 * RT_HB + 2 * RT_MB;
  static bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_thr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    return true;
  }();
  (void)attr;
  // persistent: two workgroups per CU (66 KB of LDS, <= 256 registers), each over ntiles / grid tiles
  int grid = 2 * tg_num_cus();
  if (grid > p.ntiles) grid = p.ntiles;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double px = (double)N * H * W;
  TG_LAUNCH("resblock_thr", 2.0 * 2.0 * px * 64.0 * 576.0, px * 128.0 * 2 + 2.0 * 73728.0, resblock_thr_kernel, dim3(grid), dim3(256), LDS, st, p);
  TG_CHECK_LAUNCH();
}
