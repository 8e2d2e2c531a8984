// Input-gradient chain of generator_F's HR tail -- the frame gradient through `(.)*2-1`, the output conv (64 -> 3) and the second
// transposed conv (k3 s2), reference lib/frvsr.py:73-87 under tf.gradients (lib/Teco.py:441-449) -- as ONE launch, for the latency
// regime of the training recurrence's BPTT (one launch per frame, on the step's critical path):
//
//     g_out = bf16(scale * d_frame), 3 channels zero-padded to 8             (was tg_concat2_pad,             4.6 + 1.6 us per frame)
//     g_t2  = bwd_data(output conv)(g_out) * relu'(t2)        [N,2H2,2W2,64]  (was conv3x3_c8<4>,             10.9 us)
//     g_t1  = bwd_data(conv_tran2)(g_t2)   * relu'(t1)        [N,H2,W2,64]    (was conv_igemm<2,2,4,2> gather, 21.2 us)
//
// (node costs: profiles/r04g_node_costs.txt).  g_out and g_t2 still go to HBM -- the weight gradients of the two layers read
// them after the BPTT -- but g_t2 is not read back: the strided gather of the transposed conv's gradient takes it from LDS.
//   * a workgroup (4 waves) owns a 4 x 8 tile of g_t1 (t1 resolution): 512 workgroups at [4,64,64];
//   * level 1 computes g_t2 on the 9 x 17 HR pixels the tile's gather touches: K packs the TAPS exactly as conv3x3_c8 does (one
//     v_mfma_f32_16x16x32_bf16 = 4 taps x 8 channels, 3 K-steps), so g_t2 is BIT-IDENTICAL to the two launches it replaces; the
//     waves split the PIXEL tiles (ten of 16 pixels) and each does all 64 channels, so every 128-byte row of t2 (the mask) is
//     requested by one wave only -- line requests are what bounds a CU (DESIGN lesson 17);
//   * level 2 is the k3 s2 gather over the LDS region: a wave owns 16 channels of g_t1 and both 16-pixel tiles; its 18 weight
//     fragments come straight into registers from the FRAGMENT-order copy (tg_pack_weights_frag) as one stream with a prefetch
//     distance, as in resblock_lat.hip;
//   * LDS: the g_t2 region with a 160-byte pixel pitch, 18-position rows and the 16-byte chunk index XORed by 2 * (column bit 2):
//     conflict-free for the stride-2 fragment reads of every tap under the gfx950 ds_read_b128 lane grouping (brute-force
//     search, tools/lds_layout_search.py); 29 KB in all, so the node fits beside a resident VGG workgroup (117 KB).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct HbP {
  const float* d_out;   // [N, 2H2, 2W2, 3] fp32: gradient w.r.t. the HR frame
  const void* w_out;    // [9][64][8] bf16: the output conv's HWIO weights, 3 output channels zero-padded to 8 ([tap][in][out])
  const void* t2;       // [N, 2H2, 2W2, 64] bf16: relu output of conv_tran2 (mask)
  const void* w_tr;     // conv_tran2's [tap][in][out] operand in fragment order (tg_pack_weights_frag, dst_t)
  const void* t1;       // [N, H2, W2, 64] bf16: relu output of conv_tran1 (mask)
  void* g_out;          // [N, 2H2, 2W2, 8] bf16
  void* g_t2;           // [N, 2H2, 2W2, 64] bf16
  void* g_t1;           // [N, H2, W2, 64] bf16
  int N, H2, W2;
  float scale;
  int tiles_i, tiles_j, ntiles;
  unsigned hr64_bytes, hr8_bytes, dout_bytes, t1_bytes;
  int prio;
};

typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3b __attribute__((ext_vector_type(3)));
typedef unsigned int u32x2b __attribute__((ext_vector_type(2)));

namespace {
constexpr int HB_TI = 4, HB_TJ = 8;                         // tile of g_t1
constexpr int HB_RH = 2 * HB_TI + 1, HB_RW = 2 * HB_TJ + 1;  // g_t2 region: 9 x 17 HR pixels
constexpr int HB_RP = 18, HB_P = 160;                        // region row pitch (positions), bytes per position
constexpr int HB_NPX = HB_RH * HB_RW;                        // 153
constexpr int HB_NT1 = (HB_NPX + 15) / 16;                   // 10 level-1 pixel tiles
constexpr int HB_GH = HB_RH + 2, HB_GW = HB_RW + 2, HB_GP = 20;   // g_out staging: 11 x 19 positions of 16 bytes
constexpr int HB_REGION = (HB_RH * HB_RP + 2) * HB_P;         // + a dump position for the lanes of the last partial pixel tile
constexpr unsigned HB_OOB = 0x80000000u;
constexpr int HB_DIST = 10;
}  // namespace

template <int I, int N, typename F>
__device__ __forceinline__ void hb_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    hb_static_for<I + 1, N>(f);
  }
}

// The k3 s2 gather over a staged 9 x 17 region (160-byte positions, 18-position rows, chunk index XORed by 2 * (column bit 2)):
// out[i, j, ci] = sum_{ky, kx, co} region[2 i + ky, 2 j + kx, co] * W[ky, kx, co, ci] for the 4 x 8 tile, this wave's 16 output
// channels.  lane = pixel (2 t + frow / 8, frow % 8) of pixel tile t; fragment (tap, kk): region position (2 i + ky, 2 j + kx), chunk
// (4 kk + fg) ^ 2 * bit 2 of the column -- which is (j >> 1) & 1 for kx < 2 and ((j + 1) >> 1) & 1 for kx = 2.  The first ISSUED
// weight fragments are already requested; the rest is requested here, one per step (prefetch distance ISSUED).
template <int ISSUED, typename RS>
__device__ __forceinline__ void hb_gather_level(const unsigned char* rs, const RS& rsW, int wlane, u32x4b (&wB)[18], int frow, int fg,
                                                f32x4 (&acc2)[2]) {
  const int pj = frow & 7;
  const unsigned char* rb = rs + ((2 * (frow >> 3)) * HB_RP + 2 * pj) * HB_P;
  int ch[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    ch[kk][0] = ((kk * 4 + fg) ^ (((pj >> 1) & 1) * 2)) << 4;
    ch[kk][1] = ((kk * 4 + fg) ^ ((((pj + 1) >> 1) & 1) * 2)) << 4;
  }
  auto rfrag = [&](int s, int t) {
    const int tap = s >> 1, kk = s & 1, ky = tap / 3, kx = tap % 3;
    return *reinterpret_cast<const uint4*>(rb + ((4 * t + ky) * HB_RP + kx) * HB_P + ch[kk][kx == 2 ? 1 : 0]);
  };
  acc2[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 bf[2], nbf[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) bf[t] = rfrag(0, t);
  hb_static_for<0, 18>([&](auto sv) {
    constexpr int s = decltype(sv)::value;
    if constexpr (s + ISSUED < 18) wB[s + ISSUED < 18 ? s + ISSUED : 0] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wlane, (s + ISSUED) * 4096, 0);
    if constexpr (s < 17) {
#pragma unroll
      for (int t = 0; t < 2; ++t) nbf[t] = rfrag(s + 1, t);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
      acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wB[s]), *reinterpret_cast<bf16x8*>(&bf[t]),
                                                        acc2[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) bf[t] = nbf[t];
  });
}

__global__ __launch_bounds__(256, 2) void hr_bwd_lat_kernel(HbP p) {
  __shared__ __attribute__((aligned(16))) unsigned char rs[HB_REGION];
  __shared__ __attribute__((aligned(16))) unsigned char gs[HB_GH * HB_GP * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fg = lane >> 4;
  if (p.prio) __builtin_amdgcn_s_setprio(3);

  int b = blockIdx.x;
  if ((p.ntiles & 7) == 0) b = (b & 7) * (p.ntiles >> 3) + (b >> 3);      // an XCD owns a contiguous range of tiles
  const int tj = b % p.tiles_j, tq = b / p.tiles_j;
  const int ti = tq % p.tiles_i, n = tq / p.tiles_i;
  const int i0 = ti * HB_TI, j0 = tj * HB_TJ, Y0 = 2 * i0, X0 = 2 * j0;
  const int Ho = 2 * p.H2, Wo = 2 * p.W2;

  const auto rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.d_out), 0, (int)p.dout_bytes, 0x00020000);
  const auto rsWo = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_out), 0, 9 * 64 * 16, 0x00020000);
  const auto rsT2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.t2), 0, (int)p.hr64_bytes, 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_tr), 0, 9 * 64 * 64 * 2, 0x00020000);
  const auto rsT1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.t1), 0, (int)p.t1_bytes, 0x00020000);
  const auto rsGo = __builtin_amdgcn_make_buffer_rsrc(p.g_out, 0, (int)p.hr8_bytes, 0x00020000);
  const auto rsG2 = __builtin_amdgcn_make_buffer_rsrc(p.g_t2, 0, (int)p.hr64_bytes, 0x00020000);
  const auto rsG1 = __builtin_amdgcn_make_buffer_rsrc(p.g_t1, 0, (int)p.t1_bytes, 0x00020000);

  // ---- global loads, in consumption order; none behind a branch ----------------------------------------------------------
  // (1) frame gradient on the 11 x 19 positions around the region (one position per thread)
  const int gp = min(tid, HB_GH * HB_GW - 1), gpy = gp / HB_GW, gpx = gp - gpy * HB_GW;
  const int gY = Y0 - 1 + gpy, gX = X0 - 1 + gpx;
  const bool g_in = tid < HB_GH * HB_GW && (unsigned)gY < (unsigned)Ho && (unsigned)gX < (unsigned)Wo;
  const int g_pix = (n * Ho + gY) * Wo + gX;
  const u32x3b dq = __builtin_amdgcn_raw_buffer_load_b96(rsD, (int)(g_in ? (unsigned)(g_pix * 12) : HB_OOB), 0, 0);
  // (2) output conv, input-gradient form: lane (frow, fg) of K-step kk and channel tile j holds w_out[8 - t][16 j + frow][0..7],
  //     t = 4 kk + fg the tap of slot fg (slots >= 9: zero weights) -- the operand packing of conv3x3_c8_kernel
  u32x4b wo[3][4];
#pragma unroll
  for (int kk = 0; kk < 3; ++kk)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = 4 * kk + fg;
      wo[kk][j] = __builtin_amdgcn_raw_buffer_load_b128(rsWo, (int)(t < 9 ? (unsigned)(((8 - t) * 64 + j * 16 + frow) * 16) : HB_OOB), 0, 0);
    }
  // (3) relu'(t2) for the pixel tiles of this wave (tiles wave, wave + 4, wave + 8): 4 channels x 4 channel tiles per pixel
  u32x2b msk[3][4];
  int l1y[3], l1x[3];
  bool l1ok[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int m = (wave + 4 * q) * 16 + frow;
    l1ok[q] = m < HB_NPX;
    const int mm = min(m, HB_NPX - 1);
    l1y[q] = mm / HB_RW;
    l1x[q] = mm - l1y[q] * HB_RW;
    const int Y = Y0 + l1y[q], X = X0 + l1x[q];
    const bool ok = l1ok[q] && Y < Ho && X < Wo;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      msk[q][j] = __builtin_amdgcn_raw_buffer_load_b64(rsT2, (int)(ok ? (unsigned)(((n * Ho + Y) * Wo + X) * 128 + (j * 16 + fg * 4) * 2) : HB_OOB), 0, 0);
  }
  // (4) the transposed conv's weight stream: lane's 16 bytes of step s = (tap, K-half) at [s][wave][lane]
  u32x4b wB[18];
  const int wlane = wave * 1024 + lane * 16;
#define HB_WISSUE(i)                                                                  \
  do {                                                                                \
    if constexpr ((i) < 18) wB[(i) < 18 ? (i) : 0] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wlane, (i) * 4096, 0); \
  } while (0)
  hb_static_for<0, HB_DIST>([&](auto i) { HB_WISSUE(decltype(i)::value); });
  // (5) relu'(t1) for the two pixel tiles of level 2: lane = pixel (2 t + frow / 8, frow % 8), channels 16 wave + 4 fg ..
  u32x2b mt1[2];
  unsigned o1[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int i = i0 + 2 * t + (frow >> 3), j = j0 + (frow & 7);
    const bool ok = i < p.H2 && j < p.W2;
    o1[t] = ok ? (unsigned)(((n * p.H2 + i) * p.W2 + j) * 128 + (wave * 16 + fg * 4) * 2) : HB_OOB;
    mt1[t] = __builtin_amdgcn_raw_buffer_load_b64(rsT1, (int)o1[t], 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- g_out = bf16(scale * d_frame) -> LDS (all positions) and HBM (the tile's own 8 x 16 HR pixels) ----------------------
  {
    const unsigned c0 = f2bf(__uint_as_float(dq.x) * p.scale), c1 = f2bf(__uint_as_float(dq.y) * p.scale),
                   c2 = f2bf(__uint_as_float(dq.z) * p.scale);
    const u32x4b v = {c0 | (c1 << 16), c2, 0u, 0u};
    if (tid < HB_GH * HB_GW) *reinterpret_cast<u32x4b*>(gs + (gpy * HB_GP + gpx) * 16) = v;
    const bool own = g_in && gpy >= 1 && gpy <= 2 * HB_TI && gpx >= 1 && gpx <= 2 * HB_TJ;
    __builtin_amdgcn_raw_buffer_store_b128(v, rsGo, (int)(own ? (unsigned)(g_pix * 16) : HB_OOB), 0, 0);
  }
  __syncthreads();

  // ---- level 1: g_t2 on the region.  K-step kk, slot fg <-> tap t = min(4 kk + fg, 8) (kh, kw): input position (y + kh, x + kw)
  int toff[3];
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    const int t = min(4 * kk + fg, 8), kh = t / 3, kw = t - 3 * kh;
    toff[kk] = (kh * HB_GP + kw) * 16;
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const unsigned char* gb = gs + (l1y[q] * HB_GP + l1x[q]) * 16;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      const uint4 bf = *reinterpret_cast<const uint4*>(gb + toff[kk]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wo[kk][j]), *reinterpret_cast<const bf16x8*>(&bf),
                                                         acc[j], 0, 0, 0);
    }
    // epilogue: * relu'(t2) (zero outside the image: t2 reads zeros there), bf16 -> LDS region, own pixels -> HBM
    const int Y = Y0 + l1y[q], X = X0 + l1x[q];
    const bool own = l1ok[q] && l1y[q] < 2 * HB_TI && l1x[q] < 2 * HB_TJ && Y < Ho && X < Wo;
    const int pos = l1ok[q] ? l1y[q] * HB_RP + l1x[q] : HB_RH * HB_RP;                 // (lanes past the region: the dump position)
    const int sw = ((l1x[q] >> 2) & 1) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4];
      const float a[4] = {__uint_as_float(msk[q][j].x << 16), __uint_as_float(msk[q][j].x & 0xffff0000u),
                          __uint_as_float(msk[q][j].y << 16), __uint_as_float(msk[q][j].y & 0xffff0000u)};
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = bf2f(f2bf(acc[j][r] + 0.f)) * (a[r] > 0.f ? 1.f : 0.f);
      u32x2b o;
      o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
      o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
      *reinterpret_cast<u32x2b*>(rs + pos * HB_P + (((2 * j + (fg >> 1)) ^ sw) << 4) + (fg & 1) * 8) = o;
      __builtin_amdgcn_raw_buffer_store_b64(o, rsG2, (int)(own ? (unsigned)(((n * Ho + Y) * Wo + X) * 128 + (j * 16 + fg * 4) * 2) : HB_OOB), 0, 0);
    }
    // the weight stream keeps flowing: one fragment per finished pixel tile ... (the rest inside level 2)
    if (q == 0) HB_WISSUE(HB_DIST);
    if (q == 1) HB_WISSUE(HB_DIST + 1);
    if (q == 2) HB_WISSUE(HB_DIST + 2);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();

  // ---- level 2: g_t1[i, j, ci] = sum_{ky, kx, co} g_t2[2 i + ky, 2 j + kx, co] * W[ky, kx, co, ci]  (gather, stride 2) ------------
  f32x4 acc2[2];
  hb_gather_level<HB_DIST + 3>(rs, rsW, wlane, wB, frow, fg, acc2);
#undef HB_WISSUE
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float a[4] = {__uint_as_float(mt1[t].x << 16), __uint_as_float(mt1[t].x & 0xffff0000u),
                        __uint_as_float(mt1[t].y << 16), __uint_as_float(mt1[t].y & 0xffff0000u)};
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc2[t][r] * (a[r] > 0.f ? 1.f : 0.f);
    u32x2b o;
    o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    __builtin_amdgcn_raw_buffer_store_b64(o, rsG1, (int)o1[t], 0, 0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same gather as its own launch: input gradient of a k3 s2 transposed conv (conv_tran1 in the BPTT: [B,64,64,64] -> [B,32,32,64],
// was conv_igemm gather, 9.6 us): the 9 x 17 region of the output gradient comes from HBM (whole 128-byte rows, cooperatively).
struct DbP {
  const void* dy;       // [N, 2H, 2W, 64] bf16
  const void* w_frag;   // [tap][in][out] operand in fragment order (tg_pack_weights_frag, dst_t)
  const void* aux;      // nullable [N,H,W,64]: result *= (aux > 0)
  void* dx;             // [N, H, W, 64] bf16
  int N, H, W;
  int tiles_i, tiles_j, ntiles;
  unsigned dy_bytes, dx_bytes;
  int prio;
};

template <bool HAS_AUX>
__global__ __launch_bounds__(256, 2) void deconv_bwd_lat_kernel(DbP p) {
  __shared__ __attribute__((aligned(16))) unsigned char rs[HB_REGION];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fg = lane >> 4;
  if (p.prio) __builtin_amdgcn_s_setprio(3);
  int b = blockIdx.x;
  if ((p.ntiles & 7) == 0) b = (b & 7) * (p.ntiles >> 3) + (b >> 3);
  const int tj = b % p.tiles_j, tq = b / p.tiles_j;
  const int ti = tq % p.tiles_i, n = tq / p.tiles_i;
  const int i0 = ti * HB_TI, j0 = tj * HB_TJ, Y0 = 2 * i0, X0 = 2 * j0;
  const int Ho = 2 * p.H, Wo = 2 * p.W;
  const auto rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_frag), 0, 9 * 64 * 64 * 2, 0x00020000);
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_AUX ? p.aux : p.dx), 0, (int)p.dx_bytes, 0x00020000);
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc(p.dx, 0, (int)p.dx_bytes, 0x00020000);
  constexpr int ITEMS = HB_NPX * 8, NL = (ITEMS + 255) / 256;            // 1224 16-byte items, 5 per thread
  u32x4b rr[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int item = tid + k * 256;
    const int pix = min(item >> 3, HB_NPX - 1), c = item & 7;
    const int y = pix / HB_RW, x = pix - y * HB_RW;
    const bool ok = item < ITEMS && Y0 + y < Ho && X0 + x < Wo;
    rr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)(ok ? (unsigned)(((n * Ho + Y0 + y) * Wo + X0 + x) * 128 + c * 16) : HB_OOB), 0, 0);
  }
  u32x4b wB[18];
  const int wlane = wave * 1024 + lane * 16;
  hb_static_for<0, HB_DIST + 3>([&](auto i) { wB[decltype(i)::value] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wlane, decltype(i)::value * 4096, 0); });
  u32x2b mk[2];
  unsigned o1[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int i = i0 + 2 * t + (frow >> 3), j = j0 + (frow & 7);
    const bool ok = i < p.H && j < p.W;
    o1[t] = ok ? (unsigned)(((n * p.H + i) * p.W + j) * 128 + (wave * 16 + fg * 4) * 2) : HB_OOB;
    if constexpr (HAS_AUX) mk[t] = __builtin_amdgcn_raw_buffer_load_b64(rsA, (int)o1[t], 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int item = tid + k * 256;
    const int pix = item >> 3, c = item & 7;
    const int y = pix / HB_RW, x = pix - y * HB_RW;
    if (item < ITEMS) *reinterpret_cast<u32x4b*>(rs + (y * HB_RP + x) * HB_P + ((c ^ (((x >> 2) & 1) * 2)) << 4)) = rr[k];
  }
  __syncthreads();
  f32x4 acc2[2];
  hb_gather_level<HB_DIST + 3>(rs, rsW, wlane, wB, frow, fg, acc2);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float v[4] = {acc2[t][0], acc2[t][1], acc2[t][2], acc2[t][3]};
    if constexpr (HAS_AUX) {
      const float a[4] = {__uint_as_float(mk[t].x << 16), __uint_as_float(mk[t].x & 0xffff0000u),
                          __uint_as_float(mk[t].y << 16), __uint_as_float(mk[t].y & 0xffff0000u)};
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= a[r] > 0.f ? 1.f : 0.f;
    }
    u32x2b o;
    o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    __builtin_amdgcn_raw_buffer_store_b64(o, rsX, (int)o1[t], 0, 0);
  }
}

// dx = bwd_data(conv2d_transpose k3 s2)(dy) [* relu'(aux)]: dy [N,2H,2W,64] bf16 -> dx [N,H,W,64] bf16; w_frag = the [tap][in][out]
// operand in fragment order (tg_pack_weights_frag, dst_t)
extern "C" int tg_deconv_lat_backward(const void* dy, const void* w_frag, const void* aux, void* dx, int N, int H, int W, void* stream) {
  TG_CHECK_ARG(dy && w_frag && dx && N > 0 && H > 0 && W > 0, "bad argument");
  TG_CHECK_ARG((((uintptr_t)dy | (uintptr_t)w_frag | (uintptr_t)aux | (uintptr_t)dx) & 15) == 0, "alignment");
  const int64_t px = (int64_t)N * H * W;
  TG_CHECK_ARG(px * 4 * 128 < ((int64_t)1 << 31), "tensor too large for 32-bit buffer offsets");
  DbP p;
  p.dy = dy; p.w_frag = w_frag; p.aux = aux; p.dx = dx; p.N = N; p.H = H; p.W = W;
  p.tiles_i = (H + HB_TI - 1) / HB_TI; p.tiles_j = (W + HB_TJ - 1) / HB_TJ;
  const int64_t nt = (int64_t)N * p.tiles_i * p.tiles_j;
  TG_CHECK_ARG(nt < ((int64_t)1 << 24), "too many tiles: this is the latency-regime kernel");
  p.ntiles = (int)nt;
  p.dy_bytes = (unsigned)(px * 4 * 128); p.dx_bytes = (unsigned)(px * 128);
  p.prio = 1;                                   // s_setprio 3 in the chain kernels (measured in round 2, see conv3x3.hip)
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double fl = 2.0 * px * 64 * 576, by = px * 128.0 * (5 + (aux != nullptr)) + 73728.0;
  if (aux) TG_LAUNCH("deconv_bwd_lat<aux>", fl, by, (deconv_bwd_lat_kernel<true>), dim3(p.ntiles), dim3(256), 0, st, p);
  else TG_LAUNCH("deconv_bwd_lat<>", fl, by, (deconv_bwd_lat_kernel<false>), dim3(p.ntiles), dim3(256), 0, st, p);
  TG_CHECK_LAUNCH();
}


// d_frame [N,2H2,2W2,3] fp32 -> g_out [.,8] bf16 = bf16(scale * d_frame) (zero-padded), g_t2 = bwd_data(output conv)(g_out) *
// relu'(t2), g_t1 = bwd_data(conv_tran2, k3 s2)(g_t2) * relu'(t1).  w_out: the output conv's HWIO weights with the outputs padded
// to 8 ([9][64][8]); w_tr_frag: conv_tran2's [tap][in][out] operand in fragment order (tg_pack_weights_frag, dst_t).
extern "C" int tg_hr_tail_backward(const float* d_frame, float scale, const void* w_out, const void* t2, const void* w_tr_frag,
                                   const void* t1, void* g_out, void* g_t2, void* g_t1, int N, int H2, int W2, void* stream) {
  TG_CHECK_ARG(d_frame && w_out && t2 && w_tr_frag && t1 && g_out && g_t2 && g_t1, "null pointer");
  TG_CHECK_ARG(N > 0 && H2 > 0 && W2 > 0, "bad shape");
  TG_CHECK_ARG((((uintptr_t)w_out | (uintptr_t)t2 | (uintptr_t)w_tr_frag | (uintptr_t)t1 | (uintptr_t)g_out | (uintptr_t)g_t2 | (uintptr_t)g_t1) & 15) == 0 &&
                   ((uintptr_t)d_frame & 3) == 0, "alignment");
  const int64_t hr = (int64_t)N * 4 * H2 * W2;
  TG_CHECK_ARG(hr * 128 < ((int64_t)1 << 31), "tensor too large for 32-bit buffer offsets");
  HbP p;
  p.d_out = d_frame; p.w_out = w_out; p.t2 = t2; p.w_tr = w_tr_frag; p.t1 = t1; p.g_out = g_out; p.g_t2 = g_t2; p.g_t1 = g_t1;
  p.N = N; p.H2 = H2; p.W2 = W2; p.scale = scale;
  p.tiles_i = (H2 + HB_TI - 1) / HB_TI; p.tiles_j = (W2 + HB_TJ - 1) / HB_TJ;
  const int64_t nt = (int64_t)N * p.tiles_i * p.tiles_j;
  TG_CHECK_ARG(nt < ((int64_t)1 << 24), "too many tiles: this is the latency-regime kernel");
  p.ntiles = (int)nt;
  p.hr64_bytes = (unsigned)(hr * 128); p.hr8_bytes = (unsigned)(hr * 16); p.dout_bytes = (unsigned)(hr * 12);
  p.t1_bytes = (unsigned)((int64_t)N * H2 * W2 * 128);
  p.prio = 1;                                   // s_setprio 3 in the chain kernels (measured in round 2, see conv3x3.hip)
  const double px2 = (double)hr, px1 = (double)N * H2 * W2;
  TG_LAUNCH("hr_bwd_lat", 2.0 * px2 * 64 * 27 + 2.0 * px1 * 64 * 576, px2 * (12 + 16 + 128 + 128) + px1 * 256 + 73728.0 + 9216.0,
            hr_bwd_lat_kernel, dim3(p.ntiles), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  TG_CHECK_LAUNCH();
}
