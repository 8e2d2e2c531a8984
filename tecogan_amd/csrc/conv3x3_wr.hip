// 3x3 stride-1 SAME convolution for the wide FROZEN layers (Cin a multiple of 32, >= 64; bf16) with the weight operand
// streamed into REGISTERS from a fragment-order copy -- gfx950.
//
// Covers VGG-19 conv2_2 ... conv4_4 (reference lib/ops.py:319-327 through lib/Teco.py:5-24,174-178: 76 % of the perceptual
// loss's MACs, which are 72 % of the TecoGAN step's), conv5_x as packed tiles (two 8 x 8 images per tile, see WrGeo / KS below)
// and -- taps mirrored at pack time -- their input gradients.
//
// Why a fourth 3x3 kernel.  conv3x3_dma.hip brings BOTH operands of a 32-channel stage in by LDS-DMA: 20 KB of halo and
// 36 KB of weight panel per 576 MFMAs per CU, and its stage trace (profiles/r02m_trace_dma.txt, DESIGN lesson 9) shows the
// stage waiting for that stream: the LDS-DMA path moves ~15 B/clk/CU (9 TB/s chip-wide) where the MFMA block would need 25.
// A wave that loads 16-byte vectors into REGISTERS from whole cache lines streams 43 B/clk/CU (lesson 17, resblock_lat.hip).
// The VGG weights never change (frozen network, lib/Teco.py:421: only generator / fnet / discriminator variables train), so
// a fragment-order copy costs one pack per process.  Here:
//   * a workgroup = 4 waves owns a 16 x TH pixel tile (TH = 16 or 8) x 64 output channels; a WAVE owns ALL pixels of the
//     tile x 16 output channels: its nine weight fragments of a 32-channel stage (9 KB, contiguous in the copy = 72 whole
//     lines) go global -> registers, each fragment re-requested for the NEXT stage right after its last MFMA of this one
//     (a whole stage of prefetch distance, no second register set);
//   * only the halo ((TH+2) x 18 pixels x 64 bytes = 20 KB) goes through LDS, by LDS-DMA into a double buffer, with the
//     swizzled 64-byte rows of conv3x3_dma.hip (conflict-free ds_read_b128 under the gfx950 lane grouping): 36 % of the
//     bytes the DMA path carried; every halo fragment is read once per (row, kw) and feeds three MFMAs (54 reads per 144);
//   * 48 KB of LDS and 152 registers (16-row tiles; 24 KB / 113 for 8-row tiles): three / four workgroups per CU whose stages
//     drift apart, so one's barrier / DMA wait runs under the others' MFMAs (lesson 25); beside the latency-bound recurrent
//     chain (TG_CONV_COEXIST) the launch caps itself at two per CU so that a chain workgroup always finds room (see the host code);
//   * one workgroup per (tile, channel block), no persistent loop: the hardware dispatcher balances the tail, units are
//     numbered so that an XCD owns a contiguous range (the four channel blocks of a tile run concurrently on ONE L2);
//   * accumulation order per output element = conv3x3_dma.hip's (chunk, kw, kh ascending; same MFMA, same operand roles):
//     results are BIT-IDENTICAL to tg_conv_forward's (tests/test_kernels_gpu.py holds that).
#include "common.h"
#include <mutex>
#include <type_traits>

struct ConvWrP {
  const void* in;
  const void* wf;     // fragment order: [Cout/16][Cin/32][9][64 lanes][8 bf16]
  const float* bias;
  const void* res;
  const void* aux;
  void* out;
  int N, H, W, Cin, Cout;
  float nslope;       // none: 1, ReLU: 0, LeakyReLU: alpha  -> act(v) = max(v, v*nslope)
  float mslope;       // act-grad mask: aux > 0 ? 1 : mslope
  int tiles_y, tiles_x, nblk, nunits, u8;
  unsigned in_bytes, w_bytes, out_bytes;
  int lds_floor;      // host only: dynamic LDS request floor (residency cap beside the latency-bound chain, TG_CONV_COEXIST)
};

typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_w;

namespace {
constexpr unsigned WR_OOB = 0x80000000u;
#ifndef WR_DIST
#define WR_DIST 2      // LDS fragment prefetch distance (fragments ahead of their MFMAs); 3 / 4 measured within 1.4 % (profiles/r05m_ab.txt)
#endif
// PK = images per tile row.  PK = 1: a 16 x TH tile of a larger image, (TH + 2) x 18 halo.  PK = 2 (PACKED tiles, images of exactly
// 8 x 8 pixels: VGG conv5): the 8 x 16 output pixels are two whole images side by side, each with its own zero border in a
// 10 x 20 halo; output pixel (r, c) reads halo (r + kh, c + 2 (c / 8) + kw) -- a per-lane column shift on the fragment base.
template <int TH, int PK> struct WrGeo {
  static_assert(PK == 1 || (PK == 2 && TH == 8), "packed tiles: two 8 x 8 images");
  static constexpr int HW = PK == 1 ? 18 : 20;                       // halo row pitch in pixels
  static constexpr int HR = TH + 2;                                  // halo rows
  static constexpr int HALO = HR * HW;                               // 324 / 180 / 200 halo pixels
  static constexpr int INST = (HALO * 4 + 63) / 64;                  // 21 / 12 / 13 wave-wide DMA instructions (1 KB each)
  static constexpr int ROUNDS = (INST + 3) / 4;                      // 6 / 3 / 4 rounds of 4 waves
  static constexpr int BYTES = ROUNDS * 4 * 1024;                    // 24576 / 12288 / 16384 per buffer: every wave issues every round
                                                                     // (the slots past the halo take out-of-range lanes = zeros), no branch
};
template <int I, int N, typename F>
__device__ __forceinline__ void wr_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    wr_static_for<I + 1, N>(f);
  }
}
}  // namespace

// KS = K split: the workgroup has 4 KS waves; wave (kpart, cg) accumulates input-channel chunks [kpart nchunk / KS, ...) for channel
// group cg from its kpart's OWN halo double buffer, the partial sums meet in LDS at the end (fixed order: deterministic).  For
// launches with few pixels and many channels (conv5: 3072 pixels x 512 x 4608) it is the only parallelism left: 192 units of 4
// waves would put one wave on three quarters of the SIMDs and none on the rest.
template <bool HAS_RES, bool HAS_AUX, int TH, int PK, int KS>
__global__ __launch_bounds__(256 * KS, KS == 1 ? 3 : (KS == 2 ? 2 : 1)) void conv3x3_wr_kernel(ConvWrP p) {
  using G = WrGeo<TH, PK>;
  constexpr int HW = G::HW, HR = G::HR, ROUNDS = G::ROUNDS, HB = G::BYTES;
  constexpr int NS = 3 * HR;                                          // halo fragments of a stage: s = kw * HR + hr
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // KS x 2 x HB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kpart = KS == 1 ? 0 : wave >> 2, cg = wave & 3;
  const int frow = lane & 15, fg = lane >> 4;

  // unit -> (tile, channel block): XCD x (linear workgroup id % 8) owns units [x u8, (x + 1) u8)
  const int lin = blockIdx.x;
  const int u = (lin & 7) * p.u8 + (lin >> 3);
  if (u >= p.nunits) return;
  const int tile = u / p.nblk, blk = u - tile * p.nblk;
  const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
  const int ty = t1 % p.tiles_y;
  const int n = PK == 1 ? t1 / p.tiles_y : tile * 2;                  // packed: the tile's first image
  const int y0 = ty * TH - 1, x0 = tx * 16 - 1;
  const int g16 = blk * 4 + cg;                                       // this wave's group of 16 output channels
  const int cbase = g16 * 16;
  const int row_bytes = p.Cin * 2;
  const int nchunk = p.Cin >> 5;
  const int nc = nchunk / KS, c0 = kpart * nc;                        // this wave's stages: chunks [c0, c0 + nc)
  unsigned char* const kb = smem + kpart * 2 * HB;                    // this kpart's halo double buffer

  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wf), 0, (int)p.w_bytes, 0x00020000);
  const auto rsrcO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
  const auto rsrcR = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_RES ? p.res : p.out), 0, (int)p.out_bytes, 0x00020000);
  const auto rsrcM = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_AUX ? p.aux : p.out), 0, (int)p.out_bytes, 0x00020000);
  const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);

  // ---- LDS-DMA slot descriptors of this lane (stage-independent but for the channel chunk, which is the scalar offset).
  //      A wave-wide DMA instruction fills 64 consecutive 16-byte slots; slot S is halo pixel q = S / 4, position S % 4, and
  //      holds channel group (S % 4) ^ 2 * ((q >> 2) & 1) of that pixel's 32-channel chunk (conv3x3_dma.hip's swizzle).
  unsigned hoff[ROUNDS];
#pragma unroll
  for (int k = 0; k < ROUNDS; ++k) {
    const int S = (cg + 4 * k) * 64 + lane;
    const int q = S >> 2, ch = (S & 3) ^ (((S >> 4) & 1) << 1);
    const int dy = q / HW, dx = q - HW * dy;
    if constexpr (PK == 1) {
      const bool ok = q < G::HALO && (unsigned)(y0 + dy) < (unsigned)p.H && (unsigned)(x0 + dx) < (unsigned)p.W;
      hoff[k] = ok ? (unsigned)(((n * p.H + y0 + dy) * p.W + x0 + dx) * row_bytes + ch * 16) : WR_OOB;
    } else {                  // packed: image b of the pair, pixel (ry, rx) of that image or its zero border
      const int b = dx / 10, rx = dx - 10 * b - 1, ry = dy - 1;
      const bool ok = q < G::HALO && (unsigned)ry < 8u && (unsigned)rx < 8u && n + b < p.N;
      hoff[k] = ok ? (unsigned)((((n + b) * 8 + ry) * 8 + rx) * row_bytes + ch * 16) : WR_OOB;
    }
  }
  auto dma_round = [&](int k, int chunk, int buf, bool live) {
    const int inst = cg + 4 * k;                                      // wave-uniform
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void_w*)(kb + buf * HB + inst * 1024), 16,
                                             (int)(live ? hoff[k] : WR_OOB), chunk * 64, 0, 0);
  };

  // ---- prologue: bias, the first stage's halo, the first stage's nine weight fragments (this order: the counted wait
  //      at the top of a stage relies on the nine weight loads being the YOUNGEST vector-memory operations of the wave)
  const u32x4w bq = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (cbase + fg * 4) * 4, 0, 0);
#pragma unroll
  for (int k = 0; k < ROUNDS; ++k) dma_round(k, c0, 0, true);
  // weight fragment (chunk c, tap t) of this wave: bytes [((g16 * nchunk + c) * 9 + t) * 1024, + 1024); lane l holds
  // w[t][cbase + l % 16][32 c + 8 (l / 16) .. + 8]
  const int wlane = lane * 16;
  int wsoff = (g16 * nchunk + c0) * 9216;                             // scalar: this stage's nine fragments
  u32x4w wf[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int t = (j % 3) * 3 + j / 3;                                // issue order = consumption order: kw outer, kh inner
    wf[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, wlane, wsoff + t * 1024, 0);
  }
  __builtin_amdgcn_sched_barrier(0);

  // halo fragment (row hr, tap column kw): pixels q = K + Q0 with K = HW hr + kw (compile time) and Q0 = the lane's column
  // (packed: + 2 border columns once past the first image).  The swizzle bit (q >> 2) & 1 depends only on (Q0 + K) mod 8:
  // eight lane bases cover every K, the read is base[K & 7] + 64 K.
  const int Q0 = PK == 1 ? frow : frow + 2 * (frow >> 3);
  int abase[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) abase[d] = Q0 * 64 + ((fg ^ (((((Q0 & 7) + d) >> 2) & 1) << 1)) << 4);

  f32x4 acc[TH];
#pragma unroll
  for (int i = 0; i < TH; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c = 0; c < nc; ++c) {
    const int buf = c & 1;
    // This wave's DMA slots of the stage have landed: the vector-memory queue retires in order and holds (oldest first) the
    // stage's DMA and the stage's nine weight loads, so a counted wait covers the DMA without draining the weight stream.
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // every wave's slots landed; nobody still reads the other buffer
    const bool has_next = c + 1 < nc;
    const unsigned wl_next = has_next ? (unsigned)wlane : WR_OOB;     // past the last stage the loads read zeros (no branch)
    wsoff += 9216;
    const unsigned char* sb = kb + buf * HB;
    auto rd = [&](int s) {
      const int kw = s / HR, hr = s - kw * HR;
      const int K = hr * HW + kw;
      return *reinterpret_cast<const u32x4w*>(sb + abase[K & 7] + K * 64);
    };
    u32x4w F[WR_DIST + 1];
#pragma unroll
    for (int k = 0; k < WR_DIST; ++k) F[k] = rd(k);
    wr_static_for<0, NS>([&](auto sv) {
      constexpr int s = decltype(sv)::value;
      constexpr int kw = s / HR, hr = s - kw * HR;
      if constexpr (s + WR_DIST < NS) F[(s + WR_DIST) % (WR_DIST + 1)] = rd(s + WR_DIST);   // WR_DIST fragments ahead of the MFMAs that use them
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int i = hr - kh;
        if (i >= 0 && i < TH)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[kh * 3 + kw]),
                                                           __builtin_bit_cast(bf16x8, F[s % (WR_DIST + 1)]), acc[i], 0, 0, 0);
      }
      // the next stage's halo: one DMA round after each of the first fragments' MFMAs (an LDS-DMA instruction takes 60-180
      // cycles to issue: back to back they would stall this wave's MFMA stream)
      if constexpr (s < ROUNDS) {
        __builtin_amdgcn_sched_barrier(0);
        dma_round(s, c0 + c + 1, buf ^ 1, has_next);
        __builtin_amdgcn_sched_barrier(0);
      }
      // tap (kh, kw) was used for the last time at hr = TH - 1 + kh: request the next stage's fragment into the same registers
      if constexpr (hr >= TH - 1) {
        constexpr int kh = hr - (TH - 1);
        __builtin_amdgcn_sched_barrier(0);
        wf[kh * 3 + kw] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, (int)wl_next, wsoff + (kh * 3 + kw) * 1024, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  }

  // ---- K split: partial sums of kparts 1 .. KS-1 -> LDS (over the halo buffers, once the last stage's trailing zero-fill DMA
  //      has landed and every wave has left its last stage), kpart 0 adds them in a fixed order and finishes the tile
  if constexpr (KS > 1) {
    static_assert((KS - 1) * 4 * TH * 1024 <= KS * 2 * HB, "partial sums must fit over the halo buffers");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kpart > 0) {
#pragma unroll
      for (int i = 0; i < TH; ++i)
        *reinterpret_cast<f32x4*>(smem + (((kpart - 1) * 4 + cg) * TH + i) * 1024 + lane * 16) = acc[i];
    }
    __syncthreads();
    if (kpart > 0) return;
#pragma unroll 1
    for (int k = 1; k < KS; ++k)
#pragma unroll
      for (int i = 0; i < TH; ++i) acc[i] += *reinterpret_cast<const f32x4*>(smem + (((k - 1) * 4 + cg) * TH + i) * 1024 + lane * 16);
  }

  // ---- epilogue in registers: accumulator r of lane (frow, fg) at row i = pixel (ty TH + i, tx 16 + frow) (packed: pixel
  //      (i, frow % 8) of image n + frow / 8), channel cbase + 4 fg + r
  const float bv[4] = {__uint_as_float(bq.x), __uint_as_float(bq.y), __uint_as_float(bq.z), __uint_as_float(bq.w)};
  const int x = PK == 1 ? tx * 16 + frow : (frow & 7), ybase = PK == 1 ? ty * TH : 0;
  const int ni = PK == 1 ? n : n + (frow >> 3);
  const int co = cbase + fg * 4;
  unsigned offs[TH];
#pragma unroll
  for (int i = 0; i < TH; ++i) {
    const int y = ybase + i;
    offs[i] = (y < p.H && x < p.W && ni < p.N) ? (unsigned)((((ni * p.H + y) * p.W + x) * p.Cout + co) * 2) : WR_OOB;
  }
  u32x2w rr[HAS_RES ? TH : 1], aa[HAS_AUX ? TH : 1];
  if constexpr (HAS_RES) {
#pragma unroll
    for (int i = 0; i < TH; ++i) rr[i] = __builtin_amdgcn_raw_buffer_load_b64(rsrcR, (int)offs[i], 0, 0);
  }
  if constexpr (HAS_AUX) {
#pragma unroll
    for (int i = 0; i < TH; ++i) aa[i] = __builtin_amdgcn_raw_buffer_load_b64(rsrcM, (int)offs[i], 0, 0);
  }
#pragma unroll
  for (int i = 0; i < TH; ++i) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = acc[i][r] + bv[r];
      v[r] = fmaxf(v[r], v[r] * p.nslope);
    }
    if constexpr (HAS_RES) {
      v[0] += __uint_as_float(rr[i].x << 16);
      v[1] += __uint_as_float(rr[i].x & 0xffff0000u);
      v[2] += __uint_as_float(rr[i].y << 16);
      v[3] += __uint_as_float(rr[i].y & 0xffff0000u);
    }
    if constexpr (HAS_AUX) {
      v[0] *= __uint_as_float(aa[i].x << 16) > 0.f ? 1.f : p.mslope;
      v[1] *= __uint_as_float(aa[i].x & 0xffff0000u) > 0.f ? 1.f : p.mslope;
      v[2] *= __uint_as_float(aa[i].y << 16) > 0.f ? 1.f : p.mslope;
      v[3] *= __uint_as_float(aa[i].y & 0xffff0000u) > 0.f ? 1.f : p.mslope;
    }
    u32x2w o;
    o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    __builtin_amdgcn_raw_buffer_store_b64(o, rsrcO, (int)offs[i], 0, 0);
  }
}

// ---- fragment-order copy of a [9][Cout][Cin] bf16 operand (the W^T copy for the forward conv, the natural copy for the input
//      gradient -- whose taps are mirrored HERE, flip = 1, so the kernel never sees a direction)
__global__ void pack_wide_frag_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int Cout, int Cin, int flip, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;         // one 16-byte vector of the copy
  if (i >= total) return;
  const int lane = (int)(i & 63);
  long r = i >> 6;
  const int t = (int)(r % 9);
  r /= 9;
  const int nchunk = Cin >> 5;
  const int c = (int)(r % nchunk), g = (int)(r / nchunk);
  const int frow = lane & 15, fg = lane >> 4;
  const int wt = flip ? 8 - t : t;
  dst[i] = src[(((long)wt * Cout + g * 16 + frow) * Cin + c * 32 + fg * 8) >> 3];
}

extern "C" int tg_pack_wide_frag(const void* w, void* w_frag, int Cout, int Cin, int flip, void* stream) {
  TG_CHECK_ARG(w && w_frag && Cout > 0 && Cout % 16 == 0 && Cin > 0 && Cin % 32 == 0, "bf16 [9][Cout][Cin] with Cout % 16 == 0, Cin % 32 == 0");
  TG_CHECK_ARG((((uintptr_t)w | (uintptr_t)w_frag) & 15) == 0, "pointers must be 16-byte aligned");
  const long total = (long)9 * Cout * Cin / 8;
  hipLaunchKernelGGL(pack_wide_frag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint4*>(w), static_cast<uint4*>(w_frag), Cout, Cin, flip, total);
  TG_CHECK_LAUNCH();
}

template <bool HAS_RES, bool HAS_AUX, int TH, int PK, int KS>
static void launch_wr(const ConvWrP& p, hipStream_t st) {
  auto kern = conv3x3_wr_kernel<HAS_RES, HAS_AUX, TH, PK, KS>;
  constexpr int LDS = KS * 2 * WrGeo<TH, PK>::BYTES;
  if constexpr (LDS > 65536) {
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
  }
  static const char* const pname =
      PK == 2 ? (HAS_AUX ? "conv3x3_wr_pack2<aux>" : "conv3x3_wr_pack2<>")
      : TH == 16 ? (HAS_RES ? (HAS_AUX ? "conv3x3_wr<res,aux>" : "conv3x3_wr<res>") : (HAS_AUX ? "conv3x3_wr<aux>" : "conv3x3_wr<>"))
                 : (HAS_RES ? (HAS_AUX ? "conv3x3_wr8<res,aux>" : "conv3x3_wr8<res>") : (HAS_AUX ? "conv3x3_wr8<aux>" : "conv3x3_wr8<>"));
  const double px = (double)p.N * p.H * p.W;
  const int lds = LDS > p.lds_floor ? LDS : p.lds_floor;
  TG_LAUNCH(pname, 2.0 * px * p.Cout * 9.0 * p.Cin,
            px * (p.Cin * 2.0 + p.Cout * 2.0 * (1 + HAS_RES + HAS_AUX)) + 18.0 * p.Cin * p.Cout, kern, dim3(8 * p.u8), dim3(256 * KS),
            lds, st, p);
}

template <int TH, int PK, int KS>
static void launch_wr_th(const ConvWrP& p, bool res, bool aux, hipStream_t st) {
  if constexpr (PK == 2) {                  // packed tiles: the residual form is not instantiated (no caller: VGG conv5 has none)
    if (aux) launch_wr<false, true, TH, PK, KS>(p, st);
    else launch_wr<false, false, TH, PK, KS>(p, st);
  } else {
    if (res && aux) launch_wr<true, true, TH, PK, KS>(p, st);
    else if (res) launch_wr<true, false, TH, PK, KS>(p, st);
    else if (aux) launch_wr<false, true, TH, PK, KS>(p, st);
    else launch_wr<false, false, TH, PK, KS>(p, st);
  }
}

extern "C" int tg_conv3x3_wide_frag(const tg_conv_desc* d, const void* in, const void* w_frag, const float* bias, const void* res,
                                    const void* aux, void* out, int tile_rows, int ksplit, void* stream) {
  TG_CHECK_ARG(d && in && w_frag && out, "null pointer");
  TG_CHECK_ARG(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 && d->Hin == d->Hout && d->Win == d->Wout,
               "3x3 stride-1 SAME convolutions only (either direction: the fragment-order copy carries the tap order)");
  TG_CHECK_ARG(d->in_dtype == TG_BF16 && d->out_dtype == TG_BF16, "bf16 tensors only");
  TG_CHECK_ARG(d->Cin % 32 == 0 && d->Cin >= 64 && d->Cout % 64 == 0, "Cin % 32 == 0, Cin >= 64, Cout % 64 == 0");
  TG_CHECK_ARG(d->act < TG_ACT_TANH, "epilogue activations: none / ReLU / LeakyReLU");
  TG_CHECK_ARG(tile_rows == 0 || tile_rows == 8 || tile_rows == 16, "tile_rows: 0 (auto), 8 or 16");
  TG_CHECK_ARG(ksplit == 0 || ksplit == 1 || ksplit == 2 || ksplit == 4, "ksplit: 0 (auto), 1, 2 or 4");
  TG_CHECK_ARG((((uintptr_t)in | (uintptr_t)w_frag | (uintptr_t)out | (uintptr_t)res | (uintptr_t)aux) & 15) == 0,
               "pointers must be 16-byte aligned");
  const int64_t px = (int64_t)d->N * d->Hin * d->Win;
  const int64_t in_bytes = px * d->Cin * 2, out_bytes = px * d->Cout * 2, w_bytes = (int64_t)9 * d->Cout * d->Cin * 2;
  TG_CHECK_ARG(in_bytes < ((int64_t)1 << 31) && out_bytes < ((int64_t)1 << 31), "tensor too large for 32-bit buffer offsets");
  ConvWrP p;
  p.in = in; p.wf = w_frag; p.bias = bias; p.res = res; p.aux = aux; p.out = out;
  p.N = d->N; p.H = d->Hin; p.W = d->Win; p.Cin = d->Cin; p.Cout = d->Cout;
  p.nslope = d->act == TG_ACT_RELU ? 0.f : (d->act == TG_ACT_LRELU ? d->act_alpha : 1.f);
  p.mslope = d->mask_act == TG_ACT_RELU ? 0.f : (d->mask_act == TG_ACT_LRELU ? d->mask_alpha : 1.f);
  p.nblk = p.Cout / 64;
  p.tiles_x = (p.W + 15) / 16;
  const int nchunk = p.Cin / 32;
  const bool packed = p.H == 8 && p.W == 8;               // two whole images per tile (VGG conv5)
  TG_CHECK_ARG(!packed || !res, "packed tiles (8 x 8 images) take no residual operand");
  // tile height: 16 rows read 18 halo rows for 16 (8: 10 for 8), but a launch needs several units per CU for the dispatcher to
  // balance its tail: 8-row tiles below 4 units of 16 rows per CU
  int th = packed ? 8 : tile_rows;
  if (th == 0) {
    const int64_t u16 = (int64_t)p.N * ((p.H + 15) / 16) * p.tiles_x * p.nblk;
    th = u16 >= 4 * (int64_t)tg_num_cus() ? 16 : 8;
  }
  p.tiles_y = packed ? 1 : (p.H + th - 1) / th;
  const int64_t ntiles = packed ? ((int64_t)p.N + 1) / 2 : (int64_t)p.N * p.tiles_y * p.tiles_x;
  if (packed) p.tiles_x = 1;
  const int64_t nunits = ntiles * p.nblk;
  TG_CHECK_ARG(nunits < ((int64_t)1 << 28), "too many tiles");
  // K split: only when the launch cannot give every SIMD a wave otherwise (4 waves per unit), and at least two stages per part
  int ks = ksplit;
  if (ks == 0) {
    ks = 1;
    if (packed)
      while (ks < 4 && nunits * 4 * ks < 4 * (int64_t)tg_num_cus() && nchunk % (2 * ks) == 0 && nchunk / (2 * ks) >= 2) ks *= 2;
  }
  TG_CHECK_ARG(nchunk % ks == 0 && (ks == 1 || th == 8), "ksplit must divide Cin / 32 and needs 8-row tiles");
  p.nunits = (int)nunits;
  p.u8 = (int)((nunits + 7) / 8);
  p.in_bytes = (unsigned)in_bytes; p.w_bytes = (unsigned)w_bytes; p.out_bytes = (unsigned)out_bytes;
  // Beside the latency-bound recurrent chain (TG_CONV_COEXIST) the launch asks for 56 KB of LDS it does not use: TWO workgroups per
  // CU instead of three / four, so that a chain workgroup (22 KB, 132 registers) always finds room at once instead of waiting for
  // one of this kernel's 15-us units to retire.  Same box, alternating runs: TecoGAN step 8.70 / 8.67 -> 8.61 / 8.57 ms and
  // 8.62 / 8.60 -> 8.52 / 8.50 ms (bwd_b 2.47 -> 2.40); a cap at three (40 KB) or one (80 KB) gains nothing (profiles/r05m_ab.txt).
  p.lds_floor = (d->flags & TG_CONV_COEXIST) ? 56 * 1024 : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool r = res != nullptr, a = aux != nullptr;
  if (packed) {
    if (ks == 1) launch_wr_th<8, 2, 1>(p, r, a, st);
    else if (ks == 2) launch_wr_th<8, 2, 2>(p, r, a, st);
    else launch_wr_th<8, 2, 4>(p, r, a, st);
  } else if (th == 16) {
    launch_wr_th<16, 1, 1>(p, r, a, st);
  } else {
    if (ks == 1) launch_wr_th<8, 1, 1>(p, r, a, st);
    else if (ks == 2) launch_wr_th<8, 1, 2>(p, r, a, st);
    else launch_wr_th<8, 1, 4>(p, r, a, st);
  }
  TG_CHECK_LAUNCH();
}
