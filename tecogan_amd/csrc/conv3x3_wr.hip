// 3x3 stride-1 SAME convolution for the wide FROZEN layers (Cin a multiple of 32, >= 64; bf16) with the weight operand
// streamed into REGISTERS from a fragment-order copy -- gfx950.
//
// Covers VGG-19 conv2_2 ... conv4_4 (reference lib/ops.py:319-327 through lib/Teco.py:5-24,174-178: 76 % of the perceptual
// loss's MACs, which are 72 % of the TecoGAN step's) and -- taps mirrored at pack time -- their input gradients.
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
//   * 43 KB of LDS and <= 168 registers: three workgroups per CU whose stages drift apart, so one's barrier / DMA wait
//     runs under the others' MFMAs (lesson 25) and a chain workgroup still fits beside them;
//   * one workgroup per (tile, channel block), no persistent loop: the hardware dispatcher balances the tail, units are
//     numbered so that an XCD owns a contiguous range (the four channel blocks of a tile run concurrently on ONE L2);
//   * accumulation order per output element = conv3x3_dma.hip's (chunk, kw, kh ascending; same MFMA, same operand roles):
//     results are BIT-IDENTICAL to tg_conv_forward's (tests/test_kernels_gpu.py holds that).
#include "common.h"
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
};

typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_w;

namespace {
constexpr unsigned WR_OOB = 0x80000000u;
constexpr int WR_HW = 18;                                            // halo row pitch in pixels
template <int TH> struct WrGeo {
  static constexpr int HR = TH + 2;                                  // halo rows
  static constexpr int HALO = HR * WR_HW;                            // 324 / 180 halo pixels
  static constexpr int INST = (HALO * 4 + 63) / 64;                  // 21 / 12 wave-wide DMA instructions (1 KB each)
  static constexpr int ROUNDS = (INST + 3) / 4;                      // 6 / 3 rounds of 4 waves
  static constexpr int BYTES = ROUNDS * 4 * 1024;                    // 24576 / 12288 per buffer: every wave issues every round (the
                                                                     // slots past the halo take out-of-range lanes = zeros), no branch
};
template <int I, int N, typename F>
__device__ __forceinline__ void wr_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    wr_static_for<I + 1, N>(f);
  }
}
}  // namespace

template <bool HAS_RES, bool HAS_AUX, int TH>
__global__ __launch_bounds__(256, 3) void conv3x3_wr_kernel(ConvWrP p) {
  using G = WrGeo<TH>;
  constexpr int HR = G::HR, ROUNDS = G::ROUNDS, HB = G::BYTES;
  constexpr int NS = 3 * HR;                                          // halo fragments of a stage: s = kw * HR + hr
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x HB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fg = lane >> 4;

  // unit -> (tile, channel block): XCD x (linear workgroup id % 8) owns units [x u8, (x + 1) u8)
  const int lin = blockIdx.x;
  const int u = (lin & 7) * p.u8 + (lin >> 3);
  if (u >= p.nunits) return;
  const int tile = u / p.nblk, blk = u - tile * p.nblk;
  const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
  const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
  const int y0 = ty * TH - 1, x0 = tx * 16 - 1;
  const int g16 = blk * 4 + wave;                                     // this wave's group of 16 output channels
  const int cbase = g16 * 16;
  const int row_bytes = p.Cin * 2;
  const int nchunk = p.Cin >> 5;

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
    const int S = (wave + 4 * k) * 64 + lane;
    const int q = S >> 2, ch = (S & 3) ^ (((S >> 4) & 1) << 1);
    const int dy = q / WR_HW, dx = q - WR_HW * dy;
    const bool ok = q < G::HALO && (unsigned)(y0 + dy) < (unsigned)p.H && (unsigned)(x0 + dx) < (unsigned)p.W;
    hoff[k] = ok ? (unsigned)(((n * p.H + y0 + dy) * p.W + x0 + dx) * row_bytes + ch * 16) : WR_OOB;
  }
  auto dma_round = [&](int k, int chunk, int buf, bool live) {
    const int inst = wave + 4 * k;                                    // wave-uniform
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void_w*)(smem + buf * HB + inst * 1024), 16,
                                               (int)(live ? hoff[k] : WR_OOB), chunk * 64, 0, 0);
  };

  // ---- prologue: bias, the first stage's halo, the first stage's nine weight fragments (this order: the counted wait
  //      at the top of a stage relies on the nine weight loads being the YOUNGEST vector-memory operations of the wave)
  const u32x4w bq = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (cbase + fg * 4) * 4, 0, 0);
#pragma unroll
  for (int k = 0; k < ROUNDS; ++k) dma_round(k, 0, 0, true);
  // weight fragment (chunk c, tap t) of this wave: bytes [((g16 * nchunk + c) * 9 + t) * 1024, + 1024); lane l holds
  // w[t][cbase + l % 16][32 c + 8 (l / 16) .. + 8]
  const int wlane = lane * 16;
  int wsoff = g16 * nchunk * 9216;                                    // scalar: this stage's nine fragments
  u32x4w wf[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int t = (j % 3) * 3 + j / 3;                                // issue order = consumption order: kw outer, kh inner
    wf[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, wlane, wsoff + t * 1024, 0);
  }
  __builtin_amdgcn_sched_barrier(0);

  // halo fragment (row hr, tap column kw): pixels q = K + frow with K = 18 hr + kw (compile time).  The swizzle bit
  // (q >> 2) & 1 depends only on (frow + K) mod 8: eight lane bases cover every K, the read is base[K & 7] + 64 K.
  int abase[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) abase[d] = frow * 64 + ((fg ^ (((((frow & 7) + d) >> 2) & 1) << 1)) << 4);

  f32x4 acc[TH];
#pragma unroll
  for (int i = 0; i < TH; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int buf = chunk & 1;
    // This wave's DMA slots of the stage have landed: the vector-memory queue retires in order and holds (oldest first) the
    // stage's DMA and the stage's nine weight loads, so a counted wait covers the DMA without draining the weight stream.
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // every wave's slots landed; nobody still reads the other buffer
    const bool has_next = chunk + 1 < nchunk;
    const unsigned wl_next = has_next ? (unsigned)wlane : WR_OOB;     // past the last stage the loads read zeros (no branch)
    wsoff += 9216;
    const unsigned char* sb = smem + buf * HB;
    auto rd = [&](int s) {
      const int kw = s / HR, hr = s - kw * HR;
      const int K = hr * WR_HW + kw;
      return *reinterpret_cast<const u32x4w*>(sb + abase[K & 7] + K * 64);
    };
    u32x4w F[3];
    F[0] = rd(0);
    F[1] = rd(1);
    wr_static_for<0, NS>([&](auto sv) {
      constexpr int s = decltype(sv)::value;
      constexpr int kw = s / HR, hr = s - kw * HR;
      if constexpr (s + 2 < NS) F[(s + 2) % 3] = rd(s + 2);          // two fragments ahead of the MFMAs that use them
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int i = hr - kh;
        if (i >= 0 && i < TH)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[kh * 3 + kw]),
                                                           __builtin_bit_cast(bf16x8, F[s % 3]), acc[i], 0, 0, 0);
      }
      // the next stage's halo: one DMA round after each of the first fragments' MFMAs (an LDS-DMA instruction takes 60-180
      // cycles to issue: back to back they would stall this wave's MFMA stream)
      if constexpr (s < ROUNDS) {
        __builtin_amdgcn_sched_barrier(0);
        dma_round(s, chunk + 1, buf ^ 1, has_next);
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

  // ---- epilogue in registers: accumulator r of lane (frow, fg) at row i = pixel (ty TH + i, tx 16 + frow), channel
  //      cbase + 4 fg + r
  const float bv[4] = {__uint_as_float(bq.x), __uint_as_float(bq.y), __uint_as_float(bq.z), __uint_as_float(bq.w)};
  const int x = tx * 16 + frow, ybase = ty * TH;
  const int co = cbase + fg * 4;
  unsigned offs[TH];
#pragma unroll
  for (int i = 0; i < TH; ++i) {
    const int y = ybase + i;
    offs[i] = (y < p.H && x < p.W) ? (unsigned)((((n * p.H + y) * p.W + x) * p.Cout + co) * 2) : WR_OOB;
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

template <bool HAS_RES, bool HAS_AUX, int TH>
static void launch_wr(const ConvWrP& p, hipStream_t st) {
  auto kern = conv3x3_wr_kernel<HAS_RES, HAS_AUX, TH>;
  constexpr int LDS = 2 * WrGeo<TH>::BYTES;
  static const char* const pname =
      TH == 16 ? (HAS_RES ? (HAS_AUX ? "conv3x3_wr<res,aux>" : "conv3x3_wr<res>") : (HAS_AUX ? "conv3x3_wr<aux>" : "conv3x3_wr<>"))
               : (HAS_RES ? (HAS_AUX ? "conv3x3_wr8<res,aux>" : "conv3x3_wr8<res>") : (HAS_AUX ? "conv3x3_wr8<aux>" : "conv3x3_wr8<>"));
  const double px = (double)p.N * p.H * p.W;
  TG_LAUNCH(pname, 2.0 * px * p.Cout * 9.0 * p.Cin,
            px * (p.Cin * 2.0 + p.Cout * 2.0 * (1 + HAS_RES + HAS_AUX)) + 18.0 * p.Cin * p.Cout, kern, dim3(8 * p.u8), dim3(256),
            LDS, st, p);
}

template <int TH>
static void launch_wr_th(const ConvWrP& p, bool res, bool aux, hipStream_t st) {
  if (res && aux) launch_wr<true, true, TH>(p, st);
  else if (res) launch_wr<true, false, TH>(p, st);
  else if (aux) launch_wr<false, true, TH>(p, st);
  else launch_wr<false, false, TH>(p, st);
}

extern "C" int tg_conv3x3_wide_frag(const tg_conv_desc* d, const void* in, const void* w_frag, const float* bias, const void* res,
                                    const void* aux, void* out, int tile_rows, void* stream) {
  TG_CHECK_ARG(d && in && w_frag && out, "null pointer");
  TG_CHECK_ARG(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 && d->Hin == d->Hout && d->Win == d->Wout,
               "3x3 stride-1 SAME convolutions only (either direction: the fragment-order copy carries the tap order)");
  TG_CHECK_ARG(d->in_dtype == TG_BF16 && d->out_dtype == TG_BF16, "bf16 tensors only");
  TG_CHECK_ARG(d->Cin % 32 == 0 && d->Cin >= 64 && d->Cout % 64 == 0, "Cin % 32 == 0, Cin >= 64, Cout % 64 == 0");
  TG_CHECK_ARG(d->act < TG_ACT_TANH, "epilogue activations: none / ReLU / LeakyReLU");
  TG_CHECK_ARG(tile_rows == 0 || tile_rows == 8 || tile_rows == 16, "tile_rows: 0 (auto), 8 or 16");
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
  // tile height: 16 rows read 18 halo rows for 16 (8: 10 for 8), but a launch needs several units per CU for the dispatcher to
  // balance its tail: 8-row tiles below 4 units of 16 rows per CU
  int th = tile_rows;
  if (th == 0) {
    const int64_t u16 = (int64_t)p.N * ((p.H + 15) / 16) * p.tiles_x * p.nblk;
    th = (u16 >= 4 * (int64_t)tg_num_cus() || p.H <= 8) ? 16 : 8;
  }
  p.tiles_y = (p.H + th - 1) / th;
  const int64_t nunits = (int64_t)p.N * p.tiles_y * p.tiles_x * p.nblk;
  TG_CHECK_ARG(nunits < ((int64_t)1 << 28), "too many tiles");
  p.nunits = (int)nunits;
  p.u8 = (int)((nunits + 7) / 8);
  p.in_bytes = (unsigned)in_bytes; p.w_bytes = (unsigned)w_bytes; p.out_bytes = (unsigned)out_bytes;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (th == 16) launch_wr_th<16>(p, res != nullptr, aux != nullptr, st);
  else launch_wr_th<8>(p, res != nullptr, aux != nullptr, st);
  TG_CHECK_LAUNCH();
}
