// Implicit-GEMM convolution engine for gfx950 (MFMA, wave64, LDS-staged NHWC panels).
//
// Replaces slim.conv2d / slim.conv2d_transpose (reference lib/ops.py:35-56) and their
// input gradients.  One kernel template covers
//   gather mode     : out[oy] = sum_k in[oy*s - pad + k] W[k]        (conv fwd, deconv bwd_data)
//   transposed mode : out[oy] = sum_{k == oy+pad (mod s)} in[(oy+pad-k)/s] W[k]
//                                                                    (conv bwd_data, deconv fwd)
// GEMM view: M = output pixels (of one stride-phase), N = Cout, K = taps x Cin.
// Stride-2 transposed convolutions are phase-decomposed (blockIdx.z = phase) so every
// MFMA multiplies real data only.  The im2col matrix is never materialised: each K-step
// is one (tap, 128-byte channel chunk) panel gathered straight from the NHWC tensor.
//
// Data path per K-step: 16-B global loads -> registers (prefetch of step t+1 overlaps the
// MFMAs of step t) -> LDS (144-B padded rows, double buffered, one barrier per step)
// -> ds_read_b128 fragments -> v_mfma_f32_16x16x4_f32 (exact fp32) or
// v_mfma_f32_16x16x32_bf16 -> fused epilogue (bias, activation, residual, act-grad mask).
//
// fp32 K-permutation: the 16x16x4 MFMA wants A[i][k=lane>>4]; we give lane group g the
// four k's {4g..4g+3} of a 16-wide chunk (one ds_read_b128) and issue 4 MFMAs, j-th using
// element j from both operands -- a bijection of k, so the sum is unchanged.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

// conv3x3.hip: halo-tile kernel for 3x3 stride-1 SAME convolutions (returns 1 when it took the launch)
int tg_conv3x3_try(const tg_conv_desc* d, const void* in, const void* weight, const float* bias, const void* res,
                   const void* aux, void* out, hipStream_t st);

struct ConvP {
  const void* in;
  const void* w;
  const float* bias;
  const void* res;
  const void* aux;
  void* out;
  int N, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, s, pt, pl, mode;
  int act;
  float act_alpha;
  int mask_act;
  float mask_alpha;
  int vec;  // Cin % (16B worth) == 0 -> 16-byte loads
  float nslope, mslope;  // act(v) = max(v, v*nslope); mask = aux > 0 ? 1 : mslope
  int direct_epi;        // per-lane fp32 epilogue: launches with a residual / LeakyReLU-mask operand (one rounding), and the fp32 path
  unsigned in_bytes, w_bytes;   // != 0: both operands < 2^31 bytes -> bounds-checked buffer loads (see load_vec)
  int sinv;                     // ceil(2^16 / stride)
};
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// BUF: operand fetch through bounds-checked buffer loads (16-byte aligned channel runs, operands < 2^31 bytes);
// !BUF: the general path (any channel count / alignment / size) with clamped addresses and selects.
template <typename TIn, typename TOut, int WAVES_M, int WAVES_N, int TM, int TN, bool BUF>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
  constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16;
  constexpr int EPV = 16 / (int)sizeof(TIn);  // elements per 16 bytes
  constexpr int BK = 8 * EPV;                 // 128 bytes of channels per K-step
  constexpr int ROWB = 144;                   // LDS row pitch (128 + 16 pad): 9*16 -> b128 conflict-light
  constexpr int APASS = BM / 32, BPASS = (BN + 31) / 32;
  constexpr bool F32 = sizeof(TIn) == 4;

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * ROWB + BM * 4];
  unsigned char* As = smem;
  unsigned char* Bs = smem + 2 * BM * ROWB;
  int* out_pix = reinterpret_cast<int*>(smem + 2 * (BM + BN) * ROWB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- phase geometry -----------------------------------------------------
  int oy0 = 0, ox0 = 0, ostep = 1, kh0 = 0, kw0 = 0, kstep = 1;
  if (p.mode == 1 && p.s > 1) {
    const int phy = blockIdx.z / p.s, phx = blockIdx.z % p.s;
    ostep = p.s;
    kstep = p.s;
    kh0 = phy;
    kw0 = phx;
    oy0 = ((phy - p.pt) % p.s + p.s) % p.s;
    ox0 = ((phx - p.pl) % p.s + p.s) % p.s;
  }
  const int Hq = (p.Hout - oy0 + ostep - 1) / ostep, Wq = (p.Wout - ox0 + ostep - 1) / ostep;
  const int Mq = p.N * Hq * Wq;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  if (m0 >= Mq) return;
  const int nkh = (p.KH - kh0 + kstep - 1) / kstep, nkw = (p.KW - kw0 + kstep - 1) / kstep;
  const int nchunk = (p.Cin + BK - 1) / BK;
  const int T = nkh * nkw * nchunk;

  // ---- per-thread gather rows ----------------------------------------------
  const int lrow = tid >> 3, lchunk = tid & 7;
  int a_n[APASS], a_oy[APASS], a_ox[APASS];
#pragma unroll
  for (int ps = 0; ps < APASS; ++ps) {
    const int m = m0 + ps * 32 + lrow;
    if (m < Mq) {
      const int qx = m % Wq, t = m / Wq;
      a_ox[ps] = ox0 + qx * ostep;
      a_oy[ps] = oy0 + (t % Hq) * ostep;
      a_n[ps] = t / Hq;
    } else {
      a_n[ps] = -1;
      a_oy[ps] = 0;
      a_ox[ps] = 0;
    }
  }
  for (int r = tid; r < BM; r += 256) {
    const int m = m0 + r;
    int o = -1;
    if (m < Mq) {
      const int qx = m % Wq, t = m / Wq;
      o = ((t / Hq) * p.Hout + oy0 + (t % Hq) * ostep) * p.Wout + ox0 + qx * ostep;
    }
    out_pix[r] = o;
  }

  const TIn* __restrict__ gin = static_cast<const TIn*>(p.in);
  const TIn* __restrict__ gw = static_cast<const TIn*>(p.w);

  uint4 ra[APASS], rb[BPASS];
  int ikh = 0, ikw = 0, ic = 0;  // counters of the step being LOADED

  // Fast path: buffer loads through a descriptor; a lane that must read zero (padding, tails, "no next step") gets an
  // out-of-range offset and the hardware returns 0 -- no select behind the load, so the prefetch issued ahead of the
  // MFMA block really stays in flight across it (a select in the issuing block made hipcc wait for the data at once).
  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  auto load_vec = [&](const TIn* base, bool is_a, int64_t off, int c, bool ok) -> uint4 {
    ok = ok & (c < p.Cin);
    if constexpr (BUF) {
      const unsigned boff = ok ? (unsigned)off * (unsigned)sizeof(TIn) : 0x80000000u;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(is_a ? rsrcA : rsrcB, (int)boff, 0, 0);
      return make_uint4(v.x, v.y, v.z, v.w);
    } else {
    if (p.vec) {
      // unconditional load from a clamped address + select: a branch per load would make hipcc wait
      // vmcnt(0) after every load and serialise the panel fetch
      uint4 v = *reinterpret_cast<const uint4*>(base + (ok ? off : 0));
      if (!ok) v = make_uint4(0, 0, 0, 0);
      return v;
    }
    TIn tmp[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const bool oke = ok && (c + e < p.Cin);
      const TIn t = base[oke ? off + e : 0];
      tmp[e] = oke ? t : (TIn)0;
    }
    return *reinterpret_cast<uint4*>(tmp);
    }
  };

  auto load_global = [&](bool live) {
    const int kh = kh0 + ikh * kstep, kw = kw0 + ikw * kstep;
    const int c = ic * BK + lchunk * EPV;
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      // branch-free (selects, bitwise ands): control flow between the loads would split the burst into basic blocks
      const bool fwd = p.mode == 0;
      const int ny = a_oy[ps] + p.pt - kh, nx = a_ox[ps] + p.pl - kw;
      // x / s as (x * ceil(2^16 / s)) >> 16: exact for x < 16384 (image extents), and cheap enough to compute on both
      // paths; blended with a bit mask (hipcc turns a uniform ?: into a branch, which would end the basic block)
      const int fm = fwd ? -1 : 0;
      const int iy = ((a_oy[ps] * p.s - p.pt + kh) & fm) | (((max(ny, 0) * p.sinv) >> 16) & ~fm);
      const int ix = ((a_ox[ps] * p.s - p.pl + kw) & fm) | (((max(nx, 0) * p.sinv) >> 16) & ~fm);
      const bool ok = live & (a_n[ps] >= 0) & (fwd | ((ny >= 0) & (nx >= 0))) & ((unsigned)iy < (unsigned)p.Hin) &
                      ((unsigned)ix < (unsigned)p.Win);
      const int64_t off = ((int64_t)(a_n[ps] * p.Hin + iy) * p.Win + ix) * p.Cin + c;
      ra[ps] = load_vec(gin, true, off, c, ok);
    }
#pragma unroll
    for (int ps = 0; ps < BPASS; ++ps) {
      const int row = ps * 32 + lrow;
      const bool ok = live & (row < BN) & ((n0 + row) < p.Cout);
      const int64_t off = ((int64_t)(kh * p.KW + kw) * p.Cout + n0 + row) * p.Cin + c;
      rb[ps] = load_vec(gw, false, off, c, ok);
    }
    if (++ic == nchunk) {
      ic = 0;
      if (++ikw == nkw) {
        ikw = 0;
        ++ikh;
      }
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps)
      *reinterpret_cast<uint4*>(As + (buf * BM + ps * 32 + lrow) * ROWB + lchunk * 16) = ra[ps];
#pragma unroll
    for (int ps = 0; ps < BPASS; ++ps) {
      const int row = ps * 32 + lrow;
      if (row < BN) *reinterpret_cast<uint4*>(Bs + (buf * BN + row) * ROWB + lchunk * 16) = rb[ps];
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_global(true);
  store_lds(0);
  __syncthreads();

  const int frow = lane & 15, fg = lane >> 4;
  for (int step = 0; step < T; ++step) {
    const int buf = step & 1;
    const bool more = step + 1 < T;
    if constexpr (BUF) load_global(more);    // unconditional: one straight-line block, exact vmcnt bookkeeping
    else if (more) load_global(true);
    const unsigned char* Ab = As + (buf * BM + wm * TM * 16 + frow) * ROWB + fg * 16;
    const unsigned char* Bb = Bs + (buf * BN + wn * TN * 16 + frow) * ROWB + fg * 16;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(Ab + i * 16 * ROWB + kk * 64);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const uint4*>(Bb + j * 16 * ROWB + kk * 64);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (F32) {
            const float* a = reinterpret_cast<const float*>(&af[i]);
            const float* b = reinterpret_cast<const float*>(&bfr[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&af[i]),
                                                                *reinterpret_cast<bf16x8*>(&bfr[j]),
                                                                acc[i][j], 0, 0, 0);
          }
        }
    }
    if (more) store_lds(buf ^ 1);
    __syncthreads();
  }

  // ---- fused epilogue -------------------------------------------------------
  // 32-bit offsets; none/ReLU/LeakyReLU as the branch-free max(v, v*slope); tanh/sigmoid and partial tiles
  // take separate uniformly-selected copies (no per-element branches on the common path).
  TOut* __restrict__ gout = static_cast<TOut*>(p.out);
  const TOut* __restrict__ gres = static_cast<const TOut*>(p.res);
  const TOut* __restrict__ gaux = static_cast<const TOut*>(p.aux);
  const bool has_res = gres != nullptr, has_aux = gaux != nullptr;
  const int col0 = n0 + wn * TN * 16 + frow;
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bv[j] = (p.bias && col0 + j * 16 < p.Cout) ? p.bias[col0 + j * 16] : 0.f;
  const bool slow = p.act >= TG_ACT_TANH || (has_aux && p.mask_act != TG_ACT_RELU && p.mask_act != TG_ACT_LRELU);
  if constexpr (sizeof(TOut) == 2) {
    if ((p.Cout & 7) == 0 && !slow && !p.direct_epi) {
      // bf16 outputs: stage the activated tile in LDS (the A panels are idle after the K loop) and move 16-byte
      // rows; residual / mask operands arrive as vector loads (see conv3x3.hip for the measured effect).
      u16* stage = reinterpret_cast<u16*>(As);
      constexpr int SP = 72;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pl = ((wm * TM + i) * 16 + fg * 4 + r) * SP + wn * TN * 16 + frow;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float v = acc[i][j][r] + bv[j];
            stage[pl + j * 16] = f2bf(fmaxf(v, v * p.nslope));
          }
        }
      __syncthreads();
      constexpr int VPP = BN / 8, NV = BM * VPP;
      for (int it = tid; it < NV; it += 256) {
        const int row = it / VPP, cv = it % VPP;
        const int pix = out_pix[row], c = n0 + cv * 8;
        if (pix < 0 || c >= p.Cout) continue;
        uint4 o = *reinterpret_cast<const uint4*>(stage + row * SP + cv * 8);
        const int idx = pix * p.Cout + c;
        if (has_res || has_aux) {
          uint4 rr = make_uint4(0, 0, 0, 0), aa = rr;
          if (has_res) rr = *reinterpret_cast<const uint4*>(gres + idx);
          if (has_aux) aa = *reinterpret_cast<const uint4*>(gaux + idx);
          u16* ov = reinterpret_cast<u16*>(&o);
          const u16* rv = reinterpret_cast<const u16*>(&rr);
          const u16* av = reinterpret_cast<const u16*>(&aa);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = bf2f(ov[e]);
            if (has_res) v += bf2f(rv[e]);
            if (has_aux) v *= bf2f(av[e]) > 0.f ? 1.f : p.mslope;
            ov[e] = f2bf(v);
          }
        }
        *reinterpret_cast<uint4*>(gout + idx) = o;
      }
      return;
    }
  }
  auto epilogue = [&](auto check_tag, auto slow_tag) {
    constexpr bool CHECK = decltype(check_tag)::value, SLOW = decltype(slow_tag)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pix = out_pix[(wm * TM + i) * 16 + fg * 4 + r];
        if (CHECK && pix < 0) continue;
        const int off = pix * p.Cout + col0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (CHECK && col0 + j * 16 >= p.Cout) continue;
          const int idx = off + j * 16;
          float v = acc[i][j][r] + bv[j];
          if constexpr (SLOW) v = act_fwd(v, p.act, p.act_alpha);
          else v = fmaxf(v, v * p.nslope);
          if (has_res) v += Elem<TOut>::ld(gres + idx);
          if (has_aux) {
            if constexpr (SLOW) v *= act_grad_from_out(Elem<TOut>::ld(gaux + idx), p.mask_act, p.mask_alpha);
            else v *= Elem<TOut>::ld(gaux + idx) > 0.f ? 1.f : p.mslope;
          }
          Elem<TOut>::st(gout + idx, v);
        }
      }
    }
  };
  if (slow) epilogue(std::true_type{}, std::true_type{});
  else if (m0 + BM <= Mq && n0 + BN <= p.Cout) epilogue(std::false_type{}, std::false_type{});
  else epilogue(std::true_type{}, std::false_type{});
}

int tg_deconv3x3s2_ws_try(const tg_conv_desc* d, const void* in, const void* weight, const float* bias, const void* res,
                          const void* aux, void* out, hipStream_t st);      // conv3x3_ws.hip

template <typename TIn, typename TOut, int WM, int WN, int TM, int TN>
static void launch_cfg(const ConvP& p, int nphase, int mq_max, hipStream_t st) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  dim3 grid((mq_max + BM - 1) / BM, (p.Cout + BN - 1) / BN, nphase);
  static const char* const pname = [] {
    static char b[96];
    snprintf(b, sizeof(b), "conv_igemm<%s,%s,%d,%d,%d,%d>", sizeof(TIn) == 2 ? "bf16" : "f32",
             sizeof(TOut) == 2 ? "bf16" : "f32", WM, WN, TM, TN);
    return (const char*)b;
  }();
  // algorithmic MACs: gather = out pixels * taps; transposed stride s = in pixels * taps (no zero MACs)
  const double opx = (double)p.N * p.Hout * p.Wout, ipx = (double)p.N * p.Hin * p.Win;
  const double macs = (p.mode == 0 ? opx : ipx) * p.KH * p.KW * (double)p.Cin * p.Cout;
  const double by = ipx * p.Cin * sizeof(TIn) + opx * p.Cout * sizeof(TOut) * (1 + (p.res != nullptr) + (p.aux != nullptr)) +
                    (double)p.KH * p.KW * p.Cin * p.Cout * sizeof(TIn);
  if (p.vec && p.in_bytes != 0)
    TG_LAUNCH(pname, 2.0 * macs, by, (conv_igemm_kernel<TIn, TOut, WM, WN, TM, TN, true>), grid, dim3(256), 0, st, p);
  else
    TG_LAUNCH(pname, 2.0 * macs, by, (conv_igemm_kernel<TIn, TOut, WM, WN, TM, TN, false>), grid, dim3(256), 0, st, p);
}

template <typename TIn, typename TOut>
static void launch_typed(const ConvP& p, int nphase, int mq_max, hipStream_t st, bool coexist) {
  const int64_t M = mq_max;
  if (p.Cout <= 16) {
    if (M * nphase >= 128 * 512) launch_cfg<TIn, TOut, 4, 1, 2, 1>(p, nphase, mq_max, st);
    else launch_cfg<TIn, TOut, 4, 1, 1, 1>(p, nphase, mq_max, st);
  } else if (p.Cout <= 32) {
    if (M * nphase >= 128 * 512) launch_cfg<TIn, TOut, 4, 1, 2, 2>(p, nphase, mq_max, st);
    else launch_cfg<TIn, TOut, 4, 1, 1, 2>(p, nphase, mq_max, st);
  } else {
    const int64_t ntile = (p.Cout + 63) / 64;
    // TG_CONV_COEXIST: the 128-row tile (56 KB LDS, 132 registers) does not fit beside a resident <8,64> 3x3 workgroup
    // (109 KB); the 64-row tile (37 KB) does
    if (!coexist && M * nphase * ntile >= (int64_t)128 * 512) launch_cfg<TIn, TOut, 2, 2, 4, 2>(p, nphase, mq_max, st);
    else if (M * nphase * ntile >= (int64_t)64 * 512) launch_cfg<TIn, TOut, 2, 2, 2, 2>(p, nphase, mq_max, st);
    else launch_cfg<TIn, TOut, 2, 2, 1, 2>(p, nphase, mq_max, st);
  }
}

extern "C" int tg_conv_forward(const tg_conv_desc* d, const void* in, const void* weight, const float* bias,
                               const void* res, const void* aux, void* out, void* stream) {
  TG_CHECK_ARG(d && in && weight && out, "null pointer");
  TG_CHECK_ARG(d->N > 0 && d->Hin > 0 && d->Win > 0 && d->Cin > 0 && d->Hout > 0 && d->Wout > 0 && d->Cout > 0,
               "non-positive dimension");
  TG_CHECK_ARG(d->KH >= 1 && d->KH <= 11 && d->KW >= 1 && d->KW <= 11, "kernel size out of range");
  TG_CHECK_ARG(d->stride >= 1 && d->stride <= 4, "stride must be 1..4");
  TG_CHECK_ARG(d->mode == 0 || d->mode == 1, "mode must be 0 (gather) or 1 (transposed)");
  TG_CHECK_ARG(d->mode == 0 || (d->Hout < 16000 && d->Wout < 16000), "transposed mode: output extent must be < 16000");
  TG_CHECK_ARG((d->in_dtype == TG_F32 || d->in_dtype == TG_BF16) && (d->out_dtype == TG_F32 || d->out_dtype == TG_BF16),
               "bad dtype");
  TG_CHECK_ARG(!(d->in_dtype == TG_F32 && d->out_dtype == TG_BF16), "f32 in / bf16 out is not built");
  TG_CHECK_ARG((int64_t)d->N * d->Hout * d->Wout * d->Cout < (1ll << 31) &&
                   (int64_t)d->N * d->Hin * d->Win < (1ll << 31),
               "tensor too large for 32-bit pixel indexing");
  hipStream_t st0 = static_cast<hipStream_t>(stream);
  const bool use_tile3 = true;
  if (use_tile3 && tg_conv3x3_try(d, in, weight, bias, res, aux, out, st0)) TG_CHECK_LAUNCH();
  if (tg_deconv3x3s2_ws_try(d, in, weight, bias, res, aux, out, st0)) TG_CHECK_LAUNCH();
  ConvP p;
  p.in = in; p.w = weight; p.bias = bias; p.res = res; p.aux = aux; p.out = out;
  p.N = d->N; p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin;
  p.Hout = d->Hout; p.Wout = d->Wout; p.Cout = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.s = d->stride; p.pt = d->pad_t; p.pl = d->pad_l; p.mode = d->mode;
  p.act = d->act; p.act_alpha = d->act_alpha; p.mask_act = d->mask_act; p.mask_alpha = d->mask_alpha;
  // (a residual or LeakyReLU-mask operand: the fp32 register epilogue, ONE rounding -- the staged one would round the tile to bf16
  //  first and the sum again; see conv3x3.hip)
  const int direct = (res != nullptr || (aux != nullptr && d->mask_act == TG_ACT_LRELU)) ? 1 : 0;
  p.direct_epi = direct;
  p.nslope = d->act == TG_ACT_RELU ? 0.f : (d->act == TG_ACT_LRELU ? d->act_alpha : 1.f);
  p.mslope = d->mask_act == TG_ACT_RELU ? 0.f : (d->mask_act == TG_ACT_LRELU ? d->mask_alpha : 1.f);
  const int epv = d->in_dtype == TG_F32 ? 4 : 8;
  p.vec = (d->Cin % epv == 0) && (((uintptr_t)in | (uintptr_t)weight) % 16 == 0);
  {
    const int64_t esz = d->in_dtype == TG_F32 ? 4 : 2;
    const int64_t ib = (int64_t)d->N * d->Hin * d->Win * d->Cin * esz, wb = (int64_t)d->KH * d->KW * d->Cout * d->Cin * esz;
    const bool no_buf = false;
    const bool fits = ib < ((int64_t)1 << 31) && wb < ((int64_t)1 << 31) && !no_buf;
    p.sinv = (65536 + d->stride - 1) / d->stride;
    p.in_bytes = fits ? (unsigned)ib : 0u;
    p.w_bytes = fits ? (unsigned)wb : 0u;
  }
  int nphase = 1, mq_max = d->N * d->Hout * d->Wout;
  if (d->mode == 1 && d->stride > 1) {
    nphase = d->stride * d->stride;
    mq_max = d->N * ((d->Hout + d->stride - 1) / d->stride) * ((d->Wout + d->stride - 1) / d->stride);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool coexist = (d->flags & TG_CONV_COEXIST) != 0;
  if (d->in_dtype == TG_F32) launch_typed<float, float>(p, nphase, mq_max, st, coexist);
  else if (d->out_dtype == TG_BF16) launch_typed<u16, u16>(p, nphase, mq_max, st, coexist);
  else launch_typed<u16, float>(p, nphase, mq_max, st, coexist);
  TG_CHECK_LAUNCH();
}
