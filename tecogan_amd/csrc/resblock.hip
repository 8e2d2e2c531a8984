// Fused residual block for the recurrent generator (gfx950, bf16 MFMA):
//
//     mid = act1( conv3x3(x, W1) + b1 ) [* relu'(m1)]          (written out: needed by the backward / wgrad)
//     out =       conv3x3(mid, W2) + b2 + x [* relu'(m2)]
//
// forward  (reference lib/frvsr.py:50-57): x = a_{i-1}, act1 = ReLU            -> mid = r_i, out = a_i
// backward (same block, taps flipped):     x = d a_i,   m1 = r_i, m2 = a_0|null -> mid = d(pre-ReLU r_i), out = d a_{i-1}
//
// Why: at the training shape [4,32,32,64] one conv is 0.3 GFLOP -- each launch is latency, not math
// (launch + load round trip + store/write-back ~5 us, MFMA block < 0.5 us), and the 20 res-block convs per frame sit
// on the strictly sequential recurrent chain.  One workgroup computes a TH x 16 output tile end to end: the
// intermediate is produced for the (TH+2) x 18 halo region (recomputed at tile borders, 2.25x MFMA work for TH=2 --
// irrelevant here), kept in LDS, and consumed by the second conv without leaving the CU.  That removes one kernel
// boundary and one global write->read round trip per block in both directions of the BPTT chain.
//
// LDS: x halo tile (TH+4) x 20 px, mid tile (TH+2) x 18 px and a TH x 16 staging tile for 16-byte output rows (32 KB);
// rows are pixels with a 144-byte pitch.  Weights never touch LDS: wave w owns output channels [16w, 16w+16) in both
// stages, so its 18 B-fragments per conv (9 taps x 2 K-halves x 16 B per lane) are loaded straight from L2 into VGPRs --
// W2's loads are issued up front and land during stage 1.  (Staging the two full 83 KB panels through LDS, as the
// unfused kernel does, made the fused kernel slower than two narrow-tile launches.)  64 channels only.
#include "common.h"
#include <mutex>

struct ResP {
  const u16* x;
  const u16* w1;
  const u16* w2;
  const float* b1;
  const float* b2;
  const u16* m1;
  const u16* m2;
  u16* mid;
  u16* out;
  int N, H, W, flip, relu1, tiles_y, tiles_x;
};

// B fragments of one conv for this wave's 16 output channels: lane (frow, fg) holds, for every (tap, K-half), the 8
// input channels [32*kk + 8*fg, +8) of output channel 16*w + frow -- 16 contiguous bytes of the [tap][out][in] panel.
// Macros (straight-line code) + an empty-asm pin keep the 18 vectors in VGPRs; as arrays handed to a helper/lambda the
// compiler parked them in scratch memory across stage 1 (304 B/lane, 3x slower kernel).
#define RES_LOAD_B(ARR, WSRC)                                                                       \
  _Pragma("unroll") for (int t_ = 0; t_ < 18; ++t_) {                                               \
    const int tap_ = t_ >> 1, kk_ = t_ & 1;                                                         \
    const int wtap_ = p.flip ? 8 - tap_ : tap_;                                                     \
    ARR[t_] = *reinterpret_cast<const uint4*>((WSRC) + (wtap_ * 64 + c) * 64 + kk_ * 32 + fg * 8);  \
  }
#define RES_PIN(ARR)                                                                                \
  _Pragma("unroll") for (int t_ = 0; t_ < 18; ++t_)                                                 \
      asm volatile("" : "+v"(ARR[t_].x), "+v"(ARR[t_].y), "+v"(ARR[t_].z), "+v"(ARR[t_].w));

// Cycle stamps (tools/trace_resblock.py builds a private -DTG_RES_TRACE copy; the product build has none of it).
#ifdef TG_RES_TRACE
__device__ unsigned long long tg_res_trace_buf[16];
#define RES_STAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) tg_res_trace_buf[(i)] = (unsigned long long)clock64(); } while (0)
extern "C" int tg_debug_res_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_res_trace_buf), sizeof(unsigned long long) * 16);
}
#else
#define RES_STAMP(i) do { } while (0)
#endif

template <int TH>
__global__ __launch_bounds__(256, 1) void resblock_fused_kernel(ResP p) {
  constexpr int XW = 20, XH = TH + 4, NX = XH * XW;
  constexpr int RW = 18, RH = TH + 2, NR = RH * RW;
  constexpr int M1 = (NR + 15) / 16;
  constexpr int ROWB = 144, RU = 72;                      // row pitch in bytes / in u16
  constexpr int X_ITEMS = NX * 8, X_LOADS = (X_ITEMS + 255) / 256;
  constexpr int O_ITEMS = TH * 16 * 8;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Xs = smem;
  unsigned char* Rs = Xs + NX * ROWB;
  unsigned char* Os = Rs + NR * ROWB;

  RES_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  const int tile = blockIdx.x;
  const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
  const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * 16;
  const int c = w * 16 + frow;                            // this lane's output channel (both stages)

  // ---- global -> registers: x halo tile and W1 (unconditional clamped loads + select) ----------------------
  uint4 rx[X_LOADS], bw1[18], bw2[18];
#pragma unroll
  for (int k = 0; k < X_LOADS; ++k) {
    const int item = min(tid + k * 256, X_ITEMS - 1);
    const int pix = item >> 3, ch = item & 7;
    const int gy = y0 - 2 + pix / XW, gx = x0 - 2 + pix % XW;
    const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    uint4 v = *reinterpret_cast<const uint4*>(p.x + (ok ? ((n * p.H + gy) * p.W + gx) * 64 + ch * 8 : 0));
    if (!ok) v = make_uint4(0, 0, 0, 0);
    rx[k] = v;
  }
  RES_LOAD_B(bw1, p.w1)
  RES_LOAD_B(bw2, p.w2)                                    // lands during stage 1
  RES_STAMP(1);
#pragma unroll
  for (int k = 0; k < X_LOADS; ++k) {
    const int item = tid + k * 256;
    if (item < X_ITEMS) *reinterpret_cast<uint4*>(Xs + (item >> 3) * ROWB + (item & 7) * 16) = rx[k];
  }
  __syncthreads();
  RES_STAMP(2);

  // ---- stage 1: mid over the (TH+2) x 18 region; wave w owns channels [16w, 16w+16) ------------------------
  f32x4 acc1[M1];
  int base1[M1];
#pragma unroll
  for (int mt = 0; mt < M1; ++mt) {
    acc1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q = min(mt * 16 + frow, NR - 1);
    base1[mt] = ((q / RW) * XW + q % RW) * ROWB + fg * 16;
  }
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int kh = tap / 3, kw = tap % 3;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 bfr = bw1[tap * 2 + kk];
#pragma unroll
      for (int mt = 0; mt < M1; ++mt) {
        uint4 af = *reinterpret_cast<const uint4*>(Xs + base1[mt] + (kh * XW + kw) * ROWB + kk * 64);
        acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&af),
                                                           *reinterpret_cast<bf16x8*>(&bfr), acc1[mt], 0, 0, 0);
      }
    }
  }
  RES_STAMP(3);
  {
    const float bias1 = p.b1 ? p.b1[c] : 0.f;
    u16* R16 = reinterpret_cast<u16*>(Rs);
#pragma unroll
    for (int mt = 0; mt < M1; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = mt * 16 + fg * 4 + r;
        if (q >= NR) continue;
        const int gy = y0 - 1 + q / RW, gx = x0 - 1 + q % RW;
        const bool inside = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        float v = acc1[mt][r] + bias1;
        if (p.relu1) v = fmaxf(v, 0.f);
        if (p.m1) {
          const u16 mv = p.m1[(inside ? ((n * p.H + gy) * p.W + gx) * 64 : 0) + c];
          v = bf2f(mv) > 0.f ? v : 0.f;
        }
        R16[q * RU + c] = inside ? f2bf(v) : (u16)0;       // SAME padding of the second conv: zero outside the image
      }
  }
  __syncthreads();                                          // mid tile complete
  RES_STAMP(4);
  for (int item = tid; item < O_ITEMS; item += 256) {       // interior of mid -> global, 16-byte rows
    const int pl = item >> 3, cv = item & 7;
    const int i = pl >> 4, xx = pl & 15;
    const int gy = y0 + i, gx = x0 + xx;
    if (gy < p.H && gx < p.W)
      *reinterpret_cast<uint4*>(p.mid + ((n * p.H + gy) * p.W + gx) * 64 + cv * 8) =
          *reinterpret_cast<const uint4*>(Rs + ((i + 1) * RW + xx + 1) * ROWB + cv * 16);
  }
  RES_PIN(bw2)
  RES_STAMP(5);

  // ---- stage 2: out over the TH x 16 tile ------------------------------------------------------------------
  f32x4 acc2[TH];
#pragma unroll
  for (int i = 0; i < TH; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* Afrag2 = Rs + frow * ROWB + fg * 16;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int kh = tap / 3, kw = tap % 3;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 bfr = bw2[tap * 2 + kk];
#pragma unroll
      for (int i = 0; i < TH; ++i) {
        uint4 af = *reinterpret_cast<const uint4*>(Afrag2 + ((i + kh) * RW + kw) * ROWB + kk * 64);
        acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&af),
                                                          *reinterpret_cast<bf16x8*>(&bfr), acc2[i], 0, 0, 0);
      }
    }
  }
  RES_STAMP(6);
  {
    const float bias2 = p.b2 ? p.b2[c] : 0.f;
    const u16* X16 = reinterpret_cast<const u16*>(Xs);
    u16* O16 = reinterpret_cast<u16*>(Os);
#pragma unroll
    for (int i = 0; i < TH; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int xx = fg * 4 + r;
        const float v = acc2[i][r] + bias2 + bf2f(X16[((i + 2) * XW + xx + 2) * RU + c]);   // + skip (x, from LDS)
        O16[(i * 16 + xx) * RU + c] = f2bf(v);
      }
  }
  __syncthreads();
  RES_STAMP(7);
  for (int item = tid; item < O_ITEMS; item += 256) {
    const int pl = item >> 3, cv = item & 7;
    const int i = pl >> 4, xx = pl & 15;
    const int gy = y0 + i, gx = x0 + xx;
    if (gy >= p.H || gx >= p.W) continue;
    uint4 o = *reinterpret_cast<const uint4*>(Os + pl * ROWB + cv * 16);
    const int idx = ((n * p.H + gy) * p.W + gx) * 64 + cv * 8;
    if (p.m2) {                                             // ReLU mask of the producer of x (exact: 0/1 factor)
      const uint4 mm = *reinterpret_cast<const uint4*>(p.m2 + idx);
      u16* ov = reinterpret_cast<u16*>(&o);
      const u16* mv = reinterpret_cast<const u16*>(&mm);
#pragma unroll
      for (int e = 0; e < 8; ++e) ov[e] = bf2f(mv[e]) > 0.f ? ov[e] : (u16)0;
    }
    *reinterpret_cast<uint4*>(p.out + idx) = o;
  }
  RES_STAMP(8);
#ifdef TG_RES_TRACE
  __builtin_amdgcn_s_waitcnt(0);
  RES_STAMP(9);
#endif
}

template <int TH>
static void launch_res(ResP p, hipStream_t st) {
  constexpr int LDS = ((TH + 4) * 20 + (TH + 2) * 18 + TH * 16) * 144;
  auto kern = resblock_fused_kernel<TH>;
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  p.tiles_y = (p.H + TH - 1) / TH;
  p.tiles_x = (p.W + 15) / 16;
  hipLaunchKernelGGL(kern, dim3(p.N * p.tiles_y * p.tiles_x), dim3(256), LDS, st, p);
}

extern "C" int tg_resblock_fused(const void* x, const void* w1, const float* b1, const void* m1, void* mid,
                                 const void* w2, const float* b2, const void* m2, void* out, int N, int H, int W,
                                 int flip, int relu1, void* stream) {
  TG_CHECK_ARG(x && w1 && w2 && mid && out, "null pointer");
  TG_CHECK_ARG(N > 0 && H > 0 && W > 0 && (int64_t)N * H * W * 64 < (1ll << 31), "bad shape");
  TG_CHECK_ARG((((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)mid | (uintptr_t)out) & 15) == 0,
               "pointers must be 16-byte aligned");
  ResP p;
  p.x = (const u16*)x; p.w1 = (const u16*)w1; p.w2 = (const u16*)w2; p.b1 = b1; p.b2 = b2;
  p.m1 = (const u16*)m1; p.m2 = (const u16*)m2; p.mid = (u16*)mid; p.out = (u16*)out;
  p.N = N; p.H = H; p.W = W; p.flip = flip; p.relu1 = relu1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t tiles2 = (int64_t)N * ((H + 1) / 2) * ((W + 15) / 16);
  if (tiles2 <= 512) launch_res<2>(p, st);      // small problems: more, smaller workgroups
  else launch_res<4>(p, st);
  TG_CHECK_LAUNCH();
}
