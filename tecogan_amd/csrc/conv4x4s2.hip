// 4x4 stride-2 SAME convolution and its input gradient for discriminator_F (bf16) -- gfx950.
//
// Covers the four `conv2(net, 4, C, 2)` layers of reference lib/Teco.py:52-66 (conv2 = lib/ops.py:47-56, slim.conv2d k4 s2
// SAME: pad 1 on every side for even sizes) in both directions: the forward conv of the real and the fake triplets and the
// input gradients of D's own backward pass and of the generator-side pass (tf.gradients through lib/Teco.py:393-449).
//
// Why.  Until round 5 these ran on the generic implicit-GEMM kernel (conv_igemm.hip): [24,128,128,64] -> [24,64,64,64] took
// 51 us for 12.9 GFLOP and 63 MB (MFMA floor 5 us, HBM floor 10 us), the discriminator was 2.3 ms of the 8.3 ms step's main
// stream at 3.4 % of the MFMA peak.  Both kernels here are built like conv3x3_wr.hip:
//   * a wave owns ALL pixels of the workgroup's tile x 16 output channels; its 16 weight fragments of a 32-channel stage go
//     global -> registers from a fragment-order copy (tg_pack_taps_frag; 16 KB contiguous per wave and stage = whole cache
//     lines), each re-requested for the next stage right after its last MFMA of this one;
//   * only the activation halo goes through LDS, by LDS-DMA into a double buffer, 64-byte rows XOR-swizzled as in
//     conv3x3_dma.hip (conflict-free ds_read_b128 under the gfx950 lane grouping);
//   * FORWARD (gather, stride 2): tile = 16 x 8 output pixels, halo = 18 rows x 34 columns of the input, stored as TWO column
//     -parity planes of 18 x 17 pixels -- the DMA picks each slot's source pixel, so the stride-2 gather becomes a stride-1
//     fragment read: tap column kw reads plane kw & 1 at column offset kw >> 1.  A fragment (row, kw) feeds the two kernel rows
//     of its parity: 72 reads per 128 MFMAs;
//   * INPUT GRADIENT (transposed, stride 2): the four output phases (y & 1, x & 1) are 2x2-tap stride-1 convolutions over the
//     SAME (4 + 2) x 18 halo of dY: tile = 16 x 4 positions of dY = 32 x 8 pixels of dX, 16 accumulator tiles per wave
//     (4 phases x 4 rows), a fragment (row, column shift) feeds up to 8 MFMAs: 18 reads per 64 MFMAs, no zero MACs.
#include "common.h"
#include <type_traits>

struct C4P {
  const void* in;     // forward: x [N,H,W,Cin]; input gradient: dY [N,H,W,Cin] (Cin = the reduction channels either way)
  const void* wf;     // fragment order: [Cout/16][Cin/32][16][64 lanes][8 bf16]
  const float* bias;
  const void* res;
  const void* aux;
  void* out;          // forward: [N,H/2,W/2,Cout]; input gradient: [N,2H,2W,Cout]
  int N, H, W, Cin, Cout;
  int Ho, Wo;
  float nslope, mslope;
  int tiles_y, tiles_x, nblk, nunits, u8;
  unsigned in_bytes, w_bytes, out_bytes;
  float* stats;       // forward, nullable: [2][Cout] += per-channel mean and second moment of the (pre-activation) result
  float inv_rows;     // 1 / (N Ho Wo)
};

typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2c __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_c;

namespace {
constexpr unsigned C4_OOB = 0x80000000u;
template <int I, int N, typename F>
__device__ __forceinline__ void c4_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    c4_static_for<I + 1, N>(f);
  }
}
// forward geometry: 8 output rows x 16 output columns; halo 18 input rows x 2 parity planes x 17 columns
constexpr int F_TH = 8, F_ROWS = 2 * F_TH + 2, F_PC = 17, F_PLANE = F_ROWS * F_PC, F_HALO = 2 * F_PLANE;   // 18, 17, 306, 612
constexpr int F_INST = (F_HALO * 4 + 63) / 64, F_ROUNDS = (F_INST + 3) / 4, F_HB = F_INST * 1024;       // 39, 10, 39936
// input-gradient geometry: 4 x 16 positions of dY; halo 6 x 18
constexpr int B_TH = 4, B_HR = B_TH + 2, B_HW = 18, B_HALO = B_HR * B_HW;                               // 108
constexpr int B_INST = (B_HALO * 4 + 63) / 64, B_ROUNDS = (B_INST + 3) / 4, B_HB = B_ROUNDS * 4 * 1024; // 7, 2, 8192

// common epilogue of one accumulator tile: bias, activation, residual, mask, bf16 pack, 8-byte store.  The residual / mask
// operands are loaded by the CALLER for all tiles before the first store (a load issued between two stores waits for the store
// ahead of it: 16 dependent round trips per wave, measured as 89 us in the step for a launch that takes 27 us alone).
template <bool HAS_RES, bool HAS_AUX, typename RS>
__device__ __forceinline__ void c4_store(const f32x4& a, const float (&bv)[4], float nslope, float mslope, unsigned off, const RS& rsO,
                                         const u32x2c& rr, const u32x2c& aa) {
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r] = a[r] + bv[r];
    v[r] = fmaxf(v[r], v[r] * nslope);
  }
  if constexpr (HAS_RES) {
    v[0] += __uint_as_float(rr.x << 16);
    v[1] += __uint_as_float(rr.x & 0xffff0000u);
    v[2] += __uint_as_float(rr.y << 16);
    v[3] += __uint_as_float(rr.y & 0xffff0000u);
  }
  if constexpr (HAS_AUX) {
    v[0] *= __uint_as_float(aa.x << 16) > 0.f ? 1.f : mslope;
    v[1] *= __uint_as_float(aa.x & 0xffff0000u) > 0.f ? 1.f : mslope;
    v[2] *= __uint_as_float(aa.y << 16) > 0.f ? 1.f : mslope;
    v[3] *= __uint_as_float(aa.y & 0xffff0000u) > 0.f ? 1.f : mslope;
  }
  u32x2c o;
  o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
  o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
  __builtin_amdgcn_raw_buffer_store_b64(o, rsO, (int)off, 0, 0);
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// forward: out[n, oy, ox, :] = sum_{kh, kw} in[n, 2 oy - 1 + kh, 2 ox - 1 + kw, :] . W[kh, kw]
// ---------------------------------------------------------------------------------------------------------------------------
template <bool HAS_RES>
__global__ __launch_bounds__(256, 2) void conv4x4s2_fwd_kernel(C4P p) {
  constexpr int NS = 4 * F_ROWS;                                      // halo fragments of a stage: s = kw * 18 + r
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x F_HB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fg = lane >> 4;

  const int lin = blockIdx.x;
  const int u = (lin & 7) * p.u8 + (lin >> 3);                        // XCD x owns units [x u8, (x + 1) u8)
  if (u >= p.nunits) return;
  const int tile = u / p.nblk, blk = u - tile * p.nblk;
  const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
  const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
  const int y0 = 2 * ty * F_TH - 1, x0 = 2 * tx * 16 - 1;             // input pixel of halo (row 0, plane 0, column 0)
  const int g16 = blk * 4 + wave;
  const int cbase = g16 * 16;
  const int row_bytes = p.Cin * 2;
  const int nchunk = p.Cin >> 5;

  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wf), 0, (int)p.w_bytes, 0x00020000);
  const auto rsrcO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
  const auto rsrcR = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_RES ? p.res : p.out), 0, (int)p.out_bytes, 0x00020000);
  const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);

  // LDS-DMA slot descriptors: slot S = halo position q = S / 4 (plane q / 306, row (q % 306) / 17, column q % 17), 16-byte
  // group (S % 4) ^ 2 * ((q >> 2) & 1) of that pixel's 32-channel chunk.  Instruction 39 does not exist: the fourth wave's
  // last round fetches instruction 38 again (same bytes to the same slots) instead of branching around the DMA.
  unsigned hoff[F_ROUNDS];
  int hinst[F_ROUNDS];
#pragma unroll
  for (int k = 0; k < F_ROUNDS; ++k) {
    const int inst = (wave + 4 * k) < F_INST ? (wave + 4 * k) : F_INST - 1;
    hinst[k] = inst;
    const int S = inst * 64 + lane;
    const int q = S >> 2, ch = (S & 3) ^ (((S >> 4) & 1) << 1);
    const int pl = q / F_PLANE, rem = q - pl * F_PLANE;
    const int r = rem / F_PC, c = rem - r * F_PC;
    const int iy = y0 + r, ix = x0 + 2 * c + pl;
    const bool ok = q < F_HALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    hoff[k] = ok ? (unsigned)(((n * p.H + iy) * p.W + ix) * row_bytes + ch * 16) : C4_OOB;
  }
  auto dma_round = [&](int k, int chunk, int buf, bool live) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void_c*)(smem + buf * F_HB + hinst[k] * 1024), 16,
                                             (int)(live ? hoff[k] : C4_OOB), chunk * 64, 0, 0);
  };

  // prologue: bias, first halo, first 16 weight fragments (the counted wait at the top of a stage relies on the 16 weight loads
  // being the youngest vector-memory operations of the wave)
  const u32x4c bq = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (cbase + fg * 4) * 4, 0, 0);
#pragma unroll
  for (int k = 0; k < F_ROUNDS; ++k) dma_round(k, 0, 0, true);
  const int wlane = lane * 16;
  int wsoff = g16 * nchunk * 16384;                                   // scalar: this stage's 16 fragments (tap t at + 1024 t)
  u32x4c wf[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int kw = j >> 2, kh = j & 3;                                // consumption order: kw outer, kh inner
    wf[kh * 4 + kw] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, wlane, wsoff + (kh * 4 + kw) * 1024, 0);
  }
  __builtin_amdgcn_sched_barrier(0);

  int abase[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) abase[d] = frow * 64 + ((fg ^ (((((frow & 7) + d) >> 2) & 1) << 1)) << 4);

  f32x4 acc[F_TH];
#pragma unroll
  for (int i = 0; i < F_TH; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int buf = chunk & 1;
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                 // this wave's DMA slots of the stage have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool has_next = chunk + 1 < nchunk;
    const unsigned wl_next = has_next ? (unsigned)wlane : C4_OOB;
    wsoff += 16384;
    const unsigned char* sb = smem + buf * F_HB;
    auto rd = [&](int s) {                                            // fragment (input row r, tap column kw): plane kw & 1, shift kw >> 1
      const int kw = s / F_ROWS, r = s - kw * F_ROWS;
      const int K = ((kw & 1) * F_ROWS + r) * F_PC + (kw >> 1);
      return *reinterpret_cast<const u32x4c*>(sb + abase[K & 7] + K * 64);
    };
    u32x4c F[3];
    F[0] = rd(0);
    F[1] = rd(1);
    c4_static_for<0, NS>([&](auto sv) {
      constexpr int s = decltype(sv)::value;
      constexpr int kw = s / F_ROWS, r = s - kw * F_ROWS;
      if constexpr (s + 2 < NS) F[(s + 2) % 3] = rd(s + 2);
      // input row r = 2 i + kh: the two kernel rows of r's parity
#pragma unroll
      for (int kh = (r & 1); kh < 4; kh += 2) {
        const int i = (r - kh) / 2;
        if (r >= kh && i < F_TH)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[kh * 4 + kw]),
                                                           __builtin_bit_cast(bf16x8, F[s % 3]), acc[i], 0, 0, 0);
      }
      if constexpr (s < F_ROUNDS) {
        __builtin_amdgcn_sched_barrier(0);
        dma_round(s, chunk + 1, buf ^ 1, has_next);
        __builtin_amdgcn_sched_barrier(0);
      }
      // kernel row kh is used for the last time at r = 2 (F_TH - 1) + kh
      if constexpr (r >= 2 * (F_TH - 1)) {
        constexpr int kh = r - 2 * (F_TH - 1);
        __builtin_amdgcn_sched_barrier(0);
        wf[kh * 4 + kw] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, (int)wl_next, wsoff + (kh * 4 + kw) * 1024, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  }

  const float bv[4] = {__uint_as_float(bq.x), __uint_as_float(bq.y), __uint_as_float(bq.z), __uint_as_float(bq.w)};
  const int x = tx * 16 + frow, ybase = ty * F_TH;
  const int co = cbase + fg * 4;
  unsigned offs[F_TH];
  u32x2c rr[HAS_RES ? F_TH : 1];
#pragma unroll
  for (int i = 0; i < F_TH; ++i) {
    const int y = ybase + i;
    offs[i] = (y < p.Ho && x < p.Wo) ? (unsigned)((((n * p.Ho + y) * p.Wo + x) * p.Cout + co) * 2) : C4_OOB;
    if constexpr (HAS_RES) rr[i] = __builtin_amdgcn_raw_buffer_load_b64(rsrcR, (int)offs[i], 0, 0);
  }
#pragma unroll
  for (int i = 0; i < F_TH; ++i) c4_store<HAS_RES, false>(acc[i], bv, p.nslope, 1.f, offs[i], rsrcO, rr[HAS_RES ? i : 0], rr[0]);
  // Batch-norm statistics of the layer (lib/ops.py:88-90 after lib/Teco.py:37): per-channel sum and sum of squares of conv + bias
  // from the fp32 accumulators -- the 16 lanes of a channel quad hold the tile's 16 columns -- one atomic pair per channel and wave.
  // (tg_bn_lrelu_forward then skips its two reduction launches: 2 x 9.6 us per layer and pass.)
  if (p.stats) {
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < F_TH; ++i) {
      const bool ok = ybase + i < p.Ho && x < p.Wo;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = ok ? acc[i][r] + bv[r] : 0.f;
        s1[r] += v;
        s2[r] += v * v;
      }
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s1[r] += __shfl_xor(s1[r], m, 64);
        s2[r] += __shfl_xor(s2[r], m, 64);
      }
    if (frow == 0) {
      // TG_BN_STAT_REPLICAS accumulator sets, picked by the unit: 768 workgroups adding to the SAME 2 x 64 addresses serialise in
      // the L2 (the launch went from 22 to 39 us in the step); tg_bn_lrelu_forward(prezeroed = 2) sums the replicas
      float* __restrict__ st = p.stats + (size_t)(u % TG_BN_STAT_REPLICAS) * 2 * p.Cout;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        unsafeAtomicAdd(st + co + r, s1[r] * p.inv_rows);
        unsafeAtomicAdd(st + p.Cout + co + r, s2[r] * p.inv_rows);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// input gradient: dX[n, y, x, :] = sum_{kh, kw : (y + 1 - kh) % 2 == 0, (x + 1 - kw) % 2 == 0} dY[n, (y + 1 - kh) / 2, (x + 1 - kw) / 2, :] . W[kh, kw]
//   phase py = y & 1: (kh, row shift) = (1, 0), (3, -1) for py = 0; (0, +1), (2, 0) for py = 1 -- the same for columns.
// ---------------------------------------------------------------------------------------------------------------------------
template <bool HAS_RES, bool HAS_AUX>
__global__ __launch_bounds__(256, 2) void conv4x4s2_bwd_kernel(C4P p) {
  constexpr int NS = 3 * B_HR;                                        // halo fragments of a stage: s = kc * 6 + hr
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x B_HB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fg = lane >> 4;

  const int lin = blockIdx.x;
  const int u = (lin & 7) * p.u8 + (lin >> 3);
  if (u >= p.nunits) return;
  const int tile = u / p.nblk, blk = u - tile * p.nblk;
  const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
  const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
  const int y0 = ty * B_TH - 1, x0 = tx * 16 - 1;                     // dY position of halo (0, 0)
  const int g16 = blk * 4 + wave;
  const int cbase = g16 * 16;
  const int row_bytes = p.Cin * 2;
  const int nchunk = p.Cin >> 5;

  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wf), 0, (int)p.w_bytes, 0x00020000);
  const auto rsrcO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
  const auto rsrcR = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_RES ? p.res : p.out), 0, (int)p.out_bytes, 0x00020000);
  const auto rsrcM = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_AUX ? p.aux : p.out), 0, (int)p.out_bytes, 0x00020000);

  unsigned hoff[B_ROUNDS];
#pragma unroll
  for (int k = 0; k < B_ROUNDS; ++k) {
    const int S = (wave + 4 * k) * 64 + lane;
    const int q = S >> 2, ch = (S & 3) ^ (((S >> 4) & 1) << 1);
    const int dy = q / B_HW, dx = q - B_HW * dy;
    const bool ok = q < B_HALO && (unsigned)(y0 + dy) < (unsigned)p.H && (unsigned)(x0 + dx) < (unsigned)p.W;
    hoff[k] = ok ? (unsigned)(((n * p.H + y0 + dy) * p.W + x0 + dx) * row_bytes + ch * 16) : C4_OOB;
  }
  auto dma_round = [&](int k, int chunk, int buf, bool live) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void_c*)(smem + buf * B_HB + (wave + 4 * k) * 1024), 16,
                                             (int)(live ? hoff[k] : C4_OOB), chunk * 64, 0, 0);
  };

#pragma unroll
  for (int k = 0; k < B_ROUNDS; ++k) dma_round(k, 0, 0, true);
  const int wlane = lane * 16;
  int wsoff = g16 * nchunk * 16384;
  u32x4c wf[16];
  // consumption order: column shift kc = 0 uses kw = 3, kc = 1 uses kw = 1, 2, kc = 2 uses kw = 0
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int kwo[4] = {3, 1, 2, 0};
    const int kw = kwo[j >> 2], kh = j & 3;
    wf[kh * 4 + kw] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, wlane, wsoff + (kh * 4 + kw) * 1024, 0);
  }
  __builtin_amdgcn_sched_barrier(0);

  int abase[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) abase[d] = frow * 64 + ((fg ^ (((((frow & 7) + d) >> 2) & 1) << 1)) << 4);

  f32x4 acc[2][2][B_TH];                                              // [py][px][row]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < B_TH; ++i) acc[a][b][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int buf = chunk & 1;
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool has_next = chunk + 1 < nchunk;
    const unsigned wl_next = has_next ? (unsigned)wlane : C4_OOB;
    wsoff += 16384;
    const unsigned char* sb = smem + buf * B_HB;
    auto rd = [&](int s) {
      const int kc = s / B_HR, hr = s - kc * B_HR;
      const int K = hr * B_HW + kc;
      return *reinterpret_cast<const u32x4c*>(sb + abase[K & 7] + K * 64);
    };
    u32x4c F[3];
    F[0] = rd(0);
    F[1] = rd(1);
    c4_static_for<0, NS>([&](auto sv) {
      constexpr int s = decltype(sv)::value;
      constexpr int kc = s / B_HR, hr = s - kc * B_HR;
      if constexpr (s + 2 < NS) F[(s + 2) % 3] = rd(s + 2);
      // halo row hr = i + 1 + dr, halo column = frow + 1 + dc with dc = kc - 1
      constexpr int PYK[4][3] = {{0, 1, 0}, {0, 3, -1}, {1, 0, 1}, {1, 2, 0}};       // (phase, kernel index, shift)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int py = PYK[a][0], kh = PYK[a][1], dr = PYK[a][2];
        const int i = hr - 1 - dr;
        if (i < 0 || i >= B_TH) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int px = PYK[b][0], kw = PYK[b][1], dc = PYK[b][2];
          if (dc != kc - 1) continue;
          acc[py][px][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[kh * 4 + kw]),
                                                                  __builtin_bit_cast(bf16x8, F[s % 3]), acc[py][px][i], 0, 0, 0);
        }
      }
      if constexpr (s < B_ROUNDS) {
        __builtin_amdgcn_sched_barrier(0);
        dma_round(s, chunk + 1, buf ^ 1, has_next);
        __builtin_amdgcn_sched_barrier(0);
      }
      // at the end of a column shift its kernel columns are done: kc = 0 -> kw = 3, kc = 1 -> kw = 1, 2, kc = 2 -> kw = 0
      if constexpr (hr == B_HR - 1) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
          const int dc = kw == 3 ? -1 : (kw == 0 ? 1 : 0);
          if (dc != kc - 1) continue;
#pragma unroll
          for (int kh = 0; kh < 4; ++kh)
            wf[kh * 4 + kw] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, (int)wl_next, wsoff + (kh * 4 + kw) * 1024, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  }

  const float bv[4] = {0.f, 0.f, 0.f, 0.f};
  const int co = cbase + fg * 4;
  unsigned offs[B_TH][2][2];
  u32x2c rr[HAS_RES ? B_TH : 1][2][2], aa[HAS_AUX ? B_TH : 1][2][2];
#pragma unroll
  for (int i = 0; i < B_TH; ++i)
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int y = 2 * (ty * B_TH + i) + py, x = 2 * (tx * 16 + frow) + px;
        offs[i][py][px] = (y < p.Ho && x < p.Wo) ? (unsigned)((((n * p.Ho + y) * p.Wo + x) * p.Cout + co) * 2) : C4_OOB;
        if constexpr (HAS_RES) rr[i][py][px] = __builtin_amdgcn_raw_buffer_load_b64(rsrcR, (int)offs[i][py][px], 0, 0);
        if constexpr (HAS_AUX) aa[i][py][px] = __builtin_amdgcn_raw_buffer_load_b64(rsrcM, (int)offs[i][py][px], 0, 0);
      }
#pragma unroll
  for (int i = 0; i < B_TH; ++i)
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px)
        c4_store<HAS_RES, HAS_AUX>(acc[py][px][i], bv, 1.f, p.mslope, offs[i][py][px], rsrcO, rr[HAS_RES ? i : 0][py][px],
                                   aa[HAS_AUX ? i : 0][py][px]);
}

// ---- fragment-order copy of a [taps][Cout][Cin] bf16 operand: dst[g][c][t][l][j] = src[t][16 g + l % 16][32 c + 8 (l / 16) + j]
__global__ void pack_taps_frag_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int taps, int Cout, int Cin, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  long r = i >> 6;
  const int t = (int)(r % taps);
  r /= taps;
  const int nchunk = Cin >> 5;
  const int c = (int)(r % nchunk), g = (int)(r / nchunk);
  dst[i] = src[(((long)t * Cout + g * 16 + (lane & 15)) * Cin + c * 32 + (lane >> 4) * 8) >> 3];
}

// several tensors in one launch (blockIdx.y = tensor): tab = 5 x int64 per tensor {source element offset in src_t (which = 0) or
// src_n (which = 1), destination element offset in dst, taps, Cout, Cin} with which = bit 62 of the first entry
__global__ void pack_taps_frag_multi_kernel(const u16* __restrict__ src_t, const u16* __restrict__ src_n, u16* __restrict__ dst,
                                            const int64_t* __restrict__ tab) {
  const int64_t* t = tab + (int64_t)blockIdx.y * 5;
  const int64_t so = t[0] & ~((int64_t)1 << 62);
  const uint4* __restrict__ src = reinterpret_cast<const uint4*>(((t[0] >> 62) & 1 ? src_n : src_t) + so);
  uint4* __restrict__ d = reinterpret_cast<uint4*>(dst + t[1]);
  const int taps = (int)t[2], Cout = (int)t[3], Cin = (int)t[4];
  const long total = (long)taps * Cout * Cin / 8;
  const int nchunk = Cin >> 5;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    long r = i >> 6;
    const int tp = (int)(r % taps);
    r /= taps;
    const int c = (int)(r % nchunk), g = (int)(r / nchunk);
    d[i] = src[(((long)tp * Cout + g * 16 + (lane & 15)) * Cin + c * 32 + (lane >> 4) * 8) >> 3];
  }
}

extern "C" int tg_pack_taps_frag_multi(const void* src_t, const void* src_n, void* dst, const int64_t* tab, int count, void* stream) {
  TG_CHECK_ARG(src_t && src_n && dst && tab && count > 0, "bad argument");
  TG_CHECK_ARG((((uintptr_t)src_t | (uintptr_t)src_n | (uintptr_t)dst) & 15) == 0, "pointers must be 16-byte aligned");
  hipLaunchKernelGGL(pack_taps_frag_multi_kernel, dim3(32, count), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const u16*>(src_t), static_cast<const u16*>(src_n), static_cast<u16*>(dst), tab);
  TG_CHECK_LAUNCH();
}

extern "C" int tg_pack_taps_frag(const void* w, void* w_frag, int taps, int Cout, int Cin, void* stream) {
  TG_CHECK_ARG(w && w_frag && taps > 0 && Cout > 0 && Cout % 16 == 0 && Cin > 0 && Cin % 32 == 0,
               "bf16 [taps][Cout][Cin] with Cout % 16 == 0, Cin % 32 == 0");
  TG_CHECK_ARG((((uintptr_t)w | (uintptr_t)w_frag) & 15) == 0, "pointers must be 16-byte aligned");
  const long total = (long)taps * Cout * Cin / 8;
  hipLaunchKernelGGL(pack_taps_frag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint4*>(w), static_cast<uint4*>(w_frag), taps, Cout, Cin, total);
  TG_CHECK_LAUNCH();
}

extern "C" int tg_conv4x4s2_frag(const tg_conv_desc* d, const void* in, const void* w_frag, const float* bias, const void* res,
                                 const void* aux, void* out, float* bn_stats, void* stream) {
  TG_CHECK_ARG(d && in && w_frag && out, "null pointer");
  TG_CHECK_ARG(d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad_t == 1 && d->pad_l == 1, "4x4 stride-2 convolutions with pad 1 only");
  TG_CHECK_ARG(d->in_dtype == TG_BF16 && d->out_dtype == TG_BF16, "bf16 tensors only");
  TG_CHECK_ARG(d->Cin % 32 == 0 && d->Cout % 64 == 0, "Cin % 32 == 0, Cout % 64 == 0");
  TG_CHECK_ARG((((uintptr_t)in | (uintptr_t)w_frag | (uintptr_t)out | (uintptr_t)res | (uintptr_t)aux) & 15) == 0,
               "pointers must be 16-byte aligned");
  const bool fwd = d->mode == 0;
  if (fwd) {
    TG_CHECK_ARG(d->Hin % 2 == 0 && d->Win % 2 == 0 && d->Hout == d->Hin / 2 && d->Wout == d->Win / 2, "forward: even input, output = input / 2");
    TG_CHECK_ARG(!aux && d->act < TG_ACT_TANH, "forward epilogue: bias, none / ReLU / LeakyReLU, residual");
  } else {
    TG_CHECK_ARG(d->Hout == 2 * d->Hin && d->Wout == 2 * d->Win, "input gradient: output = 2 x input");
    TG_CHECK_ARG(!bias && d->act == TG_ACT_NONE, "input-gradient epilogue: residual and activation mask only");
    TG_CHECK_ARG(!bn_stats, "batch-norm statistics belong to the forward conv");
  }
  TG_CHECK_ARG(!bn_stats || (d->act == TG_ACT_NONE && !res), "batch-norm statistics: of conv + bias (no activation, no residual)");
  const int64_t in_bytes = (int64_t)d->N * d->Hin * d->Win * d->Cin * 2, out_bytes = (int64_t)d->N * d->Hout * d->Wout * d->Cout * 2;
  const int64_t w_bytes = (int64_t)16 * d->Cout * d->Cin * 2;
  TG_CHECK_ARG(in_bytes < ((int64_t)1 << 31) && out_bytes < ((int64_t)1 << 31), "tensor too large for 32-bit buffer offsets");
  C4P p;
  p.in = in; p.wf = w_frag; p.bias = bias; p.res = res; p.aux = aux; p.out = out;
  p.N = d->N; p.H = d->Hin; p.W = d->Win; p.Cin = d->Cin; p.Cout = d->Cout; p.Ho = d->Hout; p.Wo = d->Wout;
  p.nslope = d->act == TG_ACT_RELU ? 0.f : (d->act == TG_ACT_LRELU ? d->act_alpha : 1.f);
  p.mslope = d->mask_act == TG_ACT_RELU ? 0.f : (d->mask_act == TG_ACT_LRELU ? d->mask_alpha : 1.f);
  p.stats = bn_stats;
  p.inv_rows = (float)(1.0 / ((double)d->N * d->Hout * d->Wout));
  p.nblk = p.Cout / 64;
  if (fwd) {
    p.tiles_x = (p.Wo + 15) / 16;
    p.tiles_y = (p.Ho + F_TH - 1) / F_TH;
  } else {
    p.tiles_x = (p.W + 15) / 16;
    p.tiles_y = (p.H + B_TH - 1) / B_TH;
  }
  const int64_t nunits = (int64_t)p.N * p.tiles_y * p.tiles_x * p.nblk;
  TG_CHECK_ARG(nunits < ((int64_t)1 << 28), "too many tiles");
  p.nunits = (int)nunits;
  p.u8 = (int)((nunits + 7) / 8);
  p.in_bytes = (unsigned)in_bytes; p.w_bytes = (unsigned)w_bytes; p.out_bytes = (unsigned)out_bytes;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double fl = 2.0 * (fwd ? (double)p.N * p.Ho * p.Wo : (double)p.N * p.H * p.W) * 16.0 * p.Cin * p.Cout;
  const double by = (double)in_bytes + (double)out_bytes * (1 + (res != nullptr) + (aux != nullptr)) + (double)w_bytes;
  if (fwd) {
    constexpr int LDS0 = 2 * F_HB, LDS_MAX = 96 * 1024;
    static bool attr = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4x4s2_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv4x4s2_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
      return true;
    }();
    (void)attr;
    // TG_CONV_COEXIST (the real-triplet pass runs beside the forward chain): ONE workgroup per CU -- two of them (156 KB) leave no
    // room for a chain workgroup's 22 KB (the residency cap of conv3x3_wr.hip, same reason): fwd_0 0.89 -> 0.865 ms, step 8.534 /
    // 8.539 -> 8.511 / 8.514 ms, alternating runs on one box (profiles/r05m_ab.txt)
    const int LDS = (d->flags & TG_CONV_COEXIST) ? 88 * 1024 : LDS0;
    if (res) TG_LAUNCH("conv4x4s2_fwd<res>", fl, by, (conv4x4s2_fwd_kernel<true>), dim3(8 * p.u8), dim3(256), LDS, st, p);
    else TG_LAUNCH("conv4x4s2_fwd<>", fl, by, (conv4x4s2_fwd_kernel<false>), dim3(8 * p.u8), dim3(256), LDS, st, p);
  } else {
    constexpr int LDS = 2 * B_HB;
    if (res && aux) TG_LAUNCH("conv4x4s2_bwd<res,aux>", fl, by, (conv4x4s2_bwd_kernel<true, true>), dim3(8 * p.u8), dim3(256), LDS, st, p);
    else if (res) TG_LAUNCH("conv4x4s2_bwd<res>", fl, by, (conv4x4s2_bwd_kernel<true, false>), dim3(8 * p.u8), dim3(256), LDS, st, p);
    else if (aux) TG_LAUNCH("conv4x4s2_bwd<aux>", fl, by, (conv4x4s2_bwd_kernel<false, true>), dim3(8 * p.u8), dim3(256), LDS, st, p);
    else TG_LAUNCH("conv4x4s2_bwd<>", fl, by, (conv4x4s2_bwd_kernel<false, false>), dim3(8 * p.u8), dim3(256), LDS, st, p);
  }
  TG_CHECK_LAUNCH();
}
