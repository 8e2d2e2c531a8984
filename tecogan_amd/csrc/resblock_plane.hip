// The residual TRUNK of generator_F -- `for i in range(1, num_resblock + 1): net = residual_block(net, 64, 1, ...)`, reference
// lib/frvsr.py:50-57,66-70 -- of one frame of the INFERENCE step (main.py:195-216) as ONE persistent launch in the THROUGHPUT regime:
// [1,270,480,64] = 17 x 15 tiles of 16 x 32 pixels, one workgroup per tile and per compute unit, the activations of ALL blocks
// resident in LDS.
//
// Why.  csrc/resblock_thr.hip runs a block as one launch: 31-33 us for 19.1 GFLOP (MFMA floor 7.6 us) -- 1.33 x the MACs (the
// first conv is recomputed on the region the second needs), weights resident in 144 registers per wave (two waves per SIMD whose
// epilogues, barriers and LDS round trips nothing covers, lesson 32), a 33 MB tensor read and written per block and a kernel
// boundary per block.  16 such launches are 0.5 of the 0.75 ms frame.  The hand-off measured for the training trunk (lessons 33-35:
// 0.24-0.39 us workgroup to workgroup) makes the other construction possible at this size too:
//   * a workgroup keeps its 16 x 32 tile for all blocks: two LDS planes of (16+2) x (32+2) pixels x 128 B -- X (the block input,
//     updated in place by the second conv's epilogue: it is also the skip operand) and M (relu(conv_1)); 157 KB of the 160;
//   * what a conv needs from outside is the ONE-pixel ring around the tile (100 pixels x 128 B) from up to 8 neighbour workgroups,
//     after EVERY conv (two hand-offs per block, no recompute: exactly the MACs of the two-launch path).  The border pixels are
//     published as tagged 16-byte granules {v01, tag, v23, tag} straight from the epilogue registers (rows: a store instruction
//     covers one contiguous KiB) into a two-slot ring and swept by the neighbours (handoff.h; rings P / S and the XCC exchange as
//     in resblock_chain.hip);
//   * eight waves = (pixel quarter pq: tile rows 4 pq .. 4 pq + 3 = 8 MFMA pixel tiles) x (channel half ch: 2 tiles of 16 output
//     channels): a 32-channel K step is 8 LDS fragments (ds_read_b128, each feeding two MFMAs) and 2 weight fragments (each feeding
//     eight) for 16 MFMAs -- per conv and CU 1.15 MB of LDS reads (half of what 256 B/clk carries in the MFMA time) and 0.29 MB
//     of weight fragments through the vector L1 (half of its 64 B/clk; the four waves of a channel half request the same lines);
//   * the weights are ONE stream over all blocks in consumption order, global -> registers in fragment order, each fragment
//     re-requested D steps ahead right after its last MFMA;
//   * pixel pitch 128 B with the 16-byte slots XOR-swizzled by ((column >> 1) & 3) << 1: under the gfx950 ds_read_b128 lane
//     grouping ({0-3,12-15,20-27}, ... : MI355X_MICROARCH.md, LDS) the 16 lanes of a group cover all 64 banks once for any
//     tap column;
//   * MFMA order per output element = resblock_lat.hip's (tap-major, K half minor, same operand roles), bias / ReLU / skip and the
//     bf16 roundings at the same points: results are BIT-IDENTICAL to nb x tg_resblock.
// Requires all workgroups co-resident: ntiles <= number of CUs (checked on the host); every spin is bounded (give-ups in ctrl[2]).
#include "common.h"
#include "handoff.h"

#define RP_MAXB 16

struct RpP {
  const void* x;                // [N,H,W,64] bf16
  const void* w1[RP_MAXB];      // per block: fragment-order weights (tg_pack_weights_frag) of conv_1
  const void* w2[RP_MAXB];      // ... of conv_2
  const float* b1[RP_MAXB];     // nullable
  const float* b2[RP_MAXB];     // nullable
  void* out;                    // [N,H,W,64] bf16: output of the last block
  const void* pre_x;            // nullable: [N,H,W,pre_cpad] bf16 -- the generator input; the input-stage conv + ReLU (lib/frvsr.py:
  const void* pre_w;            //   60-63) runs in this launch in front of the first block (x is ignored then); pre_w: fragment-order
  const float* pre_b;           //   [tap][64][64-padded Cin] copy
  int pre_cpad;
  unsigned* ctrl;               // [0] epoch base  [1] arrivals  [2] give-ups (sticky)
  unsigned long long* xccw;     // [tiles] {tag, XCC id} words
  void* gran;                   // granule rings P, S: 2 x 2 slots x [tiles] x RP_RING
  int nb, N, H, W;
  int tiles_y, tiles_x, ntiles, nwg;
  unsigned bytes;               // extent of the [N,H,W,64] tensors
  unsigned gslot;               // bytes of one ring slot = tiles * RP_RING
  unsigned spin_limit;
  int prio;
};

namespace {
constexpr int RP_PW = 34, RP_PH = 18;                 // plane: 18 rows x 34 columns of 128-byte pixels
constexpr int RP_PLANE = RP_PH * RP_PW * 128;         // 78336
constexpr int RP_ROW_T = 0, RP_ROW_B = 8192, RP_COL = 16384;
constexpr int RP_RING = 24576;                        // per tile and slot: top / bottom row [half 2][ct 4][fg 4][px 16], columns [wave 8][side 2][r 4][j 2][fg 4], x 16 B
constexpr int RP_STG = 2 * RP_PLANE;                  // column staging: [wave 8][side 2][r 4][j 2][fg 4] x 8 payload bytes
constexpr int RP_LDS = RP_STG + 8 * 512;              // 160768
}  // namespace

#ifdef TG_RP_TRACE
// [wave 0..7][block 0..15][stamp 0..15] of the middle workgroup
__device__ unsigned long long tg_rp_trace_buf[8 * 16 * 16];
#define RP_STAMP(k, i)                                                                                                  \
  do {                                                                                                                  \
    if (blockIdx.x == gridDim.x / 2 && lane == 0) tg_rp_trace_buf[(wave * 16 + (k)) * 16 + (i)] = (unsigned long long)clock64(); \
  } while (0)
extern "C" int tg_debug_rp_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_rp_trace_buf), sizeof(unsigned long long) * 8 * 16 * 16);
}
#else
#define RP_STAMP(k, i) do { } while (0)
#endif

typedef float f32x2p __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2p __attribute__((ext_vector_type(2)));
// two fp32 -> two bf16 in one v_cvt_pk_bf16_f32 (round to nearest even, as f2bf)
__device__ __forceinline__ unsigned rp_cvt2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2p{a, b}, bf16x2p));
}

// byte offset of 16-byte slot `slot` (8 channels) of plane position (r, c)
__device__ __forceinline__ int rp_lds(int r, int c, int slot) { return (r * RP_PW + c) * 128 + ((slot ^ (((c >> 1) & 3) << 1)) << 4); }

// D: prefetch distance of the weight stream in K steps (divides 18); L: LDS fragment look-ahead
// PRE: the input-stage conv of generator_F in front of the first block: one more conv pass (generator input, staged into the M plane
// with its channels zero-padded to 64, -> relu(conv + b) -> X) and one more hand-off (the ring of X)
template <int D, int L, bool PRE = false>
__global__ __launch_bounds__(512) void resblock_plane_kernel(RpP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(18 % D == 0 && L >= 1 && L <= 16, "weight ring slots are reused across convs");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pq = wave >> 1, ch = wave & 1;
  const int frow = lane & 15, fg = lane >> 4;

  int b = blockIdx.x;
  b = (b & 7) * (p.nwg >> 3) + (b >> 3);                 // an XCD owns a contiguous range of tiles
  const unsigned epoch0 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  const int nb = p.nb;
  if (b < p.ntiles) {
    const int tx = b % p.tiles_x, t1 = b / p.tiles_x;
    const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
    const int y0 = ty * 16, x0 = tx * 32;
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc(p.gran, 0, (int)(4u * p.gslot), 0x00020000);
    const unsigned my_xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;   // HW_REG_XCC_ID[3:0]
    if (tid == 0)
      __hip_atomic_store(p.xccw + b, ((unsigned long long)(epoch0 + 1u) << 32) | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned limit = p.spin_limit;

    // ---- the input plane: 612 positions x 8 slots = 4896 16-byte items, 10 per thread (outside the image: zeros) --------------
    const int xpix = PRE ? p.pre_cpad * 2 : 128;          // bytes per pixel of the tensor staged first (PRE: the chunks beyond it are zeros)
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(PRE ? p.pre_x : p.x), 0,
                                                       (int)((unsigned)(p.N * p.H * p.W) * (unsigned)xpix), 0x00020000);
    u32x4c xr[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int item = tid + k * 512;
      const int pos = item >> 3, slot = item & 7;
      const int r = pos / RP_PW, c = pos % RP_PW;
      const int gy = y0 - 1 + r, gx = x0 - 1 + c;
      const bool ok = item < RP_PH * RP_PW * 8 && slot * 16 < xpix && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(ok ? (unsigned)(((n * p.H + gy) * p.W + gx) * xpix + slot * 16) : RC_OOB), 0, 0);
    }

    // ---- the weight stream: position g of a block: g < 18 K step g of conv_1, else step g - 18 of conv_2; 36 .. 53 = the next
    //      block's conv_1 (same registers: the slot of step s is re-requested for step s + D right after its last MFMA)
    auto rsrc_w = [&](const void* w) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, w ? 9 * 64 * 64 * 2 : 0, 0x00020000); };
    auto rsrc_b = [&](const float* q) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q), 0, q ? 256 : 0, 0x00020000); };
    // (PRE: the input conv takes the place of a "second conv" in front of block 0: positions 18 .. 35, block 0's conv_1 behind it)
    auto rsW1 = rsrc_w(PRE ? nullptr : p.w1[0]), rsW2 = rsrc_w(PRE ? p.pre_w : p.w2[0]);
    auto rsW1n = rsrc_w(PRE ? p.w1[0] : (nb > 1 ? p.w1[1] : nullptr));
    u32x4c wq[D][2];
    const int wlane = ch * 2048 + lane * 16;             // this wave's two channel tiles: 2 ch, 2 ch + 1
#define RP_WISSUE(g)                                                                                                           \
  do {                                                                                                                         \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                                         \
      if constexpr ((g) < 18) wq[(g) % D][j_] = __builtin_amdgcn_raw_buffer_load_b128(rsW1, wlane + j_ * 1024, (g) * 4096, 0);  \
      else if constexpr ((g) < 36) wq[(g) % D][j_] = __builtin_amdgcn_raw_buffer_load_b128(rsW2, wlane + j_ * 1024, ((g) - 18) * 4096, 0); \
      else wq[(g) % D][j_] = __builtin_amdgcn_raw_buffer_load_b128(rsW1n, wlane + j_ * 1024, ((g) - 36) * 4096, 0);             \
    }                                                                                                                          \
  } while (0)
    rc_static_for<0, D>([&](auto i) { RP_WISSUE((PRE ? 18 : 0) + decltype(i)::value); });

    // ---- who is where: lanes 0 .. 7 of every wave read the XCC word of neighbour d = lane --------------------------------------
    //      d: 0 up, 1 down, 2 left, 3 right, 4 up-left, 5 up-right, 6 down-left, 7 down-right
    auto nbtile = [&](int d) {
      const int dy = d == 0 || d == 4 || d == 5 ? -1 : (d == 1 || d == 6 || d == 7 ? 1 : 0);
      const int dx = d == 2 || d == 4 || d == 6 ? -1 : (d == 3 || d == 5 || d == 7 ? 1 : 0);
      const bool ok = (unsigned)(ty + dy) < (unsigned)p.tiles_y && (unsigned)(tx + dx) < (unsigned)p.tiles_x;
      return ok ? (n * p.tiles_y + ty + dy) * p.tiles_x + tx + dx : -1;
    };
    unsigned cross = 0;                                  // bit d: neighbour d exists and runs on another XCD
    {
      const auto rsXW = __builtin_amdgcn_make_buffer_rsrc(p.xccw, 0, p.ntiles * 8, 0x00020000);
      const int t = lane < 8 ? nbtile(lane) : -1;
      const unsigned off = t >= 0 ? (unsigned)t * 8u : RC_OOB;
      u32x2c xw = u32x2c{0u, 0u};
      bool got = false;
      const unsigned lim = __builtin_amdgcn_readfirstlane(limit);
      for (unsigned spins = 0; spins <= lim; ++spins) {
        asm volatile("" ::: "memory");
        xw = __builtin_amdgcn_raw_buffer_load_b64(rsXW, (int)off, 0, RC_SC1);
        if (!__any(off != RC_OOB && xw.y != epoch0 + 1u)) { got = true; break; }
        if (spins > 32) __builtin_amdgcn_s_sleep(32);
        else __builtin_amdgcn_s_sleep(1);
      }
      if (!got) {
        limit = 0;
        if (lane == 0) __hip_atomic_fetch_add(p.ctrl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      cross = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__ballot(off != RC_OOB && (xw.x & 15u) != my_xcc)) & 0xffu;
    }
    auto is_cross = [&](int d) { return ((cross >> d) & 1u) != 0u; };

    // ---- the sweep: ALL 1600 items belong to the first four waves (7 per thread: 2 of the top row, 2 of the bottom row, one of each
    //      side column, a corner), recomputed at every sweep (no registers held across the matrix phases).  The two waves of a SIMD
    //      do not interleave -- the older one runs its MFMAs first (lesson 40) -- so waves 0..3 are through their conv and epilogue
    //      thousands of cycles before waves 4..7 and poll in what would be barrier time, while the late waves go from their
    //      epilogue straight to the barrier instead of starting a sweep of their own then (round trip + delivery off the critical path)
    auto sweep_items = [&](unsigned (&goff)[7], int (&lpos)[7]) {
      auto item = [&](int k, int d, int src, int r, int c, int q) {
        const int t = nbtile(d);
        goff[k] = t >= 0 ? (unsigned)t * (unsigned)RP_RING + (unsigned)src + (is_cross(d) ? 2u * p.gslot : 0u) : RC_OOB;
        lpos[k] = rp_lds(r, c, q >> 1) + (q & 1) * 8;
      };
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 256;                      // a row in the order it was stored: [half][ct][fg][px]
        const int px = i & 15, q = ((i >> 6) & 3) * 4 + ((i >> 4) & 3), c = 1 + ((i >> 8) & 1) * 16 + px;
        item(u, 0, RP_ROW_B + i * 16, 0, c, q);            // the upper neighbour's bottom row
        item(2 + u, 1, RP_ROW_T + i * 16, 17, c, q);
      }
      {
        // a side column in the order it was stored: [wave][side][r][j][fg] (the left halo = the left neighbour's RIGHT side)
        const int i = tid, wv = i >> 5, row = 4 * (wv >> 1) + ((i >> 3) & 3), q = ((wv & 1) * 2 + ((i >> 2) & 1)) * 4 + (i & 3);
        item(4, 2, RP_COL + wv * 1024 + 512 + (i & 31) * 16, row + 1, 0, q);
        item(5, 3, RP_COL + wv * 1024 + (i & 31) * 16, row + 1, 33, q);
      }
      {
        const int cid = (tid >> 4) & 3, q = tid & 15;
        const int src = (cid < 2 ? RP_ROW_B : RP_ROW_T) + ((cid & 1) == 0 ? 4096 + 240 : 0) + (q >> 2) * 1024 + (q & 3) * 256;
        item(6, 4 + cid, src, cid < 2 ? 0 : 17, (cid & 1) == 0 ? 0 : 33, q);
        if (tid >= 64) goff[6] = RC_OOB;
      }
    };
    // ---- what this lane publishes ----------------------------------------------------------------------------------------------
    // rows (waves of pq 0: the top row, pq 3: the bottom row): granule of pixel tile (row, half h), channel tile 2 ch + j at
    // rowbase + h * 4096 + j * 1024, straight from the epilogue registers (a store instruction = one contiguous KiB).
    // columns (every wave: 4 rows x 32 channels of the left and of the right column): the 8 lanes that hold them (frow 0 from the
    // h = 0 tiles, frow 15 from the h = 1 tiles) put the payload into the wave's 512-byte LDS staging area, all 64 lanes read it back
    // in storage order and store ONE contiguous KiB (whole lines: lesson 34; the first form stored 64-byte pieces straight from those
    // lanes, and its wrong pixels -- first read as torn granules -- were the store-data hazard described at the row stores below).
    const bool rowpub = pq == 0 || pq == 3;
    const unsigned rowbase = (unsigned)b * RP_RING + (pq == 0 ? RP_ROW_T : RP_ROW_B) + ch * 2048 + lane * 16;
    bool rowS[2];
    {
      const bool cv = is_cross(pq == 0 ? 0 : 1), cl = is_cross(pq == 0 ? 4 : 6), cr = is_cross(pq == 0 ? 5 : 7);
      rowS[0] = rowpub && (cv || (frow == 0 && cl));
      rowS[1] = rowpub && (cv || (frow == 15 && cr));
    }
    const bool colsrc = frow == 0 || frow == 15;
    unsigned char* const stg = smem + RP_STG + wave * 512;
    unsigned char* const stg_w = stg + ((frow == 0 ? 0 : 32) + fg) * 8;        // + (r * 8 + j * 4) * 8
    const unsigned colbase = (unsigned)b * RP_RING + RP_COL + wave * 1024 + lane * 16;
    const bool colS = is_cross(lane < 32 ? 2 : 3);
    const unsigned sring = 2u * p.gslot;

    // ---- LDS addresses -----------------------------------------------------------------------------------------------------------
    // fragment of pixel tile (r, h), tap (ky, kx), K half kk: plane rows 4 pq + r + ky, columns 16 h + frow + kx
    int fa[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fa[kx][kk] = rp_lds(4 * pq, frow + kx, kk * 4 + fg);
    // epilogue element of this lane in pixel tile (r, h), channel tile j: + (r * 34 + 16 h) * 128 + j * 32 (slot 4 ch + 2 j + fg / 2:
    // the swizzle touches slot bits 1, 2 -- j is bit 1 -> XOR instead of add)
    const int ea = rp_lds(4 * pq + 1, frow + 1, 4 * ch + (fg >> 1)) + (fg & 1) * 8;
    const int gyb = y0 + 4 * pq, gxb = x0 + frow;

    // ---- stage the input plane, clear the other one (PRE: the generator input goes to M, X starts as zeros) -----------------------
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int item = tid + k * 512;
      const int pos = item >> 3, slot = item & 7;
      if (item < RP_PH * RP_PW * 8) {
        *reinterpret_cast<u32x4c*>(smem + (PRE ? RP_PLANE : 0) + rp_lds(pos / RP_PW, pos % RP_PW, slot)) = xr[k];
        *reinterpret_cast<u32x4c*>(smem + (PRE ? 0 : RP_PLANE) + item * 16) = u32x4c{0u, 0u, 0u, 0u};
      }
    }
    __syncthreads();
    if (p.prio) __builtin_amdgcn_s_setprio(3);

    constexpr unsigned EB = PRE ? 1u : 0u;               // hand-offs in front of block 0
    int k = 0;
    bool last = false;
    u32x4c bq1[2], bq2[2];
    auto load_b = [&](u32x4c (&bq)[2], const float* b) {
      bq[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b(b), (ch * 32 + fg * 4) * 4, 0, 0);
      bq[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b(b), (ch * 32 + 16 + fg * 4) * 4, 0, 0);
    };
    // one conv: MODE 0: X -> relu(. + b1) -> M; 1: M -> . + b2 + X (skip) -> X in place; 2 (PRE): M -> relu(. + pre_b) -> X
    auto conv = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool SECOND = MODE == 1, FROM_M = MODE != 0;
        const unsigned char* src = smem + (FROM_M ? RP_PLANE : 0);
        unsigned char* dst = smem + (FROM_M ? 0 : RP_PLANE);
        constexpr int G0 = FROM_M ? 18 : 0;
        RP_STAMP(k, SECOND ? 5 : 0);
        f32x4 acc[8][2];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto rd = [&](auto qv) {
          constexpr int q = decltype(qv)::value;
          constexpr int s = q >> 3, i = q & 7, r = i >> 1, h = i & 1;
          constexpr int tap = s >> 1, kk = s & 1, ky = tap / 3, kx = tap % 3;
          return *reinterpret_cast<const uint4*>(src + fa[kx][kk] + ((r + ky) * RP_PW + 16 * h) * 128);
        };
        uint4 fr[L];
        rc_static_for<0, L>([&](auto qv) { fr[decltype(qv)::value] = rd(qv); });
        rc_static_for<0, 144>([&](auto qv) {
          constexpr int q = decltype(qv)::value;
          constexpr int s = q >> 3, i = q & 7;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wq[s % D][j]), *reinterpret_cast<bf16x8*>(&fr[q % L]),
                                                                acc[i][j], 0, 0, 0);
          if constexpr (q + L < 144) fr[q % L] = rd(std::integral_constant<int, (q + L < 144 ? q + L : 0)>{});
          if constexpr (i == 7) RP_WISSUE(G0 + s + D);
          __builtin_amdgcn_sched_barrier(0);
        });
        RP_STAMP(k, SECOND ? 6 : 1);

        // ---- epilogue: bias, ReLU / skip, one rounding; pixels outside the image are the next conv's zero padding.  All 16 results
        //      first (the skip operands requested up front), then the publishes, then the LDS writes: what the neighbours wait for
        //      leaves as early as it can
        const bool pub = !(SECOND && last);
        const unsigned e = MODE == 2 ? 0u : EB + 2u * (unsigned)k + (SECOND ? 1u : 0u);
        const unsigned tag = epoch0 + e + 1u;
        const unsigned soff = (e & 1u) * p.gslot;
        float bv[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const u32x4c q4 = SECOND ? bq2[j] : bq1[j];
          bv[j][0] = __uint_as_float(q4.x); bv[j][1] = __uint_as_float(q4.y);
          bv[j][2] = __uint_as_float(q4.z); bv[j][3] = __uint_as_float(q4.w);
        }
        auto elem = [&](int r, int h, int j) { return dst + (ea ^ (j * 32)) + (r * RP_PW + 16 * h) * 128; };
        u32x2c o[4][2][2];
        if constexpr (SECOND) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int j = 0; j < 2; ++j) o[r][h][j] = *reinterpret_cast<const u32x2c*>(elem(r, h, j));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const bool inimg = gyb + r < p.H && gxb + 16 * h < p.W;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              float v[4];
              if constexpr (SECOND) {
                float sk[4];
                rc_unpack4(o[r][h][j], sk);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  v[c] = acc[r * 2 + h][j][c] + bv[j][c];
                  v[c] += sk[c];
                }
              } else {
                // ReLU as resblock_lat.hip's fmaxf(v, v * 0) bit for bit (-0 for a negative sum, +0 for +0: v_max_f32 orders -0 < +0)
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(acc[r * 2 + h][j][c] + bv[j][c], -0.f);
              }
              o[r][h][j] = inimg ? u32x2c{rp_cvt2(v[0], v[1]), rp_cvt2(v[2], v[3])} : u32x2c{0u, 0u};
            }
          }
        RP_STAMP(k, SECOND ? 7 : 2);
        if (pub) {
          if (rowpub) {
            const int rr = pq == 0 ? 0 : 3;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const u32x2c ov = rr == 0 ? o[0][h][j] : o[3][h][j];
                const u32x4c gr = u32x4c{ov.x, tag, ov.y, tag};
                // (the slot offset goes into the VECTOR offset, soffset = 0: with an SGPR soffset hipcc leaves out the wait state
                //  between a 16-byte store and the next write of its data registers -- LLVM's hazard recognizer holds that MUBUF
                //  hazard not to exist then -- and on gfx950 the odd hand-offs (slot 1) then published granules whose last lanes
                //  carried the NEXT granule's payload under the right tags: profiles/r06aj_trace_plane.txt, lesson 41)
                const unsigned off = rowbase + h * 4096 + j * 1024 + soff;
                __builtin_amdgcn_raw_buffer_store_b128(gr, rsG, (int)(rowS[h] ? off + sring : RC_OOB), 0, RC_SC1);
                __builtin_amdgcn_raw_buffer_store_b128(gr, rsG, (int)off, 0, 0);
              }
          }
          if (colsrc) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x2c*>(stg_w + (r * 8 + j * 4) * 8) = frow == 0 ? o[r][0][j] : o[r][1][j];
          }
          __builtin_amdgcn_wave_barrier();                 // same-wave LDS operations are ordered; keep the compiler from mixing them
          const u32x2c oc = *reinterpret_cast<const u32x2c*>(stg + lane * 8);
          const u32x4c gr = u32x4c{oc.x, tag, oc.y, tag};
          __builtin_amdgcn_raw_buffer_store_b128(gr, rsG, (int)(colS ? colbase + soff + sring : RC_OOB), 0, RC_SC1);
          __builtin_amdgcn_raw_buffer_store_b128(gr, rsG, (int)(colbase + soff), 0, 0);
          __builtin_amdgcn_wave_barrier();
        }
        RP_STAMP(k, SECOND ? 8 : 3);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x2c*>(elem(r, h, j)) = o[r][h][j];
        RP_STAMP(k, SECOND ? 9 : 4);
        // ---- hand-off: the ring of the destination plane from the neighbours' same conv ------------------------------------------
        if (pub && limit && wave < 4) {
          unsigned goff[7];
          int lpos[7];
          sweep_items(goff, lpos);
          if (!rc_sweep<7, 0>(rsG, goff, lpos, dst, soff, tag, limit, nullptr)) {
            limit = 0;
            if (lane == 0) __hip_atomic_fetch_add(p.ctrl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (p.prio) __builtin_amdgcn_s_setprio(3);
        }
        __syncthreads();
    };
    if constexpr (PRE) {
      load_b(bq1, p.pre_b);
      conv(std::integral_constant<int, 2>{});
      rsW1 = rsW1n;
      rsW2 = rsrc_w(p.w2[0]);
      rsW1n = rsrc_w(nb > 1 ? p.w1[1] : nullptr);
    }
    for (k = 0; k < nb; ++k) {
      last = k + 1 >= nb;
      load_b(bq1, p.b1[k]);
      load_b(bq2, p.b2[k]);
      conv(std::integral_constant<int, 0>{});
      conv(std::integral_constant<int, 1>{});
      RP_STAMP(k, 10);
      rsW1 = rsW1n;
      rsW2 = rsrc_w(last ? nullptr : p.w2[last ? 0 : k + 1]);
      rsW1n = rsrc_w(k + 2 < nb ? p.w1[k + 2 < nb ? k + 2 : 0] : nullptr);
    }
#undef RP_WISSUE

    // ---- the result: the centre of X, whole 128-byte pixels ----------------------------------------------------------------------
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int item = tid + k * 512;
      const int px = item >> 3, slot = item & 7;
      const int r = px >> 5, c = px & 31;
      const int gy = y0 + r, gx = x0 + c;
      const u32x4c v = *reinterpret_cast<const u32x4c*>(smem + rp_lds(r + 1, c + 1, slot));
      __builtin_amdgcn_raw_buffer_store_b128(v, rsO, (int)(gy < p.H && gx < p.W ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + slot * 16) : RC_OOB), 0, 0);
    }
  }

  // ---- arrival: the last workgroup advances the epoch base for the next launch -------------------------------------------------
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(p.ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)p.nwg - 1u) {
      __hip_atomic_store(p.ctrl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.ctrl, epoch0 + 2u * (unsigned)nb + (PRE ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static int64_t rp_tiles(int N, int H, int W) { return (int64_t)N * ((H + 15) / 16) * ((W + 31) / 32); }

extern "C" int tg_resblock_plane_scratch_bytes(int N, int H, int W, int64_t* bytes) {
  TG_CHECK_ARG(bytes && N > 0 && H > 0 && W > 0, "null pointer / empty tensor");
  const int64_t nt = rp_tiles(N, H, W);
  *bytes = 256 + ((nt * 8 + 255) / 256) * 256 + 4 * nt * RP_RING;      // control words, XCC words, rings P and S (2 slots each)
  return TG_OK;
}

extern "C" int tg_resblock_plane(const void* x, int nblocks, const void* const* w1, const float* const* b1, const void* const* w2,
                                 const float* const* b2, void* out, void* scratch, const void* pre_x, int pre_cpad, const void* pre_w_frag,
                                 const float* pre_b, int N, int H, int W, int C, int dtype, int variant, void* stream) {
  TG_CHECK_ARG(dtype == TG_BF16 && C == 64, "bf16 tensors with 64 channels only");
  TG_CHECK_ARG(nblocks >= 1 && nblocks <= RP_MAXB, "1 .. 16 blocks per launch");
  TG_CHECK_ARG((x || pre_x) && w1 && w2 && out && scratch && N > 0 && H > 0 && W > 0, "null pointer / empty tensor");
  TG_CHECK_ARG(!pre_x || (pre_w_frag && pre_cpad >= 8 && pre_cpad <= 64 && pre_cpad % 8 == 0),
               "the input-stage conv in front of the trunk: fragment-order weights, 8 .. 64 padded input channels");
  TG_CHECK_ARG((((uintptr_t)x | (uintptr_t)out | (uintptr_t)scratch | (uintptr_t)pre_x | (uintptr_t)pre_w_frag) & 15) == 0,
               "pointers must be 16-byte aligned");
  const int64_t bytes = (int64_t)N * H * W * 128;
  TG_CHECK_ARG(bytes < ((int64_t)1 << 31), "tensor too large for 32-bit buffer offsets");
  RpP p;
  p.x = x; p.out = out;
  p.pre_x = pre_x; p.pre_w = pre_w_frag; p.pre_b = pre_b; p.pre_cpad = pre_cpad;
  for (int k = 0; k < RP_MAXB; ++k) {
    const bool on = k < nblocks;
    p.w1[k] = on ? w1[k] : nullptr; p.w2[k] = on ? w2[k] : nullptr;
    p.b1[k] = on && b1 ? b1[k] : nullptr; p.b2[k] = on && b2 ? b2[k] : nullptr;
    if (on) {
      TG_CHECK_ARG(p.w1[k] && p.w2[k], "null per-block pointer");
      TG_CHECK_ARG((((uintptr_t)p.w1[k] | (uintptr_t)p.w2[k]) & 15) == 0, "pointers must be 16-byte aligned");
    }
  }
  p.ctrl = static_cast<unsigned*>(scratch);
  p.nb = nblocks; p.N = N; p.H = H; p.W = W;
  p.tiles_y = (H + 15) / 16; p.tiles_x = (W + 31) / 32;
  const int64_t nt = rp_tiles(N, H, W);
  TG_CHECK_ARG(nt <= tg_num_cus(), "more tiles than compute units: the hand-offs need every workgroup resident (use tg_resblock_c64_thr)");
  TG_CHECK_ARG(nt * RP_RING * 4 < ((int64_t)1 << 31), "ring too large for 32-bit buffer offsets");
  p.ntiles = (int)nt;
  p.nwg = (int)((nt + 7) / 8 * 8);
  p.bytes = (unsigned)bytes;
  p.gslot = (unsigned)(nt * RP_RING);
  p.xccw = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(scratch) + 256);
  p.gran = static_cast<unsigned char*>(scratch) + 256 + ((nt * 8 + 255) / 256) * 256;
  p.spin_limit = 1u << 16;
  p.prio = 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double px = (double)N * H * W;
  const double fl = 2.0 * 2.0 * px * 64.0 * 576.0 * nblocks + (pre_x ? 2.0 * px * 64.0 * 9.0 * pre_cpad : 0.0);
  const double by = px * (128.0 + (pre_x ? 2.0 * pre_cpad : 128.0)) + (nblocks * 2.0 + (pre_x ? 1.0 : 0.0)) * 73728.0;
  constexpr int LDS = RP_LDS;
  auto go = [&](auto dtag, auto ltag, auto ptag) {
    constexpr int D = decltype(dtag)::value, L = decltype(ltag)::value;
    constexpr bool PRE = decltype(ptag)::value;
    auto kern = resblock_plane_kernel<D, L, PRE>;
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      attr_done = true;
    }
    TG_LAUNCH(PRE ? "resblock_plane<in>" : "resblock_plane", fl, by, kern, dim3(p.nwg), dim3(512), LDS, st, p);
  };
  using I6 = std::integral_constant<int, 6>;
  using I8 = std::integral_constant<int, 8>;
  using I9 = std::integral_constant<int, 9>;
  if (pre_x) go(I6{}, I8{}, std::true_type{});
  else if (variant == 1) go(I9{}, I8{}, std::false_type{});
  else go(I6{}, I8{}, std::false_type{});
  TG_CHECK_LAUNCH();
}
