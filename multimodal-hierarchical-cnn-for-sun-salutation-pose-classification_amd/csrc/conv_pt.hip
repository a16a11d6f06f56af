// 3x3 / stride-1 convolution of the 28x28, 14x14 and 7x7 stages (layer2..4 of the ResNet-18 stack, forward and
// data gradient) with the INPUT PATCH RESIDENT IN LDS and a ping-pong MFMA schedule.
//
// Replaces the same ATen calls as conv_igemm.hip (conv2d forward / backward-input reached from
// /root/reference/Quadtree_from scratch/models.py:222-243,289 and Quadtree_train.py:65).
//
// What bounds the implicit GEMM (conv_igemm.hip, and the ping-pong variant of it measured in round 2):
// a CU moves ~65-70 GB/s from L2 into LDS, and an implicit GEMM stages every input pixel once PER TAP:
// 128x128 tiles 64 FLOP per staged byte, 256x128 tiles 85 -> 4.5-6 TFLOP/s per CU however the MFMAs are scheduled.
// Here a workgroup owns 196 output pixels that form whole image rows --
//     28x28 maps: a quarter image (7 rows),   14x14: one image,   7x7: four images
// (256 images -> exactly 1024 / 256 / 64 pixel tiles: every CU busy, no ragged last round) -- and keeps their input
// patch (rows + halo, zero rows for the padding) for one 128-byte channel chunk in LDS: all nine taps read it at a
// constant row shift.  Per K-tile (one tap of one chunk) only the weight tile is staged (+ 1/9 of the next patch):
// 196x256 outputs: 176 FLOP/B, 196x128: 160 FLOP/B.  The MFMA pipe becomes the bound.
//
// Schedule (as the ring kernels of MI355X_MICROARCH.md, two wave groups instead of loader / consumer waves):
// 8 waves = 2 pixel halves x 4 channel quarters; waves 0-3 (group 0) and 4-7 (group 1) share the SIMDs pairwise and run
// the same program one barrier apart: while one group issues the MFMAs of K-tile t the other issues LDS-DMA (weight
// tile t+D, one 64-row pass of the next chunk's patch) and reads its fragments of K-tile t.  Counted vmcnt waits keep
// D = 2..3 weight tiles and the patch passes in flight across the raw s_barriers.
//   group 0:  b0 | L0 | b1 | C0 | b2 | L1 | b3 | C1 | ...
//   group 1:  b0 |    | b1 | L0 | b2 | C0 | b3 | L1 | ...
//   RAW  a wave waits for ITS DMA of weight tile t+1 (and of everything older: vmcnt is in issue order, the patch passes
//        of the next chunk are older than its first weight tile) at the end of L_t, before a barrier; group 0 reads
//        K-tile t after b_2t, by which both groups have passed such a wait;
//   WAR  weight tile t+D lands in the slot of tile t-1, patch chunk c+1 in the buffer of chunk c-1; both are issued
//        after b_2t / b_18c, and the last reads of the old contents (group 1's L_{t-1}) were retired by lgkmcnt(0)
//        before that barrier.
// The epilogue runs straight from the accumulators (the weight rows are staged permuted so that a lane owns eight
// consecutive channels of a pixel): no LDS round trip.
#include <stdlib.h>

#include <type_traits>

#include "conv_args.h"

namespace {

using qtc::ConvArgs;

constexpr int kNT = 512;        // 8 waves: two per SIMD, one of each group
constexpr int kKB = 128;        // bytes of K per row and K-tile (one channel chunk)
// Tile rows are PATCH POSITIONS (pad positions are computed and discarded), so a tap is one row shift for all rows and
// 16-row tiles are a constant 2 KB apart: one fragment address per K sub-step.  196 output pixels per workgroup:
//   GEO_ROWS   R rows of one W-wide image, row pitch PW = W + 2 (28x28: R = 7 -> 210 positions, 14x14: R = 14 -> 224):
//              224 rows = 2 wave rows x 7 tiles;
//   GEO_STACK  four 7x7 images, row pitch 8 (the right pad of a row is the left pad of the next), image pitch 64 (the
//              bottom pad row of an image is the top pad row of the next): 256 rows = 2 x 8 tiles.
enum { GEO_ROWS = 0, GEO_STACK = 1 };
constexpr int kMaxPass = 5;     // 64-row DMA passes of a patch (<= 320 positions)

struct PtArgs {
  ConvArgs c;
  int G, R, W, H;               // sub-images per tile, rows per sub-image part, image width / height
  int PW, PP, npos, npass;      // patch: row pitch W+2, positions per sub-image (R+2)*(W+2), G*PP, passes of 64 rows
  int tpi, tiles_m, batch;      // pixel tiles per image (GEO_ROWS), pixel tiles in all, images
  int items, ipw;               // (channel tile, pixel tile) items in all / per workgroup (persistent walk)
#ifdef QT_KERNEL_PROF
  unsigned long long* prof;     // experiment build only (qt_set_pt_prof): [workgroup][32] s_memrealtime stamps (10 ns)
#endif
  int stagger;                  // start delay of workgroup class k = (blockIdx / 8) % 4: k * stagger * 1024 cycles (0: none)
  int nchunks;                  // 128-byte channel chunks of the source
  unsigned src_bytes, wgt_bytes; // extents of the two operands (buffer resources: range-checked DMA)
  // GEO_STACK on the 2x2 quadrants of 14x14 maps (the quadrant conv, Quadtree_from scratch/models.py:277-287): "image"
  // n*4 + q is quadrant q of map n.  1: forward, the SOURCE is the un-split map (zero halo at the seam comes for free:
  // the patch's pad positions); 2: data gradient, the DESTINATION is the un-split map.
  int quad;
  FastDiv div_pw, div_hw, div_w;   // (div_hw / div_w: merged destination mapping)
};

// eight consecutive elements as loaded (decoded to f32 only where they are consumed)
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  uint4 u;
  __device__ __forceinline__ void load(const bf16_t* p) { u = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void get(float (&f)[8]) const {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
  }
};
template <> struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const float4*>(p);
    b = *reinterpret_cast<const float4*>(p + 4);
  }
  __device__ __forceinline__ void get(float (&f)[8]) const {
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
};

// two / four transfers 8 KiB apart in LDS (consecutive 64-row passes of a tile), M0 saved once
__device__ __forceinline__ void blds16x2(const i32x4& rsrc, unsigned v0, unsigned v1, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
      "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "s"(rsrc), "s"(soff), "s"(lds_addr)
      : "memory", "scc");
}
__device__ __forceinline__ void blds16x4(const i32x4& rsrc, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned soff,
                                         unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, %6 offen lds\n\t"
      "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %6 offen lds\n\t"
      "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
      "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(rsrc), "s"(soff), "s"(lds_addr)
      : "memory", "scc");
}

// DMA instructions a wave issues in the L segment of tap `tp` (-1: tap 8 of the previous chunk, or the prologue's last
// weight tile): one patch pass of the next chunk on the first NPASS taps (never in the last chunk) + RW for weight tile
// t+D (not on the last D taps of the last chunk)
template <int NPASS, int RW, int D>
constexpr int issued_in(int tp, bool last) {
  if (tp < 0) return RW;
  return ((!last && tp < NPASS) ? 1 : 0) + ((!last || tp + D < 9) ? RW : 0);
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose counter is a constant expression
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

// patch passes of the NEXT chunk issued in the L segment of tap `tap`: all of them no later than (and, inside their segment,
// before) weight tile 0 of the next chunk, which is issued in L_{NTAPS-D}.  Nine taps: one pass on each of the first NPASS
// taps; four taps (the merged stride-2 data gradient): two on the first one or two taps, one on the next.
template <int NTAPS, int NPASS, int D>
constexpr int passes_in(int tap) {
  if (NTAPS == 9) return tap < NPASS ? 1 : 0;
  const int slots = NTAPS - D + 1;                       // segments 0 .. NTAPS-D
  if (tap >= slots) return 0;
  const int base = NPASS / slots, extra = NPASS - base * slots;
  return base + (tap < extra ? 1 : 0);
}
template <int NTAPS, int NPASS, int D>
constexpr int passes_before(int tap) {
  int n = 0;
  for (int t = 0; t < tap; ++t) n += passes_in<NTAPS, NPASS, D>(t);
  return n;
}

// NTAPS = 9: the 3x3 / stride-1 convolution (forward, or with DGRAD its data gradient: taps mirrored).
// NTAPS = 4 ("merged"): the data gradient of a 3x3 / STRIDE-2 convolution as one 2x2-tap gather over the gradient map
// (qt_conv_desc.dst_merge, operand qt_pack_dgrad_s2_merged): the window (r..r+1, c..c+1) = taps (1,1) (1,2) (2,1) (2,2) of
// this kernel's pad-1 numbering, the 4 C output channels are the four parity classes of destination pixel
// (2r + class/2, 2c + class%2) of the C-channel map, where residual / mask / BatchNorm links are read.  All four slots are
// multiplied for every class (the slots a class does not use hold zero weights: 16/9 of the products) -- still 2-3 x
// faster than the generic tile on these short-K launches.
template <typename T, int BN, int NBW, int NPASS, int GEO, bool DGRAD, int NTAPS = 9, bool KS = false>
__global__ __launch_bounds__(kNT, 2) void conv_pt_kernel(PtArgs q) {
  static_assert(NTAPS == 9 || (NTAPS == 4 && !DGRAD && BN == 128 && NBW == 3), "tap sets");
  static_assert(NPASS >= 1 && NPASS <= kMaxPass && (NTAPS != 9 || NPASS - 1 <= 9 - (NBW - 1)),
                "the last patch pass is issued no later than (and, in its L segment, before) weight tile 0 of the next chunk");
  constexpr bool MERGE = NTAPS == 4;
  constexpr bool BWD = DGRAD || MERGE;      // which epilogue the instantiation carries
  static_assert(BN == 128 || BN == 256, "channel tile");
  static_assert(NBW == 3 || NBW == 4, "weight ring slots");
  constexpr int TM = GEO == GEO_STACK ? 8 : 7;   // 16-row tiles per wave (two wave rows)
  // KS ("K-split", 128-channel tile): a wave group owns the WHOLE 224 x 128 tile for one of the two 32-wide k-steps of every
  // K-tile -- wave tile 112 pixels x 64 channels, 11 fragment reads per 28 MFMAs instead of 18 (the 112 x 32 wave tile is
  // bound by its load segments: DESIGN.md 5) -- and the two groups add their partial sums through LDS before the epilogue,
  // each finalising one 32-channel half of the wave tile.  One item per workgroup (the exchange needs the LDS the
  // persistent walk keeps busy with the next item's tiles).
  static_assert(!KS || BN == 128, "K-split: 128-channel tile");
  constexpr int CW = KS ? 64 : BN / 4;      // channels per wave
  constexpr int TN = CW / 16;               // 16-channel tiles per wave
  constexpr int NP = KS ? 1 : TN / 2;       // 32-channel groups a wave FINALISES
  constexpr int NKK = KS ? 1 : 2;           // k-steps of a K-tile a wave multiplies
  constexpr int RW = BN / 64;               // LDS-DMA instructions per wave and weight tile (64 rows per pass)
  constexpr int D = NBW - 1;                // weight tiles in flight
  constexpr bool WFIRST = NTAPS == 9;       // L segments issue the weight tile before the patch pass (see the K loop)
#ifdef QT_PT_ABLATE   // experiment build only (scripts/pt_ablate.sh): K-loop parts switched off by bits of -stagger
  const int abl = q.stagger < 0 ? -q.stagger : 0;
#define QT_ABL(bit) (abl & (bit))
#else
#define QT_ABL(bit) false
#endif
  constexpr int WSLOT = BN * kKB;           // bytes of a weight ring slot
  constexpr bool AFF_LDS = BN == 128;       // room behind the rings for the per-channel vectors of the epilogue
  static_assert(TN % 2 == 0 && TN >= 2, "a wave owns whole 32-channel groups");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ConvArgs& p = q.c;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;               // waves w and w+4 share a SIMD: one of each group per SIMD
  const int wm = KS ? (wave >> 1) & 1 : grp, wn = KS ? wave & 1 : wave & 3;   // pixel half, channel quarter (K-split: half)
  const int frow = lane & 15, fk = lane >> 4;

  // PERSISTENT (round 3): a workgroup walks `ipw` consecutive items = (channel tile nt, pixel tile mt), nt-major; the
  // K-tile stream never stops at an item boundary (the next item's first patch and weight tiles are requested during
  // the last chunk of the current one) and the epilogue runs from the accumulators between two K-tiles.  Consecutive
  // logical workgroup ids share an XCD (its L2): a weight slice and neighbouring pixel tiles per XCD.
#ifdef QT_KERNEL_PROF
  if (q.prof && threadIdx.x == 0) q.prof[(long long)blockIdx.x * 32 + 31] = wall_clock64();   // kernel entry
#endif
  const int wg = qt_xcd_remap(blockIdx.x, gridDim.x);
  // (the 256-channel tile has no register room for the walk's state next to its epilogue: one item per workgroup there --
  // which is what 256 images give its 14x14 stage anyway; the loop below then runs once and its state dies in the epilogue)
  constexpr bool PERSIST = BN == 128 && !KS;
  const int item_begin = __builtin_amdgcn_readfirstlane(PERSIST ? wg * q.ipw : wg);
  const int item_end = PERSIST ? min(item_begin + q.ipw, q.items) : item_begin + 1;
  if (item_begin >= q.items) return;   // (uniform, before any barrier)
  if constexpr (PERSIST) {
    // de-phase the workgroups: all items cost the same, so without this every CU reaches its epilogue at the same moment
    // and the launch alternates between a phase that only issues MFMAs and a phase that only moves epilogue operands
    const int cls = (blockIdx.x >> 3) & 3;
    for (int i = 0; i < cls * q.stagger; ++i) __builtin_amdgcn_s_sleep(16);   // (negative: no delay)
  }

  constexpr int patch_bytes = NPASS * 64 * kKB;
  const unsigned smem_base = lds_addr_of(smem);
  const unsigned wring = smem_base + 2 * patch_bytes;
  // per-channel vectors of the epilogue ([6][N] f32: scale, shift, mean / invstd of two BatchNorm links) in LDS: an
  // epilogue in the middle of the K-tile stream must not wait for global loads behind the DMA queue
  float* aff = reinterpret_cast<float*>(smem + 2 * patch_bytes + NBW * WSLOT);
  const int NC = MERGE ? p.dst_merge : p.N;     // channels of the per-channel vectors (merged: of the destination map)
  if constexpr (AFF_LDS) {
    for (int i = tid; i < 6 * NC; i += kNT) {
      const int v = i / NC, c = i - v * NC;
      const float* src = v == 0 ? p.scale : v == 1 ? p.shift : v == 2 ? p.bn_mean[0] : v == 3 ? p.bn_invstd[0]
                                                                    : v == 4 ? p.bn_mean[1] : p.bn_invstd[1];
      aff[i] = src ? src[c] : (v == 0 ? 1.f : 0.f);
    }
  }

  // ---- per-thread staging rows -------------------------------------------------
  // a DMA wave instruction fills 8 rows x 128 B, lane-linear; LDS slot (lane & 7) of row r receives source chunk
  // (lane & 7) ^ (r & 7), the fragment reads apply the same XOR (conflict-free ds_read_b128)
  const int rbase = tid >> 3;                         // row inside a 64-row pass
  const int chunk = (tid & 7) ^ (rbase & 7);          // source 16-byte chunk of this lane (64 | pass stride)
  const int ce = chunk * (16 / (int)sizeof(T));       // ... in elements
  // patch position i*64 + rbase of pass i: sub-image, patch row / column -> source pixel relative to the TILE's origin (the
  // origin travels in the scalar offset, so these are invariant over the items of a workgroup) or kOob = out of the
  // buffer's range: the DMA writes zeros there.  GEO_ROWS: the resource's base sits one image row above the tensor, so the
  // halo row above the tile (real pixels unless the tile starts its image) has a non-negative offset; rows past the end of
  // the image below the last tile are switched off per item (top / bottom lanes).  32-bit BYTE offsets.
  unsigned pp_off[NPASS];
  unsigned edge = 0;      // bit i: this lane's row of pass i is the halo row above the tile; bit 8 + i: the one below it
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    pp_off[i] = kOob;
    const int pos = i * 64 + rbase;
    if constexpr (GEO == GEO_ROWS) {
      const unsigned pr = fdiv((unsigned)pos, q.div_pw);
      const unsigned pc = (unsigned)pos - pr * (unsigned)q.PW;
      if (pos < q.npos && pc >= 1 && (int)pc <= q.W) {
        pp_off[i] = (unsigned)((long long)pr * p.src_row_stride + (long long)((int)pc - 1) * p.src_pix_stride + ce) * (unsigned)sizeof(T);
        if (pr == 0) edge |= 1u << i;
        if ((int)pr == q.R + 1) edge |= 256u << i;
      }
    } else {
      const int pr = (pos >> 3) & 7, pc = pos & 7;          // pr = 0 / pc = 0: shared pad row / column
      int ir = pr - 1, ic = pc - 1;
      const int img = pos >> 6;
      if (pos < 256 && pr >= 1 && pc >= 1) {                 // (positions 256..: the last image's bottom pad row)
        long long simg = img;
        if (q.quad == 1) {   // quadrant img of the tile's map
          simg = 0;
          ir += ((img >> 1) & 1) * 7;
          ic += (img & 1) * 7;
        }
        pp_off[i] = (unsigned)((simg * p.src_img_stride + (long long)ir * p.src_row_stride +
                                (long long)ic * p.src_pix_stride) + ce) * (unsigned)sizeof(T);
      }
    }
  }
  // scalar part: the origin of pixel tile mt (bytes), and which halo rows of it are padding
  auto tile_origin = [&](int mt, unsigned& soff, int& top, int& bot) {
    if constexpr (GEO == GEO_ROWS) {
      const int img0 = mt / q.tpi, part = mt - img0 * q.tpi;
      soff = (unsigned)(((long long)img0 * p.src_img_stride + (long long)(part * q.R) * p.src_row_stride) * (int)sizeof(T));
      top = part == 0;
      bot = part == q.tpi - 1;
    } else {   // images 4*mt .. 4*mt + 3 (images past the batch are past the buffer's range), or map mt (quadrant mode)
      soff = (unsigned)((long long)mt * (q.quad == 1 ? 1 : 4) * p.src_img_stride * (int)sizeof(T));
      top = bot = 0;
    }
  };
  // LDS weight row rho = 32*g + 16*i + x holds output channel 32*g + 8*(x>>2) + 4*i + (x&3): a lane's two 16x16
  // tiles (i = 0, 1) of a 32-channel group then own eight consecutive channels 8*fk .. 8*fk+7 of a pixel
  // (lane part: the row inside the channel tile; the tile's first filter travels in the scalar offset)
  unsigned w_off[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int rho = rbase + 64 * i;
    const int x = rho & 15, ii = (rho >> 4) & 1;
    const int n = (rho & ~31) + 8 * (x >> 2) + 4 * ii + (x & 3);
    w_off[i] = (unsigned)(n * (NTAPS * p.KC) + ce) * (unsigned)sizeof(T);    // N % BN == 0 (checked by the launcher)
  }
  const i32x4 rs_src = make_rsrc(static_cast<const unsigned char*>(p.src) -
                                     (GEO == GEO_ROWS ? (long long)p.src_row_stride * (int)sizeof(T) : 0ll), q.src_bytes),
              rs_wgt = make_rsrc(p.wgt, q.wgt_bytes);
  const unsigned wtile_bytes = (unsigned)((long long)BN * NTAPS * p.KC * (int)sizeof(T));   // filters of one channel tile
  const unsigned tap_bytes = (unsigned)(p.KC * (int)sizeof(T));

  // rows [pass*64, pass*64+64) of a chunk into patch buffer `buf` (`off` = this thread's pp_off[pass], `coff` = the
  // chunk's byte offset inside a pixel)
  auto dma_patch_pass = [&](const i32x4& rs, int pass, unsigned off, unsigned soff, int buf) {
    blds16(rs, off, soff, smem_base + buf * patch_bytes + pass * (64 * kKB) + wave * 1024);
  };
  // this lane's offset in pass i for a tile whose halo rows above / below are padding (top / bot)
  auto pass_off = [&](int i, int top, int bot) {
    const unsigned m = (top ? 1u : 0u) | (bot ? 256u : 0u);
    return (GEO == GEO_ROWS && ((edge >> i) & m & 0x101u)) ? kOob : pp_off[i];
  };
  auto dma_weights = [&](const i32x4& rs, unsigned soff, int slot) {
    const unsigned sw = wring + slot * WSLOT + wave * 1024;
    if constexpr (RW == 2) blds16x2(rs, w_off[0], w_off[1], soff, sw);
    else blds16x4(rs, w_off[0], w_off[1], w_off[2], w_off[3], soff, sw);
  };

  // ---- per-lane fragment addressing ------------------------------------------------
  // pixel rows of this lane: m = wm*112 + j*16 + frow; its patch row for tap (0,0) is p0 = s*PP + r*PW + c
  // row m of the tile reads patch row m + shift(tap); tiles of 16 rows are 2 KB apart (row & 7 unchanged)
  const int a_lane = (wm * (TM * 16) + frow) * kKB;
  int b_off[2];       // lane part of a weight-fragment address: row frow, chunk (kk*4 + fk) ^ (frow & 7)
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) b_off[kk] = (wn * CW + frow) * kKB + (((kk * 4 + fk) ^ (frow & 7)) << 4);
  if constexpr (KS) b_off[0] = grp ? b_off[1] : b_off[0];   // (the group's k-step)

  f32x4 acc[TN][TM];

  // item = (nt, mt), nt-major: the items of a workgroup share their channel tile wherever ipw divides tiles_m
  int cur_nt = item_begin / q.tiles_m, cur_mt = item_begin - cur_nt * q.tiles_m;
  int pb = 0;                               // patch buffer of the current chunk
  int rd = 0, wr = D % NBW;                 // weight ring slots
  // ---- prologue: patch of chunk 0, D weight tiles; everything older than weight tile 1 landed ----
  unsigned cur_soff;
  int cur_top, cur_bot;
  tile_origin(cur_mt, cur_soff, cur_top, cur_bot);
#pragma unroll
  for (int i = 0; i < NPASS; ++i) dma_patch_pass(rs_src, i, pass_off(i, cur_top, cur_bot), cur_soff, 0);
#pragma unroll
  for (int s = 0; s < D; ++s) dma_weights(rs_wgt, (unsigned)cur_nt * wtile_bytes + (unsigned)s * tap_bytes, s);   // (D < 9: all in chunk 0)
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(RW * (D - 1)) : "memory");   // (lgkmcnt: the vectors above)
  if (grp == 1) asm volatile("s_barrier" ::: "memory");
#ifdef QT_KERNEL_PROF
  int prof_i = 0;
  auto stamp = [&]() {
    if (q.prof && tid == 0 && prof_i < 32) q.prof[(long long)blockIdx.x * 32 + prof_i] = wall_clock64();
    ++prof_i;
  };
#else
  auto stamp = []() {};
#endif
  stamp();   // 0: prologue done

  T* __restrict__ dst = static_cast<T*>(p.dst);
  const T* __restrict__ res = static_cast<const T*>(p.residual);
  const T* __restrict__ msk = static_cast<const T*>(p.relu_mask);
  const unsigned char* __restrict__ mbits = p.relu_mask_bits;
  const bool bwd_stats = BWD && p.bn_y[0] != nullptr;
  const bool want_stats = BWD ? bwd_stats : p.stats_partial != nullptr;
  // merged classes: dense pixel index of the gradient map -> pixel (2r + class/2, 2c + class%2) of the destination map
  auto merged_row = [&](int dr, int cls) -> long long {
    const unsigned img = fdiv((unsigned)dr, q.div_hw);
    const unsigned rem = (unsigned)dr - img * (unsigned)(q.H * q.W);
    const unsigned r = fdiv(rem, q.div_w), c = rem - r * (unsigned)q.W;
    return ((long long)img * p.dst_h + 2 * r + (cls >> 1)) * p.dst_w + 2 * c + (cls & 1);
  };

  for (int it = item_begin; it < item_end; ++it) {
    const bool nlive = PERSIST && it + 1 < item_end;
    const int nit = nlive ? it + 1 : item_begin;
    const int nxt_nt = nit / q.tiles_m, nxt_mt = nit - nxt_nt * q.tiles_m;
    unsigned nxt_soff;
    int nxt_top, nxt_bot;
    tile_origin(nxt_mt, nxt_soff, nxt_top, nxt_bot);
    // (zeroed here, not where the epilogue consumes them: the accumulators are dead across the epilogue's operand loads)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int cidx = 0; cidx < q.nchunks; ++cidx) {
      const bool lastc = cidx + 1 == q.nchunks;
      // what follows this chunk in the K-tile stream: the next chunk of the item, or chunk 0 of the next item (a slot
      // with nothing to fetch writes zeros into a dead buffer: out-of-range offsets / a zero-record resource, so that
      // the counted waits below are compile-time constants)
      unsigned n_psoff = lastc ? (nlive ? nxt_soff : 0u) : cur_soff + (unsigned)(cidx + 1) * kKB;
      const int n_top = lastc ? nxt_top : cur_top, n_bot = lastc ? nxt_bot : cur_bot;
      i32x4 rs_src_n = rs_src;
      rs_src_n.z = __builtin_amdgcn_readfirstlane((lastc && !nlive) ? 0 : rs_src.z);
      unsigned n_woff = lastc ? (nlive ? (unsigned)nxt_nt * wtile_bytes : 0u) : (unsigned)cur_nt * wtile_bytes + (unsigned)(cidx + 1) * kKB;
      unsigned c_woff = (unsigned)cur_nt * wtile_bytes + (unsigned)cidx * kKB;
      asm volatile("" : "+s"(n_psoff), "+s"(n_woff), "+s"(c_woff));   // (opaque: per-chunk address arithmetic stays out of the loop invariants)
      i32x4 rs_wgt_n = rs_wgt;
      rs_wgt_n.z = __builtin_amdgcn_readfirstlane((lastc && !nlive) ? 0 : rs_wgt.z);
      // the stores of an epilogue sit in the vmcnt queue between the DMA of the item before and after it; everything the
      // first D-1 K-tiles of an item need was waited for in front of that epilogue (below)
      const bool after_epilogue = cidx == 0 && it != item_begin;
      const unsigned char* patch = smem + pb * patch_bytes;
      static_for<NTAPS>([&](auto tap_tag) {
        constexpr int tap = decltype(tap_tag)::value;
        // ---- L_t ----
        // Nine taps: the weight tile FIRST, then the patch pass.  vmcnt retires in issue order, so the wait for a weight tile
        // also waits for everything issued before it: a patch pass comes from HBM / the memory-side cache (2.5-3 us measured
        // in the prologue) while a weight tile comes from the XCD's L2; behind the weight tile of its segment the pass has
        // D K-tiles instead of D - 1 before a wait covers it (pt_phases.py: the K loop stalled ~0.9 us per chunk on it).
        auto issue_patch = [&]() {
          constexpr int np = passes_in<NTAPS, NPASS, D>(tap), p0 = passes_before<NTAPS, NPASS, D>(tap);
          static_for<np>([&](auto pass_tag) {
            constexpr int ps = p0 + decltype(pass_tag)::value;
            dma_patch_pass(rs_src_n, ps, pass_off(ps, n_top, n_bot), n_psoff, pb ^ 1);
          });
        };
        if constexpr (!WFIRST) issue_patch();
        if (!QT_ABL(1)) {
          constexpr int u = (tap + D) % NTAPS;
          constexpr bool wrap = tap + D >= NTAPS;
          dma_weights(wrap ? rs_wgt_n : rs_wgt, (wrap ? n_woff : c_woff) + (unsigned)u * tap_bytes, wr);
        }
        if constexpr (WFIRST) {
          if (!QT_ABL(2)) issue_patch();
        }
        wr = wr + 1 == NBW ? 0 : wr + 1;
        // fragments of K-tile t: patch rows shifted by the tap, this K-tile's weight slot
        const int kh = MERGE ? (tap >> 1) + 1 : tap / 3, kw = MERGE ? (tap & 1) + 1 : tap % 3;
        const int pitch = GEO == GEO_STACK ? 8 : q.PW;
        int sh = (DGRAD ? ((2 - kh) * pitch + (2 - kw)) : (kh * pitch + kw)) * kKB;   // bytes
        // (opaque to the optimiser: otherwise the 7 x 9 fragment addresses, invariant across chunks, are hoisted out
        // of the chunk loop and 63 live registers spill the accumulators)
        asm volatile("" : "+s"(sh));
        const unsigned char* pa = patch + sh;
        const unsigned char* pw = smem + 2 * patch_bytes + rd * WSLOT;
        rd = rd + 1 == NBW ? 0 : rd + 1;
        uint4 fw[NKK][TN], fa[NKK][TM];
#ifdef QT_PT_ABLATE   // (fragments that are not read: whatever the registers hold, no instruction)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
          for (int i = 0; i < TN; ++i) asm volatile("" : "=v"(fw[kk][i].x), "=v"(fw[kk][i].y), "=v"(fw[kk][i].z), "=v"(fw[kk][i].w));
#pragma unroll
          for (int j = 0; j < TM; ++j) asm volatile("" : "=v"(fa[kk][j].x), "=v"(fa[kk][j].y), "=v"(fa[kk][j].z), "=v"(fa[kk][j].w));
        }
#endif
        if (!QT_ABL(4)) {
#pragma unroll
          for (int i = 0; i < TN; ++i) {
            fw[0][i] = *reinterpret_cast<const uint4*>(pw + i * (16 * kKB) + b_off[0]);
            if constexpr (!KS) fw[1][i] = *reinterpret_cast<const uint4*>(pw + i * (16 * kKB) + b_off[1]);
          }
        }
        if (!QT_ABL(8)) {
          const int rowb = a_lane + sh;
          const int a0 = a_lane + (((fk ^ (rowb >> 7)) & 7) << 4);   // chunk fk ^ (row & 7); chunk 4+fk is that ^ 64 bytes
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            if constexpr (KS) {
              fa[0][j] = *reinterpret_cast<const uint4*>(pa + (a0 ^ (grp << 6)) + j * (16 * kKB));   // (the group's k-step)
            } else {
              fa[0][j] = *reinterpret_cast<const uint4*>(pa + a0 + j * (16 * kKB));
              fa[1][j] = *reinterpret_cast<const uint4*>(pa + (a0 ^ 64) + j * (16 * kKB));
            }
          }
        }
        // Weight tile t+1 has landed once at most the instructions issued AFTER it are outstanding: those of
        // L_{t+2-D} .. L_t (vmcnt retires in issue order; at a chunk boundary this also covers the next patch, whose
        // last pass was issued on tap NPASS-1 < 9-D, i.e. before weight tile t+1 of tap 8).
        constexpr int in_t = passes_in<NTAPS, NPASS, D>(tap) + RW;
        constexpr int in_tm1 = passes_in<NTAPS, NPASS, D>((tap + NTAPS - 1) % NTAPS) + RW;
        // (weights first: the patch passes of segment t + 1 - D sit behind the weight tile waited for and may stay in flight)
        constexpr int behind = WFIRST ? passes_in<NTAPS, NPASS, D>((tap + 1 - D + 2 * NTAPS) % NTAPS) : 0;
        constexpr int allowed = in_t + (D == 3 ? in_tm1 : 0) + behind;
        if ((tap < D - 1 && after_epilogue) || QT_ABL(16)) {   // (uniform)
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(allowed) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- C_t ----
        __builtin_amdgcn_s_setprio(1);
        if (!QT_ABL(32)) {
#pragma unroll
          for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
              for (int j = 0; j < TM; ++j) QtMma<T>::run(acc[i][j], fw[kk][i], fa[kk][j]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
      });
      pb ^= 1;
    }

    // ---- epilogue: straight from the accumulators, no LDS scratch (the next item's DMA is in flight) -----------
    // Everything the next item's first D-1 K-tiles need (its patch, weight tiles 0 .. D-1: all issued at least a
    // segment ago) is waited for here, in front of the stores.
    // Item boundary of the ping-pong.  Group 1 runs one barrier behind group 0: without the next two lines its last barrier of
    // the item pairs with group 0's FIRST barrier of the next item (or the one behind the loop), i.e. group 1 sits out group 0's
    // whole epilogue and group 0 then waits for group 1's -- the two epilogues ran one after the other (pt_phases.py: a
    // data gradient's epilogue cost 5.7 us for group 0 and the following K loop was 4.8 us longer).  Group 0 takes one extra
    // barrier BEFORE its epilogue (pairs with group 1's last), so both epilogues run side by side; group 1 takes one extra
    // barrier AFTER its epilogue when another item follows, which restores the one-barrier offset exactly as at the start.
    if (grp == 0) asm volatile("s_barrier" ::: "memory");
    stamp();   // 1 + 3k: K loop of item k done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();   // 2 + 3k: the next item's first tiles have landed
    if constexpr (KS) {
      // K-split: the groups hold partial sums of the same 112 x 64 wave tiles.  Group 0 finalises the lower 32-channel
      // half (accumulator tiles 0, 1), group 1 the upper (tiles 2, 3): each writes the half it does NOT finalise
      // ((2 TM) x 1 KB per wave, lane-linear 16-byte stores: conflict free) and adds the partner's copy of its own half.
      // The whole LDS is free: every wave has finished its K loop (barrier) and its DMA has landed (vmcnt(0) above).
      asm volatile("s_barrier" ::: "memory");
      unsigned char* xch = smem + (unsigned)((grp * 4 + (wave & 3)) * (2 * TM)) * 1024u + lane * 16;
      if (grp) {   // (uniform branches: no per-lane selects, no second copy of the tiles)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) *reinterpret_cast<f32x4*>(xch + (i * TM + j) * 1024) = acc[i][j];
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) *reinterpret_cast<f32x4*>(xch + (i * TM + j) * 1024) = acc[2 + i][j];
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const unsigned char* xin = smem + (unsigned)(((grp ^ 1) * 4 + (wave & 3)) * (2 * TM)) * 1024u + lane * 16;
      if (grp) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = acc[2 + i][j] + *reinterpret_cast<const f32x4*>(xin + (i * TM + j) * 1024);
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(xin + (i * TM + j) * 1024);
      }
    }
    const int mt = cur_mt, n0 = cur_nt * BN;
    const int img0 = GEO == GEO_ROWS ? mt / q.tpi : mt * 4;
    const int row0 = GEO == GEO_ROWS ? (mt - img0 * q.tpi) * q.R : 0;
    // a lane's two 16x16 tiles (i = 2*pi, 2*pi+1) hold EIGHT consecutive channels c0 .. c0+7 of one pixel,
    // c0 = wave's first channel + pi*32 + fk*8: one 16-byte (bf16) access per lane and operand.
    int drow[TM];       // destination row (dense pixel index) of this lane's pixels, -1: none
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      int m = wm * (TM * 16) + j * 16 + frow;
      asm volatile("" : "+v"(m));   // (opaque: the row / column split must not be hoisted out of the item loop into spilled registers)
      drow[j] = -1;
      if constexpr (GEO == GEO_ROWS) {
        const unsigned r = fdiv((unsigned)m, q.div_pw);
        const unsigned c = (unsigned)m - r * (unsigned)q.PW;
        if ((int)r < q.R && (int)c < q.W && img0 < q.batch) drow[j] = (img0 * q.H + row0 + (int)r) * q.W + (int)c;
      } else {
        const int r = (m >> 3) & 7, c = m & 7, img = img0 + (m >> 6);
        if (r < 7 && c < 7 && img < q.batch) {
          if (q.quad == 2) drow[j] = ((img >> 2) * 14 + ((img >> 1) & 1) * 7 + r) * 14 + (img & 1) * 7 + c;
          else drow[j] = (img * 7 + r) * 7 + c;
        }
      }
    }
    // A FORWARD launch's epilogue is scale / shift, residual, ReLU and the BatchNorm statistics; a DATA-GRADIENT launch's is
    // residual, ReLU mask and the BatchNorm-backward links (other combinations take the generic kernel, qt_pt_eligible):
    // each instantiation carries only its own operands in registers.
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
      __builtin_amdgcn_sched_barrier(0);
      const int cl = wn * CW + (KS ? grp : pi) * 32 + fk * 8;   // channel inside the tile (K-split: the group's half)
      // merged parity classes: channel n0 + cl of the 4 C outputs is channel c0 of class mcls (uniform per wave: 32 | C)
      const int mcls = MERGE ? (n0 + cl) / NC : 0;
      const int c0 = MERGE ? (n0 + cl) - mcls * NC : n0 + cl;
      const T* __restrict__ res_c = (MERGE && p.dst_merge_res0 && mcls != 0) ? nullptr : res;
      float va[BWD ? 4 : 2][8];   // forward: scale, shift;  backward: mean / invstd of the two links
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if constexpr (!BWD) {
          va[0][e] = AFF_LDS ? aff[c0 + e] : (p.scale ? p.scale[c0 + e] : 1.f);
          va[1][e] = AFF_LDS ? aff[NC + c0 + e] : (p.shift ? p.shift[c0 + e] : 0.f);
        } else {
          va[0][e] = AFF_LDS ? aff[2 * NC + c0 + e] : (bwd_stats ? p.bn_mean[0][c0 + e] : 0.f);
          va[1][e] = AFF_LDS ? aff[3 * NC + c0 + e] : (bwd_stats ? p.bn_invstd[0][c0 + e] : 0.f);
          va[2][e] = AFF_LDS ? aff[4 * NC + c0 + e] : (p.bn_y[1] ? p.bn_mean[1][c0 + e] : 0.f);
          va[3][e] = AFF_LDS ? aff[5 * NC + c0 + e] : (p.bn_y[1] ? p.bn_invstd[1][c0 + e] : 0.f);
        }
      }
      float s1[8], s2[8], s3[BWD ? 8 : 1];  // forward: sum v, sum v^2;  backward: sum g, sum g*xhat0, sum g*xhat1
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
      for (int e = 0; e < (BWD ? 8 : 1); ++e) s3[e] = 0.f;
      // pixel tiles in batches: the batch's memory operands are requested before its first row is finished
      constexpr int JB = sizeof(T) == 2 ? (BN == 256 ? 2 : (BWD ? 4 : 8)) : 1;   // (BN == 256: 112 accumulator registers are live; the
      // 128-channel forward instantiation has the registers for ALL residual loads of a tile in one batch: one exposed latency)
#pragma unroll
      for (int jb = 0; jb < TM; jb += JB) {
        __builtin_amdgcn_sched_barrier(0);   // (keeps the loads of later batches from being hoisted over this one)
        Raw8<T> rres[JB], rmsk[BWD ? JB : 1], ry0[BWD ? JB : 1], ry1[BWD ? JB : 1];
        bool ok[JB];
        unsigned rbits[BWD ? JB : 1];   // (the mask as one byte: a bit per channel of the lane's 8)
#pragma unroll
        for (int u = 0; u < JB; ++u) {
          ok[u] = jb + u < TM && drow[jb + u < TM ? jb + u : 0] >= 0;
          if (ok[u]) {
            const long long off = (MERGE ? merged_row(drow[jb + u], mcls) : (long long)drow[jb + u]) * NC + c0;
            if (res_c) rres[u].load(res_c + off);
            if constexpr (BWD) {
              if (msk) rmsk[u].load(msk + off);
              if (mbits) rbits[u] = mbits[off >> 3];
              if (bwd_stats) ry0[u].load(static_cast<const T*>(p.bn_y[0]) + off);
              if (p.bn_y[1]) ry1[u].load(static_cast<const T*>(p.bn_y[1]) + off);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < JB; ++u) {
          if (jb + u >= TM) continue;
          const int j = jb + u;
          float v[8] = {acc[2 * pi][j][0],     acc[2 * pi][j][1],     acc[2 * pi][j][2],     acc[2 * pi][j][3],
                        acc[2 * pi + 1][j][0], acc[2 * pi + 1][j][1], acc[2 * pi + 1][j][2], acc[2 * pi + 1][j][3]};
          if (!ok[u]) continue;
          const long long off = (MERGE ? merged_row(drow[j], mcls) : (long long)drow[j]) * NC + c0;
          if constexpr (!BWD) {
            if (p.stats_partial != nullptr) {  // (uniform: an eval forward keeps no sums)
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                s1[e] += v[e];
                s2[e] += v[e] * v[e];
              }
            }
            if (p.scale || p.shift) {  // (uniform)
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = v[e] * va[0][e] + va[1][e];
            }
          }
          if (res_c) {
            float rv[8];
            rres[u].get(rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[e];
          }
          if constexpr (!BWD) {
            if (p.relu) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
          } else {
            if (msk) {
              float mv[8];
              rmsk[u].get(mv);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
            }
            if (mbits) qt_apply_mask_bits(rbits[u], v);
          }
          QtVec8<T>::store(dst + off, v);
          if constexpr (BWD) {
            if (bwd_stats) {
              float yv[8];
              ry0[u].get(yv);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                s1[e] += v[e];
                s2[e] += v[e] * (yv[e] - va[0][e]) * va[1][e];
              }
              if (p.bn_y[1]) {
                ry1[u].get(yv);
#pragma unroll
                for (int e = 0; e < 8; ++e) s3[e] += v[e] * (yv[e] - va[2][e]) * va[3][e];
              }
            }
          }
        }
      }
      if (want_stats) {
        // sum over the 16 pixels (lanes with equal fk) of the wave, four DPP adds in a fixed order -> deterministic; one partial
        // row per (pixel tile, wave row): the two wave rows are added by the finalize kernels, not here
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] = qt_row16_sum(s1[e]);
          s2[e] = qt_row16_sum(s2[e]);
          if constexpr (BWD) s3[e] = qt_row16_sum(s3[e]);
        }
        if (frow == 0) {
          float* o0 = BWD ? p.bn_partial[0] : p.stats_partial;
          const long long row = MERGE ? ((long long)mt * 2 + wm) * 4 + mcls : (long long)mt * 2 + wm;   // (merged: one row per class too)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            o0[(row * 2 + 0) * NC + c0 + e] = s1[e];
            o0[(row * 2 + 1) * NC + c0 + e] = s2[e];
          }
          if constexpr (BWD) {
            if (p.bn_y[1]) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                p.bn_partial[1][(row * 2 + 0) * NC + c0 + e] = s1[e];
                p.bn_partial[1][(row * 2 + 1) * NC + c0 + e] = s3[e];
              }
            }
          }
        }
      }
    }
    cur_nt = nxt_nt;
    cur_mt = nxt_mt;
    cur_soff = nxt_soff; cur_top = nxt_top; cur_bot = nxt_bot;
    if (grp == 1 && it + 1 < item_end) asm volatile("s_barrier" ::: "memory");   // (see the item boundary above)
    stamp();   // 3 + 3k: epilogue of item k done
  }
  // the branch-free slots of the last chunk wrote zeros into dead buffers: landed before the LDS changes hands
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// geometry of the 196-pixel tiles for an H x W image; false: shape not covered
bool pt_geometry(int H, int W, PtArgs& q) {
  if (H == 28 && W == 28) { q.G = 1; q.R = 7; }
  else if (H == 14 && W == 14) { q.G = 1; q.R = 14; }
  else if (H == 7 && W == 7) { q.G = 4; q.R = 7; }
  else return false;
  q.H = H; q.W = W;
  if (q.G == 1) {
    q.PW = W + 2;
    q.PP = (q.R + 2) * q.PW;
    q.npos = q.PP;
    q.tpi = H / q.R;
  } else {                       // stacked 7x7 images: pitch 8 / 64, one more pad row and the wrap-around pad at the end
    q.PW = 8;
    q.PP = 64;
    q.npos = 4 * 64 + 9;
    q.tpi = 0;
  }
  q.npass = (q.npos + 63) / 64;
  q.div_pw = make_fastdiv((unsigned)q.PW);
  return q.npass == 4 || q.npass == 5;
}
inline int pt_tiles_m(const PtArgs& q, int batch) { return q.G == 1 ? batch * q.tpi : (batch + q.G - 1) / q.G; }
// 256-channel tiles where the channel count allows it and 256 pixel tiles still cover the chip (14x14: one per image)
inline int pt_bn(const ConvArgs& a) { return (a.N % 256 == 0 && a.OH == 14) ? 256 : 128; }

// persistent grid: one workgroup per CU (qt_set_pt_conv_max_workgroups caps it: tests walk several items per workgroup
// at small batches; QTCNN_PT_PERSIST=0: one item per workgroup as in round 2, same-box A/B)
int g_pt_max_wgs_fwd = 0;
#ifdef QT_KERNEL_PROF
unsigned long long* g_pt_prof = nullptr;
#endif
int g_pt_stagger[2] = {-1, -1};   // forward, backward
inline int pt_stagger(bool bwd) {
  if (g_pt_stagger[0] < 0) {
    const char* e = getenv("QTCNN_PT_STAGGER_FWD");
    g_pt_stagger[0] = e ? atoi(e) : 0;
    e = getenv("QTCNN_PT_STAGGER_BWD");
    g_pt_stagger[1] = e ? atoi(e) : 0;
  }
  return g_pt_stagger[bwd ? 1 : 0];
}
inline int pt_workgroups() {
  static int persist = -1;
  if (persist < 0) {
    const char* e = getenv("QTCNN_PT_PERSIST");
    persist = e ? atoi(e) : 1;
  }
  if (g_pt_max_wgs_fwd > 0) return g_pt_max_wgs_fwd;
  if (!persist) return 1 << 30;
  int dev = 0, cus = 256, v = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
    cus = v;
  return cus;
}

template <typename T, int BN, int NBW, int NPASS, int GEO, bool DGRAD, int NTAPS = 9, bool KS = false>
int launch(PtArgs q, hipStream_t stream) {
  constexpr int lds = 2 * NPASS * 64 * kKB + NBW * BN * kKB + (BN == 128 ? 6 * 512 * 4 : 0);   // (+ the epilogue's vectors)
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = conv_pt_kernel<T, BN, NBW, NPASS, GEO, DGRAD, NTAPS, KS>;
  static std::atomic<unsigned long long> lds_limit_set{0};  // per device
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(kern), lds, lds_limit_set)) return rc;
  q.c.gridN = q.c.N / BN;
  q.c.gridM = q.tiles_m;
  q.items = q.tiles_m * q.c.gridN;
  q.ipw = (BN == 128 && !KS) ? qt_cdiv(q.items, pt_workgroups()) : 1;
  q.stagger = (q.ipw >= 2 || pt_stagger(DGRAD || NTAPS == 4) < 0) ? pt_stagger(DGRAD || NTAPS == 4) : 0;   // (< 0: experiment builds)
#ifdef QT_KERNEL_PROF
  q.prof = g_pt_prof;
#endif
  hipLaunchKernelGGL(kern, dim3(qt_cdiv(q.items, q.ipw)), dim3(kNT), lds, stream, q);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

// QTCNN_PT_KSPLIT (default 1): the 128-channel tile of the bf16 build in its K-split form (see the kernel); 0: the persistent
// M-split form
inline bool ksplit_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("QTCNN_PT_KSPLIT");
    v = e ? atoi(e) : 1;
  }
  return v != 0;
}

// merged stride-2 data gradient (four tap slots, 128-channel tiles of the 4 C class channels)
template <typename T>
int dispatch_merged(const PtArgs& q, hipStream_t stream) {
  if constexpr (sizeof(T) == 2) {
    if (q.G == 4 && ksplit_enabled() && g_pt_max_wgs_fwd == 0) return launch<T, 128, 3, 5, GEO_STACK, false, 4, true>(q, stream);
  }
  if (q.G == 4) return launch<T, 128, 3, 5, GEO_STACK, false, 4>(q, stream);
  // (the GEO_ROWS forms - 28x28 and 14x14 gradient maps - were measured and are not instantiated: 8 / 16 K-tiles per item do not
  // cover an epilogue that scatters 64-byte runs, 276 / 229 us in the train step against 156 / 118 us of the generic tile whose
  // three workgroups per CU overlap their epilogues; on the 7x7 map, 32 K-tiles per item, this kernel wins 121 against 136 us)
  qt_set_error("conv_pt (merged): only the 7x7 gradient map is instantiated");
  return QT_ERR_UNSUPPORTED;
}

inline bool ksplit_enabled_rows() {   // QTCNN_PT_KSPLIT=2: also the 28x28 stage (measurement)
  const char* e = getenv("QTCNN_PT_KSPLIT");
  return e && atoi(e) >= 2;
}

template <typename T, bool DGRAD>
int dispatch(const PtArgs& q, hipStream_t stream) {
  if constexpr (sizeof(T) == 2) {
    if (ksplit_enabled() && g_pt_max_wgs_fwd == 0) {
      // (7x7 stage: one item per workgroup in either form -- 50-57 -> 45-49 us of K loop per launch.  On the 28x28 stage the
      // K loop gains as much, 14 -> 12 us per item, but four one-item workgroups per CU with their un-hidden prologues are
      // slower than the persistent walk: 74 / 84 / 88 us against 69 / 79 / 86 (plain / eval epilogue / mask + link); not used)
      if (q.G == 4) return launch<T, 128, 4, 5, GEO_STACK, DGRAD, 9, true>(q, stream);
      if (q.npass == 5 && ksplit_enabled_rows()) return launch<T, 128, 4, 5, GEO_ROWS, DGRAD, 9, true>(q, stream);
    }
  }
  if (q.G == 4) return launch<T, 128, 4, 5, GEO_STACK, DGRAD>(q, stream);            // 7x7: 265 patch positions
  if (q.npass == 5) return launch<T, 128, 4, 5, GEO_ROWS, DGRAD>(q, stream);        // 28x28 quarter: 270
  if (q.npass == 4) {                                                                // 14x14: 256
    if constexpr (sizeof(T) == 2) {   // (the f32 instantiation of the 256-channel tile spills 8 registers: 128 there)
      if (pt_bn(q.c) == 256) return launch<T, 256, 3, 4, GEO_ROWS, DGRAD>(q, stream);
    }
    return launch<T, 128, 4, 4, GEO_ROWS, DGRAD>(q, stream);
  }
  qt_set_error("conv_pt: %d patch passes not instantiated", q.npass);
  return QT_ERR_UNSUPPORTED;
}

// QTCNN_PT_MERGED (default 1): the merged stride-2 data gradients take this kernel's four-tap instantiation; 0: the generic
// tile (same-box A/B)
inline bool merged_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("QTCNN_PT_MERGED");
    v = e ? atoi(e) : 1;
  }
  return v != 0;
}

// 0: off, 1: on (default).  QTCNN_PT_CONV / qt_set_pt_conv: same-box A/B against the generic kernel.
int g_pt_enabled = -1;
inline int pt_enabled() {
  if (g_pt_enabled < 0) {
    const char* e = getenv("QTCNN_PT_CONV");
    g_pt_enabled = e ? atoi(e) : 1;
  }
  return g_pt_enabled;
}

}  // namespace

extern "C" void qt_set_pt_conv(int mode) { g_pt_enabled = mode < 0 ? 1 : mode; }
extern "C" void qt_set_pt_conv_max_workgroups(int n) { g_pt_max_wgs_fwd = n > 0 ? n : 0; }
#ifdef QT_KERNEL_PROF
extern "C" void qt_set_pt_prof(unsigned long long* buf) { g_pt_prof = buf; }   // experiment build only, not in the header
#endif

// images the kernel walks (quadrant modes: four 7x7 region images per map) and the quadrant mode, -1: not covered
static int pt_images(const ConvArgs& a, bool dgrad, int* quad) {
  *quad = 0;
  if (a.quad) {
    if (a.quad != 2 || a.IH != 7 || a.IW != 7) return -1;
    if (!dgrad && a.OH == 7 && a.OW == 7) { *quad = 1; return a.M / 49; }            // M = maps * 4 * 49
    if (dgrad && a.OH == 14 && a.OW == 14) { *quad = 2; return a.M / 196 * 4; }     // M = maps * 196
    return -1;
  }
  if (a.OH != a.IH || a.OW != a.IW || a.M % (a.OH * a.OW) != 0) return -1;
  return a.M / (a.OH * a.OW);
}

// the merged stride-2 data gradient (qt_conv_desc.dst_merge): a 2x2 / stride 1 / pad 0 gather over a 7x7
// gradient map whose 4 C outputs scatter to the (2H x 2W) C-channel map
static bool pt_merged_shape(const ConvArgs& a, int esz) {
  return a.dst_merge > 0 && a.ntaps == 4 && a.KW == 2 && a.stride == 1 && a.pad == 0 && a.dst_sub == 2 && !a.quad &&
         a.IH == a.OH && a.IW == a.OW && a.dst_h == 2 * a.OH && a.dst_w == 2 * a.OW && a.dst_merge % 32 == 0 &&
         a.N == 4 * a.dst_merge && a.N % 128 == 0 && (a.KC * esz) % kKB == 0 && a.dst_merge <= 512;
}

// the descriptor must be a 3x3 / stride 1 / pad 1 convolution on 28x28, 14x14 or 7x7 images (or 7x7 quadrants of 14x14 maps)
bool qt_pt_eligible(const ConvArgs& a, int dtype, bool dgrad) {
  if (!pt_enabled()) return false;
  const int esz = dtype == QT_F32 ? 4 : 2;
  if (a.dst_merge) {
    if (dgrad || !pt_merged_shape(a, esz) || !merged_enabled()) return false;
    if (a.scale || a.shift || a.stats_partial || a.relu) return false;
    PtArgs q;
    if (!pt_geometry(a.OH, a.OW, q)) return false;
    const int batch = a.M / (a.OH * a.OW);
    if (batch < 16 || q.G != 4 || batch % 4 != 0) return false;   // (7x7 gradient maps only, see dispatch_merged)
    if ((long long)batch * a.src_img_stride * esz >= (1ll << 31) || (long long)a.N * 4 * a.KC * esz >= (1ll << 31)) return false;
    return true;
  }
  if (a.ntaps != 9 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.dst_sub) return false;
  // epilogue operands each instantiation carries (the generic kernel takes every combination): forward = scale / shift,
  // residual, ReLU, statistics; data gradient = residual, ReLU mask, BatchNorm-backward links
  if (!dgrad && (a.bn_y[0] || a.relu_mask || a.relu_mask_bits)) return false;
  if (dgrad && (a.scale || a.shift || a.stats_partial || a.relu)) return false;
  if (a.N > 512) return false;   // (per-channel vectors of the epilogue in LDS)
  int quad;
  const int batch = pt_images(a, dgrad, &quad);
  if (batch < 0) return false;
  PtArgs q;
  if (!pt_geometry(a.IH, a.IW, q)) return false;
  if ((a.KC * esz) % kKB != 0 || a.N % 128 != 0) return false;
  // 32-bit byte offsets below kOob
  const long long simgs = quad == 1 ? batch / 4 : batch;
  if (simgs * a.src_img_stride * esz >= (1ll << 31) || (long long)a.N * 9 * a.KC * esz >= (1ll << 31)) return false;
  if (batch < 16) return false;   // a handful of tiles: the generic kernel's small tiles cover the chip better
  return true;
}

int qt_pt_stats_rows(const ConvArgs& a, bool dgrad) {
  PtArgs q;
  int quad;
  if (a.dst_merge) {   // (pixel tile, wave row, class)
    pt_geometry(a.OH, a.OW, q);
    return 2 * 4 * pt_tiles_m(q, a.M / (a.OH * a.OW));
  }
  pt_geometry(a.IH, a.IW, q);
  return 2 * pt_tiles_m(q, pt_images(a, dgrad, &quad));   // one partial row per (pixel tile, wave row)
}

int qt_pt_launch(const ConvArgs& a, int dtype, bool dgrad, hipStream_t stream) {
  PtArgs q;
  q.c = a;
  if (a.dst_merge) {
    const int esz = dtype == QT_F32 ? 4 : 2;
    pt_geometry(a.OH, a.OW, q);
    q.quad = 0;
    q.batch = a.M / (a.OH * a.OW);
    q.tiles_m = pt_tiles_m(q, q.batch);
    q.nchunks = a.KC * esz / kKB;
    q.src_bytes = (unsigned)((((long long)q.batch - 1) * a.src_img_stride + ((long long)a.OH - 1) * a.src_row_stride +
                              ((long long)a.OW - 1) * a.src_pix_stride + a.KC + (q.G == 1 ? a.src_row_stride : 0)) * esz);
    q.wgt_bytes = (unsigned)((long long)a.N * 4 * a.KC * esz);
    q.div_hw = make_fastdiv((unsigned)(a.OH * a.OW));
    q.div_w = make_fastdiv((unsigned)a.OW);
    return dtype == QT_F32 ? dispatch_merged<float>(q, stream) : dispatch_merged<bf16_t>(q, stream);
  }
  pt_geometry(a.IH, a.IW, q);
  q.batch = pt_images(a, dgrad, &q.quad);
  q.tiles_m = pt_tiles_m(q, q.batch);
  const int esz = dtype == QT_F32 ? 4 : 2;
  q.nchunks = a.KC * esz / kKB;
  const long long simgs = q.quad == 1 ? q.batch / 4 : q.batch;
  const int sh = q.quad == 1 ? 14 : a.IH, sw = q.quad == 1 ? 14 : a.IW;
  // (the range check covers vector + scalar offset; row geometry: the base sits one image row above the tensor)
  q.src_bytes = (unsigned)(((simgs - 1) * a.src_img_stride + ((long long)sh - 1) * a.src_row_stride +
                            ((long long)sw - 1) * a.src_pix_stride + a.KC + (q.G == 1 ? a.src_row_stride : 0)) * esz);
  q.wgt_bytes = (unsigned)((long long)a.N * 9 * a.KC * esz);
  if (dtype == QT_F32) return dgrad ? dispatch<float, true>(q, stream) : dispatch<float, false>(q, stream);
  return dgrad ? dispatch<bf16_t, true>(q, stream) : dispatch<bf16_t, false>(q, stream);
}
