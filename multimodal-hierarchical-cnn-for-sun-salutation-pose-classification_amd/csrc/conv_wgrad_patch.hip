// Weight gradient of a 3x3 / stride 1 / pad 1 convolution as a streaming "patch" kernel (bf16).
//
// Replaces the conv2d backward-weight ATen calls of loss.backward()
// (/root/reference/Quadtree_from scratch/Quadtree_train.py:65) for the BasicBlock convolutions of the
// backbone (/root/reference/Quadtree_from scratch/models.py:222-229 -> torchvision resnet18 layer1..3).
//
//   dW[o][kh][kw][i] = sum_g dY[g][o] * X[g + (kh-1)*PW + (kw-1)][i]
//
// g runs over ZERO-PADDED positions, so a filter tap is a constant shift of the position index and
// all nine taps read the SAME rows of X.  One pad column and one pad row per image are enough
// (PW = W+1, PH = H+1, images back to back): the right neighbour of a row's last pixel is the next
// row's pad column, the row below an image is the next image's pad row, and taps move by at most one
// in each direction -- 13 % fewer positions than (W+2)(H+2) at 14x14, 21 % at 7x7.  The generic kernel (conv_wgrad.hip) makes one workgroup per (tap, tile) and therefore
// fetches dY and X nine times (32 FLOP per byte staged for 64x64 tiles); here one workgroup owns a
// 64(o) x 64(i) x 9(taps) accumulator (144 VGPRs per lane over 4 waves) and streams its range of
// positions once: 288 FLOP per staged byte.
//
//   * X rows live in an LDS ring of NX 32-row chunks (row = position, 128 B = 64 channels), dY rows in
//     a ring of ND chunks; both are filled by LDS-DMA (global_load_lds_dwordx4, halo / out-of-image
//     positions come from the zero page), D chunks ahead of the MFMAs, one barrier per chunk.
//   * pixel-major MFMA fragments come from ds_read_b64_tr_b16; the 32-byte-block XOR swizzle of
//     conv_wgrad.hip sits on the DMA source side.  A tap only moves the row a lane reads.
//   * the range of positions is split over workgroups (one per CU); partial filters are added
//     with f32 atomics into the zeroed gradient, as in the generic kernel.
#include <stdlib.h>

#include "qt_common.h"

namespace {

struct WPArgs {
  const bf16_t* dy;
  const bf16_t* x;
  float* dw;
  float* part;            // [nsplit][N][9][KC] partial filters (plain stores) or NULL: atomics into dw
  long long x_is, dy_is;  // image strides (elements)
  int x_rs, x_ps, dy_rs, dy_ps;
  int N, KC, H, W, PW, PH, B, PP;
  int total;              // B * PP padded positions
  int pps, nsplit, tiles, tilesC;
  int reg, regR, regS;    // region mode (qt_conv_desc.quad, tile kernel only): S x S regions of R x R per map, each with its
                          // own pad row / column in front: reg = R + 1 is the pitch of a region on the padded grid (0: plain)
  int halo;               // rows of X kept on each side of a chunk: ceil32(PW + 1)
  int adv_h, adv_w;       // 32 positions = adv_h padded rows + adv_w columns
  FastDiv div_pp, div_pw;
};

// walks padded positions 32 at a time and keeps the element offset of the (unpadded) source pixel
struct PosWalk {
  int img, ph, pw;
  long long off;
  __device__ __forceinline__ void init(int g, const WPArgs& p, long long is, int rs, int ps, int coff) {
    const unsigned gg = (unsigned)(g + p.PP);  // callers keep g >= -PP
    const unsigned q = fdiv(gg, p.div_pp);
    const unsigned rem = gg - q * (unsigned)p.PP;
    img = (int)q - 1;
    ph = (int)fdiv(rem, p.div_pw);
    pw = (int)rem - ph * p.PW;
    off = (long long)img * is + (long long)(ph - 1) * rs + (long long)(pw - 1) * ps + coff;
  }
  __device__ __forceinline__ bool valid(const WPArgs& p) const {
    return (unsigned)img < (unsigned)p.B && (unsigned)(ph - 1) < (unsigned)p.H && (unsigned)(pw - 1) < (unsigned)p.W;
  }
  __device__ __forceinline__ void advance(const WPArgs& p, long long adv_off, long long wrap_w, long long wrap_h) {
    pw += p.adv_w;
    ph += p.adv_h;
    off += adv_off;
    if (pw >= p.PW) {
      pw -= p.PW;
      ++ph;
      off += wrap_w;
    }
    if (ph >= p.PH) {
      ph -= p.PH;
      ++img;
      off += wrap_h;
    }
  }
};

__device__ __forceinline__ QT_LDS_AS s16x4* lds_tr_ptr(unsigned lds_byte) {
  return (QT_LDS_AS s16x4*)(size_t)lds_byte;
}

constexpr int WP_CH = 32;  // positions per MFMA K-step

// G = wave groups per workgroup (4 waves each).  A macro-chunk is 32*G consecutive positions: all
// waves stage it together, group g contracts rows [32g, 32g+32) of it, so with G = 2 every SIMD
// holds two waves that cover each other's LDS reads and address arithmetic.  The groups' partial
// filters are summed through LDS before they leave the workgroup.
template <int G, int D, int NX, int ND>
__global__ __launch_bounds__(256 * G, 2 / G) void conv_wgrad_patch_kernel(WPArgs p) {
  static_assert((NX & (NX - 1)) == 0 && (ND & (ND - 1)) == 0, "rings are powers of two");
  static_assert(G == 1 || G == 2, "one or two wave groups");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MCR = WP_CH * G;              // rows of a macro-chunk
  constexpr int MCB = MCR * 128;              // its bytes (64 bf16 channels per row)
  constexpr unsigned XRING = NX * MCB;        // X ring at LDS byte 0 (power of two: wrap = AND)
  const unsigned smem_base = lds_addr_of(smem);
  if (smem_base != 0) return;  // no static LDS in this kernel; addresses below are absolute

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = wave >> 2, wq = wave & 3;

  // whole position ranges per XCD; the tiles of one range run back to back on it (they share
  // the dY / X slices in that XCD's L2)
  int split, tile;
  if ((p.nsplit & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    split = (slot / p.tiles) * 8 + xcd;
    tile = slot % p.tiles;
  } else {
    split = blockIdx.x % p.nsplit;
    tile = blockIdx.x / p.nsplit;
  }
  const int tc = tile % p.tilesC, tn = tile / p.tilesC;
  const int n0 = tn * 64, c0 = tc * 64;
  const int p0 = split * p.pps;
  if (p0 >= p.total) return;
  const int pend = min(p.total, p0 + p.pps);
  const int nsteps = (pend - p0 + MCR - 1) / MCR;
  const int L = (2 * p.halo) / MCR;  // X macro-chunks ahead of the dY macro-chunk

  // ---- DMA side: thread = (row of the macro-chunk, 16-byte LDS slot); swizzle on the source ----
  const int drow = tid >> 3, dslot = tid & 7;
  const int dkey = (drow >> 1) & 3;
  const int dchunk = (((dslot >> 1) ^ dkey) << 1) | (dslot & 1);
  const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(qt_zero_page);
  PosWalk wx, wy;
  wx.init(p0 - p.halo + drow, p, p.x_is, p.x_rs, p.x_ps, c0 + dchunk * 8);
  wy.init(p0 - 2 * p.halo + drow, p, p.dy_is, p.dy_rs, p.dy_ps, n0 + dchunk * 8);
  const long long x_adv = (long long)p.adv_h * p.x_rs + (long long)p.adv_w * p.x_ps;
  const long long x_ww = (long long)p.x_rs - (long long)p.PW * p.x_ps;
  const long long x_wh = p.x_is - (long long)p.PH * p.x_rs;
  const long long y_adv = (long long)p.adv_h * p.dy_rs + (long long)p.adv_w * p.dy_ps;
  const long long y_ww = (long long)p.dy_rs - (long long)p.PW * p.dy_ps;
  const long long y_wh = p.dy_is - (long long)p.PH * p.dy_rs;
  int unit = 0;  // next unit to issue: X macro-chunk `unit`, dY macro-chunk `unit - L`
  auto issue = [&]() {
    const bf16_t* gx = wx.valid(p) ? p.x + wx.off : zero_src;
    glds16(gx, (unsigned)((unit & (NX - 1)) * MCB + wave * 1024));
    const bf16_t* gy = (unit >= L && wy.valid(p)) ? p.dy + wy.off : zero_src;
    glds16(gy, XRING + (unsigned)(((unit - L) & (ND - 1)) * MCB + wave * 1024));
    wx.advance(p, x_adv, x_ww, x_wh);
    wy.advance(p, y_adv, y_ww, y_wh);
    ++unit;
  };

  // ---- MFMA side ----
  const int li = lane & 15, lg = lane >> 4;
  const int q = li >> 2, pp = li & 3;
  const int lrow = WP_CH * group + 4 * lg + q;  // second transposing read: lrow + 16 (same swizzle key)
  // dY fragment addresses inside a stage (hi half = +2048 B)
  unsigned a_rel[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_rel[i] = XRING + lrow * 128 + ((i ^ ((lrow >> 1) & 3)) << 5) + pp * 8;
  // X fragment addresses per tap (absolute LDS bytes, advanced by one macro-chunk per step)
  unsigned b_lo[9], b_hi[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int s = (t / 3 - 1) * p.PW + (t % 3 - 1);
    const int r = p.halo + s + lrow;  // >= 0: halo >= PW + 1
    const unsigned k = (unsigned)(r >> 1) & 3u;
    const unsigned col = (((unsigned)wq ^ k) << 5) + pp * 8;
    b_lo[t] = (((unsigned)r * 128u) & (XRING - 1)) + col;
    b_hi[t] = (((unsigned)(r + 16) * 128u) & (XRING - 1)) + col;
  }

  f32x4 acc[4][9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int u = 0; u < L + D; ++u) issue();

  for (int c = 0; c < nsteps; ++c) {
    issue();
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * D) : "memory");
    const unsigned sbase = (unsigned)((c & (ND - 1)) * MCB);
    uint4 fa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(lds_tr_ptr(a_rel[i] + sbase));
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(lds_tr_ptr(a_rel[i] + sbase + 2048));
      uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      fa[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(lds_tr_ptr(b_lo[t]));
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(lds_tr_ptr(b_hi[t]));
      uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      const uint4 fb = make_uint4(l2.x, l2.y, h2.x, h2.y);
      b_lo[t] = (b_lo[t] + MCB) & (XRING - 1);
      b_hi[t] = (b_hi[t] + MCB) & (XRING - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                            __builtin_bit_cast(bf16x8, fb), acc[i][t], 0, 0, 0);
    }
  }
  // the look-ahead units are still in flight: they must land before this workgroup's LDS is reused
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- two groups: group 1 hands o-blocks 0,1 to group 0 and takes o-blocks 2,3 from it ----
  if constexpr (G == 2) {
    f32x4* xch = reinterpret_cast<f32x4*>(smem) + (wq * 18) * 64 + lane;  // [wq][18][lane] x 16 B
    __syncthreads();  // every wave is done reading the rings
    if (group == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) xch[(i * 9 + t) * 64] = acc[i][t];
    }
    __syncthreads();
    if (group == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] += xch[(i * 9 + t) * 64];
    }
    __syncthreads();
    if (group == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) xch[(i * 9 + t) * 64] = acc[2 + i][t];
    }
    __syncthreads();
    if (group == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[2 + i][t] += xch[(i * 9 + t) * 64];
    }
  }

  // ---- accumulate: lane holds o = 16*i + 4*lg + r, input channel 16*wq + li of every tap ----
  const int cc = c0 + wq * 16 + li;
  auto flush = [&](int i, const f32x4 (&a)[9]) {
    if (p.part) {  // deterministic path: this range's partial filter, summed by wgrad_partial_sum_kernel
      float* base = p.part + (long long)split * p.N * 9 * p.KC;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + i * 16 + lg * 4 + r;
        float* row = base + (long long)n * 9 * p.KC + cc;
#pragma unroll
        for (int t = 0; t < 9; ++t) row[t * p.KC] = a[t][r];
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + i * 16 + lg * 4 + r;
      float* row = p.dw + (long long)n * 9 * p.KC + cc;
#pragma unroll
      for (int t = 0; t < 9; ++t) atomicAdd(row + t * p.KC, a[t][r]);
    }
  };
  if constexpr (G == 2) {
    if (group == 0) {
      flush(0, acc[0]);
      flush(1, acc[1]);
    } else {
      flush(2, acc[2]);
      flush(3, acc[3]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) flush(i, acc[i]);
  }
}

// ---------------------------------------------------------------------------------------------
// Tile-resident variant (round 2).  What bounded the ring kernel above (PMC): nothing of the machine --
// 28 % MFMA-busy with no LDS conflicts and 9 GB/s of DMA per CU -- but its instruction stream: a barrier per 32
// positions with the fragment reads behind it (their latency exposed 100+ times per workgroup), ~100 vector
// instructions of position walking and ring wrapping per 36 MFMAs, and, with one wave group and 512 registers to
// allocate from, 208 v_accvgpr copies per iteration (fixed by the launch bound above).  Here
//   * a TILE of NT*32 consecutive padded positions (+ PW+1 rows of X on either side) is resident in LDS, double
//     buffered: one barrier per tile (288 MFMAs per wave pair), every fragment address is a per-lane base + an
//     immediate (no ring to wrap), and the fragment reads run two taps ahead of the MFMAs, the next chunk's dY
//     fragments four taps ahead -- the kernel rarely waits on a read it has just issued;
//   * the source offset of a padded position comes from a table in LDS (PP + 8 words per operand, built once per
//     workgroup: byte offset inside the image or an out-of-range marker for pad positions): a DMA instruction costs
//     one table read and one v_add3 per lane, image index and remainder of its first row are scalar arithmetic;
//     pad rows and rows past the tensor are zero-filled by the buffer range check;
//   * 8 waves, two per SIMD: group g of four waves contracts chunks g, g+2, ... of the tile; the groups' partial
//     filters are exchanged through LDS as in the G = 2 ring kernel;
//   * 222 VGPRs (allocated: 224 -- two such waves leave 64 registers per SIMD, one wave of the main stream's BatchNorm
//     kernels; with the X fragments three taps ahead it is 226 -> 232 and the step is 2 % slower, 6.45 vs 6.33 ms, at the
//     same kernel time) and (by default) up to 160 KB of LDS.  Measured alone (B = 256, the 13 layers of the model): 58-70 us
//     per launch against 86-93 (ring, one group) and 75-84 (ring, two groups); 42-50 % MFMA-busy.  What is left: the
//     two waves of a SIMD share one matrix pipe and their streams do not interleave (fragment reads, DMA slots and
//     MFMAs measured additive: 9 + 7 + 28 us of a 50 us loop), ~12 us of prologue (table, first tile) and epilogue
//     (exchange, 37.7 MB of partial filters), and the launch that sums the partial filters (8-12 us).
// ---------------------------------------------------------------------------------------------
struct WTArgs {
  WPArgs p;
  int HL;                      // rows of X kept on each side of a tile (multiple of 8, >= PW + 1)
  unsigned xb, bufb, tab;      // bytes of a buffer's X region / of a buffer; LDS byte address of the offset tables
  unsigned x_bytes, dy_bytes;  // extents of the two tensors (buffer resources)
};

template <int NT>
__global__ __launch_bounds__(512) void conv_wgrad_tile_kernel(WTArgs a) {
  static_assert(NT % 2 == 0 && NT >= 2 && NT <= 8, "chunks per tile");
  constexpr int NJ = NT / 2;  // chunks per wave group and tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const WPArgs& p = a.p;
  if (lds_addr_of(smem) != 0) return;  // no static LDS in this kernel; addresses below are absolute

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = wave >> 2, wq = wave & 3;

  int split, tile;
  if ((p.nsplit & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    split = (slot / p.tiles) * 8 + xcd;
    tile = slot % p.tiles;
  } else {
    split = blockIdx.x % p.nsplit;
    tile = blockIdx.x / p.nsplit;
  }
  const int tc = tile % p.tilesC, tn = tile / p.tilesC;
  const int n0 = tn * 64, c0 = tc * 64;
  const int p0 = split * p.pps;
  if (p0 >= p.total) return;
  const int pend = min(p.total, p0 + p.pps);
  const int nch = (pend - p0 + WP_CH - 1) / WP_CH;
  const int ntiles = (nch + NT - 1) / NT;

  // ---- offset tables: entry r < PP is padded position r of an image, entries PP..PP+7 are the first positions of
  // the NEXT image (an 8-row DMA unit may straddle two images) ----
  const int tabn = p.PP + 8;
  {
    unsigned* tx = reinterpret_cast<unsigned*>(smem + a.tab);
    unsigned* ty = tx + tabn;
    for (int r = tid; r < tabn; r += 512) {
      const int rr = r >= p.PP ? r - p.PP : r;
      const int ph = (int)fdiv((unsigned)rr, p.div_pw), pw = rr - ph * p.PW;
      const unsigned nx = r >= p.PP ? (unsigned)p.x_is * 2u : 0u, ny = r >= p.PP ? (unsigned)p.dy_is * 2u : 0u;
      bool ok;
      unsigned xo, yo;
      if (p.reg == 0) {
        ok = ph >= 1 && pw >= 1;
        xo = (unsigned)((ph - 1) * p.x_rs + (pw - 1) * p.x_ps);
        yo = (unsigned)((ph - 1) * p.dy_rs + (pw - 1) * p.dy_ps);
      } else {
        // S x S regions of a shared map (the quadrant / sub-quadrant heads: Quadtree_from scratch/models.py:277-287, :62-78),
        // every region behind its own pad row and column: a tap that leaves the region reads zeros, i.e. the zero halo at
        // the seams.  X is the un-split map, dY the dense per-region images [map][S*S][R][R][N].
        const int qr = ph / p.reg, qc = pw / p.reg;
        const int rr2 = ph - qr * p.reg - 1, cc2 = pw - qc * p.reg - 1;
        ok = rr2 >= 0 && cc2 >= 0;
        xo = (unsigned)((qr * p.regR + rr2) * p.x_rs + (qc * p.regR + cc2) * p.x_ps);
        yo = (unsigned)((((qr * p.regS + qc) * p.regR + rr2) * p.regR + cc2) * p.N);
      }
      tx[r] = ok ? xo * 2u + nx : kOob;
      ty[r] = ok ? yo * 2u + ny : kOob;
    }
  }

  // ---- DMA side: a unit = 8 rows = 1 KiB; lane = (row of the unit, 16-byte LDS slot); swizzle on the source ----
  const i32x4 rx = make_rsrc(p.x, a.x_bytes), ry = make_rsrc(p.dy, a.dy_bytes);
  const int drow = lane >> 3, dslot = lane & 7;
  const int dkey = (drow >> 1) & 3;
  const unsigned lane_c = (unsigned)((((dslot >> 1) ^ dkey) << 1) | (dslot & 1)) * 16u;
  const unsigned x_is2 = (unsigned)p.x_is * 2u, y_is2 = (unsigned)p.dy_is * 2u;

  // This wave's share of a tile: units wave, wave + 8, ... of the X region and of the dY region.  A walk keeps the
  // (scalar) image base and remainder of its unit's first row: 64 positions further per step, at most one image wrap
  // (PP >= 64), and this lane's table entry of its pending unit.  A slot has NO branch (a branch makes the compiler
  // drain every outstanding fragment read with lgkmcnt(0) at the join, three times per chunk): when the walk has
  // nothing left the transfer goes through a resource of zero records (no memory traffic) into a spare KiB of LDS.
  int x_left = 0, y_left = 0;                   // units still to issue
  unsigned x_rem = 0, x_base = 0, x_dst = 0;    // first row's position inside its image, image base + channel offset
  unsigned y_rem = 0, y_base = 0, y_dst = 0;    // (bytes), LDS address of the unit
  unsigned x_tv = 0, y_tv = 0;                  // this lane's table entry of the pending unit of either walk
  const unsigned tab_x = a.tab + (unsigned)drow * 4u, tab_y = tab_x + (unsigned)tabn * 4u;
  const unsigned spare = a.tab + 2u * (unsigned)tabn * 4u;
  auto sgpr = [](unsigned v) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
#define QT_WALK_BEGIN(W, units, pos0, is2, coff, dst0)                                        \
  {                                                                                           \
    const int u_ = (units);                                                                   \
    W##_left = __builtin_amdgcn_readfirstlane(u_ > wave ? (u_ - wave + 7) >> 3 : 0);          \
    const unsigned g_ = (unsigned)((pos0) + 8 * wave + p.PP);                                 \
    const unsigned q_ = fdiv(g_, p.div_pp);                                                   \
    W##_rem = sgpr(g_ - q_ * (unsigned)p.PP);                                                 \
    W##_base = sgpr((q_ - 1u) * (is2) + (coff));                                              \
    W##_dst = sgpr((dst0) + (unsigned)wave * 1024u);                                          \
  }
#define QT_WALK_ISSUE(W, tv_, rsrc, bytes, is2)                                               \
  {                                                                                           \
    const bool any_ = W##_left > 0;                                                           \
    i32x4 r_ = rsrc;                                                                          \
    r_.z = (int)sgpr(any_ ? (bytes) : 0u);                                                    \
    blds16(r_, (tv_) + W##_base + lane_c, 0u, sgpr(any_ ? W##_dst : spare));                  \
    const unsigned r1_ = W##_rem + 64u;                                                       \
    const bool wrap_ = r1_ >= (unsigned)p.PP;                                                 \
    W##_left = __builtin_amdgcn_readfirstlane(W##_left - (any_ ? 1 : 0));                     \
    W##_dst = sgpr(W##_dst + 8192u);                                                          \
    W##_rem = sgpr(wrap_ ? r1_ - (unsigned)p.PP : r1_);                                       \
    W##_base = sgpr(wrap_ ? W##_base + (is2) : W##_base);                                     \
  }
  auto tile_begin = [&](int k) {  // tile k -> buffer k & 1
    const int nt = min(NT, nch - k * NT);
    const int q0 = p0 + k * NT * WP_CH;
    const unsigned bufbase = (unsigned)(k & 1) * a.bufb;
    QT_WALK_BEGIN(x, (nt * WP_CH + 2 * a.HL) >> 3, q0 - a.HL, x_is2, (unsigned)c0 * 2u, bufbase)
    QT_WALK_BEGIN(y, nt * (WP_CH / 8), q0, y_is2, (unsigned)n0 * 2u, bufbase + a.xb)
    x_tv = *reinterpret_cast<const unsigned*>(smem + tab_x + x_rem * 4u);
    y_tv = *reinterpret_cast<const unsigned*>(smem + tab_y + y_rem * 4u);
  };
  // One slot: the transfer of the walk's pending unit, then the table entry of the one after it.  The read is pinned
  // here (sched_barrier) so that it is OLD when the next slot of the walk consumes it: LDS returns in order, a wait for
  // a fresh read would wait for every fragment read issued before it.
  auto dma_x = [&]() {
    QT_WALK_ISSUE(x, x_tv, rx, a.x_bytes, x_is2)
    x_tv = *reinterpret_cast<const unsigned*>(smem + tab_x + x_rem * 4u);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto dma_y = [&]() {
    QT_WALK_ISSUE(y, y_tv, ry, a.dy_bytes, y_is2)
    y_tv = *reinterpret_cast<const unsigned*>(smem + tab_y + y_rem * 4u);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto dma_rest = [&]() {   // whatever the slots of a tile did not cover (the first tile; short tiles, wide halos)
    while (x_left > 0) dma_x();
    while (y_left > 0) dma_y();
  };
#undef QT_WALK_BEGIN
#undef QT_WALK_ISSUE

  // ---- MFMA side ----
  const int li = lane & 15, lg = lane >> 4;
  const int qq = li >> 2, pp = li & 3;
  const int lrow = 4 * lg + qq;  // second transposing read: lrow + 16 (same swizzle key)
  unsigned a_base[4], b_base[9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    a_base[i] = a.xb + (unsigned)(group * WP_CH + lrow) * 128u + ((unsigned)(i ^ ((lrow >> 1) & 3)) << 5) + pp * 8;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int r = a.HL + (t / 3 - 1) * p.PW + (t % 3 - 1) + group * WP_CH + lrow;  // >= 0: HL >= PW + 1
    b_base[t] = (unsigned)r * 128u + ((unsigned)(wq ^ ((r >> 1) & 3)) << 5) + pp * 8;
  }
  auto frag = [&](unsigned addr) -> uint4 {   // one 16-byte MFMA operand: rows lrow and lrow + 16 of a chunk
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(lds_tr_ptr(addr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(lds_tr_ptr(addr + 2048));
    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
  };

  f32x4 acc[4][9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // tables visible, then tile 0 into buffer 0
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  tile_begin(0);
  dma_rest();
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  // One barrier per tile; inside it a wave runs its NJ chunks as one software pipeline: the fragments of X two taps
  // ahead of the MFMAs, the next chunk's dY fragments one per tap, a DMA slot of the next tile on taps 0 and 4 (X) and
  // 7 (dY).  (A barrier-separated ping-pong of the two groups -- fragment reads of one under the MFMAs of the other --
  // measured SLOWER here, 71-76 us against 58-62: its load segment is paced by the LDS (26 transposing reads x 4 waves
  // ~ 500 cycles) and every DMA instruction costs the segment that issues it ~65 cycles, so neither segment fits under
  // the 576 cycles of the other's MFMAs.)
  for (int k = 0; k < ntiles; ++k) {
    const int nt = min(NT, nch - k * NT);
    const int nj = (nt - group + 1) >> 1;  // chunks group, group + 2, ... < nt
    if (k + 1 < ntiles) {
      tile_begin(k + 1);
    } else {
      x_left = y_left = 0;
    }
    constexpr unsigned boff = 0;  // the fragment bases point into the tile's buffer (moved at the end of the tile)
    uint4 fa[2][4], fb[3];
    if (nj > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[0][i] = frag(a_base[i] + boff);
      fb[0] = frag(b_base[0] + boff);
      fb[1] = frag(b_base[1] + boff);
    }
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      if (jj < nj) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          constexpr int kStride = 2 * WP_CH * 128;  // bytes between a group's consecutive chunks
          const int s2 = jj * 9 + t + 2;            // fragment of X two taps ahead (a third costs 4 registers)
          if (s2 / 9 < NJ) fb[s2 % 3] = frag(b_base[s2 % 9] + boff + (unsigned)(s2 / 9) * kStride);
          if (t >= 2 && t < 6 && jj + 1 < NJ)       // the next chunk's dY fragments, one per tap
            fa[(jj + 1) & 1][t - 2] = frag(a_base[t - 2] + boff + (unsigned)(jj + 1) * kStride);
          if (t == 0 || t == 4) dma_x();            // 2 NJ slots for X, NJ for dY (exactly its share when the next
          if (t == 7) dma_y();                      // tile is full)
          // (raised priority for the four MFMAs of a tap: the SIMD's other wave gets its reads and DMA slots in between,
          // not in the middle of a burst -- 1-3 % per launch; a start skew between the two wave groups measured slower)
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[jj & 1][i]),
                                                                __builtin_bit_cast(bf16x8, fb[(jj * 9 + t) % 3]),
                                                                acc[i][t], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
        }
      }
    }
    dma_rest();
    {  // fragment bases -> the other buffer (in place: a second set of 13 registers is what the kernel cannot afford
       // next to the small kernels of the main stream, see the launcher)
      const unsigned delta = (k & 1) ? 0u - a.bufb : a.bufb;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a_base[i] += delta;
        asm volatile("" : "+v"(a_base[i]));
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        b_base[t] += delta;
        asm volatile("" : "+v"(b_base[t]));
      }
    }
    // the next tile has landed (this wave's share; the barrier makes it everybody's) and nobody reads this buffer again
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // ---- group 1 hands o-blocks 0,1 to group 0 and takes o-blocks 2,3 from it ----
  {
    f32x4* xch = reinterpret_cast<f32x4*>(smem) + (wq * 18) * 64 + lane;  // [wq][18][lane] x 16 B
    if (group == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) xch[(i * 9 + t) * 64] = acc[i][t];
    }
    __syncthreads();
    if (group == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] += xch[(i * 9 + t) * 64];
    }
    __syncthreads();
    if (group == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) xch[(i * 9 + t) * 64] = acc[2 + i][t];
    }
    __syncthreads();
    if (group == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[2 + i][t] += xch[(i * 9 + t) * 64];
    }
  }

  // ---- accumulate: lane holds o = 16*i + 4*lg + r, input channel 16*wq + li of every tap ----
  const int cc = c0 + wq * 16 + li;
  auto flush = [&](int i, const f32x4 (&v)[9]) {
    if (p.part) {
      float* base = p.part + (long long)split * p.N * 9 * p.KC;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + i * 16 + lg * 4 + r;
        float* row = base + (long long)n * 9 * p.KC + cc;
#pragma unroll
        for (int t = 0; t < 9; ++t) row[t * p.KC] = v[t][r];
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + i * 16 + lg * 4 + r;
      float* row = p.dw + (long long)n * 9 * p.KC + cc;
#pragma unroll
      for (int t = 0; t < 9; ++t) atomicAdd(row + t * p.KC, v[t][r]);
    }
  };
  if (group == 0) {
    flush(0, acc[0]);
    flush(1, acc[1]);
  } else {
    flush(2, acc[2]);
    flush(3, acc[3]);
  }
}

// dw[j] += sum over ranges of part[range][j].  A block sums JQ = 256/SG float4 columns; thread
// group g adds its contiguous share of the ranges in ascending order and the groups are added in
// ascending order, so the result does not depend on timing.  SG is chosen so that a thread has
// few (<= 8) dependent loads: the launch overlaps HBM-heavy kernels and must not be a latency chain.
template <int SG>
__global__ __launch_bounds__(256) void wgrad_partial_sum_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                int nq, int nsplit, long long stride_q, int KC, int oihw) {
  constexpr int JQ = 256 / SG;
  __shared__ float4 red[SG][JQ];
  const int col = threadIdx.x % JQ, sg = threadIdx.x / JQ;
  const int jq = blockIdx.x * JQ + col;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (jq < nq) {
    const float4* src = reinterpret_cast<const float4*>(part) + jq;
    const int per = (nsplit + SG - 1) / SG;
    const int beg = sg * per, end = min(nsplit, beg + per);
#pragma unroll 8
    for (int k = beg; k < end; ++k) {
      const float4 v = src[(long long)k * stride_q];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[sg][col] = s;
  __syncthreads();
  if (sg == 0 && jq < nq) {
    float4 o = oihw ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(dw)[jq];
#pragma unroll
    for (int g = 0; g < SG; ++g) {
      const float4 v = red[g][col];
      o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    }
    if (oihw != 1) {   // 0: accumulated into dw;  2: written as is
      reinterpret_cast<float4*>(dw)[jq] = o;
    } else {
      // j = (n*9 + tap)*KC + c  ->  OIHW element (n*KC + c)*9 + tap: the gradient is WRITTEN in the
      // reference's layout, no [O][kh][kw][I] scratch, no zero fill, no separate permutation launch
      const long long j = (long long)jq * 4;
      const int c = (int)(j % KC);
      const long long nt = j / KC;
      const int tap = (int)(nt % 9);
      const long long n = nt / 9;
      float* dst = dw + (n * KC + c) * 9 + tap;
      dst[0] = o.x; dst[9] = o.y; dst[18] = o.z; dst[27] = o.w;
    }
  }
}

// Light-footprint form (round 4).  The sum runs on the weight-gradient stream BESIDE the main stream's data-gradient kernels,
// which hold 460-512 of a SIMD's 512 registers per lane and all but 4 KB of a CU's LDS: the kernel above (66 VGPRs, 4 KB of
// LDS) cannot start a wave until one of their whole-CU workgroups retires, and in the step it measures 20-70 us per launch
// against 9-14 us alone (13 launches: ~0.6 ms of the stream that bounds the backward pass).  Here a wave owns 8 float4
// columns x 8 groups of ranges: lane = group * 8 + column adds ranges group, group + 8, ... in ascending order, four loads
// in flight, and the eight groups are added by three butterfly exchanges in a fixed order -- no LDS, <= 32 VGPRs, so its
// waves fit beside anything.  Bit-reproducible (fixed order), not bit-identical to the form above (another order).
__global__ __launch_bounds__(256) void wgrad_partial_sum_light_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                      int nq, int nsplit, long long stride_q, int KC, int oihw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 7, grp = lane >> 3;
  const int jq = (blockIdx.x * 4 + wave) * 8 + col;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (jq < nq) {
    const float4* src = reinterpret_cast<const float4*>(part) + jq + (long long)grp * stride_q;
    const long long step = 8 * stride_q;
    int k = grp;
    for (; k + 24 < nsplit; k += 32) {
      const float4 v0 = src[0], v1 = src[step], v2 = src[2 * step], v3 = src[3 * step];
      src += 4 * step;
      s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
      s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
      s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
      s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
    }
    for (; k < nsplit; k += 8) {
      const float4 v = src[0];
      src += step;
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
#pragma unroll
  for (int m = 8; m < 64; m <<= 1) {   // groups 0..7 -> every lane of a column holds the total (same order on every lane pair)
    s.x += __shfl_xor(s.x, m, 64);
    s.y += __shfl_xor(s.y, m, 64);
    s.z += __shfl_xor(s.z, m, 64);
    s.w += __shfl_xor(s.w, m, 64);
  }
  if (grp != 0 || jq >= nq) return;
  if (oihw != 1) {
    float4* o = reinterpret_cast<float4*>(dw) + jq;
    if (oihw == 0) {
      const float4 v = *o;
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *o = s;
  } else {
    const long long j = (long long)jq * 4;
    const int c = (int)(j % KC);
    const long long nt = j / KC;
    const int tap = (int)(nt % 9);
    const long long n = nt / 9;
    float* dst = dw + (n * KC + c) * 9 + tap;
    dst[0] = s.x; dst[9] = s.y; dst[18] = s.z; dst[27] = s.w;
  }
}

// The sum of a weight gradient's partial filters may run on ANOTHER stream than the kernel that wrote them
// (qt_conv2d_wgrad_oihw_on): set around that one call, consumed here.
thread_local hipStream_t g_sum_stream = nullptr;

// QTCNN_WGRAD_SUM (default 1 = the light form; 0 = the LDS form above, same-box A/B)
int sum_partials(const float* part, float* dw, int nq, int nsplit, int KC, int layout, hipStream_t stream) {
  if (g_sum_stream && g_sum_stream != stream) {   // behind the kernel on `stream`, but not in ITS queue
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, stream) != hipSuccess ||
        hipStreamWaitEvent(g_sum_stream, ev, 0) != hipSuccess) {
      qt_set_error("qt_conv2d_wgrad_oihw_on: HIP event error");
      return QT_ERR_LAUNCH;
    }
    (void)hipEventDestroy(ev);   // (deferred until the event has completed)
    stream = g_sum_stream;
  }
  static int light = -1;
  if (light < 0) {
    const char* e = getenv("QTCNN_WGRAD_SUM");
    light = e ? atoi(e) : 1;
  }
  if (light && nsplit > 16)   // (few ranges: half of the light form's lanes would idle; r04v trace: 78-83 us against 45-71 in the step)
    hipLaunchKernelGGL(wgrad_partial_sum_light_kernel, dim3(qt_cdiv(nq, 32)), dim3(256), 0, stream, part, dw, nq, nsplit,
                       (long long)nq, KC, layout);
  else if (nsplit > 16)
    hipLaunchKernelGGL(wgrad_partial_sum_kernel<32>, dim3(qt_cdiv(nq, 8)), dim3(256), 0, stream, part, dw, nq, nsplit,
                       (long long)nq, KC, layout);
  else
    hipLaunchKernelGGL(wgrad_partial_sum_kernel<8>, dim3(qt_cdiv(nq, 32)), dim3(256), 0, stream, part, dw, nq, nsplit,
                       (long long)nq, KC, layout);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

int g_wgrad_patch_min_w = -1;  // smallest image width routed here; 0 = off

int min_w() {
  if (g_wgrad_patch_min_w < 0) {
    const char* e = getenv("QTCNN_WGRAD_PATCH_MIN_W");
    g_wgrad_patch_min_w = e ? atoi(e) : 7;
  }
  return g_wgrad_patch_min_w;
}

// ranges of positions the grid is cut into for `total` positions and `tiles` channel tiles
void split_ranges(int total, int tiles, int mcr, int* pps_out, int* nsplit_out) {
  int nsplit = 256 / tiles;
  if (nsplit < 1) nsplit = 1;
  int pps = qt_cdiv(total, nsplit);
  pps = qt_cdiv(pps, mcr) * mcr;
  *pps_out = pps;
  *nsplit_out = qt_cdiv(total, pps);
}

constexpr int kGroups = 1;  // wave groups of the default variant (split_ranges depends on it)

template <int G, int D, int NX, int ND>
int launch_patch(WPArgs a, size_t part_bytes, int oihw, hipStream_t stream) {
  constexpr int MCR = WP_CH * G;
  constexpr int LDS = (NX + ND) * MCR * 128;
  if (2 * a.halo > a.PP || (2 * a.halo) % MCR != 0 || (2 * a.halo) / MCR + D + 2 > NX || D + 2 > ND) {
    qt_set_error("qt_conv2d_wgrad: image %dx%d does not fit the streaming kernel", a.H, a.W);
    return QT_ERR_INVALID_ARG;
  }
  a.adv_h = MCR / a.PW;
  a.adv_w = MCR % a.PW;
  int real_split;
  split_ranges(a.total, a.tiles, MCR, &a.pps, &real_split);
  a.nsplit = real_split;
  if (a.nsplit >= 6 && (a.nsplit & 7)) a.nsplit = qt_cdiv(a.nsplit, 8) * 8;  // empty tail ranges exit at once
  const size_t filt = (size_t)a.N * 9 * a.KC;
  if (a.part && part_bytes < (size_t)real_split * filt * 4) a.part = nullptr;  // too small: atomics
  if (oihw && !a.part) {
    qt_set_error("qt_conv2d_wgrad_oihw: workspace of %zu bytes needed", (size_t)real_split * filt * 4);
    return QT_ERR_INVALID_ARG;
  }
  auto kern = conv_wgrad_patch_kernel<G, D, NX, ND>;
  static std::atomic<unsigned long long> lds_limit_set{0};  // per device
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(kern), LDS, lds_limit_set)) return rc;
  hipLaunchKernelGGL(kern, dim3(a.tiles * a.nsplit), dim3(256 * G), LDS, stream, a);
  QT_CHECK_LAUNCH();
  if (a.part) return sum_partials(a.part, a.dw, (int)(filt / 4), real_split, a.KC, oihw, stream);
  return QT_OK;
}

// geometry of the tile-resident kernel for this image size: chunks per tile (0 = does not fit), halo rows, LDS bytes
struct TileGeo {
  int nt, hl;
  unsigned xb, bufb, tab, lds;
};
TileGeo tile_geometry(int PW, int PP) {
  TileGeo g{0, (PW + 1 + 7) / 8 * 8, 0, 0, 0, 0};
  if (g.hl > PP) return g;
  static int lds_kb = -1;  // LDS the kernel may take: what it leaves is what the main stream's small kernels find
  if (lds_kb < 0) {
    const char* e = getenv("QTCNN_WT_LDS_KB");
    lds_kb = e ? atoi(e) : 160;
    if (lds_kb < 80 || lds_kb > 160) lds_kb = 160;
  }
  for (int nt = 8; nt >= 4; nt -= 2) {
    const unsigned xb = (unsigned)(nt * WP_CH + 2 * g.hl) * 128u, bufb = xb + (unsigned)nt * WP_CH * 128u;
    const unsigned lds = 2u * bufb + 2u * (unsigned)(PP + 8) * 4u + 1024u;  // tables + the spare KiB of empty DMA slots
    if (lds <= (unsigned)lds_kb * 1024u && 2u * bufb >= 4u * 18u * 64u * 16u) {
      g.nt = nt; g.xb = xb; g.bufb = bufb; g.tab = 2u * bufb; g.lds = lds;
      return g;
    }
  }
  return g;
}

template <int NT>
int launch_tile_nt(const WTArgs& t, unsigned lds, hipStream_t stream) {
  auto kern = conv_wgrad_tile_kernel<NT>;
  static std::atomic<unsigned long long> lds_limit_set{0};  // per device
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(kern), (int)lds, lds_limit_set)) return rc;
  hipLaunchKernelGGL(kern, dim3(t.p.tiles * t.p.nsplit), dim3(512), lds, stream, t);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

// -1: the shape is not for the tile kernel (caller takes the ring kernel), else a status
int launch_tile(WPArgs a, size_t part_bytes, int oihw, hipStream_t stream) {
  const TileGeo g = tile_geometry(a.PW, a.PP);
  const unsigned long long xbytes = (unsigned long long)a.B * (unsigned long long)a.x_is * 2ull;
  const unsigned long long ybytes = (unsigned long long)a.B * (unsigned long long)a.dy_is * 2ull;
  if (!g.nt || xbytes >= (1ull << 30) || ybytes >= (1ull << 30)) return -1;
  int real_split;
  split_ranges(a.total, a.tiles, WP_CH, &a.pps, &real_split);
  a.nsplit = real_split;
  if (a.nsplit >= 6 && (a.nsplit & 7)) a.nsplit = qt_cdiv(a.nsplit, 8) * 8;  // empty tail ranges exit at once
  const size_t filt = (size_t)a.N * 9 * a.KC;
  if (a.part && part_bytes < (size_t)real_split * filt * 4) a.part = nullptr;  // too small: atomics
  if (oihw && !a.part) {
    qt_set_error("qt_conv2d_wgrad_oihw: workspace of %zu bytes needed", (size_t)real_split * filt * 4);
    return QT_ERR_INVALID_ARG;
  }
  WTArgs t;
  t.p = a;
  t.HL = g.hl; t.xb = g.xb; t.bufb = g.bufb; t.tab = g.tab;
  t.x_bytes = (unsigned)xbytes; t.dy_bytes = (unsigned)ybytes;
  int rc = g.nt == 8 ? launch_tile_nt<8>(t, g.lds, stream)
         : g.nt == 6 ? launch_tile_nt<6>(t, g.lds, stream) : launch_tile_nt<4>(t, g.lds, stream);
  if (rc != QT_OK) return rc;
  if (a.part) return sum_partials(a.part, a.dw, (int)(filt / 4), real_split, a.KC, oihw, stream);
  return QT_OK;
}

int g_wp_variant = -1;  // 3: tile-resident kernel (default), 0: ring kernel, one wave group, 2: ring kernel, two groups
int wp_variant() {
  if (g_wp_variant < 0) {
    const char* e = getenv("QTCNN_WP_VARIANT");
    g_wp_variant = e ? atoi(e) : 3;
  }
  return g_wp_variant;
}

}  // namespace

// dw = sum over `nsplit` ranges of part[range][filt] in a fixed order (conv_wgrad_s2.hip shares the reduction).  layout 0:
// added to dw ([O][taps][I]); 1: written to OIHW from [O][9][I]; 2: written as is.
void qt_wgrad_set_sum_stream(void* s) { g_sum_stream = static_cast<hipStream_t>(s); }
int qt_wgrad_partial_sum_launch(const float* part, float* dw, size_t filt, int nsplit, int KC, int layout, hipStream_t stream) {
  return sum_partials(part, dw, (int)(filt / 4), nsplit, KC, layout, stream);
}

// Smallest image width whose 3x3 stride-1 weight gradients take the streaming kernel
// (0 = never; default 7 = every eligible layer of the model, the 7x7 stage included; env QTCNN_WGRAD_PATCH_MIN_W).
extern "C" void qt_set_wgrad_patch_min_width(int w) { g_wgrad_patch_min_w = w < 0 ? 7 : w; }
// Which streaming kernel: 3 = tile-resident (default), 0 = ring with one wave group, 2 = ring with two (env
// QTCNN_WP_VARIANT; negative = default).  Same sums in a different order: results agree to f32 rounding.
extern "C" void qt_set_wgrad_patch_variant(int v) { g_wp_variant = v < 0 ? 3 : v; }

bool qt_wgrad_patch_eligible(const qt_conv_desc* d) {
  const int mw = min_w();
  if (mw <= 0 || d->dtype != QT_BF16) return false;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1) return false;
  if (d->in_h != d->out_h || d->in_w != d->out_w) return false;
  if (d->n_out % 64 || d->k_per_tap % 64) return false;
  if (d->out_w < mw || d->out_w < 7 || d->out_h < 7 || d->out_w > 120) return false;
  if (d->quad) {   // region heads: the tile kernel only (offset tables), square regions
    const int S = qt_quad_split(d->quad);
    if (wp_variant() != 3 || d->in_h != d->in_w) return false;
    const int PW = S * (d->in_w + 1);
    if (!tile_geometry(PW, PW * PW).nt) return false;
    if ((long long)d->batch * PW * PW >= (1ll << 30)) return false;
    if ((long long)d->batch * d->src_img_stride * 2 >= (1ll << 30) ||
        (long long)d->batch * S * S * d->out_h * d->out_w * d->n_out * 2 >= (1ll << 30)) return false;
    return true;
  }
  if ((long long)d->batch * (d->out_h + 1) * (d->out_w + 1) >= (1ll << 30)) return false;
  return true;
}

// padded positions per image (per map in region mode)
static int padded_positions(const qt_conv_desc* d) {
  if (d->quad) {
    const int PW = qt_quad_split(d->quad) * (d->in_w + 1);
    return PW * PW;
  }
  return (d->out_h + 1) * (d->out_w + 1);
}

// bytes of partial-filter workspace the deterministic path wants for this convolution
size_t qt_wgrad_patch_workspace_bytes(const qt_conv_desc* d) {
  if (!qt_wgrad_patch_eligible(d)) return 0;
  const int total = d->batch * padded_positions(d);
  int pps, nsplit;
  split_ranges(total, (d->n_out / 64) * (d->k_per_tap / 64), WP_CH * kGroups, &pps, &nsplit);
  return (size_t)nsplit * d->n_out * 9 * d->k_per_tap * 4;
}

int qt_wgrad_patch_launch(const qt_conv_desc* d, const void* dy, const void* x, float* dw, void* workspace,
                          size_t workspace_bytes, int oihw, void* stream) {
  WPArgs a;
  a.part = static_cast<float*>(workspace);
  a.dy = static_cast<const bf16_t*>(dy);
  a.x = static_cast<const bf16_t*>(x);
  a.dw = dw;
  a.N = d->n_out; a.KC = d->k_per_tap; a.H = d->out_h; a.W = d->out_w; a.B = d->batch;
  a.reg = a.regR = a.regS = 0;
  a.PW = a.W + 1; a.PH = a.H + 1; a.PP = a.PW * a.PH;
  a.x_is = d->src_img_stride; a.x_rs = d->src_row_stride; a.x_ps = d->src_pix_stride;
  a.dy_ps = a.N; a.dy_rs = a.W * a.N; a.dy_is = (long long)a.H * a.dy_rs;
  if (d->quad) {   // (eligibility checked the tile kernel is on and fits)
    a.regS = qt_quad_split(d->quad); a.regR = d->in_w; a.reg = a.regR + 1;
    a.PW = a.PH = a.regS * a.reg; a.PP = a.PW * a.PH;
    a.W = a.PW - 1; a.H = a.PH - 1;   // (what the geometry helpers below derive the pitch from)
    a.dy_is = (long long)a.regS * a.regS * a.regR * a.regR * a.N;
  }
  a.total = a.B * a.PP;
  a.halo = (a.PW + 1 + 31) / 32 * 32;
  a.div_pp = make_fastdiv((unsigned)a.PP);
  a.div_pw = make_fastdiv((unsigned)a.PW);
  a.tilesC = a.KC / 64;
  a.tiles = (a.N / 64) * a.tilesC;
  a.adv_h = a.adv_w = a.pps = a.nsplit = 0;
  const int variant = wp_variant();
  static int use_ws = -1;
  if (use_ws < 0) {
    const char* e = getenv("QTCNN_WGRAD_WS");
    use_ws = e ? atoi(e) : 1;
  }
  if (!use_ws && !oihw) a.part = nullptr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // Alone, two wave groups win (56x56 64->64, atomics: 99 us vs 122 us): a second wave per SIMD
  // covers LDS reads and address arithmetic.  Inside the training step the launch overlaps the
  // main stream's data-gradient / BatchNorm kernels, and two 252-VGPR waves per SIMD leave no room
  // for any of their waves: short kernels then queue behind whole workgroups.  One group per
  // workgroup (one wave per SIMD, 96 KB LDS) keeps half of every CU's registers free and measures
  // 1.7 % faster per step (7.43 vs 7.56 ms); 48 KB of LDS with a shallower ring measures slower (7.60).
  if (variant == 3) {
    const int rc = launch_tile(a, workspace_bytes, oihw, s);
    if (rc >= 0) return rc;
  }
  if (a.reg) {   // (the ring kernels walk plain images only)
    qt_set_error("qt_conv2d_wgrad: region mode needs the tile-resident kernel");
    return QT_ERR_UNSUPPORTED;
  }
  if (variant == 2) return launch_patch<2, 2, 8, 4>(a, workspace_bytes, oihw, s);
  return launch_patch<kGroups, 4, 16, 8>(a, workspace_bytes, oihw, s);
}
