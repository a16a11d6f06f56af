// Weight-gradient of a convolution / linear layer on MFMA for gfx950.
//
// Replaces the conv2d / linear backward-weight ATen calls that loss.backward()
// makes (/root/reference/Quadtree_from scratch/Quadtree_train.py:65) for the
// layers built at Quadtree_from scratch/models.py:222-243,234-238,266-271.
//
//   dW[n][tap][c] += sum_pix dY[pix][n] * X[pix moved by tap][c]
//
// The contraction runs over PIXELS, which is the strided axis of both NHWC
// operands.  Tiles are staged [pixel][channel] (channel contiguous, coalesced
// 16-byte loads, zero fill of the halo in the register stage) and the
// pixel-major MFMA fragments are produced by the hardware transposing LDS read
// ds_read_b64_tr_b16 (bf16) or by plain ds_read_b32 (f32: one k per lane).  The
// LDS image is XOR-swizzled per 32-byte block so both kinds of read are bank
// conflict free.  The pixel axis is split over workgroups; partial tiles are
// accumulated with global_atomic_add_f32 into the zero-initialised f32 gradient.
#include <string.h>

#include "qt_common.h"

namespace {

struct WgradArgs {
  const void* dy;   // [M][N]
  const void* x;    // source activations
  float* dw;        // [N][ntaps][KC] f32, accumulated into
  long long x_img_stride;
  int x_row_stride, x_pix_stride;
  int M, N, KC;
  int OH, OW, IH, IW;
  int ntaps, KW, stride, pad;
  int quad;
  int tilesN, tilesC, gtaps, ksplit, pix_per_split;
  int tap_stride;   // STEM: element distance between virtual taps (rows of the padded image)
  FastDiv div_ohw, div_ow;
  // pixel walk of the X loader (element offsets): one pixel / column wrap / row wrap / the
  // jump over the KP - LB pixels other threads stage
  long long step1, wrap_w, wrap_h, adv_off;
  int adv_h, adv_w;
  int overwrite;    // 1 (only with ksplit == 1): every element of dw has ONE producer -- plain stores, dw need not be zeroed
};

constexpr int KP = 64;  // pixels per K-step

template <typename T, int BMW, int BNW, bool STEM>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
  constexpr int ES = (int)sizeof(T);
  constexpr int EPC = 16 / ES;                 // elements per 16-byte chunk
  constexpr int RBA = BMW * ES;                // dY tile row bytes
  constexpr int NBA = RBA / 32;                // 32-byte blocks per dY row
  constexpr int BNWP = STEM ? 256 : BNW;       // padded X tile width (elements)
  constexpr int RBB = BNWP * ES;
  constexpr int NBB = RBB / 32;
  constexpr int CPRA = RBA / 16, CPRB = RBB / 16;  // 16-byte LDS slots per row
  constexpr int CPB = (BNW * ES) / 16;             // slots of an X row that carry data
  constexpr int LA = (KP * CPRA) / 256;            // dY DMA instructions per wave per step
  constexpr int LB = (KP * CPRB) / 256;            // X DMA instructions per wave per step
  constexpr int STAGE = KP * (RBA + RBB);
  constexpr int TMW = BMW / 32, TNW = BNW / 32;
  static_assert(NBA == 4 || NBA == 8 || NBA == 16, "dY row must be 128/256/512 bytes");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const T* __restrict__ dy = static_cast<const T*>(p.dy);
  const T* __restrict__ x = static_cast<const T*>(p.x);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;

  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so XCD x runs block ids
  // x, x+8, ...  Give each XCD whole pixel splits and run all (tap, tile) members of a split
  // back to back on it: the 9 taps re-read the same dY / X rows from that XCD's L2.
  // (only when the split count is a multiple of 8; few-split shapes keep the linear order.)
  const int members = p.tilesN * p.tilesC * p.gtaps;
  int split, bid;
  if ((p.ksplit & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    split = (slot / members) * 8 + xcd;
    bid = slot % members;
  } else {
    split = blockIdx.x % p.ksplit;
    bid = blockIdx.x / p.ksplit;
  }
  const int tc = bid % p.tilesC; bid /= p.tilesC;
  const int tn = bid % p.tilesN; bid /= p.tilesN;
  const int tap = bid;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int n0 = tn * BMW, c0 = tc * BNW;
  const int pbeg = split * p.pix_per_split;
  const int pend = min(p.M, pbeg + p.pix_per_split);
  const int nsteps = (pend - pbeg + KP - 1) / KP;

  auto key = [](int row, int nblk) { return nblk >= 8 ? (row & 7) : ((row >> 1) & 3); };

  // LDS-DMA staging (see conv_igemm.hip): thread = (row in pass, 16-byte LDS slot); the
  // 32-byte-block swizzle sits on the source side, slot s of row r holds source chunk
  // (((s>>1) ^ key(r)) << 1) | (s & 1).  Rows advance by a multiple of 8 per pass, so the
  // source chunk of a thread is fixed.
  constexpr int RPA = 256 / CPRA, RPB = 256 / CPRB;  // rows per pass
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sA = tid % CPRA, rA = tid / CPRA, sB = tid % CPRB, rB = tid / CPRB;
  const int cA = (((sA >> 1) ^ key(rA, NBA)) << 1) | (sA & 1);
  const bool a_col_ok = n0 + cA * EPC < p.N;
  // X rows: the source chunk is fixed per thread when a pass covers a multiple of 8 rows,
  // otherwise (f32 stem: 4 rows per pass) it is recomputed per pass from the row's key
  auto b_chunk = [&](int row) { return (((sB >> 1) ^ key(row, NBB)) << 1) | (sB & 1); };
  auto b_chunk_off = [&](int cB, bool& ok) -> long long {
    if (STEM) {
      ok = cB < CPB;
      const int v0 = cB * EPC;  // virtual channel: (tap row, element)
      return (long long)(v0 / p.KC) * p.tap_stride + (v0 % p.KC);
    }
    ok = cB < CPB && c0 + cB * EPC < p.KC;
    return c0 + cB * EPC;
  };
  bool b_col_ok;
  long long b_coff = b_chunk_off(b_chunk(rB), b_col_ok);
  const T* zero_src = reinterpret_cast<const T*>(qt_zero_page);
  const unsigned smem_base = lds_addr_of(smem);

  // Pixel <-> LDS-row assignment.  The contraction index is the pixel, so the order of the
  // pixels inside a K-step is free as long as both tiles agree.  It is chosen so that every
  // thread stages LB CONSECUTIVE pixels of the X tile: LDS row rho = i*RPB + t  holds pixel
  // t*LB + i  (t = the thread's row slot in a pass, i = pass).  The walk over consecutive
  // pixels then needs no division and no multiplication: (oh, ow, offset) advance by one pixel
  // with two wrap tests, and by the rest of the K-step with host-computed constants.
  const int tB = rB;                                   // 0..RPB-1
  int pixA[LA], rowoffA[LA];                           // dY tile rows follow the same assignment
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int rho = i * RPA + rA;
    pixA[i] = (rho % RPB) * LB + rho / RPB;
    rowoffA[i] = pixA[i] * p.N;
  }
  // state of the walk: first pixel of this thread in the current K-step
  int st_m = pbeg + tB * LB, st_oh, st_ow;
  long long st_off;
  {
    const unsigned mm = (unsigned)st_m;
    const unsigned img = fdiv(mm, p.div_ohw);
    const unsigned rem = mm - img * (unsigned)(p.OH * p.OW);
    st_oh = (int)fdiv(rem, p.div_ow);
    st_ow = (int)(rem - (unsigned)st_oh * (unsigned)p.OW);
    st_off = (long long)img * p.x_img_stride + (long long)st_oh * p.stride * p.x_row_stride +
             (long long)st_ow * p.stride * p.x_pix_stride;
  }
  const int dh0 = kh - p.pad, dw0 = kw - p.pad;
  const long long tapoff = (long long)dh0 * p.x_row_stride + (long long)dw0 * p.x_pix_stride;
  long long dy_off = (long long)pbeg * p.N + n0 + cA * EPC;

  auto dma_stage = [&](int ks, int buf) {
    const int pb = pbeg + ks * KP;
    const unsigned sa = smem_base + buf * STAGE + wave_u * 1024;
    const unsigned sb = sa + KP * RBA;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const T* g = (pb + pixA[i] < pend && a_col_ok) ? dy + (dy_off + rowoffA[i]) : zero_src;
      glds16(g, sa + i * (RPA * RBA));
    }
    dy_off += (long long)KP * p.N;
    if (p.quad) {
      // quadrant images are not equally spaced in memory: general per-pixel decode
#pragma unroll
      for (int i = 0; i < LB; ++i) {
        const int m = pb + tB * LB + i;
        const T* g = zero_src;
        if (RPB % 8 != 0) b_coff = b_chunk_off(b_chunk(i * RPB + rB), b_col_ok);
        if (m < pend && b_col_ok) {
          unsigned img = fdiv((unsigned)m, p.div_ohw);
          unsigned rem = (unsigned)m - img * (unsigned)(p.OH * p.OW);
          unsigned oh = fdiv(rem, p.div_ow);
          unsigned ow = rem - oh * (unsigned)p.OW;
          const int S = p.quad, R = S * S;
          const int n = (int)img / R, q = (int)img - n * R;
          const int qr = q / S, qc = q - qr * S;
          const long long base = (long long)n * p.x_img_stride + (long long)qr * p.IH * p.x_row_stride +
                                 (long long)qc * p.IW * p.x_pix_stride;
          const int ih = (int)oh * p.stride + dh0, iw = (int)ow * p.stride + dw0;
          if ((unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW)
            g = x + base + (long long)ih * p.x_row_stride + (long long)iw * p.x_pix_stride + b_coff;
        }
        glds16(g, sb + i * (RPB * RBB));
      }
      return;
    }
    int oh = st_oh, ow = st_ow;
    long long off = st_off;
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      if (RPB % 8 != 0) b_coff = b_chunk_off(b_chunk(i * RPB + rB), b_col_ok);
      const int ih = oh * p.stride + dh0, iw = ow * p.stride + dw0;
      const bool ok = st_m + i < pend && b_col_ok && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      const T* g = ok ? x + (off + tapoff + b_coff) : zero_src;
      glds16(g, sb + i * (RPB * RBB));
      // next pixel
      ++ow;
      off += p.step1;
      if (ow == p.OW) {
        ow = 0;
        ++oh;
        off += p.wrap_w;
        if (oh == p.OH) {
          oh = 0;
          off += p.wrap_h;
        }
      }
    }
    // the remaining KP - LB pixels of the K-step in one jump
    st_m += KP;
    ow += p.adv_w;
    off += p.adv_off;
    if (ow >= p.OW) {
      ow -= p.OW;
      ++oh;
      off += p.wrap_w;
    }
    oh += p.adv_h;
    if (oh >= p.OH) {
      oh -= p.OH;
      off += p.wrap_h;
    }
    st_oh = oh;
    st_ow = ow;
    st_off = off;
  };

  f32x4 acc[TMW][TNW];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, lg = lane >> 4;

  if (nsteps > 0) dma_stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  for (int ks = 0; ks < nsteps; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nsteps) dma_stage(ks + 1, buf ^ 1);
    const unsigned char* sa = smem + buf * STAGE;
    const unsigned char* sb = sa + KP * RBA;
    if constexpr (sizeof(T) == 2) {
      // bf16: two transposing reads give the 8 k (pixels) of one lane:
      //   j=0..3 -> pixel kb+4g+j, j=4..7 -> pixel kb+16+4g+j-4
      const int q = li >> 2, pp = li & 3;
#pragma unroll
      for (int kb = 0; kb < KP; kb += 32) {
        uint4 fa[TMW], fb[TNW];
        const int r1 = kb + 4 * lg + q, r2 = r1 + 16;
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
          const int blk = (wm * (BMW / 2) + i * 16) >> 4;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (QT_LDS_AS s16x4*)(sa + r1 * RBA + ((blk ^ key(r1, NBA)) << 5) + pp * 8));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (QT_LDS_AS s16x4*)(sa + r2 * RBA + ((blk ^ key(r2, NBA)) << 5) + pp * 8));
          uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          fa[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
        }
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
          const int blk = (wn * (BNW / 2) + j * 16) >> 4;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (QT_LDS_AS s16x4*)(sb + r1 * RBB + ((blk ^ key(r1, NBB)) << 5) + pp * 8));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (QT_LDS_AS s16x4*)(sb + r2 * RBB + ((blk ^ key(r2, NBB)) << 5) + pp * 8));
          uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          fb[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
        }
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
          for (int j = 0; j < TNW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      }
    } else {
      // f32: lane supplies (channel li, pixel kb+lg) directly
#pragma unroll 4
      for (int kb = 0; kb < KP; kb += 4) {
        const int r = kb + lg;
        float fa[TMW], fb[TNW];
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
          const int ch = wm * (BMW / 2) + i * 16 + li;
          fa[i] = *reinterpret_cast<const float*>(sa + r * RBA + ((((ch >> 3) ^ key(r, NBA)) << 5) | ((ch & 7) << 2)));
        }
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
          const int ch = wn * (BNW / 2) + j * 16 + li;
          fb[j] = *reinterpret_cast<const float*>(sb + r * RBB + ((((ch >> 3) ^ key(r, NBB)) << 5) | ((ch & 7) << 2)));
        }
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
          for (int j = 0; j < TNW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // ---- accumulate: lane holds rows n = 4*lg + r, column c = li of each tile ----
  if (nsteps > 0) {
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < TNW; ++j) {
        const int cc = wn * (BNW / 2) + j * 16 + li;
        int tap_o, c_o;
        if (STEM) {
          tap_o = cc / p.KC;
          c_o = cc - tap_o * p.KC;
        } else {
          tap_o = tap;
          c_o = c0 + cc;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wm * (BMW / 2) + i * 16 + lg * 4 + r;
          if (n < p.N && c_o < p.KC && tap_o < p.ntaps) {
            float* o = p.dw + ((long long)n * p.ntaps + tap_o) * p.KC + c_o;
            if (p.overwrite) *o = acc[i][j][r];   // (uniform)
            else atomicAdd(o, acc[i][j][r]);
          }
        }
      }
  }
}

template <typename T, int BMW, int BNW, bool STEM>
int launch(WgradArgs a, hipStream_t stream) {
  constexpr int ES = (int)sizeof(T);
  constexpr int BNWP = STEM ? 256 : BNW;
  constexpr int LDS = 2 * KP * (BMW * ES + BNWP * ES);
  auto kern = conv_wgrad_kernel<T, BMW, BNW, STEM>;
  static std::atomic<unsigned long long> lds_limit_set{0};  // per device
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(kern), LDS, lds_limit_set)) return rc;
  a.tilesN = qt_cdiv(a.N, BMW);
  a.tilesC = STEM ? 1 : qt_cdiv(a.KC, BNW);
  const int gtaps = STEM ? 1 : a.ntaps;
  const int base_blocks = a.tilesN * a.tilesC * gtaps;
  // every workgroup ends with tile-sized f32 atomics (chip-wide ~1.3 TB/s): keep the grid at
  // about two workgroups per CU instead of maximising the split.
  // Occupancy was measured both ways for the 128x128 shape: single-buffered tiles (32 KB, three
  // workgroups per CU) run layer4's 3x3 weight gradient in 125 instead of 135 us alone but cost 3 % of the
  // training step (7.15 vs 6.94 ms) because they crowd the main stream's kernels off the CUs; padding LDS
  // to ONE workgroup per CU also costs 2.4 %.  Two per CU is the balance point of the two streams.
  int ksplit = qt_cdiv(512, base_blocks);
  const int max_split = qt_cdiv(a.M, KP * 4);
  if (ksplit > max_split) ksplit = max_split;
  if (ksplit < 1) ksplit = 1;
  if (ksplit >= 6) ksplit = qt_cdiv(ksplit, 8) * 8;  // whole splits per XCD (see the kernel)
  int pps = qt_cdiv(a.M, ksplit);
  pps = qt_cdiv(pps, KP) * KP;
  a.ksplit = qt_cdiv(a.M, pps);
  if (a.ksplit >= 6 && (a.ksplit & 7)) a.ksplit = qt_cdiv(a.ksplit, 8) * 8;  // empty tail splits exit at once
  a.pix_per_split = pps;
  a.gtaps = gtaps;
  if (a.overwrite && a.ksplit != 1) {   // several pixel ranges per tile: zero, then accumulate
    if (hipMemsetAsync(a.dw, 0, (size_t)a.N * a.ntaps * a.KC * 4, stream) != hipSuccess) {
      qt_set_error("qt_linear_wgrad: memset failed");
      return QT_ERR_LAUNCH;
    }
    a.overwrite = 0;
  }
  {
    constexpr int CPRBh = (STEM ? 256 : BNW) * ES / 16, LBh = KP / (256 / CPRBh);
    const long long srs = (long long)a.stride * a.x_row_stride, sps = (long long)a.stride * a.x_pix_stride;
    a.step1 = sps;
    a.wrap_w = srs - (long long)a.OW * sps;
    a.wrap_h = a.x_img_stride - (long long)a.OH * srs;
    const int jump = KP - LBh;  // pixels
    const int ohw = a.OH * a.OW;
    const int di = jump / ohw, r1 = jump % ohw;
    a.adv_h = r1 / a.OW;
    a.adv_w = r1 % a.OW;
    a.adv_off = (long long)di * a.x_img_stride + (long long)a.adv_h * srs + (long long)a.adv_w * sps;
  }
  hipLaunchKernelGGL(kern, dim3(base_blocks * a.ksplit), dim3(256), LDS, stream, a);
  QT_CHECK_LAUNCH();
  return QT_OK;
}


// ---------------------------------------------------------------------------------------------
// Stem (conv1 7x7 / 2, packed NHWC4 input) weight gradient from RAW input rows (round 3, bf16).
//
// The generic kernel above stages, per output pixel, its 7 x 64-byte window rows (40 KB per 64 pixels: ten LDS-DMA
// instructions per wave and K-step for 28 MFMAs -- DMA-issue bound, 184 us) although neighbouring pixels share 3/4 of
// every window row.  Here a workgroup stages the nine packed input rows of TWO output rows once (16.7 KB, one contiguous
// block of the [B][230][232][4] image) next to their 224 x 64 gradient tile (28 KB): 45 KB per 224 pixels.  The pixel-major
// MFMA fragments come from the transposing LDS read at PER-LANE addresses, so the im2col overlap costs nothing: the
// fragment of pixel ow, window row kh, columns kw = 4h .. 4h+3 is the 32 bytes at byte 16 ow + 32 h of row 2 oh + kh.
// Wave w < 7 owns window row kh = w: dW[64][kh][32] = 4 x 2 accumulator tiles; the pixel axis is the MFMA K axis
// (7 blocks of 32 per tile).  Workgroups walk tiles persistently and add their filters with f32 atomics at the end.
// ---------------------------------------------------------------------------------------------
constexpr int SW_ROWB = QT_STEM_PAD_W * 4 * 2;     // 1856 bytes per packed input row
constexpr int SW_XB = 17 * 1024;                   // >= 9 rows
constexpr int SW_DYB = 224 * 128;                  // 28 KB: two output rows x 112 pixels x 64 channels
constexpr int SW_BUF = SW_XB + SW_DYB;
constexpr int SW_LDS = 2 * SW_BUF;
static_assert(9 * SW_ROWB <= SW_XB, "nine input rows fit their block");

// FUSED form: the gradient tile is not read but COMPUTED in LDS -- max-pool backward (gather of the <= 4 pooled cells whose
// argmax is the position), ReLU mask and BatchNorm backward, exactly the arithmetic of stem_bn_bwd_apply2x2_kernel
// (elementwise.hip) -- from the raw conv1 output tile (same shape, staged in its place), two rows of the pooled gradient
// and of the argmax map: d(loss)/d(conv1 output), 411 MB at 256 images, is neither written nor read.
constexpr int SW_DPB = 2 * 56 * 128;               // two pooled-gradient rows
constexpr int SW_AMB = 2 * 56 * 64;                // two argmax rows
constexpr int SW_BUF_F = SW_BUF + SW_DPB + SW_AMB; // 66 KB
constexpr int SW_LDS_F = 2 * SW_BUF_F + 7 * 64 * 4;   // (+ the BatchNorm constants: scale, shift, mean, invstd, a, b, c)
static_assert(SW_LDS_F <= 160 * 1024, "fused stem backward fits the LDS");

struct StemWgArgs {
  const bf16_t* dy;   // [B][112][112][64]  (FUSED: the raw conv1 output y)
  const bf16_t* x;    // [B][230][232][4]
  float* dw;          // [64][7][32], accumulated into
  int ntiles;         // B * 56
  // FUSED
  const bf16_t* dpooled;            // [B][56][56][64]
  const unsigned char* argmax;      // [B][56][56][64]
  const float *scale, *shift, *mean, *invstd, *coef;   // [64] each, coef = [3][64] (a, b, c of qt_bn_bwd_finalize)
};

template <bool FUSED>
__global__ __launch_bounds__(512) void stem_wgrad_rows_kernel(StemWgArgs p) {
  constexpr int BUF = FUSED ? SW_BUF_F : SW_BUF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4, q = li >> 2, pp = li & 3;
  const unsigned smem_base = lds_addr_of(smem);
  const int t_beg = (int)((long long)blockIdx.x * p.ntiles / gridDim.x);
  const int t_end = (int)((long long)(blockIdx.x + 1) * p.ntiles / gridDim.x);
  if (t_beg >= t_end) return;

  // 45 transfers of 1 KB per tile, dealt over the 8 waves: 17 of the input rows (a linear copy), 28 of the gradient tile
  // (its 32-byte blocks XOR-swizzled by (row >> 1) & 3 on the source side, as in conv_wgrad_kernel: conflict-free reads)
  auto dma_tile = [&](int tile, int buf) {
    const int img = tile / 56, rp = tile - img * 56;
    const unsigned char* xs = reinterpret_cast<const unsigned char*>(p.x) + ((size_t)img * QT_STEM_PAD_H + 4 * rp) * SW_ROWB;
    const unsigned char* ds = reinterpret_cast<const unsigned char*>(p.dy) + ((size_t)img * 112 + 2 * rp) * (112 * 128);
#pragma unroll
    for (int i = 0; i < (FUSED ? 9 : 6); ++i) {
      const int b = wave + 8 * i;
      if (b < 17) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + buf * BUF + b * 1024);
        if (b * 1024 + lane * 16 < 9 * SW_ROWB) glds16(xs + b * 1024 + lane * 16, dst);
      } else if (b < 45) {
        const int d = b - 17, r = d * 8 + (lane >> 3), sl = lane & 7;
        const int c = (((sl >> 1) ^ ((r >> 1) & 3)) << 1) | (sl & 1);
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + buf * BUF + SW_XB + d * 1024);
        glds16(ds + r * 128 + c * 16, dst);
      } else if (FUSED && b < 66) {
        // pooled rows rp, rp + 1 (the second does not exist for rp = 55): 14 KB of gradient, 7 KB of argmax, linear
        const int e = b - 45;
        const bool two = rp + 1 < 56;
        if (e < 14) {
          const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + buf * BUF + SW_BUF + e * 1024);
          if (two || e < 7)
            glds16(reinterpret_cast<const unsigned char*>(p.dpooled) + ((size_t)img * 56 + rp) * (56 * 128) + e * 1024 + lane * 16, dst);
        } else {
          const int f = e - 14;
          const unsigned dst = __builtin_amdgcn_readfirstlane(smem_base + buf * BUF + SW_BUF + SW_DPB + f * 1024);
          if (two || f * 1024 + lane * 16 < 56 * 64)
            glds16(p.argmax + ((size_t)img * 56 + rp) * (56 * 64) + f * 1024 + lane * 16, dst);
        }
      }
    }
  };
  // d(loss)/d(conv1 output) of the tile's 2 x 112 positions, in place over the staged conv1 output (same swizzle)
  auto grad_pass = [&](int tile, int buf) {
    if (tid >= 448) return;
    const int rp = tile % 56;
    const int b = tid >> 3, cg = tid & 7;
    unsigned char* tb = smem + buf * BUF + SW_XB;
    const unsigned char* dp = smem + buf * BUF + SW_BUF;
    const unsigned char* am = dp + SW_DPB;
    const bool a1 = rp + 1 < 56, b1 = b + 1 < 56;
    // tile row r = 112 * (conv row parity) + column; the 16 bytes of channel group cg sit in slot ((cg >> 1) ^ key(r)) * 2 + (cg & 1)
    auto at = [&](int r) -> unsigned char* { return tb + r * 128 + (((((cg >> 1) ^ ((r >> 1) & 3)) << 1) | (cg & 1)) << 4); };
    unsigned char* pos[4] = {at(2 * b), at(2 * b + 1), at(112 + 2 * b), at(112 + 2 * b + 1)};
    auto ld4 = [](const unsigned char* q, float (&f)[4]) {   // four bf16 -> f32
      const uint2 u = *reinterpret_cast<const uint2*>(q);
      f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
      f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    };
    // two halves of four channels each (the whole group at once needs more registers than the MFMA phase leaves)
#pragma unroll 1
    for (int hf = 0; hf < 2; ++hf) {
      const int c0 = cg * 8 + hf * 4;
      // (LDS copy of the BatchNorm constants, folded per channel at kernel start:  a (g - b - (y - mean) invstd c)
      //  = k1 g - k3 y + k4  with k1 = a, k3 = a invstd c, k4 = k3 mean - a b: two FMAs per element instead of six operations;
      //  the pass is bound by its vector instructions -- 2.6 us per tile measured, twice the MFMA phase)
      const float* cf = reinterpret_cast<const float*>(smem + 2 * BUF) + c0;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(cf), sh = *reinterpret_cast<const f32x4*>(cf + 64);
      const f32x4 k1 = *reinterpret_cast<const f32x4*>(cf + 128), k3 = *reinterpret_cast<const f32x4*>(cf + 192);
      const f32x4 k4 = *reinterpret_cast<const f32x4*>(cf + 256);
      unsigned k[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};   // argmax bytes of cells (a,b) (a,b+1) (a+1,b) (a+1,b+1)
      float d[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) d[c][e] = 0.f;
      const int cell[4] = {b, b + 1, 56 + b, 56 + b + 1};
      const bool have[4] = {true, b1, a1, a1 && b1};
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (have[c]) {
          k[c] = *reinterpret_cast<const unsigned*>(am + cell[c] * 64 + cg * 8 + hf * 4);
          ld4(dp + cell[c] * 128 + cg * 16 + hf * 8, d[c]);
        }
      float y[4][4], o[4][4];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) ld4(pos[q4] + hf * 8, y[q4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int sft = e * 8;
        const int i00 = (k[0] >> sft) & 0xff, i01 = (k[1] >> sft) & 0xff, i10 = (k[2] >> sft) & 0xff, i11 = (k[3] >> sft) & 0xff;
        // taps kh*3+kw with kh = h - (2*ph - 1), kw = w - (2*pw - 1)  (stem_bn_bwd_apply2x2_kernel: same sums, same order)
        float g[4];
        g[0] = i00 == 4 ? d[0][e] : 0.f;
        g[1] = (i00 == 5 ? d[0][e] : 0.f) + (i01 == 3 ? d[1][e] : 0.f);
        g[2] = (i00 == 7 ? d[0][e] : 0.f) + (i10 == 1 ? d[2][e] : 0.f);
        g[3] = (i00 == 8 ? d[0][e] : 0.f);
        g[3] += (i01 == 6 ? d[1][e] : 0.f);
        g[3] += (i10 == 2 ? d[2][e] : 0.f);
        g[3] += (i11 == 0 ? d[3][e] : 0.f);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float gm = (y[q4][e] * sc[e] + sh[e] > 0.f) ? g[q4] : 0.f;
          o[q4][e] = __builtin_fmaf(k1[e], gm, __builtin_fmaf(-k3[e], y[q4][e], k4[e]));
        }
      }
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const bf16x4 v = {(bf16_t)o[q4][0], (bf16_t)o[q4][1], (bf16_t)o[q4][2], (bf16_t)o[q4][3]};
        *reinterpret_cast<bf16x4*>(pos[q4] + hf * 8) = v;
      }
    }
  };

  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[i][h] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if constexpr (FUSED) {
    float* cf = reinterpret_cast<float*>(smem + 2 * BUF);
    if (tid < 64) {
      const int c = tid;
      const float a = p.coef[c], b = p.coef[64 + c], cc = p.coef[128 + c];
      const float k3 = a * p.invstd[c] * cc;
      cf[c] = p.scale[c];
      cf[64 + c] = p.shift[c];
      cf[128 + c] = a;
      cf[192 + c] = k3;
      cf[256 + c] = k3 * p.mean[c] - a * b;
    }
    // (visible to every thread behind the first tile's barrier)
  }
  dma_tile(t_beg, 0);
  for (int t = t_beg; t < t_end; ++t) {
    const int buf = (t - t_beg) & 1;
    // tile t has landed (every wave's share), and nobody reads the other buffer any more
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + 1 < t_end) dma_tile(t + 1, buf ^ 1);
    if constexpr (FUSED) {
      grad_pass(t, buf);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the gradient tile is complete
    }
    if (wave < 7) {
      const unsigned char* xb = smem + buf * BUF;
      const unsigned char* db = xb + SW_XB;
      const int kh = wave;
      // fragments of K block kb: the lane's two pixel rows are r1 = 32 kb + 4 lg + q and r1 + 16
      auto load_frags = [&](int kb, uint4 (&fa)[4], uint4 (&fb)[2]) {
        const int r1 = kb * 32 + 4 * lg + q, r2 = r1 + 16;
        const int key = (r1 >> 1) & 3;   // (== the key of r2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS_AS s16x4*)(db + r1 * 128 + ((i ^ key) << 5) + pp * 8));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS_AS s16x4*)(db + r2 * 128 + ((i ^ key) << 5) + pp * 8));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          fa[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
        }
        const int o1 = r1 >= 112, o2 = r2 >= 112;
        const unsigned char* x1 = xb + (2 * o1 + kh) * SW_ROWB + (2 * (r1 - 112 * o1) + pp) * 8;
        const unsigned char* x2 = xb + (2 * o2 + kh) * SW_ROWB + (2 * (r2 - 112 * o2) + pp) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS_AS s16x4*)(x1 + h * 32));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS_AS s16x4*)(x2 + h * 32));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          fb[h] = make_uint4(l2.x, l2.y, h2.x, h2.y);
        }
      };
      // (the next block's twelve transposing reads are in flight under the eight MFMAs of the current one)
      uint4 fa[2][4], fb[2][2];
      load_frags(0, fa[0], fb[0]);
#pragma unroll
      for (int kb = 0; kb < 7; ++kb) {
        if (kb + 1 < 7) load_frags(kb + 1, fa[(kb + 1) & 1], fb[(kb + 1) & 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            acc[i][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[kb & 1][i]),
                                                                __builtin_bit_cast(bf16x8, fb[kb & 1][h]), acc[i][h], 0, 0, 0);
      }
    }
  }
  // lane holds rows co = 16 i + 4 lg + r, column n = 16 h + li of its window row
  if (wave < 7) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = i * 16 + lg * 4 + r, n = h * 16 + li;
          atomicAdd(p.dw + (co * 7 + wave) * 32 + n, acc[i][h][r]);
        }
  }
}

int launch_stem_rows(const void* dy, const void* x, float* dw, int batch, hipStream_t stream) {
  StemWgArgs a = {};
  a.dy = static_cast<const bf16_t*>(dy); a.x = static_cast<const bf16_t*>(x); a.dw = dw; a.ntiles = batch * 56;
  static std::atomic<unsigned long long> lds_limit_set{0};  // per device
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(stem_wgrad_rows_kernel<false>), SW_LDS, lds_limit_set)) return rc;
  const int grid = a.ntiles < 256 ? a.ntiles : 256;
  hipLaunchKernelGGL(stem_wgrad_rows_kernel<false>, dim3(grid), dim3(512), SW_LDS, stream, a);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

}  // namespace

// conv_wgrad_patch.hip: streaming kernel for 3x3 / stride 1 / pad 1 (bf16)
bool qt_wgrad_patch_eligible(const qt_conv_desc* d);
size_t qt_wgrad_patch_workspace_bytes(const qt_conv_desc* d);
int qt_wgrad_patch_launch(const qt_conv_desc* d, const void* dy, const void* x, float* dw, void* workspace,
                          size_t workspace_bytes, int oihw, void* stream);

// Stem backward in one launch (bf16): max-pool backward + ReLU mask + BatchNorm backward of conv1's output computed tile by
// tile in LDS and contracted with the packed input at once: dw[64][7][32] += conv1's weight gradient (qt_unpack_stem_wgrad
// layout).  Inputs as qt_stem_bn_bwd_apply; d(loss)/d(conv1 output) is never materialised.
extern "C" int qt_stem_bn_bwd_wgrad(int dtype, const void* dpooled, const unsigned char* argmax, const void* y,
                                    const float* scale, const float* shift, const float* mean, const float* invstd,
                                    const float* coef, const void* xpad, float* dw, int batch, void* stream) {
  QT_CHECK_ARG(dpooled && argmax && y && scale && shift && mean && invstd && coef && xpad && dw && batch > 0,
               "qt_stem_bn_bwd_wgrad: bad argument");
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("QTCNN_STEM_BWD_FUSED");
    on = e ? atoi(e) : 1;
  }
  if (dtype != QT_BF16 || !on) {
    qt_set_error("qt_stem_bn_bwd_wgrad: bf16 only (use qt_stem_bn_bwd_apply + qt_conv2d_wgrad)");
    return QT_ERR_UNSUPPORTED;
  }
  StemWgArgs a;
  a.dy = static_cast<const bf16_t*>(y); a.x = static_cast<const bf16_t*>(xpad); a.dw = dw; a.ntiles = batch * 56;
  a.dpooled = static_cast<const bf16_t*>(dpooled); a.argmax = argmax;
  a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.coef = coef;
  static std::atomic<unsigned long long> lds_limit_set{0};  // per device
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(stem_wgrad_rows_kernel<true>), SW_LDS_F, lds_limit_set)) return rc;
  const int grid = a.ntiles < 256 ? a.ntiles : 256;
  hipLaunchKernelGGL(stem_wgrad_rows_kernel<true>, dim3(grid), dim3(512), SW_LDS_F, static_cast<hipStream_t>(stream), a);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

// conv_wgrad_s2.hip: the stride-2 convolutions of a transition block (3x3 / 2 pad 1, 1x1 / 2) on parity planes (bf16)
bool qt_wgrad_s2_eligible(const qt_conv_desc* d);
size_t qt_wgrad_s2_workspace_bytes(const qt_conv_desc* d);
int qt_wgrad_s2_launch(const qt_conv_desc* d, const void* dy, const void* x, float* grad_oihw, void* workspace,
                       size_t workspace_bytes, void* stream);

extern "C" size_t qt_conv2d_wgrad_workspace_bytes(const qt_conv_desc* d) {
  if (!d) return 0;
  if (d->mode == QT_CONV_FWD && qt_wgrad_s2_eligible(d)) return qt_wgrad_s2_workspace_bytes(d);
  return qt_wgrad_patch_workspace_bytes(d);
}

extern "C" int qt_conv2d_wgrad(const qt_conv_desc* d, const void* dy, const void* x, float* dw, void* stream) {
  return qt_conv2d_wgrad_ws(d, dy, x, dw, nullptr, 0, stream);
}

extern "C" int qt_conv2d_wgrad_oihw(const qt_conv_desc* d, const void* dy, const void* x, float* grad_oihw, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  QT_CHECK_ARG(d && dy && x && grad_oihw && workspace, "qt_conv2d_wgrad_oihw: null argument");
  QT_CHECK_ARG(((uintptr_t)dy % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)grad_oihw % 4) == 0 &&
                   ((uintptr_t)workspace % 16) == 0,
               "qt_conv2d_wgrad_oihw: misaligned pointer");
  if (d->mode == QT_CONV_FWD && qt_wgrad_s2_eligible(d))
    return qt_wgrad_s2_launch(d, dy, x, grad_oihw, workspace, workspace_bytes, stream);
  if (d->mode != QT_CONV_FWD || !qt_wgrad_patch_eligible(d)) {
    qt_set_error("qt_conv2d_wgrad_oihw: only the shapes of the streaming kernels (bf16: 3x3 stride 1; 3x3 / 1x1 stride 2)");
    return QT_ERR_UNSUPPORTED;
  }
  return qt_wgrad_patch_launch(d, dy, x, grad_oihw, workspace, workspace_bytes, 1, stream);
}

static thread_local int g_wgrad_overwrite = 0;   // set around the one call of qt_linear_wgrad below

// conv_wgrad_patch.hip: stream of the next partial-filter sum of this thread (NULL: the kernel's own)
void qt_wgrad_set_sum_stream(void* s);

// qt_conv2d_wgrad_oihw with the fixed-order sum of the partial filters on `sum_stream` (ordered behind the kernel by an
// event): the kernel's stream goes straight on to its next launch.  The caller keeps `workspace` untouched until the sum has
// run (e.g. two workspaces used in turn, the reuse ordered behind an event on sum_stream) -- csrc/plan.hip does.
extern "C" int qt_conv2d_wgrad_oihw_on(const qt_conv_desc* d, const void* dy, const void* x, float* grad_oihw, void* workspace,
                                       size_t workspace_bytes, void* stream, void* sum_stream) {
  qt_wgrad_set_sum_stream(sum_stream);
  const int st = qt_conv2d_wgrad_oihw(d, dy, x, grad_oihw, workspace, workspace_bytes, stream);
  qt_wgrad_set_sum_stream(nullptr);
  return st;
}

extern "C" int qt_conv2d_wgrad_ws(const qt_conv_desc* d, const void* dy, const void* x, float* dw, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  QT_CHECK_ARG(d && dy && x && dw, "qt_conv2d_wgrad: null argument");
  QT_CHECK_ARG(((uintptr_t)workspace % 16) == 0, "qt_conv2d_wgrad_ws: misaligned workspace");
  QT_CHECK_ARG(d->dtype == QT_F32 || d->dtype == QT_BF16, "qt_conv2d_wgrad: bad dtype %d", d->dtype);
  QT_CHECK_ARG(d->mode == QT_CONV_FWD, "qt_conv2d_wgrad: describe the FORWARD convolution (mode QT_CONV_FWD)");
  QT_CHECK_ARG(d->n_out > 0 && d->n_out % 8 == 0 && d->k_per_tap > 0 && d->k_per_tap % 8 == 0,
               "qt_conv2d_wgrad: channel counts must be multiples of 8 (n_out=%d k_per_tap=%d)", d->n_out, d->k_per_tap);
  QT_CHECK_ARG(((uintptr_t)dy % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dw % 4) == 0,
               "qt_conv2d_wgrad: misaligned pointer");
  WgradArgs a;
  a.dy = dy; a.x = x; a.dw = dw;
  a.x_img_stride = d->src_img_stride; a.x_row_stride = d->src_row_stride; a.x_pix_stride = d->src_pix_stride;
  const long long M = (long long)d->batch * qt_quad_regions(d->quad) * d->out_h * d->out_w;
  QT_CHECK_ARG(M > 0 && M < (1ll << 31), "qt_conv2d_wgrad: bad pixel count");
  a.M = (int)M; a.N = d->n_out; a.KC = d->k_per_tap;
  a.OH = d->out_h; a.OW = d->out_w; a.IH = d->in_h; a.IW = d->in_w;
  a.ntaps = d->kh * d->kw; a.KW = d->kw; a.stride = d->stride; a.pad = d->pad; a.quad = qt_quad_split(d->quad);
  a.tap_stride = d->src_row_stride;
  a.div_ohw = make_fastdiv((unsigned)(d->out_h * d->out_w));
  a.div_ow = make_fastdiv((unsigned)d->out_w);
  a.tilesN = a.tilesC = a.gtaps = a.ksplit = a.pix_per_split = 0;
  a.overwrite = g_wgrad_overwrite;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (qt_wgrad_patch_eligible(d)) return qt_wgrad_patch_launch(d, dy, x, dw, workspace, workspace_bytes, 0, stream);
  // packed stem: 7 row taps x 32 elements form one 224-wide virtual channel axis
  const bool stem = d->k_per_tap == 32 && d->kw == 1 && d->kh == 7 && d->n_out == 64;
  if (d->dtype == QT_BF16) {
    if (stem) {
      // QTCNN_STEM_WGRAD_ROWS (default 1): the raw-row kernel for the canonical stem geometry; 0: the generic kernel
      static int rows_on = -1;
      if (rows_on < 0) {
        const char* e = getenv("QTCNN_STEM_WGRAD_ROWS");
        rows_on = e ? atoi(e) : 1;
      }
      if (rows_on && d->kh == 7 && d->stride == 2 && d->pad == 0 && d->out_h == 112 && d->out_w == 112 &&
          d->in_h == QT_STEM_PAD_H && d->in_w == QT_STEM_PAD_W && d->src_pix_stride == 4 &&
          d->src_row_stride == QT_STEM_PAD_W * 4 && d->src_img_stride == (long long)QT_STEM_PAD_H * QT_STEM_PAD_W * 4 && !d->quad)
        return launch_stem_rows(dy, x, dw, d->batch, s);
      return launch<bf16_t, 64, 224, true>(a, s);
    }
    const bool n64 = d->n_out <= 64, c64 = d->k_per_tap <= 64;
    if (n64 && c64) return launch<bf16_t, 64, 64, false>(a, s);
    if (n64) return launch<bf16_t, 64, 128, false>(a, s);
    if (c64) return launch<bf16_t, 128, 64, false>(a, s);
    return launch<bf16_t, 128, 128, false>(a, s);
  }
  if (stem) return launch<float, 64, 224, true>(a, s);
  return launch<float, 64, 64, false>(a, s);
}

// dw [out][in] f32 = dy^T x for a Linear layer, WRITTEN (not accumulated): the generic kernel on 1x1 images; when the grid has
// a single range of rows per tile -- classifier.0 of the reference (Linear 5376 -> 2688 at 256 rows: 882 tiles;
// /root/reference/Quadtree_from scratch/models.py:264-271) -- every element has one producer and is stored plainly: no
// 58 MB zero fill, no float atomics; otherwise zero + accumulate as qt_conv2d_wgrad.
extern "C" int qt_linear_wgrad(int dtype, const void* dy, const void* x, float* dw, int rows, int out, int in, void* stream) {
  QT_CHECK_ARG(dy && x && dw && rows > 0 && out > 0 && in > 0, "qt_linear_wgrad: bad argument");
  qt_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.dtype = dtype; d.mode = QT_CONV_FWD; d.batch = rows; d.in_h = d.in_w = d.out_h = d.out_w = 1;
  d.kh = d.kw = 1; d.stride = 1; d.pad = 0;
  d.k_per_tap = in; d.n_out = out; d.src_pix_stride = in; d.src_row_stride = in; d.src_img_stride = in;
  g_wgrad_overwrite = 1;
  const int st = qt_conv2d_wgrad_ws(&d, dy, x, dw, nullptr, 0, stream);
  g_wgrad_overwrite = 0;
  return st;
}
