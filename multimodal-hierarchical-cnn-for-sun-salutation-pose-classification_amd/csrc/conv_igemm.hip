// Implicit-GEMM convolution on MFMA for gfx950 (forward gather and data-gradient
// gather), NHWC activations, [N][taps][K] weights, f32 accumulate.
//
// Replaces the ATen conv2d / conv2d-backward-input calls made by
//   /root/reference/Quadtree_from scratch/models.py:222-243,284-289 (forward) and
//   loss.backward() at Quadtree_from scratch/Quadtree_train.py:65 (dgrad).
//
// GEMM view:  D[n][m] = sum_{tap,k} W[n][tap][k] * X[pixel(m) shifted by tap][k]
//   m = destination pixel (image, oh, ow) flattened, n = destination channel.
// Workgroup = WM x WN waves, tile BM pixels x BN channels (256x128 / 8 waves or 128x128 / 4),
// K-step = 128 bytes of K per row (64 bf16 / 32 f32).  Both operand tiles are
// staged global -> registers -> LDS (zero fill of the halo happens in the
// register stage), double buffered, one barrier per K-step; the LDS image is
// XOR-swizzled per 16-byte chunk so that the ds_read_b128 fragment reads are
// conflict free.  The MFMA is issued with the WEIGHT tile as the A operand, so
// each lane ends up with 4 consecutive channels of one pixel: the epilogue
// stages the f32 tile through LDS and writes whole NHWC rows (16 B per lane),
// fusing scale/shift (+residual) (+ReLU) (+ReLU-mask) and BatchNorm partial sums.
#include <stdlib.h>

#include "conv_args.h"

namespace {

using qtc::ConvArgs;

constexpr int kRowBytes = 128;  // bytes of K per row per K-step
// s_setprio 1 around a K-step's MFMA block: with two workgroups per CU the wave that has its fragments goes
// first (alone +3 % on layer3 / layer4; inside the step within noise, eval forward +0.4 %)
constexpr bool kPrio = true;

// NSTAGE == 1: single-buffered tiles (every fragment of a K-step is read into registers, a second
// barrier, then the next K-step's DMA is issued under the MFMAs) + EH == 2: the f32 epilogue staging
// holds half of the rows at a time -> 32 KB of LDS, three workgroups per CU.
template <typename T, int BM, int BN, int WM, int WN, int NSTAGE, bool DGRAD, int EH = 1>
__global__ __launch_bounds__(64 * WM * WN, NSTAGE == 1 ? 3 : 1) void conv_igemm_kernel(ConvArgs p) {
  static_assert(EH == 1 || (EH == 2 && WM == 2), "the split epilogue takes one wave row per pass");
  constexpr int NT = 64 * WM * WN;   // threads
  constexpr int RG = NT / 8;         // rows staged per pass of the whole workgroup
  constexpr int BK = kRowBytes / (int)sizeof(T);
  constexpr int RA = BM / RG;        // pixel rows staged per thread
  constexpr int RW = BN / RG;        // weight rows staged per thread
  constexpr int TM = BM / WM / 16;   // 16-wide pixel tiles per wave
  constexpr int TN = BN / WN / 16;   // 16-wide channel tiles per wave
  constexpr int STAGE_BYTES = (BM + BN) * kRowBytes;
  static_assert(RA >= 1 && RW >= 1 && BM % RG == 0 && BN % RG == 0, "tile / thread-count mismatch");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const T* __restrict__ src = static_cast<const T*>(p.src);
  const T* __restrict__ wgt = static_cast<const T*>(p.wgt);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;

  // phase stamps exist only in an experiment build (-DQT_KERNEL_PROF, scripts/prof_build.sh): the production kernel carries
  // no profiling pointer it could write through
#ifdef QT_KERNEL_PROF
  auto stamp = [&](int i) {
    if (p.prof && tid == 0) p.prof[(long long)blockIdx.x * 4 + i] = wall_clock64();
  };
#else
  auto stamp = [](int) {};
#endif
  stamp(0);   // entry
  const int bid = qt_xcd_remap(blockIdx.x, p.gridM * p.gridN);
  const int mt = bid / p.gridN, nt = bid - mt * p.gridN;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread staging rows -------------------------------------------------
  // Staging is LDS-DMA (global_load_lds_dwordx4): one wave instruction fills 8 rows x
  // 128 B, lane-linear in LDS.  The XOR chunk swizzle therefore sits on the SOURCE side:
  // LDS slot (lane&7) of row r receives source chunk (lane&7)^(r&7).
  const int rbase = tid >> 3;  // row inside an RG-row group
  const int chunk = (tid & 7) ^ (rbase & 7);  // source 16-byte chunk of this lane
  // Per staged pixel row: a base pointer (tap (0,0), this lane's chunk) and one validity
  // bit per tap, both computed once; a K-step then costs one shift/test and one 64-bit
  // add per row.  (Stride-2 data gradients keep the general per-step gather.)
  long long a_base[RA];
  int a_oh[RA], a_ow[RA];
  const T* a_ptr[RA];
  unsigned a_mask[RA];
  const int OHW = p.OH * p.OW;
  const bool slow = DGRAD && NSTAGE != 1 && p.stride == 2;  // (the single-buffered shape is never dispatched for it)
  auto axis_ok = [&](int o, int k, int extent) -> bool {
    if (!DGRAD) return (unsigned)(o * p.stride - p.pad + k) < (unsigned)extent;
    const int t = o + p.pad - k;
    return t >= 0 && !(t & (p.stride - 1)) && (t >> (p.stride - 1)) < extent;  // stride is 1 or 2
  };
  auto tap_src = [&](int oh, int ow, int kh, int kw, int& ih, int& iw) -> bool {
    bool ok;
    if (!DGRAD) {
      ih = oh * p.stride - p.pad + kh;
      iw = ow * p.stride - p.pad + kw;
      ok = (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
    } else {
      const int th = oh + p.pad - kh, tw = ow + p.pad - kw;
      ok = th >= 0 && tw >= 0;
      if (p.stride == 2) {
        ok = ok && !((th | tw) & 1);
        ih = th >> 1;
        iw = tw >> 1;
      } else {
        ih = th;
        iw = tw;
      }
      ok = ok && ih < p.IH && iw < p.IW;
    }
    return ok;
  };
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + rbase + RG * i;
    a_base[i] = 0;
    a_oh[i] = a_ow[i] = 0;
    a_mask[i] = 0;
    a_ptr[i] = src;
    if (m < p.M) {
      int img = (int)fdiv((unsigned)m, p.div_ohw);   // (multiply-shift: two runtime divisions per staged row were ~2 us of every workgroup)
      int rem = m - img * OHW;
      int oh = (int)fdiv((unsigned)rem, p.div_ow);
      int ow = rem - oh * p.OW;
      long long base;
      if (p.quad) {
        if (!DGRAD) {
          // destination = quadrant-local 7x7 pixel of quadrant q of image n;
          // source = the un-split map, offset to the quadrant's corner.
          // (p.quad = S: the map is split into S x S regions, numbered row-major)
          const int S = p.quad, R = S * S;
          const int n = img / R, q = img - n * R;
          const int qr = q / S, qc = q - qr * S;
          base = (long long)n * p.src_img_stride +
                 (long long)qr * p.IH * p.src_row_stride +
                 (long long)qc * p.IW * p.src_pix_stride;
        } else {
          // destination = pixel of the un-split (S*IH x S*IW) map; source = the
          // dense per-region gradient image of the region that owns it.
          const int S = p.quad;
          int qh = 0, qw = 0;
          while (oh >= p.IH) { oh -= p.IH; ++qh; }
          while (ow >= p.IW) { ow -= p.IW; ++qw; }
          base = (long long)((img * S + qh) * S + qw) * p.src_img_stride;
        }
      } else {
        base = (long long)img * p.src_img_stride;
      }
      a_base[i] = base;
      a_oh[i] = oh;
      a_ow[i] = ow;
      // validity is separable: tap (kh,kw) is valid iff row tap kh and column tap kw are
      unsigned vw = 0, mask = 0;
      const int KH = p.ntaps / (p.KW * (p.KT > 1 ? p.KT : 1));
      for (int kw = 0; kw < p.KW; ++kw)
        if (axis_ok(ow, kw, p.IW)) vw |= 1u << kw;
      for (int kh = 0; kh < KH; ++kh)
        if (axis_ok(oh, kh, p.IH)) mask |= vw << (kh * p.KW);
      long long f0 = 0;
      if (p.KT > 1) {   // frame taps: the spatial mask repeats once per frame tap whose source frame exists
        const int t = img / p.imgs_per_frame, pt = p.KT >> 1;
        const int taps2d = KH * p.KW;
        unsigned m3 = 0;
        for (int kt = 0; kt < p.KT; ++kt) {
          const int ts = DGRAD ? t + pt - kt : t - pt + kt;
          if (ts >= 0 && ts < p.frames) m3 |= mask << (kt * taps2d);
        }
        mask = m3;
        f0 = (DGRAD ? pt : -pt) * p.frame_stride;
      }
      if (p.dst_merge && p.ntaps == 5) mask |= (mask & 1u) << 4;   // the fifth slot reads the window's first pixel (of the second map)
      a_mask[i] = mask;
      const long long h0 = DGRAD ? (oh + p.pad) : (oh * p.stride - p.pad);
      const long long w0 = DGRAD ? (ow + p.pad) : (ow * p.stride - p.pad);
      a_ptr[i] = src + base + f0 + h0 * p.src_row_stride + w0 * p.src_pix_stride + chunk * (16 / (int)sizeof(T));
    }
  }
  const int ktot = p.ntaps * p.KC;
  const T* w_ptr[RW];
  bool w_ok[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int n = n0 + rbase + RG * i;
    w_ok[i] = n < p.N;
    w_ptr[i] = wgt + (long long)(w_ok[i] ? n : 0) * ktot + chunk * (16 / (int)sizeof(T));
  }

  // Merged parity classes (dst_merge): a class uses 1, 2, 2 or 4 of the four tap slots and the others hold zero weights:
  // the K loop of this channel tile visits only the slots one of its classes uses (tmask; all slots otherwise).
  unsigned tmask = 0xffffffffu;
  if (p.dst_merge) {
    const int c_lo = n0 / p.dst_merge, c_hi = (min(p.N, n0 + BN) - 1) / p.dst_merge;
    tmask = 0u;
    for (int c = c_lo; c <= c_hi; ++c) tmask |= ((c >> 1) ? 0x5u : 0x1u) * ((c & 1) ? 0x3u : 0x1u);
    if (p.ntaps == 5 && c_lo == 0) tmask |= 0x10u;   // (the downsample's slot: class (0,0) only)
  }
  const int nk = __builtin_popcount(tmask & (p.ntaps >= 32 ? ~0u : (1u << p.ntaps) - 1u)) * p.KC / BK;

  // A K-step is 128 bytes of K per row.  Normally that is a slice of one tap
  // (tap uniform over the workgroup); when a tap is only 64 bytes (the packed
  // bf16 stem) a K-step spans two taps and the tap depends on the chunk.
  const int sub = p.KC * (int)sizeof(T) < kRowBytes;
  const T* zero_src = reinterpret_cast<const T*>(qt_zero_page);
  const unsigned smem_base = lds_addr_of(smem);
  // uniform (tap, channel offset) of the NEXT stage to issue; stages are issued in order
  int nx_kh = 0, nx_kw = 0, nx_kt = 0, nx_tap = 0, nx_c0 = 0;
  const int KH2 = p.KT > 1 ? p.ntaps / (p.KT * p.KW) : 1 << 30;   // rows of a 2-D filter slice (frame taps only)
  auto next_tap = [&]() {  // the next tap slot this channel tile uses
    do {
      ++nx_tap;
      if (++nx_kw == p.KW) {
        nx_kw = 0;
        if (++nx_kh == KH2) {
          nx_kh = 0;
          ++nx_kt;
        }
      }
    } while (nx_tap < p.ntaps && !((tmask >> nx_tap) & 1u));
  };
  if (!(tmask & 1u)) next_tap();
  auto dma_stage = [&](int ks, int buf) {
    const long long woff = sub ? (long long)ks * BK : (long long)nx_tap * p.KC + nx_c0;  // K offset of this stage's weights

    const unsigned sa = smem_base + buf * STAGE_BYTES + wave * (8 * kRowBytes);  // LDS byte addresses
    const unsigned sw = sa + BM * kRowBytes;
    if (slow) {
      // stride-2 data gradient: general gather
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        int ih, iw;
        const bool ok = ((a_mask[i] >> nx_tap) & 1u) && tap_src(a_oh[i], a_ow[i], nx_kh, nx_kw, ih, iw);
        const T* g = ok ? src + a_base[i] + (long long)ih * p.src_row_stride + (long long)iw * p.src_pix_stride +
                              nx_c0 + chunk * (16 / (int)sizeof(T))
                        : zero_src;
        glds16(g, sa + i * (RG * kRowBytes));
      }
    } else {
      int tap;
      long long toff;
      if (sub) {  // two 64-byte taps per K-step (KW == 1): the tap depends on the lane's chunk
        tap = ks * 2 + (chunk >> 2);
        toff = (long long)tap * p.src_row_stride - (chunk >> 2) * (BK / 2);
      } else {
        tap = nx_tap;
        toff = (long long)nx_kh * p.src_row_stride + (long long)nx_kw * p.src_pix_stride;
        if (p.KT > 1) toff += (long long)nx_kt * p.frame_stride;
        if (p.dst_merge && nx_tap == 4) toff = p.extra_off;   // (uniform: the fifth slot of a merged launch, same pixel as slot 0)
        toff = (DGRAD ? -toff : toff) + nx_c0;
      }
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const T* g = ((a_mask[i] >> tap) & 1u) ? a_ptr[i] + toff : zero_src;
        glds16(g, sa + i * (RG * kRowBytes));
      }
    }
    nx_c0 += BK;
    if (nx_c0 >= p.KC) {
      nx_c0 = 0;
      next_tap();
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const T* g = w_ok[i] ? w_ptr[i] + woff : zero_src;
      glds16(g, sw + i * (RG * kRowBytes));
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fk = lane >> 4;

  // NSTAGE-deep LDS ring.  Stage s lives in buffer s % NSTAGE; NSTAGE-1 stages are in
  // flight.  Per K-step: wait until this wave's DMAs of stage ks have landed (counted
  // vmcnt leaves the younger stages in flight), barrier (=> every wave's part of stage ks
  // is visible and nobody still reads the buffer about to be refilled), issue stage
  // ks+NSTAGE-1, then the MFMAs of stage ks.
  constexpr int DPS = RA + RW;  // DMA instructions per wave per stage
  stamp(1);   // addressing done
  if constexpr (NSTAGE == 1) {
    if (nk > 0) dma_stage(0, 0);
    for (int ks = 0; ks < nk; ++ks) {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // this K-step's tiles have landed
      const unsigned char* sa = smem;
      const unsigned char* sw = sa + BM * kRowBytes;
      uint4 fw[2][TN], fa[2][TM];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const int r = wn * (BN / WN) + i * 16 + frow;
          fw[kk][i] = *reinterpret_cast<const uint4*>(sw + r * kRowBytes + (((kk * 4 + fk) ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int r = wm * (BM / WM) + j * 16 + frow;
          fa[kk][j] = *reinterpret_cast<const uint4*>(sa + r * kRowBytes + (((kk * 4 + fk) ^ (r & 7)) << 4));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave holds its fragments: the tiles are free
      if (ks + 1 < nk) dma_stage(ks + 1, 0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) QtMma<T>::run(acc[i][j], fw[kk][i], fa[kk][j]);
    }
  } else {
#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
      if (s0 < nk) dma_stage(s0, s0);
    for (int ks = 0; ks < nk; ++ks) {
      const int buf = ks % NSTAGE;
      if (NSTAGE >= 3 && ks + NSTAGE - 2 < nk) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NSTAGE - 2) * DPS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      if (ks + NSTAGE - 1 < nk) dma_stage(ks + NSTAGE - 1, (ks + NSTAGE - 1) % NSTAGE);
      const unsigned char* sa = smem + buf * STAGE_BYTES;
      const unsigned char* sw = sa + BM * kRowBytes;
  #pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        uint4 fw[TN], fa[TM];
  #pragma unroll
        for (int i = 0; i < TN; ++i) {
          const int r = wn * (BN / WN) + i * 16 + frow;
          fw[i] = *reinterpret_cast<const uint4*>(sw + r * kRowBytes + (((kk * 4 + fk) ^ (r & 7)) << 4));
        }
  #pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int r = wm * (BM / WM) + j * 16 + frow;
          fa[j] = *reinterpret_cast<const uint4*>(sa + r * kRowBytes + (((kk * 4 + fk) ^ (r & 7)) << 4));
        }
        if (kPrio) __builtin_amdgcn_s_setprio(1);
  #pragma unroll
        for (int i = 0; i < TN; ++i)
  #pragma unroll
          for (int j = 0; j < TM; ++j) QtMma<T>::run(acc[i][j], fw[i], fa[j]);
        if (kPrio) __builtin_amdgcn_s_setprio(0);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // all reads done before the epilogue reuses LDS

  stamp(2);   // K loop done
  // ---- epilogue: accumulators -> LDS f32 [BM][BN] (chunk-swizzled) -------------
  // lane holds channels n = 4*(lane>>4)+r of pixel (lane&15) for each 16x16 tile.
  constexpr int ROWB = BN * 4;  // bytes per staged pixel row
  constexpr int EROWS = BM / EH;  // rows staged at a time
  auto stage_acc = [&](int h) {
    if (EH == 2 && wm != h) return;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int pm = (EH == 2 ? 0 : wm * (BM / WM)) + j * 16 + frow;
        const int c16 = (wn * (BN / WN) + i * 16 + fk * 4) >> 2;  // 16-byte chunk index
        *reinterpret_cast<f32x4*>(smem + pm * ROWB + ((c16 ^ (pm & 7)) << 4)) = acc[i][j];
      }
  };
  stage_acc(0);
  __syncthreads();

  constexpr int TPR = BN / 8;          // threads per pixel row (8 channels each)
  constexpr int RPP = NT / TPR;        // rows per pass
  constexpr int NPASS = BM / RPP;
  const int cg = tid % TPR, r0 = tid / TPR;
  const int nbase = n0 + cg * 8;
  const bool n_ok = nbase < p.N;  // N is a multiple of 8
  // merged parity classes (dst_merge): this thread's 8 channels are channels cbase.. of class nbase / C
  const int dN = p.dst_merge ? p.dst_merge : p.N;
  const int mcls = p.dst_merge ? nbase / p.dst_merge : 0;
  const int cbase = p.dst_merge ? nbase - mcls * p.dst_merge : nbase;
  const int d_oh = p.dst_merge ? (mcls >> 1) : p.dst_oh, d_ow = p.dst_merge ? (mcls & 1) : p.dst_ow;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = (p.scale && n_ok) ? p.scale[nbase + e] : 1.f;
    sh[e] = (p.shift && n_ok) ? p.shift[nbase + e] : 0.f;
  }
  float s1[8], s2[8], s3[8];  // forward: sum v, sum v^2;  backward: sum g, sum g*xhat0, sum g*xhat1
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = s3[e] = 0.f;
  const bool bwd_stats = p.bn_y[0] != nullptr;
  float mu0[8], is0[8], mu1[8], is1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu0[e] = (bwd_stats && n_ok) ? p.bn_mean[0][cbase + e] : 0.f;
    is0[e] = (bwd_stats && n_ok) ? p.bn_invstd[0][cbase + e] : 0.f;
    mu1[e] = (p.bn_y[1] && n_ok) ? p.bn_mean[1][cbase + e] : 0.f;
    is1[e] = (p.bn_y[1] && n_ok) ? p.bn_invstd[1][cbase + e] : 0.f;
  }
  T* __restrict__ dst = static_cast<T*>(p.dst);
  const T* __restrict__ res = (p.dst_merge && p.dst_merge_res0 && mcls != 0) ? nullptr : static_cast<const T*>(p.residual);
  const T* __restrict__ msk = static_cast<const T*>(p.relu_mask);
  const unsigned char* __restrict__ mbits = p.relu_mask_bits;
  // Passes run in batches of PB rows: the batch's memory operands (residual, ReLU mask, saved BatchNorm
  // inputs; 16 bytes per thread and row each) are all requested before the first row is finished, so a tile
  // exposes one memory latency per batch instead of one per row (alone, layer2 data gradient with mask +
  // BatchNorm link: 112 us against 84 us without operands, 31 us of which is the operands' HBM time).
  constexpr int PB = sizeof(T) == 2 ? (NSTAGE == 1 ? 1 : 4) : 1;  // (the 168-VGPR single-buffered shape has no room for a batch)
  static_assert(NPASS % PB == 0 && (EH == 1 || (NPASS / 2) % PB == 0), "pass batches");
  if constexpr (PB > 1) {
#pragma unroll
  for (int pb = 0; pb < NPASS; pb += PB) {
    if (EH == 2 && pb == NPASS / 2) {  // second half of the rows replaces the first in the staging
      __syncthreads();
      stage_acc(1);
      __syncthreads();
    }
    long long offs[PB];
    bool oks[PB];
    uint4 raw_res[PB], raw_msk[PB], raw_y0[PB], raw_y1[PB];
    unsigned raw_bits[PB];
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      const int m = m0 + r0 + (pb + u) * RPP;
      oks[u] = m < p.M && n_ok;
      long long drow = m;
      if (p.dst_sub && oks[u]) {
        const unsigned img = fdiv((unsigned)m, p.div_ohw);
        const unsigned rem = (unsigned)m - img * (unsigned)(p.OH * p.OW);
        const unsigned oh = fdiv(rem, p.div_ow);
        const unsigned ow = rem - oh * (unsigned)p.OW;
        drow = ((long long)img * p.dst_h + oh * p.dst_sub + d_oh) * p.dst_w + ow * p.dst_sub + d_ow;
      }
      offs[u] = drow * dN + cbase;
      if constexpr (PB > 1) {
        raw_res[u] = raw_msk[u] = raw_y0[u] = raw_y1[u] = make_uint4(0u, 0u, 0u, 0u);
        if (oks[u]) {
          if (res) raw_res[u] = *reinterpret_cast<const uint4*>(res + offs[u]);
          if (msk) raw_msk[u] = *reinterpret_cast<const uint4*>(msk + offs[u]);
          if (mbits) raw_bits[u] = mbits[offs[u] >> 3];
          if (bwd_stats) raw_y0[u] = *reinterpret_cast<const uint4*>(static_cast<const T*>(p.bn_y[0]) + offs[u]);
          if (p.bn_y[1]) raw_y1[u] = *reinterpret_cast<const uint4*>(static_cast<const T*>(p.bn_y[1]) + offs[u]);
        }
      }
    }
    auto operand = [&](const T* base, const uint4& raw, long long off, float (&f)[8]) {
      if constexpr (PB > 1) {
        f[0] = __uint_as_float(raw.x << 16); f[1] = __uint_as_float(raw.x & 0xffff0000u);
        f[2] = __uint_as_float(raw.y << 16); f[3] = __uint_as_float(raw.y & 0xffff0000u);
        f[4] = __uint_as_float(raw.z << 16); f[5] = __uint_as_float(raw.z & 0xffff0000u);
        f[6] = __uint_as_float(raw.w << 16); f[7] = __uint_as_float(raw.w & 0xffff0000u);
      } else {
        QtVec8<T>::load(base + off, f);
      }
    };
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      const int r = r0 + (pb + u) * RPP;
      const int rl = r % EROWS;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + rl * ROWB + (((2 * cg) ^ (rl & 7)) << 4));
      const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + rl * ROWB + (((2 * cg + 1) ^ (rl & 7)) << 4));
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (!oks[u]) continue;
      const long long off = offs[u];
      if (p.stats_partial != nullptr && !bwd_stats) {  // (uniform: an eval forward keeps no sums)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] += v[e];
          s2[e] += v[e] * v[e];
        }
      }
if (p.scale || p.shift) {  // (uniform)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
      }
      if (res) {
        float rv[8];
        operand(res, raw_res[u], off, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (msk) {
        float mv[8];
        operand(msk, raw_msk[u], off, mv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
      }
      if (mbits) qt_apply_mask_bits(PB > 1 ? raw_bits[u] : (unsigned)mbits[off >> 3], v);
      QtVec8<T>::store(dst + off, v);
      if (bwd_stats) {
        float yv[8];
        operand(static_cast<const T*>(p.bn_y[0]), raw_y0[u], off, yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] += v[e];
          s2[e] += v[e] * (yv[e] - mu0[e]) * is0[e];
        }
        if (p.bn_y[1]) {
          operand(static_cast<const T*>(p.bn_y[1]), raw_y1[u], off, yv);
#pragma unroll
          for (int e = 0; e < 8; ++e) s3[e] += v[e] * (yv[e] - mu1[e]) * is1[e];
        }
      }
    }
  }
  } else {
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int r = r0 + ps * RPP;
    const int m = m0 + r;
    if (EH == 2 && ps == NPASS / 2) {  // second half of the rows replaces the first in the staging
      __syncthreads();
      stage_acc(1);
      __syncthreads();
    }
    const int rl = r % EROWS;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + rl * ROWB + (((2 * cg) ^ (rl & 7)) << 4));
    const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + rl * ROWB + (((2 * cg + 1) ^ (rl & 7)) << 4));
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if (m < p.M && n_ok) {
      long long drow = m;
      if (p.dst_sub) {
        const unsigned img = fdiv((unsigned)m, p.div_ohw);
        const unsigned rem = (unsigned)m - img * (unsigned)(p.OH * p.OW);
        const unsigned oh = fdiv(rem, p.div_ow);
        const unsigned ow = rem - oh * (unsigned)p.OW;
        drow = ((long long)img * p.dst_h + oh * p.dst_sub + d_oh) * p.dst_w + ow * p.dst_sub + d_ow;
      }
      const long long off = drow * dN + cbase;
      if (p.stats_partial != nullptr && !bwd_stats) {  // (uniform: an eval forward keeps no sums)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] += v[e];
          s2[e] += v[e] * v[e];
        }
      }
if (p.scale || p.shift) {  // (uniform)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
      }
      if (res) {
        float rv[8];
        QtVec8<T>::load(res + off, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (msk) {
        float mv[8];
        QtVec8<T>::load(msk + off, mv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
      }
      if (mbits) qt_apply_mask_bits(mbits[off >> 3], v);
      QtVec8<T>::store(dst + off, v);
      if (bwd_stats) {
        float yv[8];
        QtVec8<T>::load(static_cast<const T*>(p.bn_y[0]) + off, yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] += v[e];
          s2[e] += v[e] * (yv[e] - mu0[e]) * is0[e];
        }
        if (p.bn_y[1]) {
          QtVec8<T>::load(static_cast<const T*>(p.bn_y[1]) + off, yv);
#pragma unroll
          for (int e = 0; e < 8; ++e) s3[e] += v[e] * (yv[e] - mu1[e]) * is1[e];
        }
      }
    }
  }
  }

  if (p.stats_partial || bwd_stats) {
    // reduce the per-thread sums over the threads that share a channel group
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [RPP][BN][3]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(r0 * BN + cg * 8 + e) * 3 + 0] = s1[e];
      red[(r0 * BN + cg * 8 + e) * 3 + 1] = s2[e];
      red[(r0 * BN + cg * 8 + e) * 3 + 2] = s3[e];
    }
    __syncthreads();
    if (tid < BN) {
      float a = 0.f, b = 0.f, c = 0.f;
      for (int r = 0; r < RPP; ++r) {
        a += red[(r * BN + tid) * 3 + 0];
        b += red[(r * BN + tid) * 3 + 1];
        c += red[(r * BN + tid) * 3 + 2];
      }
      if (n0 + tid < p.N) {
        float* o0 = bwd_stats ? p.bn_partial[0] : p.stats_partial;
        // partial row of this pixel tile (merged parity classes: one row of C channels per class) and column
        const int n = n0 + tid;
        const int cls = p.dst_merge ? n / p.dst_merge : 0;
        const long long row = p.dst_merge ? (long long)mt * (p.N / p.dst_merge) + cls : mt;
        const int col = n - cls * (p.dst_merge ? p.dst_merge : 0);
        o0[(row * 2 + 0) * dN + col] = a;
        o0[(row * 2 + 1) * dN + col] = b;
        if (bwd_stats && p.bn_y[1]) {
          p.bn_partial[1][(row * 2 + 0) * dN + col] = a;
          p.bn_partial[1][(row * 2 + 1) * dN + col] = c;
        }
      }
    }
  }
  stamp(3);   // epilogue done
}

template <typename T, int BM, int BN, int WM, int WN, int NSTAGE, bool DGRAD, int EH = 1>
int launch(const ConvArgs& a, hipStream_t stream) {
  constexpr int STAGE = (BM + BN) * kRowBytes;
  constexpr int EPI = BM * BN * 4 / EH;
  constexpr int RED = (64 * WM * WN / (BN / 8)) * BN * 3 * 4;  // statistics reduction scratch
  constexpr int LDS0 = (NSTAGE * STAGE > EPI) ? NSTAGE * STAGE : EPI;
  constexpr int LDS = LDS0 > RED ? LDS0 : RED;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  auto kern = conv_igemm_kernel<T, BM, BN, WM, WN, NSTAGE, DGRAD, EH>;
  static std::atomic<unsigned long long> lds_limit_set{0};  // per device
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(kern), LDS, lds_limit_set)) return rc;
  ConvArgs args = a;
  args.gridM = qt_cdiv(a.M, BM);
  args.gridN = qt_cdiv(a.N, BN);
  hipLaunchKernelGGL(kern, dim3(args.gridM * args.gridN), dim3(64 * WM * WN), LDS, stream, args);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

// Tile selection.  Large pixel counts use 256-pixel tiles, 8 waves and a 3-deep ring
// (one workgroup per CU); small ones (layer4, the classifier) keep 128-pixel tiles with
// 4 waves and two workgroups per CU so that the grid still covers the chip.
// Measured (B=256, bf16): the 256-pixel tile only wins for K-loops of >= 36 steps; the
// per-CU L2->LDS rate (~70 GB/s) bounds both shapes, and the short-K layers are dominated
// by the per-tile prologue / epilogue, where two resident workgroups overlap better.
// Also measured for N = 64: a 256x64 tile with four 64x64 wave tiles (fewer LDS reads per MFMA,
// two workgroups per CU) runs 130 us against 117 us for the 128x64 tile (three per CU).
// More shapes measured alone on the 3x3 layers of layer2 / layer3 / layer4 (B=256, bf16; this kernel: 91 / 87 / 78 us):
//   256x128 tile, four 128x64 wave tiles, 3 stages (fewest LDS reads per MFMA, one workgroup per CU): 130 / 105 / 94 us;
//   128x128 with 3 or 4 stages (one workgroup per CU, deeper prefetch): 142 / 133 / 117 us, no better with 4 than with 3;
//   128x64 tiles for N >= 128 (three workgroups per CU, 64x32 wave tiles): 98 / 102 / 108 us;
//   64-byte K-steps (half-deep; 3 stages + half-tile epilogue = 48 KB, three workgroups per CU): 102 / 110 / 95 us;
//   64-byte K-steps, 4 stages, fragments of step ks+1 read into a second register set under the MFMAs of step ks
//   (two workgroups per CU, 216-228 VGPRs): 120 / 114 / 104 us.
// Workgroups per CU on the 128x64 kernel (LDS padded): 1 / 2 / 3 -> 242 / 149 / 121 us.  A second wave per SIMD is worth
// 1.6x, a third 1.2x; neither deeper DMA prefetch nor fewer LDS reads per MFMA pays while a wave's own LDS-read -> MFMA ->
// barrier chain is exposed, and halving the K-step costs more in barriers than the third workgroup returns.
// (Round 1 shipped a single-buffered instantiation -- NSTAGE == 1, half-tile epilogue, three workgroups per CU -- for
// layer2's 3x3 convs; it needed 168 VGPRs and spilled 20.  Those convs now take conv_pt.hip; the one launch that still
// reached it (layer2.0.conv1, stride 2) measures the same on the two-stage tile, so the dispatch no longer uses it.)
// Single-buffered tiles + half-tile epilogue (32 KB of LDS, three workgroups per CU): every fragment of
// a K-step is read into registers, a second barrier frees the tiles, and the next K-step's DMA runs
// under the MFMAs.  Measured alone (B=256, bf16): layer2's 3x3 convs 93 -> 81 us forward, 93 -> 82 us data
// gradient; layer3 / layer4 (784 / 392 workgroups: no better fit on 768 slots than on 512) unchanged; short
// K loops (1x1 downsamples, classifier) slower.  Used only where it wins: 9 taps, N == 128, >= 100 k pixels
// (train step -0.5 %, eval forward -1.0 %).  QTCNN_IGEMM_SINGLE_BUFFER=0 turns it off.
inline int tile_m(long long M, int N, int ksteps) { return (M >= 256 * 256 && ksteps >= 36 && N > 64) ? 256 : 128; }

template <typename T, bool DGRAD>
int dispatch2(const ConvArgs& a, hipStream_t stream) {
  const int bm = tile_m(a.M, a.N, a.ntaps * a.KC * (int)sizeof(T) / kRowBytes);
  if (a.N <= 64) return launch<T, 128, 64, 2, 2, 2, DGRAD>(a, stream);
  if (bm == 256) return launch<T, 256, 128, 4, 2, 3, DGRAD>(a, stream);
  return launch<T, 128, 128, 2, 2, 2, DGRAD>(a, stream);
}

template <typename T>
int dispatch(const qt_conv_desc* d, const ConvArgs& a, hipStream_t stream) {
  return d->mode == QT_CONV_DGRAD ? dispatch2<T, true>(a, stream) : dispatch2<T, false>(a, stream);
}

}  // namespace

// conv_patch.hip: 3x3 stride-1 convs of the 56x56 / 28x28 stages with the input patch held in LDS
bool qt_patch_eligible(const qt_conv_desc* d);
int qt_patch_stats_rows(const qt_conv_desc* d);
int qt_patch_launch(const qt_conv_desc* d, const qt_conv_io* io, void* stream);
// conv_stem.hip: the packed 7x7/2 stem convolution (bf16) with its input rows held in LDS
bool qt_stem_eligible(const qt_conv_desc* d, const qt_conv_io* io);
int qt_stem_stats_rows(const qt_conv_desc* d);
int qt_stem_launch(const qt_conv_desc* d, const qt_conv_io* io, void* stream);

extern "C" int qt_conv2d_stats_rows(const qt_conv_desc* d) {
  if (!d) return QT_ERR_INVALID_ARG;
  const bool kt3 = d->kt > 1;   // (frame taps: always the generic tile)
  if (!kt3 && qt_patch_eligible(d)) return qt_patch_stats_rows(d);
  if (!kt3 && qt_stem_eligible(d, nullptr)) return qt_stem_stats_rows(d);
  const long long M = (long long)d->batch * (d->mode == QT_CONV_FWD ? qt_quad_regions(d->quad) : 1) * d->out_h * d->out_w;
  const int esz = d->dtype == QT_F32 ? 4 : 2;
  {
    ConvArgs a = {};
    a.M = (int)M; a.N = d->n_out; a.KC = d->k_per_tap; a.ntaps = d->kh * d->kw + (d->dst_merge_extra ? 1 : 0); a.KW = d->kw; a.stride = d->stride;
    a.pad = d->pad; a.quad = qt_quad_split(d->quad); a.dst_sub = d->dst_sub;
    a.OH = d->out_h; a.OW = d->out_w; a.IH = d->in_h; a.IW = d->in_w;
    a.src_img_stride = d->src_img_stride; a.src_row_stride = d->src_row_stride; a.src_pix_stride = d->src_pix_stride;
    a.dst_merge = d->dst_merge; a.dst_h = d->dst_h; a.dst_w = d->dst_w;
    if (!kt3 && qt_pt_eligible(a, d->dtype, d->mode == QT_CONV_DGRAD)) return qt_pt_stats_rows(a, d->mode == QT_CONV_DGRAD);
  }
  const int rows = qt_cdiv(M, tile_m(M, d->n_out, ((kt3 ? d->kt : 1) * d->kh * d->kw + (d->dst_merge_extra ? 1 : 0)) * d->k_per_tap * esz / kRowBytes));
  return d->dst_merge > 0 ? rows * (d->n_out / d->dst_merge) : rows;  // merged parity classes: one row per class
}

#ifdef QT_KERNEL_PROF   // experiment build only (scripts/prof_build.sh): [workgroup][4] stamps; the caller owns the buffer
unsigned long long* g_igemm_prof = nullptr;
extern "C" void qt_set_igemm_prof(unsigned long long* buf) { g_igemm_prof = buf; }
#endif

extern "C" int qt_conv2d_igemm(const qt_conv_desc* d, const qt_conv_io* io, void* stream) {
  QT_CHECK_ARG(d && io, "qt_conv2d_igemm: null descriptor");
  QT_CHECK_ARG(d->dtype == QT_F32 || d->dtype == QT_BF16, "qt_conv2d_igemm: bad dtype %d", d->dtype);
  QT_CHECK_ARG(d->mode == QT_CONV_FWD || d->mode == QT_CONV_DGRAD, "qt_conv2d_igemm: bad mode %d", d->mode);
  QT_CHECK_ARG(io->src && io->weight && io->dst, "qt_conv2d_igemm: null src/weight/dst");
  const int bk = d->dtype == QT_F32 ? 32 : 64;
  QT_CHECK_ARG(d->k_per_tap > 0 && (d->k_per_tap % bk == 0 || (2 * d->k_per_tap == bk && (d->kh * d->kw) % 2 == 0)),
               "qt_conv2d_igemm: k_per_tap=%d must be a multiple of %d (or half of it with an even tap count)",
               d->k_per_tap, bk);
  QT_CHECK_ARG(d->n_out > 0 && d->n_out % 8 == 0, "qt_conv2d_igemm: n_out=%d must be a multiple of 8", d->n_out);
  QT_CHECK_ARG(d->batch > 0 && d->out_h > 0 && d->out_w > 0 && d->in_h > 0 && d->in_w > 0,
               "qt_conv2d_igemm: bad geometry");
  QT_CHECK_ARG(d->kh > 0 && d->kw > 0 && (d->stride == 1 || d->stride == 2) && d->pad >= 0,
               "qt_conv2d_igemm: bad filter geometry kh=%d kw=%d stride=%d pad=%d", d->kh, d->kw, d->stride, d->pad);
  QT_CHECK_ARG(!(d->quad && d->stride != 1), "qt_conv2d_igemm: quadrant mode needs stride 1");
  QT_CHECK_ARG(d->quad == 0 || d->quad == 1 || d->quad == 2 || d->quad == 4, "qt_conv2d_igemm: quad must be 0, 1 (= 2), 2 or 4");
  QT_CHECK_ARG(d->kh * d->kw <= 32, "qt_conv2d_igemm: at most 32 taps (kh*kw=%d)", d->kh * d->kw);
  QT_CHECK_ARG(2 * d->k_per_tap != bk || d->kw == 1, "qt_conv2d_igemm: half-K-step taps need kw == 1");
  const int esz = d->dtype == QT_F32 ? 4 : 2;
  QT_CHECK_ARG(((uintptr_t)io->src % 16) == 0 && ((uintptr_t)io->weight % 16) == 0 && ((uintptr_t)io->dst % 16) == 0,
               "qt_conv2d_igemm: pointers must be 16-byte aligned");
  QT_CHECK_ARG((d->src_pix_stride * esz) % 8 == 0 && ((long long)d->src_row_stride * esz) % 16 == 0 &&
                   (d->src_img_stride * esz) % 16 == 0,
               "qt_conv2d_igemm: source strides break 16-byte alignment");
  // 16-byte loads start at pixel*pix_stride + c0 + chunk*16B: pixel stride must keep that aligned
  QT_CHECK_ARG((d->src_pix_stride * esz) % 16 == 0 || d->stride * d->src_pix_stride * esz % 16 == 0,
               "qt_conv2d_igemm: pixel stride %d breaks 16-byte alignment", d->src_pix_stride);

  QT_CHECK_ARG(!(io->relu_mask && io->relu_mask_bits), "qt_conv2d_igemm: relu_mask and relu_mask_bits are exclusive");
  QT_CHECK_ARG(!io->relu_mask_bits || d->n_out % 8 == 0, "qt_conv2d_igemm: relu_mask_bits needs n_out %% 8 == 0");
  const int KT = d->kt > 1 ? d->kt : 1;
  QT_CHECK_ARG(KT == 1 || (KT == 3 && d->frames > 0 && d->batch % d->frames == 0 && d->stride == 1 && !d->quad && !d->dst_sub &&
                           !d->dst_merge && d->kh * d->kw * KT <= 32 && d->k_per_tap % bk == 0),
               "qt_conv2d_igemm: frame taps need kt = 3, frames dividing batch (time-major clips), stride 1, no region / "
               "strided-destination mode, kt*kh*kw <= 32 and whole K-steps per tap");
  if (KT == 1 && qt_stem_eligible(d, io)) return qt_stem_launch(d, io, stream);
  if (KT == 1 && qt_patch_eligible(d)) {
    for (int k = 0; k < 2; ++k)
      QT_CHECK_ARG(!io->bwd_bn[k].y || (io->bwd_bn[k].mean && io->bwd_bn[k].invstd && io->bwd_bn[k].partial &&
                                        !io->stats_partial),
                   "qt_conv2d_igemm: incomplete bwd_bn[%d]", k);
    return qt_patch_launch(d, io, stream);
  }
  ConvArgs a;
  a.src = io->src; a.wgt = io->weight; a.dst = io->dst;
  a.scale = io->scale; a.shift = io->shift;
  a.residual = io->residual; a.relu_mask = io->relu_mask; a.relu_mask_bits = io->relu_mask_bits;
#ifdef QT_KERNEL_PROF
  a.prof = g_igemm_prof;
#endif
  a.stats_partial = io->stats_partial;
  for (int k = 0; k < 2; ++k) {
    a.bn_y[k] = io->bwd_bn[k].y; a.bn_mean[k] = io->bwd_bn[k].mean; a.bn_invstd[k] = io->bwd_bn[k].invstd;
    a.bn_partial[k] = io->bwd_bn[k].partial;
    QT_CHECK_ARG(!a.bn_y[k] || (a.bn_mean[k] && a.bn_invstd[k] && a.bn_partial[k] && !io->stats_partial),
                 "qt_conv2d_igemm: incomplete bwd_bn[%d]", k);
  }
  QT_CHECK_ARG(!(a.bn_y[1] && !a.bn_y[0]), "qt_conv2d_igemm: bwd_bn[1] without bwd_bn[0]");
  a.src_img_stride = d->src_img_stride;
  a.src_row_stride = d->src_row_stride;
  a.src_pix_stride = d->src_pix_stride;
  const int imgs = d->batch * (d->mode == QT_CONV_FWD ? qt_quad_regions(d->quad) : 1);
  const long long M = (long long)imgs * d->out_h * d->out_w;
  QT_CHECK_ARG(M < (1ll << 31) && M * d->n_out < (1ll << 40), "qt_conv2d_igemm: problem too large");
  a.M = (int)M; a.N = d->n_out;
  a.OH = d->out_h; a.OW = d->out_w; a.IH = d->in_h; a.IW = d->in_w;
  a.KC = d->k_per_tap; a.ntaps = d->kh * d->kw * KT; a.KW = d->kw;
  a.KT = KT; a.frames = KT > 1 ? d->frames : 1; a.imgs_per_frame = KT > 1 ? d->batch / d->frames : d->batch;
  a.frame_stride = (long long)a.imgs_per_frame * d->src_img_stride;
  a.stride = d->stride; a.pad = d->pad; a.quad = qt_quad_split(d->quad); a.relu = d->relu;
  a.gridM = a.gridN = 0;
  a.dst_sub = d->dst_sub; a.dst_h = d->dst_h; a.dst_w = d->dst_w; a.dst_oh = d->dst_off_h; a.dst_ow = d->dst_off_w;
  a.dst_merge = d->dst_merge; a.dst_merge_res0 = d->dst_merge_res0;
  a.extra_off = 0;
  if (d->dst_merge_extra) {
    QT_CHECK_ARG(d->dst_merge > 0 && io->extra_src && ((uintptr_t)io->extra_src % 16) == 0,
                 "qt_conv2d_igemm: dst_merge_extra needs dst_merge and a 16-byte aligned extra_src");
    const long long diff = (long long)((intptr_t)io->extra_src - (intptr_t)io->src);
    QT_CHECK_ARG(diff % esz == 0, "qt_conv2d_igemm: extra_src is not element-aligned to src");
    a.extra_off = diff / esz;
    a.ntaps = 5;
  }
  QT_CHECK_ARG(d->dst_merge == 0 || (d->dst_merge > 0 && d->dst_merge % 8 == 0 && d->dst_sub == 2 &&
                                     d->n_out == 4 * d->dst_merge && d->dst_off_h == 0 && d->dst_off_w == 0 &&
                                     !io->scale && !io->shift && d->mode == QT_CONV_FWD &&
                                     d->kh == 2 && d->kw == 2 && d->stride == 1 && d->pad == 0 &&
                                     d->dst_h >= 2 * d->out_h && d->dst_w >= 2 * d->out_w),
               "qt_conv2d_igemm: dst_merge = C needs a 2x2 / stride 1 / pad 0 gather, dst_sub = 2, n_out = 4 C, C %% 8 == 0, "
               "no offsets / affine");
  a.div_ohw = make_fastdiv((unsigned)(d->out_h * d->out_w));
  a.div_ow = make_fastdiv((unsigned)d->out_w);
  QT_CHECK_ARG(d->dst_sub == 0 || (d->dst_sub >= 1 && d->dst_h > 0 && d->dst_w > 0 && !io->stats_partial &&
                                   (d->out_h - 1) * d->dst_sub + d->dst_off_h < d->dst_h &&
                                   (d->out_w - 1) * d->dst_sub + d->dst_off_w < d->dst_w),
               "qt_conv2d_igemm: bad destination mapping");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (KT == 1 && qt_pt_eligible(a, d->dtype, d->mode == QT_CONV_DGRAD)) return qt_pt_launch(a, d->dtype, d->mode == QT_CONV_DGRAD, s);
  return d->dtype == QT_F32 ? dispatch<float>(d, a, s) : dispatch<bf16_t>(d, a, s);
}
