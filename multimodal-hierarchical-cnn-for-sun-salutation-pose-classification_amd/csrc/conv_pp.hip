// Ping-pong implicit-GEMM convolution for the long 3x3 layers (layer2..4 of the ResNet-18 stack, forward
// and stride-1 data gradient): ONE 8-wave workgroup per CU, 256-pixel tiles, a ring of NB K-tiles filled
// by LDS-DMA with NB-1 K-tiles in flight, and the two wave groups of the workgroup half a period apart.
//
// Replaces the same ATen calls as conv_igemm.hip (conv2d forward / backward-input reached from
// /root/reference/Quadtree_from scratch/models.py:222-243,284-289 and Quadtree_train.py:65).
//
// Why a second kernel.  conv_igemm.hip's 4-wave workgroups each do load -> barrier -> read -> MFMA in
// lockstep; two of them per CU hide part of the chain, and the K-tile about to be used was requested only
// one K-step earlier (a DMA round trip under load is ~1 us: the measured time per K-step).  Here
//   * waves 0-3 (group 0) and waves 4-7 (group 1) share the four SIMDs pairwise and run the SAME program
//     one barrier apart: while one group issues its 32 MFMAs of K-tile t, the other issues its LDS-DMA
//     for K-tile t+NB-1 and reads its fragments of K-tile t from LDS -- the matrix pipe of every SIMD
//     always has a wave with operands in registers;
//   * a K-tile is requested NB-1 (2 or 3) periods before it is read, waits are counted (s_waitcnt
//     vmcnt(N) never drains the ring), barriers are raw s_barrier;
//   * tiles are 256 pixels x 128 channels x 128 B of K (85 FLOP per staged byte) or 256 x 256 x 64 B
//     (128 FLOP/B): the L2 -> LDS rate of a CU (~65-70 GB/s) stops bounding the MFMA rate.
//
// Timeline (b = workgroup barrier, L_t = issue DMA of K-tile t+D, read fragments of K-tile t, C_t = MFMAs):
//   group 0:  b0 | L0 | b1 | C0 | b2 | L1 | b3 | C1 | ...
//   group 1:  b0 |    | b1 | L0 | b2 | C0 | b3 | L1 | ...
// Ordering argument (D = NB-1 tiles in flight):
//   RAW  a wave waits for ITS DMA of K-tile t+1 at the end of L_t (before a barrier); group 0 reads K-tile
//        t after b_2t, by which both groups have passed such a wait for it (group 1's L_{t-1} ends at b_2t);
//   WAR  the DMA of K-tile t+D overwrites the buffer of K-tile t-1 and is issued after b_2t; the last
//        reads of K-tile t-1 (group 1's L_{t-1}) were retired by lgkmcnt(0) before b_2t.
#include "conv_args.h"

namespace {

using qtc::ConvArgs;

constexpr int kNT = 512;  // 8 waves: two per SIMD, one of each group

// eight consecutive elements as loaded (decoded to f32 only where they are consumed: a batch of epilogue
// operands in flight costs 4 registers per operand and row in bf16, not 8)
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

template <typename T, int BM, int BN, int KB, int WM, int WN, int NB, bool DGRAD>
__global__ __launch_bounds__(kNT, 2) void conv_pp_kernel(ConvArgs p) {
  static_assert(WM * WN == 8, "eight waves");
  static_assert(KB == 128 || KB == 64, "K-tile rows are 128 or 64 bytes");
  static_assert((BN / WN) % 32 == 0, "a wave owns whole 32-channel groups (epilogue pairs)");
  constexpr int NT = kNT;
  constexpr int CH = KB / 16;               // 16-byte chunks per staged row
  constexpr int RG = NT / CH;               // rows staged per pass of the whole workgroup
  constexpr int BK = KB / (int)sizeof(T);   // K elements per K-tile
  constexpr int KK = KB / 64;               // MFMA sub-steps (64 B of K per row each) per K-tile
  constexpr int RA = BM / RG, RW = BN / RG;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int STAGE = (BM + BN) * KB;
  constexpr int D = NB - 1;                 // K-tiles in flight
  constexpr int P = RA + RW;                // LDS-DMA instructions per wave and K-tile
  static_assert(RA >= 1 && RW >= 1 && BM % RG == 0 && BN % RG == 0, "tile / thread-count mismatch");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const T* __restrict__ src = static_cast<const T*>(p.src);
  const T* __restrict__ wgt = static_cast<const T*>(p.wgt);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;               // waves w and w+4 share a SIMD: one of each group per SIMD
  const int wm = wave % WM, wn = wave / WM;

  const int bid = qt_xcd_remap(blockIdx.x, p.gridM * p.gridN);
  const int mt = bid / p.gridN, nt = bid - mt * p.gridN;
  const int m0 = mt * BM, n0 = nt * BN;

  // chunk swizzle: LDS slot s of row r holds source chunk s ^ swz(r) (the DMA writes lane-linearly, so
  // the permutation sits on the source address; the fragment reads apply the same XOR).  128-byte rows:
  // r & 7 (conflict-free ds_read_b128 over 16 rows x 8 chunks); 64-byte rows: four rows share a 256-byte
  // bank row, (4 - (r >> 2)) & 3 makes the 16-lane read groups hit 16 different 16-byte slots.
  auto swz = [](int r) -> int { return KB == 128 ? (r & 7) : ((4 - ((r >> 2) & 3)) & 3); };

  // ---- per-thread staging rows -------------------------------------------------
  const int rbase = tid / CH;                            // row inside an RG-row group (RG is a multiple of 16)
  const int chunk = (tid % CH) ^ swz(rbase);             // source 16-byte chunk of this lane
  const T* a_ptr[RA];
  unsigned a_mask[RA];
  const int OHW = p.OH * p.OW;
  auto axis_ok = [&](int o, int k, int extent) -> bool {
    if (!DGRAD) return (unsigned)(o * p.stride - p.pad + k) < (unsigned)extent;
    return (unsigned)(o + p.pad - k) < (unsigned)extent;   // stride 1 only
  };
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + rbase + RG * i;
    a_mask[i] = 0;
    a_ptr[i] = src;
    if (m < p.M) {
      int img = m / OHW;
      int rem = m - img * OHW;
      int oh = rem / p.OW;
      int ow = rem - oh * p.OW;
      long long base;
      if (p.quad) {
        if (!DGRAD) {
          const int S = p.quad, R = S * S;
          const int n = img / R, q = img - n * R;
          const int qr = q / S, qc = q - qr * S;
          base = (long long)n * p.src_img_stride + (long long)qr * p.IH * p.src_row_stride +
                 (long long)qc * p.IW * p.src_pix_stride;
        } else {
          const int S = p.quad;
          int qh = 0, qw = 0;
          while (oh >= p.IH) { oh -= p.IH; ++qh; }
          while (ow >= p.IW) { ow -= p.IW; ++qw; }
          base = (long long)((img * S + qh) * S + qw) * p.src_img_stride;
        }
      } else {
        base = (long long)img * p.src_img_stride;
      }
      unsigned vw = 0, mask = 0;
      const int KH = p.ntaps / p.KW;
      for (int kw = 0; kw < p.KW; ++kw)
        if (axis_ok(ow, kw, p.IW)) vw |= 1u << kw;
      for (int kh = 0; kh < KH; ++kh)
        if (axis_ok(oh, kh, p.IH)) mask |= vw << (kh * p.KW);
      a_mask[i] = mask;
      const long long h0 = DGRAD ? (oh + p.pad) : (oh * p.stride - p.pad);
      const long long w0 = DGRAD ? (ow + p.pad) : (ow * p.stride - p.pad);
      a_ptr[i] = src + base + h0 * p.src_row_stride + w0 * p.src_pix_stride + chunk * (16 / (int)sizeof(T));
    }
  }
  const int ktot = p.ntaps * p.KC;
  const T* w_ptr[RW];
  bool w_ok[RW];
  // LDS weight row rho = 32*g + 16*i + x holds output channel 32*g + 8*(x>>2) + 4*i + (x&3): a lane's two 16x16
  // tiles (i = 0, 1) of a 32-channel group then own eight consecutive channels 8*fk .. 8*fk+7 of a pixel
  auto perm32 = [](int rho) -> int {
    const int x = rho & 15, i = (rho >> 4) & 1;
    return (rho & ~31) + 8 * (x >> 2) + 4 * i + (x & 3);
  };
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int n = n0 + perm32(rbase + RG * i);
    w_ok[i] = n < p.N;
    w_ptr[i] = wgt + (long long)(w_ok[i] ? n : 0) * ktot + chunk * (16 / (int)sizeof(T));
  }

  const int nk = ktot / BK;   // >= NB (checked by the launcher)
  const T* zero_src = reinterpret_cast<const T*>(qt_zero_page);
  const unsigned smem_base = lds_addr_of(smem);
  // uniform (tap, channel offset) of the NEXT K-tile to issue; K-tiles are issued in order
  int nx_kh = 0, nx_kw = 0, nx_tap = 0, nx_c0 = 0;
  auto dma_stage = [&](int ks, int buf) {
    const unsigned sa = smem_base + buf * STAGE + wave * 1024;   // a wave instruction fills 1 KiB, lane-linear
    const unsigned sw = sa + BM * KB;
    long long toff = (long long)nx_kh * p.src_row_stride + (long long)nx_kw * p.src_pix_stride;
    toff = (DGRAD ? -toff : toff) + nx_c0;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const T* g = ((a_mask[i] >> nx_tap) & 1u) ? a_ptr[i] + toff : zero_src;
      glds16(g, sa + i * (RG * KB));
    }
    nx_c0 += BK;
    if (nx_c0 >= p.KC) {
      nx_c0 = 0;
      ++nx_tap;
      if (++nx_kw == p.KW) {
        nx_kw = 0;
        ++nx_kh;
      }
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const T* g = w_ok[i] ? w_ptr[i] + (long long)ks * BK : zero_src;
      glds16(g, sw + i * (RG * KB));
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fk = lane >> 4;
  // lane part of a fragment address: row frow of a 16-row tile, chunk (kk * 4 + fk) ^ swz(frow)
  // (16-row tile offsets do not change the swizzle: swz only looks at r & 15)
  int foff[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) foff[kk] = frow * KB + (((kk * 4 + fk) ^ swz(frow)) << 4);
  const int fw0 = BM * KB + wn * (BN / WN) * KB;   // this wave's first weight row, relative to the stage
  const int fa0 = wm * (BM / WM) * KB;             // this wave's first pixel row

  // ---- prologue: D K-tiles in flight, K-tile 0 landed for everybody, group 1 half a period behind ----
#pragma unroll
  for (int s = 0; s < D; ++s) dma_stage(s, s);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(P * (D - 1)) : "memory");
  if (grp == 1) asm volatile("s_barrier" ::: "memory");

  int rd = 0, wr = D;
  for (int t = 0; t < nk; ++t) {
    // ---- L_t: request K-tile t+D, read this wave's fragments of K-tile t ----
    const bool more = t + D < nk;
    if (more) dma_stage(t + D, wr);
    wr = wr + 1 == NB ? 0 : wr + 1;
    const unsigned char* st = smem + rd * STAGE;
    rd = rd + 1 == NB ? 0 : rd + 1;
    uint4 fw[KK][TN], fa[KK][TM];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
        fw[kk][i] = *reinterpret_cast<const uint4*>(st + fw0 + i * (16 * KB) + foff[kk]);
#pragma unroll
      for (int j = 0; j < TM; ++j)
        fa[kk][j] = *reinterpret_cast<const uint4*>(st + fa0 + j * (16 * KB) + foff[kk]);
    }
    // this wave's part of K-tile t+1 has landed (the younger K-tiles stay in flight) ...
    if (more) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P * (D - 1)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // ... and its fragments are in registers: the other group may start overwriting / reading
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- C_t ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) QtMma<T>::run(acc[i][j], fw[kk][i], fa[kk][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
  }
  if (grp == 0) asm volatile("s_barrier" ::: "memory");   // (group 1's last MFMA segment)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // all reads done before the epilogue reuses LDS

  // ---- epilogue: straight from the accumulators ------------------------------------------------
  // The weight rows were staged permuted (perm32 above): the two 16x16 tiles (i = 2*pi, 2*pi+1) of a lane
  // hold EIGHT consecutive channels c0 .. c0+7 of one pixel, c0 = wave's first channel + pi*32 + fk*8.  Every
  // memory access of the epilogue is one 16-byte (bf16) / 32-byte (f32) access per lane; the 4 lanes of a
  // pixel cover 32 channels = 64 contiguous bytes.  No LDS round trip, no workgroup barrier per pass.
  constexpr int NP = TN / 2;   // channel pairs of 16x16 tiles per wave
  T* __restrict__ dst = static_cast<T*>(p.dst);
  const T* __restrict__ res = static_cast<const T*>(p.residual);
  const T* __restrict__ msk = static_cast<const T*>(p.relu_mask);
  const bool bwd_stats = p.bn_y[0] != nullptr;
  const bool want_stats = p.stats_partial != nullptr || bwd_stats;
  // destination row of each of this lane's TM pixels (-1: past the end; rows < 2^31, checked by the caller)
  int drow[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm * (BM / WM) + j * 16 + frow;
    drow[j] = -1;
    if (m < p.M) {
      drow[j] = m;
      if (p.dst_sub) {
        const unsigned img = fdiv((unsigned)m, p.div_ohw);
        const unsigned rem = (unsigned)m - img * (unsigned)(p.OH * p.OW);
        const unsigned oh = fdiv(rem, p.div_ow);
        const unsigned ow = rem - oh * (unsigned)p.OW;
        drow[j] = (int)((img * (unsigned)p.dst_h + oh * p.dst_sub + p.dst_oh) * (unsigned)p.dst_w + ow * p.dst_sub + p.dst_ow);
      }
    }
  }
  float* red = reinterpret_cast<float*>(smem);   // [WM][BN][3] per-wave-row sums (the ring is dead: barrier above)
#pragma unroll
  for (int pi = 0; pi < NP; ++pi) {
    const int cl = wn * (BN / WN) + pi * 32 + fk * 8;   // channel inside the tile
    const int c0 = n0 + cl;
    const bool n_ok = c0 < p.N;                          // N is a multiple of 8
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = (p.scale && n_ok) ? p.scale[c0 + e] : 1.f;
      sh[e] = (p.shift && n_ok) ? p.shift[c0 + e] : 0.f;
    }
    float mu0[8], is0[8], mu1[8], is1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mu0[e] = (bwd_stats && n_ok) ? p.bn_mean[0][c0 + e] : 0.f;
      is0[e] = (bwd_stats && n_ok) ? p.bn_invstd[0][c0 + e] : 0.f;
      mu1[e] = (p.bn_y[1] && n_ok) ? p.bn_mean[1][c0 + e] : 0.f;
      is1[e] = (p.bn_y[1] && n_ok) ? p.bn_invstd[1][c0 + e] : 0.f;
    }
    float s1[8], s2[8], s3[8];  // forward: sum v, sum v^2;  backward: sum g, sum g*xhat0, sum g*xhat1
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = s3[e] = 0.f;
    // pixel tiles in batches of JB: the batch's memory operands are requested before its first row is finished
    constexpr int JB = sizeof(T) == 2 ? 4 : 1;
    static_assert(TM % JB == 0, "pixel-tile batches");
#pragma unroll
    for (int jb = 0; jb < TM; jb += JB) {
      __builtin_amdgcn_sched_barrier(0);   // (keeps the loads of later batches from being hoisted over this one)
      Raw8<T> rres[JB], rmsk[JB], ry0[JB], ry1[JB];
      bool ok[JB];
#pragma unroll
      for (int u = 0; u < JB; ++u) {
        ok[u] = n_ok && drow[jb + u] >= 0;
        if (ok[u]) {
          const long long off = (long long)drow[jb + u] * p.N + c0;
          if (res) rres[u].load(res + off);
          if (msk) rmsk[u].load(msk + off);
          if (bwd_stats) ry0[u].load(static_cast<const T*>(p.bn_y[0]) + off);
          if (p.bn_y[1]) ry1[u].load(static_cast<const T*>(p.bn_y[1]) + off);
        }
      }
#pragma unroll
      for (int u = 0; u < JB; ++u) {
        const int j = jb + u;
        float v[8] = {acc[2 * pi][j][0],     acc[2 * pi][j][1],     acc[2 * pi][j][2],     acc[2 * pi][j][3],
                      acc[2 * pi + 1][j][0], acc[2 * pi + 1][j][1], acc[2 * pi + 1][j][2], acc[2 * pi + 1][j][3]};
        if (!ok[u]) continue;
        const long long off = (long long)drow[j] * p.N + c0;
        if (!bwd_stats) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[e] += v[e];
            s2[e] += v[e] * v[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
        if (res) {
          float rv[8];
          rres[u].get(rv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rv[e];
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (msk) {
          float mv[8];
          rmsk[u].get(mv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
        }
        QtVec8<T>::store(dst + off, v);
        if (bwd_stats) {
          float yv[8];
          ry0[u].get(yv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[e] += v[e];
            s2[e] += v[e] * (yv[e] - mu0[e]) * is0[e];
          }
          if (p.bn_y[1]) {
            ry1[u].get(yv);
#pragma unroll
            for (int e = 0; e < 8; ++e) s3[e] += v[e] * (yv[e] - mu1[e]) * is1[e];
          }
        }
      }
    }
    if (want_stats) {
      // sum over the 16 pixels (lanes with equal fk) of the wave, fixed butterfly order -> deterministic
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int sft = 1; sft < 16; sft <<= 1) {
          s1[e] += __shfl_xor(s1[e], sft);
          s2[e] += __shfl_xor(s2[e], sft);
          s3[e] += __shfl_xor(s3[e], sft);
        }
      }
      if (frow == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          red[(wm * BN + cl + e) * 3 + 0] = s1[e];
          red[(wm * BN + cl + e) * 3 + 1] = s2[e];
          red[(wm * BN + cl + e) * 3 + 2] = s3[e];
        }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
      for (int r = 0; r < WM; ++r) {   // fixed order over the wave rows
        a += red[(r * BN + tid) * 3 + 0];
        b += red[(r * BN + tid) * 3 + 1];
        c += red[(r * BN + tid) * 3 + 2];
      }
      float* o0 = bwd_stats ? p.bn_partial[0] : p.stats_partial;
      o0[((long long)mt * 2 + 0) * p.N + n0 + tid] = a;
      o0[((long long)mt * 2 + 1) * p.N + n0 + tid] = b;
      if (bwd_stats && p.bn_y[1]) {
        p.bn_partial[1][((long long)mt * 2 + 0) * p.N + n0 + tid] = a;
        p.bn_partial[1][((long long)mt * 2 + 1) * p.N + n0 + tid] = c;
      }
    }
  }
}

template <typename T, int BM, int BN, int KB, int WM, int WN, int NB, bool DGRAD>
int launch(const ConvArgs& a, hipStream_t stream) {
  constexpr int LDS = NB * (BM + BN) * KB;   // the ring (the statistics scratch of the epilogue, WM*BN*12 B, reuses it)
  static_assert(LDS <= 160 * 1024 && WM * BN * 12 <= LDS, "LDS budget");
  auto kern = conv_pp_kernel<T, BM, BN, KB, WM, WN, NB, DGRAD>;
  static std::atomic<unsigned long long> lds_limit_set{0};  // per device
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(kern), LDS, lds_limit_set)) return rc;
  ConvArgs args = a;
  args.gridM = qt_cdiv(a.M, BM);
  args.gridN = qt_cdiv(a.N, BN);
  hipLaunchKernelGGL(kern, dim3(args.gridM * args.gridN), dim3(kNT), LDS, stream, args);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

// 0: off, 1: on (default).  QTCNN_PP_CONV / qt_set_pp_conv: same-box A/B against the generic kernel.
int g_pp_enabled = -1;
inline int pp_enabled() {
  if (g_pp_enabled < 0) {
    const char* e = getenv("QTCNN_PP_CONV");
    g_pp_enabled = e ? atoi(e) : 1;
  }
  return g_pp_enabled;
}

// wide tile (256 channels, 64-byte K rows) or narrow (128 channels, 128-byte K rows)
inline bool wide(const ConvArgs&) { return false; }   // (the 256 x 256 x 64 B shape: see DESIGN.md)

template <typename T, bool DGRAD>
int dispatch(const ConvArgs& a, hipStream_t stream) {
  return launch<T, 256, 128, 128, 4, 2, 3, DGRAD>(a, stream);
}

}  // namespace

extern "C" void qt_set_pp_conv(int mode) { g_pp_enabled = mode < 0 ? 1 : mode; }

bool qt_pp_eligible(const ConvArgs& a, int dtype, bool dgrad) {
  if (!pp_enabled()) return false;
  const int esz = dtype == QT_F32 ? 4 : 2;
  const int kb = wide(a) ? 64 : 128;
  if (a.stride != 1) return false;                          // (forward stride 2 would work; not measured yet)
  if (a.N < 128 || a.M < 8192) return false;                // thin problems keep the 128-row tiles
  if ((a.KC * esz) % kb != 0) return false;                 // a K-tile lies inside one tap
  if ((long long)a.ntaps * a.KC * esz / kb < 8) return false;  // short K loops: the ring's prologue does not pay
  (void)dgrad;
  return true;
}

int qt_pp_tile_m(const ConvArgs&, int) { return 256; }

int qt_pp_launch(const ConvArgs& a, int dtype, bool dgrad, hipStream_t stream) {
  if (dtype == QT_F32) return dgrad ? dispatch<float, true>(a, stream) : dispatch<float, false>(a, stream);
  return dgrad ? dispatch<bf16_t, true>(a, stream) : dispatch<bf16_t, false>(a, stream);
}
