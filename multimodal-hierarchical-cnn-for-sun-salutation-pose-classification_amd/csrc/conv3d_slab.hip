// Conv3d(32 -> 64, 3x3x3, pad 1) of the clip models' second block with the input frame slabs resident in LDS, bf16 MFMA.
//
// Replaces nn.Conv3d(32, 64, kernel_size=(3,3,3), padding=(1,1,1)) of conv3d_block2 (/root/reference/3dcnn/models.py:115,
// reached from Quadtree3DCNN.forward, models.py:189-193) in the forward pass.
//
// The 27-tap implicit GEMM (conv_igemm.hip, qt_conv_desc.kt = 3) fetches every 128-byte K row 27 times through L2 --
// 11 GB per launch at 32 clips x 8 frames of 112 x 112, 7.9 TB/s of L2 -> LDS, 1.44 ms -- and half of each row is the
// zero padding 32 -> 64 channels.  Here a workgroup of 8 waves walks the frames of one (clip, 2-row slab):
//   * a frame slab = 4 rows x (W + 2) pixels x the 32 REAL channels (64 B per pixel) arrives by LDS-DMA straight from the
//     time-major NHWC map (first half of its 128-byte rows), once per slab (2x with the row halo), into a ring of four
//     (frames t - 1, t, t + 1 in use, frame t + 2 landing under the MFMAs of frame t), ONE barrier per frame;
//   * a tap (kt, kh, kw) is a byte offset into the ring: the B fragment of 16 pixels is one ds_read_b128 per lane;
//   * wave (n, half) owns output channels 16 n .. 16 n + 15 and every second block of 16 pixels; its 27 filter fragments
//     (108 VGPRs) stay in registers for the whole walk, so the K loop is one LDS read + one MFMA per tap, two pixel blocks
//     interleaved; frames outside the clip are skipped (no MFMA work on padding frames);
//   * epilogue: bias-free accumulator + BatchNorm3d partial sums (train) or scale / shift / ReLU (eval), 8-byte stores.
// Bound (round 3's reading): LDS reads (one 1 KB fragment read per MFMA, 4 SIMDs share 128 B/clk: at most 50 % of the matrix pipe).
// Round 4 measured it: with two channel blocks per wave (CPW = 2: half the fragment reads per MFMA) and the reads issued a
// group ahead of their MFMAs the forward goes 534 -> 464 us, no further; counters of that form (eval forward, 32 clips x 8
// frames of 112 x 112): matrix pipes busy 40 % of the CU-busy cycles, LDS active 40 % (half of it the 2-way conflict of a
// 64-byte pixel pitch under ds_read_b128), 2.7 VALU + 1.5 SALU instructions per MFMA.  Neither pipe is the bound: four waves
// per CU issue-limited by the per-block address / epilogue arithmetic around 27 CPW MFMAs, and one barrier per frame.
#include <stdint.h>
#include <stdlib.h>

#include "qt_common.h"

namespace {

constexpr int S3_R = 2;        // output rows per slab
constexpr int S3_RING = 4;     // frames t - 1, t, t + 1 under the MFMAs + frame t + 2 landing

struct S3Args {
  const bf16_t* x;        // [T][B][H][W][xc], channels 0..31 read
  const bf16_t* w;        // [64][27][64]: element tap * 64 + c (qt_pack_conv3d_block), c < 32 read
  bf16_t* y;              // [T][B][H][W][64]
  const float* scale;     // nullable: y = conv * scale + shift (+ ReLU)
  const float* shift;
  float* stats;           // [gridDim.x][2][64] or NULL
  float* scratch;         // data gradient: f32 [T][B][H][W][32] partial sums between its two passes
  int relu, B, T, H, W, xc, items;
  int yc;                 // OUT == 2: channels per row of y (64: channels 32..63 written as zeros; 32: none)
  int coff;               // first of the 32 channels of x (and of the filter's K rows) this launch reads
};

// NB = 16-channel output blocks (4: the forward's 64 channels; 2: the data gradient's 32 input channels, waves (block, quarter
// of the pixel blocks)).  FLIP: tap (kt, kh, kw) multiplies filter tap 26 - index (the data gradient is the same walk over dy
// with the flipped filter [I][27][O]).  OUT: 0 = bf16 rows of 64 (NB * 16 channels written); 1 = f32 partial sums to
// scratch; 2 = scratch + accumulator -> bf16 rows of yc = 64 (channels 32..63 zero) or 32 channels.  The data gradient contracts over 64 channels =
// two launches of 32 (coff 0 / 32: an LDS ring of 64-channel slabs would need 233 KB), joined through the f32 scratch.
// CPW (round 4) = 16-channel blocks per WAVE.  1: eight waves, every B fragment (16 pixels x 32 channels of a tap, 1 KB) feeds one
// MFMA -- the LDS read path (128 B / clk per CU) then holds the matrix pipe at 50 %.  2: four waves of 512 registers each keep the
// filter fragments of two channel blocks (216 VGPRs), a fragment read feeds two MFMAs: half the LDS traffic per flop.
template <bool AFF, bool STATS, int NB = 4, bool FLIP = false, int OUT = 0, int CPW = 1>
__global__ __launch_bounds__(512 / CPW) void conv3d_c32_kernel(S3Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  constexpr int NW = 8 / CPW;                // waves
  constexpr int NG = NB / CPW;               // groups of CPW channel blocks
  constexpr int PARTS = NW / NG;             // waves that share a channel group split the pixel blocks
  static_assert(NB % CPW == 0 && NW % NG == 0, "channel blocks per wave");
  const int ng = wave % NG, half = wave / NG;
  const int W = p.W, H = p.H, T = p.T;
  const int rowb = (W + 2) * 64, slab = (S3_R + 2) * rowb;
  const unsigned smem_base = lds_addr_of(smem);

  for (int i = tid * 16; i < S3_RING * slab; i += NW * 64 * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);

  // filter fragments (A operand): row li = output channel 16 (ng CPW + cb) + li, k-group lg = input channels 8 lg .. 8 lg + 7 of the tap
  uint4 wf[CPW][27];
#pragma unroll
  for (int cb = 0; cb < CPW; ++cb)
#pragma unroll
    for (int tap = 0; tap < 27; ++tap)
      wf[cb][tap] = *reinterpret_cast<const uint4*>(p.w + ((size_t)((ng * CPW + cb) * 16 + li) * 27 + (FLIP ? 26 - tap : tap)) * 64 +
                                                    p.coff + lg * 8);

  float sc[CPW][4], sh[CPW][4], s1[CPW][4], s2[CPW][4];
#pragma unroll
  for (int cb = 0; cb < CPW; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = (ng * CPW + cb) * 16 + lg * 4 + r;
      sc[cb][r] = AFF ? p.scale[ch] : 1.f;
      sh[cb][r] = AFF ? p.shift[ch] : 0.f;
      s1[cb][r] = s2[cb][r] = 0.f;
    }

  const int nbw = W >> 4;                    // blocks of 16 pixels per row
  const int ndma = (S3_R + 2) * nbw;         // 1 KB DMA instructions per frame slab
  const unsigned char* zero_src = reinterpret_cast<const unsigned char*>(qt_zero_page) + (lane & 7) * 16;
  // frame f of (clip b, rows h0 - 1 .. h0 + S3_R) -> ring slot f % 4; instruction j = (slab row, block of 16 pixels)
  auto dma_frame = [&](int b, int f, int h0) {
    const unsigned dst0 = smem_base + (unsigned)(f % S3_RING) * slab + 64;   // pixel 0 of a slab row is the left halo
    for (int j = wave; j < ndma; j += NW) {
      const int r = j / nbw, blk = j - r * nbw;
      const int hh = h0 - 1 + r;
      const unsigned char* src =
          (unsigned)hh < (unsigned)H
              ? reinterpret_cast<const unsigned char*>(p.x + ((((size_t)f * p.B + b) * H + hh) * W + blk * 16 + (lane >> 2)) * p.xc +
                                                       p.coff) +
                    (lane & 3) * 16
              : zero_src;
      glds16(src, dst0 + (unsigned)r * rowb + (unsigned)blk * 1024);
    }
  };

  const int slabs_per_img = H / S3_R, nmb = S3_R * nbw;
  // OUT == 2: the first pass's partial sums of a frame's pixel blocks (W <= 128: at most two rounds of two blocks per wave) are
  // requested a frame ahead (loaded where they are added they cost a memory latency per pixel block with nothing to hide it)
  float4 qc[4][CPW], qn[4][CPW];
  auto load_partials = [&](int b, int t, int h0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int m = half + (k >> 1) * 2 * PARTS + (k & 1) * PARTS;
      if (m < nmb) {
        const int rr = m / nbw, cc = m - rr * nbw;
        const size_t pos = (((size_t)t * p.B + b) * H + h0 + rr) * W + cc * 16 + li;
#pragma unroll
        for (int cb = 0; cb < CPW; ++cb)
          qn[k][cb] = *reinterpret_cast<const float4*>(p.scratch + pos * 32 + (ng * CPW + cb) * 16 + lg * 4);
      }
    }
  };
  __syncthreads();   // the zero fill (halo columns) is complete before any DMA lands
  for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
    const int b = item / slabs_per_img, h0 = (item - b * slabs_per_img) * S3_R;
    // (every wave passed the barrier after the previous item's last frame: the ring is free)
    if (OUT == 2) load_partials(b, 0, h0);
    dma_frame(b, 0, h0);
    if (T > 1) dma_frame(b, 1, h0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      if (OUT == 2) {   // (in front of the DMA in issue order: the frame's closing vmcnt(0) then waits for loads a frame old)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int cb = 0; cb < CPW; ++cb) qc[k][cb] = qn[k][cb];
        if (t + 1 < T) load_partials(b, t + 1, h0);
      }
      if (t + 2 < T) dma_frame(b, t + 2, h0);   // into the slot of frame t - 2, which every wave finished before the last barrier

      const int kt_lo = t == 0 ? 1 : 0, kt_hi = t + 1 < T ? 2 : 1;
      auto round = [&](const int it, const int mb) {   // two pixel blocks (mb, mb + PARTS) interleaved
        const int mb1 = mb + PARTS;
        const bool two = mb1 < nmb;
        const int r0 = mb / nbw, c0 = mb - r0 * nbw;
        const int r1 = two ? mb1 / nbw : r0, c1 = two ? mb1 - r1 * nbw : c0;
        const unsigned char* a0 = smem + r0 * rowb + (c0 * 16 + li) * 64 + lg * 16;
        const unsigned char* a1 = smem + r1 * rowb + (c1 * 16 + li) * 64 + lg * 16;
        f32x4 acc0[CPW], acc1[CPW];
#pragma unroll
        for (int cb = 0; cb < CPW; ++cb) acc0[cb] = acc1[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Nine groups (frame tap kt, row tap kh) of 6 fragment reads + 6 CPW MFMAs, software-pipelined by hand (round 4): the
        // reads of group g + 1 are issued BEFORE the MFMAs of group g.  Written tap by tap the compiler waits for every read in
        // front of its MFMA (the ISA was `ds_read_b128, s_waitcnt lgkmcnt(0), v_mfma` 54 times per round: one LDS latency per
        // 16-cycle MFMA), and grouping the reads in the source alone does not survive its scheduler.  Frames outside the clip:
        // a uniform skip of the group's reads and MFMAs.
        uint4 xs[2][2][3];
        auto live = [&](int g) { return g / 3 >= kt_lo && g / 3 <= kt_hi; };
        auto read_group = [&](int g, uint4 (&x)[2][3]) {
          const int kt = g / 3, kh = g - kt * 3;
          const int off = ((t + kt + S3_RING - 1) % S3_RING) * slab + kh * rowb;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            x[0][kw] = *reinterpret_cast<const uint4*>(a0 + off + kw * 64);
            x[1][kw] = *reinterpret_cast<const uint4*>(a1 + off + kw * 64);
          }
        };
        if (live(0)) read_group(0, xs[0]);
#pragma unroll
        for (int g = 0; g < 9; ++g) {
          if (g + 1 < 9 && live(g + 1)) read_group(g + 1, xs[(g + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          if (live(g)) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
              for (int cb = 0; cb < CPW; ++cb) {
                const uint4 wk = wf[cb][g * 3 + kw];
                acc0[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wk),
                                                                   __builtin_bit_cast(bf16x8, xs[g & 1][0][kw]), acc0[cb], 0, 0, 0);
                acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wk),
                                                                   __builtin_bit_cast(bf16x8, xs[g & 1][1][kw]), acc1[cb], 0, 0, 0);
              }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u == 1 && !two) break;
          const int rr = u ? r1 : r0, cc = u ? c1 : c0;
          const size_t pos = (((size_t)t * p.B + b) * H + h0 + rr) * W + cc * 16 + li;
#pragma unroll
          for (int cb = 0; cb < CPW; ++cb) {
            const f32x4 acc = u ? acc1[cb] : acc0[cb];
            const int chb = (ng * CPW + cb) * 16 + lg * 4;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float a = acc[r];
              if (STATS) {
                s1[cb][r] += a;
                s2[cb][r] += a * a;
              }
              v[r] = a;
              if (AFF) {
                v[r] = a * sc[cb][r] + sh[cb][r];
                if (p.relu) v[r] = fmaxf(v[r], 0.f);
              }
            }
            if (OUT == 1) {
              *reinterpret_cast<float4*>(p.scratch + pos * 32 + chb) = make_float4(v[0], v[1], v[2], v[3]);
              continue;
            }
            if (OUT == 2) {
              const float4 q = qc[it * 2 + u][cb];
              v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
            }
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
            const int yc = OUT == 2 ? p.yc : 64;
            *reinterpret_cast<bf16x4*>(p.y + pos * yc + chb) = o;
            if (OUT == 2 && yc > 32) *reinterpret_cast<uint2*>(p.y + pos * 64 + 32 + chb) = make_uint2(0u, 0u);
          }
        }
      };
      if constexpr (OUT == 2) {   // (unrolled: the prefetched partial sums are registers)
        static_assert(NB == 2, "the joining pass: two rounds per wave and frame at W <= 128");
        if (half < nmb) round(0, half);
        if (half + 2 * PARTS < nmb) round(1, half + 2 * PARTS);
      } else {
        for (int mb = half; mb < nmb; mb += 2 * PARTS) round(0, mb);
      }
      // this wave's share of frame t + 2 has landed.  (A counted wait that leaves the frame's stores outstanding -- the DMA is
      // issued in front of them and the counter retires in order -- measured the same: 5.39 / 5.39 against 5.38 / 5.40 ms per
      // Quadtree3DCNN step; the store acknowledgements are not what a frame waits for.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                     // ... everyone's has, and everyone is done with output frame t
    }
  }

  if (STATS) {
    static_assert(!STATS || (NB == 4 && PARTS == 2), "statistics: the forward form");
    // 16 lanes of a row hold 16 pixels of the same 4 channels; the two waves of a channel block through LDS, fixed order
    float* red = reinterpret_cast<float*>(smem);   // [2 halves][2][64]
#pragma unroll
    for (int cb = 0; cb < CPW; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a1 = qt_row16_sum(s1[cb][r]), a2 = qt_row16_sum(s2[cb][r]);
        if (li == 0) {
          red[(half * 2 + 0) * 64 + (ng * CPW + cb) * 16 + lg * 4 + r] = a1;
          red[(half * 2 + 1) * 64 + (ng * CPW + cb) * 16 + lg * 4 + r] = a2;
        }
      }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, ch = tid & 63;
      p.stats[((size_t)blockIdx.x * 2 + which) * 64 + ch] = red[(0 * 2 + which) * 64 + ch] + red[(1 * 2 + which) * 64 + ch];
    }
  }
}

// ---- weight gradient of the same layer ------------------------------------------------------------------------------------
//   dW[co][ci][kt][kh][kw] = sum over positions of dy[pos][co] * x[pos + tap][ci]
// Same walk, the same ring of 32-channel x slabs.  The contraction runs over positions: MFMA k = a chunk of 32 consecutive
// positions of the 2-row tile.  dy chunks (32 positions x 64 channels = 4 KB, contiguous in memory) stream through a shared
// ring of six by LDS-DMA, three ahead, one barrier per chunk.  Both operands are position-major in LDS and come out k-major
// through ds_read_b64_tr_b16 at per-lane addresses: dy as 16 output channels x 32 positions, x as 16 input channels x 32
// positions at the tap's byte offset.  Wave (cb, ib) owns output channels 16 cb.., input channels 16 ib.. of ALL 27 taps:
// 27 accumulator tiles (108 VGPRs), 56 transposing reads per 27 MFMAs -- LDS-read bound like the forward.  Partial filters
// per workgroup, added in a fixed order straight into nn.Conv3d's [64][32][3][3][3] layout: deterministic, no atomics.
// Measured 548 us at 32 clips x 8 frames of 112 x 112 (three launches of the tile kernel over 64-channel-padded rows: 720).
// (Swapping the two 32-byte halves of a slab pixel on every second group of four pixels -- positions 4 apart are 256 B
// apart -- measured 633 us: the extra per-tap address arithmetic costs more than any bank conflict it removes.)
constexpr int S3W_NCH = 6, S3W_D = 3;
constexpr int S3W_PART = 64 * 27 * 32;

struct S3WArgs {
  const bf16_t* x;        // [T][B][H][W][xc], channels 0..31
  const bf16_t* dy;       // [T][B][H][W][64]
  float* part;            // [gridDim.x][64][27][32]
  int B, T, H, W, xc, items;
};

__global__ __launch_bounds__(512) void conv3d_c32_wgrad_kernel(S3WArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int lrow = 4 * lg + (li >> 2), pp = li & 3;
  const int cb = wave & 3, ib = wave >> 2;
  const int W = p.W, H = p.H, T = p.T;
  const int rowb = (W + 2) * 64, slab = (S3_R + 2) * rowb;
  const unsigned smem_base = lds_addr_of(smem);
  const unsigned dyring = smem_base + S3_RING * slab;

  for (int i = tid * 16; i < S3_RING * slab; i += 512 * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);

  f32x4 acc[27];
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) acc[tap] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nbw = W >> 4, ndma = (S3_R + 2) * nbw;
  const unsigned char* zero_src = reinterpret_cast<const unsigned char*>(qt_zero_page) + (lane & 7) * 16;
  auto dma_frame = [&](int b, int f, int h0) {
    const unsigned dst0 = smem_base + (unsigned)(f % S3_RING) * slab + 64;
    for (int j = wave; j < ndma; j += 8) {
      const int r = j / nbw, blk = j - r * nbw;
      const int hh = h0 - 1 + r;
      const unsigned char* src =
          (unsigned)hh < (unsigned)H
              ? reinterpret_cast<const unsigned char*>(p.x + ((((size_t)f * p.B + b) * H + hh) * W + blk * 16 + (lane >> 2)) * p.xc) +
                    (lane & 3) * 16
              : zero_src;
      glds16(src, dst0 + (unsigned)r * rowb + (unsigned)blk * 1024);
    }
  };

  // dy chunk stream: (item, frame, chunk of 32 positions of the 2 x W tile); waves 0..3 move one KB each per chunk
  const int slabs_per_img = H / S3_R, nck = (S3_R * W) >> 5;
  int is_item = blockIdx.x, is_t = 0, is_c = 0, n_issued = 0, n_used = 0;
  auto issue = [&]() {
    if (is_item >= p.items) return;
    if (wave < 4) {
      const int b = is_item / slabs_per_img, h0 = (is_item - b * slabs_per_img) * S3_R;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(
          p.dy + ((((size_t)is_t * p.B + b) * H + h0) * W + is_c * 32) * 64);
      glds16(src + wave * 1024 + lane * 16, dyring + (unsigned)(n_issued % S3W_NCH) * 4096u + (unsigned)wave * 1024u);
    }
    ++n_issued;
    if (++is_c == nck) {
      is_c = 0;
      if (++is_t == T) {
        is_t = 0;
        is_item += gridDim.x;
      }
    }
  };

  __syncthreads();   // the zero fill is complete before any DMA lands
#pragma unroll
  for (int d = 0; d < S3W_D; ++d) issue();

  for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
    const int b = item / slabs_per_img, h0 = (item - b * slabs_per_img) * S3_R;
    dma_frame(b, 0, h0);
    if (T > 1) dma_frame(b, 1, h0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      if (t + 2 < T) dma_frame(b, t + 2, h0);
      const int kt_lo = t == 0 ? 1 : 0, kt_hi = t + 1 < T ? 2 : 1;
      for (int c = 0; c < nck; ++c) {
        issue();   // chunk n_used + S3W_D into the slot of chunk n_used - 3, which every wave finished before the last barrier
        // positions lrow and lrow + 16 of the chunk -> (row, column) of the tile
        const int p0 = c * 32 + lrow, p1 = p0 + 16;
        const int r0 = p0 >= W ? 1 : 0, r1 = p1 >= W ? 1 : 0;      // (S3_R == 2)
        const unsigned xo0 = (unsigned)(r0 * rowb + (p0 - r0 * W) * 64 + ib * 32 + pp * 8);
        const unsigned xo1 = (unsigned)(r1 * rowb + (p1 - r1 * W) * 64 + ib * 32 + pp * 8);
        const unsigned da = dyring + (unsigned)(n_used % S3W_NCH) * 4096u + lrow * 128 + cb * 32 + pp * 8;
        ++n_used;
        const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS_AS s16x4*)(size_t)da);
        const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS_AS s16x4*)(size_t)(da + 16 * 128));
        const uint2 al = __builtin_bit_cast(uint2, alo), ah = __builtin_bit_cast(uint2, ahi);
        const uint4 fa = make_uint4(al.x, al.y, ah.x, ah.y);
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
          if (kt < kt_lo || kt > kt_hi) continue;
          const unsigned so = smem_base + (unsigned)(((t + kt + S3_RING - 1) % S3_RING) * slab);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const unsigned off = so + kh * rowb + kw * 64;
              const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS_AS s16x4*)(size_t)(off + xo0));
              const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS_AS s16x4*)(size_t)(off + xo1));
              const uint2 bl = __builtin_bit_cast(uint2, blo), bh = __builtin_bit_cast(uint2, bhi);
              const uint4 fb = make_uint4(bl.x, bl.y, bh.x, bh.y);
              acc[(kt * 3 + kh) * 3 + kw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc[(kt * 3 + kh) * 3 + kw], 0, 0, 0);
            }
        }
        // chunk n_used (next) was issued three iterations ago: at most this wave's two younger dy instructions (and whatever
        // x slab instructions it issued since) may still be in flight; at a frame's end everything has to have landed
        const int younger = n_issued - n_used - 1;   // dy chunks issued after the next one to be read
        if (c + 1 < nck && younger >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (c + 1 < nck && younger == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
  }

  // lane: output channels 16 cb + 4 lg + r, input channel 16 ib + li of every tap
  float* out = p.part + (size_t)blockIdx.x * S3W_PART;
#pragma unroll
  for (int tap = 0; tap < 27; ++tap)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[((cb * 16 + lg * 4 + r) * 27 + tap) * 32 + ib * 16 + li] = acc[tap][r];
}

// dW [64][32 ci][27 taps] (nn.Conv3d weight layout [co][ci][kt][kh][kw]) = sum over the workgroups' partial filters, ascending
__global__ __launch_bounds__(256) void s3_wgrad_sum_kernel(const float* __restrict__ part, float* __restrict__ dw, int nparts) {
  __shared__ float red[8][32];
  const int col = threadIdx.x & 31, sg = threadIdx.x >> 5;
  // element i = (co, tap, ci) of a partial filter: consecutive lanes read consecutive floats (reading in the OUTPUT order
  // (co, ci, tap) fetched a 64-byte sector per 4-byte element: 573 MB and 81 us per step by the PMC counters for 57 MB
  // of partial filters); the 221 KB result is scattered instead
  const int o = blockIdx.x * 32 + col;
  float s = 0.f;
  if (o < 64 * 32 * 27) {
    const float* src = part + o;
    const int per = (nparts + 7) / 8;
    const int beg = sg * per, end = min(nparts, beg + per);
#pragma unroll 8
    for (int k = beg; k < end; ++k) s += src[(size_t)k * S3W_PART];
  }
  red[sg][col] = s;
  __syncthreads();
  if (sg == 0 && o < 64 * 32 * 27) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][col];
    const int co = o / (27 * 32), rem = o - co * (27 * 32);
    const int tap = rem >> 5, ci = rem & 31;
    dw[(co * 32 + ci) * 27 + tap] = t;
  }
}

int s3_grid(int items) {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t pr;
      if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n = pr.multiProcessorCount;
    }
    return n;
  }();
  return items < cus ? items : cus;
}

bool s3_enabled() {
  static const bool on = [] {
    const char* e = getenv("QTCNN_CONV3D_SLAB");
    return !(e && e[0] == '0');
  }();
  return on;
}

// QTCNN_S3_CPW (default 2): 16-channel blocks per wave of the forward / data-gradient slab kernel (1: the eight-wave form of round 3)
int s3_cpw() {
  static const int v = [] {
    const char* e = getenv("QTCNN_S3_CPW");
    return (e && e[0] == '1') ? 1 : 2;
  }();
  return v;
}

bool s3_shape_ok(int batch, int frames, int h, int w) {
  return s3_enabled() && batch > 0 && frames > 0 && h >= S3_R && h % S3_R == 0 && w >= 16 && w % 16 == 0 && w <= 128;
}

}  // namespace

// rows of BatchNorm partial sums qt_conv3d_c32_fwd writes ([rows][2][64]); 0 = shape not covered (take qt_conv2d_igemm)
extern "C" int qt_conv3d_c32_stats_rows(int batch, int frames, int h, int w) {
  return s3_shape_ok(batch, frames, h, w) ? s3_grid(batch * (h / S3_R)) : 0;
}

extern "C" int qt_conv3d_c32_fwd(int dtype, const void* x, int x_channels, const void* w_packed, void* y, const float* scale,
                                 const float* shift, int relu, float* stats, int batch, int frames, int h, int w, void* stream) {
  QT_CHECK_ARG(x && w_packed && y && batch > 0 && frames > 0 && h > 0 && w > 0, "qt_conv3d_c32_fwd: bad argument");
  QT_CHECK_ARG(x_channels >= 32 && x_channels % 8 == 0, "qt_conv3d_c32_fwd: x rows of %d channels", x_channels);
  QT_CHECK_ARG(!(scale && stats), "qt_conv3d_c32_fwd: scale / shift and statistics are exclusive");
  QT_CHECK_ARG(!scale || shift, "qt_conv3d_c32_fwd: scale without shift");
  if (dtype != QT_BF16 || !s3_shape_ok(batch, frames, h, w) || ((uintptr_t)x % 16) != 0 || ((uintptr_t)w_packed % 16) != 0) {
    qt_set_error("qt_conv3d_c32_fwd: bf16, H %% 2 == 0, W %% 16 == 0, W <= 128, 16-byte aligned operands only (use qt_conv2d_igemm "
                 "with kt = 3)");
    return QT_ERR_UNSUPPORTED;
  }
  S3Args a;
  a.x = (const bf16_t*)x; a.w = (const bf16_t*)w_packed; a.y = (bf16_t*)y; a.scale = scale; a.shift = shift; a.stats = stats;
  a.relu = relu; a.B = batch; a.T = frames; a.H = h; a.W = w; a.xc = x_channels; a.items = batch * (h / S3_R);
  a.scratch = nullptr; a.coff = 0; a.yc = 64;
  int lds = S3_RING * (S3_R + 2) * (w + 2) * 64;
  if (lds < 2 * 2 * 64 * 4) lds = 2 * 2 * 64 * 4;
  const dim3 grid(s3_grid(a.items));
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = QT_OK;
#define QT_S3_LAUNCH(...)                                                                                    \
  do {                                                                                                       \
    static std::atomic<unsigned long long> done{0};                                                          \
    if ((rc = qt_raise_lds_limit((const void*)conv3d_c32_kernel<__VA_ARGS__>, lds, done)) != QT_OK) return rc; \
    hipLaunchKernelGGL((conv3d_c32_kernel<__VA_ARGS__>), grid, blk, lds, s, a);                              \
  } while (0)
  if (s3_cpw() == 2) {
    const dim3 blk(256);
    if (scale) QT_S3_LAUNCH(true, false, 4, false, 0, 2);
    else if (stats) QT_S3_LAUNCH(false, true, 4, false, 0, 2);
    else QT_S3_LAUNCH(false, false, 4, false, 0, 2);
  } else {
    const dim3 blk(512);
    if (scale) QT_S3_LAUNCH(true, false);
    else if (stats) QT_S3_LAUNCH(false, true);
    else QT_S3_LAUNCH(false, false);
  }
  QT_CHECK_LAUNCH();
  return QT_OK;
}

// Data gradient of the same layer: dx [T][B][H][W][dx_channels] (channels 0..31 = d(loss)/d(input); dx_channels = 64: 32..63 zero) from
// dy [T][B][H][W][64] and qt_pack_conv3d_block's data-gradient filter [64][27][64] ([input channel][tap][output channel]).
// Two launches of the slab kernel over dy's channels 0..31 / 32..63 joined through `scratch` (f32 [positions][32]).
extern "C" size_t qt_conv3d_c32_dgrad_scratch_bytes(int batch, int frames, int h, int w) {
  return s3_shape_ok(batch, frames, h, w) ? (size_t)frames * batch * h * w * 32 * sizeof(float) : 0;
}

extern "C" int qt_conv3d_c32_dgrad(int dtype, const void* dy, const void* w_dgrad_packed, void* dx, int dx_channels, void* scratch,
                                   size_t scratch_bytes, int batch, int frames, int h, int w, void* stream) {
  QT_CHECK_ARG(dy && w_dgrad_packed && dx && batch > 0 && frames > 0 && h > 0 && w > 0, "qt_conv3d_c32_dgrad: bad argument");
  QT_CHECK_ARG(dx_channels == 32 || dx_channels == 64, "qt_conv3d_c32_dgrad: dx rows of %d channels (32 or 64)", dx_channels);
  const size_t need = qt_conv3d_c32_dgrad_scratch_bytes(batch, frames, h, w);
  if (dtype != QT_BF16 || need == 0 || ((uintptr_t)dy % 16) != 0 || ((uintptr_t)w_dgrad_packed % 16) != 0 || ((uintptr_t)dx % 16) != 0) {
    qt_set_error("qt_conv3d_c32_dgrad: bf16, even H, W %% 16 == 0, W <= 128, 16-byte aligned operands only (use qt_conv2d_igemm "
                 "with kt = 3, mode QT_CONV_DGRAD)");
    return QT_ERR_UNSUPPORTED;
  }
  QT_CHECK_ARG(scratch && scratch_bytes >= need && ((uintptr_t)scratch % 16) == 0,
               "qt_conv3d_c32_dgrad: scratch of %zu bytes, %zu needed", scratch_bytes, need);
  S3Args a;
  a.x = (const bf16_t*)dy; a.w = (const bf16_t*)w_dgrad_packed; a.y = (bf16_t*)dx; a.scale = nullptr; a.shift = nullptr;
  a.stats = nullptr; a.scratch = (float*)scratch;
  a.relu = 0; a.B = batch; a.T = frames; a.H = h; a.W = w; a.xc = 64; a.yc = dx_channels; a.items = batch * (h / S3_R);
  const int lds = S3_RING * (S3_R + 2) * (w + 2) * 64;
  const dim3 grid(s3_grid(a.items));
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = QT_OK;
  if (s3_cpw() == 2) {
    const dim3 blk(256);
    a.coff = 0;
    QT_S3_LAUNCH(false, false, 2, true, 1, 2);
    QT_CHECK_LAUNCH();
    a.coff = 32;
    QT_S3_LAUNCH(false, false, 2, true, 2, 2);
    QT_CHECK_LAUNCH();
  } else {
    const dim3 blk(512);
    a.coff = 0;
    QT_S3_LAUNCH(false, false, 2, true, 1);
    QT_CHECK_LAUNCH();
    a.coff = 32;
    QT_S3_LAUNCH(false, false, 2, true, 2);
    QT_CHECK_LAUNCH();
  }
#undef QT_S3_LAUNCH
  return QT_OK;
}

// Weight gradient of the same layer: dweight [64][32][3][3][3] f32 (nn.Conv3d's layout, every element written) from
// x [T][B][H][W][x_channels] (channels 0..31) and dy [T][B][H][W][64]; per-workgroup partial filters in `workspace`.
extern "C" size_t qt_conv3d_c32_wgrad_workspace_bytes(int batch, int frames, int h, int w) {
  return s3_shape_ok(batch, frames, h, w) ? (size_t)s3_grid(batch * (h / S3_R)) * S3W_PART * sizeof(float) : 0;
}

extern "C" int qt_conv3d_c32_wgrad(int dtype, const void* x, int x_channels, const void* dy, float* dweight, void* workspace,
                                   size_t workspace_bytes, int batch, int frames, int h, int w, void* stream) {
  QT_CHECK_ARG(x && dy && dweight && batch > 0 && frames > 0 && h > 0 && w > 0, "qt_conv3d_c32_wgrad: bad argument");
  QT_CHECK_ARG(x_channels >= 32 && x_channels % 8 == 0, "qt_conv3d_c32_wgrad: x rows of %d channels", x_channels);
  const size_t need = qt_conv3d_c32_wgrad_workspace_bytes(batch, frames, h, w);
  if (dtype != QT_BF16 || need == 0 || ((uintptr_t)x % 16) != 0 || ((uintptr_t)dy % 16) != 0) {
    qt_set_error("qt_conv3d_c32_wgrad: bf16, even H, W %% 16 == 0, W <= 128, 16-byte aligned operands only (use qt_conv2d_wgrad per "
                 "frame tap)");
    return QT_ERR_UNSUPPORTED;
  }
  QT_CHECK_ARG(workspace && workspace_bytes >= need, "qt_conv3d_c32_wgrad: workspace of %zu bytes, %zu needed", workspace_bytes, need);
  S3WArgs a;
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.part = (float*)workspace;
  a.B = batch; a.T = frames; a.H = h; a.W = w; a.xc = x_channels; a.items = batch * (h / S3_R);
  const int grid = s3_grid(a.items);
  const int lds = S3_RING * (S3_R + 2) * (w + 2) * 64 + S3W_NCH * 4096;
  static std::atomic<unsigned long long> done{0};
  if (int rc = qt_raise_lds_limit((const void*)conv3d_c32_wgrad_kernel, lds, done)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(conv3d_c32_wgrad_kernel, dim3(grid), dim3(512), lds, s, a);
  QT_CHECK_LAUNCH();
  hipLaunchKernelGGL(s3_wgrad_sum_kernel, dim3((64 * 32 * 27 + 31) / 32), dim3(256), 0, s, (const float*)workspace, dweight, grid);
  QT_CHECK_LAUNCH();
  return QT_OK;
}
