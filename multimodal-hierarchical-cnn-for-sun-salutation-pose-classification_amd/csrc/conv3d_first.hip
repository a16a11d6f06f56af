// First Conv3d of the clip models (3 -> 32 channels, 3x3x3, pad 1) straight from the f32 clip, bf16 MFMA.
//
// Replaces nn.Conv3d(3, 32, kernel_size=(3,3,3), padding=(1,1,1)) of conv3d_block1
// (/root/reference/3dcnn/models.py:108, reached from Quadtree3DCNN.forward, models.py:189-193, after the permute to
// B,C,T,H,W) and its weight gradient (loss.backward(), /root/reference/3dcnn/train_3D_Quadtree_cnn_model.py).
//
// Rounds 1-3 packed the clip into one 128-wide K row per pixel (qt_pack_clip27: 27 taps x 3 channels, 256 B per pixel,
// 3.3 GB for 32 clips x 8 frames of 224 x 224) and ran a 1x1 convolution over it: 1.8 ms to write the rows, 1.6 ms to read
// them back, and the output padded to 64 channels.  Here a workgroup walks the frames of one (clip, 4-row slab): every input
// frame slab is read once from the f32 clip (1.5x with the row halo), converted to bf16 and kept in LDS as [6 rows][W + 4
// pixels][4 channels] for the three output frames that use it.  The im2col overlap is expressed by overlapping LDS reads:
// the K-chunk of pixel w for (frame tap kt, row tap kh) is the 24 bytes at pixel w - 1 of row h + kh - 1 of frame t + kt - 1;
// one MFMA k-step of 32 covers two (kt, kh) rows (4 pixels x 4 channels each, the filter is zero on the fourth pixel and
// the fourth channel), five k-steps cover the nine rows.  The filter (10 fragments) lives in registers.  The output
// channel of MFMA row 4q + r of block cb is 8q + 4cb + r, so a lane holds 8 consecutive channels of its pixel and a wave
// writes one contiguous 1 KB run per 16 pixels: no transpose.  y is [T][B][H][W][32]: no channel padding.
// Bound: the 0.8 GB output write (HBM); 10 MFMAs per 16 pixels are 50 us of matrix pipe for the whole clip batch.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "qt_common.h"

namespace {

constexpr int C3_R = 4;          // output rows per slab (one per wave)
constexpr int C3_SLABS = 5;      // four frame slabs in rotation + one that stays zero (frames outside the clip)

struct C3Args {
  const float* x;         // [B][T][3][H][W]
  const bf16_t* w;        // [>= 32][128]: element ((kt*3 + kh)*3 + kw)*3 + c (qt_pack_conv3d_block, first = 1)
  bf16_t* y;              // [T][B][H][W][32]
  const float* scale;     // nullable: y = conv * scale + shift (+ ReLU)
  const float* shift;
  float* stats;           // [gridDim.x][2][64] or NULL (channels 32..63 written as zero)
  int relu, B, T, H, W, items;
  int pc;                 // POOL: channels per pooled row (64: channels 32..63 written as zeros; 32: none)
};

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  const bf16_t x = (bf16_t)a, y = (bf16_t)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}

// POOL (eval, needs AFF): MaxPool3d((1,2,2)) of relu(conv * scale + shift) in registers -- a wave owns a row PAIR of the slab
// and every second block of 16 pixels; vertical maximum between its two accumulator sets, horizontal between neighbour lanes
// (DPP); only the pooled map [T][B][H/2][W/2][64] is written (channels 32..63 zero: the rows block 2 reads), y never exists.
template <bool AFF, bool STATS, bool POOL = false>
__global__ __launch_bounds__(256, 2) void conv3d_first_kernel(C3Args p) {
  static_assert(!POOL || (AFF && !STATS), "the pooled form is the eval form");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int W = p.W, H = p.H, T = p.T;
  const int rowb = (W + 4) * 8, slab = (C3_R + 2) * rowb;

  for (int i = tid * 16; i < C3_SLABS * slab; i += 256 * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);

  // filter fragments (A operand): MFMA row li -> channel 8*(li >> 2) + 4*cb + (li & 3); k-group lg of k-step s: the tap row
  // rr = 2s + (lg >> 1) = kt*3 + kh, pixels 2*(lg & 1) + {0, 1} of its four, 4 channel slots each
  uint4 wf[5][2];
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int ch = (li >> 2) * 8 + cb * 4 + (li & 3);
      const int rr = 2 * s + (lg >> 1);
      unsigned short e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int dp = 2 * (lg & 1) + (j >> 2), c = j & 3;
        const bool ok = rr < 9 && dp < 3 && c < 3;
        const bf16_t v = ok ? p.w[(size_t)ch * 128 + (rr * 3 + dp) * 3 + c] : (bf16_t)0.f;
        e[j] = __builtin_bit_cast(unsigned short, v);
      }
      wf[s][cb] = make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16),
                             e[6] | ((unsigned)e[7] << 16));
    }

  float sc[8], sh[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = AFF ? p.scale[lg * 8 + j] : 1.f;
    sh[j] = AFF ? p.shift[lg * 8 + j] : 0.f;
    s1[j] = s2[j] = 0.f;
  }

  // staging: a frame slab is 6 rows x W/4 groups of four pixels; a thread owns at most two groups
  const int quads = W >> 2, nslot = (C3_R + 2) * quads;
  int sr[2], sq[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int sidx = tid + j * 256;
    sr[j] = sidx / quads;
    sq[j] = sidx - sr[j] * quads;
  }
  float4 pre[2][3];
  const size_t plane = (size_t)H * W;
  auto stage = [&](int b, int f, int h0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int hh = h0 - 1 + sr[j];
      const bool ok = tid + j * 256 < nslot && (unsigned)hh < (unsigned)H;
      const float* src = p.x + ((size_t)(b * T + f) * 3) * plane + (size_t)(ok ? hh : 0) * W + (ok ? sq[j] : 0) * 4;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        pre[j][c] = ok ? *reinterpret_cast<const float4*>(src + c * plane) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit = [&](int f) {
    unsigned char* base = smem + (f & 3) * slab;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (tid + j * 256 < nslot) {
        uint2* dst = reinterpret_cast<uint2*>(base + sr[j] * rowb + (1 + sq[j] * 4) * 8);
        dst[0] = make_uint2(pack_bf16x2(pre[j][0].x, pre[j][1].x), pack_bf16x2(pre[j][2].x, 0.f));
        dst[1] = make_uint2(pack_bf16x2(pre[j][0].y, pre[j][1].y), pack_bf16x2(pre[j][2].y, 0.f));
        dst[2] = make_uint2(pack_bf16x2(pre[j][0].z, pre[j][1].z), pack_bf16x2(pre[j][2].z, 0.f));
        dst[3] = make_uint2(pack_bf16x2(pre[j][0].w, pre[j][1].w), pack_bf16x2(pre[j][2].w, 0.f));
      }
  };

  const int slabs_per_img = H / C3_R, nblk = W >> 4;
  for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
    const int b = item / slabs_per_img, h0 = (item - b * slabs_per_img) * C3_R;
    __syncthreads();                    // the previous item's last frames are done with their slabs
    stage(b, 0, h0);
    commit(0);
    if (T > 1) stage(b, 1, h0);
    for (int t = 0; t < T; ++t) {
      if (t + 1 < T) commit(t + 1);
      __syncthreads();                  // frame t + 1 is visible; every wave finished frame t - 1: slab (t + 2) & 3 is free
      if (t + 2 < T) stage(b, t + 2, h0);

      const unsigned char* a[5];
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int rr = 2 * s + (lg >> 1);
        const int kt = rr / 3, kh = rr - kt * 3;
        const int f = t + kt - 1;
        const int sl = (rr < 9 && (unsigned)f < (unsigned)T) ? (f & 3) : 4;
        a[s] = smem + sl * slab + (wave + (rr < 9 ? kh : 0)) * rowb + (li + 2 * (lg & 1)) * 8;
      }
      if (POOL) {
        const int pr = wave & 1;   // row pair of the slab; blocks (wave >> 1), (wave >> 1) + 2, ...
        // the addresses above are for slab row `wave`: move them to row 2 * pr
        const int back = (wave - 2 * pr) * rowb;
        const int pc = p.pc;
        bf16_t* orow = p.y + ((((size_t)t * p.B + b) * (H >> 1) + (h0 >> 1) + pr) * (W >> 1) + (li >> 1)) * pc +
                       ((li & 1) ? 32 : 0) + lg * 8;
        const bool writes = pc > 32 || !(li & 1);   // (32-channel rows have no zero half for the odd lanes to write)
        for (int blk = wave >> 1; blk < nblk; blk += 2) {
          f32x4 acc[2][2];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            uint4 xf[5];
#pragma unroll
            for (int s = 0; s < 5; ++s) {
              const unsigned char* q = a[s] - back + r * rowb + blk * 128;
              const uint2 lo = *reinterpret_cast<const uint2*>(q);
              const uint2 hi = *reinterpret_cast<const uint2*>(q + 8);
              xf[s] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
            acc[r][0] = acc[r][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 5; ++s)
#pragma unroll
              for (int cb = 0; cb < 2; ++cb)
                acc[r][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[s][cb]),
                                                                     __builtin_bit_cast(bf16x8, xf[s]), acc[r][cb], 0, 0, 0);
          }
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float v0 = fmaxf(acc[0][j >> 2][j & 3] * sc[j] + sh[j], 0.f);
            const float v1 = fmaxf(acc[1][j >> 2][j & 3] * sc[j] + sh[j], 0.f);
            const float m = fmaxf(v0, v1);
            const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xb1, 0xf, 0xf, false));
            v[j] = (li & 1) ? 0.f : fmaxf(m, o);   // odd lanes write the zero half of their left neighbour's pooled row
          }
          if (writes)
            *reinterpret_cast<uint4*>(orow + (size_t)blk * 8 * pc) =
                make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        }
        continue;
      }
      bf16_t* yrow = p.y + ((((size_t)t * p.B + b) * H + h0 + wave) * W + li) * 32 + lg * 8;
#pragma unroll 2
      for (int blk = 0; blk < nblk; ++blk) {
        uint4 xf[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          const uint2 lo = *reinterpret_cast<const uint2*>(a[s] + blk * 128);
          const uint2 hi = *reinterpret_cast<const uint2*>(a[s] + blk * 128 + 8);
          xf[s] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        f32x4 acc[2];
        acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[s][cb]),
                                                              __builtin_bit_cast(bf16x8, xf[s]), acc[cb], 0, 0, 0);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float acv = acc[j >> 2][j & 3];
          if (STATS) {
            s1[j] += acv;
            s2[j] += acv * acv;
          }
          v[j] = acv;
          if (AFF) {
            v[j] = acv * sc[j] + sh[j];
            if (p.relu) v[j] = fmaxf(v[j], 0.f);
          }
        }
        *reinterpret_cast<uint4*>(yrow + (size_t)blk * 16 * 32) =
            make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
      }
    }
  }

  if (STATS) {
    // lanes of a 16-lane row hold different pixels of the same 8 channels; then the four waves through LDS, fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][2][32]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a1 = qt_row16_sum(s1[j]), a2 = qt_row16_sum(s2[j]);
      if (li == 0) {
        red[(wave * 2 + 0) * 32 + lg * 8 + j] = a1;
        red[(wave * 2 + 1) * 32 + lg * 8 + j] = a2;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, ch = tid & 63;
      float v = 0.f;
      if (ch < 32) v = (red[(0 * 2 + which) * 32 + ch] + red[(1 * 2 + which) * 32 + ch]) +
                       (red[(2 * 2 + which) * 32 + ch] + red[(3 * 2 + which) * 32 + ch]);
      p.stats[((size_t)blockIdx.x * 2 + which) * 64 + ch] = v;
    }
  }
}

// ---- weight gradient of the same layer from the f32 clip and d(loss)/dy [T][B][H][W][32] ---------------------------------
//   dW[ch][kt][kh][kw][c] = sum over positions of dy[pos][ch] * x[t + kt - 1][c][h + kh - 1][w + kw - 1]
// The contraction runs over positions (MFMA k = 32 consecutive pixels of a row).  Same walk and the same bf16 frame slabs as the
// forward; a wave owns one of the slab's four rows and streams its dy row through a private LDS ring by LDS-DMA (2 KB =
// 32 positions per chunk, two chunks ahead, counted vmcnt waits, no barrier).  Both operands are position-major in LDS and
// come out k-major through ds_read_b64_tr_b16: dy as 16 channels x 32 positions, the clip as (4 window pixels x 4 channel
// slots) x 32 positions per tap row -- the window of position w starts at staged pixel w, so the im2col overlap is again
// only an address (lane (row, pp) reads the 8 bytes of pixel w0 + row + pp).  18 accumulator tiles per wave:
// 2 channel blocks x 9 tap rows, 22 transposing reads per 18 MFMAs.  Every workgroup writes its partial filter
// [32][9][16]; c3_wgrad_sum_kernel adds them in a fixed order straight into the reference's [32][3][3][3][3] layout:
// deterministic, no atomics, no unpack launch.
constexpr int C3W_D = 2;             // dy chunks in flight ahead of the one being multiplied
constexpr int C3W_NCH = C3W_D + 1;   // ring slots per wave
constexpr int C3W_PART = 32 * 9 * 16;

struct C3WArgs {
  const float* x;         // [B][T][3][H][W]
  const bf16_t* dy;       // [T][B][H][W][32]                     (FUSED: the raw conv output y, same layout)
  float* part;            // [gridDim.x][32][9][16]
  int B, T, H, W, items;
  // FUSED: d(loss)/dy is formed on the way in from the pooled side (MaxPool3d((1,2,2)) + ReLU + BatchNorm3d backward)
  const bf16_t* dout;          // [T][B][H/2][W/2][cp]: gradient of the pooled map
  const unsigned char* arg;    // same shape: window position of the maximum
  const float *mean, *invstd, *scale, *shift, *coef;   // [32] each; coef [3][cp] = qt_bn_bwd_finalize's (a, b, c)
  int cp;
};

// FUSED (round 4): the kernel reads the raw conv output y where it read dy and forms
//   dy = a (g - b - xhat c),   g = dout at the window's argmax where relu(bn(y)) > 0, else 0
// in registers (what qt_pool3d_bn_bwd_apply wrote to memory and this kernel read back: 2 x 0.8 GB at 32 clips x 8 frames of
// 224 x 224).  A lane owns 8 channels of two positions of a chunk -- the 16 bytes LDS-DMA dropped at lane * 16 and
// 1024 + lane * 16 -- so the chunk's LDS image is the same and the contraction below does not change.  The loads of a chunk
// (y 2 x 16 B, dout 2 x 16 B, argmax 2 x 8 B per lane) are issued two chunks ahead into registers.
struct C3WChunk {
  uint4 y0, y1, d0, d1;
  uint2 a0, a1;
};

template <bool FUSED>
__global__ __launch_bounds__(256, 2) void conv3d_first_wgrad_kernel(C3WArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int lrow = 4 * lg + (li >> 2), pp = li & 3;
  const int W = p.W, H = p.H, T = p.T;
  const int rowb = (W + 4) * 8, slab = (C3_R + 2) * rowb;
  const unsigned smem_base = lds_addr_of(smem);
  const unsigned ring = smem_base + C3_SLABS * slab + wave * (C3W_NCH * 2048);

  for (int i = tid * 16; i < C3_SLABS * slab; i += 256 * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);

  f32x4 acc[2][9];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int rr = 0; rr < 9; ++rr) acc[cb][rr] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int quads = W >> 2, nslot = (C3_R + 2) * quads;
  int sr[2], sq[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int sidx = tid + j * 256;
    sr[j] = sidx / quads;
    sq[j] = sidx - sr[j] * quads;
  }
  float4 pre[2][3];
  const size_t plane = (size_t)H * W;
  auto stage = [&](int b, int f, int h0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int hh = h0 - 1 + sr[j];
      const bool ok = tid + j * 256 < nslot && (unsigned)hh < (unsigned)H;
      const float* src = p.x + ((size_t)(b * T + f) * 3) * plane + (size_t)(ok ? hh : 0) * W + (ok ? sq[j] : 0) * 4;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        pre[j][c] = ok ? *reinterpret_cast<const float4*>(src + c * plane) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit = [&](int f) {
    unsigned char* base = smem + (f & 3) * slab;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (tid + j * 256 < nslot) {
        uint2* dst = reinterpret_cast<uint2*>(base + sr[j] * rowb + (1 + sq[j] * 4) * 8);
        dst[0] = make_uint2(pack_bf16x2(pre[j][0].x, pre[j][1].x), pack_bf16x2(pre[j][2].x, 0.f));
        dst[1] = make_uint2(pack_bf16x2(pre[j][0].y, pre[j][1].y), pack_bf16x2(pre[j][2].y, 0.f));
        dst[2] = make_uint2(pack_bf16x2(pre[j][0].z, pre[j][1].z), pack_bf16x2(pre[j][2].z, 0.f));
        dst[3] = make_uint2(pack_bf16x2(pre[j][0].w, pre[j][1].w), pack_bf16x2(pre[j][2].w, 0.f));
      }
  };

  // the wave's stream of dy chunks: (item, frame, chunk of 32 pixels of row h0 + wave), issued C3W_D ahead of use
  const int slabs_per_img = H / C3_R, nck = W >> 5;
  int is_item = blockIdx.x, is_t = 0, is_c = 0;   // next chunk to issue
  int n_issued = 0, n_used = 0;
  // FUSED: per-lane constants of the lane's 8 channels: dy = ka g + kb - y kc  (= a (g - b - (y - mean) invstd c))
  float ka[8], kb[8], kc[8], ksc[8], ksh[8];
  const unsigned me = (unsigned)((wave & 1) * 2 + ((lane >> 2) & 1));   // the position's place in its 2 x 2 window (h0 % 4 == 0)
  if (FUSED) {
    const int cg = (lane & 3) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = p.coef[cg + j], bb = p.coef[p.cp + cg + j], cc = p.coef[2 * p.cp + cg + j];
      const float mu = p.mean[cg + j], is = p.invstd[cg + j];
      ka[j] = a;
      kc[j] = a * is * cc;
      kb[j] = a * (mu * is * cc - bb);
      ksc[j] = p.scale[cg + j];
      ksh[j] = p.shift[cg + j];
    }
  }
  C3WChunk r0, r1, rn;   // FUSED: chunks n_used, n_used + 1 and the one being requested
  r0 = r1 = rn = C3WChunk{};
  auto issue = [&]() {
    if (is_item >= p.items) return;
    const int b = is_item / slabs_per_img, h0 = (is_item - b * slabs_per_img) * C3_R;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(
        p.dy + ((((size_t)is_t * p.B + b) * H + h0 + wave) * W + is_c * 32) * 32);
    if (FUSED) {
      rn.y0 = *reinterpret_cast<const uint4*>(src + lane * 16);
      rn.y1 = *reinterpret_cast<const uint4*>(src + 1024 + lane * 16);
      const size_t cell = (((size_t)is_t * p.B + b) * (H >> 1) + ((h0 + wave) >> 1)) * (W >> 1) + is_c * 16 + (lane >> 3);
      const size_t e0 = cell * p.cp + (lane & 3) * 8, e1 = e0 + (size_t)8 * p.cp;
      rn.d0 = *reinterpret_cast<const uint4*>(p.dout + e0);
      rn.d1 = *reinterpret_cast<const uint4*>(p.dout + e1);
      rn.a0 = *reinterpret_cast<const uint2*>(p.arg + e0);
      rn.a1 = *reinterpret_cast<const uint2*>(p.arg + e1);
    } else {
      const unsigned dst = ring + (unsigned)(n_issued % C3W_NCH) * 2048u;
      glds16(src + lane * 16, dst);
      glds16(src + 1024 + lane * 16, dst + 1024);
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
  static_assert(C3W_D == 2, "two chunks ahead: the register sets r0, r1 of the fused form");
  issue();
  r0 = rn;
  issue();
  r1 = rn;
  // eight channels of one position: bf16 y, dout (16 B each), eight argmax bytes -> eight bf16 dy
  auto form_dy = [&](const uint4& yv, const uint4& dv, const uint2& av) -> uint4 {
    const unsigned yw[4] = {yv.x, yv.y, yv.z, yv.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
    unsigned o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float r[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = 2 * q + u;
        const float yf = __builtin_bit_cast(float, u ? (yw[q] & 0xffff0000u) : (yw[q] << 16));
        const float df = __builtin_bit_cast(float, u ? (dw[q] & 0xffff0000u) : (dw[q] << 16));
        const unsigned a = ((j < 4 ? av.x : av.y) >> (8 * (j & 3))) & 0xffu;
        const float g = (a == me && yf * ksc[j] + ksh[j] > 0.f) ? df : 0.f;
        r[u] = ka[j] * g + kb[j] - yf * kc[j];
      }
      o[q] = pack_bf16x2(r[0], r[1]);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
  };

  for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
    const int b = item / slabs_per_img, h0 = (item - b * slabs_per_img) * C3_R;
    __syncthreads();
    stage(b, 0, h0);
    commit(0);
    if (T > 1) stage(b, 1, h0);
    for (int t = 0; t < T; ++t) {
      if (t + 1 < T) commit(t + 1);
      __syncthreads();
      if (t + 2 < T) stage(b, t + 2, h0);

      unsigned xa[9];   // tap row rr: pixel (lrow + pp) of row wave + kh of frame t + kt - 1 (the zero slab outside the clip)
#pragma unroll
      for (int rr = 0; rr < 9; ++rr) {
        const int kt = rr / 3, kh = rr - kt * 3;
        const int f = t + kt - 1;
        const int sl = (unsigned)f < (unsigned)T ? (f & 3) : 4;
        xa[rr] = smem_base + sl * slab + (wave + kh) * rowb + (lrow + pp) * 8;
      }
      for (int c = 0; c < nck; ++c) {
        issue();
        if (FUSED) {   // (ordinary loads: the compiler counts them)
          unsigned char* slot = smem + (ring - smem_base) + (unsigned)(n_used % C3W_NCH) * 2048u;
          *reinterpret_cast<uint4*>(slot + lane * 16) = form_dy(r0.y0, r0.d0, r0.a0);
          *reinterpret_cast<uint4*>(slot + 1024 + lane * 16) = form_dy(r0.y1, r0.d1, r0.a1);
          r0 = r1;
          r1 = rn;
        } else {
          const int ahead = n_issued - n_used - 1;   // chunks issued after the one about to be read
          if (ahead >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else if (ahead == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned da = ring + (unsigned)(n_used % C3W_NCH) * 2048u + lrow * 64 + pp * 8;
        ++n_used;
        uint4 fa[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS_AS s16x4*)(size_t)(da + cb * 32));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS_AS s16x4*)(size_t)(da + cb * 32 + 1024));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          fa[cb] = make_uint4(l2.x, l2.y, h2.x, h2.y);
        }
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS_AS s16x4*)(size_t)(xa[rr] + c * 256));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS_AS s16x4*)(size_t)(xa[rr] + c * 256 + 128));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          const uint4 fb = make_uint4(l2.x, l2.y, h2.x, h2.y);
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
            acc[cb][rr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[cb]),
                                                                  __builtin_bit_cast(bf16x8, fb), acc[cb][rr], 0, 0, 0);
        }
      }
    }
  }

  // the four waves' filters through LDS in a fixed order; lane: channels 16 cb + 4 lg + r, column li of tap row rr
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);   // [4][C3W_PART]
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int rr = 0; rr < 9; ++rr)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave * C3W_PART + ((cb * 16 + lg * 4 + r) * 9 + rr) * 16 + li] = acc[cb][rr][r];
  __syncthreads();
  for (int i = tid; i < C3W_PART; i += 256)
    p.part[(size_t)blockIdx.x * C3W_PART + i] = (red[i] + red[C3W_PART + i]) + (red[2 * C3W_PART + i] + red[3 * C3W_PART + i]);
}

// dW [32][3 c][3 kt][3 kh][3 kw] (nn.Conv3d weight layout) = sum over the workgroups' partial filters, ascending order
__global__ __launch_bounds__(256) void c3_wgrad_sum_kernel(const float* __restrict__ part, float* __restrict__ dw, int nparts) {
  __shared__ float red[8][32];
  const int col = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int o = blockIdx.x * 32 + col;     // (ch, c, kt, kh, kw)
  float s = 0.f;
  if (o < 32 * 81) {
    const int ch = o / 81, rem = o - ch * 81;
    const int c = rem / 27, tap = rem - c * 27;      // tap = (kt*3 + kh)*3 + kw
    const int rr = tap / 3, kw = tap - rr * 3;
    const float* src = part + (ch * 9 + rr) * 16 + kw * 4 + c;
    const int per = (nparts + 7) / 8;
    const int beg = sg * per, end = min(nparts, beg + per);
#pragma unroll 8
    for (int k = beg; k < end; ++k) s += src[(size_t)k * C3W_PART];
  }
  red[sg][col] = s;
  __syncthreads();
  if (sg == 0 && o < 32 * 81) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][col];
    dw[o] = t;
  }
}

int c3_grid(int items) {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t pr;
      if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n = pr.multiProcessorCount;
    }
    return n;
  }();
  return items < 2 * cus ? items : 2 * cus;
}

bool c3_enabled() {
  static const bool on = [] {
    const char* e = getenv("QTCNN_CONV3D_FIRST");
    return !(e && e[0] == '0');
  }();
  return on;
}

template <bool FUSED>
int c3_wgrad_launch(const C3WArgs& a, float* dweight, void* stream) {
  const int grid = c3_grid(a.items);
  int lds = C3_SLABS * (C3_R + 2) * (a.W + 4) * 8 + 4 * C3W_NCH * 2048;
  if (lds < 4 * C3W_PART * 4) lds = 4 * C3W_PART * 4;   // the four waves' filters at the end
  static std::atomic<unsigned long long> done{0};
  int rc = qt_raise_lds_limit((const void*)conv3d_first_wgrad_kernel<FUSED>, lds, done);
  if (rc != QT_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(conv3d_first_wgrad_kernel<FUSED>, dim3(grid), dim3(256), lds, s, a);
  QT_CHECK_LAUNCH();
  hipLaunchKernelGGL(c3_wgrad_sum_kernel, dim3((32 * 81 + 31) / 32), dim3(256), 0, s, (const float*)a.part, dweight, grid);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

bool c3_shape_ok(int dtype, const void* clips, int batch, int frames, int h, int w) {
  return dtype == QT_BF16 && c3_enabled() && batch > 0 && frames > 0 && h >= C3_R && h % C3_R == 0 && w >= 16 && w % 16 == 0 &&
         w <= 256 && ((uintptr_t)clips % 16) == 0;
}

}  // namespace

extern "C" size_t qt_conv3d_first_wgrad_workspace_bytes(int batch, int frames, int h, int w) {
  if (!c3_enabled() || batch <= 0 || frames <= 0 || h < C3_R || h % C3_R || w < 32 || w % 32 || w > 256) return 0;
  return (size_t)c3_grid(batch * (h / C3_R)) * C3W_PART * sizeof(float);
}

extern "C" int qt_conv3d_first_wgrad(int dtype, const float* clips, const void* dy, float* dweight, void* workspace,
                                     size_t workspace_bytes, int batch, int frames, int h, int w, void* stream) {
  QT_CHECK_ARG(clips && dy && dweight && batch > 0 && frames > 0 && h > 0 && w > 0, "qt_conv3d_first_wgrad: bad argument");
  const size_t need = qt_conv3d_first_wgrad_workspace_bytes(batch, frames, h, w);
  if (dtype != QT_BF16 || need == 0 || ((uintptr_t)clips % 16) != 0 || ((uintptr_t)dy % 16) != 0) {
    qt_set_error("qt_conv3d_first_wgrad: bf16, H %% 4 == 0, W %% 32 == 0, W <= 256, 16-byte aligned operands only "
                 "(use qt_pack_clip27 + qt_conv2d_wgrad)");
    return QT_ERR_UNSUPPORTED;
  }
  QT_CHECK_ARG(workspace && workspace_bytes >= need, "qt_conv3d_first_wgrad: workspace of %zu bytes, %zu needed", workspace_bytes,
               need);
  C3WArgs a;
  memset(&a, 0, sizeof(a));
  a.x = clips; a.dy = (const bf16_t*)dy; a.part = (float*)workspace;
  a.B = batch; a.T = frames; a.H = h; a.W = w; a.items = batch * (h / C3_R);
  return c3_wgrad_launch<false>(a, dweight, stream);
}

// The same weight gradient with d(loss)/dy formed on the way in (round 4): y = the raw conv output [T][B][H][W][32] of
// qt_conv3d_first_fwd (statistics form), dout / argmax [T][B][H/2][W/2][pooled_channels] = the gradient of conv3d_block1's
// pooled map and qt_pool3d_bn_relu_max's argmax (pool_t = 1), mean / invstd / scale / shift = qt_bn_finalize's vectors of the
// block's BatchNorm3d, coef = qt_bn_bwd_finalize's [3][pooled_channels].  Same result as qt_pool3d_bn_bwd_apply (dy_channels
// = 32) followed by qt_conv3d_first_wgrad, up to the rounding of the folded dy expression; dy never reaches memory.
extern "C" int qt_conv3d_first_wgrad_fused(int dtype, const float* clips, const void* y, const void* dout,
                                           const unsigned char* argmax, int pooled_channels, const float* mean, const float* invstd,
                                           const float* scale, const float* shift, const float* coef, float* dweight, void* workspace,
                                           size_t workspace_bytes, int batch, int frames, int h, int w, void* stream) {
  QT_CHECK_ARG(clips && y && dout && argmax && mean && invstd && scale && shift && coef && dweight && batch > 0 && frames > 0 &&
                   h > 0 && w > 0,
               "qt_conv3d_first_wgrad_fused: bad argument");
  QT_CHECK_ARG(pooled_channels == 32 || pooled_channels == 64, "qt_conv3d_first_wgrad_fused: pooled rows of %d channels (32 or 64)",
               pooled_channels);
  const size_t need = qt_conv3d_first_wgrad_workspace_bytes(batch, frames, h, w);
  if (dtype != QT_BF16 || need == 0 || ((uintptr_t)clips % 16) != 0 || ((uintptr_t)y % 16) != 0 || ((uintptr_t)dout % 16) != 0 ||
      ((uintptr_t)argmax % 8) != 0) {
    qt_set_error("qt_conv3d_first_wgrad_fused: bf16, H %% 4 == 0, W %% 32 == 0, W <= 256, 16-byte aligned operands only "
                 "(use qt_pool3d_bn_bwd_apply + qt_conv3d_first_wgrad)");
    return QT_ERR_UNSUPPORTED;
  }
  QT_CHECK_ARG(workspace && workspace_bytes >= need, "qt_conv3d_first_wgrad_fused: workspace of %zu bytes, %zu needed",
               workspace_bytes, need);
  C3WArgs a;
  memset(&a, 0, sizeof(a));
  a.x = clips; a.dy = (const bf16_t*)y; a.part = (float*)workspace;
  a.B = batch; a.T = frames; a.H = h; a.W = w; a.items = batch * (h / C3_R);
  a.dout = (const bf16_t*)dout; a.arg = argmax; a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift; a.coef = coef;
  a.cp = pooled_channels;
  return c3_wgrad_launch<true>(a, dweight, stream);
}

// Eval forward of conv3d_block1 in one launch: Conv3d + folded BatchNorm3d (scale / shift of 32 channels, the conv bias folded
// into shift) + ReLU + MaxPool3d((1,2,2)); pooled [T][B][H/2][W/2][pooled_channels] (64: channels 32..63 zero).  Shapes as
// qt_conv3d_first_fwd.
extern "C" int qt_conv3d_first_fwd_pool(int dtype, const float* clips, const void* w_packed, void* pooled, int pooled_channels,
                                        const float* scale, const float* shift, int batch, int frames, int h, int w, void* stream) {
  QT_CHECK_ARG(clips && w_packed && pooled && scale && shift && batch > 0 && frames > 0 && h > 0 && w > 0,
               "qt_conv3d_first_fwd_pool: bad argument");
  QT_CHECK_ARG(pooled_channels == 32 || pooled_channels == 64, "qt_conv3d_first_fwd_pool: pooled rows of %d channels (32 or 64)",
               pooled_channels);
  if (!c3_shape_ok(dtype, clips, batch, frames, h, w)) {
    qt_set_error("qt_conv3d_first_fwd_pool: bf16, H %% 4 == 0, W %% 16 == 0, W <= 256 and a 16-byte aligned clip only "
                 "(use qt_conv3d_first_fwd / qt_pack_clip27 + qt_conv2d_igemm, then qt_pool3d_max)");
    return QT_ERR_UNSUPPORTED;
  }
  C3Args a;
  a.x = clips; a.w = (const bf16_t*)w_packed; a.y = (bf16_t*)pooled; a.scale = scale; a.shift = shift; a.stats = nullptr;
  a.relu = 1; a.B = batch; a.T = frames; a.H = h; a.W = w; a.items = batch * (h / C3_R); a.pc = pooled_channels;
  const int lds = C3_SLABS * (C3_R + 2) * (w + 4) * 8;
  static std::atomic<unsigned long long> done{0};
  if (int rc = qt_raise_lds_limit((const void*)conv3d_first_kernel<true, false, true>, lds, done)) return rc;
  hipLaunchKernelGGL((conv3d_first_kernel<true, false, true>), dim3(c3_grid(a.items)), dim3(256), lds,
                     static_cast<hipStream_t>(stream), a);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

// rows of BatchNorm partial sums qt_conv3d_first_fwd writes ([rows][2][64]), 0 = shape not covered
extern "C" int qt_conv3d_first_stats_rows(int batch, int frames, int h, int w) {
  if (batch <= 0 || frames <= 0 || h < C3_R || h % C3_R || w < 16 || w % 16 || w > 256) return 0;
  return c3_grid(batch * (h / C3_R));
}

extern "C" int qt_conv3d_first_fwd(int dtype, const float* clips, const void* w_packed, void* y, const float* scale,
                                   const float* shift, int relu, float* stats, int batch, int frames, int h, int w,
                                   void* stream) {
  QT_CHECK_ARG(clips && w_packed && y && batch > 0 && frames > 0 && h > 0 && w > 0, "qt_conv3d_first_fwd: bad argument");
  QT_CHECK_ARG(!(scale && stats), "qt_conv3d_first_fwd: scale / shift and statistics are exclusive");
  QT_CHECK_ARG(!scale || shift, "qt_conv3d_first_fwd: scale without shift");
  if (!c3_shape_ok(dtype, clips, batch, frames, h, w)) {
    qt_set_error("qt_conv3d_first_fwd: bf16, H %% 4 == 0, W %% 16 == 0, W <= 256 and a 16-byte aligned clip only "
                 "(use qt_pack_clip27 + qt_conv2d_igemm)");
    return QT_ERR_UNSUPPORTED;
  }
  C3Args a;
  a.x = clips; a.w = (const bf16_t*)w_packed; a.y = (bf16_t*)y; a.scale = scale; a.shift = shift; a.stats = stats;
  a.relu = relu; a.B = batch; a.T = frames; a.H = h; a.W = w; a.items = batch * (h / C3_R); a.pc = 64;
  const int lds = C3_SLABS * (C3_R + 2) * (w + 4) * 8;
  const dim3 grid(c3_grid(a.items)), blk(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = QT_OK;
  if (scale) {
    static std::atomic<unsigned long long> done{0};
    if ((rc = qt_raise_lds_limit((const void*)conv3d_first_kernel<true, false>, lds, done)) != QT_OK) return rc;
    hipLaunchKernelGGL((conv3d_first_kernel<true, false>), grid, blk, lds, s, a);
  } else if (stats) {
    static std::atomic<unsigned long long> done{0};
    if ((rc = qt_raise_lds_limit((const void*)conv3d_first_kernel<false, true>, lds, done)) != QT_OK) return rc;
    hipLaunchKernelGGL((conv3d_first_kernel<false, true>), grid, blk, lds, s, a);
  } else {
    static std::atomic<unsigned long long> done{0};
    if ((rc = qt_raise_lds_limit((const void*)conv3d_first_kernel<false, false>, lds, done)) != QT_OK) return rc;
    hipLaunchKernelGGL((conv3d_first_kernel<false, false>), grid, blk, lds, s, a);
  }
  QT_CHECK_LAUNCH();
  return QT_OK;
}
