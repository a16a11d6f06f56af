// 3x3 / stride 1 / pad 1 convolution of ResNet layer1 (56x56 map, 64 -> 64 channels, bf16; forward and data gradient)
// with the input window resident in LDS: every input pixel is fetched from L2 ONCE per workgroup and reused by all nine
// taps, instead of once per tap as in the generic implicit GEMM (conv_igemm.hip).
//
// Replaces the conv2d forward / backward-input ATen calls reached from /root/reference/Quadtree_from scratch/models.py:222-229
// (torchvision resnet18 layer1) and Quadtree_train.py:65.
//
// Formulation.  Pixels are numbered in ZERO-PADDED coordinates q = (img*(H+2) + hp)*(W+2) + wp; the border positions hold
// zeros and their outputs are discarded.  In that numbering a tap is a CONSTANT row shift:
// source(q, kh, kw) = q + (kh-1)*(W+2) + (kw-1).  Tensors in HBM stay dense NHWC: padding exists only in the tile's index
// space (7 % more positions at 56x56), so no other kernel changes.
//
// (Rounds 1-3 also carried an experimental one-tile-per-workgroup kernel for the 56x56 / 28x28 stages behind
// QTCNN_PATCH_CONV=1; it only ever tied the generic kernel -- EXPERIMENTS.md, round 1 -- and was removed in round 4.  The
// persistent ring kernel below is what the models run.)
#include <stdlib.h>

#include <type_traits>

#include "qt_common.h"

namespace {

struct PatchArgs {
  const void* src;
  const void* wgt;   // [N][9][C]
  void* dst;
  const float* scale;
  const float* shift;
  const void* residual;
  const void* relu_mask;
  const unsigned char* relu_mask_bits;   // one bit per element (qt_conv_io.relu_mask_bits)
  float* stats_partial;
  const void* bn_y[2];
  const float* bn_mean[2];
  const float* bn_invstd[2];
  float* bn_partial[2];
  int B, H, W, C, N;
  int PW, PP;          // W+2, (H+2)*(W+2)
  long long Q;         // B*PP padded positions
  int flip;            // 1: data gradient (taps mirrored)
  int relu;
  int gridM, gridN;
  FastDiv div_pp, div_pw;
};

constexpr int kRowBytes = 128;
constexpr int BM = 256;
constexpr int NT = 512;

// ---------------------------------------------------------------------------------------------
// Persistent variant for ResNet layer1 (56x56 map, 64 -> 64 channels, bf16; forward and data
// gradient).  One 8-wave workgroup per CU walks a CONTIGUOUS range of 256-position tiles:
//   * the whole 3x3x64x64 filter stays in LDS (72 KB), loaded once per workgroup;
//   * the input patch is a sliding window in a 640-row LDS ring (80 KB): consecutive tiles share
//     their halo, so a tile fetches only its 256 new positions (32 KB), and the fetch of tile k+1
//     is issued at the start of tile k (it lands behind 144 MFMAs per wave);
//   * no LDS staging in the epilogue: a lane owns 4 consecutive channels of a position and
//     stores them (8 bytes) itself; BatchNorm partial sums accumulate in registers over all tiles
//     of the workgroup and are reduced once (one partial row per workgroup);
//   * one barrier per tile.
// Per tile and CU: 32 KB from L2 for 37.7 MFLOP (1150 FLOP/B) -- the kernel is no longer bound by
// the L2->LDS path, and workgroup dispatch gaps / exposed prologues (what holds the one-tile-per-
// workgroup kernel above back) disappear.
// ---------------------------------------------------------------------------------------------
// The ring is followed by a MIRROR of its first 48 rows: a lane's four 16-row tiles of a tap then sit at rr0, rr0 + 16,
// rr0 + 32, rr0 + 48 with ONE wrap of rr0 per tap instead of one per fragment (the fragment addresses were 4.7 vector
// instructions per MFMA, PMC: 22.1 M VALU beside 3.9 M MFMAs per launch).  The statistics scratch aliases the filter.
constexpr int L1_PW = 58, L1_RING = 640, L1_MIRROR = 48, L1_WBYTES = 9 * 64 * kRowBytes;
constexpr int L1_RBYTES = (L1_RING + L1_MIRROR) * kRowBytes;
constexpr int L1_RED = 0;                               // [8 waves][2 i][4 fk][4 r][3] floats, over the (dead) filter
constexpr int L1_LDS = L1_WBYTES + L1_RBYTES;
static_assert(L1_LDS <= 160 * 1024 && 8 * 2 * 4 * 4 * 3 * 4 <= L1_WBYTES, "layer1 ring kernel LDS");

// Halo positions of a tile store their (meaningless) 16 bytes here instead of branching around the store: every wave then
// issues exactly four stores per tile, and the next tile's barrier can wait with a COUNTED vmcnt for the window fetch alone
// instead of for the acknowledgement of the tile's stores.
__device__ uint4 l1_store_sink[NT];

// OPS: the epilogue has memory operands (residual / ReLU mask / BatchNorm links), prefetched per tile.
template <bool FLIP, bool OPS>
__global__ __launch_bounds__(NT) void conv_l1_ring_kernel(PatchArgs p) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const T* __restrict__ src = static_cast<const T*>(p.src);
  const T* __restrict__ wgt = static_cast<const T*>(p.wgt);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int frow = lane & 15, fk = lane >> 4;
  const int rbase = tid >> 3, chunk = (tid & 7) ^ (rbase & 7);
  const unsigned smem_base = lds_addr_of(smem);
  const T* zero_src = reinterpret_cast<const T*>(qt_zero_page);

  // contiguous tile range of this workgroup (static split.  A dynamic queue of 4-tile chunks was
  // measured: it absorbs late-starting workgroups under the concurrent weight-gradient stream, but
  // re-priming the window per chunk and the coarser quantisation cost more: 112 vs 96 us alone,
  // 30.35 vs 30.55 k images/s end to end.)
  const int ntiles = p.gridM, G = gridDim.x, b = blockIdx.x;
  const int t0 = (int)((long long)b * ntiles / G), t1 = (int)((long long)(b + 1) * ntiles / G);
  const int nt = t1 - t0;
  if (nt <= 0) return;
  const long long qstart = (long long)t0 * BM - (L1_PW + 1);   // padded position of ring index u = 0

  // ring pass: 64 consecutive window indices [ub, ub+64), ub a multiple of 64
  auto dma_ring = [&](int ub) {
    const long long q = qstart + ub + rbase;
    const T* g = zero_src;
    if (q >= 0 && q < p.Q) {
      const unsigned uq = (unsigned)q;
      const unsigned img = fdiv(uq, p.div_pp);
      const unsigned rem = uq - img * (unsigned)p.PP;
      const unsigned hp = fdiv(rem, p.div_pw);
      const unsigned wp = rem - hp * (unsigned)L1_PW;
      if (hp >= 1 && hp <= 56u && wp >= 1 && wp <= 56u)
        g = src + (((long long)img * 56 + (hp - 1)) * 56 + (wp - 1)) * 64 + chunk * 8;
    }
    const int rrow = __builtin_amdgcn_readfirstlane(ub % L1_RING);
    glds16(g, smem_base + L1_WBYTES + (rrow + wave * 8) * kRowBytes);
    if (rrow == 0 && wave < L1_MIRROR / 8)  // rows 0..47 also behind the ring (same source, same pass)
      glds16(g, smem_base + L1_WBYTES + (L1_RING + wave * 8) * kRowBytes);
  };

  // ---- prologue: filter (LDS row t*64 + n) and the first window ----
#pragma unroll
  for (int t = 0; t < 9; ++t)
  {
    // LDS filter row rho = wn*32 + i*16 + x holds output channel wn*32 + 8*(x>>2) + 4*i + (x&3): a lane's two
    // 16x16 tiles (i = 0, 1) then own EIGHT consecutive channels fk*8 .. fk*8+7 of a position, so every epilogue
    // access is one 16-byte load / store per lane instead of two 8-byte ones.
    const int x = rbase & 15, ii = (rbase >> 4) & 1;
    const int chan = (rbase & 32) + 8 * (x >> 2) + 4 * ii + (x & 3);
    glds16(wgt + ((long long)chan * 9 + t) * 64 + chunk * 8, smem_base + (t * 64 + wave * 8) * kRowBytes);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) dma_ring(i * 64);

  // per-lane epilogue constants: channels n = wn*32 + i*16 + fk*4 + r
  float sc[2][4], sh[2][4], mu0[2][4], is0[2][4], mu1[2][4], is1[2][4];
  const bool bwd_stats = p.bn_y[0] != nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = wn * 32 + fk * 8 + i * 4 + r;
      sc[i][r] = p.scale ? p.scale[n] : 1.f;
      sh[i][r] = p.shift ? p.shift[n] : 0.f;
      mu0[i][r] = bwd_stats ? p.bn_mean[0][n] : 0.f;
      is0[i][r] = bwd_stats ? p.bn_invstd[0][n] : 0.f;
      mu1[i][r] = p.bn_y[1] ? p.bn_mean[1][n] : 0.f;
      is1[i][r] = p.bn_y[1] ? p.bn_invstd[1][n] : 0.f;
    }
  float s1[2][4], s2[2][4], s3[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = s3[i][r] = 0.f;
  T* __restrict__ dst = static_cast<T*>(p.dst);
  const T* __restrict__ res = static_cast<const T*>(p.residual);
  const T* __restrict__ msk = static_cast<const T*>(p.relu_mask);
  const unsigned char* __restrict__ mbits = p.relu_mask_bits;
  const unsigned char* ring = smem + L1_WBYTES;
  const int w_lane = (wn * 32 + frow) * kRowBytes + ((fk ^ (frow & 7)) << 4);  // this lane's filter fragment, tap 0 / tile 0 / kk 0

  int base = 0;  // (256*k) mod 640
  for (int k = 0; k < nt; ++k) {
    // this tile's window has landed; every wave is done reading the rows the next fetch replaces.  Vector-memory
    // operations retire in issue order: the window fetch of this tile was issued before the previous tile's operand loads
    // (all consumed by its epilogue) and its four stores, so "at most four outstanding" means the window is in LDS while
    // the stores may still be on their way.
    if (k == 0)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (k + 1 < nt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_ring(256 * k + 384 + i * 64);
    }
    // The epilogue's operands (residual, ReLU mask, saved BatchNorm inputs: 8 bytes per lane and 16x16 tile
    // each) are requested NOW, behind the window fetch and in front of the 144 MFMAs of the tile: read in the
    // epilogue itself they were the exposed latency of a one-workgroup-per-CU kernel (alone: 98 -> 148 us with
    // a residual, 95 -> 179 us with mask + one BatchNorm link).
    auto out_row = [&](int j) -> int {  // NHWC row of this lane's position in 16x16 tile j, or -1 (halo / past the end)
      const long long q = (long long)(t0 + k) * BM + wm * 64 + j * 16 + frow;
      if (q >= p.Q) return -1;
      const unsigned uq = (unsigned)q;
      const unsigned img = fdiv(uq, p.div_pp);
      const unsigned rem = uq - img * (unsigned)p.PP;
      const unsigned hp = fdiv(rem, p.div_pw);
      const unsigned wp = rem - hp * (unsigned)L1_PW;
      return (hp >= 1 && hp <= 56u && wp >= 1 && wp <= 56u) ? (int)((img * 56 + (hp - 1)) * 56 + (wp - 1)) : -1;
    };
    int prow[4] = {-1, -1, -1, -1};
    uint4 pre_res[4], pre_msk[4], pre_y0[4], pre_y1[4];  // 8 channels wn*32 + fk*8 .. +7 of tile j's position
    if constexpr (OPS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        prow[j] = out_row(j);
        pre_res[j] = pre_msk[j] = pre_y0[j] = pre_y1[j] = make_uint4(0u, 0u, 0u, 0u);
        if (prow[j] >= 0) {
          const long long off = (long long)prow[j] * 64 + wn * 32 + fk * 8;
          if (res) pre_res[j] = *reinterpret_cast<const uint4*>(res + off);
          if (msk) pre_msk[j] = *reinterpret_cast<const uint4*>(msk + off);
          if (mbits) pre_msk[j].x = mbits[off >> 3];   // (the mask as one byte, a bit per channel, in the same registers)
          if (bwd_stats) pre_y0[j] = *reinterpret_cast<const uint4*>(static_cast<const T*>(p.bn_y[0]) + off);
          if (p.bn_y[1]) pre_y1[j] = *reinterpret_cast<const uint4*>(static_cast<const T*>(p.bn_y[1]) + off);
        }
      }
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane_row = base + wm * 64 + frow;
    // Eighteen (tap, K half) groups of 6 fragment reads + 8 MFMAs, software-pipelined by hand (round 3): the fragments of
    // group g+1 are requested BEFORE the MFMAs of group g issue (two register sets), so their LDS latency runs under
    // 8 MFMAs instead of in front of them (the compiler's own schedule read each group right before its first use).
    uint4 fwp[2][2], fap[2][4];
    auto read_group = [&](int g, uint4 (&fw)[2], uint4 (&fa)[4]) {
      constexpr int kPW = L1_PW;
      const int t = g >> 1, kk = g & 1;
      const int kh = t / 3, kw = t % 3;
      const int shift = FLIP ? (2 - kh) * kPW + (2 - kw) : kh * kPW + kw;
      // first row of this lane's four tiles for the tap, wrapped once; (row & 7) = (frow + shift) & 7 for all of them
      // (base, wm * 64, j * 16 and the ring size are multiples of 8): the swizzled 16-byte column is a per-tap constant
      int rr0 = lane_row + shift;
      rr0 = rr0 >= L1_RING ? rr0 - L1_RING : rr0;
      const int a0 = rr0 * kRowBytes + ((fk ^ ((frow + shift) & 7)) << 4);
#pragma unroll
      for (int i = 0; i < 2; ++i)   // filter row t*64 + wn*32 + i*16 + frow: (row & 7) = frow & 7
        fw[i] = *reinterpret_cast<const uint4*>(smem + (w_lane ^ (kk << 6)) + (t * 64 + i * 16) * kRowBytes);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        fa[j] = *reinterpret_cast<const uint4*>(ring + (a0 ^ (kk << 6)) + j * (16 * kRowBytes));
    };
    read_group(0, fwp[0], fap[0]);
#pragma unroll
    for (int g = 0; g < 18; ++g) {
      if (g + 1 < 18) read_group(g + 1, fwp[(g + 1) & 1], fap[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // (the reads of group g+1 stay in front of the MFMAs of group g)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) QtMma<T>::run(acc[i][j], fwp[g & 1][i], fap[g & 1][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    base += BM;
    if (base >= L1_RING) base -= L1_RING;

    // ---- epilogue straight from the accumulators ----
    auto bf4 = [](unsigned lo, unsigned hi, float (&f)[4]) {
      f[0] = __uint_as_float(lo << 16); f[1] = __uint_as_float(lo & 0xffff0000u);
      f[2] = __uint_as_float(hi << 16); f[3] = __uint_as_float(hi & 0xffff0000u);
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (!OPS) prow[j] = out_row(j);
      const bool live = prow[j] >= 0;
      const float lm = live ? 1.f : 0.f;  // halo positions take no part in the statistics
      const long long off = (long long)(live ? prow[j] : 0) * 64 + wn * 32 + fk * 8;
      bf16x8 o;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if (p.stats_partial != nullptr && !bwd_stats) {  // (uniform: an eval forward keeps no sums)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1[i][r] += lm * v[r];
            s2[i][r] += lm * v[r] * v[r];
          }
        }
if (p.scale || p.shift) {  // (uniform)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[i][r] + sh[i][r];
        }
        if (OPS && res) {
          float rv[4];
          bf4(i ? pre_res[j].z : pre_res[j].x, i ? pre_res[j].w : pre_res[j].y, rv);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rv[r];
        }
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (OPS && msk) {
          float mv[4];
          bf4(i ? pre_msk[j].z : pre_msk[j].x, i ? pre_msk[j].w : pre_msk[j].y, mv);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = mv[r] > 0.f ? v[r] : 0.f;
        }
        if (OPS && mbits) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (pre_msk[j].x >> (i * 4 + r)) & 1u ? v[r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) o[i * 4 + r] = (bf16_t)v[r];
        if (OPS && bwd_stats) {
          float yv[4];
          bf4(i ? pre_y0[j].z : pre_y0[j].x, i ? pre_y0[j].w : pre_y0[j].y, yv);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1[i][r] += lm * v[r];
            s2[i][r] += lm * v[r] * (yv[r] - mu0[i][r]) * is0[i][r];
          }
          if (p.bn_y[1]) {
            float y2[4];
            bf4(i ? pre_y1[j].z : pre_y1[j].x, i ? pre_y1[j].w : pre_y1[j].y, y2);
#pragma unroll
            for (int r = 0; r < 4; ++r) s3[i][r] += lm * v[r] * (y2[r] - mu1[i][r]) * is1[i][r];
          }
        }
      }
      // One store instruction per j, spelled out: the counted vmcnt(4) at the next tile's barrier relies on exactly four
      // vector-memory operations being issued here, which a predicated / merged compiler store would not guarantee.
      bf16x8* out = live ? reinterpret_cast<bf16x8*>(dst + off) : reinterpret_cast<bf16x8*>(&l1_store_sink[tid]);
      // (s_nop: a store of more than 8 bytes needs wait states before its data registers are written again -- the
      // compiler pads its own stores, not ours)
      asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(out), "v"(o) : "memory");
    }
  }

  // ---- BatchNorm partial sums: one row per workgroup ----
  if (p.stats_partial || bwd_stats) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {  // over the 16 positions (lanes with equal fk)
          s1[i][r] += __shfl_xor(s1[i][r], m, 64);
          s2[i][r] += __shfl_xor(s2[i][r], m, 64);
          s3[i][r] += __shfl_xor(s3[i][r], m, 64);
        }
      }
    float* red = reinterpret_cast<float*>(smem + L1_RED);  // [wave][i][fk][r][3]
    __syncthreads();  // the scratch lies over the filter: every wave has issued its last MFMA
    if (frow == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* d = red + ((((wave * 2 + i) * 4 + fk) * 4 + r) * 3);
          d[0] = s1[i][r];
          d[1] = s2[i][r];
          d[2] = s3[i][r];
        }
    }
    __syncthreads();
    if (tid < 64) {
      const int n = tid, wn_ = n >> 5, fk_ = (n >> 3) & 3, i = (n >> 2) & 1, r = n & 3;  // n = wn*32 + fk*8 + i*4 + r
      float a = 0.f, bb = 0.f, c = 0.f;
      for (int w4 = 0; w4 < 4; ++w4) {
        const float* d = red + (((((wn_ * 4 + w4) * 2 + i) * 4 + fk_) * 4 + r) * 3);
        a += d[0];
        bb += d[1];
        c += d[2];
      }
      float* o0 = bwd_stats ? p.bn_partial[0] : p.stats_partial;
      o0[((long long)b * 2 + 0) * 64 + n] = a;
      o0[((long long)b * 2 + 1) * 64 + n] = bb;
      if (bwd_stats && p.bn_y[1]) {
        p.bn_partial[1][((long long)b * 2 + 0) * 64 + n] = a;
        p.bn_partial[1][((long long)b * 2 + 1) * 64 + n] = c;
      }
    }
  }
}

// Round 3, measured and REMOVED (git history: "layer1: ping-pong cut of the persistent kernel"): the same kernel re-cut on
// conv_pt.hip's ping-pong schedule (waves 0-3 / 4-7 one barrier apart, three taps = 36 fragment reads + 48 MFMAs per
// segment, the epilogue of tile k-1 / the window fetch of tile k+1 inside the first load segment of tile k).  The micro-
// benchmark of the bare schedule promised 1617 cycles per three taps against 2820 in lockstep (scripts/ubench/pingpong.hip),
// the kernel measured 92-97 us against 65-76 us for this one (forward, alone; 117-130 against 85-94 with a residual).
// Ablation of it (QTCNN_L1_DEBUG at the time): bare fragment-read / MFMA loop 58 us, + window fetch 16 us, + epilogue
// 25 us, the wait for the window 0 us.  Per tile a wave spends ~460 vector instructions on position arithmetic (two
// divisions per DMA pass and per output row), accumulator clearing and the epilogue; in a ping-pong they sit in ONE
// wave group's load segment while the partner's 48 MFMAs (768 cycles) hold half of the SIMD's issue slots, and the other
// group repeats them one segment later: paid twice per tile, where the lockstep kernel's two waves per SIMD overlap them
// with each other's MFMAs.  A ping-pong cut of this layer needs tile-invariant addressing (whole-row tiles as conv_pt.hip:
// per-lane offsets + a scalar origin) and an epilogue spread over the segments behind a second accumulator set -- a
// different kernel, not this one with barriers moved.
// A SECOND ping-pong cut, built on those lessons, was measured and removed as well (never committed; its numbers are in
// gpurun_out/r03r_*.json, r03s_l1ab.log and DESIGN.md): whole-row
// tiles (4 rows of a 56-wide image per tile: tile-invariant addressing, one image per workgroup), filter resident (taps 0-7
// in LDS, tap 8 in registers), two accumulator sets with the epilogue of tile k-1 spread over the load segments of tile k,
// residual slices requested by inline-asm loads behind counted waits; bit-identical outputs.  Alone (forward, 256 images):
// 97-100 us against 69-76 us for this kernel (113-120 against 85-94 with a residual).  Ablation: fragment reads +
// barriers + prologue 42 us; + MFMAs 54 us; + patch DMA 74-80 us; + epilogue 97-100 us.  Two conclusions: (i) even with
// DMA and epilogue hidden completely the read / MFMA skeleton of a 64-channel tile (24 fragment reads per 32 MFMAs, one
// barrier pair per two taps) is 54 us -- the gain over this kernel would be 15-20 us; (ii) they do not hide: a wave's
// ~320 vector instructions per tile (epilogue, accumulator clearing, addresses) issue at half rate beside the partner's
// MFMAs and exceed the partner's 2,300 MFMA cycles.  The 64 -> 64 layer has too few MFMAs per staged byte and per output
// element for the two-group schedule; this lockstep kernel (two waves per SIMD overlapping each other's vector work)
// stays the layer1 kernel.
inline int l1_ring_grid(long long Q) {
  const int ntiles = qt_cdiv(Q, BM);
  return ntiles < 256 ? ntiles : 256;
}

template <bool FLIP, bool OPS>
int launch_l1_ring(PatchArgs a, hipStream_t stream) {
  auto kern = conv_l1_ring_kernel<FLIP, OPS>;
  static std::atomic<unsigned long long> lds_limit_set{0};  // per device
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(kern), L1_LDS, lds_limit_set)) return rc;
  a.gridM = qt_cdiv(a.Q, BM);
  a.gridN = 1;
  hipLaunchKernelGGL(kern, dim3(l1_ring_grid(a.Q)), dim3(NT), L1_LDS, stream, a);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

}  // namespace


// Shapes this kernel takes over from the generic implicit GEMM (see qt_conv2d_igemm).

// Mostly off by default (mode 2 = ring kernel only): measured on MI355X (B=256, bf16) it only ties the generic kernel
// (layer1 127 vs 119 us, layer2 109 vs 100 us): with one 8-wave workgroup per CU the load,
// MFMA and epilogue phases of a tile do not overlap, while the generic kernel runs two
// workgroups per CU.  A persistent filter-in-registers variant for the 64->64 layer (4 waves,
// 288 weight VGPRs, double-buffered patch) was also measured: 200 us, slower still; it is in
// the git history (commit "persistent layer1 kernel"), not in the tree.  QTCNN_PATCH_CONV=1 or qt_set_patch_conv(1) turns it on.
static int g_patch_enabled = -1;
extern "C" void qt_set_patch_conv(int mode) { g_patch_enabled = mode < 0 ? 2 : (mode > 2 ? 2 : mode); }   // 0 = off, else on

static bool l1_ring_shape(const qt_conv_desc* d);

bool qt_patch_eligible(const qt_conv_desc* d) {
  if (g_patch_enabled < 0) {
    const char* v = getenv("QTCNN_PATCH_CONV");
    g_patch_enabled = v ? atoi(v) : 2;
  }
  // 0: never; otherwise the shape served by the persistent ring kernel
  if (!g_patch_enabled) return false;
  return l1_ring_shape(d) && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && !d->quad && !d->dst_sub &&
         d->in_h == d->out_h && d->in_w == d->out_w && d->in_h == d->in_w && d->src_pix_stride == d->k_per_tap &&
         d->src_row_stride == d->in_w * d->k_per_tap &&
         d->src_img_stride == (long long)d->in_h * d->in_w * d->k_per_tap &&
         (long long)d->batch * (d->in_h + 2) * (d->in_w + 2) < (1ll << 31);
}

static bool l1_ring_shape(const qt_conv_desc* d) {
  return d->dtype == QT_BF16 && d->in_w == 56 && d->k_per_tap == 64 && d->n_out == 64;
}

int qt_patch_stats_rows(const qt_conv_desc* d) {
  const long long Q = (long long)d->batch * (d->in_h + 2) * (d->in_w + 2);
  return l1_ring_grid(Q);  // the persistent kernel emits one row per workgroup
}

int qt_patch_launch(const qt_conv_desc* d, const qt_conv_io* io, void* stream) {
  PatchArgs a;
  a.src = io->src; a.wgt = io->weight; a.dst = io->dst;
  a.scale = io->scale; a.shift = io->shift; a.residual = io->residual; a.relu_mask = io->relu_mask;
  a.relu_mask_bits = io->relu_mask_bits;
  a.stats_partial = io->stats_partial;
  for (int k = 0; k < 2; ++k) {
    a.bn_y[k] = io->bwd_bn[k].y; a.bn_mean[k] = io->bwd_bn[k].mean; a.bn_invstd[k] = io->bwd_bn[k].invstd;
    a.bn_partial[k] = io->bwd_bn[k].partial;
  }
  a.B = d->batch; a.H = d->in_h; a.W = d->in_w; a.C = d->k_per_tap; a.N = d->n_out;
  a.PW = a.W + 2; a.PP = (a.H + 2) * a.PW;
  a.Q = (long long)a.B * a.PP;
  a.flip = d->mode == QT_CONV_DGRAD;
  a.relu = d->relu;
  a.div_pp = make_fastdiv((unsigned)a.PP);
  a.div_pw = make_fastdiv((unsigned)a.PW);
  a.gridM = a.gridN = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool ops = a.residual || a.relu_mask || a.relu_mask_bits || a.bn_y[0];
  if (a.flip) return ops ? launch_l1_ring<true, true>(a, s) : launch_l1_ring<true, false>(a, s);
  return ops ? launch_l1_ring<false, true>(a, s) : launch_l1_ring<false, false>(a, s);
}
