// Stem convolution (7x7 / stride 2 / pad 3, 3 -> 64 channels) on the packed NHWC4 input, bf16.
//
// Replaces the F.conv2d behind torchvision resnet18.conv1, reached at
// /root/reference/Quadtree_from scratch/models.py:223 (self.base_cnn.conv1) in
// forward() (models.py:274-276) -- first op of the QuadtreeCNN forward pass.
//
// The generic implicit GEMM stages, per output pixel, 7 row taps x 64 B through LDS although
// neighbouring pixels share 3/4 of every tap row, and runs 8 taps (K = 256) because its K-step
// is two taps wide.  Here a workgroup keeps the INPUT rows of a 4 x 112 pixel output tile in LDS
// (13 packed rows = one contiguous 24 KB block of the [B][230][232][4] image), and
//   * an MFMA B-fragment of pixel ow / tap kh / k-group g is the 16 bytes at row (2*oh+kh),
//     byte 16*(ow+g): the im2col overlap is expressed by overlapping LDS reads, nothing is copied;
//   * the whole filter (64 x 7 x 32 bf16 = 28 KB) lives in registers, 28 fragments per wave, for all
//     the tiles a workgroup walks (persistent grid), so one LDS read feeds four MFMAs (K = 224);
//   * the next tile's rows arrive by LDS-DMA while the current tile is multiplied;
//   * the epilogue transposes each 16-pixel block through LDS into one contiguous 2 KB store and
//     keeps BatchNorm sums in registers across tiles (one partial row per workgroup).
// Bound: the 411 MB output write (HBM), not MFMA.
#include <stdlib.h>

#include "qt_common.h"

namespace {

struct StemArgs {
  const bf16_t* x;        // [B][230][232][4]
  const bf16_t* w;        // [64][taps][32]
  bf16_t* y;              // [B][112][112][64]
  const float* scale;     // nullable
  const float* shift;
  float* stats;           // [gridDim.x][2][64] or NULL
  int relu, taps, ntiles;
};

constexpr int ST_TH = 4;                          // output rows per tile
constexpr int ST_ROWB = QT_STEM_PAD_W * 4 * 2;    // 1856 bytes per packed input row
constexpr int ST_IN_ROWS = 2 * ST_TH + 5;         // 13
constexpr int ST_BUF = 24 * 1024;                 // whole 1 KB DMA instructions
static_assert(ST_IN_ROWS * ST_ROWB <= ST_BUF, "a tile's input rows fit one buffer");
constexpr int ST_STG_ROW = 144;                   // 128 B of channels + 16 B pad per pixel
constexpr int ST_STG = 16 * ST_STG_ROW;           // per-wave epilogue staging
constexpr int ST_LDS = 2 * ST_BUF + 4 * ST_STG;

// AFF: scale / shift (+ReLU) epilogue (eval mode); STATS: BatchNorm partial sums (train mode).  The epilogue's VALU work
// is comparable to a block's 28 MFMAs, so neither half is compiled in unless it is used.
template <bool AFF, bool STATS>
__global__ __launch_bounds__(256) void conv_stem_kernel(StemArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const unsigned smem_base = lds_addr_of(smem);

  // contiguous share of the tiles (tile = image * 28 + row group)
  const int t_beg = (int)((long long)blockIdx.x * p.ntiles / gridDim.x);
  const int t_end = (int)((long long)(blockIdx.x + 1) * p.ntiles / gridDim.x);

  // filter fragments: A operand, row = output channel 16*cb + li, k = 8*lg .. 8*lg+7 of tap kh
  uint4 wf[7][4];
#pragma unroll
  for (int kh = 0; kh < 7; ++kh)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
      wf[kh][cb] = *reinterpret_cast<const uint4*>(p.w + ((size_t)(cb * 16 + li) * p.taps + kh) * 32 + lg * 8);

  float sc[4][4], sh[4][4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = cb * 16 + lg * 4 + r;
      sc[cb][r] = p.scale ? p.scale[ch] : 1.f;
      sh[cb][r] = p.scale ? p.shift[ch] : 0.f;
    }
  float s1[4][4], s2[4][4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[cb][r] = s2[cb][r] = 0.f;

  // 24 DMA instructions of 1 KB per tile, 6 per wave
  auto dma_tile = [&](int tile, int buf) {
    const int img = tile / 28, rg = tile - img * 28;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.x) +
                               ((size_t)img * QT_STEM_PAD_H + (size_t)rg * 2 * ST_TH) * ST_ROWB;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int blk = wave * 6 + i;
      glds16(src + blk * 1024 + lane * 16, smem_base + buf * ST_BUF + blk * 1024);
    }
  };

  unsigned char* stg = smem + 2 * ST_BUF + wave * ST_STG;
  if (t_beg < t_end) dma_tile(t_beg, 0);
  for (int t = t_beg; t < t_end; ++t) {
    const int buf = (t - t_beg) & 1;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // tile t landed; buffer buf^1 is free
    if (t + 1 < t_end) dma_tile(t + 1, buf ^ 1);
    const unsigned char* in = smem + buf * ST_BUF;
    const int img = t / 28, rg = t - img * 28;
#pragma unroll 1
    for (int j = 0; j < 7; ++j) {
      const int rb = wave + 4 * j;       // 28 blocks of 16 pixels: 4 rows x 7
      const int orow = rb / 7, ob = rb - orow * 7;
      const unsigned char* a0 = in + (2 * orow) * ST_ROWB + 16 * (ob * 16 + li + lg);
      uint4 xf[7];
#pragma unroll
      for (int kh = 0; kh < 7; ++kh) xf[kh] = *reinterpret_cast<const uint4*>(a0 + kh * ST_ROWB);
      f32x4 acc[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[kh][cb]),
                                                            __builtin_bit_cast(bf16x8, xf[kh]), acc[cb], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      // lane: pixel li of the block, channels 16*cb + 4*lg + r
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = acc[cb][r];
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
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
        *reinterpret_cast<bf16x4*>(stg + li * ST_STG_ROW + cb * 32 + lg * 8) = o;
      }
      // the 16 pixels x 128 B of this block are contiguous in y: two 16-byte stores per lane
      bf16_t* dst = p.y + (((size_t)img * 112 + rg * ST_TH + orow) * 112 + ob * 16) * 64;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int px = h * 8 + (lane >> 3), ck = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(stg + px * ST_STG_ROW + ck * 16);
        *reinterpret_cast<uint4*>(dst + px * 64 + ck * 8) = v;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (STATS && p.stats) {
    // [wave][lane][32] f32 over the (now idle) input buffers, then 128 threads add 64 values each
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[(tid * 32) + cb * 4 + r] = s1[cb][r];
        red[(tid * 32) + 16 + cb * 4 + r] = s2[cb][r];
      }
    __syncthreads();
    if (tid < 128) {
      const int stat = tid >> 6, ch = tid & 63;
      const int cb = ch >> 4, g = (ch >> 2) & 3, r = ch & 3;
      float s = 0.f;
      for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 16; ++l) s += red[((w * 64 + g * 16 + l) * 32) + stat * 16 + cb * 4 + r];
      p.stats[((size_t)blockIdx.x * 2 + stat) * 64 + ch] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Eval mode: conv1 + folded BatchNorm + ReLU + MaxPool2d(3,2,1) in one kernel (the 411 MB conv1 map
// is neither written nor read back; SURVEY.md 8(d) lists this fusion).  One 8-wave workgroup per CU
// walks tiles of TWO pooled rows: the five conv1 rows they need (4k-1 .. 4k+3) are computed exactly as
// above from 15 packed input rows, written as bf16 into an LDS tile [5][112][64] (144-byte pixel
// stride), and after a barrier every thread reduces 3x3 windows of that tile to pooled pixels.
// Post-ReLU values are >= 0, so out-of-image window positions are simply skipped.
// ---------------------------------------------------------------------------------------------
constexpr int SP_IN_ROWS = 15;
constexpr int SP_BUF = 28 * 1024;                  // >= 15 * 1856 = 27840
static_assert(SP_IN_ROWS * ST_ROWB <= SP_BUF, "input rows of a fused tile fit one buffer");
constexpr int SP_PXB = 144;                        // bytes per pixel in the conv tile (128 + pad)
constexpr int SP_TILE = 5 * 112 * SP_PXB;          // 80640
constexpr int SP_LDS = 2 * SP_BUF + SP_TILE;
// RAW form (round 3): the kernel reads the f32 NCHW image itself -- no qt_pack_stem_input pass, no packed copy in HBM.
// The 45 image rows a tile needs (15 rows x 3 planes, 896 B each) arrive by LDS-DMA in a 1 KB-per-row staging area and
// one conversion pass rewrites them as the 15 packed bf16 rows (zero border included) the conv phase reads; staging is
// single buffered (the next tile's rows are requested right after the conversion, behind conv + pool of this tile).
constexpr int SP_RAW_STG = 45 * 1024;
constexpr int SP_LDS_RAW = SP_BUF + SP_RAW_STG + SP_TILE;
static_assert(SP_LDS_RAW <= 160 * 1024, "raw-input form fits the LDS");

struct StemPoolArgs {
  const bf16_t* x;        // [B][230][232][4]
  const bf16_t* w;        // [64][taps][32]
  bf16_t* pooled;         // [B][56][56][64]
  const float* scale;
  const float* shift;
  int taps, ntiles;       // ntiles = B * 28
  const float* img;       // RAW: [B][3][224][224] f32 (x unused)
};

template <bool RAW>
__global__ __launch_bounds__(512) void conv_stem_pool_kernel(StemPoolArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const unsigned smem_base = lds_addr_of(smem);
  unsigned char* ctile = smem + (RAW ? SP_BUF + SP_RAW_STG : 2 * SP_BUF);

  const int t_beg = (int)((long long)blockIdx.x * p.ntiles / gridDim.x);
  const int t_end = (int)((long long)(blockIdx.x + 1) * p.ntiles / gridDim.x);

  uint4 wf[7][4];
#pragma unroll
  for (int kh = 0; kh < 7; ++kh)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
      wf[kh][cb] = *reinterpret_cast<const uint4*>(p.w + ((size_t)(cb * 16 + li) * p.taps + kh) * 32 + lg * 8);
  float sc[4][4], sh[4][4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = cb * 16 + lg * 4 + r;
      sc[cb][r] = p.scale[ch];
      sh[cb][r] = p.shift[ch];
    }

  // tile (img, k): LDS row j of the input buffer = packed input row 8k - 2 + j (rows -2, -1 of the first tile of an
  // image are never read: conv row -1 is skipped).  28 DMA instructions of 1 KB, waves 0..6 take four each.
  auto dma_tile = [&](int tile, int buf) {
    if (wave >= 7) return;
    const int img = tile / 28, k = tile - img * 28;
    const int row0 = k == 0 ? 0 : 8 * k - 2;   // first packed row fetched
    const int skip = k == 0 ? 2 : 0;           // LDS rows left untouched in front of it
    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.x) + ((size_t)img * QT_STEM_PAD_H + row0) * ST_ROWB;
    const int nblk = k == 0 ? 24 : 28;         // 13 rows (24128 B) instead of 15 for the first tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int blk = wave * 4 + i;
      if (blk < nblk) glds16(src + blk * 1024 + lane * 16, smem_base + buf * SP_BUF + skip * ST_ROWB + blk * 1024);
    }
  };

  // RAW: staging row (c, j) <- image row 8k - 5 + j of plane c (packed row 8k - 2 + j is image row 8k - 5 + j), where it
  // exists; 56 lanes x 16 B per row, 45 rows dealt over the 8 waves
  auto dma_raw = [&](int tile) {
    const int img = tile / 28, k = tile - img * 28;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int r = wave + 8 * i;           // c * 15 + j
      if (r < 45) {
        const int c = r / 15, j = r - c * 15;
        const int h = 8 * k - 5 + j;
        if ((unsigned)h < 224u && lane < 56)
          glds16(p.img + (((size_t)img * 3 + c) * 224 + h) * 224 + lane * 4, smem_base + SP_BUF + r * 1024);
      }
    }
  };
  // staging -> packed rows [15][232][4] bf16 in buffer 0 (same rounding as qt_pack_stem_input).  One pixel per thread and
  // round: consecutive lanes read consecutive floats of a plane row and write consecutive 8-byte pixels (no bank conflicts)
  auto convert_raw = [&](int tile) {
    const int k = tile % 28;
    // four pixels per thread and round: image columns 4g .. 4g+3 (one aligned 16-byte read per plane) are packed pixels
    // 4g+3 .. 4g+6; 15 x 56 = 840 items in two rounds, both rounds' reads in flight together
    f32x4 v[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int it = tid + r * 512;
      const int j = it / 56, g = it - j * 56;
      const int h = 8 * k - 5 + j;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (it < 840 && (unsigned)h < 224u) {
        const unsigned char* s0 = smem + SP_BUF + j * 1024 + g * 16;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[r][c] = *reinterpret_cast<const f32x4*>(s0 + c * 15 * 1024);
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int it = tid + r * 512;
      const int j = it / 56, g = it - j * 56;
      if (it < 840) {
        unsigned char* d = smem + j * ST_ROWB + (4 * g + 3) * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bf16x4 o = {(bf16_t)v[r][0][q], (bf16_t)v[r][1][q], (bf16_t)v[r][2][q], (bf16_t)0.f};
          *reinterpret_cast<bf16x4*>(d + q * 8) = o;
        }
      }
    }
    // the zero border: packed pixels 0..2 and 227..231 of the 15 rows
    if (tid < 15 * 8) {
      const int j = tid >> 3, b = tid & 7;
      const bf16x4 z = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
      *reinterpret_cast<bf16x4*>(smem + j * ST_ROWB + (b < 3 ? b : 224 + b) * 8) = z;
    }
  };

  if (t_beg < t_end) {
    if constexpr (RAW) dma_raw(t_beg);
    else dma_tile(t_beg, 0);
  }
  for (int t = t_beg; t < t_end; ++t) {
    const int buf = RAW ? 0 : (t - t_beg) & 1;
    // this tile's rows have landed; every wave is done pooling the previous tile (conv tile free) and reading its input
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (RAW) {
      convert_raw(t);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // packed rows complete, staging free
      if (t + 1 < t_end) dma_raw(t + 1);
    } else {
      if (t + 1 < t_end) dma_tile(t + 1, buf ^ 1);
    }
    const unsigned char* in = smem + buf * SP_BUF;
    const int img = t / 28, k = t - img * 28;
    // ---- conv phase: 35 blocks of 16 pixels (tile row tr = 0..4 <-> conv row 4k - 1 + tr), wave w takes w, w+8, ... ----
#pragma unroll 1
    for (int rb = wave; rb < 35; rb += 8) {
      const int tr = rb / 7, ob = rb - tr * 7;
      if (k == 0 && tr == 0) continue;           // conv row -1 does not exist
      const unsigned char* a0 = in + (2 * tr) * ST_ROWB + 16 * (ob * 16 + li + lg);
      uint4 xf[7];
#pragma unroll
      for (int kh = 0; kh < 7; ++kh) xf[kh] = *reinterpret_cast<const uint4*>(a0 + kh * ST_ROWB);
      f32x4 acc[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[kh][cb]),
                                                            __builtin_bit_cast(bf16x8, xf[kh]), acc[cb], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)fmaxf(acc[cb][r] * sc[cb][r] + sh[cb][r], 0.f);
        *reinterpret_cast<bf16x4*>(ctile + ((tr * 112 + ob * 16 + li) * SP_PXB) + cb * 32 + lg * 8) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the conv tile is complete
    // ---- pool phase: one thread per (pooled column, 8-channel group) does BOTH pooled rows of the tile: the five conv rows
    // x three columns it needs are 15 reads in flight at once (two rounds of nine-read items before: 18 reads per pair and
    // two exposed LDS latencies), the middle row's column maximum is shared ----
    if (tid < 448) {
      const int cg = tid & 7, pw = tid >> 3;
      // The tile holds ReLU outputs: non-negative bf16 values order like their bit patterns, so the 3x3 maximum is a
      // packed SIGNED 16-bit maximum on the raw words (a -0 the ReLU may have let through is then the smallest value,
      // as it is for the float maximum against the +0 start): 4 instructions per 8 channels and window cell instead of
      // 8 unpacks + 8 float maxima, and the result is the same bf16 value, bit for bit.
      typedef short us2 __attribute__((ext_vector_type(2)));
      auto pkmax = [](unsigned a, unsigned b) -> unsigned {
        return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
      };
      auto max4 = [&](const uint4& a, const uint4& b) -> uint4 {
        return make_uint4(pkmax(a.x, b.x), pkmax(a.y, b.y), pkmax(a.z, b.z), pkmax(a.w, b.w));
      };
      const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
      uint4 raw[5][3];
#pragma unroll
      for (int tr = 0; tr < 5; ++tr)
#pragma unroll
        for (int dc = 0; dc < 3; ++dc) {
          const int col = 2 * pw - 1 + dc;
          raw[tr][dc] = zero;                       // (out-of-image cells: 0 never beats a ReLU output)
          if (!(k == 0 && tr == 0) && (unsigned)col < 112u)
            raw[tr][dc] = *reinterpret_cast<const uint4*>(ctile + (tr * 112 + col) * SP_PXB + cg * 16);
        }
      uint4 rowmax[5];
#pragma unroll
      for (int tr = 0; tr < 5; ++tr) rowmax[tr] = max4(max4(raw[tr][0], raw[tr][1]), raw[tr][2]);
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const uint4 best = max4(max4(rowmax[2 * pi], rowmax[2 * pi + 1]), rowmax[2 * pi + 2]);
        *reinterpret_cast<uint4*>(p.pooled + (((size_t)img * 56 + 2 * k + pi) * 56 + pw) * 64 + cg * 8) = best;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int g_stem_enabled = -1;
bool stem_enabled() {
  if (g_stem_enabled < 0) {
    const char* e = getenv("QTCNN_STEM_CONV");
    g_stem_enabled = e ? atoi(e) : 1;
  }
  return g_stem_enabled != 0;
}

int stem_grid(int batch) {
  const int ntiles = batch * (112 / ST_TH);
  return ntiles < 512 ? ntiles : 512;  // two workgroups per CU
}

}  // namespace

// 1 (default): the packed bf16 stem convolution takes the dedicated kernel; 0: generic implicit GEMM
extern "C" void qt_set_stem_conv(int mode) { g_stem_enabled = mode < 0 ? 1 : (mode != 0); }

bool qt_stem_eligible(const qt_conv_desc* d, const qt_conv_io* io) {
  if (!stem_enabled() || d->dtype != QT_BF16 || d->mode != QT_CONV_FWD) return false;
  if (d->k_per_tap != 32 || d->kw != 1 || (d->kh != 7 && d->kh != 8) || d->stride != 2 || d->pad != 0) return false;
  if (d->n_out != 64 || d->out_h != 112 || d->out_w != 112 || d->in_h != QT_STEM_PAD_H || d->in_w != QT_STEM_PAD_W)
    return false;
  if (d->src_pix_stride != 4 || d->src_row_stride != QT_STEM_PAD_W * 4 ||
      d->src_img_stride != (long long)QT_STEM_PAD_H * QT_STEM_PAD_W * 4)
    return false;
  if (d->quad || d->dst_sub) return false;
  if (io && (io->residual || io->relu_mask || io->relu_mask_bits || io->bwd_bn[0].y || io->bwd_bn[1].y)) return false;
  return true;
}

int qt_stem_stats_rows(const qt_conv_desc* d) { return stem_grid(d->batch); }

int qt_stem_launch(const qt_conv_desc* d, const qt_conv_io* io, void* stream) {
  QT_CHECK_ARG((io->scale == nullptr) == (io->shift == nullptr), "qt_conv2d_igemm: scale and shift come together");
  StemArgs a;
  a.x = static_cast<const bf16_t*>(io->src);
  a.w = static_cast<const bf16_t*>(io->weight);
  a.y = static_cast<bf16_t*>(io->dst);
  a.scale = io->scale; a.shift = io->shift; a.stats = io->stats_partial;
  a.relu = d->relu; a.taps = d->kh; a.ntiles = d->batch * (112 / ST_TH);
  const bool aff = a.scale != nullptr || a.relu, stats = a.stats != nullptr;
  void (*kern)(StemArgs) = aff ? (stats ? conv_stem_kernel<true, true> : conv_stem_kernel<true, false>)
                               : (stats ? conv_stem_kernel<false, true> : conv_stem_kernel<false, false>);
  static std::atomic<unsigned long long> lds_limit_set[4];  // per instantiation, per device
  const int ki = (aff ? 2 : 0) + (stats ? 1 : 0);
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(kern), ST_LDS, lds_limit_set[ki])) return rc;
  hipLaunchKernelGGL(kern, dim3(stem_grid(d->batch)), dim3(256), ST_LDS, static_cast<hipStream_t>(stream), a);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

// conv1 + BatchNorm (as scale / shift) + ReLU + MaxPool2d(3,2,1) of the eval forward in one launch:
// xpad [B][230][232][4] bf16, weights [64][taps][32] (qt_pack_stem_weight), pooled [B][56][56][64].
extern "C" int qt_stem_conv_pool(int dtype, const void* xpad, const void* weight, int taps, const float* scale,
                                 const float* shift, void* pooled, int batch, void* stream) {
  QT_CHECK_ARG(xpad && weight && scale && shift && pooled && batch > 0 && (taps == 7 || taps == 8),
               "qt_stem_conv_pool: bad argument");
  if (dtype != QT_BF16 || !stem_enabled()) {
    qt_set_error("qt_stem_conv_pool: bf16 only (use qt_conv2d_igemm + qt_stem_pool)");
    return QT_ERR_UNSUPPORTED;
  }
  StemPoolArgs a;
  a.x = static_cast<const bf16_t*>(xpad); a.w = static_cast<const bf16_t*>(weight); a.img = nullptr;
  a.pooled = static_cast<bf16_t*>(pooled); a.scale = scale; a.shift = shift; a.taps = taps; a.ntiles = batch * 28;
  static std::atomic<unsigned long long> lds_limit_set{0};  // per device
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(conv_stem_pool_kernel<false>), SP_LDS, lds_limit_set)) return rc;
  const int grid = a.ntiles < 256 ? a.ntiles : 256;
  hipLaunchKernelGGL(conv_stem_pool_kernel<false>, dim3(grid), dim3(512), SP_LDS, static_cast<hipStream_t>(stream), a);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

// The same from the f32 NCHW image the reference's dataloader hands over ([B][3][224][224]): packing, conv1, BatchNorm,
// ReLU and the max pool in one launch; bit-identical to qt_pack_stem_input + qt_stem_conv_pool.
extern "C" int qt_stem_conv_pool_nchw(int dtype, const float* image_nchw, const void* weight, int taps, const float* scale,
                                      const float* shift, void* pooled, int batch, void* stream) {
  QT_CHECK_ARG(image_nchw && weight && scale && shift && pooled && batch > 0 && (taps == 7 || taps == 8),
               "qt_stem_conv_pool_nchw: bad argument");
  static int raw_on = -1;
  if (raw_on < 0) {
    const char* e = getenv("QTCNN_STEM_NCHW");
    raw_on = e ? atoi(e) : 1;
  }
  if (dtype != QT_BF16 || !stem_enabled() || !raw_on || ((uintptr_t)image_nchw % 16) != 0) {
    qt_set_error("qt_stem_conv_pool_nchw: bf16 and a 16-byte aligned image only (use qt_pack_stem_input + qt_stem_conv_pool)");
    return QT_ERR_UNSUPPORTED;
  }
  StemPoolArgs a;
  a.x = nullptr; a.img = image_nchw; a.w = static_cast<const bf16_t*>(weight);
  a.pooled = static_cast<bf16_t*>(pooled); a.scale = scale; a.shift = shift; a.taps = taps; a.ntiles = batch * 28;
  static std::atomic<unsigned long long> lds_limit_set{0};  // per device
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(conv_stem_pool_kernel<true>), SP_LDS_RAW, lds_limit_set)) return rc;
  const int grid = a.ntiles < 256 ? a.ntiles : 256;
  hipLaunchKernelGGL(conv_stem_pool_kernel<true>, dim3(grid), dim3(512), SP_LDS_RAW, static_cast<hipStream_t>(stream), a);
  QT_CHECK_LAUNCH();
  return QT_OK;
}
