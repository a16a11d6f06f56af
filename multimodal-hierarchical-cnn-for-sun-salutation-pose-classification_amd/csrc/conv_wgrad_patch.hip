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
__global__ __launch_bounds__(256 * G) void conv_wgrad_patch_kernel(WPArgs p) {
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
    if (!oihw) {
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

int g_wgrad_patch_min_w = -1;  // smallest image width routed here; 0 = off

int min_w() {
  if (g_wgrad_patch_min_w < 0) {
    const char* e = getenv("QTCNN_WGRAD_PATCH_MIN_W");
    g_wgrad_patch_min_w = e ? atoi(e) : 14;
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
  if (a.part) {
    const int nq = (int)(filt / 4);
    if (real_split > 16)
      hipLaunchKernelGGL(wgrad_partial_sum_kernel<32>, dim3(qt_cdiv(nq, 8)), dim3(256), 0, stream, a.part, a.dw, nq,
                         real_split, (long long)nq, a.KC, oihw);
    else
      hipLaunchKernelGGL(wgrad_partial_sum_kernel<8>, dim3(qt_cdiv(nq, 32)), dim3(256), 0, stream, a.part, a.dw, nq,
                         real_split, (long long)nq, a.KC, oihw);
    QT_CHECK_LAUNCH();
  }
  return QT_OK;
}

}  // namespace

// Smallest image width whose 3x3 stride-1 weight gradients take the streaming kernel
// (0 = never; default 14 = every eligible layer, env QTCNN_WGRAD_PATCH_MIN_W).
extern "C" void qt_set_wgrad_patch_min_width(int w) { g_wgrad_patch_min_w = w < 0 ? 14 : w; }

bool qt_wgrad_patch_eligible(const qt_conv_desc* d) {
  const int mw = min_w();
  if (mw <= 0 || d->dtype != QT_BF16) return false;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->quad) return false;
  if (d->in_h != d->out_h || d->in_w != d->out_w) return false;
  if (d->n_out % 64 || d->k_per_tap % 64) return false;
  if (d->out_w < mw || d->out_w < 7 || d->out_h < 7 || d->out_w > 120) return false;
  if ((long long)d->batch * (d->out_h + 1) * (d->out_w + 1) >= (1ll << 30)) return false;
  return true;
}

// bytes of partial-filter workspace the deterministic path wants for this convolution
size_t qt_wgrad_patch_workspace_bytes(const qt_conv_desc* d) {
  if (!qt_wgrad_patch_eligible(d)) return 0;
  const int total = d->batch * (d->out_h + 1) * (d->out_w + 1);
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
  a.PW = a.W + 1; a.PH = a.H + 1; a.PP = a.PW * a.PH;
  a.x_is = d->src_img_stride; a.x_rs = d->src_row_stride; a.x_ps = d->src_pix_stride;
  a.dy_ps = a.N; a.dy_rs = a.W * a.N; a.dy_is = (long long)a.H * a.dy_rs;
  a.total = a.B * a.PP;
  a.halo = (a.PW + 1 + 31) / 32 * 32;
  a.div_pp = make_fastdiv((unsigned)a.PP);
  a.div_pw = make_fastdiv((unsigned)a.PW);
  a.tilesC = a.KC / 64;
  a.tiles = (a.N / 64) * a.tilesC;
  a.adv_h = a.adv_w = a.pps = a.nsplit = 0;
  static int variant = -1;
  if (variant < 0) {
    const char* e = getenv("QTCNN_WP_VARIANT");
    variant = e ? atoi(e) : 0;
  }
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
  if (variant == 2) return launch_patch<2, 2, 8, 4>(a, workspace_bytes, oihw, s);
  return launch_patch<kGroups, 4, 16, 8>(a, workspace_bytes, oihw, s);
}
