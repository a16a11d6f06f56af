// Implicit-GEMM convolution on MFMA for gfx950 (forward gather and data-gradient
// gather), NHWC activations, [N][taps][K] weights, f32 accumulate.
//
// Replaces the ATen conv2d / conv2d-backward-input calls made by
//   /root/reference/Quadtree_from scratch/models.py:222-243,284-289 (forward) and
//   loss.backward() at Quadtree_from scratch/Quadtree_train.py:65 (dgrad).
//
// GEMM view:  D[n][m] = sum_{tap,k} W[n][tap][k] * X[pixel(m) shifted by tap][k]
//   m = destination pixel (image, oh, ow) flattened, n = destination channel.
// Workgroup = 256 threads = 4 waves (2x2), tile BM pixels x BN channels,
// K-step = 128 bytes of K per row (64 bf16 / 32 f32).  Both operand tiles are
// staged global -> registers -> LDS (zero fill of the halo happens in the
// register stage), double buffered, one barrier per K-step; the LDS image is
// XOR-swizzled per 16-byte chunk so that the ds_read_b128 fragment reads are
// conflict free.  The MFMA is issued with the WEIGHT tile as the A operand, so
// each lane ends up with 4 consecutive channels of one pixel: the epilogue
// stages the f32 tile through LDS and writes whole NHWC rows (16 B per lane),
// fusing scale/shift (+residual) (+ReLU) (+ReLU-mask) and BatchNorm partial sums.
#include "qt_common.h"

namespace {

struct ConvArgs {
  const void* src;
  const void* wgt;
  void* dst;
  const float* scale;      // per destination channel, nullable
  const float* shift;      // per destination channel, nullable
  const void* residual;    // [M][N] same dtype, nullable
  const void* relu_mask;   // [M][N] same dtype: result *= (mask > 0), nullable
  float* stats_partial;    // [gridM][2][N] per-tile sum / sum of squares, nullable
  long long src_img_stride;
  int src_row_stride, src_pix_stride;
  int M, N;
  int OH, OW, IH, IW;
  int KC;                  // K elements per tap
  int ntaps, KW;
  int stride, pad;
  int quad;
  int relu;
  int gridM, gridN;
};

constexpr int kRowBytes = 128;  // bytes of K per row per K-step

template <typename T, int BM, int BN, bool DGRAD>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
  constexpr int BK = kRowBytes / (int)sizeof(T);
  constexpr int RA = BM / 32;  // pixel rows staged per thread
  constexpr int RW = BN / 32;  // weight rows staged per thread
  constexpr int TM = BM / 32;  // 16-wide pixel tiles per wave
  constexpr int TN = BN / 32;  // 16-wide channel tiles per wave
  constexpr int STAGE_BYTES = (BM + BN) * kRowBytes;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const T* __restrict__ src = static_cast<const T*>(p.src);
  const T* __restrict__ wgt = static_cast<const T*>(p.wgt);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;

  const int bid = qt_xcd_remap(blockIdx.x, p.gridM * p.gridN);
  const int mt = bid / p.gridN, nt = bid - mt * p.gridN;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread staging rows -------------------------------------------------
  const int chunk = tid & 7;  // 16-byte chunk inside the 128-byte row
  const int rbase = tid >> 3;  // 0..31
  long long a_base[RA];
  int a_oh[RA], a_ow[RA];
  const int OHW = p.OH * p.OW;
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + rbase + 32 * i;
    if (m < p.M) {
      int img = m / OHW;
      int rem = m - img * OHW;
      int oh = rem / p.OW;
      int ow = rem - oh * p.OW;
      long long base;
      if (p.quad) {
        if (!DGRAD) {
          // destination = quadrant-local 7x7 pixel of quadrant q of image n;
          // source = the un-split map, offset to the quadrant's corner.
          const int n = img >> 2, q = img & 3;
          base = (long long)n * p.src_img_stride +
                 (long long)(q >> 1) * p.IH * p.src_row_stride +
                 (long long)(q & 1) * p.IW * p.src_pix_stride;
        } else {
          // destination = pixel of the un-split (2*IH x 2*IW) map; source = the
          // dense per-quadrant gradient image of the quadrant that owns it.
          const int qh = oh >= p.IH, qw = ow >= p.IW;
          base = (long long)(img * 4 + qh * 2 + qw) * p.src_img_stride;
          oh -= qh * p.IH;
          ow -= qw * p.IW;
        }
      } else {
        base = (long long)img * p.src_img_stride;
      }
      a_base[i] = base;
      a_oh[i] = oh;
      a_ow[i] = ow;
    } else {
      a_base[i] = 0;
      a_oh[i] = -100000;  // every tap invalid
      a_ow[i] = -100000;
    }
  }
  const int ktot = p.ntaps * p.KC;
  const T* w_ptr[RW];
  bool w_ok[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int n = n0 + rbase + 32 * i;
    w_ok[i] = n < p.N;
    w_ptr[i] = wgt + (long long)(w_ok[i] ? n : 0) * ktot + chunk * (16 / (int)sizeof(T));
  }

  uint4 ra[RA], rw[RW];
  const int ksteps_per_tap = p.KC / BK;
  const int nk = (p.ntaps * p.KC) / BK;

  // A K-step is 128 bytes of K per row.  Normally that is a slice of one tap
  // (tap uniform over the workgroup); when a tap is only 64 bytes (the packed
  // bf16 stem) a K-step spans two taps and the tap depends on the chunk.
  const int sub = p.KC * (int)sizeof(T) < kRowBytes;
  auto load_stage = [&](int ks) {
    int tap, c0;
    if (sub) {
      tap = ks * 2 + (chunk >> 2);
      c0 = -(chunk >> 2) * (BK / 2);  // chunk*EPC + c0 = element inside the tap
    } else {
      tap = ks / ksteps_per_tap;
      c0 = (ks - tap * ksteps_per_tap) * BK;
    }
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int ih, iw;
      bool ok;
      if (!DGRAD) {
        ih = a_oh[i] * p.stride - p.pad + kh;
        iw = a_ow[i] * p.stride - p.pad + kw;
        ok = (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      } else {
        const int th = a_oh[i] + p.pad - kh, tw = a_ow[i] + p.pad - kw;
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
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) {
        const T* g = src + a_base[i] + (long long)ih * p.src_row_stride +
                     (long long)iw * p.src_pix_stride + c0 + chunk * (16 / (int)sizeof(T));
        v = *reinterpret_cast<const uint4*>(g);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (w_ok[i]) v = *reinterpret_cast<const uint4*>(w_ptr[i] + (long long)ks * BK);
      rw[i] = v;
    }
  };
  auto store_stage = [&](int buf) {
    unsigned char* sa = smem + buf * STAGE_BYTES;
    unsigned char* sw = sa + BM * kRowBytes;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int r = rbase + 32 * i;
      *reinterpret_cast<uint4*>(sa + r * kRowBytes + ((chunk ^ (r & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int r = rbase + 32 * i;
      *reinterpret_cast<uint4*>(sw + r * kRowBytes + ((chunk ^ (r & 7)) << 4)) = rw[i];
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fk = lane >> 4;

  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) load_stage(ks + 1);
    const unsigned char* sa = smem + buf * STAGE_BYTES;
    const unsigned char* sw = sa + BM * kRowBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 fw[TN], fa[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int r = wn * (BN / 2) + i * 16 + frow;
        fw[i] = *reinterpret_cast<const uint4*>(sw + r * kRowBytes + (((kk * 4 + fk) ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int r = wm * (BM / 2) + j * 16 + frow;
        fa[j] = *reinterpret_cast<const uint4*>(sa + r * kRowBytes + (((kk * 4 + fk) ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) QtMma<T>::run(acc[i][j], fw[i], fa[j]);
    }
    if (ks + 1 < nk) store_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS f32 [BM][BN] (chunk-swizzled) -------------
  // lane holds channels n = 4*(lane>>4)+r of pixel (lane&15) for each 16x16 tile.
  constexpr int ROWB = BN * 4;  // bytes per staged pixel row
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int pm = wm * (BM / 2) + j * 16 + frow;
      const int c16 = (wn * (BN / 2) + i * 16 + fk * 4) >> 2;  // 16-byte chunk index
      *reinterpret_cast<f32x4*>(smem + pm * ROWB + ((c16 ^ (pm & 7)) << 4)) = acc[i][j];
    }
  __syncthreads();

  constexpr int TPR = BN / 8;          // threads per pixel row (8 channels each)
  constexpr int RPP = 256 / TPR;       // rows per pass
  constexpr int NPASS = BM / RPP;
  const int cg = tid % TPR, r0 = tid / TPR;
  const int nbase = n0 + cg * 8;
  const bool n_ok = nbase < p.N;  // N is a multiple of 8
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = (p.scale && n_ok) ? p.scale[nbase + e] : 1.f;
    sh[e] = (p.shift && n_ok) ? p.shift[nbase + e] : 0.f;
  }
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  T* __restrict__ dst = static_cast<T*>(p.dst);
  const T* __restrict__ res = static_cast<const T*>(p.residual);
  const T* __restrict__ msk = static_cast<const T*>(p.relu_mask);
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int r = r0 + ps * RPP;
    const int m = m0 + r;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + r * ROWB + (((2 * cg) ^ (r & 7)) << 4));
    const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + r * ROWB + (((2 * cg + 1) ^ (r & 7)) << 4));
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if (m < p.M && n_ok) {
      const long long off = (long long)m * p.N + nbase;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += v[e];
        s2[e] += v[e] * v[e];
        v[e] = v[e] * sc[e] + sh[e];
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
      QtVec8<T>::store(dst + off, v);
    }
  }

  if (p.stats_partial) {
    // reduce the per-thread sums over the threads that share a channel group
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [RPP][BN][2]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(r0 * BN + cg * 8 + e) * 2 + 0] = s1[e];
      red[(r0 * BN + cg * 8 + e) * 2 + 1] = s2[e];
    }
    __syncthreads();
    if (tid < BN) {
      float a = 0.f, b = 0.f;
      for (int r = 0; r < RPP; ++r) {
        a += red[(r * BN + tid) * 2 + 0];
        b += red[(r * BN + tid) * 2 + 1];
      }
      if (n0 + tid < p.N) {
        p.stats_partial[((long long)mt * 2 + 0) * p.N + n0 + tid] = a;
        p.stats_partial[((long long)mt * 2 + 1) * p.N + n0 + tid] = b;
      }
    }
  }
}

template <typename T, int BM, int BN, bool DGRAD>
int launch(const ConvArgs& a, hipStream_t stream) {
  constexpr int STAGE = (BM + BN) * kRowBytes;
  constexpr int EPI = BM * BN * 4;
  constexpr int LDS = (2 * STAGE > EPI) ? 2 * STAGE : EPI;
  auto kern = conv_igemm_kernel<T, BM, BN, DGRAD>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      qt_set_error("hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
      return QT_ERR_LAUNCH;
    }
    attr_done = true;
  }
  ConvArgs args = a;
  args.gridM = qt_cdiv(a.M, BM);
  args.gridN = qt_cdiv(a.N, BN);
  hipLaunchKernelGGL(kern, dim3(args.gridM * args.gridN), dim3(256), LDS, stream, args);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

template <typename T>
int dispatch(const qt_conv_desc* d, const ConvArgs& a, hipStream_t stream) {
  const bool dg = d->mode == QT_CONV_DGRAD;
  if (a.N <= 64) return dg ? launch<T, 128, 64, true>(a, stream) : launch<T, 128, 64, false>(a, stream);
  return dg ? launch<T, 128, 128, true>(a, stream) : launch<T, 128, 128, false>(a, stream);
}

}  // namespace

extern "C" int qt_conv2d_stats_rows(const qt_conv_desc* d) {
  if (!d) return QT_ERR_INVALID_ARG;
  const long long M = (long long)d->batch * (d->quad && d->mode == QT_CONV_FWD ? 4 : 1) * d->out_h * d->out_w;
  return qt_cdiv(M, 128);
}

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
  const int esz = d->dtype == QT_F32 ? 4 : 2;
  QT_CHECK_ARG(((uintptr_t)io->src % 16) == 0 && ((uintptr_t)io->weight % 16) == 0 && ((uintptr_t)io->dst % 16) == 0,
               "qt_conv2d_igemm: pointers must be 16-byte aligned");
  QT_CHECK_ARG((d->src_pix_stride * esz) % 8 == 0 && ((long long)d->src_row_stride * esz) % 16 == 0 &&
                   (d->src_img_stride * esz) % 16 == 0,
               "qt_conv2d_igemm: source strides break 16-byte alignment");
  // 16-byte loads start at pixel*pix_stride + c0 + chunk*16B: pixel stride must keep that aligned
  QT_CHECK_ARG((d->src_pix_stride * esz) % 16 == 0 || d->stride * d->src_pix_stride * esz % 16 == 0,
               "qt_conv2d_igemm: pixel stride %d breaks 16-byte alignment", d->src_pix_stride);

  ConvArgs a;
  a.src = io->src; a.wgt = io->weight; a.dst = io->dst;
  a.scale = io->scale; a.shift = io->shift;
  a.residual = io->residual; a.relu_mask = io->relu_mask;
  a.stats_partial = io->stats_partial;
  a.src_img_stride = d->src_img_stride;
  a.src_row_stride = d->src_row_stride;
  a.src_pix_stride = d->src_pix_stride;
  const int imgs = d->batch * ((d->quad && d->mode == QT_CONV_FWD) ? 4 : 1);
  const long long M = (long long)imgs * d->out_h * d->out_w;
  QT_CHECK_ARG(M < (1ll << 31) && M * d->n_out < (1ll << 40), "qt_conv2d_igemm: problem too large");
  a.M = (int)M; a.N = d->n_out;
  a.OH = d->out_h; a.OW = d->out_w; a.IH = d->in_h; a.IW = d->in_w;
  a.KC = d->k_per_tap; a.ntaps = d->kh * d->kw; a.KW = d->kw;
  a.stride = d->stride; a.pad = d->pad; a.quad = d->quad; a.relu = d->relu;
  a.gridM = a.gridN = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return d->dtype == QT_F32 ? dispatch<float>(d, a, s) : dispatch<bf16_t>(d, a, s);
}
