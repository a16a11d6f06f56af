// Layout packing kernels (HBM-bound, coalesced): input image, weights, weight gradients.
//
//  * NCHW f32 image -> zero-padded NHWC4 tensor for the packed 7x7/2 stem
//    (input contract: /root/reference/Quadtree_from scratch/dataloader.py:72-91).
//  * OIHW f32 master weights (the reference's state_dict layout)
//    -> [O][kh][kw][I] forward operand and [I][kh][kw][O] data-gradient operand.
//  * [O][kh][kw][I] f32 weight gradients -> OIHW f32 .grad tensors.
#include <math.h>
#include <string.h>

#include "qt_common.h"

namespace {

template <typename T> __device__ __forceinline__ T qt_from_f32(float v);
template <> __device__ __forceinline__ float qt_from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t qt_from_f32<bf16_t>(float v) { return (bf16_t)v; }

// dst[n][230][232][4]: rows/cols shifted by +3, channel 3 and the border are zero.
template <typename T>
__global__ void pack_stem_input_kernel(const float* __restrict__ img, T* __restrict__ dst, int batch) {
  constexpr int PH = QT_STEM_PAD_H, PW = QT_STEM_PAD_W;
  const long long total = (long long)batch * PH * PW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int pw = (int)(i % PW);
    const int ph = (int)((i / PW) % PH);
    const int n = (int)(i / ((long long)PW * PH));
    const int h = ph - 3, w = pw - 3;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)h < 224u && (unsigned)w < 224u) {
      const float* s = img + (long long)n * 3 * 224 * 224 + h * 224 + w;
      v[0] = s[0];
      v[1] = s[224 * 224];
      v[2] = s[2 * 224 * 224];
    }
    T* d = dst + i * 4;
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      *reinterpret_cast<bf16x4*>(d) = o;
    }
  }
}

// one thread per (o, tap, i): reads OIHW, writes both packed layouts
template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ fwd, T* __restrict__ dgrad,
                                        int O, int I, int taps) {
  const long long total = (long long)O * I * taps;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    // idx enumerates the forward layout [o][tap][i] (coalesced writes)
    const int i = (int)(idx % I);
    const int tap = (int)((idx / I) % taps);
    const int o = (int)(idx / ((long long)I * taps));
    const float v = w[((long long)o * I + i) * taps + tap];
    if (fwd) fwd[idx] = qt_from_f32<T>(v);
    if (dgrad) dgrad[((long long)i * taps + tap) * O + o] = qt_from_f32<T>(v);
  }
}

// All conv / linear weights of a model in ONE launch.  A block transposes a 32(o) x 32(i) x taps
// tile through LDS: the OIHW rows are read as contiguous runs of 32*taps floats, the forward operand
// [o][tap][i] is written in runs of 32 i, the data-gradient operand [i][tap][o] (or its parity-class
// form for stride-2 convs, see pack_dgrad_s2_kernel) in runs of 32 o.  The per-element kernels above
// scatter 2-byte stores with a stride of taps*O elements and need one launch per layer (0.42 ms per
// training step for the 26 M parameters of QuadtreeCNN); this one moves the same bytes in ~60 us.
// Adam with L2-in-gradient weight decay, torch.optim.Adam semantics (amsgrad / maximize off):
//   g += wd*p;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// (the reference's optimizer: Quadtree_from scratch/Quadtree_train.py:45, resnet/train_cnn_model.py:65)
struct AdamScalars {
  float lr, b1, b2, eps, wd;
  float step_size;     // lr / (1 - b1^t)
  float inv_sqrt_bc2;  // 1 / sqrt(1 - b2^t)
  float grad_scale;    // multiplies the incoming gradient (1 = none)
  int enabled;
};
__device__ __forceinline__ float adam_update(float p, float g, float& m, float& v, const AdamScalars& h) {
  g = g * h.grad_scale + h.wd * p;
  m = h.b1 * m + (1.f - h.b1) * g;
  v = h.b2 * v + (1.f - h.b2) * g * g;
  const float denom = sqrtf(v) * h.inv_sqrt_bc2 + h.eps;
  return p - h.step_size * (m / denom);
}

constexpr int PK_MAX_ITEMS = 32;
struct PackBatchArgs {
  AdamScalars adam;                  // enabled: the masters are updated in the same pass
  const float* g[PK_MAX_ITEMS];
  float* m[PK_MAX_ITEMS];
  float* v[PK_MAX_ITEMS];
  const float* w[PK_MAX_ITEMS];
  void* fwd[PK_MAX_ITEMS];
  void* dgrad[PK_MAX_ITEMS];
  int O[PK_MAX_ITEMS], I[PK_MAX_ITEMS];
  unsigned char k[PK_MAX_ITEMS], s2[PK_MAX_ITEMS];
  int first_block[PK_MAX_ITEMS + 1];
  int n;
};

template <typename T> struct PackPair;  // two consecutive elements as one store
template <> struct PackPair<float> {
  static __device__ __forceinline__ void store(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
};
template <> struct PackPair<bf16_t> {
  static __device__ __forceinline__ void store(bf16_t* p, float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v = {(bf16_t)a, (bf16_t)b};
    *reinterpret_cast<bf16x2*>(p) = v;
  }
};

// one TO(o) x TI(i) x TAPS tile; `tile` holds it as [o][i*TAPS + tap] with an odd row stride
template <typename T, int TAPS, int TO, int TI>
__device__ __forceinline__ void pack_tile(float* tile, const float* w, T* __restrict__ fwd, T* __restrict__ dg,
                                          int O, int I, int o0, int i0, int s2, const AdamScalars& adam,
                                          const float* __restrict__ g, float* m, float* v) {
  constexpr int RUN = TI * TAPS, S = RUN + 1, TOTAL = TO * RUN;
  for (int e = threadIdx.x; e < TOTAL; e += 256) {
    const int o = e / RUN, r = e - o * RUN;
    const long long idx = ((long long)(o0 + o) * I + i0) * TAPS + r;
    float pv = w[idx];
    if (adam.enabled) {  // optimizer step on the f32 master, then the fresh value is what gets packed
      float mv = m[idx], vv = v[idx];
      pv = adam_update(pv, g[idx], mv, vv, adam);
      m[idx] = mv;
      v[idx] = vv;
      const_cast<float*>(w)[idx] = pv;
    }
    tile[o * S + r] = pv;
  }
  __syncthreads();
  if (fwd) {
    for (int e = threadIdx.x; e < TOTAL / 2; e += 256) {
      const int i = (e % (TI / 2)) * 2, ot = e / (TI / 2);
      const int o = ot / TAPS, tap = ot - o * TAPS;
      const float* t = tile + o * S + i * TAPS + tap;
      PackPair<T>::store(fwd + ((long long)(o0 + o) * TAPS + tap) * I + i0 + i, t[0], t[TAPS]);
    }
  }
  if (dg) {
    for (int e = threadIdx.x; e < TOTAL / 2; e += 256) {
      const int o = (e % (TO / 2)) * 2, it2 = e / (TO / 2);
      const int i = it2 / TAPS, tap = it2 - i * TAPS;
      const float* t = tile + o * S + i * TAPS + tap;
      long long dst;
      if (TAPS == 1 && s2 == 4) {
        // the fifth tap slot of the block's merged stride-2 operand (rows of class (0,0), five slots per row)
        dst = ((long long)(i0 + i) * 5 + 4) * O + o0 + o;
      } else if (TAPS == 1 || !s2) {
        dst = ((long long)(i0 + i) * TAPS + tap) * O + o0 + o;
      } else {
        // parity-class layout of pack_dgrad_s2_kernel: class (ph,pw), taps {1} / {2,0} per axis
        const int kh = tap / 3, kw = tap - kh * 3;
        const int ph = kh != 1, pw = kw != 1;
        const int th = ph ? (kh == 2 ? 0 : 1) : 0, tw = pw ? (kw == 2 ? 0 : 1) : 0;
        const int nh = ph ? 2 : 1, nw = pw ? 2 : 1;
        const int cls = ph * 2 + pw;
        const long long IO = (long long)I * O;
        const long long base = cls == 0 ? 0 : (cls == 1 ? IO : (cls == 2 ? 3 * IO : 5 * IO));
        dst = base + ((long long)(i0 + i) * (nh * nw) + th * nw + tw) * O + o0 + o;
        // merged layout of pack_dgrad_s2m_kernel: rows (class, i), 2 x 2 tap slots (the unused ones stay zero)
        if (s2 == 2) dst = (((long long)cls * I + i0 + i) * 4 + th * 2 + tw) * O + o0 + o;
        if (s2 == 3) dst = (((long long)cls * I + i0 + i) * 5 + th * 2 + tw) * O + o0 + o;   // (five slots per row, slot 4: see above)
      }
      PackPair<T>::store(dg + dst, t[0], t[S]);
    }
  }
}

constexpr int PK_T1 = 64;  // tile edge of 1x1 filters; 3x3 filters use 32

template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(PackBatchArgs a) {
  __shared__ float tile[32 * (32 * 9 + 1)];
  static_assert(PK_T1 * (PK_T1 + 1) <= 32 * (32 * 9 + 1), "1x1 tile fits the 3x3 tile's LDS");
  int it = 0;
  while (it + 1 < a.n && (int)blockIdx.x >= a.first_block[it + 1]) ++it;
  const int O = a.O[it], I = a.I[it];
  const int b = blockIdx.x - a.first_block[it];
  T* fwd = static_cast<T*>(a.fwd[it]);
  T* dg = static_cast<T*>(a.dgrad[it]);
  if (a.k[it] == 1) {
    const int tiles_i = I / PK_T1;
    pack_tile<T, 1, PK_T1, PK_T1>(tile, a.w[it], fwd, dg, O, I, (b / tiles_i) * PK_T1, (b % tiles_i) * PK_T1, (int)a.s2[it], a.adam,
                                  a.g[it], a.m[it], a.v[it]);
  } else {
    const int tiles_i = I >> 5;
    pack_tile<T, 9, 32, 32>(tile, a.w[it], fwd, dg, O, I, (b / tiles_i) << 5, (b % tiles_i) << 5, (int)a.s2[it], a.adam,
                            a.g[it], a.m[it], a.v[it]);
  }
}

// Plain multi-tensor Adam for everything that has no packed copy (biases, BatchNorm affine
// parameters, the small linears, conv1): one launch, 4096 elements per block.
constexpr int AD_MAX_ITEMS = 48, AD_CHUNK = 4096;
struct AdamBatchArgs {
  AdamScalars adam;
  float* p[AD_MAX_ITEMS];
  const float* g[AD_MAX_ITEMS];
  float* m[AD_MAX_ITEMS];
  float* v[AD_MAX_ITEMS];
  long long numel[AD_MAX_ITEMS];
  int first_block[AD_MAX_ITEMS + 1];
  int n;
};
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamBatchArgs a) {
  int it = 0;
  while (it + 1 < a.n && (int)blockIdx.x >= a.first_block[it + 1]) ++it;
  const long long base = (long long)(blockIdx.x - a.first_block[it]) * AD_CHUNK;
  const long long n = a.numel[it];
  float* __restrict__ p = a.p[it];
  const float* __restrict__ g = a.g[it];
  float* __restrict__ m = a.m[it];
  float* __restrict__ v = a.v[it];
#pragma unroll 4
  for (int k = 0; k < AD_CHUNK / 256; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    if (i < n) {
      float mv = m[i], vv = v[i];
      p[i] = adam_update(p[i], g[i], mv, vv, a.adam);
      m[i] = mv;
      v[i] = vv;
    }
  }
}

// stem: [64][3][7][7] -> [64][taps rows][8 cols][4 ch] (rows >= 7, col 7 and ch 3 are zero)
template <typename T>
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int taps) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 64 * taps * 32) return;
  const int c = idx & 3, kw = (idx >> 2) & 7, kh = (idx >> 5) % taps, o = idx / (taps * 32);
  float v = 0.f;
  if (c < 3 && kw < 7 && kh < 7) v = w[((o * 3 + c) * 7 + kh) * 7 + kw];
  dst[idx] = qt_from_f32<T>(v);
}

// class-wise data-gradient operand of a stride-2 conv: dst[class][i][tap'][o]
template <typename T>
__global__ void pack_dgrad_s2_kernel(const float* __restrict__ w, T* __restrict__ dst, int O, int I, int k) {
  // taps per parity: k=3 -> {1} / {2,0};  k=1 -> {0} / {}
  const int n0 = 1, n1 = k == 3 ? 2 : 0;
  long long base = 0;
  for (int cls = 0; cls < 4; ++cls) {
    const int ph = cls >> 1, pw = cls & 1;
    const int nh = ph ? n1 : n0, nw = pw ? n1 : n0;
    const long long total = (long long)I * nh * nw * O;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
      const int o = (int)(idx % O);
      const int t = (int)((idx / O) % (nh * nw));
      const int i = (int)(idx / ((long long)O * nh * nw));
      const int th = t / nw, tw = t - th * nw;
      const int kh = k == 1 ? 0 : (ph ? (th == 0 ? 2 : 0) : 1);
      const int kw = k == 1 ? 0 : (pw ? (tw == 0 ? 2 : 0) : 1);
      dst[base + idx] = qt_from_f32<T>(w[(((long long)o * I + i) * k + kh) * k + kw]);
    }
    base += total;
  }
}

// all four parity classes as ONE operand of a 2x2-tap gather over the gradient map (qt_conv_desc.dst_merge):
// dst[(class*I + i)][th*2 + tw][o]; th / tw = 0: the tap on the row / column itself (kh / kw = 1 on even, 2 on odd
// pixels), 1: the tap on the next row / column (kh / kw = 0, odd pixels only).  Slots no tap maps to stay zero.
template <typename T>
__global__ void pack_dgrad_s2m_kernel(const float* __restrict__ w, T* __restrict__ dst, int O, int I) {
  const long long total = (long long)O * I * 9;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int o = (int)(idx % O);
    const int tap = (int)((idx / O) % 9);
    const int i = (int)(idx / ((long long)O * 9));
    const int kh = tap / 3, kw = tap - kh * 3;
    const int ph = kh != 1, pw = kw != 1;
    const int th = ph ? (kh == 2 ? 0 : 1) : 0, tw = pw ? (kw == 2 ? 0 : 1) : 0;
    const int cls = ph * 2 + pw;
    dst[(((long long)cls * I + i) * 4 + th * 2 + tw) * O + o] = qt_from_f32<T>(w[(((long long)o * I + i) * 3 + kh) * 3 + kw]);
  }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ dw, float* __restrict__ grad, int O, int I, int taps,
                                    int accumulate) {
  const long long total = (long long)O * I * taps;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(idx % I);
    const int tap = (int)((idx / I) % taps);
    const int o = (int)(idx / ((long long)I * taps));
    const long long g = ((long long)o * I + i) * taps + tap;
    grad[g] = accumulate ? grad[g] + dw[idx] : dw[idx];
  }
}

__global__ void unpack_stem_wgrad_kernel(const float* __restrict__ dw, float* __restrict__ grad, int accumulate) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over [64][3][7][7]
  if (idx >= 64 * 3 * 49) return;
  const int kw = idx % 7, kh = (idx / 7) % 7, c = (idx / 49) % 3, o = idx / 147;
  const float v = dw[((o * 7 + kh) * 8 + kw) * 4 + c];
  grad[idx] = accumulate ? grad[idx] + v : v;
}

int grid_for(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int qt_pack_stem_input(int dtype, const float* image_nchw, void* dst, int batch, void* stream) {
  QT_CHECK_ARG(image_nchw && dst && batch > 0, "qt_pack_stem_input: bad argument");
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_pack_stem_input: bad dtype %d", dtype);
  const long long total = (long long)batch * QT_STEM_PAD_H * QT_STEM_PAD_W;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == QT_F32)
    hipLaunchKernelGGL(pack_stem_input_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, image_nchw,
                       static_cast<float*>(dst), batch);
  else
    hipLaunchKernelGGL(pack_stem_input_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, image_nchw,
                       static_cast<bf16_t*>(dst), batch);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_pack_conv_weight(int dtype, const float* w_oihw, void* w_fwd, void* w_dgrad, int O, int I, int kh,
                                   int kw, void* stream) {
  QT_CHECK_ARG(w_oihw && (w_fwd || w_dgrad) && O > 0 && I > 0 && kh > 0 && kw > 0, "qt_pack_conv_weight: bad argument");
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_pack_conv_weight: bad dtype %d", dtype);
  const long long total = (long long)O * I * kh * kw;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == QT_F32)
    hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, w_oihw,
                       static_cast<float*>(w_fwd), static_cast<float*>(w_dgrad), O, I, kh * kw);
  else
    hipLaunchKernelGGL(pack_conv_weight_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, w_oihw,
                       static_cast<bf16_t*>(w_fwd), static_cast<bf16_t*>(w_dgrad), O, I, kh * kw);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

static int make_adam_scalars(const qt_adam_desc* d, AdamScalars* h) {
  QT_CHECK_ARG(d && d->step >= 1 && d->lr >= 0.f && d->beta1 >= 0.f && d->beta1 < 1.f && d->beta2 >= 0.f && d->beta2 < 1.f &&
                   d->eps >= 0.f && d->weight_decay >= 0.f,
               "qt_adam: bad hyper-parameters (step counts from 1)");
  h->lr = d->lr; h->b1 = d->beta1; h->b2 = d->beta2; h->eps = d->eps; h->wd = d->weight_decay;
  const double bc1 = 1.0 - pow((double)d->beta1, (double)d->step), bc2 = 1.0 - pow((double)d->beta2, (double)d->step);
  h->step_size = (float)((double)d->lr / bc1);
  h->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  h->grad_scale = d->grad_scale == 0.f ? 1.f : d->grad_scale;
  h->enabled = 1;
  return QT_OK;
}

static int pack_batched(int dtype, const qt_pack_item* items, const qt_adam_item* opt, const qt_adam_desc* adam, int n,
                        void* stream);

extern "C" int qt_pack_weights_batched(int dtype, const qt_pack_item* items, int n, void* stream) {
  return pack_batched(dtype, items, nullptr, nullptr, n, stream);
}

extern "C" int qt_adam_pack_weights_batched(int dtype, const qt_pack_item* items, const qt_adam_item* opt,
                                            const qt_adam_desc* adam, int n, void* stream) {
  QT_CHECK_ARG(opt && adam, "qt_adam_pack_weights_batched: null optimizer state");
  return pack_batched(dtype, items, opt, adam, n, stream);
}

extern "C" int qt_adam_multi(const qt_adam_item* items, int n, const qt_adam_desc* adam, void* stream) {
  QT_CHECK_ARG(items && n > 0, "qt_adam_multi: no tensors");
  AdamScalars h;
  if (int st = make_adam_scalars(adam, &h)) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int j0 = 0; j0 < n; j0 += AD_MAX_ITEMS) {
    AdamBatchArgs a;
    memset(&a, 0, sizeof(a));
    a.adam = h;
    const int cnt = n - j0 < AD_MAX_ITEMS ? n - j0 : AD_MAX_ITEMS;
    int blocks = 0;
    for (int j = 0; j < cnt; ++j) {
      const qt_adam_item& q = items[j0 + j];
      QT_CHECK_ARG(q.param && q.grad && q.exp_avg && q.exp_avg_sq && q.numel > 0, "qt_adam_multi: item %d incomplete", j0 + j);
      a.p[j] = q.param; a.g[j] = q.grad; a.m[j] = q.exp_avg; a.v[j] = q.exp_avg_sq; a.numel[j] = q.numel;
      a.first_block[j] = blocks;
      blocks += (int)((q.numel + AD_CHUNK - 1) / AD_CHUNK);
    }
    a.first_block[cnt] = blocks;
    a.n = cnt;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, s, a);
    QT_CHECK_LAUNCH();
  }
  return QT_OK;
}

static int pack_batched(int dtype, const qt_pack_item* items, const qt_adam_item* opt, const qt_adam_desc* adam, int n,
                        void* stream) {
  QT_CHECK_ARG(items && n > 0 && n <= PK_MAX_ITEMS, "qt_pack_weights_batched: 1..%d items", PK_MAX_ITEMS);
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_pack_weights_batched: bad dtype %d", dtype);
  PackBatchArgs a;
  memset(&a, 0, sizeof(a));
  if (opt) {
    if (int st = make_adam_scalars(adam, &a.adam)) return st;
  }
  int blocks = 0;
  for (int j = 0; j < n; ++j) {
    const qt_pack_item& q = items[j];
    const int te = q.k == 1 ? PK_T1 : 32;
    QT_CHECK_ARG(q.w_oihw && (q.w_fwd || q.w_dgrad) && (q.k == 1 || q.k == 3) && q.O > 0 && q.I > 0 && q.O % te == 0 &&
                     q.I % te == 0,
                 "qt_pack_weights_batched: item %d: O=%d I=%d must be multiples of %d for k=%d (k in {1,3})", j, q.O, q.I,
                 te, q.k);
    a.w[j] = q.w_oihw; a.fwd[j] = q.w_fwd; a.dgrad[j] = q.w_dgrad;
    if (opt) {
      const qt_adam_item& u = opt[j];
      QT_CHECK_ARG(u.param == q.w_oihw && u.grad && u.exp_avg && u.exp_avg_sq &&
                       u.numel == (long long)q.O * q.I * q.k * q.k,
                   "qt_adam_pack_weights_batched: item %d: optimizer state does not match the weight", j);
      a.g[j] = u.grad; a.m[j] = u.exp_avg; a.v[j] = u.exp_avg_sq;
    }
    a.O[j] = q.O; a.I[j] = q.I; a.k[j] = (unsigned char)q.k; a.s2[j] = (unsigned char)((q.stride2_dgrad >= 2 && q.stride2_dgrad <= 4) ? q.stride2_dgrad : (q.stride2_dgrad ? 1 : 0));
    QT_CHECK_ARG(q.stride2_dgrad != 4 || q.k == 1, "qt_pack_weights_batched: item %d: stride2_dgrad = 4 is for 1x1 filters", j);
    QT_CHECK_ARG(q.stride2_dgrad != 3 || q.k == 3, "qt_pack_weights_batched: item %d: stride2_dgrad = 3 is for 3x3 filters", j);
    a.first_block[j] = blocks;
    blocks += (q.O / te) * (q.I / te);
  }
  a.first_block[n] = blocks;
  a.n = n;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == QT_F32)
    hipLaunchKernelGGL(pack_weights_batched_kernel<float>, dim3(blocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(pack_weights_batched_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, a);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_pack_dgrad_s2(int dtype, const float* w_oihw, void* dst, int O, int I, int k, long long* class_offset,
                                int* class_kh, int* class_kw, void* stream) {
  QT_CHECK_ARG(w_oihw && O > 0 && I > 0 && (k == 1 || k == 3), "qt_pack_dgrad_s2: bad argument");
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_pack_dgrad_s2: bad dtype %d", dtype);
  long long base = 0;
  for (int cls = 0; cls < 4; ++cls) {
    const int nh = (cls >> 1) ? (k == 3 ? 2 : 0) : 1, nw = (cls & 1) ? (k == 3 ? 2 : 0) : 1;
    if (class_offset) class_offset[cls] = base;
    if (class_kh) class_kh[cls] = nh;
    if (class_kw) class_kw[cls] = nw;
    base += (long long)I * nh * nw * O;
  }
  if (!dst) return QT_OK;  // layout query only
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long total = (long long)O * I * k * k;
  if (dtype == QT_F32)
    hipLaunchKernelGGL(pack_dgrad_s2_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, w_oihw,
                       static_cast<float*>(dst), O, I, k);
  else
    hipLaunchKernelGGL(pack_dgrad_s2_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, w_oihw,
                       static_cast<bf16_t*>(dst), O, I, k);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_pack_dgrad_s2_merged(int dtype, const float* w_oihw, void* dst, int O, int I, void* stream) {
  QT_CHECK_ARG(w_oihw && dst && O > 0 && I > 0, "qt_pack_dgrad_s2_merged: bad argument");
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_pack_dgrad_s2_merged: bad dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t esz = dtype == QT_F32 ? 4 : 2;
  if (hipMemsetAsync(dst, 0, (size_t)16 * O * I * esz, s) != hipSuccess) {
    qt_set_error("qt_pack_dgrad_s2_merged: memset failed");
    return QT_ERR_LAUNCH;
  }
  const long long total = (long long)O * I * 9;
  if (dtype == QT_F32)
    hipLaunchKernelGGL(pack_dgrad_s2m_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, w_oihw, static_cast<float*>(dst), O, I);
  else
    hipLaunchKernelGGL(pack_dgrad_s2m_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, w_oihw, static_cast<bf16_t*>(dst), O, I);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_pack_stem_weight(int dtype, const float* w_oihw, void* dst, int taps, void* stream) {
  QT_CHECK_ARG(w_oihw && dst && (taps == 7 || taps == 8), "qt_pack_stem_weight: bad argument");
  QT_CHECK_ARG(dtype == QT_F32 || dtype == QT_BF16, "qt_pack_stem_weight: bad dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int total = 64 * taps * 32;
  if (dtype == QT_F32)
    hipLaunchKernelGGL(pack_stem_weight_kernel<float>, dim3(qt_cdiv(total, 256)), dim3(256), 0, s, w_oihw,
                       static_cast<float*>(dst), taps);
  else
    hipLaunchKernelGGL(pack_stem_weight_kernel<bf16_t>, dim3(qt_cdiv(total, 256)), dim3(256), 0, s, w_oihw,
                       static_cast<bf16_t*>(dst), taps);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_unpack_conv_wgrad(const float* dw, float* grad_oihw, int O, int I, int kh, int kw, int accumulate,
                                    void* stream) {
  QT_CHECK_ARG(dw && grad_oihw && O > 0 && I > 0 && kh > 0 && kw > 0, "qt_unpack_conv_wgrad: bad argument");
  const long long total = (long long)O * I * kh * kw;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), dw,
                     grad_oihw, O, I, kh * kw, accumulate);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_unpack_stem_wgrad(const float* dw, float* grad_oihw, int accumulate, void* stream) {
  QT_CHECK_ARG(dw && grad_oihw, "qt_unpack_stem_wgrad: null argument");
  hipLaunchKernelGGL(unpack_stem_wgrad_kernel, dim3(qt_cdiv(64 * 147, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dw, grad_oihw, accumulate);
  QT_CHECK_LAUNCH();
  return QT_OK;
}
