// Layout packing kernels (HBM-bound, coalesced): input image, weights, weight gradients.
//
//  * NCHW f32 image -> zero-padded NHWC4 tensor for the packed 7x7/2 stem
//    (input contract: /root/reference/Quadtree_from scratch/dataloader.py:72-91).
//  * OIHW f32 master weights (the reference's state_dict layout)
//    -> [O][kh][kw][I] forward operand and [I][kh][kw][O] data-gradient operand.
//  * [O][kh][kw][I] f32 weight gradients -> OIHW f32 .grad tensors.
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
