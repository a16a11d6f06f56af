// Head kernels of AttentionHierarchicalCNN (/root/reference/Quadtree_from scratch/models.py:6-101):
//   * region average pool: AdaptiveAvgPool2d((1,1)) of the per-region conv+ReLU maps of the quadrant
//     (:21-24, 4 regions of 14x14) and sub-quadrant (:27-30, 16 regions of 7x7) processors, written in the
//     reference's concatenation order (:62-78), and its backward fused with the ReLU mask;
//   * attention gate (:34-38, :81-89): Linear(64,32) -> ReLU -> Linear(32,1) per sub-quadrant vector,
//     softmax over the 16 scores, weighted sum of the vectors -- forward and backward, one wave per image;
//   * a strided ReLU/dropout mask for the single-layer numerical MLP (:43-46) that lives inside the fused
//     feature matrix.
// All HBM / latency bound (the 16x64 vectors of an image fit in 4 KB of LDS); nothing here is GEMM-shaped.
#include <type_traits>

#include "qt_common.h"

namespace {

int grid_for(long long total, int block = 256, int cap = 16384) {
  long long g = (total + block - 1) / block;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

// The convolution numbers the S x S regions of an image row-major (rr*S + rc).  The reference appends
// quadrants in the order TL, TR, BL, BR (:62-65) and, for each quadrant in that order, its four
// sub-quadrants in the same order (:70-78): slot = quadrant*4 + sub-quadrant.
__device__ __forceinline__ int region_slot(int r, int S) {
  if (S == 2) return r;
  const int rr = r >> 2, rc = r & 3;
  return (((rr >> 1) * 2 + (rc >> 1)) << 2) + (rr & 1) * 2 + (rc & 1);
}

template <typename D> __device__ __forceinline__ void store8(D* p, const float (&v)[8]) { QtVec8<D>::store(p, v); }
template <typename D> __device__ __forceinline__ void load8(const D* p, float (&v)[8]) { QtVec8<D>::load(p, v); }

// x [B*R][HW][C] (conv + bias + ReLU applied) -> dst[b*ld + col0 + slot(r)*C + c] = mean over HW.
// Four lanes share one (image, 8-channel group): each sums a quarter of the positions.
template <typename T, typename D>
__global__ __launch_bounds__(256) void region_avgpool_kernel(const T* __restrict__ x, D* __restrict__ dst, int nimg, int S,
                                                             int HW, int C, int ld, int col0) {
  const int cgs = C >> 3, R = S * S;
  const long long total = (long long)nimg * cgs * 4;
  const float inv = 1.f / (float)HW;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const bool live = i < total;
  const int part = (int)(i & 3);
  const long long u = live ? (i >> 2) : 0;
  const int c0 = (int)(u % cgs) * 8, img = (int)(u / cgs);
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (live) {
    const T* base = x + (long long)img * HW * C + c0;
#pragma unroll 7
    for (int p = part; p < HW; p += 4) {
      float v[8];
      QtVec8<T>::load(base + (long long)p * C, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s[e] += __shfl_xor(s[e], 1, 64);
    s[e] += __shfl_xor(s[e], 2, 64);
    s[e] *= inv;
  }
  if (live && part == 0) {
    const int b = img / R, r = img - b * R;
    store8<D>(dst + (long long)b * ld + col0 + region_slot(r, S) * C + c0, s);
  }
}

// g[img][p][c] = x > 0 ? d[b*ld + col0 + slot*C + c] / HW : 0
template <typename T, typename D>
__global__ void region_avgpool_bwd_kernel(const D* __restrict__ d, const T* __restrict__ x, T* __restrict__ g, int nimg, int S,
                                          int HW, int C, int ld, int col0) {
  const int cgs = C >> 3, R = S * S;
  const long long total = (long long)nimg * HW * cgs;
  const float inv = 1.f / (float)HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cgs) * 8;
    const int img = (int)(i / ((long long)cgs * HW));
    const int b = img / R, r = img - b * R;
    float dv[8], xv[8];
    load8<D>(d + (long long)b * ld + col0 + region_slot(r, S) * C + c0, dv);
    QtVec8<T>::load(x + i * 8, xv);
#pragma unroll
    for (int e = 0; e < 8; ++e) dv[e] = xv[e] > 0.f ? dv[e] * inv : 0.f;
    QtVec8<T>::store(g + i * 8, dv);
  }
}

constexpr int NV = 16, DV = 64, DH = 32;  // vectors per image, vector length, hidden width of the gate
constexpr int VP = DV + 1;                // padded LDS row (lanes that walk j with a fixed column hit distinct banks)

// One wave per image.  v [B][16][64] f32 -> act [B][16][32] (post-ReLU hidden), alpha [B][16] (softmax weights),
// out[b*ld + col0 + c] = sum_j alpha_j v_j[c].
template <typename T>
__global__ __launch_bounds__(64) void attention_fwd_kernel(const float* __restrict__ v, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, float* __restrict__ act,
                                                           float* __restrict__ alpha, T* __restrict__ out, int ld,
                                                           int col0) {
  __shared__ float sv[NV * VP];
  __shared__ float sw[DV * (DH + 1)];  // w1 transposed: [k][h]
  __shared__ float sa[NV * (DH + 1)];
  __shared__ float ss[NV];
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* vb = v + (long long)b * NV * DV;
#pragma unroll
  for (int j = 0; j < NV; ++j) sv[j * VP + lane] = vb[j * DV + lane];
#pragma unroll
  for (int h = 0; h < DH; ++h) sw[lane * (DH + 1) + h] = w1[h * DV + lane];
  __syncthreads();
  {
    const int h = lane & 31, j0 = (lane >> 5) * 8;
    float a[8];
    const float bias = b1[h];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = bias;
    for (int k = 0; k < DV; ++k) {
      const float w = sw[k * (DH + 1) + h];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += w * sv[(j0 + j) * VP + k];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float r = fmaxf(a[j], 0.f);
      sa[(j0 + j) * (DH + 1) + h] = r;
      act[((long long)b * NV + j0 + j) * DH + h] = r;
    }
  }
  __syncthreads();
  if (lane < NV) {
    float s = b2[0];
    for (int h = 0; h < DH; ++h) s += w2[h] * sa[lane * (DH + 1) + h];
    ss[lane] = s;
  }
  __syncthreads();
  float mx = ss[0];
#pragma unroll
  for (int j = 1; j < NV; ++j) mx = fmaxf(mx, ss[j]);
  float e[NV], den = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    e[j] = __expf(ss[j] - mx);
    den += e[j];
  }
  const float rden = 1.f / den;
  float o = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) o += (e[j] * rden) * sv[j * VP + lane];
  out[(long long)b * ld + col0 + lane] = (T)o;
  if (lane < NV) alpha[(long long)b * NV + lane] = e[lane] * rden;
}

// One wave per image.  dout = d[b*ld + col0 + c].
//   dalpha_j = dout . v_j;  ds_j = alpha_j (dalpha_j - sum_k alpha_k dalpha_k);  dpre_j = ds_j w2 (act_j > 0);
//   dv_j = alpha_j dout + W1^T dpre_j.
// ds [B][16] and dpre [B][16][32] leave for the parameter gradients (thin products over B*16 rows).
template <typename T>
__global__ __launch_bounds__(64) void attention_bwd_kernel(const T* __restrict__ d, const float* __restrict__ v,
                                                           const float* __restrict__ act, const float* __restrict__ alpha,
                                                           const float* __restrict__ w1, const float* __restrict__ w2,
                                                           float* __restrict__ ds, float* __restrict__ dpre,
                                                           float* __restrict__ dv, int ld, int col0) {
  __shared__ float sv[NV * VP];
  __shared__ float sp[NV * (DH + 1)];
  __shared__ float sd[DV];
  __shared__ float sda[NV];
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* vb = v + (long long)b * NV * DV;
#pragma unroll
  for (int j = 0; j < NV; ++j) sv[j * VP + lane] = vb[j * DV + lane];
  const float dout = qt_to_f32<T>(d[(long long)b * ld + col0 + lane]);
  sd[lane] = dout;
  __syncthreads();
  if (lane < NV) {
    float s = 0.f;
    for (int c = 0; c < DV; ++c) s += sd[c] * sv[lane * VP + c];
    sda[lane] = s;
  }
  __syncthreads();
  float al[NV], t = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    al[j] = alpha[(long long)b * NV + j];
    t += al[j] * sda[j];
  }
  float dsj[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) dsj[j] = al[j] * (sda[j] - t);
  if (lane < NV) {
    float mine = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) mine = lane == j ? dsj[j] : mine;
    ds[(long long)b * NV + lane] = mine;
  }
  {
    const int h = lane & 31, j0 = (lane >> 5) * 8;
    const float w = w2[h];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float dj = 0.f;
#pragma unroll
      for (int q = 0; q < NV; ++q) dj = (j0 + j) == q ? dsj[q] : dj;
      const long long idx = ((long long)b * NV + j0 + j) * DH + h;
      const float p = act[idx] > 0.f ? dj * w : 0.f;
      sp[(j0 + j) * (DH + 1) + h] = p;
      dpre[idx] = p;
    }
  }
  __syncthreads();
  float g[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) g[j] = al[j] * dout;
  for (int h = 0; h < DH; ++h) {
    const float w = w1[h * DV + lane];
#pragma unroll
    for (int j = 0; j < NV; ++j) g[j] += w * sp[j * (DH + 1) + h];
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) dv[((long long)b * NV + j) * DV + lane] = g[j];
}

// out[r][c] = act[r*ld + col0 + c] > 0 ? d[r*ld + col0 + c] * mul : 0     (f32 out, dense [rows][cols])
template <typename T>
__global__ void relu_mask_cols_kernel(const T* __restrict__ d, const T* __restrict__ act, float* __restrict__ out,
                                      long long rows, int cols, int ld, int col0, float mul) {
  const long long total = rows * cols;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols);
    const long long r = i / cols, src = r * ld + col0 + c;
    out[i] = qt_to_f32<T>(act[src]) > 0.f ? qt_to_f32<T>(d[src]) * mul : 0.f;
  }
}

template <typename F> void by_dtype(int dtype, F&& f) {
  if (dtype == QT_F32)
    f(static_cast<float*>(nullptr));
  else
    f(static_cast<bf16_t*>(nullptr));
}
#define QT_T(tag) std::remove_pointer_t<decltype(tag)>
#define QT_DT_OK(dtype, name) QT_CHECK_ARG((dtype) == QT_F32 || (dtype) == QT_BF16, name ": bad dtype %d", (dtype))

}  // namespace

extern "C" int qt_region_avgpool(int dtype, const void* x, void* dst, int dst_dtype, int batch, int split, int hw, int C,
                                 int ld, int col0, void* stream) {
  QT_DT_OK(dtype, "qt_region_avgpool");
  QT_CHECK_ARG(dst_dtype == dtype || dst_dtype == QT_F32, "qt_region_avgpool: dst_dtype must be the map's dtype or f32");
  QT_CHECK_ARG(x && dst && batch > 0 && (split == 2 || split == 4) && hw > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0 &&
                   col0 % 8 == 0 && col0 >= 0 && col0 + split * split * C <= ld,
               "qt_region_avgpool: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nimg = batch * split * split;
  const int grid = (int)(((long long)nimg * (C / 8) * 4 + 255) / 256);
  by_dtype(dtype, [&](auto tag) {
    using T = QT_T(tag);
    if (dst_dtype == dtype)
      hipLaunchKernelGGL((region_avgpool_kernel<T, T>), dim3(grid), dim3(256), 0, s, (const T*)x, (T*)dst, nimg, split, hw,
                         C, ld, col0);
    else
      hipLaunchKernelGGL((region_avgpool_kernel<T, float>), dim3(grid), dim3(256), 0, s, (const T*)x, (float*)dst, nimg,
                         split, hw, C, ld, col0);
  });
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_region_avgpool_bwd(int dtype, const void* d, int d_dtype, const void* x, void* g, int batch, int split,
                                     int hw, int C, int ld, int col0, void* stream) {
  QT_DT_OK(dtype, "qt_region_avgpool_bwd");
  QT_CHECK_ARG(d_dtype == dtype || d_dtype == QT_F32, "qt_region_avgpool_bwd: d_dtype must be the map's dtype or f32");
  QT_CHECK_ARG(d && x && g && batch > 0 && (split == 2 || split == 4) && hw > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0 &&
                   col0 % 8 == 0 && col0 >= 0 && col0 + split * split * C <= ld,
               "qt_region_avgpool_bwd: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nimg = batch * split * split;
  const int grid = grid_for((long long)nimg * hw * (C / 8));
  by_dtype(dtype, [&](auto tag) {
    using T = QT_T(tag);
    if (d_dtype == dtype)
      hipLaunchKernelGGL((region_avgpool_bwd_kernel<T, T>), dim3(grid), dim3(256), 0, s, (const T*)d, (const T*)x, (T*)g,
                         nimg, split, hw, C, ld, col0);
    else
      hipLaunchKernelGGL((region_avgpool_bwd_kernel<T, float>), dim3(grid), dim3(256), 0, s, (const float*)d, (const T*)x,
                         (T*)g, nimg, split, hw, C, ld, col0);
  });
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_attention_gate(int dtype, const float* v, const float* w1, const float* b1, const float* w2,
                                 const float* b2, float* act, float* alpha, void* out, int batch, int ld, int col0,
                                 void* stream) {
  QT_DT_OK(dtype, "qt_attention_gate");
  QT_CHECK_ARG(v && w1 && b1 && w2 && b2 && act && alpha && out && batch > 0 && col0 >= 0 && col0 + DV <= ld,
               "qt_attention_gate: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  by_dtype(dtype, [&](auto tag) {
    using T = QT_T(tag);
    hipLaunchKernelGGL(attention_fwd_kernel<T>, dim3(batch), dim3(64), 0, s, v, w1, b1, w2, b2, act, alpha, (T*)out, ld,
                       col0);
  });
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_attention_gate_bwd(int dtype, const void* d, const float* v, const float* act, const float* alpha,
                                     const float* w1, const float* w2, float* ds, float* dpre, float* dv, int batch, int ld,
                                     int col0, void* stream) {
  QT_DT_OK(dtype, "qt_attention_gate_bwd");
  QT_CHECK_ARG(d && v && act && alpha && w1 && w2 && ds && dpre && dv && batch > 0 && col0 >= 0 && col0 + DV <= ld,
               "qt_attention_gate_bwd: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  by_dtype(dtype, [&](auto tag) {
    using T = QT_T(tag);
    hipLaunchKernelGGL(attention_bwd_kernel<T>, dim3(batch), dim3(64), 0, s, (const T*)d, v, act, alpha, w1, w2, ds, dpre,
                       dv, ld, col0);
  });
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_relu_mask_cols(int dtype, const void* d, const void* act, float* out, long long rows, int cols, int ld,
                                 int col0, float mul, void* stream) {
  QT_DT_OK(dtype, "qt_relu_mask_cols");
  QT_CHECK_ARG(d && act && out && rows > 0 && cols > 0 && col0 >= 0 && col0 + cols <= ld, "qt_relu_mask_cols: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  by_dtype(dtype, [&](auto tag) {
    using T = QT_T(tag);
    hipLaunchKernelGGL(relu_mask_cols_kernel<T>, dim3(grid_for(rows * cols)), dim3(256), 0, s, (const T*)d, (const T*)act,
                       out, rows, cols, ld, col0, mul);
  });
  QT_CHECK_LAUNCH();
  return QT_OK;
}
