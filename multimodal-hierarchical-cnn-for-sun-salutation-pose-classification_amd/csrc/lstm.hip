// nn.LSTM (batch_first, one direction) recurrence of the CNN+LSTM sequence model
// (/root/reference/cnn+lstm/models.py:43-49,81-86), forward and backward-through-time, f32.
//
// The input products of all time steps (x_t W_ih^T + b_ih) and the parameter gradients are dense products done
// outside (plan.hip: the f32 MFMA instantiations of the conv kernels on 1x1 "images"); what is sequential lives
// here: ONE launch per layer and direction of time, one workgroup of 4H threads per sequence, the state stays in
// registers / LDS for all T steps and W_hh (H x 4H f32, 1 MB at H = 256) is re-read from L2 every step.
// torch gate order: i, f, g, o (rows [0,H), [H,2H), [2H,3H), [3H,4H) of W_ih / W_hh / biases).
// Latency bound by construction (T dependent steps); the per-frame ResNet-18 in front of it is >99.9 % of the FLOPs.
#include "qt_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// xproj [B][T][4H] (input product + b_ih; b_hh is added here when given) -> gates [B][T][4H] (post-activation i,f,g,o),
// cell [B][T][H], hprev [B][T][H] (h_{t-1}, zeros at t = 0), hout [B][T][H].  whhT = W_hh^T [H][4H].
// 4H threads: thread (k, u) owns gate row k*H + u (one coalesced stream over W_hh^T: a step is bound by how fast ONE
// CU pulls the 4H x H matrix from L2, so all four gate rows are in flight at once); threads of k = 0 then own unit u.
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_fwd_kernel(const float* __restrict__ xproj, const float* __restrict__ whhT,
                                                         const float* __restrict__ bhh, float* __restrict__ gates,
                                                         float* __restrict__ cell, float* __restrict__ hprev,
                                                         float* __restrict__ hout, int T) {
  __shared__ float sh[H];
  __shared__ float sa[4 * H];
  const int b = blockIdx.x, tid = threadIdx.x, u = tid % H;
  const bool owner = tid < H;
  float c = 0.f, h = 0.f;
  const float bias = bhh ? bhh[tid] : 0.f;
  for (int t = 0; t < T; ++t) {
    const long long row = (long long)b * T + t;
    if (owner) {
      sh[u] = h;
      hprev[row * H + u] = h;
    }
    __syncthreads();
    float a = xproj[row * 4 * H + tid] + bias;
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const float* w = whhT + tid;  // consecutive tid: coalesced
#pragma unroll 4
    for (int m = 0; m < H; m += 4) {
      a += w[(long long)m * 4 * H] * sh[m];
      a1 += w[(long long)(m + 1) * 4 * H] * sh[m + 1];
      a2 += w[(long long)(m + 2) * 4 * H] * sh[m + 2];
      a3 += w[(long long)(m + 3) * 4 * H] * sh[m + 3];
    }
    a = (a + a1) + (a2 + a3);
    const float act = (tid >= 2 * H && tid < 3 * H) ? tanhf(a) : sigmoidf_(a);
    sa[tid] = act;
    gates[row * 4 * H + tid] = act;
    __syncthreads();
    if (owner) {
      const float gi = sa[u], gf = sa[H + u], gg = sa[2 * H + u], go = sa[3 * H + u];
      c = gf * c + gi * gg;
      h = go * tanhf(c);
      cell[row * H + u] = c;
      hout[row * H + u] = h;
    }
  }
}

// dhout [B][T][H] (gradient w.r.t. every h_t from above; NULL = zero) and dlast [B][H] (extra gradient of the
// last step; NULL = zero) -> dgates [B][T][4H] (w.r.t. the PRE-activation gates).  whh = W_hh [4H][H].
// 4H threads: the owners (k = 0) do the cell arithmetic of unit u; the recurrent product dh = dgates W_hh is split
// over the four gate blocks (thread (k, u) sums rows k*H .. k*H+H-1 of column u) and folded through LDS.
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_bwd_kernel(const float* __restrict__ dhout, const float* __restrict__ dlast,
                                                         const float* __restrict__ gates, const float* __restrict__ cell,
                                                         const float* __restrict__ whh, float* __restrict__ dgates, int T) {
  __shared__ float sg[4 * H];
  __shared__ float sp[4 * H];
  const int b = blockIdx.x, tid = threadIdx.x, u = tid % H, k = tid / H;
  const bool owner = tid < H;
  float dh_rec = 0.f, dc_next = 0.f;
  for (int t = T - 1; t >= 0; --t) {
    const long long row = (long long)b * T + t;
    if (owner) {
      float dh = dh_rec;
      if (dhout) dh += dhout[row * H + u];
      if (dlast && t == T - 1) dh += dlast[(long long)b * H + u];
      const float* g = gates + row * 4 * H;
      const float gi = g[u], gf = g[H + u], gg = g[2 * H + u], go = g[3 * H + u];
      const float c = cell[row * H + u];
      const float cp = t > 0 ? cell[(row - 1) * H + u] : 0.f;
      const float tc = tanhf(c);
      const float dc = dc_next + dh * go * (1.f - tc * tc);
      const float dai = dc * gg * gi * (1.f - gi);
      const float daf = dc * cp * gf * (1.f - gf);
      const float dag = dc * gi * (1.f - gg * gg);
      const float dao = dh * tc * go * (1.f - go);
      dc_next = dc * gf;
      float* dg = dgates + row * 4 * H;
      dg[u] = dai; dg[H + u] = daf; dg[2 * H + u] = dag; dg[3 * H + u] = dao;
      sg[u] = dai; sg[H + u] = daf; sg[2 * H + u] = dag; sg[3 * H + u] = dao;
    }
    __syncthreads();
    float acc = 0.f, acc1 = 0.f;
    const float* w = whh + (long long)k * H * H + u;  // consecutive u: coalesced
#pragma unroll 4
    for (int j = 0; j < H; j += 2) {
      acc += sg[k * H + j] * w[(long long)j * H];
      acc1 += sg[k * H + j + 1] * w[(long long)(j + 1) * H];
    }
    sp[tid] = acc + acc1;
    __syncthreads();
    if (owner) dh_rec = (sp[u] + sp[H + u]) + (sp[2 * H + u] + sp[3 * H + u]);
    // (sg / sp are rewritten only after the next step's barriers: owners write sg before the first one, and every
    //  thread has passed the second barrier of this step before any owner gets there)
  }
}

__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[i][threadIdx.x] = src[(long long)r * cols + c];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < rows && c < cols) dst[(long long)c * rows + r] = tile[threadIdx.x][i];
  }
}

template <typename T>
__global__ void cast_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = qt_to_f32<T>(src[i]);
}

// g[i] = x[i] != 0 ? g[i] * mul : 0 : backward of the dropout nn.LSTM applies between its layers, from the
// dropped activations themselves (an h_t that is exactly 0.0f before dropout has measure zero)
__global__ void scale_by_nonzero_kernel(float* __restrict__ g, const float* __restrict__ x, long long n, float mul) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    g[i] = x[i] != 0.f ? g[i] * mul : 0.f;
}

}  // namespace

extern "C" int qt_lstm_forward(const float* xproj, const float* whh_t, const float* bhh, float* gates, float* cell,
                               float* hprev, float* hout, int batch, int T, int H, void* stream) {
  QT_CHECK_ARG(xproj && whh_t && gates && cell && hprev && hout && batch > 0 && T > 0, "qt_lstm_forward: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (H == 256)
    hipLaunchKernelGGL(lstm_fwd_kernel<256>, dim3(batch), dim3(1024), 0, s, xproj, whh_t, bhh, gates, cell, hprev, hout, T);
  else if (H == 64)
    hipLaunchKernelGGL(lstm_fwd_kernel<64>, dim3(batch), dim3(256), 0, s, xproj, whh_t, bhh, gates, cell, hprev, hout, T);
  else if (H == 188)   // Quadtree3DCNN: nn.LSTM(47, 47 * 4) (3dcnn/models.py:146-152)
    hipLaunchKernelGGL(lstm_fwd_kernel<188>, dim3(batch), dim3(752), 0, s, xproj, whh_t, bhh, gates, cell, hprev, hout, T);
  else {
    qt_set_error("qt_lstm_forward: hidden size %d is not instantiated (256, 188, 64)", H);
    return QT_ERR_UNSUPPORTED;
  }
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_lstm_backward(const float* dhout, const float* dlast, const float* gates, const float* cell,
                                const float* whh, float* dgates, int batch, int T, int H, void* stream) {
  QT_CHECK_ARG(gates && cell && whh && dgates && batch > 0 && T > 0, "qt_lstm_backward: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (H == 256)
    hipLaunchKernelGGL(lstm_bwd_kernel<256>, dim3(batch), dim3(1024), 0, s, dhout, dlast, gates, cell, whh, dgates, T);
  else if (H == 64)
    hipLaunchKernelGGL(lstm_bwd_kernel<64>, dim3(batch), dim3(256), 0, s, dhout, dlast, gates, cell, whh, dgates, T);
  else if (H == 188)
    hipLaunchKernelGGL(lstm_bwd_kernel<188>, dim3(batch), dim3(752), 0, s, dhout, dlast, gates, cell, whh, dgates, T);
  else {
    qt_set_error("qt_lstm_backward: hidden size %d is not instantiated (256, 188, 64)", H);
    return QT_ERR_UNSUPPORTED;
  }
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_transpose_f32(const float* src, float* dst, int rows, int cols, void* stream) {
  QT_CHECK_ARG(src && dst && rows > 0 && cols > 0, "qt_transpose_f32: bad argument");
  hipLaunchKernelGGL(transpose_kernel, dim3(qt_cdiv(cols, 32), qt_cdiv(rows, 32)), dim3(32, 8), 0,
                     static_cast<hipStream_t>(stream), src, dst, rows, cols);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_scale_by_nonzero(float* g, const float* x, long long n, float mul, void* stream) {
  QT_CHECK_ARG(g && x && n > 0, "qt_scale_by_nonzero: bad argument");
  long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(scale_by_nonzero_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g, x, n, mul);
  QT_CHECK_LAUNCH();
  return QT_OK;
}

extern "C" int qt_cast_f32(int dtype, const void* src, float* dst, long long n, void* stream) {
  QT_CHECK_ARG((dtype == QT_F32 || dtype == QT_BF16) && src && dst && n > 0, "qt_cast_f32: bad argument");
  long long blocks = (n + 255) / 256;
  const dim3 grid((unsigned)(blocks > 8192 ? 8192 : blocks));
  if (dtype == QT_F32)
    hipLaunchKernelGGL(cast_f32_kernel<float>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), (const float*)src, dst, n);
  else
    hipLaunchKernelGGL(cast_f32_kernel<bf16_t>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16_t*)src, dst, n);
  QT_CHECK_LAUNCH();
  return QT_OK;
}
