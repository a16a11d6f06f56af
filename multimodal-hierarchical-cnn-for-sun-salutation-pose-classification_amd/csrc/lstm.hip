// nn.LSTM (batch_first, one direction) recurrence of the CNN+LSTM sequence model
// (/root/reference/cnn+lstm/models.py:43-49,81-86), forward and backward-through-time, f32.
//
// The input products of all time steps (x_t W_ih^T + b_ih + b_hh) and the parameter gradients are thin
// dense products done outside (qt_gemm_small / qt_col_sum); what is sequential lives here: ONE launch per
// layer and direction of time, one workgroup per sequence, thread u owns hidden unit u, the state stays in
// registers / LDS for all T steps and W_hh (H x 4H f32, 1 MB at H = 256) is re-read from L2 every step.
// torch gate order: i, f, g, o (rows [0,H), [H,2H), [2H,3H), [3H,4H) of W_ih / W_hh / biases).
// Latency bound by construction (T dependent steps); the per-frame ResNet-18 in front of it is >99.9 % of the FLOPs.
#include "qt_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// xproj [B][T][4H] (input product + b_ih; b_hh is added here when given) -> gates [B][T][4H] (post-activation i,f,g,o),
// cell [B][T][H], hprev [B][T][H] (h_{t-1}, zeros at t = 0), hout [B][T][H].  whhT = W_hh^T [H][4H].
template <int H>
__global__ __launch_bounds__(H) void lstm_fwd_kernel(const float* __restrict__ xproj, const float* __restrict__ whhT,
                                                     const float* __restrict__ bhh, float* __restrict__ gates, float* __restrict__ cell,
                                                     float* __restrict__ hprev, float* __restrict__ hout, int T) {
  __shared__ float sh[H];
  const int b = blockIdx.x, u = threadIdx.x;
  float c = 0.f, h = 0.f;
  const float b0 = bhh ? bhh[u] : 0.f, b1 = bhh ? bhh[H + u] : 0.f, b2 = bhh ? bhh[2 * H + u] : 0.f,
              b3 = bhh ? bhh[3 * H + u] : 0.f;
  for (int t = 0; t < T; ++t) {
    const long long row = (long long)b * T + t;
    sh[u] = h;
    hprev[row * H + u] = h;
    __syncthreads();
    const float* xp = xproj + row * 4 * H;
    float a0 = xp[u] + b0, a1 = xp[H + u] + b1, a2 = xp[2 * H + u] + b2, a3 = xp[3 * H + u] + b3;
#pragma unroll 4
    for (int m = 0; m < H; ++m) {
      const float hm = sh[m];
      const float* w = whhT + (long long)m * 4 * H + u;  // consecutive u: coalesced
      a0 += w[0] * hm;
      a1 += w[H] * hm;
      a2 += w[2 * H] * hm;
      a3 += w[3 * H] * hm;
    }
    const float gi = sigmoidf_(a0), gf = sigmoidf_(a1), gg = tanhf(a2), go = sigmoidf_(a3);
    c = gf * c + gi * gg;
    h = go * tanhf(c);
    float* g = gates + row * 4 * H;
    g[u] = gi; g[H + u] = gf; g[2 * H + u] = gg; g[3 * H + u] = go;
    cell[row * H + u] = c;
    hout[row * H + u] = h;
    __syncthreads();  // sh is rewritten at the top of the next step
  }
}

// dhout [B][T][H] (gradient w.r.t. every h_t from above; NULL = zero) and dlast [B][H] (extra gradient of the
// last step; NULL = zero) -> dgates [B][T][4H] (w.r.t. the PRE-activation gates).  whh = W_hh [4H][H].
template <int H>
__global__ __launch_bounds__(H) void lstm_bwd_kernel(const float* __restrict__ dhout, const float* __restrict__ dlast,
                                                     const float* __restrict__ gates, const float* __restrict__ cell,
                                                     const float* __restrict__ whh, float* __restrict__ dgates, int T) {
  __shared__ float sg[4 * H];
  const int b = blockIdx.x, u = threadIdx.x;
  float dh_rec = 0.f, dc_next = 0.f;
  for (int t = T - 1; t >= 0; --t) {
    const long long row = (long long)b * T + t;
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
    __syncthreads();  // the previous step's readers of sg are done
    sg[u] = dai; sg[H + u] = daf; sg[2 * H + u] = dag; sg[3 * H + u] = dao;
    __syncthreads();
    float acc = 0.f;
#pragma unroll 4
    for (int j = 0; j < 4 * H; ++j) acc += sg[j] * whh[(long long)j * H + u];  // consecutive u: coalesced
    dh_rec = acc;
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
    hipLaunchKernelGGL(lstm_fwd_kernel<256>, dim3(batch), dim3(256), 0, s, xproj, whh_t, bhh, gates, cell, hprev, hout, T);
  else if (H == 64)
    hipLaunchKernelGGL(lstm_fwd_kernel<64>, dim3(batch), dim3(64), 0, s, xproj, whh_t, bhh, gates, cell, hprev, hout, T);
  else {
    qt_set_error("qt_lstm_forward: hidden size %d is not instantiated (256, 64)", H);
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
    hipLaunchKernelGGL(lstm_bwd_kernel<256>, dim3(batch), dim3(256), 0, s, dhout, dlast, gates, cell, whh, dgates, T);
  else if (H == 64)
    hipLaunchKernelGGL(lstm_bwd_kernel<64>, dim3(batch), dim3(64), 0, s, dhout, dlast, gates, cell, whh, dgates, T);
  else {
    qt_set_error("qt_lstm_backward: hidden size %d is not instantiated (256, 64)", H);
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
