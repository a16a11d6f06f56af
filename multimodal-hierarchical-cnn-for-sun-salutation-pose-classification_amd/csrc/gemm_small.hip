// Small dense products that are too thin for MFMA tiles (K=47 MLP input, N=12
// logits, their gradients).  f32 arithmetic; operands may be f32 or bf16 with
// arbitrary element strides, so the same kernel serves forward, data-gradient
// and weight-gradient of
//   numerical_mlp  (/root/reference/Quadtree_from scratch/models.py:255-260)
//   classifier.3   (/root/reference/Quadtree_from scratch/models.py:270).
#include "qt_common.h"

namespace {

struct SmallArgs {
  const void* A;
  const void* B;
  const float* bias;
  void* C;
  long long ars, aks, brs, bks, crs;
  int M, N, K;
  int a_bf16, b_bf16, c_bf16;
  int relu, accumulate;
};

__device__ __forceinline__ float ld(const void* p, long long i, int is_bf16) {
  return is_bf16 ? (float)static_cast<const bf16_t*>(p)[i] : static_cast<const float*>(p)[i];
}

__device__ __forceinline__ void finish(const SmallArgs& a, int m, int n, float acc) {
  if (a.bias) acc += a.bias[n];
  const long long ci = (long long)m * a.crs + n;
  if (a.accumulate) acc += ld(a.C, ci, a.c_bf16);
  if (a.relu) acc = fmaxf(acc, 0.f);
  if (a.c_bf16)
    static_cast<bf16_t*>(a.C)[ci] = (bf16_t)acc;
  else
    static_cast<float*>(a.C)[ci] = acc;
}

// one thread per output element (short K)
__global__ void gemm_small_thread_kernel(SmallArgs a) {
  const long long total = (long long)a.M * a.N;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % a.N), m = (int)(i / a.N);
    float acc4[4] = {0.f, 0.f, 0.f, 0.f};
    const long long ab = (long long)m * a.ars, bb = (long long)n * a.brs;
    int k = 0;
    for (; k + 3 < a.K; k += 4) {
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = ld(a.A, ab + (long long)(k + u) * a.aks, a.a_bf16);
        bv[u] = ld(a.B, bb + (long long)(k + u) * a.bks, a.b_bf16);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc4[u] += av[u] * bv[u];
    }
    for (; k < a.K; ++k) acc4[0] += ld(a.A, ab + (long long)k * a.aks, a.a_bf16) * ld(a.B, bb + (long long)k * a.bks, a.b_bf16);
    finish(a, m, n, (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]));
  }
}

// one wave per output element (long K), 64-lane shuffle reduction
__global__ __launch_bounds__(256) void gemm_small_wave_kernel(SmallArgs a) {
  const int lane = threadIdx.x & 63;
  const long long total = (long long)a.M * a.N;
  for (long long i = blockIdx.x * 4ll + (threadIdx.x >> 6); i < total; i += (long long)gridDim.x * 4) {
    const int n = (int)(i % a.N), m = (int)(i / a.N);
    // four independent partial sums: the loop is a chain of dependent scalar loads otherwise (classifier.3
    // forward, K = 2688: 42 trips of one memory latency each)
    float acc4[4] = {0.f, 0.f, 0.f, 0.f};
    const long long ab = (long long)m * a.ars, bb = (long long)n * a.brs;
    int k = lane;
    for (; k + 192 < a.K; k += 256) {
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = ld(a.A, ab + (long long)(k + 64 * u) * a.aks, a.a_bf16);
        bv[u] = ld(a.B, bb + (long long)(k + 64 * u) * a.bks, a.b_bf16);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc4[u] += av[u] * bv[u];
    }
    for (; k < a.K; k += 64) acc4[0] += ld(a.A, ab + (long long)k * a.aks, a.a_bf16) * ld(a.B, bb + (long long)k * a.bks, a.b_bf16);
    float acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) finish(a, m, n, acc);
  }
}

// 32 x 32 output tile per 256-thread workgroup, K in steps of 32 through LDS (f32 math, fixed summation order): the products
// that are neither short in K (thread kernel) nor thin in M or N (wave kernels: one wave per output element) -- the LSTM
// input products and weight gradients of the clip models (M = 256..752, N = 188..752, K = 188..256) took 102 us per launch
// with a wave per output element; 141 k waves of four loads each for 72 MFLOP.
__global__ __launch_bounds__(256) void gemm_small_tile_kernel(SmallArgs a) {
  __shared__ float As[32][33], Bs[32][33];
  const int tn = blockIdx.x % ((a.N + 31) / 32), tm = blockIdx.x / ((a.N + 31) / 32);
  const int m0 = tm * 32, n0 = tn * 32;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // outputs (m0 + ty*2 + i, n0 + tx*2 + j)
  const int lr = threadIdx.x >> 3, lc = (threadIdx.x & 7) * 4; // staging: row lr, k columns lc .. lc+3
  const bool a_kfast = a.aks <= a.ars, b_kfast = a.bks <= a.brs;
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  // staging follows the operand's contiguous axis (the weight-gradient products read their operands k-strided); the next
  // K step's eight values per thread are requested before the current step is multiplied (one global latency per step otherwise)
  float ra[4], rb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      {
        const int r = a_kfast ? lr : (int)(threadIdx.x & 31), c = a_kfast ? lc + u : (int)(threadIdx.x >> 5) + 8 * u;
        const int m = m0 + r, k = k0 + c;
        ra[u] = (m < a.M && k < a.K) ? ld(a.A, (long long)m * a.ars + (long long)k * a.aks, a.a_bf16) : 0.f;
      }
      {
        const int r = b_kfast ? lr : (int)(threadIdx.x & 31), c = b_kfast ? lc + u : (int)(threadIdx.x >> 5) + 8 * u;
        const int n = n0 + r, k = k0 + c;
        rb[u] = (n < a.N && k < a.K) ? ld(a.B, (long long)n * a.brs + (long long)k * a.bks, a.b_bf16) : 0.f;
      }
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < a.K; k0 += 32) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      As[a_kfast ? lr : (int)(threadIdx.x & 31)][a_kfast ? lc + u : (int)(threadIdx.x >> 5) + 8 * u] = ra[u];
      Bs[b_kfast ? lr : (int)(threadIdx.x & 31)][b_kfast ? lc + u : (int)(threadIdx.x >> 5) + 8 * u] = rb[u];
    }
    __syncthreads();
    if (k0 + 32 < a.K) fetch(k0 + 32);
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const float a0 = As[ty * 2][k], a1 = As[ty * 2 + 1][k], b0 = Bs[tx * 2][k], b1 = Bs[tx * 2 + 1][k];
      acc[0][0] += a0 * b0; acc[0][1] += a0 * b1;
      acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + ty * 2 + i, n = n0 + tx * 2 + j;
      if (m < a.M && n < a.N) finish(a, m, n, acc[i][j]);
    }
}

// one wave per output element, K contiguous in both operands and a multiple of 8: every lane requests all its 8-element
// pieces (16 B of bf16 / 32 B of f32) before the first multiply -- classifier.3's forward (K = 2688: 22 us as 42 trips of
// two scalar loads per lane, one memory latency per group of four) is then ONE latency
template <bool A16, bool B16>
__global__ __launch_bounds__(256) void gemm_small_wave_vec_kernel(SmallArgs a) {
  const int lane = threadIdx.x & 63;
  const long long total = (long long)a.M * a.N;
  const int nchunk = a.K >> 3;
  for (long long i = blockIdx.x * 4ll + (threadIdx.x >> 6); i < total; i += (long long)gridDim.x * 4) {
    const int n = (int)(i % a.N), m = (int)(i / a.N);
    const long long ab = (long long)m * a.ars, bb = (long long)n * a.brs;
    float acc = 0.f;
    for (int c0 = 0; c0 < nchunk; c0 += 64 * 4) {
      float av[4][8], bv[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * 64 + lane;
        const bool in = c < nchunk;
        const long long ka = ab + (long long)(in ? c : 0) * 8, kb = bb + (long long)(in ? c : 0) * 8;
        if constexpr (A16) QtVec8<bf16_t>::load(static_cast<const bf16_t*>(a.A) + ka, av[u]);
        else QtVec8<float>::load(static_cast<const float*>(a.A) + ka, av[u]);
        if constexpr (B16) QtVec8<bf16_t>::load(static_cast<const bf16_t*>(a.B) + kb, bv[u]);
        else QtVec8<float>::load(static_cast<const float*>(a.B) + kb, bv[u]);
        if (!in) {
#pragma unroll
          for (int e = 0; e < 8; ++e) av[u][e] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += av[u][e] * bv[u][e];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) finish(a, m, n, acc);
  }
}

}  // namespace

extern "C" int qt_gemm_small(const qt_gemm_small_desc* d, const void* A, const void* B, const float* bias, void* C,
                             void* stream) {
  QT_CHECK_ARG(d && A && B && C, "qt_gemm_small: null argument");
  QT_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "qt_gemm_small: bad shape %dx%dx%d", d->M, d->N, d->K);
  for (int t : {d->a_dtype, d->b_dtype, d->c_dtype})
    QT_CHECK_ARG(t == QT_F32 || t == QT_BF16, "qt_gemm_small: bad dtype %d", t);
  SmallArgs a;
  a.A = A; a.B = B; a.bias = bias; a.C = C;
  a.ars = d->a_row_stride; a.aks = d->a_k_stride; a.brs = d->b_row_stride; a.bks = d->b_k_stride; a.crs = d->c_row_stride;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.a_bf16 = d->a_dtype == QT_BF16; a.b_bf16 = d->b_dtype == QT_BF16; a.c_bf16 = d->c_dtype == QT_BF16;
  a.relu = d->relu; a.accumulate = d->accumulate;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long total = (long long)d->M * d->N;
  if (d->K > 96 && d->M >= 32 && d->N >= 32) {
    const int tiles = ((d->M + 31) / 32) * ((d->N + 31) / 32);
    hipLaunchKernelGGL(gemm_small_tile_kernel, dim3(tiles), dim3(256), 0, s, a);
  } else if (d->K <= 96) {
    long long g = (total + 255) / 256;
    hipLaunchKernelGGL(gemm_small_thread_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, s, a);
  } else {
    long long g = (total + 3) / 4;
    const dim3 grid((unsigned)(g > 16384 ? 16384 : g));
    const int ea = a.a_bf16 ? 2 : 4, eb = a.b_bf16 ? 2 : 4;
    const bool vec = a.aks == 1 && a.bks == 1 && d->K % 8 == 0 && a.ars % 8 == 0 && a.brs % 8 == 0 &&
                     ((uintptr_t)A % (8 * ea)) == 0 && ((uintptr_t)B % (8 * eb)) == 0 && !(a.a_bf16 == 0 && a.b_bf16 == 1);
    if (vec && a.a_bf16 && a.b_bf16) hipLaunchKernelGGL((gemm_small_wave_vec_kernel<true, true>), grid, dim3(256), 0, s, a);
    else if (vec && a.a_bf16) hipLaunchKernelGGL((gemm_small_wave_vec_kernel<true, false>), grid, dim3(256), 0, s, a);
    else if (vec) hipLaunchKernelGGL((gemm_small_wave_vec_kernel<false, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gemm_small_wave_kernel, grid, dim3(256), 0, s, a);
  }
  QT_CHECK_LAUNCH();
  return QT_OK;
}
