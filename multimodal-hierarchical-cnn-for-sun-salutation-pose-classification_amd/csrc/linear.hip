// nn.Linear on a small batch (M <= 256 rows) with a large weight matrix, bf16: split-K MFMA GEMM.
//
// Replaces F.linear of classifier.0 (/root/reference/Quadtree_from scratch/models.py:266-268,
// Linear(5376 -> 2688) + ReLU) in forward() (models.py:303) and its backward-input product under
// loss.backward() (Quadtree_train.py:65).
//
//   y[m][n] = act( bias[n] + sum_k x[m][k] * w[n][k] )        x [M][K], w [N][K]  (K contiguous)
//
// With M = 256 the implicit-GEMM conv kernel has 2 x N/128 = 42 tiles for 256 CUs and walks the
// whole K = 5376 in each of them (82 us forward, 124 us backward-input): the layer is bound by
// streaming its 29 MB of weights exactly once, which takes every CU.  Here a workgroup owns all M
// rows x 64 output columns x ONE SLICE of K, so N/64 x S workgroups (S chosen to reach ~256) each
// stream a disjoint 64 x K/S block of the weights; partial products go to an f32 workspace
// [S][M][N] with plain stores and a second kernel adds the S slices in order, applies bias / ReLU
// and converts (deterministic, no atomics).  Staging is LDS-DMA with the XOR chunk swizzle of
// conv_igemm.hip, three stages, two K-steps in flight.
#include <stdlib.h>

#include "qt_common.h"

namespace {

struct LinArgs {
  const bf16_t* x;
  const bf16_t* w;
  float* part;  // [S][M][N]
  int M, N, K;
  int ksteps;      // K / 64
  int steps_per;   // K-steps per slice
  int nsplit, ntn;
};

constexpr int LN_BM = 256, LN_BN = 64, LN_STAGE = (LN_BM + LN_BN) * 128, LN_NST = 3;

__global__ __launch_bounds__(256) void linear_splitk_kernel(LinArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x / p.ntn, tn = blockIdx.x - split * p.ntn;
  const int n0 = tn * LN_BN;
  const int ks0 = split * p.steps_per;
  const int nst = min(p.steps_per, p.ksteps - ks0);
  if (nst <= 0) return;

  // staging: thread = (row in a 32-row pass, 16-byte LDS slot); source chunk = slot ^ (row & 7)
  const int rbase = tid >> 3, chunk = (tid & 7) ^ (rbase & 7);
  const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(qt_zero_page);
  const bf16_t* a_ptr[8];
  int a_adv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = rbase + 32 * i;
    const bool ok = m < p.M;
    a_ptr[i] = ok ? p.x + (long long)m * p.K + ks0 * 64 + chunk * 8 : zero_src;
    a_adv[i] = ok ? 64 : 0;
  }
  const bf16_t* w_ptr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) w_ptr[i] = p.w + (long long)(n0 + rbase + 32 * i) * p.K + ks0 * 64 + chunk * 8;
  const unsigned smem_base = lds_addr_of(smem);
  auto dma = [&](int buf) {
    const unsigned sa = smem_base + buf * LN_STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      glds16(a_ptr[i], sa + i * 4096);
      a_ptr[i] += a_adv[i];
    }
    const unsigned sw = sa + LN_BM * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      glds16(w_ptr[i], sw + i * 4096);
      w_ptr[i] += 64;
    }
  };

  // wave w: rows [64w, 64w+64) x all 64 columns; weights are the MFMA A operand
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, lg = lane >> 4;

  dma(0);
  if (nst > 1) dma(1);
  for (int ks = 0; ks < nst; ++ks) {
    const int buf = ks % LN_NST;
    // stage ks has landed (10 DMA instructions per thread and stage; stage ks+1 may still fly) ...
    if (ks + 1 < nst)
      asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    // ... and every wave is done with stage ks-1, whose buffer the next fill reuses
    if (ks + 2 < nst) dma((ks + 2) % LN_NST);
    const unsigned char* sa = smem + buf * LN_STAGE;
    const unsigned char* sw = sa + LN_BM * 128;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      uint4 fx[4], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wave * 64 + i * 16 + li;
        fx[i] = *reinterpret_cast<const uint4*>(sa + r * 128 + (((kc * 4 + lg) ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = j * 16 + li;
        fw[j] = *reinterpret_cast<const uint4*>(sw + r * 128 + (((kc * 4 + lg) ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) QtMma<bf16_t>::run(acc[i][j], fw[j], fx[i]);
    }
  }

  // lane: row m = 64w + 16i + li, columns n0 + 16j + 4lg .. +3
  float* out = p.part + (long long)split * p.M * p.N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = wave * 64 + i * 16 + li;
    if (m < p.M) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(out + (long long)m * p.N + n0 + j * 16 + lg * 4) = acc[i][j];
    }
  }
}

// y[m][n] = act(bias[n] + sum_s part[s][m][n]), 8 columns per thread
__global__ __launch_bounds__(256) void linear_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                            bf16_t* __restrict__ y, long long MN, int N, int nsplit,
                                                            int relu) {
  const long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * 8;
  if (i >= MN) return;
  float v[8];
  const int n = (int)(i % N);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = bias ? bias[n + e] : 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float4 a = *reinterpret_cast<const float4*>(part + s * MN + i);
    const float4 b = *reinterpret_cast<const float4*>(part + s * MN + i + 4);
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
    v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  if (relu) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  QtVec8<bf16_t>::store(y + i, v);
}

int pick_split(int ntn, int ksteps) {
  int s = 256 / ntn;
  if (s < 1) s = 1;
  if (s > ksteps) s = ksteps;
  while (s > 1 && ksteps % s) --s;  // equal slices
  return s;
}

int g_linear_enabled = -1;
bool linear_enabled() {
  if (g_linear_enabled < 0) {
    const char* e = getenv("QTCNN_LINEAR_SPLITK");
    g_linear_enabled = e ? atoi(e) : 1;
  }
  return g_linear_enabled != 0;
}

}  // namespace

extern "C" size_t qt_linear_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || M > LN_BM || N <= 0 || N % LN_BN || K <= 0 || K % 64) return 0;
  return (size_t)pick_split(N / LN_BN, K / 64) * M * N * 4;
}

// bf16, M <= 256, N % 64 == 0, K % 64 == 0 and a workspace of qt_linear_workspace_bytes: split-K kernel.
// Returns QT_ERR_UNSUPPORTED for shapes it does not cover (callers fall back to qt_conv2d_igemm).
extern "C" int qt_linear_bf16(const void* x, const void* w, const float* bias, int relu, void* y, int M, int N, int K,
                              void* workspace, size_t workspace_bytes, void* stream) {
  QT_CHECK_ARG(x && w && y, "qt_linear_bf16: null argument");
  const size_t need = qt_linear_workspace_bytes(M, N, K);
  if (!linear_enabled() || need == 0 || !workspace || workspace_bytes < need) {
    qt_set_error("qt_linear_bf16: shape M=%d N=%d K=%d / workspace %zu B not covered", M, N, K, workspace_bytes);
    return QT_ERR_UNSUPPORTED;
  }
  QT_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                   ((uintptr_t)workspace % 16) == 0,
               "qt_linear_bf16: pointers must be 16-byte aligned");
  LinArgs a;
  a.x = static_cast<const bf16_t*>(x);
  a.w = static_cast<const bf16_t*>(w);
  a.part = static_cast<float*>(workspace);
  a.M = M; a.N = N; a.K = K;
  a.ksteps = K / 64;
  a.ntn = N / LN_BN;
  a.nsplit = pick_split(a.ntn, a.ksteps);
  a.steps_per = a.ksteps / a.nsplit;
  constexpr int LDS = LN_NST * LN_STAGE;
  static std::atomic<unsigned long long> lds_limit_set{0};  // per device
  if (int rc = qt_raise_lds_limit(reinterpret_cast<const void*>(linear_splitk_kernel), LDS, lds_limit_set)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(linear_splitk_kernel, dim3(a.ntn * a.nsplit), dim3(256), LDS, s, a);
  QT_CHECK_LAUNCH();
  const long long MN = (long long)M * N;
  hipLaunchKernelGGL(linear_finish_kernel, dim3(qt_cdiv(MN / 8, 256)), dim3(256), 0, s, a.part, bias,
                     static_cast<bf16_t*>(y), MN, N, a.nsplit, relu);
  QT_CHECK_LAUNCH();
  return QT_OK;
}
