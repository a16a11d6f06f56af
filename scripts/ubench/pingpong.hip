// Developer aid (GPU box): what one K-tile of the 8-wave ping-pong schedule (conv_pt.hip / conv_s2.hip) costs, piece by piece.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/pingpong.hip -o /tmp/pingpong && /tmp/pingpong
// Each variant runs ITERS K-tiles: two segments per wave (L then C), each ended by a raw s_barrier; waves 4-7 run one
// barrier behind waves 0-3.  Cycles per K-tile = s_memtime delta / ITERS (median over workgroups), 256 workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
constexpr int ITERS = 400;

__device__ __forceinline__ void blds(const i32x4& rsrc, unsigned voff, unsigned soff, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds) : "memory");
}

// NM MFMAs per C segment, NR ds_read_b128 per L segment, ND LDS-DMA instructions per L segment, PING: two groups offset
template <int NM, int NR, int ND, bool PING, int NB>
__global__ __launch_bounds__(512, 2) void kern(const uint4* src, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2;
  f32x4 acc[14];
  for (int i = 0; i < 14; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  uint4 fr[NR > 0 ? NR : 1];
  for (int i = 0; i < (NR > 0 ? NR : 1); ++i) fr[i] = make_uint4(tid, i, 3, 4);
  i32x4 rs;
  unsigned long long b = (unsigned long long)src;
  rs.x = (int)(unsigned)b; rs.y = (int)((b >> 32) & 0xffff); rs.z = 1 << 20; rs.w = 0x00020000;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const unsigned voff = (unsigned)(tid * 16);
  __syncthreads();
  if (PING && grp == 1) asm volatile("s_barrier" ::: "memory");
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    // ---- L ----
#pragma unroll
    for (int d = 0; d < ND; ++d) blds(rs, voff, (unsigned)(((it * ND + d) & 63) * 8192), lds0 + 98304 + ((it + d) & 3) * 8192 + wave * 1024);
#pragma unroll
    for (int r = 0; r < NR; ++r)
      fr[r] = *reinterpret_cast<const uint4*>(smem + ((it & 1) * 32768) + r * 2048 + ((lane & 15) * 128) + (((lane >> 4) ^ (lane & 7)) << 4));
    if (ND > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int k = 0; k < NB; ++k) asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- C ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < NM; ++m)
      acc[m % 14] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[m % (NR > 0 ? NR : 1)]),
                                                            __builtin_bit_cast(bf16x8, fr[(m + 1) % (NR > 0 ? NR : 1)]), acc[m % 14], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    for (int k = 0; k < NB; ++k) asm volatile("s_barrier" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (PING && grp == 0) asm volatile("s_barrier" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0;
  for (int i = 0; i < 14; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) sink[tid] = s;
  if (tid == 0) out[blockIdx.x] = t1 - t0;
}

// Register-pipelined variant: the fragments of K-tile t+1 are read WHILE the MFMAs of K-tile t issue (two fragment sets),
// DMA first.  PING: the two wave groups still alternate (L = DMA issue only); else all 8 waves in lockstep, ONE barrier
// per K-tile.
template <int NM, int NR, int ND, bool PING>
__global__ __launch_bounds__(512, 2) void kern_pipe(const uint4* src, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2;
  f32x4 acc[14];
  for (int i = 0; i < 14; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  uint4 fa[NR], fb[NR];
  for (int i = 0; i < NR; ++i) { fa[i] = make_uint4(tid, i, 3, 4); fb[i] = make_uint4(tid, i, 5, 6); }
  i32x4 rs;
  unsigned long long b = (unsigned long long)src;
  rs.x = (int)(unsigned)b; rs.y = (int)((b >> 32) & 0xffff); rs.z = 1 << 20; rs.w = 0x00020000;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const unsigned voff = (unsigned)(tid * 16);
  const unsigned char* rbase = smem + ((lane & 15) * 128) + (((lane >> 4) ^ (lane & 7)) << 4);
  __syncthreads();
  if (PING && grp == 1) asm volatile("s_barrier" ::: "memory");
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  auto ktile = [&](int it, uint4 (&cur)[NR], uint4 (&nxt)[NR]) {
#pragma unroll
    for (int d = 0; d < ND; ++d) blds(rs, voff, (unsigned)(((it * ND + d) & 63) * 8192), lds0 + 98304 + ((it + d) & 3) * 8192 + wave * 1024);
    if (PING) {
      if (ND > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
      asm volatile("s_barrier" ::: "memory");
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      acc[m % 14] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur[m % NR]),
                                                            __builtin_bit_cast(bf16x8, cur[(m + 1) % NR]), acc[m % 14], 0, 0, 0);
      if (m < NR) nxt[m] = *reinterpret_cast<const uint4*>(rbase + (((it + 1) & 1) * 32768) + m * 2048);
    }
    __builtin_amdgcn_s_setprio(0);
    if (!PING && ND > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  for (int it = 0; it < ITERS; it += 2) {
    ktile(it, fa, fb);
    ktile(it + 1, fb, fa);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (PING && grp == 0) asm volatile("s_barrier" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0;
  for (int i = 0; i < 14; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) sink[tid] = s;
  if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <typename K>
void run_k(const char* what, K k, const uint4* src, unsigned long long* out, float* sink) {
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  std::vector<unsigned long long> h(256);
  double best = 1e30;
  float ms = 0, bestms = 1e30;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 147456, 0, src, out, sink);
    hipEventRecord(b);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    best = std::min(best, (double)h[128] / ITERS);
    bestms = std::min(bestms, ms);
  }
  printf("%-72s : %7.0f cycles / K-tile   (%.1f us / %d K-tiles)\n", what, best, bestms * 1e3, ITERS);
}

template <int NM, int NR, int ND, bool PING, int NB = 1>
void run(const char* what, const uint4* src, unsigned long long* out, float* sink) {
  auto k = kern<NM, NR, ND, PING, NB>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  std::vector<unsigned long long> h(256);
  double best = 1e30;
  float ms = 0;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 147456, 0, src, out, sink);
    hipEventRecord(b);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    best = std::min(best, (double)h[128] / ITERS);
  }
  printf("%-64s NM %2d NR %2d ND %d ping %d bars %d : %7.0f cycles / K-tile   (%.1f us / %d K-tiles)\n", what, NM, NR, ND, (int)PING, NB,
         best, ms * 1e3, ITERS);
}

int main() {
  uint4* src; unsigned long long* out; float* sink;
  hipMalloc(&src, 1 << 21); hipMalloc(&out, 256 * 8); hipMalloc(&sink, 4096);
  std::vector<unsigned> h((1 << 21) / 4);
  for (auto& v : h) v = (unsigned)rand() * 2654435761u;
  hipMemcpy(src, h.data(), 1 << 21, hipMemcpyHostToDevice);
  run<0, 0, 0, true>("bare: 2 barriers per wave and K-tile, groups offset", src, out, sink);
  run<0, 0, 0, false>("bare, all 8 waves in lockstep", src, out, sink);
  run<28, 0, 0, true>("+ 28 MFMA per wave in C", src, out, sink);
  run<56, 0, 0, true>("+ 56 MFMA", src, out, sink);
  run<28, 18, 0, true>("+ 28 MFMA, 18 ds_read_b128 in L", src, out, sink);
  run<28, 18, 2, true>("+ 28 MFMA, 18 reads, 2 LDS-DMA", src, out, sink);
  run<28, 18, 4, true>("+ 28 MFMA, 18 reads, 4 LDS-DMA (conv_s2 today)", src, out, sink);
  run<32, 20, 4, true>("+ 32 MFMA, 20 reads, 4 LDS-DMA (7x7 stack)", src, out, sink);
  run<56, 22, 4, true>("+ 56 MFMA, 22 reads, 4 LDS-DMA (BN = 256)", src, out, sink);
  run<56, 36, 8, true>("two K-tiles per segment pair: 56 MFMA, 36 reads, 8 DMA", src, out, sink);
  run<28, 18, 4, false>("lockstep (no ping-pong): 28 MFMA, 18 reads, 4 DMA", src, out, sink);
  run_k("PIPELINED lockstep, 1 barrier: 28 MFMA, 18 reads under them, 2 DMA", kern_pipe<28, 18, 2, false>, src, out, sink);
  run_k("PIPELINED lockstep, 1 barrier: 28 MFMA, 18 reads under them, 3 DMA", kern_pipe<28, 18, 3, false>, src, out, sink);
  run_k("PIPELINED lockstep, 1 barrier: 28 MFMA, 18 reads under them, 4 DMA", kern_pipe<28, 18, 4, false>, src, out, sink);
  run_k("PIPELINED lockstep, 1 barrier: 32 MFMA, 20 reads under them, 4 DMA", kern_pipe<32, 20, 4, false>, src, out, sink);
  run_k("PIPELINED ping-pong (L = DMA only): 28 MFMA, 18 reads, 3 DMA", kern_pipe<28, 18, 3, true>, src, out, sink);
  run_k("PIPELINED ping-pong (L = DMA only): 28 MFMA, 18 reads, 4 DMA", kern_pipe<28, 18, 4, true>, src, out, sink);
  run_k("PIPELINED lockstep: 28 MFMA, 18 reads, 0 DMA", kern_pipe<28, 18, 0, false>, src, out, sink);
  run<48, 36, 1, true>("layer1 re-cut: 3 taps per segment: 48 MFMA, 36 reads, 1 DMA", src, out, sink);
  run<48, 36, 2, true>("layer1 re-cut: 3 taps per segment: 48 MFMA, 36 reads, 2 DMA", src, out, sink);
  run<32, 24, 1, true>("layer1 re-cut: 2 taps per segment: 32 MFMA, 24 reads, 1 DMA", src, out, sink);
  run<48, 36, 0, false>("lockstep 48 MFMA, 36 reads (today's ring kernel, 3 taps at a time)", src, out, sink);
  run<0, 18, 0, true>("reads only", src, out, sink);
  run<0, 0, 4, true>("DMA only", src, out, sink);
  return 0;
}
