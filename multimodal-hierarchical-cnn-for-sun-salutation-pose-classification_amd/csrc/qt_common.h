// Shared device/host definitions for the QuadtreeCNN HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qtcnn.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define QT_WAVE 64
#define QT_LDS_AS __attribute__((address_space(3)))

// thread-local last-error text (qt_last_error)
void qt_set_error(const char* fmt, ...);

#define QT_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      qt_set_error(__VA_ARGS__);           \
      return QT_ERR_INVALID_ARG;           \
    }                                      \
  } while (0)

#define QT_CHECK_LAUNCH()                                              \
  do {                                                                 \
    hipError_t e_ = hipGetLastError();                                 \
    if (e_ != hipSuccess) {                                            \
      qt_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,     \
                   hipGetErrorString(e_));                             \
      return QT_ERR_LAUNCH;                                            \
    }                                                                  \
  } while (0)

template <typename T> struct QtElem;
template <> struct QtElem<float> { static constexpr int kDtype = QT_F32; };
template <> struct QtElem<bf16_t> { static constexpr int kDtype = QT_BF16; };

// ---- 16-byte vector <-> 4 or 8 elements -----------------------------------
__device__ __forceinline__ float qt_bf16_bits_to_f32(uint32_t bits16) {
  return __uint_as_float(bits16 << 16);
}

// load N consecutive T as float (N*sizeof(T) must be 16 or 32 bytes aligned accordingly)
template <typename T> struct QtVec8;  // 8 consecutive elements <-> float[8]
template <> struct QtVec8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
template <> struct QtVec8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = o;
  }
};

// ReLU mask, one bit per element ([M][N/8] bytes): zero the channels of the 8-channel group at element offset `off`
// (a multiple of 8) whose bit is clear.  `byte` was loaded from bits[off >> 3].
__device__ __forceinline__ void qt_apply_mask_bits(unsigned byte, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (byte >> e) & 1u ? v[e] : 0.f;
}

// Sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), every lane receives the total; four DPP adds in a fixed
// order (row mirror, half-row mirror, quad reversal, neighbour swap) -> deterministic, no LDS crossbar traffic
// (__shfl_xor compiles to ds_bpermute_b32).
__device__ __forceinline__ float qt_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x1b, 0xf, 0xf, false));   // quad_perm [3,2,1,0]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  return v;
}

template <typename T> __device__ __forceinline__ float qt_to_f32(T v);
template <> __device__ __forceinline__ float qt_to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float qt_to_f32<bf16_t>(bf16_t v) { return (float)v; }

// ---- MFMA: one 16-byte K-chunk per lane for both operands ---------------------
// D[row][col] += sum_k A[row][k] * B[col][k]; lane l supplies row/col (l&15) and
// the K-chunk (l>>4).  bf16: 8 k per chunk -> one 16x16x32 MFMA.  f32: 4 k per
// chunk -> four 16x16x4 MFMAs (k-order inside the step is a permutation shared
// by A and B, which a dot product does not see).
template <typename T> struct QtMma;
template <> struct QtMma<bf16_t> {
  static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
        __builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};
template <> struct QtMma<float> {
  static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};

// XCD-aware bijective block remap (8 XCDs, blocks dealt round-robin): blocks
// that end up with consecutive logical ids share an XCD (and its L2).
__device__ __forceinline__ int qt_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// 128 bytes of zeros: the LDS-DMA source of halo pixels and of rows past the tensor's end
static __device__ uint4 qt_zero_page[8];

typedef __attribute__((address_space(3))) void* lptr_t;

// One LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to the 1 KiB
// at LDS byte address `lds_addr` (wave-uniform).  Issued from inline asm on purpose: the
// compiler then neither counts it in vmcnt nor fences later LDS reads with vmcnt(0), so
// the ring below can keep stages in flight across barriers with counted waits (the
// kernel's own s_waitcnt vmcnt(N) + s_barrier order the data before it is read).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}
// LDS-DMA through a buffer resource: 64 lanes x 16 B from  base + voff (per lane, bytes) + soff (wave-uniform)  to the
// 1 KiB at LDS byte address `lds_addr`.  Lanes whose voff + soff is outside [0, num_records) write ZEROS (hardware range
// check of raw buffers; measured in round 3: the scalar offset DOES take part -- a tile-sized num_records with the tile's
// origin in soff zeroed every tile but the first -- so num_records spans the whole tensor): halo rows and rows past the tensor need no second pointer, and a
// transfer costs no vector instruction -- the per-lane offsets are loop invariants, tap / chunk offsets are scalar.
// (Issued from inline asm, like glds16: the compiler neither counts it in vmcnt nor fences LDS reads with vmcnt(0).)
typedef __attribute__((ext_vector_type(4))) int i32x4;
constexpr unsigned kOob = 0x80000000u;   // a per-lane offset no tensor here reaches (< 2 GiB, checked by the launcher)
__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r.y = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));   // stride 0: raw buffer
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;                                                  // 32-bit data format, no swizzle (gfx94x / gfx950)
  return r;
}
__device__ __forceinline__ void blds16(const i32x4& rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr)
      : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)p);
}


// exact unsigned division by a runtime constant (n < 2^31): q = (umulhi(n, mul) + n) >> shr
struct FastDiv {
  unsigned mul, shr;
};
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  f.shr = s;
  f.mul = (unsigned)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, FastDiv f) {
  return (__umulhi(n, f.mul) + n) >> f.shr;  // exact for n < 2^31
}


// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: remember it per (kernel, device) --
// one bit per device ordinal in an atomic owned by the call site -- so a process that runs plans on cuda:1
// after cuda:0 raises the limit there too, and two host threads may race here harmlessly.
#include <atomic>
static inline int qt_raise_lds_limit(const void* kern, int lds_bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return QT_OK;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    qt_set_error("hipFuncSetAttribute(%d B of LDS, device %d): %s", lds_bytes, dev, hipGetErrorString(e));
    return QT_ERR_LAUNCH;
  }
  done.fetch_or(bit, std::memory_order_release);
  return QT_OK;
}

static inline int qt_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
// qt_conv_desc::quad: 0 = whole images, 1 or 2 = 2 x 2 regions per image, 4 = 4 x 4 regions
static inline int qt_quad_split(int quad) { return quad == 1 ? 2 : quad; }
static inline int qt_quad_regions(int quad) { return quad ? qt_quad_split(quad) * qt_quad_split(quad) : 1; }
