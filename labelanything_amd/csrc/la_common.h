// Common device/host helpers for the LabelAnything gfx950 kernels.
// CDNA4 only: wave = 64 lanes, MFMA 32x32x16 (f16/bf16 in, f32 accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Measurement / debugging switches (environment overrides of kernel selection, ablation bits of la_gemm_variant) exist only in a
// -DLA_DEBUG build (`make DEBUG=1` -> libla_hip_dbg.so, loaded by tools/ through LA_HIP_LIB): the product library never reads the
// environment and its kernels carry no "results wrong by construction" paths.
#ifdef LA_DEBUG
#include <cstdlib>
static inline const char* la_dbg_env(const char* name) { return getenv(name); }
#else
static inline constexpr const char* la_dbg_env(const char*) { return nullptr; }
#endif

namespace la {

typedef _Float16 f16_t;
typedef __bf16 bf16_t;

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- 16-bit storage types -----------------------------------------------------------------
template <typename T> struct Half16;
template <> struct Half16<f16_t> {
  static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
  }
};
template <> struct Half16<bf16_t> {
  static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8_t, a), __builtin_bit_cast(b8_t, b), c, 0, 0, 0);
  }
};

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

template <typename T> __device__ __forceinline__ uint16_t bits16(T v) { return __builtin_bit_cast(uint16_t, v); }
template <typename T> __device__ __forceinline__ T from_bits16(uint16_t v) { return __builtin_bit_cast(T, v); }

// two floats -> one dword of 2 x 16 bit, round-to-nearest-even: a single v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32 on gfx950
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef T t2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, t2));
}
template <typename T> __device__ __forceinline__ float unpack_lo(uint32_t v) { return (float)from_bits16<T>((uint16_t)(v & 0xffffu)); }
template <typename T> __device__ __forceinline__ float unpack_hi(uint32_t v) { return (float)from_bits16<T>((uint16_t)(v >> 16)); }

// store 2 / 4 consecutive values in the storage type (16-bit packed, or plain fp32)
template <typename T> __device__ __forceinline__ void store2(T* p, float a, float b) { *reinterpret_cast<uint32_t*>(p) = pack2<T>(a, b); }
template <> __device__ __forceinline__ void store2<float>(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
template <typename T> __device__ __forceinline__ void store4v(T* p, float a, float b, float c, float d) {
  uint2 v;
  v.x = pack2<T>(a, b);
  v.y = pack2<T>(c, d);
  *reinterpret_cast<uint2*>(p) = v;
}
template <> __device__ __forceinline__ void store4v<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

// ---- LA_F16X2 operand rows: two 16-bit planes [hi (E) | lo (E)] per row, hi = rn(v), lo = rn(v - hi) --------------------------------
// A GEMM that multiplies [A_hi | A_lo | A_hi] by [W_hi | W_hi | W_lo] (la_gemm a_kmod = 2 K over 3 K columns) then carries ~21
// mantissa bits through the fast MFMA: fp32-level accuracy at 3 of its 16 passes per fp32-MFMA pass.
template <typename T> __device__ __forceinline__ void store_split(T* row2e, int E, int col, float v) {
  const T h = (T)v;
  row2e[col] = h;
  row2e[E + col] = (T)(v - (float)h);
}
template <typename T> __device__ __forceinline__ void store4_split(T* row2e, int E, int col, float a, float b, float c, float d) {
  const T ha = (T)a, hb = (T)b, hc = (T)c, hd = (T)d;
  store4v<T>(row2e + col, a, b, c, d);
  store4v<T>(row2e + E + col, a - (float)ha, b - (float)hb, c - (float)hc, d - (float)hd);
}

// ---- activations ----------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// GELU(erf) for epilogues whose result is rounded to 16 bit anyway: erf by Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7,
// three orders below the fp16 rounding step) = 1 v_rcp + 1 v_exp + 8 FMA instead of libm's ~40-instruction erff.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.0f - p * t * __expf(-z * z);          // erf(|x|/sqrt2)
  return 0.5f * x * (1.0f + copysignf(e, x));
}

// Two GELUs at once on the packed-fp32 pipe (v_pk_mul_f32 / v_pk_fma_f32), for epilogues that are VALU-bound and round the result to
// 16 bit.  GELU(x) = x Phi(x) with Phi(c) = 0.5 + c R(c^2) on |c| <= 3 sqrt 2, R a degree-9 minimax polynomial in c^2 (Lawson iteration
// on (Phi(c) - 0.5) / c with weight c, checked in fp32 Horner arithmetic; beyond the clamp Phi stays at Phi(+-3 sqrt 2) = 1 - 1.1e-5 /
// 1.1e-5).  |GELU error|: 1.6e-6 on |x| <= 1, 3.1e-6 on |x| <= 2, 4.9e-6 on |x| <= 3, 1.4e-5 on |x| <= 4, 4.6e-5 beyond - in 12 packed + 2
// clamp instructions per pair.  The form of rounds 3 - 5 (erf(u) = u P(u^2), degree 9, then 0.5 x (1 + erf): 14 + 2 instructions) had
// 8e-6 / 2.2e-5 / 3.4e-5 / 4.9e-5 / 5.9e-5 on the same ranges; a degree-8 R (11 + 2) has 8e-6 / 1.4e-5 / 2.4e-5 / 4.7e-5 / 5.6e-5 and
// measured 8.5e-4 on the cfg1 logits where this one measures (profiles/r06_notes.md 6).  The GELU is a quarter of the lin1 launch.
constexpr float GELU_CLAMP = 4.2426405f;
constexpr float GELU_R0 = 3.9893408094e-01f, GELU_R1 = -6.6454246054e-02f, GELU_R2 = 9.9263965487e-03f, GELU_R3 = -1.1587673958e-03f,
                GELU_R4 = 1.0572972350e-04f, GELU_R5 = -7.3937967560e-06f, GELU_R6 = 3.7849368483e-07f, GELU_R7 = -1.3177804174e-08f,
                GELU_R8 = 2.7561278625e-10f, GELU_R9 = -2.5916132844e-12f;
__device__ __forceinline__ f32x2 gelu_erf_pk(f32x2 x) {
  f32x2 c;
  c.x = __builtin_amdgcn_fmed3f(x.x, -GELU_CLAMP, GELU_CLAMP);
  c.y = __builtin_amdgcn_fmed3f(x.y, -GELU_CLAMP, GELU_CLAMP);
  const f32x2 t = c * c;
  f32x2 p = t * GELU_R9 + GELU_R8;
  p = p * t + GELU_R7;
  p = p * t + GELU_R6;
  p = p * t + GELU_R5;
  p = p * t + GELU_R4;
  p = p * t + GELU_R3;
  p = p * t + GELU_R2;
  p = p * t + GELU_R1;
  p = p * t + GELU_R0;
  p = p * c + 0.5f;
  return x * p;
}

// ---- wave reductions (64 lanes) ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v, int width = 64) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v, int width = 64) {
  for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- DPP reductions -------------------------------------------------------------------------------------------------------------
// __shfl_xor lowers to ds_bpermute_b32 (an LDS-crossbar round trip, ~200 cycles when the next step depends on it).  A wave that
// runs alone on its SIMD with a serial chain of them (the fused two-way kernels: 1 workgroup per CU) is latency-bound on exactly
// that; the DPP forms below are plain VALU moves.  row16_sum: every lane ends up with the sum of its 16-lane row (xor 1, xor 2 as
// quad permutes, then rotations by 4 and 8 inside the row).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
  v += dpp_mov<0x124>(v);      // row_ror:4
  v += dpp_mov<0x128>(v);      // row_ror:8
  return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = row16_sum(v);
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}

// ---- LDS tile swizzle -----------------------------------------------------------------------
// Tiles are [rows][64 halfs] = 128 B per row = 8 chunks of 16 B.  ds_read_b128 is serviced in 16-lane
// groups over a 256-B bank row (MI355X_MICROARCH LDS table); XOR-ing the chunk with (row>>1)&7 makes
// the 16 rows of every group land on 16 distinct 16-B slots.
__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---- LDS-DMA ---------------------------------------------------------------------------------
// 16 bytes per lane, global -> LDS without touching VGPRs: global_load_lds_dwordx4, destination = M0 + lane*16.
// Issued through inline asm ON PURPOSE: hipcc treats the builtin as a pending LDS write and inserts s_waitcnt vmcnt(0)
// in front of the next ds_read, which serialises every prefetch pipeline (guide: "glds pipelining across barriers").
// With the asm form the compiler does not know about the transfer, so the caller MUST order it by hand:
// dma_wait<N>() (counted s_waitcnt vmcnt) followed by a barrier before any wave reads the bytes.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst_uniform);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}
// Same, with the source as a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: one address VGPR instead of
// two and no 64-bit VALU add per piece.
__device__ __forceinline__ void dma16s(const void* gbase_uniform, unsigned voff, unsigned lds_dst_uniform) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst_uniform);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(gbase_uniform), "s"(dst)
               : "memory");
}
template <int N> __device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device: remember it per (call site, device), so that a process
// that drives several GPUs raises the limit on each of them (mask: one static 64-bit word per call site).
static inline void ensure_dyn_lds(const void* kernel, int bytes, unsigned long long& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && ((mask >> dev) & 1ull)) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (dev >= 0 && dev < 64) mask |= 1ull << dev;
}

// XCD-aware tile order: consecutive workgroups round-robin over the 8 XCDs, so give each XCD a
// contiguous chunk of the tile sequence (bijective also when n % 8 != 0).
__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Linear tile index -> (tile_m, tile_n) in groups of GM row-panels with M fastest inside a group, so that the ~64 tiles
// one XCD has in flight cover a GM x (64/GM) patch: both the A row-panels and the W column-panels of the patch stay
// resident in the 4 MiB L2 instead of W being re-fetched from the fabric for every row-panel.
__device__ __forceinline__ void tile_coords(int t, int ntm, int ntn, int gm, int& tm, int& tn) {
  if (gm <= 1) {
    tm = t / ntn;
    tn = t % ntn;
    return;
  }
  const int per_group = gm * ntn;
  const int g = t / per_group;
  const int first = g * gm;
  const int rows = min(gm, ntm - first);
  const int r = t - g * per_group;
  tm = first + r % rows;
  tn = r / rows;
}

}  // namespace la

// ---- host side error plumbing ---------------------------------------------------------------
void la_set_error(const char* fmt, ...);
#define LA_CHECK_ARG(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      la_set_error(__VA_ARGS__);         \
      return -1;                         \
    }                                    \
  } while (0)
#define LA_CHECK_LAUNCH(name)                                          \
  do {                                                                 \
    hipError_t _e = hipGetLastError();                                 \
    if (_e != hipSuccess) {                                            \
      la_set_error("%s: %s", name, hipGetErrorString(_e));             \
      return -2;                                                       \
    }                                                                  \
  } while (0)
