// "t256w": the persistent 256 x 256 x 64 tile on FOUR waves of 512 registers (one wave per SIMD, 2 x 2, 128 x 128 per wave).
//
// Why a second persistent kernel beside gemm_t256q (8 waves x 256 registers, 128 x 64 per wave): with 128 x 128 per wave a k-tile is
// 64 MFMA 32x32x16 against 32 fragment reads (t256q: 32 against 28), there is no second wave group and therefore no phase barrier
// (2 barriers per k-tile instead of 8), and the 16 LDS-DMA pieces a wave issues per k-tile sit one at a time between its own MFMAs
// instead of arriving at the vector-memory front end eight at a time at the head of an interval.  Same LDS image, same piece shape
// (8 rows x 128 B), same XOR swizzle, same MFMA order over k as t256q: results are bit-identical.
//
// A wave's k-tile (rows [wr*128, +128) of A against rows [wc*128, +128) of W, 64 deep) runs as two halves of four 8-MFMA blocks:
//     H1  ks = 0..3:  acc[i = 0, 1][j = 0..3] += A_i(ks) . W_j(ks)      reads W_j(ks) (kept: 16 fragments) and A_0, A_1
//     H2  ks = 0..3:  acc[i = 2, 3][j = 0..3] += A_i(ks) . W_j(ks)      reads A_2, A_3 only
// so the W rows and the A rows of sub-tile 0 (48 of a k-tile's 64 KiB) are dead after H1 and the rest after H2: with TWO k-tile
// buffers the stream still runs a full k-tile ahead of its first reader -
//     blocks 0-2 of k-tile t:  A1(t+1)   4 pieces per wave into buffer (t+1)&1   (free since barrier B2 of t-1)
//     blocks 3-7 of k-tile t:  A0W(t+2) 12 pieces per wave into buffer t&1        (free since barrier B1 of t)
//     B1 (between blocks 2 and 3): every wave has finished its H1 reads of t and waited for its own A1(t) pieces
//     B2 (between blocks 6 and 7): every wave has finished reading t and waited for its own A0W(t+1) pieces;
//                                  block 7 reads the first fragments of t+1
// Counted waits (in-order retirement): before B1 vmcnt(16) = A0W(t+1) + A1(t+1) may stay out; before B2 vmcnt(15) = A1(t+1) + the 11
// pieces of A0W(t+2) issued so far.  The stream never stops at an output-tile seam (the next tile's first two k-tiles are in flight
// during the epilogue) and never branches at the end of the tile sequence: the last two k-tiles re-request valid bytes into dead
// regions.  Epilogue: epilogue_wave (gemm_shared.h) on the two 128 x 64 halves of the wave's block.
// LDS: [A buf0 | A buf1 | W buf0 | W buf1] x 32 KiB (dynamic LDS of a kernel without static LDS starts at 0: buffers toggle by XOR 32 KiB),
// then a 2 KiB slab and a 512 B row table per wave.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "gemm_shared.h"

namespace la {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));      // an operand fragment as a plain register tuple ("v" constraint)

__device__ __forceinline__ u32x4 lds_read16(unsigned addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((uintptr_t)addr);
}

// one LDS-DMA piece: M0 = ldsbase + IMM, 16 bytes per lane from gbase + voff
template <int IMM>
__device__ __forceinline__ void dma_piece(const void* gbase_uniform, unsigned voff, unsigned ldsbase_uniform) {
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(gbase_uniform), "s"(ldsbase_uniform), "n"(IMM)
               : "memory", "scc");
}

// MFMA with the accumulator pinned to the AGPR half of the register file ("+a"): with the builtin hipcc keeps the 256 loop-carried
// accumulator registers in VGPRs and copies 16 of them in and out of AGPRs around every MFMA (v_accvgpr_write / _read, spills).
// The asm form is invisible to the hazard recognizer: operands come from ds_read (s_waitcnt, no VALU producer), the accumulate chain
// needs no wait states, and the first VALU reader of an accumulator (epilogue) sits behind an explicit s_nop.
template <typename T> struct MfmaA;
template <> struct MfmaA<f16_t> {
  static __device__ __forceinline__ void go(f32x16& c, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  }
};
template <> struct MfmaA<bf16_t> {
  static __device__ __forceinline__ void go(f32x16& c, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  }
};

// accumulator := 0, defined in the AGPR class as well (0 . 0 + 0 on the matrix pipe): a v_mov / v_accvgpr_write form makes the loop-carried
// accumulators VGPR-class values again
__device__ __forceinline__ void acc_zero(f32x16& c) {
  const u32x4 z = {0u, 0u, 0u, 0u};
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %1, 0" : "=a"(c) : "v"(z));
}

#ifdef LA_DEBUG
constexpr int LA_W4_NSTAMP = 128;
__device__ unsigned long long g_w4_stamps[4 * LA_W4_NSTAMP];      // [wave][i] = s_memtime << 8 | tag (workgroup 0 of the last stamped launch)
#endif

template <int N> __device__ __forceinline__ void wait_vm_lgkm0() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// ---- the k-tile body is written out slot by slot: one MFMA, at most one memory instruction, pinned by sched_barrier(0) ----------------
#define LA_W4_SB __builtin_amdgcn_sched_barrier(0);
// ABL (measurement builds only, results wrong by construction): 1 no LDS-DMA pieces in the loop, 2 no fragment reads, 4 no waits / barriers,
// 8 no MFMAs
#define LA_W4_MF(IB, ii, j, PAR, KS)                                                                                            \
  if constexpr (!(ABL & 8)) MfmaA<T>::go(acc[(j) >> 1][(IB) + (ii)][(j) & 1], af[PAR][ii], wf[KS][j]); \
  LA_W4_SB
#define LA_W4_RW(KS, j) if constexpr (!(ABL & 2)) wf[KS][j] = lds_read16(waddr[KS] + (j) * 4096); LA_W4_SB
#define LA_W4_RA(PAR, ii, i, KS) if constexpr (!(ABL & 2)) af[PAR][ii] = lds_read16(aaddr[KS] + (i) * 4096); LA_W4_SB
#define LA_W4_PA0(src, i, bo) if constexpr (!(ABL & 1)) dma_piece<(i) * 1024>(src, soA0[i], dstA + (bo)); LA_W4_SB
#define LA_W4_PA1(src, i, bo) if constexpr (!(ABL & 1)) dma_piece<8192 + (i) * 1024>(src, soA1[i], dstA + (bo)); LA_W4_SB
#define LA_W4_PW(src, i, bo) if constexpr (!(ABL & 1)) dma_piece<(i) * 1024>(src, soW[i], dstW + (bo)); LA_W4_SB
#define LA_W4_QA0(src, i, bo) dma_piece<(i) * 1024>(src, soA0[i], dstA + (bo));      // (prologue: never ablated)
#define LA_W4_QA1(src, i, bo) dma_piece<8192 + (i) * 1024>(src, soA1[i], dstA + (bo));
#define LA_W4_QW(src, i, bo) dma_piece<(i) * 1024>(src, soW[i], dstW + (bo));
#define LA_W4_WAIT(N)                                                             \
  if constexpr (!(ABL & 4)) {                                                     \
    if (seam) wait_vm_lgkm0<((N) + SEAM > 63 ? 63 : (N) + SEAM)>();               \
    else wait_vm_lgkm0<(N)>();                                                    \
  }
#define LA_W4_BAR if constexpr (!(ABL & 4)) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
// how many memory instructions share one MFMA -> MFMA gap (tools/gen/w4_ktile.py; measured in profiles/r05_notes.md)
#ifndef LA_W4_GROUP
#define LA_W4_GROUP 1
#endif
#include "gemm_w4_ktile.inc"

// ---- epilogue of an interior tile without row / column maps: the wave's 128 x 128 block through an 8 KiB fp32 slab ---------------------
// One wave per SIMD issues one instruction every ~4-5 cycles and nothing else fills its slots: the epilogue costs what its instruction
// COUNT costs.  epilogue_wave (a 2 KiB 16-bit slab: bias, convert, 32-bit LDS stores, unzip) is ~1500 instructions per wave here;
// this form is ~450: the accumulators leave the AGPRs as 16-byte LDS stores of 4 rows x 1 column (no AGPR -> VGPR copy), come back as
// 4 rows x 8 columns per lane, and bias / activation / conversion run on row segments that go out as whole 128-byte lines.
//   round (i, jp) = rows [32 i, +32) x columns [64 jp, +64): 2 accumulator tiles = 8 KiB
//   slab unit (16 B) of row quad q (0..7), column c (0..63) at q * 64 + (c & ~7) + ((c & 7) ^ ((c >> 3) ^ q) & 7)
//   write: lane (fr, fh), register quad g of tile jj -> q = 2 g + fh, c = 32 jj + fr          (8 stores per round, conflict-free)
//   read:  lane -> q = lane >> 3, column group cg = lane & 7: columns 8 cg + t, t = 0..7      (8 loads per round, conflict-free)
// The LDS queue of a wave is in order: round r + 1 is written right behind the read instructions of round r.
template <typename T, int EPI>
__device__ __forceinline__ void epilogue_w4(char* slab, f32x16 (&acc)[2][4][2], int row0, int col0, const LaGemmEpilogue& e, int lane, int M, int N) {
  const int fr = lane & 31, fh = lane >> 5;
  const int rq = lane >> 3, cg = lane & 7;
  const unsigned sl = lds_addr_of(slab);
  // write address of (g = 0, jj = 0): + g * 2048 (two quads) + jj * 512 (four column groups)
  const unsigned waddr = sl + (unsigned)((fh * 64 + (fr & ~7) + ((fr & 7) ^ (((fr >> 3) ^ fh) & 7))) << 4);
  // (the swizzle term of tile jj, quad 2 g + fh: ((4 jj + (fr >> 3)) ^ (2 g + fh)) & 7 = ((fr >> 3) ^ fh) ^ (4 jj) ^ (2 g): XOR constants)
  unsigned raddr[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) raddr[t] = sl + (unsigned)((rq * 64 + cg * 8 + (t ^ ((cg ^ rq) & 7))) << 4);
  // EPI 7 / 10: EPI 3 as the PRODUCER side of a folded LayerNorm (LaGemmEpilogue.nstat_out): + out16, + the partial row sums of every
  // round, + the per-group vector rvec (7: every row of the tile in one group - the vector joins the bias; 10: groups of >= 128 rows that
  // do not end on tile edges - the wave's 128 rows lie in at most two groups: two bias sets, selected per row).
  // EPI 8 / 9: EPI 1 / 2 as the CONSUMER side (nstat_in): the normalisation applied to the product.
  // EPI 11 / 12: EPI 7 / 10 on a stream that is itself a pair of fp16 planes [hi | lo] (out16 / aux16, same bytes as fp32): the residual
  // is read from the two planes IN PLACE and written back as planes - the hi plane IS the next GEMM's operand, so the 16-bit copy costs no
  // extra bytes (a producer epilogue is bound by HBM round trips: + 2 E bytes per row were + 20 % on proj, + 8 % on lin2).
  constexpr bool RESP = EPI == 11 || EPI == 12;
  constexpr bool TWOG = EPI == 10 || EPI == 12;
  constexpr bool PROD = EPI == 7 || EPI == 10 || RESP;
  constexpr bool RES = EPI == 3 || PROD;
  constexpr bool NORM = EPI == 8 || EPI == 9;
  constexpr bool GELU = EPI == 2 || EPI == 5 || EPI == 9;
  float bias[2][8];
  float biasB[2][8];      // (two-group forms only)
  float ncol[2][8];
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    if (EPI == 6) continue;
    const float4 b0 = e.bias ? *reinterpret_cast<const float4*>(e.bias + col0 + jp * 64 + cg * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b1 = e.bias ? *reinterpret_cast<const float4*>(e.bias + col0 + jp * 64 + cg * 8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    bias[jp][0] = b0.x; bias[jp][1] = b0.y; bias[jp][2] = b0.z; bias[jp][3] = b0.w;
    bias[jp][4] = b1.x; bias[jp][5] = b1.y; bias[jp][6] = b1.z; bias[jp][7] = b1.w;
    if (TWOG) {      // a second copy for the rows of the NEXT group (selected per row: a group's rows never see another group's vector)
#pragma unroll
      for (int t = 0; t < 8; ++t) biasB[jp][t] = bias[jp][t];
      const int gB = min(row0 / e.rvec_rpg + 1, (M - 1) / e.rvec_rpg);
      const float* rv = e.rvec + (size_t)gB * N + col0 + jp * 64 + cg * 8;
      const float4 r0 = *reinterpret_cast<const float4*>(rv), r1 = *reinterpret_cast<const float4*>(rv + 4);
      biasB[jp][0] += r0.x; biasB[jp][1] += r0.y; biasB[jp][2] += r0.z; biasB[jp][3] += r0.w;
      biasB[jp][4] += r1.x; biasB[jp][5] += r1.y; biasB[jp][6] += r1.z; biasB[jp][7] += r1.w;
    }
    if (PROD && e.rvec) {      // (a wave that lies entirely beyond M in the ragged last row tile: clamped to the last group, nothing of it is stored)
      const float* rv = e.rvec + (size_t)min(row0 / e.rvec_rpg, (M - 1) / e.rvec_rpg) * N + col0 + jp * 64 + cg * 8;
      const float4 r0 = *reinterpret_cast<const float4*>(rv), r1 = *reinterpret_cast<const float4*>(rv + 4);
      bias[jp][0] += r0.x; bias[jp][1] += r0.y; bias[jp][2] += r0.z; bias[jp][3] += r0.w;
      bias[jp][4] += r1.x; bias[jp][5] += r1.y; bias[jp][6] += r1.z; bias[jp][7] += r1.w;
    }
    if (NORM) {
      const float4 c0 = *reinterpret_cast<const float4*>(e.ncol + col0 + jp * 64 + cg * 8);
      const float4 c1 = *reinterpret_cast<const float4*>(e.ncol + col0 + jp * 64 + cg * 8 + 4);
      ncol[jp][0] = c0.x; ncol[jp][1] = c0.y; ncol[jp][2] = c0.z; ncol[jp][3] = c0.w;
      ncol[jp][4] = c1.x; ncol[jp][5] = c1.y; ncol[jp][6] = c1.z; ncol[jp][7] = c1.w;
    }
  }
  // (mean, rstd) of the wave's rows 32 i + 4 rq + s (nstat_in is padded to whole row tiles: no row predicate): a ring of two row
  // blocks, block i + 2 requested when block i is done
  float4 mrv[2][2];
  auto ldmr = [&](int i) {
#ifdef LA_W4_NO_MRLOAD      // (measurement builds only, results wrong: what the statistics loads cost the epilogue - tools/normfold_ab.py)
    mrv[i & 1][0] = mrv[i & 1][1] = make_float4(0.f, 1.f, 0.f, 1.f);
    return;
#endif
    const float4* mp = reinterpret_cast<const float4*>(e.nstat_in + (size_t)(row0 + i * 32 + rq * 4) * 2);
    mrv[i & 1][0] = mp[0];
    mrv[i & 1][1] = mp[1];
  };
  if (NORM) {
    ldmr(0);
    ldmr(1);
  }
  const int rowB = TWOG ? (row0 / e.rvec_rpg + 1) * e.rvec_rpg : 0;      // first row of the next group
  const int nslots = N >> 6;      // PROD: one partial per row and 64-column round (nothing carried from round to round: registers)
  T* out16 = reinterpret_cast<T*>(e.out16);
  auto wr = [&](auto ic, auto jpc) {
    constexpr int i = decltype(ic)::value, jp = decltype(jpc)::value;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = {acc[jp][i][jj][4 * g], acc[jp][i][jj][4 * g + 1], acc[jp][i][jj][4 * g + 2], acc[jp][i][jj][4 * g + 3]};
        // XOR constants of the swizzle: units (4 jj) ^ (2 g) inside the 8-unit group -> byte offset ((4 jj ^ 2 g) & 7) * 16
        const unsigned a = (waddr ^ (unsigned)((((4 * jj) ^ (2 * g)) & 7) << 4)) + (unsigned)(g * 2048 + jj * 512);
        *reinterpret_cast<__attribute__((address_space(3))) f32x4*>((uintptr_t)a) = v;
      }
  };
  f32x4 rd0[8];
  auto rd = [&](f32x4 (&r)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t) r[t] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((uintptr_t)raddr[t]);
  };
  // rows 4 rq + s of block i, columns col0 + 64 jp + 8 cg + t
  // EPI 3: the residual rows of round (i, jp) - requested RING rounds ahead of their use (HBM round trips, nothing else to hide them)
  // (res_mod: the residual repeats every res_mod rows; the host sends such a call here only when res_mod % 256 == 0 - a tile is inside one period)
  const int rrow0 = e.res_mod > 0 ? row0 % e.res_mod : row0;
  auto ldres = [&](int i, int jp, float4 (&res)[4][2]) {
    const int row = rrow0 + i * 32 + rq * 4, col = col0 + jp * 64 + cg * 8;
    if constexpr (RESP) {      // res[s][0] = the 8 hi halves, res[s][1] = the 8 lo halves of the row segment (16 bytes each, as raw bits)
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const bool ok = row0 + i * 32 + rq * 4 + s_ < M;
        res[s_][0] = ok ? *reinterpret_cast<const float4*>(reinterpret_cast<const T*>(e.out16) + (size_t)(row + s_) * e.ld16 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        res[s_][1] = ok ? *reinterpret_cast<const float4*>(reinterpret_cast<const T*>(e.aux16) + (size_t)(row + s_) * e.ldaux + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        res[s_][h] = (e.res && row0 + i * 32 + rq * 4 + s_ < M) ? *reinterpret_cast<const float4*>(e.res + (size_t)(row + s_) * e.ldr + col + 4 * h)
                                                                  : make_float4(0.f, 0.f, 0.f, 0.f);      // (rows beyond M: the last, ragged row tile)
  };
  // EPI 5 / 6 (training): aux16 = a second 16-bit matrix of the output's shape.  5: the pre-activation (bias added, before the GELU) is
  // written there beside out16 = GELU - the forward that keeps what gelu' needs.  6: it is READ - out16 = acc * gelu'(aux16), the data
  // gradient of the layer in front of a GELU (dX = dY W of fc2 times gelu'(pre): no fp32 d-activation round trip, no la_gelu_bwd16 pass);
  // its row segments are requested a round ahead like the residual's
  T* aux16 = reinterpret_cast<T*>(e.aux16);
  auto ldaux = [&](int i, int jp, uint4 (&ax)[4]) {
    const int row = row0 + i * 32 + rq * 4, col = col0 + jp * 64 + cg * 8;
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
      ax[s_] = row + s_ < M ? *reinterpret_cast<const uint4*>(aux16 + (size_t)(row + s_) * e.ldaux + col) : make_uint4(0u, 0u, 0u, 0u);
  };
  auto out = [&](int i, int jp, const f32x4 (&r)[8], const float4 (&res)[4][2], const uint4 (&ax)[4]) {
    const int row = row0 + i * 32 + rq * 4, col = col0 + jp * 64 + cg * 8;
    if (RES) {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        float4 o0, o1;
        if constexpr (RESP) {
          // (words unpacked with bit operations: a pointer into the register array would send it to scratch memory)
          const uint32_t hw_[4] = {__builtin_bit_cast(uint32_t, res[s_][0].x), __builtin_bit_cast(uint32_t, res[s_][0].y),
                                   __builtin_bit_cast(uint32_t, res[s_][0].z), __builtin_bit_cast(uint32_t, res[s_][0].w)};
          const uint32_t lw_[4] = {__builtin_bit_cast(uint32_t, res[s_][1].x), __builtin_bit_cast(uint32_t, res[s_][1].y),
                                   __builtin_bit_cast(uint32_t, res[s_][1].z), __builtin_bit_cast(uint32_t, res[s_][1].w)};
          const bool nb = TWOG && row + s_ >= rowB;
          float c[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float hv = (t & 1) ? unpack_hi<T>(hw_[t >> 1]) : unpack_lo<T>(hw_[t >> 1]);
            const float lv = (t & 1) ? unpack_hi<T>(lw_[t >> 1]) : unpack_lo<T>(lw_[t >> 1]);
            c[t] = r[t][s_] + (nb ? biasB[jp][t] : bias[jp][t]) + (hv + lv);
          }
          o0 = make_float4(c[0], c[1], c[2], c[3]);
          o1 = make_float4(c[4], c[5], c[6], c[7]);
        } else if (EPI == 10) {
          const bool nb = row + s_ >= rowB;
          o0.x = r[0][s_] + (nb ? biasB[jp][0] : bias[jp][0]) + res[s_][0].x; o0.y = r[1][s_] + (nb ? biasB[jp][1] : bias[jp][1]) + res[s_][0].y;
          o0.z = r[2][s_] + (nb ? biasB[jp][2] : bias[jp][2]) + res[s_][0].z; o0.w = r[3][s_] + (nb ? biasB[jp][3] : bias[jp][3]) + res[s_][0].w;
          o1.x = r[4][s_] + (nb ? biasB[jp][4] : bias[jp][4]) + res[s_][1].x; o1.y = r[5][s_] + (nb ? biasB[jp][5] : bias[jp][5]) + res[s_][1].y;
          o1.z = r[6][s_] + (nb ? biasB[jp][6] : bias[jp][6]) + res[s_][1].z; o1.w = r[7][s_] + (nb ? biasB[jp][7] : bias[jp][7]) + res[s_][1].w;
        } else {
          o0.x = r[0][s_] + bias[jp][0] + res[s_][0].x; o0.y = r[1][s_] + bias[jp][1] + res[s_][0].y;
          o0.z = r[2][s_] + bias[jp][2] + res[s_][0].z; o0.w = r[3][s_] + bias[jp][3] + res[s_][0].w;
          o1.x = r[4][s_] + bias[jp][4] + res[s_][1].x; o1.y = r[5][s_] + bias[jp][5] + res[s_][1].y;
          o1.z = r[6][s_] + bias[jp][6] + res[s_][1].z; o1.w = r[7][s_] + bias[jp][7] + res[s_][1].w;
        }
        if (PROD) {      // sum x, sum x^2 over the round's 64 columns of this row: the 8 lanes of a row quad folded by DPP, lane cg == 0 stores
          float a = ((o0.x + o0.y) + (o0.z + o0.w)) + ((o1.x + o1.y) + (o1.z + o1.w));
          float b = ((o0.x * o0.x + o0.y * o0.y) + (o0.z * o0.z + o0.w * o0.w)) + ((o1.x * o1.x + o1.y * o1.y) + (o1.z * o1.z + o1.w * o1.w));
          a += dpp_mov<0xB1>(a); b += dpp_mov<0xB1>(b);         // quad_perm [1,0,3,2]
          a += dpp_mov<0x4E>(a); b += dpp_mov<0x4E>(b);         // quad_perm [2,3,0,1]
          a += dpp_mov<0x141>(a); b += dpp_mov<0x141>(b);       // row_half_mirror: the other quad of the 8-lane group
          if (cg == 0 && row + s_ < M) *reinterpret_cast<float2*>(e.nstat_out + ((size_t)(row + s_) * nslots + (col >> 6)) * 2) = make_float2(a, b);
        }
        if (row + s_ >= M) continue;
        if (!RESP && (!PROD || e.out32 != nullptr)) {      // (a producer that leaves plane pairs may skip the fp32 matrix: the patch embedding)
          float* op = e.out32 + (size_t)(row + s_) * e.ld32 + col;
          *reinterpret_cast<float4*>(op) = o0;
          *reinterpret_cast<float4*>(op + 4) = o1;
        }
        if (PROD) {
          // the 16-bit copy is an MFMA operand of the next GEMM: it SATURATES at the fp16 range instead of turning into inf (an un-normalised
          // stream has no range guarantee; SAM checkpoints stay three orders of magnitude below).  aux16 (optional): the lo plane
          // rn16(x - hi) beside it - [hi | lo] rows, the LA_F16X2 operand of the SAM neck's 1 x 1 convolution, without a pass of its own
          constexpr float HMAX = 65504.0f;
          float c[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
          for (int t = 0; t < 8; ++t) c[t] = __builtin_amdgcn_fmed3f(c[t], -HMAX, HMAX);
          uint4 pk;
          pk.x = pack2<T>(c[0], c[1]); pk.y = pack2<T>(c[2], c[3]); pk.z = pack2<T>(c[4], c[5]); pk.w = pack2<T>(c[6], c[7]);
          *reinterpret_cast<uint4*>(out16 + (size_t)(row + s_) * e.ld16 + col) = pk;
          if (aux16) {
            uint4 lo;
            lo.x = pack2<T>(c[0] - unpack_lo<T>(pk.x), c[1] - unpack_hi<T>(pk.x)); lo.y = pack2<T>(c[2] - unpack_lo<T>(pk.y), c[3] - unpack_hi<T>(pk.y));
            lo.z = pack2<T>(c[4] - unpack_lo<T>(pk.z), c[5] - unpack_hi<T>(pk.z)); lo.w = pack2<T>(c[6] - unpack_lo<T>(pk.w), c[7] - unpack_hi<T>(pk.w));
            *reinterpret_cast<uint4*>(aux16 + (size_t)(row + s_) * e.ldaux + col) = lo;
          }
        } else if (out16) {
          uint4 pk;
          pk.x = pack2<T>(o0.x, o0.y); pk.y = pack2<T>(o0.z, o0.w); pk.z = pack2<T>(o1.x, o1.y); pk.w = pack2<T>(o1.z, o1.w);
          *reinterpret_cast<uint4*>(out16 + (size_t)(row + s_) * e.ld16 + col) = pk;
        }
      }
    } else {
      // bias (+ GELU) on row pairs (s, s + 1) of one column: adjacent registers = one packed-fp32 operand.  The GELU runs on eight
      // pairs IN LOCKSTEP (same arithmetic as gelu_erf_pk, one polynomial step of all eight at a time): written pair by pair, hipcc
      // emits each 13-deep dependent chain on its own with an s_nop behind every packed instruction - a third of the epilogue's
      // instructions - and a wave that is alone on its SIMD has nothing else to put there.
      f32x2 v[8][2];
      if constexpr (NORM) {      // rstd (acc - mean ncol) + bias on the row pairs (2 sp, 2 sp + 1); both pairs first: the slab registers die here
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          const float4 m4 = mrv[i & 1][sp];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            v[t][sp] = f32x2{r[t][2 * sp], r[t][2 * sp + 1]} - f32x2{m4.x, m4.z} * f32x2{ncol[jp][t], ncol[jp][t]};
            v[t][sp] = v[t][sp] * f32x2{m4.y, m4.w} + f32x2{bias[jp][t], bias[jp][t]};
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if constexpr (NORM) continue;
          v[t][sp] = f32x2{r[t][2 * sp], r[t][2 * sp + 1]};
          if (EPI != 6) v[t][sp] = v[t][sp] + f32x2{bias[jp][t], bias[jp][t]};        // (a data gradient has no bias)
        }
        if (EPI == 5) {                        // the two rows of this pair as they are before the activation
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            uint4 pk;
            pk.x = pack2<T>(v[0][sp][h], v[1][sp][h]);
            pk.y = pack2<T>(v[2][sp][h], v[3][sp][h]);
            pk.z = pack2<T>(v[4][sp][h], v[5][sp][h]);
            pk.w = pack2<T>(v[6][sp][h], v[7][sp][h]);
            if (row + 2 * sp + h < M) *reinterpret_cast<uint4*>(aux16 + (size_t)(row + 2 * sp + h) * e.ldaux + col) = pk;
          }
        }
        if (EPI == 6) {                        // v *= gelu'(x), x = the saved pre-activation: 0.5 (1 + erf(x / sqrt 2)) + x exp(-x^2 / 2) / sqrt(2 pi)
          f32x2 x[8], u[8], tt[8], p[8], ex[8];
          const T* a0 = reinterpret_cast<const T*>(&ax[2 * sp]);
          const T* a1 = reinterpret_cast<const T*>(&ax[2 * sp + 1]);
#define LA_W4_STEP(expr)                          \
  _Pragma("unroll") for (int t = 0; t < 8; ++t) { \
    expr;                                         \
  }                                               \
  __builtin_amdgcn_sched_barrier(0);
          LA_W4_STEP(x[t] = (f32x2{(float)a0[t], (float)a1[t]}))
          LA_W4_STEP(u[t] = x[t] * 0.70710678118654752440f)
          LA_W4_STEP(u[t].x = __builtin_amdgcn_fmed3f(u[t].x, -3.0f, 3.0f); u[t].y = __builtin_amdgcn_fmed3f(u[t].y, -3.0f, 3.0f))
          LA_W4_STEP(tt[t] = u[t] * u[t])
          LA_W4_STEP(p[t] = tt[t] * -3.753537037e-09f + 1.995845196e-07f)
          LA_W4_STEP(p[t] = p[t] * tt[t] + -4.771217391e-06f)
          LA_W4_STEP(p[t] = p[t] * tt[t] + 6.851813669e-05f)
          LA_W4_STEP(p[t] = p[t] * tt[t] + -6.692335592e-04f)
          LA_W4_STEP(p[t] = p[t] * tt[t] + 4.784903489e-03f)
          LA_W4_STEP(p[t] = p[t] * tt[t] + -2.622046508e-02f)
          LA_W4_STEP(p[t] = p[t] * tt[t] + 1.123065501e-01f)
          LA_W4_STEP(p[t] = p[t] * tt[t] + -3.759292066e-01f)
          LA_W4_STEP(p[t] = p[t] * tt[t] + 1.128377676e+00f)
          LA_W4_STEP(p[t] = p[t] * u[t])                                   // erf(x / sqrt 2)
          LA_W4_STEP(ex[t] = x[t] * x[t])
          LA_W4_STEP(ex[t] = ex[t] * -0.72134752044448170368f)            // -x^2 / 2 in base 2
          LA_W4_STEP(ex[t].x = __builtin_amdgcn_exp2f(ex[t].x); ex[t].y = __builtin_amdgcn_exp2f(ex[t].y))
          LA_W4_STEP(p[t] = p[t] * 0.5f + 0.5f)
          LA_W4_STEP(x[t] = x[t] * 0.39894228040143267794f)
          LA_W4_STEP(p[t] = x[t] * ex[t] + p[t])
          LA_W4_STEP(v[t][sp] = v[t][sp] * p[t])
#undef LA_W4_STEP
        }
        if (GELU) {        // x Phi(clamp(x)), Phi(c) = 0.5 + c R(c^2): gelu_erf_pk's arithmetic, 14 instructions per pair
          f32x2 u[8], tt[8], p[8];
#define LA_W4_STEP(expr)                          \
  _Pragma("unroll") for (int t = 0; t < 8; ++t) { \
    expr;                                         \
  }                                               \
  __builtin_amdgcn_sched_barrier(0);
          LA_W4_STEP(u[t].x = __builtin_amdgcn_fmed3f(v[t][sp].x, -GELU_CLAMP, GELU_CLAMP); u[t].y = __builtin_amdgcn_fmed3f(v[t][sp].y, -GELU_CLAMP, GELU_CLAMP))
          LA_W4_STEP(tt[t] = u[t] * u[t])
          LA_W4_STEP(p[t] = tt[t] * GELU_R9 + GELU_R8)
          LA_W4_STEP(p[t] = p[t] * tt[t] + GELU_R7)
          LA_W4_STEP(p[t] = p[t] * tt[t] + GELU_R6)
          LA_W4_STEP(p[t] = p[t] * tt[t] + GELU_R5)
          LA_W4_STEP(p[t] = p[t] * tt[t] + GELU_R4)
          LA_W4_STEP(p[t] = p[t] * tt[t] + GELU_R3)
          LA_W4_STEP(p[t] = p[t] * tt[t] + GELU_R2)
          LA_W4_STEP(p[t] = p[t] * tt[t] + GELU_R1)
          LA_W4_STEP(p[t] = p[t] * tt[t] + GELU_R0)
          LA_W4_STEP(p[t] = p[t] * u[t] + 0.5f)
          LA_W4_STEP(v[t][sp] = v[t][sp] * p[t])
#undef LA_W4_STEP
        }
      }
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        uint4 pk;
        pk.x = pack2<T>(v[0][s_ >> 1][s_ & 1], v[1][s_ >> 1][s_ & 1]);
        pk.y = pack2<T>(v[2][s_ >> 1][s_ & 1], v[3][s_ >> 1][s_ & 1]);
        pk.z = pack2<T>(v[4][s_ >> 1][s_ & 1], v[5][s_ >> 1][s_ & 1]);
        pk.w = pack2<T>(v[6][s_ >> 1][s_ & 1], v[7][s_ >> 1][s_ & 1]);
        if (row + s_ < M) *reinterpret_cast<uint4*>(out16 + (size_t)(row + s_) * e.ld16 + col) = pk;
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  // rounds r = 2 i + jp.  ONE fragment set: the wave must not spill here - a reload is a VMEM load behind the round's stores, i.e. an
  // s_waitcnt vmcnt(0) that drains them (and the LDS-DMA pieces of the next tile) once per round
  float4 res0[4][2], res1[4][2];
  uint4 aux0[4], aux1[4];
  if (RES) {
    ldres(0, 0, res0);
    ldres(0, 1, res1);
  }

  if (EPI == 6) {
    ldaux(0, 0, aux0);
    ldaux(0, 1, aux1);
  }
  wr(I0{}, I0{});
  rd(rd0);
  wr(I0{}, I1{});
  __builtin_amdgcn_sched_barrier(0);
  out(0, 0, rd0, res0, aux0);
  if (RES) ldres(1, 0, res0);
  if (EPI == 6) ldaux(1, 0, aux0);
  __builtin_amdgcn_sched_barrier(0);
  rd(rd0);
  wr(I1{}, I0{});
  __builtin_amdgcn_sched_barrier(0);
  out(0, 1, rd0, res1, aux1);
  if (NORM) ldmr(2);
  if (RES) ldres(1, 1, res1);
  if (EPI == 6) ldaux(1, 1, aux1);
  __builtin_amdgcn_sched_barrier(0);
  rd(rd0);
  wr(I1{}, I1{});
  __builtin_amdgcn_sched_barrier(0);
  out(1, 0, rd0, res0, aux0);
  if (RES) ldres(2, 0, res0);
  if (EPI == 6) ldaux(2, 0, aux0);
  __builtin_amdgcn_sched_barrier(0);
  rd(rd0);
  wr(I2{}, I0{});
  __builtin_amdgcn_sched_barrier(0);
  out(1, 1, rd0, res1, aux1);
  if (NORM) ldmr(3);
  if (RES) ldres(2, 1, res1);
  if (EPI == 6) ldaux(2, 1, aux1);
  __builtin_amdgcn_sched_barrier(0);
  rd(rd0);
  wr(I2{}, I1{});
  __builtin_amdgcn_sched_barrier(0);
  out(2, 0, rd0, res0, aux0);
  if (RES) ldres(3, 0, res0);
  if (EPI == 6) ldaux(3, 0, aux0);
  __builtin_amdgcn_sched_barrier(0);
  rd(rd0);
  wr(I3{}, I0{});
  __builtin_amdgcn_sched_barrier(0);
  out(2, 1, rd0, res1, aux1);
  if (RES) ldres(3, 1, res1);
  if (EPI == 6) ldaux(3, 1, aux1);
  __builtin_amdgcn_sched_barrier(0);
  rd(rd0);
  wr(I3{}, I1{});
  __builtin_amdgcn_sched_barrier(0);
  out(3, 0, rd0, res0, aux0);
  __builtin_amdgcn_sched_barrier(0);
  rd(rd0);
  __builtin_amdgcn_sched_barrier(0);
  out(3, 1, rd0, res1, aux1);
  __builtin_amdgcn_sched_barrier(0);
}

// DIRECT: every tile is interior and unmapped (M % 256 == 0, no output row map, no V^T columns): epilogue_w4 only.  The two epilogues
// do not share a kernel: with both behind a branch hipcc spills 128 accumulator registers to scratch in front of it.
template <typename T, int EPI, int ABL, bool DIRECT>
__global__ __launch_bounds__(256, 1) void gemm_t256w_kernel(const T* __restrict__ A, int lda, const T* __restrict__ Wt, int ldw, int M, int N,
                                                             int K, LaGemmEpilogue e, int gm, int stg) {
  constexpr int BK_ = 64;
  constexpr unsigned REG = 32768;                     // one operand of one k-tile
  constexpr int SEAM = (EPI == 3 || EPI == 5 || EPI == 7 || EPI >= 10) ? 47 : 32;      // epilogue stores per wave that the first two waits of a tile may leave outstanding
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef LA_DEBUG
  const bool stamps = ((gm >> 10) & 1) && blockIdx.x == 0;      // seam timeline of workgroup 0 (la_dbg_w4_stamps)
  int stamp_i = 0;
  auto stamp = [&](int tag) {
    if (stamps && stamp_i < LA_W4_NSTAMP) {
      unsigned long long t_;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
      if (lane == 0) g_w4_stamps[wave * LA_W4_NSTAMP + stamp_i] = (t_ << 8) | (unsigned)tag;
      ++stamp_i;
    }
  };
#else
  auto stamp = [](int) {};
#endif
#ifdef LA_DEBUG
  const bool nostore = (gm >> 8) & 1;                 // measurement ablation (la_gemm_variant bit 8): the epilogue's global stores are skipped
#else
  constexpr bool nostore = false;
#endif
  gm &= 0xff;
  // start stagger: workgroup class (blockIdx / 8) % P waits class x D x 1024 cycles, so that the epilogues (HBM bursts with the matrix
  // pipe idle) of different CUs fall into each other's main loops instead of all at once
  if ((stg >> 16) > 1) {
    const int cls = (blockIdx.x >> 3) % (stg >> 16);
    for (int i = 0; i < cls * (stg & 0xffff); ++i) __builtin_amdgcn_s_sleep(16);      // 16 x 64 cycles
  }
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  const int ntn = N / 256, ntm = (M + 255) / 256, ntiles = ntm * ntn;
  const unsigned lds0 = lds_addr_of(smem);
  char* slab = smem + 4 * REG + wave * 8192;          // 8 KiB per wave: epilogue_w4's fp32 slab, or epilogue_wave's 2 KiB slab + row table
  unsigned* rtab = nullptr;
  if (EPI == 1 && e.map != LA_MAP_NONE) rtab = reinterpret_cast<unsigned*>(slab + 2048);

  // ---- per-lane fragment addresses: row * 128 + swizzled chunk; + i * 4096 selects the 32-row fragment -----------------------------
  unsigned aaddr[4], waddr[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const unsigned coff = (unsigned)(((ks * 2 + fh) ^ ((fr >> 1) & 7)) << 4);
    aaddr[ks] = lds0 + (wr * 128 + fr) * 128 + coff;
    waddr[ks] = lds0 + 2 * REG + (wc * 128 + fr) * 128 + coff;
  }
  // ---- LDS-DMA: destination bases of this wave's pieces (buffer 0) and per-lane source offsets ------------------------------------
  //   A0 piece i (4): rows (wave >> 1) * 128 + ((wave & 1) * 4 + i) * 8      A1: the same + 64      W piece i (8): rows (wave * 8 + i) * 8
  const unsigned dstA = lds0 + (wave >> 1) * 16384 + (wave & 1) * 4096;
  const unsigned dstW = lds0 + 2 * REG + wave * 8192;
  unsigned soA0[4], soA1[4], soA1n[4], soW[8];
  int m0 = 0, n0 = 0, m0n = 0, n0n = 0;
  auto plan = [&](int tile, int& pm0, int& pn0, unsigned (&a0)[4], unsigned (&a1)[4], unsigned (&w)[8]) {
    int tm_, tn_;
    tile_coords(xcd_remap(tile, ntiles), ntm, ntn, gm, tm_, tn_);
    pm0 = tm_ * 256;
    pn0 = tn_ * 256;
    // (EPI >= 7: the lane-dependent row / chunk terms below are loop invariants that hipcc hoists and then - these epilogues leave no
    // register for them - spills, with a reload + s_waitcnt vmcnt(0) per use; an opaque copy of the lane keeps them local to the call)
    int lane_ = lane;
    if constexpr (EPI >= 7) asm volatile("" : "+v"(lane_));
    const int lane = lane_;
    const int lr = lane >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (wave >> 1) * 128 + ((wave & 1) * 4 + i) * 8 + lr;
      const int ch = ((lane & 7) ^ ((r >> 1) & 7)) << 3;                 // (r + 64 has the same swizzle term)
      a0[i] = (unsigned)(((size_t)a_row(e, min(pm0 + r, M - 1)) * lda + ch) * sizeof(T));
      a1[i] = (unsigned)(((size_t)a_row(e, min(pm0 + r + 64, M - 1)) * lda + ch) * sizeof(T));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = (wave * 8 + i) * 8 + lr;
      const int ch = ((lane & 7) ^ ((r >> 1) & 7)) << 3;
      w[i] = (unsigned)(((size_t)(pn0 + r) * ldw + ch) * sizeof(T));
    }
  };

  f32x16 acc[2][4][2];                                // [column half][i][j in half]: acc[h] is one epilogue_wave block
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc_zero(acc[h][i][j]);
  u32x4 wf[4][4], af[2][2];

  int tile = blockIdx.x;
  plan(tile, m0, n0, soA0, soA1, soW);
  const int nk = K / BK_;                             // >= 2 (host side)
  unsigned bofs = 0;                                  // buffer of the current k-tile: 0 / REG
  // ---- prologue: A0W(0), A1(0) into buffer 0, A0W(1) into buffer 1 ---------------------------------------------------------------
  {
    const T* sa = A + a_koff(e, 0);
    const T* sw = Wt;
    LA_W4_QA0(sa, 0, 0) LA_W4_QA0(sa, 1, 0) LA_W4_QA0(sa, 2, 0) LA_W4_QA0(sa, 3, 0)
    LA_W4_QW(sw, 0, 0) LA_W4_QW(sw, 1, 0) LA_W4_QW(sw, 2, 0) LA_W4_QW(sw, 3, 0)
    LA_W4_QW(sw, 4, 0) LA_W4_QW(sw, 5, 0) LA_W4_QW(sw, 6, 0) LA_W4_QW(sw, 7, 0)
    LA_W4_QA1(sa, 0, 0) LA_W4_QA1(sa, 1, 0) LA_W4_QA1(sa, 2, 0) LA_W4_QA1(sa, 3, 0)
    sa = A + a_koff(e, BK_);
    sw = Wt + BK_;
    LA_W4_QA0(sa, 0, REG) LA_W4_QA0(sa, 1, REG) LA_W4_QA0(sa, 2, REG) LA_W4_QA0(sa, 3, REG)
    LA_W4_QW(sw, 0, REG) LA_W4_QW(sw, 1, REG) LA_W4_QW(sw, 2, REG) LA_W4_QW(sw, 3, REG)
    LA_W4_QW(sw, 4, REG) LA_W4_QW(sw, 5, REG) LA_W4_QW(sw, 6, REG) LA_W4_QW(sw, 7, REG)
  }
  wait_vm_lgkm0<16>();                                // A0W(0) landed (A1(0), A0W(1) may still be out)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < 4; ++j) wf[0][j] = lds_read16(waddr[0] + j * 4096);
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) af[0][ii] = lds_read16(aaddr[0] + ii * 4096);
  // aaddr[0] / waddr[0] point at the buffer their NEXT read comes from: block 3 (A of this k-tile), then block 7 (the next k-tile)
  waddr[0] ^= REG;

  int kt = 0;
  stamp(1);
  bool seam = false;                                  // the previous k-tile ended in an epilogue whose stores are exactly counted
  for (;;) {
    const int next = tile + gridDim.x;
    const bool more = next < ntiles;
    const bool last = kt + 1 == nk, prelast = kt + 2 == nk;
    // sources of this k-tile's requests: A1 of stream position t+1, A0 / W of t+2 (at the end of the tile sequence: valid bytes into dead regions)
    const int kA1 = last ? 0 : kt + 1;
    const int kAW = kt + 2 < nk ? kt + 2 : kt + 2 - nk;
    if (last && more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) soA1[i] = soA1n[i];
    }
    const T* sa1 = A + a_koff(e, kA1 * BK_);
    const unsigned bo1 = bofs ^ REG;                  // A1(t+1) lands in the OTHER buffer
    // ================= the k-tile: LA_W4_BLOCK_0 .. 7 (gemm_w4_ktile.inc, generated by tools/gen/w4_ktile.py) =========================
    // H1 blocks 0-2 (ks 0-2): reads of ks + 1 in the order of their first use (A_i0, W_0 .. W_3, A_i1), the A1 pieces; block 2 ends in wait + B1
    LA_W4_BLOCK_0
    LA_W4_BLOCK_1
    LA_W4_BLOCK_2
    if (prelast && more) plan(next, m0n, n0n, soA0, soA1n, soW);         // A0W(t+2) is the next tile's first k-tile
    const T* saw = A + a_koff(e, kAW * BK_);
    const T* sww = Wt + kAW * BK_;
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) waddr[ks] ^= REG;  // (their next use is H1 of the next k-tile)
    // block 3 (ks 3): reads of H2 ks 0, A0 pieces 0, 1
    LA_W4_BLOCK_3
    aaddr[0] ^= REG;                                  // next use: block 7, the first fragments of k-tile t+1
    // H2 blocks 4-6 (ks 0-2): reads of ks + 1 (A_2, A_3 only), A0 pieces 2, 3, W pieces 0-6; block 6 ends in wait + B2
    LA_W4_BLOCK_4
    LA_W4_BLOCK_5
    LA_W4_BLOCK_6
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) aaddr[ks] ^= REG;
    // block 7 (ks 3): the first fragments of k-tile t+1 (other buffer), W piece 7
    LA_W4_BLOCK_7
    waddr[0] ^= REG;
    bofs ^= REG;
    seam = false;
    ++kt;
    if (!last) continue;
    // ================= seam: epilogue of the finished tile =========================================================================
    stamp(2);                                          // main loop done
    asm volatile("s_nop 15" ::: "memory");             // the last MFMA's result -> first VALU reader (12 wait states; see MfmaA)
    // (the empty asm re-defines the accumulators in the AGPR class HERE: without it hipcc hoists the 128 AGPR -> VGPR copies the epilogue
    // needs into the k-tile loop, which then spills its DMA offsets - every reload is an s_waitcnt vmcnt(0) in front of a piece)
    const bool vtile = !DIRECT && EPI == 1 && e.vt != nullptr && n0 >= e.vt_col0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+a"(acc[0][i][j]));
    // EPI >= 7 (folded LayerNorm: more live values in the epilogue than the register file has beside the main loop's): the 16 DMA source
    // offsets of the next tile wait out the epilogue in LDS - the wave's own A1 region of the buffer the finished k-tile used (4 KiB, 64 B
    // per lane) is dead from barrier B2 of that k-tile until the next tile's first A1 pieces, which are issued behind the epilogue.
    // Without it hipcc spills them to scratch and reloads them INSIDE the k-tile loop (a vmcnt(0) in front of every piece).
    constexpr bool STASH = DIRECT && EPI >= 7;
    const unsigned stash = dstA + 8192 + (bofs ^ REG) + (unsigned)lane * 64;
    if constexpr (STASH) {
      const u32x4 q0 = {soA0[0], soA0[1], soA0[2], soA0[3]}, q1 = {soA1[0], soA1[1], soA1[2], soA1[3]};
      const u32x4 q2 = {soW[0], soW[1], soW[2], soW[3]}, q3 = {soW[4], soW[5], soW[6], soW[7]};
      asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\tds_write_b128 %0, %3 offset:32\n\tds_write_b128 %0, %4 offset:48"
                   :
                   : "v"(stash), "v"(q0), "v"(q1), "v"(q2), "v"(q3)
                   : "memory");
    }
    if constexpr (DIRECT) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+a"(acc[1][i][j]));
      epilogue_w4<T, EPI>(slab, acc, m0 + wr * 128, n0 + wc * 128, e, lane, M, N);
    } else {
      epilogue_wave<T, EPI>(slab, rtab, acc[0], m0 + wr * 128, n0 + wc * 128, n0, M, e, lane, nostore);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+a"(acc[1][i][j]));
      epilogue_wave<T, EPI>(slab, rtab, acc[1], m0 + wr * 128, n0 + wc * 128 + 64, n0, M, e, lane, nostore);
    }
    seam = (m0 + 256 <= M) && !vtile;
    stamp(3);                                          // epilogue issued
    if (!more) break;
    if constexpr (STASH) {
      u32x4 q0, q1, q2, q3;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
                   : "v"(stash)
                   : "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        soA0[i] = q0[i];
        soA1[i] = q1[i];
        soW[i] = q2[i];
        soW[4 + i] = q3[i];
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc_zero(acc[h][i][j]);
    m0 = m0n;
    n0 = n0n;
    tile = next;
    kt = 0;
    if constexpr (DIRECT) {
      // the first fragments of the next tile again (block 7 already read them): 24 registers that are then dead across the epilogue,
      // which needs them for its residual ring - six LDS reads per tile against spills in the epilogue
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[0][j] = lds_read16((waddr[0] ^ REG) + j * 4096);
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) af[0][ii] = lds_read16((aaddr[0]) + ii * 4096);
    }
    stamp(1);                                          // accumulators cleared: the next tile's main loop starts
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail's surplus requests must have landed before the LDS is released
}

template <typename T, int EPI, int ABL, bool DIRECT>
static void launch_t256w_abl(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, int gm, hipStream_t st) {
  constexpr int LDS = 4 * 32768 + 4 * 8192;      // two k-tile buffers + 8 KiB slab per wave: all 160 KiB
  static unsigned long long attr_mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(gemm_t256w_kernel<T, EPI, ABL, DIRECT>), LDS, attr_mask);
  static int ncu_of[64] = {0};                        // per device: a process may drive several GPUs
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& ncu = ncu_of[dev & 63];
  if (ncu == 0) {
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (ncu <= 0) ncu = 256;
  }
  const int ntiles = ((M + 255) / 256) * (N / 256);
  int grid = ntiles < ncu ? ntiles : ncu;
  int stg = 0;
  static const char* genv = la_dbg_env("LA_W4_GRID");        // debugging: workgroups launched
  if (genv && atoi(genv) > 0 && atoi(genv) < grid) grid = atoi(genv);
  static const char* senv = la_dbg_env("LA_W4_STAGGER");     // debugging: "P,D" = P start classes, D x 1024 cycles apart
  if (senv) stg = (atoi(senv) << 16) | (strchr(senv, ',') ? atoi(strchr(senv, ',') + 1) : 0);
  hipLaunchKernelGGL((gemm_t256w_kernel<T, EPI, ABL, DIRECT>), dim3(grid), dim3(256), LDS, st, reinterpret_cast<const T*>(A), lda,
                     reinterpret_cast<const T*>(W), ldw, M, N, K, e, gm & 0x5ff, stg);
}

template <typename T, int EPI>
void launch_t256w(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, int gm, hipStream_t st) {
#ifdef LA_DEBUG
  // la_gemm_variant bits 12-15 = ABL (EPI 1 only)
  if (EPI == 1) {
    switch ((gm >> 12) & 15) {
      case 1: return launch_t256w_abl<T, 1, 1, false>(A, lda, W, ldw, M, N, K, e, gm, st);
      case 2: return launch_t256w_abl<T, 1, 2, false>(A, lda, W, ldw, M, N, K, e, gm, st);
      case 3: return launch_t256w_abl<T, 1, 3, false>(A, lda, W, ldw, M, N, K, e, gm, st);
      case 4: return launch_t256w_abl<T, 1, 4, false>(A, lda, W, ldw, M, N, K, e, gm, st);
      case 5: return launch_t256w_abl<T, 1, 5, false>(A, lda, W, ldw, M, N, K, e, gm, st);
      case 7: return launch_t256w_abl<T, 1, 7, false>(A, lda, W, ldw, M, N, K, e, gm, st);
      case 8: return launch_t256w_abl<T, 1, 8, false>(A, lda, W, ldw, M, N, K, e, gm, st);
      case 14: return launch_t256w_abl<T, 1, 14, false>(A, lda, W, ldw, M, N, K, e, gm, st);
      default: break;
    }
  }
#endif
  // (a ragged last row tile - M % 256 != 0: the HF encoders' 57664 = 225.25 tiles - stays on the direct epilogue: its loads and stores are
  // predicated on the row; a residual modulo res_mod needs whole tiles inside a period and is only sent here with M % 256 == 0)
  const bool direct = e.map == LA_MAP_NONE && !e.vt && !((gm >> 8) & 1);
  if (direct) launch_t256w_abl<T, EPI, 0, true>(A, lda, W, ldw, M, N, K, e, gm, st);
  else launch_t256w_abl<T, EPI, 0, false>(A, lda, W, ldw, M, N, K, e, gm, st);
}

// EPI 5 / 6 (GELU forward that also keeps the pre-activation; data gradient times gelu' - training only) exist on the direct epilogue alone
template <typename T, int EPI>
void launch_t256w_fused(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, int gm, hipStream_t st) {
  launch_t256w_abl<T, EPI, 0, true>(A, lda, W, ldw, M, N, K, e, gm & ~0x100, st);
}
template void launch_t256w_fused<f16_t, 5>(const void*, int, const void*, int, int, int, int, const LaGemmEpilogue&, int, hipStream_t);
template void launch_t256w_fused<f16_t, 6>(const void*, int, const void*, int, int, int, int, const LaGemmEpilogue&, int, hipStream_t);
template void launch_t256w_fused<bf16_t, 5>(const void*, int, const void*, int, int, int, int, const LaGemmEpilogue&, int, hipStream_t);
template void launch_t256w_fused<bf16_t, 6>(const void*, int, const void*, int, int, int, int, const LaGemmEpilogue&, int, hipStream_t);
// EPI 7 / 10 / 8 / 9: the producer and consumer sides of a folded LayerNorm (LaGemmEpilogue.nstat_out / nstat_in), fp16 operands
template void launch_t256w_fused<f16_t, 7>(const void*, int, const void*, int, int, int, int, const LaGemmEpilogue&, int, hipStream_t);
template void launch_t256w_fused<f16_t, 10>(const void*, int, const void*, int, int, int, int, const LaGemmEpilogue&, int, hipStream_t);
template void launch_t256w_fused<f16_t, 8>(const void*, int, const void*, int, int, int, int, const LaGemmEpilogue&, int, hipStream_t);
template void launch_t256w_fused<f16_t, 9>(const void*, int, const void*, int, int, int, int, const LaGemmEpilogue&, int, hipStream_t);
template void launch_t256w_fused<f16_t, 11>(const void*, int, const void*, int, int, int, int, const LaGemmEpilogue&, int, hipStream_t);
template void launch_t256w_fused<f16_t, 12>(const void*, int, const void*, int, int, int, int, const LaGemmEpilogue&, int, hipStream_t);

#define LA_W4_INST(T, EPI) \
  template void launch_t256w<T, EPI>(const void*, int, const void*, int, int, int, int, const LaGemmEpilogue&, int, hipStream_t);
LA_W4_INST(f16_t, 1) LA_W4_INST(f16_t, 2) LA_W4_INST(f16_t, 3)
LA_W4_INST(bf16_t, 1) LA_W4_INST(bf16_t, 2) LA_W4_INST(bf16_t, 3)

}  // namespace la

#ifdef LA_DEBUG
// seam timeline of workgroup 0 of the last gemm_t256w launch made with la_gemm_variant bit 10: 4 waves x 128 entries of (s_memtime << 8 | tag)
extern "C" int la_dbg_w4_stamps(unsigned long long* host_out) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(la::g_w4_stamps), sizeof(unsigned long long) * 4 * la::LA_W4_NSTAMP);
}
extern "C" int la_dbg_w4_stamps_clear() {
  static unsigned long long z[4 * la::LA_W4_NSTAMP] = {0};
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(la::g_w4_stamps), z, sizeof(z));
}
#endif
