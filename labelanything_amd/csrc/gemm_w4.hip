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
#include <type_traits>
#include "gemm_shared.h"

namespace la {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));      // an operand fragment as a plain register tuple ("v" constraint)

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

template <int N> __device__ __forceinline__ void wait_vm_lgkm0() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// ---- the k-tile body is written out slot by slot: one MFMA, at most one memory instruction, pinned by sched_barrier(0) ----------------
#define LA_W4_SB __builtin_amdgcn_sched_barrier(0);
#define LA_W4_MF(IB, ii, j, PAR, KS)                                                                                            \
  MfmaA<T>::go(acc[(j) >> 1][(IB) + (ii)][(j) & 1], af[PAR][ii], wf[KS][j]); \
  LA_W4_SB
#define LA_W4_RW(KS, j) wf[KS][j] = lds_read16(waddr[KS] + (j) * 4096); LA_W4_SB
#define LA_W4_RA(PAR, ii, i, KS) af[PAR][ii] = lds_read16(aaddr[KS] + (i) * 4096); LA_W4_SB
#define LA_W4_PA0(src, i, bo) dma_piece<(i) * 1024>(src, soA0[i], dstA + (bo)); LA_W4_SB
#define LA_W4_PA1(src, i, bo) dma_piece<8192 + (i) * 1024>(src, soA1[i], dstA + (bo)); LA_W4_SB
#define LA_W4_PW(src, i, bo) dma_piece<(i) * 1024>(src, soW[i], dstW + (bo)); LA_W4_SB

template <typename T, int EPI>
__global__ __launch_bounds__(256, 1) void gemm_t256w_kernel(const T* __restrict__ A, int lda, const T* __restrict__ Wt, int ldw, int M, int N,
                                                             int K, LaGemmEpilogue e, int gm) {
  constexpr int BK_ = 64;
  constexpr unsigned REG = 32768;                     // one operand of one k-tile
  constexpr int SEAM = (EPI == 3) ? 47 : 32;          // epilogue stores per wave that the first two waits of a tile may leave outstanding
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  const int ntn = N / 256, ntm = (M + 255) / 256, ntiles = ntm * ntn;
  const unsigned lds0 = lds_addr_of(smem);
  char* slab = smem + 4 * REG + wave * 2048;
  unsigned* rtab = nullptr;
  if (EPI == 1 && e.map != LA_MAP_NONE) rtab = reinterpret_cast<unsigned*>(smem + 4 * REG + 4 * 2048 + wave * 512);

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
    LA_W4_PA0(sa, 0, 0) LA_W4_PA0(sa, 1, 0) LA_W4_PA0(sa, 2, 0) LA_W4_PA0(sa, 3, 0)
    LA_W4_PW(sw, 0, 0) LA_W4_PW(sw, 1, 0) LA_W4_PW(sw, 2, 0) LA_W4_PW(sw, 3, 0)
    LA_W4_PW(sw, 4, 0) LA_W4_PW(sw, 5, 0) LA_W4_PW(sw, 6, 0) LA_W4_PW(sw, 7, 0)
    LA_W4_PA1(sa, 0, 0) LA_W4_PA1(sa, 1, 0) LA_W4_PA1(sa, 2, 0) LA_W4_PA1(sa, 3, 0)
    sa = A + a_koff(e, BK_);
    sw = Wt + BK_;
    LA_W4_PA0(sa, 0, REG) LA_W4_PA0(sa, 1, REG) LA_W4_PA0(sa, 2, REG) LA_W4_PA0(sa, 3, REG)
    LA_W4_PW(sw, 0, REG) LA_W4_PW(sw, 1, REG) LA_W4_PW(sw, 2, REG) LA_W4_PW(sw, 3, REG)
    LA_W4_PW(sw, 4, REG) LA_W4_PW(sw, 5, REG) LA_W4_PW(sw, 6, REG) LA_W4_PW(sw, 7, REG)
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
    // ================= H1 ==========================================================================================================
    // block 0 (ks 0): reads of ks 1, A1 pieces 0, 1
    LA_W4_MF(0, 0, 0, 0, 0) LA_W4_RW(1, 0) LA_W4_MF(0, 0, 1, 0, 0) LA_W4_RW(1, 1) LA_W4_MF(0, 0, 2, 0, 0) LA_W4_RW(1, 2)
    LA_W4_MF(0, 0, 3, 0, 0) LA_W4_RW(1, 3) LA_W4_MF(0, 1, 0, 0, 0) LA_W4_RA(1, 0, 0, 1) LA_W4_MF(0, 1, 1, 0, 0) LA_W4_RA(1, 1, 1, 1)
    LA_W4_MF(0, 1, 2, 0, 0) LA_W4_PA1(sa1, 0, bo1) LA_W4_MF(0, 1, 3, 0, 0) LA_W4_PA1(sa1, 1, bo1)
    // block 1 (ks 1): reads of ks 2, A1 piece 2
    LA_W4_MF(0, 0, 0, 1, 1) LA_W4_RW(2, 0) LA_W4_MF(0, 0, 1, 1, 1) LA_W4_RW(2, 1) LA_W4_MF(0, 0, 2, 1, 1) LA_W4_RW(2, 2)
    LA_W4_MF(0, 0, 3, 1, 1) LA_W4_RW(2, 3) LA_W4_MF(0, 1, 0, 1, 1) LA_W4_RA(0, 0, 0, 2) LA_W4_MF(0, 1, 1, 1, 1) LA_W4_RA(0, 1, 1, 2)
    LA_W4_MF(0, 1, 2, 1, 1) LA_W4_PA1(sa1, 2, bo1) LA_W4_MF(0, 1, 3, 1, 1)
    // block 2 (ks 2): reads of ks 3 (the last H1 reads), A1 piece 3, then wait + B1
    LA_W4_MF(0, 0, 0, 0, 2) LA_W4_RW(3, 0) LA_W4_MF(0, 0, 1, 0, 2) LA_W4_RW(3, 1) LA_W4_MF(0, 0, 2, 0, 2) LA_W4_RW(3, 2)
    LA_W4_MF(0, 0, 3, 0, 2) LA_W4_RW(3, 3) LA_W4_MF(0, 1, 0, 0, 2) LA_W4_RA(1, 0, 0, 3) LA_W4_MF(0, 1, 1, 0, 2) LA_W4_RA(1, 1, 1, 3)
    LA_W4_MF(0, 1, 2, 0, 2) LA_W4_PA1(sa1, 3, bo1)
    if (seam) wait_vm_lgkm0<(16 + SEAM > 63 ? 63 : 16 + SEAM)>();
    else wait_vm_lgkm0<16>();
    LA_W4_SB LA_W4_MF(0, 1, 3, 0, 2)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    LA_W4_SB
    if (prelast && more) plan(next, m0n, n0n, soA0, soA1n, soW);         // A0W(t+2) is the next tile's first k-tile
    const T* saw = A + a_koff(e, kAW * BK_);
    const T* sww = Wt + kAW * BK_;
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) waddr[ks] ^= REG;  // (their next use is H1 of the next k-tile)
    // block 3 (ks 3): reads of H2 ks 0, A0 pieces 0, 1
    LA_W4_MF(0, 0, 0, 1, 3) LA_W4_RA(0, 0, 2, 0) LA_W4_MF(0, 0, 1, 1, 3) LA_W4_RA(0, 1, 3, 0) LA_W4_MF(0, 0, 2, 1, 3) LA_W4_PA0(saw, 0, bofs)
    LA_W4_MF(0, 0, 3, 1, 3) LA_W4_MF(0, 1, 0, 1, 3) LA_W4_PA0(saw, 1, bofs) LA_W4_MF(0, 1, 1, 1, 3) LA_W4_MF(0, 1, 2, 1, 3)
    LA_W4_MF(0, 1, 3, 1, 3)
    aaddr[0] ^= REG;                                  // next use: block 7, the first fragments of k-tile t+1
    // ================= H2 ==========================================================================================================
    // block 4 (ks 0): reads of ks 1, A0 pieces 2, 3, W piece 0
    LA_W4_MF(2, 0, 0, 0, 0) LA_W4_RA(1, 0, 2, 1) LA_W4_MF(2, 0, 1, 0, 0) LA_W4_RA(1, 1, 3, 1) LA_W4_MF(2, 0, 2, 0, 0) LA_W4_PA0(saw, 2, bofs)
    LA_W4_MF(2, 0, 3, 0, 0) LA_W4_MF(2, 1, 0, 0, 0) LA_W4_PA0(saw, 3, bofs) LA_W4_MF(2, 1, 1, 0, 0) LA_W4_MF(2, 1, 2, 0, 0)
    LA_W4_PW(sww, 0, bofs) LA_W4_MF(2, 1, 3, 0, 0)
    // block 5 (ks 1): reads of ks 2, W pieces 1, 2, 3
    LA_W4_MF(2, 0, 0, 1, 1) LA_W4_RA(0, 0, 2, 2) LA_W4_MF(2, 0, 1, 1, 1) LA_W4_RA(0, 1, 3, 2) LA_W4_MF(2, 0, 2, 1, 1) LA_W4_PW(sww, 1, bofs)
    LA_W4_MF(2, 0, 3, 1, 1) LA_W4_MF(2, 1, 0, 1, 1) LA_W4_PW(sww, 2, bofs) LA_W4_MF(2, 1, 1, 1, 1) LA_W4_MF(2, 1, 2, 1, 1)
    LA_W4_PW(sww, 3, bofs) LA_W4_MF(2, 1, 3, 1, 1)
    // block 6 (ks 2): reads of ks 3 (the last reads of this k-tile), W pieces 4, 5, 6, then wait + B2
    LA_W4_MF(2, 0, 0, 0, 2) LA_W4_RA(1, 0, 2, 3) LA_W4_MF(2, 0, 1, 0, 2) LA_W4_RA(1, 1, 3, 3) LA_W4_MF(2, 0, 2, 0, 2) LA_W4_PW(sww, 4, bofs)
    LA_W4_MF(2, 0, 3, 0, 2) LA_W4_MF(2, 1, 0, 0, 2) LA_W4_PW(sww, 5, bofs) LA_W4_MF(2, 1, 1, 0, 2) LA_W4_MF(2, 1, 2, 0, 2)
    LA_W4_PW(sww, 6, bofs)
    if (seam) wait_vm_lgkm0<(15 + SEAM > 63 ? 63 : 15 + SEAM)>();
    else wait_vm_lgkm0<15>();
    LA_W4_SB LA_W4_MF(2, 1, 3, 0, 2)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    LA_W4_SB
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) aaddr[ks] ^= REG;
    // block 7 (ks 3): the first fragments of k-tile t+1 (other buffer), W piece 7
    LA_W4_MF(2, 0, 0, 1, 3) LA_W4_RW(0, 0) LA_W4_MF(2, 0, 1, 1, 3) LA_W4_RW(0, 1) LA_W4_MF(2, 0, 2, 1, 3) LA_W4_RW(0, 2)
    LA_W4_MF(2, 0, 3, 1, 3) LA_W4_RW(0, 3) LA_W4_MF(2, 1, 0, 1, 3) LA_W4_RA(0, 0, 0, 0) LA_W4_MF(2, 1, 1, 1, 3) LA_W4_RA(0, 1, 1, 0)
    LA_W4_MF(2, 1, 2, 1, 3) LA_W4_PW(sww, 7, bofs) LA_W4_MF(2, 1, 3, 1, 3)
    waddr[0] ^= REG;
    bofs ^= REG;
    seam = false;
    ++kt;
    if (!last) continue;
    // ================= seam: epilogue of the finished tile =========================================================================
    asm volatile("s_nop 15" ::: "memory");             // the last MFMA's result -> first VALU reader (12 wait states; see MfmaA)
    // (the empty asm re-defines the accumulators in the AGPR class HERE: without it hipcc hoists the 128 AGPR -> VGPR copies the epilogue
    // needs into the k-tile loop, which then spills its DMA offsets - every reload is an s_waitcnt vmcnt(0) in front of a piece)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+a"(acc[0][i][j]));
    epilogue_wave<T, EPI>(slab, rtab, acc[0], m0 + wr * 128, n0 + wc * 128, n0, M, e, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+a"(acc[1][i][j]));
    epilogue_wave<T, EPI>(slab, rtab, acc[1], m0 + wr * 128, n0 + wc * 128 + 64, n0, M, e, lane);
    seam = (m0 + 256 <= M) && !(EPI == 1 && e.vt != nullptr && n0 >= e.vt_col0);
    if (!more) break;
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
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail's surplus requests must have landed before the LDS is released
}

template <typename T, int EPI>
void launch_t256w(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, int gm, hipStream_t st) {
  constexpr int LDS = 4 * 32768 + 4 * 2048 + 4 * 512;      // two k-tile buffers + 2 KiB slab per wave + row tables: 138 KiB
  static unsigned long long attr_mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(gemm_t256w_kernel<T, EPI>), LDS, attr_mask);
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (ncu <= 0) ncu = 256;
  }
  const int ntiles = ((M + 255) / 256) * (N / 256);
  const int grid = ntiles < ncu ? ntiles : ncu;
  hipLaunchKernelGGL((gemm_t256w_kernel<T, EPI>), dim3(grid), dim3(256), LDS, st, reinterpret_cast<const T*>(A), lda,
                     reinterpret_cast<const T*>(W), ldw, M, N, K, e, gm);
}

#define LA_W4_INST(T, EPI) \
  template void launch_t256w<T, EPI>(const void*, int, const void*, int, int, int, int, const LaGemmEpilogue&, int, hipStream_t);
LA_W4_INST(f16_t, 1) LA_W4_INST(f16_t, 2) LA_W4_INST(f16_t, 3)
LA_W4_INST(bf16_t, 1) LA_W4_INST(bf16_t, 2) LA_W4_INST(bf16_t, 3)

}  // namespace la
