// Fused image-side kernels of the TwoWayTransformer (models/transformer.py:255-329, common.py:57-148): the (groups, hw, D) fp32
// stream of the prompt encoder / mask decoder is read ONCE per attention instead of once per projection, score pass and norm.
//
//   la_twoway_t2i  tokens -> image attention: per 64-row tile  K = x Wk^T + PEK,  V = x Wv^T + bv  on the fast MFMA with operands split
//                  on the fly (x = x_hi + x_lo fp16 planes, weights pre-split: hi.hi + lo.hi + hi.lo, ~21 mantissa bits), then the
//                  tile's softmax partials (m, l, sum p V) for every (token, head) straight from the accumulator layout - K and V never
//                  leave the registers.  la_twoway_merge folds the partials of a group's tiles.
//   la_twoway_i2t  image -> tokens attention + out_proj + residual + LayerNorm: Q = x Wq^T + PEQ per tile, online softmax over the
//                  group's tokens per head, O -> [hi | lo] planes in LDS -> Y = O Wo^T + bo + x -> LN -> written back in place.
//                  One read and one write of the stream per layer.
// The positional encoding never enters the kernels: (x + pe) W^T + b = x W^T + (pe W^T + b), and PEK / PEQ = pe W^T + b is a constant
// [hw, DI] table per layer (the caller computes it once).
//
// Round 4 shape (rounds 1-3: one 128-row tile per CU, 4 waves of 512 registers).  A tile is 64 rows; its 4 waves form a 2 x 2 grid -
// wave (rt, ch) multiplies row tile rt (32 rows) against column half ch of every product - and, for the published decoder width
// D = 256, a workgroup needs 80 KiB of LDS and <= 256 registers per lane, so TWO tiles are in flight per CU: one tile's row burst /
// store burst overlaps the other's MFMA phases, and every SIMD holds two waves (a lone wave issues a VALU instruction every ~8 cycles,
// two waves every ~4: profiles/r03_notes.md 3).  The same code instantiated for D = 512 (the published SAM-1024 decoder geometry,
// parameters/validation/old/COCO_Fold0_sam.yaml:255-270; internal width 256, 8 heads of 32) takes 160 KiB and one tile per CU.
// Rows are requested in ONE burst and live in registers for the whole kernel (wave w owns rows 16 w .. 16 w + 15 for loading, staging
// and the LayerNorm epilogue; lane l the columns 4 l .. 4 l + 3 of every 256-column block: 1 KiB coalesced per row and block).
// Weights stream through LDS in "k16 slabs": for 16 consecutive K columns, [plane][rows][32 B] (LDS-DMA pieces of 32 rows), ring of three.
#include <cstdlib>
#include "la_common.h"
#include "../../include/la_hip.h"

namespace la {

constexpr int TW_ROWS = 64;       // stream rows per workgroup
constexpr int TW_APL = TW_ROWS * 64;        // one 32-column plane chunk of the tile: 64 rows x 64 B = 4 KiB
constexpr int TW_XPL = 2 * 4 * TW_APL;      // x planes in LDS: [hi | lo][4 chunks] = 128 columns at a time, 32 KiB
constexpr int TW_MAXT = 32;

template <int DI> struct TwCfg {
  static constexpr int D = 2 * DI;          // stream width
  static constexpr int HD = DI / 8;         // head width (8 heads)
  static constexpr int NV = D / 256;        // float4 per lane and row
  static constexpr int NJ = DI / 64;        // 32-column accumulator tiles per wave in the DI-wide products (column half ch)
  static constexpr int NY = D / 64;         // ... in the D-wide product Y = O Wo^T
  static constexpr int NCO = DI / 32;       // 32-column chunks of O
  static constexpr int YLD = D + 4;
};

// 64-byte rows (32 halfs): 16-byte chunk c of row r in slot c ^ ((r >> 2) & 3) - conflict-free ds_read_b128 fragments
__device__ __forceinline__ int tw_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }
// 32-byte rows of a k16 weight slab: 16-byte half c of row r in slot c ^ ((r >> 3) & 1) (16 consecutive rows = 512 B = all 64 banks twice;
// the flip moves rows 8-15 of every 16 onto the other half of the bank row)
__device__ __forceinline__ int tw_woff(int row, int c) { return row * 32 + ((c ^ ((row >> 3) & 1)) << 4); }

// hi / lo planes of 4 consecutive values -> two 8-byte LDS stores
__device__ __forceinline__ void split_store4(char* hi_plane, char* lo_plane, int off, float4 v) {
  const f16_t ha = (f16_t)v.x, hb = (f16_t)v.y, hc = (f16_t)v.z, hd = (f16_t)v.w;
  uint2 h, l;
  h.x = pack2<f16_t>(v.x, v.y);
  h.y = pack2<f16_t>(v.z, v.w);
  l.x = pack2<f16_t>(v.x - (float)ha, v.y - (float)hb);
  l.y = pack2<f16_t>(v.z - (float)hc, v.w - (float)hd);
  *reinterpret_cast<uint2*>(hi_plane + off) = h;
  *reinterpret_cast<uint2*>(lo_plane + off) = l;
}

// this wave's 16 rows of the tile: row rr, 256-column block v in xs[rr][v], lane -> columns 256 v + 4 lane .. + 3 (rows beyond hw read as zero)
template <int NV>
__device__ __forceinline__ void load_rows(const float* xg, int D, int row0, int hw, int lane, float4 (&xs)[16][NV]) {
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int row = row0 + rr;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      xs[rr][v] = *reinterpret_cast<const float4*>(xg + (size_t)min(row, hw - 1) * D + v * 256 + lane * 4);
      if (row >= hw) xs[rr][v] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// columns [128 sg, +128) of this wave's 16 rows -> A-operand plane chunks [4][64 rows x 64 B] (hi at pa, lo at pa + 4 chunks); the 32
// lanes that hold those columns do the work
template <int NV, int V>
__device__ __forceinline__ void stage_block(char* pa, int wave, int lane, int half, const float4 (&xs)[16][NV]) {
  if ((lane >> 5) == half) {
    const int li = lane & 31;
    char* hi = pa + (li >> 3) * TW_APL;
    char* lo = hi + 4 * TW_APL;
    const int sub = (li & 1) * 8, c16 = (li & 7) >> 1;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) split_store4(hi, lo, tw_off(wave * 16 + rr, c16) + sub, xs[rr][V]);
  }
}
// (the 256-column block is selected by a wave-uniform branch over compile-time indices: a run-time index into xs would move the whole
// row set into scratch memory - which is what the first D = 512 build did, 528 bytes per lane)
template <int NV>
__device__ __forceinline__ void stage_group(char* pa, int wave, int lane, int sg, const float4 (&xs)[16][NV]) {
  if (NV == 1 || (sg >> 1) == 0) stage_block<NV, 0>(pa, wave, lane, sg & 1, xs);
  else stage_block<NV, (NV > 1 ? 1 : 0)>(pa, wave, lane, sg & 1, xs);
}

// One k16 slab of NP weight planes with NR rows each: K columns [16 kk, +16) of every row, [plane][row][32 B] (tw_woff) at LDS byte
// address dst.  Pieces of 32 rows (1 KiB); the workgroup's 4 waves take pieces wave, wave + 4, ...  (NP * NR / 32 pieces, a multiple of 4)
template <int NP, int NR>
__device__ __forceinline__ void dma_slab(const f16_t* const (&planes)[NP], int ldw, int kk, unsigned dst, int wave, int lane) {
  constexpr int PIECES = NP * NR / 32;
#pragma unroll
  for (int i = 0; i < PIECES / 4; ++i) {
    const int piece = i * 4 + wave;
    const int plane = piece / (NR / 32), r = (piece % (NR / 32)) * 32 + (lane >> 1);
    const int c = (lane & 1) ^ ((r >> 3) & 1);
    dma16(planes[plane] + (size_t)r * ldw + kk * 16 + c * 8, dst + piece * 1024);
  }
}

// head sum over HD consecutive lanes of the 32-lane half
template <int HD> __device__ __forceinline__ float head_sum(float v) {
  v = row16_sum(v);
  if (HD == 32) v += __shfl_xor(v, 16, 64);
  return v;
}

// PE tables (pe Wk^T + bk, pe Wq^T + bq: fp32 [hw, DI], constant per layer and grid) are consumed in the ACCUMULATOR layout of the tile
// kernels - lane (fr, fh) of wave (rt, ch), tile j, register r <-> row 64 s + 32 rt + (r & 3) + 8 (r >> 2) + 4 fh, column ch DI/2 + 32 j + fr.
// Read row-major that is 16 NJ quarter-full requests per lane and tile (128 of the ~512 requests a 64-row tile issues); la_twoway_pe_layout
// writes the table once in the order the kernels read it, [tile s][wave][j][k = r / 4][lane] float4: 4 NJ full requests per lane.
template <int DI>
__global__ __launch_bounds__(256) void twoway_pe_layout_kernel(const float* __restrict__ src, int hw, float* __restrict__ dst) {
  constexpr int NJ = DI / 64;
  const int s = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rt = wave >> 1, ch = wave & 1, fr = lane & 31, fh = lane >> 5;
  for (int j = 0; j < NJ; ++j)
    for (int k = 0; k < 4; ++k) {
      float v[4];
      for (int c = 0; c < 4; ++c) {
        const int r = 4 * k + c;
        const int row = min(s * TW_ROWS + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh, hw - 1);
        v[c] = src[(size_t)row * DI + ch * (DI / 2) + j * 32 + fr];
      }
      reinterpret_cast<float4*>(dst)[((((size_t)s * 4 + wave) * NJ + j) * 4 + k) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// this lane's 16 NJ entries of a laid-out PE table
template <int NJ>
__device__ __forceinline__ void load_pe(const float* tab, int split, int wave, int lane, float (&p)[NJ][16]) {
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = reinterpret_cast<const float4*>(tab)[((((size_t)split * 4 + wave) * NJ + j) * 4 + k) * 64 + lane];
      p[j][4 * k + 0] = v.x;
      p[j][4 * k + 1] = v.y;
      p[j][4 * k + 2] = v.z;
      p[j][4 * k + 3] = v.w;
    }
}

struct TwT2iArgs {
  const float* img;      // [G * hw, D]
  const f16_t *wk_hi, *wk_lo, *wv_hi, *wv_lo;   // [DI, D] each
  const float* pek;      // pe Wk^T + bk in the layout of la_twoway_pe_layout
  const float* bv;       // [DI]
  const float* q;        // [G * nt, DI] projected queries (bias included)
  float* part;           // [G][S][2][nt][8][2 + HD]
  int G, hw, nt, S;
  float scale;
};

// K / V projections of a 64-row tile held in registers + the tile's softmax partials (the compute part of la_twoway_t2i; also the tail
// of the fused i2t + t2i kernel, which enters with the freshly normalised rows).  The caller has issued slabs 0 and 1 of the weight ring
// at smem + ring0; the x planes live at smem[0, 32K).  LDS: 32 KiB + ring of three k16 slabs (wk_hi, wk_lo, wv_hi, wv_lo x DI rows x 32 B).
template <int DI>
__device__ __forceinline__ void t2i_compute(const TwT2iArgs& a, char* smem, int ring0, int g, int split, float4 (&xs)[16][TwCfg<DI>::NV],
                                            int wave, int lane) {
  using C = TwCfg<DI>;
  constexpr int SLAB = 4 * DI * 32, NKK = C::D / 16;
  constexpr int PW = 4 * DI / 32 / 4;               // DMA pieces per wave and slab
  char* pa = smem;
  const int rt = wave >> 1, ch = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  const int row0 = split * TW_ROWS;
  const f16_t* const planes[4] = {a.wk_hi, a.wk_lo, a.wv_hi, a.wv_lo};
  const unsigned lds_w = lds_addr_of(smem + ring0);
  float pk[C::NJ][16];               // this lane's entries of the PEK table (accumulator layout): requested BEHIND the row burst (below)

  f32x16 kacc[C::NJ], vacc[C::NJ];
#pragma unroll
  for (int j = 0; j < C::NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) kacc[j][r] = vacc[j][r] = 0.f;

  for (int kk = 0; kk < NKK; ++kk) {
    if (kk + 1 < NKK) dma_wait<PW>();                // slab kk landed (slab kk + 1 may still travel)
    else dma_wait<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // ... for every wave, and every wave is done with step kk - 1
    asm volatile("" ::: "memory");
    if (kk + 2 < NKK) dma_slab<4, DI>(planes, C::D, kk + 2, lds_w + ((kk + 2) % 3) * SLAB, wave, lane);
    if ((kk & 7) == 0) {                             // next 128 columns of the rows -> planes (the first time: waits for the row burst)
      stage_group<C::NV>(pa, wave, lane, kk >> 3, xs);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kk == 0) load_pe<C::NJ>(a.pek, split, wave, lane, pk);     // only needed behind the loop: requested behind the row burst
    }
    const char* pw = smem + ring0 + (kk % 3) * SLAB;
    const char* pah = pa + ((kk >> 1) & 3) * TW_APL;
    const uint4 ah = *reinterpret_cast<const uint4*>(pah + tw_off(rt * 32 + fr, (kk & 1) * 2 + fh));
    const uint4 al = *reinterpret_cast<const uint4*>(pah + 4 * TW_APL + tw_off(rt * 32 + fr, (kk & 1) * 2 + fh));
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) {
      const int wr = ch * (DI / 2) + j * 32 + fr;
      uint4 wf[4];
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) wf[pl] = *reinterpret_cast<const uint4*>(pw + pl * DI * 32 + tw_woff(wr, fh));
      kacc[j] = Half16<f16_t>::mfma32(ah, wf[0], kacc[j]);
      kacc[j] = Half16<f16_t>::mfma32(al, wf[0], kacc[j]);
      kacc[j] = Half16<f16_t>::mfma32(ah, wf[1], kacc[j]);
      vacc[j] = Half16<f16_t>::mfma32(ah, wf[2], vacc[j]);
      vacc[j] = Half16<f16_t>::mfma32(al, wf[2], vacc[j]);
      vacc[j] = Half16<f16_t>::mfma32(ah, wf[3], vacc[j]);
    }
  }
  // ---- + PEK / bv; accumulator layout: lane -> column ch DI/2 + j 32 + fr, register r -> row (r & 3) + 8 (r >> 2) + 4 fh of row tile rt --
  bool rv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rv[r] = row0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh < a.hw;
#pragma unroll
  for (int j = 0; j < C::NJ; ++j) {
    const float bvv = a.bv[ch * (DI / 2) + j * 32 + fr];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      kacc[j][r] += pk[j][r];
      vacc[j][r] += bvv;
    }
  }
  // ---- softmax partials of every (token, head of this column half) over the wave's 32 rows ----------------------------------------------
  // a head = HD consecutive columns = HD consecutive lanes: q.k is a lane butterfly, the row set of a lane is (r, fh)
  float* pbase = a.part + (((size_t)g * a.S + split) * 2 + rt) * a.nt * 8 * (2 + C::HD);
  for (int t = 0; t < a.nt; ++t) {
    const float* qt = a.q + ((size_t)g * a.nt + t) * DI + ch * (DI / 2);
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) {
      const float qv = qt[j * 32 + fr] * a.scale;
      float s[16];
      float m = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = head_sum<C::HD>(kacc[j][r] * qv);
        s[r] = rv[r] ? v : -3.0e38f;
        m = fmaxf(m, s[r]);
      }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float l = 0.f, o = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = rv[r] ? __expf(s[r] - m) : 0.f;
        l += p;
        o += p * vacc[j][r];
      }
      l += __shfl_xor(l, 32, 64);
      o += __shfl_xor(o, 32, 64);
      if (fh == 0) {
        const int col = j * 32 + fr;
        const int head = ch * 4 + col / C::HD;
        float* dst = pbase + ((size_t)t * 8 + head) * (2 + C::HD);
        dst[2 + (col % C::HD)] = o;
        if ((col % C::HD) == 0) {
          dst[0] = m;
          dst[1] = l;
        }
      }
    }
  }
}

// LDS: x planes [0, 32K) (128 columns of K at a time) | ring of three k16 slabs behind them
template <int DI>
__global__ __launch_bounds__(256, (DI == 128 ? 2 : 1)) void twoway_t2i_kernel(TwT2iArgs a) {
  using C = TwCfg<DI>;
  constexpr int SLAB = 4 * DI * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.y, split = blockIdx.x;
  const float* xg = a.img + (size_t)g * a.hw * C::D;
  const f16_t* const planes[4] = {a.wk_hi, a.wk_lo, a.wv_hi, a.wv_lo};
  const unsigned lds_w = lds_addr_of(smem + TW_XPL);
  dma_slab<4, DI>(planes, C::D, 0, lds_w, wave, lane);
  dma_slab<4, DI>(planes, C::D, 1, lds_w + SLAB, wave, lane);
  float4 xs[16][C::NV];
  load_rows<C::NV>(xg, C::D, split * TW_ROWS + wave * 16, a.hw, lane, xs);
  t2i_compute<DI>(a, smem, TW_XPL, g, split, xs, wave, lane);
}

// fold the partials of a group's tiles: out[g, t, head * HD + c] = sum_p e^{m_p - M} o_p / sum_p e^{m_p - M} l_p   (8 HD threads)
template <int HD>
__global__ __launch_bounds__(8 * HD) void twoway_merge_kernel(const float* __restrict__ part, int nparts, int nt, float* __restrict__ out) {
  const int g = blockIdx.y, t = blockIdx.x;
  const int head = threadIdx.x / HD, c = threadIdx.x % HD;
  const float* p0 = part + ((size_t)g * nparts * nt + t) * 8 * (2 + HD) + head * (2 + HD);
  const size_t pstride = (size_t)nt * 8 * (2 + HD);
  float M = -3.0e38f;
  for (int p = 0; p < nparts; ++p) M = fmaxf(M, p0[p * pstride]);
  float l = 0.f, o = 0.f;
  for (int p = 0; p < nparts; ++p) {
    const float f = __expf(p0[p * pstride] - M);
    l += f * p0[p * pstride + 1];
    o += f * p0[p * pstride + 2 + c];
  }
  out[((size_t)g * nt + t) * (8 * HD) + head * HD + c] = o / l;
}

// =================================================================================================================================
// image -> tokens attention + out_proj + residual + LayerNorm, in place on the stream.
//   phase 1  Q = x Wq^T (+ PEQ)          D / 16 k16 slabs of Wq, two per barrier (ring of three pairs), x planes staged 128 columns of K at a time
//   phase 2  per 32-column tile: online softmax over the group's tokens (their k / v rows come straight from global memory: a few KiB,
//            L1-resident), lane butterfly for q.k; O = softmax . v goes to LDS as [hi | lo] planes in the MFMA A-operand image
//   phase 3  Y = O Wo^T                  DI / 16 k16 slabs of Wo; D / 64 accumulator tiles per wave
//   phase 4  Y + bo -> LDS (row-major fp32), then one WAVE per row: + x (still in registers), LayerNorm, coalesced store
// LDS: x planes [0, 32K), later the O planes [0, DI / 32 x 8K) | slab ring behind max(x planes, O planes); the Y tile of phase 4 reuses
// everything.  D = 256: 80 KiB (two workgroups per CU); D = 512: 160 KiB.  nt <= 32.
// =================================================================================================================================
struct TwI2tArgs {
  float* img;            // [G * hw, D] in / out
  const f16_t *wq_hi, *wq_lo;      // [DI, D]
  const float* peq;      // pe Wq^T + bq in the layout of la_twoway_pe_layout
  const float *k, *v;    // [G * nt, DI] projected token keys / values (bias included)
  const f16_t *wo_hi, *wo_lo;      // [D, DI]
  const float *bo, *gamma, *beta;  // [D]
  float eps, scale;
  int G, hw, nt;
};

#ifdef LA_DEBUG
__device__ unsigned long long g_tw_stamps[4 * 16];      // [wave][i]: s_memtime of workgroup (0, 0) at the phase boundaries of twoway_i2t_kernel
#define TW_STAMP(i)                                                                                   \
  do {                                                                                                \
    if (blockIdx.x == 0 && blockIdx.y == 0) {                                                         \
      unsigned long long t_;                                                                          \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
      if ((threadIdx.x & 63) == 0) g_tw_stamps[(threadIdx.x >> 6) * 16 + (i)] = t_;                   \
    }                                                                                                 \
  } while (0)
#else
#define TW_STAMP(i) do {} while (0)
#endif

template <int DI>
__global__ __launch_bounds__(256, (DI == 128 ? 2 : 1)) void twoway_i2t_kernel(TwI2tArgs a) {
  using C = TwCfg<DI>;
  constexpr int D = C::D;
  constexpr int SLAB1 = 2 * DI * 32, SLAB3 = 2 * D * 32, NKK1 = D / 16, NKK3 = DI / 16;
  constexpr int OPL = 2 * C::NCO * TW_APL;                          // O planes: 32 KiB (DI 128) / 64 KiB (DI 256)
  constexpr int RING0 = OPL > TW_XPL ? OPL : TW_XPL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* pa = smem;                                   // x planes (phase 1), O planes (phases 2-3)
  float* yt = reinterpret_cast<float*>(smem);        // phase 4: [64][YLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = wave >> 1, ch = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  const int g = blockIdx.y, row0 = blockIdx.x * TW_ROWS;
  float* xg = a.img + (size_t)g * a.hw * D;

  const f16_t* const wq[2] = {a.wq_hi, a.wq_lo};
  const f16_t* const wo[2] = {a.wo_hi, a.wo_lo};
  const unsigned lds_w = lds_addr_of(smem + RING0);
  constexpr int PW1 = 2 * DI / 32 / 4, PW3 = 2 * D / 32 / 4;        // DMA pieces per wave and slab
  TW_STAMP(0);
#pragma unroll
  for (int k0 = 0; k0 < 4; ++k0) dma_slab<2, DI>(wq, D, k0, lds_w + k0 * SLAB1, wave, lane);
  float4 xs[16][C::NV];
  load_rows<C::NV>(xg, D, row0 + wave * 16, a.hw, lane, xs);
  float pq[C::NJ][16];               // this lane's entries of the PEQ table (accumulator layout): requested BEHIND the row burst (below)
  // ---- phase 1: k32 steps (two k16 slabs per barrier), ring of three slab pairs ---------------------------------------------------------
  f32x16 qacc[C::NJ];
#pragma unroll
  for (int j = 0; j < C::NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) qacc[j][r] = 0.f;
  for (int kp = 0; kp < NKK1 / 2; ++kp) {
    if (kp + 1 < NKK1 / 2) dma_wait<2 * PW1>();      // pair kp landed (pair kp + 1 may still travel)
    else dma_wait<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kp + 2 < NKK1 / 2) {
      dma_slab<2, DI>(wq, D, 2 * kp + 4, lds_w + ((2 * kp + 4) % 6) * SLAB1, wave, lane);
      dma_slab<2, DI>(wq, D, 2 * kp + 5, lds_w + ((2 * kp + 5) % 6) * SLAB1, wave, lane);
    }
    if ((kp & 3) == 0) {
      stage_group<C::NV>(pa, wave, lane, kp >> 2, xs);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kp == 0) {
        TW_STAMP(1);                                 // row burst landed, first 128 columns staged
        load_pe<C::NJ>(a.peq, blockIdx.x, wave, lane, pq);     // only needed in phase 2: requested behind the row burst
      }
    }
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int kk = 2 * kp + h2;
      const char* pw = smem + RING0 + (kk % 6) * SLAB1;
      const char* pah = pa + (kp & 3) * TW_APL;
      const uint4 ah = *reinterpret_cast<const uint4*>(pah + tw_off(rt * 32 + fr, h2 * 2 + fh));
      const uint4 al = *reinterpret_cast<const uint4*>(pah + 4 * TW_APL + tw_off(rt * 32 + fr, h2 * 2 + fh));
#pragma unroll
      for (int j = 0; j < C::NJ; ++j) {
        const int wr = ch * (DI / 2) + j * 32 + fr;
        const uint4 wh = *reinterpret_cast<const uint4*>(pw + tw_woff(wr, fh));
        const uint4 wl = *reinterpret_cast<const uint4*>(pw + DI * 32 + tw_woff(wr, fh));
        qacc[j] = Half16<f16_t>::mfma32(ah, wh, qacc[j]);
        qacc[j] = Half16<f16_t>::mfma32(al, wh, qacc[j]);
        qacc[j] = Half16<f16_t>::mfma32(ah, wl, qacc[j]);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // every wave is done with the x planes and the phase-1 ring
  asm volatile("" ::: "memory");
  TW_STAMP(2);                                       // phase 1 done
  // phase 3's first two weight slabs travel during the attention phase
  dma_slab<2, D>(wo, DI, 0, lds_w, wave, lane);
  dma_slab<2, D>(wo, DI, 1, lds_w + SLAB3, wave, lane);
  // ---- phase 2: attention over the tokens, one 32-column tile at a time ----------------------------------------------------------
  const float* tkg = a.k + (size_t)g * a.nt * DI + ch * (DI / 2);
  const float* tvg = a.v + (size_t)g * a.nt * DI + ch * (DI / 2);
#pragma unroll
  for (int j = 0; j < C::NJ; ++j) {
    float m[16], l[16], o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      qacc[j][r] = (qacc[j][r] + pq[j][r]) * a.scale;
      m[r] = -3.0e38f;
      l[r] = o[r] = 0.f;
    }
    for (int t = 0; t < a.nt; ++t) {
      const float kv = tkg[t * DI + j * 32 + fr], vv = tvg[t * DI + j * 32 + fr];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s = head_sum<C::HD>(qacc[j][r] * kv);
        const float mn = fmaxf(m[r], s);
        const float corr = __expf(m[r] - mn), p = __expf(s - mn);
        l[r] = l[r] * corr + p;
        o[r] = o[r] * corr + p * vv;
        m[r] = mn;
      }
    }
    // O -> [hi | lo] planes in the A-operand image of O's 32-column chunk oc: row = rt 32 + (r & 3) + 8 (r >> 2) + 4 fh, column fr.  Lane
    // pairs (fr, fr ^ 1) hold neighbouring columns: they trade halves over DPP, the even lane stores the hi dword and the odd lane the lo
    // dword of the pair (one 32-bit LDS store per lane instead of two 16-bit ones)
    const int oc = ch * C::NJ + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float ov = o[r] * __builtin_amdgcn_rcpf(l[r]);
      const float hv = (float)(f16_t)ov, lv = ov - hv;
      const float hn = dpp_mov<0xB1>(hv), ln = dpp_mov<0xB1>(lv);          // the neighbour's values
      const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      const int off = oc * TW_APL + tw_off(row, fr >> 3) + (fr & 6) * 2;
      const bool odd = fr & 1;
      const uint32_t w = odd ? pack2<f16_t>(ln, lv) : pack2<f16_t>(hv, hn);
      *reinterpret_cast<uint32_t*>(pa + (odd ? C::NCO * TW_APL : 0) + off) = w;
    }
  }
  TW_STAMP(3);                                       // phase 2 done
  // ---- phase 3: Y = O Wo^T, k16 slabs of Wo in the ring ------------------------------------------------------------------------------
  f32x16 yacc[C::NY];
#pragma unroll
  for (int j = 0; j < C::NY; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) yacc[j][r] = 0.f;
  for (int kk = 0; kk < NKK3; ++kk) {
    if (kk + 1 < NKK3) dma_wait<PW3>();
    else dma_wait<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // (kk == 0: also orders the O planes of every wave before the first read)
    asm volatile("" ::: "memory");
    if (kk + 2 < NKK3) dma_slab<2, D>(wo, DI, kk + 2, lds_w + ((kk + 2) % 3) * SLAB3, wave, lane);
    const char* pw = smem + RING0 + (kk % 3) * SLAB3;
    const char* pah = pa + (kk >> 1) * TW_APL;
    const uint4 ah = *reinterpret_cast<const uint4*>(pah + tw_off(rt * 32 + fr, (kk & 1) * 2 + fh));
    const uint4 al = *reinterpret_cast<const uint4*>(pah + C::NCO * TW_APL + tw_off(rt * 32 + fr, (kk & 1) * 2 + fh));
#pragma unroll
    for (int j = 0; j < C::NY; ++j) {
      const int wr = ch * (D / 2) + j * 32 + fr;
      const uint4 wh = *reinterpret_cast<const uint4*>(pw + tw_woff(wr, fh));
      const uint4 wl = *reinterpret_cast<const uint4*>(pw + D * 32 + tw_woff(wr, fh));
      yacc[j] = Half16<f16_t>::mfma32(ah, wh, yacc[j]);
      yacc[j] = Half16<f16_t>::mfma32(al, wh, yacc[j]);
      yacc[j] = Half16<f16_t>::mfma32(ah, wl, yacc[j]);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // every wave is done with the O planes and the ring: the Y tile may overwrite them
  asm volatile("" ::: "memory");
  TW_STAMP(4);                                       // phase 3 done
  // ---- phase 4: + bias -> LDS row-major, then a wave per row: residual (registers), LayerNorm, store -----------------------------------
#pragma unroll
  for (int j = 0; j < C::NY; ++j) {
    const int col = ch * (D / 2) + j * 32 + fr;
    const float bov = a.bo[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) yt[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * C::YLD + col] = yacc[j][r] + bov;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // a row's columns come from two waves (ch 0 / 1); its reader is a third
  asm volatile("" ::: "memory");
  float4 gm[C::NV], bt[C::NV];
#pragma unroll
  for (int v = 0; v < C::NV; ++v) {
    gm[v] = reinterpret_cast<const float4*>(a.gamma)[v * 64 + lane];
    bt[v] = reinterpret_cast<const float4*>(a.beta)[v * 64 + lane];
  }
#pragma unroll
  for (int rr = 0; rr < 16; ++rr)
#pragma unroll
    for (int v = 0; v < C::NV; ++v) {
      const float4 y = *reinterpret_cast<const float4*>(&yt[(wave * 16 + rr) * C::YLD + v * 256 + lane * 4]);
      xs[rr][v].x += y.x; xs[rr][v].y += y.y; xs[rr][v].z += y.z; xs[rr][v].w += y.w;
    }
  float mu[16], rs[16];
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < C::NV; ++v) sum += (xs[rr][v].x + xs[rr][v].y) + (xs[rr][v].z + xs[rr][v].w);
    mu[rr] = wave_sum_dpp(sum) * (1.0f / D);
  }
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < C::NV; ++v) {
      xs[rr][v].x -= mu[rr]; xs[rr][v].y -= mu[rr]; xs[rr][v].z -= mu[rr]; xs[rr][v].w -= mu[rr];
      sq += (xs[rr][v].x * xs[rr][v].x + xs[rr][v].y * xs[rr][v].y) + (xs[rr][v].z * xs[rr][v].z + xs[rr][v].w * xs[rr][v].w);
    }
    rs[rr] = wave_sum_dpp(sq);
  }
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int row = row0 + wave * 16 + rr;
    const float rstd = 1.0f / sqrtf(rs[rr] * (1.0f / D) + a.eps);
    if (row < a.hw) {
#pragma unroll
      for (int v = 0; v < C::NV; ++v)
        *reinterpret_cast<float4*>(xg + (size_t)row * D + v * 256 + lane * 4) =
            make_float4(xs[rr][v].x * rstd * gm[v].x + bt[v].x, xs[rr][v].y * rstd * gm[v].y + bt[v].y, xs[rr][v].z * rstd * gm[v].z + bt[v].z,
                        xs[rr][v].w * rstd * gm[v].w + bt[v].w);
    }
  }
  TW_STAMP(5);                                       // stores issued
}

template <int DI> static void launch_t2i(const TwT2iArgs& a, float* out, hipStream_t st) {
  constexpr int LDS = TW_XPL + 3 * 4 * DI * 32;
  static unsigned long long attr_mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(twoway_t2i_kernel<DI>), LDS, attr_mask);
  hipLaunchKernelGGL(twoway_t2i_kernel<DI>, dim3(a.S, a.G), dim3(256), LDS, st, a);
  hipLaunchKernelGGL(twoway_merge_kernel<DI / 8>, dim3(a.nt, a.G), dim3(DI), 0, st, a.part, a.S * 2, a.nt, out);
}

template <int DI> static void launch_i2t(const TwI2tArgs& a, hipStream_t st) {
  using C = TwCfg<DI>;
  constexpr int OPL = 2 * C::NCO * TW_APL, RING0 = OPL > TW_XPL ? OPL : TW_XPL;
  constexpr int RING = 3 * 2 * C::D * 32, YT = TW_ROWS * C::YLD * 4;
  constexpr int LDS = (RING0 + RING) > YT ? (RING0 + RING) : YT;
  static_assert(LDS <= 160 * 1024, "two-way tile does not fit the LDS");
  static unsigned long long attr_mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(twoway_i2t_kernel<DI>), LDS, attr_mask);
  hipLaunchKernelGGL(twoway_i2t_kernel<DI>, dim3((a.hw + TW_ROWS - 1) / TW_ROWS, a.G), dim3(256), LDS, st, a);
}

}  // namespace la

extern "C" int la_twoway_pe_layout(const float* table, int hw, int DI, float* out, void* stream) {
  LA_CHECK_ARG(table && out && hw > 0 && (DI == 128 || DI == 256), "la_twoway_pe_layout: table [hw, DI] with DI = 128 / 256 (got hw=%d DI=%d)", hw, DI);
  const int S = (hw + la::TW_ROWS - 1) / la::TW_ROWS;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (DI == 128) hipLaunchKernelGGL(la::twoway_pe_layout_kernel<128>, dim3(S), dim3(256), 0, st, table, hw, out);
  else hipLaunchKernelGGL(la::twoway_pe_layout_kernel<256>, dim3(S), dim3(256), 0, st, table, hw, out);
  LA_CHECK_LAUNCH("la_twoway_pe_layout");
  return 0;
}

#ifdef LA_DEBUG
extern "C" int la_dbg_twoway_stamps(unsigned long long* host_out) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(la::g_tw_stamps), sizeof(unsigned long long) * 64);
}
#endif

extern "C" int la_twoway_t2i(const float* img, const void* wk_hi, const void* wk_lo, const void* wv_hi, const void* wv_lo, const float* pek,
                             const float* bv, const float* q, int G, int hw, int nt, int D, int heads, float* part, float* out, void* stream) {
  LA_CHECK_ARG(img && wk_hi && wk_lo && wv_hi && wv_lo && pek && bv && q && part && out, "la_twoway_t2i: null pointer");
  LA_CHECK_ARG((D == 256 || D == 512) && heads == 8, "la_twoway_t2i: built for D = 256 / 512 with 8 heads (got D=%d heads=%d)", D, heads);
  LA_CHECK_ARG(G > 0 && hw > 0 && nt > 0, "la_twoway_t2i: bad shape");
  const int S = (hw + la::TW_ROWS - 1) / la::TW_ROWS;
  la::TwT2iArgs a{img, (const la::f16_t*)wk_hi, (const la::f16_t*)wk_lo, (const la::f16_t*)wv_hi, (const la::f16_t*)wv_lo, pek, bv, q,
                  part, G, hw, nt, S, 1.0f / sqrtf((float)(D / 16))};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (D == 256) la::launch_t2i<128>(a, out, st);
  else la::launch_t2i<256>(a, out, st);
  LA_CHECK_LAUNCH("la_twoway_t2i");
  return 0;
}

extern "C" int la_twoway_i2t(float* img, const void* wq_hi, const void* wq_lo, const float* peq, const float* k, const float* v,
                             const void* wo_hi, const void* wo_lo, const float* bo, const float* gamma, const float* beta, float eps, int G,
                             int hw, int nt, int D, int heads, void* stream) {
  LA_CHECK_ARG(img && wq_hi && wq_lo && peq && k && v && wo_hi && wo_lo && bo && gamma && beta, "la_twoway_i2t: null pointer");
  LA_CHECK_ARG((D == 256 || D == 512) && heads == 8, "la_twoway_i2t: built for D = 256 / 512 with 8 heads (got D=%d heads=%d)", D, heads);
  LA_CHECK_ARG(G > 0 && hw > 0 && nt > 0 && nt <= la::TW_MAXT, "la_twoway_i2t: nt=%d out of range (1..%d)", nt, la::TW_MAXT);
  la::TwI2tArgs a{img, (const la::f16_t*)wq_hi, (const la::f16_t*)wq_lo, peq, k, v, (const la::f16_t*)wo_hi, (const la::f16_t*)wo_lo, bo, gamma,
                  beta, eps, 1.0f / sqrtf((float)(D / 16)), G, hw, nt};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (D == 256) la::launch_i2t<128>(a, st);
  else la::launch_i2t<256>(a, st);
  LA_CHECK_LAUNCH("la_twoway_i2t");
  return 0;
}
