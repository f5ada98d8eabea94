// Fused image-side kernels of the TwoWayTransformer (models/transformer.py:255-329, common.py:57-148): the (groups, hw, D) fp32
// stream of the prompt encoder / mask decoder is read ONCE per attention instead of once per projection, score pass and norm.
//
//   la_twoway_t2i  tokens -> image attention: per 128-row tile  K = x Wk^T + PEK,  V = x Wv^T + bv  on the fast MFMA with operands split
//                  on the fly (x = x_hi + x_lo fp16 planes, weights pre-split: hi.hi + lo.hi + hi.lo, ~21 mantissa bits), then the
//                  tile's softmax partials (m, l, sum p V) for every (token, head) straight from the accumulator layout - K and V never
//                  leave the registers.  la_twoway_merge folds the partials of a group's tiles.
//   la_twoway_i2t  image -> tokens attention + out_proj + residual + LayerNorm: Q = x Wq^T + PEQ per tile, online softmax over the
//                  group's tokens per head, O -> [hi | lo] planes in LDS -> Y = O Wo^T + bo + x -> LN -> written back in place.
//                  One read and one write of the stream per layer.
// The positional encoding never enters the kernels: (x + pe) W^T + b = x W^T + (pe W^T + b), and PEK / PEQ = pe W^T + b is a constant
// [hw, 128] table per layer (the caller computes it once).
//
// A tile's 128 rows are requested in ONE burst and live in registers for the whole kernel (wave w owns rows 32 w .. 32 w + 31, lane l
// the columns 4 l .. 4 l + 3: 1 KiB coalesced per row): one memory latency per tile instead of one per k-chunk, and the residual of
// the LayerNorm epilogue needs no second read.  One workgroup of 4 waves per CU (512 registers per lane).
// D = 256 (internal 128, 8 heads of 16): the geometry of every published model (other widths take the unfused path).
#include <cstdlib>
#include "la_common.h"
#include "../../include/la_hip.h"

namespace la {

constexpr int TW_ROWS = 128;      // stream rows per workgroup
constexpr int TW_BK = 32;         // columns per weight chunk
constexpr int TW_DI = 128;        // internal width of the cross attentions (D / 2)
constexpr int TW_HD = 16;         // head width
constexpr int TW_APLANE = TW_ROWS * 64;     // one 32-column plane chunk: 128 rows x 64 B = 8 KiB
constexpr int TW_MAXT = 32;
constexpr int TW_YLD = 256 + 4;

// 64-byte rows (32 halfs): 16-byte chunk c of row r in slot c ^ ((r >> 2) & 3) - conflict-free ds_read_b128 fragments
__device__ __forceinline__ int tw_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

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

// this wave's 32 rows of the tile: row rr in xs[rr], lane -> columns 4 lane .. 4 lane + 3 (rows beyond hw read as zero)
__device__ __forceinline__ void load_rows(const float* xg, int D, int row0, int hw, int lane, float4 (&xs)[32]) {
#pragma unroll
  for (int rr = 0; rr < 32; ++rr) {
    const int row = row0 + rr;
    xs[rr] = *reinterpret_cast<const float4*>(xg + (size_t)min(row, hw - 1) * D + lane * 4);
    if (row >= hw) xs[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// columns [128 half, +128) of this wave's rows -> A-operand plane chunks [4][128 rows x 64 B] (hi at pa, lo at pa + 4 planes); the 32
// lanes that hold those columns do the work
__device__ __forceinline__ void stage_half(char* pa, int wave, int lane, int half, const float4 (&xs)[32]) {
  if ((lane >> 5) == half) {
    const int li = lane & 31;
    char* hi = pa + (li >> 3) * TW_APLANE;
    char* lo = hi + 4 * TW_APLANE;
    const int sub = (li & 1) * 8, c16 = (li & 7) >> 1;
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) split_store4(hi, lo, tw_off(wave * 32 + rr, c16) + sub, xs[rr]);
  }
}

struct TwT2iArgs {
  const float* img;      // [G * hw, D]
  const f16_t *wk_hi, *wk_lo, *wv_hi, *wv_lo;   // [DI, D] each
  const float* pek;      // [hw, DI]  pe Wk^T + bk
  const float* bv;       // [DI]
  const float* q;        // [G * nt, DI] projected queries (bias included)
  float* part;           // [G][S][4][nt][8][2 + HD]
  int G, hw, nt, D, S;
  float scale;
};

// LDS: x planes [0, 64K) (half of K at a time) | weight ring: three chunks of 32 KiB (wk_hi, wk_lo, wv_hi, wv_lo x 128 rows x 64 B) at [64K, 160K)
__global__ __launch_bounds__(256, 1) void twoway_t2i_kernel(TwT2iArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* pa = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int g = blockIdx.y, split = blockIdx.x;
  const int row0 = split * TW_ROWS;
  const float* xg = a.img + (size_t)g * a.hw * a.D;

  // weight chunk = 4 planes x 128 rows = 32 pieces of 16 rows, 8 per wave; ring of three chunks, all three issued BEFORE the row burst so
  // that they are on chip by the time the rows are (vmcnt retires in order)
  unsigned wsoff[8];
  const f16_t* wbase[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int piece = wave * 8 + i;
    const int r = (piece & 7) * 16 + (lane >> 2);
    wsoff[i] = (unsigned)(((size_t)r * a.D + (((lane & 3) ^ ((r >> 2) & 3)) << 3)) * 2);
    const int plane = piece >> 3;
    wbase[i] = plane == 0 ? a.wk_hi : plane == 1 ? a.wk_lo : plane == 2 ? a.wv_hi : a.wv_lo;
  }
  const unsigned lds_w = lds_addr_of(smem + 65536);
  auto dma_w = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dma16s(wbase[i] + kc * TW_BK, wsoff[i], lds_w + (kc % 3) * 32768 + (wave * 8 + i) * 1024);
  };
  dma_w(0);
  dma_w(1);
  dma_w(2);
  float4 xs[32];
  load_rows(xg, a.D, row0 + wave * 32, a.hw, lane, xs);
  float pk[4][16];                  // this lane's 64 entries of the PEK table (accumulator layout), requested with the rows
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      pk[j][r] = a.pek[(size_t)min(row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh, a.hw - 1) * TW_DI + j * 32 + fr];

  f32x16 kacc[4], vacc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) kacc[j][r] = vacc[j][r] = 0.f;

  const int nkc = a.D / TW_BK;       // 8
  for (int kc = 0; kc < nkc; ++kc) {
    if ((kc & 3) == 0) stage_half(pa, wave, lane, kc >> 2, xs);      // (waits for the row burst the first time, hence for chunks 0-2)
    if (kc >= 1 && kc + 2 < nkc) dma_w(kc + 2);                      // into the slot chunk kc - 1 left at the last barrier
    {                                                                // chunk kc landed; the (up to two) younger ones stay in flight
      const int younger = min(2, nkc - 1 - kc);
      if (younger == 2) dma_wait<16>();
      else if (younger == 1) dma_wait<8>();
      else dma_wait<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* pw = smem + 65536 + (kc % 3) * 32768;
    const char* pah = pa + (kc & 3) * TW_APLANE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 ah = *reinterpret_cast<const uint4*>(pah + tw_off(wave * 32 + fr, ks * 2 + fh));
      const uint4 al = *reinterpret_cast<const uint4*>(pah + 4 * TW_APLANE + tw_off(wave * 32 + fr, ks * 2 + fh));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4 wf[4];
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) wf[pl] = *reinterpret_cast<const uint4*>(pw + pl * TW_APLANE + tw_off(j * 32 + fr, ks * 2 + fh));
        kacc[j] = Half16<f16_t>::mfma32(ah, wf[0], kacc[j]);
        kacc[j] = Half16<f16_t>::mfma32(al, wf[0], kacc[j]);
        kacc[j] = Half16<f16_t>::mfma32(ah, wf[1], kacc[j]);
        vacc[j] = Half16<f16_t>::mfma32(ah, wf[2], vacc[j]);
        vacc[j] = Half16<f16_t>::mfma32(al, wf[2], vacc[j]);
        vacc[j] = Half16<f16_t>::mfma32(ah, wf[3], vacc[j]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();    // weight slot kc % 3 (and, every 4th chunk, the x planes) may be restaged
    asm volatile("" ::: "memory");
  }
  // ---- + PEK / bv; accumulator layout: lane -> column j*32 + fr, register r -> row (r & 3) + 8 (r >> 2) + 4 fh of the wave's 32 rows -------
  bool rv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rv[r] = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh < a.hw;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float bvv = a.bv[j * 32 + fr];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      kacc[j][r] += pk[j][r];
      vacc[j][r] += bvv;
    }
  }
  // ---- softmax partials of every (token, head) over this wave's 32 rows -------------------------------------------------------------------
  // a head = 16 consecutive columns = one 16-lane row of the wave: q.k is a 16-lane butterfly sum, the row set of a lane is (r, fh)
  float* pbase = a.part + (((size_t)g * a.S + split) * 4 + wave) * a.nt * 8 * (2 + TW_HD);
  for (int t = 0; t < a.nt; ++t) {
    const float* qt = a.q + ((size_t)g * a.nt + t) * TW_DI;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float qv = qt[j * 32 + fr] * a.scale;
      float s[16];
      float m = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = row16_sum(kacc[j][r] * qv);
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
        const int head = j * 2 + (fr >> 4);
        float* dst = pbase + ((size_t)t * 8 + head) * (2 + TW_HD);
        dst[2 + (fr & 15)] = o;
        if ((fr & 15) == 0) {
          dst[0] = m;
          dst[1] = l;
        }
      }
    }
  }
}

// fold the partials of a group's tiles: out[g, t, head * HD + c] = sum_p e^{m_p - M} o_p / sum_p e^{m_p - M} l_p
__global__ __launch_bounds__(128) void twoway_merge_kernel(const float* __restrict__ part, int nparts, int nt, float* __restrict__ out) {
  const int g = blockIdx.y, t = blockIdx.x;
  const int head = threadIdx.x >> 4, c = threadIdx.x & 15;
  const float* p0 = part + ((size_t)g * nparts * nt + t) * 8 * (2 + TW_HD) + head * (2 + TW_HD);
  const size_t pstride = (size_t)nt * 8 * (2 + TW_HD);
  float M = -3.0e38f;
  for (int p = 0; p < nparts; ++p) M = fmaxf(M, p0[p * pstride]);
  float l = 0.f, o = 0.f;
  for (int p = 0; p < nparts; ++p) {
    const float f = __expf(p0[p * pstride] - M);
    l += f * p0[p * pstride + 1];
    o += f * p0[p * pstride + 2 + c];
  }
  out[((size_t)g * nt + t) * TW_DI + head * TW_HD + c] = o / l;
}

// =================================================================================================================================
// image -> tokens attention + out_proj + residual + LayerNorm, in place on the stream.
//   phase 1  Q = x Wq^T (+ PEQ)          8 weight chunks of 32 columns (LDS-DMA ring of two), x planes staged half of K at a time
//   phase 2  per 32-column tile (= 2 heads): online softmax over the group's tokens (token k / v in LDS as fp32), 16-lane butterfly for
//            q.k; O = softmax . v goes to LDS as [hi | lo] planes in the MFMA A-operand image (each wave reads back only its own rows)
//   phase 3  Y = O Wo^T                  4 weight chunks of 32 columns of O; 8 accumulator tiles per wave
//   phase 4  Y + bo -> LDS (row-major fp32), then one WAVE per row: + x (still in registers), LayerNorm, 1 KiB coalesced store
// LDS (160 KiB): x planes, later O planes [0, 64K) | weight ring [64K, 128K) (2 x 16 KiB in phase 1, 2 x 32 KiB in phase 3) | token
// k / v fp32 [128K, 160K); the Y tile of phase 4 (130 KiB) reuses everything.  nt <= 32.
// =================================================================================================================================
struct TwI2tArgs {
  float* img;            // [G * hw, D] in / out
  const f16_t *wq_hi, *wq_lo;      // [DI, D]
  const float* peq;      // [hw, DI]  pe Wq^T + bq
  const float *k, *v;    // [G * nt, DI] projected token keys / values (bias included)
  const f16_t *wo_hi, *wo_lo;      // [D, DI]
  const float *bo, *gamma, *beta;  // [D]
  float eps, scale;
  int G, hw, nt, D;
};

__global__ __launch_bounds__(256, 1) void twoway_i2t_kernel(TwI2tArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* pa = smem;                                   // x planes (phase 1), O planes (phases 2-3): [hi | lo][4 chunks][128 rows x 64 B]
  float* tk = reinterpret_cast<float*>(smem + 131072);            // [nt][128]
  float* tv = tk + TW_MAXT * TW_DI;                               // [nt][128]
  float* yt = reinterpret_cast<float*>(smem);        // phase 4: [128][TW_YLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int g = blockIdx.y, row0 = blockIdx.x * TW_ROWS;
  float* xg = a.img + (size_t)g * a.hw * a.D;

  // phase-1 weight ring: four chunks of 16 KiB (wq_hi, wq_lo x 128 rows x 64 B); the first four leave BEFORE the row burst, so they are
  // on chip by the time the rows are (vmcnt retires in order)
  unsigned wsoff[4];
  const f16_t* wbase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;                  // 2 planes x 8 groups of 16 rows
    const int r = (piece & 7) * 16 + (lane >> 2);
    wsoff[i] = (unsigned)(((size_t)r * a.D + (((lane & 3) ^ ((r >> 2) & 3)) << 3)) * 2);
    wbase[i] = (piece >> 3) ? a.wq_lo : a.wq_hi;
  }
  const unsigned lds_w = lds_addr_of(smem + 65536);
  auto dma_wq = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16s(wbase[i] + kc * TW_BK, wsoff[i], lds_w + (kc & 3) * 16384 + (wave * 4 + i) * 1024);
  };
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) dma_wq(kc);
  float4 xs[32];
  load_rows(xg, a.D, row0 + wave * 32, a.hw, lane, xs);
  // this lane's 64 entries of the PEQ table (accumulator layout), requested together with the rows: one latency, not 64
  float pq[4][16];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = min(row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh, a.hw - 1);
      pq[j][r] = a.peq[(size_t)row * TW_DI + j * 32 + fr];
    }
  for (int i = tid; i < a.nt * TW_DI; i += 256) {
    tk[i] = a.k[(size_t)g * a.nt * TW_DI + i];
    tv[i] = a.v[(size_t)g * a.nt * TW_DI + i];
  }
  // ---- phase 1 ------------------------------------------------------------------------------------------------------------------
  f32x16 qacc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) qacc[j][r] = 0.f;
  const int nkc = a.D / TW_BK;       // 8
  for (int kc = 0; kc < nkc; ++kc) {
    if ((kc & 3) == 0) stage_half(pa, wave, lane, kc >> 2, xs);
    if (kc >= 1 && kc + 3 < nkc) dma_wq(kc + 3);                      // into the slot chunk kc - 1 left at the last barrier
    {                                                                 // chunk kc landed; the (up to three) younger ones stay in flight
      const int younger = min(3, nkc - 1 - kc);
      if (younger == 3) dma_wait<12>();
      else if (younger == 2) dma_wait<8>();
      else if (younger == 1) dma_wait<4>();
      else dma_wait<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* pw = smem + 65536 + (kc & 3) * 16384;
    const char* pah = pa + (kc & 3) * TW_APLANE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 ah = *reinterpret_cast<const uint4*>(pah + tw_off(wave * 32 + fr, ks * 2 + fh));
      const uint4 al = *reinterpret_cast<const uint4*>(pah + 4 * TW_APLANE + tw_off(wave * 32 + fr, ks * 2 + fh));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint4 wh = *reinterpret_cast<const uint4*>(pw + tw_off(j * 32 + fr, ks * 2 + fh));
        const uint4 wl = *reinterpret_cast<const uint4*>(pw + TW_APLANE + tw_off(j * 32 + fr, ks * 2 + fh));
        qacc[j] = Half16<f16_t>::mfma32(ah, wh, qacc[j]);
        qacc[j] = Half16<f16_t>::mfma32(al, wh, qacc[j]);
        qacc[j] = Half16<f16_t>::mfma32(ah, wl, qacc[j]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  // phase 3's first two weight chunks travel during the attention phase (the ring is free: every wave passed the last barrier)
  unsigned osoff[8];
  const f16_t* obase[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int piece = wave * 8 + i;                  // 2 planes x 16 groups of 16 rows (Wo has D = 256 rows)
    const int r = (piece & 15) * 16 + (lane >> 2);
    osoff[i] = (unsigned)(((size_t)r * TW_DI + (((lane & 3) ^ ((r >> 2) & 3)) << 3)) * 2);
    obase[i] = (piece >> 4) ? a.wo_lo : a.wo_hi;
  }
  auto dma_wo = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dma16s(obase[i] + kc * TW_BK, osoff[i], lds_w + (kc & 1) * 32768 + (wave * 8 + i) * 1024);
  };
  dma_wo(0);
  dma_wo(1);
  // ---- phase 2: attention over the tokens, one 32-column tile (two heads) at a time -------------------------------------------------------
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float m[16], l[16], o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      qacc[j][r] = (qacc[j][r] + pq[j][r]) * a.scale;
      m[r] = -3.0e38f;
      l[r] = o[r] = 0.f;
    }
    for (int t = 0; t < a.nt; ++t) {
      const float kv = tk[t * TW_DI + j * 32 + fr], vv = tv[t * TW_DI + j * 32 + fr];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s = row16_sum(qacc[j][r] * kv);
        const float mn = fmaxf(m[r], s);
        const float corr = __expf(m[r] - mn), p = __expf(s - mn);
        l[r] = l[r] * corr + p;
        o[r] = o[r] * corr + p * vv;
        m[r] = mn;
      }
    }
    // O -> [hi | lo] planes in the A-operand image of k-chunk j: row = wave * 32 + (r & 3) + 8 (r >> 2) + 4 fh, column fr.  Lane pairs
    // (fr, fr ^ 1) hold neighbouring columns: they trade halves over DPP, the even lane stores the hi dword and the odd lane the lo
    // dword of the pair (one 32-bit LDS store per lane instead of two 16-bit ones)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float ov = o[r] * __builtin_amdgcn_rcpf(l[r]);
      const float hv = (float)(f16_t)ov, lv = ov - hv;
      const float hn = dpp_mov<0xB1>(hv), ln = dpp_mov<0xB1>(lv);          // the neighbour's values
      const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      const int off = j * TW_APLANE + tw_off(row, fr >> 3) + (fr & 6) * 2;
      const bool odd = fr & 1;
      const uint32_t w = odd ? pack2<f16_t>(ln, lv) : pack2<f16_t>(hv, hn);
      *reinterpret_cast<uint32_t*>(pa + (odd ? 4 * TW_APLANE : 0) + off) = w;
    }
  }
  // ---- phase 3: Y = O Wo^T, weight chunks double buffered ---------------------------------------------------------------------------------
  f32x16 yacc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) yacc[j][r] = 0.f;
  for (int kc = 0; kc < TW_DI / TW_BK; ++kc) {
    if (kc >= 1 && kc + 1 < TW_DI / TW_BK) dma_wo(kc + 1);            // chunks 0 and 1 left during the attention phase
    if (kc + 1 < TW_DI / TW_BK) dma_wait<8>();
    else dma_wait<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* pw = smem + 65536 + (kc & 1) * 32768;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 ah = *reinterpret_cast<const uint4*>(pa + kc * TW_APLANE + tw_off(wave * 32 + fr, ks * 2 + fh));
      const uint4 al = *reinterpret_cast<const uint4*>(pa + 4 * TW_APLANE + kc * TW_APLANE + tw_off(wave * 32 + fr, ks * 2 + fh));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint4 wh = *reinterpret_cast<const uint4*>(pw + tw_off(j * 32 + fr, ks * 2 + fh));
        const uint4 wl = *reinterpret_cast<const uint4*>(pw + 256 * 64 + tw_off(j * 32 + fr, ks * 2 + fh));
        yacc[j] = Half16<f16_t>::mfma32(ah, wh, yacc[j]);
        yacc[j] = Half16<f16_t>::mfma32(al, wh, yacc[j]);
        yacc[j] = Half16<f16_t>::mfma32(ah, wl, yacc[j]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  // ---- phase 4: + bias -> LDS row-major, then a wave per row: residual (registers), LayerNorm, store -----------------------------------------
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float bov = a.bo[j * 32 + fr];
#pragma unroll
    for (int r = 0; r < 16; ++r) yt[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * TW_YLD + j * 32 + fr] = yacc[j][r] + bov;
  }
  // (each wave wrote and now reads only its own 32 rows; the barrier above already retired every other use of this LDS)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float4 gm = reinterpret_cast<const float4*>(a.gamma)[lane], bt = reinterpret_cast<const float4*>(a.beta)[lane];
  // four passes over the wave's 32 rows, each with 32 independent chains (LDS reads, DPP reductions, stores all pipeline)
#pragma unroll
  for (int rr = 0; rr < 32; ++rr) {
    const float4 y = *reinterpret_cast<const float4*>(&yt[(wave * 32 + rr) * TW_YLD + lane * 4]);
    xs[rr].x += y.x; xs[rr].y += y.y; xs[rr].z += y.z; xs[rr].w += y.w;
  }
  float mu[32], rs[32];
#pragma unroll
  for (int rr = 0; rr < 32; ++rr) mu[rr] = wave_sum_dpp((xs[rr].x + xs[rr].y) + (xs[rr].z + xs[rr].w)) * (1.0f / 256.0f);
#pragma unroll
  for (int rr = 0; rr < 32; ++rr) {
    xs[rr].x -= mu[rr]; xs[rr].y -= mu[rr]; xs[rr].z -= mu[rr]; xs[rr].w -= mu[rr];
    rs[rr] = wave_sum_dpp((xs[rr].x * xs[rr].x + xs[rr].y * xs[rr].y) + (xs[rr].z * xs[rr].z + xs[rr].w * xs[rr].w));
  }
#pragma unroll
  for (int rr = 0; rr < 32; ++rr) {
    const int row = row0 + wave * 32 + rr;
    const float rstd = 1.0f / sqrtf(rs[rr] * (1.0f / 256.0f) + a.eps);
    if (row < a.hw)
      *reinterpret_cast<float4*>(xg + (size_t)row * a.D + lane * 4) =
          make_float4(xs[rr].x * rstd * gm.x + bt.x, xs[rr].y * rstd * gm.y + bt.y, xs[rr].z * rstd * gm.z + bt.z, xs[rr].w * rstd * gm.w + bt.w);
  }
}

}  // namespace la

extern "C" int la_twoway_t2i(const float* img, const void* wk_hi, const void* wk_lo, const void* wv_hi, const void* wv_lo, const float* pek,
                             const float* bv, const float* q, int G, int hw, int nt, int D, int heads, float* part, float* out, void* stream) {
  LA_CHECK_ARG(img && wk_hi && wk_lo && wv_hi && wv_lo && pek && bv && q && part && out, "la_twoway_t2i: null pointer");
  LA_CHECK_ARG(D == 2 * la::TW_DI && heads == 8, "la_twoway_t2i: built for D = 256 with 8 heads (got D=%d heads=%d)", D, heads);
  LA_CHECK_ARG(G > 0 && hw > 0 && nt > 0, "la_twoway_t2i: bad shape");
  const int S = (hw + la::TW_ROWS - 1) / la::TW_ROWS;
  la::TwT2iArgs a{img, (const la::f16_t*)wk_hi, (const la::f16_t*)wk_lo, (const la::f16_t*)wv_hi, (const la::f16_t*)wv_lo, pek, bv, q,
                  part, G, hw, nt, D, S, 1.0f / sqrtf((float)la::TW_HD)};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  constexpr int LDS = 160 * 1024;
  static unsigned long long attr_mask = 0;
  la::ensure_dyn_lds(reinterpret_cast<const void*>(la::twoway_t2i_kernel), LDS, attr_mask);
  hipLaunchKernelGGL(la::twoway_t2i_kernel, dim3(S, G), dim3(256), LDS, st, a);
  hipLaunchKernelGGL(la::twoway_merge_kernel, dim3(nt, G), dim3(128), 0, st, part, S * 4, nt, out);
  LA_CHECK_LAUNCH("la_twoway_t2i");
  return 0;
}

extern "C" int la_twoway_i2t(float* img, const void* wq_hi, const void* wq_lo, const float* peq, const float* k, const float* v,
                             const void* wo_hi, const void* wo_lo, const float* bo, const float* gamma, const float* beta, float eps, int G,
                             int hw, int nt, int D, int heads, void* stream) {
  LA_CHECK_ARG(img && wq_hi && wq_lo && peq && k && v && wo_hi && wo_lo && bo && gamma && beta, "la_twoway_i2t: null pointer");
  LA_CHECK_ARG(D == 2 * la::TW_DI && heads == 8, "la_twoway_i2t: built for D = 256 with 8 heads (got D=%d heads=%d)", D, heads);
  LA_CHECK_ARG(G > 0 && hw > 0 && nt > 0 && nt <= la::TW_MAXT, "la_twoway_i2t: nt=%d out of range (1..%d)", nt, la::TW_MAXT);
  la::TwI2tArgs a{img, (const la::f16_t*)wq_hi, (const la::f16_t*)wq_lo, peq, k, v, (const la::f16_t*)wo_hi, (const la::f16_t*)wo_lo, bo, gamma,
                  beta, eps, 1.0f / sqrtf((float)la::TW_HD), G, hw, nt, D};
  constexpr int LDS = 160 * 1024;
  static unsigned long long attr_mask = 0;
  la::ensure_dyn_lds(reinterpret_cast<const void*>(la::twoway_i2t_kernel), LDS, attr_mask);
  hipLaunchKernelGGL(la::twoway_i2t_kernel, dim3((hw + la::TW_ROWS - 1) / la::TW_ROWS, G), dim3(256), LDS, reinterpret_cast<hipStream_t>(stream), a);
  LA_CHECK_LAUNCH("la_twoway_i2t");
  return 0;
}
