// Backward of the plain (HF ViT) flash attention of la_attn_fwd for gfx950, head_dim 64 - the gradient of
//   O = softmax(Q K^T * scale) V      (transformers ViTSelfAttention under build_encoder.py:83-100; autograd of what
//                                      label_anything/models/lam.py:321-347 trains when the backbone is not frozen)
// Two kernels, no atomics, the T x T matrices never exist:
//   attn_bwd_dq_kernel   one workgroup = 4 waves = 128 QUERY rows of one (image, head), loop over 64-key tiles.  Same orientation as
//                        the forward (scores transposed, S^T = K Q^T, one query per lane): P^T = exp2(S^T c - LSE) with the row's LSE
//                        from the forward, dP^T = V dO^T, dS^T = P^T (dP^T - D), dQ^T += K^T dS^T.  Also writes D = rowsum(dO * O).
//   attn_bwd_dkv_kernel  one workgroup = 128 KEY rows, loop over 64-query tiles, one key per lane: S = Q K^T, dP = dO V^T,
//                        dV^T += dO^T P, dK^T += Q^T dS.
// In both, the operand that is reduced over tokens is needed token-contiguous, so the caller passes TRANSPOSED copies (la_head_transpose:
// K^T for dQ; Q^T and dO^T for dK / dV), laid out like the forward's V^T: [B*heads, 64, Tpad], zero padded.  P and dS go from the
// accumulator layout into the B operand of the next MFMA with v_permlane32_swap, exactly as P does in the forward kernel.
// All 16-bit operands share one dtype; gradients arrive pre-scaled (loss scaling is the host's business: train_encoder.py).
#include <cstdlib>
#include "la_common.h"
#include "../../include/la_hip.h"

namespace la {

constexpr float BWD_NEG_BIG = -1.0e30f;
constexpr int TILE_B = 64 * 64 * 2;      // one 64 x 64 16-bit tile: 8 KiB, 128-byte rows, XOR swizzled (swz_off)

struct AttnBwdEncArgs {
  const void* qkv;      // [B*T, 3E] 16-bit: q | k | v rows
  const void* dout;     // [B*T, E] 16-bit: dO
  const void* out;      // [B*T, E] 16-bit: O (forward output)             (dq kernel)
  const void* kt;       // [B*heads, 64, Tpad]: K^T                          (dq kernel)
  const void* qt;       // [B*heads, 64, Tpad]: Q^T                          (dkv kernel)
  const void* dot;      // [B*heads, 64, Tpad]: dO^T                         (dkv kernel)
  float* lse;           // [B*heads, Tpad] log2-domain LSE of the forward, +BIG beyond T
  float* dvec;          // [B*heads, Tpad] D = rowsum(dO * O), 0 beyond T (written by the dq kernel)
  void* dqkv;           // [B*T, 3E] 16-bit: dq | dk | dv rows (written)
  int B, heads, T, Tpad, E;
  float scale;
  // decomposed relative-position bias (SAM ViTDet blocks, image_encoder.py:340-376): S = scale q.k + relh[q][key / G] + relw[q][key % G]
  const float* relh;    // [B*heads, T, G] fp32 (la_relpos_terms), nullptr = plain attention
  const float* relw;
  float* drelh;         // [B*heads, T, G] fp32: d relh[q][kh] = sum_kw dS[q][(kh, kw)]   (written by the dq kernel)
  float* drelw;         //                       d relw[q][kw] = sum_kh dS[q][(kh, kw)]
  int G;
};

// Workgroup -> (image-head, 128-row block), as in the forward kernel (attn_enc.hip): workgroup b runs on XCD b % 8, so inside one XCD the
// row block runs fastest and the workgroups an XCD has in flight share the tiles they loop over in its L2 (with the image-head fastest
// every workgroup streams its own image-head's K / V or Q / dO from the fabric: 2.4 GB per launch at 52 x 12 x 901)
__device__ __forceinline__ void bwd_block_map(const AttnBwdEncArgs& a, int& bh, int& blk) {
  const int BH = a.B * a.heads, nb = (a.T + 127) / 128;
  if ((BH & 7) == 0) {
    const int idx = blockIdx.x >> 3;
    blk = idx % nb;
    bh = (idx / nb) * 8 + (blockIdx.x & 7);
  } else {
    bh = blockIdx.x % BH;
    blk = blockIdx.x / BH;
  }
}

// stage rows [row0, row0 + 64) (clamped to maxrow) x 64 columns of a row-major 16-bit matrix into one swizzled LDS tile
template <typename T>
__device__ __forceinline__ void stage_rows(const T* base, size_t ld, int row0, int maxrow, unsigned lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (i * 4 + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    dma16(base + (size_t)min(row0 + row, maxrow) * ld + chunk * 8, lds_tile + (i * 4 + wave) * 1024);
  }
}

// Tiles that are read BOTH ways - by rows (ds_read_b128: the A operand of S / dP) and transposed (ds_read_b64_tr_b16: the A operand of the
// product that reduces over the tile's 64 tokens, K^T / Q^T / dO^T of rounds 3 - 4's la_head_transpose copies) - use the chunk swizzle
// swzp(r) = 4 ((r >> 1) & 1) + ((r >> 2) & 3): a bit permutation of the row-read swizzle (r >> 1) & 7 (so the 16-lane groups of a
// ds_read_b128 still hit 16 different 16-byte slots), whose bit 2 separates rows r and r + 2 - a 32-lane pass of the transpose read
// (4 token rows x 64 bytes) covers every bank once.
__device__ __forceinline__ int swzp(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int swzp_off(int row, int chunk) { return row * 128 + ((chunk ^ swzp(row)) << 4); }
template <typename T>
__device__ __forceinline__ void stage_rows_p(const T* base, size_t ld, int row0, int maxrow, unsigned lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (i * 4 + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ swzp(row);
    dma16(base + (size_t)min(row0 + row, maxrow) * ld + chunk * 8, lds_tile + (i * 4 + wave) * 1024);
  }
}
// byte offsets of this lane's two transpose reads of a tile [64 tokens][64 dims] for dimension block dd (32 dims): tokens 8 fh + 4 h + j
// of every 16-token step (+ ks * 2048), see attn_enc.hip VROW: in a 16-lane group lane 4 j + c fetches (token row j, dims D + 4 c ..)
__device__ __forceinline__ void tr_offsets(int lane, unsigned (&tro)[2][2]) {
  const int fh = lane >> 5, j = (lane & 15) >> 2, c = lane & 3, gd = (lane >> 4) & 1;
#pragma unroll
  for (int dd = 0; dd < 2; ++dd)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 8 * fh + 4 * h + j;
      tro[dd][h] = (unsigned)(row * 128 + (((4 * dd + 2 * gd + (c >> 1)) ^ swzp(row)) << 4) + (c & 1) * 8);
    }
}
// A operand (32 dims of block dd x tokens 16 ks .. 16 ks + 15) of a product that reduces over the tile's tokens
__device__ __forceinline__ uint4 tr_frag(unsigned tile_lds, const unsigned (&tro)[2][2], int dd, int ks) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(tile_lds + tro[dd][0] + ks * 2048));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(tile_lds + tro[dd][1] + ks * 2048));
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// accumulator tile pair (rows = 64 reduction tokens of one lane's column) -> the four B-operand k-slices, as in the forward kernel
template <typename T>
__device__ __forceinline__ void to_b_frags(const f32x16 (&s)[2], uint4 (&pf)[4]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int t = ks >> 1, g0 = (ks & 1) * 8;
    const uint32_t x0 = pack2<T>(s[t][g0 + 0], s[t][g0 + 1]);
    const uint32_t x1 = pack2<T>(s[t][g0 + 2], s[t][g0 + 3]);
    const uint32_t y0 = pack2<T>(s[t][g0 + 4], s[t][g0 + 5]);
    const uint32_t y1 = pack2<T>(s[t][g0 + 6], s[t][g0 + 7]);
    const auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
    pf[ks] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
  }
}

template <typename T> __device__ __forceinline__ float dot8(uint4 a, uint4 b) {
  const T* pa = reinterpret_cast<const T*>(&a);
  const T* pb = reinterpret_cast<const T*>(&b);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)pa[i] * (float)pb[i];
  return s;
}

// ---------------------------------------------------------------------------------------------------------------------------
// BIAS 0: plain.  BIAS 1: rel-pos bias, any G <= 32 (per-wave LDS tables of the wave's 32 query rows, as the forward's MODE 1; the bias
// gradients are scatter-added into a second pair of tables with ds_add_f32 - the two half-lanes of a query row hit the same entries).
// BIAS 3: G <= 16 (the 14 x 14 windows) - no per-score work at all: the bias enters S^T as the product E R on the matrix pipe (R in hi + lo
// 16-bit parts: exact to 2^-22), and d relh | d relw = E^T dS^T accumulates beside dQ^T from the same dS^T fragments.
// BIAS 2: G == 64 - a 64-key tile is exactly one key row: relh[q][tile] is one scalar per tile, relw[q][0..63] lives in the 32 score
// registers' positions for the whole kernel, d relw accumulates in 32 more registers and d relh[q][tile] is one store per tile.
// NH = head width / 64 (1 or 2): a 128-wide head (SAM ViT-H's 80 zero-padded by the packed weights, engine.py head_pad) is two 64-wide
// halves everywhere - NH K tiles + NH V tiles per stage, 4 NH k-slices of S^T / dP^T, 2 NH accumulator tiles of dQ^T - on one wave per SIMD.
template <typename T, int BIAS, int NH>
__global__ __launch_bounds__(256, (NH == 1 ? 2 : 1)) void attn_bwd_dq_kernel(AttnBwdEncArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 2 * NH * TILE_B;        // K rows (read by rows and transposed; NH tiles of 64 dims) | V rows
  constexpr int HDT = 64 * NH, KS = 4 * NH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  int bh, qblk;
  bwd_block_map(a, bh, qblk);
  const int h = bh % a.heads, b = bh / a.heads;
  const int T_ = a.T, E3 = 3 * a.E;
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const int q = qblk * 128 + wave * 32 + fr, qc = min(q, T_ - 1);
  const float c2 = a.scale * 1.44269504088896340736f;

  uint4 qf[KS], dof[KS];
  float dsum = 0.f;
  {
    const size_t row = (size_t)b * T_ + qc;
    const T* pq = qkv + row * E3 + h * HDT + fh * 8;
    const T* pd = reinterpret_cast<const T*>(a.dout) + row * a.E + h * HDT + fh * 8;
    const T* po = reinterpret_cast<const T*>(a.out) + row * a.E + h * HDT + fh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = *reinterpret_cast<const uint4*>(pq + ks * 16);
      dof[ks] = *reinterpret_cast<const uint4*>(pd + ks * 16);
      dsum += dot8<T>(dof[ks], *reinterpret_cast<const uint4*>(po + ks * 16));
    }
  }
  dsum += __shfl_xor(dsum, 32, 64);               // D of this lane pair's query
  const float lse2 = a.lse[(size_t)bh * a.Tpad + qc];
  // rows in [T, Tpad) are written here too (D = 0, LSE = +BIG: p = 0 in the dK / dV kernel) - the C entry point does not depend on
  // how the caller initialised its buffers (an uninitialised D there would give 0 * NaN in dK)
  if (fh == 0 && q < a.Tpad) {
    a.dvec[(size_t)bh * a.Tpad + q] = q < T_ ? dsum : 0.f;
    if (q >= T_) a.lse[(size_t)bh * a.Tpad + q] = 1e30f;
  }

  const unsigned lds0 = lds_addr_of(smem);
  const T* kbase = qkv + (size_t)b * T_ * E3 + a.E + h * HDT;
  const T* vbase = kbase + a.E;
  auto dma = [&](int j, int stage) {
    const unsigned s0 = lds0 + stage * STAGE;
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
      stage_rows_p<T>(kbase + hh * 64, E3, j * 64, T_ - 1, s0 + hh * TILE_B, wave, lane);
      stage_rows<T>(vbase + hh * 64, E3, j * 64, T_ - 1, s0 + (NH + hh) * TILE_B, wave, lane);
    }
  };
  unsigned tro[2][2];
  tr_offsets(lane, tro);

  f32x16 acc[2 * NH];
#pragma unroll
  for (int d = 0; d < 2 * NH; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  const int ntiles = (T_ + 63) >> 6;
  // the row blocks of an image-head run side by side on one XCD: each starts its loop at a different key tile so that they do not all
  // ask the L2 for the same 24 KiB at the same moment (the sums are over all tiles either way; only the fp32 summation order moves)
  const int rot = (int)((long)qblk * ntiles / ((T_ + 127) >> 7));
  auto tile_of = [&](int it) { const int j = it + rot; return j >= ntiles ? j - ntiles : j; };
  dma(tile_of(0), 0);
  // ---- bias tables -------------------------------------------------------------------------------------------------------------
  const int G = a.G, GS = G + 1;
  float* my_bh = nullptr;       // BIAS 1: [32][GS] relh rows of the wave's queries | [32][GS] relw | [32][GS] d relh | [32][GS] d relw
  const int* keyinfo = nullptr;
  f32x16 rw[2], drw[2];         // BIAS 2: relw / d relw of the lane's 32 key columns
  const float* rhq = nullptr;
  if (BIAS == 1) {
    float* tab = reinterpret_cast<float*>(smem + 2 * STAGE);
    my_bh = tab + wave * 4 * 32 * GS;
    for (int idx = lane; idx < 32 * G; idx += 64) {
      const int i = idx / G, k = idx % G;
      const size_t src = ((size_t)bh * T_ + min(qblk * 128 + wave * 32 + i, T_ - 1)) * G + k;
      my_bh[i * GS + k] = a.relh[src];
      my_bh[(32 + i) * GS + k] = a.relw[src];
      my_bh[(64 + i) * GS + k] = 0.f;
      my_bh[(96 + i) * GS + k] = 0.f;
    }
    int* ki = reinterpret_cast<int*>(tab + 4 * 4 * 32 * GS);
    for (int k = tid; k < a.Tpad; k += 256) {
      const int kk = min(k, T_ - 1);
      ki[k] = ((kk / G) << 16) | (kk % G);
    }
    keyinfo = ki;
  }
  // BIAS 3: the bias and its gradient as products with the 0 / 1 matrix E[key][c] (c < 16: key row == c; c >= 16: key column == c - 16)
  uint4 rfh[2], rfl[2];         // R[c][q] = relh[q][c] | relw[q][c - 16], divided by the scale, hi + lo 16-bit parts: B operand of S^T += E R
  f32x16 drel;                  // d R[c][q] = sum_key E^T[c][key] dS^T[key][q]
  const char* et = nullptr;     // E^T [32][Tpad] 16-bit in LDS, rows padded by 16 bytes
  int ETS = 0;
  if (BIAS == 3) {
    ETS = a.Tpad * 2 + 16;
    char* etw = smem + 2 * STAGE;
    int* ki = reinterpret_cast<int*>(etw + 32 * ETS);
    for (int o = tid * 16; o < 32 * ETS; o += 256 * 16) *reinterpret_cast<uint4*>(etw + o) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const uint16_t one = (uint16_t)pack2<T>(1.0f, 0.0f);
    for (int k = tid; k < a.Tpad; k += 256) {
      const int kk = min(k, T_ - 1), kh = kk / G, kw = kk % G;
      ki[k] = (kh << 16) | kw;
      if (k < T_) {
        *reinterpret_cast<uint16_t*>(etw + kh * ETS + k * 2) = one;
        *reinterpret_cast<uint16_t*>(etw + (16 + kw) * ETS + k * 2) = one;
      }
    }
    et = etw;
    keyinfo = ki;
    const float inv = 1.0f / a.scale;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float* src = (ks == 0 ? a.relh : a.relw) + ((size_t)bh * T_ + qc) * G;
      uint32_t hi[4], lo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = fh * 8 + 2 * i;
        const float v0 = c < G ? src[c] * inv : 0.f, v1 = c + 1 < G ? src[c + 1] * inv : 0.f;
        const T h0 = (T)v0, h1 = (T)v1;
        hi[i] = pack2<T>((float)h0, (float)h1);
        lo[i] = pack2<T>(v0 - (float)h0, v1 - (float)h1);
      }
      rfh[ks] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      rfl[ks] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) drel[r] = 0.f;
  }
  if (BIAS == 2) {
    rhq = a.relh + ((size_t)bh * T_ + qc) * 64;
    const float* p = a.relw + ((size_t)bh * T_ + qc) * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 v = *reinterpret_cast<const float4*>(p + t * 32 + 8 * g4 + 4 * fh);
        rw[t][g4 * 4 + 0] = v.x;
        rw[t][g4 * 4 + 1] = v.y;
        rw[t][g4 * 4 + 2] = v.z;
        rw[t][g4 * 4 + 3] = v.w;
        drw[t][g4 * 4 + 0] = drw[t][g4 * 4 + 1] = drw[t][g4 * 4 + 2] = drw[t][g4 * 4 + 3] = 0.f;
      }
  }
  const float inv_c = 1.0f / a.scale;            // the bias enters in units of the raw score (score = scale * (q.k + bias / scale))
  dma_wait<0>();
  __syncthreads();
  // a wave whose 32 query rows all lie beyond T (T = 901: three of the 32 waves of an image-head) only helps staging the tiles
  const bool idle_wave = qblk * 128 + wave * 32 >= T_;
  for (int it = 0; it < ntiles; ++it) {
    const int j = tile_of(it);
    if (it + 1 < ntiles) dma(tile_of(it + 1), (it + 1) & 1);
    if (idle_wave) {
      dma_wait<0>();
      __syncthreads();
      continue;
    }
    const char* sk = smem + (it & 1) * STAGE;
    const char* sv = sk + NH * TILE_B;
    const unsigned sk_lds = lds0 + (it & 1) * STAGE;
    f32x16 s[2], dp[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = dp[t][r] = 0.f;
      if (BIAS == 3) {
        // the lane's key row of E: a one at c = key row (first k-step) and at c = 16 + key column (second); this lane holds c = fh*8 .. +8
        const int info = keyinfo[j * 64 + t * 32 + fr];
        const uint32_t one = pack2<T>(1.0f, 0.0f);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int pos = (ks == 0 ? (info >> 16) : (info & 0xffff)) - fh * 8;
          const int d = pos >> 1;
          const uint32_t val = one << ((pos & 1) * 16);
          const uint4 ef = make_uint4(d == 0 ? val : 0u, d == 1 ? val : 0u, d == 2 ? val : 0u, d == 3 ? val : 0u);
          s[t] = Half16<T>::mfma32(ef, rfh[ks], s[t]);
          s[t] = Half16<T>::mfma32(ef, rfl[ks], s[t]);
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 kf = *reinterpret_cast<const uint4*>(sk + (ks >> 2) * TILE_B + swzp_off(t * 32 + fr, (ks & 3) * 2 + fh));
        const uint4 vf = *reinterpret_cast<const uint4*>(sv + (ks >> 2) * TILE_B + swz_off(t * 32 + fr, (ks & 3) * 2 + fh));
        s[t] = Half16<T>::mfma32(kf, qf[ks], s[t]);
        dp[t] = Half16<T>::mfma32(vf, dof[ks], dp[t]);
      }
    }
    const bool tail = j * 64 + 64 > T_;
    float rh = 0.f, drh = 0.f;
    if (BIAS == 2) rh = rhq[j];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = j * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        float sraw = s[t][r];
        int info = 0;
        if (BIAS == 1) {
          info = keyinfo[key];
          sraw += (my_bh[fr * GS + (info >> 16)] + my_bh[(32 + fr) * GS + (info & 0xffff)]) * inv_c;
        }
        if (BIAS == 2) sraw += (rh + rw[t][r]) * inv_c;
        const float sv_ = (tail && key >= T_) ? BWD_NEG_BIG : sraw;
        const float p = __builtin_amdgcn_exp2f(fmaf(sv_, c2, -lse2));
        const float dsv = p * (dp[t][r] - dsum);   // dS^T (before the 1/sqrt(d) factor, applied at the end)
        s[t][r] = dsv;
        if (BIAS == 1 && q < T_ && !(tail && key >= T_)) {
          atomicAdd(&my_bh[(64 + fr) * GS + (info >> 16)], dsv);
          atomicAdd(&my_bh[(96 + fr) * GS + (info & 0xffff)], dsv);
        }
        if (BIAS == 2) {
          drh += dsv;
          drw[t][r] += dsv;
        }
      }
    if (BIAS == 2) {
      drh += __shfl_xor(drh, 32, 64);
      if (q < T_ && fh == 0) a.drelh[((size_t)bh * T_ + q) * 64 + j] = drh;
    }
    uint4 dsf[4];
    to_b_frags<T>(s, dsf);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int d = 0; d < 2 * NH; ++d) {
        const uint4 ktf = tr_frag(sk_lds + (d >> 1) * TILE_B, tro, d & 1, ks);      // K^T[32 d .. + 32][keys 16 ks .. + 16] out of the row-major K tiles
        acc[d] = Half16<T>::mfma32(ktf, dsf[ks], acc[d]);
      }
    if (BIAS == 3) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 etf = *reinterpret_cast<const uint4*>(et + fr * ETS + (j * 64 + ks * 16 + fh * 8) * 2);
        drel = Half16<T>::mfma32(etf, dsf[ks], drel);
      }
    }
    dma_wait<0>();
    __syncthreads();
  }
  if (q < T_) {
    T* op = reinterpret_cast<T*>(a.dqkv) + ((size_t)b * T_ + q) * E3 + h * HDT;
#pragma unroll
    for (int d = 0; d < 2 * NH; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 v;
        v.x = pack2<T>(acc[d][g4 * 4 + 0] * a.scale, acc[d][g4 * 4 + 1] * a.scale);
        v.y = pack2<T>(acc[d][g4 * 4 + 2] * a.scale, acc[d][g4 * 4 + 3] * a.scale);
        *reinterpret_cast<uint2*>(op + d * 32 + 8 * g4 + 4 * fh) = v;
      }
  }
  if (BIAS == 2 && q < T_) {
    float* p = a.drelw + ((size_t)bh * T_ + q) * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<float4*>(p + t * 32 + 8 * g4 + 4 * fh) =
            make_float4(drw[t][g4 * 4 + 0], drw[t][g4 * 4 + 1], drw[t][g4 * 4 + 2], drw[t][g4 * 4 + 3]);
  }
  if (BIAS == 3 && q < T_) {
    // register r of the accumulator is c = (r & 3) + 8 (r >> 2) + 4 fh: r < 8 the key rows, r >= 8 the key columns
    float* ph = a.drelh + ((size_t)bh * T_ + q) * G;
    float* pw = a.drelw + ((size_t)bh * T_ + q) * G;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int c = (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (c < G) {
        ph[c] = drel[r];
        pw[c] = drel[8 + r];
      }
    }
  }
  if (BIAS == 1) {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int idx = lane; idx < 32 * G; idx += 64) {
      const int i = idx / G, k = idx % G;
      const int qi = qblk * 128 + wave * 32 + i;
      if (qi < T_) {
        a.drelh[((size_t)bh * T_ + qi) * G + k] = my_bh[(64 + i) * GS + k];
        a.drelw[((size_t)bh * T_ + qi) * G + k] = my_bh[(96 + i) * GS + k];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// BIAS: the rel-pos terms of (query register, key lane) are read straight from the fp32 term arrays (a fixed register = one query row:
// the 64 keys of the wave read inside one <= 256-byte row of relh / relw)
// BIAS 0: none.  1: the two terms of a score read from global memory (odd G: rows of 64 queries are not 16-byte aligned).  2: staged in LDS.
template <typename T, int BIAS, int NH>
__global__ __launch_bounds__(256, (NH == 1 ? 2 : 1)) void attn_bwd_dkv_kernel(AttnBwdEncArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Q rows | dO rows (both read by rows and transposed) | LSE (64 floats) | D (64 floats) | BIAS: the rel-pos terms of the tile's 64
  // queries - relw rows [64][G] fp32 (16 KiB at G = 64) and relh: [64][G] for G <= 32, the 4-column group that holds this workgroup's
  // two key rows [64][4] for G = 64 (a workgroup = 128 keys = two key rows) - staged by LDS-DMA like the tiles (round 4 read two terms per
  // score from global memory: 64 vector-memory instructions per tile and wave)
  constexpr int BIAS_B = BIAS == 2 ? 16384 + 1024 : 0;      // G = 64: relw 16 KiB + relh 1 KiB; G <= 32: relw <= 8 KiB, relh <= 8 KiB behind it
  constexpr int STAGE = 2 * NH * TILE_B + 512 + BIAS_B;      // (NH tiles of 64 dims each for Q and for dO)
  constexpr int HDT = 64 * NH, KS = 4 * NH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  int bh, kblk;
  bwd_block_map(a, bh, kblk);
  const int h = bh % a.heads, b = bh / a.heads;
  const int T_ = a.T, E3 = 3 * a.E;
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const int key = kblk * 128 + wave * 32 + fr, kc = min(key, T_ - 1);
  const float c2 = a.scale * 1.44269504088896340736f;

  uint4 kf[KS], vf[KS];
  {
    const T* pk = qkv + ((size_t)b * T_ + kc) * E3 + a.E + h * HDT + fh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = *reinterpret_cast<const uint4*>(pk + ks * 16);
      vf[ks] = *reinterpret_cast<const uint4*>(pk + a.E + ks * 16);
    }
  }
  const int G = BIAS ? a.G : 1;
  const int kh_l = kc / G, kw_l = kc % G;
  const float inv_c = 1.0f / a.scale;
  // staged bias terms need 16-byte aligned rows of 64 queries: (bh T + 64 i) G % 4 == 0 - every even G (the 14 x 14 windows, 64 x 64)
  constexpr bool stage_bias = BIAS == 2;
  const int relh_cols = G == 64 ? 4 : G;                                       // staged relh row length
  const int relh_c0 = G == 64 ? ((2 * kblk) & ~3) : 0;                          // first staged relh column
  const long rel_total = (long)a.B * a.heads * T_ * G;                          // floats in relh / relw
  const int relh_at = G == 64 ? 16384 : 8192;                                   // byte offset of the staged relh block behind relw
  const unsigned lds0 = lds_addr_of(smem);
  const T* qbase = qkv + (size_t)b * T_ * E3 + h * HDT;
  const T* dobase = reinterpret_cast<const T*>(a.dout) + (size_t)b * T_ * a.E + h * HDT;
  const float* lseb = a.lse + (size_t)bh * a.Tpad;
  const float* dvb = a.dvec + (size_t)bh * a.Tpad;
  auto dma = [&](int i, int stage) {
    const unsigned s0 = lds0 + stage * STAGE;
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
      stage_rows_p<T>(qbase + hh * 64, E3, i * 64, T_ - 1, s0 + hh * TILE_B, wave, lane);
      stage_rows_p<T>(dobase + hh * 64, a.E, i * 64, T_ - 1, s0 + (NH + hh) * TILE_B, wave, lane);
    }
    // the 64 LSE / D values of the query tile: 16 lanes x 16 bytes each (wave 0: LSE, wave 1: D); every wave issues the same
    // NUMBER of pieces per tile only matters for counted waits - this kernel waits for all of them (dma_wait<0>)
    if (wave < 2 && lane < 16) dma16((wave == 0 ? lseb : dvb) + i * 64 + lane * 4, s0 + 2 * NH * TILE_B + wave * 256);
    if (stage_bias) {
      const long row0 = (long)bh * T_ + i * 64;                // (rows beyond T: whatever follows in the array - their probabilities are 0)
      const unsigned sb = s0 + 2 * NH * TILE_B + 512;
      const int npw = (64 * G * 4 + 1023) >> 10;               // 1 KiB pieces of the relw block [64][G] (contiguous rows)
      for (int pc = wave; pc < npw; pc += 4) {
        const long off = min(row0 * G + pc * 256 + lane * 4, rel_total - 4);
        dma16(a.relw + off, sb + pc * 1024);
      }
      if (G == 64) {
        if (wave == 3) dma16(a.relh + min((row0 + lane) * 64 + relh_c0, rel_total - 4), sb + relh_at);      // lane = query: 4 columns
      } else {
        for (int pc = wave; pc < npw; pc += 4) {
          const long off = min(row0 * G + pc * 256 + lane * 4, rel_total - 4);
          dma16(a.relh + off, sb + relh_at + pc * 1024);
        }
      }
    }
  };
  unsigned tro[2][2];
  tr_offsets(lane, tro);

  f32x16 dv[2 * NH], dk[2 * NH];
#pragma unroll
  for (int d = 0; d < 2 * NH; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[d][r] = dk[d][r] = 0.f;

  const int ntiles = (T_ + 63) >> 6;
  const int rot = (int)((long)kblk * ntiles / ((T_ + 127) >> 7));      // (staggered start, as in the dQ kernel)
  auto tile_of = [&](int it) { const int i = it + rot; return i >= ntiles ? i - ntiles : i; };
  dma(tile_of(0), 0);
  dma_wait<0>();
  __syncthreads();
  const bool idle_wave = kblk * 128 + wave * 32 >= T_;       // (as in the dQ kernel: key rows beyond T)
  for (int it = 0; it < ntiles; ++it) {
    const int i = tile_of(it);
    if (it + 1 < ntiles) dma(tile_of(it + 1), (it + 1) & 1);
    if (idle_wave) {
      dma_wait<0>();
      __syncthreads();
      continue;
    }
    const char* sq = smem + (it & 1) * STAGE;
    const char* sdo = sq + NH * TILE_B;
    const unsigned sq_lds = lds0 + (it & 1) * STAGE, sdo_lds = sq_lds + NH * TILE_B;
    const float* slse = reinterpret_cast<const float*>(sq + 2 * NH * TILE_B);
    const float* sdv = slse + 64;
    const float* srw = reinterpret_cast<const float*>(sq + 2 * NH * TILE_B + 512);
    const float* srh = srw + relh_at / 4;
    f32x16 s[2], dp[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = dp[t][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 qf = *reinterpret_cast<const uint4*>(sq + (ks >> 2) * TILE_B + swzp_off(t * 32 + fr, (ks & 3) * 2 + fh));
        const uint4 df = *reinterpret_cast<const uint4*>(sdo + (ks >> 2) * TILE_B + swzp_off(t * 32 + fr, (ks & 3) * 2 + fh));
        s[t] = Half16<T>::mfma32(qf, kf[ks], s[t]);          // S[i][j]: lane = key j, registers = queries
        dp[t] = Half16<T>::mfma32(df, vf[ks], dp[t]);        // dP[i][j]
      }
    }
    f32x16 ds[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 l4 = *reinterpret_cast<const float4*>(slse + t * 32 + 8 * g4 + 4 * fh);
        const float4 d4 = *reinterpret_cast<const float4*>(sdv + t * 32 + 8 * g4 + 4 * fh);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = g4 * 4 + k;
          if (BIAS) {
            const int ql = t * 32 + 8 * g4 + 4 * fh + k;                         // the register's query inside the tile
            if (stage_bias) {
              s[t][r] += (srh[ql * relh_cols + kh_l - relh_c0] + srw[ql * G + kw_l]) * inv_c;
            } else {
              const size_t row = ((size_t)bh * T_ + min(i * 64 + ql, T_ - 1)) * G;
              s[t][r] += (a.relh[row + kh_l] + a.relw[row + kw_l]) * inv_c;
            }
          }
          const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], c2, -lv[k]));    // rows beyond T carry LSE = +BIG: p = 0
          s[t][r] = p;
          ds[t][r] = p * (dp[t][r] - dvv[k]);
        }
      }
    uint4 pf[4], dsf[4];
    to_b_frags<T>(s, pf);
    to_b_frags<T>(ds, dsf);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int d = 0; d < 2 * NH; ++d) {
        const uint4 dotf = tr_frag(sdo_lds + (d >> 1) * TILE_B, tro, d & 1, ks);      // dO^T / Q^T[32 d .. + 32][queries 16 ks .. + 16] out of the row-major tiles
        const uint4 qtf = tr_frag(sq_lds + (d >> 1) * TILE_B, tro, d & 1, ks);
        dv[d] = Half16<T>::mfma32(dotf, pf[ks], dv[d]);      // dV^T[d][j] += dO^T[d][i] P[i][j]
        dk[d] = Half16<T>::mfma32(qtf, dsf[ks], dk[d]);      // dK^T[d][j] += Q^T[d][i] dS[i][j]
      }
    dma_wait<0>();
    __syncthreads();
  }
  if (key < T_) {
    T* op = reinterpret_cast<T*>(a.dqkv) + ((size_t)b * T_ + key) * E3 + a.E + h * HDT;
#pragma unroll
    for (int d = 0; d < 2 * NH; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 vk, vv;
        vk.x = pack2<T>(dk[d][g4 * 4 + 0] * a.scale, dk[d][g4 * 4 + 1] * a.scale);
        vk.y = pack2<T>(dk[d][g4 * 4 + 2] * a.scale, dk[d][g4 * 4 + 3] * a.scale);
        vv.x = pack2<T>(dv[d][g4 * 4 + 0], dv[d][g4 * 4 + 1]);
        vv.y = pack2<T>(dv[d][g4 * 4 + 2], dv[d][g4 * 4 + 3]);
        *reinterpret_cast<uint2*>(op + d * 32 + 8 * g4 + 4 * fh) = vk;
        *reinterpret_cast<uint2*>(op + a.E + d * 32 + 8 * g4 + 4 * fh) = vv;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// dst[(b * heads + h) * 64 + d][t] = src[b * T + t][col0 + h * 64 + d], t < T; zero for T <= t < Tpad.  One workgroup per 64-token tile.
template <typename T>
__global__ __launch_bounds__(256) void head_transpose_kernel(const T* __restrict__ src, int ld, int col0, int heads, int T_, int Tpad,
                                                             T* __restrict__ dst) {
  __shared__ T tile[64][64 + 2];
  const int bh = blockIdx.y, t0 = blockIdx.x * 64;
  const int b = bh / heads, h = bh % heads;
  const int tid = threadIdx.x;
  {
    const int r = tid >> 2, c8 = (tid & 3) * 16;           // 64 rows x 4 threads x 16 halves
    const int t = t0 + r;
    uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
    if (t < T_) {
      const T* p = src + ((size_t)b * T_ + t) * ld + col0 + h * 64 + c8;
      v0 = *reinterpret_cast<const uint4*>(p);
      v1 = *reinterpret_cast<const uint4*>(p + 8);
    }
    const T* e0 = reinterpret_cast<const T*>(&v0);
    const T* e1 = reinterpret_cast<const T*>(&v1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      tile[r][c8 + i] = e0[i];
      tile[r][c8 + 8 + i] = e1[i];
    }
  }
  __syncthreads();
  {
    const int d = tid >> 2, t8 = (tid & 3) * 16;
    T o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = tile[t8 + i][d];
    T* p = dst + ((size_t)bh * 64 + d) * Tpad + t0 + t8;
    *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(&o[0]);
    *reinterpret_cast<uint4*>(p + 8) = *reinterpret_cast<const uint4*>(&o[8]);
  }
}

// launch both backward kernels (dq first: it writes D); bias 0 plain, 1 rel-pos G <= 32, 2 rel-pos G == 64
template <typename T, int BIAS, int NH>
static void launch_attn_bwd_nh(const AttnBwdEncArgs& a, hipStream_t st) {
  const int nblk = (a.T + 127) / 128 * a.B * a.heads;
  const int lds_dq = 2 * 2 * NH * TILE_B + (BIAS == 1   ? 4 * 4 * 32 * (a.G + 1) * (int)sizeof(float) + a.Tpad * (int)sizeof(int)
                                           : BIAS == 3 ? 32 * (a.Tpad * 2 + 16) + a.Tpad * (int)sizeof(int)
                                                       : 0);
  static unsigned long long m1 = 0, m2 = 0, m3 = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(attn_bwd_dq_kernel<T, BIAS, NH>), 160 * 1024, m1);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, BIAS, NH>), dim3(nblk), dim3(256), lds_dq, st, a);
  if (BIAS != 0 && (a.G & 1) == 0) {           // staged bias terms: even G (16-byte aligned rows of 64 queries)
    constexpr int LDS_DKV = 2 * (2 * NH * TILE_B + 512 + 16384 + 1024);      // 67 KiB: two workgroups per CU (wide heads: 99 KiB, one)
    ensure_dyn_lds(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<T, 2, NH>), LDS_DKV, m3);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, 2, NH>), dim3(nblk), dim3(256), LDS_DKV, st, a);
  } else {
    constexpr int LDS_DKV = 2 * (2 * NH * TILE_B + 512);
    ensure_dyn_lds(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<T, (BIAS != 0 ? 1 : 0), NH>), LDS_DKV, m2);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, (BIAS != 0 ? 1 : 0), NH>), dim3(nblk), dim3(256), LDS_DKV, st, a);
  }
}
template <typename T, int BIAS>
static void launch_attn_bwd_t(const AttnBwdEncArgs& a, hipStream_t st) {
  if (a.E == a.heads * 128) launch_attn_bwd_nh<T, BIAS, 2>(a, st);
  else launch_attn_bwd_nh<T, BIAS, 1>(a, st);
}
static void launch_attn_bwd(const AttnBwdEncArgs& a, int bias, int dt, hipStream_t st) {
  if (dt == LA_F16) {
    if (bias == 0) launch_attn_bwd_t<f16_t, 0>(a, st);
    else if (bias == 1) launch_attn_bwd_t<f16_t, 1>(a, st);
    else if (bias == 3) launch_attn_bwd_t<f16_t, 3>(a, st);
    else launch_attn_bwd_t<f16_t, 2>(a, st);
  } else {
    if (bias == 0) launch_attn_bwd_t<bf16_t, 0>(a, st);
    else if (bias == 1) launch_attn_bwd_t<bf16_t, 1>(a, st);
    else if (bias == 3) launch_attn_bwd_t<bf16_t, 3>(a, st);
    else launch_attn_bwd_t<bf16_t, 2>(a, st);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Backward of the decomposed rel-pos terms (image_encoder.py:340-376: rel_h[q][kh] = q . Rh[qy - kh + G - 1], rel_w[q][kw] =
// q . Rw[qx - kw + G - 1], q UNSCALED) given d rel_h / d rel_w from attn_bwd_dq_kernel.  One workgroup per (image-head, query row y):
//   dq[x][d]     += sum_kh drelh[x][kh] Rh[y - kh + G - 1][d] + sum_kw drelw[x][kw] Rw[x - kw + G - 1][d]      (added to the 16-bit dq rows)
//   dRh[y - kh + G - 1][d] += sum_x drelh[x][kh] q[x][d]        dRw[r][d] += sum_{x - kw + G - 1 = r} drelw[x][kw] q[x][d]   (fp32 atomics:
//   the tables are shared by every image, head and row).  Tables / table gradients are fp32 [(2 G - 1), 64].
// k-loop of an exact-fp32 MFMA product whose operands are gathered element by element: the 8 + 8 gathers of a chunk are issued before its
// 8 MFMAs, so one load latency is paid per chunk instead of per step (NS2 = number of k pairs, a multiple of 8)
template <typename FA, typename FB>
__device__ __forceinline__ void mfma32_gather(f32x16& acc, int ns2, FA&& fa, FB&& fb) {
  for (int c = 0; c < ns2; c += 8) {
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      av[u] = fa(c + u);
      bv[u] = fb(c + u);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
  }
}

template <typename T>
__global__ __launch_bounds__(256, 3) void relpos_bwd_kernel(const T* __restrict__ qkv, T* __restrict__ dqkv, const float* __restrict__ drelh,
                                                         const float* __restrict__ drelw, const float* __restrict__ tabh,
                                                         const float* __restrict__ tabw, float* __restrict__ dtabh, float* __restrict__ dtabw,
                                                         int B, int heads, int G, int E, float gscale, int RY, int c0) {
  // (heads wider than 64 - zero-padded 80 -> 128: SAM ViT-H - are walked in 64-column blocks: every output column is its own problem; HDs =
  // the head stride of the qkv rows AND the row stride of the fp32 tables, c0 = this launch's first column inside the head)
  const int HDs = E / heads;
  // Four small products per query row on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: A lane (fr, fh) = A[row fr][k fh], B lane = B[k fh][col fr],
  // accumulator register r / lane = row (r & 3) + 8 (r >> 2) + 4 fh / column fr), operands gathered from LDS with the index arithmetic of
  // the decomposition (the vector-ALU form of round 4's first version spent 7.8 ms per SAM-B training step in these loops):
  //   dq [x][d]  = drelh[x][:] . RhSel[:][d]  +  U[x][:] . Rw[:][d]       RhSel[kh] = Rh[y - kh + G - 1],  U[x][r] = drelw[x][x + G - 1 - r]
  //   dRhSel[kh][d] = drelh[:][kh]^T . q[:][d]                            -> dRh[y - kh + G - 1]
  //   dRw [r][d] = U[:][r]^T . q[:][d]
  // One workgroup walks RY consecutive rows of one image-head and folds their table gradients on chip - dRw stays in its MFMA accumulators
  // (its row index does not depend on y), dRh goes through an LDS table - so the tables receive ONE pass of atomics per workgroup: device-scope
  // atomics of every workgroup on the same 12 K addresses are what the one-row-per-workgroup form was waiting for.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int GS = G + 1;                                // row stride of the two G x G arrays (a lane walks a column of drelh)
  const int GP = (G + 31) & ~31, NREL = 2 * G - 1, RP = (NREL + 31) & ~31;
  float* sdh = reinterpret_cast<float*>(smem);         // [G][GS] drelh of the row's queries
  float* sdw = sdh + G * GS;                           // [G][GS] drelw
  float* sah = sdw + G * GS;                           // [NREL][64] dRh of this workgroup's rows
  T* sq = reinterpret_cast<T*>(sah + NREL * 64);       // [G][64] q of the row (16 bit, as stored)
  const int nchunk = (G + RY - 1) / RY;
  const int yc = blockIdx.x % nchunk, bh = blockIdx.x / nchunk, h = bh % heads, b = bh / heads;
  const int T_ = G * G, E3 = 3 * E, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int ta = wave >> 1, tb = wave & 1, dcol = tb * 32 + fr;
  f32x16 accw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[i][r] = 0.f;
  for (int i = tid; i < NREL * 64; i += 256) sah[i] = 0.f;
  for (int y = yc * RY; y < min(G, yc * RY + RY); ++y) {
    const size_t row0 = (size_t)b * T_ + (size_t)y * G;
    __syncthreads();                                   // the previous row's readers are done (first row: the zero fill has landed)
    for (int i = tid; i < G * 64; i += 256) sq[i] = qkv[(row0 + i / 64) * E3 + h * HDs + c0 + (i & 63)];
    for (int i = tid; i < G * G; i += 256) {
      sdh[(i / G) * GS + i % G] = drelh[((size_t)bh * T_ + (size_t)y * G) * G + i];
      sdw[(i / G) * GS + i % G] = drelw[((size_t)bh * T_ + (size_t)y * G) * G + i];
    }
    __syncthreads();
    // ---- dq rows: tile (x rows ta, d columns tb) ---------------------------------------------------------------------------------
    if (ta * 32 < G) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int x = ta * 32 + fr;
      mfma32_gather(acc, GP / 2,
                    [&](int s2) { const int kh = 2 * s2 + fh; return (x < G && kh < G) ? sdh[x * GS + kh] : 0.f; },
                    [&](int s2) { const int kh = 2 * s2 + fh; return kh < G ? tabh[(size_t)(y - kh + G - 1) * HDs + c0 + dcol] : 0.f; });
      mfma32_gather(acc, RP / 2,
                    [&](int s2) { const int r = 2 * s2 + fh, kw = x + G - 1 - r; return (x < G && r < NREL && kw >= 0 && kw < G) ? sdw[x * GS + kw] : 0.f; },
                    [&](int s2) { const int r = 2 * s2 + fh; return r < NREL ? tabw[(size_t)r * HDs + c0 + dcol] : 0.f; });
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int xo = ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (xo < G) {
          T* p = dqkv + (row0 + xo) * E3 + h * HDs + c0 + dcol;
          *p = (T)((float)*p + acc[r]);
        }
      }
      // ---- dRh[y - kh + G - 1][d]: tile (kh rows ta, d columns tb); for a fixed y every (kh, d) is its own table entry ------------------
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int kh = ta * 32 + fr;
      mfma32_gather(acc, GP / 2,
                    [&](int s2) { const int xx = 2 * s2 + fh; return (kh < G && xx < G) ? sdh[xx * GS + kh] : 0.f; },
                    [&](int s2) { const int xx = 2 * s2 + fh; return xx < G ? (float)sq[xx * 64 + dcol] : 0.f; });
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ko = ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (ko < G) sah[(y - ko + G - 1) * 64 + dcol] += acc[r];
      }
    }
    // ---- dRw[r][d]: 2 RP / 32 tiles (r rows, d columns), wave w owns tiles w and w + 4; accumulated over the workgroup's rows --------------
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const int t = wave + 4 * ti;
      if (t < 2 * (RP / 32)) {
        const int tr = t >> 1, tc = (t & 1) * 32 + fr, rr = tr * 32 + fr;
        mfma32_gather(accw[ti], GP / 2,
                      [&](int s2) { const int xx = 2 * s2 + fh, kw = xx + G - 1 - rr; return (rr < NREL && xx < G && kw >= 0 && kw < G) ? sdw[xx * GS + kw] : 0.f; },
                      [&](int s2) { const int xx = 2 * s2 + fh; return xx < G ? (float)sq[xx * 64 + tc] : 0.f; });
      }
    }
  }
  __syncthreads();
  // (gscale is a plain multiplier: the caller's buffers carry the loss scale like every other gradient of the backward pass)
  for (int i = tid; i < NREL * 64; i += 256) atomicAdd(&dtabh[(size_t)(i >> 6) * HDs + c0 + (i & 63)], sah[i] * gscale);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int t = wave + 4 * ti;
    if (t < 2 * (RP / 32)) {
      const int tr = t >> 1, tc = (t & 1) * 32 + fr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = tr * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (ro < NREL) atomicAdd(&dtabw[(size_t)ro * HDs + c0 + tc], accw[ti][r] * gscale);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same gradients for small grids (G <= 16: the 14 x 14 windows, 1200 image-heads of 196 queries per SAM-B block) as two dense
// products per 32 queries, without the per-row workgroups above (16 800 of them per launch, each a chain of three dependent global round
// trips: 0.5 ms for 90 MB of traffic).  With U[q][c] = drelh[q][y_q + G - 1 - c] (c < 32) | drelw[q][x_q + G - 1 - (c - 32)] (c >= 32)
// - the terms' gradients shifted onto the table row they multiply, zero outside the grid - and Rcat = [Rh ; Rw] padded to 2 x 32 rows:
//   dq[q][:]    += U[q][:] . Rcat           (computed transposed, dq^T = Rcat^T U^T: a lane ends up with 4-channel groups of its query)
//   dRcat[c][:] += sum_q U[q][c] q[q][:]
// on the exact-fp32 MFMA.  One wave = one unit of 32 consecutive queries of an image-head at a time (its drelh / drelw / q rows staged in a
// private LDS area, no workgroup barriers in the loop); dRcat stays in 64 accumulator registers over all the units a wave walks, the
// four waves meet in LDS at the end and the tables receive one atomic per entry and workgroup.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void relpos_bwd_rows_kernel(const T* __restrict__ qkv, T* __restrict__ dqkv, const float* __restrict__ drelh,
                                                                 const float* __restrict__ drelw, const float* __restrict__ tabh,
                                                                 const float* __restrict__ tabw, float* __restrict__ dtabh,
                                                                 float* __restrict__ dtabw, int B, int heads, int G, int E, float gscale, int c0) {
  const int HDs = E / heads;                          // (head stride = table row stride; c0: see relpos_bwd_kernel)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, fh = lane >> 5;
  const int T_ = G * G, E3 = 3 * E, NREL = 2 * G - 1, NS = (NREL + 1) >> 1;
  float* stab = reinterpret_cast<float*>(smem);                                   // [64][64]: Rh rows (zero beyond NREL) | Rw rows
  const int wave_lds = 2 * 32 * G * (int)sizeof(float) + 32 * 64 * (int)sizeof(T);
  float* sdh = reinterpret_cast<float*>(smem + 64 * 64 * sizeof(float) + wave * wave_lds);    // [32][G] drelh of the unit's queries
  float* sdw = sdh + 32 * G;
  T* sq = reinterpret_cast<T*>(sdw + 32 * G);                                     // [32][64] q rows
  for (int i = tid; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = c & 31;
    stab[i] = r < NREL ? (c < 32 ? tabh : tabw)[r * HDs + c0 + (i & 63)] : 0.f;
  }
  __syncthreads();
  f32x16 dR[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dR[i][j][r] = 0.f;
  const int units_per = (T_ + 31) >> 5, nunits = B * heads * units_per;
  const int ginv = 65536 / G + 1;                          // q / G == (q * ginv) >> 16 for q < 256
  for (int unit = blockIdx.x * 4 + wave; unit < nunits; unit += gridDim.x * 4) {
    const int bh = unit / units_per, q0 = (unit % units_per) * 32, b = bh / heads, h = bh % heads;
    {
      const size_t base = ((size_t)bh * T_ + q0) * G;
      const int n = min(32, T_ - q0) * G;
      for (int i = lane; i < 32 * G; i += 64) {
        sdh[i] = i < n ? drelh[base + i] : 0.f;
        sdw[i] = i < n ? drelw[base + i] : 0.f;
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = (lane >> 3) + 8 * p;
        uint4 v = *reinterpret_cast<const uint4*>(qkv + ((size_t)b * T_ + min(q0 + row, T_ - 1)) * E3 + h * HDs + c0 + (lane & 7) * 8);
        if (q0 + row >= T_) v = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(sq + row * 64 + (lane & 7) * 8) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // ---- dq^T[d][q] = sum_c Rcat[c][d] U[q][c]: A = table column block, B = the lane's query --------------------------------------
    const int q = q0 + fr, qv = min(q, T_ - 1), qy = (qv * ginv) >> 16, qx = qv - qy * G;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const float* sd = half ? sdw : sdh;
      const int pos = (half ? qx : qy) + G - 1;
      for (int s2 = 0; s2 < NS; ++s2) {
        const int c = 2 * s2 + fh, k = pos - c;
        const float bv = (k >= 0 && k < G) ? sd[fr * G + k] : 0.f;        // (c == NREL, the odd pad, has k = pos - NREL < 0 ... or a zero table row)
        const float* tr = stab + (half * 32 + c) * 64;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(tr[fr], bv, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(tr[32 + fr], bv, acc[1], 0, 0, 0);
      }
    }
    if (q < T_) {
      T* p = dqkv + ((size_t)b * T_ + q) * E3 + h * HDs + c0;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          T* pp = p + dt * 32 + 8 * g4 + 4 * fh;
          uint2 v = *reinterpret_cast<const uint2*>(pp);
          const T* cur = reinterpret_cast<const T*>(&v);
          uint2 o;
          o.x = pack2<T>((float)cur[0] + acc[dt][g4 * 4 + 0], (float)cur[1] + acc[dt][g4 * 4 + 1]);
          o.y = pack2<T>((float)cur[2] + acc[dt][g4 * 4 + 2], (float)cur[3] + acc[dt][g4 * 4 + 3]);
          *reinterpret_cast<uint2*>(pp) = o;
        }
    }
    // ---- dRcat[c][d] += sum_q U[q][c] q[q][d]: A = the lane's table row over the unit's queries, B = q rows ---------------------------
#pragma unroll 4
    for (int s2 = 0; s2 < 16; ++s2) {
      const int ql = 2 * s2 + fh, qg = min(q0 + ql, T_ - 1), yq = (qg * ginv) >> 16, xq = qg - yq * G;
      const int kh = yq + G - 1 - fr, kw = xq + G - 1 - fr;
      const float ah = (kh >= 0 && kh < G) ? sdh[ql * G + kh] : 0.f;      // (rows beyond T are zero in the staged arrays; fr >= NREL gives k < 0)
      const float aw = (kw >= 0 && kw < G) ? sdw[ql * G + kw] : 0.f;
      const float b0 = (float)sq[ql * 64 + fr], b1 = (float)sq[ql * 64 + 32 + fr];
      dR[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ah, b0, dR[0][0], 0, 0, 0);
      dR[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ah, b1, dR[0][1], 0, 0, 0);
      dR[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, b0, dR[1][0], 0, 0, 0);
      dR[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, b1, dR[1][1], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();                       // (the staging area is rewritten by the next unit)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // the four waves' table gradients meet in the table's own LDS area, then one atomic per entry
  __syncthreads();
  for (int i = tid; i < 64 * 64; i += 256) stab[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        atomicAdd(&stab[c * 64 + dt * 32 + fr], dR[ct][dt][r]);
      }
  __syncthreads();
  for (int i = tid; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = c & 31;
    if (r < NREL) atomicAdd(&(c < 32 ? dtabh : dtabw)[r * HDs + c0 + (i & 63)], stab[i] * gscale);
  }
}

}  // namespace la

extern "C" int la_head_transpose(const void* src, int ld, int col0, int B, int heads, int T, int Tpad, void* dst, int dt, void* stream) {
  LA_CHECK_ARG(src && dst && B > 0 && heads > 0 && T > 0, "la_head_transpose: bad arguments");
  LA_CHECK_ARG(Tpad >= T && (Tpad % 64) == 0 && (ld % 8) == 0 && (col0 % 8) == 0, "la_head_transpose: Tpad %% 64, ld %% 8, col0 %% 8 (Tpad=%d ld=%d col0=%d)",
               Tpad, ld, col0);
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16, "la_head_transpose: bad dtype %d", dt);
  const dim3 grid(Tpad / 64, B * heads), blk(256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dt == LA_F16)
    hipLaunchKernelGGL(la::head_transpose_kernel<la::f16_t>, grid, blk, 0, st, (const la::f16_t*)src, ld, col0, heads, T, Tpad, (la::f16_t*)dst);
  else
    hipLaunchKernelGGL(la::head_transpose_kernel<la::bf16_t>, grid, blk, 0, st, (const la::bf16_t*)src, ld, col0, heads, T, Tpad, (la::bf16_t*)dst);
  LA_CHECK_LAUNCH("la_head_transpose");
  return 0;
}

extern "C" int la_attn_bwd(const void* qkv, const void* out16, const void* dout16, const void* kt, const void* qt, const void* dot, float* lse,
                           float* dvec, void* dqkv, int B, int heads, int T, int Tpad, int E, float scale, int dt, void* stream) {
  LA_CHECK_ARG(qkv && out16 && dout16 && lse && dvec && dqkv, "la_attn_bwd: null pointer");      // (kt / qt / dot: unused since round 5)
  LA_CHECK_ARG(B > 0 && heads > 0 && T > 0 && (E == heads * 64 || E == heads * 128),
               "la_attn_bwd: needs head_dim 64 or 128 - other widths zero-padded (E=%d heads=%d)", E, heads);
  LA_CHECK_ARG(Tpad >= T && (Tpad % 64) == 0, "la_attn_bwd: Tpad=%d must be a multiple of 64 covering T=%d", Tpad, T);
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16, "la_attn_bwd: bad dtype %d", dt);
  la::AttnBwdEncArgs a{qkv, dout16, out16, kt, qt, dot, lse, dvec, dqkv, B, heads, T, Tpad, E, scale, nullptr, nullptr, nullptr, nullptr, 0};
  la::launch_attn_bwd(a, 0, dt, reinterpret_cast<hipStream_t>(stream));
  LA_CHECK_LAUNCH("la_attn_bwd");
  return 0;
}

extern "C" int la_attn_bwd_relpos(const void* qkv, const void* out16, const void* dout16, const void* kt, const void* qt, const void* dot,
                                  float* lse, float* dvec, void* dqkv, const float* relh, const float* relw, float* drelh, float* drelw, int B,
                                  int heads, int T, int Tpad, int G, int E, float scale, int dt, void* stream) {
  LA_CHECK_ARG(qkv && out16 && dout16 && lse && dvec && dqkv && relh && relw && drelh && drelw, "la_attn_bwd_relpos: null pointer");
  LA_CHECK_ARG(B > 0 && heads > 0 && T > 0 && (E == heads * 64 || E == heads * 128),
               "la_attn_bwd_relpos: needs head_dim 64 or 128 - other widths zero-padded (E=%d heads=%d)", E, heads);
  LA_CHECK_ARG(Tpad >= T && (Tpad % 64) == 0, "la_attn_bwd_relpos: Tpad=%d must be a multiple of 64 covering T=%d", Tpad, T);
  LA_CHECK_ARG(G * G == T && (G <= 32 || G == 64), "la_attn_bwd_relpos: T == G*G with G <= 32 or G == 64 (T=%d G=%d)", T, G);
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16, "la_attn_bwd_relpos: bad dtype %d", dt);
  la::AttnBwdEncArgs a{qkv, dout16, out16, kt, qt, dot, lse, dvec, dqkv, B, heads, T, Tpad, E, scale, relh, relw, drelh, drelw, G};
  la::launch_attn_bwd(a, G == 64 ? 2 : G <= 16 ? 3 : 1, dt, reinterpret_cast<hipStream_t>(stream));
  LA_CHECK_LAUNCH("la_attn_bwd_relpos");
  return 0;
}

extern "C" int la_relpos_bwd(const void* qkv, void* dqkv, const float* drelh, const float* drelw, const float* tabh, const float* tabw,
                             float* dtabh, float* dtabw, int B, int heads, int G, int E, float gscale, int dt, void* stream) {
  LA_CHECK_ARG(qkv && dqkv && drelh && drelw && tabh && tabw && dtabh && dtabw, "la_relpos_bwd: null pointer");
  LA_CHECK_ARG(B > 0 && heads > 0 && G > 0 && G <= 64 && (E == heads * 64 || E == heads * 128),
               "la_relpos_bwd: needs head_dim 64 or 128 (tables [(2 G - 1), head_dim] fp32), G <= 64 (E=%d heads=%d G=%d)", E, heads, G);
  const int hdw = E / heads;
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16, "la_relpos_bwd: bad dtype %d", dt);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (G <= 16) {                                     // the windows: dense products over units of 32 queries
    const int lds_rows = 64 * 64 * (int)sizeof(float) + 4 * (2 * 32 * G * (int)sizeof(float) + 32 * 64 * 2);
    const int nunits = B * heads * ((G * G + 31) / 32);
    const int grid_rows = nunits / 4 < 512 ? (nunits + 3) / 4 : 512;          // (256 / 512 / 768 workgroups on 100 x 12 windows: 135 / 117 / 132 us)
    for (int c0 = 0; c0 < hdw; c0 += 64) {
      if (dt == LA_F16)
        hipLaunchKernelGGL(la::relpos_bwd_rows_kernel<la::f16_t>, dim3(grid_rows), dim3(256), lds_rows, st, (const la::f16_t*)qkv, (la::f16_t*)dqkv,
                           drelh, drelw, tabh, tabw, dtabh, dtabw, B, heads, G, E, gscale, c0);
      else
        hipLaunchKernelGGL(la::relpos_bwd_rows_kernel<la::bf16_t>, dim3(grid_rows), dim3(256), lds_rows, st, (const la::bf16_t*)qkv,
                           (la::bf16_t*)dqkv, drelh, drelw, tabh, tabw, dtabh, dtabw, B, heads, G, E, gscale, c0);
    }
    LA_CHECK_LAUNCH("la_relpos_bwd");
    return 0;
  }
  const int lds = (2 * G * (G + 1) + (2 * G - 1) * 64) * (int)sizeof(float) + G * 64 * 2;        // 73 KiB at G = 64: two workgroups per CU
  // rows per workgroup.  Large grids (G == 64: 4 image-heads x 64 rows per CU): as many as keep >= ~3 workgroups per CU in the launch - the
  // table atomics shrink by that factor.  Windows (G <= 32, tables of a few KiB, workgroups of 10 KiB): one row each - a row's three
  // dependent global round trips (operands in, dq read-modify-write) are what it waits for, and only more resident workgroups hide them
  // (measured on 100 x 12 windows of 14 x 14: 471 us with 14 rows per workgroup)
  int ry = 1;
  while (G > 32 && ry < G && (long)B * heads * ((G + 2 * ry - 1) / (2 * ry)) >= 768) ry *= 2;
  const int grid = B * heads * ((G + ry - 1) / ry);
  static unsigned long long m1 = 0, m2 = 0;
  for (int c0 = 0; c0 < hdw; c0 += 64) {
    if (dt == LA_F16) {
      la::ensure_dyn_lds(reinterpret_cast<const void*>(la::relpos_bwd_kernel<la::f16_t>), 80 * 1024, m1);
      hipLaunchKernelGGL(la::relpos_bwd_kernel<la::f16_t>, dim3(grid), dim3(256), lds, st, (const la::f16_t*)qkv, (la::f16_t*)dqkv, drelh,
                         drelw, tabh, tabw, dtabh, dtabw, B, heads, G, E, gscale, ry, c0);
    } else {
      la::ensure_dyn_lds(reinterpret_cast<const void*>(la::relpos_bwd_kernel<la::bf16_t>), 80 * 1024, m2);
      hipLaunchKernelGGL(la::relpos_bwd_kernel<la::bf16_t>, dim3(grid), dim3(256), lds, st, (const la::bf16_t*)qkv, (la::bf16_t*)dqkv, drelh,
                         drelw, tabh, tabw, dtabh, dtabw, B, heads, G, E, gscale, ry, c0);
    }
  }
  LA_CHECK_LAUNCH("la_relpos_bwd");
  return 0;
}
