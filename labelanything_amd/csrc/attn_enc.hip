// Encoder attention for gfx950: decomposed rel-pos terms + flash attention (head_dim 64).
//
// la_attn_fwd   one workgroup = 4 waves = 128 query rows of one (image|window, head); each wave owns 32 query
//               rows.  Scores are computed TRANSPOSED, S^T = K . Q^T with MFMA 32x32x16 (A = K tile from LDS,
//               B = Q held in registers), so that every lane holds 32 of the 64 scores of ONE query row:
//               the online-softmax state (m, l, alpha) is lane-local and the row max needs a single
//               cross-lane exchange (lane ^ 32).  P is converted to 16 bit in registers and redistributed
//               with v_permlane32_swap into the B operand of O^T = V^T . P^T (A = V^T tile from LDS), so the
//               output accumulator is again one query row per lane.  V arrives pre-transposed ([hd][T], written
//               by the qkv GEMM epilogue), K/V tiles of 64 keys are double buffered in LDS (register staged,
//               XOR swizzled).  The T x T matrix never exists.  The SAM decomposed relative position bias
//               (image_encoder.py:340-376) enters as the INITIAL VALUE of the score accumulator:
//                 fast path (G == 64): a 64-key tile is exactly one key row, so the bias is
//                   relw[q][0..63] (held in registers for the whole kernel) + relh[q][tile] (one LDS read);
//                 generic path (windows, other grids): per-wave LDS tables indexed by (key / G, key % G).
// la_relpos_terms  one wave per (batch, head, query row y): relh = Q_y . Rh_y^T and U = Q_y . Rw^T on MFMA
//               straight from global memory, U scattered to relw[q][qx - r + G - 1].
#include <cstdlib>
#include "la_common.h"
#ifndef LA_ATTN_ABL
#ifndef LA_ATTN_DMA_RUN
#define LA_ATTN_DMA_RUN 1   // K / V tile sources as running 32-bit offsets (0: the per-tile clamp + multiply-add form, tools/attn_ab.sh A/B)
#endif
#define LA_ATTN_ABL 0       // measurement ablations of attn_fwd_kernel (results wrong): 1 no exp, 2 no S MFMAs, 4 no PV MFMAs, 8 no staging / barrier
#endif
#ifndef LA_WIN_ROT
#define LA_WIN_ROT 1        // SAM windows: rotate the wave -> query tile assignment with the (window, head) (0: fixed, A/B)
#endif
#ifndef LA_WIN_HALF
#define LA_WIN_HALF 1       // SAM windows: the last key tile runs its lower 32 slots only when the upper 32 are padding (0: full tile, A/B)
#endif
#ifndef LA_ATTN_X
#define LA_ATTN_X 0         // timing experiments (results wrong beyond tile 0): 2 no maximum pass after the first tile, 4 exp2 of the bare score, 8 no row sums
#endif
#include "../../include/la_hip.h"

namespace la {

constexpr int HD = 64;
constexpr int KV_STAGE = 2 * 64 * HD * 2;  // K tile + V^T tile, 16 KiB
constexpr float NEG_BIG = -1.0e30f;
constexpr float RESCALE_THR = 8.0f;   // log2 units

// ------------------------------------------------------------------------------------------------
template <typename T, int NH>
__global__ __launch_bounds__(256) void relpos_kernel(const T* __restrict__ qkv, int B, int heads, int G, int E,
                                                     const T* __restrict__ tabh, const T* __restrict__ tabw,
                                                     float* __restrict__ relh, float* __restrict__ relw) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= B * heads * G) return;
  const int lane = threadIdx.x & 63, fr = lane & 31, fh = lane >> 5;
  const int y = w % G, bh = w / G, h = bh % heads, b = bh / heads;
  const int T_ = G * G;
  const int ntx = (G + 31) >> 5;
  const uint4 zero = make_uint4(0, 0, 0, 0);

  constexpr int HDT = 64 * NH, KS = 4 * NH;      // head dim (64 or 128) and MFMA k-slices over it
  uint4 af[2][KS];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int x = ti * 32 + fr;
    const bool ok = (ti < ntx) && (x < G);
    const T* p = qkv + ((size_t)b * T_ + (size_t)y * G + (ok ? x : 0)) * (3 * E) + h * HDT + fh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) af[ti][ks] = ok ? *reinterpret_cast<const uint4*>(p + ks * 16) : zero;
  }
  const size_t obase = ((size_t)bh * T_ + (size_t)y * G) * G;

  // relh[x][kh] = q_x . Rh[y - kh + G - 1]
  for (int tj = 0; tj < ntx; ++tj) {
    const int j = tj * 32 + fr;
    const bool okj = j < G;
    const T* p = tabh + (size_t)(okj ? (y - j + G - 1) : 0) * HDT + fh * 8;
    uint4 wf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = okj ? *reinterpret_cast<const uint4*>(p + ks * 16) : zero;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      if (ti >= ntx) continue;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = Half16<T>::mfma32(af[ti][ks], wf[ks], acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int x = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (x < G && okj) relh[obase + (size_t)x * G + j] = acc[r];
      }
    }
  }
  // U[x][r'] = q_x . Rw[r'];  relw[x][kw] = U[x][x - kw + G - 1]
  const int nrel = 2 * G - 1;
  const int ntr = (nrel + 31) >> 5;
  for (int tj = 0; tj < ntr; ++tj) {
    const int rp = tj * 32 + fr;
    const bool okr = rp < nrel;
    const T* p = tabw + (size_t)(okr ? rp : 0) * HDT + fh * 8;
    uint4 wf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = okr ? *reinterpret_cast<const uint4*>(p + ks * 16) : zero;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      if (ti >= ntx) continue;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = Half16<T>::mfma32(af[ti][ks], wf[ks], acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int x = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        const int kw = x + G - 1 - rp;
        if (x < G && okr && kw >= 0 && kw < G) relw[obase + (size_t)x * G + kw] = acc[r];
      }
    }
  }
}


// max / sum of a value over the lane pair (l, l ^ 32): one v_permlane32_swap (both halves end up holding both values) instead of a
// ds_bpermute round trip through the LDS queue in the middle of every key tile's dependency chain
__device__ __forceinline__ float xhalf_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#ifdef LA_DEBUG
// phase timeline of the window instance of attn_fwd_kernel (MODE 5): s_memtime of the workgroup in the MIDDLE of the launch (steady state),
// [wave][i]; tools/win_phases.py
__device__ unsigned long long g_win_stamps[4 * 16];
#define WIN_STAMP(i)                                                                                  \
  do {                                                                                                \
    if (MODE == 5 && blockIdx.x == gridDim.x / 2) {                                                   \
      unsigned long long t_;                                                                          \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
      if ((threadIdx.x & 63) == 0) g_win_stamps[(threadIdx.x >> 6) * 16 + (i)] = t_;                   \
    }                                                                                                 \
  } while (0)
#else
#define WIN_STAMP(i) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------
struct AttnArgs {
  const void* qkv;
  const void* vt;
  void* out;
  const float* relh;
  const float* relw;
  const void* tabh;   // MODE 3: 16-bit rel-pos tables [(2G-1), 64]
  const void* tabw;
  int B, heads, T, Tpad, G, E;
  float scale;
  float* lse;         // optional [B*heads, Tpad] (row stride Tpad): log2-domain log-sum-exp of every query row (la_attn_fwd_lse: the training forward)
  float* cspart;      // optional [B * ceil(T / 128), E]: column sums of the 16-bit output rows of every 128-query block (the token means of
  int csH, csW;       // the proj operand, LamEngine mean planes); WIN16 with csH x csW = the image's token grid: padded window rows left out
  // la_attn_fwd_rows (VROW instances): vt == nullptr, V tiles are staged ROW-MAJOR from the v columns of qkv like the K tiles and reach
  // the MFMA through ds_read_b64_tr_b16.  WIN16 with imgH > 0: q | k | v rows and the output are in IMAGE order ([images, imgH, imgW]
  // tokens; B = images * windows per image); a window token beyond the image reads ``padrow`` ([3E]: q | k | v of a pad-after-norm
  // token = the qkv bias) and its output row does not exist.
  int imgH, imgW;
  const void* padrow;
  // WIN16: divisors of the index arithmetic as multiply-high constants (set_win_magic).  A window workgroup lives ~15 us, and the ~10
  // run-time integer divisions on its way to the first load (float-reciprocal sequences of ~20 dependent instructions each, through
  // v_readfirstlane for the wave-uniform ones) were a fifth of that (profiles/r06_window_phases.log).
  int nwx, nwin;                                   // windows per image row / per image (image order)
  unsigned mg_heads, mg_nwin, mg_nwx, mg_G;
};

// n / d for 0 <= n, n * d < 2^32, with magic = ceil(2^32 / d) from the host (d == 1: magic unused): one s_mul_hi_u32 / v_mul_hi_u32
__device__ __forceinline__ int udiv_magic(int n, int d, unsigned magic) { return d == 1 ? n : (int)__umulhi((unsigned)n, magic); }
static inline unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }
static inline void set_win_magic(AttnArgs& a) {
  a.nwx = a.imgH > 0 ? (a.imgW + a.G - 1) / a.G : 1;
  a.nwin = a.imgH > 0 ? a.nwx * ((a.imgH + a.G - 1) / a.G) : 1;
  a.mg_heads = magic_of(a.heads);
  a.mg_nwin = magic_of(a.nwin);
  a.mg_nwx = magic_of(a.nwx);
  a.mg_G = magic_of(a.G);
}

// MODE 0: no bias.  MODE 1: rel-pos, generic G (LDS tables filled from la_relpos_terms output).
// MODE 2: rel-pos, G == 64 (tile == key row), terms from la_relpos_terms.  MODE 4: same with the terms computed in-kernel.
// MODE 3: rel-pos, G <= 16 (SAM windows): the decomposed terms are computed
// IN the kernel (U[r][q] = R[r] . q on MFMA, 8 extra MFMAs per 32-query tile) - no la_relpos_terms pass, no global bias.
// NH = head_dim / 64 (1 or 2).  A 128-wide head is two 64-wide halves everywhere: the K tile is two [64 keys][64 dims]
// sub-tiles, the V^T tile two [64 dims][64 keys] sub-tiles (same 128-byte rows, same swizzle, same DMA pieces), Q has
// 4 NH MFMA k-slices and O^T 2 NH accumulator tiles.  (Heads that are not a multiple of 64 wide - SAM ViT-H has 80 - are
// zero-padded to the next multiple by the host when the weights are packed.)
// VROW: V tiles row-major [64 keys][64 dims] from the v columns of qkv (no V^T copy anywhere): 16-byte chunk c of key row r sits at slot
// c ^ 4 ((r >> 1) & 1), and the A operand of O^T += V^T P^T - lane (d, fh): keys 8 fh .. 8 fh + 7 of dimension d - is two transpose
// reads: in a 16-lane group lane 4 j + c fetches the 8 bytes (row k0 + j, dims D + 4 c ..), lane p receives dimension D + p of the four
// rows (probed on the part: tools/micro/tr_probe.hip).  A 32-lane pass covers 4 rows x 64 bytes = every bank once.
template <typename T, int MODE, int NH, bool VROW = false>
__global__ __launch_bounds__(256, (((MODE == 4 || MODE == 5) && NH == 1) ? 3 : 2)) void attn_fwd_kernel(AttnArgs a) {
  constexpr int HDT = 64 * NH, KS = 4 * NH, SUB = 64 * 64 * 2, KVS = KV_STAGE * NH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WIN_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int BH = a.B * a.heads;
  // Workgroup -> (image*head, query block).  Workgroup b runs on XCD b % 8; the ~64 workgroups an XCD has in flight must
  // share their K / V^T (1 MiB per image-head at T = 4096) or every tile streams from the fabric (PMC: 6 GB FETCH per
  // 16-image launch with bh fastest).  So inside one XCD the query block runs fastest: 64 resident workgroups = 2 heads.
  int bh, qblk;
  {
    const int nq = (a.T + 127) / 128;
    if (MODE == 5) {             // (windows: T <= 256 = one or two query blocks - shifts and multiply-highs, no run-time division)
      const bool two = nq == 2;
      if ((BH & 7) == 0) {
        const int idx = blockIdx.x >> 3;
        qblk = two ? (idx & 1) : 0;
        bh = ((two ? idx >> 1 : idx) << 3) + (blockIdx.x & 7);
      } else {
        qblk = (int)blockIdx.x >= BH;
        bh = blockIdx.x - (qblk ? BH : 0);
      }
    } else if ((BH & 7) == 0) {
      const int idx = blockIdx.x >> 3;
      qblk = idx % nq;
      bh = (idx / nq) * 8 + (blockIdx.x & 7);
    } else {
      bh = blockIdx.x % BH;
      qblk = blockIdx.x / BH;
    }
  }
  const int b = MODE == 5 ? udiv_magic(bh, a.heads, a.mg_heads) : bh / a.heads;
  const int h = bh - b * a.heads;
  const int T_ = a.T, E3 = 3 * a.E;
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const T* vt = reinterpret_cast<const T*>(a.vt);
  // (windows: wave w works on query tile (w + bh) % 4 of the block - T = 196 leaves the second block's fourth tile empty, and a fixed
  // assignment puts that hole on the same SIMD every time)
  const int q0 = qblk * 128 + ((MODE == 5 && LA_WIN_ROT) ? ((wave + bh) & 3) : wave) * 32;
  const int q = q0 + fr;
  const int qc = min(q, T_ - 1);
  const float inv_scale = 1.0f / a.scale;
  const float c2 = a.scale * 1.44269504088896340736f;  // logits -> log2 domain
  // image-order windows (MODE 5, imgH > 0): window b = (image, wy, wx); token (ty, tx) of the window is image token (wy G + ty, wx G + tx)
  const bool img_order = MODE == 5 && VROW && a.imgH > 0;
  int wimg = 0, wy0 = 0, wx0 = 0;
  if (img_order) {
    wimg = udiv_magic(b, a.nwin, a.mg_nwin);
    const int w_ = b - wimg * a.nwin;
    const int wy = udiv_magic(w_, a.nwx, a.mg_nwx);
    wy0 = wy * a.G;
    wx0 = (w_ - wy * a.nwx) * a.G;
  }
  // windows: the lane's query as window coordinates
  const int qy5 = MODE == 5 ? udiv_magic(qc, a.G, a.mg_G) : 0, qx5 = qc - qy5 * a.G;
  // row of window token (ty, tx) in the image-order buffers, or -1 beyond the image
  auto img_row = [&](int ty, int tx) -> long {
    const int y = wy0 + ty, x = wx0 + tx;
    return (y < a.imgH && x < a.imgW) ? ((long)wimg * a.imgH + y) * a.imgW + x : -1;
  };
  long qrow = (long)b * T_ + qc;                       // row of this lane's query in qkv / out
  if (img_order) qrow = img_row(qy5, qx5);
  const T* padrow = reinterpret_cast<const T*>(a.padrow);

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q][ks*16 + fh*8 .. +8] -----------------
  uint4 qf[KS];
  {
    const T* p = (qrow >= 0 ? qkv + (size_t)qrow * E3 : padrow) + h * HDT + fh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(p + ks * 16);
  }

  // ---- K / V^T tile staging by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 tile rows per wave instruction) ----------
  // The DMA destination is lane-linear, so the XOR swizzle goes on the per-lane SOURCE chunk (row = lane/8, slot = lane%8
  // holds logical chunk slot ^ ((row>>1)&7)).  No staging registers: hipcc parked the register-staged variant in
  // scratch and exposed the whole load latency every tile.
  const T* ksrc[2];
  const T* vsrc[2];
  int krow[2], koff[2], voff[2];                             // (VROW: element offsets of the lane's K / V chunk inside a qkv row)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (i * 4 + wave) * 8 + (lane >> 3);        // tile row 0..63
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    krow[i] = row;
    koff[i] = a.E + h * HDT + chunk * 8;
    voff[i] = 2 * a.E + h * HDT + ((lane & 7) ^ (4 * ((row >> 1) & 1))) * 8;
    ksrc[i] = qkv + (size_t)b * T_ * E3 + koff[i];
    vsrc[i] = VROW ? qkv + (size_t)b * T_ * E3 + voff[i] : vt + ((size_t)bh * HDT + row) * a.Tpad + chunk * 8;
  }
  const unsigned lds0 = lds_addr_of(smem);
  // plain / global modes: the lane's K (and V) chunk of tile j is 64 rows behind tile j - 1's - RUNNING pointers (one 64-bit add per piece
  // and tile) instead of clamp + 64-bit multiply-add per piece (22 -> 8 vector instructions per tile beside ~150 of softmax; dma() is
  // called for j = 0, 1, 2, ... in order).  Only a tile that reaches beyond the last row takes the clamped form.
  // (as 32-bit byte offsets from the image's q | k | v block, a wave-uniform base: dma16s - one address register per piece)
  const T* img_base = qkv + (size_t)b * T_ * E3;
  unsigned krun[2], vrun[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    krun[i] = (unsigned)(((size_t)krow[i] * E3 + koff[i]) * sizeof(T));
    vrun[i] = (unsigned)(((size_t)krow[i] * E3 + voff[i]) * sizeof(T));
  }
  const unsigned tile_step = (unsigned)((size_t)64 * E3 * sizeof(T));
  auto dma = [&](int j, int stage) {
    const unsigned sk = lds0 + stage * KVS;
    const unsigned sv = sk + NH * SUB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = 0;
      const T* kp = nullptr;
      const T* vp = nullptr;
      if (MODE != 5 && !LA_ATTN_DMA_RUN) {      // (A/B build: clamp + 64-bit multiply-add per piece, the form before round 6)
        key = min(j * 64 + krow[i], T_ - 1);
        kp = ksrc[i] + (size_t)key * E3;
        if (VROW) vp = vsrc[i] + (size_t)key * E3;
      } else if (MODE != 5) {
        unsigned ko = krun[i], vo = vrun[i];
        if (j * 64 + 64 > T_) {          // (wave-uniform: the last, ragged tile) rows beyond T - 1 re-read row T - 1, masked later
          const unsigned back = (unsigned)max(j * 64 + krow[i] - (T_ - 1), 0) * (unsigned)(E3 * sizeof(T));
          ko -= back;
          vo -= back;
        }
        krun[i] += tile_step;
        vrun[i] += tile_step;
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
          dma16s(img_base, ko + hh * 128, sk + hh * SUB + (i * 4 + wave) * 1024);
          if (VROW) dma16s(img_base, vo + hh * 128, sv + hh * SUB + (i * 4 + wave) * 1024);
          else dma16(vsrc[i] + (size_t)hh * 64 * a.Tpad + j * 64, sv + hh * SUB + (i * 4 + wave) * 1024);
        }
        continue;
      } else if (MODE == 5) {   // slot -> token of the ws x ws window (padded slots read a clamped, later masked, row)
        const int slot = j * 64 + krow[i];
        const int ty = min(slot >> 4, a.G - 1), tx = min(slot & 15, a.G - 1);
        key = ty * a.G + tx;
        if (img_order) {
          const long r = img_row(ty, tx);
          const T* rp = r >= 0 ? qkv + (size_t)r * E3 : padrow;
          kp = rp + koff[i];
          vp = rp + voff[i];
        }
      }
      if (MODE == 5 && !img_order) {
        kp = ksrc[i] + (size_t)key * E3;
        if (VROW) vp = vsrc[i] + (size_t)key * E3;
      }
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {            // dims 64 hh .. of K (and of the V rows), rows 64 hh .. of V^T
        dma16(kp + hh * 64, sk + hh * SUB + (i * 4 + wave) * 1024);
        if (VROW) dma16(vp + hh * 64, sv + hh * SUB + (i * 4 + wave) * 1024);
        else dma16(vsrc[i] + (size_t)hh * 64 * a.Tpad + j * 64, sv + hh * SUB + (i * 4 + wave) * 1024);
      }
    }
  };

  // the first K / V^T tile goes on its way BEFORE the bias tables are built: a window's key loop is four tiles long, and two thirds of
  // that kernel's time were the per-workgroup prologue chain (loads -> table MFMAs -> LDS -> first tile) - the stages and the bias
  // tables do not overlap in LDS
  const int ntiles = (MODE == 5) ? ((16 * a.G + 63) >> 6) : ((T_ + 63) >> 6);
  // windows: the last tile's upper 32 key slots are all padding (G = 14: key rows 14 / 15) - their scores carry NEG_BIG, exp2 gives exactly 0,
  // so the tile runs its lower half only and no bit changes
  const bool half_last = MODE == 5 && LA_WIN_HALF && ((16 * a.G) & 63) != 0 && ((16 * a.G) & 63) <= 32;
  dma(0, 0);
  WIN_STAMP(1);

  // ---- bias staging ---------------------------------------------------------------------------------
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * KVS);
  f32x16 rw[2];
  float bw8[8];
  float* my_bh = nullptr;
  float* my_bw = nullptr;
  const int* keyinfo = nullptr;
  if (MODE == 2) {
    my_bh = bias_lds + wave * 32 * 65;
    for (int i = 0; i < 32; ++i) {
      const int qi = min(q0 + i, T_ - 1);
      my_bh[i * 65 + lane] = a.relh[((size_t)bh * T_ + qi) * 64 + lane] * inv_scale;
    }
    const float* p = a.relw + ((size_t)bh * T_ + qc) * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 v = *reinterpret_cast<const float4*>(p + t * 32 + 8 * g4 + 4 * fh);
        rw[t][g4 * 4 + 0] = v.x * inv_scale;
        rw[t][g4 * 4 + 1] = v.y * inv_scale;
        rw[t][g4 * 4 + 2] = v.z * inv_scale;
        rw[t][g4 * 4 + 3] = v.w * inv_scale;   // (vector element writes with constant indices stay in registers)
      }
  } else if (MODE == 5) {
    // windows in 16-wide slot order: U tables as in MODE 3, then the bias of score register (t, r) in tile j is
    //   bh4[2t + (r >> 3)]  (key row 4j + 2t + (r>>3))  +  bw8[((r >> 2) & 1) * 4 + (r & 3)]  (key column 8((r>>2)&1) + 4fh + (r&3)),
    // with padded rows / columns carrying NEG_BIG (no separate masking pass).
    // One 32 x 33 table per wave, used twice: first for U_w (only needed to pick the lane's 8 column terms bw8), then
    // overwritten with U_h, which the key loop reads - 48.5 KiB per workgroup, three workgroups per CU.
    const int G = a.G, nrel = 2 * G - 1;
    my_bh = bias_lds + wave * 32 * 33;
    const T* th = reinterpret_cast<const T*>(a.tabh) + (size_t)min(fr, nrel - 1) * HDT + fh * 8;
    const T* tw = reinterpret_cast<const T*>(a.tabw) + (size_t)min(fr, nrel - 1) * HDT + fh * 8;
    f32x16 uh, uw;
#pragma unroll
    for (int r = 0; r < 16; ++r) uh[r] = uw[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uh = Half16<T>::mfma32(*reinterpret_cast<const uint4*>(th + ks * 16), qf[ks], uh);
      uw = Half16<T>::mfma32(*reinterpret_cast<const uint4*>(tw + ks * 16), qf[ks], uw);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) my_bh[fr * 33 + (r & 3) + 8 * (r >> 2) + 4 * fh] = uw[r] * inv_scale;
    __builtin_amdgcn_wave_barrier();
    const int qx = qx5;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kw = 8 * (i >> 2) + 4 * fh + (i & 3);
      bw8[i] = (kw < G) ? my_bh[fr * 33 + qx + G - 1 - kw] : NEG_BIG;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) my_bh[fr * 33 + (r & 3) + 8 * (r >> 2) + 4 * fh] = uh[r] * inv_scale;
    __builtin_amdgcn_wave_barrier();
  } else if (MODE == 4) {
    // G == 64, terms computed in-kernel.  The wave's 32 queries share the image row y and cover columns x0 .. x0+31.
    //   relw[q][kw] = q . Rw[x - kw + 63] = Uw[(x - x0) + 63 - kw][q],  Uw[i][q] = Rw[x0 + i] . q,  i < 96   (12 MFMAs)
    //   relh[q][j]  = q . Rh[y - j + 63]  = Uh[63 - j][q],              Uh[i][q] = Rh[y + i] . q,   i < 64   ( 8 MFMAs)
    // Uw tiles are bounced through a per-wave LDS scratch to reach the score-accumulator register layout.
    // Only HALF of the relh row (32 tiles) is kept in LDS at a time - the second half is recomputed at tile 32 (4 MFMAs) -
    // so a workgroup needs 48.5 KiB instead of 65 and three of them (12 waves) fit a CU.
    my_bh = bias_lds + wave * 32 * 33;
    const int x0 = q0 & 63;
    const T* tabw = reinterpret_cast<const T*>(a.tabw);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) rw[t][r] = 0.f;
#pragma unroll 1
    for (int tt = 0; tt < 3; ++tt) {
      const T* tp = tabw + (size_t)min(x0 + tt * 32 + fr, 126) * HDT + fh * 8;
      f32x16 u;
#pragma unroll
      for (int r = 0; r < 16; ++r) u[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) u = Half16<T>::mfma32(*reinterpret_cast<const uint4*>(tp + ks * 16), qf[ks], u);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 16; ++r) my_bh[fr * 33 + (r & 3) + 8 * (r >> 2) + 4 * fh] = u[r] * inv_scale;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kw = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
          const int i = fr + 63 - kw - tt * 32;
          if (i >= 0 && i < 32) rw[t][r] = my_bh[fr * 33 + i];
        }
    }
    __builtin_amdgcn_wave_barrier();
  } else if (MODE == 1) {
    const int G = a.G, GS = G + 1;
    my_bh = bias_lds + wave * 2 * 32 * GS;
    my_bw = my_bh + 32 * GS;
    for (int idx = lane; idx < 32 * G; idx += 64) {
      const int i = idx / G, k = idx % G;
      const size_t src = ((size_t)bh * T_ + min(q0 + i, T_ - 1)) * G + k;
      my_bh[i * GS + k] = a.relh[src] * inv_scale;
      my_bw[i * GS + k] = a.relw[src] * inv_scale;
    }
    int* ki = reinterpret_cast<int*>(bias_lds + 4 * 2 * 32 * GS);
    for (int k = tid; k < a.Tpad; k += 256) ki[k] = ((k / G) << 16) | (k % G);
    keyinfo = ki;
  } else if (MODE == 3) {
    // U_h[q][r] = q . Rh[r], U_w[q][r] = q . Rw[r] for r < 2G-1 <= 32; bias(q, k) = U_h[q][qy - kh + G-1] + U_w[q][qx - kw + G-1]
    const int G = a.G, nrel = 2 * G - 1;
    my_bh = bias_lds + wave * 2 * 32 * 33;
    my_bw = my_bh + 32 * 33;
    const T* th = reinterpret_cast<const T*>(a.tabh) + (size_t)min(fr, nrel - 1) * HDT + fh * 8;
    const T* tw = reinterpret_cast<const T*>(a.tabw) + (size_t)min(fr, nrel - 1) * HDT + fh * 8;
    f32x16 uh, uw;
#pragma unroll
    for (int r = 0; r < 16; ++r) uh[r] = uw[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uh = Half16<T>::mfma32(*reinterpret_cast<const uint4*>(th + ks * 16), qf[ks], uh);
      uw = Half16<T>::mfma32(*reinterpret_cast<const uint4*>(tw + ks * 16), qf[ks], uw);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {      // accumulator: column = query fr, row = table row
      const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
      my_bh[fr * 33 + row] = uh[r] * inv_scale;
      my_bw[fr * 33 + row] = uw[r] * inv_scale;
    }
    int* ki = reinterpret_cast<int*>(bias_lds + 4 * 2 * 32 * 33);
    for (int k = tid; k < a.Tpad; k += 256) {
      const int kk = min(k, T_ - 1);       // padded keys are masked later; keep their lookups in range
      ki[k] = ((G - 1 - kk / G) << 16) | (G - 1 - kk % G);
    }
    keyinfo = ki;
  }

  f32x16 oacc[2 * NH];
#pragma unroll
  for (int d = 0; d < 2 * NH; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;
  // VROW: byte offsets of this lane's transpose reads inside a V stage, for the two 32-dimension blocks (+ ks * 2048, + 512 for keys 4 .. 7)
  unsigned vtr[2] = {0u, 0u};
  if (VROW) {
    const int j = (lane & 15) >> 2, c = lane & 3, jb = (j >> 1) & 1, gd = (lane >> 4) & 1;
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) vtr[dd] = (unsigned)((8 * fh + j) * 128 + ((4 * (dd ^ jb) + 2 * gd + (c >> 1)) << 4) + (c & 1) * 8);
  }

  WIN_STAMP(2);
  dma_wait<0>();
  // MODE 4: relh[q][j] = Uh[63 - j][q], Uh[i][q] = Rh[y + i] . q; half hf covers tiles j in [32 hf, 32 hf + 32) = rows
  // i in [32 (1 - hf), +32): my_bh[q][j & 31]
  auto fill_relh_half = [&](int hf) {
    const int y = q0 >> 6;
    const T* tp = reinterpret_cast<const T*>(a.tabh) + (size_t)min(y + (1 - hf) * 32 + fr, 126) * HDT + fh * 8;
    f32x16 u;
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) u = Half16<T>::mfma32(*reinterpret_cast<const uint4*>(tp + ks * 16), qf[ks], u);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) my_bh[fr * 33 + 31 - ((r & 3) + 8 * (r >> 2) + 4 * fh)] = u[r] * inv_scale;
    __builtin_amdgcn_wave_barrier();
  };
  if (MODE == 4) fill_relh_half(0);
  __syncthreads();
  WIN_STAMP(3);
  // a wave whose 32 query rows all lie beyond T (T = 901: three of the 32 waves of an image-head) only helps staging the tiles
  const bool idle_wave = q0 >= T_;
  for (int j = 0; j < ntiles; ++j) {
#if !(LA_ATTN_ABL & 8)
    if (j + 1 < ntiles) dma(j + 1, (j + 1) & 1);
#endif
    if (idle_wave) {
      dma_wait<0>();
      __syncthreads();
      continue;
    }
    const char* sk = smem + (j & 1) * KVS;
    const char* sv = sk + NH * SUB;
    if (MODE == 4 && j == 32) fill_relh_half(1);

    // ---- S^T tile: 64 keys x 32 queries.  G == 64: the C operand of the first MFMA IS the relw register block
    // (no per-tile initialisation pass); the per-row scalar relh[q][tile] is folded into the softmax constants below.
    f32x16 s[2];
    float rh = 0.f;
    const bool half = MODE == 5 && half_last && j == ntiles - 1;      // (wave-uniform)
    if (MODE == 2) rh = my_bh[fr * 65 + j];
    if (MODE == 4) rh = my_bh[fr * 33 + (j & 31)];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (MODE == 5 && t == 1 && half) continue;
      const uint4 kf0 = *reinterpret_cast<const uint4*>(sk + swz_off(t * 32 + fr, fh));
      if (MODE == 2 || MODE == 4) {
        s[t] = Half16<T>::mfma32(kf0, qf[0], rw[t]);
      } else if (MODE == 5) {
        f32x16 z;
        const int qy = qy5;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int kh = 4 * j + 2 * t + hh;
          const float bh = (kh < a.G) ? my_bh[fr * 33 + qy + a.G - 1 - kh] : NEG_BIG;
#pragma unroll
          for (int i = 0; i < 8; ++i) z[hh * 8 + i] = bh + bw8[i];
        }
        s[t] = Half16<T>::mfma32(kf0, qf[0], z);
      } else {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        s[t] = Half16<T>::mfma32(kf0, qf[0], z);
      }
    }
#if !(LA_ATTN_ABL & 2)
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (MODE == 5 && t == 1 && half) continue;
        const uint4 kf = *reinterpret_cast<const uint4*>(sk + (ks >> 2) * SUB + swz_off(t * 32 + fr, (ks & 3) * 2 + fh));
        s[t] = Half16<T>::mfma32(kf, qf[ks], s[t]);
      }
    }
#endif
    if (MODE == 1) {
      const int GS = a.G + 1;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
          const int info = keyinfo[key];
          s[t][r] += my_bh[fr * GS + (info >> 16)] + my_bw[fr * GS + (info & 0xffff)];
        }
    }
    if (MODE == 3) {
      const int qy = qc / a.G, qx = qc % a.G;
      const float* bh_q = my_bh + fr * 33 + qy;
      const float* bw_q = my_bw + fr * 33 + qx;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
          const int info = keyinfo[key];
          s[t][r] += bh_q[info >> 16] + bw_q[info & 0xffff];
        }
    }
    if (MODE != 5 && j * 64 + 64 > T_) {  // tail tile: mask keys >= T (wave-uniform branch)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
          if (key >= T_) s[t][r] = NEG_BIG;
        }
    }

    // ---- online softmax, one query row per lane pair (lane, lane ^ 32) -----------------------------------
    // Scores live in "raw" units (logit / scale); true score = s + rh.  The running max is only advanced when some row
    // of the wave outgrows it by more than RESCALE_THR (in log2 units): P then stays <= 2^THR, which fp16/bf16 hold at
    // full relative precision, and the O / l rescale pass disappears from almost every tile.
#if LA_ATTN_X & 2
    if (j == 0) {
#endif
    float mx = s[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (MODE == 5 && t == 1 && half) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
    }
    mx = xhalf_max(mx) + rh;
    if (!__all((mx - m_run) * c2 <= RESCALE_THR)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 2 * NH; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
      m_run = m_new;
    }
#if LA_ATTN_X & 2
    }
#endif
    const float mc = (rh - m_run) * c2;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (MODE == 5 && t == 1 && half) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // (scalar FMAs on purpose: v_pk_fma_f32 does not overlap with another wave's MFMA stream at all, v_fma_f32 partly, v_exp_f32
        // fully - tools/probes/coissue.hip; the packed form of this loop measured 0.7 % slower)
#if LA_ATTN_ABL & 1
        const float p = fmaf(s[t][r], c2, mc);
#else
#if LA_ATTN_X & 4
        const float p = __builtin_amdgcn_exp2f(s[t][r]);
#else
        const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], c2, mc));   // raw v_exp_f32 (no denormal fix-up)
#endif
#endif
        s[t][r] = p;
#if !(LA_ATTN_X & 8)
        psum += p;
#endif
      }
    }
    l_run += psum;

    // ---- P^T fragments: lane (q, fh) needs keys ks*16 + fh*8 .. +8 -------------------------------------------
    uint4 pf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (MODE == 5 && ks >= 2 && half) continue;
      const int t = ks >> 1, g0 = (ks & 1) * 8;  // register group base (4 regs per group)
      const uint32_t x0 = pack2<T>(s[t][g0 + 0], s[t][g0 + 1]);
      const uint32_t x1 = pack2<T>(s[t][g0 + 2], s[t][g0 + 3]);
      const uint32_t y0 = pack2<T>(s[t][g0 + 4], s[t][g0 + 5]);
      const uint32_t y1 = pack2<T>(s[t][g0 + 6], s[t][g0 + 7]);
      const auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
      pf[ks] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }

    // ---- O^T += V^T P^T -----------------------------------------------------------------------------------
#if LA_ATTN_ABL & 4
    asm volatile("" ::"v"(pf[0].x), "v"(pf[0].y), "v"(pf[0].z), "v"(pf[0].w), "v"(pf[1].x), "v"(pf[1].y), "v"(pf[1].z), "v"(pf[1].w));
    asm volatile("" ::"v"(pf[2].x), "v"(pf[2].y), "v"(pf[2].z), "v"(pf[2].w), "v"(pf[3].x), "v"(pf[3].y), "v"(pf[3].z), "v"(pf[3].w));
#else
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (MODE == 5 && ks >= 2 && half) continue;
#pragma unroll
      for (int d = 0; d < 2 * NH; ++d) {
        uint4 vf;
        if (VROW) {
          typedef short s16x4 __attribute__((ext_vector_type(4)));
          typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
          const unsigned va = lds_addr_of(sv) + (d >> 1) * SUB + vtr[d & 1] + ks * 2048;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)va);
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(va + 512));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          vf = make_uint4(l2.x, l2.y, h2.x, h2.y);
        } else {
          vf = *reinterpret_cast<const uint4*>(sv + (d >> 1) * SUB + swz_off((d & 1) * 32 + fr, ks * 2 + fh));
        }
        oacc[d] = Half16<T>::mfma32(vf, pf[ks], oacc[d]);
      }
    }
#endif

#if !(LA_ATTN_ABL & 8)
    if (j < 4) WIN_STAMP(8 + j);      // (the tile's own work is done; what follows is the wait for the next tile + the barrier)
    dma_wait<0>();     // next tile (issued before this tile's MFMAs) has landed for this wave ...
    __syncthreads();   // ... and for all waves; orders the stage swap
#endif
    if (j < 4) WIN_STAMP(4 + j);
  }

  // ---- normalise and store: lane holds O[q][d*32 + 8*g + 4*fh + 0..3] ---------------------------------------
  const float l_tot = xhalf_sum(l_run);
  const float inv_l = 1.0f / l_tot;
  if (a.lse != nullptr && q < T_ && fh == 0) a.lse[(size_t)bh * a.Tpad + q] = m_run * c2 + __builtin_amdgcn_logf(l_tot);   // v_log_f32 = log2
  if (q < T_ && qrow >= 0) {
    T* op = reinterpret_cast<T*>(a.out) + (size_t)qrow * a.E + h * HDT;
#pragma unroll
    for (int d = 0; d < 2 * NH; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 v;
        v.x = pack2<T>(oacc[d][g4 * 4 + 0] * inv_l, oacc[d][g4 * 4 + 1] * inv_l);
        v.y = pack2<T>(oacc[d][g4 * 4 + 2] * inv_l, oacc[d][g4 * 4 + 3] * inv_l);
        *reinterpret_cast<uint2*>(op + d * 32 + 8 * g4 + 4 * fh) = v;
      }
  }
  WIN_STAMP(12);
  if (a.cspart != nullptr) {
    // Column sums of the block's stored rows ON THE MATRIX PIPE: a wave writes its 32 normalised 16-bit rows into LDS in the V tile's
    // layout (the K / V stages are idle: the key loop ended with a barrier) and runs the P . V step on them with P = 1 -
    // sum_k V^T[d][k] = the column sums of exactly the stored values, accumulated in fp32.  8 NH LDS writes, 4 NH transpose-read pairs and
    // 4 NH MFMAs per wave instead of 32 NH four-step DPP sums (128 vector instructions of a window workgroup's ~1200: with the DPP form the
    // window blocks' sums cost as much as the la_colmean16 pass they replace).  The four waves' sums meet in LDS, fixed order.
    bool valid = q < T_ && qrow >= 0;
    if (MODE == 5 && a.csH > 0 && !img_order) {
      const int nwx = (a.csW + a.G - 1) / a.G, nwy = (a.csH + a.G - 1) / a.G;
      const int w = b % (nwx * nwy);
      valid = valid && (w / nwx) * a.G + qy5 < a.csH && (w % nwx) * a.G + qx5 < a.csW;
    }
    char* tile = smem + wave * (NH * 4096);                    // [NH sub-tiles][32 rows][128 B], chunk c of row r at slot c ^ 4 ((r >> 1) & 1)
#pragma unroll
    for (int d = 0; d < 2 * NH; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 v;
        v.x = pack2<T>(oacc[d][g4 * 4 + 0] * inv_l, oacc[d][g4 * 4 + 1] * inv_l);
        v.y = pack2<T>(oacc[d][g4 * 4 + 2] * inv_l, oacc[d][g4 * 4 + 3] * inv_l);
        if (!valid) v = make_uint2(0u, 0u);
        const int c16 = 4 * (d & 1) + g4;                      // dims 32 (d & 1) + 8 g4 + 4 fh .. + 3 of the sub-tile
        *reinterpret_cast<uint2*>(tile + (d >> 1) * 4096 + fr * 128 + ((c16 ^ (4 * ((fr >> 1) & 1))) << 4) + 8 * fh) = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    unsigned ctr[2];
    {
      const int j = (lane & 15) >> 2, c = lane & 3, jb = (j >> 1) & 1, gd = (lane >> 4) & 1;
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) ctr[dd] = (unsigned)((8 * fh + j) * 128 + ((4 * (dd ^ jb) + 2 * gd + (c >> 1)) << 4) + (c & 1) * 8);
    }
    const uint32_t one2 = pack2<T>(1.0f, 1.0f);
    const uint4 ones = make_uint4(one2, one2, one2, one2);
    float* red = reinterpret_cast<float*>(smem + 4 * NH * 4096);
#pragma unroll
    for (int d = 0; d < 2 * NH; ++d) {
      f32x16 cs;
#pragma unroll
      for (int r = 0; r < 16; ++r) cs[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
        const unsigned va = lds_addr_of(tile) + (d >> 1) * 4096 + ctr[d & 1] + ks * 2048;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)va);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(va + 512));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        cs = Half16<T>::mfma32(make_uint4(l2.x, l2.y, h2.x, h2.y), ones, cs);
      }
      if (fr == 0) {                                           // (every column of the product is the same sum) lane (0, fh): dims + 4 fh
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * HDT + d * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh] = cs[r];
      }
    }
    __syncthreads();
    if (tid < HDT) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += red[w * HDT + tid];
      const int nq = (T_ + 127) / 128;
      a.cspart[((size_t)b * nq + qblk) * a.E + h * HDT + tid] = t;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// SAM window attention (T = ws^2 <= 256 tokens, e.g. 196): latency / bandwidth-bound, so a different shape:
// ONE WAVE per (window, head, 32-query tile), no LDS staging of K / V and no workgroup barriers in the key loop - the
// MFMA A operands (K rows, V^T rows) are fetched straight from global memory (the 7 query tiles of a window-head hit
// the same lines in L1/L2).  16 independent waves per CU hide the load latency.  Rel-pos terms are computed in the
// prologue (U[r][q] = R[r] . q, 8 MFMAs) and bounced through a small per-wave LDS table, as in MODE 3 above.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 3) void attn_window_kernel(AttnArgs a, int qtiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int G = a.G, T_ = a.T, E3 = 3 * a.E;
  // key -> (G-1-kh, G-1-kw) lookup, shared by the 4 waves
  int* keyinfo = reinterpret_cast<int*>(smem);
  for (int k = tid; k < a.Tpad; k += 256) {
    const int kk = min(k, T_ - 1);
    keyinfo[k] = ((G - 1 - kk / G) << 16) | (G - 1 - kk % G);
  }
  __syncthreads();
  const int job = blockIdx.x * 4 + wave;
  const int njobs = a.B * a.heads * qtiles;
  if (job >= njobs) return;
  const int qt = job % qtiles, bh = job / qtiles;
  const int h = bh % a.heads, b = bh / a.heads;
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const T* vt = reinterpret_cast<const T*>(a.vt);
  const int q0 = qt * 32, q = q0 + fr, qc = min(q, T_ - 1);
  const float inv_scale = 1.0f / a.scale;
  const float c2 = a.scale * 1.44269504088896340736f;

  uint4 qf[4];
  {
    const T* p = qkv + ((size_t)b * T_ + qc) * E3 + h * HD + fh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(p + ks * 16);
  }
  // ---- decomposed rel-pos terms -> per-wave LDS tables [32 queries][33] --------------------------------------
  float* my_bh = reinterpret_cast<float*>(smem + a.Tpad * sizeof(int)) + wave * 2 * 32 * 33;
  float* my_bw = my_bh + 32 * 33;
  {
    const int nrel = 2 * G - 1;
    const T* th = reinterpret_cast<const T*>(a.tabh) + (size_t)min(fr, nrel - 1) * HD + fh * 8;
    const T* tw = reinterpret_cast<const T*>(a.tabw) + (size_t)min(fr, nrel - 1) * HD + fh * 8;
    f32x16 uh, uw;
#pragma unroll
    for (int r = 0; r < 16; ++r) uh[r] = uw[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uh = Half16<T>::mfma32(*reinterpret_cast<const uint4*>(th + ks * 16), qf[ks], uh);
      uw = Half16<T>::mfma32(*reinterpret_cast<const uint4*>(tw + ks * 16), qf[ks], uw);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
      my_bh[fr * 33 + row] = uh[r] * inv_scale;
      my_bw[fr * 33 + row] = uw[r] * inv_scale;
    }
    __builtin_amdgcn_wave_barrier();
  }
  const float* bh_q = my_bh + fr * 33 + qc / G;
  const float* bw_q = my_bw + fr * 33 + qc % G;

  const T* kbase = qkv + (size_t)b * T_ * E3 + a.E + h * HD + fh * 8;      // + key * E3 + ks * 16
  const T* vbase = vt + ((size_t)bh * HD + fr) * a.Tpad + fh * 8;          // + d_half * 32 * Tpad + key0 + ks * 16
  f32x16 oacc[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;
  const int ntiles = (T_ + 63) >> 6;
  for (int j = 0; j < ntiles; ++j) {
    // ---- S^T = K Q^T (A operand straight from global) -----------------------------------------------------
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const T* kp = kbase + (size_t)min(j * 64 + t * 32 + fr, T_ - 1) * E3;
      uint4 kf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const uint4*>(kp + ks * 16);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[t] = Half16<T>::mfma32(kf[ks], qf[ks], s[t]);
    }
    // V^T fragments for this tile: issue the loads now, they land while the softmax runs
    uint4 vf[2][4];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) vf[d][ks] = *reinterpret_cast<const uint4*>(vbase + (size_t)d * 32 * a.Tpad + j * 64 + ks * 16);
    // ---- bias + mask ------------------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = j * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        const int info = keyinfo[key];
        const float sv = s[t][r] + bh_q[info >> 16] + bw_q[info & 0xffff];
        s[t][r] = (key < T_) ? sv : NEG_BIG;
      }
    float mx = s[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
    mx = xhalf_max(mx);
    if (!__all((mx - m_run) * c2 <= RESCALE_THR)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
      m_run = m_new;
    }
    const float mc = -m_run * c2;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], c2, mc));
        s[t][r] = p;
        psum += p;
      }
    l_run += psum;
    uint4 pf[4];
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
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int d = 0; d < 2; ++d) oacc[d] = Half16<T>::mfma32(vf[d][ks], pf[ks], oacc[d]);
  }
  const float l_tot = xhalf_sum(l_run);
  const float inv_l = 1.0f / l_tot;
  if (q < T_) {
    T* op = reinterpret_cast<T*>(a.out) + ((size_t)b * T_ + q) * a.E + h * HD;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 v;
        v.x = pack2<T>(oacc[d][g4 * 4 + 0] * inv_l, oacc[d][g4 * 4 + 1] * inv_l);
        v.y = pack2<T>(oacc[d][g4 * 4 + 2] * inv_l, oacc[d][g4 * 4 + 3] * inv_l);
        *reinterpret_cast<uint2*>(op + d * 32 + 8 * g4 + 4 * fh) = v;
      }
  }
}

template <typename T>
static void launch_window(const AttnArgs& a, hipStream_t st) {
  const int qtiles = (a.T + 31) / 32;
  const int njobs = a.B * a.heads * qtiles;
  const size_t lds = (size_t)a.Tpad * sizeof(int) + 4 * 2 * 32 * 33 * sizeof(float);
  hipLaunchKernelGGL((attn_window_kernel<T>), dim3((njobs + 3) / 4), dim3(256), lds, st, a, qtiles);
}

template <typename T, int MODE, int NH, bool VROW = false>
static void launch_attn_nh(const AttnArgs& a, size_t lds, hipStream_t st) {
  // raise the dynamic-LDS limit to the most any request of this variant can need (160 KiB), once per device
  static unsigned long long attr_mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(attn_fwd_kernel<T, MODE, NH, VROW>), 160 * 1024, attr_mask);
  const int nq = (a.T + 127) / 128;
  hipLaunchKernelGGL((attn_fwd_kernel<T, MODE, NH, VROW>), dim3(nq * a.B * a.heads), dim3(256), lds, st, a);
}

// extra = LDS beyond the two K / V^T stages (bias tables); head_dim = E / heads is 64 or 128
template <typename T, int MODE, bool VROW = false>
static void launch_attn(const AttnArgs& a, size_t extra, hipStream_t st) {
  if (a.E == a.heads * 128) launch_attn_nh<T, MODE, 2, VROW>(a, 2 * (size_t)KV_STAGE * 2 + extra, st);
  else launch_attn_nh<T, MODE, 1, VROW>(a, 2 * (size_t)KV_STAGE + extra, st);
}


// ------------------------------------------------------------------------------------------------
// fp8 QK^T variant of the plain attention (BASELINE configs[4]: "fp8 MFMA attention"; opt-in, la_attn_fwd_fp8).
// Scores S^T = K Q^T on v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 operands, unit block scales): ONE 64-deep instruction per 32 x 32
// score tile at twice the 16-bit MFMA rate instead of four 16-deep ones, K tiles of 4 KiB instead of 8.  Softmax and O^T = V^T P^T
// are exactly the 16-bit kernel's (P in 16 bit, fp32 accumulate).  Q and K arrive as e4m3 bytes [B*T, 2E] (la_qk_fp8).  Both
// operands are loaded the same way - lane (row, half) holds bytes [32 half, 32 half + 32) of its row - so whatever order the
// instruction gives the 64 k-slots, q and k elements of equal head dimension meet.
// ------------------------------------------------------------------------------------------------
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int swz64b_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd_fp8_kernel(AttnArgs a, const unsigned char* __restrict__ qk8) {
  constexpr int K8 = 64 * 64, VT = 64 * 64 * 2, STG = K8 + VT;      // K tile (fp8) 4 KiB + V^T tile (16 bit) 8 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int BH = a.B * a.heads;
  const int bh = blockIdx.x % BH, qblk = blockIdx.x / BH;
  const int h = bh % a.heads, b = bh / a.heads;
  const int T_ = a.T, E2 = 2 * a.E;
  const T* vt = reinterpret_cast<const T*>(a.vt);
  const int q0 = qblk * 128 + wave * 32, q = q0 + fr, qc = min(q, T_ - 1);
  const float c2 = a.scale * 1.44269504088896340736f;

  i32x8 qf;
  {
    const uint4* p = reinterpret_cast<const uint4*>(qk8 + ((size_t)b * T_ + qc) * E2 + h * 64 + fh * 32);
    const uint4 lo = p[0], hi = p[1];
    qf = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
  }
  const unsigned lds0 = lds_addr_of(smem);
  const unsigned char* kbase = qk8 + (size_t)b * T_ * E2 + a.E + h * 64;
  auto dma = [&](int j, int stage) {
    const unsigned sk = lds0 + stage * STG, sv = sk + K8;
    {   // K tile: 64 rows x 64 bytes = 4 pieces of 16 rows, one per wave
      const int row = wave * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      dma16(kbase + (size_t)min(j * 64 + row, T_ - 1) * E2 + chunk * 16, sk + wave * 1024);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // V^T tile as in attn_fwd_kernel
      const int row = (i * 4 + wave) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      dma16(vt + ((size_t)bh * 64 + row) * a.Tpad + chunk * 8 + j * 64, sv + (i * 4 + wave) * 1024);
    }
  };
  f32x16 oacc[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;
  const int ntiles = (T_ + 63) >> 6;
  dma(0, 0);
  dma_wait<0>();
  __syncthreads();
  for (int j = 0; j < ntiles; ++j) {
    if (j + 1 < ntiles) dma(j + 1, (j + 1) & 1);
    const char* sk = smem + (j & 1) * STG;
    const char* sv = sk + K8;
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint4 lo = *reinterpret_cast<const uint4*>(sk + swz64b_off(t * 32 + fr, fh * 2));
      const uint4 hi = *reinterpret_cast<const uint4*>(sk + swz64b_off(t * 32 + fr, fh * 2 + 1));
      const i32x8 kf{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
      f32x16 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
      s[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf, z, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    if (j * 64 + 64 > T_) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
          if (key >= T_) s[t][r] = NEG_BIG;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
    mx = xhalf_max(mx);
    if (!__all((mx - m_run) * c2 <= RESCALE_THR)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
      m_run = m_new;
    }
    const float mc = -m_run * c2;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], c2, mc));
        s[t][r] = p;
        psum += p;
      }
    l_run += psum;
    uint4 pf[4];
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
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const uint4 vf = *reinterpret_cast<const uint4*>(sv + swz_off(d * 32 + fr, ks * 2 + fh));
        oacc[d] = Half16<T>::mfma32(vf, pf[ks], oacc[d]);
      }
    dma_wait<0>();
    __syncthreads();
  }
  const float l_tot = xhalf_sum(l_run);
  const float inv_l = 1.0f / l_tot;
  if (q < T_) {
    T* op = reinterpret_cast<T*>(a.out) + ((size_t)b * T_ + q) * a.E + h * 64;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 v;
        v.x = pack2<T>(oacc[d][g4 * 4 + 0] * inv_l, oacc[d][g4 * 4 + 1] * inv_l);
        v.y = pack2<T>(oacc[d][g4 * 4 + 2] * inv_l, oacc[d][g4 * 4 + 3] * inv_l);
        *reinterpret_cast<uint2*>(op + d * 32 + 8 * g4 + 4 * fh) = v;
      }
  }
}

// q | k columns of qkv16 [rows, 3E] -> e4m3 bytes [rows, 2E] (v_cvt_pk_fp8_f32: OCP e4m3 with saturation on gfx950)
template <typename T>
__global__ __launch_bounds__(256) void qk_fp8_kernel(const T* __restrict__ qkv, long rows, int E, unsigned char* __restrict__ out) {
  const int per_row = (2 * E) / 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * per_row; i += (long)gridDim.x * 256) {
    const long r = i / per_row;
    const int c = (int)(i % per_row) * 8;
    const uint4 v = *reinterpret_cast<const uint4*>(qkv + r * 3 * E + c);
    const T* e = reinterpret_cast<const T*>(&v);
    unsigned w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)e[0], (float)e[1], w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)e[2], (float)e[3], w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)e[4], (float)e[5], w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)e[6], (float)e[7], w1, true);
    *reinterpret_cast<uint2*>(out + r * 2 * E + c) = make_uint2(w0, w1);
  }
}

}  // namespace la

extern "C" int la_relpos_terms(const void* qkv, int B, int heads, int G, int E, const void* tabh, const void* tabw, float* relh,
                               float* relw, int dt, void* stream) {
  LA_CHECK_ARG(qkv && tabh && tabw && relh && relw, "la_relpos_terms: null pointer");
  LA_CHECK_ARG(B > 0 && heads > 0 && G > 0 && G <= 64 && (E == heads * 64 || E == heads * 128),
               "la_relpos_terms: needs head_dim 64 or 128 and G <= 64 (G=%d E=%d heads=%d)", G, E, heads);
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16, "la_relpos_terms: bad dtype %d", dt);
  const int waves = B * heads * G;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((waves + 3) / 4), blk(256);
  const bool wide = E == heads * 128;
  if (dt == LA_F16) {
    if (wide) hipLaunchKernelGGL((la::relpos_kernel<la::f16_t, 2>), grid, blk, 0, st, (const la::f16_t*)qkv, B, heads, G, E, (const la::f16_t*)tabh, (const la::f16_t*)tabw, relh, relw);
    else hipLaunchKernelGGL((la::relpos_kernel<la::f16_t, 1>), grid, blk, 0, st, (const la::f16_t*)qkv, B, heads, G, E, (const la::f16_t*)tabh, (const la::f16_t*)tabw, relh, relw);
  } else {
    if (wide) hipLaunchKernelGGL((la::relpos_kernel<la::bf16_t, 2>), grid, blk, 0, st, (const la::bf16_t*)qkv, B, heads, G, E, (const la::bf16_t*)tabh, (const la::bf16_t*)tabw, relh, relw);
    else hipLaunchKernelGGL((la::relpos_kernel<la::bf16_t, 1>), grid, blk, 0, st, (const la::bf16_t*)qkv, B, heads, G, E, (const la::bf16_t*)tabh, (const la::bf16_t*)tabw, relh, relw);
  }
  LA_CHECK_LAUNCH("la_relpos_terms");
  return 0;
}

extern "C" int la_attn_fwd_cs(const void* qkv, const void* vt, void* out16, const float* relh, const float* relw, const void* tabh,
                              const void* tabw, int B, int heads, int T, int Tpad, int G, int E, float scale, int mode, float* cspart, int csH,
                              int csW, int dt, void* stream);

#ifdef LA_DEBUG
extern "C" int la_dbg_win_stamps(unsigned long long* host_out) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(la::g_win_stamps), sizeof(unsigned long long) * 64);
}
#endif

extern "C" int la_attn_fwd(const void* qkv, const void* vt, void* out16, const float* relh, const float* relw, const void* tabh,
                           const void* tabw, int B, int heads, int T, int Tpad, int G, int E, float scale, int mode, int dt, void* stream) {
  return la_attn_fwd_cs(qkv, vt, out16, relh, relw, tabh, tabw, B, heads, T, Tpad, G, E, scale, mode, nullptr, 0, 0, dt, stream);
}

extern "C" int la_attn_fwd_cs(const void* qkv, const void* vt, void* out16, const float* relh, const float* relw, const void* tabh,
                              const void* tabw, int B, int heads, int T, int Tpad, int G, int E, float scale, int mode, float* cspart, int csH,
                              int csW, int dt, void* stream) {
  LA_CHECK_ARG(qkv && vt && out16, "la_attn_fwd: null pointer");
  LA_CHECK_ARG(B > 0 && heads > 0 && T > 0 && (E == heads * 64 || E == heads * 128),
               "la_attn_fwd: needs head_dim 64 or 128 - pad other widths with zero columns (E=%d heads=%d)", E, heads);
  LA_CHECK_ARG(Tpad >= T && (Tpad % 64) == 0, "la_attn_fwd: Tpad=%d must be a multiple of 64 covering T=%d", Tpad, T);
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16, "la_attn_fwd: bad dtype %d", dt);
  la::AttnArgs a{qkv, vt, out16, relh, relw, tabh, tabw, B, heads, T, Tpad, G, E, scale, nullptr, cspart, csH, csW};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t kv = 0;       // launch_attn adds the K / V^T stages for the head width; the sizes below are the bias tables
  if (mode == LA_ATTN_PLAIN) {
    if (dt == LA_F16) la::launch_attn<la::f16_t, 0>(a, kv, st);
    else la::launch_attn<la::bf16_t, 0>(a, kv, st);
  } else if (mode == LA_ATTN_RELPOS_WIN16) {
    LA_CHECK_ARG(tabh && tabw && G > 0 && G <= 16 && G * G == T && Tpad >= 16 * G,
                 "la_attn_fwd: WIN16 needs the tables, T == G*G, G <= 16 and Tpad >= 16*G (T=%d G=%d Tpad=%d)", T, G, Tpad);
    const size_t lds = kv + 4 * 32 * 33 * sizeof(float);
    la::set_win_magic(a);
    if (dt == LA_F16) la::launch_attn<la::f16_t, 5>(a, lds, st);
    else la::launch_attn<la::bf16_t, 5>(a, lds, st);
  } else if (mode == LA_ATTN_RELPOS) {
    LA_CHECK_ARG(G > 0 && G <= 64 && G * G == T, "la_attn_fwd: rel-pos needs T == G*G, G <= 64 (T=%d G=%d)", T, G);
    if (tabh && tabw && G <= 16) {        // windows: bias terms computed in-kernel from the tables
      static const char* wforce = la_dbg_env("LA_WINDOW_PATH");     // debugging: "lds" selects the LDS-staged MODE 3 kernel
      if (!(wforce && wforce[0] == 'l') && E == heads * 64) {   // (the no-LDS window kernel is 64-wide only)
        if (dt == LA_F16) la::launch_window<la::f16_t>(a, st);
        else la::launch_window<la::bf16_t>(a, st);
        LA_CHECK_LAUNCH("la_attn_fwd");
        return 0;
      }
      const size_t lds = kv + 4 * 2 * 32 * 33 * sizeof(float) + (size_t)Tpad * sizeof(int);
      if (dt == LA_F16) la::launch_attn<la::f16_t, 3>(a, lds, st);
      else la::launch_attn<la::bf16_t, 3>(a, lds, st);
      LA_CHECK_LAUNCH("la_attn_fwd");
      return 0;
    }
    if (tabh && tabw && G == 64) {        // global blocks: terms computed in the prologue of each query tile
      const size_t lds = kv + 4 * 32 * 33 * sizeof(float);
      if (dt == LA_F16) la::launch_attn<la::f16_t, 4>(a, lds, st);
      else la::launch_attn<la::bf16_t, 4>(a, lds, st);
      LA_CHECK_LAUNCH("la_attn_fwd");
      return 0;
    }
    LA_CHECK_ARG(relh && relw, "la_attn_fwd: rel-pos terms missing (run la_relpos_terms, or pass the tables for G <= 16 / G == 64)");
    if (G == 64) {
      const size_t lds = kv + 4 * 32 * 65 * sizeof(float);
      if (dt == LA_F16) la::launch_attn<la::f16_t, 2>(a, lds, st);
      else la::launch_attn<la::bf16_t, 2>(a, lds, st);
    } else {
      const size_t lds = kv + 4 * 2 * 32 * (G + 1) * sizeof(float) + (size_t)Tpad * sizeof(int);
      if (dt == LA_F16) la::launch_attn<la::f16_t, 1>(a, lds, st);
      else la::launch_attn<la::bf16_t, 1>(a, lds, st);
    }
  } else {
    LA_CHECK_ARG(false, "la_attn_fwd: bad mode %d", mode);
  }
  LA_CHECK_LAUNCH("la_attn_fwd");
  return 0;
}

// The same attention without a V^T copy: V tiles come row-major from the v columns of qkv (ds_read_b64_tr_b16 feeds the MFMA).  Modes:
// LA_ATTN_PLAIN; LA_ATTN_RELPOS with the 16-bit tables and G == 64 (terms in-kernel); LA_ATTN_RELPOS_WIN16 (G <= 16) - with imgH > 0 the
// windows are addressed in IMAGE order: qkv / out16 are [images * imgH * imgW] token rows, B = images * ceil(imgH / G) * ceil(imgW / G)
// windows, padrow = [3E] 16-bit q | k | v of a padded token (the qkv bias: pad-after-norm, image_encoder.py:160-172), no window buffers.
extern "C" int la_attn_fwd_rows(const void* qkv, void* out16, const void* tabh, const void* tabw, int B, int heads, int T, int Tpad, int G, int E,
                                float scale, int mode, float* cspart, int imgH, int imgW, const void* padrow, int dt, void* stream) {
  LA_CHECK_ARG(qkv && out16, "la_attn_fwd_rows: null pointer");
  LA_CHECK_ARG(B > 0 && heads > 0 && T > 0 && (E == heads * 64 || E == heads * 128),
               "la_attn_fwd_rows: needs head_dim 64 or 128 - pad other widths with zero columns (E=%d heads=%d)", E, heads);
  LA_CHECK_ARG(Tpad >= T && (Tpad % 64) == 0, "la_attn_fwd_rows: Tpad=%d must be a multiple of 64 covering T=%d", Tpad, T);
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16, "la_attn_fwd_rows: bad dtype %d", dt);
  la::AttnArgs a{qkv, nullptr, out16, nullptr, nullptr, tabh, tabw, B, heads, T, Tpad, G, E, scale, nullptr, cspart, 0, 0, 0, 0, nullptr};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (mode == LA_ATTN_PLAIN) {
    if (dt == LA_F16) la::launch_attn<la::f16_t, 0, true>(a, 0, st);
    else la::launch_attn<la::bf16_t, 0, true>(a, 0, st);
  } else if (mode == LA_ATTN_RELPOS_WIN16) {
    LA_CHECK_ARG(tabh && tabw && G > 0 && G <= 16 && G * G == T && Tpad >= 16 * G,
                 "la_attn_fwd_rows: WIN16 needs the tables, T == G*G, G <= 16 and Tpad >= 16*G (T=%d G=%d Tpad=%d)", T, G, Tpad);
    if (imgH > 0) {
      const int nw = ((imgH + G - 1) / G) * ((imgW + G - 1) / G);
      LA_CHECK_ARG(imgW > 0 && padrow && B % nw == 0, "la_attn_fwd_rows: image-order windows need imgW, padrow and B = images * %d windows", nw);
      a.imgH = imgH;
      a.imgW = imgW;
      a.padrow = padrow;
    }
    la::set_win_magic(a);
    const size_t lds = 4 * 32 * 33 * sizeof(float);
    if (dt == LA_F16) la::launch_attn<la::f16_t, 5, true>(a, lds, st);
    else la::launch_attn<la::bf16_t, 5, true>(a, lds, st);
  } else if (mode == LA_ATTN_RELPOS) {
    LA_CHECK_ARG(tabh && tabw && G == 64 && T == 4096, "la_attn_fwd_rows: rel-pos form needs the tables and the 64 x 64 grid (G=%d T=%d)", G, T);
    const size_t lds = 4 * 32 * 33 * sizeof(float);
    if (dt == LA_F16) la::launch_attn<la::f16_t, 4, true>(a, lds, st);
    else la::launch_attn<la::bf16_t, 4, true>(a, lds, st);
  } else {
    LA_CHECK_ARG(false, "la_attn_fwd_rows: bad mode %d", mode);
  }
  LA_CHECK_LAUNCH("la_attn_fwd_rows");
  return 0;
}

extern "C" int la_attn_fwd_lse(const void* qkv, const void* vt, void* out16, float* lse, int B, int heads, int T, int Tpad, int E, float scale,
                               int dt, void* stream) {
  LA_CHECK_ARG(qkv && vt && out16 && lse, "la_attn_fwd_lse: null pointer");
  LA_CHECK_ARG(B > 0 && heads > 0 && T > 0 && (E == heads * 64 || E == heads * 128),
               "la_attn_fwd_lse: needs head_dim 64 or 128 - other widths zero-padded (E=%d heads=%d)", E, heads);
  LA_CHECK_ARG(Tpad >= T && (Tpad % 64) == 0, "la_attn_fwd_lse: Tpad=%d must be a multiple of 64 covering T=%d", Tpad, T);
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16, "la_attn_fwd_lse: bad dtype %d", dt);
  la::AttnArgs a{qkv, vt, out16, nullptr, nullptr, nullptr, nullptr, B, heads, T, Tpad, 0, E, scale, lse};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dt == LA_F16) la::launch_attn<la::f16_t, 0>(a, 0, st);
  else la::launch_attn<la::bf16_t, 0>(a, 0, st);
  LA_CHECK_LAUNCH("la_attn_fwd_lse");
  return 0;
}

extern "C" int la_attn_fwd_relpos_lse(const void* qkv, const void* vt, void* out16, const float* relh, const float* relw, float* lse, int B, int heads,
                                      int T, int Tpad, int G, int E, float scale, int dt, void* stream) {
  LA_CHECK_ARG(qkv && vt && out16 && relh && relw && lse, "la_attn_fwd_relpos_lse: null pointer");
  LA_CHECK_ARG(B > 0 && heads > 0 && T > 0 && (E == heads * 64 || E == heads * 128),
               "la_attn_fwd_relpos_lse: needs head_dim 64 or 128 - other widths zero-padded (E=%d heads=%d)", E, heads);
  LA_CHECK_ARG(Tpad >= T && (Tpad % 64) == 0, "la_attn_fwd_relpos_lse: Tpad=%d must be a multiple of 64 covering T=%d", Tpad, T);
  LA_CHECK_ARG(G > 0 && G <= 64 && G * G == T, "la_attn_fwd_relpos_lse: rel-pos needs T == G*G, G <= 64 (T=%d G=%d)", T, G);
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16, "la_attn_fwd_relpos_lse: bad dtype %d", dt);
  la::AttnArgs a{qkv, vt, out16, relh, relw, nullptr, nullptr, B, heads, T, Tpad, G, E, scale, lse};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (G == 64) {
    const size_t lds = 4 * 32 * 65 * sizeof(float);
    if (dt == LA_F16) la::launch_attn<la::f16_t, 2>(a, lds, st);
    else la::launch_attn<la::bf16_t, 2>(a, lds, st);
  } else {
    const size_t lds = 4 * 2 * 32 * (G + 1) * sizeof(float) + (size_t)Tpad * sizeof(int);
    if (dt == LA_F16) la::launch_attn<la::f16_t, 1>(a, lds, st);
    else la::launch_attn<la::bf16_t, 1>(a, lds, st);
  }
  LA_CHECK_LAUNCH("la_attn_fwd_relpos_lse");
  return 0;
}

extern "C" int la_qk_fp8(const void* qkv, long rows, int E, void* qk8, int dt, void* stream) {
  LA_CHECK_ARG(qkv && qk8 && rows > 0 && E > 0 && (E % 8) == 0 && (dt == LA_F16 || dt == LA_BF16), "la_qk_fp8: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long n = rows * (2 * E / 8);
  const dim3 grid((unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256)), blk(256);
  if (dt == LA_F16) hipLaunchKernelGGL(la::qk_fp8_kernel<la::f16_t>, grid, blk, 0, st, (const la::f16_t*)qkv, rows, E, (unsigned char*)qk8);
  else hipLaunchKernelGGL(la::qk_fp8_kernel<la::bf16_t>, grid, blk, 0, st, (const la::bf16_t*)qkv, rows, E, (unsigned char*)qk8);
  LA_CHECK_LAUNCH("la_qk_fp8");
  return 0;
}

extern "C" int la_attn_fwd_fp8(const void* qk8, const void* vt, void* out16, int B, int heads, int T, int Tpad, int E, float scale, int dt,
                               void* stream) {
  LA_CHECK_ARG(qk8 && vt && out16, "la_attn_fwd_fp8: null pointer");
  LA_CHECK_ARG(B > 0 && heads > 0 && T > 0 && E == heads * 64, "la_attn_fwd_fp8: needs head_dim 64 (E=%d heads=%d)", E, heads);
  LA_CHECK_ARG(Tpad >= T && (Tpad % 64) == 0, "la_attn_fwd_fp8: Tpad=%d must be a multiple of 64 covering T=%d", Tpad, T);
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16, "la_attn_fwd_fp8: bad dtype %d", dt);
  la::AttnArgs a{nullptr, vt, out16, nullptr, nullptr, nullptr, nullptr, B, heads, T, Tpad, 0, E, scale, nullptr};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nblk = (T + 127) / 128 * B * heads;
  constexpr int LDS = 2 * (64 * 64 + 64 * 64 * 2);
  if (dt == LA_F16) hipLaunchKernelGGL(la::attn_fwd_fp8_kernel<la::f16_t>, dim3(nblk), dim3(256), LDS, st, a, (const unsigned char*)qk8);
  else hipLaunchKernelGGL(la::attn_fwd_fp8_kernel<la::bf16_t>, dim3(nblk), dim3(256), LDS, st, a, (const unsigned char*)qk8);
  LA_CHECK_LAUNCH("la_attn_fwd_fp8");
  return 0;
}
