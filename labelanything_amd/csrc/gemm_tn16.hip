// Weight gradient of an nn.Linear on the 16-bit MFMA WITHOUT transposed operand copies (gfx950):
//   dW[N, K] += dY[R, N]^T . X[R, K]        (and db[N] += column sums of dY)
// autograd of `x @ W^T + b` in label_anything/models/common.py:19-37, image_encoder.py:200-255 and transformers' ViT layers
// (build_encoder.py:83-100) when the backbone trains (models/lam.py:321-347).  Both operands are ROW-major over the reduction index (the
// token rows R) - the layout the forward / backward passes leave them in.  Rounds 3 - 4 made token-contiguous copies (la_transpose16, 98
// launches = 5.6 ms per cfg3 training step) for a split-K la_gemm; here the [64 rows][256 columns] tiles are staged row-major by LDS-DMA
// and BOTH MFMA operands come out of them with ds_read_b64_tr_b16 (the LDS transpose read the attention kernels use for V / K^T / Q^T):
// a lane receives 4 consecutive reduction rows of one column per read, two reads = the 8-element k-slice of v_mfma_f32_32x32x16.
//
// One workgroup = 4 waves x 512 registers (one wave per SIMD), a 256 (n) x 256 (k) tile of dW, wave (wn, wk) its 128 x 128 quadrant = 16
// accumulator tiles in the AGPRs; the reduction runs over one CHUNK of the rows in 64-row steps (two LDS stages of 64 KiB), the chunks of a
// tile add into dW with fp32 atomics (split over R exactly like la_gemm's ksplit path; the caller zero-fills or pre-loads dW).
// LDS tile: 64 rows of 512 B, the eight 64-byte segments of row r stored at segment s ^ (r & 7): a 32-lane pass of the transpose read
// (4 rows x 64 B) covers every bank once.  db: the workgroups of k-tile 0 sum their dY fragments on the vector ALU beside the MFMAs
// (v_dot2 with ones, co-issued) - no second pass over dY.
#include "la_common.h"
#include "../../include/la_hip.h"

namespace la {

constexpr int TN_TILE_B = 64 * 512;         // one [64 rows][256 columns] 16-bit tile
constexpr int TN_STAGE_B = 2 * TN_TILE_B;   // dY tile | X tile

struct GemmTn16Args {
  const void* dy;
  const void* x;
  float* dw;
  float* db;
  int ldy, ldx, lddw, R, N, K;
  int gsize, gstride;      // output row n lands in dW row (n / gsize) * gstride + n % gsize (gsize == 0: n) - la_gemm's LA_MAP_GROUP
  int S, chunk_rows;       // number of row chunks, rows per chunk (a multiple of 64)
};

template <typename T> __device__ __forceinline__ float dot2_ones(uint32_t v, float acc);
template <> __device__ __forceinline__ float dot2_ones<f16_t>(uint32_t v, float acc) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 ones = {(_Float16)1.0f, (_Float16)1.0f};
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, v), ones, acc, false);
}
template <> __device__ __forceinline__ float dot2_ones<bf16_t>(uint32_t v, float acc) {
  return acc + __builtin_bit_cast(float, v << 16) + __builtin_bit_cast(float, v & 0xffff0000u);
}

template <typename T>
__global__ __launch_bounds__(256, 1) void gemm_tn16_kernel(GemmTn16Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int fh = lane >> 5;
  const int nt_n = a.N >> 8, nt_k = a.K >> 8, ntiles = nt_n * nt_k;
  // workgroup b runs on XCD b % 8; xcd_remap hands every XCD a contiguous run of the (chunk, tile) sequence: the tiles of one row chunk
  // sit on ONE XCD next to each other, so the chunk's dY / X rows are fetched from HBM once and served to the other tiles by that XCD's L2
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int chunk = lin / ntiles, tile = lin % ntiles;
  const int tn = tile / nt_k, tk = tile % nt_k;           // (k-tile fastest: the workgroups that share a dY tile are neighbours)
  const int r_begin = chunk * a.chunk_rows, r_end = min(a.R, r_begin + a.chunk_rows);
  if (r_begin >= r_end) return;
  const int nsteps = (r_end - r_begin + 63) >> 6;
  const T* dyb = reinterpret_cast<const T*>(a.dy) + tn * 256;
  const T* xb = reinterpret_cast<const T*>(a.x) + tk * 256;
  const unsigned lds0 = lds_addr_of(smem);

  // LDS-DMA: a 1 KiB piece = 2 rows x 512 B; lane l fills 16-byte slot l & 31 of row 2 piece + (l >> 5) with the source chunk whose
  // swizzled position that slot is (slot ^ ((row & 7) << 2)): 32 pieces per tile, wave w takes pieces w, w + 4, ...
  auto dma = [&](int step, int stage) {
    const unsigned s0 = lds0 + stage * TN_STAGE_B;
    const int row0 = r_begin + step * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int piece = i * 4 + wave, row = piece * 2 + (lane >> 5);
      const int src_chunk = (lane & 31) ^ ((row & 7) << 2);
      const size_t grow = (size_t)min(row0 + row, a.R - 1);
      dma16(dyb + grow * a.ldy + src_chunk * 8, s0 + piece * 1024);
      dma16(xb + grow * a.ldx + src_chunk * 8, s0 + TN_TILE_B + piece * 1024);
    }
  };
  // transpose-read offsets (see attn_bwd.hip tr_offsets): in a 16-lane group lane 4 j + c fetches (row j of 4, columns 4 c .. 4 c + 3 of the
  // group's 16); lane p then holds column p of the 16, rows 0 .. 3.  Lanes 16 - 31: the next 16 columns; lane half fh: rows 8 fh ..
  // tro[h]: rows 8 fh + 4 h + j of every 16-row k-slice (+ ks * 16 * 512 bytes), column block cb (+ swizzled (cb << 6))
  const int j = (lane & 15) >> 2, c = lane & 3, gd = (lane >> 4) & 1;
  unsigned tro[2], rsw[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * fh + 4 * h + j;                   // (row & 7) is the same for every k-slice: 16 ks does not touch the low 3 bits
    tro[h] = (unsigned)(row * 512 + 32 * gd + 8 * c);
    rsw[h] = (unsigned)(row & 7);
  }
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  auto frag = [&](unsigned tile_lds, int cb, int ks) -> uint4 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(tile_lds + tro[0] + (((unsigned)cb ^ rsw[0]) << 6) + ks * 8192));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(tile_lds + tro[1] + (((unsigned)cb ^ rsw[1]) << 6) + ks * 8192));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_db = a.db != nullptr && tk == 0 && wk == 0;

  dma(0, 0);
  dma_wait<0>();
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    if (step + 1 < nsteps) dma(step + 1, (step + 1) & 1);
    const unsigned sy = lds0 + (step & 1) * TN_STAGE_B, sx = sy + TN_TILE_B;
    const int rows_left = r_end - (r_begin + step * 64);          // < 64 only in the chunk's last step
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag(sy, wn * 4 + i, ks);
#pragma unroll
      for (int q = 0; q < 4; ++q) bf[q] = frag(sx, wk * 4 + q, ks);
      if (rows_left < 64) {                    // rows beyond the chunk (clamped reads of real rows) must not count: zero them in dY
        const int cnt = rows_left - (16 * ks + 8 * fh);             // valid ones among this lane's 8 reduction rows
        uint32_t m[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) m[w] = cnt > 2 * w + 1 ? 0xffffffffu : cnt > 2 * w ? 0x0000ffffu : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          af[i].x &= m[0];
          af[i].y &= m[1];
          af[i].z &= m[2];
          af[i].w &= m[3];
        }
      }
      if (do_db) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bsum[i] = dot2_ones<T>(af[i].x, bsum[i]);
          bsum[i] = dot2_ones<T>(af[i].y, bsum[i]);
          bsum[i] = dot2_ones<T>(af[i].z, bsum[i]);
          bsum[i] = dot2_ones<T>(af[i].w, bsum[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][q] = Half16<T>::mfma32(af[i], bf[q], acc[i][q]);
    }
    dma_wait<0>();
    __syncthreads();
  }
  // accumulator register r of lane (fr, fh): row (r & 3) + 8 (r >> 2) + 4 fh of the 32 x 32 tile, column fr
  const int fr = lane & 31;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = tn * 256 + wn * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      const size_t orow = a.gsize > 0 ? (size_t)(n / a.gsize) * a.gstride + n % a.gsize : (size_t)n;
      float* dst = a.dw + orow * a.lddw + tk * 256 + wk * 128 + fr;
#pragma unroll
      for (int q = 0; q < 4; ++q) atomicAdd(dst + q * 32, acc[i][q][r]);
    }
  }
  if (do_db) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);        // the two lane halves hold rows 8 fh .. of every k-slice
      if (fh == 0) atomicAdd(a.db + tn * 256 + wn * 128 + i * 32 + fr, v);
    }
  }
}

}  // namespace la

extern "C" int la_gemm_tn16(const void* dy, int ldy, const void* x, int ldx, float* dw, int lddw, int R, int N, int K, int gsize, int gstride,
                            float* db, int dt, void* stream) {
  LA_CHECK_ARG(dy && x && dw, "la_gemm_tn16: null pointer");
  LA_CHECK_ARG(R > 0 && N > 0 && K > 0 && (N % 256) == 0 && (K % 256) == 0, "la_gemm_tn16: N and K must be multiples of 256 (R=%d N=%d K=%d)", R, N, K);
  LA_CHECK_ARG(ldy >= N && ldx >= K && (ldy % 8) == 0 && (ldx % 8) == 0 && lddw >= K, "la_gemm_tn16: row strides (ldy=%d ldx=%d lddw=%d)", ldy, ldx, lddw);
  LA_CHECK_ARG((reinterpret_cast<uintptr_t>(dy) % 16) == 0 && (reinterpret_cast<uintptr_t>(x) % 16) == 0, "la_gemm_tn16: operands must be 16-byte aligned");
  LA_CHECK_ARG(gsize == 0 || (gsize > 0 && gstride >= gsize && N % gsize == 0), "la_gemm_tn16: bad row grouping (gsize=%d gstride=%d)", gsize, gstride);
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16, "la_gemm_tn16: bad dtype %d", dt);
  const int ntiles = (N / 256) * (K / 256);
  // row chunks: a workgroup owns a whole CU (512 registers per lane), so ONE round of at most 256 workgroups - as many chunks as fit,
  // at least 4 steps of 64 rows each (every chunk ends in a 256 KiB pass of atomics: fewer chunks = less of that)
  const int steps = (R + 63) / 64;
  int S = 256 / ntiles;
  if (S < 1) S = 1;
  while (S > 1 && (steps + S - 1) / S < 4) --S;
  const int chunk_rows = ((steps + S - 1) / S) * 64;
  S = (R + chunk_rows - 1) / chunk_rows;
  la::GemmTn16Args a{dy, x, dw, db, ldy, ldx, lddw, R, N, K, gsize, gstride, S, chunk_rows};
  const int grid = S * ntiles;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static unsigned long long m1 = 0, m2 = 0;
  if (dt == LA_F16) {
    la::ensure_dyn_lds(reinterpret_cast<const void*>(la::gemm_tn16_kernel<la::f16_t>), 2 * la::TN_STAGE_B, m1);
    hipLaunchKernelGGL(la::gemm_tn16_kernel<la::f16_t>, dim3(grid), dim3(256), 2 * la::TN_STAGE_B, st, a);
  } else {
    la::ensure_dyn_lds(reinterpret_cast<const void*>(la::gemm_tn16_kernel<la::bf16_t>), 2 * la::TN_STAGE_B, m2);
    hipLaunchKernelGGL(la::gemm_tn16_kernel<la::bf16_t>, dim3(grid), dim3(256), 2 * la::TN_STAGE_B, st, a);
  }
  LA_CHECK_LAUNCH("la_gemm_tn16");
  return 0;
}
