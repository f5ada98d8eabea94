// la_gemm: C[M,N] = epilogue(A[M,K] . W[N,K]^T) on gfx950 MFMA (32x32x16, f16/bf16 in, fp32 accumulate).
//
// Workgroup = 256 threads = 4 waves (2 x 2), tile 128 x 128 x 64.  Each wave owns a 64 x 64 sub-tile as
// 2 x 2 MFMA 32x32 accumulators (64 acc VGPRs).  Operand tiles are staged global -> registers -> LDS
// (16-B chunks, XOR-swizzled so every ds_read_b128 lane group is bank-conflict free), double buffered with
// ONE barrier per K-step: the next tile's global loads are issued before the MFMAs of the current one and
// written to the other LDS stage after them.  Tiles are handed to XCDs in contiguous chunks (xcd_remap) so
// the workgroups sharing an A row-panel / the whole W hit the same L2.
#include <cstdlib>
#include <type_traits>
#include "la_common.h"
#include "../../include/la_hip.h"
#include "gemm_shared.h"

namespace la {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB


template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const T* __restrict__ A, int lda, const T* __restrict__ Wt, int ldw,
                                                          int M, int N, int K, LaGemmEpilogue e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (N + BN - 1) / BN, ntm = (M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

  // ---- global -> register staging: thread t owns chunk (t & 7) of rows (t >> 3) + 32 i ------------
  const int lc = tid & 7, lr = tid >> 3;
  const T* a_ptr[4];
  const T* w_ptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = a_row(e, min(m0 + lr + 32 * i, M - 1));
    const int rw = min(n0 + lr + 32 * i, N - 1);
    a_ptr[i] = A + (size_t)ra * lda + lc * 8;
    w_ptr[i] = Wt + (size_t)rw * ldw + lc * 8;
  }
  uint4 ra_[4], rw_[4];
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
    const int ka = a_koff(e, k0);
    const bool ok = (k0 + lc * 8) < K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra_[i] = ok ? *reinterpret_cast<const uint4*>(a_ptr[i] + ka) : make_uint4(0, 0, 0, 0);
      rw_[i] = ok ? *reinterpret_cast<const uint4*>(w_ptr[i] + k0) : make_uint4(0, 0, 0, 0);
    }
  };
  auto swrite = [&](int stage) {
    char* sa = smem + stage * STAGE_BYTES;
    char* sw = sa + BM * BK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = lr + 32 * i;
      *reinterpret_cast<uint4*>(sa + swz_off(r, lc)) = ra_[i];
      *reinterpret_cast<uint4*>(sw + swz_off(r, lc)) = rw_[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (K + BK - 1) / BK;
  gload(0);
  swrite(0);
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
    const char* sa = smem + (kt & 1) * STAGE_BYTES;
    const char* sw = sa + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const uint4*>(sa + swz_off(wm * 64 + i * 32 + fr, ks * 2 + fh));
        wf[i] = *reinterpret_cast<const uint4*>(sw + swz_off(wn * 64 + i * 32 + fr, ks * 2 + fh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Half16<T>::mfma32(af[i], wf[j], acc[i][j]);
    }
    if (kt + 1 < nk) swrite((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  const RowMap rm{e.map, e.p0, e.p1, e.p2, e.p3, e.p4};
  T* out16 = reinterpret_cast<T*>(e.out16);
  T* vt = reinterpret_cast<T*>(e.vt);
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const int col = n0 + wn * 64 + tj * 32 + fr;
    if (col >= N) continue;
    int dcol = col, row_add = 0, bcol = col;
    if (e.map == LA_MAP_CONVT2X2) {
      const int kyx = col / e.p2;
      dcol = col % e.p2;
      bcol = dcol;
      row_add = (kyx >> 1) * 2 * e.p0 + (kyx & 1);
    }
    const float bias = e.bias ? e.bias[bcol] : 0.f;
    const bool to_vt = (vt != nullptr) && (col >= e.vt_col0);
    int vhead = 0, vd = 0;
    if (to_vt) {
      const int cv = col - e.vt_col0;
      vhead = cv / e.vt_hd;
      vd = cv % e.vt_hd;
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (row >= M) continue;
        float v = acc[ti][tj][r] + bias;
        if (e.act == LA_ACT_GELU) v = gelu_erf(v);
        else if (e.act == LA_ACT_RELU) v = fmaxf(v, 0.f);
        if (to_vt) {
          const int vr = map_row(rm, row);            // token index in the destination order
          if (vr < 0) continue;
          const int b = vr / e.vt_T, t = vr % e.vt_T;
          vt[((size_t)(b * e.vt_heads + vhead) * e.vt_hd + vd) * e.vt_Tpad + vt_slot(t, e.vt_ws)] = (T)v;
          continue;
        }
        int drow = map_row(rm, row);
        if (drow < 0) continue;
        drow += row_add;
        if (e.res) {
          const int rr = e.res_mod ? drow % e.res_mod : drow;
          v += e.res[(size_t)rr * e.ldr + dcol];
        }
        if (e.out32) e.out32[(size_t)drow * e.ld32 + dcol] = v;
        if (out16) out16[(size_t)drow * e.ld16 + dcol] = (T)v;
      }
    }
  }
}



// -----------------------------------------------------------------------------------------------------------------
// Shared LDS-staged epilogue.  Each wave owns TI x TJ MFMA 32x32 accumulators at (wrow0, wcol0) inside a 128-row chunk
// of the block tile; the chunk is staged as fp32 [128][BN_ + 4] and then every thread handles 8 consecutive columns of
// one row: float4 bias / residual loads, 16-byte (16-bit) or 2 x 16-byte (fp32) stores.  Tiles of the transposed-V
// region are staged column-major instead so that consecutive tokens are contiguous in the V^T buffer.
// -----------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void store8(T* p, const float* v) {
  uint4 o;
  o.x = pack2<T>(v[0], v[1]); o.y = pack2<T>(v[2], v[3]); o.z = pack2<T>(v[4], v[5]); o.w = pack2<T>(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = o;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float* v) {
  reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// Transposed-V store of the 4 consecutive GEMM rows of one V column whose destination rows (token index in the output
// order, -1 = none) are d[0..3].  When the four are consecutive tokens of one batch item whose slots are contiguous they go
// out as one 8-byte store; a run that straddles a window row of the 16-slot layout as two 4-byte pairs; anything else
// (a row map that jumps to another window, the ragged end of M) element by element.
template <typename T>
__device__ __forceinline__ void vt_store4(T* vt, const LaGemmEpilogue& e, const int (&d)[4], int vhead, int vd, float4 v, float bias) {
  const float vv[4] = {v.x + bias, v.y + bias, v.z + bias, v.w + bias};
  auto slot_ptr = [&](int drow) {
    const int bj = drow / e.vt_T, tj = drow % e.vt_T;
    return vt + ((size_t)(bj * e.vt_heads + vhead) * e.vt_hd + vd) * e.vt_Tpad + vt_slot(tj, e.vt_ws);
  };
  const bool run = d[0] >= 0 && d[1] == d[0] + 1 && d[2] == d[0] + 2 && d[3] == d[0] + 3;
  const int t0 = run ? d[0] % e.vt_T : 0;
  if (run && (t0 & 1) == 0 && t0 + 3 < e.vt_T) {
    if (e.vt_ws == 0 || (t0 % e.vt_ws) <= e.vt_ws - 4) {
      store4v<T>(slot_ptr(d[0]), vv[0], vv[1], vv[2], vv[3]);
      return;
    }
    if ((e.vt_ws & 1) == 0) {           // two pairs, each inside one window row
      store2<T>(slot_ptr(d[0]), vv[0], vv[1]);
      store2<T>(slot_ptr(d[2]), vv[2], vv[3]);
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (d[j] >= 0) *slot_ptr(d[j]) = (T)vv[j];
}

template <typename T, int TI, int TJ, int BN_, int NT, int CR = 128>
__device__ __forceinline__ void epilogue_lds(float* epi, const f32x16 (&acc)[TI][TJ], int nchunks, int my_chunk, int wrow0, int wcol0,
                                             int m0, int n0, int M, int N, const LaGemmEpilogue& e, int tid) {
  constexpr int LD = BN_ + 4;
  __shared__ int drow_lds[CR];
  const int lane = tid & 63, fr = lane & 31, fh = lane >> 5;
  const RowMap rm{e.map, e.p0, e.p1, e.p2, e.p3, e.p4};
  T* outT = reinterpret_cast<T*>(e.out16);
  T* vt = reinterpret_cast<T*>(e.vt);
  const bool vt_tile = (vt != nullptr) && (n0 >= e.vt_col0);
#pragma unroll 1
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (chunk > 0) __syncthreads();
    if (my_chunk == chunk) {
      if (!vt_tile) {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              epi[(wrow0 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * LD + wcol0 + tj * 32 + fr] = acc[ti][tj][r];
      } else {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const float4 v = make_float4(acc[ti][tj][g4 * 4], acc[ti][tj][g4 * 4 + 1], acc[ti][tj][g4 * 4 + 2], acc[ti][tj][g4 * 4 + 3]);
              *reinterpret_cast<float4*>(&epi[(wcol0 + tj * 32 + fr) * LD + wrow0 + ti * 32 + 8 * g4 + 4 * fh]) = v;
            }
      }
    }
    // destination row of every GEMM row of this chunk (-1: dropped / beyond M), computed once: the maps cost several
    // integer divisions each and every row is needed by BN_/8 (row-major pass) or BN_ (V^T pass) items
    const int mrow0 = m0 + chunk * CR;
    for (int i = tid; i < CR; i += NT) drow_lds[i] = (mrow0 + i < M) ? map_row(rm, mrow0 + i) : -1;
    __syncthreads();
    if (vt_tile) {
      constexpr int RG = CR / 4;
      for (int it = tid; it < BN_ * RG; it += NT) {
        const int rg = it % RG, c = it / RG;
        const int col = n0 + c;
        if (col >= N) continue;
        const float4 v = *reinterpret_cast<const float4*>(&epi[c * LD + rg * 4]);
        const float bias = e.bias ? e.bias[col] : 0.f;
        const int cv = col - e.vt_col0;
        const int vhead = cv / e.vt_hd, vd = cv % e.vt_hd;
        if (rm.mode != LA_MAP_NONE) {        // mapped rows: destinations from the table
          const int d[4] = {drow_lds[rg * 4], drow_lds[rg * 4 + 1], drow_lds[rg * 4 + 2], drow_lds[rg * 4 + 3]};
          vt_store4<T>(vt, e, d, vhead, vd, v, bias);
          continue;
        }
        const int row = mrow0 + rg * 4;
        if (row >= M) continue;
        const int b = row / e.vt_T, t = row % e.vt_T;
        T* dst = vt + ((size_t)(b * e.vt_heads + vhead) * e.vt_hd + vd) * e.vt_Tpad + vt_slot(t, e.vt_ws);
        if ((e.vt_T & 3) == 0 && row + 3 < M && (e.vt_ws == 0 || ((e.vt_ws & 3) == 0) || (t % e.vt_ws) <= e.vt_ws - 4)) {
          // 4 consecutive tokens of one batch item (and, for windows, of one window row: contiguous slots)
          store4v<T>(dst, v.x + bias, v.y + bias, v.z + bias, v.w + bias);
        } else if ((e.vt_T & 3) == 0 && row + 3 < M && e.vt_ws > 0 && (e.vt_ws & 1) == 0) {
          // group straddles a window row: two token pairs, each inside one row (ws even), 4-byte stores
          store2<T>(dst, v.x + bias, v.y + bias);
          store2<T>(vt + ((size_t)(b * e.vt_heads + vhead) * e.vt_hd + vd) * e.vt_Tpad + vt_slot(t + 2, e.vt_ws), v.z + bias, v.w + bias);
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
          for (int j = 0; j < 4; ++j) {
            const int rj = row + j;
            if (rj >= M) break;
            const int bj = rj / e.vt_T, tj2 = rj % e.vt_T;
            vt[((size_t)(bj * e.vt_heads + vhead) * e.vt_hd + vd) * e.vt_Tpad + vt_slot(tj2, e.vt_ws)] = (T)(vv[j] + bias);
          }
        }
      }
      continue;
    }
    // Row-major pass: thread -> (8-column group cg, rows r0 + k * NT / CG).  cg is the same for every item of a thread
    // (NT % CG == 0), so bias / column mapping are per-thread constants, and the loop is fully unrolled so that the LDS
    // reads and residual loads of all items are in flight together.
    constexpr int CG = BN_ / 8;
    static_assert(NT % CG == 0 && (CR * CG) % NT == 0, "epilogue item mapping");
    constexpr int RSTEP = NT / CG, ITERS = CR * CG / NT;
    const int cg = tid % CG, r0 = tid / CG;
    const int col0 = n0 + cg * 8;
    if (col0 >= N) continue;
    int dcol = col0, row_add = 0, bcol = col0;
    if (e.map == LA_MAP_CONVT2X2) {
      const int kyx = col0 / e.p2;
      dcol = col0 % e.p2;
      bcol = dcol;
      row_add = (kyx >> 1) * 2 * e.p0 + (kyx & 1);
    }
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(e.bias + bcol);
      const float4 b1 = *reinterpret_cast<const float4*>(e.bias + bcol + 4);
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
    const bool fast_gelu = sizeof(T) == 2 && !e.out32;      // result only survives as a 16-bit value
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
      const int r = r0 + k * RSTEP;
      int drow = drow_lds[r];
      if (drow < 0) continue;
      drow += row_add;
      float v[8];
      {
        const float4 a0 = *reinterpret_cast<const float4*>(&epi[r * LD + cg * 8]);
        const float4 a1 = *reinterpret_cast<const float4*>(&epi[r * LD + cg * 8 + 4]);
        v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
      }
      float4 r0v = make_float4(0.f, 0.f, 0.f, 0.f), r1v = r0v;
      if (e.res) {
        const int rr = e.res_mod ? drow % e.res_mod : drow;
        r0v = *reinterpret_cast<const float4*>(e.res + (size_t)rr * e.ldr + dcol);
        r1v = *reinterpret_cast<const float4*>(e.res + (size_t)rr * e.ldr + dcol + 4);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += bv[j];
      if (e.act == LA_ACT_GELU) {
        if (fast_gelu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_erf_fast(v[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_erf(v[j]);
        }
      } else if (e.act == LA_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      v[0] += r0v.x; v[1] += r0v.y; v[2] += r0v.z; v[3] += r0v.w; v[4] += r1v.x; v[5] += r1v.y; v[6] += r1v.z; v[7] += r1v.w;
      if (e.out32) store8<float>(e.out32 + (size_t)drow * e.ld32 + dcol, v);
      if (outT) store8<T>(outT + (size_t)drow * e.ld16 + dcol, v);
    }
  }
}

static bool epi_vec_ok(int N, const LaGemmEpilogue& e, int elt_bytes) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if ((N % 8) != 0) return false;
  if (e.bias && !al16(e.bias)) return false;
  if (e.res && (!al16(e.res) || (e.ldr % 4) != 0)) return false;
  if (e.out32 && (!al16(e.out32) || (e.ld32 % 4) != 0)) return false;
  if (e.out16 && (!al16(e.out16) || (e.ld16 % (16 / elt_bytes)) != 0)) return false;
  if (e.map == LA_MAP_CONVT2X2 && (e.p2 % 8) != 0) return false;
  if (e.vt && ((e.vt_col0 % BN) != 0 || (e.vt_Tpad % 4) != 0 || !al16(e.vt))) return false;
  return true;
}


// One 64-deep K tile for a 64 x 64 wave tile: 4 MFMA k-steps, fragments of step ks+1 are fetched from LDS before the
// MFMAs of step ks are issued (two register sets), so the ds_read latency hides behind matrix work.
template <typename T>
__device__ __forceinline__ void mma_ktile(const char* sa, const char* sw, int arow, int wrow, int fr, int fh, f32x16 (&acc)[2][2]) {
  uint4 af[2][2], wf[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    af[0][i] = *reinterpret_cast<const uint4*>(sa + swz_off(arow + i * 32 + fr, fh));
    wf[0][i] = *reinterpret_cast<const uint4*>(sw + swz_off(wrow + i * 32 + fr, fh));
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int cur = ks & 1, nxt = cur ^ 1;
    if (ks < 3) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[nxt][i] = *reinterpret_cast<const uint4*>(sa + swz_off(arow + i * 32 + fr, (ks + 1) * 2 + fh));
        wf[nxt][i] = *reinterpret_cast<const uint4*>(sw + swz_off(wrow + i * 32 + fr, (ks + 1) * 2 + fh));
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = Half16<T>::mfma32(af[cur][i], wf[cur][j], acc[i][j]);
  }
}

// =================================================================================================================
// v2 fast path (K % 64 == 0, N % 8 == 0, 16-byte aligned rows): operand tiles go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass; the XOR swizzle is applied to the per-lane SOURCE
// address because the DMA destination is lane-linear), and the epilogue is staged through LDS so that every lane
// handles 8 consecutive output columns (16-byte stores, float4 residual / bias loads).  Tile BM x BN with one
// 64 x 64 sub-tile per wave: 128x128 (4 waves, 2 blocks/CU).
// =================================================================================================================
constexpr int EPI_LD = BN + 4;                      // fp32 row stride of the staged output tile
constexpr int EPI_BYTES = 128 * EPI_LD * 4;         // one 128-row chunk

template <int BM_>
constexpr int fast_lds_bytes() {
  return (2 * (BM_ + BN) * BK * 2) > EPI_BYTES ? (2 * (BM_ + BN) * BK * 2) : EPI_BYTES;
}

template <typename T, int BM_>
__global__ __launch_bounds__(BM_ * 2, 2) void gemm_dma_kernel(const T* __restrict__ A, int lda, const T* __restrict__ Wt, int ldw,
                                                               int M, int N, int K, LaGemmEpilogue e, int gm) {
  constexpr int NW = (BM_ / 64) * 2;               // waves: (BM_/64) x 2
  constexpr int NT = NW * 64;
  constexpr int STAGE = (BM_ + BN) * BK * 2;       // bytes
  constexpr int NDMA = (BM_ + BN) / 8 / NW;        // wave-level DMA instructions per k-tile (8 rows each)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (N + BN - 1) / BN, ntm = (M + BM_ - 1) / BM_;
  int tm_, tn_;
  tile_coords(xcd_remap(blockIdx.x, ntm * ntn), ntm, ntn, gm, tm_, tn_);
  const int m0 = tm_ * BM_, n0 = tn_ * BN;

  // ---- DMA plan: instruction i of this wave moves 8 tile rows (1 KiB); lane -> (row lane/8, slot lane%8) --------
  // source = wave-uniform base (A or W, advanced by the k-tile) + 32-bit per-lane byte offset (launcher checks < 4 GiB)
  constexpr int NA = BM_ / 8 / NW;                 // pieces i < NA are A rows, the rest W rows
  static_assert((BM_ / 8) % NW == 0, "A / W pieces must split per instruction index");
  unsigned soff[NDMA];
  int ldsoff[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int grp = i * NW + wave;                 // row group index over [A rows | W rows]
    const int trow = grp * 8 + (lane >> 3);        // tile row in the concatenated (BM_ + BN) row space
    const int slot = lane & 7;
    if (i < NA) {
      const int r = trow;
      soff[i] = (unsigned)(((size_t)a_row(e, min(m0 + r, M - 1)) * lda + ((slot ^ ((r >> 1) & 7)) << 3)) * sizeof(T));
    } else {
      const int r = trow - BM_;
      soff[i] = (unsigned)(((size_t)min(n0 + r, N - 1) * ldw + ((slot ^ ((r >> 1) & 7)) << 3)) * sizeof(T));
    }
    ldsoff[i] = grp * 1024;                        // A tile occupies [0, BM_*128), W tile follows: same linear space
  }
  const unsigned lds0 = lds_addr_of(smem);
  auto dma = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) dma16s(i < NA ? A + a_koff(e, kt * BK) : Wt + kt * BK, soff[i], lds0 + stage * STAGE + ldsoff[i]);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / BK;
  const int fr = lane & 31, fh = lane >> 5;
  dma(0, 0);
  dma_wait<0>();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) dma(kt + 1, (kt + 1) & 1);
    const char* sa = smem + (kt & 1) * STAGE;
    const char* sw = sa + BM_ * BK * 2;
    mma_ktile<T>(sa, sw, wm * 64, wn * 64, fr, fh, acc);
    dma_wait<0>();     // this wave's part of the next tile has landed (issued before the MFMAs above) ...
    __syncthreads();   // ... and so has everyone else's; also orders the stage swap
  }

  // ---- epilogue through LDS, 128 rows at a time ----------------------------------------------------------------
  epilogue_lds<T, 2, 2, BN, NT>(reinterpret_cast<float*>(smem), acc, BM_ / 128, wm >> 1, (wm & 1) * 64, wn * 64, m0, n0, M, N, e, tid);
}


// =================================================================================================================
// v4: 256 x 128 block tile, 4 waves (2 x 2) with a 128 x 64 tile PER WAVE (4 x 2 MFMA 32x32 accumulators = 128 VGPRs),
// BK = 32, three LDS stages (72 KiB -> 2 workgroups per CU), prefetch distance 2 with counted vmcnt.
// Rationale (PMC: 0 bank conflicts, 41 % of wave time parked in waitcnt/barrier, MFMA busy 40 %): the 128x128 kernel moves
// 1.5 KiB through the LDS port per MFMA (1.0 fragment reads + 0.5 DMA writes); this shape moves 1.125 KiB.
// Rows are 64 B here, so four tile rows share a 256-B bank row: chunk c of row r lives in slot c ^ ((r >> 2) & 3).
// =================================================================================================================
__device__ __forceinline__ int swz64_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_dma4_kernel(const T* __restrict__ A, int lda, const T* __restrict__ Wt, int ldw,
                                                            int M, int N, int K, LaGemmEpilogue e, int gm) {
  constexpr int BM_ = 256, BK_ = 32;
  constexpr int NW = 4, NT = 256;
  constexpr int STAGE = (BM_ + BN) * BK_ * 2;      // 24 KiB
  constexpr int NDMA = (BM_ + BN) / 16 / NW;       // 6 wave-level DMA instructions (16 rows x 64 B each) per k-step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (N + BN - 1) / BN, ntm = (M + BM_ - 1) / BM_;
  int tm_, tn_;
  tile_coords(xcd_remap(blockIdx.x, ntm * ntn), ntm, ntn, gm, tm_, tn_);
  const int m0 = tm_ * BM_, n0 = tn_ * BN;

  constexpr int NA = BM_ / 16 / NW;                // pieces i < NA are A rows, the rest W rows
  unsigned soff[NDMA];                             // 32-bit byte offsets from A / W (launcher checks < 4 GiB)
  int ldsoff[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int grp = i * NW + wave;                 // 16-row group over [A rows | W rows]
    const int trow = grp * 16 + (lane >> 2);
    const int slot = lane & 3;
    if (i < NA) {
      const int r = trow;
      soff[i] = (unsigned)(((size_t)a_row(e, min(m0 + r, M - 1)) * lda + ((slot ^ ((r >> 2) & 3)) << 3)) * sizeof(T));
    } else {
      const int r = trow - BM_;
      soff[i] = (unsigned)(((size_t)min(n0 + r, N - 1) * ldw + ((slot ^ ((r >> 2) & 3)) << 3)) * sizeof(T));
    }
    ldsoff[i] = grp * 1024;
  }
  const unsigned lds0 = lds_addr_of(smem);
  auto dma = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) dma16s(i < NA ? A + a_koff(e, kt * BK_) : Wt + kt * BK_, soff[i], lds0 + stage * STAGE + ldsoff[i]);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / BK_;
  const int fr = lane & 31, fh = lane >> 5;
  dma(0, 0);
  if (nk > 1) dma(1, 1);
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) dma_wait<NDMA>();
    else dma_wait<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < nk) dma(kt + 2, (stage + 2 >= 3) ? stage - 1 : stage + 2);
    const char* sa = smem + stage * STAGE;
    const char* sw = sa + BM_ * BK_ * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 af[4], wf[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const uint4*>(sa + swz64_off(wm * 128 + i * 32 + fr, ks * 2 + fh));
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const uint4*>(sw + swz64_off(wn * 64 + j * 32 + fr, ks * 2 + fh));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Half16<T>::mfma32(af[i], wf[j], acc[i][j]);
    }
    stage = (stage == 2) ? 0 : stage + 1;
  }
  __syncthreads();
  epilogue_lds<T, 4, 2, BN, NT>(reinterpret_cast<float*>(smem), acc, 2, wm, 0, wn * 64, m0, n0, M, N, e, tid);
}

// row-panels per tile group (tile_coords); measured flat within +-2 % for 1..16 on the 256 x 128 / 128 x 128 kernels (default 8)
// and ~2 % better at 1..4 for the 256 x 256 kernel (dflt = 2 there) - except with >= 10 column tiles (lin1: N = 3072), where 8 row
// panels per group fetch 25 % less through the L2 (1.77 -> 1.33 M KiB of FETCH_SIZE per launch, tools/gemm_group_m.sh: with 2 row panels
// per group an XCD streams the whole 4.7 MB weight for every pair of panels) and run 1.3 % faster; lin2 / proj (3 column tiles) fetch
// and run worse beyond 2.  LA_GEMM_GROUP_M overrides both
static int tile_group_m(int dflt = 8) {
  static int forced = -2;
  if (forced == -2) {
    const char* v = la_dbg_env("LA_GEMM_GROUP_M");
    forced = v ? atoi(v) : -1;
  }
  return forced >= 0 ? forced : dflt;
}

template <typename T>
static void launch_fast4(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, hipStream_t st) {
  constexpr int LDS = 3 * (256 + BN) * 32 * 2;     // 72 KiB >= 67.5 KiB epilogue chunk
  static unsigned long long attr_mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(gemm_dma4_kernel<T>), LDS, attr_mask);
  const int ntm = (M + 255) / 256, ntn = (N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_dma4_kernel<T>), dim3(ntm * ntn), dim3(256), LDS, st, reinterpret_cast<const T*>(A), lda,
                     reinterpret_cast<const T*>(W), ldw, M, N, K, e, tile_group_m());
}


constexpr int PP_BM = 256, PP_BN = 256;
constexpr int PP_STAGE = (PP_BM + PP_BN) * BK * 2;     // 64 KiB
constexpr int PP_HALF = 128 * BK * 2;                   // 16 KiB half-tile

// Epilogue of the 256 x 256 tile kernels (8 waves, wave (grp, wi) holds rows [128 grp, +128) x columns [64 wi, +64) as
// acc[4][2]): four 64-row chunks staged through LDS (each wave's 128 rows span two chunks).
template <typename T>
__device__ __forceinline__ void epilogue_256(char* smem, const f32x16 (&acc)[4][2], int grp, int wrow, int m0, int n0, int M, int N,
                                             const LaGemmEpilogue& e, int tid) {
  constexpr int NT = 512;
  const int lane = tid & 63, fr = lane & 31, fh = lane >> 5;
  float* epi = reinterpret_cast<float*>(smem);
  constexpr int LD = PP_BN + 4;
  const RowMap rm{e.map, e.p0, e.p1, e.p2, e.p3, e.p4};
  T* outT = reinterpret_cast<T*>(e.out16);
  T* vt = reinterpret_cast<T*>(e.vt);
  const bool vt_tile = (vt != nullptr) && (n0 >= e.vt_col0);
#pragma unroll
  for (int chunk = 0; chunk < 4; ++chunk) {
    if (chunk > 0) __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (grp * 2 + half != chunk) continue;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int ti = half * 2 + t2;                 // static: accumulator row-tile of this wave
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          if (!vt_tile) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              epi[(t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * LD + wrow + tj * 32 + fr] = acc[ti][tj][r];
          } else {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const float4 v = make_float4(acc[ti][tj][g4 * 4], acc[ti][tj][g4 * 4 + 1], acc[ti][tj][g4 * 4 + 2], acc[ti][tj][g4 * 4 + 3]);
              *reinterpret_cast<float4*>(&epi[(wrow + tj * 32 + fr) * (64 + 4) + t2 * 32 + 8 * g4 + 4 * fh]) = v;
            }
          }
        }
      }
    }
    __syncthreads();
    const int mrow0 = m0 + chunk * 64;
    if (vt_tile) {      // transposed staging [col][64 rows + 4]: items = (col, 4-row group)
      for (int it = tid; it < PP_BN * 16; it += NT) {
        const int rg = it & 15, c = it >> 4;
        const int col = n0 + c, row = mrow0 + rg * 4;
        if (col >= N || row >= M) continue;
        const float4 v = *reinterpret_cast<const float4*>(&epi[c * 68 + rg * 4]);
        const float bias = e.bias ? e.bias[col] : 0.f;
        const int cv = col - e.vt_col0;
        int d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = (row + j < M) ? map_row(rm, row + j) : -1;
        vt_store4<T>(vt, e, d, cv / e.vt_hd, cv % e.vt_hd, v, bias);
      }
      continue;
    }
    constexpr int CG = PP_BN / 8;
    for (int it = tid; it < 64 * CG; it += NT) {
      const int cg = it % CG, r = it / CG;
      const int row = mrow0 + r, col0 = n0 + cg * 8;
      if (row >= M || col0 >= N) continue;
      float v[8];
      {
        const float4 a0 = *reinterpret_cast<const float4*>(&epi[r * LD + cg * 8]);
        const float4 a1 = *reinterpret_cast<const float4*>(&epi[r * LD + cg * 8 + 4]);
        v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
      }
      int dcol = col0, row_add = 0, bcol = col0;
      if (e.map == LA_MAP_CONVT2X2) {
        const int kyx = col0 / e.p2;
        dcol = col0 % e.p2;
        bcol = dcol;
        row_add = (kyx >> 1) * 2 * e.p0 + (kyx & 1);
      }
      if (e.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(e.bias + bcol);
        const float4 b1 = *reinterpret_cast<const float4*>(e.bias + bcol + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (e.act == LA_ACT_GELU) {
        if (!e.out32) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_erf_fast(v[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_erf(v[j]);
        }
      } else if (e.act == LA_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      int drow = map_row(rm, row);
      if (drow < 0) continue;
      drow += row_add;
      if (e.res) {
        const int rr = e.res_mod ? drow % e.res_mod : drow;
        const float4 r0 = *reinterpret_cast<const float4*>(e.res + (size_t)rr * e.ldr + dcol);
        const float4 r1 = *reinterpret_cast<const float4*>(e.res + (size_t)rr * e.ldr + dcol + 4);
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
      }
      if (e.out32) store8<float>(e.out32 + (size_t)drow * e.ld32 + dcol, v);
      if (outT) store8<T>(outT + (size_t)drow * e.ld16 + dcol, v);
    }
  }}

// =================================================================================================================
// v6 "ping-pong": 256 x 256 x 64 tile, 8 waves = 2 groups of 4; group g owns rows [128 g, +128), wave i of a group the
// columns [64 i, +64) -> 128 x 64 per wave (4 x 2 MFMA 32x32 accumulators).  Every SIMD hosts one wave of each group.
// A k-step (64) is walked QUADRANT-major: the wave's tile is 2 x 2 quadrants of 64 x 32 and one burst = one quadrant
// over the whole k-step (2 row tiles x 4 k-slices = 8 MFMAs = 256 matrix-pipe cycles); quadrant order (0,0) (0,1)
// (1,1) (1,0) so that each burst loads only one new operand half.  The two groups run ONE BARRIER OUT OF PHASE: while
// group 0 issues a burst, group 1 fetches fragments from LDS and issues its share of the LDS-DMA, then they swap.
//
//   interval      8t     8t+1    8t+2    8t+3    8t+4    8t+5    8t+6    8t+7        (one s_barrier between intervals)
//   group 0       L1      M1      L2      M2      L3      M3      L4      M4
//   group 1     M4(t-1)   L1      M1      L2      M2      L3      M3      L4
//   L1: read A0,W0  L2: read W1  L3: read A1  L4: read W0        M1: q(0,0)  M2: q(0,1)  M3: q(1,1)  M4: q(1,0)
//
// LDS: 2 stages x (A 256x64 + W 256x64) x 2 B = 128 KiB, k-step t in stage t & 1; full 128-byte rows so every DMA piece
// (8 rows) moves whole cache lines.  Rows are stored HALF-major: operand half h (the rows of quadrant index h of every
// wave) is one contiguous 16 KiB "half-tile" = 16 DMA pieces = 2 per wave, and a half-tile is restaged as soon as its
// last reader is done instead of waiting for the whole stage:
//   L1 issues A1(t+1)   L2 issues W0(t+1)   L3 issues A0(t+2)   L4 issues W1(t+2)
// Hazards (g1 runs one interval later than g0; reads are retired by the lgkmcnt(0) at the top of the following M):
//   * WAR: each half-tile is restaged two L phases after the L phase that read it last (A0: L1 -> L3, W1: L2 -> L4,
//     A1: L3 -> next L1, W0: L4 -> next L2), i.e. >= 1 full interval after the slower group's reads retired.
//   * RAW: k-step t+1 is first read in interval 8t+8.  Every wave retires its own pieces of k-step t+1 with a COUNTED
//     vmcnt before the barrier that closes interval 8t+7 (group 0 at the end of M4, group 1 in L4): only the younger
//     A0(t+2), W1(t+2) may stay in flight -> vmcnt(4).
// Operand traffic per MFMA: 0.75 KiB of fragment reads + 0.25 KiB of DMA (vs 1.0 + 0.5 in the 128x128 kernel) and half
// the L2 reads per FLOP.
// =================================================================================================================

template <typename T>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const T* __restrict__ A, int lda, const T* __restrict__ Wt, int ldw,
                                                          int M, int N, int K, LaGemmEpilogue e, int gm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2), wi = wave & 3;
  const int fr = lane & 31, fh = lane >> 5;
  const int ntn = (N + PP_BN - 1) / PP_BN, ntm = (M + PP_BM - 1) / PP_BM;
  int tm_, tn_;
  tile_coords(xcd_remap(blockIdx.x, ntm * ntn), ntm, ntn, gm, tm_, tn_);
  const int m0 = tm_ * PP_BM, n0 = tn_ * PP_BN;

  // DMA pieces of this wave: for half-tile (operand o, half h) the 8-row groups p = wave and wave + 8 of its 128 LDS rows.
  // LDS row lr of A half h holds tile row (lr / 64) * 128 + h * 64 + lr % 64; of W half h tile column (lr / 32) * 64 +
  // h * 32 + lr % 32.  The 16-byte chunk c of a row sits in slot c ^ ((lr >> 1) & 7) (conflict-free ds_read_b128).
  unsigned soff[2][2][2];    // [operand][half][piece] byte offsets from A / W (launcher checks < 4 GiB)
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = (wave + 8 * i) * 8 + (lane >> 3);
      const int slot = lane & 7;
      const int ch = (slot ^ ((lr >> 1) & 7)) << 3;
      const int ra = (lr >> 6) * 128 + h * 64 + (lr & 63);
      const int rw = (lr >> 5) * 64 + h * 32 + (lr & 31);
      soff[0][h][i] = (unsigned)(((size_t)a_row(e, min(m0 + ra, M - 1)) * lda + ch) * sizeof(T));
      soff[1][h][i] = (unsigned)(((size_t)min(n0 + rw, N - 1) * ldw + ch) * sizeof(T));
    }
  const unsigned lds0 = lds_addr_of(smem);
  const int nk = K / BK;
  // half-tile (o, h) of k-step kt: stage kt & 1, operand o at + o * 32 KiB, half h at + h * 16 KiB; this wave's pieces at
  // + wave KiB and + (8 + wave) KiB.  Source = wave-uniform base advanced by the k-step + 32-bit lane offset.
  auto dma_ht = [&](int kt, int o, int h) {
    if (kt >= nk) return;
    const unsigned base = lds0 + (kt & 1) * PP_STAGE + o * (2 * PP_HALF) + h * PP_HALF + wave * 1024;
    const T* sb = o ? Wt + kt * BK : A + a_koff(e, kt * BK);
    dma16s(sb, soff[o][h][0], base);
    dma16s(sb, soff[o][h][1], base + 8 * 1024);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 af[4][2], wf[4];                              // fragments [k-slice][row tile] / [k-slice] of the current halves
  const int wrow = wi * 64;                           // first tile column of this wave (epilogue)
  auto load_a = [&](int kt, int h) {
    const char* sa = smem + (kt & 1) * PP_STAGE + h * PP_HALF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) af[ks][i] = *reinterpret_cast<const uint4*>(sa + swz_off(grp * 64 + i * 32 + fr, ks * 2 + fh));
  };
  auto load_w = [&](int kt, int h) {
    const char* sw = smem + (kt & 1) * PP_STAGE + 2 * PP_HALF + h * PP_HALF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wf[ks] = *reinterpret_cast<const uint4*>(sw + swz_off(wi * 32 + fr, ks * 2 + fh));
  };
  auto bar = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
#define LA_PP_BURST(HA, HB)                                                                \
  do {                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                               \
      acc[2 * HA][HB] = Half16<T>::mfma32(af[ks][0], wf[ks], acc[2 * HA][HB]);                       \
      acc[2 * HA + 1][HB] = Half16<T>::mfma32(af[ks][1], wf[ks], acc[2 * HA + 1][HB]);               \
    }                                                                                                \
    __builtin_amdgcn_s_setprio(0);                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  } while (0)

  // prologue: all of k-step 0, then A0(1), W1(1) (what L3/L4 of a "k-step -1" would have issued)
  dma_ht(0, 0, 0);
  dma_ht(0, 1, 0);
  dma_ht(0, 0, 1);
  dma_ht(0, 1, 1);
  dma_ht(1, 0, 0);
  dma_ht(1, 1, 1);
  if (nk > 1) dma_wait<4>();
  else dma_wait<0>();
  bar();                                   // k-step 0 resident
  if (grp == 1) bar();                     // group 1 runs one interval behind
  for (int kt = 0; kt < nk; ++kt) {
    // ---- L1 / M1: quadrant (0,0) -------------------------------------------------------------------------------------
    dma_ht(kt + 1, 0, 1);
    load_w(kt, 0);
    load_a(kt, 0);
    bar();
    LA_PP_BURST(0, 0);
    bar();
    // ---- L2 / M2: quadrant (0,1) -------------------------------------------------------------------------------------
    dma_ht(kt + 1, 1, 0);
    load_w(kt, 1);
    bar();
    LA_PP_BURST(0, 1);
    bar();
    // ---- L3 / M3: quadrant (1,1) -------------------------------------------------------------------------------------
    dma_ht(kt + 2, 0, 0);
    load_a(kt, 1);
    bar();
    LA_PP_BURST(1, 1);
    bar();
    // ---- L4 / M4: quadrant (1,0) -------------------------------------------------------------------------------------
    dma_ht(kt + 2, 1, 1);
    load_w(kt, 0);
    if (grp == 1) {                        // group 1 closes interval 8 kt + 7 here: k-step kt+1 must be complete
      if (kt + 2 < nk) dma_wait<4>();
      else dma_wait<0>();
    }
    bar();
    LA_PP_BURST(1, 0);
    if (grp == 0) {
      if (kt + 2 < nk) dma_wait<4>();
      else dma_wait<0>();
    }
    bar();
  }
#undef LA_PP_BURST
  if (grp == 0) bar();                     // re-align the two groups
  __syncthreads();

  epilogue_256<T>(smem, acc, grp, wrow, m0, n0, M, N, e, tid);
}

template <typename T>
static void launch_pp(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, hipStream_t st) {
  constexpr int LDS = 2 * PP_STAGE;      // 128 KiB (epilogue chunk 64 x 260 x 4 = 65 KiB, transposed 256 x 68 x 4 = 68 KiB)
  static unsigned long long attr_mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(gemm_pp_kernel<T>), LDS, attr_mask);
  const int ntm = (M + PP_BM - 1) / PP_BM, ntn = (N + PP_BN - 1) / PP_BN;
  hipLaunchKernelGGL((gemm_pp_kernel<T>), dim3(ntm * ntn), dim3(512), LDS, st, reinterpret_cast<const T*>(A), lda,
                     reinterpret_cast<const T*>(W), ldw, M, N, K, e, tile_group_m(2));
}

// Epilogue of gemm_t256_kernel: the tile leaves in four 64-row chunks (wave group g holds rows [128 g, +128) = chunks 2g, 2g+1)
// through TWO staging buffers, in the order g0/h0, g1/h0, g0/h1, g1/h1: while every thread turns chunk p into 16-byte global
// stores, the owning group already writes chunk p+1 into the other buffer - one barrier per chunk, and the residual rows of
// a chunk are requested before that barrier.  A thread owns 8 consecutive columns (bias, column mapping: per-thread
// constants for the whole tile) of rows r0 + 16 k.  Transposed-V tiles take the column-major path of epilogue_256.
//
// CODE SIZE is what this epilogue is tuned for: it runs once per tile, so every instruction of it is an instruction-cache
// miss (a fully unrolled generic version made the kernel 147 KB and cost 8-20 us per tile, more than the data movement).
// The chunk loop is a real loop, and EPI specialises the per-element work at compile time for the four hot GEMMs:
//   EPI 1  bias -> 16-bit                  (qkv)          EPI 3  bias + fp32 residual -> fp32 [+ 16-bit]   (proj, lin2)
//   EPI 2  bias -> GELU -> 16-bit          (lin1)         EPI 0  everything at run time (maps, ReLU, res_mod, ...)
template <int EPI> struct EpiTraits {
  static constexpr bool generic = (EPI == 0);
};

template <typename T, int EPI>
__device__ __forceinline__ void epilogue_t256(char* smem, const f32x16 (&acc)[4][2], int grp, int wcol, int m0, int n0, int M, int N,
                                              const LaGemmEpilogue& e, int tid) {
  constexpr bool GEN = (EPI == 0);
  // LDS-only barrier: __syncthreads() would also wait for every outstanding global store / residual load (its fence drains
  // vmcnt), i.e. each chunk would pay a full store round trip; here only the LDS traffic has to be ordered
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  T* vt = reinterpret_cast<T*>(e.vt);
  if (GEN && vt != nullptr && n0 >= e.vt_col0) {
    epilogue_256<T>(smem, acc, grp, wcol, m0, n0, M, N, e, tid);
    return;
  }
  if (EPI == 1 && vt != nullptr && n0 >= e.vt_col0) {
    // Transposed-V tile, identity row map: the same two-buffer chunk pipeline with COLUMN-major staging [256 cols][64 rows + 4],
    // so that 4 consecutive tokens of one V column are one float4 and leave as one 8-byte store into vt[(b, head, d)][slot].
    constexpr int LDT = 64 + 4;
    constexpr int BUFT = 256 * LDT;                   // floats per buffer (69 632 B)
    float* epi = reinterpret_cast<float*>(smem);
    const int lane = tid & 63, fr = lane & 31, fh = lane >> 5;
    auto stageT0 = [&](float* buf) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
            *reinterpret_cast<float4*>(&buf[(wcol + tj * 32 + fr) * LDT + t2 * 32 + 8 * g4 + 4 * fh]) =
                make_float4(acc[t2][tj][g4 * 4], acc[t2][tj][g4 * 4 + 1], acc[t2][tj][g4 * 4 + 2], acc[t2][tj][g4 * 4 + 3]);
    };
    auto stageT1 = [&](float* buf) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
            *reinterpret_cast<float4*>(&buf[(wcol + tj * 32 + fr) * LDT + t2 * 32 + 8 * g4 + 4 * fh]) =
                make_float4(acc[2 + t2][tj][g4 * 4], acc[2 + t2][tj][g4 * 4 + 1], acc[2 + t2][tj][g4 * 4 + 2], acc[2 + t2][tj][g4 * 4 + 3]);
    };
    if (grp == 0) stageT0(epi);
    const bool quad_ok = (e.vt_T & 3) == 0;
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
      const int mrow0 = m0 + 128 * (p & 1) + 64 * (p >> 1);
      const float* buf = epi + (p & 1) * BUFT;
      lds_barrier();
      if (p + 1 < 4 && grp == ((p + 1) & 1)) {
        float* nb = epi + ((p + 1) & 1) * BUFT;
        if (p + 1 < 2) stageT0(nb);
        else stageT1(nb);
      }
#pragma unroll 2
      for (int it = tid; it < 256 * 16; it += 512) {
        const int rg = it & 15, c = it >> 4;
        const int col = n0 + c, row = mrow0 + rg * 4;
        if (col >= N || row >= M) continue;
        const float4 v = *reinterpret_cast<const float4*>(&buf[c * LDT + rg * 4]);
        const float bias = e.bias ? e.bias[col] : 0.f;
        const int cv = col - e.vt_col0;
        const int vhead = cv / e.vt_hd, vd = cv % e.vt_hd;
        const int b = row / e.vt_T, t = row % e.vt_T;
        T* rowp = vt + ((size_t)(b * e.vt_heads + vhead) * e.vt_hd + vd) * e.vt_Tpad;
        T* dst = rowp + vt_slot(t, e.vt_ws);
        if (quad_ok && row + 3 < M && (e.vt_ws == 0 || (e.vt_ws & 3) == 0 || (t % e.vt_ws) <= e.vt_ws - 4)) {
          store4v<T>(dst, v.x + bias, v.y + bias, v.z + bias, v.w + bias);      // 4 consecutive tokens, contiguous slots
        } else if (quad_ok && row + 3 < M && e.vt_ws > 0 && (e.vt_ws & 1) == 0) {
          store2<T>(dst, v.x + bias, v.y + bias);                               // the quad straddles a window row: two pairs
          store2<T>(rowp + vt_slot(t + 2, e.vt_ws), v.z + bias, v.w + bias);
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
          for (int j = 0; j < 4; ++j) {
            const int rj = row + j;
            if (rj >= M) break;
            const int bj = rj / e.vt_T, tj2 = rj % e.vt_T;
            vt[((size_t)(bj * e.vt_heads + vhead) * e.vt_hd + vd) * e.vt_Tpad + vt_slot(tj2, e.vt_ws)] = (T)(vv[j] + bias);
          }
        }
      }
    }
    return;
  }
  constexpr int LD = 256 + 4;
  constexpr int BUF = 64 * LD;                        // floats per staging buffer (66 560 B)
  float* epi = reinterpret_cast<float*>(smem);
  const int lane = tid & 63, fr = lane & 31, fh = lane >> 5;
  const RowMap rm{e.map, e.p0, e.p1, e.p2, e.p3, e.p4};
  T* outT = reinterpret_cast<T*>(e.out16);
  float* out32 = (GEN || EPI == 3) ? e.out32 : nullptr;
  const float* res = (GEN || EPI == 3) ? e.res : nullptr;
  const int act = GEN ? e.act : (EPI == 2 ? LA_ACT_GELU : LA_ACT_NONE);
  const int cg = tid & 31, r0 = tid >> 5;             // 8 columns cg*8.., rows r0 + 16 k of every chunk
  const int col0 = n0 + cg * 8;
  const bool col_ok = col0 < N;
  int dcol = col0, row_add = 0, bcol = col0;
  if (GEN && e.map == LA_MAP_CONVT2X2) {
    const int kyx = col0 / e.p2;
    dcol = col0 % e.p2;
    bcol = dcol;
    row_add = (kyx >> 1) * 2 * e.p0 + (kyx & 1);
  }
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (e.bias && col_ok) {
    const float4 b0 = *reinterpret_cast<const float4*>(e.bias + bcol);
    const float4 b1 = *reinterpret_cast<const float4*>(e.bias + bcol + 4);
    bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
  }
  const bool fast_gelu = !out32;                      // the result only survives as a 16-bit value
  // this wave's rows [64 half, +64) of its 128 -> buf rows 0..63 (the half is static: accumulator registers cannot be indexed at
  // run time).  One opaque per-lane base + compile-time offsets: without the asm the compiler hoists all 64 addresses of both
  // buffers out of the chunk loop and spills them.
  auto stage0 = [&](float* buf) {
    float* wb = buf + (4 * fh) * LD + wcol + fr;
    asm volatile("" : "+v"(wb));
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) wb[(t2 * 32 + (r & 3) + 8 * (r >> 2)) * LD + tj * 32] = acc[t2][tj][r];
  };
  auto stage1 = [&](float* buf) {
    float* wb = buf + (4 * fh) * LD + wcol + fr;
    asm volatile("" : "+v"(wb));
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) wb[(t2 * 32 + (r & 3) + 8 * (r >> 2)) * LD + tj * 32] = acc[2 + t2][tj][r];
  };
  float4 rv[4][2];
  auto dest_row = [&](int p, int k) {
    const int row = m0 + 128 * (p & 1) + 64 * (p >> 1) + r0 + 16 * k;
    int d = (row < M && col_ok) ? ((!GEN || rm.mode == LA_MAP_NONE) ? row : map_row(rm, row)) : -1;
    if (GEN && d >= 0) d += row_add;
    return d;
  };
  auto fetch_res = [&](int d, float4 (&r_)[2]) {
    r_[0] = r_[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (res && d >= 0) {
      const int rr = (GEN && e.res_mod) ? d % e.res_mod : d;
      r_[0] = *reinterpret_cast<const float4*>(res + (size_t)rr * e.ldr + dcol);
      r_[1] = *reinterpret_cast<const float4*>(res + (size_t)rr * e.ldr + dcol + 4);
    }
  };
  if (grp == 0) stage0(epi);
#pragma unroll 1
  for (int p = 0; p < 4; ++p) {
    const float* buf = epi + (p & 1) * BUF + r0 * LD + cg * 8;      // this thread's first item (opaque: see stage0)
    asm volatile("" : "+v"(buf));
    if (GEN || EPI == 3) {                             // residual rows of this chunk, requested in front of the barrier
#pragma unroll
      for (int k = 0; k < 4; ++k) fetch_res(dest_row(p, k), rv[k]);
    }
    lds_barrier();                                     // chunk p staged; buffer (p+1)&1 no longer read by anyone
    if (p + 1 < 4 && grp == ((p + 1) & 1)) {
      float* nb = epi + ((p + 1) & 1) * BUF;
      if (p + 1 < 2) stage0(nb);
      else stage1(nb);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int d = dest_row(p, k);
      float v[8];
      {
        const float4 a0 = *reinterpret_cast<const float4*>(&buf[16 * k * LD]);
        const float4 a1 = *reinterpret_cast<const float4*>(&buf[16 * k * LD + 4]);
        v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += bv[j];
      if (act == LA_ACT_GELU) {
        if (!GEN || fast_gelu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_erf_fast(v[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_erf(v[j]);
        }
      } else if (GEN && act == LA_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (GEN || EPI == 3) {
        v[0] += rv[k][0].x; v[1] += rv[k][0].y; v[2] += rv[k][0].z; v[3] += rv[k][0].w;
        v[4] += rv[k][1].x; v[5] += rv[k][1].y; v[6] += rv[k][1].z; v[7] += rv[k][1].w;
      }
      if (d >= 0) {
        if (out32) store8<float>(out32 + (size_t)d * e.ld32 + dcol, v);
        if (outT) store8<T>(outT + (size_t)d * e.ld16 + dcol, v);
      }
      __builtin_amdgcn_sched_barrier(0);               // keep the items apart: interleaving all four costs registers (spills)
    }
  }
}

// =================================================================================================================
// v7 "t256": 256 x 256 x 32 tile, 8 waves (2 x 4, 128 x 64 per wave = 4 x 2 MFMA 32x32 accumulators), symmetric waves,
// one barrier per k-step, LDS-DMA ring with counted vmcnt, and NPL weight PLANES per k-step:
//   NPL = 1  C = A . W^T                                   ring of 4 stages x 32 KiB
//   NPL = 2  C = A . W_hi^T + A . W_lo^T  (split precision) ring of 3 stages x 48 KiB
// With two planes every A fragment feeds TWO MFMAs, so a k-step carries 32 MFMAs per wave against 16 fragment reads and
// 6 DMA pieces: 24 B/clk/CU of operand stream at the full matrix rate against the ~30 B/clk/CU LDS-DMA ceiling
// (profiles/r01_gemm_ablation.md 7) - the shape at which the split-precision GEMMs are matrix-pipe bound instead of
// stream bound, which is what makes the second plane cheaper than a second GEMM.  W = [W_hi | W_lo] per row (ldw = 2 K).
// Rows are 64 B (BK = 32): chunk c of row r lives in slot c ^ ((r >> 2) & 3) (same image as the 256 x 128 kernel).
// =================================================================================================================
template <typename T, int NPL, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_t256_kernel(const T* __restrict__ A, int lda, const T* __restrict__ Wt, int ldw,
                                                            int M, int N, int K, LaGemmEpilogue e, int gm) {
  constexpr int BK_ = 32;
  constexpr int OPB = 256 * BK_ * 2;                 // 16 KiB per operand tile
  constexpr int STAGE = OPB * (1 + NPL);
  constexpr int NST = (NPL == 2) ? 3 : 4;
  constexpr int NP = 2 * (1 + NPL);                  // DMA pieces per wave per k-step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wi = wave & 3;
  const int fr = lane & 31, fh = lane >> 5;
  const int ntn = (N + 255) / 256, ntm = (M + 255) / 256;
  int tm_, tn_;
  tile_coords(xcd_remap(blockIdx.x, ntm * ntn), ntm, ntn, gm, tm_, tn_);
  const int m0 = tm_ * 256, n0 = tn_ * 256;

  // DMA plan: operand o (0 = A, 1 = W_hi, 2 = W_lo), piece i of this wave = tile rows [(wave + 8 i) 16, +16)
  unsigned soff[1 + NPL][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave + 8 * i) * 16 + (lane >> 2);
    const int ch = ((lane & 3) ^ ((r >> 2) & 3)) << 3;
    soff[0][i] = (unsigned)(((size_t)a_row(e, min(m0 + r, M - 1)) * lda + ch) * sizeof(T));
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) soff[1 + pl][i] = (unsigned)(((size_t)min(n0 + r, N - 1) * ldw + pl * K + ch) * sizeof(T));
  }
  const unsigned lds0 = lds_addr_of(smem);
  auto dma = [&](int kt, int stage) {
    const unsigned base = lds0 + stage * STAGE + wave * 1024;
#pragma unroll
    for (int o = 0; o < 1 + NPL; ++o)
#pragma unroll
      for (int i = 0; i < 2; ++i) dma16s(o ? Wt + kt * BK_ : A + a_koff(e, kt * BK_), soff[o][i], base + o * OPB + i * 8192);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- main loop: the two wave groups (one wave of each per SIMD) run ONE BARRIER OUT OF PHASE -------------------------
  //   interval      2t      2t+1     2t+2     2t+3          L(t): issue the DMA pieces of k-step t + NST - 1, fetch ALL
  //   group 0      L(t)     M(t)    L(t+1)   M(t+1)               fragments of k-step t, retire the reads (lgkmcnt 0)
  //   group 1     M(t-1)    L(t)     M(t)    L(t+1)         M(t): the 16 NPL MFMAs of the k-step, nothing else
  // so each SIMD's matrix pipe always has exactly one wave feeding it while the other wave does the memory work.
  // Hazards (every interval ends with a workgroup barrier):
  //   RAW  k-step t+1 is first read in interval 2t+2 (group 0): every wave retires ITS pieces of t+1 with a counted vmcnt
  //        before the barrier closing interval 2t+1 - group 0 at the end of M(t), group 1 at the end of L(t).
  //   WAR  the stage of k-step t is last read in interval 2t+1 (group 1's L(t), reads retired before its barrier) and is
  //        restaged with k-step t + NST by L(t+1), i.e. from interval 2t+2 on.
  const int nk = K / BK_;
  uint4 af[2][4], wf[2][NPL][2];
  auto load_frags = [&](int stage) {
    const char* sa = smem + stage * STAGE;
    const char* sw = sa + OPB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) af[ks][i] = *reinterpret_cast<const uint4*>(sa + swz64_off(grp * 128 + i * 32 + fr, ks * 2 + fh));
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wf[ks][pl][j] = *reinterpret_cast<const uint4*>(sw + pl * OPB + swz64_off(wi * 64 + j * 32 + fr, ks * 2 + fh));
    }
  };
  auto bar = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // retire this wave's pieces of k-step kt + 1: the (up to NST - 2) younger k-steps may stay in flight
  auto retire_next = [&](int kt) {
    const int younger = min(NST - 2, nk - 2 - kt);
    if (younger >= 2) dma_wait<2 * NP>();
    else if (younger == 1) dma_wait<NP>();
    else dma_wait<0>();
  };
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) dma(s, s);
  retire_next(-1);                                   // k-step 0 resident ...
  bar();                                             // ... for everybody
  if (grp == 1) bar();                               // group 1 runs one interval behind
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // ---- L(kt) ------------------------------------------------------------------------------------------------------
    {
      int ps = stage + NST - 1;
      if (ps >= NST) ps -= NST;
      if (kt + NST - 1 < nk) dma(kt + NST - 1, ps);
    }
    load_frags(stage);
    if (grp == 1) retire_next(kt);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bar();
    // ---- M(kt) ------------------------------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = Half16<T>::mfma32(af[ks][i], wf[ks][pl][j], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 0) retire_next(kt);
    bar();
    stage = (stage == NST - 1) ? 0 : stage + 1;
  }
  if (grp == 0) bar();                               // re-align the two groups
  __syncthreads();
  epilogue_t256<T, EPI>(smem, acc, grp, wi * 64, m0, n0, M, N, e, tid);
}

// =================================================================================================================
// v8 "t256p": the same 256 x 256 x 32 main loop as gemm_t256_kernel, PERSISTENT: one workgroup per CU walks its tiles and
// the k-step stream never stops at a tile seam -
//   * the last NST - 1 L intervals of a tile issue the first NST - 1 k-steps of the NEXT tile, so the next main loop
//     starts on resident stages (no prologue bubble, no relaunch, and the finished tile's stores drain behind it instead
//     of holding the CU until the workgroup may retire);
//   * the epilogue never touches the ring and has no workgroup barrier: every wave transposes its own 128 x 64 block
//     through a PRIVATE 2 KiB slab behind the ring (16 rows x 64 columns of 16-bit output, or 8 rows of fp32) and stores
//     whole 128-B / 256-B row segments.  It sits at the head of the wave's next L interval, i.e. while the SIMD's other
//     wave runs its MFMA interval (the two groups stay one barrier out of phase across the seam).
// vmcnt bookkeeping across a seam: the epilogue's S stores sit between the DMA pieces of next-tile k-steps NST - 2 and
// NST - 1 in issue order (the counter retires in order and counts stores), so the first NST - 2 waits of the new tile
// allow S more outstanding operations (only for tiles whose store count is exact: interior rows, not a V^T tile).
// EPI: 1 bias -> 16-bit (+ V^T columns), 2 bias -> GELU -> 16-bit, 3 bias + fp32 residual -> fp32 [+ 16-bit].
// =================================================================================================================

template <typename T, int NPL, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_t256p_kernel(const T* __restrict__ A, int lda, const T* __restrict__ Wt, int ldw,
                                                             int M, int N, int K, LaGemmEpilogue e, int gm) {
  constexpr int BK_ = 32;
  constexpr int OPB = 256 * BK_ * 2;                 // 16 KiB per operand tile
  constexpr int STAGE = OPB * (1 + NPL);
  constexpr int NST = (NPL == 2) ? 3 : 4;
  constexpr int NP = 2 * (1 + NPL);                  // DMA pieces per wave per k-step
  constexpr int SDECL = (EPI == 3) ? 32 : 16;        // stores per wave of an interior, non-V^T tile (never more than are issued)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wi = wave & 3;
  const int fr = lane & 31, fh = lane >> 5;
  const int ntn = N / 256, ntm = (M + 255) / 256, ntiles = ntm * ntn;
  char* slab = smem + NST * STAGE + wave * 2048;
  // destination-row table of a mapped epilogue (512 B per wave): behind the slabs when the ring leaves room (one plane), else the
  // wave's own slab - two planes are only mapped when every tile is a V^T tile, which never touches the slab (host side)
  unsigned* rtab = nullptr;
  if (EPI == 1 && e.map != LA_MAP_NONE)
    rtab = reinterpret_cast<unsigned*>(NPL == 1 ? smem + NST * STAGE + 8 * 2048 + wave * 512 : slab);

  // DMA plan of a tile: operand o (0 = A, 1 = W_hi, 2 = W_lo), piece i of this wave = tile rows [(wave + 8 i) 16, +16)
  auto plan = [&](int tile, unsigned (&so)[1 + NPL][2], int& m0, int& n0) {
    int tm_, tn_;
    tile_coords(xcd_remap(tile, ntiles), ntm, ntn, gm, tm_, tn_);
    m0 = tm_ * 256;
    n0 = tn_ * 256;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wave + 8 * i) * 16 + (lane >> 2);
      const int ch = ((lane & 3) ^ ((r >> 2) & 3)) << 3;
      so[0][i] = (unsigned)(((size_t)a_row(e, min(m0 + r, M - 1)) * lda + ch) * sizeof(T));
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) so[1 + pl][i] = (unsigned)(((size_t)(n0 + r) * ldw + pl * K + ch) * sizeof(T));
    }
  };
  const unsigned lds0 = lds_addr_of(smem);
  auto dma = [&](const unsigned (&so)[1 + NPL][2], int kt, int stage) {
    const unsigned base = lds0 + stage * STAGE + wave * 1024;
#pragma unroll
    for (int o = 0; o < 1 + NPL; ++o)
#pragma unroll
      for (int i = 0; i < 2; ++i) dma16s(o ? Wt + kt * BK_ : A + a_koff(e, kt * BK_), so[o][i], base + o * OPB + i * 8192);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / BK_;                            // >= NST (host side)
  uint4 af[2][4], wf[2][NPL][2];
  auto load_frags = [&](int stage) {
    const char* sa = smem + stage * STAGE;
    const char* sw = sa + OPB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) af[ks][i] = *reinterpret_cast<const uint4*>(sa + swz64_off(grp * 128 + i * 32 + fr, ks * 2 + fh));
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wf[ks][pl][j] = *reinterpret_cast<const uint4*>(sw + pl * OPB + swz64_off(wi * 64 + j * 32 + fr, ks * 2 + fh));
    }
  };
  auto bar = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  unsigned soff[1 + NPL][2], soffn[1 + NPL][2];
  int m0, n0, m0n = 0, n0n = 0;
  int tile = blockIdx.x;
  plan(tile, soff, m0, n0);
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) dma(soff, s, s);
  dma_wait<(NST - 2) * NP>();                        // k-step 0 resident ...
  bar();                                             // ... for everybody
  if (grp == 1) bar();                               // group 1 runs one interval behind
  int stage = 0;
  bool seam_slack = false;                           // the finished tile's SDECL stores are still in front of this tile's late pieces
  for (;;) {
    const int next = tile + gridDim.x;
    const bool more = next < ntiles;
    if (more) plan(next, soffn, m0n, n0n);
    for (int kt = 0; kt < nk; ++kt) {
      // ---- L(kt) ------------------------------------------------------------------------------------------------------
      {
        int ps = stage + NST - 1;
        if (ps >= NST) ps -= NST;
        const int pf = kt + NST - 1;
        if (pf < nk) dma(soff, pf, ps);
        else if (more) dma(soffn, pf - nk, ps);
      }
      // retire this wave's pieces of the NEXT k-step of the stream (kt + 1, or the next tile's k-step 0)
      auto retire_next = [&]() {
        const int younger = more ? NST - 2 : min(NST - 2, nk - 2 - kt);
        if (seam_slack && kt < NST - 2) {            // (implies more-or-not irrelevant: kt + 1 <= NST - 2 < nk)
          if (NST == 4) dma_wait<2 * NP + SDECL>();
          else dma_wait<NP + SDECL>();
        } else if (younger >= 2 && NST == 4) dma_wait<2 * NP>();
        else if (younger >= 1) dma_wait<NP>();
        else dma_wait<0>();
      };
      load_frags(stage);
      if (grp == 1) retire_next();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      bar();
      // ---- M(kt) ------------------------------------------------------------------------------------------------------
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = Half16<T>::mfma32(af[ks][i], wf[ks][pl][j], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (grp == 0) retire_next();
      if (kt + 1 < nk || grp == 0) bar();             // group 1 keeps its last M interval open: both groups' epilogues share it
      stage = (stage == NST - 1) ? 0 : stage + 1;
    }
    // ---- tile done: this wave's block leaves through its private slab; the ring already holds the next tile's first k-steps.
    // Interval plan at a seam (| = barrier):   group 0   M(nk-1) | E      | L'(0) | M'(0) | ...
    //                                          group 1   L(nk-1) | M(nk-1) E | -  | L'(0) | ...
    // i.e. both epilogues run side by side (they are latency-, not issue-bound) and the one-interval stagger is restored after it. ----
    epilogue_wave<T, EPI>(slab, rtab, acc, m0 + grp * 128, n0 + wi * 64, n0, M, e, lane);
    seam_slack = (m0 + 256 <= M) && !(EPI == 1 && e.vt != nullptr && n0 >= e.vt_col0);
    bar();
    if (!more) break;
    if (grp == 1) bar();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int o = 0; o < 1 + NPL; ++o)
#pragma unroll
      for (int i = 0; i < 2; ++i) soff[o][i] = soffn[o][i];
    m0 = m0n;
    n0 = n0n;
    tile = next;
  }
}

// =================================================================================================================
// v9 "t256q": the persistent 256 x 256 tile with a 64-deep k-step in four QUADRANT phases (one weight plane).
// Why: with BK = 32 an LDS-DMA piece is 16 rows x 64 B - half a cache line per row, every 128-B line of A and W passes the
// vector-memory front end twice, and the k-step was paced by exactly that (profiles/r02_gemm_notes.md: the pieces, not their
// source, not the fragment reads).  Here a piece is 8 rows x 128 B = 8 whole lines.
// Per wave (128 x 64 block = acc[4][2]) a 64-deep k-tile is four phases of 8 MFMA 32x32x16 (256 cycles) each:
//     q0  acc[0..1][0] += A_sub0 . W_j0      reads A_sub0 (8 x b128) + W_j0 (4)          DMA quarter Q0 of the NEXT k-tile
//     q1  acc[0..1][1] += A_sub0 . W_j1      reads W_j1 (4)                               Q1
//     q2  acc[2..3][1] += A_sub1 . W_j1      reads A_sub1 (8, same registers as A_sub0)   Q2
//     q3  acc[2..3][0] += A_sub1 . W_j0      (W_j0 kept in registers)                      Q3
// Quarters are ordered by FIRST USE: Q0 = the A_sub0 rows of both wave groups, Q1 = the W_j0 rows of the four wave columns,
// Q2 = W_j1, Q3 = A_sub1 - so a quarter issued in phase q of k-tile kt is first read 3-4 phases later and TWO k-tile buffers
// (2 x 64 KiB) are enough.  Counted waits (each wave issues 2 pieces per phase; in steady state 8 are outstanding before phase 3's wait):
//     end of q3: vmcnt(4) -> Q0, Q1 of the next k-tile landed      end of q0: vmcnt(4) -> Q2 of this one      end of q1: vmcnt(4) -> Q3
// always one barrier ahead of the first reader.  The two wave groups run one barrier out of phase (group 1's L interval = group
// 0's M interval) exactly as in t256p; the seam (epilogue through the private slabs, next tile's first k-tile already in flight,
// store slack in the first two waits of a tile) is the same as well.
// =================================================================================================================
// EPI 4 (split-K weight gradients): no bias / activation / residual - the accumulators are ADDED to out32 with fp32 atomics, straight
// from the accumulator layout (a register is 32 consecutive columns of one row per half wave: 128-byte segments).  ksplit > 1 cuts the
// K range into chunks of kchunk (a multiple of 64) that run as independent tiles: dW[N, K] = dY^T X over 10^4 - 10^5 tokens has
// 9 - 36 output tiles only, the chunks are what fills the chip.
static int g_gemm_variant = 2;       // see launch_t256p (0: the BK = 32 persistent kernel everywhere)

#ifdef LA_DEBUG
constexpr int LA_DBG_NSTAMP = 64;
__device__ unsigned long long g_dbg_stamps[8 * LA_DBG_NSTAMP];      // [wave][i] = s_memtime << 8 | tag (workgroup 0 of the last stamped launch)
#endif

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_t256q_kernel(const T* __restrict__ A, int lda, const T* __restrict__ Wt, int ldw,
                                                             int M, int N, int K, LaGemmEpilogue e, int gm, int ksplit, int kchunk) {
  constexpr int BK_ = 64;
  constexpr int OPB = 256 * BK_ * 2;                 // 32 KiB per operand k-tile
  constexpr int BUFB = 2 * OPB;                      // 64 KiB per k-tile
  constexpr int SDECL = (EPI == 3) ? 32 : 16;        // stores per wave of an interior, non-V^T tile (never more than are issued)
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef LA_DEBUG
  const bool stamps = ((gm >> 10) & 1) && blockIdx.x == 0;      // seam timeline of workgroup 0 (la_dbg_gemm_stamps)
  int stamp_i = 0;
  auto stamp = [&](int tag) {
    if (stamps && stamp_i < LA_DBG_NSTAMP) {
      unsigned long long t_;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
      if ((threadIdx.x & 63) == 0) g_dbg_stamps[(threadIdx.x >> 6) * LA_DBG_NSTAMP + stamp_i] = (t_ << 8) | (unsigned)tag;
      ++stamp_i;
    }
  };
  const bool nostore = (gm >> 8) & 1;                // measurement ablations (la_gemm_variant bits 8 / 23: results wrong by construction);
  const bool noepi = (gm >> 23) & 1;                 // they exist in the -DLA_DEBUG library only
#else
  constexpr bool nostore = false, noepi = false;
  auto stamp = [](int) {};
#endif
  gm &= 0xff;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wi = wave & 3;
  const int fr = lane & 31, fh = lane >> 5;
  const int ntn = N / 256, ntm = (M + 255) / 256, ntmn = ntm * ntn, ntiles = ntmn * ksplit;
  char* slab = smem + 2 * BUFB + wave * 2048;
  unsigned* rtab = nullptr;
  if (EPI == 1 && e.map != LA_MAP_NONE) rtab = reinterpret_cast<unsigned*>(smem + 2 * BUFB + 8 * 2048 + wave * 512);

  // first tile row of piece i (0, 1) of this wave in quarter q
  auto piece_row0 = [&](int q, int i) {
    const int p = wave * 2 + i;
    if (q == 0 || q == 3) return (p >> 3) * 128 + (p & 7) * 8 + (q == 3 ? 64 : 0);
    return (p >> 2) * 64 + (p & 3) * 8 + (q == 2 ? 32 : 0);
  };
  // per-lane source offsets of the 8 pieces of a k-tile (ONE set: the next tile's offsets replace them at the head of a tile's last
  // k-tile, when every piece of the current tile has been issued - a second set cost 8 VGPRs and, at 256, spills whose reloads the
  // compiler guards with s_waitcnt vmcnt(0): each one drains the DMA ring and the epilogue's own store burst)
  unsigned soff[4][2];
  int kb_issue = 0;                                  // first k of the chunk the offsets in soff belong to
  auto plan = [&](int tile, int& m0, int& n0, int& kb, int& nkt) {
    int tm_, tn_;
    const int tt = xcd_remap(tile, ntiles);
    const int chunk = tt / ntmn;
    tile_coords(tt - chunk * ntmn, ntm, ntn, gm, tm_, tn_);
    m0 = tm_ * 256;
    n0 = tn_ * 256;
    kb = chunk * kchunk;
    nkt = (min(K, kb + kchunk) - kb) / BK_;
    kb_issue = kb;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = piece_row0(q, i) + (lane >> 3);
        const int ch = ((lane & 7) ^ ((r >> 1) & 7)) << 3;
        if (q == 0 || q == 3) soff[q][i] = (unsigned)(((size_t)a_row(e, min(m0 + r, M - 1)) * lda + ch) * sizeof(T));
        else soff[q][i] = (unsigned)(((size_t)(n0 + r) * ldw + ch) * sizeof(T));
      }
  };
  const unsigned lds0 = lds_addr_of(smem);
  auto dma_q = [&](int q, int kt, int buf) {
    const bool isw = (q == 1 || q == 2);
    const T* src = isw ? Wt + kb_issue + kt * BK_ : A + a_koff(e, kb_issue + kt * BK_);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16s(src, soff[q][i], lds0 + buf * BUFB + (isw ? OPB : 0) + piece_row0(q, i) * 128);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 af[2][4], wf[4];
  auto read_a = [&](int buf, int sub) {
    const char* sa = smem + buf * BUFB;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        af[i][ks] = *reinterpret_cast<const uint4*>(sa + swz_off(grp * 128 + (sub * 2 + i) * 32 + fr, ks * 2 + fh));
  };
  auto read_w = [&](int buf, int j) {
    const char* sw = smem + buf * BUFB + OPB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wf[ks] = *reinterpret_cast<const uint4*>(sw + swz_off(wi * 64 + j * 32 + fr, ks * 2 + fh));
  };
  auto bar = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  int m0, n0, kb0, nk, m0n = 0, n0n = 0, kbn = 0, nkn = 0;       // nk: 64-deep k-tiles of the current tile (>= 2, host side)
  int tile = blockIdx.x;
  plan(tile, m0, n0, kb0, nk);
#pragma unroll
  for (int q = 0; q < 4; ++q) dma_q(q, 0, 0);
  dma_wait<4>();                                     // Q0, Q1 of k-tile 0 ...
  bar();                                             // ... for everybody
  if (grp == 1) bar();                               // group 1 runs one interval behind
  int buf = 0;
  bool seam_slack = false;
  for (;;) {
    const int next = tile + gridDim.x;
    const bool more = next < ntiles;
    for (int kt = 0; kt < nk; ++kt) {
      const bool last = kt + 1 == nk;
      const bool feed = !last || more;               // a k-tile follows in the stream
      if (last && more) plan(next, m0n, n0n, kbn, nkn);        // every piece of this tile is on its way: the offsets now describe the next tile
      // counted wait at the end of phase q (q = 3, 0, 1); see the header
      auto retire = [&](int q) {
        if (q == 3) {
          if (feed) dma_wait<4>();
        } else if (!feed) {
          if (q == 0) dma_wait<2>();
          else dma_wait<0>();
        } else if (seam_slack && kt == 0) {
          dma_wait<4 + SDECL>();
        } else {
          dma_wait<4>();
        }
      };
      auto phase = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        // ---- L: fragment reads first, then the two DMA pieces (their issue stall overlaps the LDS latency) -------------------
        if (q == 0) {
          read_w(buf, 0);
          read_a(buf, 0);
        } else if (q == 1) {
          read_w(buf, 1);
        } else if (q == 2) {
          read_a(buf, 1);
        } else {
          read_w(buf, 0);                            // W_j0 again: 4 reads in the otherwise empty L(q3) instead of 16 registers held
        }
        __builtin_amdgcn_sched_barrier(0);
        if (feed) dma_q(q, last ? 0 : kt + 1, buf ^ 1);
        if (grp == 1 && q != 2) retire(q);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        bar();
        // ---- M ----------------------------------------------------------------------------------------------------------
        __builtin_amdgcn_s_setprio(1);
        constexpr int ib = (q >= 2) ? 2 : 0, jb = (q == 1 || q == 2) ? 1 : 0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[ib + i][jb] = Half16<T>::mfma32(af[i][ks], wf[ks], acc[ib + i][jb]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0 && q != 2) retire(q);
        if (q != 3 || !last || grp == 0) bar();      // group 1 keeps its last M interval open: both groups' epilogues share it
      };
      if (last) stamp(1);                            // head of the tile's last k-tile
      phase(std::integral_constant<int, 0>{});
      phase(std::integral_constant<int, 1>{});
      phase(std::integral_constant<int, 2>{});
      phase(std::integral_constant<int, 3>{});
      buf ^= 1;
      if (kt < 2) stamp(4 + kt);                     // end of the tile's first / second k-tile
    }
    stamp(2);                                        // main loop done
    if (noepi) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else if (EPI == 4) {
      const int fr_ = lane & 31, fh_ = lane >> 5;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + grp * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh_;
          if (row < M) {
            // (LA_MAP_GROUP: the row blocks of several weight gradients that sit a fixed stride apart in one flat gradient buffer)
            const int drow = e.map == LA_MAP_GROUP ? (row / e.p0) * e.p1 + row % e.p0 + e.p2 : row;
            float* op = e.out32 + (size_t)drow * e.ld32 + n0 + wi * 64 + fr_;
            unsafeAtomicAdd(op, acc[i][0][r]);
            unsafeAtomicAdd(op + 32, acc[i][1][r]);
          }
        }
    } else {
      epilogue_wave<T, (EPI >= 4 ? 1 : EPI)>(slab, rtab, acc, m0 + grp * 128, n0 + wi * 64, n0, M, e, lane, nostore);
    }
    stamp(3);                                        // epilogue issued
    seam_slack = EPI != 4 && (m0 + 256 <= M) && !(EPI == 1 && e.vt != nullptr && n0 >= e.vt_col0) && !nostore && !noepi;
    bar();
    if (!more) break;
    if (grp == 1) bar();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    m0 = m0n;
    n0 = n0n;
    nk = nkn;
    tile = next;
    stamp(6);                                        // accumulators cleared: the next tile's main loop starts
  }
}


template <typename T, int EPI>
static void launch_t256q(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, hipStream_t st) {
  constexpr int LDS = 2 * 65536 + 8 * 2048 + 8 * 512;       // two k-tile buffers + 2 KiB slab per wave + row tables: 148 KiB
  static unsigned long long attr_mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(gemm_t256q_kernel<T, EPI>), LDS, attr_mask);
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (ncu <= 0) ncu = 256;
  }
  int ksplit = 1, kchunk = K;
  if (EPI == 4) {
    // chunks of c k-tiles (every chunk, the last included, at least 2 deep): ONE round of tiles over the chip.  Every chunk ends in
    // 64 K fp32 atomics on its output tile, and the chunks of a tile serialise on them in L2: with K = 46912 and 9 output tiles,
    // 16 / 28 / 32 / 64 / 114 chunks measured 97 / - / 131 / 155 / 210 us (more than one round also pays the tile quantisation)
    const int tmn = ((M + 255) / 256) * (N / 256), nkt = K / 64;
    int want = ncu / tmn;
    static const char* wenv = la_dbg_env("LA_KSPLIT_WANT");      // debugging: force the number of K chunks
    if (wenv) want = atoi(wenv);
    if (want > nkt / 2) want = nkt / 2;
    if (want < 1) want = 1;
    int c = (nkt + want - 1) / want;
    if (c < 2) c = 2;
    while (c < nkt && (nkt % c) == 1) ++c;
    if (c > nkt) c = nkt;
    kchunk = c * 64;
    ksplit = (nkt + c - 1) / c;
  }
  const int ntiles = ((M + 255) / 256) * (N / 256) * ksplit;
  int grid = ntiles < ncu ? ntiles : ncu;
  static const char* genv = la_dbg_env("LA_KSPLIT_GRID");      // debugging: workgroups launched (0 = one per tile)
  if (EPI == 4 && genv) grid = atoi(genv) > 0 ? atoi(genv) : ntiles;
  hipLaunchKernelGGL((gemm_t256q_kernel<T, EPI>), dim3(grid), dim3(512), LDS, st, reinterpret_cast<const T*>(A), lda,
                     reinterpret_cast<const T*>(W), ldw, M, N, K, e, tile_group_m(N >= 2560 ? 8 : 2) | (g_gemm_variant & 0x800500), ksplit, kchunk);
}

template <typename T, int NPL, int EPI>
static void launch_t256p(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, hipStream_t st) {
  // main loop of the single-plane 64-deep shapes (la_gemm_variant): 2 (default) = the four-wave kernel (gemm_w4.hip; its own epilogue on
  // interior unmapped tiles - lin1, lin2, proj - and epilogue_wave elsewhere: measured ahead of the eight-wave kernel on every encoder
  // shape, profiles/r05_notes.md), 1 = the eight-wave quadrant-phase kernel
  const int var = g_gemm_variant & 0xff;
  const bool k64 = NPL == 1 && (K % 64) == 0 && K >= 128 && (e.a_kmod == 0 || (e.a_kmod % 64) == 0);
  if (k64 && var == 2) return launch_t256w<T, EPI>(A, lda, W, ldw, M, N, K, e, tile_group_m(N >= 2560 ? 8 : 2) | (g_gemm_variant & 0xf500), st);
  if (k64 && var >= 1) return launch_t256q<T, EPI>(A, lda, W, ldw, M, N, K, e, st);
  constexpr int LDS = ((NPL == 2) ? 3 * 49152 : 4 * 32768 + 8 * 512) + 8 * 2048;      // ring + 2 KiB slab per wave (+ row tables): 148 / 160 KiB
  static unsigned long long attr_mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(gemm_t256p_kernel<T, NPL, EPI>), LDS, attr_mask);
  static int ncu = 0;                                // (the GPUs of one host are the same part)
  if (ncu == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (ncu <= 0) ncu = 256;
  }
  const int ntiles = ((M + 255) / 256) * (N / 256);
  const int grid = ntiles < ncu ? ntiles : ncu;
  hipLaunchKernelGGL((gemm_t256p_kernel<T, NPL, EPI>), dim3(grid), dim3(512), LDS, st, reinterpret_cast<const T*>(A), lda,
                     reinterpret_cast<const T*>(W), ldw, M, N, K, e, tile_group_m(2));
}

template <typename T, int NPL, int EPI>
static void launch_t256_epi(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, hipStream_t st) {
  constexpr int LDS = (NPL == 2) ? 3 * 49152 : 136 * 1024;     // ring 144 / 128 KiB; epilogue: two staging buffers of 65 KiB
  static unsigned long long attr_mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(gemm_t256_kernel<T, NPL, EPI>), LDS, attr_mask);
  const int ntm = (M + 255) / 256, ntn = (N + 255) / 256;
  hipLaunchKernelGGL((gemm_t256_kernel<T, NPL, EPI>), dim3(ntm * ntn), dim3(512), LDS, st, reinterpret_cast<const T*>(A), lda,
                     reinterpret_cast<const T*>(W), ldw, M, N, K, e, tile_group_m(2));
}

// pick the compile-time epilogue variant that covers this call (see epilogue_t256)
template <typename T, int NPL>
static void launch_t256(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, hipStream_t st) {
  const bool plain = e.map == LA_MAP_NONE && e.res_mod == 0;
  static const char* nop = la_dbg_env("LA_GEMM_NO_PERSISTENT");
  const bool al = (N % 256) == 0 && K / 32 >= 8 && !nop && (e.ld16 % 8) == 0;
  if (al && plain && e.act == LA_ACT_NONE && !e.res && !e.out32 && e.out16) return launch_t256p<T, NPL, 1>(A, lda, W, ldw, M, N, K, e, st);
  // qkv of a SAM window block from image-order tokens: rows scattered into window order by the epilogue (no padded rows multiplied)
  const bool scatter = e.map == LA_MAP_WINDOW_PART && e.res_mod == 0 && e.amap == LA_MAP_NONE && e.act == LA_ACT_NONE && !e.res && !e.out32 &&
                       e.out16 && (NPL == 1 || (e.vt && e.vt_col0 == 0)) &&
                       (!e.vt || (size_t)((M + e.vt_T - 1) / e.vt_T + 4096) * e.vt_heads * e.vt_hd * e.vt_Tpad < (1ull << 32));
  if (al && scatter) return launch_t256p<T, NPL, 1>(A, lda, W, ldw, M, N, K, e, st);
  if (al && plain && e.act == LA_ACT_GELU && !e.res && !e.out32 && e.out16 && !e.vt) return launch_t256p<T, NPL, 2>(A, lda, W, ldw, M, N, K, e, st);
  // a residual that repeats every res_mod rows (the patch embedding's position table: one row per token of the image) stays on the
  // persistent four-wave kernel when whole 256-row tiles sit inside one period - its direct epilogue takes the residual rows modulo
  const bool w4_resmod = NPL == 1 && e.map == LA_MAP_NONE && e.res_mod > 0 && (e.res_mod % 256) == 0 && (M % 256) == 0 && e.res && (g_gemm_variant & 0xff) == 2 &&
                         (K % 64) == 0 && K >= 128 && (e.a_kmod == 0 || (e.a_kmod % 64) == 0) && !((g_gemm_variant >> 8) & 1);
  if (al && (plain || w4_resmod) && e.act == LA_ACT_NONE && e.out32 && !e.vt && (e.ld32 % 4) == 0 && (!e.res || (e.ldr % 4) == 0)) {
    // (fp32 atomics from the accumulator layout instead of the read-modify-write through the slab were measured in round 4 and are
    // slower: profiles/r04_notes.md 1)
    return launch_t256p<T, NPL, 3>(A, lda, W, ldw, M, N, K, e, st);
  }
  if (plain && e.act == LA_ACT_NONE && !e.res && !e.out32 && e.out16) launch_t256_epi<T, NPL, 1>(A, lda, W, ldw, M, N, K, e, st);
  else if (plain && e.act == LA_ACT_GELU && !e.res && !e.out32 && e.out16 && !e.vt) launch_t256_epi<T, NPL, 2>(A, lda, W, ldw, M, N, K, e, st);
  else if (plain && e.act == LA_ACT_NONE && e.res && e.out32 && !e.vt) launch_t256_epi<T, NPL, 3>(A, lda, W, ldw, M, N, K, e, st);
  else launch_t256_epi<T, NPL, 0>(A, lda, W, ldw, M, N, K, e, st);
}

template <typename T, int BM_>
static void launch_fast(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, hipStream_t st) {
  constexpr int LDS = fast_lds_bytes<BM_>();
  static unsigned long long attr_mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(gemm_dma_kernel<T, BM_>), LDS, attr_mask);
  const int ntm = (M + BM_ - 1) / BM_, ntn = (N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_dma_kernel<T, BM_>), dim3(ntm * ntn), dim3(BM_ * 2), LDS, st, reinterpret_cast<const T*>(A), lda,
                     reinterpret_cast<const T*>(W), ldw, M, N, K, e, tile_group_m());
}

static bool fast_ok(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if ((K % 64) != 0) return false;
  if ((size_t)M * lda * 2 >= (1ull << 32) || (size_t)N * ldw * 2 >= (1ull << 32)) return false;   // 32-bit DMA offsets
  if (!al16(A) || !al16(W)) return false;
  return epi_vec_ok(N, e, 2);
}


// =================================================================================================================
// fp32 path (dt == LA_F32): exact-fp32 MFMA v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain, 157 TF/s peak).
// Same 128 x 128 tile / 4 waves / register-staged double buffer as gemm_nt_kernel with BK = 32 floats (128-byte
// rows, identical swizzle).  A 16-byte chunk holds k = 4c..4c+3; lane half fh consumes k = 2s + fh of MFMA step s.
// Used for the decoder stages only (a few GFLOP per episode), so simplicity wins over the last TF/s.
// =================================================================================================================
constexpr int BKF = 32;

// Implicit im2col for a 3x3 / pad 1 convolution over an NHWC fp32 map with Cin % 32 == 0: k-tile kt covers input
// channels [32*(kt % cpt), +32) of tap kt / cpt (cpt = Cin / 32); A row = output pixel (b, y, x).
struct ConvA {
  int enabled, H, W, Cin;
};

template <int BN_>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Wt, int ldw,
                                                          int M, int N, int K, LaGemmEpilogue e, ConvA cv, int vec_epi) {
  // wave layout: BN_ = 128 -> 2 x 2 waves of 64 x 64; BN_ = 32 -> 4 x 1 waves of 32 x 32
  constexpr int TI = (BN_ == 128) ? 2 : 1;   // 32-row MFMA tiles per wave
  constexpr int TJ = (BN_ == 128) ? 2 : 1;
  constexpr int WROWS = TI * 32, WCOLS = TJ * 32;
  constexpr int NWR = BN_ / 32;              // W-tile 32-row groups loaded per k-tile (4 or 1)
  constexpr int STAGE = (BM + BN_) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (BN_ == 128) ? (wave >> 1) : wave, wn = (BN_ == 128) ? (wave & 1) : 0;
  const int ntn = (N + BN_ - 1) / BN_, ntm = (M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN_;
  const int lc = tid & 7, lr = tid >> 3;
  const float* a_ptr[4];
  const float* w_ptr[NWR];
  int py[4], px[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = min(m0 + lr + 32 * i, M - 1);
    if (cv.enabled) {
      px[i] = row % cv.W;
      py[i] = (row / cv.W) % cv.H;
      a_ptr[i] = A + (size_t)row * cv.Cin + lc * 4;     // centre pixel
    } else {
      px[i] = py[i] = 0;
      a_ptr[i] = A + (size_t)row * lda + lc * 4;
    }
  }
#pragma unroll
  for (int i = 0; i < NWR; ++i) w_ptr[i] = Wt + (size_t)min(n0 + lr + 32 * i, N - 1) * ldw + lc * 4;
  uint4 ra_[4], rw_[NWR];
  const int cpt = cv.enabled ? cv.Cin / 32 : 1;
  auto gload = [&](int kt) {
    const int k0 = kt * BKF;
    const bool ok = (k0 + lc * 4) < K;
    if (cv.enabled) {
      const int tap = kt / cpt, c0 = (kt % cpt) * 32;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int yy = py[i] + dy, xx = px[i] + dx;
        const bool in = (yy >= 0) && (yy < cv.H) && (xx >= 0) && (xx < cv.W);
        ra_[i] = in ? *reinterpret_cast<const uint4*>(a_ptr[i] + (long)(dy * cv.W + dx) * cv.Cin + c0) : make_uint4(0, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) ra_[i] = ok ? *reinterpret_cast<const uint4*>(a_ptr[i] + k0) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NWR; ++i) rw_[i] = ok ? *reinterpret_cast<const uint4*>(w_ptr[i] + k0) : make_uint4(0, 0, 0, 0);
  };
  auto swrite = [&](int stage) {
    char* sa = smem + stage * STAGE;
    char* sw = sa + BM * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(sa + swz_off(lr + 32 * i, lc)) = ra_[i];
#pragma unroll
    for (int i = 0; i < NWR; ++i) *reinterpret_cast<uint4*>(sw + swz_off(lr + 32 * i, lc)) = rw_[i];
  };
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = (K + BKF - 1) / BKF;
  gload(0);
  swrite(0);
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
    const char* sa = smem + (kt & 1) * STAGE;
    const char* sw = sa + BM * 128;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      // lane half fh owns floats {2fh, 2fh+1} of the 16-byte chunk (any k permutation is fine as long as A and W agree):
      // one 8-byte LDS read per operand and NO runtime vector indexing (which hipcc lowers through scratch memory).
      float2 af[TI], wf[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const float2*>(sa + swz_off(wm * WROWS + i * 32 + fr, c) + fh * 8);
#pragma unroll
      for (int j = 0; j < TJ; ++j) wf[j] = *reinterpret_cast<const float2*>(sw + swz_off(wn * WCOLS + j * 32 + fr, c) + fh * 8);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, wf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, wf[j].y, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) swrite((kt + 1) & 1);
    __syncthreads();
  }
  if (vec_epi) {   // block-uniform: LDS-staged 16-byte epilogue (the stage buffers are free after the last barrier)
    epilogue_lds<float, TI, TJ, BN_, 256>(reinterpret_cast<float*>(smem), acc, 1, 0, wm * WROWS, wn * WCOLS, m0, n0, M, N, e, tid);
    return;
  }
  const RowMap rm{e.map, e.p0, e.p1, e.p2, e.p3, e.p4};
  float* outT = reinterpret_cast<float*>(e.out16);
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
    const int col = n0 + wn * WCOLS + tj * 32 + fr;
    if (col >= N) continue;
    int dcol = col, row_add = 0, bcol = col;
    if (e.map == LA_MAP_CONVT2X2) {
      const int kyx = col / e.p2;
      dcol = col % e.p2;
      bcol = dcol;
      row_add = (kyx >> 1) * 2 * e.p0 + (kyx & 1);
    }
    const float bias = e.bias ? e.bias[bcol] : 0.f;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WROWS + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (row >= M) continue;
        float v = acc[ti][tj][r] + bias;
        if (e.act == LA_ACT_GELU) v = gelu_erf(v);
        else if (e.act == LA_ACT_RELU) v = fmaxf(v, 0.f);
        int drow = map_row(rm, row);
        if (drow < 0) continue;
        drow += row_add;
        if (e.res) {
          const int rr = e.res_mod ? drow % e.res_mod : drow;
          v += e.res[(size_t)rr * e.ldr + dcol];
        }
        if (e.out32) e.out32[(size_t)drow * e.ld32 + dcol] = v;
        if (outT) outT[(size_t)drow * e.ld16 + dcol] = v;
      }
    }
  }
}

template <int BN_>
static void launch_f32(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, const ConvA& cv,
                       hipStream_t st) {
  constexpr int EPI = 128 * (BN_ + 4) * 4;
  constexpr int LDS = (2 * (BM + BN_) * 128) > EPI ? (2 * (BM + BN_) * 128) : EPI;
  const int vec_epi = epi_vec_ok(N, e, 4) ? 1 : 0;
  static unsigned long long attr_mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(gemm_f32_kernel<BN_>), LDS, attr_mask);
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN_ - 1) / BN_;
  hipLaunchKernelGGL(gemm_f32_kernel<BN_>, dim3(ntm * ntn), dim3(256), LDS, st, reinterpret_cast<const float*>(A), lda,
                     reinterpret_cast<const float*>(W), ldw, M, N, K, e, cv, vec_epi);
}

// =================================================================================================================
// skinny GEMM (M <= 32 rows: decoder tokens / class prototypes): y[m][n] = sum_k x[m][k] W[n][k] on the VALU in fp32.
// One wave per 4 output columns; lanes stride over K in float4 / 8-half chunks (W streamed once, coalesced; the few x
// rows stay in L1), per-lane partial sums for every (m, column) are reduced across the wave at the end.
// =================================================================================================================
template <typename T> struct Load8;     // 8 consecutive k values as floats
template <> struct Load8<float> {
  static __device__ __forceinline__ void ld(const float* p, float* o) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  }
};
template <typename T> struct Load8 {
  static __device__ __forceinline__ void ld(const T* p, float* o) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    o[0] = unpack_lo<T>(v.x); o[1] = unpack_hi<T>(v.x); o[2] = unpack_lo<T>(v.y); o[3] = unpack_hi<T>(v.y);
    o[4] = unpack_lo<T>(v.z); o[5] = unpack_hi<T>(v.z); o[6] = unpack_lo<T>(v.w); o[7] = unpack_hi<T>(v.w);
  }
};

constexpr int SK_COLS = 4;

template <typename T, int MR, int KW>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const T* __restrict__ A, int lda, const T* __restrict__ Wt, int ldw, int M, int N,
                                                          int K, LaGemmEpilogue e) {
  // A wave owns SK_COLS output columns x MR rows over a share of K; lanes split a 512-wide k-slab (8 values each).
  //   KW = 1 (short K): the four waves of a workgroup take four column groups, each walks all of K;
  //   KW = 4 (K >= 1024): the four waves share ONE column group and take the k-slabs w, w + 4, ... - with K <= 2048 a single batch of
  //                        loads and MR * 32 FMAs per lane instead of a four-deep serial loop.
  // Partials are folded inside 16-lane rows with DPP (no ds_bpermute chains); the 4 rows (x 4 waves) of every output meet in LDS
  // and are summed in a fixed order.
  __shared__ float part[4][MR * SK_COLS][5];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (KW == 4 ? blockIdx.x : blockIdx.x * 4 + wave) * SK_COLS;
  // blockIdx.y: chunk of MR rows (a few hundred token rows - many prompt pairs - are still far too few for an MFMA tile grid:
  // 240 x 256 x 2048 is FOUR 128 x 128 tiles; here it is 8 chunks x 64 workgroups re-streaming a 2 MB weight from L2)
  const int m_base = blockIdx.y * MR;
  A += (size_t)m_base * lda;
  M -= m_base;
  float acc[MR][SK_COLS];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int c = 0; c < SK_COLS; ++c) acc[m][c] = 0.f;
  if (n0 < N) {
    for (int k = (KW == 4 ? wave * 512 : 0) + lane * 8; k < K; k += KW * 512) {
      float w[SK_COLS][8];
#pragma unroll
      for (int c = 0; c < SK_COLS; ++c) Load8<T>::ld(Wt + (size_t)min(n0 + c, N - 1) * ldw + k, w[c]);
      // no "if (m < M)" around the loads: hipcc would branch around every load and wait for each one in turn (one full memory
      // latency per row); rows >= M re-read row M-1 and are simply not stored.
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        float x[8];
        Load8<T>::ld(A + (size_t)min(m, M - 1) * lda + k, x);
#pragma unroll
        for (int c = 0; c < SK_COLS; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[m][c] = fmaf(x[j], w[c][j], acc[m][c]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int c = 0; c < SK_COLS; ++c) {
      const float r = row16_sum(acc[m][c]);
      if ((lane & 15) == 0) part[wave][m * SK_COLS + c][lane >> 4] = r;
    }
  __syncthreads();
  T* outT = reinterpret_cast<T*>(e.out16);
  constexpr int OUTS = MR * SK_COLS * (KW == 4 ? 1 : 4);
  for (int o = threadIdx.x; o < OUTS; o += 256) {
    const int idx = o % (MR * SK_COLS), ow = o / (MR * SK_COLS);               // ow: owning wave (KW = 1 only)
    const int ml = idx / SK_COLS;
    const int col = (KW == 4 ? blockIdx.x : blockIdx.x * 4 + ow) * SK_COLS + idx % SK_COLS;
    if (ml < M && col < N) {
      float v = 0.f;
      if (KW == 4) {
#pragma unroll
        for (int w = 0; w < 4; ++w) v += (part[w][idx][0] + part[w][idx][1]) + (part[w][idx][2] + part[w][idx][3]);
      } else {
        v = (part[ow][idx][0] + part[ow][idx][1]) + (part[ow][idx][2] + part[ow][idx][3]);
      }
      const int m = m_base + ml;
      v += e.bias ? e.bias[col] : 0.f;
      if (e.act == LA_ACT_GELU) v = gelu_erf(v);
      else if (e.act == LA_ACT_RELU) v = fmaxf(v, 0.f);
      if (e.res) v += e.res[(size_t)(e.res_mod ? m % e.res_mod : m) * e.ldr + col];
      if (e.out32) e.out32[(size_t)m * e.ld32 + col] = v;
      if (outT) outT[(size_t)m * e.ld16 + col] = (T)v;
    }
  }
}

template <typename T, int KW>
static void launch_skinny_kw(const T* a, int lda, const T* w, int ldw, int M, int N, int K, const LaGemmEpilogue& e, hipStream_t st) {
  const int groups = (N + SK_COLS - 1) / SK_COLS;
  const dim3 grid(KW == 4 ? groups : (groups + 3) / 4, M > 32 ? (M + 31) / 32 : 1), block(256);
  if (M <= 8) hipLaunchKernelGGL((gemm_skinny_kernel<T, 8, KW>), grid, block, 0, st, a, lda, w, ldw, M, N, K, e);
  else if (M <= 16) hipLaunchKernelGGL((gemm_skinny_kernel<T, 16, KW>), grid, block, 0, st, a, lda, w, ldw, M, N, K, e);
  else hipLaunchKernelGGL((gemm_skinny_kernel<T, 32, KW>), grid, block, 0, st, a, lda, w, ldw, M, N, K, e);
}

template <typename T>
static void launch_skinny(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, hipStream_t st) {
  const T* a = reinterpret_cast<const T*>(A);
  const T* w = reinterpret_cast<const T*>(W);
  if (K >= 1024) launch_skinny_kw<T, 4>(a, lda, w, ldw, M, N, K, e, st);
  else launch_skinny_kw<T, 1>(a, lda, w, ldw, M, N, K, e, st);
}

// =================================================================================================================
// A few hundred fp32 rows (128 < M <= 512: the decoder tokens of many prompt pairs - 240 x 2048 x 256 is the token MLP of cfg4) on the
// exact-fp32 MFMA with 32 x 32 tiles, one per WAVE, operands straight from global memory: 512 wave tiles fill the chip where a 128 x 128
// grid has 32 workgroups, and the row-chunked VALU kernel above spends 71 us on this shape with half its lanes idle (K = 256 is 32 lanes
// of 8).  A lane loads float4s - k = 8 j + 4 fh .. + 3 of chunk j for its row of A and of W - and the four elements feed four MFMA steps:
// any assignment of k to (step, half-wave) is fine as long as A and W agree.
//   KS = 1: the four waves of a workgroup own 2 x 2 tiles, each over all of K.
//   KS = 4 (K >= 1024): the four waves split the 8-wide k-chunks of ONE tile and meet in LDS (fixed summation order).
// =================================================================================================================
template <int KS>
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Wt, int ldw, int M,
                                                             int N, int K, LaGemmEpilogue e) {
  __shared__ float part[KS == 4 ? 4 : 1][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, fh = lane >> 5;
  const int tm = KS == 4 ? blockIdx.y : blockIdx.y * 2 + (wave >> 1), tn = KS == 4 ? blockIdx.x : blockIdx.x * 2 + (wave & 1);
  const bool active = tm * 32 < M && tn * 32 < N;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (active) {
    const float* ap = A + (size_t)min(tm * 32 + fr, M - 1) * lda + 4 * fh;
    const float* wp = Wt + (size_t)min(tn * 32 + fr, N - 1) * ldw + 4 * fh;
    const int nchunk = K >> 3;
    // batches of 8 chunks: 16 independent 16-byte loads in flight, then 32 MFMAs
    for (int j0 = (KS == 4 ? wave : 0); j0 < nchunk; j0 += 8 * KS) {
      float4 a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = min(j0 + u * KS, nchunk - 1);       // (beyond K: a valid address, the products are skipped below)
        a[u] = *reinterpret_cast<const float4*>(ap + 8 * j);
        b[u] = *reinterpret_cast<const float4*>(wp + 8 * j);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (j0 + u * KS < nchunk) {                          // wave-uniform
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
        }
      }
    }
  }
  if (KS == 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc[r];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = (part[0][r][lane] + part[1][r][lane]) + (part[2][r][lane] + part[3][r][lane]);
  }
  if (!active) return;
  const int col = tn * 32 + fr;
  if (col >= N) return;
  float* outT = reinterpret_cast<float*>(e.out16);
  const float bias = e.bias ? e.bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
    if (m >= M) continue;
    float v = acc[r] + bias;
    if (e.act == LA_ACT_GELU) v = gelu_erf(v);
    else if (e.act == LA_ACT_RELU) v = fmaxf(v, 0.f);
    if (e.res) v += e.res[(size_t)(e.res_mod ? m % e.res_mod : m) * e.ldr + col];
    if (e.out32) e.out32[(size_t)m * e.ld32 + col] = v;
    if (outT) outT[(size_t)m * e.ld16 + col] = v;
  }
}

static void launch_f32_small(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e, hipStream_t st) {
  const int tm = (M + 31) / 32, tn = (N + 31) / 32;
  if (K >= 1024)
    hipLaunchKernelGGL(gemm_f32_small_kernel<4>, dim3(tn, tm), dim3(256), 0, st, reinterpret_cast<const float*>(A), lda,
                       reinterpret_cast<const float*>(W), ldw, M, N, K, e);
  else
    hipLaunchKernelGGL(gemm_f32_small_kernel<1>, dim3((tn + 1) / 2, (tm + 1) / 2), dim3(256), 0, st, reinterpret_cast<const float*>(A), lda,
                       reinterpret_cast<const float*>(W), ldw, M, N, K, e);
}

template <typename T>
static int launch_gemm(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e,
                       hipStream_t st) {
  static unsigned long long attr_mask = 0;
  ensure_dyn_lds(reinterpret_cast<const void*>(gemm_nt_kernel<T>), 2 * STAGE_BYTES, attr_mask);
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  hipLaunchKernelGGL(gemm_nt_kernel<T>, dim3(ntm * ntn), dim3(256), 2 * STAGE_BYTES, st, reinterpret_cast<const T*>(A), lda,
                     reinterpret_cast<const T*>(W), ldw, M, N, K, e);
  return 0;
}

}  // namespace la

extern "C" int la_gemm_variant(int v) {
  const int prev = la::g_gemm_variant;
#ifdef LA_DEBUG
  if (v >= 0) la::g_gemm_variant = v;       // bit 8 no stores, bit 10 seam stamps (la_dbg_gemm_stamps), bit 23 no epilogue
#else
  if (v >= 0 && v <= 2) la::g_gemm_variant = v;       // the product library only knows the bit-identical main loops
#endif
  return prev;
}

#ifdef LA_DEBUG
// seam timeline of workgroup 0 of the last la_gemm launch made with la_gemm_variant bit 10: 8 waves x 64 entries of (s_memtime << 8 | tag)
extern "C" int la_dbg_gemm_stamps(unsigned long long* host_out) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(la::g_dbg_stamps), sizeof(unsigned long long) * 8 * la::LA_DBG_NSTAMP);
}
extern "C" int la_dbg_gemm_stamps_clear() {
  static unsigned long long z[8 * la::LA_DBG_NSTAMP] = {0};
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(la::g_dbg_stamps), z, sizeof(z));
}
#endif

// (the persistent kernel pays off from one tile per CU of the CURRENT device - not a constant, not cached across devices)
extern "C" int la_gemm_fused_act_ok(int M, int N, int K) {
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
  return M > 0 && (N % 256) == 0 && (K % 64) == 0 && K >= 128 && (long)((M + 255) / 256) * (N / 256) >= ncu && (la::g_gemm_variant & 0xff) == 2;
}

extern "C" int la_gemm(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue* epi, int dt,
                       void* stream) {
  LA_CHECK_ARG(A && W && epi, "la_gemm: null pointer");
  LA_CHECK_ARG(M > 0 && N > 0 && K > 0, "la_gemm: bad shape M=%d N=%d K=%d", M, N, K);
  const int kq = (dt == LA_F32) ? 4 : 8;
  LA_CHECK_ARG((K % kq) == 0 && (lda % kq) == 0 && (ldw % kq) == 0, "la_gemm: K, lda, ldw must be multiples of %d (K=%d lda=%d ldw=%d)", kq, K,
               lda, ldw);
  LA_CHECK_ARG(epi->out32 || epi->out16 || epi->vt, "la_gemm: no output");
  LA_CHECK_ARG(epi->a_kmod == 0 || (dt != LA_F32 && epi->a_kmod > 0 && (epi->a_kmod % 64) == 0 && epi->a_kmod <= K && lda >= epi->a_kmod && M > 32),
               "la_gemm: a_kmod=%d must be a multiple of 64, <= K=%d and <= lda=%d (16-bit operands, M > 32)", epi->a_kmod, K, lda);
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16 || dt == LA_F32, "la_gemm: bad dtype %d", dt);
  LA_CHECK_ARG(epi->amap == LA_MAP_NONE || (epi->amap == LA_MAP_WINDOW_PART && epi->map == LA_MAP_NONE && dt != LA_F32) ||
                   (epi->amap == LA_MAP_CONV3X3 && epi->map == LA_MAP_NONE && dt == LA_F16 && epi->a_kmod == 0 && epi->p1 > 0 && (epi->p1 % 64) == 0 &&
                    K == 27 * epi->p1 && epi->p2 == lda && lda == 2 * epi->p1 && epi->p0 > 2 && (N % 256) == 0 && !epi->vt && epi->ksplit == 0 && M > 512),
               "la_gemm: amap must be LA_MAP_NONE, LA_MAP_WINDOW_PART (16-bit operands, no output map) or LA_MAP_CONV3X3 (fp16 plane pairs, p0 = padded "
               "width, p1 = C %% 64 == 0, p2 = lda = 2 C, K = 27 C, N %% 256 == 0), got amap=%d map=%d dt=%d", epi->amap, epi->map, dt);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  LA_CHECK_ARG(epi->act != LA_ACT_GELU_BWD || epi->aux16, "la_gemm: LA_ACT_GELU_BWD needs aux16 (the saved pre-activation)");
  if (epi->aux16 && !epi->nstat_out) {
    // the training forms of the MLP's GELU (see LaGemmEpilogue.aux16): the direct epilogue of the persistent four-wave kernel only
    LA_CHECK_ARG(epi->act == LA_ACT_GELU || epi->act == LA_ACT_GELU_BWD, "la_gemm: aux16 goes with LA_ACT_GELU (written) or LA_ACT_GELU_BWD (read)");
    LA_CHECK_ARG(dt != LA_F32 && la_gemm_fused_act_ok(M, N, K) && la::fast_ok(A, lda, W, ldw, M, N, K, *epi),
                 "la_gemm: aux16 needs 16-bit operands and a shape la_gemm_fused_act_ok() accepts (M=%d N=%d K=%d)", M, N, K);
    LA_CHECK_ARG(epi->out16 && !epi->out32 && !epi->res && !epi->vt && epi->map == LA_MAP_NONE && epi->amap == LA_MAP_NONE && epi->a_kmod == 0 &&
                     epi->ksplit == 0 && (epi->ld16 % 8) == 0 && (epi->ldaux % 8) == 0 && epi->ld16 >= N && epi->ldaux >= N &&
                     ((reinterpret_cast<uintptr_t>(epi->out16) | reinterpret_cast<uintptr_t>(epi->aux16)) & 15) == 0,
                 "la_gemm: aux16 forms write out16 only (no residual / fp32 output / maps / V^T / planes), rows 16-byte aligned");
    LA_CHECK_ARG(epi->act == LA_ACT_GELU || !epi->bias, "la_gemm: LA_ACT_GELU_BWD takes no bias");
    const int gm = la::tile_group_m(N >= 2560 ? 8 : 2);
    if (epi->act == LA_ACT_GELU) {
      if (dt == LA_F16) la::launch_t256w_fused<la::f16_t, 5>(A, lda, W, ldw, M, N, K, *epi, gm, st);
      else la::launch_t256w_fused<la::bf16_t, 5>(A, lda, W, ldw, M, N, K, *epi, gm, st);
    } else {
      if (dt == LA_F16) la::launch_t256w_fused<la::f16_t, 6>(A, lda, W, ldw, M, N, K, *epi, gm, st);
      else la::launch_t256w_fused<la::bf16_t, 6>(A, lda, W, ldw, M, N, K, *epi, gm, st);
    }
    LA_CHECK_LAUNCH("la_gemm");
    return 0;
  }
  if (epi->nstat_out || epi->nstat_in || epi->rvec) {
    // LayerNorm folded into its neighbour GEMMs (see LaGemmEpilogue.nstat_out): the direct epilogue of the persistent four-wave kernel only
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    LA_CHECK_ARG(dt == LA_F16 && (N % 256) == 0 && K >= 128 && (epi->a_kmod == 0 || (epi->a_kmod % 64) == 0) &&
                     la::fast_ok(A, lda, W, ldw, M, N, K, *epi) && (la::g_gemm_variant & 0xff) == 2,
                 "la_gemm: nstat_out / nstat_in need fp16 operands, N %% 256 == 0, K %% 64 == 0, K >= 128, 16-byte aligned rows (M=%d N=%d K=%d)", M,
                 N, K);
    LA_CHECK_ARG(!epi->vt && epi->map == LA_MAP_NONE && epi->amap == LA_MAP_NONE && epi->ksplit == 0 && !(epi->nstat_out && epi->nstat_in) &&
                     (!epi->aux16 || (epi->nstat_out && (epi->ldaux % 8) == 0 && epi->ldaux >= N && (reinterpret_cast<uintptr_t>(epi->aux16) & 15) == 0)),
                 "la_gemm: nstat_out / nstat_in take no row maps / V^T / ksplit, not both at once; aux16 only with nstat_out (the lo plane, 16-byte aligned rows)");
    const int gm = la::tile_group_m(N >= 2560 ? 8 : 2);
    if (epi->nstat_in) {
      LA_CHECK_ARG(epi->ncol && epi->out16 && !epi->out32 && !epi->res && !epi->rvec && (epi->act == LA_ACT_NONE || epi->act == LA_ACT_GELU) &&
                       (epi->ld16 % 8) == 0 && epi->ld16 >= N && al16(epi->out16) && al16(epi->nstat_in) && al16(epi->ncol) && epi->a_kmod == 0,
                   "la_gemm: nstat_in writes out16 only (act NONE / GELU), needs ncol, one weight plane");
      if (epi->act == LA_ACT_GELU) la::launch_t256w_fused<la::f16_t, 9>(A, lda, W, ldw, M, N, K, *epi, gm, st);
      else la::launch_t256w_fused<la::f16_t, 8>(A, lda, W, ldw, M, N, K, *epi, gm, st);
    } else if (epi->nstat_out && !epi->out32 && epi->res && epi->aux16) {
      // fp32 residual in (the position table of the patch embedding), plane pairs out, no fp32 matrix at all
      LA_CHECK_ARG(epi->out16 && epi->act == LA_ACT_NONE && (epi->ld16 % 8) == 0 && epi->ld16 >= N && al16(epi->out16) && (epi->ldr % 4) == 0 &&
                       (reinterpret_cast<uintptr_t>(epi->nstat_out) & 7) == 0 && !epi->rvec,
                   "la_gemm: nstat_out with a residual and no out32 writes plane pairs only (out16 + aux16), no group vector");
      LA_CHECK_ARG(epi->res_mod == 0 || ((epi->res_mod % 256) == 0 && (M % 256) == 0),
                   "la_gemm: nstat_out with a periodic residual needs res_mod %% 256 == 0 and M %% 256 == 0 (res_mod=%d M=%d)", epi->res_mod, M);
      la::launch_t256w_fused<la::f16_t, 7>(A, lda, W, ldw, M, N, K, *epi, gm, st);
    } else if (epi->nstat_out && !epi->out32 && !epi->res) {
      // the stream as fp16 plane pairs, read-modify-written in place: out16 = hi plane, aux16 = lo plane (see LaGemmEpilogue.nstat_out)
      LA_CHECK_ARG(epi->out16 && epi->aux16 && epi->act == LA_ACT_NONE && (epi->ld16 % 8) == 0 && (epi->ldaux % 8) == 0 && epi->ld16 >= N &&
                       epi->ldaux >= N && al16(epi->out16) && al16(epi->aux16) && (reinterpret_cast<uintptr_t>(epi->nstat_out) & 7) == 0 && epi->res_mod == 0,
                   "la_gemm: nstat_out without out32 / res updates a plane-pair stream in place: out16 (hi) and aux16 (lo), 16-byte aligned rows");
      LA_CHECK_ARG(!epi->rvec || (epi->rvec_rpg > 0 && al16(epi->rvec) && ((epi->rvec_rpg % 256) == 0 || epi->rvec_rpg >= 128)),
                   "la_gemm: rvec needs a 16-byte aligned vector and groups of whole 256-row tiles or of at least 128 rows (rvec_rpg=%d)", epi->rvec_rpg);
      if (epi->rvec && (epi->rvec_rpg % 256) != 0) la::launch_t256w_fused<la::f16_t, 12>(A, lda, W, ldw, M, N, K, *epi, gm, st);
      else la::launch_t256w_fused<la::f16_t, 11>(A, lda, W, ldw, M, N, K, *epi, gm, st);
    } else {
      LA_CHECK_ARG(epi->nstat_out && epi->out32 && epi->out16 && epi->act == LA_ACT_NONE && (epi->ld16 % 8) == 0 && epi->ld16 >= N &&
                       (epi->ld32 % 4) == 0 && epi->ld32 >= N && al16(epi->out16) && al16(epi->out32) && (!epi->res || (epi->ldr % 4) == 0) &&
                       (reinterpret_cast<uintptr_t>(epi->nstat_out) & 7) == 0,
                   "la_gemm: nstat_out goes with out32 + out16 (no activation), 16-byte aligned rows");
      LA_CHECK_ARG(epi->res_mod == 0 || ((epi->res_mod % 256) == 0 && (M % 256) == 0 && epi->res),
                   "la_gemm: nstat_out with a periodic residual needs res_mod %% 256 == 0 and M %% 256 == 0 (res_mod=%d M=%d)", epi->res_mod, M);
      LA_CHECK_ARG(!epi->rvec || (epi->rvec_rpg > 0 && al16(epi->rvec) && ((epi->rvec_rpg % 256) == 0 || epi->rvec_rpg >= 128)),
                   "la_gemm: rvec needs a 16-byte aligned vector and groups of whole 256-row tiles or of at least 128 rows (rvec_rpg=%d)", epi->rvec_rpg);
      if (epi->rvec && (epi->rvec_rpg % 256) != 0) la::launch_t256w_fused<la::f16_t, 10>(A, lda, W, ldw, M, N, K, *epi, gm, st);
      else la::launch_t256w_fused<la::f16_t, 7>(A, lda, W, ldw, M, N, K, *epi, gm, st);
    }
    LA_CHECK_LAUNCH("la_gemm");
    return 0;
  }
  // up to 512 fp32 rows (decoder tokens of many prompt pairs): an MFMA grid of 128 x 128 tiles is a handful of workgroups and leaves the
  // chip idle (240 x 256 x 2048: 155 us on four tiles) - 32 x 32 wave tiles (gemm_f32_small_kernel) above 128 rows, the VALU kernel below
  const bool few_rows = M <= 32 || (dt == LA_F32 && M <= 512 && (long)((M + 127) / 128) * ((N + 127) / 128) < 64);
  const bool skinny = few_rows && (K % 8) == 0 && epi->map == LA_MAP_NONE && epi->amap == LA_MAP_NONE && !epi->vt;
  static const char* nosmall = la_dbg_env("LA_NO_F32_SMALL");   // debugging: the row-chunked VALU kernel for every few-row fp32 shape
  // (from 129 rows: up to 128 rows - the per-image vectors of the encoder's token-mean corrections, one row per image - stay on the VALU
  // kernel, whose lane-parallel partial sums round 4 x closer to fp64 than the MFMA's serial chain (9e-8 against 4e-7 on 52 x 768 x 1536);
  // the corrections accumulate over every block of the encoder and a 26- and a 52-image batch must not take different kernels:
  // tests/test_model_gpu.py::test_full_geometry_episode_properties measured 2.7e-4 on the cfg3 logits between the two)
  if (skinny && dt == LA_F32 && M > 128 && !nosmall && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) == 0) {
    la::launch_f32_small(A, lda, W, ldw, M, N, K, *epi, st);       // (K, lda, ldw are multiples of 4 here: float4 operand loads)
    LA_CHECK_LAUNCH("la_gemm");
    return 0;
  }
  if (skinny) {
    if (dt == LA_F32) la::launch_skinny<float>(A, lda, W, ldw, M, N, K, *epi, st);
    else if (dt == LA_F16) la::launch_skinny<la::f16_t>(A, lda, W, ldw, M, N, K, *epi, st);
    else la::launch_skinny<la::bf16_t>(A, lda, W, ldw, M, N, K, *epi, st);
    LA_CHECK_LAUNCH("la_gemm");
    return 0;
  }
  if (dt == LA_F32) {
    LA_CHECK_ARG(!epi->vt, "la_gemm: the transposed-V epilogue is 16-bit only");
    const la::ConvA nocv{0, 0, 0, 0};
    if (N <= 32) la::launch_f32<32>(A, lda, W, ldw, M, N, K, *epi, nocv, st);
    else la::launch_f32<128>(A, lda, W, ldw, M, N, K, *epi, nocv, st);
    LA_CHECK_LAUNCH("la_gemm");
    return 0;
  }
  if (epi->ksplit > 0) {
    // split-K accumulate (weight gradients): out32 += A . W^T with fp32 atomics, K cut into independent chunks
    LA_CHECK_ARG(epi->out32 && !epi->out16 && !epi->res && !epi->bias && !epi->vt && epi->act == LA_ACT_NONE &&
                     (epi->map == LA_MAP_NONE || (epi->map == LA_MAP_GROUP && epi->p0 > 0)) && epi->amap == LA_MAP_NONE && epi->a_kmod == 0,
                 "la_gemm: ksplit accumulates the bare product into out32 (no bias / residual / activation / second output; row map none or LA_MAP_GROUP)");
    LA_CHECK_ARG((N % 256) == 0 && (K % 64) == 0 && K >= 128 && (size_t)M * lda * 2 < (1ull << 32) && (size_t)N * ldw * 2 < (1ull << 32) &&
                     (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0,
                 "la_gemm: ksplit needs N %% 256 == 0, K %% 64 == 0, K >= 128, 16-byte aligned operands below 4 GiB (M=%d N=%d K=%d)", M, N, K);
    if (dt == LA_F16) la::launch_t256q<la::f16_t, 4>(A, lda, W, ldw, M, N, K, *epi, st);
    else la::launch_t256q<la::bf16_t, 4>(A, lda, W, ldw, M, N, K, *epi, st);
    LA_CHECK_LAUNCH("la_gemm");
    return 0;
  }
  static const char* force = la_dbg_env("LA_GEMM_PATH");   // debugging: "v1" (register staged), "2" (128x128), "4" (256x128), "6" (256x256)
  bool fast = la::fast_ok(A, lda, W, ldw, M, N, K, *epi) && !(force && force[0] == 'v');
  if (fast) {
    // measured on MI355X (profiles/r01_gemm_variants.log): the 256x128 / 128x64-per-wave kernel wins by ~5 % on the short-K
    // (K = 768) shapes once there are >= 2 full waves of tiles; the 128x128 kernel wins on long K and small grids.
    const long tiles256 = (long)((M + 255) / 256) * ((N + la::BN - 1) / la::BN);
    bool v4 = (K <= 1536) && (tiles256 >= 512);
    if (force && force[0] == '4') v4 = true;
    if (force && force[0] == '2') v4 = false;
    // long-K shapes with >= 2 full waves of 256 x 256 tiles: the ping-pong kernel (+15 % on 65536x768x3072)
    const long tiles_pp = (long)((M + la::PP_BM - 1) / la::PP_BM) * ((N + la::PP_BN - 1) / la::PP_BN);
    bool pp = (K > 1536) && (tiles_pp >= 512);
    if (force) pp = force[0] == '6';
    pp = pp && (!epi->vt || (epi->vt_col0 % la::PP_BN) == 0);
    // two weight planes against one A ([W_hi | W_lo], a_kmod = K / 2): the 256 x 256 two-plane kernel, which reuses every A fragment
    // for both planes; LA_GEMM_PATH=7 also sends single-plane shapes through its NPL = 1 form (A/B experiments)
    const bool planes2 = epi->a_kmod > 0 && K == 2 * epi->a_kmod;
    // ... and, measured on MI355X (tools/gemm_planes_bench.py), its single-plane form beats the 256 x 128, the 128 x 128 and the older
    // ping-pong kernel on every shape with >= 2 full rounds of 256 x 256 tiles (K = 768: +10-15 %, K = 3072: equal)
    bool t256 = ((planes2 && tiles_pp >= 128) || (!planes2 && tiles_pp >= 512)) && (!epi->vt || (epi->vt_col0 % 256) == 0);
    if (force) t256 = (force[0] == '7') && (!epi->vt || (epi->vt_col0 % 256) == 0);
    if (t256) {
      if (planes2) {
        if (dt == LA_F16) la::launch_t256<la::f16_t, 2>(A, lda, W, ldw, M, N, K / 2, *epi, st);
        else la::launch_t256<la::bf16_t, 2>(A, lda, W, ldw, M, N, K / 2, *epi, st);
      } else {
        if (dt == LA_F16) la::launch_t256<la::f16_t, 1>(A, lda, W, ldw, M, N, K, *epi, st);
        else la::launch_t256<la::bf16_t, 1>(A, lda, W, ldw, M, N, K, *epi, st);
      }
    } else if (pp) {
      if (dt == LA_F16) la::launch_pp<la::f16_t>(A, lda, W, ldw, M, N, K, *epi, st);
      else la::launch_pp<la::bf16_t>(A, lda, W, ldw, M, N, K, *epi, st);
    } else if (v4) {
      if (dt == LA_F16) la::launch_fast4<la::f16_t>(A, lda, W, ldw, M, N, K, *epi, st);
      else la::launch_fast4<la::bf16_t>(A, lda, W, ldw, M, N, K, *epi, st);
    } else {
      if (dt == LA_F16) la::launch_fast<la::f16_t, 128>(A, lda, W, ldw, M, N, K, *epi, st);
      else la::launch_fast<la::bf16_t, 128>(A, lda, W, ldw, M, N, K, *epi, st);
    }
  } else {
    if (dt == LA_F16) la::launch_gemm<la::f16_t>(A, lda, W, ldw, M, N, K, *epi, st);
    else la::launch_gemm<la::bf16_t>(A, lda, W, ldw, M, N, K, *epi, st);
  }
  LA_CHECK_LAUNCH("la_gemm");
  return 0;
}

extern "C" int la_conv3x3_f32(const float* in, int B, int H, int W, int Cin, const float* wt, const float* bias, int Cout, float* out32,
                              void* stream) {
  LA_CHECK_ARG(in && wt && out32, "la_conv3x3_f32: null pointer");
  LA_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && (Cin % 32) == 0 && Cout > 0, "la_conv3x3_f32: Cin=%d must be a multiple of 32", Cin);
  LaGemmEpilogue e{};
  e.bias = bias;
  e.out32 = out32;
  e.ld32 = Cout;
  const la::ConvA cv{1, H, W, Cin};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int M = B * H * W, K = 9 * Cin;
  if (Cout <= 32) la::launch_f32<32>(in, Cin, wt, K, M, Cout, K, e, cv, st);
  else la::launch_f32<128>(in, Cin, wt, K, M, Cout, K, e, cv, st);
  LA_CHECK_LAUNCH("la_conv3x3_f32");
  return 0;
}
