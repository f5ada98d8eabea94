// la_gemm: C[M,N] = epilogue(A[M,K] . W[N,K]^T) on gfx950 MFMA (32x32x16, f16/bf16 in, fp32 accumulate).
//
// Workgroup = 256 threads = 4 waves (2 x 2), tile 128 x 128 x 64.  Each wave owns a 64 x 64 sub-tile as
// 2 x 2 MFMA 32x32 accumulators (64 acc VGPRs).  Operand tiles are staged global -> registers -> LDS
// (16-B chunks, XOR-swizzled so every ds_read_b128 lane group is bank-conflict free), double buffered with
// ONE barrier per K-step: the next tile's global loads are issued before the MFMAs of the current one and
// written to the other LDS stage after them.  Tiles are handed to XCDs in contiguous chunks (xcd_remap) so
// the workgroups sharing an A row-panel / the whole W hit the same L2.
#include "la_common.h"
#include "../../include/la_hip.h"

namespace la {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB

struct RowMap {
  int mode, p0, p1, p2, p3, p4;
};

// returns destination row or -1 (dropped)
__device__ __forceinline__ int map_row(const RowMap& m, int row) {
  switch (m.mode) {
    case LA_MAP_GROUP:
      return (row / m.p0) * m.p1 + (row % m.p0) + m.p2;
    case LA_MAP_WINDOW_MERGE: {
      const int ws = m.p0, nwy = m.p1, nwx = m.p2, H = m.p3, W = m.p4;
      const int tok = row % (ws * ws);
      int win = row / (ws * ws);
      const int wx = win % nwx;
      win /= nwx;
      const int wy = win % nwy;
      const int b = win / nwy;
      const int y = wy * ws + tok / ws, x = wx * ws + tok % ws;
      return (y < H && x < W) ? (b * H + y) * W + x : -1;
    }
    case LA_MAP_CONVT2X2: {
      const int W = m.p0, H = m.p1;
      const int x = row % W;
      const int y = (row / W) % H;
      const int b = row / (W * H);
      return (b * 2 * H + 2 * y) * (2 * W) + 2 * x;  // + ky*2W + kx added per column
    }
    default:
      return row;
  }
}

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
    const int ra = min(m0 + lr + 32 * i, M - 1);
    const int rw = min(n0 + lr + 32 * i, N - 1);
    a_ptr[i] = A + (size_t)ra * lda + lc * 8;
    w_ptr[i] = Wt + (size_t)rw * ldw + lc * 8;
  }
  uint4 ra_[4], rw_[4];
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
    const bool ok = (k0 + lc * 8) < K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra_[i] = ok ? *reinterpret_cast<const uint4*>(a_ptr[i] + k0) : make_uint4(0, 0, 0, 0);
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
          const int b = row / e.vt_T, t = row % e.vt_T;
          vt[((size_t)(b * e.vt_heads + vhead) * e.vt_hd + vd) * e.vt_Tpad + t] = (T)v;
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

template <typename T>
static int launch_gemm(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue& e,
                       hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              2 * STAGE_BYTES);
    attr_set = true;
  }
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  hipLaunchKernelGGL(gemm_nt_kernel<T>, dim3(ntm * ntn), dim3(256), 2 * STAGE_BYTES, st, reinterpret_cast<const T*>(A), lda,
                     reinterpret_cast<const T*>(W), ldw, M, N, K, e);
  return 0;
}

}  // namespace la

extern "C" int la_gemm(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const LaGemmEpilogue* epi, int dt,
                       void* stream) {
  LA_CHECK_ARG(A && W && epi, "la_gemm: null pointer");
  LA_CHECK_ARG(M > 0 && N > 0 && K > 0, "la_gemm: bad shape M=%d N=%d K=%d", M, N, K);
  LA_CHECK_ARG((K % 8) == 0 && (lda % 8) == 0 && (ldw % 8) == 0, "la_gemm: K, lda, ldw must be multiples of 8 (K=%d lda=%d ldw=%d)", K,
               lda, ldw);
  LA_CHECK_ARG(epi->out32 || epi->out16 || epi->vt, "la_gemm: no output");
  LA_CHECK_ARG(dt == LA_F16 || dt == LA_BF16, "la_gemm: bad dtype %d", dt);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dt == LA_F16) la::launch_gemm<la::f16_t>(A, lda, W, ldw, M, N, K, *epi, st);
  else la::launch_gemm<la::bf16_t>(A, lda, W, ldw, M, N, K, *epi, st);
  LA_CHECK_LAUNCH("la_gemm");
  return 0;
}
