// la_confmat_update: prediction / ground-truth label maps -> accumulated confusion matrices, without the label maps
// leaving HBM (reference experiment/run.py:697-704 = argmax -> to_global_multiclass -> StrictMeanIoU /
// DistributedBinaryJaccardIndex updates, utils/metrics.py:28-53, data/utils.py:567-590).
//   * per batch item b a lookup table lut[b][0..L) turns episode-local labels into dataset labels (the chained
//     torch.where of to_global_multiclass collapsed on the host); labels outside [0, L) - background 0 is inside, the
//     ignore label -100 is not - pass through unchanged.
//   * multiclass matrix (K x K, row = target, column = prediction): pixels with target == ignore are dropped, a label
//     outside [0, K) is counted in counters[0] (torchmetrics would raise; the host turns a non-zero count into an error).
//   * binary matrix (2 x 2): labels > 0 become 1 first (DistributedBinaryJaccardIndex.update), then the same rule.
// Pure HBM streaming, 16 B per pixel.  Label maps are piecewise constant, so every lane first folds its 4 consecutive
// pixels into runs and a wave whose lanes all hold the same single key issues ONE atomic; the K x K histogram lives in LDS
// when it fits (K <= 128) and is flushed with 64-bit global atomics at the end.
#include "la_common.h"
#include "../../include/la_hip.h"

namespace la {

constexpr int CM_PIX = 4;              // pixels per lane per iteration (2 x 16-byte loads per map)
constexpr int CM_LDS_K = 128;          // K*K*4 = 64 KiB

struct CmArgs {
  const long long* pred;
  const long long* gt;
  int B;
  long HW;
  const int* lut;
  int L, K;
  long long ignore;
  unsigned long long* confmat;
  unsigned long long* confbin;
  unsigned long long* counters;
};

template <bool USE_LDS>
__global__ __launch_bounds__(256) void confmat_kernel(CmArgs a) {
  extern __shared__ unsigned int hist[];                 // K*K (+4 binary) when USE_LDS
  const int tid = threadIdx.x;
  const int KK = a.K * a.K;
  unsigned int* hbin = hist + (USE_LDS ? KK : 0);
  if (USE_LDS)
    for (int i = tid; i < KK; i += 256) hist[i] = 0;
  if (tid < 4) hbin[tid] = 0;
  __syncthreads();
  unsigned int bad = 0;

  auto add_mc = [&](int key, unsigned int n) {
    if (USE_LDS) atomicAdd(&hist[key], n);
    else atomicAdd(&a.confmat[key], (unsigned long long)n);
  };
  // one (multiclass key, binary key) run with multiplicity n; key < 0 = not counted
  auto emit_lane = [&](int kmc, int kb, unsigned int n) {
    if (kmc >= 0) add_mc(kmc, n);
    if (kb >= 0) atomicAdd(&hbin[kb], n);
  };
  // same, called with the whole wave converged: when every lane carries the same pair, one atomic serves all 64
  auto emit_wave = [&](int kmc, int kb, unsigned int n) {
    const int f_mc = __builtin_amdgcn_readfirstlane(kmc), f_b = __builtin_amdgcn_readfirstlane(kb);
    if (__ballot(kmc == f_mc && kb == f_b) == ~0ull) {
      unsigned int tot = n;
      for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
      if ((tid & 63) == 0) emit_lane(f_mc, f_b, tot);
    } else {
      emit_lane(kmc, kb, n);
    }
  };

  const long per_item = a.HW;
  const long groups = (per_item + CM_PIX - 1) / CM_PIX;            // groups of CM_PIX pixels never straddle batch items
  const long total = (long)a.B * groups;
  const long stride = (long)gridDim.x * 256;
  const long iters = (total + stride - 1) / stride;                 // same trip count for every lane: waves stay converged
  for (long it = 0; it < iters; ++it) {
    const long gidx = it * stride + (long)blockIdx.x * 256 + tid;
    const bool live = gidx < total;
    const long gi = live ? gidx : 0;
    const int b = (int)(gi / groups);
    const long p0 = (gi % groups) * CM_PIX;
    const long base = (long)b * per_item + p0;
    const int npix = live ? (int)min((long)CM_PIX, per_item - p0) : 0;
    long long pv[CM_PIX], tv[CM_PIX];
    if (npix == CM_PIX && ((base & 1) == 0)) {                       // 16-byte aligned pairs
      const longlong2 p01 = *reinterpret_cast<const longlong2*>(a.pred + base), p23 = *reinterpret_cast<const longlong2*>(a.pred + base + 2);
      const longlong2 t01 = *reinterpret_cast<const longlong2*>(a.gt + base), t23 = *reinterpret_cast<const longlong2*>(a.gt + base + 2);
      pv[0] = p01.x; pv[1] = p01.y; pv[2] = p23.x; pv[3] = p23.y;
      tv[0] = t01.x; tv[1] = t01.y; tv[2] = t23.x; tv[3] = t23.y;
    } else {
#pragma unroll
      for (int i = 0; i < CM_PIX; ++i) {
        pv[i] = (i < npix) ? a.pred[base + i] : 0;
        tv[i] = (i < npix) ? a.gt[base + i] : 0;
      }
    }
    const int* lut = a.lut ? a.lut + (size_t)b * a.L : nullptr;
    int run_mc = -1, run_b = -1;
    unsigned int run_n = 0;
    bool have = false;
#pragma unroll
    for (int i = 0; i < CM_PIX; ++i) {
      int kmc = -1, kb = -1;
      if (i < npix) {
        long long p = pv[i], t = tv[i];
        if (lut) {
          if (p >= 0 && p < a.L) p = lut[p];
          if (t >= 0 && t < a.L) t = lut[t];
        }
        if (t != a.ignore) {
          if (p < 0 || p >= a.K || t < 0 || t >= a.K) ++bad;
          else kmc = (int)t * a.K + (int)p;
        }
        const long long pb = p > 0 ? 1 : p, tb = t > 0 ? 1 : t;
        if (tb != a.ignore) {
          if (pb < 0 || tb < 0) ++bad;
          else kb = (int)tb * 2 + (int)pb;
        }
      }
      if (have && kmc == run_mc && kb == run_b) {
        ++run_n;
      } else {
        if (have) emit_lane(run_mc, run_b, run_n);       // (divergent: lanes close runs at different i)
        run_mc = kmc; run_b = kb; run_n = 1; have = true;
      }
    }
    emit_wave(run_mc, run_b, run_n);                      // converged again: every lane has exactly one open run
  }
  __syncthreads();
  if (USE_LDS)
    for (int i = tid; i < KK; i += 256)
      if (hist[i]) atomicAdd(&a.confmat[i], (unsigned long long)hist[i]);
  if (tid < 4 && hbin[tid]) atomicAdd(&a.confbin[tid], (unsigned long long)hbin[tid]);
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((tid & 63) == 0 && bad) atomicAdd(&a.counters[0], (unsigned long long)bad);
}

}  // namespace la

extern "C" int la_confmat_update(const long long* pred, const long long* gt, int B, long HW, const int* lut, int L, int K,
                                 long long ignore_index, unsigned long long* confmat, unsigned long long* confbin,
                                 unsigned long long* counters, void* stream) {
  LA_CHECK_ARG(pred && gt && confmat && confbin && counters, "la_confmat_update: null pointer");
  LA_CHECK_ARG(B > 0 && HW > 0 && K >= 2 && K <= 32768 && (lut == nullptr || L > 0), "la_confmat_update: bad shape B=%d HW=%ld K=%d L=%d", B, HW, K, L);
  la::CmArgs a{pred, gt, B, HW, lut, L, K, ignore_index, confmat, confbin, counters};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long groups = (long)B * ((HW + la::CM_PIX - 1) / la::CM_PIX);
  long blocks = (groups + 255) / 256;
  if (blocks > 512) blocks = 512;          // 2 per CU: the per-block histogram init + flush (K*K entries) must stay small next to the pixels
  if (K <= la::CM_LDS_K) {
    const size_t lds = ((size_t)K * K + 4) * sizeof(unsigned int);
    static unsigned long long attr_mask = 0;
    la::ensure_dyn_lds(reinterpret_cast<const void*>(la::confmat_kernel<true>), (la::CM_LDS_K * la::CM_LDS_K + 4) * (int)sizeof(unsigned int),
                       attr_mask);
    hipLaunchKernelGGL(la::confmat_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, st, a);
  } else {
    hipLaunchKernelGGL(la::confmat_kernel<false>, dim3((unsigned)blocks), dim3(256), 4 * sizeof(unsigned int), st, a);
  }
  LA_CHECK_LAUNCH("la_confmat_update");
  return 0;
}
