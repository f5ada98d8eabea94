// Focal loss of the training objective (first link: loss + its gradient w.r.t. the logits).
// Reference: LabelAnythingLoss.logits_loss with components {focal: {weight}} and class_weighting
// (loss/__init__.py:67-89), FocalLoss (loss/focal.py:17-26), get_weight_matrix_from_labels (loss/utils.py:17-43):
//   ce = cross_entropy(x, t) (0 where t == ignore),  pt = exp(-ce),  fl = (1 - pt)^gamma * w[t] * ce,  loss = mean over ALL
//   B*H*W pixels (ignored ones included in the divisor), w[c] = 1 / log(1.1 + count_c / (B*H*W)) for the classes that occur,
//   1 for those that do not, 0 for the ignore label.
// Three launches: label histogram -> fused forward + backward (one pass over the logits, per-workgroup partial sums) ->
// deterministic fold.  d loss / d x_j = scale / N * w[t] * ((1-pt)^g + g * ce * pt * (1-pt)^(g-1)) * (softmax_j - [j == t]).
#include "la_common.h"
#include "../../include/la_hip.h"

#pragma clang fp contract(off)

namespace la {

constexpr int FL_MAXC = 64;

__global__ __launch_bounds__(256) void focal_hist_kernel(const long long* __restrict__ target, long n, int C, long long ignore,
                                                         unsigned long long* __restrict__ counts /*[C + 2]: 0 = ignore, C + 1 = out of range*/) {
  __shared__ unsigned int h[FL_MAXC + 2];
  for (int i = threadIdx.x; i <= C + 1; i += 256) h[i] = 0;
  __syncthreads();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long long t = target[i];
    if (t == ignore) atomicAdd(&h[0], 1u);
    else if (t >= 0 && t < C) atomicAdd(&h[1 + (int)t], 1u);
    else atomicAdd(&h[C + 1], 1u);               // torch's cross_entropy raises on these; here they are counted (and contribute nothing)
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= C + 1; i += 256)
    if (h[i]) atomicAdd(&counts[i], (unsigned long long)h[i]);
}

__global__ __launch_bounds__(256) void focal_fwd_bwd_kernel(const float* __restrict__ x, const long long* __restrict__ target, int B, int C,
                                                            long HW, float gamma, int class_weighting, float scale, long long ignore,
                                                            const unsigned long long* __restrict__ counts, float* __restrict__ wout,
                                                            float* __restrict__ dx, double* __restrict__ partial) {
  __shared__ float w[FL_MAXC];
  __shared__ double red[256];
  const long n = (long)B * HW;
  for (int c = threadIdx.x; c < C; c += 256) {
    float v = 1.0f;
    if (class_weighting && counts[1 + c] > 0) v = 1.0f / logf(1.1f + (float)counts[1 + c] / (float)n);
    w[c] = v;
    if (blockIdx.x == 0 && wout) wout[c] = v;
  }
  __syncthreads();
  const float inv_n = scale / (float)n;
  double acc = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / HW, p = i % HW;
    const float* xp = x + b * C * HW + p;
    float* dp = dx ? dx + b * C * HW + p : nullptr;
    const long long t = target[i];
    if (t == ignore || t < 0 || t >= C) {
      if (dp)
        for (int c = 0; c < C; ++c) dp[(long)c * HW] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, xp[(long)c * HW]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(xp[(long)c * HW] - mx);
    const float lse = mx + logf(se);
    const float ce = lse - xp[t * HW];
    const float pt = expf(-ce);
    const float om = 1.0f - pt;
    const float wt = w[(int)t];
    const float fl = powf(om, gamma) * wt * ce;
    acc += (double)fl;
    if (dp) {
      // d/dce [ (1 - e^-ce)^g * ce ] = (1-pt)^g + g * ce * pt * (1-pt)^(g-1)
      const float g = powf(om, gamma) + (om > 0.f ? gamma * ce * pt * powf(om, gamma - 1.0f) : 0.f);
      const float k = inv_n * wt * g;
      for (int c = 0; c < C; ++c) {
        const float sm = expf(xp[(long)c * HW] - lse);
        dp[(long)c * HW] = k * (sm - (c == (int)t ? 1.0f : 0.0f));
      }
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void focal_fold_kernel(const double* __restrict__ partial, int nblocks, long n, float scale, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nblocks; ++i) s += partial[i];
    loss[0] = (float)(s / (double)n * (double)scale);
  }
}

}  // namespace la

extern "C" int la_focal_loss(const float* logits, const long long* target, int B, int C, long HW, float gamma, int class_weighting, float scale,
                             long long ignore_index, float* loss, float* dlogits, float* class_weights, void* scratch, long scratch_bytes,
                             void* stream) {
  LA_CHECK_ARG(logits && target && loss && scratch, "la_focal_loss: null pointer");
  LA_CHECK_ARG(B > 0 && HW > 0 && C >= 2 && C <= la::FL_MAXC, "la_focal_loss: bad shape B=%d C=%d HW=%ld (C <= %d)", B, C, HW, la::FL_MAXC);
  const long n = (long)B * HW;
  int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  const long need = (long)(C + 2) * 8 + (long)blocks * 8;
  LA_CHECK_ARG(scratch_bytes >= need, "la_focal_loss: scratch needs %ld bytes", need);
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* counts = (unsigned long long*)scratch;
  double* partial = (double*)((char*)scratch + (long)(C + 2) * 8);
  if (hipMemsetAsync(counts, 0, (size_t)(C + 2) * 8, st) != hipSuccess) {
    la_set_error("la_focal_loss: memset failed");
    return -2;
  }
  hipLaunchKernelGGL(la::focal_hist_kernel, dim3(blocks), dim3(256), 0, st, target, n, C, ignore_index, counts);
  hipLaunchKernelGGL(la::focal_fwd_bwd_kernel, dim3(blocks), dim3(256), 0, st, logits, target, B, C, HW, gamma, class_weighting, scale,
                     ignore_index, counts, class_weights, dlogits, partial);
  hipLaunchKernelGGL(la::focal_fold_kernel, dim3(1), dim3(64), 0, st, partial, blocks, n, scale, loss);
  LA_CHECK_LAUNCH("la_focal_loss");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// AdamW step on a flat fp32 parameter buffer (torch.optim.AdamW, the reference's optimizer: experiment/utils.py:53-76,
// mae_noembs.yaml:32-33), same update order as torch's single-tensor path:
//   p *= 1 - lr * wd;  m += (g - m) * (1 - b1);  v = v * b2 + (1 - b2) * g * g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// g is multiplied by grad_scale first (1 / world size after the SUM all-reduce of the data-parallel ranks).
// ---------------------------------------------------------------------------------------------------------------------
namespace la {
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                    float bc2_sqrt, float grad_scale) {
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * grad_scale;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] * b2 + ((1.0f - b2) * gi) * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}
}  // namespace la

extern "C" int la_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, float grad_scale, void* stream) {
  LA_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "la_adamw_step: bad arguments");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(la::adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr, beta1,
                     beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
  LA_CHECK_LAUNCH("la_adamw_step");
  return 0;
}
