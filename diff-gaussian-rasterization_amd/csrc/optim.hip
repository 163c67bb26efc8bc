// optim.hip -- fused sparse Adam step for the Gaussian parameters.
//
// A mapping iteration ends with an optimiser step over the per-Gaussian tensors whose gradients the backward just
// wrote (for all views: after the all-reduce).  Only Gaussians some view saw have a gradient; as in 3DGS's sparse Adam
// the rows of the others are left alone: parameter AND both moments untouched.  One launch per tensor, element-
// parallel (row = element / k), so consecutive lanes touch consecutive addresses of all four streams.
// Traffic per updated element: param, grad, exp_avg, exp_avg_sq in; param, exp_avg, exp_avg_sq out = 28 bytes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "kernels.h"

namespace dgr {
namespace {

__global__ void __launch_bounds__(256) sparse_adam_kernel(size_t n, int k, float* __restrict__ param,
                                                          const float* __restrict__ grad, float* __restrict__ exp_avg,
                                                          float* __restrict__ exp_avg_sq, const int* __restrict__ visible,
                                                          float step_size, float beta1, float beta2, float eps,
                                                          float inv_sqrt_bias2, float lr, const int* __restrict__ step_dev) {
    if (step_dev) {  // capturable form: the step count lives on the device, the bias corrections are formed here
        const float st = (float)*step_dev;
        step_size = lr / (1.0f - powf(beta1, st));
        inv_sqrt_bias2 = 1.0f / sqrtf(1.0f - powf(beta2, st));
    }
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
        if (visible && visible[e / (size_t)k] <= 0) continue;
        const float g = grad[e];
        const float m = beta1 * exp_avg[e] + (1.0f - beta1) * g;
        const float v = beta2 * exp_avg_sq[e] + (1.0f - beta2) * g * g;
        exp_avg[e] = m;
        exp_avg_sq[e] = v;
        // torch.optim.Adam: p -= lr / bias1 * m / (sqrt(v) / sqrt(bias2) + eps)
        param[e] -= step_size * m / (sqrtf(v) * inv_sqrt_bias2 + eps);
    }
}

// 3DGS's densification bookkeeping after a view's backward (GaussianModel.add_densification_stats and the
// max_radii2D update of the training loop), for the rows the view saw: one pass instead of five indexed torch ops.
__global__ void __launch_bounds__(256) densification_stats_kernel(int rows, const float* __restrict__ dmeans2D,
                                                                  const int* __restrict__ radii,
                                                                  float* __restrict__ grad_accum, float* __restrict__ denom,
                                                                  float* __restrict__ max_radii2D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    const int r = radii[i];
    if (r <= 0) return;
    if (grad_accum) {
        const float gx = dmeans2D[3 * (size_t)i], gy = dmeans2D[3 * (size_t)i + 1];
        grad_accum[i] += sqrtf(gx * gx + gy * gy);
    }
    if (denom) denom[i] += 1.0f;
    if (max_radii2D) max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
}

}  // namespace

hipError_t launch_densification_stats(int rows, const float* dmeans2D, const int* radii, float* grad_accum, float* denom,
                                      float* max_radii2D, hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    launch(densification_stats_kernel, dim3((rows + 255) / 256), dim3(256), stream, rows, dmeans2D, radii, grad_accum, denom,
           max_radii2D);
    return hipGetLastError();
}

hipError_t launch_sparse_adam(size_t rows, int k, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                              const int* visible, float lr, float beta1, float beta2, float eps, int step,
                              const int* step_dev, hipStream_t stream) {
    const size_t n = rows * (size_t)k;
    if (n == 0) return hipSuccess;
    if (step < 1) step = 1;  // (unused with step_dev)
    const double bias1 = 1.0 - pow((double)beta1, (double)step), bias2 = 1.0 - pow((double)beta2, (double)step);
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32);
    launch(sparse_adam_kernel, dim3(blocks), dim3(256), stream, n, k, param, grad, exp_avg, exp_avg_sq, visible,
           (float)((double)lr / bias1), beta1, beta2, eps, (float)(1.0 / sqrt(bias2)), lr, step_dev);
    return hipGetLastError();
}

}  // namespace dgr
