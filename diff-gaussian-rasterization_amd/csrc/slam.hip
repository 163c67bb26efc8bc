// slam.hip -- the small per-iteration pieces of a tracking step around the rasterizer, each as
// ONE launch instead of a dozen elementwise torch kernels: a 640x480 tracking iteration is launch-bound (the rasterizer's
// kernels are 0.17 ms of a 0.43 ms hipGraph replay when pose, loss and their backward run as torch ops).
//   pose_forward / pose_backward : (quaternion, translation) <-> the rasterizer's camera tensors and dL/dviewmatrix
//   l1_loss_forward / _backward  : w_c mean|C - C_obs| + w_d mean|D - D_obs| and its two gradient images
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"

namespace dgr {
namespace {

// R = (r^2 - |v|^2) I + 2 v v^T + 2 r [v]x of the NORMALISED quaternion (r, x, y, z) -- dgr_amd.slam.quat_to_rotmat
__device__ void rotation(const float* q, float (&R)[3][3], float& inv_norm, float (&qh)[4]) {
    inv_norm = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) qh[i] = q[i] * inv_norm;
    const float r = qh[0], x = qh[1], y = qh[2], z = qh[3];
    const float d = r * r - (x * x + y * y + z * z);
    R[0][0] = d + 2.f * x * x;      R[0][1] = 2.f * x * y - 2.f * r * z;  R[0][2] = 2.f * x * z + 2.f * r * y;
    R[1][0] = 2.f * x * y + 2.f * r * z;  R[1][1] = d + 2.f * y * y;      R[1][2] = 2.f * y * z - 2.f * r * x;
    R[2][0] = 2.f * x * z - 2.f * r * y;  R[2][1] = 2.f * y * z + 2.f * r * x;  R[2][2] = d + 2.f * z * z;
}

// viewmatrix = W2C^T, projmatrix = W2C^T Proj^T, campos = -R^T t  (cuda_rasterizer/auxiliary.h:58-77 reads all three
// column-major, i.e. as the transposes stored row-major)
__global__ void pose_forward_kernel(const float* q, const float* t, const float* perspec, float* view, float* proj, float* campos) {
    if (threadIdx.x != 0) return;
    float R[3][3], inv_norm, qh[4];
    rotation(q, R, inv_norm, qh);
    float V[4][4];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) V[i][j] = R[j][i];
        V[i][3] = 0.f;
        V[3][i] = t[i];
    }
    V[3][3] = 1.f;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            view[4 * i + j] = V[i][j];
            float s = 0.f;
            for (int k = 0; k < 4; k++) s += V[i][k] * perspec[4 * k + j];
            proj[4 * i + j] = s;
        }
    for (int i = 0; i < 3; i++) campos[i] = -(R[0][i] * t[0] + R[1][i] * t[1] + R[2][i] * t[2]);
}

// dL/dviewmatrix -> dL/dq, dL/dt.  projmatrix and campos enter the rasterizer as constants (the reference's backward adds
// their dependence on the pose inside its kernels: L/cuda_rasterizer/backward.cu:633-651, 683-751).
__global__ void pose_backward_kernel(const float* q, const float* dview, float* dq, float* dt) {
    if (threadIdx.x != 0) return;
    float R[3][3], inv_norm, qh[4];
    rotation(q, R, inv_norm, qh);
    float G[3][3];  // dL/dR[a][b] = dL/dviewmatrix[b][a]
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) G[a][b] = dview[4 * b + a];
    for (int a = 0; a < 3; a++) dt[a] = dview[12 + a];
    const float r = qh[0], v[3] = {qh[1], qh[2], qh[3]};
    const float tr = G[0][0] + G[1][1] + G[2][2];
    float g[4];
    g[0] = 2.f * r * tr + 2.f * (-v[2] * G[0][1] + v[1] * G[0][2] + v[2] * G[1][0] - v[0] * G[1][2] - v[1] * G[2][0] + v[0] * G[2][1]);
    const float skew[3] = {G[2][1] - G[1][2], G[0][2] - G[2][0], G[1][0] - G[0][1]};
    for (int k = 0; k < 3; k++) {
        float gv = 0.f;
        for (int j = 0; j < 3; j++) gv += (G[k][j] + G[j][k]) * v[j];
        g[1 + k] = -2.f * v[k] * tr + 2.f * gv + 2.f * r * skew[k];
    }
    const float along = qh[0] * g[0] + qh[1] * g[1] + qh[2] * g[2] + qh[3] * g[3];
    for (int i = 0; i < 4; i++) dq[i] = (g[i] - qh[i] * along) * inv_norm;  // through q / |q|
}

constexpr int LOSS_BLOCKS = 128;

// partial[b] = block b's share of  w_c / n_c * sum|C - C_obs| + w_d / n_d * sum|D - D_obs|
__global__ void __launch_bounds__(256) l1_partial_kernel(long n_c, const float* c, const float* c_obs, long n_d, const float* d,
                                                         const float* d_obs, float k_c, float k_d, float* partial) {
    float s = 0.f;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_c; i += stride) s += k_c * fabsf(c[i] - c_obs[i]);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_d; i += stride) s += k_d * fabsf(d[i] - d_obs[i]);
    __shared__ float red[4];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(64) l1_final_kernel(const float* partial, int n, float* loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) *loss = s;
}
__device__ __forceinline__ float sign0(float x) { return (x > 0.f) ? 1.f : (x < 0.f) ? -1.f : 0.f; }  // torch.sign
__global__ void __launch_bounds__(256) l1_backward_kernel(long n_c, const float* c, const float* c_obs, long n_d, const float* d,
                                                          const float* d_obs, float k_c, float k_d, const float* upstream,
                                                          float* dc, float* dd) {
    const float up = upstream ? *upstream : 1.f;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_c; i += stride) dc[i] = (up * k_c) * sign0(c[i] - c_obs[i]);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_d; i += stride) dd[i] = (up * k_d) * sign0(d[i] - d_obs[i]);
}

}  // namespace

hipError_t launch_pose_forward(const float* q, const float* t, const float* perspec, float* view, float* proj, float* campos,
                               hipStream_t stream) {
    launch(pose_forward_kernel, dim3(1), dim3(64), stream, q, t, perspec, view, proj, campos);
    return hipGetLastError();
}
hipError_t launch_pose_backward(const float* q, const float* dview, float* dq, float* dt, hipStream_t stream) {
    launch(pose_backward_kernel, dim3(1), dim3(64), stream, q, dview, dq, dt);
    return hipGetLastError();
}
int l1_loss_partials() { return LOSS_BLOCKS; }
hipError_t launch_l1_loss_forward(long n_c, const float* c, const float* c_obs, long n_d, const float* d, const float* d_obs,
                                  float w_c, float w_d, float* partial, float* loss, hipStream_t stream) {
    const float k_c = n_c > 0 ? w_c / (float)n_c : 0.f, k_d = n_d > 0 ? w_d / (float)n_d : 0.f;
    launch(l1_partial_kernel, dim3(LOSS_BLOCKS), dim3(256), stream, n_c, c, c_obs, n_d, d, d_obs, k_c, k_d, partial);
    launch(l1_final_kernel, dim3(1), dim3(64), stream, (const float*)partial, LOSS_BLOCKS, loss);
    return hipGetLastError();
}
hipError_t launch_l1_loss_backward(long n_c, const float* c, const float* c_obs, long n_d, const float* d, const float* d_obs,
                                   float w_c, float w_d, const float* upstream, float* dc, float* dd, hipStream_t stream) {
    const float k_c = n_c > 0 ? w_c / (float)n_c : 0.f, k_d = n_d > 0 ? w_d / (float)n_d : 0.f;
    const long n = std::max(n_c, n_d);
    if (n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<long>((n + 255) / 256, 2048);
    launch(l1_backward_kernel, dim3(blocks), dim3(256), stream, n_c, c, c_obs, n_d, d, d_obs, k_c, k_d, upstream, dc, dd);
    return hipGetLastError();
}

}  // namespace dgr
