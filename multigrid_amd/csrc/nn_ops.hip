// Fused epilogues of the PPO trainer's two convolutions (f1, the caller of the env path: v8_robust_ppo.py:69-95,
// `Conv2d -> ReLU -> Conv2d -> ReLU -> AdaptiveAvgPool2d((1,1))`).  The convolutions themselves stay MIOpen's; what is
// here are the passes around them that are pure HBM traffic over the 1.3 GB / 2.6 GB activation tensors of an update
// batch: bias add, ReLU, the global average and their backward passes (mask, expand, bias-gradient reduction), each
// fused into ONE read (+ at most one write) of the activation.  fp32, channels-last storage ([N][H*W][C], C innermost:
// what MIOpen's nhwc kernels produce for the trainer's inputs), C a multiple of 4 so a lane moves 16 bytes.
// Bias gradients are reduced deterministically: every workgroup writes its partial sums to partial[block][C] and the
// caller adds the rows up (same result every run, unlike float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgrid_nn.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Block-wide sum of one float4 per thread over the threads that share a channel group (tid % G4), G4 = C/4 | kBlock.
__device__ __forceinline__ void reduce_groups(float4 acc, int G4, float* partial_row) {
  __shared__ float4 red[kBlock];
  const int tid = threadIdx.x;
  red[tid] = acc;
  __syncthreads();
  for (int s = kBlock / 2; s >= G4; s >>= 1) {       // kBlock and G4 are powers of two, tid % G4 is preserved by +s
    if (tid < s) red[tid] = f4_add(red[tid], red[tid + s]);
    __syncthreads();
  }
  if (tid < G4) reinterpret_cast<float4*>(partial_row)[tid] = red[tid];
}

// y[row][c] = max(y[row][c] + bias[c], 0), in place.  One float4 per thread and pass.
__global__ __launch_bounds__(kBlock) void bias_relu_kernel(float4* y, const float4* bias, int64_t n_vec, int G4) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  // kBlock % G4 == 0 and the grid stride is a multiple of kBlock: a thread keeps its channel group
  const float4 b = bias[threadIdx.x % G4];
  for (; i < n_vec; i += stride) {
    float4 v = y[i];
    v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f); v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
    y[i] = v;
  }
}

// grad_in = grad_out * (out > 0); partial[block][c] = sum over this block's rows of grad_in (threshold_backward + the
// convolution's bias gradient in one pass)
__global__ __launch_bounds__(kBlock) void bias_relu_bwd_kernel(const float4* out, const float4* grad_out, float4* grad_in,
                                                               float* partial, int64_t n_vec, int G4) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (; i < n_vec; i += stride) {
    const float4 o = out[i], g = grad_out[i];
    float4 r;
    r.x = o.x > 0.f ? g.x : 0.f; r.y = o.y > 0.f ? g.y : 0.f; r.z = o.z > 0.f ? g.z : 0.f; r.w = o.w > 0.f ? g.w : 0.f;
    grad_in[i] = r;
    acc = f4_add(acc, r);
  }
  reduce_groups(acc, G4, partial + (int64_t)blockIdx.x * G4 * 4);
}

// out[n][c] = mean over hw of max(y[n][hw][c] + bias[c], 0); y is only read.  One (n, channel group) per thread.
__global__ __launch_bounds__(kBlock) void bias_relu_mean_kernel(const float4* y, const float4* bias, float4* out, int64_t n_items,
                                                                int HW, int G4, float inv_hw) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const float4 b = bias[threadIdx.x % G4];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_items; i += stride) {
    const int64_t n = i / G4;
    const float4* row = y + n * (int64_t)HW * G4 + (i - n * G4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = 0; h < HW; ++h) {
      const float4 v = row[(int64_t)h * G4];
      acc.x += fmaxf(v.x + b.x, 0.f); acc.y += fmaxf(v.y + b.y, 0.f); acc.z += fmaxf(v.z + b.z, 0.f); acc.w += fmaxf(v.w + b.w, 0.f);
    }
    out[i] = make_float4(acc.x * inv_hw, acc.y * inv_hw, acc.z * inv_hw, acc.w * inv_hw);
  }
}

// grad_y[n][hw][c] = (y + bias > 0) ? grad_out[n][c] / HW : 0; partial[block][c] = this block's share of the bias gradient
__global__ __launch_bounds__(kBlock) void bias_relu_mean_bwd_kernel(const float4* y, const float4* bias, const float4* grad_out,
                                                                    float4* grad_y, float* partial, int64_t n_items, int HW,
                                                                    int G4, float inv_hw) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const float4 b = bias[threadIdx.x % G4];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_items; i += stride) {
    const int64_t n = i / G4;
    const int64_t off = n * (int64_t)HW * G4 + (i - n * G4);
    float4 g = grad_out[i];
    g.x *= inv_hw; g.y *= inv_hw; g.z *= inv_hw; g.w *= inv_hw;
    for (int h = 0; h < HW; ++h) {
      const float4 v = y[off + (int64_t)h * G4];
      float4 r;
      r.x = v.x + b.x > 0.f ? g.x : 0.f; r.y = v.y + b.y > 0.f ? g.y : 0.f;
      r.z = v.z + b.z > 0.f ? g.z : 0.f; r.w = v.w + b.w > 0.f ? g.w : 0.f;
      grad_y[off + (int64_t)h * G4] = r;
      acc = f4_add(acc, r);
    }
  }
  reduce_groups(acc, G4, partial + (int64_t)blockIdx.x * G4 * 4);
}

int check_shape(int64_t rows, int C) {
  if (rows < 0 || C < 4 || (C & 3) != 0) return MGNN_ERR_SHAPE;
  const int G4 = C / 4;
  if (G4 > kBlock || (G4 & (G4 - 1)) != 0) return MGNN_ERR_SHAPE;      // a power of two that divides the block
  return MGNN_OK;
}

int grid_for(int64_t work_items, int blocks_cap) {
  int64_t b = (work_items + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > blocks_cap) b = blocks_cap;
  return (int)b;
}

int launched() { return hipGetLastError() == hipSuccess ? MGNN_OK : MGNN_ERR_LAUNCH; }

}  // namespace

extern "C" {

int mgnn_partial_rows(void) { return MGNN_PARTIAL_ROWS; }

int mgnn_bias_relu(float* y, const float* bias, int64_t n_rows, int C, void* stream) {
  if (int rc = check_shape(n_rows, C)) return rc;
  if (n_rows == 0) return MGNN_OK;
  const int64_t n_vec = n_rows * (C / 4);
  hipLaunchKernelGGL(bias_relu_kernel, dim3(grid_for(n_vec, 8192)), dim3(kBlock), 0, (hipStream_t)stream,
                     reinterpret_cast<float4*>(y), reinterpret_cast<const float4*>(bias), n_vec, C / 4);
  return launched();
}

int mgnn_bias_relu_backward(const float* out, const float* grad_out, float* grad_in, float* partial, int64_t n_rows, int C,
                            void* stream) {
  if (int rc = check_shape(n_rows, C)) return rc;
  const int64_t n_vec = n_rows * (C / 4);
  // every row of partial is written (blocks without work write zeros): always the full MGNN_PARTIAL_ROWS grid
  hipLaunchKernelGGL(bias_relu_bwd_kernel, dim3(MGNN_PARTIAL_ROWS), dim3(kBlock), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(out), reinterpret_cast<const float4*>(grad_out),
                     reinterpret_cast<float4*>(grad_in), partial, n_vec, C / 4);
  return launched();
}

int mgnn_bias_relu_mean(const float* y, const float* bias, float* out, int64_t N, int HW, int C, void* stream) {
  if (int rc = check_shape(N, C)) return rc;
  if (HW < 1) return MGNN_ERR_SHAPE;
  if (N == 0) return MGNN_OK;
  const int64_t n_items = N * (C / 4);
  hipLaunchKernelGGL(bias_relu_mean_kernel, dim3(grid_for(n_items, 1 << 20)), dim3(kBlock), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(y), reinterpret_cast<const float4*>(bias), reinterpret_cast<float4*>(out),
                     n_items, HW, C / 4, 1.0f / (float)HW);
  return launched();
}

int mgnn_bias_relu_mean_backward(const float* y, const float* bias, const float* grad_out, float* grad_y, float* partial,
                                 int64_t N, int HW, int C, void* stream) {
  if (int rc = check_shape(N, C)) return rc;
  if (HW < 1) return MGNN_ERR_SHAPE;
  const int64_t n_items = N * (C / 4);
  hipLaunchKernelGGL(bias_relu_mean_bwd_kernel, dim3(MGNN_PARTIAL_ROWS), dim3(kBlock), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(y), reinterpret_cast<const float4*>(bias),
                     reinterpret_cast<const float4*>(grad_out), reinterpret_cast<float4*>(grad_y), partial, n_items, HW, C / 4,
                     1.0f / (float)HW);
  return launched();
}

}  // extern "C"
