/* Fused convolution epilogues for the PPO trainer that drives the env path (the reference's
 * v8_robust_ppo.py:69-95 network, `Conv2d -> ReLU -> Conv2d -> ReLU -> AdaptiveAvgPool2d((1,1))`, and the update loop
 * :295-345 that back-propagates through it).  NOT part of the env drop-in boundary (include/mgrid.h): the reference has
 * no native interface here — it calls torch.nn modules — so these entry points replace the bias add / ReLU / pooling
 * modules of that network and their autograd, and multigrid_amd/fused_ops.py binds them as torch.autograd.Functions.
 *
 * All tensors are fp32 device memory in channels-last storage, [N][H*W][C] with C innermost (what MIOpen's nhwc
 * convolutions produce), C a power-of-two multiple of 4 up to 1024.  Every call enqueues one kernel on `stream` (a
 * hipStream_t; NULL = the default stream) of the CURRENT device and returns without synchronising; nothing is
 * allocated.  Return value: MGNN_OK or an MGNN_ERR_*.
 */
#ifndef MGRID_NN_H
#define MGRID_NN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGNN_OK 0
#define MGNN_ERR_SHAPE 1   /* C not a power-of-two multiple of 4, negative sizes */
#define MGNN_ERR_LAUNCH 2  /* the kernel launch failed (hipGetLastError) */

/* rows of a bias-gradient partial buffer: the backward kernels run MGNN_PARTIAL_ROWS workgroups and workgroup b writes
 * its share of the bias gradient to partial[b][0..C) (zeros if it had no work); the caller sums the rows.  Deterministic. */
#define MGNN_PARTIAL_ROWS 2048
int mgnn_partial_rows(void);

/* nn.Conv2d's bias add + nn.ReLU (v8_robust_ppo.py:70-71), in place: y[r][c] = max(y[r][c] + bias[c], 0), r < n_rows = N*H*W */
int mgnn_bias_relu(float* y, const float* bias, int64_t n_rows, int C, void* stream);

/* its backward: grad_in = grad_out * (out > 0)  (ReLU) and partial[b][c] = workgroup b's sum of grad_in over rows (the
 * convolution's bias gradient).  out = what mgnn_bias_relu left in y. */
int mgnn_bias_relu_backward(const float* out, const float* grad_out, float* grad_in, float* partial, int64_t n_rows, int C,
                            void* stream);

/* bias add + ReLU + AdaptiveAvgPool2d((1,1)) + Flatten (v8_robust_ppo.py:72-75): out[n][c] = mean_hw max(y[n][hw][c] + bias[c], 0).
 * y (the convolution's output without bias) is only read. */
int mgnn_bias_relu_mean(const float* y, const float* bias, float* out, int64_t N, int HW, int C, void* stream);

/* its backward: grad_y[n][hw][c] = (y[n][hw][c] + bias[c] > 0) ? grad_out[n][c] / HW : 0, partial as above */
int mgnn_bias_relu_mean_backward(const float* y, const float* bias, const float* grad_out, float* grad_y, float* partial,
                                 int64_t N, int HW, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif
