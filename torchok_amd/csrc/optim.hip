// Flat-arena optimizers: one launch updates a contiguous run of fp32 master parameters from the
// fp32 gradient arena and rewrites the bf16 shadow the MFMA kernels read (fp32 master + bf16
// shadow in ONE pass: 4 reads/writes of 4 B + one 2 B write per parameter).
// Arithmetic order follows torch.optim.SGD / Adam / AdamW (single-tensor reference path).
#include "tok_common.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ mbuf, bf16* __restrict__ shadow,
                                                  size_t n, float lr, float momentum, float dampening,
                                                  float wd, int nesterov, int first, int maximize) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    float w = p[i];
    float d = maximize ? -g[i] : g[i];
    if (wd != 0.f) d = d + wd * w;
    if (momentum != 0.f) {
      float b;
      if (first) b = d;
      else b = momentum * mbuf[i] + (1.f - dampening) * d;
      mbuf[i] = b;
      d = nesterov ? d + momentum * b : b;
    }
    w = w - lr * d;
    p[i] = w;
    if (shadow != nullptr) shadow[i] = f2bf(w);
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   bf16* __restrict__ shadow, size_t n, float lr, float beta1,
                                                   float beta2, float eps, float wd, int decoupled,
                                                   float step_size, float bc2_sqrt, int maximize) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    float w = p[i];
    float d = maximize ? -g[i] : g[i];
    if (wd != 0.f) {
      if (decoupled) w = w * (1.f - lr * wd);
      else d = d + wd * w;
    }
    float mi = m[i], vi = v[i];
    mi = mi + (d - mi) * (1.f - beta1);           // lerp, as torch
    vi = beta2 * vi + (1.f - beta2) * d * d;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    w = w - step_size * (mi / denom);
    p[i] = w;
    if (shadow != nullptr) shadow[i] = f2bf(w);
  }
}

// torch.optim.RMSprop (_single_tensor_rmsprop): g += wd * p; sq = alpha * sq + (1 - alpha) g^2; centered: ga = lerp(ga, g,
// 1 - alpha), avg = sqrt(sq - ga^2) + eps, else avg = sqrt(sq) + eps; momentum: buf = m * buf + g / avg, p -= lr * buf,
// else p -= lr * g / avg.  State arrays start at zero (the arena allocates them zeroed).
__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq,
                               float* __restrict__ buf, float* __restrict__ ga, size_t n, float lr, float alpha, float eps,
                               float wd, float momentum, int centered, int maximize) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float w = p[i];
    float d = maximize ? -g[i] : g[i];
    if (wd != 0.f) d = fmaf(wd, w, d);
    const float s = alpha * sq[i] + (1.f - alpha) * d * d;
    sq[i] = s;
    float avg;
    if (centered) {
      const float a = ga[i] + (d - ga[i]) * (1.f - alpha);
      ga[i] = a;
      avg = sqrtf(s - a * a) + eps;
    } else {
      avg = sqrtf(s) + eps;
    }
    if (momentum > 0.f) {
      const float b = momentum * buf[i] + d / avg;
      buf[i] = b;
      w -= lr * b;
    } else {
      w -= lr * (d / avg);
    }
    p[i] = w;
  }
}

__global__ void fill_kernel(float* dst, float v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = v;
}
__global__ void scale_kernel(float* dst, float f, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] *= f;
}

inline int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int tok_sgd_step(float* param, const float* grad, float* momentum_buf, void* shadow_bf16,
                            size_t count, float lr, float momentum, float dampening, float weight_decay,
                            int nesterov, int first_step, int maximize, void* stream) {
  TOK_CHECK_ARG(param && grad && count > 0, "tok_sgd_step: bad args");
  TOK_CHECK_ARG(momentum == 0.f || momentum_buf != nullptr, "tok_sgd_step: momentum buffer missing");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(count)), dim3(256), 0, tok_stream(stream), param, grad,
                     momentum_buf, (bf16*)shadow_bf16, count, lr, momentum, dampening, weight_decay,
                     nesterov, first_step, maximize);
  TOK_CHECK_LAUNCH("tok_sgd_step");
  return TOK_OK;
}

extern "C" int tok_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                             void* shadow_bf16, size_t count, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int decoupled, int64_t step, int maximize, void* stream) {
  TOK_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && count > 0 && step >= 1, "tok_adam_step: bad args");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(count)), dim3(256), 0, tok_stream(stream), param, grad,
                     exp_avg, exp_avg_sq, (bf16*)shadow_bf16, count, lr, beta1, beta2, eps, weight_decay,
                     decoupled, (float)((double)lr / bc1), (float)sqrt(bc2), maximize);
  TOK_CHECK_LAUNCH("tok_adam_step");
  return TOK_OK;
}

// ---- capturable Adam: the step count lives on the device (torch.optim.Adam(capturable=True)) ----------------------------------
// Same update as adam_kernel with the bias corrections derived in the kernel from *step (the number of steps ALREADY taken:
// this launch is step *step + 1), in double like the host path; tok_step_advance increments the counter afterwards.  Nothing
// in the launch arguments changes from step to step, so a hipGraph recording of the optimizer step replays correctly.
namespace {
__global__ __launch_bounds__(256) void adam_capturable_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                              float* __restrict__ m, float* __restrict__ v,
                                                              bf16* __restrict__ shadow, size_t n, float lr, float beta1,
                                                              float beta2, float eps, float wd, int decoupled,
                                                              const int64_t* __restrict__ step, int maximize) {
  const double t = (double)(*step + 1);
  const double bc1 = 1.0 - pow((double)beta1, t);
  const double bc2 = 1.0 - pow((double)beta2, t);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float w = p[i];
    float d = maximize ? -g[i] : g[i];
    if (wd != 0.f) {
      if (decoupled) w = w * (1.f - lr * wd);
      else d = d + wd * w;
    }
    float mi = m[i], vi = v[i];
    mi = mi + (d - mi) * (1.f - beta1);
    vi = beta2 * vi + (1.f - beta2) * d * d;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    w = w - step_size * (mi / denom);
    p[i] = w;
    if (shadow != nullptr) shadow[i] = f2bf(w);
  }
}
__global__ void step_advance_kernel(int64_t* step) { *step += 1; }
}  // namespace

extern "C" int tok_adam_step_capturable(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                                        size_t count, float lr, float beta1, float beta2, float eps, float weight_decay,
                                        int decoupled, const int64_t* step_dev, int maximize, void* stream) {
  TOK_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step_dev && count > 0, "tok_adam_step_capturable: bad args");
  hipLaunchKernelGGL(adam_capturable_kernel, dim3(grid_for(count)), dim3(256), 0, tok_stream(stream), param, grad, exp_avg,
                     exp_avg_sq, (bf16*)shadow_bf16, count, lr, beta1, beta2, eps, weight_decay, decoupled, step_dev, maximize);
  TOK_CHECK_LAUNCH("tok_adam_step_capturable");
  return TOK_OK;
}

extern "C" int tok_step_advance(int64_t* step_dev, void* stream) {
  TOK_CHECK_ARG(step_dev != nullptr, "tok_step_advance: null pointer");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, tok_stream(stream), step_dev);
  TOK_CHECK_LAUNCH("tok_step_advance");
  return TOK_OK;
}

extern "C" int tok_rmsprop_step(float* param, const float* grad, float* square_avg, float* momentum_buf, float* grad_avg,
                                size_t count, float lr, float alpha, float eps, float weight_decay, float momentum,
                                int centered, int maximize, void* stream) {
  TOK_CHECK_ARG(param && grad && square_avg && count > 0, "tok_rmsprop_step: bad args");
  TOK_CHECK_ARG((momentum <= 0.f || momentum_buf) && (!centered || grad_avg), "tok_rmsprop_step: state buffer missing");
  hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(count)), dim3(256), 0, tok_stream(stream), param, grad, square_avg,
                     momentum_buf, grad_avg, count, lr, alpha, eps, weight_decay, momentum, centered, maximize);
  TOK_CHECK_LAUNCH("tok_rmsprop_step");
  return TOK_OK;
}

extern "C" int tok_fill_f32(float* dst, float value, size_t count, void* stream) {
  TOK_CHECK_ARG(dst && count > 0, "tok_fill_f32: bad args");
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(count)), dim3(256), 0, tok_stream(stream), dst, value, count);
  TOK_CHECK_LAUNCH("tok_fill_f32");
  return TOK_OK;
}

extern "C" int tok_scale_f32(float* dst, float factor, size_t count, void* stream) {
  TOK_CHECK_ARG(dst && count > 0, "tok_scale_f32: bad args");
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(count)), dim3(256), 0, tok_stream(stream), dst, factor, count);
  TOK_CHECK_LAUNCH("tok_scale_f32");
  return TOK_OK;
}
