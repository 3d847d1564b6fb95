// Host input pipeline of the reference moved onto the GPU: Dataset_2.py label2vec and the
// DataAugs.py augmentations, AS EXECUTED by the reference (oracle/input_oracle.py spells out the quirks), fused into one
// elementwise pass per batch.  The reference runs them as pure-Python O(H*W) loops per sample (tens of milliseconds per
// image), which would starve a 5 ms training step; here a 16-image batch is one ~10 us launch.  The random draws stay on
// the host in the reference's order (Python `random`), so a seeded run picks the same boxes and shifts.
#include "common.h"

// label -> soft class maps (Dataset_2.py:6-20)
__device__ __forceinline__ void label2vec_dev(float label, int C, float* out) {
  if (C == 3) {
    float c2 = label >= 1.05f ? label - 1.f : 0.f;
    c2 = c2 > 1.f ? 1.f : c2;
    out[0] = label <= 0.95f ? 1.f : 0.f;
    out[1] = label > 0.95f ? 1.f - c2 : 0.f;
    out[2] = c2;
  } else {
    out[0] = 1.f - label;
    out[1] = label;
  }
}

__global__ __launch_bounds__(256) void label2vec_kernel(const float* label, int64_t M, int C, float* out) {
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    float v[3];
    label2vec_dev(label[m], C, v);
    for (int c = 0; c < C; ++c) out[m * C + c] = v[c];
  }
}

extern "C" int usseg_label2vec(const float* label, int64_t M, int32_t num_classes, float* out, usseg_stream_t stream) {
  USSEG_CHECK_ARG(label && out && (num_classes == 2 || num_classes == 3), "label2vec: num_classes must be 2 or 3");
  if (M <= 0) return USSEG_OK;
  int64_t g = cdiv64(M, 256 * 4);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(label2vec_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, label, M, num_classes, out);
  return usseg_check_launch("label2vec");
}

__device__ __forceinline__ uint32_t aug_hash(uint64_t v) {
  v ^= v >> 33; v *= 0xff51afd7ed558ccdULL; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ULL; v ^= v >> 33;
  return (uint32_t)v;
}

// dataAug (DataAugs.py:82-102) for a whole batch + label2vec + the cast to the model's bf16 NHWC input.
template <typename T>
__global__ __launch_bounds__(256) void augment_kernel(const UssegAugDesc d, const UssegAugSample* samples, const T* x, const float* y,
                                                      const float* noise, bf16_t* x_out, float* x_out_f32, float* y_out, float* y_vec) {
  const int64_t total = (int64_t)d.B * d.H * d.W;
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < total; m += (int64_t)gridDim.x * 256) {
    const int b = (int)(m / ((int64_t)d.H * d.W));
    const int rem = (int)(m - (int64_t)b * d.H * d.W);
    const int i = rem / d.W, j = rem - i * d.W;
    const UssegAugSample s = samples[b];
    int si = i, sj = j;
    bool valid = true;
    if (s.do_shift) {                                     // DataAugs.py:6-24: the loops stop at si-1, the rest stays zero
      valid = i < d.H - 1 && j < d.W - 1;
      si = s.shift_dir ? i + s.shift_r : i - s.shift_r;
      sj = s.shift_dir ? j + s.shift_c : j - s.shift_c;
      valid = valid && (unsigned)si < (unsigned)d.H && (unsigned)sj < (unsigned)d.W;
    }
    float lab = 0.f;
    bool zero_img = !valid;
    int64_t src = 0;
    if (valid) {
      src = ((int64_t)b * d.H + si) * d.W + sj;
      lab = y[src];
      if (s.do_reduc && lab == 0.f) zero_img = true;      // imageReduc as executed (DataAugs.py:76-78)
      for (int k = 0; k < s.nclip; ++k) {                  // clip (DataAugs.py:27-38), in source coordinates
        const int r = s.clip[k][0], c = s.clip[k][1], ra = s.clip[k][2], ca = s.clip[k][3];
        if (r + ra > si && si > r - ra && c + ca > sj && sj > c - ca && si < d.H - 1 && sj < d.W - 1) {
          zero_img = true;
          lab = 0.f;
        }
      }
    }
    for (int c = 0; c < d.Cphys; ++c) {
      double v = 0.0;
      if (c < d.C) {
        if (!zero_img) v = (double)x[src * d.C + c];
        if (s.do_noise) {                                  // noisy (DataAugs.py:41-51): unit Gaussian / 5000 on every pixel
          double g;
          if (noise) g = (double)noise[m * d.C + c];
          else {
            uint64_t ctr = s.seed * 0x9e3779b97f4a7c15ULL + (uint64_t)(m * d.C + c) * 2;
            float u1 = (aug_hash(ctr) + 1.0f) * (1.0f / 4294967296.0f), u2 = aug_hash(ctr + 1) * (1.0f / 4294967296.0f);
            g = (double)(sqrtf(-2.f * __logf(u1)) * __cosf(6.28318530718f * u2));
          }
          v += g / 5000.0;
        }
      }
      if (x_out) {   // double -> float -> bf16 in two separately rounded steps, like the reference's float32 model input followed by
                     // the bf16 cast (integer RNE on the float bits: the compiler must not fuse the two conversions)
        const uint32_t u = __float_as_uint((float)v);
        x_out[m * d.Cphys + c] = (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
      }
      if (x_out_f32 && c < d.C) x_out_f32[m * d.C + c] = (float)v;
    }
    if (y_out) y_out[m] = lab;
    if (y_vec) {
      float v[3];
      label2vec_dev(lab, d.num_classes, v);
      for (int c = 0; c < d.num_classes; ++c) y_vec[m * d.num_classes + c] = v[c];
    }
  }
}

extern "C" int usseg_augment(const UssegAugDesc* d, const UssegAugSample* samples_dev, const void* x, int32_t x_is_f64, const float* y,
                             const float* noise, void* x_out_bf16, float* x_out_f32, float* y_out, float* y_vec, usseg_stream_t stream) {
  USSEG_CHECK_ARG(d && samples_dev && x && y && (x_out_bf16 || x_out_f32), "augment: null pointer");
  USSEG_CHECK_ARG(d->B > 0 && d->H > 1 && d->W > 1 && d->C > 0 && d->Cphys >= d->C && d->Cphys % 8 == 0, "augment: bad geometry");
  USSEG_CHECK_ARG(!y_vec || d->num_classes == 2 || d->num_classes == 3, "augment: num_classes must be 2 or 3");
  int64_t total = (int64_t)d->B * d->H * d->W;
  int64_t g = cdiv64(total, 256 * 2);
  if (g > 4096) g = 4096;
  if (x_is_f64)
    hipLaunchKernelGGL(augment_kernel<double>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *d, samples_dev, (const double*)x, y, noise,
                       (bf16_t*)x_out_bf16, x_out_f32, y_out, y_vec);
  else
    hipLaunchKernelGGL(augment_kernel<float>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *d, samples_dev, (const float*)x, y, noise,
                       (bf16_t*)x_out_bf16, x_out_f32, y_out, y_vec);
  return usseg_check_launch("augment");
}
