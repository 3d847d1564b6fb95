// Split-attention (global average pool, tiny per-(image,path) MLP, channel re-weighting), head softmax + loss,
// and the optimiser (global-norm clip + Adam) for gfx950.  The big tensors are streamed 16 bytes per lane; the
// MLP is a few hundred FLOPs per workgroup and runs in fp32 out of LDS.
#include "common.h"

static inline int lanes_per_pixel(int chunks) {
  int l = 1;
  while (l < chunks) l <<= 1;
  return l;
}

// g[b][c] += sum_hw y[b,hw,c]   (and, with dout != NULL, ds[b][cy] += mult * sum_hw y[b,hw,cy] * dout[b,hw,co(cy)])
__global__ __launch_bounds__(256) void sa_reduce_kernel(const bf16_t* y, const bf16_t* dout, int HW, int Cy, int ldy, int lddo, int R, int Cg,
                                                         int LPP, float* ws) {
  __shared__ float s_red[4 * 512];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ppw = 64 / LPP, chunk = lane & (LPP - 1), slot = lane / LPP;
  const bool chunk_ok = chunk * 8 < Cy;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  // output-channel index of each of this lane's y channels (only needed for R > 1)
  int co[8];
  float cok[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int cy = chunk * 8 + j;
    cok[j] = cy < Cy ? 1.f : 0.f;
    cy = min(cy, Cy - 1);   // clamped: the loads below are unconditional (a load under a per-element condition serialises)
    int pr = cy / Cg;  // = p*R + r
    co[j] = (pr / R) * Cg + (cy - pr * Cg);
  }
  const int64_t ppb = 4 * ppw;
  const bf16_t* yb = y + (int64_t)b * HW * ldy;
  const bf16_t* db = dout ? dout + (int64_t)b * HW * lddo : nullptr;
  // four pixels per lane and trip with unconditional loads at clamped pixels (masked afterwards): the loop is a chain of HBM round trips, one
  // load pair per trip made it 8 dependent latencies per workgroup at stage 1 (12-15 us for 34 MB); fixed summation order
  const bool fastform = db == nullptr || R == 1 || (Cg & 7) == 0;
  const int64_t stride = (int64_t)gridDim.x * ppb;
  if (fastform) {
    const int dcol = (db && R != 1) ? co[0] : chunk * 8;
    for (int64_t base = (int64_t)blockIdx.x * ppb; base < HW; base += 4 * stride) {
      uint4 yv[4], dv4[4];
      float okf[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t m = base + u * stride + wv * ppw + slot;
        const bool ok = chunk_ok && m < HW;
        const int64_t mc = ok ? m : 0;
        okf[u] = ok ? 1.f : 0.f;
        yv[u] = *reinterpret_cast<const uint4*>(yb + mc * ldy + (chunk_ok ? chunk * 8 : 0));
        if (db) dv4[u] = *reinterpret_cast<const uint4*>(db + mc * lddo + (chunk_ok ? dcol : 0));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8];
        unpack8(yv[u], v);
        if (db) {
          float d[8];
          unpack8(dv4[u], d);
#pragma unroll
          for (int j = 0; j < 8; ++j) s[j] += okf[u] * cok[j] * v[j] * d[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) s[j] += okf[u] * v[j];
        }
      }
    }
  } else {
    for (int64_t base = (int64_t)blockIdx.x * ppb; base < HW; base += stride) {
      const int64_t m = base + wv * ppw + slot;
      if (chunk_ok && m < HW) {
        float v[8], dv[8];
        unpack8(*reinterpret_cast<const uint4*>(yb + m * ldy + chunk * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[j] = bf2f(db[m * lddo + co[j]]);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += cok[j] * v[j] * dv[j];
      }
    }
  }
  // one partial row per workgroup (no atomics); usseg_launch_reduce_finish adds the rows of image b
  const int Cp = (Cy + 7) & ~7;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = s[j];
    for (int msk = LPP; msk < 64; msk <<= 1) v += __shfl_xor(v, msk, 64);
    s[j] = v;
  }
  if (lane < LPP && chunk_ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s_red[wv * 512 + chunk * 8 + j] = s[j];
  }
  __syncthreads();
  float* row = ws + ((int64_t)b * gridDim.x + blockIdx.x) * Cp;
  for (int c = threadIdx.x; c < Cp; c += 256) row[c] = s_red[c] + s_red[512 + c] + s_red[1024 + c] + s_red[1536 + c];
}

static int sa_check(const UssegSplitAttnDesc* d) {
  USSEG_CHECK_ARG(d, "null descriptor");
  USSEG_CHECK_ARG(d->B > 0 && d->HW > 0 && d->P > 0 && d->R > 0 && d->Cg > 0 && d->Hd > 0, "splitattn: bad sizes");
  USSEG_CHECK_ARG(d->Cg <= 128 && d->Hd <= 64 && d->R <= 4, "splitattn: Cg <= 128, Hd <= 64, R <= 4");
  USSEG_CHECK_ARG(d->Cy_phys % 8 == 0 && d->Co_phys % 8 == 0 && d->Cy_phys >= d->P * d->R * d->Cg && d->Co_phys >= d->P * d->Cg,
                  "splitattn: physical widths");
  USSEG_CHECK_ARG(d->Cy_phys <= 512 && d->ldy % 8 == 0 && d->ldo % 8 == 0, "splitattn: strides");
  return USSEG_OK;
}

extern "C" int usseg_splitattn_gap(const UssegSplitAttnDesc* d, const void* y, float* g, float* ws, usseg_stream_t stream) {
  int rc = sa_check(d);
  if (rc) return rc;
  USSEG_CHECK_ARG(y && g && ws, "null pointer");
  int Cy = d->P * d->R * d->Cg;
  int LPP = lanes_per_pixel(roundup(Cy, 8) / 8);
  int ppb = 4 * (64 / LPP);
  int gx = (int)cdiv64(d->HW, (int64_t)ppb * 8);
  int gmax = USSEG_REDUCE_MAX_BLOCKS / d->B;
  if (gmax > 64) gmax = 64;
  if (gmax < 1) gmax = 1;
  if (gx > gmax) gx = gmax;
  hipLaunchKernelGGL(sa_reduce_kernel, dim3(gx, d->B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, (const bf16_t*)nullptr, d->HW, Cy,
                     d->ldy, 0, d->R, d->Cg, LPP, ws);
  usseg_launch_reduce_finish(ws, d->B, gx, 1, roundup(Cy, 8), Cy, 1.0f, g, nullptr, nullptr, (hipStream_t)stream, 1);   // g is OVERWRITTEN
  return usseg_check_launch("splitattn_gap");
}

extern "C" int usseg_splitattn_apply_bwd_reduce(const UssegSplitAttnDesc* d, const void* y, const void* dout, int32_t lddo, float* ds,
                                                float* ws, usseg_stream_t stream) {
  int rc = sa_check(d);
  if (rc) return rc;
  USSEG_CHECK_ARG(y && dout && ds && ws && lddo % 8 == 0, "null pointer");
  int Cy = d->P * d->R * d->Cg;
  int LPP = lanes_per_pixel(roundup(Cy, 8) / 8);
  int ppb = 4 * (64 / LPP);
  int gx = (int)cdiv64(d->HW, (int64_t)ppb * 8);
  int gmax = USSEG_REDUCE_MAX_BLOCKS / d->B;
  if (gmax > 64) gmax = 64;
  if (gmax < 1) gmax = 1;
  if (gx > gmax) gx = gmax;
  hipLaunchKernelGGL(sa_reduce_kernel, dim3(gx, d->B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, (const bf16_t*)dout, d->HW, Cy,
                     d->ldy, lddo, d->R, d->Cg, LPP, ws);
  usseg_launch_reduce_finish(ws, d->B, gx, 1, roundup(Cy, 8), Cy, d->mult, ds, nullptr, nullptr, (hipStream_t)stream, 1);   // ds is OVERWRITTEN
  return usseg_check_launch("splitattn_apply_bwd_reduce");
}

// ---- the tiny MLP: one 128-thread workgroup per (image b, path p) ----------------------------------------
struct SaMlp {
  UssegSplitAttnDesc d;
  UssegSplitAttnParams p;
  UssegSplitAttnGrads gr;
  const float* g; float* s; float* ws; const float* ds; float* dg;
  int32_t g_rows, g_stride;   // g = [B][g_rows][g_stride] partial rows of the pooled sums (1 row of Cy floats: already summed)
  float* gws;      // backward: per-(path, image) rows of parameter-gradient partials [w1 | b1 | gamma | beta | w2 | b2] (no atomics)
  int32_t Ctot;
  // backward, fused entry: ds arrives as the reduce kernel's partial rows [B][ds_nb][ds_cp] (summed here in row order: no finishing launch)
  const float* ds_rows; int32_t ds_nb, ds_cp; float ds_mult;
};

extern "C" int64_t usseg_splitattn_ws_floats(const UssegSplitAttnDesc* d) {
  if (!d) return 0;
  return (int64_t)d->B * d->P * (d->Cg + 2 * d->Hd);
}

// sum / max over up to 128 LDS floats by wave 0 (the serial tid==0 loops were 64-128 dependent LDS round trips each)
__device__ __forceinline__ float wave_sum(float v) {
  for (int msk = 32; msk >= 1; msk >>= 1) v += __shfl_xor(v, msk, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int msk = 32; msk >= 1; msk >>= 1) v = fmaxf(v, __shfl_xor(v, msk, 64));
  return v;
}

// y[o] = sum_i x[i] * W[i*s_i + o*s_o] for o < No (<= 128), i < Ni, by all SA_THR threads: the input range is cut into
// SA_THR / pow2(No) slices that run side by side (a 128-thread workgroup with one thread per output walked Ni dependent-latency
// batches of L2 loads in a row: 5 us per matrix), partial sums land in part[slice][o] and are added in slice order by sa_mv_sum.
#define SA_THR 256
__device__ __forceinline__ int sa_mv(const float* __restrict__ W, int s_i, int s_o, const float* x, int Ni, int No, float* part /* [8][128] */) {
  const int tid = threadIdx.x;
  const int Np = No <= 32 ? 32 : (No <= 64 ? 64 : 128), parts = SA_THR / Np;
  const int o = tid & (Np - 1), pt = tid / Np;
  const int per = (Ni + parts - 1) / parts, i0 = pt * per, i1 = i0 + per < Ni ? i0 + per : Ni;
  // eight independent weight loads per trip (the kernel is a chain of L2 round trips: what matters is how many loads fly together);
  // fixed summation order
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  if (o < No) {
    const float* w = W + (int64_t)o * s_o;
    int i = i0;
    for (; i + 7 < i1; i += 8) {
      const float w0 = w[(int64_t)i * s_i], w1 = w[(int64_t)(i + 1) * s_i], w2 = w[(int64_t)(i + 2) * s_i], w3 = w[(int64_t)(i + 3) * s_i];
      const float w4 = w[(int64_t)(i + 4) * s_i], w5 = w[(int64_t)(i + 5) * s_i], w6 = w[(int64_t)(i + 6) * s_i], w7 = w[(int64_t)(i + 7) * s_i];
      v0 = fmaf(x[i], w0, v0); v1 = fmaf(x[i + 1], w1, v1); v2 = fmaf(x[i + 2], w2, v2); v3 = fmaf(x[i + 3], w3, v3);
      v0 = fmaf(x[i + 4], w4, v0); v1 = fmaf(x[i + 5], w5, v1); v2 = fmaf(x[i + 6], w6, v2); v3 = fmaf(x[i + 7], w7, v3);
    }
    for (; i + 1 < i1; i += 2) {
      v0 = fmaf(x[i], w[(int64_t)i * s_i], v0);
      v1 = fmaf(x[i + 1], w[(int64_t)(i + 1) * s_i], v1);
    }
    if (i < i1) v0 = fmaf(x[i], w[(int64_t)i * s_i], v0);
  }
  part[pt * 128 + o] = (v0 + v1) + (v2 + v3);
  return parts;
}
__device__ __forceinline__ float sa_mv_sum(const float* part, int parts, int o) {
  float v = 0.f;
  for (int q = 0; q < parts; ++q) v += part[q * 128 + o];
  return v;
}

template <bool BWD>
__global__ __launch_bounds__(SA_THR) void sa_mlp_kernel(const SaMlp a) {
  const UssegSplitAttnDesc& d = a.d;
  const int b = blockIdx.x / d.P, p = blockIdx.x - b * d.P;
  const int tid = threadIdx.x;
  const int Cg = d.Cg, Hd = d.Hd, R = d.R;
  __shared__ float gin[128], h1[64], xh[64], av[64], red[4], dz[4 * 128], da[64], dh[64], part[8 * 128];
  __shared__ float dsl[BWD ? 4 * 128 : 1];
  const int Cy = d.P * R * Cg;
  float* wsb = a.ws + (int64_t)(b * d.P + p) * (Cg + 2 * Hd);
  const float* w1 = a.p.w1 + (int64_t)p * Cg * Hd;
  const float* w2 = a.p.w2 + (int64_t)p * R * Hd * Cg;
  const float gscale = d.mult / (float)d.HW;
  // backward: this (path, image)'s row of parameter-gradient partials; usseg_launch_reduce_finish adds the rows of a path over
  // the images in a fixed order (float atomics made these gradients differ in the last bits from run to run at B > 2)
  float* grow = BWD ? a.gws + ((int64_t)p * d.B + b) * a.Ctot : nullptr;
  float* g_w1 = grow, *g_b1 = g_w1 + Cg * Hd, *g_gamma = g_b1 + Hd, *g_beta = g_gamma + Hd, *g_w2 = g_beta + Hd, *g_b2 = g_w2 + R * Hd * Cg;

  // gin[c] = mult/HW * sum_r g[b][(p*R+r)*Cg + c]     (ResNest.py:173-180); the partial rows of the pooled sums are split over
  // SA_THR / pow2(Cg) thread slices and added in slice order
  {
    const int Np = Cg <= 32 ? 32 : (Cg <= 64 ? 64 : 128), parts = SA_THR / Np;
    const int c = tid & (Np - 1), pt = tid / Np;
    float v = 0.f;
    if (c < Cg)
      for (int r = 0; r < R; ++r) {
        const float* gp = a.g + (int64_t)b * a.g_rows * a.g_stride + (p * R + r) * Cg + c;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;          // the rows are independent loads: eight of them in flight per trip (stage 1 has 256 rows per image)
        int jr = pt;
        const int64_t st = (int64_t)parts * a.g_stride;
        for (; jr + 7 * parts < a.g_rows; jr += 8 * parts) {
          const float* q = gp + (int64_t)jr * a.g_stride;
          const float r0 = q[0], r1 = q[st], r2 = q[2 * st], r3 = q[3 * st], r4 = q[4 * st], r5 = q[5 * st], r6 = q[6 * st], r7 = q[7 * st];
          v0 += r0; v1 += r1; v2 += r2; v3 += r3; v0 += r4; v1 += r5; v2 += r6; v3 += r7;
        }
        for (; jr + parts < a.g_rows; jr += 2 * parts) { v0 += gp[(int64_t)jr * a.g_stride]; v1 += gp[(int64_t)(jr + parts) * a.g_stride]; }
        if (jr < a.g_rows) v0 += gp[(int64_t)jr * a.g_stride];
        v += (v0 + v1) + (v2 + v3);
      }
    part[pt * 128 + c] = v;
    __syncthreads();
    if (tid < Cg) gin[tid] = sa_mv_sum(part, parts, tid) * gscale;
  }
  __syncthreads();
  // dense1 (ResNest.py:182)
  {
    const int parts = sa_mv(w1, Hd, 1, gin, Cg, Hd, part);
    __syncthreads();
    if (tid < Hd) h1[tid] = a.p.b1[p * Hd + tid] + sa_mv_sum(part, parts, tid);
  }
  __syncthreads();
  // norm (LN over Hd <= 64, ResNest.py:183 / BN inference, TBI_ResNest.py:190) + act
  if (tid < 64 && d.norm_mode == 0) {
    float hv = tid < Hd ? h1[tid] : 0.f;
    float mu = wave_sum(hv) / (float)Hd;
    float dv = tid < Hd ? hv - mu : 0.f;
    float var = wave_sum(dv * dv) / (float)Hd;
    if (tid == 0) { red[0] = mu; red[1] = rsqrtf(var + d.eps); }
  }
  __syncthreads();
  for (int j = tid; j < Hd; j += SA_THR) {
    float x;
    if (d.norm_mode == 0) x = (h1[j] - red[0]) * red[1];
    else x = (h1[j] - a.p.mean[p * Hd + j]) * rsqrtf(a.p.var[p * Hd + j] + d.eps);
    xh[j] = x;
    av[j] = apply_act(a.p.gamma[p * Hd + j] * x + a.p.beta[p * Hd + j], d.act, d.alpha);
  }
  __syncthreads();

  if (!BWD) {
    for (int c = tid; c < Cg; c += SA_THR) wsb[c] = gin[c];
    for (int j = tid; j < Hd; j += SA_THR) { wsb[Cg + j] = h1[j]; wsb[Cg + Hd + j] = av[j]; }
    // dense2 per radix branch + softmax over channels (ResNest.py:187-192; TBI_ResNest.py:194-200)
    for (int r = 0; r < R; ++r) {
      float* z = dz;  // reuse as scratch
      {
        const int parts = sa_mv(w2 + (int64_t)r * Hd * Cg, Cg, 1, av, Hd, Cg, part);
        __syncthreads();
        if (tid < Cg) z[tid] = a.p.b2[(p * R + r) * Cg + tid] + sa_mv_sum(part, parts, tid);
      }
      __syncthreads();
      if (tid < 64) {   // Cg <= 128: two channels per lane
        float z0 = tid < Cg ? z[tid] : -INFINITY, z1 = tid + 64 < Cg ? z[tid + 64] : -INFINITY;
        float mx = wave_max(fmaxf(z0, z1));
        float sum = wave_sum((tid < Cg ? __expf(z0 - mx) : 0.f) + (tid + 64 < Cg ? __expf(z1 - mx) : 0.f));
        if (tid == 0) { red[2] = mx; red[3] = 1.f / sum; }
      }
      __syncthreads();
      for (int c = tid; c < Cg; c += SA_THR) {
        float sv = d.use_sigmoid ? 1.f / (1.f + __expf(-z[c])) : __expf(z[c] - red[2]) * red[3];
        a.s[((int64_t)(b * d.P + p) * R + r) * Cg + c] = sv;
      }
      __syncthreads();
    }
  } else {
    if (a.ds_rows) {      // ds[b][cy] = mult * sum over the reduce kernel's partial rows, in row order
      const float r_cg = fdiv_rcp(Cg);
      for (int i = tid; i < R * Cg; i += SA_THR) {
        int r, c;
        fdivmod(i, Cg, r_cg, r, c);
        const float* src = a.ds_rows + (int64_t)b * a.ds_nb * a.ds_cp + (p * R + r) * Cg + c;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;     // four rows in flight per trip, fixed order
        int j = 0;
        for (; j + 3 < a.ds_nb; j += 4) {
          const float q0 = src[(int64_t)j * a.ds_cp], q1 = src[(int64_t)(j + 1) * a.ds_cp], q2 = src[(int64_t)(j + 2) * a.ds_cp], q3 = src[(int64_t)(j + 3) * a.ds_cp];
          t0 += q0; t1 += q1; t2 += q2; t3 += q3;
        }
        for (; j < a.ds_nb; ++j) t0 += src[(int64_t)j * a.ds_cp];
        dsl[r * 128 + c] = a.ds_mult * ((t0 + t1) + (t2 + t3));
      }
      __syncthreads();
    }
    // softmax / sigmoid backward -> dz[r][c]
    for (int r = 0; r < R; ++r) {
      const float* sv = a.s + ((int64_t)(b * d.P + p) * R + r) * Cg;
      const float* dsv = a.ds_rows ? dsl + r * 128 : a.ds + (int64_t)b * Cy + (p * R + r) * Cg;
      if (tid < 64) {
        float t0 = (!d.use_sigmoid && tid < Cg) ? dsv[tid] * sv[tid] : 0.f;
        float t1 = (!d.use_sigmoid && tid + 64 < Cg) ? dsv[tid + 64] * sv[tid + 64] : 0.f;
        float dot = wave_sum(t0 + t1);
        if (tid == 0) red[2] = dot;
      }
      __syncthreads();
      for (int c = tid; c < Cg; c += SA_THR) {
        float v = d.use_sigmoid ? dsv[c] * sv[c] * (1.f - sv[c]) : sv[c] * (dsv[c] - red[2]);
        dz[r * 128 + c] = v;
        g_b2[r * Cg + c] = v;
      }
      __syncthreads();
    }
    // dW2[p][r][j][c] += a[j]*dz[r][c];  da[j] = sum_{r,c} w2*dz
    {
      const float r_cg = fdiv_rcp(Cg), r_hd = fdiv_rcp(Hd);       // (indices < 2^20: fdiv, common.h)
      for (int idx = tid; idx < R * Hd * Cg; idx += SA_THR) {
        int rj, c, r, j;
        fdivmod(idx, Cg, r_cg, rj, c);
        fdivmod(rj, Hd, r_hd, r, j);
        g_w2[idx] = av[j] * dz[r * 128 + c];
      }
    }
    {
      float v = 0.f;
      for (int r = 0; r < R; ++r) {          // da[j] = sum_{r,c} w2[r][j][c] * dz[r][c]
        const int parts = sa_mv(w2 + (int64_t)r * Hd * Cg, 1, Cg, dz + r * 128, Cg, Hd, part);
        __syncthreads();
        if (tid < Hd) v += sa_mv_sum(part, parts, tid);
        __syncthreads();
      }
      if (tid < Hd) da[tid] = v;
    }
    __syncthreads();
    // act + norm backward
    for (int j = tid; j < Hd; j += SA_THR) {
      float ga = a.p.gamma[p * Hd + j];
      float pre = ga * xh[j] + a.p.beta[p * Hd + j];
      float dpre = da[j] * act_grad(pre, d.act, d.alpha);
      g_gamma[j] = dpre * xh[j];
      g_beta[j] = dpre;
      dh[j] = dpre * ga;  // d xhat
    }
    __syncthreads();
    if (d.norm_mode == 0) {
      if (tid < 64) {
        float dj = tid < Hd ? dh[tid] : 0.f, xj = tid < Hd ? xh[tid] : 0.f;
        float s1 = wave_sum(dj), s2 = wave_sum(dj * xj);
        if (tid == 0) { red[2] = s1 / (float)Hd; red[3] = s2 / (float)Hd; }
      }
      __syncthreads();
      for (int j = tid; j < Hd; j += SA_THR) dh[j] = red[1] * (dh[j] - red[2] - xh[j] * red[3]);
    } else {
      for (int j = tid; j < Hd; j += SA_THR) dh[j] = dh[j] * rsqrtf(a.p.var[p * Hd + j] + d.eps);
    }
    __syncthreads();
    for (int j = tid; j < Hd; j += SA_THR) g_b1[j] = dh[j];
    {
      const float r_hd = fdiv_rcp(Hd);
      for (int idx = tid; idx < Cg * Hd; idx += SA_THR) {
        int c, j;
        fdivmod(idx, Hd, r_hd, c, j);
        g_w1[idx] = gin[c] * dh[j];
      }
    }
    // d g_sum[b][(p*R+r)*Cg + c] = mult/HW * sum_j w1[c][j] dh[j]
    {
      const int parts = sa_mv(w1, 1, Hd, dh, Hd, Cg, part);
      __syncthreads();
      if (tid < Cg) {
        const float v = sa_mv_sum(part, parts, tid) * gscale;
        for (int r = 0; r < R; ++r) a.dg[(int64_t)b * Cy + (p * R + r) * Cg + tid] = v;
      }
    }
  }
}

extern "C" int usseg_splitattn_mlp_fwd(const UssegSplitAttnDesc* d, const float* g, int32_t g_rows, int32_t g_stride, const UssegSplitAttnParams* p,
                                       float* s, float* ws, usseg_stream_t stream) {
  int rc = sa_check(d);
  if (rc) return rc;
  USSEG_CHECK_ARG(g && p && s && ws && p->w1 && p->b1 && p->gamma && p->beta && p->w2 && p->b2, "null pointer");
  USSEG_CHECK_ARG(d->norm_mode == 0 || (p->mean && p->var), "affine norm needs mean/var");
  USSEG_CHECK_ARG(g_rows >= 1 && g_stride >= d->P * d->R * d->Cg, "splitattn_mlp: bad pooled-row layout");
  SaMlp a = {};
  a.d = *d; a.p = *p; a.g = g; a.s = s; a.ws = ws; a.g_rows = g_rows; a.g_stride = g_stride;
  hipLaunchKernelGGL(sa_mlp_kernel<false>, dim3(d->B * d->P), dim3(SA_THR), 0, (hipStream_t)stream, a);
  return usseg_check_launch("splitattn_mlp_fwd");
}

extern "C" int64_t usseg_splitattn_mlp_bwd_ws_floats(const UssegSplitAttnDesc* d) {
  if (!d) return 0;
  return (int64_t)d->B * d->P * (d->Cg * d->Hd + 3 * d->Hd + d->R * d->Hd * d->Cg + d->R * d->Cg);
}

extern "C" int usseg_splitattn_mlp_bwd(const UssegSplitAttnDesc* d, const float* g, int32_t g_rows, int32_t g_stride, const UssegSplitAttnParams* p,
                                       const float* s, const float* ws, const float* ds, float* dg, const UssegSplitAttnGrads* grads,
                                       float* grad_ws, usseg_stream_t stream) {
  int rc = sa_check(d);
  if (rc) return rc;
  USSEG_CHECK_ARG(g && p && s && ds && dg && grads && grad_ws, "null pointer");
  USSEG_CHECK_ARG(g_rows >= 1 && g_stride >= d->P * d->R * d->Cg, "splitattn_mlp: bad pooled-row layout");
  USSEG_CHECK_ARG(grads->w1 && grads->b1 && grads->gamma && grads->beta && grads->w2 && grads->b2, "null grad pointer");
  SaMlp a = {};
  a.d = *d; a.p = *p; a.gr = *grads; a.g = g; a.s = const_cast<float*>(s); a.ws = const_cast<float*>(ws); a.ds = ds; a.dg = dg;
  a.g_rows = g_rows; a.g_stride = g_stride;
  const int Cg = d->Cg, Hd = d->Hd, R = d->R;
  a.Ctot = Cg * Hd + 3 * Hd + R * Hd * Cg + R * Cg;
  a.gws = usseg_defer_reduce_ws((hipStream_t)stream, grad_ws, (int64_t)d->B * d->P * a.Ctot);   // a private region while finishes are deferred
  hipLaunchKernelGGL(sa_mlp_kernel<true>, dim3(d->B * d->P), dim3(SA_THR), 0, (hipStream_t)stream, a);
  // per parameter: rows (path, image) of Ctot floats -> the path's variable (per-path variables are adjacent: [P][numel])
  const int sizes[6] = {Cg * Hd, Hd, Hd, Hd, R * Hd * Cg, R * Cg};
  float* dst[6] = {grads->w1, grads->b1, grads->gamma, grads->beta, grads->w2, grads->b2};
  int off = 0;
  for (int i = 0; i < 6; ++i) {
    usseg_launch_reduce_finish(a.gws + off, d->P, d->B, 1, a.Ctot, sizes[i], 1.f, dst[i], nullptr, nullptr, (hipStream_t)stream);
    off += sizes[i];
  }
  return usseg_check_launch("splitattn_mlp_bwd");
}

// usseg_splitattn_apply_bwd_reduce + usseg_splitattn_mlp_bwd without the finishing launch between them: the MLP kernel sums the reduce
// kernel's partial rows itself (ResNest.py:194-197 backward into :186-192)
extern "C" int usseg_splitattn_bwd_fused(const UssegSplitAttnDesc* d, const void* y, const void* dout, int32_t lddo, const float* g, int32_t g_rows,
                                         int32_t g_stride, const UssegSplitAttnParams* p, const float* s, const float* ws, float* dg,
                                         const UssegSplitAttnGrads* grads, float* reduce_ws, float* grad_ws, usseg_stream_t stream) {
  const float* ds = reduce_ws;
  int rc = sa_check(d);
  if (rc) return rc;
  USSEG_CHECK_ARG(g && p && s && ds && dg && grads && grad_ws && y && dout && lddo % 8 == 0, "null pointer");
  const int Cy = d->P * d->R * d->Cg;
  const int LPPr = lanes_per_pixel(roundup(Cy, 8) / 8);
  int gx = (int)cdiv64(d->HW, (int64_t)(4 * (64 / LPPr)) * 8);
  int gmax = USSEG_REDUCE_MAX_BLOCKS / d->B;
  if (gmax > 64) gmax = 64;
  if (gmax < 1) gmax = 1;
  if (gx > gmax) gx = gmax;
  hipLaunchKernelGGL(sa_reduce_kernel, dim3(gx, d->B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, (const bf16_t*)dout, d->HW, Cy,
                     d->ldy, lddo, d->R, d->Cg, LPPr, reduce_ws);
  USSEG_CHECK_ARG(g_rows >= 1 && g_stride >= d->P * d->R * d->Cg, "splitattn_mlp: bad pooled-row layout");
  USSEG_CHECK_ARG(grads->w1 && grads->b1 && grads->gamma && grads->beta && grads->w2 && grads->b2, "null grad pointer");
  SaMlp a = {};
  a.d = *d; a.p = *p; a.gr = *grads; a.g = g; a.s = const_cast<float*>(s); a.ws = const_cast<float*>(ws); a.ds = nullptr; a.dg = dg;
  a.ds_rows = reduce_ws; a.ds_nb = gx; a.ds_cp = roundup(Cy, 8); a.ds_mult = d->mult;
  a.g_rows = g_rows; a.g_stride = g_stride;
  const int Cg = d->Cg, Hd = d->Hd, R = d->R;
  a.Ctot = Cg * Hd + 3 * Hd + R * Hd * Cg + R * Cg;
  a.gws = usseg_defer_reduce_ws((hipStream_t)stream, grad_ws, (int64_t)d->B * d->P * a.Ctot);   // a private region while finishes are deferred
  hipLaunchKernelGGL(sa_mlp_kernel<true>, dim3(d->B * d->P), dim3(SA_THR), 0, (hipStream_t)stream, a);
  // per parameter: rows (path, image) of Ctot floats -> the path's variable (per-path variables are adjacent: [P][numel])
  const int sizes[6] = {Cg * Hd, Hd, Hd, Hd, R * Hd * Cg, R * Cg};
  float* dst[6] = {grads->w1, grads->b1, grads->gamma, grads->beta, grads->w2, grads->b2};
  int off = 0;
  for (int i = 0; i < 6; ++i) {
    usseg_launch_reduce_finish(a.gws + off, d->P, d->B, 1, a.Ctot, sizes[i], 1.f, dst[i], nullptr, nullptr, (hipStream_t)stream);
    off += sizes[i];
  }
  return usseg_check_launch("splitattn_bwd_fused");
}

// out[b,hw,p*Cg+c] = mult * sum_r y[b,hw,(p*R+r)*Cg+c] * s[b][p][r][c]   (ResNest.py:194-197)
// BWD: dy[b,hw,cy] = mult * s[b][cy] * dout[b,hw,co(cy)] + dg[b][cy]
template <bool BWD>
__global__ __launch_bounds__(256) void sa_apply_kernel(const bf16_t* in, const float* s, const float* dg, int B, int HW, int P, int R, int Cg,
                                                        int ldi, int ldo, int CHo, float mult, bf16_t* out) {
  // CHo = chunks of the OUTPUT row (fwd: Co_phys/8, bwd: Cy_phys/8)
  const int64_t total = (int64_t)B * HW * CHo;
  const int Cy = P * R * Cg, Co = P * Cg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t pix = i / CHo;
    int c0 = (int)(i - pix * CHo) * 8;
    int b = (int)(pix / HW);
    float o[8];
    if (!BWD) {
      if (R == 1) {
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(in + pix * ldi + c0), v);
        float sv[8];   // unconditional loads at clamped indices (a load under a per-element condition serialises: one L2 round trip each)
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[j] = s[(int64_t)b * Cy + min(c0 + j, Co - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (c0 + j < Co) ? mult * v[j] * sv[j] : 0.f;
      } else if ((Cg & 7) == 0) {
        // groups are whole 8-channel chunks: the chunk lies in one path and each branch's operands are one 16-byte load of y
        // and two of s (the element-wise form below issues 64 scalar loads per thread: 1.8 TB/s on Arch A's stage-1 tensor)
        const int p = c0 / Cg, cc = c0 - p * Cg;
        const bool ok = c0 < Co;
        uint4 raw[4];
        float4 w0[4], w1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cy = ok ? (p * R + min(r, R - 1)) * Cg + cc : 0;
          raw[r] = *reinterpret_cast<const uint4*>(in + pix * ldi + cy);
          w0[r] = *reinterpret_cast<const float4*>(s + (int64_t)b * Cy + cy);
          w1[r] = *reinterpret_cast<const float4*>(s + (int64_t)b * Cy + cy + 4);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r < R) {
            float v[8];
            unpack8(raw[r], v);
            o[0] += v[0] * w0[r].x; o[1] += v[1] * w0[r].y; o[2] += v[2] * w0[r].z; o[3] += v[3] * w0[r].w;
            o[4] += v[4] * w1[r].x; o[5] += v[5] * w1[r].y; o[6] += v[6] * w1[r].z; o[7] += v[7] * w1[r].w;
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = ok ? mult * o[j] : 0.f;
      } else {
        // R <= 4 distinct branches: all R*8 (value, weight) pairs are loaded unconditionally at clamped indices first
        float xv[4][8], wv[4][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = min(c0 + j, Co - 1);
          const int p = c / Cg, cc = c - p * Cg;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int cy = (p * R + min(r, R - 1)) * Cg + cc;
            xv[r][j] = bf2f(in[pix * ldi + cy]);
            wv[r][j] = s[(int64_t)b * Cy + cy];
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float acc = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc += r < R ? xv[r][j] * wv[r][j] : 0.f;
          o[j] = (c0 + j < Co) ? mult * acc : 0.f;
        }
      }
    } else {
      if (R == 1) {
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(in + pix * ldi + c0), v);
        float sv[8], dv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int64_t ix = (int64_t)b * Cy + min(c0 + j, Cy - 1);
          sv[j] = s[ix];
          dv[j] = dg[ix];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (c0 + j < Cy) ? mult * v[j] * sv[j] + dv[j] : 0.f;
      } else if ((Cg & 7) == 0) {
        const bool ok = c0 < Cy;
        const int cy = ok ? c0 : 0;
        const int pr = cy / Cg;
        const int co = (pr / R) * Cg + (cy - pr * Cg);
        float dv[8];
        unpack8(*reinterpret_cast<const uint4*>(in + pix * ldi + co), dv);
        const float4 s0 = *reinterpret_cast<const float4*>(s + (int64_t)b * Cy + cy), s1 = *reinterpret_cast<const float4*>(s + (int64_t)b * Cy + cy + 4);
        const float4 g0 = *reinterpret_cast<const float4*>(dg + (int64_t)b * Cy + cy), g1 = *reinterpret_cast<const float4*>(dg + (int64_t)b * Cy + cy + 4);
        o[0] = mult * dv[0] * s0.x + g0.x; o[1] = mult * dv[1] * s0.y + g0.y; o[2] = mult * dv[2] * s0.z + g0.z; o[3] = mult * dv[3] * s0.w + g0.w;
        o[4] = mult * dv[4] * s1.x + g1.x; o[5] = mult * dv[5] * s1.y + g1.y; o[6] = mult * dv[6] * s1.z + g1.z; o[7] = mult * dv[7] * s1.w + g1.w;
        if (!ok) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = 0.f;
        }
      } else {
        float dv[8], sv[8], gv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int cy = min(c0 + j, Cy - 1);
          const int pr = cy / Cg;
          const int co = (pr / R) * Cg + (cy - pr * Cg);
          dv[j] = bf2f(in[pix * ldi + co]);
          sv[j] = s[(int64_t)b * Cy + cy];
          gv[j] = dg[(int64_t)b * Cy + cy];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (c0 + j < Cy) ? mult * dv[j] * sv[j] + gv[j] : 0.f;
      }
    }
    *reinterpret_cast<uint4*>(out + pix * ldo + c0) = pack8(o);
  }
}

extern "C" int usseg_splitattn_apply_fwd(const UssegSplitAttnDesc* d, const void* y, const float* s, void* out, usseg_stream_t stream) {
  int rc = sa_check(d);
  if (rc) return rc;
  USSEG_CHECK_ARG(y && s && out, "null pointer");
  int64_t total = (int64_t)d->B * d->HW * (d->Co_phys / 8);
  int64_t g = cdiv64(total, 256 * 4);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(sa_apply_kernel<false>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, s, (const float*)nullptr,
                     d->B, d->HW, d->P, d->R, d->Cg, d->ldy, d->ldo, d->Co_phys / 8, d->mult, (bf16_t*)out);
  return usseg_check_launch("splitattn_apply_fwd");
}

extern "C" int usseg_splitattn_apply_bwd_dy(const UssegSplitAttnDesc* d, const void* dout, int32_t lddo, const float* s, const float* dg,
                                            void* dy, int32_t lddy, usseg_stream_t stream) {
  int rc = sa_check(d);
  if (rc) return rc;
  USSEG_CHECK_ARG(dout && s && dg && dy && lddo % 8 == 0 && lddy % 8 == 0, "null pointer");
  int64_t total = (int64_t)d->B * d->HW * (d->Cy_phys / 8);
  int64_t g = cdiv64(total, 256 * 4);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(sa_apply_kernel<true>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout, s, dg, d->B, d->HW,
                     d->P, d->R, d->Cg, lddo, lddy, d->Cy_phys / 8, d->mult, (bf16_t*)dy);
  return usseg_check_launch("splitattn_apply_bwd_dy");
}

// ------------------------------------------------------------------------------------------ head softmax + loss
// One pixel: softmax, probabilities out, loss term (returned) and d(sum loss)/d logits.
// CC = compile-time class count (0 = read d.C): with a run-time C the per-class register arrays are indexed through select chains
// (600 v_cndmask + 480 s_cselect in the ISA, 36 us for 1M pixels); the reference's 3 classes get the unrolled form.
template <int CC>
__device__ __forceinline__ float softmax_loss_pixel(const UssegLossDesc& d, int b, int hw, const float* __restrict__ logits,
                                                    const float* __restrict__ y_true, const float* __restrict__ scale, float* __restrict__ probs,
                                                    bf16_t* __restrict__ dlogits) {
  const int C = CC ? CC : d.C;
  const int64_t m = (int64_t)b * d.HW + hw;      // (image, pixel) come from the grid: no 64-bit division per pixel
  float z[8], p[8], yt[8];
  float mx = -INFINITY;
  // quad (space-to-depth) layout of the head's output and its gradient: pixel (y, x) of the full-resolution map lives in
  // slot 4*((y&1)*2 + (x&1)) of the 16-channel pixel (y/2, x/2) - what the 2x2-tap form of the stride-2 head produces
  int64_t lbase = m * d.ldl, dbase = m * d.lddl;
  if (d.quad_w) {
    const int y = (d.quad_w & (d.quad_w - 1)) == 0 ? hw >> (31 - __clz(d.quad_w)) : hw / d.quad_w, x = hw - y * d.quad_w;
    const int64_t q = (b * (d.HW / d.quad_w / 2) + (y >> 1)) * (d.quad_w / 2) + (x >> 1);
    const int slot = 4 * ((y & 1) * 2 + (x & 1));
    lbase = q * d.ldl + slot;
    dbase = q * d.lddl + slot;
  }
  _Pragma("unroll") for (int c = 0; c < C; ++c) { z[c] = logits[lbase + c]; mx = fmaxf(mx, z[c]); }
  float sum = 0.f;
  _Pragma("unroll") for (int c = 0; c < C; ++c) { p[c] = __expf(z[c] - mx); sum += p[c]; }
  float inv = 1.f / sum;
  _Pragma("unroll") for (int c = 0; c < C; ++c) { p[c] *= inv; probs[m * C + c] = p[c]; }
  if (!y_true) return 0.f;
  _Pragma("unroll") for (int c = 0; c < C; ++c) yt[c] = y_true[m * C + c];
  float dLdp[8];
  float l = 0.f;
  if (d.loss_kind == 0) {
    // CategoricalCrossentropy(label_smoothing) on probabilities (VisionTransformer.py:205)
    float S = 0.f;
    _Pragma("unroll") for (int c = 0; c < C; ++c) S += p[c];
    float u[8], ubar = 0.f;
    _Pragma("unroll") for (int c = 0; c < C; ++c) {
      float ys = yt[c] * (1.f - d.label_smoothing) + d.label_smoothing / (float)C;
      float q = p[c] / S;
      float qc = fminf(fmaxf(q, d.clip_eps), 1.f - d.clip_eps);
      l -= ys * __logf(qc);
      u[c] = (q > d.clip_eps && q < 1.f - d.clip_eps) ? -ys / qc : 0.f;
      ubar += u[c] * q;
    }
    l *= d.inv_global_batch;
    _Pragma("unroll") for (int c = 0; c < C; ++c) dLdp[c] = (u[c] - ubar) / S * d.inv_global_batch;
  } else {
    // my_loss_cat (TBI_ResNest.py:234-248): -sum_b y*log(p+1e-7)*scale[hw][c]; the [H,W] map is summed for the gradient
    _Pragma("unroll") for (int c = 0; c < C; ++c) {
      float sc = scale[(int64_t)hw * C + c];
      l -= yt[c] * __logf(p[c] + 1e-7f) * sc;
      dLdp[c] = -yt[c] * sc / (p[c] + 1e-7f);
    }
  }
  if (dlogits) {
    float dot = 0.f;
    _Pragma("unroll") for (int c = 0; c < C; ++c) dot += dLdp[c] * p[c];
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (c < C) ? p[c] * (dLdp[c] - dot) : 0.f;
    if (d.quad_w) {
      uint2 v;
      v.x = pack2bf(o[0], o[1]); v.y = pack2bf(o[2], o[3]);
      *reinterpret_cast<uint2*>(dlogits + dbase) = v;
    } else {
      *reinterpret_cast<uint4*>(dlogits + dbase) = pack8(o);
    }
  }
  return l;
}

// Reproducible reductions: loss_kind 0 sums the workgroup totals in workgroup order (grid_ordered_sum, loss = [USSEG_ACC_FLOATS]);
// loss_kind 1 gives every pixel of the [H,W] map to ONE thread that walks the batch in order (no atomics on the map).
template <int CC>
__global__ __launch_bounds__(256) void softmax_loss_kernel(const UssegLossDesc d, const float* __restrict__ logits, const float* __restrict__ y_true,
                                                            const float* __restrict__ scale, float* __restrict__ probs, float* loss,
                                                            bf16_t* __restrict__ dlogits) {
  __shared__ float red[4];
  if (d.loss_kind == 1 && y_true) {
    const int nb = (int)(d.M / d.HW);
    for (int64_t hw = (int64_t)blockIdx.x * 256 + threadIdx.x; hw < d.HW; hw += (int64_t)gridDim.x * 256) {
      float l = 0.f;
      for (int b = 0; b < nb; ++b) l += softmax_loss_pixel<CC>(d, b, (int)hw, logits, y_true, scale, probs, dlogits);
      loss[hw] = l;
    }
    return;
  }
  // grid (pixel blocks, images); four independent pixels in flight per thread (the per-pixel chain of dependent loads is latency-bound)
  float lsum = 0.f;
  const int b = blockIdx.y, stride = gridDim.x * 256;
  int hw = blockIdx.x * 256 + threadIdx.x;
  for (; hw + 3 * stride < d.HW; hw += 4 * stride) {
    float l4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) l4[u] = softmax_loss_pixel<CC>(d, b, hw + u * stride, logits, y_true, scale, probs, dlogits);
    lsum += (l4[0] + l4[1]) + (l4[2] + l4[3]);
  }
  for (; hw < d.HW; hw += stride) lsum += softmax_loss_pixel<CC>(d, b, hw, logits, y_true, scale, probs, dlogits);
  if (y_true) {
    for (int msk = 32; msk >= 1; msk >>= 1) lsum += __shfl_xor(lsum, msk, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
    __syncthreads();
    grid_ordered_sum((red[0] + red[1]) + (red[2] + red[3]), loss, gridDim.x * gridDim.y, blockIdx.y * gridDim.x + blockIdx.x);
  }
}

extern "C" int usseg_softmax_loss_fwd_bwd(const UssegLossDesc* d, const float* logits, const float* y_true, const float* scale, float* probs,
                                          float* loss, void* dlogits, usseg_stream_t stream) {
  USSEG_CHECK_ARG(d && logits && probs, "null pointer");
  USSEG_CHECK_ARG(d->C >= 1 && d->C <= 8 && d->ldl >= d->C, "softmax_loss: 1 <= C <= 8");
  USSEG_CHECK_ARG(!y_true || loss, "loss pointer required with y_true");
  USSEG_CHECK_ARG(d->quad_w == 0 || (d->C <= 4 && d->ldl == 16 && d->quad_w % 2 == 0 && d->HW % (2 * d->quad_w) == 0),
                  "softmax_loss: quad layout needs C <= 4, ldl == 16, even H and W");
  USSEG_CHECK_ARG(!dlogits || (d->lddl == (d->quad_w ? 16 : 8)), "dlogits stride must be 8 (16 in the quad layout)");
  USSEG_CHECK_ARG(d->loss_kind == 0 || (d->loss_kind == 1 && scale), "loss_kind 1 needs scale");
  if (d->M <= 0) return USSEG_OK;
  USSEG_CHECK_ARG(d->HW > 0 && d->M % d->HW == 0, "softmax_loss: M must be a multiple of HW");
  USSEG_CHECK_ARG(d->HW < (1 << 30) && d->M / d->HW <= 65535, "softmax_loss: image too large");
  const int nb = (int)(d->M / d->HW);
  dim3 grid;
  if (y_true && d->loss_kind == 1) {
    int64_t g = cdiv64(d->HW, 256);
    grid = dim3((unsigned)(g > 2048 ? 2048 : g));
  } else {
    static const int ppt = getenv("USSEG_LOSS_PPT") ? atoi(getenv("USSEG_LOSS_PPT")) : 4;
    int64_t g = cdiv64(d->HW, 256 * ppt);          // <= 2048 workgroups in all: the slots of the ordered sum
    if (g * nb > 2048) g = 2048 / nb > 0 ? 2048 / nb : 1;
    USSEG_CHECK_ARG(g * nb <= 2048, "softmax_loss: more than 2048 images");
    grid = dim3((unsigned)g, (unsigned)nb);
  }
  if (d->C == 3) hipLaunchKernelGGL(softmax_loss_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, *d, logits, y_true, scale, probs, loss, (bf16_t*)dlogits);
  else hipLaunchKernelGGL(softmax_loss_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, *d, logits, y_true, scale, probs, loss, (bf16_t*)dlogits);
  return usseg_check_launch("softmax_loss");
}

// ---- the quad-form head, its softmax and the loss in ONE launch -------------------------------------------------------------------
// DecoderCup's head (Conv2DTranspose(classes <= 4, 3 or 4 taps, strides 2), Decoder.py:119-121,142) in its quad form is a 3x3 stride-1 conv at
// the INPUT resolution with 16 outputs = (output parity class)*4 + class (usseg.h, "quad form"); softmax and CategoricalCrossentropy follow
// per output pixel (VisionTransformer.py:205,225-229).  Unfused that was conv (writes 64 B of fp32 logits per input pixel) -> softmax + loss
// (reads them back): 26 + 28 us and 134 MB per 16-image step for 0.6 GFLOP.  Here a workgroup stages a 16x16-pixel tile (+1 halo) of the
// 16-channel input in LDS, runs K = 144 on MFMA with the weight rows as the A operand - so a lane ends up with the 4 consecutive outputs
// of ONE parity class of ONE pixel, i.e. the logits of one output pixel - and finishes softmax, probabilities, loss term and d loss / d logits in
// registers.  The loss is the ordered grid-wide sum of the workgroup totals (bitwise reproducible).
struct HeadQuad {
  const bf16_t* x; const bf16_t* wq; const float* bias; const float* y_true;
  float* probs; float* loss; bf16_t* dl;
  int32_t B, h, w, ldx, tiles_x, C;
  float label_smoothing, clip_eps, inv_global_batch;
};
template <int CC, int CP>
__global__ __launch_bounds__(256) void head_quad_loss_kernel(const HeadQuad p) {      // (72 channels: 192 VGPRs, two workgroups per CU; a bound of three spills)
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TW = 16, HWD = TW + 2, CH = CP / 8, KTOT = 9 * CP, KS = (KTOT + 31) / 32, NIT = (HWD * HWD * CH + 255) / 256;
  __shared__ __attribute__((aligned(16))) bf16_t XT[HWD * HWD * CP];        // the halo tile [18*18][CP]
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, q = lane >> 4;
  const int b = blockIdx.y, tyi = blockIdx.x / p.tiles_x, txi = blockIdx.x - tyi * p.tiles_x;
  const int ty0 = tyi * TW, tx0 = txi * TW;
  // every global load of the tile is issued before the first LDS store (a load -> store loop pays one memory latency per trip: 12 trips);
  // zeros outside the image: the conv's 'same' padding
  uint4 xv[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int it = tid + i * 256;
    const int hp = it / CH, c = it - hp * CH;
    const int hy = hp / HWD, hx = hp - hy * HWD;
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
    const bool ok = it < HWD * HWD * CH && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
    xv[i] = ok ? *reinterpret_cast<const uint4*>(p.x + (((int64_t)b * p.h + gy) * p.w + gx) * p.ldx + c * 8) : make_uint4(0, 0, 0, 0);
  }
  // the labels of this lane's four output pixels: independent of the conv, so their loads fly under the tile staging and the MFMAs
  const int pa = q >> 1, pb = q & 1;           // this lane's output parity
  const int H2 = 2 * p.h, W2 = 2 * p.w;
  float ytv[4][CC];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gy = ty0 + wv + 4 * i, gx = tx0 + r;
    const bool ok = p.y_true != nullptr && gy < p.h && gx < p.w;
    const int64_t m = ok ? ((int64_t)b * H2 + 2 * gy + pa) * W2 + 2 * gx + pb : 0;
#pragma unroll
    for (int c = 0; c < CC; ++c) ytv[i][c] = ok ? p.y_true[m * CC + c] : 0.f;
  }
  // weight fragments, all K steps, in registers: A[n = r][k = ks*32 + q*8 ..]; k = tap*CP + channel; the K tail (k >= KTOT) is zero
  bf16x8_t a[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int k0 = ks * 32 + q * 8;
    a[ks] = *reinterpret_cast<const bf16x8_t*>(p.wq + r * KTOT + (k0 < KTOT ? k0 : 0));
  }
  if ((KTOT & 31) != 0) {      // (only the last step has a tail; zeroed here, once, not behind every load)
    const int k0 = (KS - 1) * 32 + q * 8;
    if (k0 >= KTOT) {
#pragma unroll
      for (int e = 0; e < 8; ++e) a[KS - 1][e] = (__bf16)0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int it = tid + i * 256;
    const int hp = it / CH, c = it - hp * CH;
    if (it < HWD * HWD * CH) *reinterpret_cast<uint4*>(XT + hp * CP + c * 8) = xv[i];
  }
  __syncthreads();
  float bias[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) bias[c] = c < CC ? p.bias[c] : 0.f;
  float lsum = 0.f;
  f32x4_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // B[k][pixel r of row wv + 4i]: the K tail reads a valid (clamped) pixel chunk against a zero weight fragment
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int k0 = ks * 32 + q * 8;
    const int kc = k0 < KTOT ? k0 : 0;
    const int tap = kc / CP, c0 = kc - tap * CP, ty = tap / 3, tx = tap - ty * 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x8_t bx = *reinterpret_cast<const bf16x8_t*>(XT + ((wv + 4 * i + ty) * HWD + (r + tx)) * CP + c0);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], bx, acc[i], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gy = ty0 + wv + 4 * i, gx = tx0 + r;
    if (gy < p.h && gx < p.w) {
      const int64_t m = ((int64_t)b * H2 + 2 * gy + pa) * W2 + 2 * gx + pb;
      float z[4], pr[4];
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < CC; ++c) { z[c] = acc[i][c] + bias[c]; mx = fmaxf(mx, z[c]); }
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < CC; ++c) { pr[c] = __expf(z[c] - mx); sum += pr[c]; }
      const float inv = 1.f / sum;
#pragma unroll
      for (int c = 0; c < CC; ++c) { pr[c] *= inv; p.probs[m * CC + c] = pr[c]; }
      if (p.y_true) {       // CategoricalCrossentropy(label_smoothing) on probabilities: the arithmetic of softmax_loss_pixel, loss_kind 0
        float u[4], dLdp[4];
        const float* yt = ytv[i];
        float S = 0.f, ubar = 0.f, l = 0.f;
#pragma unroll
        for (int c = 0; c < CC; ++c) S += pr[c];
#pragma unroll
        for (int c = 0; c < CC; ++c) {
          const float ys = yt[c] * (1.f - p.label_smoothing) + p.label_smoothing / (float)CC;
          const float qq = pr[c] / S;
          const float qc = fminf(fmaxf(qq, p.clip_eps), 1.f - p.clip_eps);
          l -= ys * __logf(qc);
          u[c] = (qq > p.clip_eps && qq < 1.f - p.clip_eps) ? -ys / qc : 0.f;
          ubar += u[c] * qq;
        }
        lsum += l * p.inv_global_batch;
        if (p.dl) {
          float dot = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < CC; ++c) { dLdp[c] = (u[c] - ubar) / S * p.inv_global_batch; dot += dLdp[c] * pr[c]; }
#pragma unroll
          for (int c = 0; c < CC; ++c) o[c] = pr[c] * (dLdp[c] - dot);
          uint2 v;
          v.x = pack2bf(o[0], o[1]); v.y = pack2bf(o[2], o[3]);
          *reinterpret_cast<uint2*>(p.dl + (((int64_t)b * p.h + gy) * p.w + gx) * 16 + 4 * q) = v;
        }
      }
    }
  }
  if (p.y_true) {
    for (int msk = 32; msk >= 1; msk >>= 1) lsum += __shfl_xor(lsum, msk, 64);
    if (lane == 0) red[wv] = lsum;
    __syncthreads();
    grid_ordered_sum((red[0] + red[1]) + (red[2] + red[3]), p.loss, gridDim.x * gridDim.y, blockIdx.y * gridDim.x + blockIdx.x);
  }
#endif
}
template <int CC, int CP>
static void head_quad_launch(const HeadQuad& p, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((head_quad_loss_kernel<CC, CP>), grid, dim3(256), 0, s, p);
}
template <int CP>
static void head_quad_launch_c(const HeadQuad& p, dim3 grid, hipStream_t s) {
  switch (p.C) {
    case 1: head_quad_launch<1, CP>(p, grid, s); break;
    case 2: head_quad_launch<2, CP>(p, grid, s); break;
    case 3: head_quad_launch<3, CP>(p, grid, s); break;
    default: head_quad_launch<4, CP>(p, grid, s); break;
  }
}
extern "C" int usseg_head_quad_softmax_loss(const void* x, int32_t B, int32_t h, int32_t w, int32_t Cin_phys, int32_t ldx, const void* wq, const float* bias, int32_t C,
                                            const float* y_true, float* probs, float* loss, void* dlogits, float label_smoothing, float clip_eps,
                                            float inv_global_batch, usseg_stream_t stream) {
  USSEG_CHECK_ARG(x && wq && bias && probs && B > 0 && h > 0 && w > 0 && Cin_phys >= 8 && Cin_phys % 8 == 0 && ldx >= Cin_phys && ldx % 8 == 0 && C >= 1 && C <= 4,
                  "head_quad_softmax_loss: bad arguments");
  USSEG_CHECK_ARG(!y_true || loss, "head_quad_softmax_loss: loss pointer required with y_true");
  USSEG_CHECK_ARG(!dlogits || y_true, "head_quad_softmax_loss: dlogits needs y_true");
  HeadQuad p = {};
  p.x = (const bf16_t*)x; p.wq = (const bf16_t*)wq; p.bias = bias; p.y_true = y_true; p.probs = probs; p.loss = loss; p.dl = (bf16_t*)dlogits;
  p.B = B; p.h = h; p.w = w; p.ldx = ldx; p.C = C; p.tiles_x = (w + 15) / 16;
  p.label_smoothing = label_smoothing; p.clip_eps = clip_eps; p.inv_global_batch = inv_global_batch;
  const int64_t tiles = (int64_t)p.tiles_x * ((h + 15) / 16);
  // instantiated for the head widths of the models (Decoder.py: 16 channels + the re-injected 56 = 72; 16 for a plain head); more workgroups
  // than the ordered sum has slots, or another width: the caller runs conv + softmax_loss
  if (tiles * B > USSEG_ACC_FLOATS - 2 || B > 65535 || (Cin_phys != 16 && Cin_phys != 72)) {
    usseg_set_error("head_quad_softmax_loss: no fused kernel for this size");
    return USSEG_ERR_UNSUPPORTED;
  }
  const dim3 grid((unsigned)tiles, (unsigned)B);
  hipStream_t s = (hipStream_t)stream;
  const int slot = usseg_prof_start(4, s);       // timed with the fused tile kernels (a conv + what follows it in one launch)
  if (Cin_phys == 72) head_quad_launch_c<72>(p, grid, s);
  else head_quad_launch_c<16>(p, grid, s);
  usseg_prof_stop(4, slot, s);
  return usseg_check_launch("head_quad_softmax_loss");
}

// compute_loss / my_loss_cat on probabilities (the reference's public loss methods): same arithmetic as the fused kernel above
__device__ __forceinline__ float loss_from_probs_pixel(const UssegLossDesc& d, int64_t m, const float* probs, const float* y_true,
                                                       const float* scale) {
  const int C = d.C;
  float p[8], yt[8], l = 0.f;
  for (int c = 0; c < C; ++c) { p[c] = probs[m * C + c]; yt[c] = y_true[m * C + c]; }
  if (d.loss_kind == 0) {
    float S = 0.f;
    for (int c = 0; c < C; ++c) S += p[c];
    for (int c = 0; c < C; ++c) {
      float ys = yt[c] * (1.f - d.label_smoothing) + d.label_smoothing / (float)C;
      float qc = fminf(fmaxf(p[c] / S, d.clip_eps), 1.f - d.clip_eps);
      l -= ys * __logf(qc);
    }
    return l * d.inv_global_batch;
  }
  const int hw = (int)(m % d.HW);
  for (int c = 0; c < C; ++c) l -= yt[c] * __logf(p[c] + 1e-7f) * scale[(int64_t)hw * C + c];
  return l;
}
__global__ __launch_bounds__(256) void loss_from_probs_kernel(const UssegLossDesc d, const float* probs, const float* y_true, const float* scale,
                                                               float* loss) {
  __shared__ float red[4];
  if (d.loss_kind == 1) {
    const int nb = (int)(d.M / d.HW);
    for (int64_t hw = (int64_t)blockIdx.x * 256 + threadIdx.x; hw < d.HW; hw += (int64_t)gridDim.x * 256) {
      float l = 0.f;
      for (int b = 0; b < nb; ++b) l += loss_from_probs_pixel(d, (int64_t)b * d.HW + hw, probs, y_true, scale);
      loss[hw] = l;
    }
    return;
  }
  float lsum = 0.f;
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < d.M; m += (int64_t)gridDim.x * 256)
    lsum += loss_from_probs_pixel(d, m, probs, y_true, scale);
  for (int msk = 32; msk >= 1; msk >>= 1) lsum += __shfl_xor(lsum, msk, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  __syncthreads();
  grid_ordered_sum((red[0] + red[1]) + (red[2] + red[3]), loss, gridDim.x);
}
extern "C" int usseg_loss_from_probs(const UssegLossDesc* d, const float* probs, const float* y_true, const float* scale, float* loss,
                                     usseg_stream_t stream) {
  USSEG_CHECK_ARG(d && probs && y_true && loss, "loss_from_probs: null pointer");
  USSEG_CHECK_ARG(d->C >= 1 && d->C <= 8 && d->HW > 0, "loss_from_probs: 1 <= C <= 8");
  USSEG_CHECK_ARG(d->loss_kind == 0 || (d->loss_kind == 1 && scale), "loss_kind 1 needs scale");
  if (d->M <= 0) return USSEG_OK;
  USSEG_CHECK_ARG(d->M % d->HW == 0, "loss_from_probs: M must be a multiple of HW");
  int64_t g = d->loss_kind == 1 ? cdiv64(d->HW, 256) : cdiv64(d->M, 256 * 4);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(loss_from_probs_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *d, probs, y_true, scale, loss);
  return usseg_check_launch("loss_from_probs");
}

// Accuracy metric of a step (TBI_ResNest.py:48-51: mean over the pixels of argmax(probs) == argmax(y_true); tf.argmax takes the
// FIRST maximum): one pass, one count - the reference's four reductions (two argmax, a comparison, a mean) were four framework
// launches over intermediate tensors.  acc[0] = matching pixels / M (fixed-order block sums: reproducible); acc needs
// USSEG_ACC_FLOATS floats (grid_ordered_sum slots).
__global__ __launch_bounds__(256) void accuracy_kernel(const float* probs, const float* y, int64_t M, int C, float inv_m, float* acc) {
  __shared__ float red[4];
  float cnt = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (C == 3) {      // the reference's three classes: four pixels per thread and trip, all 24 loads issued before the first compare
    for (; m + 3 * stride < M; m += 4 * stride) {
      float pv[4][3], tv[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 3; ++c) { pv[u][c] = probs[(m + u * stride) * 3 + c]; tv[u][c] = y[(m + u * stride) * 3 + c]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ip = pv[u][1] > pv[u][0] ? (pv[u][2] > pv[u][1] ? 2 : 1) : (pv[u][2] > pv[u][0] ? 2 : 0);   // FIRST maximum, as the loop below
        const int it = tv[u][1] > tv[u][0] ? (tv[u][2] > tv[u][1] ? 2 : 1) : (tv[u][2] > tv[u][0] ? 2 : 0);
        cnt += ip == it ? 1.f : 0.f;
      }
    }
  }
  for (; m < M; m += stride) {
    const float* p = probs + m * C;
    const float* t = y + m * C;
    int ip = 0, it = 0;
    float bp = p[0], bt = t[0];
    for (int c = 1; c < C; ++c) {
      const float vp = p[c], vt = t[c];
      if (vp > bp) { bp = vp; ip = c; }
      if (vt > bt) { bt = vt; it = c; }
    }
    cnt += ip == it ? 1.f : 0.f;
  }
  for (int msk = 32; msk >= 1; msk >>= 1) cnt += __shfl_xor(cnt, msk, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  grid_ordered_sum(((red[0] + red[1]) + (red[2] + red[3])) * inv_m, acc, gridDim.x);
}
extern "C" int usseg_accuracy(const float* probs, const float* y_true, int64_t M, int32_t C, float* acc, usseg_stream_t stream) {
  USSEG_CHECK_ARG(probs && y_true && acc && C >= 1 && C <= 64, "accuracy: bad arguments");
  if (M <= 0) return USSEG_OK;
  int64_t g = cdiv64(M, 256 * 4);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(accuracy_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, probs, y_true, M, (int)C, 1.f / (float)M, acc);
  return usseg_check_launch("accuracy");
}

__global__ __launch_bounds__(256) void loss_cat_scale_kernel(const float* y, int B, int HW, int C, float* scale) {
  int64_t total = (int64_t)HW * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += y[(int64_t)b * total + i];
    scale[i] = 1.f / (s + 1.f) / (float)HW;
  }
}
extern "C" int usseg_loss_cat_scale(const float* y_true, int32_t B, int32_t HW, int32_t C, float* scale, usseg_stream_t stream) {
  USSEG_CHECK_ARG(y_true && scale && B > 0 && HW > 0 && C > 0, "loss_cat_scale: bad args");
  int64_t g = cdiv64((int64_t)HW * C, 256);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(loss_cat_scale_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, y_true, B, HW, C, scale);
  return usseg_check_launch("loss_cat_scale");
}

// out[0] = sum of n floats, workgroup partials added in workgroup order (grid_ordered_sum): the scalar of a loss MAP (TBI_ResNest.py:234-248
// returns [H,W]; MainParallel.py:131-134 reduces a scalar) without a framework reduction on the step path
__global__ __launch_bounds__(256) void sum_f32_kernel(const float* x, int64_t n, float* out) {
  __shared__ float red[4];
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float s = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const float a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
    s += a; s1 += b; s2 += c; s3 += d;
  }
  for (; i < n; i += stride) s += x[i];
  s = (s + s1) + (s2 + s3);
  for (int msk = 32; msk >= 1; msk >>= 1) s += __shfl_xor(s, msk, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  grid_ordered_sum((red[0] + red[1]) + (red[2] + red[3]), out, gridDim.x);
}
extern "C" int usseg_sum_f32(const float* x, int64_t n, float* out, usseg_stream_t stream) {
  USSEG_CHECK_ARG(x && out && n >= 0, "sum_f32: bad arguments");
  int64_t g = cdiv64(n > 0 ? n : 1, 256 * 4);
  if (g > 256) g = 256;
  hipLaunchKernelGGL(sum_f32_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, n, out);
  return usseg_check_launch("sum_f32");
}

// ------------------------------------------------------------------------------------------ optimiser
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, int64_t n, float* out, int32_t* step, float* lr_t, float lr, float b1, float b2) {
  __shared__ float red[4];
  float s = 0.f;
  int64_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (; i + 3 * stride < n4; i += 4 * stride) {   // four loads in flight per thread
    float4 a = g4[i], b = g4[i + stride], c = g4[i + 2 * stride], d = g4[i + 3 * stride];
    s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    s1 += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
    s2 += c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
    s3 += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
  }
  for (; i < n4; i += stride) {
    float4 v = g4[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  s += s1 + s2 + s3;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
  for (int msk = 32; msk >= 1; msk >>= 1) s += __shfl_xor(s, msk, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const bool last = grid_ordered_sum((red[0] + red[1]) + (red[2] + red[3]), out, gridDim.x);   // reproducible: the clip factor scales every update
  if (last && step && threadIdx.x == 0) {   // usseg_sumsq_advance: the optimiser's step counter moves on in the same launch
    const int t = *step + 1;
    *step = t;
    *lr_t = lr * sqrtf(1.f - powf(b2, (float)t)) / (1.f - powf(b1, (float)t));
  }
}
static int sumsq_launch(const float* g, int64_t n, float* out, int32_t* step, float* lr_t, float lr, float b1, float b2, usseg_stream_t stream) {
  USSEG_CHECK_ARG(g && out && n > 0 && ((uintptr_t)g % 16) == 0, "sumsq: bad args (g must be 16-byte aligned, n > 0)");   // out[0] is OVERWRITTEN
  int64_t grid = cdiv64(n, 256 * 16);
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, g, n, out, step, lr_t, lr, b1, b2);
  return usseg_check_launch("sumsq");
}
extern "C" int usseg_sumsq(const float* g, int64_t n, float* out, usseg_stream_t stream) {
  return sumsq_launch(g, n, out, nullptr, nullptr, 0.f, 0.f, 0.f, stream);
}
extern "C" int usseg_sumsq_advance(const float* g, int64_t n, float* out, int32_t* step, float* lr_t_dev, float lr, float beta1, float beta2,
                                   usseg_stream_t stream) {
  USSEG_CHECK_ARG(step && lr_t_dev, "sumsq_advance: null pointer");
  return sumsq_launch(g, n, out, step, lr_t_dev, lr, beta1, beta2, stream);
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float clip_norm,
                                                    const float* lr_t_dev, float b1, float b2, float eps) {
  float scale = 1.f;
  if (clip_norm > 0.f) scale = clip_norm / fmaxf(sqrtf(*sumsq), clip_norm);  // tf.clip_by_global_norm
  const float lr_t = *lr_t_dev;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * scale;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}
extern "C" int usseg_adam_clip_step(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float clip_norm,
                                    const float* lr_t_dev, float beta1, float beta2, float eps, usseg_stream_t stream) {
  USSEG_CHECK_ARG(p && g && m && v && lr_t_dev && (clip_norm <= 0.f || sumsq), "adam: null pointer");
  if (n <= 0) return USSEG_OK;
  int64_t grid = cdiv64(n, 256 * 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, sumsq, clip_norm, lr_t_dev, beta1,
                     beta2, eps);
  return usseg_check_launch("adam");
}

__global__ void adam_advance_kernel(int32_t* step, float* lr_t, float lr, float b1, float b2) {
  int t = *step + 1;
  *step = t;
  *lr_t = lr * sqrtf(1.f - powf(b2, (float)t)) / (1.f - powf(b1, (float)t));
}
extern "C" int usseg_adam_advance(int32_t* step, float* lr_t_dev, float lr, float beta1, float beta2, usseg_stream_t stream) {
  USSEG_CHECK_ARG(step && lr_t_dev, "adam_advance: null pointer");
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, lr_t_dev, lr, beta1, beta2);
  return usseg_check_launch("adam_advance");
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}
extern "C" int usseg_fill_f32(float* p, int64_t n, float value, usseg_stream_t stream) {
  USSEG_CHECK_ARG(p || n == 0, "fill: null pointer");
  if (n <= 0) return USSEG_OK;
  int64_t grid = cdiv64(n, 256 * 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, n, value);
  return usseg_check_launch("fill");
}

__global__ __launch_bounds__(256) void scale_kernel(float* p, int64_t n, const float* sumsq, float clip_norm) {
  float scale = clip_norm / fmaxf(sqrtf(*sumsq), clip_norm);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] *= scale;
}
extern "C" int usseg_scale_f32(float* p, int64_t n, const float* sumsq, float clip_norm, usseg_stream_t stream) {
  USSEG_CHECK_ARG(p && sumsq && clip_norm > 0.f, "scale: bad args");
  if (n <= 0) return USSEG_OK;
  int64_t grid = cdiv64(n, 256 * 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, n, sumsq, clip_norm);
  return usseg_check_launch("scale");
}

// ------------------------------------------------------------------------------------------ ViT attention helpers
// One wave per row (n <= 2048): softmax(scale * s) with fp32 statistics; writes the fp32 weights and a bf16 copy.
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const float* s, int64_t rows, int n, float scale, float* p32, bf16_t* pbf) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sr = s + row * n;
  float mx = -INFINITY;
  for (int c = lane; c < n; c += 64) mx = fmaxf(mx, sr[c] * scale);
  for (int msk = 32; msk >= 1; msk >>= 1) mx = fmaxf(mx, __shfl_xor(mx, msk, 64));
  float sum = 0.f;
  for (int c = lane; c < n; c += 64) sum += __expf(sr[c] * scale - mx);
  for (int msk = 32; msk >= 1; msk >>= 1) sum += __shfl_xor(sum, msk, 64);
  const float inv = 1.f / sum;
  for (int c = lane; c < n; c += 64) {
    float v = __expf(sr[c] * scale - mx) * inv;
    p32[row * n + c] = v;
    pbf[row * n + c] = f2bf(v);
  }
}
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* p32, const float* dp, int64_t rows, int n, float scale, bf16_t* ds) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* pr = p32 + row * n;
  const float* dr = dp + row * n;
  float dot = 0.f;
  for (int c = lane; c < n; c += 64) dot += pr[c] * dr[c];
  for (int msk = 32; msk >= 1; msk >>= 1) dot += __shfl_xor(dot, msk, 64);
  for (int c = lane; c < n; c += 64) ds[row * n + c] = f2bf(scale * pr[c] * (dr[c] - dot));
}
extern "C" int usseg_softmax_rows_fwd(const float* s, int64_t rows, int32_t n, float scale, float* p32, void* pbf, usseg_stream_t stream) {
  USSEG_CHECK_ARG(s && p32 && pbf && rows > 0 && n > 0, "softmax_rows_fwd: bad args");
  hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, s, rows, n, scale, p32, (bf16_t*)pbf);
  return usseg_check_launch("softmax_rows_fwd");
}
extern "C" int usseg_softmax_rows_bwd(const float* p32, const float* dp, int64_t rows, int32_t n, float scale, void* ds_bf, usseg_stream_t stream) {
  USSEG_CHECK_ARG(p32 && dp && ds_bf && rows > 0 && n > 0, "softmax_rows_bwd: bad args");
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, p32, dp, rows, n, scale, (bf16_t*)ds_bf);
  return usseg_check_launch("softmax_rows_bwd");
}

// 32x32 tiles through LDS: dst[b][c][r] = src[b][r][c]
__global__ __launch_bounds__(256) void transpose_batched_kernel(const bf16_t* src, int R, int C, int lds_, int nb2, int64_t ss1, int64_t ss2, bf16_t* dst) {
  __shared__ bf16_t tile[32][33];
  const int bz = blockIdx.z, b1 = bz / nb2, b2 = bz - b1 * nb2;
  const bf16_t* s = src + b1 * ss1 + b2 * ss2;
  bf16_t* d = dst + (int64_t)bz * R * C;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < R && c0 + tx < C) tile[i][tx] = s[(int64_t)(r0 + i) * lds_ + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < C && r0 + tx < R) d[(int64_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}
extern "C" int usseg_transpose_batched(const void* src, int32_t R, int32_t C, int32_t lds_, int32_t nb1, int32_t nb2, int64_t ss1, int64_t ss2,
                                       void* dst, usseg_stream_t stream) {
  USSEG_CHECK_ARG(src && dst && R > 0 && C > 0 && nb1 > 0 && nb2 > 0 && (int64_t)nb1 * nb2 < 65536, "transpose_batched: bad args");
  hipLaunchKernelGGL(transpose_batched_kernel, dim3((C + 31) / 32, (R + 31) / 32, nb1 * nb2), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, R, C, lds_, nb2, ss1, ss2, (bf16_t*)dst);
  return usseg_check_launch("transpose_batched");
}

__global__ __launch_bounds__(256) void cast_f32_bf16_batched_kernel(const float* src, int R, int C, int nb2, bf16_t* dst, int ldd, int64_t ds1,
                                                                     int64_t ds2, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c = (int)(i % C);
    int64_t t = i / C;
    int r = (int)(t % R);
    int bz = (int)(t / R);
    int b1 = bz / nb2, b2 = bz - b1 * nb2;
    dst[b1 * ds1 + b2 * ds2 + (int64_t)r * ldd + c] = f2bf(src[i]);
  }
}
extern "C" int usseg_cast_f32_to_bf16_batched(const float* src, int32_t R, int32_t C, int32_t nb1, int32_t nb2, void* dst, int32_t ldd, int64_t ds1,
                                              int64_t ds2, usseg_stream_t stream) {
  USSEG_CHECK_ARG(src && dst && R > 0 && C > 0 && nb1 > 0 && nb2 > 0, "cast_f32_to_bf16_batched: bad args");
  int64_t total = (int64_t)nb1 * nb2 * R * C;
  int64_t g = cdiv64(total, 256 * 4);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(cast_f32_bf16_batched_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src, R, C, nb2, (bf16_t*)dst, ldd, ds1,
                     ds2, total);
  return usseg_check_launch("cast_f32_to_bf16_batched");
}
