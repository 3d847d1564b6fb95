// 3x3 convolution (stride 1, SAME, any dilation) with an LDS-resident input halo tile, bf16 MFMA, for gfx950.
//
// Why a second conv kernel: the gather implicit GEMM (conv_igemm.hip) re-stages the 128-pixel activation tile for
// every tap, i.e. 9 x per channel chunk.  Here a workgroup stages its pixel patch PLUS a one-pixel halo ONCE per
// 32-channel chunk and all 9 taps read their shifted A fragments straight from that LDS image (per-lane addresses),
// so global->LDS traffic per MFMA drops ~5x.
//
// Dilation d is handled by decomposition: a dilated 3x3 conv is d*d independent DENSE 3x3 convs on the sub-lattices
// {(a + d*i, b + d*j)}.  Every (image, a, b) is a "virtual image" of (H/d) x (W/d) pixels read with pixel stride d,
// so the halo is always one lattice pixel wide, whatever the dilation.
//
// Workgroup = 128 output pixels (4 waves x 2 MFMA pixel-strips of 16) x 16*NT output channels.  The 128 pixels are
// NV patches of PH x PW lattice pixels (PW = min(16, W/d)); small virtual images (8x8, 4x4) are packed several to a
// workgroup.  Operands: x NHWC bf16, packed weights Wp[N][9*Cin] as for the gather kernel.  The same kernel runs the
// backward-data pass (flip = 1: tap position (kh,kw) uses weight tap 8 - (3*kh+kw), operand packed with in/out swapped).
// Pipeline: single LDS image, next chunk prefetched into registers during the MFMAs (2 workgroups per CU overlap).
#include "common.h"

struct HaloParams {
  const bf16_t* x;
  const bf16_t* w;
  void* y;
  const float* bias;
  const bf16_t* res;
  int32_t B, H, W, d;
  int32_t Hl, Wl;        // lattice size = H/d, W/d
  int32_t PH, PW, NV;    // patch size, patches per workgroup
  int32_t tiles_x, tiles_per_v, npatches;
  int32_t ldx, ldy, ldr;
  int32_t Cin, nchunks;  // physical input channels (multiple of 8), ceil(Cin/32)
  int32_t Nw, Kw, Nout;
  int32_t act;
  float alpha;
  int32_t out_f32, accumulate, flip;
};

template <int NT>
__global__ __launch_bounds__(256) void conv_halo_kernel(const HaloParams p) {
  constexpr int BN = 16 * NT, LDSS = 40;
  constexpr int MAXHP = 288;                    // halo pixels per workgroup (8 patches of 6x6 is the maximum)
  constexpr int A_IT = (MAXHP * 4 + 255) / 256;  // 5
  constexpr int W_IT = (9 * BN * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) bf16_t lds_a[MAXHP * LDSS];
  __shared__ __attribute__((aligned(16))) bf16_t lds_w[9 * BN * LDSS];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n0 = blockIdx.y * BN;
  const int HW2 = p.PW + 2, HPP = (p.PH + 2) * HW2;  // halo row length, halo pixels per patch
  const int NHP = p.NV * HPP;

  // ---- staging assignments (fixed for the whole kernel): A items = (halo pixel, 16-byte chunk q)
  int a_off[A_IT];  // element offset of the pixel in x, or -1
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    int idx = tid + 256 * it;
    int hp = idx >> 2;
    int off = -1;
    if (hp < NHP) {
      int pi = hp / HPP, rem = hp - pi * HPP;
      int hy = rem / HW2, hx = rem - hy * HW2;
      int gp = blockIdx.x * p.NV + pi;
      if (gp < p.npatches) {
        int v = gp / p.tiles_per_v, tt = gp - v * p.tiles_per_v;
        int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
        int dd = p.d * p.d;
        int b = v / dd, ab = v - b * dd;
        int la = ab / p.d, lb = ab - la * p.d;
        int ly = ty * p.PH + hy - 1, lx = tx * p.PW + hx - 1;
        if ((unsigned)ly < (unsigned)p.Hl && (unsigned)lx < (unsigned)p.Wl)
          off = ((b * p.H + la + p.d * ly) * p.W + lb + p.d * lx) * p.ldx;
      }
    }
    a_off[it] = off;
  }
  const int q = tid & 3;

  uint4 ra[A_IT], rw[W_IT];
  auto load_chunk = [&](int ck) {
    const int c = ck * 32 + q * 8;
    const bool cok = c < p.Cin;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (cok && a_off[it] >= 0) v = *reinterpret_cast<const uint4*>(p.x + a_off[it] + c);
      ra[it] = v;
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      int idx = tid + 256 * it;
      int row = idx >> 2;  // t*BN + n
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row < 9 * BN) {
        int t = row / BN, n = row - t * BN;
        int tw = p.flip ? 8 - t : t;
        if (cok && (n0 + n) < p.Nw) v = *reinterpret_cast<const uint4*>(p.w + (int64_t)(n0 + n) * p.Kw + tw * p.Cin + c);
      }
      rw[it] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      int idx = tid + 256 * it;
      if ((idx >> 2) < NHP) *reinterpret_cast<uint4*>(&lds_a[(idx >> 2) * LDSS + q * 8]) = ra[it];
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      int idx = tid + 256 * it;
      if ((idx >> 2) < 9 * BN) *reinterpret_cast<uint4*>(&lds_w[(idx >> 2) * LDSS + q * 8]) = rw[it];
    }
  };

  // ---- this lane's two output pixels (one per strip): LDS halo index of the top-left tap, and the image pixel
  const int pl = lane & 15, fk = (lane >> 4) * 8;
  const int rows_per_strip = 16 / p.PW;  // PW in {4, 8, 16}
  const int spp = (p.PH * p.PW) >> 4;    // strips per patch
  int a_base[2];
  int64_t opix[2];
  bool ovalid[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    int s = wv * 2 + s2;
    int pi = s / spp, sl = s - pi * spp;
    int r = pl / p.PW, c = pl - r * p.PW;
    int row = sl * rows_per_strip + r;
    a_base[s2] = (pi * HPP + row * HW2 + c) * LDSS + fk;
    int gp = blockIdx.x * p.NV + pi;
    ovalid[s2] = gp < p.npatches;
    int gpc = ovalid[s2] ? gp : 0;
    int v = gpc / p.tiles_per_v, tt = gpc - v * p.tiles_per_v;
    int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
    int dd = p.d * p.d;
    int b = v / dd, ab = v - b * dd;
    int la = ab / p.d, lb = ab - la * p.d;
    int iy = la + p.d * (ty * p.PH + row), ix = lb + p.d * (tx * p.PW + c);
    opix[s2] = ((int64_t)b * p.H + iy) * p.W + ix;
  }
  const int w_base = (lane & 15) * LDSS + fk;

  f32x4_t acc[2][NT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  load_chunk(0);
  for (int ck = 0; ck < p.nchunks; ++ck) {
    store_chunk();
    __syncthreads();
    if (ck + 1 < p.nchunks) load_chunk(ck + 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int kh = t / 3, kw = t - 3 * kh;
      const int toff = (kh * HW2 + kw) * LDSS;
      bf16x8_t xf[2], wf[NT];
#pragma unroll
      for (int a = 0; a < 2; ++a) xf[a] = *reinterpret_cast<const bf16x8_t*>(&lds_a[a_base[a] + toff]);
#pragma unroll
      for (int b = 0; b < NT; ++b) wf[b] = *reinterpret_cast<const bf16x8_t*>(&lds_w[(t * BN + b * 16) * LDSS + w_base]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue (same contract as the gather kernel): lane holds Y[its pixel][n = 4*(lane>>4) + j] per n-tile
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    if (!ovalid[a]) continue;
#pragma unroll
    for (int bt = 0; bt < NT; ++bt) {
      int n = n0 + bt * 16 + (lane >> 4) * 4;
      if (n >= p.Nout) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[a][bt][j];
      if (p.bias) {
        float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
      }
      if (p.act != USSEG_ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], p.act, p.alpha);
      }
      if (p.res) {
        uint2 rr = *reinterpret_cast<const uint2*>(p.res + opix[a] * p.ldr + n);
        v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
        v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
      }
      if (p.out_f32) {
        float* dst = reinterpret_cast<float*>(p.y) + opix[a] * p.ldy + n;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < p.Nout) dst[j] = p.accumulate ? dst[j] + v[j] : v[j];
      } else {
        bf16_t* dst = reinterpret_cast<bf16_t*>(p.y) + opix[a] * p.ldy + n;
        if (p.accumulate) {
          uint2 o = *reinterpret_cast<const uint2*>(dst);
          v[0] += __uint_as_float(o.x << 16); v[1] += __uint_as_float(o.x & 0xffff0000u);
          v[2] += __uint_as_float(o.y << 16); v[3] += __uint_as_float(o.y & 0xffff0000u);
        }
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(dst) = o;
      }
    }
  }
}

// Returns 1 and launches if the geometry fits the halo kernel, 0 if the caller must use the gather kernel.
int usseg_try_launch_conv_halo(const bf16_t* x, const bf16_t* w, void* y, const float* bias, const bf16_t* res, int B, int H, int W, int d,
                               int Cin, int ldx, int Nout, int ldy, int ldr, int Nw, int Kw, int act, float alpha, int out_f32,
                               int accumulate, int flip, hipStream_t s) {
  static const int disabled = getenv("USSEG_NO_HALO") != nullptr;
  if (disabled) return 0;
  if (d < 1 || H % d || W % d) return 0;
  const int Hl = H / d, Wl = W / d;
  int PW;
  if (Wl % 16 == 0) PW = 16;
  else if (Wl == 8 || Wl == 4) PW = Wl;
  else return 0;
  int PH = 128 / PW;
  if (PH > Hl) PH = Hl;
  if (Hl % PH || (PH * PW) % 16) return 0;
  const int NV = 128 / (PH * PW);
  if (NV * (PH + 2) * (PW + 2) > 288) return 0;
  if ((int64_t)B * H * W * ldx >= (1ll << 31)) return 0;
  HaloParams p = {};
  p.x = x; p.w = w; p.y = y; p.bias = bias; p.res = res;
  p.B = B; p.H = H; p.W = W; p.d = d; p.Hl = Hl; p.Wl = Wl; p.PH = PH; p.PW = PW; p.NV = NV;
  p.tiles_x = Wl / PW;
  p.tiles_per_v = (Hl / PH) * p.tiles_x;
  p.npatches = B * d * d * p.tiles_per_v;
  p.ldx = ldx; p.ldy = ldy; p.ldr = ldr; p.Cin = Cin; p.nchunks = (Cin + 31) / 32;
  p.Nw = Nw; p.Kw = Kw; p.Nout = Nout; p.act = act; p.alpha = alpha; p.out_f32 = out_f32; p.accumulate = accumulate; p.flip = flip;
  const int gx = (p.npatches + NV - 1) / NV;
  // pick the channel tile: the widest that still gives the chip a few hundred workgroups
  int nt = Nout <= 16 ? 1 : (Nout <= 32 ? 2 : 4);
  while (nt > 1 && (int64_t)gx * ((Nout + 16 * nt - 1) / (16 * nt)) < 256 && Nout > 16 * (nt / 2)) nt >>= 1;
  const int gy = (Nout + 16 * nt - 1) / (16 * nt);
  const int slot = usseg_prof_start(1, s);
  if (nt == 1) hipLaunchKernelGGL(conv_halo_kernel<1>, dim3(gx, gy), dim3(256), 0, s, p);
  else if (nt == 2) hipLaunchKernelGGL(conv_halo_kernel<2>, dim3(gx, gy), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(conv_halo_kernel<4>, dim3(gx, gy), dim3(256), 0, s, p);
  usseg_prof_stop(1, slot, s);
  return 1;
}
