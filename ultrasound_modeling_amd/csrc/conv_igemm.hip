// Gather implicit-GEMM on bf16 MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate, for gfx950.
//
// One kernel serves Conv2D fwd, Conv2D dgrad, Conv2DTranspose fwd (4 parity classes) and Conv2DTranspose dgrad:
//   Y[b, gy*osy+oay, gx*osx+oax, n] = epi( sum_{tap i} sum_{c<Cin} X[b, gy*isy+dy_i, gx*isx+dx_i, c] * Wp[n][w_i*Cin + c] )
// GEMM view: rows m = (b,gy,gx) pixels, cols n = output channels, K = (tap, channel) walked in 16-byte
// (8-channel) chunks, so any Cin that is a multiple of 8 works and taps that fall outside the image are
// zero-filled while staging (the im2col matrix never exists in memory).
//
// Tile: 128 pixels x (16*NT) channels per 256-thread workgroup, K step 32.  Each of the 4 waves owns 32 pixels
// x all 16*NT channels.  Operands are staged global -> VGPR -> LDS (80-byte padded rows), double buffered, with
// the next step's global loads issued before the current step's MFMAs (one barrier per K step).
// The MFMA is issued as D^T = W * X^T so that each lane ends up with 4 CONSECUTIVE channels of one pixel:
// the epilogue (bias, activation, residual) then stores 8 bytes per lane and 32 bytes per pixel per tile.
#include "common.h"

struct IgemmParams {
  const bf16_t* x;
  const bf16_t* w;
  void* y;
  const float* bias;
  const float* scale;   // optional per-channel multiplier applied before the bias (folded inference BatchNorm)
  const bf16_t* res;
  int32_t B, Hg, Wg;
  int64_t M;
  int32_t Hi, Wi, ldx, isy, isx;
  int32_t Ho, Wo, ldy, osy, osx, oay, oax;
  int32_t ldr;
  int32_t cpt;      // 16-byte chunks per tap = Cin/8
  int32_t ntaps;
  int32_t nchunks;  // ntaps * cpt
  int32_t Nw;       // rows of the packed weight matrix
  int32_t Kw;       // its row stride (elements)
  int32_t Nout;     // channels to store
  int32_t act;
  float alpha;
  int32_t out_f32;
  int32_t accumulate;
  int16_t tap_dy[32], tap_dx[32], tap_w[32], tap_c[32];   // per tap: source offset, weight tap index, source channel base
  // Conv2DTranspose forward: the four output-parity classes run as blockIdx.z of ONE launch; class c owns the tap-table
  // entries [4c, 4c+4) (cls_ntaps[c] of them) and writes output pixels (2*gy + (c>>1), 2*gx + (c&1)).
  int32_t cls_mode;
  int16_t cls_ntaps[4];
  // batched GEMM (attention): blockIdx.z = b1*nb2 + b2 selects operand bases x + b1*xs1 + b2*xs2 etc. (elements)
  int32_t nb2, nbatch;
  // LDS-DMA kernels: workgroups go to the 8 XCDs round-robin in dispatch order (x = pixel tile fastest), so the channel tiles / parity classes
  // of ONE pixel tile - which stage the same activation rows - ran in different XCDs and each pulled those rows through HBM / MALL again (the
  // dense layers of the Swin / ViT configurations: 2-4 channel tiles per token tile).  Remapped, XCD c owns a contiguous run of the
  // (pixel tile, channel tile, class) items in class- and channel-fastest order (see WgradParams::xcd_remap).
  int32_t xcd_remap;
  int64_t xs1, xs2, ws1, ws2, ys1, ys2;
};

// KC = 16-byte K chunks per step: 4 (K step 32, one MFMA k-step per barrier) or 8 (K step 64: half the barriers and twice
// the bytes in flight per thread; used when the K loop is long).
template <int NT, int KC>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  constexpr int BM = 128, BN = 16 * NT, LDSS = KC * 8 + 8;  // LDS row stride in elements (data + 16 B pad)
  constexpr int WCH = (BN * KC + 255) / 256;         // weight chunks per thread per K step
  constexpr int AH = BM * KC / 256, RSTEP = 256 / KC; // A rows per thread and their spacing
  constexpr int BUF = (BM + BN) * LDSS;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds_dyn[];   // [2][BUF] operand stages, then the tap table
  bf16_t* const lds0 = lds_dyn;
  int* const s_tap = reinterpret_cast<int*>(lds_dyn + 2 * BUF);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  if (tid < 32) {
    s_tap[tid] = p.tap_dy[tid];
    s_tap[32 + tid] = p.tap_dx[tid];
    s_tap[64 + tid] = p.tap_w[tid];
    s_tap[96 + tid] = p.tap_c[tid];
  }
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int HWg = p.Hg * p.Wg;
  const int cls = p.cls_mode ? (int)blockIdx.z : 0;
  const int tbase = cls * 4;
  const int ntaps = p.cls_mode ? (int)p.cls_ntaps[cls] : p.ntaps;
  const int oay = p.cls_mode ? (cls >> 1) : p.oay, oax = p.cls_mode ? (cls & 1) : p.oax;
  const int bz = (!p.cls_mode && p.nb2 > 0) ? (int)blockIdx.z : 0;
  const int bz1 = p.nb2 > 0 ? bz / p.nb2 : 0, bz2 = p.nb2 > 0 ? bz - bz1 * p.nb2 : 0;
  const bf16_t* const xbase = p.x + bz1 * p.xs1 + bz2 * p.xs2;
  const bf16_t* const wbase = p.w + bz1 * p.ws1 + bz2 * p.ws2;
  const int64_t ybatch = bz1 * p.ys1 + bz2 * p.ys2;

  // --- per-thread staging coordinates: chunk column q (0..KC-1) of rows r0 + RSTEP*h
  const int q = tid & (KC - 1);
  const int r0 = tid / KC;
  int py[AH], px[AH], pb[AH];
  bool pv[AH];
#pragma unroll
  for (int h = 0; h < AH; ++h) {
    int64_t m = m0 + r0 + RSTEP * h;
    pv[h] = m < p.M;
    int mm = pv[h] ? (int)m : 0;
    int b = mm / HWg;
    int rem = mm - b * HWg;
    int gy = rem / p.Wg;
    int gx = rem - gy * p.Wg;
    pb[h] = b * p.Hi;
    py[h] = gy * p.isy;
    px[h] = gx * p.isx;
  }
  // (tap, chunk-in-tap) of this thread's chunk column, advanced by KC chunks per K step
  int ti = q / p.cpt;
  int c8 = q - ti * p.cpt;
  const int nks = (ntaps * p.cpt + KC - 1) / KC;
  const int Cin = p.cpt * 8;

  uint4 ra[AH], rw[WCH];
  __syncthreads();  // tap table visible

  auto load_step = [&]() {
    const bool tv = ti < ntaps;
    int dy = 0, dx = 0, tw = 0, tc = 0;
    if (tv) { dy = s_tap[tbase + ti]; dx = s_tap[32 + tbase + ti]; tw = s_tap[64 + tbase + ti]; tc = s_tap[96 + tbase + ti]; }
#pragma unroll
    for (int h = 0; h < AH; ++h) {
      uint4 v = make_uint4(0, 0, 0, 0);
      int iy = py[h] + dy, ix = px[h] + dx;
      if (tv && pv[h] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) {
        const bf16_t* src = xbase + ((int64_t)(pb[h] + iy) * p.Wi + ix) * p.ldx + tc + c8 * 8;
        v = *reinterpret_cast<const uint4*>(src);
      }
      ra[h] = v;
    }
#pragma unroll
    for (int j = 0; j < WCH; ++j) {
      int idx = tid + 256 * j;
      int n = idx / KC;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (idx < BN * KC && tv && (n0 + n) < p.Nw) {
        const bf16_t* src = wbase + (int64_t)(n0 + n) * p.Kw + tw * Cin + c8 * 8;
        v = *reinterpret_cast<const uint4*>(src);
      }
      rw[j] = v;
    }
    // advance to the next K step
    c8 += KC;
    while (c8 >= p.cpt) { c8 -= p.cpt; ++ti; }
  };
  auto store_step = [&](int buf) {
    bf16_t* const L = lds0 + buf * BUF;
#pragma unroll
    for (int h = 0; h < AH; ++h)
      *reinterpret_cast<uint4*>(&L[(r0 + RSTEP * h) * LDSS + q * 8]) = ra[h];
#pragma unroll
    for (int j = 0; j < WCH; ++j) {
      int idx = tid + 256 * j;
      if (idx < BN * KC) *reinterpret_cast<uint4*>(&L[(BM + idx / KC) * LDSS + q * 8]) = rw[j];
    }
  };

  f32x4_t acc[2][NT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  load_step();
  store_step(0);
  __syncthreads();

  const int frow = lane & 15, fk = (lane >> 4) * 8;
  for (int ks = 0; ks < nks; ++ks) {
    const int cur = ks & 1;
    const bool more = (ks + 1) < nks;
    if (more) load_step();
    const bf16_t* const L = lds0 + cur * BUF;
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      bf16x8_t xf[2], wf[NT];
#pragma unroll
      for (int a = 0; a < 2; ++a)
        xf[a] = *reinterpret_cast<const bf16x8_t*>(&L[(wv * 32 + a * 16 + frow) * LDSS + fk + 32 * kk]);
#pragma unroll
      for (int b = 0; b < NT; ++b)
        wf[b] = *reinterpret_cast<const bf16x8_t*>(&L[(BM + b * 16 + frow) * LDSS + fk + 32 * kk]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
    }
    if (more) store_step(cur ^ 1);
    __syncthreads();
  }

  // --- epilogue: lane holds Y[pixel = lane&15][n = 4*(lane>>4) + j]
  int64_t opix[2];
  bool ovalid[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    int64_t m = m0 + wv * 32 + a * 16 + frow;
    ovalid[a] = m < p.M;
    int mm = ovalid[a] ? (int)m : 0;
    int b = mm / HWg;
    int rem = mm - b * HWg;
    int gy = rem / p.Wg;
    int gx = rem - gy * p.Wg;
    opix[a] = ((int64_t)(b * p.Ho + gy * p.osy + oay)) * p.Wo + gx * p.osx + oax;
  }
  const EpiArgs e = {p.scale, p.bias, p.res, p.y, p.ldy, p.ldr, p.Nout, p.act, p.alpha, p.out_f32, p.accumulate};
  const int nbase = n0 + (lane >> 4) * 4;
  EpiConst<NT> ec;
  epi_const_load<NT>(ec, p.bias, nbase, p.Nout);
  conv_epilogue<2, NT>(e, ec, acc, opix, ovalid, nbase, ybatch);
}

// ---- LDS-DMA variant (everything but the batched attention GEMMs) -------------------------------------------------------
// Same tiling and tap tables, but the operand pieces go global -> LDS directly (buffer_load ... lds: no staging registers,
// no ds_write pass, out-of-image / out-of-range pieces are zero-filled by the buffer range check) into a ring of NSTAGE
// stages, so the pieces of K step ks+NSTAGE-1 are in flight under the MFMAs of step ks, and the only wait of a step is
// `vmcnt(pieces of the younger steps)`.  The register-staged kernel keeps one step in flight and then waits for it
// (measured on the fused branch backward-data: 1.5k cycles issuing + 0.7k waiting per step around 0.6k of MFMA).
// LDS image of a stage: [128 + max(BN,32) rows][8 chunks of 16 B], dense 128-B rows; slot j of row R holds chunk
// j ^ ((R >> 1) & 7): a fragment read (16 consecutive rows, one chunk) then touches every bank exactly once, and a DMA
// wave-instruction (8 rows x 8 slots, lane -> row lane/8, slot lane%8) keeps a fixed chunk per lane.
typedef __attribute__((address_space(3))) void* igemm_lds_ptr_t;
#define IGEMM_OOB 0x80000000u
// The tap-table entry of a K step, read by inline assembly.  As a C++ load it is an LDS read that follows LDS-DMA writes the
// compiler cannot tell apart from it, and SIInsertWaitcnts then puts `s_waitcnt vmcnt(0)` in front of it: every K step waited
// for ALL stages in flight before issuing the next one - the ring never held more than one stage in flight, whatever NSTAGE
// (the table lives in its own static array; the DMA only ever writes the dynamic stage ring).  Worth 3-5 % on the long-K launches.
typedef int igemm_i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ igemm_i32x4_t igemm_tap_read(const int* entry) {
  igemm_i32x4_t e;
  const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const int*)entry;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(addr) : "memory");
  return e;
}

template <int NT, int NSTAGE>
__global__ __launch_bounds__(256, 2) void igemm_dma_kernel(const IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, BN = 16 * NT, BNP = BN < 32 ? 32 : BN;   // weight rows padded so every wave issues W_IT pieces
  constexpr int STAGE = (BM + BNP) * 128;                           // bytes
  constexpr int A_IT = BM / 32, W_IT = BNP / 32;                    // 8-row pieces per wave per step
  constexpr int NI = A_IT + W_IT;                                   // DMA instructions per wave per step
  constexpr int D = NSTAGE - 1;                                     // prefetch distance
  extern __shared__ __attribute__((aligned(1024))) char lds_raw[];
  // per tap: source pixel shift and the element offsets (activation, weight) its chunks start from - decoded once per
  // workgroup; a K step then costs one 16-byte table read per lane and no integer multiplies
  __shared__ __attribute__((aligned(16))) int s_tap4[32][4];   // dy, dx, da, dw

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 32) {
    const int dy = p.tap_dy[tid], dx = p.tap_dx[tid];
    s_tap4[tid][0] = dy;
    s_tap4[tid][1] = dx;
    s_tap4[tid][2] = (dy * p.Wi + dx) * p.ldx + p.tap_c[tid];
    s_tap4[tid][3] = p.tap_w[tid] * p.cpt * 8;
  }
  int bx = blockIdx.x, by = blockIdx.y, bzz = blockIdx.z;
  if (p.xcd_remap) {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, total = gx * gy * gz;
    const int L = bx + gx * (by + gy * bzz);
    const int c = L & 7, q8 = total >> 3, r8 = total & 7;
    const int item = c * q8 + (c < r8 ? c : r8) + (L >> 3);
    const int u = __builtin_amdgcn_readfirstlane(fdiv(item, fdiv_rcp(gz)));      // (remapped grids are below 2^20 workgroups: launcher)
    bzz = item - u * gz;
    bx = __builtin_amdgcn_readfirstlane(fdiv(u, fdiv_rcp(gy)));
    by = u - bx * gy;
  }
  const int64_t m0 = (int64_t)bx * BM;
  const int n0 = by * BN;
  const int HWg = p.Hg * p.Wg;
  const float r_hw = fdiv_rcp1(HWg), r_w = fdiv_rcp1(p.Wg);   // pixel decodes: fdivmod_px (common.h), M < 2^23 by the launcher
  // Conv2DTranspose forward: blockIdx.z = output-parity class, which owns tap-table entries [4c, 4c+4) and its own output pixels
  const int cls = p.cls_mode ? bzz : 0;
  const int tbase = cls * 4;
  const int ntaps = p.cls_mode ? (int)p.cls_ntaps[cls] : p.ntaps;
  const int oay = p.cls_mode ? (cls >> 1) : p.oay, oax = p.cls_mode ? (cls & 1) : p.oax;
  // batched GEMM (attention): blockIdx.z = b1*nb2 + b2 selects the operand bases (never together with the parity classes)
  const int bz = (!p.cls_mode && p.nb2 > 0) ? bzz : 0;
  const int bz1 = p.nb2 > 0 ? bz / p.nb2 : 0, bz2 = p.nb2 > 0 ? bz - bz1 * p.nb2 : 0;
  const int64_t ybatch = bz1 * p.ys1 + bz2 * p.ys2;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + bz1 * p.xs1 + bz2 * p.xs2), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + bz1 * p.ws1 + bz2 * p.ws2), 0, 0x7fffffff, 0x00020000);

  // ---- this lane's pieces: row (lane / 8) of piece wv + 4*it, chunk column q (fixed: see the layout note)
  const int lrow = lane >> 3;
  const int q = (lane & 7) ^ (((lane >> 4) + 4 * (wv & 1)) & 7);
  int abase[A_IT], py[A_IT], px[A_IT];
  bool pv[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int64_t m = m0 + 8 * (wv + 4 * it) + lrow;
    pv[it] = m < p.M;
    const int mm = pv[it] ? (int)m : 0;
    int b, rem, gy, gx;
    fdivmod_px(mm, HWg, r_hw, b, rem);
    fdivmod_px(rem, p.Wg, r_w, gy, gx);
    py[it] = __mul24(gy, p.isy);
    px[it] = __mul24(gx, p.isx);
    abase[it] = __mul24(__mul24(__mul24(b, p.Hi) + py[it], p.Wi) + px[it], p.ldx);
  }
  int wrow[W_IT];
  bool wok[W_IT];
#pragma unroll
  for (int it = 0; it < W_IT; ++it) {
    const int n = 8 * (wv + 4 * it) + lrow;
    wok[it] = n < BN && (n0 + n) < p.Nw;
    wrow[it] = wok[it] ? (n0 + n) * p.Kw : 0;
  }
  const int nks = (ntaps * p.cpt + 7) / 8;
  int ti = q / p.cpt;               // this lane's K chunk of the step being issued: (tap, chunk of it), 8 chunks further per step
  int c8 = q - ti * p.cpt;
  const int adv_t = 8 / p.cpt, adv_c = 8 - adv_t * p.cpt;
  __syncthreads();  // tap table visible
  // dense layers / 1x1 convs: ONE tap at the pixel itself - its table entry stays in registers (no LDS round trip in front of
  // every step's pieces) and no piece can fall outside the image
  const igemm_i32x4_t e0 = igemm_tap_read(s_tap4[tbase]);
  const bool dense = __builtin_amdgcn_readfirstlane((ntaps == 1) & (e0.x == 0) & (e0.y == 0)) != 0;

  auto issue = [&](int stage) {
    char* const sb = lds_raw + stage * STAGE;
    const bool tv = ti < ntaps;
    const igemm_i32x4_t e = dense ? e0 : igemm_tap_read(s_tap4[tbase + (tv ? ti : 0)]);
    const int da = e.z + c8 * 8, dw = e.w + c8 * 8;
    if (dense) {     // one unshifted tap: no table read, no range checks
#pragma unroll
      for (int it = 0; it < A_IT; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (igemm_lds_ptr_t)(sb + (wv + 4 * it) * 1024), 16, (tv & pv[it]) ? (uint32_t)(abase[it] + da) * 2u : IGEMM_OOB, 0, 0, 0);
    } else
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int iy = py[it] + e.x, ix = px[it] + e.y;
      const bool ok = tv & pv[it] & ((unsigned)iy < (unsigned)p.Hi) & ((unsigned)ix < (unsigned)p.Wi);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (igemm_lds_ptr_t)(sb + (wv + 4 * it) * 1024), 16, ok ? (uint32_t)(abase[it] + da) * 2u : IGEMM_OOB, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const bool ok = tv & wok[it];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (igemm_lds_ptr_t)(sb + BM * 128 + (wv + 4 * it) * 1024), 16, ok ? (uint32_t)(wrow[it] + dw) * 2u : IGEMM_OOB, 0, 0, 0);
    }
    ti += adv_t;                    // 8 chunks further, branch-free
    c8 += adv_c;
    if (c8 >= p.cpt) { c8 -= p.cpt; ++ti; }
  };

  f32x4_t acc[2][NT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, g = lane >> 4;
  int a_off[2], w_off[NT];   // byte offset of this lane's fragment row; its swizzle key is (row >> 1) & 7
#pragma unroll
  for (int a = 0; a < 2; ++a) a_off[a] = (wv * 32 + a * 16 + frow) * 128;
#pragma unroll
  for (int b = 0; b < NT; ++b) w_off[b] = (BM + b * 16 + frow) * 128;
  const int key = (frow >> 1) & 7;   // all fragment rows are frow + a multiple of 16: the same key

#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nks) issue(s);
  for (int ks = 0; ks < nks; ++ks) {
    // step ks has landed once at most the pieces of the younger steps are outstanding (vmcnt retires in issue order);
    // bare s_barrier: __syncthreads() would prepend a vmcnt(0)
    // (min(D-1, steps issued after ks) stages may stay in flight: near the end of the K loop fewer younger stages exist, and a wait
    // sized for D-1 of them would let pieces of stage ks itself be outstanding)
    static_assert(D <= 3, "the wait ladder below covers up to three stages in flight");
    if (D >= 3 && ks + 2 < nks) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NI) : "memory");
    else if (D >= 2 && ks + 1 < nks) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // everyone is also done reading stage (ks-1) % NSTAGE, which the next issue overwrites
    if (ks + D < nks) issue((ks + D) % NSTAGE);
    const char* const L = lds_raw + (ks % NSTAGE) * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int sl = ((4 * kk + g) ^ key) << 4;
      bf16x8_t xf[2], wf[NT];
#pragma unroll
      for (int a = 0; a < 2; ++a) xf[a] = *reinterpret_cast<const bf16x8_t*>(L + a_off[a] + sl);
#pragma unroll
      for (int b = 0; b < NT; ++b) wf[b] = *reinterpret_cast<const bf16x8_t*>(L + w_off[b] + sl);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
    }
  }

  int64_t opix[2];
  bool ovalid[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    int64_t m = m0 + wv * 32 + a * 16 + frow;
    ovalid[a] = m < p.M;
    int mm = ovalid[a] ? (int)m : 0;
    int b, rem, gy, gx;
    fdivmod_px(mm, HWg, r_hw, b, rem);
    fdivmod_px(rem, p.Wg, r_w, gy, gx);
    opix[a] = (int64_t)(__mul24(__mul24(b, p.Ho) + __mul24(gy, p.osy) + oay, p.Wo) + __mul24(gx, p.osx) + oax);   // (output pixels < 2^23: launcher)
  }
  const EpiArgs e = {p.scale, p.bias, p.res, p.y, p.ldy, p.ldr, p.Nout, p.act, p.alpha, p.out_f32, p.accumulate};
  const int nbase = n0 + g * 4;
  EpiConst<NT> ec;
  epi_const_load<NT>(ec, p.bias, nbase, p.Nout);
  static_assert(4 * epi_lds_bytes<2, NT>() <= NSTAGE * STAGE, "the transposing epilogue reuses the stage ring");
  __syncthreads();   // every wave is done with the last stage
  if (!conv_epilogue_lds<2, NT>(lds_raw + wv * epi_lds_bytes<2, NT>(), e, ec, acc, opix, ovalid, n0, lane, ybatch))
    conv_epilogue<2, NT>(e, ec, acc, opix, ovalid, nbase, ybatch);
#endif
}

// ---- persistent form for short K loops (plain mode, bias + activation + bf16 epilogue only) -----------------------------------
// With 7-14 K steps per tile the prologue (row decode, first-stage latency) and the epilogue are 40 % of a workgroup's life.
// Here a workgroup walks over pixel tiles blockIdx.x, blockIdx.x + gridDim.x, ... and the stage ring runs on ACROSS tile
// boundaries: the pieces of the next tile's first steps are in flight under the last MFMAs and the stores of the current one
// (the issue side keeps its own tile's row constants; the compute side only needs the tile index again for the epilogue).
template <int NT, int NSTAGE>
__global__ __launch_bounds__(256, 2) void igemm_dma_persist_kernel(const IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, BN = 16 * NT, BNP = BN < 32 ? 32 : BN;
  constexpr int STAGE = (BM + BNP) * 128;
  constexpr int A_IT = BM / 32, W_IT = BNP / 32;
  constexpr int NI = A_IT + W_IT;
  constexpr int D = NSTAGE - 1;
  extern __shared__ __attribute__((aligned(1024))) char lds_raw[];
  __shared__ __attribute__((aligned(16))) int s_tap4[32][4];   // dy, dx, da, dw

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 32) {
    const int dy = p.tap_dy[tid], dx = p.tap_dx[tid];
    s_tap4[tid][0] = dy;
    s_tap4[tid][1] = dx;
    s_tap4[tid][2] = (dy * p.Wi + dx) * p.ldx + p.tap_c[tid];
    s_tap4[tid][3] = p.tap_w[tid] * p.cpt * 8;
  }
  int bx = blockIdx.x, by = blockIdx.y, bzz = blockIdx.z;
  if (p.xcd_remap) {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, total = gx * gy * gz;
    const int L = bx + gx * (by + gy * bzz);
    const int c = L & 7, q8 = total >> 3, r8 = total & 7;
    const int item = c * q8 + (c < r8 ? c : r8) + (L >> 3);
    const int u = __builtin_amdgcn_readfirstlane(fdiv(item, fdiv_rcp(gz)));      // (remapped grids are below 2^20 workgroups: launcher)
    bzz = item - u * gz;
    bx = __builtin_amdgcn_readfirstlane(fdiv(u, fdiv_rcp(gy)));
    by = u - bx * gy;
  }
  const int n0 = by * BN;
  const int HWg = p.Hg * p.Wg;
  const float r_hw = fdiv_rcp1(HWg), r_w = fdiv_rcp1(p.Wg);   // pixel decodes: fdivmod_px (common.h), M < 2^23 by the launcher
  const int ntaps = p.ntaps;
  const int ntile = (int)((p.M + BM - 1) / BM);
  const int my_tiles = (bx < ntile) ? (ntile - bx + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  if (my_tiles == 0) return;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x7fffffff, 0x00020000);

  const int lrow = lane >> 3;
  const int q = (lane & 7) ^ (((lane >> 4) + 4 * (wv & 1)) & 7);
  int wrow[W_IT];
  bool wok[W_IT];
#pragma unroll
  for (int it = 0; it < W_IT; ++it) {
    const int n = 8 * (wv + 4 * it) + lrow;
    wok[it] = n < BN && (n0 + n) < p.Nw;
    wrow[it] = wok[it] ? (n0 + n) * p.Kw : 0;
  }
  const int nks = (ntaps * p.cpt + 7) / 8;
  const int ti0 = q / p.cpt, c80 = q - ti0 * p.cpt;
  const int adv_t = 8 / p.cpt, adv_c = 8 - adv_t * p.cpt;

  // ---- issue side: the tile whose steps are being staged
  int i_tile = 0, i_ks = 0, ti = ti0, c8 = c80;
  int abase[A_IT], py[A_IT], px[A_IT];
  bool pv[A_IT];
  auto decode_rows = [&](int tile_no) {
    const int64_t m0 = ((int64_t)bx + (int64_t)tile_no * gridDim.x) * BM;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int64_t m = m0 + 8 * (wv + 4 * it) + lrow;
      pv[it] = m < p.M;
      const int mm = pv[it] ? (int)m : 0;
      int b, rem, gy, gx;
      fdivmod_px(mm, HWg, r_hw, b, rem);
      fdivmod_px(rem, p.Wg, r_w, gy, gx);
      py[it] = __mul24(gy, p.isy);
      px[it] = __mul24(gx, p.isx);
      abase[it] = __mul24(__mul24(__mul24(b, p.Hi) + py[it], p.Wi) + px[it], p.ldx);
    }
  };
  decode_rows(0);
  __syncthreads();  // tap table visible
  const igemm_i32x4_t e0 = igemm_tap_read(s_tap4[0]);      // (see igemm_dma_kernel)
  const bool dense = __builtin_amdgcn_readfirstlane((ntaps == 1) & (e0.x == 0) & (e0.y == 0)) != 0;

  auto issue = [&](int stage) {
    char* const sb = lds_raw + stage * STAGE;
    const bool tv = ti < ntaps;
    const igemm_i32x4_t e = dense ? e0 : igemm_tap_read(s_tap4[tv ? ti : 0]);
    const int da = e.z + c8 * 8, dw = e.w + c8 * 8;
    if (dense) {
#pragma unroll
      for (int it = 0; it < A_IT; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (igemm_lds_ptr_t)(sb + (wv + 4 * it) * 1024), 16, (tv & pv[it]) ? (uint32_t)(abase[it] + da) * 2u : IGEMM_OOB, 0, 0, 0);
    } else
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int iy = py[it] + e.x, ix = px[it] + e.y;
      const bool ok = tv & pv[it] & ((unsigned)iy < (unsigned)p.Hi) & ((unsigned)ix < (unsigned)p.Wi);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (igemm_lds_ptr_t)(sb + (wv + 4 * it) * 1024), 16, ok ? (uint32_t)(abase[it] + da) * 2u : IGEMM_OOB, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const bool ok = tv & wok[it];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (igemm_lds_ptr_t)(sb + BM * 128 + (wv + 4 * it) * 1024), 16, ok ? (uint32_t)(wrow[it] + dw) * 2u : IGEMM_OOB, 0, 0, 0);
    }
    ti += adv_t;
    c8 += adv_c;
    if (c8 >= p.cpt) { c8 -= p.cpt; ++ti; }
    if (++i_ks == nks) {            // next step belongs to the next tile of this workgroup
      i_ks = 0;
      ti = ti0;
      c8 = c80;
      if (++i_tile < my_tiles) decode_rows(i_tile);
    }
  };

  f32x4_t acc[2][NT];
  const int frow = lane & 15, g = lane >> 4;
  int a_off[2], w_off[NT];
#pragma unroll
  for (int a = 0; a < 2; ++a) a_off[a] = (wv * 32 + a * 16 + frow) * 128;
#pragma unroll
  for (int b = 0; b < NT; ++b) w_off[b] = (BM + b * 16 + frow) * 128;
  const int key = (frow >> 1) & 7;
  const EpiArgs e = {nullptr, p.bias, nullptr, p.y, p.ldy, 0, p.Nout, p.act, p.alpha, 0, 0};
  const int nbase = n0 + g * 4;
  EpiConst<NT> ec;
  epi_const_load<NT>(ec, p.bias, nbase, p.Nout);

  const int G = my_tiles * nks;
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < G) issue(s);
  int c_ks = 0, c_tile = 0;
  for (int gs = 0; gs < G; ++gs) {
    // at most the pieces of the younger steps (and, right after a tile end, that tile's stores - younger still) outstanding
    static_assert(D <= 3, "the wait ladder below covers up to three stages in flight");
    if (D >= 3 && gs + 2 < G) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NI) : "memory");
    else if (D >= 2 && gs + 1 < G) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (gs + D < G) issue((gs + D) % NSTAGE);
    if (c_ks == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const char* const L = lds_raw + (gs % NSTAGE) * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int sl = ((4 * kk + g) ^ key) << 4;
      bf16x8_t xf[2], wf[NT];
#pragma unroll
      for (int a = 0; a < 2; ++a) xf[a] = *reinterpret_cast<const bf16x8_t*>(L + a_off[a] + sl);
#pragma unroll
      for (int b = 0; b < NT; ++b) wf[b] = *reinterpret_cast<const bf16x8_t*>(L + w_off[b] + sl);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
    }
    if (++c_ks < nks) continue;
    c_ks = 0;
    // ---- epilogue of this tile: no loads (bias in registers), the stores fly under the next tile's steps
    const int64_t m0 = ((int64_t)bx + (int64_t)c_tile * gridDim.x) * BM;
    ++c_tile;
    int64_t opix[2];
    bool ovalid[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      int64_t m = m0 + wv * 32 + a * 16 + frow;
      ovalid[a] = m < p.M;
      int mm = ovalid[a] ? (int)m : 0;
      int b, rem, gy, gx;
      fdivmod_px(mm, HWg, r_hw, b, rem);
      fdivmod_px(rem, p.Wg, r_w, gy, gx);
      opix[a] = (int64_t)(__mul24(__mul24(b, p.Ho) + __mul24(gy, p.osy) + p.oay, p.Wo) + __mul24(gx, p.osx) + p.oax);
    }
    conv_epilogue<2, NT, true>(e, ec, acc, opix, ovalid, nbase, 0);
  }
#endif
}

template <int NT, int NSTAGE>
static void igemm_dma_persist_launch_t(const IgemmParams& p, dim3 grid, hipStream_t s) {
  constexpr int BNP = 16 * NT < 32 ? 32 : 16 * NT;
  const size_t dyn = (size_t)NSTAGE * (128 + BNP) * 128;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)igemm_dma_persist_kernel<NT, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((igemm_dma_persist_kernel<NT, NSTAGE>), grid, dim3(256), dyn, s, p);
}

template <int NT, int NSTAGE>
static void igemm_dma_launch_t(const IgemmParams& p, dim3 grid, hipStream_t s) {
  constexpr int BNP = 16 * NT < 32 ? 32 : 16 * NT;
  const size_t dyn = (size_t)NSTAGE * (128 + BNP) * 128;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)igemm_dma_kernel<NT, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((igemm_dma_kernel<NT, NSTAGE>), grid, dim3(256), dyn, s, p);
}

static int launch_igemm(const IgemmParams& p_in, hipStream_t s) {
  IgemmParams p = p_in;
  if (p.M <= 0) return USSEG_OK;
  dim3 block(256);
  int64_t gx = cdiv64(p.M, 128);
  USSEG_CHECK_ARG(gx < (1ll << 31), "igemm: too many pixel tiles");
  const unsigned gz = p.cls_mode ? 4 : (p.nb2 > 0 ? (unsigned)p.nbatch : 1);
  // channel tile: the widest that covers Nout, narrowed while the launch has fewer workgroups than CUs (the 16x16 stages
  // are 32 pixel tiles: a 128-channel tile leaves 3/4 of the chip idle - the b0.up backward-data ran 63 us on 64 workgroups)
  static const int wg_min = getenv("USSEG_IGEMM_WG_MIN") ? atoi(getenv("USSEG_IGEMM_WG_MIN")) : 256;
  int nt = p.Nout <= 16 ? 1 : (p.Nout <= 32 ? 2 : (p.Nout <= 64 ? 4 : 8));
  while (nt > 2 && gx * ((p.Nout + 16 * nt - 1) / (16 * nt)) * gz < wg_min) nt >>= 1;
  const unsigned gy = (unsigned)((p.Nout + 16 * nt - 1) / (16 * nt));
  // K step 64 when the K loop is long enough to amortise the bigger stage (>= 16 chunks of 8 channels)
  static const int kc_env = getenv("USSEG_IGEMM_KC") ? atoi(getenv("USSEG_IGEMM_KC")) : 0;
  const int kc = kc_env ? kc_env : (p.ntaps * p.cpt >= 16 ? 8 : 4);
  const dim3 grid((unsigned)gx, gy, gz);
  const int slot = usseg_prof_start(1, s);
  {   // plain mode with a K loop worth a stage ring: the LDS-DMA variant
    static const int dma = getenv("USSEG_IGEMM_DMA") ? atoi(getenv("USSEG_IGEMM_DMA")) : 1;
    const int64_t nb = cdiv64(p.M, (int64_t)p.Hg * p.Wg);
    const bool fits = nb * p.Hi * p.Wi * p.ldx * 2 < 0x7fff0000ll && (int64_t)p.Nw * p.Kw * 2 < 0x7fff0000ll &&
                      p.M < (1 << 23) && nb * p.Hi * p.Wi < (1 << 23) && nb * p.Ho * p.Wo < (1 << 23);   // fdivmod_px / signed 24-bit multiplies in the row decodes
    const int k_chunks = p.cls_mode ? 4 * p.cpt : p.ntaps * p.cpt;   // a parity class has up to four taps
    static const int dma_min = getenv("USSEG_IGEMM_DMA_MIN") ? atoi(getenv("USSEG_IGEMM_DMA_MIN")) : 4;
    static const int dma_batched = getenv("USSEG_IGEMM_DMA_BATCHED") ? atoi(getenv("USSEG_IGEMM_DMA_BATCHED")) : 1;
    const bool bat_ok = p.nb2 <= 0 || (dma_batched && !p.cls_mode && (p.xs1 | p.xs2 | p.ws1 | p.ws2) % 8 == 0);   // 16-byte aligned batch bases
    static const int xcd_env = getenv("USSEG_IGEMM_XCD") ? atoi(getenv("USSEG_IGEMM_XCD")) : 1;
    p.xcd_remap = 0;
    if (dma && bat_ok && k_chunks >= dma_min && fits) {
      // activation-heavy launches only (pixels >= 16 x output channels: the operand worth sharing is the activation tile).  Measured per
      // configuration (same-box pairs): Arch A 6.71 / 6.75 -> 6.54 / 6.61 ms, cfg5 10.54 / 10.51 -> 10.41 / 10.44, Arch B neutral; on cfg4's token
      // GEMMs (8192 tokens x 512-2048 channels, weight-heavy) it cost 0.5 %, hence the rule
      p.xcd_remap = xcd_env && p.nb2 <= 0 && (int64_t)gx * gy * gz < (1ll << 20) && (int64_t)gy * gz > 1 && (xcd_env > 1 || p.M >= 16 * (int64_t)p.Nout);
      // short K loops with several pixel tiles per resident workgroup slot: the persistent form
      static const int persist = getenv("USSEG_IGEMM_PERSIST") ? atoi(getenv("USSEG_IGEMM_PERSIST")) : 1;
      const int nks = (p.ntaps * p.cpt + 7) / 8;
      const bool plain = !p.cls_mode && p.nb2 <= 0 && !p.res && !p.accumulate && !p.scale && !p.out_f32;
      const int64_t slots_x = 512 / (int64_t)gy;      // two workgroups per CU resident
      if (persist && plain && nks <= 16 && slots_x >= 32 && gx >= 2 * slots_x) {
        const dim3 pgrid((unsigned)slots_x, gy, 1);
        if (nt == 1) igemm_dma_persist_launch_t<1, 3>(p, pgrid, s);
        else if (nt == 2) igemm_dma_persist_launch_t<2, 3>(p, pgrid, s);
        else if (nt == 4) igemm_dma_persist_launch_t<4, 3>(p, pgrid, s);
        else igemm_dma_persist_launch_t<8, 2>(p, pgrid, s);
        usseg_prof_stop(1, slot, s);
        return usseg_check_launch("igemm_dma_persist");
      }
      // 128-channel tiles: two stages (64 KB, two workgroups per CU) by default; a launch that puts at most one workgroup on a CU
      // anyway and has a long K loop (the dense layers of the ViT / Swin: 8192 tokens x 2048 -> 512 is 256 workgroups x 32 steps)
      // is bound by the DMA latency of its single prefetched stage: it gets a deeper ring instead
      static const int st8 = getenv("USSEG_IGEMM_STAGES8") ? atoi(getenv("USSEG_IGEMM_STAGES8")) : 4;   // 4 stages: cfg4 11.01 -> 10.92 ms, the others unchanged (2 = off)
      static const int st8_wg = getenv("USSEG_IGEMM_STAGES8_WG") ? atoi(getenv("USSEG_IGEMM_STAGES8_WG")) : 256;
      const bool deep = nt == 8 && st8 > 2 && (int64_t)gx * gy * gz <= st8_wg && nks >= 8;
      if (nt == 1) igemm_dma_launch_t<1, 3>(p, grid, s);
      else if (nt == 2) igemm_dma_launch_t<2, 3>(p, grid, s);
      else if (nt == 4) igemm_dma_launch_t<4, 3>(p, grid, s);
      else if (deep && st8 == 3) igemm_dma_launch_t<8, 3>(p, grid, s);
      else if (deep) igemm_dma_launch_t<8, 4>(p, grid, s);
      else igemm_dma_launch_t<8, 2>(p, grid, s);
      usseg_prof_stop(1, slot, s);
      return usseg_check_launch("igemm_dma");
    }
  }
#define USSEG_IGEMM_LAUNCH(NT_, KC_)                                                                                         \
  do {                                                                                                                         \
    const size_t dyn = (size_t)2 * (128 + 16 * NT_) * (KC_ * 8 + 8) * sizeof(bf16_t) + 128 * sizeof(int);                      \
    static bool attr = false;                                                                                                  \
    if (!attr) {                                                                                                               \
      (void)hipFuncSetAttribute((const void*)igemm_kernel<NT_, KC_>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);   \
      attr = true;                                                                                                             \
    }                                                                                                                          \
    hipLaunchKernelGGL((igemm_kernel<NT_, KC_>), grid, block, dyn, s, p);                                                      \
  } while (0)
  if (kc == 8) {
    if (nt == 1) USSEG_IGEMM_LAUNCH(1, 8); else if (nt == 2) USSEG_IGEMM_LAUNCH(2, 8); else if (nt == 4) USSEG_IGEMM_LAUNCH(4, 8); else USSEG_IGEMM_LAUNCH(8, 8);
  } else {
    if (nt == 1) USSEG_IGEMM_LAUNCH(1, 4); else if (nt == 2) USSEG_IGEMM_LAUNCH(2, 4); else if (nt == 4) USSEG_IGEMM_LAUNCH(4, 4); else USSEG_IGEMM_LAUNCH(8, 4);
  }
#undef USSEG_IGEMM_LAUNCH
  usseg_prof_stop(1, slot, s);
  return usseg_check_launch("igemm");
}

static int check_desc(const UssegConvDesc* d, bool tconv) {
  USSEG_CHECK_ARG(d != nullptr, "null descriptor");
  USSEG_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0, "bad B/H/W");
  USSEG_CHECK_ARG(d->Cin > 0 && d->Cin % 8 == 0 && d->ldx % 8 == 0 && d->ldx >= d->Cin, "Cin/ldx must be multiples of 8");
  USSEG_CHECK_ARG(d->Cout > 0 && d->ldy >= d->Cout, "bad Cout/ldy");
  if (!(d->flags & USSEG_OUT_F32)) USSEG_CHECK_ARG(d->Cout % 8 == 0 && d->ldy % 8 == 0, "bf16 Cout/ldy must be multiples of 8");
  else USSEG_CHECK_ARG(d->ldy % 4 == 0, "f32 ldy must be a multiple of 4");
  if (tconv) USSEG_CHECK_ARG(d->ksize == 3 || d->ksize == 4, "tconv ksize must be 3 or 4");
  else USSEG_CHECK_ARG((d->ksize == 1 || d->ksize == 3) && d->dilation >= 1, "conv ksize must be 1 or 3");
  return USSEG_OK;
}

extern "C" int usseg_conv2d_fwd(const UssegConvDesc* d, const void* x, const void* wp, const float* bias,
                                const void* residual, int32_t ldr, void* y, usseg_stream_t stream) {
  int rc = check_desc(d, false);
  if (rc) return rc;
  USSEG_CHECK_ARG(x && wp && y, "null pointer");
  if (d->ksize == 3 &&
      usseg_try_launch_conv_big((const bf16_t*)x, (const bf16_t*)wp, y, bias, (const bf16_t*)residual, d->B, d->H, d->W, d->dilation, d->Cin,
                                d->ldx, d->Cout, d->ldy, ldr, roundup(d->Cout, 16), 9 * d->Cin, d->act, d->alpha,
                                (d->flags & USSEG_OUT_F32) ? 1 : 0, (d->flags & USSEG_ACCUMULATE) ? 1 : 0, 0, (hipStream_t)stream))
    return usseg_check_launch("conv_big");
  if (d->ksize == 3 &&
      usseg_try_launch_conv_halo((const bf16_t*)x, (const bf16_t*)wp, y, bias, (const bf16_t*)residual, d->B, d->H, d->W, d->dilation, d->Cin,
                                 d->ldx, d->Cout, d->ldy, ldr, roundup(d->Cout, 16), 9 * d->Cin, d->act, d->alpha,
                                 (d->flags & USSEG_OUT_F32) ? 1 : 0, (d->flags & USSEG_ACCUMULATE) ? 1 : 0, 0, (hipStream_t)stream))
    return usseg_check_launch("conv_halo");
  IgemmParams p = {};
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)wp; p.y = y; p.bias = bias; p.res = (const bf16_t*)residual; p.ldr = ldr;
  p.scale = usseg_epi_scale[0];
  p.B = d->B; p.Hg = d->H; p.Wg = d->W; p.M = (int64_t)d->B * d->H * d->W;
  p.Hi = d->H; p.Wi = d->W; p.ldx = d->ldx; p.isy = p.isx = 1;
  p.Ho = d->H; p.Wo = d->W; p.ldy = d->ldy; p.osy = p.osx = 1; p.oay = p.oax = 0;
  p.cpt = d->Cin / 8;
  const int k = d->ksize, half = k / 2;
  p.ntaps = k * k;
  for (int kh = 0; kh < k; ++kh)
    for (int kw = 0; kw < k; ++kw) {
      int t = kh * k + kw;
      p.tap_dy[t] = (int16_t)((kh - half) * d->dilation);
      p.tap_dx[t] = (int16_t)((kw - half) * d->dilation);
      p.tap_w[t] = (int16_t)t;
    }
  p.nchunks = p.ntaps * p.cpt;
  p.Nw = roundup(d->Cout, 16); p.Kw = p.ntaps * d->Cin; p.Nout = d->Cout;
  p.act = d->act; p.alpha = d->alpha; p.out_f32 = (d->flags & USSEG_OUT_F32) ? 1 : 0;
  p.accumulate = (d->flags & USSEG_ACCUMULATE) ? 1 : 0;
  return launch_igemm(p, (hipStream_t)stream);
}

extern "C" int usseg_conv2d_dgrad(const UssegConvDesc* d, const void* dy, const void* wp, const void* residual,
                                  int32_t ldr, void* dx, usseg_stream_t stream) {
  int rc = check_desc(d, false);
  if (rc) return rc;
  USSEG_CHECK_ARG(dy && wp && dx, "null pointer");
  USSEG_CHECK_ARG(!(d->flags & USSEG_OUT_F32) && d->Cout % 8 == 0, "dgrad needs bf16 dy with Cout % 8 == 0");
  if (d->ksize == 3 &&
      usseg_try_launch_conv_big((const bf16_t*)dy, (const bf16_t*)wp, dx, nullptr, (const bf16_t*)residual, d->B, d->H, d->W, d->dilation,
                                d->Cout, d->ldy, d->Cin, d->ldx, ldr, roundup(d->Cin, 16), 9 * d->Cout, USSEG_ACT_NONE, 0.f, 0,
                                (d->flags & USSEG_ACCUMULATE) ? 1 : 0, 1, (hipStream_t)stream))
    return usseg_check_launch("conv_big_dgrad");
  if (d->ksize == 3 &&
      usseg_try_launch_conv_halo((const bf16_t*)dy, (const bf16_t*)wp, dx, nullptr, (const bf16_t*)residual, d->B, d->H, d->W, d->dilation,
                                 d->Cout, d->ldy, d->Cin, d->ldx, ldr, roundup(d->Cin, 16), 9 * d->Cout, USSEG_ACT_NONE, 0.f, 0,
                                 (d->flags & USSEG_ACCUMULATE) ? 1 : 0, 1, (hipStream_t)stream))
    return usseg_check_launch("conv_halo_dgrad");
  IgemmParams p = {};
  p.x = (const bf16_t*)dy; p.w = (const bf16_t*)wp; p.y = dx; p.bias = nullptr; p.res = (const bf16_t*)residual; p.ldr = ldr;
  p.B = d->B; p.Hg = d->H; p.Wg = d->W; p.M = (int64_t)d->B * d->H * d->W;
  p.Hi = d->H; p.Wi = d->W; p.ldx = d->ldy; p.isy = p.isx = 1;
  p.Ho = d->H; p.Wo = d->W; p.ldy = d->ldx; p.osy = p.osx = 1; p.oay = p.oax = 0;
  p.cpt = d->Cout / 8;
  const int k = d->ksize, half = k / 2;
  p.ntaps = k * k;
  for (int kh = 0; kh < k; ++kh)
    for (int kw = 0; kw < k; ++kw) {
      int t = kh * k + kw;  // y[p] uses x[p + off_t]  =>  dx[q] gathers dy[q - off_t] with the same W[t]
      p.tap_dy[t] = (int16_t)(-(kh - half) * d->dilation);
      p.tap_dx[t] = (int16_t)(-(kw - half) * d->dilation);
      p.tap_w[t] = (int16_t)t;
    }
  p.nchunks = p.ntaps * p.cpt;
  p.Nw = roundup(d->Cin, 16); p.Kw = p.ntaps * d->Cout; p.Nout = d->Cin;
  p.act = USSEG_ACT_NONE; p.alpha = 0.f; p.out_f32 = 0; p.accumulate = (d->flags & USSEG_ACCUMULATE) ? 1 : 0;
  return launch_igemm(p, (hipStream_t)stream);
}

static int conv_multi(int32_t njobs, const UssegConvJob* jobs, int flip, usseg_stream_t stream) {
  USSEG_CHECK_ARG(jobs && njobs >= 1 && njobs <= 4, "conv multi: 1 <= njobs <= 4");
  bool all3 = true;     // every job a 3x3 - or, forward only, a 1x1 beside at least one 3x3 (it rides along as a centre-tap job: conv_big.hip)
  int n3 = 0, n1 = 0;
  for (int j = 0; j < njobs; ++j) {
    int rc = check_desc(&jobs[j].desc, false);
    if (rc) return rc;
    USSEG_CHECK_ARG(jobs[j].x && jobs[j].wp && jobs[j].y, "conv multi: null pointer");
    if (flip) USSEG_CHECK_ARG(!(jobs[j].desc.flags & USSEG_OUT_F32) && jobs[j].desc.Cout % 8 == 0, "dgrad needs bf16 dy with Cout % 8 == 0");
    all3 = all3 && (jobs[j].desc.ksize == 3 || (jobs[j].desc.ksize == 1 && !flip && jobs[j].desc.dilation == 1));
    n3 += jobs[j].desc.ksize == 3; n1 += jobs[j].desc.ksize == 1;
  }
  all3 = all3 && n3 >= 1;
  struct ScaleGuard {   // the per-job epilogue multipliers are visible to the launchers only during this call
    ~ScaleGuard() { for (int j = 0; j < 4; ++j) usseg_epi_scale[j] = nullptr; }
  } guard;
  if (njobs > 1 && all3) {
    for (int j = 0; j < njobs; ++j) usseg_epi_scale[j] = flip ? nullptr : jobs[j].scale;
    if (usseg_try_launch_conv_big_multi(njobs, jobs, flip, (hipStream_t)stream)) return usseg_check_launch("conv_big_multi");
    if (n1 == 0 && usseg_try_launch_conv_halo_multi(njobs, jobs, flip, (hipStream_t)stream)) return usseg_check_launch("conv_halo_multi");
  }
  for (int j = 0; j < njobs; ++j) {
    const UssegConvJob& q = jobs[j];
    for (int k = 0; k < 4; ++k) usseg_epi_scale[k] = nullptr;
    usseg_epi_scale[0] = flip ? nullptr : q.scale;
    int rc = flip ? usseg_conv2d_dgrad(&q.desc, q.x, q.wp, q.residual, q.ldr, q.y, stream)
                  : usseg_conv2d_fwd(&q.desc, q.x, q.wp, q.bias, q.residual, q.ldr, q.y, stream);
    if (rc) return rc;
  }
  return USSEG_OK;
}
extern "C" int usseg_conv2d_fwd_affine(const UssegConvDesc* d, const void* x, const void* wp, const float* scale, const float* shift,
                                       const void* residual, int32_t ldr, void* y, usseg_stream_t stream) {
  USSEG_CHECK_ARG(scale && shift && ((((uintptr_t)scale) | ((uintptr_t)shift)) & 15) == 0, "conv2d_fwd_affine: scale/shift must be non-null, 16-byte aligned");
  usseg_epi_scale[0] = scale;
  int rc = usseg_conv2d_fwd(d, x, wp, shift, residual, ldr, y, stream);
  usseg_epi_scale[0] = nullptr;
  return rc;
}
extern "C" int usseg_conv2d_fwd_multi(int32_t njobs, const UssegConvJob* jobs, usseg_stream_t stream) {
  return conv_multi(njobs, jobs, 0, stream);
}
extern "C" int usseg_conv2d_dgrad_multi(int32_t njobs, const UssegConvJob* jobs, usseg_stream_t stream) {
  return conv_multi(njobs, jobs, 1, stream);
}

// dx = sum over parallel conv branches of their backward-data passes, as ONE implicit GEMM whose K axis walks every
// (branch, tap, channel): dx is written once instead of once + a read-modify-write per further branch (for the 128-channel,
// 128x128 decoder stage that is 470 MB of traffic -> 100 MB).
extern "C" int usseg_conv2d_dgrad_branches(const UssegConvDesc* d, int32_t nbranches, const int32_t* ksize, const int32_t* dilation,
                                           const int32_t* ch_off, int32_t Cb, const void* dy, const void* wp_cat, const void* residual,
                                           int32_t ldr, void* dx, usseg_stream_t stream) {
  USSEG_CHECK_ARG(d && ksize && dilation && ch_off && dy && wp_cat && dx, "null pointer");
  USSEG_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cin % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0, "bad geometry");
  USSEG_CHECK_ARG(nbranches >= 1 && Cb > 0 && Cb % 8 == 0, "dgrad_branches: Cb must be a multiple of 8");
  IgemmParams p = {};
  p.x = (const bf16_t*)dy; p.w = (const bf16_t*)wp_cat; p.y = dx; p.bias = nullptr; p.res = (const bf16_t*)residual; p.ldr = ldr;
  p.B = d->B; p.Hg = d->H; p.Wg = d->W; p.M = (int64_t)d->B * d->H * d->W;
  p.Hi = d->H; p.Wi = d->W; p.ldx = d->ldy; p.isy = p.isx = 1;
  p.Ho = d->H; p.Wo = d->W; p.ldy = d->ldx; p.osy = p.osx = 1; p.oay = p.oax = 0;
  p.cpt = Cb / 8;
  int t = 0;
  for (int b = 0; b < nbranches; ++b) {
    const int k = ksize[b], half = k / 2;
    USSEG_CHECK_ARG((k == 1 || k == 3) && dilation[b] >= 1 && ch_off[b] % 8 == 0 && ch_off[b] + Cb <= d->ldy, "dgrad_branches: bad branch");
    for (int kh = 0; kh < k; ++kh)
      for (int kw = 0; kw < k; ++kw) {
        USSEG_CHECK_ARG(t < 32, "dgrad_branches: at most 32 taps in total");
        p.tap_dy[t] = (int16_t)(-(kh - half) * dilation[b]);     // y[p] uses x[p + off]  =>  dx[q] gathers dy[q - off]
        p.tap_dx[t] = (int16_t)(-(kw - half) * dilation[b]);
        p.tap_w[t] = (int16_t)t;                                  // K index = (running tap)*Cb + co in the concatenated operand
        p.tap_c[t] = (int16_t)ch_off[b];
        ++t;
      }
  }
  p.ntaps = t;
  p.nchunks = p.ntaps * p.cpt;
  p.Nw = roundup(d->Cin, 16); p.Kw = p.ntaps * Cb; p.Nout = d->Cin;
  p.act = USSEG_ACT_NONE; p.alpha = 0.f; p.out_f32 = 0; p.accumulate = (d->flags & USSEG_ACCUMULATE) ? 1 : 0;
  return launch_igemm(p, (hipStream_t)stream);
}

// Conv2DTranspose stride 2 'same': out[2i + kh - pad] += x[i] * w[kh], pad = 0 (k=3, crop end) / 1 (k=4).
extern "C" int usseg_tconv2d_fwd(const UssegConvDesc* d, const void* x, const void* wp, const float* bias, void* y,
                                 usseg_stream_t stream) {
  int rc = check_desc(d, true);
  if (rc) return rc;
  USSEG_CHECK_ARG(x && wp && y, "null pointer");
  const int k = d->ksize, pad = (k == 4) ? 1 : 0;
  IgemmParams p = {};
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)wp; p.y = y; p.bias = bias; p.res = nullptr; p.ldr = 0;
  p.B = d->B; p.Hg = d->H; p.Wg = d->W; p.M = (int64_t)d->B * d->H * d->W;
  p.Hi = d->H; p.Wi = d->W; p.ldx = d->ldx; p.isy = p.isx = 1;
  p.Ho = 2 * d->H; p.Wo = 2 * d->W; p.ldy = d->ldy; p.osy = p.osx = 2;
  p.cpt = d->Cin / 8;
  p.cls_mode = 1;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      const int c = a * 2 + b;
      int nt = 0;
      for (int kh = 0; kh < k; ++kh) {
        if (((kh - pad) & 1) != a) continue;      // 2*i + kh - pad == 2*gy + a
        int dyo = (a - (kh - pad)) / 2;           // i = gy + dyo
        for (int kw = 0; kw < k; ++kw) {
          if (((kw - pad) & 1) != b) continue;
          int dxo = (b - (kw - pad)) / 2;
          p.tap_dy[4 * c + nt] = (int16_t)dyo; p.tap_dx[4 * c + nt] = (int16_t)dxo; p.tap_w[4 * c + nt] = (int16_t)(kh * k + kw);
          ++nt;
        }
      }
      p.cls_ntaps[c] = (int16_t)nt;
    }
  p.ntaps = 4; p.nchunks = 4 * p.cpt;
  p.Nw = roundup(d->Cout, 16); p.Kw = k * k * d->Cin; p.Nout = d->Cout;
  p.act = d->act; p.alpha = d->alpha; p.out_f32 = (d->flags & USSEG_OUT_F32) ? 1 : 0; p.accumulate = 0;
  return launch_igemm(p, (hipStream_t)stream);
}

// dx[i] = sum_k dy[2i + kh - pad] * w[kh]: a stride-2 gather over dy (2H x 2W).
extern "C" int usseg_tconv2d_dgrad(const UssegConvDesc* d, const void* dy, const void* wp, const void* residual,
                                   int32_t ldr, void* dx, usseg_stream_t stream) {
  int rc = check_desc(d, true);
  if (rc) return rc;
  USSEG_CHECK_ARG(dy && wp && dx, "null pointer");
  USSEG_CHECK_ARG(!(d->flags & USSEG_OUT_F32) && d->Cout % 8 == 0, "tconv dgrad needs bf16 dy with Cout % 8 == 0");
  const int k = d->ksize, pad = (k == 4) ? 1 : 0;
  IgemmParams p = {};
  p.x = (const bf16_t*)dy; p.w = (const bf16_t*)wp; p.y = dx; p.bias = nullptr; p.res = (const bf16_t*)residual; p.ldr = ldr;
  p.B = d->B; p.Hg = d->H; p.Wg = d->W; p.M = (int64_t)d->B * d->H * d->W;
  p.Hi = 2 * d->H; p.Wi = 2 * d->W; p.ldx = d->ldy; p.isy = p.isx = 2;
  p.Ho = d->H; p.Wo = d->W; p.ldy = d->ldx; p.osy = p.osx = 1; p.oay = p.oax = 0;
  p.cpt = d->Cout / 8;
  p.ntaps = k * k;
  for (int kh = 0; kh < k; ++kh)
    for (int kw = 0; kw < k; ++kw) {
      int t = kh * k + kw;
      p.tap_dy[t] = (int16_t)(kh - pad); p.tap_dx[t] = (int16_t)(kw - pad); p.tap_w[t] = (int16_t)t;
    }
  p.nchunks = p.ntaps * p.cpt;
  p.Nw = roundup(d->Cin, 16); p.Kw = p.ntaps * d->Cout; p.Nout = d->Cin;
  p.act = USSEG_ACT_NONE; p.out_f32 = 0; p.accumulate = (d->flags & USSEG_ACCUMULATE) ? 1 : 0;
  return launch_igemm(p, (hipStream_t)stream);
}

// ---- batched GEMM on the same kernel: Y[b1,b2][m][n] = sum_k X[b1,b2][m][k] * W[b1,b2][n][k]  (both operands K-contiguous).
// Used for the ViT attention products (VisionTransformer.py:41,47): rows m = tokens, the head split is a channel slice.
extern "C" int usseg_gemm_nt_batched(const UssegGemmDesc* d, const void* x, const void* w, void* y, usseg_stream_t stream) {
  USSEG_CHECK_ARG(d && x && w && y, "null pointer");
  USSEG_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0 && d->K % 8 == 0 && d->ldx % 8 == 0 && d->ldw % 8 == 0, "gemm: K, ldx, ldw must be multiples of 8");
  USSEG_CHECK_ARG(d->nb1 > 0 && d->nb2 > 0 && (int64_t)d->nb1 * d->nb2 < 65536, "gemm: bad batch");
  const bool f32 = (d->flags & USSEG_OUT_F32) != 0;
  USSEG_CHECK_ARG(f32 ? d->ldy % 4 == 0 : (d->ldy % 8 == 0 && d->N % 8 == 0), "gemm: ldy / N alignment");
  IgemmParams p = {};
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = y;
  p.B = 1; p.Hg = 1; p.Wg = d->M; p.M = d->M;
  p.Hi = 1; p.Wi = d->M; p.ldx = d->ldx; p.isy = p.isx = 1;
  p.Ho = 1; p.Wo = d->M; p.ldy = d->ldy; p.osy = p.osx = 1;
  p.cpt = d->K / 8; p.ntaps = 1; p.nchunks = p.cpt;
  p.Nw = d->N; p.Kw = d->ldw; p.Nout = d->N;
  p.out_f32 = f32 ? 1 : 0;
  p.nb2 = d->nb2; p.nbatch = d->nb1 * d->nb2;
  p.xs1 = d->xs1; p.xs2 = d->xs2; p.ws1 = d->ws1; p.ws2 = d->ws2; p.ys1 = d->ys1; p.ys2 = d->ys2;
  return launch_igemm(p, (hipStream_t)stream);
}
