// 3x3 convolution for the channel-rich (MFMA / L2-fill bound) layers: 256-pixel x 16*NT-channel workgroup tile,
// LDS-DMA (global_load_lds) double-buffered operand stages, bf16 MFMA, gfx950.
//
// Why a third conv kernel: conv_halo.hip moves (halo tile + 9 x BN weight rows) from L2 to LDS per 32-channel chunk
// for only 128 x BN outputs (64-86 FLOP per staged byte); at the measured ~70 GB/s per CU of L2->LDS fill that caps
// the decoder's 512->64 / 256->64 / concats_2 layers at 15-20 % of the MFMA peak.  Here a workgroup owns 256 pixels
// (halo tile <= 576 pixels) x 64 channels: 165-250 FLOP per staged byte, the operands never pass through VGPRs
// (no ds_write pass, no staging registers) and the next chunk's DMA flies under the current chunk's 144 MFMAs per wave.
//
// LDS image of a stage (no padding: a DMA wave-instruction writes 1 KiB = 16 rows x 64 B linearly):
//   A: [halo pixel][32 ch] 64-B rows;  W: [tap*BN + n][32 k] 64-B rows.
//   16-B slot j of row r holds k-group q = j ^ (2*((r>>2)&1)) - the source address is permuted, the read applies the
//   same involution; with it every ds_read_b128 fragment read (16 consecutive rows) is bank-conflict free.
// Up to 4 independent jobs (the DecoderBlock's parallel dilation branches) share one launch: blockIdx.z = job.
#include "common.h"

struct BigJob {
  const bf16_t* x;
  const bf16_t* w;
  void* y;
  const float* bias;
  const float* scale;   // optional per-channel multiplier applied before the bias (folded inference BatchNorm)
  const bf16_t* res;
  int32_t H, W, d, Hl, Wl, PH, PW, NV;
  int32_t tiles_x, tiles_per_v, npatches;
  int32_t ldx, ldy, ldr, Cin, nchunks, Nw, Kw, Nout, act;
  float alpha;
  int32_t out_f32, accumulate, flip;
  int32_t gx, gy, npa, nstages;   // this job's grid extent; A pieces (16 halo pixels each) per stage; LDS stages (1 or 2)
  uint32_t x_bytes, w_bytes;   // buffer extents for the range-checked DMA
  int32_t mask_ch;             // quad-form transposed conv: channels per parity class (0 = every tap for every channel)
  int32_t one_tap;             // a 1x1 conv riding in a multi-job launch of 3x3 convs on the same input (the DecoderBlock's first branch): only the centre
                               // tap exists - its packed operand is [N][Cin] - the other taps' weight rows are zero-filled by the DMA and their MFMAs skipped
  uint16_t tapmask[4];         // stencil taps class c uses (fwd: class of the output-channel tile; dgrad: class of the K chunk)
};
struct BigParams {
  BigJob job[4];
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
// LDS-DMA through raw buffer loads: a lane whose byte offset is out of range (>= num_records) deposits ZEROS, which is
// how the halo's out-of-image pixels, the channel tail and the weight rows past Nw are filled (no zero page, no select
// between pointers).  OOB_OFF stays out of range after a chunk offset (< 64 KB) is added.
#define OOB_OFF 0x80000000u

// NS = 16-pixel strips per wave, NW = waves: workgroup tile = 16*NS*NW pixels x 16*NT channels.  NW = 8 with NS = 2 is the
// 256-pixel tile run by eight waves: the weight stage (18-36 KB per 32-channel chunk, the larger half of what a 128-pixel
// workgroup moves through the ~33 B/clk L2->LDS path) is shared by twice the pixels while the CU keeps the same number of waves.
template <int NT, int NS, int NW = 4>
__global__ __launch_bounds__(64 * NW) void conv_big_kernel(const BigParams P) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource builtins exist in the device pass only; the host pass needs just the stub
  constexpr int BN = 16 * NT;
  constexpr int W_BYTES = 9 * BN * 64;
  constexpr int NPW = 9 * BN / 16;            // W pieces per stage
  constexpr int W_IT = (NPW + NW - 1) / NW;
  constexpr int A_IT = ((NS * NW == 16 ? 36 : 18) + NW - 1) / NW;   // A pieces per wave: halo tile <= 576 (256-pixel tile) / 288 pixels
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const BigJob& p = P.job[blockIdx.z];
  if ((int)blockIdx.x >= p.gx || (int)blockIdx.y >= p.gy) return;

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.y * BN;
  const int HW2 = p.PW + 2, HPP = (p.PH + 2) * HW2, NHP = p.NV * HPP;
  const int a_bytes = p.npa * 1024, stage_bytes = a_bytes + W_BYTES;
  const int dd = p.d * p.d;
  // (all decodes below: indices of halo pixels / patches / tiles, far below 2^21)
  const float r_hpp = fdiv_rcp(HPP), r_hw2 = fdiv_rcp(HW2), r_tpv = fdiv_rcp(p.tiles_per_v), r_tx = fdiv_rcp(p.tiles_x), r_dd = fdiv_rcp(dd),
              r_d = fdiv_rcp(p.d);

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  // ---- DMA source byte offsets, fixed for the kernel (OOB_OFF = zero fill).  Lane -> (row = 16*piece + lane/4, slot = lane%4)
  const int slot_q = (lane & 3) ^ (((lane >> 4) & 1) << 1);   // k-group this lane's slot holds (row bit 2 == lane bit 4)
  uint32_t a_off[A_IT], w_off[W_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int hp = 16 * (wv + NW * it) + (lane >> 2);
    uint32_t off = OOB_OFF;
    if (hp < NHP) {
      int pi, rem, hy, hx;
      fdivmod(hp, HPP, r_hpp, pi, rem);
      fdivmod(rem, HW2, r_hw2, hy, hx);
      int gp = blockIdx.x * p.NV + pi;
      if (gp < p.npatches) {
        int v, tt, ty, tx, b, ab, la, lb;
        fdivmod(gp, p.tiles_per_v, r_tpv, v, tt);
        fdivmod(tt, p.tiles_x, r_tx, ty, tx);
        fdivmod(v, dd, r_dd, b, ab);
        fdivmod(ab, p.d, r_d, la, lb);
        int ly = __mul24(ty, p.PH) + hy - 1, lx = __mul24(tx, p.PW) + hx - 1;
        if ((unsigned)ly < (unsigned)p.Hl && (unsigned)lx < (unsigned)p.Wl)   // (24-bit multiplies: full rate; every factor is an image coordinate, a pixel index < 2^24 or a row stride)
          off = (uint32_t)(__mul24(__mul24(__mul24(b, p.H) + la + __mul24(p.d, ly), p.W) + lb + __mul24(p.d, lx), p.ldx) + slot_q * 8) * 2u;
      }
    }
    a_off[it] = off;
  }
#pragma unroll
  for (int it = 0; it < W_IT; ++it) {
    const int row = 16 * (wv + NW * it) + (lane >> 2);   // t*BN + n
    uint32_t off = OOB_OFF;
    if (row < 9 * BN) {
      int t = row / BN, n = row - t * BN;
      int tw = p.flip ? 8 - t : t;
      if (p.one_tap) tw = t == 4 ? 0 : -1;
      if (tw >= 0 && n0 + n < p.Nw) off = (uint32_t)((n0 + n) * p.Kw + tw * p.Cin + slot_q * 8) * 2u;
    }
    w_off[it] = off;
  }
  const int cq = slot_q * 8;

  auto issue = [&](int ck, int stage) {
    char* abuf = lds + stage * stage_bytes;
    char* wbuf = abuf + a_bytes;
    const uint32_t cb = ck * 64;                 // bytes
    const bool cok = ck * 32 + cq < p.Cin;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int j = wv + NW * it;
      if (j < p.npa) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(abuf + j * 1024), 16, cok ? a_off[it] + cb : OOB_OFF, 0, 0, 0);
      }
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int j = wv + NW * it;
      if (j < NPW) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(wbuf + j * 1024), 16, cok ? w_off[it] + cb : OOB_OFF, 0, 0, 0);
      }
    }
  };

  issue(0, 0);     // the first chunk flies under the output-pixel decode below (~300 instructions that nothing before the MFMAs needs)
  // ---- this lane's NS output pixels (one per 16-pixel strip of the wave)
  const int pl = lane & 15, qk = lane >> 4;
  const int rows_per_strip = 16 / p.PW, spp = (p.PH * p.PW) >> 4;
  int hb[NS];
  int64_t opix[NS];
  bool ovalid[NS];
#pragma unroll
  for (int a = 0; a < NS; ++a) {
    int s = wv * NS + a;
    int pi, sl, r, c;
    fdivmod(s, spp, fdiv_rcp(spp), pi, sl);
    fdivmod(pl, p.PW, fdiv_rcp(p.PW), r, c);
    int row = sl * rows_per_strip + r;
    hb[a] = pi * HPP + row * HW2 + c;
    int gp = blockIdx.x * p.NV + pi;
    ovalid[a] = gp < p.npatches;
    int gpc = ovalid[a] ? gp : 0;
    int v, tt, ty, tx, b, ab, la, lb;
    fdivmod(gpc, p.tiles_per_v, r_tpv, v, tt);
    fdivmod(tt, p.tiles_x, r_tx, ty, tx);
    fdivmod(v, dd, r_dd, b, ab);
    fdivmod(ab, p.d, r_d, la, lb);
    int iy = la + __mul24(p.d, __mul24(ty, p.PH) + row), ix = lb + __mul24(p.d, __mul24(tx, p.PW) + c);
    opix[a] = (int64_t)(__mul24(__mul24(b, p.H) + iy, p.W) + ix);     // (pixel index < 2^24: the launcher's 32-bit-offset check)
  }
  const int w_lane = pl * 64 + ((qk ^ (((pl >> 2) & 1) << 1)) << 4);

  f32x4_t acc[NS][NT];
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const bool dbuf = p.nstages == 2;
  for (int ck = 0; ck < p.nchunks; ++ck) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // chunk ck has landed for every wave; everyone is done reading the other stage
    if (dbuf && ck + 1 < p.nchunks) issue(ck + 1, (ck + 1) & 1);
    const char* abuf = lds + (dbuf ? (ck & 1) * stage_bytes : 0);
    const char* wbuf = abuf + a_bytes + w_lane;
    const uint32_t tmask = p.one_tap ? 0x010u : (p.mask_ch ? p.tapmask[(p.flip ? ck * 32 : n0) / p.mask_ch] : 0x1ffu);   // wave-uniform
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (!((tmask >> t) & 1u)) continue;     // this class's weights are identically zero at tap t
      const int toff = (t / 3) * HW2 + (t % 3);
      bf16x8_t xf[NS], wf[NT];
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const int hp = hb[a] + toff;
        xf[a] = *reinterpret_cast<const bf16x8_t*>(abuf + (hp << 6) + ((qk << 4) ^ ((hp & 4) << 3)));
      }
#pragma unroll
      for (int b = 0; b < NT; ++b) wf[b] = *reinterpret_cast<const bf16x8_t*>(wbuf + (t * BN + b * 16) * 64);
#pragma unroll
      for (int a = 0; a < NS; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
    }
    if (!dbuf && ck + 1 < p.nchunks) {     // single stage (two workgroups per CU overlap each other): refill after everyone has read
      __syncthreads();
      issue(ck + 1, 0);
    }
  }

  // ---- epilogue: lane holds Y[its pixel][n = 4*(lane>>4) + j] per n-tile
  const EpiArgs e = {p.scale, p.bias, p.res, p.y, p.ldy, p.ldr, p.Nout, p.act, p.alpha, p.out_f32, p.accumulate};
  const int nbase = n0 + qk * 4;
  EpiConst<NT> ec;
  epi_const_load<NT>(ec, p.bias, nbase, p.Nout);
  conv_epilogue<NS, NT>(e, ec, acc, opix, ovalid, nbase);
#endif
}

// ---- streaming variant for the HBM-bound layers (few input channels, many pixels: the stem and stage-1 convs) -------------
// The workgroup is persistent: the whole weight operand of its channel tile (nchunks x 9 x BN x 32, <= 40 KB) is DMA'd into
// LDS once, then it walks over pixel groups with the halo tiles double-buffered: the DMA of step s+1 is in flight under the
// MFMAs and the stores of step s.  Nothing passes through VGPRs on the way in, the bias sits in registers, and the epilogue
// issues no loads, so the only wait of a step is `vmcnt(stores of the previous step)` - the stores themselves keep flying.
// One patch per pixel group (NV == 1): the group's geometry is four scalars read from a small LDS table.
// Walk: XCD x (= blockIdx.x % 8) owns a contiguous eighth of the groups and its workgroups take adjacent groups at the
// same time, so concurrently staged halo tiles are neighbours in memory and shared halo columns hit in that XCD's L2.
// RES: y = act(conv + bias) + residual.  The residual tile of a group is loaded into registers at the top of the group's LAST
// chunk step - before that step's MFMAs and in program order before the next step's DMA, so its latency hides under the MFMAs -
// and the compiler's own wait before the adds drains (only then, once per group) the prefetch DMA issued after it.
// Up to 4 jobs of identical plan (the dilation branches of a decoder stage: same tensors' shapes, other dilation / weights / output slice)
// share one launch, blockIdx.z = job: a job's workgroups are dispatched after its predecessor's, so the jobs still stream one after the other
// - what is saved is the dispatch boundary between them (cache write-back, ramp-down and ramp-up: 5-9 us each at batch 16).
template <int NT, int NS, bool RES = false>
__global__ __launch_bounds__(256, 2) void conv_stream_kernel(const BigParams P, const int fast_wait) {
#if defined(__HIP_DEVICE_COMPILE__)
  const BigJob& p = P.job[blockIdx.z];
  constexpr int BN = 16 * NT;
  constexpr int W_BYTES = 9 * BN * 64;
  constexpr int NPW = 9 * BN / 16;            // W pieces per chunk
  constexpr int A_IT = NS == 4 ? 6 : 3;       // A pieces per wave: halo tile 18x18 = 324 pixels (NS=4) / 10x18 = 180 (NS=2)
  constexpr int MAXG = 64;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  __shared__ int s_grp[MAXG][4];               // per group: DMA origin (bytes), output pixel base, ly0 - 1, lx0 - 1

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.y * BN;
  const int HW2 = p.PW + 2, NHP = (p.PH + 2) * HW2;
  const int a_bytes = p.npa * 1024;
  char* const wlds = lds;
  char* const alds = lds + p.nchunks * W_BYTES;
  const int dd = p.d * p.d;
  const float r_nch = fdiv_rcp(p.nchunks);
  const float r_hw2 = fdiv_rcp(HW2), r_tpv = fdiv_rcp(p.tiles_per_v), r_tx = fdiv_rcp(p.tiles_x), r_dd = fdiv_rcp(dd), r_d = fdiv_rcp(p.d);

  const int ngroups = p.npatches;
  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3, nj = gridDim.x >> 3;   // gridDim.x is a multiple of 8
  const int gpx = (ngroups + 7) >> 3;
  const int g_lo = xcd * gpx + wj;
  int g_hi = (xcd + 1) * gpx;
  if (g_hi > ngroups) g_hi = ngroups;
  int ng = g_lo < g_hi ? (g_hi - g_lo + nj - 1) / nj : 0;
  if (ng > MAXG) ng = MAXG;                    // the host sizes the grid so that this never truncates
  if (ng == 0) return;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  const int slot_q = (lane & 3) ^ (((lane >> 4) & 1) << 1);
  const int cq = slot_q * 8;

  // weights: every chunk, once
  for (int j = wv; j < p.nchunks * NPW; j += 4) {
    const int ck = j / NPW, r = j - ck * NPW;
    const int row = 16 * r + (lane >> 2);      // t*BN + n
    const int t = row / BN, n = row - t * BN;
    int tw = p.flip ? 8 - t : t;
    if (p.one_tap) tw = t == 4 ? 0 : -1;
    const bool ok = tw >= 0 && (n0 + n) < p.Nw && (ck * 32 + cq) < p.Cin;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(wlds + j * 1024), 16,
                                             ok ? (uint32_t)((n0 + n) * p.Kw + tw * p.Cin + ck * 32 + cq) * 2u : OOB_OFF, 0, 0, 0);
  }
  // group table
  for (int i = tid; i < ng; i += 256) {
    const int g = g_lo + i * nj;
    int v, tt, ty, tx, b, ab, la, lb;
    fdivmod(g, p.tiles_per_v, r_tpv, v, tt);
    fdivmod(tt, p.tiles_x, r_tx, ty, tx);
    fdivmod(v, dd, r_dd, b, ab);
    fdivmod(ab, p.d, r_d, la, lb);
    const int64_t org = (((int64_t)(b * p.H + la + p.d * (ty * p.PH - 1))) * p.W + lb + p.d * (tx * p.PW - 1)) * p.ldx * 2;
    s_grp[i][0] = (int)(uint32_t)org;          // may wrap below zero: only in-image lanes (true offset >= 0) use it
    s_grp[i][1] = (b * p.H + la + p.d * ty * p.PH) * p.W + lb + p.d * tx * p.PW;
    s_grp[i][2] = ty * p.PH - 1;
    s_grp[i][3] = tx * p.PW - 1;
  }
  // this lane's halo items: byte offset relative to the group origin, and (hy, hx) for the in-image test
  uint32_t a_rel[A_IT];
  int a_hy[A_IT], a_hx[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int hp = 16 * (wv + 4 * it) + (lane >> 2);
    int hy, hx;
    fdivmod(hp, HW2, r_hw2, hy, hx);
    a_hy[it] = hp < NHP ? hy : -(1 << 20);     // never in range
    a_hx[it] = hx;
    a_rel[it] = (uint32_t)(((p.d * hy) * p.W + p.d * hx) * p.ldx + cq) * 2u;
  }
  // this lane's NS output pixels
  const int pl = lane & 15, qk = lane >> 4;
  const int rows_per_strip = 16 / p.PW;
  int hb[NS], orel[NS];
#pragma unroll
  for (int a = 0; a < NS; ++a) {
    const int sl = wv * NS + a;
    int r, c;
    fdivmod(pl, p.PW, fdiv_rcp(p.PW), r, c);
    const int row = sl * rows_per_strip + r;
    hb[a] = row * HW2 + c;
    orel[a] = p.d * row * p.W + p.d * c;
  }
  const int w_lane = pl * 64 + ((qk ^ (((pl >> 2) & 1) << 1)) << 4);
  const EpiArgs e = {p.scale, p.bias, p.res, p.y, p.ldy, p.ldr, p.Nout, p.act, p.alpha, p.out_f32, p.accumulate};
  const int nbase = n0 + qk * 4;
  EpiConst<NT> ec;
  epi_const_load<NT>(ec, p.bias, nbase, p.Nout);
  __syncthreads();                             // group table visible

  auto issue_a = [&](int s) {
    const int gl = __builtin_amdgcn_readfirstlane(fdiv(s, r_nch)), ck = s - gl * p.nchunks;
    char* abuf = alds + (s & 1) * a_bytes;
    const uint32_t org = (uint32_t)s_grp[gl][0] + ck * 64;
    const int lym1 = s_grp[gl][2], lxm1 = s_grp[gl][3];
    const bool cok = ck * 32 + cq < p.Cin;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int j = wv + 4 * it;
      if (j < p.npa) {
        const bool ok = cok && (unsigned)(lym1 + a_hy[it]) < (unsigned)p.Hl && (unsigned)(lxm1 + a_hx[it]) < (unsigned)p.Wl;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(abuf + j * 1024), 16, ok ? org + a_rel[it] : OOB_OFF, 0, 0, 0);
      }
    }
  };

  f32x4_t acc[NS][NT];
  const int total = ng * p.nchunks;
  issue_a(0);
  for (int s = 0; s < total; ++s) {
    const int gl = __builtin_amdgcn_readfirstlane(fdiv(s, r_nch)), ck = s - gl * p.nchunks;
    // step s has landed: everything older than the previous step's NS*NT stores is complete (vmcnt retires in order)
    // (a bare s_barrier: __syncthreads() would prepend its own vmcnt(0) and drain the stores after all)
    if (fast_wait && s > 0 && ck == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NS * NT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ... for every wave, and everyone is done reading the other buffer
    uint2 rr[RES ? NS : 1][RES ? NT : 1];
    if constexpr (RES) {
      if (ck == p.nchunks - 1) {
        const int ob = s_grp[gl][1];
#pragma unroll
        for (int a = 0; a < NS; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b) {
            const int n = nbase + b * 16;
            rr[a][b] = *reinterpret_cast<const uint2*>(p.res + (int64_t)(ob + orel[a]) * p.ldr + (n < p.Nout ? n : 0));
          }
      }
    }
    if (s + 1 < total) issue_a(s + 1);
    if (ck == 0) {
#pragma unroll
      for (int a = 0; a < NS; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const char* abuf = alds + (s & 1) * a_bytes;
    const char* wbuf = wlds + ck * W_BYTES + w_lane;
    auto do_tap = [&](int t) {
      const int toff = (t / 3) * HW2 + (t % 3);
      bf16x8_t xf[NS], wf[NT];
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const int hp = hb[a] + toff;
        xf[a] = *reinterpret_cast<const bf16x8_t*>(abuf + (hp << 6) + ((qk << 4) ^ ((hp & 4) << 3)));
      }
#pragma unroll
      for (int b = 0; b < NT; ++b) wf[b] = *reinterpret_cast<const bf16x8_t*>(wbuf + (t * BN + b * 16) * 64);
#pragma unroll
      for (int a = 0; a < NS; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
    };
    if constexpr (RES || NT == 4) {   // (never a ride-along 1x1 - residual jobs and 64-channel tiles go elsewhere, stream_plan -: no branch here; with it the
                                      //  64-channel, 256-pixel residual variant spilled 92 registers and the stage-1 concats_2 launch went from 34 to 49 us)
#pragma unroll
      for (int t = 0; t < 9; ++t) do_tap(t);
    } else if (p.one_tap) do_tap(4);      // (one uniform branch per step: the nine-tap body stays one straight-line block)
    else {
#pragma unroll
      for (int t = 0; t < 9; ++t) do_tap(t);
    }
    if (ck != p.nchunks - 1) continue;
    int64_t opix[NS];
    bool ovalid[NS];
    const int obase = s_grp[gl][1];
#pragma unroll
    for (int a = 0; a < NS; ++a) { opix[a] = obase + orel[a]; ovalid[a] = true; }
    if constexpr (RES) {
      const float alpha = p.alpha;
      const int act = p.act;
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        bf16_t* const yrow = reinterpret_cast<bf16_t*>(p.y) + opix[a] * p.ldy;
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          const int n = nbase + b * 16;
          if (n >= p.Nout) continue;
          const float v0 = apply_act(acc[a][b][0] + ec.bb[b].x, act, alpha) + __uint_as_float(rr[a][b].x << 16);
          const float v1 = apply_act(acc[a][b][1] + ec.bb[b].y, act, alpha) + __uint_as_float(rr[a][b].x & 0xffff0000u);
          const float v2 = apply_act(acc[a][b][2] + ec.bb[b].z, act, alpha) + __uint_as_float(rr[a][b].y << 16);
          const float v3 = apply_act(acc[a][b][3] + ec.bb[b].w, act, alpha) + __uint_as_float(rr[a][b].y & 0xffff0000u);
          uint2 o;
          o.x = pack2bf(v0, v1);
          o.y = pack2bf(v2, v3);
          *reinterpret_cast<uint2*>(yrow + n) = o;
        }
      }
    } else {
      conv_epilogue<NS, NT, true>(e, ec, acc, opix, ovalid, nbase);
    }
  }
#endif
}

// Fills a job from a conv geometry; returns 0 if the geometry does not fit the 256-pixel tiling.
static int big_fill_job(BigJob& p, const bf16_t* x, const bf16_t* w, void* y, const float* bias, const bf16_t* res, int B, int H, int W,
                        int d, int Cin, int ldx, int Nout, int ldy, int ldr, int Nw, int Kw, int act, float alpha, int out_f32,
                        int accumulate, int flip, int PX, int any_cin = 0) {
  static const int min_cin = getenv("USSEG_BIG_MIN_CIN") ? atoi(getenv("USSEG_BIG_MIN_CIN")) : 33;
  if (d < 1 || H % d || W % d || (Cin < min_cin && !any_cin)) return 0;
  const int Hl = H / d, Wl = W / d;
  int PW;
  if (Wl % 16 == 0) PW = 16;
  else if (Wl == 8 || Wl == 4) PW = Wl;
  else return 0;
  int PH = PX / PW;
  if (PH > Hl) PH = Hl;
  if (Hl % PH || (PH * PW) % 16 || PX % (PH * PW)) return 0;
  const int NV = PX / (PH * PW);
  const int NHP = NV * (PH + 2) * (PW + 2);
  if (NHP > (PX == 256 ? 576 : 288)) return 0;
  if ((int64_t)B * H * W * ldx >= (1ll << 30) || (int64_t)Nw * Kw >= (1ll << 30)) return 0;   // byte offsets < 2^31
  if ((int64_t)B * H * W >= (1ll << 23) || (int64_t)B * d * d * (Hl / PH) * (Wl / PW) >= (1ll << 20)) return 0;   // signed 24-bit multiplies / fdiv() in the kernels' tile decodes
  p = {};
  p.x = x; p.w = w; p.y = y; p.bias = bias; p.res = res;
  p.H = H; p.W = W; p.d = d; p.Hl = Hl; p.Wl = Wl; p.PH = PH; p.PW = PW; p.NV = NV;
  p.tiles_x = Wl / PW;
  p.tiles_per_v = (Hl / PH) * p.tiles_x;
  p.npatches = B * d * d * p.tiles_per_v;
  p.ldx = ldx; p.ldy = ldy; p.ldr = ldr; p.Cin = Cin; p.nchunks = (Cin + 31) / 32;
  p.Nw = Nw; p.Kw = Kw; p.Nout = Nout; p.act = act; p.alpha = alpha; p.out_f32 = out_f32; p.accumulate = accumulate; p.flip = flip;
  p.gx = (p.npatches + NV - 1) / NV;
  p.x_bytes = (uint32_t)((int64_t)B * H * W * ldx * 2);
  p.w_bytes = (uint32_t)((int64_t)Nw * Kw * 2);
  p.npa = (NHP + 15) / 16;
  return 1;
}

static int nt_for(int nout) { return nout <= 16 ? 1 : (nout <= 32 ? 2 : 4); }

template <int NT, int NS, int NW = 4>
static void big_launch_t(const BigParams& P, dim3 grid, size_t dyn, hipStream_t s) {
  static bool attr_done = false;   // > 64 KB of LDS per workgroup needs the opt-in (one-time host call, never a stream op)
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_big_kernel<NT, NS, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_big_kernel<NT, NS, NW>), grid, dim3(64 * NW), dyn, s, P);
}

static void big_launch(BigParams& P, int njobs, int nt, int PX, hipStream_t s, int w8 = 0) {
  int gx = 0, gy = 0, npa = 0;
  for (int j = 0; j < njobs; ++j) {
    BigJob& p = P.job[j];
    p.gy = (p.Nout + 16 * nt - 1) / (16 * nt);
    if (p.gx > gx) gx = p.gx;
    if (p.gy > gy) gy = p.gy;
    if (p.npa > npa) npa = p.npa;
  }
  // One LDS stage with several co-resident workgroups per CU measured 1.3-1.4x faster than a double-buffered single workgroup -
  // but when the launch has at most two workgroups per CU there is nobody to overlap with, and the second stage (the next
  // chunk's DMA in flight under this chunk's MFMAs) wins: -10...-15 % on concats_2 256->512 @16x16, conv_more and the decoder's
  // 512->64 branches at batch 16 and 32 (tools/bench_conv.py, round 2), +15...+30 % when forced on the 1024-workgroup launches.
  // USSEG_BIG_STAGES: 0 = this rule (default), 1 / 2 = forced.
  static const int stages_env = getenv("USSEG_BIG_STAGES") ? atoi(getenv("USSEG_BIG_STAGES")) : 0;
  int64_t total_wg = 0;
  for (int j = 0; j < njobs; ++j) total_wg += (int64_t)P.job[j].gx * P.job[j].gy;
  const int nstages = stages_env == 2 ? 2 : (stages_env == 1 ? 1 : (total_wg <= 512 ? 2 : 1));
  for (int j = 0; j < njobs; ++j) { P.job[j].npa = npa; P.job[j].nstages = nstages; }   // one stage layout for the whole launch
  const size_t dyn = nstages * ((size_t)npa * 1024 + (size_t)9 * 16 * nt * 64);
  const dim3 grid(gx, gy, njobs);
  const int slot = usseg_prof_start(1, s);
  if (PX == 256 && w8) {          // 256-pixel tile on eight waves (two strips each)
    if (nt == 1) big_launch_t<1, 2, 8>(P, grid, dyn, s);
    else if (nt == 2) big_launch_t<2, 2, 8>(P, grid, dyn, s);
    else big_launch_t<4, 2, 8>(P, grid, dyn, s);
  } else if (PX == 256) {
    if (nt == 1) big_launch_t<1, 4>(P, grid, dyn, s);
    else if (nt == 2) big_launch_t<2, 4>(P, grid, dyn, s);
    else big_launch_t<4, 4>(P, grid, dyn, s);
  } else {
    if (nt == 1) big_launch_t<1, 2>(P, grid, dyn, s);
    else if (nt == 2) big_launch_t<2, 2>(P, grid, dyn, s);
    else big_launch_t<4, 2>(P, grid, dyn, s);
  }
  usseg_prof_stop(1, slot, s);
}

struct BigGeom {   // what big_fill_job needs, so that a job can be re-filled for another pixel tile
  const bf16_t* x; const bf16_t* w; void* y; const float* bias; const bf16_t* res;
  int B, H, W, d, Cin, ldx, Nout, ldy, ldr, Nw, Kw, act; float alpha; int out_f32, accumulate, flip;
  int one_tap = 0;
};

template <int NT, int NS, bool RES = false>
static void stream_launch_t(const BigParams& P, dim3 grid, size_t dyn, int fast_wait, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv_stream_kernel<NT, NS, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);   // + 1 KB static
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_stream_kernel<NT, NS, RES>), grid, dim3(256), dyn, s, P, fast_wait);
}

// Plan of the streaming kernel for one job whose weights fit in LDS and whose launch has enough pixel groups to amortise them.
struct StreamPlan {
  BigJob job;
  int nt, PX, res, fast_wait;
  size_t dyn;
  unsigned wgx, gy;
};
static int stream_plan(const BigGeom& g, const float* epi_scale, StreamPlan& sp) {
  static const int mode = getenv("USSEG_STREAM") ? atoi(getenv("USSEG_STREAM")) : 1;
  static const int px_env = getenv("USSEG_STREAM_PX") ? atoi(getenv("USSEG_STREAM_PX")) : 0;
  // read per call (not cached): the parity tests use these two to drive small shapes through many steps per workgroup
  const char* ms_env = getenv("USSEG_STREAM_MIN_STEPS");
  const char* wgx_env = getenv("USSEG_STREAM_WGX");
  const int min_steps = ms_env ? atoi(ms_env) : 1;
  static const int occ_env = getenv("USSEG_STREAM_OCC") ? atoi(getenv("USSEG_STREAM_OCC")) : 0;
  if (!mode || usseg_tap_mask.group_ch) return 0;
  // its epilogue is bias + activation + bf16 store (no loads inside the streaming loop), or the RES variant with a preloaded residual
  static const int res_env = getenv("USSEG_STREAM_RES") ? atoi(getenv("USSEG_STREAM_RES")) : 1;
  if (g.out_f32 || g.accumulate || (!g.flip && epi_scale)) return 0;
  if (g.res && (!res_env || (g.ldr & 3))) return 0;
  const int nt = nt_for(g.Nout);
  const int nchunks = (g.Cin + 31) / 32;
  const size_t wbytes = (size_t)nchunks * 9 * 16 * nt * 64;
  if (wbytes > 40960) return 0;
  const int gy = (g.Nout + 16 * nt - 1) / (16 * nt);
  BigJob& p = sp.job;
  for (int PX = 256; PX >= 128; PX >>= 1) {
    if (px_env && PX != px_env) continue;
    if (!big_fill_job(p, g.x, g.w, g.y, g.bias, g.res, g.B, g.H, g.W, g.d, g.Cin, g.ldx, g.Nout, g.ldy, g.ldr, g.Nw, g.Kw, g.act, g.alpha,
                      g.out_f32, g.accumulate, g.flip, PX, 1))
      continue;
    if (p.NV != 1 || p.PW != 16) continue;
    if (g.one_tap && (nt == 4 || g.res)) return 0;      // (those instantiations carry no centre-tap path: conv_big takes the launch)
    p.one_tap = g.one_tap;
    const size_t dyn = wbytes + 2 * (size_t)p.npa * 1024;
    int occ = (int)((160 * 1024 - 2048) / (dyn + 1024));
    if (occ > 4) occ = 4;
    if (occ_env > 0 && occ_env < occ) occ = occ_env;
    if (occ < 1) continue;
    int wgx = (256 * occ / gy) & ~7;             // one resident wave of workgroups, a multiple of 8 (per-XCD walk)
    if (wgx < 8) wgx = 8;
    if (wgx_env && atoi(wgx_env) >= 8) wgx = atoi(wgx_env) & ~7;
    const int gpx = (p.npatches + 7) / 8;        // groups per XCD
    int per_wg = (gpx + wgx / 8 - 1) / (wgx / 8);
    if (per_wg * nchunks < min_steps && !px_env) continue;   // too few steps to amortise the weights: the tiled kernels win
    if (per_wg > 64) { wgx = ((gpx + 63) / 64) * 8; per_wg = 64; }
    // the in-order wait on "all but the youngest NS*NT operations" needs every step to issue exactly that many stores
    sp.fast_wait = g.Nout % (16 * nt) == 0 && !g.res;   // (the residual variant's own loads are in the queue too)
    sp.nt = nt; sp.PX = PX; sp.res = g.res != nullptr; sp.dyn = dyn; sp.wgx = (unsigned)wgx; sp.gy = (unsigned)gy;
    return 1;
  }
  return 0;
}
// jobs[0..n) share one plan: one launch, blockIdx.z = job
static void stream_launch(const StreamPlan* sp, int n, hipStream_t s) {
  BigParams P;
  for (int j = 0; j < n; ++j) P.job[j] = sp[j].job;
  for (int j = n; j < 4; ++j) P.job[j] = sp[0].job;
  const dim3 grid(sp[0].wgx, sp[0].gy, (unsigned)n);
  const int nt = sp[0].nt, PX = sp[0].PX, fast_wait = sp[0].fast_wait;
  const size_t dyn = sp[0].dyn;
  const int slot = usseg_prof_start(1, s);
  if (sp[0].res) {
    if (PX == 256) {
      if (nt == 1) stream_launch_t<1, 4, true>(P, grid, dyn, fast_wait, s);
      else if (nt == 2) stream_launch_t<2, 4, true>(P, grid, dyn, fast_wait, s);
      else stream_launch_t<4, 4, true>(P, grid, dyn, fast_wait, s);
    } else {
      if (nt == 1) stream_launch_t<1, 2, true>(P, grid, dyn, fast_wait, s);
      else if (nt == 2) stream_launch_t<2, 2, true>(P, grid, dyn, fast_wait, s);
      else stream_launch_t<4, 2, true>(P, grid, dyn, fast_wait, s);
    }
  } else if (PX == 256) {
    if (nt == 1) stream_launch_t<1, 4>(P, grid, dyn, fast_wait, s);
    else if (nt == 2) stream_launch_t<2, 4>(P, grid, dyn, fast_wait, s);
    else stream_launch_t<4, 4>(P, grid, dyn, fast_wait, s);
  } else {
    if (nt == 1) stream_launch_t<1, 2>(P, grid, dyn, fast_wait, s);
    else if (nt == 2) stream_launch_t<2, 2>(P, grid, dyn, fast_wait, s);
    else stream_launch_t<4, 2>(P, grid, dyn, fast_wait, s);
  }
  usseg_prof_stop(1, slot, s);
}
static bool stream_same_plan(const StreamPlan& a, const StreamPlan& b) {
  return a.nt == b.nt && a.PX == b.PX && a.res == b.res && a.fast_wait == b.fast_wait && a.dyn == b.dyn && a.wgx == b.wgx && a.gy == b.gy;
}

// Chooses the tile (pixels per workgroup, channel tile) and launches; 0 if no tiling fits or the launch would be too small.
static int big_plan_and_launch(const BigGeom* g, int njobs, hipStream_t s) {
  static const int mode = getenv("USSEG_BIG") ? atoi(getenv("USSEG_BIG")) : 1;
  static const int px_env = getenv("USSEG_BIG_PX") ? atoi(getenv("USSEG_BIG_PX")) : 0;
  static const int nt_env = getenv("USSEG_BIG_NT") ? atoi(getenv("USSEG_BIG_NT")) : 0;
  static const int min_wg = getenv("USSEG_BIG_MIN_WG") ? atoi(getenv("USSEG_BIG_MIN_WG")) : 128;
  static const int wg_min = getenv("USSEG_BIG_WG_TILE") ? atoi(getenv("USSEG_BIG_WG_TILE")) : 512;
  if (!mode) return 0;
  {   // HBM-bound jobs whose weights fit in LDS: the streaming kernel, one full-chip launch per job
    static const int multi = getenv("USSEG_STREAM_MULTI") ? atoi(getenv("USSEG_STREAM_MULTI")) : 1;
    const char* merge_env = getenv("USSEG_STREAM_MERGE");      // read per call: the parity test flips it inside one process
    const int merge = merge_env ? atoi(merge_env) : 1;         // jobs of one plan in ONE launch (0: a launch per job)
    bool all = (njobs == 1 || multi) && njobs <= 4;
    StreamPlan sp[4];
    for (int j = 0; j < njobs && all; ++j) all = stream_plan(g[j], usseg_epi_scale[j], sp[j]) != 0;
    if (all) {
      for (int j = 0; j < njobs;) {
        int n = 1;
        while (merge && j + n < njobs && stream_same_plan(sp[j], sp[j + n])) ++n;
        stream_launch(sp + j, n, s);
        j += n;
      }
      return 1;
    }
  }
  int nt_max = 1;
  for (int j = 0; j < njobs; ++j) nt_max = nt_for(g[j].Nout) > nt_max ? nt_for(g[j].Nout) : nt_max;
  BigParams P;
  // Tile policy (measured at batch 16 and 128, tools/bench_conv.py): 128-pixel tiles unless the launch is large enough to
  // give every CU eight 256-pixel workgroups; then the widest channel tile that still yields two workgroups per CU.
  int best_px = 0, best_nt = 0;
  int64_t best_wg = 0;
  for (int PX = 256; PX >= 128 && !best_px; PX >>= 1) {
    if (px_env && PX != px_env) continue;
    bool ok = true;
    for (int j = 0; j < njobs && ok; ++j)
      ok = big_fill_job(P.job[j], g[j].x, g[j].w, g[j].y, g[j].bias, g[j].res, g[j].B, g[j].H, g[j].W, g[j].d, g[j].Cin, g[j].ldx, g[j].Nout,
                        g[j].ldy, g[j].ldr, g[j].Nw, g[j].Kw, g[j].act, g[j].alpha, g[j].out_f32, g[j].accumulate, g[j].flip, PX);
    if (!ok) continue;
    for (int nt = nt_max; nt >= 1; nt >>= 1) {
      if (nt_env && nt != nt_env) continue;
      int64_t wg = 0;
      for (int j = 0; j < njobs; ++j) wg += (int64_t)P.job[j].gx * ((P.job[j].Nout + 16 * nt - 1) / (16 * nt));
      if (PX == 256 && !px_env) {
        if (nt == nt_max && wg >= 2048) { best_px = PX; best_nt = nt; best_wg = wg; }
        break;
      }
      best_px = PX; best_nt = nt; best_wg = wg;      // narrower tiles only add workgroups: keep the last one tried
      if (wg >= wg_min || nt_env) break;
    }
  }
  if (!best_px || (best_wg < min_wg && !(px_env || nt_env))) return 0;
  // 128-pixel plan: if the 256-pixel tiling fits too and still leaves enough workgroups, run it on eight waves - the same
  // waves per CU, half the weight-stage traffic (read per call: the parity tests force it with USSEG_BIG_W8=2)
  int w8 = 0;
  {
    const char* w8e = getenv("USSEG_BIG_W8");
    const int w8_mode = w8e ? atoi(w8e) : 1;
    static const int w8_min = getenv("USSEG_BIG_W8_MIN") ? atoi(getenv("USSEG_BIG_W8_MIN")) : 128;
    if (w8_mode && best_px == 128) {
      bool ok = true;
      int64_t wg = 0;
      for (int j = 0; j < njobs && ok; ++j) {
        ok = big_fill_job(P.job[j], g[j].x, g[j].w, g[j].y, g[j].bias, g[j].res, g[j].B, g[j].H, g[j].W, g[j].d, g[j].Cin, g[j].ldx, g[j].Nout,
                          g[j].ldy, g[j].ldr, g[j].Nw, g[j].Kw, g[j].act, g[j].alpha, g[j].out_f32, g[j].accumulate, g[j].flip, 256);
        if (ok) wg += (int64_t)P.job[j].gx * ((P.job[j].Nout + 16 * best_nt - 1) / (16 * best_nt));
      }
      // measured: it pays with the 64-channel tile (36 KB weight stage); with the narrower tiles the planner falls back to
      // when workgroups are scarce, the halved workgroup count costs more than the shared stage saves
      if (ok && ((wg >= w8_min && best_nt == 4) || w8_mode == 2)) { w8 = 1; best_px = 256; }
    }
  }
  for (int j = 0; j < njobs; ++j)
    if (!big_fill_job(P.job[j], g[j].x, g[j].w, g[j].y, g[j].bias, g[j].res, g[j].B, g[j].H, g[j].W, g[j].d, g[j].Cin, g[j].ldx, g[j].Nout,
                      g[j].ldy, g[j].ldr, g[j].Nw, g[j].Kw, g[j].act, g[j].alpha, g[j].out_f32, g[j].accumulate, g[j].flip, best_px))
      return 0;
  for (int j = 0; j < njobs; ++j) { P.job[j].scale = g[j].flip ? nullptr : usseg_epi_scale[j]; P.job[j].one_tap = g[j].one_tap; }
  if (usseg_tap_mask.group_ch) {
    // the tile (fwd: 16*nt output channels; dgrad: one 32-channel K chunk) must lie inside one class
    const int unit = g[0].flip ? 32 : 16 * best_nt;
    if (njobs != 1 || usseg_tap_mask.group_ch % unit) return 0;
    P.job[0].mask_ch = usseg_tap_mask.group_ch;
    for (int c = 0; c < 4; ++c) P.job[0].tapmask[c] = usseg_tap_mask.mask[c];
  }
  big_launch(P, njobs, best_nt, best_px, s, w8);
  return 1;
}

// Returns 1 and launches if the geometry fits, 0 if the caller must use another kernel.
int usseg_try_launch_conv_big(const bf16_t* x, const bf16_t* w, void* y, const float* bias, const bf16_t* res, int B, int H, int W, int d,
                              int Cin, int ldx, int Nout, int ldy, int ldr, int Nw, int Kw, int act, float alpha, int out_f32,
                              int accumulate, int flip, hipStream_t s) {
  BigGeom g = {x, w, y, bias, res, B, H, W, d, Cin, ldx, Nout, ldy, ldr, Nw, Kw, act, alpha, out_f32, accumulate, flip};
  return big_plan_and_launch(&g, 1, s);
}

int usseg_try_launch_conv_big_multi(int njobs, const UssegConvJob* jobs, int flip, hipStream_t s) {
  if (njobs < 1 || njobs > 4) return 0;
  BigGeom g[4];
  for (int j = 0; j < njobs; ++j) {
    const UssegConvJob& q = jobs[j];
    const UssegConvDesc& d = q.desc;
    const int out_f32 = (d.flags & USSEG_OUT_F32) ? 1 : 0, acc = (d.flags & USSEG_ACCUMULATE) ? 1 : 0;
    if (flip)
      g[j] = {(const bf16_t*)q.x, (const bf16_t*)q.wp, q.y, nullptr, (const bf16_t*)q.residual, d.B, d.H, d.W, d.dilation, d.Cout, d.ldy,
              d.Cin, d.ldx, q.ldr, roundup(d.Cin, 16), 9 * d.Cout, USSEG_ACT_NONE, 0.f, 0, acc, 1};
    else if (d.ksize == 1) {     // rides along as a centre-tap-only job of the 3x3 geometry (dilation 1); its operand is the 1x1 conv's own [N][Cin]
      g[j] = {(const bf16_t*)q.x, (const bf16_t*)q.wp, q.y, q.bias, (const bf16_t*)q.residual, d.B, d.H, d.W, 1, d.Cin, d.ldx,
              d.Cout, d.ldy, q.ldr, roundup(d.Cout, 16), d.Cin, d.act, d.alpha, out_f32, acc, 0};
      g[j].one_tap = 1;
    } else
      g[j] = {(const bf16_t*)q.x, (const bf16_t*)q.wp, q.y, q.bias, (const bf16_t*)q.residual, d.B, d.H, d.W, d.dilation, d.Cin, d.ldx,
              d.Cout, d.ldy, q.ldr, roundup(d.Cout, 16), 9 * d.Cin, d.act, d.alpha, out_f32, acc, 0};
  }
  return big_plan_and_launch(g, njobs, s);
}
