// Weight-gradient GEMM on bf16 MFMA for gfx950:
//   dW[t][m][n] += sum_{pixels (b,gy,gx)} A[b, gy*asy+ady_t, gx*asx+adx_t, m] * Bm[b, gy*bsy+bdy_t, gx*bsx+bdx_t, n]
// (Conv2D: A = x shifted by the tap, Bm = dy.  Conv2DTranspose: A = x, Bm = dy read with stride 2.)
// The contraction runs over PIXELS, which is the slow (row) axis of both NHWC operands, so both MFMA operands
// need a transpose: tiles are staged row-major [32 pixels][64 channels] in LDS and the fragments are read with
// ds_read_b64_tr_b16 (hardware 4x16 transpose), two reads per 8-deep K fragment.
// Grid: x = pixel split (split-K), y = (m-tile, n-tile), z = tap.  Each workgroup reduces its pixel range into a
// 64x64 fp32 tile (4 waves as 2x2, 2x2 MFMA 16x16x32 tiles each) and adds it to the fp32 scratch with atomics.
#include "common.h"

struct WgradParams {
  const bf16_t* a;
  const bf16_t* b;
  float* out;  // [ntaps][Ma][Nb]
  int32_t B, Hg, Wg;
  int64_t M;
  int32_t Ha, Wa, lda, asy, asx;
  int32_t Hb, Wb, ldb, bsy, bsx;
  int32_t Ma, Nb;
  int32_t mtiles, ntiles;
  int64_t chunk;  // pixels per split (multiple of 32)
  int16_t ady[16], adx[16], bdy[16], bdx[16];
  // batched form (attention): blockIdx.z = b1*nb2 + b2 instead of the tap; operand bases + b1*s1 + b2*s2 (elements)
  int32_t nb2;
  int64_t as1, as2, bs1, bs2, os1, os2;
  WgMap map;   // where the result goes (identity: out[t][m][n])
  float* ws;   // partial slabs [split][ntaps][Ma][Nb] (plain stores, summed + scattered by wgrad_finish_kernel) or NULL (atomics)
  int32_t ntaps;
  // wgrad_dma_kernel: workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), so with grid = (split, tile, tap) the
  // taps / channel tiles that read the SAME pixel range land in different XCDs at different times and every one of them pulls its
  // operand rows through HBM / MALL again (Arch A's 4x4 up-convs: 16 taps x (x + a quarter of dy) = 2.1 GB for 234 MB of operands,
  // the launch ran at the memory rate).  With the remap, XCD c owns a contiguous run of the (split, tile, tap) items in tap-fastest
  // order: the items of one pixel range start together on one XCD and re-use its L2.
  int32_t xcd_remap;
  int32_t lw, lh;   // log2(Wg), log2(Hg) when both are powers of two (the pixel decode of a DMA piece is then two shifts), else -1
};

__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  constexpr int BK = 32, TS = 72;  // LDS row stride in elements (128 B data + 16 B pad)
  __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][BK * TS];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int t = p.nb2 > 0 ? 0 : (int)blockIdx.z;
  const int bz1 = p.nb2 > 0 ? (int)blockIdx.z / p.nb2 : 0, bz2 = p.nb2 > 0 ? (int)blockIdx.z - bz1 * p.nb2 : 0;
  const bf16_t* const abase = p.a + bz1 * p.as1 + bz2 * p.as2;
  const bf16_t* const bbase = p.b + bz1 * p.bs1 + bz2 * p.bs2;
  const int mt = blockIdx.y / p.ntiles, nt = blockIdx.y - mt * p.ntiles;
  const int m0 = mt * 64, n0 = nt * 64;
  const int64_t k_begin = (int64_t)blockIdx.x * p.chunk;
  int64_t k_end = k_begin + p.chunk;
  if (k_end > p.M) k_end = p.M;
  if (k_begin >= k_end) return;
  const int ady = p.ady[t], adx = p.adx[t], bdy = p.bdy[t], bdx = p.bdx[t];
  const int HWg = p.Hg * p.Wg;

  const int pr = tid >> 3, cc = tid & 7;  // staging: pixel row in tile, 8-channel chunk
  const bool a_ok = (m0 + cc * 8) < p.Ma, b_ok = (n0 + cc * 8) < p.Nb;
  uint4 ra, rb;
  auto load_step = [&](int64_t kbase) {
    int64_t m = kbase + pr;
    ra = make_uint4(0, 0, 0, 0);
    rb = ra;
    if (m < k_end) {
      int mm = (int)m;
      int b = mm / HWg;
      int rem = mm - b * HWg;
      int gy = rem / p.Wg, gx = rem - gy * p.Wg;
      int ay = gy * p.asy + ady, ax = gx * p.asx + adx;
      int by = gy * p.bsy + bdy, bx = gx * p.bsx + bdx;
      bool in_a = (unsigned)ay < (unsigned)p.Ha && (unsigned)ax < (unsigned)p.Wa;
      bool in_b = (unsigned)by < (unsigned)p.Hb && (unsigned)bx < (unsigned)p.Wb;
      if (in_a && in_b) {  // a product with a zero operand contributes nothing: skip both loads
        if (a_ok) ra = *reinterpret_cast<const uint4*>(abase + ((int64_t)(b * p.Ha + ay) * p.Wa + ax) * p.lda + m0 + cc * 8);
        if (b_ok) rb = *reinterpret_cast<const uint4*>(bbase + ((int64_t)(b * p.Hb + by) * p.Wb + bx) * p.ldb + n0 + cc * 8);
      }
    }
  };
  auto store_step = [&](int buf) {
    *reinterpret_cast<uint4*>(&lds[buf][0][pr * TS + cc * 8]) = ra;
    *reinterpret_cast<uint4*>(&lds[buf][1][pr * TS + cc * 8]) = rb;
  };

  const int wm = wv >> 1, wn = wv & 1;  // wave's 32x32 sub-tile
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // transposed-read addressing: 16-lane group g reads K rows 8g..8g+7 (two reads of 4 rows); lane 4q+pp of the
  // group supplies the address of row q, columns 4pp..4pp+3 and receives column (lane&15), rows 0..3.
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  auto frag = [&](const bf16_t* tile, int col0) -> bf16x8_t {
    const bf16_t* a0 = tile + (8 * g + tq) * TS + col0 + 4 * tp;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0 + 4 * TS));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };

  load_step(k_begin);
  store_step(0);
  __syncthreads();
  int it = 0;
  for (int64_t kb = k_begin; kb < k_end; kb += BK, ++it) {
    const int cur = it & 1;
    const bool more = (kb + BK) < k_end;
    if (more) load_step(kb + BK);
    bf16x8_t af[2], bfr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      af[i] = frag(&lds[cur][0][0], wm * 32 + i * 16);
      bfr[i] = frag(&lds[cur][1][0], wn * 32 + i * 16);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    if (more) store_step(cur ^ 1);
    __syncthreads();
  }

  // D[row = m_local = 4*(lane>>4)+j][col = n_local = lane&15]
  float* out = p.out + (int64_t)t * p.Ma * p.Nb + bz1 * p.os1 + bz2 * p.os2;
  float* slab = p.ws ? p.ws + ((int64_t)blockIdx.x * p.ntaps + t) * p.Ma * p.Nb : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n = n0 + wn * 32 + j * 16 + li;
      if (n >= p.Nb) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m = m0 + wm * 32 + i * 16 + g * 4 + r;
        if (m >= p.Ma) continue;
        if (slab) slab[(int64_t)m * p.Nb + n] = acc[i][j][r];
        else {
          float* q = wg_map_dst(p.map, out, (int64_t)m * p.Nb + n, t, m, n);
          if (q) atomicAdd(q, acc[i][j][r]);
        }
      }
    }
}

// ---- LDS-DMA variant -------------------------------------------------------------------------------------------------------
// Same per-tap GEMM over pixels, but: 64-pixel K steps whose operand rows go global -> LDS directly (buffer_load ... lds, out of
// range pixels and channel tails zero-filled by the range check) into a ring of NSTAGE stages with counted vmcnt waits and bare
// barriers; (64*TM) x (64*TM) tiles (TM = 2: 16 MFMA tiles per wave per 32 pixels instead of 4 for the same fragment reads);
// pixel coordinates by a float-reciprocal division instead of two integer divisions per load.
// LDS image of an operand stage: [64 pixels][64*TM channels], dense rows; the 32-byte block b (16 channels = one MFMA tile
// column) of pixel row r sits at block b ^ f(r), f(r) = (r >> 1) & 3 for 128-byte rows, r & 7 for 256-byte rows: a transposed
// fragment read (8 consecutive pixel rows x 32 B per 32-lane pass) then covers all 64 banks once, and the lane -> (row, block)
// map of a DMA wave-instruction keeps one source channel offset per lane.
typedef __attribute__((address_space(3))) void* wg_lds_ptr_t;
#define WGRAD_OOB 0x80000000u
// One LDS-DMA piece (buffer_load_dwordx4 ... lds: 16 B per lane, 1 KB per wave) issued by inline assembly.  Issued through the
// builtin, the compiler knows that LDS is being written behind its back and - unable to tell the stage being filled from the
// stage being read - puts `s_waitcnt vmcnt(0)` in front of the first transposed fragment read (a builtin with an LDS memory
// operand) that follows: every K step then waited for the piece it had just issued, DMA and MFMAs strictly in sequence.  The
// kernel orders the ring itself (vmcnt(pieces of the younger steps) + barrier at the top of a step), so the compiler must not
// see the DMA at all.  No other vector memory LOAD may be added to the loop of a kernel that uses this: the compiler's own vmcnt
// bookkeeping does not count these pieces (the build checks it: tools/check_dma_loops.py).
typedef int wg_i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wg_i32x4_t wg_rsrc(const void* base) {   // stride 0, no range limit, raw dword addressing: as make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000)
  const uint64_t a = (uint64_t)base;
  return (wg_i32x4_t){(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
}
__device__ __forceinline__ void wg_dma16(const wg_i32x4_t rsrc, char* lds_dst, uint32_t voff) {
  const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(la), "v"(voff), "s"(rsrc) : "memory");   // (M0 is a reserved register to this compiler: it re-materialises M0 before every use of its own, and rejects it in a clobber list)
}

template <int TM, int NSTAGE>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(const WgradParams p, const float rcp_hw, const float rcp_w) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BT = 64 * TM;                 // channels per tile side
  constexpr int ROWB = BT * 2;                // bytes per pixel row of an operand stage
  constexpr int OPB = 64 * ROWB;              // bytes per operand stage (64 pixels)
  constexpr int STAGE = 2 * OPB;
  constexpr int RPI = 1024 / ROWB;            // pixel rows per DMA instruction (8 or 4)
  constexpr int IT = 64 / RPI / 4;            // instructions per wave per operand per step (2 or 4)
  constexpr int NI = 2 * IT;
  constexpr int D = NSTAGE - 1;
  constexpr int MT = 2 * TM;                  // MFMA tiles per wave per side
  extern __shared__ __attribute__((aligned(1024))) char lds_raw[];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // batched form (attention): blockIdx.z = b1*nb2 + b2 selects the operand / output bases instead of the tap
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;    // (split, tile, tap)
  if (p.xcd_remap) {     // dispatch order L -> XCD L % 8, slot L / 8 -> item (split-major, tile, tap-fastest) of that XCD's contiguous run
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, total = gx * gy * gz;
    const int L = bx + gx * (by + gy * bz);
    const int c = L & 7, q = total >> 3, r = total & 7;
    const int item = c * q + (c < r ? c : r) + (L >> 3);
    const int u = __builtin_amdgcn_readfirstlane(fdiv(item, fdiv_rcp(gz)));     // (the launcher keeps remapped grids below 2^20 workgroups)
    bz = item - u * gz;
    bx = __builtin_amdgcn_readfirstlane(fdiv(u, fdiv_rcp(gy)));
    by = u - bx * gy;
  }
  const int t = p.nb2 > 0 ? 0 : bz;
  const int bz1 = p.nb2 > 0 ? bz / p.nb2 : 0, bz2 = p.nb2 > 0 ? bz - bz1 * p.nb2 : 0;
  const int mt = by / p.ntiles, nt = by - mt * p.ntiles;
  const int m0 = mt * BT, n0 = nt * BT;
  const int64_t k_begin = (int64_t)bx * p.chunk;
  int64_t k_end = k_begin + p.chunk;
  if (k_end > p.M) k_end = p.M;
  if (k_begin >= k_end) return;
  const int nks = (int)((k_end - k_begin + 63) >> 6);
  const int ady = p.ady[t], adx = p.adx[t], bdy = p.bdy[t], bdx = p.bdx[t];
  const int HWg = p.Hg * p.Wg;

  const wg_i32x4_t ra = wg_rsrc(p.a + bz1 * p.as1 + bz2 * p.as2);
  const wg_i32x4_t rb = wg_rsrc(p.b + bz1 * p.bs1 + bz2 * p.bs2);

  // ---- this lane's DMA pieces: row lrow of instruction wv + 4*it, one fixed 16-byte channel slot
  const int lrow = TM == 1 ? (lane >> 3) : (lane >> 4);
  const int slot = TM == 1 ? (lane & 7) : (lane & 15);                       // 16-byte slot within the row
  const int fkey = TM == 1 ? ((lane >> 4) & 3) : ((4 * (wv & 1) + (lane >> 4)) & 7);   // f(row) of every row this lane touches
  const int ch = (((slot >> 1) ^ fkey) << 4) + ((slot & 1) << 3);             // source channel of that slot
  const bool a_cok = (m0 + ch) < p.Ma, b_cok = (n0 + ch) < p.Nb;
  const int a_c = m0 + ch, b_c = n0 + ch;

  // (uniform) both operands are read at the grid pixel itself: the per-piece pixel decode (two reciprocal divisions with their
  // corrections, four range checks, two address polynomials: ~40 VALU instructions per piece, 4 pieces per wave and step at the
  // 128-channel tile - more issue time than the step's 32 MFMAs) collapses to one multiply-add per operand
  const bool linear = p.asy == 1 && p.asx == 1 && p.bsy == 1 && p.bsx == 1 && ady == 0 && adx == 0 && bdy == 0 && bdx == 0 &&
                      p.Ha == p.Hg && p.Wa == p.Wg && p.Hb == p.Hg && p.Wb == p.Wg;
  int kpos = 0;   // pixel offset (within the split) of the step being issued
  // linear path: the byte offsets of this lane's pieces advance by a constant per K step (the per-piece form cost two quarter-rate
  // v_mul_lo_u32, a 64-bit compare and an exec-mask region per operand: ~34 issue slots per piece pair against 6 here)
  int lin_m[IT];
  uint32_t lin_a[IT], lin_b[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    lin_m[it] = (int)k_begin + RPI * (wv + 4 * it) + lrow;                    // M < 2^24 (launcher)
    lin_a[it] = ((uint32_t)lin_m[it] * (uint32_t)p.lda + (uint32_t)a_c) * 2u;
    lin_b[it] = ((uint32_t)lin_m[it] * (uint32_t)p.ldb + (uint32_t)b_c) * 2u;
  }
  const uint32_t lin_sa = 128u * (uint32_t)p.lda, lin_sb = 128u * (uint32_t)p.ldb;   // 64 pixels x 2 bytes
  const int k_end_i = (int)k_end;
  auto issue = [&](int stage) {
    char* const sa = lds_raw + stage * STAGE;
    char* const sb = sa + OPB;
    if (linear) {   // dense layers / 1x1 at stride 1: grid pixel == operand pixel for both operands, nothing out of range
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int j = wv + 4 * it;
        const bool pv = lin_m[it] < k_end_i;
        wg_dma16(ra, sa + j * 1024, (pv & a_cok) ? lin_a[it] : WGRAD_OOB);
        wg_dma16(rb, sb + j * 1024, (pv & b_cok) ? lin_b[it] : WGRAD_OOB);
        lin_m[it] += 64; lin_a[it] += lin_sa; lin_b[it] += lin_sb;
      }
      kpos += 64;
      return;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int j = wv + 4 * it;
      const int64_t m = k_begin + kpos + RPI * j + lrow;
      const bool pv = m < k_end;
      const int mm = pv ? (int)m : 0;
      // (image, row, column) of grid pixel mm: float-reciprocal quotients corrected by one step either way (mm < 2^24)
      // 24-bit multiplies (v_mul_i32_i24 / v_mad_i32_i24: full rate; v_mul_lo_u32 issues at a quarter of it and there were twelve
      // per piece - more issue time than the step's MFMAs): every factor here is a pixel index or a count below 2^23 (the launcher
      // checks the pixel counts), the last products are taken modulo 2^32 like the 32-bit form
      int b, gy, gx;
      if (p.lw >= 0) {     // (uniform) power-of-two grid: every feature map of the models here
        gx = mm & (p.Wg - 1);
        gy = (mm >> p.lw) & (p.Hg - 1);
        b = mm >> (p.lw + p.lh);
      } else {
        b = (int)((float)mm * rcp_hw);
        int rem = mm - __mul24(b, HWg);
        if (rem < 0) { --b; rem += HWg; }
        if (rem >= HWg) { ++b; rem -= HWg; }
        gy = (int)((float)rem * rcp_w);
        gx = rem - __mul24(gy, p.Wg);
        if (gx < 0) { --gy; gx += p.Wg; }
        if (gx >= p.Wg) { ++gy; gx -= p.Wg; }
      }
      const int ay = __mul24(gy, p.asy) + ady, ax = __mul24(gx, p.asx) + adx;
      const int by = __mul24(gy, p.bsy) + bdy, bx = __mul24(gx, p.bsx) + bdx;
      const bool in_a = ((unsigned)ay < (unsigned)p.Ha) & ((unsigned)ax < (unsigned)p.Wa);
      const bool in_b = ((unsigned)by < (unsigned)p.Hb) & ((unsigned)bx < (unsigned)p.Wb);
      const bool oka = pv & in_a & in_b & a_cok;   // a product with a zero operand contributes nothing: one zero suffices
      const bool okb = pv & in_b & b_cok;
      const uint32_t offa = (__umul24((uint32_t)(__mul24(__mul24(b, p.Ha) + ay, p.Wa) + ax), (uint32_t)p.lda) + (uint32_t)a_c) * 2u;
      const uint32_t offb = (__umul24((uint32_t)(__mul24(__mul24(b, p.Hb) + by, p.Wb) + bx), (uint32_t)p.ldb) + (uint32_t)b_c) * 2u;
      wg_dma16(ra, sa + j * 1024, oka ? offa : WGRAD_OOB);
      wg_dma16(rb, sb + j * 1024, okb ? offb : WGRAD_OOB);
    }
    kpos += 64;
  };

  const int wm = wv >> 1, wn = wv & 1;
  f32x4_t acc[MT][MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // transposed fragment reads: K index 8g + 4h + tq of a 32-pixel sub-step is pixel 16h + 4g + tq (same permutation for both
  // operands); the lane supplies the address of 4 channels (8 B) of that pixel and receives 4 pixels of channel (lane & 15)
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  auto tr8 = [&](const char* a0, const char* a1) -> bf16x8_t {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a1));
    s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  // byte offset of (pixel row r, MFMA tile column blk) within an operand stage, this lane's 8 bytes of it
  auto frag_off = [&](int r, int blk) -> int {
    const int f = TM == 1 ? ((r >> 1) & 3) : (r & 7);
    return r * ROWB + ((blk ^ f) << 5) + (tp << 3);
  };

#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nks) issue(s);
  for (int ks = 0; ks < nks; ++ks) {
    // stage ks has landed once only the pieces of the younger ISSUED stages are outstanding: min(D-1, nks-1-ks) of them
    static_assert(D <= 3, "the wait ladder below covers up to three stages in flight");
    if (D >= 3 && ks + 2 < nks) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NI) : "memory");
    else if (D >= 2 && ks + 1 < nks) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ks + D < nks) issue((ks + D) % NSTAGE);
    const char* const LA = lds_raw + (ks % NSTAGE) * STAGE;
    const char* const LB = LA + OPB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int r0 = 32 * kk + 4 * g + tq, r1 = r0 + 16;
      bf16x8_t af[MT], bfr[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        af[i] = tr8(LA + frag_off(r0, wm * MT + i), LA + frag_off(r1, wm * MT + i));
        bfr[i] = tr8(LB + frag_off(r0, wn * MT + i), LB + frag_off(r1, wn * MT + i));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // D^T: lane holds 4 consecutive n
    }
  }

  // lane holds D[n_local = 4g + r][m_local = li] of each tile: 16 bytes of the slab, or four atomics into the (mapped) gradient
  float* out = p.out + (int64_t)t * p.Ma * p.Nb + bz1 * p.os1 + bz2 * p.os2;
  float* slab = p.ws ? p.ws + ((int64_t)bx * p.ntaps + t) * p.Ma * p.Nb : nullptr;
  const bool vec_ok = (p.Nb & 3) == 0;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m0 + (wm * MT + i) * 16 + li;
      const int n = n0 + (wn * MT + j) * 16 + g * 4;
      if (m >= p.Ma || n >= p.Nb) continue;
      if (slab && vec_ok && n + 3 < p.Nb) {
        *reinterpret_cast<float4*>(slab + (int64_t)m * p.Nb + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= p.Nb) continue;
          if (slab) slab[(int64_t)m * p.Nb + n + r] = acc[i][j][r];
          else {
            float* q = wg_map_dst(p.map, out, (int64_t)m * p.Nb + n + r, t, m, n + r);
            if (q) atomicAdd(q, acc[i][j][r]);
          }
        }
      }
    }
#endif
}

template <int TM, int NSTAGE>
static void wgrad_dma_launch_t(const WgradParams& p, dim3 grid, hipStream_t s) {
  const size_t dyn = (size_t)NSTAGE * 2 * 64 * 64 * TM * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)wgrad_dma_kernel<TM, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((wgrad_dma_kernel<TM, NSTAGE>), grid, dim3(256), dyn, s, p, 1.0f / (float)(p.Hg * p.Wg), 1.0f / (float)p.Wg);
}

static int launch_wgrad(WgradParams& p, int ntaps, float* ws, int64_t ws_floats, hipStream_t s) {
  // LDS-DMA variant: unbatched launches whose operands fit 32-bit byte offsets and whose pixel count fits the float division
  static const int dma_env = getenv("USSEG_WGRAD_DMA") ? atoi(getenv("USSEG_WGRAD_DMA")) : 1;
  const bool dma = dma_env && p.nb2 <= 0 && p.M < (1 << 24) && (int64_t)p.Hg * p.Wg < (1 << 23) && (int64_t)p.B * p.Ha * p.Wa < (1 << 24) && (int64_t)p.B * p.Hb * p.Wb < (1 << 24) &&   // (24-bit multiplies in the pixel decode)
                   (int64_t)p.B * p.Ha * p.Wa * p.lda * 2 < 0x7fff0000ll &&
                   (int64_t)p.B * p.Hb * p.Wb * p.ldb * 2 < 0x7fff0000ll && p.lda % 8 == 0 && p.ldb % 8 == 0;
  // 128x128 tiles when both channel counts exceed one 64-wide tile (split-K supplies the workgroups if the tiles are few)
  static const int tm_env = getenv("USSEG_WGRAD_TM") ? atoi(getenv("USSEG_WGRAD_TM")) : 0;
  int tm = 1;
  if (dma && p.Ma > 64 && p.Nb > 64) tm = 2;
  if (dma && tm_env) tm = tm_env;
  const int bt = 64 * tm;
  p.mtiles = (p.Ma + bt - 1) / bt;
  p.ntiles = (p.Nb + bt - 1) / bt;
  p.ntaps = ntaps;
  int64_t tiles = (int64_t)p.mtiles * p.ntiles * ntaps;
  // enough splits to give every CU a few workgroups, but at least 256 pixels of work per split
  // swept on the four bench configurations (target 2048/1024/512/384/256/128 x slab cap 8/4/2): the round-1 setting (2048, 8) made
  // 16-32 splits of the big dense gradients of the ViT / Swin layers, whose slabs then dominate (wgrad_finish 580 us per cfg4 step);
  // (512, 4): Arch B 3.735 -> 3.70 ms, cfg4 11.52 -> 11.11, cfg5 12.44 -> 11.87, Arch A unchanged; 128 is 2-10 % slower everywhere
  static const int wg_target = getenv("USSEG_WGD_TARGET") ? atoi(getenv("USSEG_WGD_TARGET")) : 512;
  static const int slab_cap = getenv("USSEG_WGD_CAP") ? atoi(getenv("USSEG_WGD_CAP")) : 4;
  int64_t want = (wg_target + tiles - 1) / tiles;
  int64_t max_splits = cdiv64(p.M, 256);
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  // With a workspace every split stores its partial [ntaps][Ma][Nb] slab and wgrad_finish_kernel sums (and scatters)
  // them: fp32 atomics from ~1000 workgroups into the few cache lines of a small gradient serialise at the L2 (the
  // 32x16 cardinal conv1 gradient took 35-120 us that way).  Slab traffic is kept within ~4x the operand bytes (USSEG_WGD_CAP).
  const int64_t slab = (int64_t)ntaps * p.Ma * p.Nb;
  p.ws = nullptr;
  int64_t cap_splits = 1;
  if (ws) ws = usseg_defer_wgrad_ws(s, ws, ws_floats, &ws_floats);
  if (ws && want > 1) {
    const int64_t in_bytes = p.M * (p.Ma + p.Nb) * 2;
    int64_t cap = slab_cap * in_bytes / (slab * 4);
    if (cap < 4) cap = 4;
    if (cap > ws_floats / slab) cap = ws_floats / slab;
    if (want > cap) want = cap;
    cap_splits = cap;
    if (want > 1) p.ws = ws;
  }
  if (want < 1) want = 1;
  {
    // wave quantisation (see wgrad_halo_launch): 3 (64x64 tiles) or 2 (128x128) workgroups per CU resident; near the target pick
    // the split count with the cheapest rounds x (K steps per workgroup + fixed cost) estimate
    static const int quant = getenv("USSEG_WG_QUANT") ? atoi(getenv("USSEG_WG_QUANT")) : 1;
    const int64_t hi_cap = p.ws ? (want * 2 < cap_splits ? want * 2 : cap_splits) : want;   // within the slab traffic cap; without slabs (atomics) never more splits
    if (quant && dma && want > 1) {
      const int64_t slots = 256 * (tm == 2 ? 2 : 3);
      int64_t best = want;
      double best_cost = 1e30;
      for (int64_t sp = want / 2 > 1 ? want / 2 : 1; sp <= hi_cap && sp <= cdiv64(p.M, 256); ++sp) {
        if (p.ws && sp * slab > ws_floats) break;
        const int64_t ch = cdiv64(cdiv64(p.M, sp), 64) * 64, real = cdiv64(p.M, ch);
        const double rounds = (double)((tiles * real + slots - 1) / slots);
        const double cost = rounds * ((double)(ch / 64) + 8.0) + 0.02 * (double)real;
        if (cost < best_cost) { best_cost = cost; best = sp; }
      }
      want = best;
    }
  }
  p.chunk = cdiv64(cdiv64(p.M, want), 64) * 64;
  int64_t splits = cdiv64(p.M, p.chunk);
  const int slot = usseg_prof_start(2, s);
  const dim3 grid((unsigned)splits, (unsigned)(p.mtiles * p.ntiles), (unsigned)ntaps);
  static const int dbg = getenv("USSEG_WGRAD_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "[wgrad] M %lld (B %d Hg %d Wg %d) Ma %d Nb %d taps %d tm %d splits %lld chunk %lld slab %s asy %d bsy %d lda %d ldb %d\n", (long long)p.M, p.B, p.Hg, p.Wg,
                   p.Ma, p.Nb, ntaps, tm, (long long)splits, (long long)p.chunk, p.ws ? "yes" : "no", p.asy, p.bsy, p.lda, p.ldb);
  {
    static const int pow2_env = getenv("USSEG_WGRAD_POW2") ? atoi(getenv("USSEG_WGRAD_POW2")) : 1;
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    p.lw = pow2_env ? lg(p.Wg) : -1;
    p.lh = pow2_env ? lg(p.Hg) : -1;
    if (p.lw < 0 || p.lh < 0) p.lw = p.lh = -1;
  }
  static const int xcd_env = getenv("USSEG_WGRAD_XCD") ? atoi(getenv("USSEG_WGRAD_XCD")) : 1;
  p.xcd_remap = dma && xcd_env && p.nb2 <= 0 && (int64_t)splits * p.mtiles * p.ntiles * ntaps < (1ll << 20) && (p.mtiles * p.ntiles * ntaps > 1);
  if (dma) {
    if (tm == 2) wgrad_dma_launch_t<2, 2>(p, grid, s);
    else wgrad_dma_launch_t<1, 3>(p, grid, s);
  } else hipLaunchKernelGGL(wgrad_kernel, grid, dim3(256), 0, s, p);
  if (p.ws) usseg_launch_wgrad_finish(p.ws, (int)splits, slab, p.out, p.map, p.Ma, p.Nb, s);
  usseg_prof_stop(2, slot, s);
  return usseg_check_launch("wgrad");
}

static int conv2d_wgrad_impl(const UssegConvDesc* d, const void* x, const void* dy, float* dw, const UssegWgradDst* dst, float* ws,
                             int64_t ws_floats, usseg_stream_t stream) {
  USSEG_CHECK_ARG(d && x && dy && (dw || dst), "null pointer");
  WgMap map;
  USSEG_CHECK_ARG(wg_map_fill(map, dst), "wgrad: bad destination map (1..4 blocks, non-null dst)");
  USSEG_CHECK_ARG(d->Cin % 8 == 0 && d->Cout % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0, "channels must be multiples of 8");
  USSEG_CHECK_ARG(d->ksize == 1 || d->ksize == 3, "conv ksize must be 1 or 3");
  if (d->ksize == 3 && usseg_try_launch_wgrad_halo((const bf16_t*)x, (const bf16_t*)dy, dw, map, d->B, d->H, d->W, d->dilation, d->Cin,
                                                   d->Cout, d->ldx, d->ldy, ws, ws_floats, (hipStream_t)stream))
    return usseg_check_launch("wgrad_halo");
  WgradParams p = {};
  p.map = map;
  p.a = (const bf16_t*)x; p.b = (const bf16_t*)dy; p.out = dw;
  p.B = d->B; p.Hg = d->H; p.Wg = d->W; p.M = (int64_t)d->B * d->H * d->W;
  p.Ha = d->H; p.Wa = d->W; p.lda = d->ldx; p.asy = p.asx = 1;
  p.Hb = d->H; p.Wb = d->W; p.ldb = d->ldy; p.bsy = p.bsx = 1;
  p.Ma = d->Cin; p.Nb = d->Cout;
  const int k = d->ksize, half = k / 2;
  for (int kh = 0; kh < k; ++kh)
    for (int kw = 0; kw < k; ++kw) {
      int t = kh * k + kw;
      p.ady[t] = (int16_t)((kh - half) * d->dilation); p.adx[t] = (int16_t)((kw - half) * d->dilation);
      p.bdy[t] = 0; p.bdx[t] = 0;
    }
  return launch_wgrad(p, k * k, ws, ws_floats, (hipStream_t)stream);
}

extern "C" int usseg_conv2d_wgrad(const UssegConvDesc* d, const void* x, const void* dy, float* dw, float* ws, int64_t ws_floats,
                                  usseg_stream_t stream) {
  USSEG_CHECK_ARG(dw, "null pointer");
  return conv2d_wgrad_impl(d, x, dy, dw, nullptr, ws, ws_floats, stream);
}
extern "C" int usseg_conv2d_wgrad_mapped(const UssegConvDesc* d, const void* x, const void* dy, const UssegWgradDst* dst, float* ws,
                                         int64_t ws_floats, usseg_stream_t stream) {
  USSEG_CHECK_ARG(dst, "null destination map");
  return conv2d_wgrad_impl(d, x, dy, nullptr, dst, ws, ws_floats, stream);
}

extern "C" int usseg_conv2d_wgrad_multi(int32_t njobs, const UssegWgradJob* jobs, float* ws, int64_t ws_floats, usseg_stream_t stream) {
  USSEG_CHECK_ARG(jobs && njobs >= 1 && njobs <= 4, "wgrad multi: 1 <= njobs <= 4");
  for (int j = 0; j < njobs; ++j) USSEG_CHECK_ARG(jobs[j].x && jobs[j].dy && (jobs[j].dw || jobs[j].dst), "wgrad multi: null pointer");
  if (njobs > 1 && usseg_try_launch_wgrad_halo_multi(njobs, jobs, ws, ws_floats, (hipStream_t)stream)) return usseg_check_launch("wgrad_halo_multi");
  for (int j = 0; j < njobs; ++j) {
    int rc = conv2d_wgrad_impl(&jobs[j].desc, jobs[j].x, jobs[j].dy, jobs[j].dst ? nullptr : jobs[j].dw, jobs[j].dst, ws, ws_floats, stream);
    if (rc) return rc;
  }
  return USSEG_OK;
}

static int tconv2d_wgrad_impl(const UssegConvDesc* d, const void* x, const void* dy, float* dw, const UssegWgradDst* dst, float* ws,
                              int64_t ws_floats, usseg_stream_t stream) {
  USSEG_CHECK_ARG(d && x && dy && (dw || dst), "null pointer");
  WgMap map;
  USSEG_CHECK_ARG(wg_map_fill(map, dst), "wgrad: bad destination map (1..4 blocks, non-null dst)");
  USSEG_CHECK_ARG(d->Cin % 8 == 0 && d->Cout % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0, "channels must be multiples of 8");
  USSEG_CHECK_ARG(d->ksize == 3 || d->ksize == 4, "tconv ksize must be 3 or 4");
  const int k = d->ksize, pad = (k == 4) ? 1 : 0;
  WgradParams p = {};
  p.B = d->B; p.Hg = d->H; p.Wg = d->W; p.M = (int64_t)d->B * d->H * d->W;
  if (!dst) {
    p.a = (const bf16_t*)x; p.b = (const bf16_t*)dy; p.out = dw;
    p.Ha = d->H; p.Wa = d->W; p.lda = d->ldx; p.asy = p.asx = 1;
    p.Hb = 2 * d->H; p.Wb = 2 * d->W; p.ldb = d->ldy; p.bsy = p.bsx = 2;
    p.Ma = d->Cin; p.Nb = d->Cout;
    for (int kh = 0; kh < k; ++kh)
      for (int kw = 0; kw < k; ++kw) {
        int t = kh * k + kw;
        p.ady[t] = 0; p.adx[t] = 0;
        p.bdy[t] = (int16_t)(kh - pad); p.bdx[t] = (int16_t)(kw - pad);
      }
  } else {
    // Mapped: the Keras variable is [k,k,Cout,Cin] (input channel fastest).  The kernel's lanes run along its n axis, so
    // the operands swap roles (A = dy, B = x): atomics of one wave-instruction then land on consecutive addresses.
    p.a = (const bf16_t*)dy; p.b = (const bf16_t*)x; p.out = nullptr;
    p.Ha = 2 * d->H; p.Wa = 2 * d->W; p.lda = d->ldy; p.asy = p.asx = 2;
    p.Hb = d->H; p.Wb = d->W; p.ldb = d->ldx; p.bsy = p.bsx = 1;
    p.Ma = d->Cout; p.Nb = d->Cin;
    for (int kh = 0; kh < k; ++kh)
      for (int kw = 0; kw < k; ++kw) {
        int t = kh * k + kw;
        p.ady[t] = (int16_t)(kh - pad); p.adx[t] = (int16_t)(kw - pad);
        p.bdy[t] = 0; p.bdx[t] = 0;
      }
    for (int b = 0; b < map.nblocks; ++b) {   // kernel axes (m, n) = (output channel, input channel)
      WgBlock u = map.blk[b];
      map.blk[b].sI = u.sO; map.blk[b].sO = u.sI;
      map.blk[b].i_off = u.o_off; map.blk[b].o_off = u.i_off;
      map.blk[b].ni = u.no; map.blk[b].no = u.ni;
    }
    if (usseg_try_launch_tconv_wgrad_halo((const bf16_t*)x, (const bf16_t*)dy, map, d->B, d->H, d->W, d->Cin, d->Cout, d->ldx, d->ldy, k, ws,
                                          ws_floats, (hipStream_t)stream))
      return usseg_check_launch("tconv_wgrad_halo");
  }
  p.map = map;
  return launch_wgrad(p, k * k, ws, ws_floats, (hipStream_t)stream);
}
extern "C" int usseg_tconv2d_wgrad(const UssegConvDesc* d, const void* x, const void* dy, float* dw, usseg_stream_t stream) {
  USSEG_CHECK_ARG(dw, "null pointer");
  return tconv2d_wgrad_impl(d, x, dy, dw, nullptr, nullptr, 0, stream);
}
extern "C" int usseg_tconv2d_wgrad_mapped(const UssegConvDesc* d, const void* x, const void* dy, const UssegWgradDst* dst, float* ws,
                                          int64_t ws_floats, usseg_stream_t stream) {
  USSEG_CHECK_ARG(dst, "null destination map");
  return tconv2d_wgrad_impl(d, x, dy, nullptr, dst, ws, ws_floats, stream);
}

// ---- batched "TN" GEMM: OUT[b1,b2][m][n] += sum_r A[b1,b2][r][m] * B[b1,b2][r][n]  (contraction over ROWS of both operands,
// fp32 atomics into a zeroed buffer).  Attention backward: dV = P^T dO, dK = dS^T Q (VisionTransformer.py:41-47 transposed).
extern "C" int usseg_gemm_tn_batched(const UssegGemmDesc* d, const void* a, const void* b, float* out, usseg_stream_t stream) {
  USSEG_CHECK_ARG(d && a && b && out, "null pointer");
  USSEG_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0 && d->M % 8 == 0 && d->N % 8 == 0 && d->ldx % 8 == 0 && d->ldw % 8 == 0,
                  "gemm_tn: M, N, lda, ldb must be multiples of 8");
  USSEG_CHECK_ARG(d->nb1 > 0 && d->nb2 > 0 && (int64_t)d->nb1 * d->nb2 < 65536, "gemm_tn: bad batch");
  WgradParams p = {};
  p.a = (const bf16_t*)a; p.b = (const bf16_t*)b; p.out = out;
  p.B = 1; p.Hg = 1; p.Wg = d->K; p.M = d->K;            // contraction length = rows
  p.Ha = 1; p.Wa = d->K; p.lda = d->ldx; p.asy = p.asx = 1;
  p.Hb = 1; p.Wb = d->K; p.ldb = d->ldw; p.bsy = p.bsx = 1;
  p.Ma = d->M; p.Nb = d->N;
  p.nb2 = d->nb2; p.as1 = d->xs1; p.as2 = d->xs2; p.bs1 = d->ws1; p.bs2 = d->ws2; p.os1 = d->ys1; p.os2 = d->ys2;
  p.mtiles = (p.Ma + 63) / 64;
  p.ntiles = (p.Nb + 63) / 64;
  p.chunk = ((d->K + 63) / 64) * 64;                     // one split: the batch already fills the chip
  p.ntaps = 1;
  const int slot = usseg_prof_start(2, (hipStream_t)stream);
  static const int dma_env = getenv("USSEG_WGRAD_DMA") ? atoi(getenv("USSEG_WGRAD_DMA")) : 1;
  const bool dma = dma_env && d->K < (1 << 23) && (int64_t)d->K * d->ldx * 2 < 0x7fff0000ll && (int64_t)d->K * d->ldw * 2 < 0x7fff0000ll &&
                   (d->xs1 | d->xs2 | d->ws1 | d->ws2) % 8 == 0;
  const dim3 grid(1, (unsigned)(p.mtiles * p.ntiles), (unsigned)(d->nb1 * d->nb2));
  if (dma) wgrad_dma_launch_t<1, 3>(p, grid, (hipStream_t)stream);
  else hipLaunchKernelGGL(wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  usseg_prof_stop(2, slot, (hipStream_t)stream);
  return usseg_check_launch("gemm_tn_batched");
}
