// First encoder layer on the uint8-resident image store (reference contract utils/load_data/base_load_data.py:39-40:
// pixels are k/255, and the exemplar rows of models/BaseModel.py:247 are read un-binarised, i.e. grey).
//
// The store keeps a pixel as the byte k (value = k * x_scale, x_scale = 1/255): 4x less HBM than fp32 rows.  A byte is an
// integer <= 255, hence EXACT in bf16 (8 significant bits), and an fp32 weight splits exactly into three bf16 terms
// w = w0 + w1 + w2 (24 = 3 x 8 significant bits).  So  sum_k x_k w_k = x_scale * sum_k k (w0 + w1 + w2)  runs as three
// bf16 products per element on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- every product k * w_j is exact in fp32
// (16 significant bits), the rounding that is left is the fp32 accumulation, as in an fp32 GEMM -- at 16/3 times the rate
// of the fp32 matrix pipe (v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate).  Not a reduced-precision path: tests
// hold it against the fp64 oracle at the fp32 kernel's bar.
//
//   evae_dense_u8_prepare   weights [N x K] fp32 (h and g banks) -> the three bf16 terms, laid out as the LDS images of the
//                           GEMM's B tiles (one 24 KB image per (column tile, K-slab): a block stages it with straight
//                           16-byte copies)
//   evae_gated_dense_fwd_u8 out = (x W_h^T + b_h) * sigmoid(x W_g^T + b_g) for M gathered rows of the byte store
//
// Kernel: block = 256 threads = 4 waves (2 x 2), block tile 128 rows x 64 gated outputs (= 128 MFMA columns [h | g]), wave
// tile 64 x (32 h + 32 g): the gate meets its h in the same lane.  K-slab = 32 bytes of a row.  LDS rows are 64 B with the
// four 16-byte slots XOR-swizzled by (row >> 2) & 3, so every ds_read_b128 fragment read and every staging write is
// conflict-free without padding; two stages of (A 8 KB + B 24 KB) = 64 KB per block, two blocks per CU.
#include "evae_common.h"
#include "evae_u8_prepare.h"
#include "evae_p6_image.h"
#include "evae_gemm_kernel.h"

namespace evae {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16u __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));    // native vector (HIP's uint4 struct does not always stay in registers)

constexpr int U8_BM = 128, U8_NT = 256;        // U8_BN = 64, U8_BK = 32: evae_u8_prepare.h
constexpr int U8_A_BYTES = U8_BM * 64;                 // 128 rows x 32 bf16
constexpr int U8_B_BYTES = 3 * 128 * 64;               // 3 terms x 128 columns x 32 bf16
constexpr int U8_STAGE = U8_A_BYTES + U8_B_BYTES;      // 32 KB


// B-tile images (u8_prepare_element, evae_u8_prepare.h): one thread per (tn, s, c, k) element, all three terms
__global__ void u8_prepare_kernel(const float* __restrict__ wh, const float* __restrict__ wg, int N, int K, int nslab,
                                  unsigned short* __restrict__ img) {
  u8_prepare_element((size_t)blockIdx.x * blockDim.x + threadIdx.x, wh, wg, N, K, nslab, img);
}

__device__ __forceinline__ bf16x8 lds_read16(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

// 16 bytes -> 16 bf16 (two 16-byte LDS slots).  A byte k as bf16 is the upper half of float(k): exact.
__device__ __forceinline__ void u8x16_to_bf16(const u32x4 v, u32x4& lo, u32x4& hi) {
  auto cvt4 = [](unsigned w, unsigned& p0, unsigned& p1) {
    const unsigned f0 = __float_as_uint((float)(w & 0xFFu)), f1 = __float_as_uint((float)((w >> 8) & 0xFFu));
    const unsigned f2 = __float_as_uint((float)((w >> 16) & 0xFFu)), f3 = __float_as_uint((float)(w >> 24));
    p0 = (f0 >> 16) | (f1 & 0xFFFF0000u);
    p1 = (f2 >> 16) | (f3 & 0xFFFF0000u);
  };
  unsigned a, b;
  cvt4(v[0], a, b); lo[0] = a; lo[1] = b;
  cvt4(v[1], a, b); lo[2] = a; lo[3] = b;
  cvt4(v[2], a, b); hi[0] = a; hi[1] = b;
  cvt4(v[3], a, b); hi[2] = a; hi[3] = b;
}

// GATED: the forward (columns = [h | g] pairs, gate epilogue).  !GATED: a plain product into split-K partial planes
// part[z][M][N] -- the weight gradient runs through it with the roles swapped (rows = pixels of the gathered, transposed
// byte rows; columns = the three-term split of dy; contraction over the batch rows).
template <bool GATED>
__global__ __launch_bounds__(U8_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void u8_gemm_kernel(
    const unsigned char* __restrict__ x, const int64_t* __restrict__ rows, int M, long long ldx, float x_scale,
    const unsigned short* __restrict__ img, int nslab_total, int ksplit, const float* __restrict__ bh,
    const float* __restrict__ bg, int N, float* __restrict__ out, float* __restrict__ save_s, int tiles_m, int tiles_n,
    const P6Sink tsink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn, zs = 0;
  if constexpr (GATED) {
    // XCD-aware bijective remap (as the fp32 GEMM): XCD x = id % 8 works on a contiguous run of tiles
    const int ntiles = tiles_m * tiles_n;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int qq = ntiles >> 3, rr = ntiles & 7;
    const int tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
    tm = tile / tiles_n; tn = tile - tm * tiles_n;
  } else {
    // Split-K weight gradient: the tiles_m row tiles (pixels) of one UNIT = (contraction slice z, column tile tn) read the same
    // three-term dy^T images -- the heavy operand, 6 bytes per element -- so a unit's blocks sit on ONE XCD, dispatched
    // back to back (ids 8 j + x -> XCD x): they walk the slice in step and all but the first find every image slab in that
    // XCD's L2.  Units are ordered z-major and dealt to the XCDs in contiguous runs, so the column tiles of a slice (which
    // share the byte rows) mostly meet on one XCD too.  (r02: every (row tile x column tile x slice) block streamed both
    // operands past its XCD's L2: 710 MB read per launch against 110 MB of operands.)
    const int nunits = ((nslab_total + ksplit - 1) / ksplit) * tiles_n;      // slices x column tiles; the grid is 1-D
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int qq = nunits >> 3, rr = nunits & 7;
    const int ul = slot / tiles_m;
    if (ul >= qq + (xcd < rr ? 1 : 0)) return;                // this XCD has one unit less than the grid allows for
    const int u = xcd * qq + (xcd < rr ? xcd : rr) + ul;
    tm = slot - ul * tiles_m;
    zs = u / tiles_n; tn = u - zs * tiles_n;
  }
  const int m0 = tm * U8_BM, n0 = tn * (GATED ? U8_BN : 2 * U8_BN);
  const int s_begin = zs * ksplit;
  const int s_end = (s_begin + ksplit < nslab_total) ? s_begin + ksplit : nslab_total;
  const int nslab = s_end - s_begin;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, lh = lane >> 5;

  // staging roles: A -- thread = (row tid >> 1, 16-byte half tid & 1); B -- six 16-byte chunks of the 24 KB image
  const int arow = tid >> 1, ahalf = tid & 1;
  const int am = m0 + arow;
  const size_t arow_g = rows ? (size_t)rows[am < M ? am : m0] : (size_t)(am < M ? am : m0);
  const unsigned char* ap = x + arow_g * ldx + ahalf * 16 + (size_t)s_begin * U8_BK;
  const unsigned short* bimg = img + ((size_t)tn * nslab_total + s_begin) * (3 * 128 * 32);
  const int a_off0 = arow * 64 + (((2 * ahalf) ^ ((arow >> 2) & 3)) << 4);
  const int a_off1 = arow * 64 + (((2 * ahalf + 1) ^ ((arow >> 2) & 3)) << 4);

  f32x16u acc[2][2];        // [mt][h | g]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 ra, rb0, rb1, rb2, rb3, rb4, rb5;
  auto load = [&](int s) {
    ra = *reinterpret_cast<const u32x4*>(ap + (size_t)s * U8_BK);
    const u32x4* bsrc = reinterpret_cast<const u32x4*>(bimg + (size_t)s * (3 * 128 * 32)) + tid;
    rb0 = bsrc[0]; rb1 = bsrc[U8_NT]; rb2 = bsrc[2 * U8_NT]; rb3 = bsrc[3 * U8_NT]; rb4 = bsrc[4 * U8_NT]; rb5 = bsrc[5 * U8_NT];
  };
  auto store = [&](int buf) {
    char* base = smem + buf * U8_STAGE;
    u32x4 lo, hi;
    u8x16_to_bf16(ra, lo, hi);
    *reinterpret_cast<u32x4*>(base + a_off0) = lo;
    *reinterpret_cast<u32x4*>(base + a_off1) = hi;
    u32x4* bdst = reinterpret_cast<u32x4*>(base + U8_A_BYTES) + tid;
    bdst[0] = rb0; bdst[U8_NT] = rb1; bdst[2 * U8_NT] = rb2; bdst[3 * U8_NT] = rb3; bdst[4 * U8_NT] = rb4; bdst[5 * U8_NT] = rb5;
  };

  if (nslab > 0) {
    load(0);
    store(0);
  }
  if (nslab > 1) load(1);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int cur = s & 1;
    const char* As = smem + cur * U8_STAGE;
    const char* Bs = As + U8_A_BYTES;
    // all sixteen fragment reads of the slab are issued before its first MFMA: the matrix pipe starts on the first k-step
    // while the reads of the second are still in flight (two waves per SIMD cannot hide an LDS round trip per k-step)
    bf16x8 af[2][2], bf[2][2][3];
#pragma unroll
    for (int step = 0; step < 2; ++step) {
      const int ks = 2 * step + lh;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int r = wr * 64 + mt * 32 + l31;
        af[step][mt] = lds_read16(As + r * 64 + ((ks ^ ((r >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int hg = 0; hg < 2; ++hg) {
        const int c = wc * 64 + hg * 32 + l31;
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[step][hg][p] = lds_read16(Bs + (p * 128 + c) * 64 + ((ks ^ ((c >> 2) & 3)) << 4));
      }
    }
    if (s + 1 < nslab) store(cur ^ 1);
    if (s + 2 < nslab) load(s + 2);
    __builtin_amdgcn_s_setprio(1);            // the MFMA run ahead of the other block's staging (measured on gemm_x6_kernel: +3-5 %)
#pragma unroll
    for (int step = 0; step < 2; ++step) {
      // smallest terms first: the partial sums of w2 and w1 are added into the accumulator before the large w0 term
#pragma unroll
      for (int p = 2; p >= 0; --p)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int hg = 0; hg < 2; ++hg)
            acc[mt][hg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[step][mt], bf[step][hg][p], acc[mt][hg], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }

  // epilogue: acc[mt][cb][r] <-> row m0 + wr*64 + mt*32 + (r&3) + 8*(r>>2) + 4*lh
  if constexpr (GATED) {
    const int n = n0 + wc * 32 + l31;           // output column; cb = 0: h, 1: g
    const bool nok = n < N;
    const bool timg = tsink.img != nullptr;     // the layer's output also as the pre-split image of its transpose (evae_p6_image.h)
    if (nok || timg) {
      float vbh = (bh && nok) ? bh[n] : 0.f, vbg = (bg && nok) ? bg[n] : 0.f;
      asm volatile("" : "+v"(vbh));        // the wait for the two loads here, not in front of every guarded row's stores
      asm volatile("" : "+v"(vbg));
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float ov[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          // values outside the row guard, stores inside: a guarded FIRST use of the bias loads makes the compiler put a full
          // s_waitcnt vmcnt(0) -- every store in flight included -- in front of each row's stores (evae_gemm_kernel.h)
          const float h = fmaf(acc[mt][0][r], x_scale, vbh);
          const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-kLog2e * fmaf(acc[mt][1][r], x_scale, vbg)));
          ov[r] = h * sg;
          if (m < M && nok) {
            const size_t o = (size_t)m * N + n;
            out[o] = ov[r];
            if (save_s) save_s[o] = sg;
          }
        }
        if (timg) p6_emit_tile(tsink, tsink.row0 + n, nok, tsink.kbase + m0 + wr * 64 + mt * 32, ov, lh);
      }
    }
  } else {
    float* part = out + (size_t)zs * M * N;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int n = n0 + wc * 64 + cb * 32 + l31;
      if (n >= N) continue;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < M) part[(size_t)m * N + n] = acc[mt][cb][r];
        }
    }
  }
}

// ---- the gated forward again, as a copy pipeline (r04) ----------------------------------------------------------------------------
// Same tile, operands, arithmetic and results as u8_gemm_kernel<true>; what differs is how a K-slab reaches the MFMAs.  There:
// seven 16-byte loads per thread into registers, the byte -> bf16 conversion of the whole A slab and eight ds_write_b128 in
// front of the slab's MFMAs, the fragments of a slab read behind its barrier (matrix pipe 45 % busy, 4 VALU per MFMA,
// profiles/r03_pmc/u8fwd1.json).  Here, as in gemm_p6_kernel: LDS-DMA copies (buffer_load / global_load ... lds: no staging
// registers, no ds_write) -- the 24 KB weight image of slab i + 2 into a ring of two, the 4 KB of gathered BYTES of slab i + 3
// into a ring of four (the image is L2-resident, the bytes come from HBM: one slab more of lead) -- the A bytes stay bytes in
// LDS and are converted per fragment (12 VALU per 8 bytes, between the MFMAs), and the fragments of slab i + 1 are read and
// converted behind the MFMAs of slab i (two fragment sets in registers).
// The weight image (three bf16 terms: 6 bytes per weight) is the heavy operand and comes through the L2 -> LDS path, which
// carries about 24 bytes per cycle per CU (profiles/r04_micro): at 128 rows per block the 588 MB of image reads of a
// 25000-row launch alone take the time of the MFMAs.  NWR = 4 (eight waves, 256 rows per block, one block per CU) halves
// the image bytes per row: 32 KB instead of 56 KB per CU and slab.
constexpr int U8P_B = 3 * 128 * 64;
constexpr int u8p_a_bytes(int nwr) { return nwr * 64 * 32; }
constexpr int u8p_lds_bytes(int nwr, int lead) { return 4 * u8p_a_bytes(nwr) + lead * U8P_B; }      // lead 2: 64 KB (two blocks per CU) | 80 KB (one)

typedef __attribute__((address_space(3))) void* u8p_lds_t;
typedef __attribute__((address_space(1))) const void* u8p_glb_t;

// 8 bytes (two dwords) -> 8 bf16: a byte k as bf16 is the upper half of float(k), exact
__device__ __forceinline__ bf16x8 u8x8_to_bf16(unsigned lo, unsigned hi) {
  auto cvt4 = [](unsigned w, unsigned& p0, unsigned& p1) {
    // v_cvt_f32_ubyteN, then v_perm_b32 takes the upper halves of two floats: six VALU per four bytes
    const unsigned f0 = __float_as_uint((float)(w & 0xFFu)), f1 = __float_as_uint((float)((w >> 8) & 0xFFu));
    const unsigned f2 = __float_as_uint((float)((w >> 16) & 0xFFu)), f3 = __float_as_uint((float)(w >> 24));
    p0 = __builtin_amdgcn_perm(f1, f0, 0x07060302u);
    p1 = __builtin_amdgcn_perm(f3, f2, 0x07060302u);
  };
  u32x4 v;
  unsigned a, b;
  cvt4(lo, a, b); v[0] = a; v[1] = b;
  cvt4(hi, a, b); v[2] = a; v[3] = b;
  return __builtin_bit_cast(bf16x8, v);
}

template <int N_>
__device__ __forceinline__ void u8p_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// L = slabs of lead of the weight-image copies = depth of their ring (the byte ring has four slots)
// (L = 3 measured at 256 rows: 74.6 us against 74.3 with L = 2 -- hipcc puts an s_waitcnt vmcnt(0) in front of each k-step's
// first byte read, a ds_read it can see behind a global_load ... lds into the same array, so the lead is one k-step whatever
// the ring holds; the reads as inline assembly with hand-counted waits took the wait away, gained 2 us and failed two parity
// tests at 10 blocks -- not kept.  profiles/r04_ab/knobs.jsonl)
template <int NWR, int L>
__global__ __launch_bounds__(NWR * 128) __attribute__((amdgpu_waves_per_eu(2, 2))) void u8p_gemm_kernel(
    const unsigned char* __restrict__ x, const int64_t* __restrict__ rows, int M, long long ldx, float x_scale,
    const unsigned short* __restrict__ img, int nslab, const float* __restrict__ bh, const float* __restrict__ bg, int N,
    float* __restrict__ out, float* __restrict__ save_s, int tiles_m, int tiles_n, const P6Sink tsink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int U8P_A = u8p_a_bytes(NWR), NWAVE = 2 * NWR, BM = 64 * NWR;
  constexpr int NB = 24 / NWAVE, NB1 = NB * 2 / 3;     // B pieces per wave and slab; of them behind k-step 1
  char* const Abuf = smem;                       // [4][BM rows][32 bytes: the two 16-byte halves swapped on odd groups of 8 rows]
  char* const Bbuf = smem + 4 * U8P_A;           // [L][3 terms][128 columns][64 B swizzled] = the weight image of a slab
  int tm, tn;
  {
    const int ntiles = tiles_m * tiles_n;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int qq = ntiles >> 3, rr = ntiles & 7;
    const int tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
    tm = tile / tiles_n; tn = tile - tm * tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * U8_BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, lh = lane >> 5;

  // copies.  A: this wave's piece = rows 32 w .. 32 w + 31 of the tile, lane -> (row, physical half); the half it FETCHES is
  // swapped on odd groups of eight rows (the 16-lane groups of the fragments' ds_read_b128 then cover all banks)
  const int crow = 32 * wave + (lane >> 1), cphys = lane & 1;
  const int am = m0 + crow;
  const size_t arow_g = rows ? (size_t)rows[am < M ? am : m0] : (size_t)(am < M ? am : m0);
  const unsigned char* const asrc = x + arow_g * ldx + ((cphys ^ ((crow >> 3) & 1)) << 4);
  const rsrc_t rB = make_rsrc(img + (size_t)tn * nslab * (3 * 128 * 32), 0x7FFFFFFFu);
  const unsigned voffb = (unsigned)lane * 16u;
  auto issue_a = [&](int s) {
    __builtin_amdgcn_global_load_lds((u8p_glb_t)(asrc + (size_t)s * U8_BK), (u8p_lds_t)(Abuf + (s & 3) * U8P_A + wave * 1024), 16, 0, 0);
  };
  auto issue_b = [&](int s, int q) {             // piece wave + NWAVE q of the 24
    const int idx = wave + NWAVE * q;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (u8p_lds_t)(Bbuf + (s % L) * U8P_B + idx * 1024), 16, voffb,
                                             (unsigned)s * (unsigned)U8P_B + idx * 1024, 0, 0);
  };

  f32x16u acc[2][2];        // [mt][h | g]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: A the lane's 8 bytes of (row, k-step), B as u8_gemm_kernel
  unsigned fa[2][2], fb[2][2];
#pragma unroll
  for (int step = 0; step < 2; ++step) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int r = wr * 64 + mt * 32 + l31;
      fa[step][mt] = (unsigned)(r * 32 + ((step ^ ((r >> 3) & 1)) << 4) + lh * 8);
    }
#pragma unroll
    for (int hg = 0; hg < 2; ++hg) {
      const int c = wc * 64 + hg * 32 + l31, ks = 2 * step + lh;
      fb[step][hg] = (unsigned)(c * 64 + ((ks ^ ((c >> 2) & 3)) << 4));
    }
  }
  // fragment set s holds k-step s of a slab: [set][mt], [set][hg][term]
  uint2 araw[2];                        // a lane's eight bytes of a k-step: ds_read_b64, each bank touched twice by the 64 lanes
  bf16x8 af[2][2], bf[2][2][3];
  auto read_araw = [&](int s, int step, int mt) {
    araw[mt] = *reinterpret_cast<const uint2*>(Abuf + (s & 3) * U8P_A + fa[step][mt]);
  };
  auto convert_a = [&](int step, int mt) { af[step][mt] = u8x8_to_bf16(araw[mt].x, araw[mt].y); };
  auto read_b = [&](int s, int step, int hg, int p) {
    bf[step][hg][p] = lds_read16(Bbuf + (s % L) * U8P_B + p * (128 * 64) + fb[step][hg]);
  };
#define EVAE_U8P_SB __builtin_amdgcn_sched_barrier(0)
  // One k-step (12 MFMAs) on fragment set STEP; behind the MFMAs the eight reads and two conversions of the NEXT k-step
  // (k-step 1 of slab i behind k-step 0; k-step 0 of slab i + 1 behind k-step 1) and this wave's share of the copies:
  //   k-step 1 of slab i: two thirds of its B(i + L) pieces  -- behind the slab's one barrier: slab i's last reads were issued in k-step 0
  //   k-step 0 of slab i: the other third of B(i + L - 1) and A(i + L)
  // Before k-step 1 reads slab i + 1, slab i + 1 must have landed: what was issued behind it may still be in flight.
  auto kstep = [&](auto step_, auto has1_, auto has2_, auto has3_, int i) {
    constexpr int step = decltype(step_)::value;
    constexpr bool HAS1 = decltype(has1_)::value, HAS2 = decltype(has2_)::value, HAS3 = decltype(has3_)::value;   // slabs i + 1 .. 3 exist
    constexpr bool HASL = L == 2 ? HAS2 : HAS3, HASL1 = L == 2 ? HAS1 : HAS2;
    constexpr bool READS = step == 0 || HAS1;
    if constexpr (step == 1 && HAS1) {
      u8p_wait_vm<(L == 3 && HAS2 ? 1 + NB : 0) + (HASL ? 1 : 0)>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    EVAE_U8P_SB;
    const int rs = step == 0 ? i : i + 1;          // the slab the reads are of
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int p = 2; p >= 0; --p)                   // smallest terms first
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int hg = 0; hg < 2; ++hg) {
          const int j = ((2 - p) * 2 + mt) * 2 + hg;          // 0 .. 11
          acc[mt][hg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[step][mt], bf[step][hg][p], acc[mt][hg], 0, 0, 0);
          EVAE_U8P_SB;
          if constexpr (READS) {
            if (j < 2) { read_araw(rs, step ^ 1, j); EVAE_U8P_SB; }
            else if (j < 8) { const int k = j - 2; read_b(rs, step ^ 1, k / 3, k % 3); EVAE_U8P_SB; }
            else if (j < 10) { convert_a(step ^ 1, j - 8); EVAE_U8P_SB; }
          }
          if constexpr (step == 1 && HASL) { if (j % 3 == 1 && j / 3 < NB1) { issue_b(i + L, j / 3); EVAE_U8P_SB; } }
          if constexpr (step == 0 && HASL1) { if ((j == 3 || j == 6) && NB1 + j / 3 - 1 < NB) { issue_b(i + L - 1, NB1 + j / 3 - 1); EVAE_U8P_SB; } }
          if constexpr (step == 0 && HASL) { if (j == 10) { issue_a(i + L); EVAE_U8P_SB; } }
        }
    __builtin_amdgcn_s_setprio(0);
  };
#undef EVAE_U8P_SB
  constexpr std::integral_constant<int, 0> S0{};
  constexpr std::integral_constant<int, 1> S1{};
  constexpr std::true_type T{};
  constexpr std::false_type F{};
  {
    // prologue (nslab >= 3: the host sends shorter contractions to u8_gemm_kernel)
#pragma unroll
    for (int sl = 0; sl < L; ++sl) {
      issue_a(sl);
#pragma unroll
      for (int q = 0; q < (sl == L - 1 ? NB1 : NB); ++q) issue_b(sl, q);
    }
    u8p_wait_vm<(L - 1) + (L - 2) * NB + NB1>();                // A(0), B(0) landed
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) { read_araw(0, 0, mt); convert_a(0, mt); }
#pragma unroll
    for (int hg = 0; hg < 2; ++hg)
#pragma unroll
      for (int p = 0; p < 3; ++p) read_b(0, 0, hg, p);
    int i = 0;
    for (; i + 3 < nslab; ++i) { kstep(S0, T, T, T, i); kstep(S1, T, T, T, i); }
    kstep(S0, T, T, F, i); kstep(S1, T, T, F, i); ++i;
    kstep(S0, T, F, F, i); kstep(S1, T, F, F, i); ++i;
    kstep(S0, F, F, F, i); kstep(S1, F, F, F, i);
  }
  // epilogue: as u8_gemm_kernel<true>
  const int n = n0 + wc * 32 + l31;
  const bool nok = n < N;
  const bool timg = tsink.img != nullptr;
  if (nok || timg) {
    float vbh = (bh && nok) ? bh[n] : 0.f, vbg = (bg && nok) ? bg[n] : 0.f;
    asm volatile("" : "+v"(vbh));
    asm volatile("" : "+v"(vbg));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float ov[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float h = fmaf(acc[mt][0][r], x_scale, vbh);
        const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-kLog2e * fmaf(acc[mt][1][r], x_scale, vbg)));
        ov[r] = h * sg;
        if (m < M && nok) {
          const size_t o = (size_t)m * N + n;
          out[o] = ov[r];
          if (save_s) save_s[o] = sg;
        }
      }
      if (timg) p6_emit_tile(tsink, tsink.row0 + n, nok, tsink.kbase + m0 + wr * 64 + mt * 32, ov, lh);
    }
  }
}

// ---- weight-gradient pre-passes -----------------------------------------------------------------------------------------
// xT [K + 1][ldt] bytes: xT[k][m] = x[rows[m]][k] for m < M (0 beyond), row K = ones (the bias-gradient row).
// Block = 64 batch rows x 64 pixels through an LDS tile.
__global__ __launch_bounds__(256) void u8_gather_transpose_kernel(const unsigned char* __restrict__ x,
                                                                  const int64_t* __restrict__ rows, int M, int K,
                                                                  long long ldx, unsigned char* __restrict__ xT,
                                                                  long long ldt) {
  __shared__ unsigned char tile[64][80];
  const int m0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int t = threadIdx.x;
  {
    const int r = t >> 2, c = (t & 3) * 16;                 // batch row r, 16 pixels from k0 + c
    const int m = m0 + r;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (m < M && k0 + c < K) {
      const unsigned char* p = x + (size_t)(rows ? rows[m] : m) * ldx + k0 + c;
      if (k0 + c + 16 <= K) v = *reinterpret_cast<const u32x4*>(p);
      else { unsigned char b[16] = {0}; for (int i = 0; i < K - k0 - c; ++i) b[i] = p[i]; v = *reinterpret_cast<u32x4*>(b); }
    }
    *reinterpret_cast<u32x4*>(&tile[r][c]) = v;
  }
  __syncthreads();
  {
    const int k = t >> 2, mc = (t & 3) * 16;                // pixel k0 + k, 16 batch rows from m0 + mc
    if (k0 + k < K && m0 + mc < ldt) {
      unsigned char b[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) b[i] = tile[mc + i][k];
      *reinterpret_cast<u32x4*>(xT + (size_t)(k0 + k) * ldt + m0 + mc) = *reinterpret_cast<u32x4*>(b);
    }
  }
  if (blockIdx.y == 0 && t < 64 && m0 + t < ldt) xT[(size_t)K * ldt + m0 + t] = (m0 + t < M) ? (unsigned char)1 : (unsigned char)0;
}

// dy [M x N] fp32 (row stride ldy) -> the three bf16 terms of dy^T as the GEMM's B-tile images: image (tn, s) covers columns
// n = tn*128 + c and contraction rows m = s*32 .. +31.  Truncation split: w0 = top 8 significant bits of w, w1 of w - w0,
// w2 = the rest -- exact (24 = 3 x 8 bits), and cheaper than rounding.
__global__ __launch_bounds__(256) void u8_prepare_dyT_kernel(const float* __restrict__ dy, int M, int N, long long ldy,
                                                             int nslab, unsigned short* __restrict__ img) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[3 * 128 * 32];
  const int tn = blockIdx.y, s = blockIdx.x;
  const int t = threadIdx.x;
  const int mi = t >> 3, m = s * 32 + mi;                   // one batch row, 16 columns
  const int c0 = (t & 7) * 16;
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const int c = c0 + q4 * 4, n = tn * 128 + c;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < M) {
      if (n + 4 <= N && (ldy & 3) == 0) {
        const float4 f = *reinterpret_cast<const float4*>(dy + (size_t)m * ldy + n);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
      } else {
        for (int i = 0; i < 4; ++i) if (n + i < N) v[i] = dy[(size_t)m * ldy + n + i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned u0 = __float_as_uint(v[i]) & 0xFFFF0000u;
      const float r1 = v[i] - __uint_as_float(u0);
      const unsigned u1 = __float_as_uint(r1) & 0xFFFF0000u;
      const float r2 = r1 - __uint_as_float(u1);
      const unsigned u2 = __float_as_uint(r2);
      const int cc = c + i;
      const int o = cc * 32 + (((mi >> 3) ^ ((cc >> 2) & 3)) << 3) + (mi & 7);
      lds[o] = (unsigned short)(u0 >> 16); lds[128 * 32 + o] = (unsigned short)(u1 >> 16); lds[2 * 128 * 32 + o] = (unsigned short)(u2 >> 16);
    }
  }
  __syncthreads();
  u32x4* dst = reinterpret_cast<u32x4*>(img + ((size_t)tn * nslab + s) * (3 * 128 * 32));
  const u32x4* src = reinterpret_cast<const u32x4*>(lds);
#pragma unroll
  for (int i = 0; i < 6; ++i) dst[t + 256 * i] = src[t + 256 * i];
}

// finish: u8_wgrad_finish_body (evae_u8_prepare.h)
__global__ __launch_bounds__(256) void u8_wgrad_finish_kernel(const U8FinishArgs u) {
  __shared__ float tile[32][33];
  u8_wgrad_finish_body(u, blockIdx.x, blockIdx.y, tile);
}

}  // namespace evae

using namespace evae;

static int u8_nslab(int K) { return cdiv(K, U8_BK); }

extern "C" int evae_dense_u8_supported(int K, long long ldx) { return (K > 0 && (ldx % 16) == 0) ? 1 : 0; }

extern "C" size_t evae_dense_u8_prepared_bytes(int N, int K) {
  if (N <= 0 || K <= 0) return 256;
  return align_up((size_t)cdiv(N, U8_BN) * u8_nslab(K) * (3 * 128 * 32) * sizeof(unsigned short), 256);
}

extern "C" int evae_dense_u8_prepare(const float* wh, const float* wg, int N, int K, void* prepared, size_t prepared_bytes,
                                     evae_stream_t stream_) {
  EVAE_REQUIRE(N > 0 && K > 0 && wh && wg && prepared, "dense_u8_prepare: bad arguments");
  EVAE_REQUIRE(prepared_bytes >= evae_dense_u8_prepared_bytes(N, K), "dense_u8_prepare: buffer too small (%zu)", prepared_bytes);
  const size_t elems = (size_t)cdiv(N, U8_BN) * u8_nslab(K) * 128 * 32;
  u8_prepare_kernel<<<(unsigned)((elems + 255) / 256), 256, 0, (hipStream_t)stream_>>>(wh, wg, N, K, u8_nslab(K),
                                                                                      (unsigned short*)prepared);
  return check_launch("u8_prepare_kernel");
}

static int gated_dense_fwd_u8_core(const unsigned char* x, const int64_t* rows, int M, int K, long long ldx, float x_scale,
                                   const void* prepared, const float* bh, const float* bg, int N, float* out,
                                   float* save_s, const P6Sink& tsink, evae_stream_t stream_) {
  EVAE_REQUIRE(M >= 0 && K > 0 && N > 0, "gated_dense_fwd_u8: bad sizes M=%d K=%d N=%d", M, K, N);
  if (M == 0) return EVAE_OK;
  EVAE_REQUIRE(x && rows && prepared && out, "gated_dense_fwd_u8: null pointer");
  EVAE_REQUIRE(evae_dense_u8_supported(K, ldx) && (((uintptr_t)x) & 15) == 0,
               "gated_dense_fwd_u8: rows of the byte store must be 16-byte aligned (ldx %% 16 == 0)");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)u8_gemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * U8_STAGE);
    attr = true;
  }
  const int tiles_m = cdiv(M, U8_BM), tiles_n = cdiv(N, U8_BN);
  static int pipe = -1, tall = 1;
  if (pipe < 0) {
    const char* e = getenv("EVAE_U8_PIPE");
    pipe = e ? atoi(e) : 1;                    // 0: never, 1: machine-filling launches, 2: whenever the contraction allows
    const char* t = getenv("EVAE_U8_TALL");
    tall = t ? atoi(t) : 1;
    (void)hipFuncSetAttribute((const void*)u8p_gemm_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, u8p_lds_bytes(2, 2));
    (void)hipFuncSetAttribute((const void*)u8p_gemm_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, u8p_lds_bytes(4, 2));
  }
  if (pipe && u8_nslab(K) >= 3 && (pipe == 2 || tiles_m * tiles_n >= 256)) {
    // a machine-filling launch: the copy-pipeline form of the same kernel; 256-row blocks when those still fill the machine
    const int tiles_m4 = cdiv(M, 256);
    if (tall && (tall == 2 || tiles_m4 * tiles_n >= 256)) {
      u8p_gemm_kernel<4, 2><<<tiles_m4 * tiles_n, 512, u8p_lds_bytes(4, 2), (hipStream_t)stream_>>>(
          x, rows, M, ldx, x_scale, (const unsigned short*)prepared, u8_nslab(K), bh, bg, N, out, save_s, tiles_m4, tiles_n, tsink);
      return check_launch("u8p_gemm_kernel<4, 2>");
    }
    u8p_gemm_kernel<2, 2><<<tiles_m * tiles_n, 256, u8p_lds_bytes(2, 2), (hipStream_t)stream_>>>(
        x, rows, M, ldx, x_scale, (const unsigned short*)prepared, u8_nslab(K), bh, bg, N, out, save_s, tiles_m, tiles_n, tsink);
    return check_launch("u8p_gemm_kernel<2, 2>");
  }
  u8_gemm_kernel<true><<<tiles_m * tiles_n, U8_NT, 2 * U8_STAGE, (hipStream_t)stream_>>>(
      x, rows, M, ldx, x_scale, (const unsigned short*)prepared, u8_nslab(K), u8_nslab(K), bh, bg, N, out, save_s, tiles_m, tiles_n,
      tsink);
  return check_launch("u8_gemm_kernel<gated>");
}

extern "C" int evae_gated_dense_fwd_u8(const unsigned char* x, const int64_t* rows, int M, int K, long long ldx, float x_scale,
                                       const void* prepared, const float* bh, const float* bg, int N, float* out,
                                       float* save_s, evae_stream_t stream_) {
  const P6Sink none = {nullptr, 0, 0, 0, 0};
  return gated_dense_fwd_u8_core(x, rows, M, K, ldx, x_scale, prepared, bh, bg, N, out, save_s, none, stream_);
}

// ... whose output also leaves as the pre-split bf16 image of out^T (as evae_gated_dense_fwd_timg)
extern "C" int evae_gated_dense_fwd_u8_timg(const unsigned char* x, const int64_t* rows, int M, int K, long long ldx, float x_scale,
                                            const void* prepared, const float* bh, const float* bg, int N, float* out,
                                            float* save_s, void* timg, int t_nks, int t_row0, int t_kbase, evae_stream_t stream_) {
  EVAE_REQUIRE(timg && t_nks > 0 && t_row0 >= 0 && t_kbase >= 0 && (t_kbase % 8) == 0 && (t_kbase + M + 15) / 16 <= t_nks,
               "gated_dense_fwd_u8_timg: bad image placement (k-steps %d, first k %d, rows %d)", t_nks, t_kbase, M);
  const P6Sink sink = {(unsigned char*)timg, t_nks, t_row0, t_kbase, t_kbase + M};
  return gated_dense_fwd_u8_core(x, rows, M, K, ldx, x_scale, prepared, bh, bg, N, out, save_s, sink, stream_);
}

// ---- weight gradient -------------------------------------------------------------------------------------------------------
struct U8WgradLayout { size_t xT, img, part, total; long long ldt; int nslab, nz, ksplit, tiles_k, tiles_n; };
static U8WgradLayout u8_wgrad_layout(int M, int N, int K) {
  U8WgradLayout L;
  L.nslab = cdiv(M, U8_BK);
  L.ldt = (long long)L.nslab * U8_BK + 32;                  // 16-byte rows, slack for the last slab
  L.tiles_k = cdiv(K + 1, U8_BM); L.tiles_n = cdiv(N, 2 * U8_BN);
  // Split the contraction so that ONE round of resident blocks is in flight and every XCD holds whole units (a unit = the
  // tiles_k row tiles of one (slice, column tile), see u8_gemm_kernel<false>): 64 block slots per XCD (32 CUs x 2) ->
  // floor(64 / tiles_k) units per XCD.  25 100 x 600 x 784: 7 row tiles, 9 units per XCD, 72 / 5 column tiles -> 14 slices
  // (r02 measured the entry point at 512 -> 163 us, 768 -> 179, 1024 -> 175, 1536 -> 187 blocks: every slice more is
  // another [785 x 600] partial plane written and read)
  static int slots = -1;
  if (slots < 0) { const char* e = getenv("EVAE_U8_WGRAD_SLOTS"); slots = e ? atoi(e) : 512; }
  const int per_xcd = std::max(1, (slots / 8) / L.tiles_k);
  int nz = (8 * per_xcd) / L.tiles_n;
  if (nz > L.nslab) nz = L.nslab;
  if (nz < 1) nz = 1;
  L.ksplit = cdiv(L.nslab, nz);
  L.nz = cdiv(L.nslab, L.ksplit);
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes, 256); return at; };
  L.xT = take((size_t)(K + 1) * L.ldt + 64);
  L.img = take((size_t)L.tiles_n * L.nslab * (3 * 128 * 32) * sizeof(unsigned short));
  L.part = take((size_t)L.nz * (K + 1) * N * sizeof(float));
  L.total = o + 256;
  return L;
}

extern "C" size_t evae_dense_bwd_weight_u8_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 256;
  return u8_wgrad_layout(M, N, K).total;
}

extern "C" int evae_dense_bwd_weight_u8_images(int M, int N, int K, size_t* offset, int* nslab) {
  EVAE_REQUIRE(M > 0 && N > 0 && K > 0 && offset && nslab, "dense_bwd_weight_u8_images: bad arguments");
  const U8WgradLayout L = u8_wgrad_layout(M, N, K);
  *offset = L.img; *nslab = L.nslab;
  return EVAE_OK;
}

// phase 0: everything; 1: the pre-passes (byte gather-transpose, dy split / transposition) into the workspace; 2: the product and
// its finish -- so that a caller can run the bandwidth-bound pre-passes on another stream, beside a matrix-bound launch;
// 3: the byte gather-transpose alone (it needs the gather list only, not dy: a training step can issue it during its forward
// pass); 4: everything but the gather-transpose (the workspace holds it)
static int dense_bwd_weight_u8_core(const float* dy, int M, int N, long long ldy, const unsigned char* x,
                                    const int64_t* rows, int K, long long ldx, float x_scale, float* dw, float* db,
                                    void* ws, size_t ws_bytes, int phase_, hipStream_t stream) {
  const bool skip_finish = (phase_ & 16) != 0;       // | 16: the GEMM leaves its partial planes, no finish launch
  const int phase = phase_ & 15;
  EVAE_REQUIRE(M >= 0 && N > 0 && K > 0 && ldy >= N && ldx >= K, "dense_bwd_weight_u8: bad sizes M=%d N=%d K=%d", M, N, K);
  EVAE_REQUIRE(dw != nullptr || phase == 3, "dense_bwd_weight_u8: null dw");
  if (M == 0) {
    if (phase == 1 || phase == 3) return EVAE_OK;
    (void)hipMemsetAsync(dw, 0, (size_t)N * K * sizeof(float), stream);
    if (db) (void)hipMemsetAsync(db, 0, (size_t)N * sizeof(float), stream);
    return check_launch("dense_bwd_weight_u8(empty)");
  }
  EVAE_REQUIRE(x, "dense_bwd_weight_u8: null pointer");      // dy == NULL: its tile images are already in the workspace
  EVAE_REQUIRE(evae_dense_u8_supported(K, ldx) && (((uintptr_t)x) & 15) == 0, "dense_bwd_weight_u8: unaligned byte store");
  const U8WgradLayout L = u8_wgrad_layout(M, N, K);
  if (ws == nullptr || ws_bytes < L.total) { set_error("dense_bwd_weight_u8: workspace too small (%zu)", ws_bytes); return EVAE_EWORKSPACE; }
  unsigned char* xT = (unsigned char*)ws + L.xT;
  unsigned short* img = (unsigned short*)((char*)ws + L.img);
  float* part = (float*)((char*)ws + L.part);
  // the padding columns m >= M of xT (up to the slab boundary + slack) must be zero: they meet dy rows that do not exist
  int rc = EVAE_OK;
  if (phase != 2) {
    if (phase != 4) u8_gather_transpose_kernel<<<dim3(cdiv((int)L.ldt, 64), cdiv(K, 64)), 256, 0, stream>>>(x, rows, M, K, ldx, xT, L.ldt);
    if (dy && phase != 3) u8_prepare_dyT_kernel<<<dim3(L.nslab, L.tiles_n), 256, 0, stream>>>(dy, M, N, ldy, L.nslab, img);
    rc = check_launch("u8 weight-gradient pre-passes");
    if (rc || phase == 1 || phase == 3) return rc;
  }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)u8_gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * U8_STAGE);
    attr = true;
  }
  const int units = L.nz * L.tiles_n;
  u8_gemm_kernel<false><<<8 * L.tiles_k * cdiv(units, 8), U8_NT, 2 * U8_STAGE, stream>>>(
      xT, nullptr, K + 1, L.ldt, 1.0f, img, L.nslab, L.ksplit, nullptr, nullptr, N, part, nullptr, L.tiles_k, L.tiles_n,
      P6Sink{nullptr, 0, 0, 0, 0});
  rc = check_launch("u8_gemm_kernel<raw>");
  if (rc) return rc;
  if (skip_finish) return EVAE_OK;            // the caller sums the planes in a grouped finish (evae_dense_bwd_weight_finish_group)
  const U8FinishArgs u = {part, L.nz, K, N, x_scale, dw, db};
  u8_wgrad_finish_kernel<<<dim3(cdiv(K + 1, 32), cdiv(N, 32)), 256, 0, stream>>>(u);
  return check_launch("u8_wgrad_finish_kernel");
}

namespace evae {
int u8_wgrad_finish_job(int M, int N, int K, float x_scale, float* dw, float* db, void* ws, size_t ws_bytes, U8FinishArgs* out,
                        int* tiles_x, int* tiles_y) {
  EVAE_REQUIRE(M > 0 && N > 0 && K > 0 && dw && ws && out, "dense_bwd_weight_finish_group: bad byte-layer job");
  const U8WgradLayout L = u8_wgrad_layout(M, N, K);
  EVAE_REQUIRE(ws_bytes >= L.total, "dense_bwd_weight_finish_group: byte-layer workspace too small (%zu)", ws_bytes);
  *out = {(const float*)((char*)ws + L.part), L.nz, K, N, x_scale, dw, db};
  *tiles_x = cdiv(K + 1, 32); *tiles_y = cdiv(N, 32);
  return EVAE_OK;
}
}  // namespace evae

extern "C" int evae_dense_bwd_weight_u8(const float* dy, int M, int N, long long ldy, const unsigned char* x,
                                        const int64_t* rows, int K, long long ldx, float x_scale, float* dw, float* db,
                                        void* ws, size_t ws_bytes, evae_stream_t stream_) {
  return dense_bwd_weight_u8_core(dy, M, N, ldy, x, rows, K, ldx, x_scale, dw, db, ws, ws_bytes, 0, (hipStream_t)stream_);
}

extern "C" int evae_dense_bwd_weight_u8_phased(const float* dy, int M, int N, long long ldy, const unsigned char* x,
                                               const int64_t* rows, int K, long long ldx, float x_scale, float* dw, float* db,
                                               void* ws, size_t ws_bytes, int phase, evae_stream_t stream_) {
  EVAE_REQUIRE((phase & 15) >= 0 && (phase & 15) <= 4 && (phase & ~31) == 0 && phase != 0, "dense_bwd_weight_u8_phased: phase must be 1 .. 4 (| 16)");
  return dense_bwd_weight_u8_core(dy, M, N, ldy, x, rows, K, ldx, x_scale, dw, db, ws, ws_bytes, phase, (hipStream_t)stream_);
}
