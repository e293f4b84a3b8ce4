// fp32 GEMM on the bf16 matrix pipe ("x6"): every fp32 operand element is split, inside the kernel, into three bf16 terms
//   a = a0 + a1 + a2,   a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)      (round-to-nearest-even, v_cvt_pk_bf16_f32)
// which carries 3 x 8 = 24 significant bits, i.e. the whole fp32 mantissa (|a - a0 - a1 - a2| <= 2^-26 |a|).  The product of
// two such sums is evaluated as the six partial products of order down to 2^-16
//   a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the three dropped terms (a1 b2, a2 b1, a2 b2) are below 2^-25 |a b|,
// half an fp32 ulp of the product, so the result carries the error of an fp32 GEMM (the accumulation rounding dominates, as
// in v_mfma_f32_32x32x2_f32) -- tests hold this kernel to the fp32 kernel's bar against the fp64 oracle.  Six bf16 MFMAs do
// the work of sixteen fp32 ones: the bf16 pipe runs 16 x the fp32 pipe per instruction-cycle, so the ceiling is 16/6 = 2.7 x
// the fp32 matrix peak (157 -> 419 TFLOP/s of fp32-equivalent work).
//
// Operands: A [M x K] and B [N x K], both contraction-contiguous ("KC": activations x nn.Linear weights, the channels-last
// convolution's pixels x permuted filters).  Same GemmArgs, tile map and epilogues (gemm_epilogue) as the fp32 kernel.
//
// Block = 256 threads = 4 waves (2 x 2), block tile 128 x 128 (gated: 128 rows x 64 outputs = [h | g] column pairs), wave
// tile 64 x 64, K-slab = 32.  LDS holds ONE slab as six planes (A0 A1 A2 B0 B1 B2) of 128 rows x 32 bf16 (64-byte rows, the
// four 16-byte slots XOR-swizzled by (row >> 2) & 3: fragment reads and staging writes are conflict-free): 48 KB, two blocks
// per CU.  Per slab a wave reads its 24 fragments into registers, the block meets at a barrier, and the 48 MFMAs then run
// with one micro-step of the NEXT slab's staging behind every second MFMA (split of a register pair, the three 8-byte LDS
// writes of a chunk, the chunk's load for the slab after next).  VALU work is not free beside the matrix pipe on this
// machine (an MFMA holds the SIMD's issue for its passes: measured, the staging adds its full issue time); pre-split B
// operands (tile images built by a pre-pass) were measured too: same kernel time, and the extra launches cost the headline
// step 28 us, so both operands are split in the kernel.
#pragma once
#include "evae_gemm_kernel.h"

namespace evae {

typedef __bf16 x6_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x6_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x6_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned x6_u32x2 __attribute__((ext_vector_type(2)));

#ifndef EVAE_X6_ABL
#define EVAE_X6_ABL 0
#endif
constexpr int X6_ABL = EVAE_X6_ABL;         // ablation builds (tools only): 1 = no staging of the next slab, 2 = no MFMAs
constexpr int X6_PLANE = 128 * 64;          // bytes of one plane: 128 rows x 32 bf16
constexpr int x6_lds_bytes(int bn) { return 3 * X6_PLANE + 3 * bn * 64; }     // 48 KB (column tile 128) or 36 KB (64)

// two consecutive fp32 -> the three bf16 terms of each, packed pairwise (9 VALU instructions: 3 v_cvt_pk_bf16_f32, 2 x (shift,
// mask, v_pk_add_f32))
__device__ __forceinline__ void x6_split2(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
  x6_f32x2 r = {x, y};
  p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, x6_bf16x2));
  x6_f32x2 h = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xFFFF0000u)};
  r = r - h;                                                   // exact: a0 carries the leading bits of a
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, x6_bf16x2));
  h[0] = __uint_as_float(p1 << 16); h[1] = __uint_as_float(p1 & 0xFFFF0000u);
  r = r - h;
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, x6_bf16x2));
}

// TM = terms per operand.  3: the fp32-accurate product above.  2 (the top-K screen only, whose distances are a filter with a
// proven error bound, re-ranked exactly afterwards): a = a0 + a1, products a0 b0 + a0 b1 + a1 b0 -- half the MFMAs, two thirds
// of the staging; what the split drops is <= 2^-15 |a b| per product (|a1 b1| <= 2^-16 |a b| dominates).
// EPI_DIST_TILEMIN with g.e0 == nullptr: the squared norms of the A rows (the cache) are accumulated while their slabs are
// staged (every row is staged exactly once by its block: no pass of its own over the cache), handed to the epilogue through
// LDS, and their maximum goes to *(unsigned*)g.aux_cnt (bit pattern, atomicMax).
template <int EPI, int CV, int BN_, int TM = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_x6_kernel(const GemmArgs g) {
  constexpr bool GATED = (EPI == EPI_GATED || EPI == EPI_RAW_GATED);
  constexpr bool NORMS = (EPI == EPI_DIST_TILEMIN);
  static_assert(BN_ == 128 || (BN_ == 64 && !GATED), "column tile: 128, or 64 for narrow plain outputs");
  static_assert(TM == 3 || (TM == 2 && BN_ == 128), "two-term products: 128-wide column tile only");
  constexpr int NW = 4, MT = 2, NT = BN_ / 64, GNT = 256, NV = 4, NVB = BN_ / 32;   // float4 chunks per thread: A 4, B 4 or 2
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);
  if constexpr (EPI == EPI_PRIOR_LSE || EPI == EPI_PRIOR_P) {
    if (g.skip_flag != nullptr && *g.skip_flag != 0u) return;
  }
  const int ntiles = g.tiles_m * g.tiles_n;
  int tile;
  {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int qq = ntiles >> 3, rr = ntiles & 7;
    tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM;
  const int n0 = GATED ? tn * 64 : tn * BN_;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int nslab[2];
  nslab[0] = (g.Kc[0] + BK - 1) / BK;
  nslab[1] = g.npairs > 1 ? (g.Kc[1] + BK - 1) / BK : 0;
  int s_begin = 0, s_end = nslab[0] + nslab[1];
  if (g.ksplit > 0) {
    s_begin = blockIdx.z * g.ksplit;
    const int e = s_begin + g.ksplit;
    if (e < s_end) s_end = e;
  }

  const rsrc_t rA0 = make_rsrc(g.A[0], 0x7FFFFFFFu);
  const rsrc_t rA1 = make_rsrc(g.npairs > 1 ? g.A[1] : g.A[0], 0x7FFFFFFFu);
  const rsrc_t rB0 = make_rsrc(g.B[0], 0x7FFFFFFFu);
  const rsrc_t rB1 = make_rsrc(GATED ? g.Bg : (g.npairs > 1 ? g.B[1] : g.B[0]), 0x7FFFFFFFu);

  // staging roles: chunk f = tid + 256 i -> tile row f >> 3, float4 (f & 7) of the 32-wide slab
  unsigned voA[2][NV], voB[2][NV];
  unsigned long long tapmask[NV];
  unsigned st_off[NV];                    // byte offset of the chunk's 8 bytes inside a plane
  const int c8 = tid & 7;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int row = (tid >> 3) + 32 * i;
    st_off[i] = (unsigned)(row * 64 + ((((c8 >> 1) ^ ((row >> 2) & 3))) << 4) + (c8 & 1) * 8);
    tapmask[i] = 0ull;
    const int r = m0 + row;
    const bool ok = r < g.M;
    if constexpr (CV == 1) {
      const unsigned rr = ok ? (unsigned)r : 0u;
      const unsigned n = fdiv(rr, g.cv.div_rhw), rem = rr - n * (unsigned)(g.cv.RH * g.cv.RW);
      const unsigned ry = fdiv(rem, g.cv.div_rw), rx = rem - ry * (unsigned)g.cv.RW;
      const int ay = (int)ry * g.cv.rs + g.cv.roy, ax = (int)rx * g.cv.rsx + g.cv.rox;
      const long long off = (((long long)n * g.cv.IH + ay) * g.cv.IW + ax) * g.cv.ps * 4 + 16 * c8 + (long long)g.cv.bias;
      voA[0][i] = voA[1][i] = ok ? (unsigned)off : OOB;
      if (ok)
        for (int t = 0; t < g.cv.ntaps; ++t) {
          const int y = ay + g.cv.tdy[t], x = ax + g.cv.tdx[t];
          if ((unsigned)y < (unsigned)g.cv.IH && (unsigned)x < (unsigned)g.cv.IW) tapmask[i] |= 1ull << t;
        }
    } else {
#pragma unroll
      for (int p = 0; p < 2; ++p) voA[p][i] = ok ? (unsigned)(r * g.lda[p] + 4 * c8) * 4u : OOB;
    }
    // B rows: gated -> LDS row [wc][h | g][32] = weight row n0 + (row >> 6) * 32 + (row & 31) of bank (row & 32)
    const int n = GATED ? n0 + (row >> 6) * 32 + (row & 31) : n0 + row;
#pragma unroll
    for (int p = 0; p < 2; ++p) voB[p][i] = (n < g.N && i < NVB) ? (unsigned)(n * g.ldb[GATED ? 0 : p] + 4 * c8) * 4u : OOB;
  }

  float4 ra[NV], rb[NV];
  // chunk q of slab s into its register (q = 0..3: A chunks, 4..7: B chunks)
  auto load_chunk = [&](int s, int q) {
    const bool p1 = s >= nslab[0];
    const int k0 = (p1 ? s - nslab[0] : s) * BK;
    const int kv = (p1 ? g.Kc[1] : g.Kc[0]) - k0;
    const bool dead = 4 * c8 + 4 > kv;                            // K tail: this chunk lies beyond the contraction
    if (q < NV) {
      const int i = q;
      if constexpr (CV == 1) {
        const int tap = k0 / g.cv.Cg;
        const int kin = k0 - tap * g.cv.Cg;
        const unsigned so = (unsigned)(g.cv.tsoff[tap] + kin * 4);
        const bool live = ((tapmask[i] >> tap) & 1ull) && (4 * c8 + 4 <= g.cv.creal - kin);
        ra[i] = buf_ld4(rA0, live ? voA[0][i] : OOB, so);
      } else {
        ra[i] = buf_ld4(p1 ? rA1 : rA0, dead ? OOB : (p1 ? voA[1][i] : voA[0][i]), (unsigned)k0 * 4u);
      }
    } else {
      const int i = q - NV;
      // gated: chunk i belongs to bank i & 1 (row & 32 = 32 (i & 1) because tid >> 3 < 32)
      const rsrc_t rB = GATED ? ((i & 1) ? rB1 : rB0) : (p1 ? rB1 : rB0);
      rb[i] = buf_ld4(rB, dead ? OOB : ((!GATED && p1) ? voB[1][i] : voB[0][i]), (unsigned)k0 * 4u);
    }
  };
  auto load = [&](int s) {
#pragma unroll
    for (int q = 0; q < NV + NVB; ++q) load_chunk(s, q);
  };
  // staging of chunk q in three micro-steps that fit between two MFMAs each: split of the first pair, split of the second
  // pair, the three 8-byte LDS writes (after which the chunk's register is free for the slab after next)
  unsigned sp[6];
  float nrm[NV] = {0.f, 0.f, 0.f, 0.f};   // NORMS: this thread's part of |row|^2 of its four A rows
  auto stage_part = [&](int q, int part) {
    const bool isb = q >= NV;
    const int i = isb ? q - NV : q;
    const float4 v = isb ? rb[i] : ra[i];
    if (part == 0) {
      x6_split2(v.x, v.y, sp[0], sp[1], sp[2]);
      if constexpr (NORMS) { if (!isb) nrm[i] = fmaf(v.y, v.y, fmaf(v.x, v.x, nrm[i])); }
    } else if (part == 1) {
      x6_split2(v.z, v.w, sp[3], sp[4], sp[5]);
      if constexpr (NORMS) { if (!isb) nrm[i] = fmaf(v.w, v.w, fmaf(v.z, v.z, nrm[i])); }
    } else {
      char* p = lds + (isb ? 3 * X6_PLANE : 0) + st_off[i];
      const int ps = isb ? BN_ * 64 : X6_PLANE;                    // bytes between the planes of this operand
      x6_u32x2 t0 = {sp[0], sp[3]}, t1 = {sp[1], sp[4]}, t2 = {sp[2], sp[5]};
      *reinterpret_cast<x6_u32x2*>(p) = t0;
      *reinterpret_cast<x6_u32x2*>(p + ps) = t1;
      if constexpr (TM == 3) *reinterpret_cast<x6_u32x2*>(p + 2 * ps) = t2;
    }
  };
  auto stage = [&](int q) { stage_part(q, 0); stage_part(q, 1); stage_part(q, 2); };

  // fragment addresses (bytes inside a plane) of this lane for the two k-steps of a slab
  unsigned fa[2][MT], fb[2][NT];
#pragma unroll
  for (int step = 0; step < 2; ++step) {
    const int ks = 2 * step + lh;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int r = wr * 64 + mt * 32 + l31;
      fa[step][mt] = (unsigned)(r * 64 + ((ks ^ ((r >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c = wc * 32 * NT + nt * 32 + l31;
      fb[step][nt] = (unsigned)(3 * X6_PLANE + c * 64 + ((ks ^ ((c >> 2) & 3)) << 4));
    }
  }

  if (s_begin < s_end) {
    load(s_begin);
#pragma unroll
    for (int q = 0; q < NV + NVB; ++q) stage(q);
    if (s_begin + 1 < s_end) load(s_begin + 1);
    __syncthreads();
#define EVAE_SB __builtin_amdgcn_sched_barrier(0)
    auto slab = [&](int s, auto ST_, auto LD_) {
      constexpr bool ST = decltype(ST_)::value, LD = decltype(LD_)::value;
      x6_bf16x8 af[2][MT][TM], bf[2][NT][TM];
#pragma unroll
      for (int step = 0; step < 2; ++step) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int p = 0; p < TM; ++p) af[step][mt][p] = *reinterpret_cast<const x6_bf16x8*>(lds + fa[step][mt] + p * X6_PLANE);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int p = 0; p < TM; ++p) bf[step][nt][p] = *reinterpret_cast<const x6_bf16x8*>(lds + fb[step][nt] + p * (BN_ * 64));
      }
      if constexpr (ST) __syncthreads();             // every wave holds its fragments: the planes may be overwritten
      __builtin_amdgcn_s_setprio(1);                 // the MFMA phase ahead of the other block's fragment reads
      // 48 MFMAs: term-major inside a k-step (four independent accumulators between two uses of one), smallest terms first.
      // Behind every second MFMA one micro-step of the staging of the next slab; the chunk's load for the slab after next
      // goes out right behind its LDS writes, a whole slab before it is needed.
      constexpr int NP = TM == 3 ? 6 : 3;                          // partial products per element pair
      constexpr int PA[6] = {TM == 3 ? 2 : 1, 0, TM == 3 ? 1 : 0, 1, 0, 0}, PB[6] = {0, TM == 3 ? 2 : 1, TM == 3 ? 1 : 0, 0, 1, 0};
      constexpr int SP = (2 * NP * MT * NT) / (3 * (NV + NVB));    // MFMAs per micro-step: 48 / 24 = 2, 24 / 18 = 1, 24 / 24 = 1
#pragma unroll
      for (int step = 0; step < 2; ++step)
#pragma unroll
        for (int t = 0; t < NP; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int j = ((step * NP + t) * MT + mt) * NT + nt;         // 0 .. 2 NP MT NT - 1
              EVAE_SB;
              if constexpr (!(X6_ABL & 2))
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[step][mt][PA[t]], bf[step][nt][PB[t]], acc[mt][nt], 0, 0, 0);
              EVAE_SB;
              if constexpr (ST) {
                if (j % SP == SP - 1 && !(X6_ABL & 1)) {
                  const int k = j / SP, q = k / 3, part = k - 3 * q;           // micro-step k
                  if (q < NV + NVB) {
                    stage_part(q, part);
                    if constexpr (LD) {
                      if (part == 2) load_chunk(s + 2, q);
                    }
                  }
                }
              }
            }
      EVAE_SB;
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();
    };
    constexpr std::true_type T{};
    constexpr std::false_type F{};
    int s = s_begin;
    for (; s + 2 < s_end; ++s) slab(s, T, T);
    if (s + 1 < s_end) { slab(s, T, F); ++s; }
    slab(s, F, F);
#undef EVAE_SB
  }
  if constexpr (NORMS) {
    if (g.e0 == nullptr) {
      // the eight threads of a row (consecutive lanes) add up their parts; smem[512 + row] is beyond the epilogue's scratch
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float t = nrm[i];
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
        if (c8 == 0) smem[512 + (tid >> 3) + 32 * i] = t;
      }
      __syncthreads();
      if (tid < BM) {
        float t = (m0 + tid < g.M) ? smem[512 + tid] : 0.f;
        t = wave_max(t);
        if (g.tile_max != nullptr) { if (lane == 0) g.tile_max[2 * tm + (tid >> 6)] = t; }
        else if (lane == 0 && t > 0.f) atomicMax(reinterpret_cast<unsigned*>(g.aux_cnt), __float_as_uint(t));
      }
    }
  }
  gemm_epilogue<EPI, BN_, NW, CV>(g, acc, m0, n0, tm, wr, wc, lane, smem, blockIdx.z);
}

// ---- both operands k-major: C [M x N] = A^T B with A [Kc x M] (row stride lda), B [Kc x N] (row stride ldb) -- the weight
// gradient dW = dy^T x (contraction over the batch rows; ones_col = virtual all-ones column of B: the bias gradient).
// Same tile, planes, MFMA schedule and epilogue as gemm_x6_kernel; what differs is the staging: a thread loads a 4 x 4 block
// (four consecutive contraction rows x four consecutive columns, one float4 per row), which holds, for each of its four
// columns, four consecutive k of one LDS row = one 8-byte write per plane -- the transposition costs no instruction.  LDS row r
// of an operand lives at physical row r ^ ((r >> 2) & 1): the 16 lanes of a ds_write_b64 pass are two column groups (rows
// 4 n4 + i and 4 (n4 + 1) + i) x the eight 8-byte chunks of a row, and the swap gives the two rows opposite parity, i.e. the
// two halves of the 128-byte bank period (without it every pass is a 2-way conflict); fragment reads see whole aligned
// groups of four rows, so their conflict-free pattern is unchanged.
// CV = 2: B is the im2col matrix of a channels-last convolution's weight gradient, [pixel][(tap, channel)] gathered as four
// channels of one tap (ConvMap, evae_gemm_kernel.h); a thread's four columns are one (tap, channel group) for the whole
// kernel, its contraction rows are decomposed into (image, y, x) per slab.
template <int EPI, int CV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_x6t_kernel(const GemmArgs g) {
  constexpr int NW = 4, MT = 2, NT = 2, BN_ = 128;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);
  const int ntiles = g.tiles_m * g.tiles_n;
  int tile;
  {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int qq = ntiles >> 3, rr = ntiles & 7;
    tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN_;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nslab = (g.Kc[0] + BK - 1) / BK;
  int s_begin = 0, s_end = nslab;
  if (g.ksplit > 0) {
    s_begin = blockIdx.z * g.ksplit;
    const int e = s_begin + g.ksplit;
    if (e < s_end) s_end = e;
  }
  const rsrc_t rA = make_rsrc(g.A[0], 0x7FFFFFFFu);
  const rsrc_t rB = make_rsrc(g.B[0], 0x7FFFFFFFu);

  // staging roles: four columns 4 n4 .. 4 n4 + 3 of the tile, contraction rows 4 m4 .. 4 m4 + 3 of the slab
  const int n4 = tid >> 3, m4 = tid & 7;
  const bool colokA = m0 + 4 * n4 + 4 <= g.M;
  const int cb = n0 + 4 * n4;
  const int nlim = g.ones_col >= 0 ? g.ones_col : g.N;
  const bool colokB = cb + 4 <= nlim;
  const bool onesB = g.ones_col >= 0 && cb == g.ones_col;
  int coltap = 0, colch = 0;
  if constexpr (CV == 2) { coltap = cb / g.cv.Cg; colch = cb - coltap * g.cv.Cg; if (!colokB) coltap = 0; }
  unsigned voA[4], voB[4], st_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    voA[j] = colokA ? (unsigned)((4 * m4 + j) * g.lda[0] + m0 + 4 * n4) * 4u : OOB;
    voB[j] = colokB ? (unsigned)((4 * m4 + j) * g.ldb[0] + cb) * 4u : OOB;
    const int pr = 4 * n4 + (j ^ (n4 & 1));                       // physical row of column j
    st_off[j] = (unsigned)(pr * 64 + ((((m4 >> 1) ^ (n4 & 3))) << 4) + (m4 & 1) * 8);
  }

  float4 ra[4], rb[4];
  auto load_a = [&](int s) {
    const int k0 = s * BK, kv = g.Kc[0] - k0;
    const unsigned so = (unsigned)k0 * (unsigned)g.lda[0] * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = buf_ld4(rA, (4 * m4 + j < kv) ? voA[j] : OOB, so);
  };
  auto load_b = [&](int s) {
    const int k0 = s * BK, kv = g.Kc[0] - k0;
    const unsigned so = (unsigned)k0 * (unsigned)g.ldb[0] * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rb[j] = buf_ld4(rB, (4 * m4 + j < kv) ? voB[j] : OOB, so);
      if (onesB) rb[j] = make_float4((4 * m4 + j < kv) ? 1.f : 0.f, 0.f, 0.f, 0.f);
    }
  };
  auto comp = [](const float4& v, int i) -> float { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); };
  // micro-step k = 0 .. 15 of the staging of one slab: operand B (k < 8) then A; within an operand the row pair (0, 1) of its
  // four columns, then the pair (2, 3).  One step = the split of two elements (column i, two consecutive contraction rows) and
  // three 4-byte LDS writes; a row pair's two registers are free after its fourth column and are reloaded at once for the
  // slab after next -- a whole slab before they are needed.  Returns 1 + (operand, pair) when a pair is done, else 0.
  auto micro = [&](int k) -> int {
    const bool isa = k >= 8;
    const int kk = isa ? k - 8 : k, pair = kk >> 2, i = kk & 3;
    unsigned p0, p1, p2;
    const float x = isa ? comp(ra[2 * pair], i) : comp(rb[2 * pair], i);
    const float y = isa ? comp(ra[2 * pair + 1], i) : comp(rb[2 * pair + 1], i);
    x6_split2(x, y, p0, p1, p2);
    char* p = lds + (isa ? 0 : 3 * X6_PLANE) + st_off[i] + 4 * pair;
    *reinterpret_cast<unsigned*>(p) = p0;
    *reinterpret_cast<unsigned*>(p + X6_PLANE) = p1;
    *reinterpret_cast<unsigned*>(p + 2 * X6_PLANE) = p2;
    return i == 3 ? 1 + (isa ? 2 : 0) + pair : 0;
  };
  auto reload = [&](int s, int what) {              // what - 1 = 2 * operand + pair
    const int k0 = s * BK, kv = g.Kc[0] - k0;
    const bool isa = what >= 3;
    const int pair = (what - 1) & 1;
    const unsigned so = (unsigned)k0 * (unsigned)(isa ? g.lda[0] : g.ldb[0]) * 4u;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * pair + jj;
      const bool live = 4 * m4 + j < kv;
      if (isa) ra[j] = buf_ld4(rA, live ? voA[j] : OOB, so);
      else {
        if constexpr (CV == 2) {
          const unsigned m = (unsigned)(k0 + 4 * m4 + j);
          const unsigned n = fdiv(m, g.cv.div_rhw), rem = m - n * (unsigned)(g.cv.RH * g.cv.RW);
          const unsigned ry = fdiv(rem, g.cv.div_rw), rx = rem - ry * (unsigned)g.cv.RW;
          const int y = (int)ry * g.cv.rs + g.cv.roy + g.cv.tdy[coltap], x = (int)rx * g.cv.rsx + g.cv.rox + g.cv.tdx[coltap];
          const bool in = live && colokB && (unsigned)y < (unsigned)g.cv.IH && (unsigned)x < (unsigned)g.cv.IW;
          const unsigned off = (unsigned)((((int)n * g.cv.IH + y) * g.cv.IW + x) * g.cv.ps + colch) * 4u;
          rb[j] = buf_ld4(rB, in ? off : OOB, 0u);
        } else {
          rb[j] = buf_ld4(rB, live ? voB[j] : OOB, so);
        }
        if (onesB) rb[j] = make_float4(live ? 1.f : 0.f, 0.f, 0.f, 0.f);
      }
    }
  };

  unsigned fa[2][MT], fb[2][NT];
#pragma unroll
  for (int step = 0; step < 2; ++step) {
    const int ks = 2 * step + lh;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int r = wr * 64 + mt * 32 + l31, pr = r ^ ((r >> 2) & 1);
      fa[step][mt] = (unsigned)(pr * 64 + ((ks ^ ((pr >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c = wc * 64 + nt * 32 + l31, pc = c ^ ((c >> 2) & 1);
      fb[step][nt] = (unsigned)(3 * X6_PLANE + pc * 64 + ((ks ^ ((pc >> 2) & 3)) << 4));
    }
  }

  if (s_begin < s_end) {
    load_a(s_begin); reload(s_begin, 1); reload(s_begin, 2);
#pragma unroll
    for (int k = 0; k < 16; ++k) micro(k);
    if (s_begin + 1 < s_end) { load_a(s_begin + 1); reload(s_begin + 1, 1); reload(s_begin + 1, 2); }
    __syncthreads();
    auto slab = [&](int s, auto ST_, auto LD_) {
      constexpr bool ST = decltype(ST_)::value, LD = decltype(LD_)::value;
      x6_bf16x8 af[2][MT][3], bf[2][NT][3];
#pragma unroll
      for (int step = 0; step < 2; ++step) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int p = 0; p < 3; ++p) af[step][mt][p] = *reinterpret_cast<const x6_bf16x8*>(lds + fa[step][mt] + p * X6_PLANE);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int p = 0; p < 3; ++p) bf[step][nt][p] = *reinterpret_cast<const x6_bf16x8*>(lds + fb[step][nt] + p * X6_PLANE);
      }
      if constexpr (ST) __syncthreads();
      __builtin_amdgcn_s_setprio(1);
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int step = 0; step < 2; ++step)
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int j = ((step * 6 + t) * MT + mt) * NT + nt;          // 0 .. 47
              __builtin_amdgcn_sched_barrier(0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[step][mt][PA[t]], bf[step][nt][PB[t]], acc[mt][nt], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (ST) {
                if (j % 3 == 2) {
                  const int done = micro(j / 3);
                  if constexpr (LD) {
                    if (done) reload(s + 2, done);
                  }
                }
              }
            }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();
    };
    constexpr std::true_type T{};
    constexpr std::false_type F{};
    int s = s_begin;
    for (; s + 2 < s_end; ++s) slab(s, T, T);
    if (s + 1 < s_end) { slab(s, T, F); ++s; }
    slab(s, F, F);
  }
  gemm_epilogue<EPI, BN_, NW, 0>(g, acc, m0, n0, tm, wr, wc, lane, smem, blockIdx.z);
}

// k-major operands the x6t kernel can take: no row gather, 16-byte aligned rows, column counts in fours, below 2 GiB
static bool gemm_x6t_ok(const GemmArgs& g, bool im2col_b = false) {
  if (g.b_krows != nullptr || g.a_rows != nullptr || g.npairs != 1) return false;
  if (g.lda[0] % 4 || g.M % 4 || (!im2col_b && g.ldb[0] % 4)) return false;
  if (g.ones_col >= 0 ? (g.ones_col % 4 != 0) : (g.N % 4 != 0)) return false;
  if (((uintptr_t)g.A[0] | (uintptr_t)g.B[0]) & 15) return false;
  return (long long)g.Kc[0] * g.lda[0] < (1ll << 29) && (im2col_b || (long long)g.Kc[0] * g.ldb[0] < (1ll << 29));
}
// how much of its 128 x 128 tiles a weight-gradient product [M x N] fills (the launch policy wants >= 0.85)
static double gemm_x6t_fill(int M, int N) {
  return ((double)M / (cdiv(M, BM) * BM)) * ((double)N / (cdiv(N, 128) * 128));
}

// split of the contraction (M_c rows) for the weight-gradient kernel: enough 128 x 128 tiles x slices to fill 512 block slots,
// at least eight K-slabs per slice
struct X6tSplit { int nz, ksplit; };
static X6tSplit x6t_split(int Mc, int N, int Kp) {
  const int slabs = cdiv(Mc, BK), tiles = cdiv(N, BM) * cdiv(Kp, 128);
  int nz = std::max(1, std::min(512 / std::max(tiles, 1), slabs / 8));
  const int ksplit = cdiv(slabs, nz);
  nz = cdiv(slabs, ksplit);
  return {nz, ksplit};
}

template <int EPI, int CV = 0>
static int launch_gemm_x6t(GemmArgs& g, int nz, hipStream_t stream, const char* what) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_x6t_kernel<EPI, CV>, hipFuncAttributeMaxDynamicSharedMemorySize, x6_lds_bytes(128));
    attr_done = true;
  }
  g.tiles_m = cdiv(g.M, BM);
  g.tiles_n = cdiv(g.N, 128);
  g.dbg = 0;
  dim3 grid(g.tiles_m * g.tiles_n, 1, nz);
  gemm_x6t_kernel<EPI, CV><<<grid, 256, x6_lds_bytes(128), stream>>>(g);
  return check_launch(what);
}

// operands the x6 kernel can take: contraction-contiguous, 16-byte aligned rows, offsets below 2 GiB
static bool gemm_x6_ok(const GemmArgs& g) {
  for (int p = 0; p < g.npairs; ++p) {
    if (g.Kc[p] % 4 || g.lda[p] % 4 || g.ldb[p] % 4) return false;
    if (((uintptr_t)g.A[p] | (uintptr_t)g.B[p]) & 15) return false;
    if ((long long)g.M * g.lda[p] >= (1ll << 29) || (long long)g.N * g.ldb[p] >= (1ll << 29)) return false;
  }
  if (g.Bg && ((uintptr_t)g.Bg & 15)) return false;
  return g.a_rows == nullptr;
}

// Policy, shared by every translation unit (defined in evae_dense.hip, set through evae_gemm_x6_configure): enabled (EVAE_X6=0
// keeps every GEMM on the fp32 matrix pipe) and the smallest row count that takes the x6 kernel (EVAE_X6_MIN_ROWS, default
// 2048: launches of a few hundred rows are latency-bound and stay on the fp32 kernel)
extern int g_x6_enabled, g_x6_min_rows;
void gemm_x6_init_policy();
static bool gemm_x6_enabled() { if (g_x6_enabled < 0) gemm_x6_init_policy(); return g_x6_enabled != 0; }
static int gemm_x6_min_rows() { if (g_x6_min_rows < 0) gemm_x6_init_policy(); return g_x6_min_rows; }
// Worth it only when the launch fills the machine: at least 384 block tiles (of 512 resident slots) of 128 x 128 or
// 128 x 64 -- narrow outputs (the 80-column heads of the encoder) leave most CUs without a block and run faster on the fp32
// kernel's smaller tiles.  min_rows == 0 (tests) takes every eligible launch.
static bool gemm_x6_fills(int M, int N, bool gated) {
  if (gemm_x6_min_rows() == 0) return true;
  const long long t128 = (long long)cdiv(M, BM) * cdiv(N, gated ? 64 : 128), t64 = (long long)cdiv(M, BM) * cdiv(N, 64);
  return M >= gemm_x6_min_rows() && (gated ? t128 : std::max(t128, t64)) >= 384;
}
static bool gemm_x6_use(const GemmArgs& g, bool gated = false) {
  return gemm_x6_enabled() && gemm_x6_fills(g.M, g.N, gated) && gemm_x6_ok(g);
}

// column tile of a plain (not gated) launch: the one that wastes less of the machine -- padding of the last column tile
// times the occupancy of the last round of blocks (two blocks per CU, 256 CUs)
static int gemm_x6_pick_bn(int M, int N) {
  auto eff = [&](int bn) {
    const int tn = cdiv(N, bn);
    const long long tiles = (long long)cdiv(M, BM) * tn;
    const long long slots = 512;
    const double fill = (double)N / ((double)tn * bn);
    const double rounds = (double)tiles / (double)(cdiv((int)tiles, (int)slots) * slots);
    return fill * rounds * (bn == 128 ? 1.0 : 0.9);              // the narrow tile reads 0.75 fragments per MFMA instead of 0.5
  };
  return eff(128) >= eff(64) ? 128 : 64;
}

template <int EPI, int CV = 0, int BN_ = 128, int TM = 3>
static int launch_gemm_x6(GemmArgs& g, int nz, hipStream_t stream, const char* what) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_x6_kernel<EPI, CV, BN_, TM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              x6_lds_bytes(BN_));
    attr_done = true;
  }
  constexpr bool GATED = (EPI == EPI_GATED || EPI == EPI_RAW_GATED);
  g.tiles_m = cdiv(g.M, BM);
  g.tiles_n = cdiv(g.N, GATED ? 64 : BN_);
  g.dbg = 0;
  dim3 grid(g.tiles_m * g.tiles_n, 1, nz);
  gemm_x6_kernel<EPI, CV, BN_, TM><<<grid, 256, x6_lds_bytes(BN_), stream>>>(g);
  return check_launch(what);
}

}  // namespace evae
