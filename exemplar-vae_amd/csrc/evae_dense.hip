// Dense layers of the encoder/decoder on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact fp32 products/accumulation, 157 TFLOP/s peak).
// Replaces utils/nn.py:29-69 (NonLinear, GatedDense) + torch.nn.Linear and their autograd.
//
// One templated LDS-tiled GEMM serves every call:
//   block tile 128 x 128 x 32, 256 threads = 4 waves in a 2 x 2 grid, wave tile 64 x 64 =
//   2 x 2 MFMA tiles of 32 x 32 (64 accumulator VGPRs), LDS double-buffered, the next K-slab's
//   global loads are issued before the current slab's MFMAs (register prefetch).
//   Operands come in two layouts, chosen per call so that global reads are always contiguous
//   float4 streams and no transposition pass is ever needed:
//     KC ("k-contiguous", [rows][k]):  x and nn.Linear weights in the forward;   fragments are read
//        as ds_read_b128 along k -- lanes 0-31 take k..k+3 and lanes 32-63 take k+4..k+7 of an
//        8-wide k-group, feeding four consecutive MFMAs; row stride 36 floats (stride/4 odd) makes
//        the 16-lane b128 groups conflict-free;
//     RC ("row-contiguous", [k][rows]): dy^T and x in the weight gradient, W in the data gradient;
//        fragments are ds_read_b32 of 32 consecutive floats (conflict-free by construction).
//   Row gathers (the exemplar gather of models/BaseModel.py:247) are folded into the tile loads.
//   Epilogues: bias + activation, the GatedDense gate h*sigmoid(g) (h and g column tiles live in
//   the same wave, so the product is register-local), the gate derivative for the layer below, and
//   raw split-K partials (weight gradient; reduced deterministically by a second kernel).
//   Workgroup ids are remapped so that each XCD owns a contiguous run of tiles (shared A row-panels
//   stay in one L2).
#include "evae_common.h"

namespace evae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, GNT = 256;
constexpr int KS = BK + 4;           // KC tile row stride (floats); 36/4 = 9 odd
constexpr int RS = BM + 4;           // RC tile row stride
constexpr int TILE_FLOATS = BM * KS; // 4608 >= BK * RS = 4224

enum { EPI_LINEAR = 0, EPI_GATED = 1, EPI_GATE_BWD = 2, EPI_PARTIAL = 3 };

struct GemmArgs {
  const float* A[2];
  const float* B[2];
  int lda[2], ldb[2];
  int Kc[2];                 // contraction length of each (A,B) pair
  int npairs;
  const int64_t* a_rows;     // KC A: gather of output rows;   RC A: unused
  const int64_t* b_krows;    // RC B: gather along the contraction index (weight gradient x rows)
  const float* Bg;           // EPI_GATED: second weight matrix (g), same layout as B[0]
  int M, N;                  // output rows / columns
  int ksplit;                // contraction steps (of BK) per blockIdx.z; 0 = no split
  const float* bias0;
  const float* bias1;
  float* out0;
  float* out1;
  float* out2;
  int ldo;
  const float* e0;           // EPI_GATE_BWD: h of the layer below
  const float* e1;           //               s of the layer below
  int act;
  float lo, hi;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ float4 ld4(const float* p, int valid, bool vec) {
  // valid = number of in-range elements (0..4) starting at p
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid >= 4 && vec) return *reinterpret_cast<const float4*>(p);
  if (valid > 0) v.x = p[0];
  if (valid > 1) v.y = p[1];
  if (valid > 2) v.z = p[2];
  if (valid > 3) v.w = p[3];
  return v;
}

// ---- tile loaders: 4 float4 per thread per operand per K-slab -------------------------------------
// KC: tile[row][k], row = r0 + (f >> 3), k = k0 + 4*(f & 7);   f = tid + 256*i
template <bool KC>
struct TileLoader {
  const float* base[4];   // KC: row base pointers (gather resolved once)
  int rowok[4];
  __device__ __forceinline__ void init(const float* src, int ld, int r0, int nrows,
                                       const int64_t* gather) {
    if (KC) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int f = threadIdx.x + GNT * i;
        int r = r0 + (f >> 3);
        rowok[i] = r < nrows;
        int64_t gr = rowok[i] ? (gather ? gather[r] : (int64_t)r) : 0;
        base[i] = src + gr * ld + 4 * (f & 7);
      }
    }
  }
  // KC load: kleft = contraction elements remaining from k0
  __device__ __forceinline__ void load_kc(float4 (&v)[4], int k0, int kend, bool vec) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int f = threadIdx.x + GNT * i;
      int k = k0 + 4 * (f & 7);
      int valid = rowok[i] ? (kend - k) : 0;
      v[i] = ld4(base[i] + k0, valid, vec);
    }
  }
  // RC load: tile[k][row], k = k0 + (f >> 5), row = r0 + 4*(f & 31)
  __device__ __forceinline__ void load_rc(float4 (&v)[4], const float* src, int ld, int r0, int nrows,
                                          int k0, int kend, const int64_t* kgather, bool vec) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int f = threadIdx.x + GNT * i;
      int k = k0 + (f >> 5);
      int r = r0 + 4 * (f & 31);
      int valid = (k < kend) ? (nrows - r) : 0;
      int64_t gk = (k < kend) ? (kgather ? kgather[k] : (int64_t)k) : 0;
      v[i] = ld4(src + gk * ld + r, valid, vec);
    }
  }
  __device__ __forceinline__ void store(float* tile, const float4 (&v)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int f = threadIdx.x + GNT * i;
      if (KC) *reinterpret_cast<float4*>(tile + (f >> 3) * KS + 4 * (f & 7)) = v[i];
      else    *reinterpret_cast<float4*>(tile + (f >> 5) * RS + 4 * (f & 31)) = v[i];
    }
  }
};

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void mma_slab(f32x16 (&acc)[2][2], const float* __restrict__ As,
                                         const float* __restrict__ Bs, int wr, int wc, int lane) {
  const int l31 = lane & 31;
  const int kh = (lane >> 5) * 4;
#pragma unroll
  for (int kg = 0; kg < BK; kg += 8) {
    float a[2][4], b[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (A_KC) {
        const float4 v = *reinterpret_cast<const float4*>(As + (wr * 64 + t * 32 + l31) * KS + kg + kh);
        a[t][0] = v.x; a[t][1] = v.y; a[t][2] = v.z; a[t][3] = v.w;
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) a[t][s] = As[(kg + kh + s) * RS + wr * 64 + t * 32 + l31];
      }
      if (B_KC) {
        const float4 v = *reinterpret_cast<const float4*>(Bs + (wc * 64 + t * 32 + l31) * KS + kg + kh);
        b[t][0] = v.x; b[t][1] = v.y; b[t][2] = v.z; b[t][3] = v.w;
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) b[t][s] = Bs[(kg + kh + s) * RS + wc * 64 + t * 32 + l31];
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);
  }
}

__device__ __forceinline__ float apply_act(float v, int act, float lo, float hi) {
  if (act == EVAE_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  if (act == EVAE_ACT_HARDTANH) return fminf(fmaxf(v, lo), hi);
  return v;
}

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(GNT, 2) void gemm_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  auto As = [&](int b) -> float* { return smem + (2 * b) * TILE_FLOATS; };
  auto Bs = [&](int b) -> float* { return smem + (2 * b + 1) * TILE_FLOATS; };

  // XCD-aware bijective remap: XCD x (= id % 8) works on a contiguous run of tiles
  const int ntiles = g.tiles_m * g.tiles_n;
  int tile;
  {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int qq = ntiles >> 3, rr = ntiles & 7;
    tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM;
  // EPI_GATED: a block covers 64 gated output columns; B tile rows = [wc][h|g][32]
  const int n0 = (EPI == EPI_GATED) ? tn * 64 : tn * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // flattened list of K-slabs over the (A,B) pairs
  int nslab[2];
  nslab[0] = (g.Kc[0] + BK - 1) / BK;
  nslab[1] = g.npairs > 1 ? (g.Kc[1] + BK - 1) / BK : 0;
  int s_begin = 0, s_end = nslab[0] + nslab[1];
  if (g.ksplit > 0) {
    s_begin = blockIdx.z * g.ksplit;
    int e = s_begin + g.ksplit;
    if (e < s_end) s_end = e;
  }

  TileLoader<A_KC> la[2];
  TileLoader<B_KC> lb[2];
  bool veca[2], vecb[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (p < g.npairs) {
      veca[p] = ((g.lda[p] & 3) == 0) && (((uintptr_t)g.A[p] & 15) == 0);
      vecb[p] = ((g.ldb[p] & 3) == 0) && (((uintptr_t)g.B[p] & 15) == 0);
      la[p].init(g.A[p], g.lda[p], m0, g.M, g.a_rows);
      if (EPI != EPI_GATED) lb[p].init(g.B[p], g.ldb[p], n0, g.N, nullptr);
    }
  }
  // EPI_GATED B tile: LDS row r -> weight row n0 + (r>>6)*32 + (r&31) of (r&32 ? Bg : B[0])
  const float* gb_base[4];
  int gb_ok[4];
  bool vecg = true;
  if (EPI == EPI_GATED) {
    vecg = ((g.ldb[0] & 3) == 0) && ((((uintptr_t)g.B[0] | (uintptr_t)g.Bg) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int f = threadIdx.x + GNT * i;
      int r = f >> 3;
      int n = n0 + (r >> 6) * 32 + (r & 31);
      gb_ok[i] = n < g.N;
      const float* w = (r & 32) ? g.Bg : g.B[0];
      gb_base[i] = w + (size_t)(gb_ok[i] ? n : 0) * g.ldb[0] + 4 * (f & 7);
    }
  }

  auto load_slab = [&](int s, float4 (&ra)[4], float4 (&rb)[4]) {
    const int p = (s < nslab[0]) ? 0 : 1;
    const int k0 = (p == 0 ? s : s - nslab[0]) * BK;
    const int kend = g.Kc[p];
    if (A_KC) la[p].load_kc(ra, k0, kend, veca[p]);
    else la[p].load_rc(ra, g.A[p], g.lda[p], m0, g.M, k0, kend, nullptr, veca[p]);
    if (EPI == EPI_GATED) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int f = threadIdx.x + GNT * i;
        int k = k0 + 4 * (f & 7);
        rb[i] = ld4(gb_base[i] + k0, gb_ok[i] ? (kend - k) : 0, vecg);
      }
    } else if (B_KC) {
      lb[p].load_kc(rb, k0, kend, vecb[p]);
    } else {
      lb[p].load_rc(rb, g.B[p], g.ldb[p], n0, g.N, k0, kend, g.b_krows, vecb[p]);
    }
  };

  if (s_begin < s_end) {
    float4 ra[4], rb[4];
    load_slab(s_begin, ra, rb);
    la[0].store(As(0), ra);
    lb[0].store(Bs(0), rb);
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
      const int cur = (s - s_begin) & 1;
      const bool more = s + 1 < s_end;
      if (more) load_slab(s + 1, ra, rb);
      mma_slab<A_KC, B_KC>(acc, As(cur), Bs(cur), wr, wc, lane);
      if (more) {
        la[0].store(As(cur ^ 1), ra);
        lb[0].store(Bs(cur ^ 1), rb);
      }
      __syncthreads();
    }
  }

  // ---- epilogue.  acc[mt][nt][r] <-> row m0 + wr*64 + mt*32 + (r&3) + 8*(r>>2) + 4*(lane>>5),
  //                                    col (within wave tile) nt*32 + (lane&31)
  const int l31 = lane & 31, lh = lane >> 5;
  if (EPI == EPI_GATED) {
    const int n = n0 + wc * 32 + l31;
    if (n < g.N) {
      const float bh = g.bias0 ? g.bias0[n] : 0.f;
      const float bg = g.bias1 ? g.bias1[n] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < g.M) {
            const float h = acc[mt][0][r] + bh;
            const float s = 1.0f / (1.0f + expf(-(acc[mt][1][r] + bg)));
            const size_t o = (size_t)m * g.ldo + n;
            g.out0[o] = h * s;
            if (g.out1) g.out1[o] = h;
            if (g.out2) g.out2[o] = s;
          }
        }
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = n0 + wc * 64 + nt * 32 + l31;
      if (n >= g.N) continue;
      const float bias = (EPI == EPI_LINEAR && g.bias0) ? g.bias0[n] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m >= g.M) continue;
          const size_t o = (size_t)m * g.ldo + n;
          const float v = acc[mt][nt][r];
          if (EPI == EPI_LINEAR) {
            const float pre = v + bias;
            if (g.out1) g.out1[o] = pre;
            g.out0[o] = apply_act(pre, g.act, g.lo, g.hi);
          } else if (EPI == EPI_GATE_BWD) {
            const float h = g.e0[o], s = g.e1[o];
            g.out0[o] = v * s;                       // dh
            g.out1[o] = v * h * s * (1.0f - s);      // dg
          } else {                                    // EPI_PARTIAL
            g.out0[(size_t)blockIdx.z * g.M * g.ldo + o] = v;
          }
        }
    }
  }
}

// out[i] = (accumulate ? out[i] : 0) + sum_z part[z][i]
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int nsplit, size_t n,
                                     float* __restrict__ out, int accumulate) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? out[i] : 0.f;
  for (int z = 0; z < nsplit; ++z) s += part[(size_t)z * n + i];
  out[i] = s;
}

// db[n] = sum_m dy[m][n]: 64 columns x 4 row-slices per block, then a second pass over row blocks.
constexpr int CS_ROWS = 512;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ dy, int M, int N,
                                                             float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rs = threadIdx.x >> 6;
  const int mbeg = blockIdx.y * CS_ROWS;
  int mend = mbeg + CS_ROWS;
  if (mend > M) mend = M;
  float s = 0.f;
  if (c < N)
    for (int m = mbeg + rs; m < mend; m += 4) s += dy[(size_t)m * N + c];
  red[rs][threadIdx.x & 63] = s;
  __syncthreads();
  if (rs == 0 && c < N)
    part[(size_t)blockIdx.y * N + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void gated_bwd_input_kernel(const float* __restrict__ dout, const float* __restrict__ h,
                                       const float* __restrict__ s, size_t n, float* __restrict__ dh,
                                       float* __restrict__ dg) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float d = dout[i], hv = h[i], sv = s[i];
    dh[i] = d * sv;
    dg[i] = d * hv * sv * (1.0f - sv);
  }
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ yp, size_t n,
                               int act, float lo, float hi, float* __restrict__ dpre) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float d = dy[i], v = yp[i];
    float r = d;
    if (act == EVAE_ACT_SIGMOID) r = d * v * (1.0f - v);                 // v = y
    else if (act == EVAE_ACT_HARDTANH) r = (v > lo && v < hi) ? d : 0.f;  // v = pre-activation
    dpre[i] = r;
  }
}

constexpr size_t GEMM_LDS = 4 * TILE_FLOATS * sizeof(float);  // 73,728 B

template <bool A_KC, bool B_KC, int EPI>
static int launch_gemm(GemmArgs& g, int nz, hipStream_t stream, const char* what) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<A_KC, B_KC, EPI>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS);
    attr = true;
  }
  g.tiles_m = cdiv(g.M, BM);
  g.tiles_n = cdiv(g.N, EPI == EPI_GATED ? 64 : BN);
  dim3 grid(g.tiles_m * g.tiles_n, 1, nz);
  gemm_kernel<A_KC, B_KC, EPI><<<grid, GNT, GEMM_LDS, stream>>>(g);
  return check_launch(what);
}

static int elt_grid(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b < 4096 ? (b ? b : 1) : 4096);
}

static void wgrad_split(int M, int N, int K, int* nz, int* ksplit) {
  int tiles = cdiv(N, BM) * cdiv(K, BN);
  int slabs = cdiv(M, BK);
  int want = cdiv(1024, tiles);           // ~4 blocks per CU in flight
  if (want > slabs) want = slabs;
  if (want < 1) want = 1;
  *ksplit = cdiv(slabs, want);
  *nz = cdiv(slabs, *ksplit);
}

}  // namespace evae

using namespace evae;

extern "C" int evae_gated_dense_fwd(const float* x, const int64_t* rows, int M, int K, int ldx,
                                    const float* wh, const float* bh, const float* wg, const float* bg,
                                    int N, float* out, float* save_h, float* save_s,
                                    evae_stream_t stream_) {
  EVAE_REQUIRE(M >= 0 && K > 0 && N > 0 && ldx >= K, "gated_dense_fwd: bad sizes M=%d K=%d N=%d ldx=%d", M, K, N, ldx);
  if (M == 0) return EVAE_OK;
  EVAE_REQUIRE(x && wh && wg && out, "gated_dense_fwd: null pointer");
  GemmArgs g = {};
  g.A[0] = x; g.B[0] = wh; g.Bg = wg; g.lda[0] = ldx; g.ldb[0] = K; g.Kc[0] = K; g.npairs = 1;
  g.a_rows = rows; g.M = M; g.N = N; g.bias0 = bh; g.bias1 = bg;
  g.out0 = out; g.out1 = save_h; g.out2 = save_s; g.ldo = N;
  return launch_gemm<true, true, EPI_GATED>(g, 1, (hipStream_t)stream_, "gated_dense_fwd");
}

extern "C" int evae_linear_fwd(const float* x, const int64_t* rows, int M, int K, int ldx,
                               const float* w, const float* b, int N, int act, float act_lo,
                               float act_hi, float* y, float* pre, evae_stream_t stream_) {
  EVAE_REQUIRE(M >= 0 && K > 0 && N > 0 && ldx >= K, "linear_fwd: bad sizes M=%d K=%d N=%d ldx=%d", M, K, N, ldx);
  EVAE_REQUIRE(act >= 0 && act <= 2, "linear_fwd: bad activation %d", act);
  if (M == 0) return EVAE_OK;
  EVAE_REQUIRE(x && w && y, "linear_fwd: null pointer");
  GemmArgs g = {};
  g.A[0] = x; g.B[0] = w; g.lda[0] = ldx; g.ldb[0] = K; g.Kc[0] = K; g.npairs = 1;
  g.a_rows = rows; g.M = M; g.N = N; g.bias0 = b; g.out0 = y; g.out1 = pre; g.ldo = N;
  g.act = act; g.lo = act_lo; g.hi = act_hi;
  return launch_gemm<true, true, EPI_LINEAR>(g, 1, (hipStream_t)stream_, "linear_fwd");
}

extern "C" int evae_dense_bwd_data(const float* dy1, const float* w1, const float* dy2, const float* w2,
                                   int M, int N, int K, const float* h_prev, const float* s_prev,
                                   float* dx_or_dh, float* dg, evae_stream_t stream_) {
  EVAE_REQUIRE(M >= 0 && N > 0 && K > 0, "dense_bwd_data: bad sizes M=%d N=%d K=%d", M, N, K);
  if (M == 0) return EVAE_OK;
  EVAE_REQUIRE(dy1 && w1 && dx_or_dh, "dense_bwd_data: null pointer");
  EVAE_REQUIRE((dy2 == nullptr) == (w2 == nullptr), "dense_bwd_data: dy2/w2 must come together");
  const bool gate = h_prev != nullptr;
  EVAE_REQUIRE(!gate || (s_prev && dg), "dense_bwd_data: gate fusion needs h_prev, s_prev and dg");
  GemmArgs g = {};
  g.A[0] = dy1; g.B[0] = w1; g.lda[0] = N; g.ldb[0] = K; g.Kc[0] = N; g.npairs = 1;
  if (dy2) { g.A[1] = dy2; g.B[1] = w2; g.lda[1] = N; g.ldb[1] = K; g.Kc[1] = N; g.npairs = 2; }
  g.M = M; g.N = K; g.out0 = dx_or_dh; g.out1 = dg; g.ldo = K; g.e0 = h_prev; g.e1 = s_prev;
  if (gate) return launch_gemm<true, false, EPI_GATE_BWD>(g, 1, (hipStream_t)stream_, "dense_bwd_data(gate)");
  g.out1 = nullptr;
  return launch_gemm<true, false, EPI_LINEAR>(g, 1, (hipStream_t)stream_, "dense_bwd_data");
}

extern "C" size_t evae_dense_bwd_weight_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 256;
  int nz, ks;
  wgrad_split(M, N, K, &nz, &ks);
  size_t part = (size_t)nz * N * K * sizeof(float);
  size_t cs = (size_t)cdiv(M, CS_ROWS) * N * sizeof(float);
  return align_up(part, 256) + align_up(cs, 256) + 256;
}

extern "C" int evae_dense_bwd_weight(const float* dy, int M, int N, const float* x, const int64_t* rows,
                                     int K, int ldx, float* dw, float* db, int accumulate, void* ws,
                                     size_t ws_bytes, evae_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  EVAE_REQUIRE(M >= 0 && N > 0 && K > 0 && ldx >= K, "dense_bwd_weight: bad sizes M=%d N=%d K=%d", M, N, K);
  EVAE_REQUIRE(dw != nullptr, "dense_bwd_weight: null dw");
  if (ws == nullptr || ws_bytes < evae_dense_bwd_weight_workspace_bytes(M, N, K)) {
    set_error("dense_bwd_weight: workspace too small (%zu)", ws_bytes);
    return EVAE_EWORKSPACE;
  }
  if (M == 0) {
    if (!accumulate) {
      (void)hipMemsetAsync(dw, 0, (size_t)N * K * sizeof(float), stream);
      if (db) (void)hipMemsetAsync(db, 0, (size_t)N * sizeof(float), stream);
    }
    return check_launch("dense_bwd_weight(empty)");
  }
  EVAE_REQUIRE(dy && x, "dense_bwd_weight: null pointer");
  int nz, ks;
  wgrad_split(M, N, K, &nz, &ks);
  float* part = (float*)ws;
  float* cs = (float*)((char*)ws + align_up((size_t)nz * N * K * sizeof(float), 256));
  GemmArgs g = {};
  g.A[0] = dy; g.B[0] = x; g.lda[0] = N; g.ldb[0] = ldx; g.Kc[0] = M; g.npairs = 1;
  g.b_krows = rows; g.M = N; g.N = K; g.ksplit = ks; g.out0 = part; g.ldo = K;
  int rc = launch_gemm<false, false, EPI_PARTIAL>(g, nz, stream, "dense_bwd_weight");
  if (rc) return rc;
  size_t n = (size_t)N * K;
  splitk_reduce_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(part, nz, n, dw, accumulate);
  rc = check_launch("splitk_reduce");
  if (rc || !db) return rc;
  int nb = cdiv(M, CS_ROWS);
  colsum_partial_kernel<<<dim3(cdiv(N, 64), nb), 256, 0, stream>>>(dy, M, N, cs);
  rc = check_launch("colsum_partial");
  if (rc) return rc;
  splitk_reduce_kernel<<<cdiv(N, 256), 256, 0, stream>>>(cs, nb, (size_t)N, db, accumulate);
  return check_launch("colsum_reduce");
}

extern "C" int evae_gated_dense_bwd_input(const float* dout, const float* h, const float* s, size_t n,
                                          float* dh, float* dg, evae_stream_t stream_) {
  if (n == 0) return EVAE_OK;
  EVAE_REQUIRE(dout && h && s && dh && dg, "gated_dense_bwd_input: null pointer");
  gated_bwd_input_kernel<<<elt_grid(n), 256, 0, (hipStream_t)stream_>>>(dout, h, s, n, dh, dg);
  return check_launch("gated_dense_bwd_input");
}

extern "C" int evae_act_bwd(const float* dy, const float* y_or_pre, size_t n, int act, float act_lo,
                            float act_hi, float* dpre, evae_stream_t stream_) {
  if (n == 0) return EVAE_OK;
  EVAE_REQUIRE(dy && dpre && (act == EVAE_ACT_NONE || y_or_pre), "act_bwd: null pointer");
  EVAE_REQUIRE(act >= 0 && act <= 2, "act_bwd: bad activation %d", act);
  act_bwd_kernel<<<elt_grid(n), 256, 0, (hipStream_t)stream_>>>(dy, y_or_pre ? y_or_pre : dy, n, act, act_lo, act_hi, dpre);
  return check_launch("act_bwd");
}
