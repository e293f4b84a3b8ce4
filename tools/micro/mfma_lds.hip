// Which ingredient of the GEMM inner loop costs MFMA issue slots?  Progressive variants of the loop
// (8 waves/block, 2 blocks/CU, wave tile 32x64, fragments via ds_read_b128) with no global traffic.
// hipcc -O3 --offload-arch=gfx950 -w tools/micro/mfma_lds.hip -o tools/micro/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// accumulators pinned to AccVGPRs: MFMA results then do not share the ArchVGPR write port with LDS / VMEM / VALU results
#define MFMA_A(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
constexpr int KS = 36, BM = 128, BN = 128;

// MODE bit0: LDS fragment reads, bit1: barrier per slab, bit2: pipelined fragment reads (2 register sets)
__global__ void fill_kernel(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xffff) / 65536.0f - 0.5f;
  }
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void loop_kernel(const float* __restrict__ src, float* out, int slabs,
                                                      const float* __restrict__ gsrc = nullptr, size_t gmask = 0,
                                                      int rstride = 0) {
  extern __shared__ float smem[];
  float* As = smem;
  float* Bs = smem + 2 * BM * KS;
  for (int i = threadIdx.x; i < 2 * BM * KS + 2 * BN * KS; i += 512) smem[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1;
  const int l31 = lane & 31, kh = (lane >> 5) * 4;
  float v0 = lane, v1 = w, v2 = lane * 2, v3 = w * 3;
  f32x16 acc[2];
  for (int j = 0; j < 16; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  float4 fa = *reinterpret_cast<const float4*>(As + (wr * 32 + l31) * KS + kh);
  float4 fb0 = *reinterpret_cast<const float4*>(Bs + (wc * 64 + l31) * KS + kh);
  float4 fb1 = *reinterpret_cast<const float4*>(Bs + (wc * 64 + 32 + l31) * KS + kh);
  float4 g0 = make_float4(0, 0, 0, 0), g1 = g0, g2 = g0, g3 = g0;
  float4 h0 = g0, h1 = g0, h2 = g0, h3 = g0;
  // per-thread float4 stream: block b, slab s -> 4 x (512 threads x 16 B) = 32 KB per slab, wrapped by gmask
  size_t gpos = ((size_t)blockIdx.x * 977 * 8192 + threadIdx.x * 4);
  auto body = [&](int s, float4& g0, float4& g1, float4& g2, float4& g3) __attribute__((always_inline)) {
    const int cur = s & 1;
    if ((MODE & 64) && !(MODE & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // asm loads: the compiler does not track them
    if (MODE & 32) {   // park the previous slab's loads in the idle LDS buffer (waits on vmcnt)
      float* d = smem + (cur ^ 1) * BM * KS + (threadIdx.x >> 3) * KS + 4 * (threadIdx.x & 7);
      *reinterpret_cast<float4*>(d) = g0;
      *reinterpret_cast<float4*>(d + 64 * KS) = g1;
      float* e = Bs + (cur ^ 1) * BN * KS + (threadIdx.x >> 3) * KS + 4 * (threadIdx.x & 7);
      *reinterpret_cast<float4*>(e) = g2;
      *reinterpret_cast<float4*>(e + 64 * KS) = g3;
    }
    if (MODE & 512) {   // same tile pattern, straight into the idle LDS buffer: wave w fills rows 8w.. of each 64-row half
      const size_t rowA = (size_t)(blockIdx.x / 5) * 128 + (threadIdx.x >> 3);
      const size_t rowB = (size_t)(blockIdx.x % 5) * 128 + (threadIdx.x >> 3);
      const float* qa = gsrc + rowA * rstride + 4 * (threadIdx.x & 7) + (s % 24) * 32;
      const float* qb = gsrc + ((size_t)1 << 27) + rowB * rstride + 4 * (threadIdx.x & 7) + (s % 24) * 32;
      typedef __attribute__((address_space(3))) void* lds_t;
      float* da = smem + (cur ^ 1) * BM * KS + w * 256;          // 1 KB per wave per instruction, unpadded
      float* db = Bs + (cur ^ 1) * BN * KS + w * 256;
      __builtin_amdgcn_global_load_lds(qa, (lds_t)da, 16, 0, 0);
      __builtin_amdgcn_global_load_lds(qa + 64 * (size_t)rstride, (lds_t)(da + 2048), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(qb, (lds_t)db, 16, 0, 0);
      __builtin_amdgcn_global_load_lds(qb + 64 * (size_t)rstride, (lds_t)(db + 2048), 16, 0, 0);
    } else if (MODE & 64) {   // tile pattern of the GEMM: 8 lanes per 128-B row segment, rows rstride floats apart
      const size_t rowA = (size_t)(blockIdx.x / 5) * 128 + (threadIdx.x >> 3);
      const size_t rowB = (size_t)(blockIdx.x % 5) * 128 + (threadIdx.x >> 3);
      const float* qa = gsrc + rowA * rstride + 4 * (threadIdx.x & 7) + (s % 24) * 32;
      const float* qb = gsrc + ((size_t)1 << 27) + rowB * rstride + 4 * (threadIdx.x & 7) + (s % 24) * 32;
      if (MODE & 32) {   // consumed by the LDS stores: plain loads, the compiler places the vmcnt waits
        if (MODE & 2048) {   // same slab every time: L1/L2-resident
          qa -= (s % 24) * 32; qb -= (s % 24) * 32;
        }
        if (MODE & 4096) {        // half the bytes: dwordx2
          const float2 a0 = *reinterpret_cast<const float2*>(qa), a1 = *reinterpret_cast<const float2*>(qa + 64 * (size_t)rstride);
          const float2 b0 = *reinterpret_cast<const float2*>(qb), b1 = *reinterpret_cast<const float2*>(qb + 64 * (size_t)rstride);
          g0.x = a0.x; g0.y = a0.y; g1.x = a1.x; g1.y = a1.y; g2.x = b0.x; g2.y = b0.y; g3.x = b1.x; g3.y = b1.y;
        } else if (MODE & 8192) {  // a quarter: dword
          g0.x = qa[0]; g1.x = qa[64 * (size_t)rstride]; g2.x = qb[0]; g3.x = qb[64 * (size_t)rstride];
        } else if (MODE & 16384) { // two instructions instead of four
          g0 = *reinterpret_cast<const float4*>(qa);
          g2 = *reinterpret_cast<const float4*>(qb);
        } else {
        g0 = *reinterpret_cast<const float4*>(qa);
        g1 = *reinterpret_cast<const float4*>(qa + 64 * (size_t)rstride);
        g2 = *reinterpret_cast<const float4*>(qb);
        g3 = *reinterpret_cast<const float4*>(qb + 64 * (size_t)rstride);
        }
      } else {
      // asm volatile so that unconsumed loads still execute every slab
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g0) : "v"(qa));
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g1) : "v"(qa + 64 * (size_t)rstride));
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g2) : "v"(qb));
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g3) : "v"(qb + 64 * (size_t)rstride));
      }
    } else if (MODE & 16) {
      const float* q = gsrc + (gpos & gmask);
      g0 = *reinterpret_cast<const float4*>(q);
      g1 = *reinterpret_cast<const float4*>(q + 2048);
      g2 = *reinterpret_cast<const float4*>(q + 4096);
      g3 = *reinterpret_cast<const float4*>(q + 6144);
      gpos += 8192;
    }
    const float* a = As + cur * BM * KS + (wr * 32 + l31) * KS + kh;
    const float* b = Bs + cur * BN * KS + (wc * 64 + l31) * KS + kh;
    if (MODE & 4) {
      float4 ga = *reinterpret_cast<const float4*>(a), gb0 = *reinterpret_cast<const float4*>(b),
             gb1 = *reinterpret_cast<const float4*>(b + 32 * KS);
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        float4 na, nb0, nb1;
        if (kg < 3) {
          na = *reinterpret_cast<const float4*>(a + (kg + 1) * 8);
          nb0 = *reinterpret_cast<const float4*>(b + (kg + 1) * 8);
          nb1 = *reinterpret_cast<const float4*>(b + 32 * KS + (kg + 1) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 256) {
          MFMA_A(acc[0], ga.x, gb0.x); MFMA_A(acc[1], ga.x, gb1.x);
          MFMA_A(acc[0], ga.y, gb0.y); MFMA_A(acc[1], ga.y, gb1.y);
          MFMA_A(acc[0], ga.z, gb0.z); MFMA_A(acc[1], ga.z, gb1.z);
          MFMA_A(acc[0], ga.w, gb0.w); MFMA_A(acc[1], ga.w, gb1.w);
        } else {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.x, gb0.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.x, gb1.x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.y, gb0.y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.y, gb1.y, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.z, gb0.z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.z, gb1.z, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.w, gb0.w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga.w, gb1.w, acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kg < 3) { ga = na; gb0 = nb0; gb1 = nb1; }
      }
    } else {
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        if (MODE & 1) {
          fa = *reinterpret_cast<const float4*>(a + kg * 8);
          fb0 = *reinterpret_cast<const float4*>(b + kg * 8);
          fb1 = *reinterpret_cast<const float4*>(b + 32 * KS + kg * 8);
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb0.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb1.x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb0.y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb1.y, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb0.z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb1.z, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb0.w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb1.w, acc[1], 0, 0, 0);
      }
    }
    if (MODE & 8) {   // 80 dependent-free VALU ops per slab per wave (address / mask arithmetic stand-in)
#pragma unroll
      for (int u = 0; u < 20; ++u) {
        v0 = v0 * 1.0001f + 0.5f; v1 = v1 * 0.9999f + 0.25f; v2 = v2 * 1.0002f - 0.5f; v3 = v3 * 0.9998f - 0.25f;
      }
    }
    if (MODE & 512) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE & 2) __syncthreads();
  };
  if (MODE & 1024) {
    for (int s = 0; s < slabs; s += 2) { body(s, g0, g1, g2, g3); body(s + 1, h0, h1, h2, h3); }
  } else {
    for (int s = 0; s < slabs; ++s) body(s, g0, g1, g2, g3);
  }
  if (MODE & 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (v0 + v1 + v2 + v3 + g0.x + g1.y + g2.z + g3.w + h0.x + h1.y + h2.z + h3.w == 1.2345f) out[0] = v0;
  if (MODE & 256) asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  float sum = 0;
  for (int j = 0; j < 16; ++j) sum += acc[0][j] + acc[1][j];
  out[blockIdx.x * 512 + threadIdx.x] = sum;
}

static const float* g_gsrc = nullptr;
static size_t g_gmask = 0;
static int g_rstride = 0;
template <int MODE>
static void run(const char* name, const float* src, float* out, int nblocks, int slabs) {
  const size_t lds = (2 * BM * KS + 2 * BN * KS) * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(loop_kernel<MODE>, dim3(nblocks), dim3(512), lds, 0, src, out, slabs, g_gsrc, g_gmask, g_rstride);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(loop_kernel<MODE>, dim3(nblocks), dim3(512), lds, 0, src, out, slabs, g_gsrc, g_gmask, g_rstride);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double fl = (double)nblocks * 8 * slabs * 32 * 4096.0;
  printf("%-44s blocks=%d: %8.1f us  %6.1f TFLOP/s\n", name, nblocks, ms * 1e3, fl / ms / 1e9);
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int n = 2 * BM * KS + 2 * BN * KS;
  std::vector<float> h(n);
  const bool rnd = argc > 1;
  for (int i = 0; i < n; ++i) h[i] = rnd ? (float)rand() / RAND_MAX - 0.5f : 1.0f;
  float *src, *out;
  hipMalloc(&src, n * 4); hipMalloc(&out, 2048 * 512 * 4);
  hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  float* gs; hipMalloc(&gs, ((size_t)1 << 30) + (1 << 20)); hipMemset(gs, 0, ((size_t)1 << 30) + (1 << 20));
  g_gsrc = gs;
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, gs, ((size_t)1 << 28) + (1 << 18));
  hipDeviceSynchronize();
  printf("LDS contents: %s\n", rnd ? "random" : "ones");
  for (int nb : {512, 980}) {
    const int slabs = 250;
    run<0>("mfma only", src, out, nb, slabs);
    run<2>("mfma + barrier/slab", src, out, nb, slabs);
    run<1>("mfma + frag reads (same regs)", src, out, nb, slabs);
    run<3>("mfma + frag reads + barrier", src, out, nb, slabs);
    run<4>("mfma + pipelined frag reads", src, out, nb, slabs);
    run<6>("mfma + pipelined frag reads + barrier", src, out, nb, slabs);
    run<14>("  ... + 80 VALU ops/slab/wave", src, out, nb, slabs);
    g_rstride = 784;
    run<6 + 64 + 32>("tile loads + stores (x4, streaming)", src, out, nb, slabs);
    run<6 + 64 + 32 + 2048>("tile loads + stores (x4, L1/L2-resident)", src, out, nb, slabs);
    run<6 + 64 + 32 + 2048 + 4096>("tile loads + stores (x2, resident)", src, out, nb, slabs);
    run<6 + 64 + 32 + 2048 + 8192>("tile loads + stores (x1, resident)", src, out, nb, slabs);
    run<6 + 64 + 32 + 2048 + 16384>("tile loads + stores (2 of 4 x4 loads, resident)", src, out, nb, slabs);
    run<6 + 32>("LDS stores only (no loads)", src, out, nb, slabs);
    run<6 + 512>("direct-to-LDS tile loads (no VGPR staging)", src, out, nb, slabs);
    run<6 + 256>("AGPR acc: mfma + pipelined reads + barrier", src, out, nb, slabs);
    run<14 + 256>("AGPR acc:   ... + 80 VALU ops/slab/wave", src, out, nb, slabs);
    g_rstride = 784;
    run<6 + 64 + 32 + 256>("AGPR acc:   ... + tile loads + LDS stores", src, out, nb, slabs);
    run<14 + 64 + 32 + 256>("AGPR acc:   ... + tile loads + stores + 80 VALU", src, out, nb, slabs);
    run<14 + 64 + 32>("VGPR acc:   ... + tile loads + stores + 80 VALU", src, out, nb, slabs);
    for (int rs : {784}) {
      g_rstride = rs;
      char nm[96];
      snprintf(nm, sizeof nm, "  ... + tile-pattern loads + LDS stores, %d", rs);
      run<6 + 64 + 32>(nm, src, out, nb, slabs);
    }
    for (size_t mb : {64}) {
      g_gmask = (mb << 18) - 1;   // floats
      char nm[96];
      snprintf(nm, sizeof nm, "  ... + 4 dwordx4 loads/slab/wave, %zu MB set", mb);
      run<6 + 16>(nm, src, out, nb, slabs);
      snprintf(nm, sizeof nm, "  ... + loads + LDS stores, %zu MB set", mb);
      run<6 + 16 + 32>(nm, src, out, nb, slabs);
    }
  }
  return 0;
}
