// What does an LDS-DMA copy (buffer_load_dwordx4 ... lds) write for a lane whose offset lies beyond the buffer resource's
// num_records?  (The window loader of csrc/evae_conv_win.h parks the zero padding of a convolution on such offsets.)
// Also: per-lane gather addresses with a lane-linear LDS destination.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -I exemplar-vae_amd/csrc tools/micro/dma_oob.hip -o tools/micro/dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lds_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // lane l gathers 16 bytes from slot (63 - l) -- reversed -- and odd lanes are parked out of range
  const unsigned voff = (lane & 1) ? 0x80000000u : (unsigned)(63 - lane) * 16u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_t)lds, 16, voff, 0, 0, 0);
  // second piece: offsets just beyond num_records (nbytes = 1024: slot 64 + l)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_t)(lds + 256), 16, (unsigned)(lane < 32 ? lane : 64 + lane) * 16u, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x1000u + i;
  unsigned *d, *o;
  CK(hipMalloc(&d, 4096)); CK(hipMalloc(&o, 2048));
  CK(hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice));
  k<<<1, 64>>>(d, 1024, o);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> r(512);
  CK(hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const unsigned want = (l & 1) ? 0u : 0x1000u + (63 - l) * 4 + j;
      if (r[l * 4 + j] != want) { if (bad < 8) printf("piece0 lane %d dword %d: got %08x want %08x\n", l, j, r[l * 4 + j], want); ++bad; }
      const unsigned want2 = l < 32 ? 0x1000u + l * 4 + j : 0u;
      if (r[256 + l * 4 + j] != want2) { if (bad < 8) printf("piece1 lane %d dword %d: got %08x want %08x\n", l, j, r[256 + l * 4 + j], want2); ++bad; }
    }
  printf("dma_oob: %s (%d mismatches): out-of-range lanes of an LDS-DMA copy %s\n", bad ? "FAIL" : "ok", bad, bad ? "do NOT write zeros" : "write zeros; gather addresses land lane-linear");
  return bad ? 1 : 0;
}
