// Probe of LDS-DMA semantics on gfx950 (global_load_lds_dwordx4 through the builtin):
//  (1) destination = wave-uniform LDS base + lane*16, source address per lane
//  (2) lanes masked by EXEC write nothing
//  (3) completion is covered by s_waitcnt vmcnt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void probe(const unsigned* src, unsigned* out, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < 4096; k += blockDim.x) lds[k] = 0xdeadbeefu;
  __syncthreads();
  // each wave copies 1 KiB: lane l fetches source chunk (63 - l) (reversed) into slot l
  unsigned* dst = lds + wave * 256;
  const unsigned* s = src + wave * 256 + (63 - lane) * 4;
  if (mode == 0 || lane < 40) {
    __builtin_amdgcn_global_load_lds(s, (lds_void*)dst, 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int k = threadIdx.x; k < 1024; k += blockDim.x) out[k] = lds[k];
}
int main() {
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i;
  unsigned *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    probe<<<1, 256, 16384>>>(d, o, mode);
    std::vector<unsigned> r(1024);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0, untouched = 0;
    for (int w = 0; w < 4; ++w)
      for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
          unsigned got = r[w * 256 + l * 4 + k], exp = w * 256 + (63 - l) * 4 + k;
          if (mode == 1 && l >= 40) { if (got == 0xdeadbeefu) untouched++; else bad++; }
          else if (got != exp) bad++;
        }
    printf("mode %d: bad %d untouched %d (expect bad 0%s)\n", mode, bad, untouched, mode ? ", untouched 384" : "");
  }
  return 0;
}
