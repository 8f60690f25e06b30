// Does global_load_lds_dwordx4 leave the LDS slots of EXEC-masked lanes untouched on gfx950?  (The persistent whole-sequence
// attention kernel keeps a constant "ones" column in V's LDS image and lets the LDS-DMA skip it.)  Build:
//   hipcc -O3 --offload-arch=gfx950 tools/gll_probe.hip -o tools/gll_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const unsigned* g, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned s[64 * 4];
  const int lane = threadIdx.x & 63;
  for (int i = 0; i < 4; ++i) s[lane * 4 + i] = 0x55555555u;
  __syncthreads();
  if (lane & 1) __builtin_amdgcn_global_load_lds((const void*)(g + lane * 4), (lds_ptr_t)s, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = s[lane * 4 + i];
}
int main() {
  unsigned h[256], *g, *o;
  for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
  hipMalloc(&g, 1024); hipMalloc(&o, 1024);
  hipMemcpy(g, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o);
  hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
  int bad_active = 0, touched_inactive = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      const unsigned v = h[l * 4 + i];
      if (l & 1) bad_active += (v != 1000u + l * 4 + i);
      else touched_inactive += (v != 0x55555555u);
    }
  printf("active lanes wrong: %d; inactive lanes' LDS slots overwritten: %d  (lane 0 slot: %08x %08x, lane 1 slot: %u %u)\n", bad_active,
         touched_inactive, h[0], h[1], h[4], h[5]);
  return 0;
}
