// What v_permlane16_swap / v_permlane32_swap deliver on gfx950 (hipcc --offload-arch=gfx950 -O2 -o permlane_swap_probe permlane_swap_probe.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  const unsigned a = 100 + lane, b = 200 + lane;
  const u32x2 r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  const u32x2 r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane] = r16[0];
  out[64 + lane] = r16[1];
  out[128 + lane] = r32[0];
  out[192 + lane] = r32[1];
}
int main() {
  unsigned* d;
  unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane16_swap(a=100+lane, b=200+lane)[0]", "permlane16_swap[1]", "permlane32_swap[0]", "permlane32_swap[1]"};
  for (int v = 0; v < 4; ++v) {
    printf("%s:", names[v]);
    for (int r = 0; r < 4; ++r) printf("  row %d: %u..%u", r, h[v * 64 + r * 16], h[v * 64 + r * 16 + 15]);
    printf("\n");
  }
  return 0;
}
