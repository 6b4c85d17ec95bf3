// experiment: where does global_load_lds_dwordx4 put the lanes' 16 bytes?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* g, unsigned* out) {
  extern __shared__ unsigned buf[];
  typedef __attribute__((address_space(1))) const void* GP;
  typedef __attribute__((address_space(3))) void* LP;
  __builtin_amdgcn_global_load_lds((GP)(g + threadIdx.x * 4), (LP)buf, 16, 0, 0);
  __builtin_amdgcn_global_load_lds((GP)(g + 256 + threadIdx.x * 4), (LP)(buf + 256), 16, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = buf[i];
}
int main() {
  unsigned h[512], *g, *o;
  for (int i = 0; i < 512; i++) h[i] = i;
  hipMalloc(&g, sizeof h); hipMalloc(&o, sizeof h);
  hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, g, o);
  unsigned r[512];
  hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 512; i++) if (r[i] != (unsigned)i) bad++;
  printf("mismatches %d; first 16 LDS dwords:", bad);
  for (int i = 0; i < 16; i++) printf(" %u", r[i]);
  printf("\ndwords 256..271:");
  for (int i = 256; i < 272; i++) printf(" %u", r[i]);
  printf("\n");
  return 0;
}
