// experiment: many workgroups per CU, each filling a 36 KB LDS strip with global_load_lds_dwordx4 (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* g, unsigned long long* bad, int rounds) {
  extern __shared__ double buf[];
  typedef __attribute__((address_space(1))) const void* GP;
  typedef __attribute__((address_space(3))) void* LP;
  const int lane = threadIdx.x;
  unsigned long long nb = 0;
  for (int r = 0; r < rounds; r++) {
    const double* row = g + ((size_t)blockIdx.x * rounds + r) * 4608;
    for (int kk = 0; kk < 36; kk++)
      __builtin_amdgcn_global_load_lds((GP)(row + 2 * lane + kk * 128), (LP)(buf + kk * 128), 16, 0, 0);
    __syncthreads();
    for (int i = 0; i < 72; i++)
      if (buf[i * 64 + lane] != row[i * 64 + lane]) nb++;
    __syncthreads();
  }
  if (nb) atomicAdd(bad, nb);
}
int main() {
  const int blocks = 4096, rounds = 4;
  const size_t n = (size_t)blocks * rounds * 4608;
  std::vector<double> h(n);
  for (size_t i = 0; i < n; i++) h[i] = (double)i;
  double* g; unsigned long long* bad, hb = 0;
  if (hipMalloc(&g, n * 8) || hipMalloc(&bad, 8)) return 1;
  (void)hipMemcpy(g, h.data(), n * 8, hipMemcpyHostToDevice);
  (void)hipMemset(bad, 0, 8);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 36864, 0, g, bad, rounds);
  (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  printf("mismatching doubles: %llu of %zu (%s)\n", hb, n, hipGetErrorString(hipGetLastError()));
  return 0;
}
