// Microbenchmark (round 4): what does a launch of 64 one-wave workgroups cost as a function of the DYNAMIC LDS each workgroup asks for?
// (pih_fly_step_kernel asks for 120 KB per wave; its waves are alive 110 us of a 163-us launch.)   hipcc --offload-arch=gfx950 -O3 -o lds_launch lds_launch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float lds[];
__global__ void __launch_bounds__(64, 1) k_touch(float* out, int words, int spin) {
  float acc = 0;
  for (int i = threadIdx.x; i < words; i += 64 * 64) lds[i] = (float)i;      // touch a little of it
  __syncthreads();
  for (int s = 0; s < spin; s++) acc = acc * 1.0001f + lds[(threadIdx.x + s) % (words > 0 ? words : 1)];
  if (acc == 123.456f) out[blockIdx.x] = acc;
}
int main() {
  float* out; hipMalloc(&out, 4096 * sizeof(float));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int sizes[] = {0, 16 << 10, 48 << 10, 64 << 10, 65 << 10, 96 << 10, 120 << 10, 160 << 10};
  for (int blocks : {64, 256, 1024}) {
    for (int sz : sizes) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_touch), hipFuncAttributeMaxDynamicSharedMemorySize, sz) != hipSuccess) { printf("blocks %d lds %d: attribute refused\n", blocks, sz); continue; }
      for (int spin : {0, 20000}) {
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(64), sz, 0, out, sz / 4, spin);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(64), sz, 0, out, sz / 4, spin);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        hipError_t e = hipGetLastError();
        printf("blocks %4d  dynamic LDS %6d B  spin %5d: %.2f us per launch  %s\n", blocks, sz, spin, ms * 10.0f, e == hipSuccess ? "" : hipGetErrorString(e));
      }
    }
  }
  return 0;
}
