// Microbenchmark: cycles per row update of the row-space solver's dependent chain (one wave, or two waves on one SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o rowchain rowchain.hip ; run: ./rowchain
#include <hip/hip_runtime.h>
#include <cstdio>
#define ROWS 60
template <int VAR> __global__ void __launch_bounds__(64) k(float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x;
  float z = lane * 0.001f, lam = 0.f, lb = -1.f - lane * 0.01f, ub = 1.f + lane * 0.01f;
  float B[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; i++) B[i] = 0.001f * (float)((lane * 7 + i * 13) % 17) - 0.008f;
  unsigned long long mnext = 1; float ind[4] = {lane == 0 ? 1.f : 0.f, lane == 1 ? 1.f : 0.f, lane == 2 ? 1.f : 0.f, lane == 3 ? 1.f : 0.f};
  __asm__ volatile("" : "+s"(mnext), "+v"(ind[0]), "+v"(ind[1]), "+v"(ind[2]), "+v"(ind[3]));
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < ROWS; g++) {
      const float cand = __builtin_amdgcn_fmed3f(z, lb, ub);
      const float dlv = cand - lam;
      if (VAR == 0) {          // readlane -> SGPR -> fmac
        const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlv), g));
        unsigned long long m; __asm__ volatile("s_lshl_b64 %1, 1, %3\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(lam), "=&s"(m) : "v"(cand), "n"(0) : "scc");
        z += B[g] * s;
      } else if (VAR == 1) {   // no broadcast at all (wrong maths, timing only)
        lam = cand;
        z += B[g] * dlv;
      } else if (VAR == 2) {   // readlane without the commit
        const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlv), g));
        z += B[g] * s;
      } else if (VAR == 3) {   // DPP row broadcast-ish substitute: quad_perm broadcast of lane 0 of each quad (timing of a DPP mov in the chain)
        const float s = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, dlv), 0x00, 0xF, 0xF, true));
        z += B[g] * s;
      } else if (VAR == 5) {   // commit mask made one row ahead (hides the SALU -> VALU latency)
        const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlv), g));
        __asm__ volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(lam) : "v"(cand), "s"(mnext));
        __asm__ volatile("s_lshl_b64 %0, 1, %1" : "=s"(mnext) : "n"(1) : "scc");
        z += B[g] * s;
      } else if (VAR == 6) {   // commit through v_cmp_eq (VALU -> VCC -> v_cndmask)
        const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlv), g));
        lam = (lane == g) ? cand : lam;
        z += B[g] * s;
      } else if (VAR == 7) {   // commit through v_writelane of the broadcast candidate
        const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlv), g));
        const int sc = __builtin_amdgcn_readlane(__builtin_bit_cast(int, cand), g);
        { int li = __builtin_bit_cast(int, lam); __asm__ volatile("v_writelane_b32 %0, %1, %2" : "+v"(li) : "s"(sc), "n"(0)); lam = __builtin_bit_cast(float, li); }
        z += B[g] * s;
      } else if (VAR == 8) {   // lam += dl at the row's lane through the broadcast step: lam = writelane(lam_g + s)... (readlane lam, add scalar? no SALU float) -> v_add on all lanes of a lane-masked step
        const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dlv), g));
        lam += ind[g % 4] * s;   // indicator vectors (timing only)
        z += B[g] * s;
      } else if (VAR == 4) {   // v-space chain: readlane -> fma -> med3 -> sub -> fmac
        const float vj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), g));
        float sum = lam + (ub - vj * lb); sum = __builtin_amdgcn_fmed3f(sum, lb, ub);
        const float dl = sum - lam; lam = sum;
        z += B[g] * dl;
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + lane] = z + lam;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int VAR> void run(const char* name, float* out, long long* cyc, int blocks) {
  const int iters = 200;
  hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
  hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { fprintf(stderr, "sync: %s\n", hipGetErrorString(e)); return; }
  static long long h[8192]; hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; i++) s += (double)h[i];
  fprintf(stderr, "%-44s blocks %5d: %.1f cycles per row\n", name, blocks, s / blocks / ((double)iters * ROWS));
}
int main() { fprintf(stderr, "start\n");
  float* out; long long* cyc; hipMalloc(&out, 8192 * 64 * 4); hipMalloc(&cyc, 8192 * 8);
  for (int blocks : {1, 2048}) {     // 1 wave alone; 2 / 4 / 8 waves per SIMD worth of work (1024 SIMDs)
    run<0>("z-space row (readlane + commit)", out, cyc, blocks);
    run<2>("z-space row (readlane, no commit)", out, cyc, blocks);
    run<1>("no broadcast", out, cyc, blocks);
    run<3>("DPP mov instead of readlane", out, cyc, blocks);
    run<4>("v-space row", out, cyc, blocks);
    run<5>("z-space, mask one row ahead", out, cyc, blocks);
    run<6>("z-space, commit by v_cmp_eq + cndmask", out, cyc, blocks);
    run<7>("z-space, commit by readlane + writelane", out, cyc, blocks);
    run<8>("z-space, lam += indicator * step", out, cyc, blocks);
  }
  return 0;
}
