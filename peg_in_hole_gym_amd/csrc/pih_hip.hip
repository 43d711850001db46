// pih_hip.hip -- kernels + C ABI (include/pih.h) of the MI355X-native peg-in-hole environment.
// gfx950 only; one wavefront (64 threads, one workgroup) per env; per-env scratch in LDS (struct pih::Shared).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libpih_hip.so pih_hip.hip   (see build.py)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "pih_device.h"
#include "pih_render.h"
#include "pih_fly.h"

using namespace pih;

// ------------------------------------------------------------------------------------------------ kernels
// state: float[n][256] (env-major records: the 64 lanes of the env's wave read/write consecutive words, so every
// access is a fully coalesced 256 B segment).
// Longest-job-first dispatch order.  The cost of an env-step grows ~linearly with its contact count (each contact is one
// 3x3 PGS block per iteration), and at 4096 envs there are only two rounds of resident waves, so a heavy env that starts
// late leaves most of the chip idle at the tail.  This single-workgroup counting sort orders the envs by the contact count
// of their PREVIOUS step (descending); pih_step_kernel maps blockIdx through it.  Results do not depend on block order.
// Launch 1 of a step.  Block 0: longest-job-first dispatch order (counting sort of the envs by their previous-step contact
// count, most contacts first).  Blocks 1..: the controller (action / state machine -> IK -> joint targets), ONE ENV PER QUAD OF LANES
// (pih_ikq.h: 64 envs per block, 16 per wavefront; rounds 1-3 ran one env per lane: 64 wavefronts walking ~1 000 dependent instructions
// per IK iteration -- the quad shortens that chain to ~600 on 256 wavefronts).  The two parts touch disjoint state words.
constexpr int PRE_THREADS = 256;
// sched_k > 0 (config.schedule = 2): "light seeds".  With n envs on m wave slots and n / m around 2, longest-job-first pairs the
// heaviest env of the launch with the lightest one in the same slot -- the launch then ends a whole light env after the heaviest one.
// Here the first wave of m blocks is the m - sched_k heaviest envs plus the sched_k LIGHTEST ones; the slots of those finish early, take
// a second and a third light env, and the slots of the heaviest envs are never handed a second one.  sched_heads = m - sched_k.
__device__ __forceinline__ void pre_sort_block(float* __restrict__ state, int* __restrict__ order, int n, int sched_heads, int sched_k) {
  const int t = threadIdx.x;
  if (!order) return;
  __shared__ int hist[64], base[64];
  if (t < 64) hist[t] = 0;
  __syncthreads();
  for (int e = t; e < n; e += PRE_THREADS) {
    int k = (int)state[(size_t)e * PIH_STATE_WORDS + PIH_S_NCONTACT];
    k = k < 0 ? 0 : (k > 63 ? 63 : k);
    atomicAdd(&hist[63 - k], 1);          // bin 0 = most contacts
  }
  __syncthreads();
  if (t == 0) { int acc = 0; for (int b = 0; b < 64; b++) { base[b] = acc; acc += hist[b]; } }
  __syncthreads();
  for (int e = t; e < n; e += PRE_THREADS) {
    int k = (int)state[(size_t)e * PIH_STATE_WORDS + PIH_S_NCONTACT];
    k = k < 0 ? 0 : (k > 63 ? 63 : k);
    int r = atomicAdd(&base[63 - k], 1);
    if (sched_k > 0 && n > sched_heads + sched_k && r >= sched_heads) r = r >= n - sched_k ? sched_heads + (n - 1 - r) : r + sched_k;
    order[r] = e;
  }
}

__global__ void __launch_bounds__(PRE_THREADS) pih_pre_kernel(Params P, float* __restrict__ state, const float* __restrict__ actions,
                                                               int* __restrict__ order, int n, int sched_heads, int sched_k) {
  const int t = threadIdx.x;
  if (blockIdx.x == 0) { pre_sort_block(state, order, n, sched_heads, sched_k); return; }
  // controller: one env per QUAD of lanes (pih_ikq.h), 64 envs per block = 4 wavefronts of 16 envs
  const int env = (blockIdx.x - 1) * 64 + (t >> 2);
  if (env >= n) return;                                   // (uniform per quad: DPP never reads a lane that has left)
  float* S = state + (size_t)env * PIH_STATE_WORDS;
  if (!P.autoreset && S[PIH_S_DONE] != 0) return;      // finished envs keep their last values (envs/base_env.py:62,66)
  float a[4] = {0, 0, 0, 0};
  if (actions) { a[0] = actions[env * 4]; a[1] = actions[env * 4 + 1]; a[2] = actions[env * 4 + 2]; a[3] = actions[env * 4 + 3]; }
  QuadDpp qd; qd.l = t & 3;
  controller_targets_quad(qd, S, P, a);
}

// The round 1-3 controller launch, kept as a MEASUREMENT SWITCH (pih_config.schedule bit 3): one env per LANE, 64 envs per block on the
// block's first wavefront (controller_targets: ik_chain, ~1 000 dependent instructions per IK iteration).  A/B partner of pih_pre_kernel.
__global__ void __launch_bounds__(PRE_THREADS) pih_pre_lane_kernel(Params P, float* __restrict__ state, const float* __restrict__ actions,
                                                                    int* __restrict__ order, int n, int sched_heads, int sched_k) {
  const int t = threadIdx.x;
  if (blockIdx.x == 0) { pre_sort_block(state, order, n, sched_heads, sched_k); return; }
  if (t >= 64) return;
  const int env = (blockIdx.x - 1) * 64 + t;
  if (env >= n) return;
  float* S = state + (size_t)env * PIH_STATE_WORDS;
  if (!P.autoreset && S[PIH_S_DONE] != 0) return;
  float a[4] = {0, 0, 0, 0};
  if (actions) { a[0] = actions[env * 4]; a[1] = actions[env * 4 + 1]; a[2] = actions[env * 4 + 2]; a[3] = actions[env * 4 + 3]; }
  controller_targets(S, P, a);
}

// ---- the fused launch (round 4): ONE launch per step.
// Rounds 1-3 ran two launches per step: pih_pre_kernel (dispatch-order sort + controller, 33 us of which ~10 us are the gap between two
// dependent launches, on 64 of the chip's 1 024 SIMDs) and then pih_step_kernel (313 us at 4 096 envs).  Here the step kernel's grid is
// G = ceil(n / 64) CONTROLLER wavefronts (blocks 0 .. G - 1, dispatched first) followed by the n env wavefronts:
//   * a controller wavefront runs controller_compute for 64 envs, one per lane, from the envs' state records as the previous launch left
//     them, writes each env's 13 controller words into its mailbox record and publishes the group with a release store of the launch's
//     epoch; it never waits for anything;
//   * an env wavefront does forward kinematics, collision detection and the articulated-body sweep (~22 us), then waits -- bounded,
//     s_sleep between polls -- for its group's flag and takes the controller words from the mailbox (Wave::await_controller); the
//     controller needs ~25 us, so the first round of env waves waits a few us and the second round not at all;
//   * the dispatch order for the NEXT launch is built by the env wavefronts themselves: at its end an env takes a slot in the bin of its
//     contact count (atomicAdd; bin 0 = most contacts) of the "next" bin buffer; at its start block b finds "its" env by a prefix sum
//     over the 64 bin counts of the "current" buffer.  Three buffers rotate (current / next / being zeroed), so no launch ever reads a
//     count another wave of the same launch is still changing.  Results do not depend on the order (tests/test_gpu_api.py).
struct FusedArgs {
  int G, n, epoch;               // G = 0: two-launch path (no controller role, no mailbox)
  const int* bcur; int* bnext; int* bzero;   // bin buffers: [64 counts][64 x n slots]; nullptr: block b = env b (schedule 0) or `order`
  float* mail; int* flags; int* err;
};
__device__ __forceinline__ int fused_lookup_env(const int* __restrict__ bcur, int b, int n, int lane) {
  int p = bcur[lane];
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(p, d); if (lane >= d) p += t; }
  const unsigned long long m = __ballot(p > b);
  if (m == 0) return b;                                      // (inconsistent bins cannot happen; stay in range if they do)
  const int k = __ffsll((long long)m) - 1;
  const int base = k > 0 ? __shfl(p, k - 1) : 0;
  const int e = bcur[64 + (size_t)k * n + (b - base)];
  return e >= 0 && e < n ? e : b;
}

__global__ void __launch_bounds__(64, 2) pih_step_kernel(Params P, float* __restrict__ state, const float* __restrict__ actions,
                                                      float* __restrict__ obs, float* __restrict__ reward,
                                                      unsigned char* __restrict__ done, float* __restrict__ dbg,
                                                      float* __restrict__ ovf, const int* __restrict__ order, FusedArgs F) {
  const int lane = threadIdx.x;
  if ((int)blockIdx.x < F.G) {
    // ---- controller role: 64 envs, one per lane (controller_compute: the strictly sequential IK)
    __builtin_amdgcn_s_setprio(3);                             // it shares its SIMD with an env wavefront that will be waiting for it
    if (blockIdx.x == 0 && F.bzero) F.bzero[lane] = 0;
    const int env = blockIdx.x * 64 + lane;
    if (env < F.n) {
      const float* S = state + (size_t)env * PIH_STATE_WORDS;
      if (P.autoreset || S[PIH_S_DONE] == 0) {               // finished envs keep their last values (envs/base_env.py:62,66): their waves do not read the mailbox
        float a[4] = {0, 0, 0, 0};
        if (actions) { a[0] = actions[env * 4]; a[1] = actions[env * 4 + 1]; a[2] = actions[env * 4 + 2]; a[3] = actions[env * 4 + 3]; }
        const CtrlOut o = controller_compute(S, P, a);
        float* m = F.mail + (size_t)env * CTRL_WORDS;
        const float ov[13] = {o.target[0], o.target[1], o.target[2], o.target[3], o.target[4], o.target[5], o.target[6], o.target[7], o.target[8], o.fsm, o.fsmt, o.grasp_angle, o.attach_qz};
#pragma unroll
        for (int i = 0; i < 13; i++) __hip_atomic_store(m + i, ov[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // The mailbox stores of all 64 lanes are acknowledged before the group is published: an explicit  s_waitcnt vmcnt(0).  The
    // workgroup-scope release fence alone compiles to NO wait (the waves of a workgroup share their CU's L1, so the memory model needs none
    // for that scope), and the flag store then overtook mailbox stores still in flight -- with 640+ workgroups in a random-fly launch a
    // step wavefront read one stale target word in ~ 1 of 200 groups (tests/test_gpu_fly.py, 12 000 envs); an agent-scope release would
    // add an L2 write-back (+ 50 us per step, section 6.0 of DESIGN.md) that write-through stores do not need.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane == 0) __hip_atomic_store(F.flags + blockIdx.x, F.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the group is published
    return;
  }
  __shared__ Shared sh;
  const int b = blockIdx.x - F.G;
  const int env = __builtin_amdgcn_readfirstlane(F.bcur ? fused_lookup_env(F.bcur, b, F.n, lane) : (order ? order[b] : b));   // (wave-uniform by construction; the scalar makes it so for the compiler)
  Wave w; w.l = lane; w.counter = 0;
  float* rec = state + (size_t)env * PIH_STATE_WORDS;
#pragma unroll
  for (int i = 0; i < PIH_STATE_WORDS / 64; i++) sh.S[lane + 64 * i] = rec[lane + 64 * i];
  __syncthreads();
  float o[5], r; unsigned char d;
  Ovf ov; ov.base = ovf + (size_t)env * OVF_WORDS;
  w.dbg = dbg ? dbg + (size_t)env * PIH_DEBUG_WORDS : nullptr; w.dbgmode = P.debug; w.prio_on = !P.noprio;
  if (F.G > 0) { w.cflags = F.flags; w.cmail = F.mail; w.cerr = F.err; w.cepoch = F.epoch; w.cenv = env; }
  step_env(w, sh, P, ov, env, nullptr, o, &r, &d, dbg ? dbg + (size_t)env * PIH_DEBUG_WORDS : nullptr);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PIH_STATE_WORDS / 64; i++) rec[lane + 64 * i] = sh.S[lane + 64 * i];
  if (lane < 5 && obs) obs[env * 5 + lane] = o[0] * (lane == 0) + o[1] * (lane == 1) + o[2] * (lane == 2) + o[3] * (lane == 3) + o[4] * (lane == 4);
  if (lane == 0) {
    if (reward) reward[env] = r; if (done) done[env] = d;
    if (F.bnext) {                                           // a slot in the next launch's dispatch order: bin 0 = most contacts
      int k = (int)sh.S[PIH_S_NCONTACT]; k = k < 0 ? 0 : (k > 63 ? 63 : k);
      const int rnk = atomicAdd(F.bnext + (63 - k), 1);
      if (rnk < F.n) F.bnext[64 + (size_t)(63 - k) * F.n + rnk] = env;
    }
  }
}

// first fill of a bin buffer (pih_create): the same counting sort, from the state records
__global__ void __launch_bounds__(256) pih_bins_init_kernel(const float* __restrict__ state, int* __restrict__ bins, int n) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    int k = (int)state[(size_t)e * PIH_STATE_WORDS + PIH_S_NCONTACT]; k = k < 0 ? 0 : (k > 63 ? 63 : k);
    const int rnk = atomicAdd(bins + (63 - k), 1);
    bins[64 + (size_t)(63 - k) * n + rnk] = e;
  }
}

// hard (resetSimulation, envs/base_env.py:85-86): a NEW scene like every reset (the reference keeps drawing from the global `random`); it
// also clears the non-finite-reset count.  rewind (explicit replay, pih_reset(seed != 0) / pih_reseed): the env's draw counter restarts.
__global__ void __launch_bounds__(64) pih_reset_kernel(Params P, float* __restrict__ state, const unsigned char* __restrict__ mask, int hard, int rewind) {
  __shared__ Shared sh;
  const int env = blockIdx.x, lane = threadIdx.x;
  if (mask && !mask[env]) return;
  Wave w; w.l = lane; w.counter = 0;
  float* rec = state + (size_t)env * PIH_STATE_WORDS;
  for (int i = 0; i < PIH_STATE_WORDS / 64; i++) sh.S[lane + 64 * i] = rec[lane + 64 * i];
  __syncthreads();
  if (hard) sh.S[PIH_S_SPARE] = 0;
  if (rewind) { sh.S[PIH_S_RNG] = 0; sh.S[PIH_S_RNG_HI] = 0; }
  __syncthreads();
  reset_state(sh.S, P, P.env0 + env);
  __syncthreads();
  fk_all(w, sh);
  float tip[7]; tip_pose(sh, tip);
  for (int i = 0; i < 7; i++) sh.S[PIH_S_TIP + i] = tip[i];
  V3 ee; M3 eR; ee_pose(sh, ee, eR);
  sh.S[PIH_S_EE] = ee.x + sh.S[PIH_S_OFFSET]; sh.S[PIH_S_EE + 1] = ee.y + sh.S[PIH_S_OFFSET + 1]; sh.S[PIH_S_EE + 2] = ee.z + sh.S[PIH_S_OFFSET + 2];
  __syncthreads();
  for (int i = 0; i < PIH_STATE_WORDS / 64; i++) rec[lane + 64 * i] = sh.S[lane + 64 * i];
}

__global__ void __launch_bounds__(64) pih_init_offsets_kernel(float* __restrict__ state, const float* __restrict__ offsets, int n) {
  int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= n) return;
  float* rec = state + (size_t)e * PIH_STATE_WORDS;
  for (int k = 0; k < 3; k++) rec[PIH_S_OFFSET + k] = offsets ? offsets[3 * e + k] : 0.f;
}

// stand-alone batched IK (envs/utils.py:67,79): one problem per QUAD of lanes (pih_ikq.h), 16 problems per wavefront
template <class C> __device__ __forceinline__ void ik_quad_problem(const Params& P, int n, int stride, const float* q0, const float* tpos, const float* tquat, float* qout) {
  const int i = blockIdx.x * 16 + (threadIdx.x >> 2);
  if (i >= n) return;
  QuadDpp qd; qd.l = threadIdx.x & 3;
  const int j0 = 2 * qd.l, j1 = 2 * qd.l + 1;
  const QuadSlots sl = ikq_slots<C>(qd);
  float a = j0 < C::N ? q0[i * stride + j0] : 0.0f, b = j1 < C::N ? q0[i * stride + j1] : 0.0f;
  Q4 tq; tq.x = tquat[4 * i]; tq.y = tquat[4 * i + 1]; tq.z = tquat[4 * i + 2]; tq.w = tquat[4 * i + 3];
  ikq_solve(qd, sl, P, mk(tpos[3 * i], tpos[3 * i + 1], tpos[3 * i + 2]), tq, a, b);
  if (j0 < C::N) qout[i * stride + j0] = a;
  if (j1 < C::N) qout[i * stride + j1] = b;
  for (int k = C::N + qd.l; k < stride; k += 4) qout[i * stride + k] = q0[i * stride + k];      // (Panda: the finger entries pass through)
}
__global__ void __launch_bounds__(64) pih_ik_kernel(Params P, int n, const float* __restrict__ q0, const float* __restrict__ tpos,
                                                    const float* __restrict__ tquat, float* __restrict__ qout) {
  ik_quad_problem<PandaChain>(P, n, 9, q0, tpos, tquat, qout);
}
// the same for the UR5 chain (envs/utils.py:79): q0 float[n,6] -> qout float[n,6]
__global__ void __launch_bounds__(64) pih_ik_ur5_kernel(Params P, int n, const float* __restrict__ q0, const float* __restrict__ tpos,
                                                        const float* __restrict__ tquat, float* __restrict__ qout) {
  ik_quad_problem<Ur5Chain>(P, n, 6, q0, tpos, tquat, qout);
}

// wrist camera (p12): grid = (strips, envs), 256 threads (4 waves); out float[count, H, W, 4] = depth, r, g, b
__global__ void __launch_bounds__(RENDER_THREADS) pih_render_kernel(const float* __restrict__ state, float* __restrict__ out,
                                                                    int env_begin, int W, int H, int rows_per_strip, int flags) {
  __shared__ Shared sh;
  __shared__ Scene sc;
  const int tid = threadIdx.x, e = blockIdx.y, env = env_begin + e;
  const int r0 = blockIdx.x * rows_per_strip, r1 = min(H, r0 + rows_per_strip);
  Wave w; w.l = tid; w.counter = 0;
  const float* rec = state + (size_t)env * PIH_STATE_WORDS;
  for (int i = tid; i < PIH_STATE_WORDS; i += RENDER_THREADS) sh.S[i] = rec[i];
  __syncthreads();
  scene_setup(w, sh, sc, tid);
  const float T = PIH_CAM_TANH2, sy = 2.0f / H, sx = 2.0f / W;
  float4* img = reinterpret_cast<float4*>(out) + (size_t)e * H * W;
  constexpr int TR = 16, TC = 64;                   // tile = 16 rows x 64 columns, one row of a tile per wave instruction
  const int lane = tid & 63, wave = tid >> 6;
  const int tcols = (W + TC - 1) / TC, trows = (r1 - r0 + TR - 1) / TR;
  for (int tile = wave; tile < tcols * trows; tile += RENDER_THREADS / 64) {
    const int ti = tile / tcols, tj = tile - ti * tcols;
    const int i0 = r0 + ti * TR, i1 = min(r1, i0 + TR), j0 = tj * TC, j1 = min(W, j0 + TC);
    // camera-plane rectangle of the tile's pixel edges (row 0 is the top of the image, v grows upwards)
    const float tu0 = (sx * j0 - 1.0f) * T, tu1 = (sx * j1 - 1.0f) * T, tv0 = (1.0f - sy * i1) * T, tv1 = (1.0f - sy * i0) * T;
    const unsigned prims = (unsigned)__ballot(prim_on_tile(sc, lane, tu0, tu1, tv0, tv1));
    const int j = j0 + lane;
    if (j < j1) {
      const float xc = (sx * (j + 0.5f) - 1.0f) * T;
      for (int i = i0; i < i1; i++) {
        real4 c = shade(sc, prims, xc, (1.0f - sy * (i + 0.5f)) * T, flags);
        img[i * W + j] = make_float4(c.x, c.y, c.z, c.w);
      }
    }
  }
}

// label images: grid = (ceil(S*S/256), envs); out[e][k][y][x], image index [cc][rr] of the reference = [x-like c][r]
__global__ void __launch_bounds__(256) pih_labels_kernel(const float* __restrict__ state, float* __restrict__ out, float* __restrict__ meta,
                                                         int env_begin, int S) {
  const int e = blockIdx.y, env = env_begin + e;
  const float angle = state[(size_t)env * PIH_STATE_WORDS + PIH_S_GRASP_ANGLE];
  const LabelRect L = label_rect(angle, S);
  // each thread owns 4 consecutive pixels and writes one 16-byte store per label plane (the kernel is pure HBM writes)
  const int SS = S * S, i4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  float* o = out + (size_t)e * 4 * SS;
  if (i4 < SS) {
    float v[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int idx = i4 + k < SS ? i4 + k : SS - 1;
      const int c = idx / S, r = idx - c * S;            // element [c][r] of the image
      const bool in = label_inside(L, (float)c, (float)r);
      v[0][k] = in ? 50.0f : 0.0f; v[1][k] = in ? L.s2 : 0.0f; v[2][k] = in ? L.c2 : 1.0f; v[3][k] = in ? L.wpx : 0.0f;
    }
    const bool vec = (i4 + 3 < SS) && ((SS & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
    for (int pl = 0; pl < 4; pl++) {
      if (vec) *reinterpret_cast<float4*>(o + (size_t)pl * SS + i4) = make_float4(v[pl][0], v[pl][1], v[pl][2], v[pl][3]);
      else for (int k = 0; k < 4 && i4 + k < SS; k++) o[(size_t)pl * SS + i4 + k] = v[pl][k];
    }
  }
  if (meta && blockIdx.x == 0 && threadIdx.x == 0) {
    float* m = meta + (size_t)e * 5;
    m[0] = 0; m[1] = 0; m[2] = angle * (180.0f / 3.14159265358979f); m[3] = L.wpx; m[4] = L.lpx;
  }
}

__global__ void pih_gather_kernel(const float* __restrict__ state, float* __restrict__ out, int n, int word0, int nwords) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * nwords) return;
  int e = idx / nwords, k = idx - e * nwords;
  out[idx] = state[(size_t)e * PIH_STATE_WORDS + word0 + k];
}

// ------------------------------------------------------------------------------------------------ 'random-fly' task kernels
// One env per LANE or per QUAD of lanes (pih_fly.h).  state: float[PIH_FLY_STATE_WORDS][n] (structure-of-arrays: word w of the envs of a
// wave is one coalesced segment).  LDS: the per-lane contact rows, [word][lane]: 120 KB per wave in the lane layout (one wave per CU), 38 KB
// in the quad layout (four per CU; the kernel needs 450 registers, i.e. one wave per SIMD, anyway).
// (Round 3, profiles/r03_fly_envs_per_wave.txt: packing FEWER envs into a wave of the lane layout did not shorten the launch, and with 120 KB
//  of LDS per wave it cost rounds.  The quad layout does not idle the other lanes: it gives them a share of the PGS sweep.)
// Launch 1 of a random-fly step: the controller ur_execute (envs/utils.py:70-82) -- getQuaternionFromEuler + calculateInverseKinematics --
// with one env per QUAD of lanes (pih_ikq.h); writes the IK targets into the state record (words PIH_F_TARGET .., structure-of-arrays).
// 256 threads = 64 envs per block.  (Rounds 2-3 ran the IK inside the one-env-per-lane step kernel, where it was 60 % of the
// instruction stream of 64 lone wavefronts.)
__global__ void __launch_bounds__(256) pih_fly_pre_kernel(Params P, float* __restrict__ state, const float* __restrict__ actions, int n) {
  const int env = blockIdx.x * 64 + (threadIdx.x >> 2);
  if (env >= n) return;
  if (!P.autoreset && state[(size_t)PIH_F_DONE * n + env] != 0) return;      // frozen (envs/base_env.py:62,66): targets stay
  QuadDpp qd; qd.l = threadIdx.x & 3;
  const int j0 = 2 * qd.l, j1 = j0 + 1;
  const QuadSlots sl = ikq_slots<Ur5Chain>(qd);
  float q0 = j0 < fly::NJ ? state[(size_t)(PIH_F_Q + j0) * n + env] : 0.0f, q1 = j1 < fly::NJ ? state[(size_t)(PIH_F_Q + j1) * n + env] : 0.0f;
  const float* a = actions + (size_t)env * PIH_FLY_ACTION_DIM;
  const Q4 tq = quat_from_euler(a[3], a[4], a[5]);
  const V3 tp = mk(a[0] - state[(size_t)PIH_F_OFFSET * n + env], a[1] - state[(size_t)(PIH_F_OFFSET + 1) * n + env], a[2] - state[(size_t)(PIH_F_OFFSET + 2) * n + env]);
  ikq_solve(qd, sl, P, tp, tq, q0, q1);
  if (j0 < fly::NJ) state[(size_t)(PIH_F_TARGET + j0) * n + env] = q0;
  if (j1 < fly::NJ) state[(size_t)(PIH_F_TARGET + j1) * n + env] = q1;
}

// The fused random-fly launch (round 4, as the peg-in-hole one): the first G = ceil(n / 64) blocks are CONTROLLER wavefronts (the IK of 64
// envs, one per lane, straight into a mailbox [group][word][lane], then the epoch into the block's flag), the following blocks the step
// wavefronts (ceil(n / 16) in the quad layout, G in the lane layout), whose lanes read their targets from the mailbox right before the PGS
// loop -- after forward kinematics, the articulated-body sweeps, collision detection and all response rows.  Every workgroup of the launch
// carries the step's dynamic LDS, and a step wavefront spins on its controller's flag, so the layout is used only while ALL workgroups are
// resident at once (pih_create: quad layout 4 per CU, lane layout 1 per CU); bigger batches run the IK inside the step wavefront (MODE 0).
// MODE 1: targets from a pre-launch (pih_fly_pre_kernel; measurement switch).
struct FlyFused { int G, epoch; float* mail; int* flags; int* err; };
struct MailboxIk {
  const int* flag; const float* mail; int* err; int epoch, env, n;
  __device__ __forceinline__ void operator()(const float*, const float*, const float*, const Params&, float* qs) const {
    int tries = 0;      // (relaxed agent-scope atomics + workgroup fences: see Wave::await_controller)
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < epoch) {
      __builtin_amdgcn_s_sleep(4);
      if (++tries > (1 << 21)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int i = 0; i < fly::NJ; i++) qs[i] = __hip_atomic_load(mail + ((size_t)(env >> 6) * fly::NJ + i) * 64 + (env & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};
// QUAD: one env per quad of lanes (pih_fly.h): a step wavefront holds 16 envs, the PGS sweep is split over the quad; the controller
// wavefronts stay one env per lane, so one flag covers four step wavefronts.
__device__ __forceinline__ int fly_wave_or(int x) {     // OR of a 6-bit mask over the wavefront's active lanes, as a scalar
  int m = 0;
#pragma unroll
  for (int k = 0; k < fly::NJ; k++) if (__builtin_amdgcn_ballot_w64((x >> k) & 1) != 0) m |= 1 << k;
  return __builtin_amdgcn_readfirstlane(m);
}
struct FlyLane : fly::NoQuad {
  __device__ __forceinline__ int wave_or(int x) const { return fly_wave_or(x); }
  __device__ __forceinline__ bool wave_any(bool x) const { return __builtin_amdgcn_ballot_w64(x) != 0; }
};
struct FlyQuad : QuadDpp {
  static constexpr bool QUAD = true;
  __device__ __forceinline__ int wave_max(int x) const {     // the largest x of the wavefront's active lanes (0 <= x <= fly::NC), as a scalar
    int m = 0;
#pragma unroll
    for (int k = 1; k <= fly::NC; k++) if (__builtin_amdgcn_ballot_w64(x >= k) != 0) m = k;
    return __builtin_amdgcn_readfirstlane(m);
  }
  __device__ __forceinline__ int wave_or(int x) const { return fly_wave_or(x); }
  __device__ __forceinline__ bool wave_any(bool x) const { return __builtin_amdgcn_ballot_w64(x) != 0; }
};
template <int MODE, bool QUAD = false>
__global__ void __launch_bounds__(64, 1) pih_fly_step_kernel(Params P, float* __restrict__ state, const float* __restrict__ actions,
                                                             float* __restrict__ obs, float* __restrict__ reward,
                                                             unsigned char* __restrict__ done, float* __restrict__ dbg, int n, FlyFused F) {
  extern __shared__ float lanemem[];
  if (MODE == 2 && (int)blockIdx.x < F.G) {
    // ---- controller role
    __builtin_amdgcn_s_setprio(3);
    const int env = blockIdx.x * 64 + threadIdx.x;
    if (env < n && (P.autoreset || state[(size_t)PIH_F_DONE * n + env] == 0)) {
      float q[fly::NJ], S3[PIH_F_OFFSET + 3], a[PIH_FLY_ACTION_DIM], qs[fly::NJ];
#pragma unroll
      for (int i = 0; i < fly::NJ; i++) q[i] = state[(size_t)(PIH_F_Q + i) * n + env];
#pragma unroll
      for (int k = 0; k < 3; k++) S3[PIH_F_OFFSET + k] = state[(size_t)(PIH_F_OFFSET + k) * n + env];
#pragma unroll
      for (int k = 0; k < PIH_FLY_ACTION_DIM; k++) a[k] = actions[(size_t)env * PIH_FLY_ACTION_DIM + k];
      fly::InlineIk()(q, S3, a, P, qs);
#pragma unroll
      for (int i = 0; i < fly::NJ; i++) __hip_atomic_store(F.mail + ((size_t)blockIdx.x * fly::NJ + i) * 64 + threadIdx.x, qs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (see the peg-in-hole controller role: the stores are acknowledged before the flag)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(F.flags + blockIdx.x, F.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int blk = MODE == 2 ? blockIdx.x - F.G : blockIdx.x;
  const int env = QUAD ? blk * 16 + (int)(threadIdx.x >> 2) : blk * 64 + (int)threadIdx.x;
  if (env >= n) return;
  const bool writer = !QUAD || (threadIdx.x & 3) == 0;      // (the four lanes of a quad hold identical results)
  // config.debug = 2: start / end of the wavefront on the chip-wide 100 MHz clock and where it ran (debug words 940 .. 947 of the wave's
  // first env; tools/fly_trace.py) -- never read by the kernel
  const bool stamp = dbg && P.debug == 2 && threadIdx.x == 0;
  long long ts0 = 0;
  if (stamp) ts0 = (long long)__builtin_amdgcn_s_memrealtime();
  float S[fly::SW];
#pragma unroll
  for (int w = 0; w < fly::SW; w++) S[w] = state[(size_t)w * n + env];
  float a[PIH_FLY_ACTION_DIM] = {0, 0, 0, 0, 0, 0};
  if (actions) {
#pragma unroll
    for (int k = 0; k < PIH_FLY_ACTION_DIM; k++) a[k] = actions[(size_t)env * PIH_FLY_ACTION_DIM + k];
  }
  float o[PIH_FLY_OBS_DIM], r; unsigned char d;
  fly::LaneMem mem; mem.p = lanemem + threadIdx.x; mem.stride = 64;
  float* dbge = dbg ? dbg + (size_t)env * PIH_DEBUG_WORDS : nullptr;
  if constexpr (MODE == 2) {
    MailboxIk mb; mb.flag = F.flags + (env >> 6); mb.mail = F.mail; mb.err = F.err; mb.epoch = F.epoch; mb.env = env; mb.n = n;
    if constexpr (QUAD) { FlyQuad qd; qd.l = threadIdx.x & 3; fly::step_env(S, P, P.env0 + env, a, o, &r, &d, mem, dbge, mb, qd); }
    else fly::step_env(S, P, P.env0 + env, a, o, &r, &d, mem, dbge, mb, FlyLane());
  } else if constexpr (MODE == 1) fly::step_env(S, P, P.env0 + env, a, o, &r, &d, mem, dbge, fly::RecordIk(), FlyLane());
  else if constexpr (QUAD) { FlyQuad qd; qd.l = threadIdx.x & 3; fly::step_env(S, P, P.env0 + env, a, o, &r, &d, mem, dbge, fly::InlineIk(), qd); }
  else fly::step_env(S, P, P.env0 + env, a, o, &r, &d, mem, dbge, fly::InlineIk(), FlyLane());
  if (writer) {
#pragma unroll
    for (int w = 0; w < fly::SW; w++) state[(size_t)w * n + env] = S[w];
    if (obs) {
#pragma unroll
      for (int k = 0; k < PIH_FLY_OBS_DIM; k++) obs[(size_t)env * PIH_FLY_OBS_DIM + k] = o[k];
    }
    if (reward) reward[env] = r;
    if (done) done[env] = d;
  }
  if (stamp) {
    const long long ts1 = (long long)__builtin_amdgcn_s_memrealtime();
    float* o2 = dbg + (size_t)env * PIH_DEBUG_WORDS;
    o2[940] = (float)(ts0 & 0xFFFF); o2[941] = (float)((ts0 >> 16) & 0xFFFF); o2[942] = (float)((ts0 >> 32) & 0xFFFF);
    o2[943] = (float)(ts1 & 0xFFFF); o2[944] = (float)((ts1 >> 16) & 0xFFFF); o2[945] = (float)((ts1 >> 32) & 0xFFFF);
    o2[946] = (float)(__builtin_amdgcn_s_getreg((15 << 11) | 4) & 0xFFFF); o2[947] = (float)__builtin_amdgcn_s_getreg((3 << 11) | 20);
  }
}

__global__ void __launch_bounds__(64) pih_fly_reset_kernel(Params P, float* __restrict__ state, const unsigned char* __restrict__ mask, int hard, int rewind, int n) {
  const int env = blockIdx.x * 64 + threadIdx.x;
  if (env >= n || (mask && !mask[env])) return;
  float S[fly::SW];
#pragma unroll
  for (int w = 0; w < fly::SW; w++) S[w] = state[(size_t)w * n + env];
  if (hard) S[PIH_F_SPARE] = 0;
  if (rewind) { S[PIH_F_RNG] = 0; S[PIH_F_RNG_HI] = 0; }
  fly::reset_state(S, P, P.env0 + env);
#pragma unroll
  for (int w = 0; w < fly::SW; w++) state[(size_t)w * n + env] = S[w];
}

// env-major float[n, W]  <->  structure-of-arrays float[W][n]
__global__ void pih_soa_to_aos_kernel(const float* __restrict__ soa, float* __restrict__ aos, int n, int W, int word0, int nwords) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * nwords) return;
  const int e = idx / nwords, k = idx - e * nwords;
  aos[idx] = soa[(size_t)(word0 + k) * n + e];
  (void)W;
}
__global__ void pih_aos_to_soa_kernel(const float* __restrict__ aos, float* __restrict__ soa, int n, int W) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * W) return;
  const int w = idx / n, e = idx - w * n;
  soa[idx] = aos[(size_t)e * W + w];
}
__global__ void pih_fly_init_offsets_kernel(float* __restrict__ state, const float* __restrict__ offsets, int n) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= n) return;
  for (int k = 0; k < 3; k++) state[(size_t)(PIH_F_OFFSET + k) * n + e] = offsets ? offsets[3 * e + k] : 0.f;
}

// ------------------------------------------------------------------------------------------------ host side
// roctx ranges around the step / reset launches (tracing): only with pih_config.debug != 0, and through dlopen so
// that the library has no link-time dependency on the profiler SDK -- without the roctx library the ranges are silently absent.
struct Roctx {
  int (*push)(const char*) = nullptr; int (*pop)() = nullptr; bool tried = false;
  void load() {
    if (tried) return;
    tried = true;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* l = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!l) continue;
      push = reinterpret_cast<int (*)(const char*)>(dlsym(l, "roctxRangePushA")); pop = reinterpret_cast<int (*)()>(dlsym(l, "roctxRangePop"));
      if (push && pop) return;
      push = nullptr; pop = nullptr;
    }
  }
};
static Roctx g_roctx;
struct RoctxRange {      // RAII: pushes when tracing is on for this handle and the library was found
  bool on = false;
  RoctxRange(bool enabled, const char* name) { if (enabled) { g_roctx.load(); if (g_roctx.push) { g_roctx.push(name); on = true; } } }
  ~RoctxRange() { if (on) g_roctx.pop(); }
};
constexpr size_t EV_POOL = 1024;   // timing event triples kept before they are folded into the running sums
struct EvTriple { hipEvent_t a, b, c; };
struct pih_handle {
  pih_config cfg;
  Params P;
  int device = 0;
  int sched_k = 0, sched_heads = 0;   // config.schedule = 2: light seeds in the first wave of blocks (see pih_pre_kernel)
  bool fly = false;         // PIH_TASK_RANDOM_FLY: structure-of-arrays state, one env per lane
  bool rewind_pending = false;   // pih_reseed: the envs reset by the next pih_reset restart their draw sequence
  int words = PIH_STATE_WORDS;
  float* state = nullptr;
  float* dbg = nullptr;
  float* ovf = nullptr;     // spill area for contacts beyond the LDS-resident CL (rarely touched)
  int* order = nullptr;     // longest-job-first block -> env map (block 0 of pih_pre_kernel; two-launch path)
  // fused launch (default for the peg-in-hole task): controller mailbox + group flags + error word, three rotating bin buffers of the
  // in-kernel dispatch order, the launch counter
  bool fused = false, flyquad = false;     // flyquad: random-fly step wavefronts hold one env per quad of lanes
  float* mail = nullptr; int* flags = nullptr; int* errw = nullptr; volatile int* errw_host = nullptr; int* bins = nullptr; size_t bin_ints = 0; int epoch = 0;
  std::string err;
  int timing = 0;           // 0 off; k >= 1: every k-th step launch is bracketed by events (pih_set_timing)
  unsigned timing_tick = 0;
  std::vector<EvTriple> ev;   // (before pre-kernel, between, after step kernel) of each timed step launch
  size_t ev_used = 0;
  double acc_pre_ms = 0, acc_step_ms = 0; int64_t acc_n = 0;
};

static thread_local std::string g_err;
static const char* const CTRL_TIMEOUT_MSG = "pih_step: an env wavefront timed out waiting for its controller wavefront (fused launch); the results of that step are invalid";

static int fail(pih_handle* h, const char* what, hipError_t e) {
  std::string m = std::string(what) + ": " + hipGetErrorString(e);
  if (h) h->err = m;
  g_err = m;
  return -1;
}
#define HIPCHK(h, x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(h, #x, _e); } while (0)

// Every entry point runs on the handle's own device ("one handle per device, handles independent"): the caller's current
// device is switched for the duration of the call and restored afterwards.
struct DevGuard {
  int prev = -1; bool switched = false; hipError_t err = hipSuccess;
  explicit DevGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); switched = err == hipSuccess; }
  }
  ~DevGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define PIH_ENTER(h) DevGuard _guard((h)->device); if (_guard.err != hipSuccess) return fail(h, "select the handle's device", _guard.err)

static Params make_params(const pih_config* c) {
  Params P;
  P.dt = c->dt; P.resid = c->residual_threshold; P.erp = c->erp; P.warm = c->warmstart; P.margin = c->contact_margin;
  P.slop = c->linear_slop; P.ikdamp = c->ik_damping; P.ikres = c->ik_residual; P.dv = c->dv; P.iters = c->solver_iters;
  P.ikiters = c->ik_iters; P.mode = c->mode; P.maxsteps = c->max_episode_steps; P.autoreset = c->auto_reset;
  P.selfcol = c->enable_self_collision; P.armcol = c->enable_arm_collision; P.debug = c->debug; P.env0 = c->env_index0; P.seed = c->seed; P.pgsmode = c->solver_path; P.attachball = c->attach_ball; P.noprio = (c->schedule & 4) != 0; P.nospec = (c->schedule & 64) != 0;
  P.checkstride = c->exit_check_stride < 1 ? 1 : c->exit_check_stride;
  P.object = c->object_id;
  return P;
}

// fold finished event triples into the running sums (called when the pool is full and by pih_timing)
static int drain_events(pih_handle* h) {
  for (size_t i = 0; i < h->ev_used; i++) {
    HIPCHK(h, hipEventSynchronize(h->ev[i].c));
    float m1 = 0, m2 = 0;
    HIPCHK(h, hipEventElapsedTime(&m1, h->ev[i].a, h->ev[i].b));
    HIPCHK(h, hipEventElapsedTime(&m2, h->ev[i].b, h->ev[i].c));
    h->acc_pre_ms += m1; h->acc_step_ms += m2; h->acc_n++;
  }
  h->ev_used = 0;
  return 0;
}

extern "C" {

void pih_default_config(pih_config* c) {
  memset(c, 0, sizeof *c);
  c->n_envs = 1; c->env_index0 = 0; c->mode = 0; c->solver_iters = 50; c->ik_iters = 20; c->max_episode_steps = 2227;
  c->auto_reset = 0; c->enable_self_collision = 1; c->enable_arm_collision = 3; c->task_id = PIH_TASK_PEG_IN_HOLE; c->debug = 0; c->schedule = 1; c->exit_check_stride = 16; c->seed = 0; c->dt = 1.0f / 240.0f; c->residual_threshold = 1e-7f;
  c->erp = 0.2f; c->warmstart = 0.85f; c->contact_margin = 0.005f; c->linear_slop = 1e-5f; c->ik_damping = 0.5f; c->ik_residual = 1e-4f;
  c->dv = 2.0f / 240.0f;
}
int pih_abi_version(void) { return PIH_ABI_VERSION; }
int pih_task_dims(int task_id, int32_t out[3]) {
  if (!out) return -2;
  if (task_id == PIH_TASK_PEG_IN_HOLE) { out[0] = PIH_ACTION_DIM; out[1] = PIH_OBS_DIM; out[2] = PIH_STATE_WORDS; return 0; }
  if (task_id == PIH_TASK_RANDOM_FLY) { out[0] = PIH_FLY_ACTION_DIM; out[1] = PIH_FLY_OBS_DIM; out[2] = PIH_FLY_STATE_WORDS; return 0; }
  return -2;
}

const char* pih_object_name(int task_id, int object_id) {
  static const char* const names[PIH_FLY_NOBJ] = PIH_FLY_OBJ_NAMES;
  return task_id == PIH_TASK_RANDOM_FLY && object_id >= 0 && object_id < PIH_FLY_NOBJ ? names[object_id] : nullptr;
}

int pih_destroy(pih_handle* h) {
  if (!h) return 0;
  DevGuard guard(h->device);
  for (auto& p : h->ev) { hipEventDestroy(p.a); hipEventDestroy(p.b); hipEventDestroy(p.c); }
  if (h->state) hipFree(h->state);
  if (h->dbg) hipFree(h->dbg);
  if (h->ovf) hipFree(h->ovf);
  if (h->order) hipFree(h->order);
  if (h->mail) hipFree(h->mail);
  if (h->flags) hipFree(h->flags);
  if (h->errw_host) hipHostFree(const_cast<int*>(h->errw_host));
  if (h->bins) hipFree(h->bins);
  delete h;
  return 0;
}

// allocation + first reset; on any failure the caller (pih_create) destroys the half-built handle
static int create_impl(pih_handle* h, const float* offsets_host, float** offd) {
  const pih_config* cfg = &h->cfg;
  size_t nb = (size_t)cfg->n_envs * h->words * sizeof(float);
  HIPCHK(h, hipMalloc(&h->state, nb));
  HIPCHK(h, hipMemset(h->state, 0, nb));
  if (cfg->debug) { HIPCHK(h, hipMalloc(&h->dbg, (size_t)cfg->n_envs * PIH_DEBUG_WORDS * sizeof(float))); HIPCHK(h, hipMemset(h->dbg, 0, (size_t)cfg->n_envs * PIH_DEBUG_WORDS * sizeof(float))); }
  if (offsets_host) {
    HIPCHK(h, hipMalloc(offd, (size_t)cfg->n_envs * 3 * sizeof(float)));
    HIPCHK(h, hipMemcpy(*offd, offsets_host, (size_t)cfg->n_envs * 3 * sizeof(float), hipMemcpyHostToDevice));
  }
  if (h->fly) {
    // (dynamic LDS beyond the default 64 KB limit: the per-lane contact rows and candidate staging of 64 envs are LANE_WORDS * 64 words)
    static_assert((size_t)fly::LANE_WORDS * 64 * sizeof(float) <= 160 * 1024, "the random-fly kernel's per-wave LDS exceeds a CU's 160 KB");
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pih_fly_step_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(fly::LANE_WORDS * 64 * sizeof(float))));
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pih_fly_step_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(fly::LANE_WORDS * 64 * sizeof(float))));
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pih_fly_step_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(fly::LANE_WORDS * 64 * sizeof(float))));
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pih_fly_step_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(fly::LANE_WORDS_Q * 64 * sizeof(float))));
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pih_fly_step_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(fly::LANE_WORDS_Q * 64 * sizeof(float))));
    {
      // fused launch (controller wavefronts + step wavefronts in one grid) while both sets of workgroups -- each with the step's 120 KB of
      // LDS -- fit the chip's CUs at once; schedule + 8: IK inside the step wavefront, + 16: IK as a quad-per-env pre-launch (switches)
      int cus = 0; HIPCHK(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
      const int G = (cfg->n_envs + 63) / 64;
      // One env per QUAD of lanes in the step wavefronts (38 KB of LDS: four workgroups per CU) unless schedule + 32; its fused launch needs
      // the G controller workgroups and the ceil(n / 16) step workgroups resident together: 4 per CU (n <= 13 104 on 256 CUs); bigger
      // batches run the IK inside the quad's step wavefront.  The lane layout (120 KB per workgroup) fuses while 2 G <= CUs.
      h->flyquad = (cfg->schedule & (32 | 16)) == 0;
      const bool nofuse = (cfg->schedule & (8 | 16)) != 0;
      h->fused = !nofuse && (h->flyquad ? G + (cfg->n_envs + 15) / 16 <= 4 * cus : 2 * G <= cus);
      if (h->fused) {
        // mailbox [group][word][64 lanes]: the two 128-byte lines of a (group, word) hold no other group's targets
        HIPCHK(h, hipMalloc(&h->mail, (size_t)G * 64 * fly::NJ * sizeof(float)));
        HIPCHK(h, hipMemset(h->mail, 0, (size_t)G * 64 * fly::NJ * sizeof(float)));
        HIPCHK(h, hipMalloc(&h->flags, (size_t)G * sizeof(int)));
        HIPCHK(h, hipMemset(h->flags, 0, (size_t)G * sizeof(int)));
        HIPCHK(h, hipHostMalloc((void**)&h->errw_host, sizeof(int), hipHostMallocMapped));
        *h->errw_host = 0;
        HIPCHK(h, hipHostGetDevicePointer((void**)&h->errw, const_cast<int*>(h->errw_host), 0));
      }
    }
    const int nb64 = (cfg->n_envs + 63) / 64;
    hipLaunchKernelGGL(pih_fly_init_offsets_kernel, dim3(nb64), dim3(64), 0, 0, h->state, *offd, cfg->n_envs);
    hipLaunchKernelGGL(pih_fly_reset_kernel, dim3(nb64), dim3(64), 0, 0, h->P, h->state, (const unsigned char*)nullptr, 0, 0, cfg->n_envs);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipDeviceSynchronize());
    return 0;
  }
  HIPCHK(h, hipMalloc(&h->ovf, ((size_t)cfg->n_envs * OVF_WORDS + OVF_PAD_WORDS) * sizeof(float)));
  HIPCHK(h, hipMemset(h->ovf, 0, ((size_t)cfg->n_envs * OVF_WORDS + OVF_PAD_WORDS) * sizeof(float)));
  if (cfg->schedule & 3) HIPCHK(h, hipMalloc(&h->order, (size_t)cfg->n_envs * sizeof(int)));
  // fused launch unless a measurement switch asks for the two-launch path (schedule + 8: controller one env per lane, + 16: one env per
  // quad) or for the experimental partner-aware order (schedule & 3 == 2, which only the pre-kernel's sort implements)
  h->fused = (cfg->schedule & (8 | 16)) == 0 && (cfg->schedule & 3) != 2;
  if (h->fused) {
    const int n = cfg->n_envs, G = (n + 63) / 64;
    HIPCHK(h, hipMalloc(&h->mail, (size_t)n * CTRL_WORDS * sizeof(float)));
    HIPCHK(h, hipMemset(h->mail, 0, (size_t)n * CTRL_WORDS * sizeof(float)));
    HIPCHK(h, hipMalloc(&h->flags, (size_t)G * sizeof(int)));
    HIPCHK(h, hipMemset(h->flags, 0, (size_t)G * sizeof(int)));
    // the error word lives in pinned, device-mapped HOST memory: an env wavefront that gives up waiting stores 1 there (system scope), and
    // the next pih_step / pih_timing2 on the handle sees it without synchronising anything
    HIPCHK(h, hipHostMalloc((void**)&h->errw_host, sizeof(int), hipHostMallocMapped));
    *h->errw_host = 0;
    HIPCHK(h, hipHostGetDevicePointer((void**)&h->errw, const_cast<int*>(h->errw_host), 0));
    if (cfg->schedule & 3) {
      h->bin_ints = 64 + (size_t)64 * n;
      HIPCHK(h, hipMalloc(&h->bins, 3 * h->bin_ints * sizeof(int)));
      HIPCHK(h, hipMemset(h->bins, 0, 3 * h->bin_ints * sizeof(int)));
    }
  }
  if ((cfg->schedule & 3) == 2) {
    int cus = 0; HIPCHK(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
    const int slots = 8 * cus;                    // two 256-VGPR wavefronts on each of the 4 SIMDs of a CU
    h->sched_k = slots / 16;                      // (0 < sched_k < slots by construction; pih_pre_kernel ignores it unless n > slots)
    h->sched_heads = slots - h->sched_k;
  }
  hipLaunchKernelGGL(pih_init_offsets_kernel, dim3((cfg->n_envs + 63) / 64), dim3(64), 0, 0, h->state, *offd, cfg->n_envs);
  hipLaunchKernelGGL(pih_reset_kernel, dim3(cfg->n_envs), dim3(64), 0, 0, h->P, h->state, (const unsigned char*)nullptr, 0, 0);
  // the first launch (epoch 1) reads bin buffer 1: every env once, ordered by the contact counts of the reset state
  if (h->bins) hipLaunchKernelGGL(pih_bins_init_kernel, dim3(64), dim3(256), 0, 0, h->state, h->bins + h->bin_ints, cfg->n_envs);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipDeviceSynchronize());
  return 0;
}

int pih_create(const pih_config* cfg, const float* offsets_host, pih_handle** out) {
  if (!cfg || !out || cfg->n_envs <= 0) { g_err = "pih_create: bad arguments"; return -2; }
  if (cfg->task_id != PIH_TASK_PEG_IN_HOLE && cfg->task_id != PIH_TASK_RANDOM_FLY) { g_err = "pih_create: unknown task_id"; return -2; }
  if (cfg->task_id == PIH_TASK_RANDOM_FLY && (cfg->object_id < 0 || cfg->object_id >= PIH_FLY_NOBJ)) { g_err = "pih_create: unknown object_id for the random-fly task"; return -2; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) { g_err = "pih_create: no HIP device (this library has no CPU path)"; return -3; }
  pih_handle* h = new pih_handle;
  h->cfg = *cfg; h->P = make_params(cfg);
  h->fly = cfg->task_id == PIH_TASK_RANDOM_FLY; h->words = h->fly ? PIH_FLY_STATE_WORDS : PIH_STATE_WORDS;
  float* offd = nullptr;
  int rc = -1;
  if (hipGetDevice(&h->device) == hipSuccess) rc = create_impl(h, offsets_host, &offd);
  else g_err = "pih_create: hipGetDevice failed";
  if (offd) hipFree(offd);
  if (rc != 0) { pih_destroy(h); return rc; }   // nothing leaks on the error path (g_err keeps the message)
  *out = h;
  return 0;
}

int pih_reset(pih_handle* h, const uint8_t* mask_dev, int hard, uint64_t seed, void* stream) {
  if (!h) return -2;
  // a new seed rewinds the draw sequences, and that must reach EVERY env of the handle: with a mask the unmasked envs would keep
  // their draw counter on a different seed stream (neither a replay nor a continuation; round-3 review) -- rejected
  if (mask_dev && (seed != 0 || h->rewind_pending)) {
    h->err = "pih_reset: a new seed (seed != 0, or a pending pih_reseed) needs a reset of all envs (mask_dev = NULL)";
    return -2;
  }
  PIH_ENTER(h);
  RoctxRange range(h->cfg.debug != 0, hard ? "pih_reset(hard)" : "pih_reset");
  if (seed != 0) { h->cfg.seed = seed; h->P.seed = seed; h->rewind_pending = true; }
  const int rewind = h->rewind_pending ? 1 : 0;
  h->rewind_pending = false;
  if (h->fly) hipLaunchKernelGGL(pih_fly_reset_kernel, dim3((h->cfg.n_envs + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->P, h->state, mask_dev, hard != 0, rewind, h->cfg.n_envs);
  else hipLaunchKernelGGL(pih_reset_kernel, dim3(h->cfg.n_envs), dim3(64), 0, (hipStream_t)stream, h->P, h->state, mask_dev, hard != 0, rewind);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int pih_reseed(pih_handle* h, uint64_t seed) {
  if (!h) return -2;
  h->cfg.seed = seed; h->P.seed = seed; h->rewind_pending = true;
  return 0;
}

static int launch_step(pih_handle* h, const float* actions, float* obs, float* reward, uint8_t* done, hipStream_t s) {
  RoctxRange range(h->cfg.debug != 0, h->fly ? "pih_step(random-fly)" : "pih_step(peg-in-hole)");
  EvTriple* t = nullptr;
  if (h->timing > 0 && (h->timing_tick++ % (unsigned)h->timing) == 0) {
    if (h->ev_used == EV_POOL) { int r = drain_events(h); if (r) return r; }
    if (h->ev_used == h->ev.size()) {
      EvTriple n;
      HIPCHK(h, hipEventCreate(&n.a)); HIPCHK(h, hipEventCreate(&n.b)); HIPCHK(h, hipEventCreate(&n.c));
      h->ev.push_back(n);
    }
    t = &h->ev[h->ev_used++];
    HIPCHK(h, hipEventRecord(t->a, s));
  }
  // measurement switches.  peg-in-hole two-launch path: + 8 = controller one env per lane, + 16 = one env per quad.  random-fly: default =
  // the fused launch (IK in controller wavefronts of the same grid; batches beyond 8192 envs: IK inside the step wavefront), + 8 = IK
  // inside the step wavefront, + 16 = the quad-per-env pre-launch
  const bool lane_ctrl = (h->cfg.schedule & 8) != 0;
  if (h->fly) {
    const int G = (h->cfg.n_envs + 63) / 64; const size_t lds = (size_t)fly::LANE_WORDS * 64 * sizeof(float);
    FlyFused FF; memset(&FF, 0, sizeof FF);
    if (h->fused) {
      if (*h->errw_host) { h->err = CTRL_TIMEOUT_MSG; return -5; }
      FF.G = G; FF.epoch = ++h->epoch; FF.mail = h->mail; FF.flags = h->flags; FF.err = h->errw;
      if (t) HIPCHK(h, hipEventRecord(t->b, s));
      if (h->flyquad)
        hipLaunchKernelGGL((pih_fly_step_kernel<2, true>), dim3(G + (h->cfg.n_envs + 15) / 16), dim3(64), (size_t)fly::LANE_WORDS_Q * 64 * sizeof(float), s, h->P, h->state, actions, obs, reward, done, h->dbg, h->cfg.n_envs, FF);
      else
        hipLaunchKernelGGL(pih_fly_step_kernel<2>, dim3(2 * G), dim3(64), lds, s, h->P, h->state, actions, obs, reward, done, h->dbg, h->cfg.n_envs, FF);
    } else if (h->cfg.schedule & 16) {
      hipLaunchKernelGGL(pih_fly_pre_kernel, dim3(G), dim3(256), 0, s, h->P, h->state, actions, h->cfg.n_envs);
      if (t) HIPCHK(h, hipEventRecord(t->b, s));
      hipLaunchKernelGGL(pih_fly_step_kernel<1>, dim3(G), dim3(64), lds, s, h->P, h->state, actions, obs, reward, done, h->dbg, h->cfg.n_envs, FF);
    } else {
      if (t) HIPCHK(h, hipEventRecord(t->b, s));
      if (h->flyquad)
        hipLaunchKernelGGL((pih_fly_step_kernel<0, true>), dim3((h->cfg.n_envs + 15) / 16), dim3(64), (size_t)fly::LANE_WORDS_Q * 64 * sizeof(float), s, h->P, h->state, actions, obs, reward, done, h->dbg, h->cfg.n_envs, FF);
      else
        hipLaunchKernelGGL(pih_fly_step_kernel<0>, dim3(G), dim3(64), lds, s, h->P, h->state, actions, obs, reward, done, h->dbg, h->cfg.n_envs, FF);
    }
    if (t) HIPCHK(h, hipEventRecord(t->c, s));
    HIPCHK(h, hipGetLastError());
    return 0;
  }
  FusedArgs F; memset(&F, 0, sizeof F);
  if (h->fused) {
    if (*h->errw_host) { h->err = CTRL_TIMEOUT_MSG; return -5; }     // an EARLIER step of this handle timed out: fail loudly from here on
    // one launch: controller wavefronts first, then the env wavefronts
    const int e = ++h->epoch;
    F.G = (h->cfg.n_envs + 63) / 64; F.n = h->cfg.n_envs; F.epoch = e; F.mail = h->mail; F.flags = h->flags; F.err = h->errw;
    if (h->bins) { F.bcur = h->bins + (size_t)(e % 3) * h->bin_ints; F.bnext = h->bins + (size_t)((e + 1) % 3) * h->bin_ints; F.bzero = h->bins + (size_t)((e + 2) % 3) * h->bin_ints; }
    if (t) HIPCHK(h, hipEventRecord(t->b, s));
    hipLaunchKernelGGL(pih_step_kernel, dim3(F.G + h->cfg.n_envs), dim3(64), 0, s, h->P, h->state, actions, obs, reward, done, h->dbg, h->ovf, (const int*)nullptr, F);
  } else {
    if (lane_ctrl) hipLaunchKernelGGL(pih_pre_lane_kernel, dim3(1 + (h->cfg.n_envs + 63) / 64), dim3(PRE_THREADS), 0, s, h->P, h->state, actions, h->order, h->cfg.n_envs, h->sched_heads, h->sched_k);
    else hipLaunchKernelGGL(pih_pre_kernel, dim3(1 + (h->cfg.n_envs + 63) / 64), dim3(PRE_THREADS), 0, s, h->P, h->state, actions, h->order, h->cfg.n_envs, h->sched_heads, h->sched_k);
    if (t) HIPCHK(h, hipEventRecord(t->b, s));
    hipLaunchKernelGGL(pih_step_kernel, dim3(h->cfg.n_envs), dim3(64), 0, s, h->P, h->state, actions, obs, reward, done, h->dbg, h->ovf, (const int*)h->order, F);
  }
  if (t) HIPCHK(h, hipEventRecord(t->c, s));
  HIPCHK(h, hipGetLastError());
  return 0;
}

int pih_step(pih_handle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream) {
  if (!h) return -2;
  if ((h->cfg.mode == 0 || h->fly) && !actions_dev) { h->err = "pih_step: actions_dev is NULL in action mode"; return -2; }
  PIH_ENTER(h);
  return launch_step(h, actions_dev, obs_dev, reward_dev, done_dev, (hipStream_t)stream);
}

int pih_step_n(pih_handle* h, int k, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream) {
  if (!h || k < 0) return -2;
  if ((h->cfg.mode == 0 || h->fly) && !actions_dev) { h->err = "pih_step_n: actions_dev is NULL in action mode"; return -2; }
  PIH_ENTER(h);
  for (int i = 0; i < k; i++) { int r = launch_step(h, actions_dev, obs_dev, reward_dev, done_dev, (hipStream_t)stream); if (r) return r; }
  return 0;
}

int pih_get_state(pih_handle* h, int field, void* out_dev, void* stream) {
  if (!h || !out_dev) return -2;
  PIH_ENTER(h);
  hipStream_t s = (hipStream_t)stream;
  const int n = h->cfg.n_envs;
  if (h->fly) {
    int w0 = 0, nw = 0;
    switch (field) {
      case PIH_FIELD_STATE: w0 = 0; nw = PIH_FLY_STATE_WORDS; break;
      case PIH_FIELD_CONTACT_FORCE: w0 = PIH_F_CFORCE; nw = 1; break;
      case PIH_FIELD_EE_POS: w0 = PIH_F_EE; nw = 3; break;
      case PIH_FIELD_DEBUG:
        if (!h->dbg) { h->err = "pih_get_state: debug buffer not enabled (config.debug = 0)"; return -4; }
        HIPCHK(h, hipMemcpyAsync(out_dev, h->dbg, (size_t)n * PIH_DEBUG_WORDS * sizeof(float), hipMemcpyDeviceToDevice, s)); return 0;
      default: h->err = "pih_get_state: field not available for the random-fly task"; return -2;
    }
    hipLaunchKernelGGL(pih_soa_to_aos_kernel, dim3((n * nw + 255) / 256), dim3(256), 0, s, h->state, (float*)out_dev, n, PIH_FLY_STATE_WORDS, w0, nw);
    HIPCHK(h, hipGetLastError());
    return 0;
  }
  switch (field) {
    case PIH_FIELD_STATE: HIPCHK(h, hipMemcpyAsync(out_dev, h->state, (size_t)n * PIH_STATE_WORDS * sizeof(float), hipMemcpyDeviceToDevice, s)); return 0;
    case PIH_FIELD_TIP_POSE: hipLaunchKernelGGL(pih_gather_kernel, dim3((n * 7 + 255) / 256), dim3(256), 0, s, h->state, (float*)out_dev, n, (int)PIH_S_TIP, 7); break;
    case PIH_FIELD_CONTACT_FORCE: hipLaunchKernelGGL(pih_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, h->state, (float*)out_dev, n, (int)PIH_S_CFORCE, 1); break;
    case PIH_FIELD_EE_POS: hipLaunchKernelGGL(pih_gather_kernel, dim3((n * 3 + 255) / 256), dim3(256), 0, s, h->state, (float*)out_dev, n, (int)PIH_S_EE, 3); break;
    case PIH_FIELD_DEBUG:
      if (!h->dbg) { h->err = "pih_get_state: debug buffer not enabled (config.debug = 0)"; return -4; }
      HIPCHK(h, hipMemcpyAsync(out_dev, h->dbg, (size_t)n * PIH_DEBUG_WORDS * sizeof(float), hipMemcpyDeviceToDevice, s)); return 0;
    default: h->err = "pih_get_state: unknown field"; return -2;
  }
  HIPCHK(h, hipGetLastError());
  return 0;
}

int pih_set_state(pih_handle* h, int field, const void* in_dev, void* stream) {
  if (!h || !in_dev) return -2;
  if (field != PIH_FIELD_STATE) { h->err = "pih_set_state: only PIH_FIELD_STATE is writable"; return -2; }
  PIH_ENTER(h);
  if (h->fly) {
    const int n = h->cfg.n_envs;
    hipLaunchKernelGGL(pih_aos_to_soa_kernel, dim3((n * PIH_FLY_STATE_WORDS + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)in_dev, h->state, n, PIH_FLY_STATE_WORDS);
    HIPCHK(h, hipGetLastError());
    return 0;
  }
  HIPCHK(h, hipMemcpyAsync(h->state, in_dev, (size_t)h->cfg.n_envs * PIH_STATE_WORDS * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int pih_ik(pih_handle* h, int n, const float* q0_dev, const float* tpos_dev, const float* tquat_dev, float* qout_dev, void* stream) {
  if (!h || n <= 0 || !q0_dev || !tpos_dev || !tquat_dev || !qout_dev) return -2;
  PIH_ENTER(h);
  hipLaunchKernelGGL(pih_ik_kernel, dim3((n + 15) / 16), dim3(64), 0, (hipStream_t)stream, h->P, n, q0_dev, tpos_dev, tquat_dev, qout_dev);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int pih_ik_ur5(pih_handle* h, int n, const float* q0_dev, const float* tpos_dev, const float* tquat_dev, float* qout_dev, void* stream) {
  if (!h || n <= 0 || !q0_dev || !tpos_dev || !tquat_dev || !qout_dev) return -2;
  PIH_ENTER(h);
  hipLaunchKernelGGL(pih_ik_ur5_kernel, dim3((n + 15) / 16), dim3(64), 0, (hipStream_t)stream, h->P, n, q0_dev, tpos_dev, tquat_dev, qout_dev);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int pih_render(pih_handle* h, float* out_dev, int width, int height, int env_begin, int env_count, void* stream) {
  return pih_render_ex(h, out_dev, width, height, env_begin, env_count, 0, stream);
}

int pih_render_ex(pih_handle* h, float* out_dev, int width, int height, int env_begin, int env_count, int flags, void* stream) {
  if (!h || !out_dev || width <= 0 || height <= 0 || env_begin < 0 || env_count <= 0 || env_begin + env_count > h->cfg.n_envs) {
    if (h) h->err = "pih_render: bad arguments";
    return -2;
  }
  if (h->fly) { h->err = "pih_render: the wrist camera belongs to the peg-in-hole task"; return -2; }
  if ((reinterpret_cast<uintptr_t>(out_dev) & 15) != 0) { h->err = "pih_render: out_dev must be 16-byte aligned"; return -2; }
  PIH_ENTER(h);
  // every workgroup runs the forward kinematics of its env once: few strips per env when the batch alone gives the chip
  // several rounds of workgroups (>= 8192: images differ ~3x in cost, the tail matters), more strips for small batches
  int strips = (8192 + env_count - 1) / env_count;
  const int max_strips = (height + 31) / 32;
  strips = strips < 1 ? 1 : (strips > max_strips ? max_strips : strips);
  const int rows = ((height + strips - 1) / strips + 15) / 16 * 16;
  strips = (height + rows - 1) / rows;
  if (env_count > 65535) { h->err = "pih_render: env_count > 65535 per call"; return -2; }
  hipLaunchKernelGGL(pih_render_kernel, dim3(strips, env_count), dim3(RENDER_THREADS), 0, (hipStream_t)stream, h->state, out_dev, env_begin, width, height, rows, flags);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int pih_grasp_labels(pih_handle* h, float* out_dev, float* meta_dev, int size, int env_begin, int env_count, void* stream) {
  if (!h || !out_dev || size <= 0 || env_begin < 0 || env_count <= 0 || env_count > 65535 || env_begin + env_count > h->cfg.n_envs) {
    if (h) h->err = "pih_grasp_labels: bad arguments";
    return -2;
  }
  if (h->fly) { h->err = "pih_grasp_labels: the grasp labels belong to the peg-in-hole task"; return -2; }
  PIH_ENTER(h);
  hipLaunchKernelGGL(pih_labels_kernel, dim3((size * size + 1023) / 1024, env_count), dim3(256), 0, (hipStream_t)stream, h->state, out_dev, meta_dev, env_begin, size);
  HIPCHK(h, hipGetLastError());
  return 0;
}

int pih_set_timing(pih_handle* h, int enable) { if (!h) return -2; h->timing = enable > 0 ? enable : 0; h->timing_tick = 0; return 0; }

int pih_timing2(pih_handle* h, int reset, double* pre_ms_out, double* step_ms_out, int64_t* launches_out) {
  if (!h) return -2;
  PIH_ENTER(h);
  int r = drain_events(h);
  if (r) return r;
  if (h->errw_host && *h->errw_host) { h->err = CTRL_TIMEOUT_MSG; return -5; }
  if (pre_ms_out) *pre_ms_out = h->acc_n ? h->acc_pre_ms / (double)h->acc_n : 0.0;
  if (step_ms_out) *step_ms_out = h->acc_n ? h->acc_step_ms / (double)h->acc_n : 0.0;
  if (launches_out) *launches_out = h->acc_n;
  if (reset) { h->acc_pre_ms = h->acc_step_ms = 0; h->acc_n = 0; }
  return 0;
}

int pih_timing(pih_handle* h, int reset, double* avg_ms_out, int64_t* launches_out) {
  double a = 0, b = 0;
  int r = pih_timing2(h, reset, &a, &b, launches_out);
  if (r == 0 && avg_ms_out) *avg_ms_out = a + b;
  return r;
}

const char* pih_last_error(pih_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

}  // extern "C"
