// pih_wave.h -- the gfx950 wave layer of the per-env step: ONE WAVEFRONT (64 lanes, one workgroup) PER ENV.
// The wave context (`Wave`: lane-parallel loops, ballot/popcount stream compaction, shader-clock stamps) and the phases written
// directly with wave primitives: DPP row reductions, ds_bpermute lane exchanges, v_readlane broadcasts, inline v_add_f32_dpp
// row_bcast, 16-byte LDS broadcasts -- forward kinematics as a prefix product of rigid transforms, link velocities as prefix
// sums, the entry-parallel inward sweep of the articulated-body algorithm, and the sequential-impulse PGS.
#pragma once
#include <type_traits>
#include "pih_common.h"

namespace pih {

struct Wave {
  int l;
  int counter;
  static constexpr bool controller_inline = false;          // the controller runs in pih_pre_kernel, one env per lane
  real* dbg = nullptr; int dbgmode = 0; long long t0 = 0, t1 = 0;   // diagnostic shader-clock stamps (config.debug == 2; never read by the kernel)
  PIH_HD void stamp(int k) { if (dbg && dbgmode == 2) { long long t = __builtin_readcyclecounter(); if (l == 0) dbg[900 + k] = (real)(t - t0); t0 = t; } }   // sub-phases
  // debug words 940..949: start / end of the env's wave on the chip-wide 100 MHz clock (s_memrealtime; three 16-bit pieces each), HW_ID, XCC_ID: the
  // launch's dispatch timeline (tools/sched_trace.py)
  PIH_HD void trace(int at, long long t) {
    if (l != 0) return;
    dbg[at] = (real)(t & 0xFFFF); dbg[at + 1] = (real)((t >> 16) & 0xFFFF); dbg[at + 2] = (real)((t >> 32) & 0xFFFF);
    if (at == 940) { dbg[946] = (real)(__builtin_amdgcn_s_getreg((15 << 11) | 4) & 0xFFFF); dbg[947] = (real)__builtin_amdgcn_s_getreg((3 << 11) | 20); }
  }
  PIH_HD void phase_begin() { if (dbg && dbgmode == 2) { t1 = __builtin_readcyclecounter(); trace(940, (long long)__builtin_amdgcn_s_memrealtime()); } }
  PIH_HD void phase(int k) { if (dbg && dbgmode == 2) { long long t = __builtin_readcyclecounter(); if (l == 0) dbg[900 + k] = (real)(t - t1); t1 = t; if (k == 7) trace(943, (long long)__builtin_amdgcn_s_memrealtime()); } }
  PIH_HD int lane() const { return l; }
  // issue priority of this wavefront among the waves of its SIMD (s_setprio 0..3).  A launch of n envs ends when its heaviest
  // env ends, and that env shares its SIMD with a light one for most of its life: the heavy wave goes first at every issue slot.
  bool prio_on = true;
  PIH_HD void priority(int contacts) {
    if (!prio_on) return;
    if (contacts > 20) __builtin_amdgcn_s_setprio(3);
    else if (contacts > 14) __builtin_amdgcn_s_setprio(2);
    else if (contacts > 10) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
  }
  // Fused launch (pih_step_kernel with a controller role, round 4): the env's controller output arrives in a 16-word mailbox record,
  // published by the controller wavefront of the env's group of 64 with a release store of the launch's epoch to the group's flag.
  // Wave-uniform bounded wait (s_sleep between polls; on time-out the error word is set and the step proceeds with the targets it has --
  // a launch can never hang on this), then the 13 controller words go into the LDS copy of the state record.  cmail == nullptr:
  // two-launch path (pih_pre_kernel wrote the record before this kernel started), nothing to do.
  const int* cflags = nullptr; const real* cmail = nullptr; int* cerr = nullptr; int cepoch = 0, cenv = 0;
  PIH_HD void await_controller(Shared& sh) {
    if (!cmail) return;
    const int* f = cflags + (cenv >> 6);
    int tries = 0;
    // (flag and mailbox words are RELAXED agent-scope atomics: loads and stores that are coherent across the chip's eight L2s by
    //  themselves.  An agent-scope acquire would invalidate, and a release write back, the whole L2 of the XCD -- measured: + 50 us per
    //  launch with 4 096 acquires.  Order: the controller completes its mailbox stores (workgroup-scope release fence = wait for the
    //  stores to be acknowledged) before it stores the flag; this wave reads the mailbox after it has seen the flag.)
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < cepoch) {
      __builtin_amdgcn_s_sleep(4);
      if (++tries > (1 << 21)) { if (l == 0) __hip_atomic_store(cerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }   // (host-mapped word: the next pih_step fails with -5)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __syncthreads();
    if (l < 13) {
      const real v = __hip_atomic_load(cmail + (size_t)cenv * CTRL_WORDS + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int word = l < 9 ? PIH_S_TARGET + l : (l == 9 ? (int)PIH_S_FSM : l == 10 ? (int)PIH_S_FSMT : l == 11 ? (int)PIH_S_GRASP_ANGLE : (int)PIH_S_ATTACH_QZ);
      sh.S[word] = v;
    }
    __syncthreads();
  }
  PIH_HD void sync() { __syncthreads(); }
  template <class F> PIH_HD void par(int n, F f) {
    __syncthreads();
    for (int b = 0; b < n; b += 64) { int i = b + l; if (i < n) f(i); }
    __syncthreads();
  }
  // all lanes call f(i, in_range) for every chunk so that wave collectives inside f are legal
  template <class F> PIH_HD void par_all(int n, F f) {
    __syncthreads();
    for (int b = 0; b < n; b += 64) { int i = b + l; f(i, i < n); }
    __syncthreads();
  }
  PIH_HD void alloc_reset(int base) { counter = base; }
  PIH_HD int alloc_count() const { return counter; }
  PIH_HD int alloc(bool valid) {   // must be reached by all 64 lanes
    unsigned long long m = __ballot(valid);
    int slot = counter + __popcll(m & ((1ull << l) - 1ull));
    counter += __popcll(m);
    return valid ? slot : -1;
  }
};
// ------------------------------------------------------------------------------------------------ cross-lane helpers
// Two floats in one 64-bit VGPR pair, so that z += b * s moves both halves in ONE instruction (v_pk_fma_f32, s a wave-uniform scalar:
// gs_row2_* below).  Written
// as asm on an integer container on purpose: with float2 vector types this toolchain emits the packed FMA but then reads the wrong
// half of the pair in a following v_readlane (checked on the MI355X); the integer form extracts sub-registers correctly.
typedef unsigned long long pk2;
PIH_HD pk2 pk_pack(real x, real y) { return (pk2)__builtin_bit_cast(unsigned, x) | ((pk2)__builtin_bit_cast(unsigned, y) << 32); }
PIH_HD real pk_lo(pk2 v) { return __builtin_bit_cast(float, (unsigned)v); }
PIH_HD real pk_hi(pk2 v) { return __builtin_bit_cast(float, (unsigned)(v >> 32)); }
// acc += J * w on both halves, w a wave-uniform value held in the LOW (pk_fma_lo) or HIGH (pk_fma_hi) half of the VGPR pair w2
PIH_HD void pk_fma_lo(pk2& acc, pk2 j, pk2 w2) { __asm__("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(j), "v"(w2)); }
PIH_HD void pk_fma_hi(pk2& acc, pk2 j, pk2 w2) { __asm__("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(j), "v"(w2)); }
PIH_HD real rdlane(real v, int lane) {   // broadcast one lane's value (lane must be wave-uniform): v_readlane_b32
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// (x[G] = y[G] for ONE lane G known at compile time -- the "commit" of a row's multiplier -- is  s_lshl_b64 mask, 1, G ; v_cndmask :
//  the lane mask is made on the spot by the scalar unit, as a 64-bit constant operand it would be hoisted out of the solver loop, one
//  SGPR pair per row, and spilled.)
// One Gauss-Seidel row of the two-rows-per-lane solver (pgs_rows2) as ONE issue-ordered block:
//     cand = med3(z, lo, hi);  dl = cand - lam;  sdl = dl[lane G];  lam[G] = cand[G];  zz += col * sdl   (both halves, v_pk_fma_f32)
// (z: the half of zz the row lives in; G: its compile-time lane.)  A wave alone on its SIMD issues one instruction of ANY kind per
// 4 cycles, so what a row costs is its instruction count -- s_nop, s_waitcnt and scalar bookkeeping included -- and the heaviest env
// of a launch, which is what the launch waits for, runs this row 4 800 times per step.  Left to the scheduler a row came out as
// med3, sub, (s_lshl, cndmask), s_nop, readlane, s_waitcnt, pk_fma, s_nop; here the hazards of the gfx940 family are covered by
// instructions the row needs anyway: a VALU result is read by v_readlane one instruction later at the earliest (s_lshl_b64 sits
// between), and the SGPR a v_readlane wrote is read by the FMA two instructions later at the earliest (v_cndmask and the normal row's
// second v_readlane, or one s_nop in a friction row).  The broadcast travels through the fixed pair s[100:101] (an asm operand cannot
// name the low half of a 64-bit scalar operand; the high half is never read: op_sel_hi 0 takes the low half for both products).
// Normal row: also s0 = cand[G], the new normal multiplier the friction bounds follow.  INVARIANT (the compiler's hazard recognizer does
// not look inside an asm block, so every block must be safe by itself): s0's v_readlane is the THIRD instruction of the block
// -- one instruction after the v_max that produced cand, four before the block ends -- so the two wait states between a VALU write of
// an SGPR and a VALU read of it have passed whatever the scheduler places right after the block (round-3 review: it used to be the
// second-to-last instruction and relied on scalar compare / branch instructions happening to follow).  Order in a normal row:
//     v_max cand ; v_sub dl ; v_readlane s0 <- cand ; v_readlane s100 <- dl ; s_lshl mask ; v_cndmask lam ; v_pk_fma (reads s100).
// (A normal row's upper bound is PIH_BIG: its clamp is max(z, lo) -- TWO vector sources.  v_med3_f32 with three different VGPRs pays a
//  register-bank conflict whenever the allocator puts two of them in one bank; with that clamp in every row, builds whose loops were
//  instruction for instruction the same ran at 1 640 or 1 716 cycles per iteration.  The motor rows' bounds are symmetric: -hi, hi.)
PIH_HD void gs_row2_normal(pk2& zz, real z, pk2 col, real lo, real& lam, int G, real& dl, real& s0) {
  real cand; unsigned long long m;
  __asm__ volatile("v_max_f32 %0, %6, %7\n\tv_sub_f32 %1, %0, %3\n\tv_readlane_b32 %2, %0, %9\n\tv_readlane_b32 s100, %1, %9\n\t"
                   "s_lshl_b64 %4, 1, %9\n\tv_cndmask_b32_e64 %3, %3, %0, %4\n\tv_pk_fma_f32 %5, s[100:101], %8, %5 op_sel_hi:[0,1,1]"
                   : "=&v"(cand), "=&v"(dl), "=&s"(s0), "+v"(lam), "=&s"(m), "+v"(zz) : "v"(z), "v"(lo), "v"(col), "n"(G) : "scc", "s100", "s101");
}
// A row with symmetric bounds and no follower rows (the pipe motor rows)
PIH_HD void gs_row2_bounded(pk2& zz, real z, pk2 col, real hi, real& lam, int G, real& dl) {      // bounds -hi .. hi
  real cand; unsigned long long m;
  __asm__ volatile("v_med3_f32 %0, %5, -%6, %6\n\tv_sub_f32 %1, %0, %2\n\ts_lshl_b64 %3, 1, %8\n\tv_readlane_b32 s100, %1, %8\n\t"
                   "v_cndmask_b32_e64 %2, %2, %0, %3\n\ts_nop 0\n\tv_pk_fma_f32 %4, s[100:101], %7, %4 op_sel_hi:[0,1,1]"
                   : "=&v"(cand), "=&v"(dl), "+v"(lam), "=&s"(m), "+v"(zz) : "v"(z), "v"(hi), "v"(col), "n"(G) : "scc", "s100", "s101");
}
// Friction row: bounds -h .. h.  (One block per row, not per pair as in pgs_rows: the second row would have to read the half of zz
// the first one has just written, and an asm operand cannot name a half of a 64-bit operand -- a separate input operand is by
// contract the value at entry.  The s_waitcnt for the row's streamed column sits between two blocks anyway.)
PIH_HD void gs_row2_friction(pk2& zz, real z, pk2 col, real h, real& lam, int G, real& dl) {
  real cand; unsigned long long m;
  __asm__ volatile("v_med3_f32 %0, %5, -%6, %6\n\tv_sub_f32 %1, %0, %2\n\ts_lshl_b64 %3, 1, %8\n\tv_readlane_b32 s100, %1, %8\n\t"
                   "v_cndmask_b32_e64 %2, %2, %0, %3\n\ts_nop 0\n\tv_pk_fma_f32 %4, s[100:101], %7, %4 op_sel_hi:[0,1,1]"
                   : "=&v"(cand), "=&v"(dl), "+v"(lam), "=&s"(m), "+v"(zz) : "v"(z), "v"(h), "v"(col), "n"(G) : "scc", "s100", "s101");
}
// The same for the one-row-per-lane solver (pgs_rows): z += b * sdl with v_fmac_f32, the broadcast in any SGPR; the two friction rows
// of a contact (Bullet runs both or neither) are one block, so that nothing is inserted between them.
PIH_HD void gs_row1_normal(real& z, real b, real lo, real& lam, int G, real& dl, real& s0) {
  real cand, sdl; unsigned long long m;
  __asm__ volatile("v_max_f32 %0, %6, %7\n\tv_sub_f32 %1, %0, %4\n\tv_readlane_b32 %3, %0, %9\n\tv_readlane_b32 %2, %1, %9\n\t"
                   "s_lshl_b64 %5, 1, %9\n\tv_cndmask_b32_e64 %4, %4, %0, %5\n\tv_fmac_f32 %6, %2, %8"
                   : "=&v"(cand), "=&v"(dl), "=&s"(sdl), "=&s"(s0), "+v"(lam), "=&s"(m), "+v"(z) : "v"(lo), "v"(b), "n"(G) : "scc");
}
PIH_HD void gs_row1_friction2(real& z, real b1, real b2, real h, real& lam, int G1, int G2, real& dl1, real& dl2) {
  real cand, sdl; unsigned long long m;
  __asm__ volatile("v_med3_f32 %0, %6, -%7, %7\n\tv_sub_f32 %1, %0, %4\n\ts_lshl_b64 %5, 1, %10\n\tv_readlane_b32 %3, %1, %10\n\t"
                   "v_cndmask_b32_e64 %4, %4, %0, %5\n\ts_nop 0\n\tv_fmac_f32 %6, %3, %8\n\t"
                   "v_med3_f32 %0, %6, -%7, %7\n\tv_sub_f32 %2, %0, %4\n\ts_lshl_b64 %5, 1, %11\n\tv_readlane_b32 %3, %2, %11\n\t"
                   "v_cndmask_b32_e64 %4, %4, %0, %5\n\ts_nop 0\n\tv_fmac_f32 %6, %3, %9"
                   : "=&v"(cand), "=&v"(dl1), "=&v"(dl2), "=&s"(sdl), "+v"(lam), "=&s"(m), "+v"(z) : "v"(h), "v"(b1), "v"(b2), "n"(G1), "n"(G2) : "scc");
}
template <int CTRL> PIH_HD real dpp_add(real x) {   // x + x[dpp-permuted lane]  (v_add_f32_dpp)
  return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
PIH_HD real sum8(real x) {               // sum over each aligned group of 8 lanes, result in all 8
  x = dpp_add<0xB1>(x);                  // quad_perm [1,0,3,2]
  x = dpp_add<0x4E>(x);                  // quad_perm [2,3,0,1]
  return dpp_add<0x141>(x);              // row_half_mirror
}
PIH_HD real from_lane(real v, int byte_addr) {   // v of lane byte_addr / 4 (per-lane source): ds_bpermute_b32, no LDS memory, no VALU slot
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
// ------------------------------------------------------------------------------------------------ forward kinematics
// GPU form: lane = link.  The world pose of a link is the product of the local transforms along its chain, i.e. an inclusive
// prefix "product" of rigid transforms: five Hillis-Steele steps (12 ds_bpermute + 39 FMA each) instead of a 33-link serial
// composition, and the pose never leaves the lane's registers before the world axes / rotated inertias are written.
// Out-of-chain sources read lane 63, which holds the identity.  Finger 8 is a child of link 6, not of finger 7: it takes
// finger 7's position in the chain order.  The arm's base rotation is folded into link 0's local transform.
PIH_HD int lane_byte(int lane) { return 4 * lane; }
PIH_HD void fk_all(Wave& w, Shared& sh) {
  w.sync();
  const int lane = w.lane();
  const bool active = lane < NL;
  const int L = active ? lane : NL - 1;
  M3 R = ldm(IDENT3); V3 o = mk(0, 0, 0);
  if (active) {
    real T[12];
    const real q = L < ANL ? sh.S[PIH_S_QARM + L] : (L == ANL ? (real)0 : sh.S[PIH_S_QJ + L - ANL - 1]);
    local_transform(L, q, sh.S, T);
    R = ldm(T); o = ld3(T + 9);
    if (L == 0) { const M3 B = ldm(ARM_BASE_R); o = mul(B, o); R = mul(B, R); }
  }
  const int cs = L < ANL ? 0 : ANL;
  const int vidx = L == ANL - 1 ? ANL - 2 : lane;
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const int src = vidx - off;
    const int sb = lane_byte((active && src >= cs) ? src : 63);
    M3 Rs; V3 os;
#pragma unroll
    for (int k = 0; k < 9; k++) Rs.m[k] = from_lane(R.m[k], sb);
    os = mk(from_lane(o.x, sb), from_lane(o.y, sb), from_lane(o.z, sb));
    o = os + mul(Rs, o); R = mul(Rs, R);
  }
  if (active) {
    stm(sh.a.LR[L], R); st3(sh.LO[L], o);
    st3(sh.LA[L], mul(R, ld3(L_AXIS[L])));
    st3(sh.a.LRC[L], mul(R, ld3(L_COM[L])));
    sts3(sh.a.LIC[L], rot_sym(R, lds3(L_INERTIA[L])));
  }
  w.sync();
}
// ------------------------------------------------------------------------------------------------ link velocities
// link velocities from the generalized velocity sh.u: omega_L and the velocity of the link-origin point
// The same recurrences as two inclusive prefix sums along the chains (lane = link): omega_L = sum over the path of the joint
// angular rates, v_L = sum over the path of (omega_parent x (o_L - o_parent) + prismatic rate).  Hillis-Steele steps through
// ds_bpermute (5 steps cover the 24-link pipe); finger 8 is a child of link 6, not of finger 7, so it drops finger 7's terms.
PIH_HD V3 from_lane3(V3 v, int byte_addr) { return mk(from_lane(v.x, byte_addr), from_lane(v.y, byte_addr), from_lane(v.z, byte_addr)); }
PIH_HD V3 chain_prefix_sum(V3 t, int lane, int chain_start) {
  V3 acc = t;
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const int src = lane - off;
    const V3 o = from_lane3(acc, 4 * (src < 0 ? 0 : src));
    if (src >= chain_start) acc = acc + o;
  }
  return acc;
}
PIH_HD void link_velocities_scan(Shared& sh, int lane) {
  const bool active = lane < NL;
  const int L = active ? lane : NL - 1;
  const int jt = L == ANL ? PIH_JT_FLOATING : ((L == ANL - 1 || L == ANL - 2) ? PIH_JT_PRISMATIC : PIH_JT_REVOLUTE);
  const int par = (L == 0 || L == ANL) ? -1 : (L == ANL - 1 ? ANL - 3 : L - 1);
  const int cs = L < ANL ? 0 : ANL, d = link_dof(L);
  const V3 aq = sh.u[d] * ld3(sh.LA[L]);
  const V3 tw = jt == PIH_JT_FLOATING ? ld3(&sh.u[d + 3]) : (jt == PIH_JT_REVOLUTE ? aq : mk(0, 0, 0));
  V3 wv = chain_prefix_sum(tw, lane, cs);
  const V3 tw7 = from_lane3(tw, 4 * (ANL - 2));
  if (L == ANL - 1) wv = wv - tw7;
  const V3 wp = from_lane3(wv, 4 * (par < 0 ? 0 : par));
  V3 tv;
  if (jt == PIH_JT_FLOATING) tv = ld3(&sh.u[d]);
  else {
    tv = jt == PIH_JT_PRISMATIC ? aq : mk(0, 0, 0);
    if (par >= 0) tv = tv + cross(wp, ld3(sh.LO[L]) - ld3(sh.LO[par]));
  }
  V3 vv = chain_prefix_sum(tv, lane, cs);
  const V3 tv7 = from_lane3(tv, 4 * (ANL - 2));
  if (L == ANL - 1) vv = vv - tv7;
  if (active) { st3(sh.VW[L], wv); st3(sh.VV[L], vv); }
}
PIH_HD void link_velocities(Wave& w, Shared& sh) { w.sync(); link_velocities_scan(sh, w.lane()); w.sync(); }

// ------------------------------------------------------------------------------------------------ ABA inward sweep
// Inward sweep of the articulated-body algorithm, LANE = ENTRY of the articulated inertia: lane l < 48 owns entry
// (i, j) = (l >> 3, l & 7) of the 6 x 8 array [ I^A (6 x 6, rows/cols 0-2 angular, 3-5 linear, i.e. [[A, B], [B^T, C]]) |
// p^A (column 6) | - ].  Per link:
//   1. m = own inertia of the link + what the child handed up
//   2. U = I^A S  (S = [a; 0] revolute, [0; a] prismatic), D = S.U, u = tau - S.p^A
//   3. I^a = I^A - U U^T / D ;  column 6 = p^a = p^A + I^a c + U u / D
//   4. translate to the parent's origin (r = o_L - o_parent):  B' = B + [r]x C,  A' = A + [r]x B^T - B' [r]x,
//      p_a' = p_a + r x p_l   (entry formulas: ([r]x X)_ij = r_i1 X_i2,j - r_i2 X_i1,j ; (X [r]x)_ij = X_i,j1 r_j2 - X_i,j2 r_j1)
// The arm's two fingers (links 7, 8) both feed link 6: finger 8's hand-up is parked in a register until finger 7 is done.
PIH_HD void aba_inward(Wave& w, Shared& sh, areal* rootp) {
  real* const Mx = sh.r_lam;      // 48 words of scratch for the root inversion (r_lam is dead until the rows are built)
  // GPU form of the same four steps: the entry stays in a register of its lane; row sums are DPP reductions over the 8-lane
  // row group, entries of other rows come through ds_bpermute (no LDS memory, no VALU slot), wave-uniform scalars (D, u, the
  // U vector) through v_readlane.  The translation is branch-free: every lane evaluates
  //     v + rA X1 - rB X2 - ( (X3 + rA X4 - rB X5) rG - (X6 + rA X7 - rB X8) rK )
  // with per-lane source lanes and 0/1 masks fixed before the loop (B, B^T and p_a lanes use the first two terms only,
  // C / p_l / idle lanes none), so one batch of 8 bpermutes and one wait serve the whole step.
  {
    const int l = w.lane(), i = l >> 3, j = l & 7;
    const int ia = i < 3 ? i : (i < 6 ? i - 3 : 0), ja = j < 3 ? j : (j < 6 ? j - 3 : 0);
    const int i1 = ia == 2 ? 0 : ia + 1, i2 = ia == 0 ? 2 : ia - 1, j1 = ja == 2 ? 0 : ja + 1, j2 = ja == 0 ? 2 : ja - 1;
    const bool typeA = i < 3 && j < 3, typeB = i < 3 && j >= 3 && j < 6, typeBt = i >= 3 && i < 6 && j < 3, typeP = i < 3 && j == 6;
    const int own_off = aba_own_word(i, j);
    // first pair of terms: coefficient indices into r and source lanes
    int kA = 0, kB = 0, s1 = l, s2 = l;
    if (typeA) { kA = i1; kB = i2; s1 = 8 * j + 3 + i2; s2 = 8 * j + 3 + i1; }
    else if (typeB) { kA = i1; kB = i2; s1 = 8 * (3 + i2) + j; s2 = 8 * (3 + i1) + j; }
    else if (typeBt) { kA = j1; kB = j2; s1 = 8 * (3 + j2) + i; s2 = 8 * (3 + j1) + i; }
    else if (typeP) { kA = i1; kB = i2; s1 = 8 * (3 + i2) + 6; s2 = 8 * (3 + i1) + 6; }
    const real m1 = (typeA || typeB || typeBt || typeP) ? (real)1 : (real)0, m2 = typeA ? (real)1 : (real)0;
    int s3 = l, s4 = l, s5 = l, s6 = l, s7 = l, s8 = l;
    if (typeA) { s3 = 8 * i + 3 + j1; s4 = 8 * (3 + i2) + 3 + j1; s5 = 8 * (3 + i1) + 3 + j1; s6 = 8 * i + 3 + j2; s7 = 8 * (3 + i2) + 3 + j2; s8 = 8 * (3 + i1) + 3 + j2; }
    const int sUj = j < 6 ? 8 * j : l;                                 // any lane of row j holds U_j
    // one link with a joint (steps 2-4): stores U, 1/D, u; returns the lane's entry of what is handed up to the parent (garbage for
    // the arm root, whose parent is the fixed world: `up` = false skips the translation)
    auto link = [&](int L, real m, bool up) __attribute__((always_inline)) -> real {
      const int jt = joint_type(L);                                    // (no table load on the chain: pih_common.h)
      const int sb = jt == PIH_JT_REVOLUTE ? 0 : 3;
      const V3 a = ld3(sh.LA[L]);
      const real rA = m1 * sh.AR[L][kA], rB = m1 * sh.AR[L][kB], rG = m2 * sh.AR[L][j2], rK = m2 * sh.AR[L][j1];
      const real cj = j < 6 ? sh.a.CB[L][j] : (real)0;
      const int js = j - sb;
      const real sj = js == 0 ? a.x : (js == 1 ? a.y : (js == 2 ? a.z : (real)0));
      const real Ui = sum8(m * sj);                                    // U_i = sum_k I^A[i][sb + k] a_k, in every lane of row i
      const real Uj = from_lane(Ui, 4 * sUj);
      const real D = a.x * rdlane(Ui, 8 * sb) + a.y * rdlane(Ui, 8 * (sb + 1)) + a.z * rdlane(Ui, 8 * (sb + 2));
      const real tau = 0;                                               // (-L_DAMPING[L] * u: no joint damping in the model, asserted in pih_common.h)
      const real u = tau - (a.x * rdlane(m, 8 * sb + 6) + a.y * rdlane(m, 8 * (sb + 1) + 6) + a.z * rdlane(m, 8 * (sb + 2) + 6));
      const real Di = (real)1 / D;
      if (j == 0 && i < 6) sh.AU[L][i] = Ui;
      if (l == 0) { sh.ADinv[L] = Di; sh.Au[L] = u; }
      if (!up) return 0;
      real ma = j < 6 ? m - Ui * Uj * Di : m;                          // I^a ; column 6 is fixed up next
      const real s = sum8(j < 6 ? ma * cj : (real)0);                  // (I^a c)_i in every lane of row i
      if (j == 6) ma = m + s + Ui * (u * Di);                          // p^a = p^A + I^a c + U u / D
      const real X1 = from_lane(ma, 4 * s1), X2 = from_lane(ma, 4 * s2), X3 = from_lane(ma, 4 * s3), X4 = from_lane(ma, 4 * s4),
                 X5 = from_lane(ma, 4 * s5), X6 = from_lane(ma, 4 * s6), X7 = from_lane(ma, 4 * s7), X8 = from_lane(ma, 4 * s8);
      return ma + rA * X1 - rB * X2 - ((X3 + rA * X4 - rB * X5) * rG - (X6 + rA * X7 - rB * X8) * rK);
    };
    // The arm chain (links 8 .. 0) and the pipe chain (33 .. 9) are independent: the first eight steps take one link of each in
    // the same basic block, so that the two dependent chains (LDS read -> DPP sums -> bpermute -> rcp -> DPP sums -> 8 bpermutes)
    // fill each other's latencies; the remaining sixteen pipe links, the floating pipe root and the arm root follow.
    real carry_p = 0, carry_a = 0, hold = 0;
#pragma nounroll
    for (int k = 0; k < 8; k++) {
      const int Lp = NL - 1 - k, La = ANL - 1 - k;                     // pipe 33 .. 26, arm 8 .. 1
      const real own_p = sh.a.IAP[Lp][own_off], own_a = sh.a.IAP[La][own_off];
      const real mp = k == 0 ? own_p : own_p + carry_p;                // link 33 is the pipe's leaf
      const real ma = k < 2 ? own_a : own_a + carry_a;                 // links 8 and 7 (the fingers) are leaves
      const real vp = link(Lp, mp, true), va = link(La, ma, true);
      carry_p = vp;
      if (k == 0) hold = va;                                           // finger 8: parked until finger 7 is done
      else if (k == 1) carry_a = va + hold;                            // finger 7: both fingers feed link 6
      else carry_a = va;
    }
#pragma nounroll
    for (int L = NL - 9; L > ANL; L--) carry_p = link(L, sh.a.IAP[L][own_off] + carry_p, true);
    {                                                                  // floating pipe root
      const real m = sh.a.IAP[ANL][own_off] + carry_p;
      w.sync(); if (l < 48) Mx[l] = m; w.sync();
      aba_root_inverse(sh, Mx, rootp);
    }
    link(0, sh.a.IAP[0][own_off] + carry_a, false);                    // arm root: parent is the fixed world
    w.sync();
  }
}

// ------------------------------------------------------------------------------------------------ motor response rows
// pull the 32 motor response rows out of their LDS staging words into registers (lane = DOF: arm lanes hold column d of the
// 9 x 9 arm block, pipe lanes column d - 9 of the 23 x 29 pipe block) before the contact rows of the second pass overwrite them
PIH_HD void pull_motor_rows(Wave& w, Shared& sh, MotorW& mw) {
  const int d = w.lane();
#pragma unroll
  for (int j = 0; j < PIH_OBJ_NJ; j++) mw.w[j] = d < 9 ? (j < 9 ? wma_row(sh, j)[d] : (real)0) : (d < ND ? wmp_row(sh, j)[d - 9] : (real)0);
}

// ------------------------------------------------------------------------------------------------ PGS
// after row16_sum3: rows 1,3 += lane 15 of the previous row (row_bcast:15), rows 2,3 += lane 31 (row_bcast:31)
// => lanes 32..47 hold the sum over lanes 0..47 (all 38 DOF lanes).  Written as inline asm because hipcc lowers the
// masked-row form to v_mov 0 + v_mov_dpp + v_add (3 instructions) instead of one fused v_add_f32_dpp; the s_nop covers the
// VALU-write -> DPP-read hazard that the compiler does not pad inside asm.
PIH_HD void rows012_total3(real& a, real& b, real& c) {
  __asm__ volatile("s_nop 1\n\t"
                   "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa\n\t"
                   "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa\n\t"
                   "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa\n\t"
                   "s_nop 1\n\t"
                   "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc\n\t"
                   "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc\n\t"
                   "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc\n\t"
                   "s_nop 1"
                   : "+v"(a), "+v"(b), "+v"(c));
}
// after this every lane holds the sum over its 16-lane row; three independent reductions interleaved for ILP
PIH_HD void row16_sum3(real& a, real& b, real& c) {
  a = dpp_add<0xB1>(a); b = dpp_add<0xB1>(b); c = dpp_add<0xB1>(c);        // quad_perm [1,0,3,2]
  a = dpp_add<0x4E>(a); b = dpp_add<0x4E>(b); c = dpp_add<0x4E>(c);        // quad_perm [2,3,0,1]
  a = dpp_add<0x141>(a); b = dpp_add<0x141>(b); c = dpp_add<0x141>(c);     // row_half_mirror
  a = dpp_add<0x140>(a); b = dpp_add<0x140>(b); c = dpp_add<0x140>(c);     // row_mirror
}


// ---- PGS in ROW space for envs with few contacts (the common case: 70-80 % of the env-steps of the benchmark workload)
// When all rows fit one wavefront -- 32 motor rows + 3 x (<= MERGED_CONTACTS) contact rows <= 62 -- lane g owns row g and holds
//   z = lambda + rhs - dinv (J_g . du)   "the multiplier the row would take if it were not clamped", given every impulse applied so
//                                        far (the arm joints' lanes, whose three rows share one lane, hold -dinv (J_g . du)), and
//   Bn[i] = [g == i] - dinv A[g][i]      from the lane's row of the Delassus matrix A = J M^-1 J^T, for EVERY row i (62 registers;
//                                        motor columns by symmetry from the lane's own response row, contact columns as J_g . W_i
//                                        against the response rows that build_rows staged in LDS).
// A row update then needs no Jacobian and no cross-lane reduction: every lane clamps its own z (v_med3), subtracts its multiplier,
// the row's lane supplies the step through one v_readlane, and ONE FMA moves every z (z += Bn[i] * step; the updated row's own z
// stays put because dinv A[i][i] = 1) -- with the same sequence of row updates as Bullet (per arm joint: motor, lower, upper
// limit; the 23 pipe motors; per contact: normal, dir1, dir2).  The DOF velocities are recovered at the end as du = sum_i W_i
// lambda_i.  The arm-row and pipe-row columns are disjoint (A[arm][pipe motor] = 0), so the motor chain runs on two accumulators
// for ILP exactly like the DOF-space chain.  (tools/micro/rowchain.hip times the row chain in isolation.)
// The iteration loop of all three solvers and the cadence of Bullet's early exit (largest squared row residual <= residual_threshold).
// Bullet evaluates the test after EVERY iteration; `stride` = pih_config.exit_check_stride selects how often the product does:
//   stride = 1: every iteration (Bullet's cadence);
//   stride = s > 1 (default 16): iterations 1..4 -- where it actually fires: envs in free flight converge in two -- then iterations
//   4 + s k and the last one; in between the body runs without the per-row compare (one v_cmp + one scalar OR per row, ~15 % of a row
//   update).  An env that would have met the threshold between two tests performs at most s - 1 further iterations whose updates are
//   all below the threshold.  The oracle has the same switch (piho_config.exit_check_stride); tests/test_gpu_defaults.py compares the
//   product at its defaults with the oracle at both cadences.
// The unchecked body is instantiated twice per trip: the multipliers are loop-carried, and with a single copy every new value is
// moved back into the register the loop header expects.
template <bool DOUBLED = true, class FC, class FN> PIH_HD int pgs_iteration_loop(int iters, int stride, FC checked, FN unchecked) {
  int it = 0;
  const int lead = stride <= 1 ? iters : 4;
  // iterations 1..lead with the test
  while (it < iters && it < lead) { it++; if (checked()) return it; }
  // then groups of `stride`: stride - 1 without, one with (the last iteration always with)
  while (it < iters) {
    const int stop = it + stride - 1 < iters - 1 ? it + stride - 1 : iters - 1;
    if (DOUBLED) {
      while (it < stop) {
        it++; unchecked();
        if (it >= stop) break;
        it++; unchecked();
      }
    } else {
#pragma nounroll
      while (it < stop) { it++; unchecked(); }
    }
    it++; if (checked()) return it;
  }
  return it;
}
constexpr int FR = NMOT + 3 * MERGED_CONTACTS;
PIH_HD int pgs_rows(Wave& w, Shared& sh, const Params& P) {
  const int nc = __builtin_amdgcn_readfirstlane(sh.nc);
  w.sync();
  const int lane = w.lane();
  const bool hasarm = __builtin_amdgcn_readfirstlane(sh.nca) != 0;       // does any contact involve an arm link?
  // ---- this lane's Jacobian row (38 entries, registers): unit vector for a motor row, point Jacobian for a contact row
  real J[ND];
  {
    const bool ismotor = lane < NMOT;
    const int row = ismotor ? 0 : lane - NMOT, c = row / 3, k = row - 3 * c;
    const bool live = !ismotor && c < nc;
    int la = -1, lb = -1; V3 p = mk(0, 0, 0), dir = mk(0, 0, 0); bool ang = false;
    if (live) { la = sh.c_la[c]; lb = sh.c_lb[c]; const real* R = sh.b.crec[c]; p = ld3(R); dir = ld3(R + 8 + 4 * k); ang = R[6] != 0; }
    const int md = lane < 9 ? lane : 15 + (lane - 9);
    // No contact involves the arm in 99.9 % of the env-steps of a random-action rollout (sh.nca, counted by collide): the nine arm
    // entries of every contact row's Jacobian and response row are then exact zeros, and both the entries and their products are
    // left out (wave-uniform branches; the sums are unchanged, a skipped term is 0 x 0)
    if (!hasarm) {
#pragma unroll
      for (int d = 0; d < 9; d++) J[d] = ismotor && d == md ? (real)1 : (real)0;
    }
    DofGeom gn = dof_geom(sh, hasarm ? 0 : 9);
#pragma unroll
    for (int d = 0; d < ND; d++) {
      if (d < 9 && !hasarm) continue;
      const DofGeom g = gn;
      if (d + 1 < ND) gn = dof_geom(sh, d + 1);           // the next DOF's axis / origin are in flight while this entry is computed
      const real jc = live ? jac_entry(g, la, lb, p, dir, ang) : (real)0;
      J[d] = ismotor ? (d == md ? (real)1 : (real)0) : jc;
      __asm__ volatile("" : "+v"(J[d]) :: "memory");     // one DOF at a time (same reason as for the columns of A below)
    }
  }
  // ---- this lane's row of A.  Motor columns need no dot product: the Delassus matrix is symmetric, A[r][motor m] = (M^-1 J_r^T)[dof(m)]
  // = W_r[dof(m)] -- a contact row reads entry dof(m) of its OWN response row, a motor row the entry of motor m's staged row at its
  // own DOF (arm x pipe entries are 0): 32 LDS reads per lane instead of 748 FMAs + 748 broadcasts
  real A[FR];
  {
    const int row = lane - NMOT;                                         // contact-row index (lanes >= 32)
    const bool crow = row >= 0 && row < 3 * nc;
    const real* wown = crow ? sh.b.Wp[row] : sh.b.Wp[0];
#pragma unroll
    for (int m = 0; m < NMOT; m++) {
      const int dm = m < 9 ? m : 15 + (m - 9);                            // the motor's DOF
      real a = 0;
      if (crow) a = wown[dm];
      else if (m < 9) { if (lane < 9) a = wma_row(sh, m)[lane]; }
      else if (lane >= 9 && lane < NMOT) a = wmp_row(sh, m - 9)[lane - 3];   // DOF of pipe motor lane: 15 + (lane - 9); row entries start at DOF 9
      if (m % 4 == 3) __asm__ volatile("" : "+v"(a) :: "memory");          // four LDS reads in flight, not one (and not all 32: spills)
      A[m] = a;
    }
  }
#pragma unroll
  for (int c = 0; c < MERGED_CONTACTS; c++) {
    if (c < nc) {
      // (the three response rows of a contact are requested together: one LDS round trip per contact instead of one per column)
      real a0 = 0, a1 = 0, a2 = 0;
      const real* wr = sh.b.Wp[3 * c];
      if (hasarm) {
#pragma unroll
        for (int d = 0; d < 9; d++) { a0 += J[d] * wr[d]; a1 += J[d] * wr[WPS + d]; a2 += J[d] * wr[2 * WPS + d]; }
      }
#pragma unroll
      for (int d = 9; d < ND; d++) { a0 += J[d] * wr[d]; a1 += J[d] * wr[WPS + d]; a2 += J[d] * wr[2 * WPS + d]; }
      __asm__ volatile("" : "+v"(a0), "+v"(a1), "+v"(a2) :: "memory");
      A[NMOT + 3 * c] = a0; A[NMOT + 3 * c + 1] = a1; A[NMOT + 3 * c + 2] = a2;
    } else { A[NMOT + 3 * c] = 0; A[NMOT + 3 * c + 1] = 0; A[NMOT + 3 * c + 2] = 0; }
  }
  // ---- per-lane row constants.  The solve runs on z = lambda + rhs - dinv (J du) of the lane's own row ("the multiplier the row
  // would take if it were not clamped"): a row update is then  lambda' = clamp(z),  d = lambda' - lambda,  and every lane's z moves
  // by  -dinv A[i] d  -- except the updated row's own z, which stays where it is (dinv A[i][i] = 1).  With Bn[i] = [lane == i] -
  // dinv A[i] (in the registers that held A) the whole update of row i is
  //     clamp (v_med3) -> subtract -> v_readlane -> one FMA on z
  // three dependent VALU operations and a readlane instead of readlane -> FMA -> clamp -> subtract -> FMA, no row constant is
  // fetched inside the loop (bounds, threshold and multiplier of a row live in its lane), and the running multipliers of the pipe
  // motors and the contacts need no wave-uniform registers.  The arm joints keep their uniform chain (motor, lower, upper limit
  // share one lane): their lanes hold z = -dinv (J du).
  w.sync();
  real di = 0, rhs = 0, thr = PIH_BIG, lbv = 0, ubv = 0, cmu = 0, cfl = 0, lam0 = 0;
  if (lane < 9) di = sh.mrec[lane][0];
  else if (lane < NMOT) { di = sh.mrec[lane][0]; rhs = sh.mrec[lane][1]; thr = sh.mrec[lane][2]; ubv = sh.mrec[lane][3]; lbv = -ubv; }
  else {
    const int row = lane - NMOT, c = row / 3, k = row - 3 * c;
    if (c < nc) {
      const real* R = sh.b.crec[c];
      di = R[11 + 4 * k]; rhs = R[20 + k]; thr = R[29 + k]; cmu = R[5]; cfl = R[4];
      if (k == 0) { lbv = R[3]; ubv = PIH_BIG; lam0 = sh.r_lam[3 * c]; }      // friction rows: bounds set per iteration from the normal multiplier
    }
  }
  // (sqrt(resid) * dinv of a contact row sits in words 29..31 of its record, the multipliers go back to sh.r_lam)
#pragma unroll
  for (int i = 0; i < FR; i++) { A[i] = (lane == i && lane >= 9 ? (real)1 : (real)0) - di * A[i]; }
  real* const Bn = A;
  unsigned angmask = 0;                    // contacts whose friction rows are always solved (mu < 0: the attach weld's rows)
#pragma unroll
  for (int c = 0; c < MERGED_CONTACTS; c++) if (c < nc && sh.b.crec[c][5] < 0) angmask |= 1u << c;
  angmask = (unsigned)__builtin_amdgcn_readfirstlane((int)angmask);
  // ---- the limit rows of arm joints 0..6 are exact no-ops for the whole solve when (a) the joint's motor row is never clamped --
  // an unclamped velocity motor sets the joint's velocity change to (target - current) whatever it was, because J W dinv = 1 -- and
  // (b) that target velocity violates neither limit speed: then  rhs_limit - (J du) dinv = (v_limit - v_target) dinv < 0  every time
  // and the multiplier stays 0.  (b) is known before the solve (skip7, wave-uniform); (a) is watched during the solve (one compare
  // per arm motor row) and, should a motor ever clamp, the solve is simply run again with every limit row in place.  The finger
  // joints (targets far outside [0, 0.04] most of the time) always keep their limit rows.
  bool skip7 = true;
  {
    const real dt = P.dt;
#pragma unroll
    for (int j = 0; j < 7; j++) {
      const real q = sh.S[PIH_S_QARM + j], uj = sh.u[j];
      const real vt = sh.mrec[j][1] * sh.lrec[j][2] + uj;                         // rhs / dinv + u = the motor's target velocity
      const real plo = q - L_LO[j], phi = L_HI[j] - q;
      const real vlo = plo > 0 ? -plo / dt : -P.erp * plo / dt, vhi = phi > 0 ? -phi / dt : -P.erp * phi / dt;
      const real tol = (real)1e-4 * ((real)1 + absr(vt) + absr(uj) + absr(vlo) + absr(vhi));
      skip7 = skip7 && (vt - vlo > tol) && (-vhi - vt > tol);
    }
    // only with the huge impulse bound of the action-mode controller (1e5 dt): the scripted controller's 1200 dt does clamp
    skip7 = skip7 && sh.mrec[0][3] >= (real)100;
    skip7 = __builtin_amdgcn_readfirstlane((int)skip7) != 0;
  }
  real armlim = sh.mrec[0][3];
#pragma unroll
  for (int j = 1; j < 7; j++) armlim = sh.mrec[j][3] < armlim ? sh.mrec[j][3] : armlim;
  if (lane < 9) sh.lrec[lane][3] = sh.lrec[lane][2] * sh.mrec[lane][0];          // (J W) dinv of the arm joint (1 up to rounding)
  w.sync();
  // ---- multipliers: arm rows wave-uniform in VGPRs, every other row in its own lane
  real lam_a[9], lam_lo[9], lam_hi[9], lamr[NMOT];     // pipe motor rows: wave-uniform multipliers (lam[g] += step: one issue slot)
  real lam = 0, z = 0;
  int it = 0;
  // commit one lane of a per-lane register: x[g] = y[g] (the lane mask is a compile-time constant in an SGPR pair)
  auto solve = [&](auto FULLTAG) __attribute__((always_inline)) -> bool {     // returns true if an arm motor row clamped
    constexpr bool FULL = decltype(FULLTAG)::value;
#pragma unroll
    for (int j = 0; j < 9; j++) { lam_a[j] = 0; lam_lo[j] = 0; lam_hi[j] = 0; }
#pragma unroll
    for (int g = 9; g < NMOT; g++) lamr[g] = 0;
    lam = lam0;
    {
      real v = 0;                          // warm start: J du of the cached normal multipliers, z = lambda + rhs - dinv (J du)
#pragma unroll
      for (int c = 0; c < MERGED_CONTACTS; c++)
        if (c < nc) v += Bn[NMOT + 3 * c] * sh.r_lam[3 * c];
      z = rhs + v;                         // (Bn = [own] - dinv A: the sum is  [own normal row] lam0 - dinv (J du))
    }
    real amax = 0;                           // largest |multiplier| any arm motor row 0..6 took during the solve (limit rows left out)
    auto iterate = [&](auto CHECKTAG, auto WELDTAG) __attribute__((always_inline)) -> bool {
      constexpr bool CHECK = decltype(CHECKTAG)::value;     // evaluate the early-exit test in this iteration? (see pgs_iteration_loop)
      constexpr bool WELD = decltype(WELDTAG)::value;       // attach / weld rows present (scripted mode): see pgs_rows2
      unsigned long long busy = 0;
      __asm__ volatile("" ::: "memory");      // keep the arm row constants in LDS (see pgs(): LICM would hoist and spill them)
      constexpr int PF = 4;
      real4 pa4[PF], pl4[PF];
#pragma unroll
      for (int k = 0; k < PF; k++) { pa4[k] = *reinterpret_cast<const real4*>(sh.mrec[k]); pl4[k] = *reinterpret_cast<const real4*>(sh.lrec[k]); }
      // the arm rows and the pipe motor rows do not see each other (A[arm row][pipe motor row] = 0): two accumulators, two chains
      real za = z, zp = z;
      // The 9 arm joint blocks are spread evenly over the 23 pipe motor rows (block k in front of row 23 k / 9): a pipe row is
      // med3, sub, <1 wait state>, v_readlane, <2 wait states>, fmac, and the instructions of an arm block are what fills those slots
      // (with all arm blocks up front the last 14 pipe rows ran with three s_nop slots each).  The two chains commute exactly.
#pragma unroll
      for (int jp = 0; jp < PIH_OBJ_NJ; jp++) {
        const int j = (jp * 9 + PIH_OBJ_NJ - 1) / PIH_OBJ_NJ;         // the arm block in front of pipe row jp, if 23 j / 9 == jp
        if (j < 9 && (PIH_OBJ_NJ * j) / 9 == jp) {   // arm joint block: motor, lower limit, upper limit; y = dinv (J du) of the joint
          const real4 ca = pa4[j % PF], cl = pl4[j % PF];
          if (j + PF < 9) { pa4[j % PF] = *reinterpret_cast<const real4*>(sh.mrec[j + PF]); pl4[j % PF] = *reinterpret_cast<const real4*>(sh.lrec[j + PF]); }
          const real rh = ca.y, th = ca.z, lim = ca.w;
          const real lor = cl.x, hir = cl.y, wd = cl.w;
          real y = -rdlane(za, j);
          real sum = lam_a[j] + (rh - y);
          sum = med3_(sum, -lim, lim);
          const real dl = sum - lam_a[j]; lam_a[j] = sum;
          if (CHECK) busy |= __ballot(absr(dl) > th);
          if (FULL || j >= 7) {
            y += dl * wd;
            real s2 = lam_lo[j] + (lor - y); s2 = max_(s2, (real)0);
            const real d2 = s2 - lam_lo[j]; lam_lo[j] = s2;
            if (CHECK) busy |= __ballot(absr(d2) > th);
            y += d2 * wd;
            real s3 = lam_hi[j] + (hir + y); s3 = max_(s3, (real)0);
            const real d3 = s3 - lam_hi[j]; lam_hi[j] = s3;
            if (CHECK) busy |= __ballot(absr(d3) > th);
            za += Bn[j] * (dl + d2 - d3);
          } else {
            amax = max_(amax, absr(sum));                         // watched in EVERY iteration (one v_max): a multiplier may touch its bound and leave it again
            za += Bn[j] * dl;
          }
        }
        // pipe motor row 9 + jp: every lane clamps its own z, the row's lane supplies the step
        const int g = 9 + jp;
        const real cand = med3_(zp, -ubv, ubv);              // (the motor rows' bounds are symmetric; two vector sources, see gs_row2_normal)
        const real dlv = cand - lamr[g];
        const real sdl = rdlane(dlv, g);
        if (CHECK) busy |= __ballot(absr(dlv) > thr) & (1ull << g);
        lamr[g] += sdl;
        zp += Bn[g] * sdl;
      }
      z = (za + zp) - z;
      // contacts: normal, dir1, dir2 -- the friction bounds follow the normal multiplier; Bullet leaves the friction rows of an
      // unloaded contact alone: a wave-uniform branch (half of the listed contacts are unloaded; collapsing their bounds onto the
      // current multiplier instead -- branch-free, step 0 -- costs the 14 VALU of two rows for nothing: 263 k vs 247 k cycles at 8-10)
      // (nc and angmask through scalars that are opaque at every contact: the loop-invariant exit / friction conditions would otherwise
      //  be precomputed as 64-bit lane masks outside the iteration loop, spilled to VGPR lanes and read back with v_readlane in every
      //  iteration -- or the loop given a run-time trip count.  One copy per iteration, "modified" in place: no instruction per contact)
      int ncl = nc; unsigned am = angmask;
#pragma unroll
      for (int c = 0; c < MERGED_CONTACTS; c++) {
        __asm__ volatile("" : "+s"(ncl), "+s"(am));
        if (c >= ncl) break;
        const int g0 = NMOT + 3 * c;
        real dl, s0;
        gs_row1_normal(z, Bn[g0], lbv, lam, g0, dl, s0);
        if (CHECK) busy |= __ballot(absr(dl) > thr) & (1ull << g0);
        // wave-uniform branch on  s0 > 0 || weld row  (as integers, on the scalar unit: see pgs_rows2)
        const int s0i = __builtin_bit_cast(int, s0), weld = WELD ? (int)((am >> c) & 1u) : 0;
        if (WELD ? (s0i > weld ? s0i : weld) > 0 : s0i > 0) {
          const real hi = max_(cmu * s0, cfl);
          real dl2;
          gs_row1_friction2(z, Bn[g0 + 1], Bn[g0 + 2], hi, lam, g0 + 1, g0 + 2, dl, dl2);
          if (CHECK) busy |= (__ballot(absr(dl) > thr) & (1ull << (g0 + 1))) | (__ballot(absr(dl2) > thr) & (1ull << (g0 + 2)));
        }
      }
      return CHECK && busy == 0;
    };
    if (angmask != 0) it = pgs_iteration_loop(P.iters, P.checkstride, [&]() __attribute__((always_inline)) { return iterate(std::true_type{}, std::true_type{}); }, [&]() __attribute__((always_inline)) { return iterate(std::false_type{}, std::true_type{}); });
    else it = pgs_iteration_loop(P.iters, P.checkstride, [&]() __attribute__((always_inline)) { return iterate(std::true_type{}, std::false_type{}); }, [&]() __attribute__((always_inline)) { return iterate(std::false_type{}, std::false_type{}); });
    return __builtin_amdgcn_readfirstlane(amax >= armlim ? 1 : 0) != 0;   // (armlim = the smallest bound of rows 0..6: conservative if they differ, exact if equal)
  };
  int variant = 1;
  if (!skip7) { solve(std::true_type{}); variant = 2; }
  else if (solve(std::false_type{})) { solve(std::true_type{}); variant = 4; }
  if (lane == 0) sh.S[PIH_S_SOLVER] = (real)variant;
  // multipliers back to LDS: contacts -> r_lam, pipe motors -> word 1 of their (now unused) constant record
  if (lane >= NMOT && lane < NMOT + 3 * nc) sh.r_lam[lane - NMOT] = lam;
#pragma unroll
  for (int g = 9; g < NMOT; g++) if (lane == 0) sh.mrec[g][1] = lamr[g];
  w.sync();
  {
    const int d = lane, dw = d < ND ? d : ND;
    real du = 0;
    if (d < 9) {
#pragma unroll
      for (int j = 0; j < 9; j++) du += wma_row(sh, j)[d] * (lam_a[j] + lam_lo[j] - lam_hi[j]);
    } else if (d < ND) {
#pragma unroll
      for (int j = 0; j < PIH_OBJ_NJ; j++) du += wmp_row(sh, j)[d - 9] * sh.mrec[9 + j][1];
    }
#pragma unroll
    for (int c = 0; c < MERGED_CONTACTS; c++)
      if (c < nc) du += sh.b.Wp[3 * c][dw] * sh.r_lam[3 * c] + sh.b.Wp[3 * c + 1][dw] * sh.r_lam[3 * c + 1] + sh.b.Wp[3 * c + 2][dw] * sh.r_lam[3 * c + 2];
    if (d < ND) sh.u[d] += du;
  }
  w.sync();
  return it;
}

// ---- the same solver with TWO rows per lane, for envs with 11 .. HC contacts (128 rows; a fifth of the env-steps of the benchmark
// workload and every one of its slowest: their DOF-space blocks were what a launch of 4096 envs waited for).  Row g lives in lane
// g % 64 of register set g / 64.  The 128 x 128 matrix no longer fits the register file: the columns of the 32 motor rows (KREG = NMOT)
// stay in registers, those of the contact rows are written once per step to the env's scratch in global memory
// ([column][lane][2], 512 B per column, L2-resident) and streamed through a ring of D prefetched columns in every iteration --
// the addresses do not depend on the solve, so the loads are issued a ring ahead and never sit on the row chain.
// Motor columns come from the symmetry of the Delassus matrix, A[r][motor m] = W_r[dof(m)], without a dot product.
PIH_HD int pgs_rows2(Wave& w, Shared& sh, const Params& P, const Ovf& ov, const MotorW& mw) {
  const int nc = __builtin_amdgcn_readfirstlane(sh.nc);
  w.sync();
  const int lane = w.lane();
  constexpr int D = 12;                                    // ring depth (columns in flight)
  real* const Bg = ov.base + OVF_B_OFF + 2 * lane;         // this lane's slot of streamed column s: Bg[s * 128 .. +1]
  // one streamed column (both rows of the lane) = one 8-byte load at  (uniform column pointer) + lane * 8: global_load_dwordx2 with the
  // pointer in an SGPR pair and the column in the instruction's immediate offset.  The pointer is made opaque once per iteration
  // (otherwise the addresses of all 96 columns are loop invariants: hoisted out of the iteration loop and spilled).  No clamp on the
  // column index: the ring runs up to D columns past the last one written, inside the env's own scratch (the parked motor rows
  // behind the column area and, for the last env of the batch, OVF_PAD_WORDS of slack), and those values are never used.
  static_assert(OVF_MW_WORDS + OVF_PAD_WORDS >= D * 128, "the ring reads D columns past the streamed-column area");
  const unsigned l8 = (unsigned)lane * 8u;
  // ---- rows of this lane: r0 = lane (motor row, or contact row lane - 32), r1 = 64 + lane (contact row 32 + lane)
  struct RowC { real di, rhs, thr, lb, ub, mu, fl, lam0; };
  auto rowconst = [&](int g) __attribute__((always_inline)) -> RowC {
    RowC r; r.di = 0; r.rhs = 0; r.thr = PIH_BIG; r.lb = 0; r.ub = 0; r.mu = 0; r.fl = 0; r.lam0 = 0;
    if (g < 9) r.di = sh.mrec[g][0];
    else if (g < NMOT) { r.di = sh.mrec[g][0]; r.rhs = sh.mrec[g][1]; r.thr = sh.mrec[g][2]; r.ub = sh.mrec[g][3]; r.lb = -r.ub; }
    else {
      const int row = g - NMOT, c = row / 3, k = row - 3 * c;
      if (c < nc) {
        const real* R = c < CL ? sh.b.crec[c] : ov.base + OVF_W_WORDS + (size_t)(c - CL) * CREC;
        r.di = R[11 + 4 * k]; r.rhs = R[20 + k]; r.thr = R[29 + k]; r.mu = R[5]; r.fl = R[4];
        if (k == 0) { r.lb = R[3]; r.ub = PIH_BIG; r.lam0 = sh.r_lam[3 * c]; }
      }
    }
    return r;
  };
  const RowC c0 = rowconst(lane), c1 = rowconst(64 + lane);
  // ---- columns.  Bn[i] = [own row] - dinv A[i]
  pk2 BB[KREG];                                            // (row of register set 0, row of register set 1) per column
  // motor columns by symmetry: A[r][m] = W_r[dof(m)].  Contact rows read their own response row; motor rows take the (lane = DOF)
  // registers of the motor rows: arm lane r holds W_m[r] itself, pipe lane r finds W_m[dof(r)] six lanes up
  {
    const int row0 = lane - NMOT, row1 = 32 + lane;                 // contact-row index of r0 (if >= 0) and r1
    const real* w0 = row0 >= 0 && row0 < 3 * nc ? wp_row(sh, ov, row0) : nullptr;
    const real* w1 = row1 < 3 * nc ? wp_row(sh, ov, row1) : nullptr;
#pragma unroll
    for (int m = 0; m < NMOT; m++) {
      const int dm = m < 9 ? m : 15 + (m - 9);
      real a0;
      if (m < 9) a0 = lane < 9 ? mw.w[m] : (real)0;
      else {
        const real up = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * ((lane + 6) & 63), __builtin_bit_cast(int, mw.w[m - 9])));
        a0 = (lane >= 9 && lane < NMOT) ? up : (real)0;
      }
      if (w0) a0 = w0[dm];
      const real a1 = w1 ? w1[dm] : (real)0;
      // (no pin between the 32 columns: the 64 reads -- LDS, or the scratch in global memory for rows of contacts >= CL -- target the
      //  registers that stay, and all are in flight together instead of 32 serialised round trips)
      BB[m] = pk_pack(((lane == m && m >= 9) ? (real)1 : (real)0) - c0.di * a0, -c1.di * a1);
    }
  }
  // the motor rows' registers are not needed again before the very end: park them in the env's scratch
  real* const Mg = ov.base + OVF_MW_OFF + lane;
#pragma unroll
  for (int j = 0; j < PIH_OBJ_NJ; j++) Mg[j * 64] = mw.w[j];
  // ---- Jacobian rows (registers) of both rows of the lane
  pk2 JJ[ND];                                              // (entry d of the row in register set 0, of the row in set 1)
  {
    auto geom = [&](int g, int& la, int& lb, V3& p, V3& dir, bool& ang, bool& live) __attribute__((always_inline)) {
      la = -1; lb = -1; p = mk(0, 0, 0); dir = mk(0, 0, 0); ang = false; live = false;
      if (g < NMOT) return;
      const int row = g - NMOT, c = row / 3, k = row - 3 * c;
      if (c >= nc) return;
      live = true; la = sh.c_la[c]; lb = sh.c_lb[c];
      const real* R = c < CL ? sh.b.crec[c] : ov.base + OVF_W_WORDS + (size_t)(c - CL) * CREC;
      p = ld3(R); dir = ld3(R + 8 + 4 * k); ang = R[6] != 0;
    };
    int la0, lb0, la1, lb1; V3 p0, d0, p1, d1; bool a0, a1, l0, l1;
    geom(lane, la0, lb0, p0, d0, a0, l0); geom(64 + lane, la1, lb1, p1, d1, a1, l1);
    const int md = lane < 9 ? lane : 15 + (lane - 9);
    DofGeom gn = dof_geom(sh, 0);
#pragma unroll
    for (int d = 0; d < ND; d++) {
      const DofGeom g = gn;
      if (d + 1 < ND) gn = dof_geom(sh, d + 1);
      real j0 = lane < NMOT ? (d == md ? (real)1 : (real)0) : (l0 ? jac_entry(g, la0, lb0, p0, d0, a0) : (real)0);
      real j1 = l1 ? jac_entry(g, la1, lb1, p1, d1, a1) : (real)0;
      __asm__ volatile("" : "+v"(j0), "+v"(j1) :: "memory");
      JJ[d] = pk_pack(j0, j1);
    }
  }
  // contact columns: J . W_i, the response row W_i broadcast from LDS (contacts < CL) or from the env's scratch
  w.sync();
  // Both rows of the lane advance with one v_pk_fma_f32 per response-row entry; the entries are read in pairs (w_d, w_d+1) and the
  // packed FMA picks the half (op_sel), so that the 228 FMAs + 114 reads of a contact become 114 + 57.  Two loops, because the rows of
  // contacts < CL are LDS words and those of contacts >= CL global words: one loop over a generic pointer made every read a flat_load
  // under a branch.  (Rows are 39 words apart: 4-byte aligned pairs, which ds_read2_b32 / global_load_dwordx2 accept.)
  static_assert(ND % 2 == 0 && KREG == NMOT, "entries are consumed in pairs; every contact column is streamed");
  real wv0 = 0, wv1 = 0;                                     // warm start: sum over the contacts of (normal column) x (cached multiplier), see below
  auto column = [&](int cb, auto pairof) __attribute__((always_inline)) {
    pk2 acc[3] = {0, 0, 0};
    // (blocks of 4 DOFs = 6 pairs, requested one block ahead of their use: left alone the scheduler, at the register limit, issues
    //  each read right in front of its FMA -- 57 serialised LDS round trips per contact)
    constexpr int NB = (ND + 3) / 4;
    pk2 wq[2][6];
    auto request = [&](int blk, pk2* q) __attribute__((always_inline)) {
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int k = 0; k < 3; k++) { const int d = 4 * blk + 2 * h; q[3 * h + k] = d < ND ? pairof(k * WPS + d) : (pk2)0; }
    };
    request(0, wq[0]);
#pragma unroll
    for (int blk = 0; blk < NB; blk++) {
      if (blk + 1 < NB) request(blk + 1, wq[(blk + 1) & 1]);
      __asm__ volatile("" ::: "memory");                   // the next block's reads stay in front of this block's FMAs
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int d = 4 * blk + 2 * h;
        if (d < ND) {
#pragma unroll
          for (int k = 0; k < 3; k++) { pk_fma_lo(acc[k], JJ[d], wq[blk & 1][3 * h + k]); pk_fma_hi(acc[k], JJ[d + 1], wq[blk & 1][3 * h + k]); }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int i = NMOT + 3 * cb + k;
      const real b0 = (lane == i ? (real)1 : (real)0) - c0.di * pk_lo(acc[k]), b1 = (64 + lane == i ? (real)1 : (real)0) - c1.di * pk_hi(acc[k]);
      Bg[(size_t)(i - KREG) * 128] = b0; Bg[(size_t)(i - KREG) * 128 + 1] = b1;
      if (k == 0) { const real l = sh.r_lam[3 * cb]; wv0 += b0 * l; wv1 += b1 * l; }   // (while the column is in registers: reading it back cost a round trip per contact)
    }
  };
  {
    const int nl = nc < CL ? nc : CL;
#pragma unroll 1
    for (int cb = 0; cb < nl; cb++) {
      const real* wr = sh.b.Wp[3 * cb];
      column(cb, [&](int o) __attribute__((always_inline)) { return pk_pack(wr[o], wr[o + 1]); });
    }
    typedef const real __attribute__((address_space(1)))* grp;
#pragma unroll 1
    for (int cb = CL; cb < nc; cb++) {
      const grp wr = (grp)(ov.base + (size_t)(3 * (cb - CL)) * WPS);
      column(cb, [&](int o) __attribute__((always_inline)) { return pk_pack(wr[o], wr[o + 1]); });
    }
  }
#pragma unroll
  for (int q = NMOT; q < KREG; q++) if (q >= NMOT + 3 * nc) BB[q] = 0;
  unsigned angmask = 0;
  for (int c = 0; c < nc; c++) { const real* R = c < CL ? sh.b.crec[c] : ov.base + OVF_W_WORDS + (size_t)(c - CL) * CREC; if (R[5] < 0) angmask |= 1u << c; }
  angmask = (unsigned)__builtin_amdgcn_readfirstlane((int)angmask);
  if (lane < 9) sh.lrec[lane][3] = sh.lrec[lane][2] * sh.mrec[lane][0];          // (J W) dinv of the arm joint (1 up to rounding)
  __threadfence_block();                                     // the streamed columns: written above, read back by the same lane
  w.sync();
  // ---- solve (every limit row in place)
  // (the arm joints' three multipliers -- motor, lower, upper limit -- are wave-uniform values in registers, as in pgs_rows: the
  //  register budget is set by pgs_rows, and three v_readlane + three lane commits per joint and iteration were the price of lanes)
  real lam0 = c0.lam0, lam1 = c1.lam0, z0, z1;
  real lam_a[9], lam_lo[9], lam_hi[9];
#pragma unroll
  for (int j = 0; j < 9; j++) { lam_a[j] = 0; lam_lo[j] = 0; lam_hi[j] = 0; }
  z0 = c0.rhs + wv0; z1 = c1.rhs + wv1;                      // warm start: z = lambda + rhs - dinv (J du) (see pgs_rows)
  // WELD: the env has attach / weld rows (scripted mode, states 4-6), whose friction rows run whatever the normal multiplier is; without
  // them (always in action mode) the per-contact test is a scalar compare of s0's bit pattern alone.  (Round 4: one instantiation per
  // case; the combined test had come out as five instructions -- v_and, v_max_i32, v_cmp, s_and, s_cbranch -- per contact.)
  auto iterate = [&](auto CHECKTAG, auto WELDTAG) __attribute__((always_inline)) -> bool {
    constexpr bool CHECK = decltype(CHECKTAG)::value;
    constexpr bool WELD = decltype(WELDTAG)::value;
    unsigned long long busy = 0;
    __asm__ volatile("" ::: "memory");
    // ring of streamed columns: the first D are requested before the motor rows
    typedef const char __attribute__((address_space(1)))* gcp;
    typedef const pk2 __attribute__((address_space(1)))* gpp;
    gcp colp = (gcp)(ov.base + OVF_B_OFF);
    __asm__ volatile("" : "+s"(colp));
    auto ldcol = [&](int sidx) __attribute__((always_inline)) -> pk2 { return *(gpp)(colp + (size_t)sidx * 512 + l8); };
    pk2 rr[D];
#pragma unroll
    for (int s = 0; s < D; s++) rr[s] = ldcol(s);
    constexpr int PF = 2;
    real4 pa4[PF], pl4[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) { pa4[k] = *reinterpret_cast<const real4*>(sh.mrec[k]); pl4[k] = *reinterpret_cast<const real4*>(sh.lrec[k]); }
    real za = z0; pk2 zp1 = pk_pack(z0, z1);               // arm chain; (pipe chain, register set 1)
#pragma unroll
    for (int j = 0; j < PIH_OBJ_NJ; j++) {
      if (j < 9) {
        const real4 ca = pa4[j % PF], cl = pl4[j % PF];
        if (j + PF < 9) { pa4[j % PF] = *reinterpret_cast<const real4*>(sh.mrec[j + PF]); pl4[j % PF] = *reinterpret_cast<const real4*>(sh.lrec[j + PF]); }
        const real rh = ca.y, th = ca.z, lim = ca.w;
        const real lor = cl.x, hir = cl.y, wd = cl.w;
        real y = -rdlane(za, j);
        real sum = lam_a[j] + (rh - y);
        sum = med3_(sum, -lim, lim);
        const real dl = sum - lam_a[j]; lam_a[j] = sum;
        if (CHECK) busy |= __ballot(absr(dl) > th);
        y += dl * wd;
        real s2 = lam_lo[j] + (lor - y); s2 = max_(s2, (real)0);
        const real d2 = s2 - lam_lo[j]; lam_lo[j] = s2;
        if (CHECK) busy |= __ballot(absr(d2) > th);
        y += d2 * wd;
        real s3 = lam_hi[j] + (hir + y); s3 = max_(s3, (real)0);
        const real d3 = s3 - lam_hi[j]; lam_hi[j] = s3;
        if (CHECK) busy |= __ballot(absr(d3) > th);
        const real tot = dl + d2 - d3;
        za += pk_lo(BB[j]) * tot; zp1 = pk_pack(pk_lo(zp1), pk_hi(zp1) + pk_hi(BB[j]) * tot);
      }
      const int g = 9 + j;
      real dlv;
      gs_row2_bounded(zp1, pk_lo(zp1), BB[g], c0.ub, lam0, g, dlv);
      if (CHECK) busy |= __ballot(absr(dlv) > c0.thr) & (1ull << g);
    }
    pk2 zz = pk_pack((za + pk_lo(zp1)) - z0, pk_hi(zp1));
    // one row (gs_row_normal / gs_row_friction): clamp the lane's own z, take the row's step from its lane, move every z
    // (nc and angmask through scalars that are opaque at every contact: the 2 x 32 loop-invariant exit / friction conditions would
    //  otherwise be precomputed as 64-bit lane masks outside the iteration loop and spilled, or the loop given a run-time trip count
    //  and not unrolled.  One copy per iteration, modified "in place" by the empty asm: no instruction per contact)
    int ncl = nc; unsigned am = angmask;
    unsigned l8c = l8;
    gcp colq = colp + D * 512;                               // column 3 c + D: what contact c requests
    __asm__ volatile("" : "+s"(colq));
#pragma unroll
    for (int c = 0; c < HC; c++) {
      __asm__ volatile("" : "+s"(ncl), "+s"(am), "+v"(l8c));   // (l8c: its zero extension must be visible in the contact's own basic block
                                                               //  for the scalar-base + 32-bit-lane-offset form of global_load)
      if (c >= ncl) break;
      const int g0 = NMOT + 3 * c;
      pk2 bb[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int sc = g0 + k - KREG;                        // (KREG = NMOT: every contact column is streamed)
        bb[k] = rr[sc % D];
        rr[sc % D] = *(gpp)(colq + ((c & 1) * 3 + k) * 512 + l8c);   // request the column a ring ahead (immediate offsets 0 .. 2 560)
      }
      if (c & 1) {
        colq += 6 * 512;
        __asm__ volatile("" : "+s"(colq));                 // (the running pointer stays in an SGPR pair: s_add_u32 / s_addc_u32 per PAIR of contacts)
      }
      real dl, s0;
      if (g0 < 64) { gs_row2_normal(zz, pk_lo(zz), bb[0], c0.lb, lam0, g0, dl, s0); if (CHECK) busy |= __ballot(absr(dl) > c0.thr) & (1ull << g0); }
      else { gs_row2_normal(zz, pk_hi(zz), bb[0], c1.lb, lam1, g0 - 64, dl, s0); if (CHECK) busy |= __ballot(absr(dl) > c1.thr) & (1ull << (g0 - 64)); }
      // Bullet leaves the friction rows of an unloaded contact alone: wave-uniform branch on  s0 > 0 || weld row  -- as integers, on
      // the scalar unit (the bit pattern of a positive float is a positive integer)
      const int s0i = __builtin_bit_cast(int, s0), weld = WELD ? (int)((am >> c) & 1u) : 0;
      if (WELD ? (s0i > weld ? s0i : weld) > 0 : s0i > 0) {
        auto friction = [&](int g, pk2 col) __attribute__((always_inline)) {      // (row g lives in register set g / 64)
          if (g < 64) { const real h0 = max_(c0.mu * s0, c0.fl); gs_row2_friction(zz, pk_lo(zz), col, h0, lam0, g, dl); if (CHECK) busy |= __ballot(absr(dl) > c0.thr) & (1ull << g); }
          else { const real h1 = max_(c1.mu * s0, c1.fl); gs_row2_friction(zz, pk_hi(zz), col, h1, lam1, g - 64, dl); if (CHECK) busy |= __ballot(absr(dl) > c1.thr) & (1ull << (g - 64)); }
        };
        friction(g0 + 1, bb[1]);
        friction(g0 + 2, bb[2]);
      }
    }
    z0 = pk_lo(zz); z1 = pk_hi(zz);
    return CHECK && busy == 0;
  };
  int it;
  if (angmask != 0) it = pgs_iteration_loop<false>(P.iters, P.checkstride, [&]() __attribute__((always_inline)) { return iterate(std::true_type{}, std::true_type{}); }, [&]() __attribute__((always_inline)) { return iterate(std::false_type{}, std::true_type{}); });
  else it = pgs_iteration_loop<false>(P.iters, P.checkstride, [&]() __attribute__((always_inline)) { return iterate(std::true_type{}, std::false_type{}); }, [&]() __attribute__((always_inline)) { return iterate(std::false_type{}, std::false_type{}); });
  if (lane == 0) sh.S[PIH_S_SOLVER] = 5;
  // ---- multipliers back to LDS, DOF velocities du = sum_i W_i lambda_i
  if (lane >= NMOT && lane < NMOT + 3 * nc) sh.r_lam[lane - NMOT] = lam0;
  if (32 + lane < 3 * nc) sh.r_lam[32 + lane] = lam1;
  if (lane >= 9 && lane < NMOT) sh.mrec[lane][1] = lam0;
#pragma unroll
  for (int j = 0; j < 9; j++) if (lane == 0) sh.mrec[j][1] = lam_a[j] + lam_lo[j] - lam_hi[j];
  w.sync();
  {
    const int d = lane, dw = d < ND ? d : ND;
    real du = 0;
#pragma unroll
    for (int j = 0; j < PIH_OBJ_NJ; j++) {
      du += Mg[j * 64] * (d < 9 ? (j < 9 ? sh.mrec[j][1] : (real)0) : sh.mrec[9 + j][1]);
    }
    // du += sum_r W_r[d] lambda_r: rows of contacts < CL from LDS, the others from the env's scratch -- two loops with known address
    // spaces, four reads in flight (one loop over the generic row pointer was a serialised round trip per row, flat loads to global
    // memory for the rows of contacts >= CL)
    {
      const int nl = 3 * (nc < CL ? nc : CL);
#pragma unroll 8
      for (int r = 0; r < nl; r++) du += sh.b.Wp[r][dw] * sh.r_lam[r];
      typedef const real __attribute__((address_space(1)))* grp;
      const grp wg = (grp)ov.base;
#pragma unroll 4
      for (int r = 3 * CL; r < 3 * nc; r++) du += wg[(size_t)(r - 3 * CL) * WPS + dw] * sh.r_lam[r];
    }
    if (d < ND) sh.u[d] += du;
  }
  w.sync();
  return it;
}

// Sequential impulse, Bullet resolveSingleConstraintRowGeneric form; row order: per arm joint (motor, lower limit, upper
// limit), the 23 pipe motors, then per contact (normal, dir1, dir2).  Returns iterations executed.
// GPU form: one lane per DOF holds its entry of the velocity change `du`; row multipliers live lane-distributed in
// registers (v_readlane to broadcast); motor response rows are preloaded into registers; the arm and pipe motor chains
// commute (disjoint DOFs) and are interleaved for ILP; each contact is solved as an exact 3x3 Gauss-Seidel block: three
// DPP row-reductions in flight at once, then the cross terms G bring dir1/dir2 up to date without touching `du`.
PIH_HD int pgs(Wave& w, Shared& sh, const Params& P, const Ovf& ov, const MotorW& mw) {
  const int nc = sh.nc;
  if (P.pgsmode == 0 && nc <= MERGED_CONTACTS) return pgs_rows(w, sh, P);   // all rows fit one wavefront: row-space solver
  if (P.pgsmode == 0 && nc <= HC) return pgs_rows2(w, sh, P, ov, mw);       // two rows per lane
  if (w.lane() == 0) sh.S[PIH_S_SOLVER] = 0;
  // early exit test without divisions: (dl / dinv)^2 <= resid  <=>  dl^2 - resid dinv^2 <= 0  for every row
  w.sync();
  const int d = w.lane();
  const DofGeom g = dof_geom(sh, d);
  const bool armlane = d < 9;
  const int dw = d < ND ? d : ND;   // response-row word of this lane (idle lanes read the zero pad)
  // motor / limit multipliers: wave-uniform values held in VGPRs (no readlane, no conditional write-back);
  // contact multipliers: lane-distributed, contact c in lane c
  real lam_p[PIH_OBJ_NJ], lam_a[9], lam_lo[9], lam_hi[9];
#pragma unroll
  for (int j = 0; j < PIH_OBJ_NJ; j++) lam_p[j] = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) { lam_a[j] = 0; lam_lo[j] = 0; lam_hi[j] = 0; }
  // per-lane sign of every contact's Jacobian column, 2 bits per contact (two's complement: 00 = 0, 01 = +1, 11 = -1, read
  // back with one v_bfe_i32): +1 if this lane's joint is an ancestor of linkA, -1 of linkB, 0 of both or neither
  unsigned sg0 = 0, sg1 = 0, sg2 = 0;
  // Jacobian column of this lane for a point p: cross(ae, p - g.o) + mp  (revolute-like: ae = axis, mp = 0; prismatic-like:
  // ae = 0, mp = axis; unused lane: both 0) -- no per-contact select
  const V3 ae = g.kind == 0 ? g.a : mk(0, 0, 0), mp = g.kind == 1 ? g.a : mk(0, 0, 0);
  real du = 0;
  int ang_c = -1;
  for (int c = 0; c < nc; c++) {
    int la = sh.c_la[c], lb = sh.c_lb[c];
    if (sh.c_mu[c] < (real)-1.5) ang_c = c;
    int sgn = g.kind != 2 ? (int)is_anc(g.L, la) - (int)is_anc(g.L, lb) : 0;
    unsigned code = (unsigned)sgn & 3u;
    if (c < 16) sg0 |= code << (2 * c); else if (c < 32) sg1 |= code << (2 * (c - 16)); else sg2 |= code << (2 * (c - 32));
    real l = sh.r_lam[3 * c];   // warm start (uniform LDS read)
    if (l != 0) {
      du += (c < CL ? sh.b.Wp[3 * c][dw] : ov.base[(size_t)(3 * (c - CL)) * WPS + dw]) * l;
    }
  }
  const int ang_cs = __builtin_amdgcn_readfirstlane(ang_c);
  // one PGS iteration; returns true when every row moved by less than its threshold
  auto iterate = [&](auto CHECKTAG) __attribute__((always_inline)) -> bool {
    constexpr bool CHECK = decltype(CHECKTAG)::value;     // evaluate the early-exit test in this iteration? (see pgs_iteration_loop)
    // Early exit (Bullet's least-squares residual test, max over rows of (d lambda / dinv)^2 <= resid) as |d lambda| >
    // sqrt(resid) dinv per row: one v_cmp into a wave mask + a scalar OR per row instead of an FMA and a max.  Bit 32 is read:
    // the motor chain is wave-uniform and the contact chain is valid in lanes 32..47.
    unsigned long long busy = 0;
    // the row constants are re-read from LDS every iteration ON PURPOSE: without this compiler barrier LICM hoists all
    // ~155 loop-invariant loads out of the iteration loop and spills them to scratch inside the hot loop
    __asm__ volatile("" ::: "memory");
    // row constants come from LDS as 16-byte broadcasts, explicitly prefetched PF records ahead: the dependent chain of one
    // motor step is ~6 VALU ops (~50 cycles) while an LDS round trip is >100, so a distance-1 prefetch stalls every step
    constexpr int PF = 6;
    real4 pm[PF], pa4[PF], pl4[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) { pm[k] = *reinterpret_cast<const real4*>(sh.mrec[9 + k]); pa4[k] = *reinterpret_cast<const real4*>(sh.mrec[k]); pl4[k] = *reinterpret_cast<const real4*>(sh.lrec[k]); }
#pragma unroll
    for (int j = 0; j < PIH_OBJ_NJ; j++) {
      real tot_a = 0;
      const real4 cm = pm[j % PF];
      if (j + PF < PIH_OBJ_NJ) pm[j % PF] = *reinterpret_cast<const real4*>(sh.mrec[9 + j + PF]);
      if (j < 9) {   // arm joint block: motor, lower limit, upper limit (wave-uniform chain)
        const real4 ca = pa4[j % PF], cl = pl4[j % PF];
        if (j + PF < 9) { pa4[j % PF] = *reinterpret_cast<const real4*>(sh.mrec[j + PF]); pl4[j % PF] = *reinterpret_cast<const real4*>(sh.lrec[j + PF]); }
        const real di = ca.x, rhs = ca.y, thr = ca.z, lim = ca.w;
        const real lor = cl.x, hir = cl.y, wjj = cl.z;
        real dj = rdlane(du, j);
        real sum = lam_a[j] + (rhs - dj * di);
        sum = med3_(sum, -lim, lim);
        real dl = sum - lam_a[j]; lam_a[j] = sum;
        if (CHECK) busy |= __ballot(absr(dl) > thr);
        dj += dl * wjj;
        real s2 = lam_lo[j] + (lor - dj * di); s2 = max_(s2, (real)0);
        real d2 = s2 - lam_lo[j]; lam_lo[j] = s2;
        if (CHECK) busy |= __ballot(absr(d2) > thr);
        dj += d2 * wjj;
        real s3 = lam_hi[j] + (hir + dj * di); s3 = max_(s3, (real)0);
        real d3 = s3 - lam_hi[j]; lam_hi[j] = s3;
        if (CHECK) busy |= __ballot(absr(d3) > thr);
        tot_a = dl + d2 - d3;
      }
      // pipe joint motor j (DOF 15 + j)
      const real di = cm.x, rhs = cm.y, thr = cm.z, lim = cm.w;
      real dj = rdlane(du, 15 + j);
      real sum = lam_p[j] + (rhs - dj * di);
      sum = med3_(sum, -lim, lim);
      real dl = sum - lam_p[j]; lam_p[j] = sum;
      if (CHECK) busy |= __ballot(absr(dl) > thr);
      du += mw.w[j] * (armlane ? tot_a : dl);
    }
    // one exact 3x3 Gauss-Seidel block per contact.  The body is instantiated twice so that the LDS-resident contacts
    // (c < CL) compile to ds_read with immediate offsets and only the rare spilled ones (c >= CL) use global loads; a
    // single loop over "LDS or global" pointers degrades every access to flat_load + vmcnt(0)/lgkmcnt(0) waits.
    // Whole record (8 x 16 B) + the three response-row entries of this lane are fetched in ONE batch, and for the
    // LDS-resident contacts the next contact's batch is issued before the current block computes (software pipelining):
    // piecemeal loads cost four serial LDS round trips per contact, which two waves per SIMD cannot hide.
    struct CRec { real4 q[8]; real w0, w1, w2; };
    auto fetch = [&](const real* R, const real* wr) __attribute__((always_inline)) -> CRec {
      CRec r;
#pragma unroll
      for (int i = 0; i < 8; i++) r.q[i] = reinterpret_cast<const real4*>(R)[i];
      r.w0 = wr[dw]; r.w1 = wr[WPS + dw]; r.w2 = wr[2 * WPS + dw];
      return r;
    };
    auto block = [&](int c, unsigned sgw, const CRec& r, real* R, bool in_lds) __attribute__((always_inline)) {
      // (the angular rows of the attach weld -- at most one such contact, scripted mode only, index ang_c -- take the DOF's axis
      //  itself as Jacobian column instead of axis x lever arm: one scalar compare per contact)
      // q0 = p.xyz, lo_n | q1 = hi_floor, mu, -, - | q2 = n, dinv_n | q3 = t1, dinv_t1 | q4 = t2, dinv_t2
      // q5 = rhs n,t1,t2, G[t1][n] | q6 = G[t2][n], G[t2][t1], lam_n, lam_t1 | q7 = lam_t2, ...
      // (lo_n = 0 / -BIG and hi_floor = 0 / +BIG make the attach rows bilateral without a select)
      const real w0 = r.w0, w1 = r.w1, w2 = r.w2;
      const V3 pr = mk(r.q[0].x - g.o.x, r.q[0].y - g.o.y, r.q[0].z - g.o.z);
      const real mu = r.q[1].y;
      const real sdu = (real)(int)__builtin_amdgcn_sbfe(sgw, 2u * (unsigned)(c & 15), 2u) * du;
      V3 cv = mk(__builtin_fmaf(ae.y, pr.z, __builtin_fmaf(-ae.z, pr.y, mp.x)), __builtin_fmaf(ae.z, pr.x, __builtin_fmaf(-ae.x, pr.z, mp.y)), __builtin_fmaf(ae.x, pr.y, __builtin_fmaf(-ae.y, pr.x, mp.z)));
      if (__builtin_expect(c == ang_cs, 0)) cv = ae;
      real jd0 = sdu * dot(mk(r.q[2].x, r.q[2].y, r.q[2].z), cv), jd1 = sdu * dot(mk(r.q[3].x, r.q[3].y, r.q[3].z), cv), jd2 = sdu * dot(mk(r.q[4].x, r.q[4].y, r.q[4].z), cv);
      // materialise the products: otherwise fast-math folds the multiply into the first reduction step as mul + mov_dpp + fmac
      // (3 instructions per value) instead of mul + v_add_f32_dpp (2)
      __asm__ volatile("" : "+v"(jd0), "+v"(jd1), "+v"(jd2));
      row16_sum3(jd0, jd1, jd2);
      rows012_total3(jd0, jd1, jd2);          // valid in lanes 32..47 from here; the scalar chain below runs in plain VGPRs
      const real l0 = r.q[6].z, l1 = r.q[6].w, l2 = r.q[7].x;
      const real di0 = r.q[2].w, di1 = r.q[3].w, di2 = r.q[4].w;
      real s0 = l0 + (r.q[5].x - jd0 * di0);
      s0 = max_(s0, r.q[0].w);
      real dl0 = s0 - l0;
      if (CHECK) busy |= __ballot(absr(dl0) > r.q[7].y);
      real dl1 = 0, dl2 = 0, s1 = l1, s2 = l2;
      if (rdlane(s0, 32) > 0 || rdlane(mu, 32) < 0) {   // wave-uniform branch (Bullet skips the friction rows of an unloaded contact)
        real hi = max_(mu * s0, r.q[1].x);
        jd1 += r.q[5].w * dl0;
        s1 = l1 + (r.q[5].y - jd1 * di1); s1 = med3_(s1, -hi, hi); dl1 = s1 - l1;
        if (CHECK) busy |= __ballot(absr(dl1) > r.q[7].z);
        jd2 += r.q[6].x * dl0 + r.q[6].y * dl1;
        s2 = l2 + (r.q[5].z - jd2 * di2); s2 = med3_(s2, -hi, hi); dl2 = s2 - l2;
        if (CHECK) busy |= __ballot(absr(dl2) > r.q[7].w);
      }
      if (d == 32) { R[26] = s0; R[27] = s1; R[28] = s2; }
      if (!in_lds) __threadfence_block();     // spilled records live in global memory: make lane 32's store visible to the wave
      du += w0 * rdlane(dl0, 32) + w1 * rdlane(dl1, 32) + w2 * rdlane(dl2, 32);
    };
    const int ncl = nc < CL ? nc : CL;
    if (ncl > 0) {
      // two-deep ping-pong (ra / rb) instead of "cur = nxt": the rotation of a 27-register record costs 27 v_mov per contact
      CRec ra = fetch(sh.b.crec[0], &sh.b.Wp[0][0]);
      int c = 0;
      for (;;) {
        const int c1 = c + 1 < ncl ? c + 1 : c;
        CRec rb = fetch(sh.b.crec[c1], &sh.b.Wp[3 * c1][0]);
        block(c, c < 16 ? sg0 : sg1, ra, sh.b.crec[c], true);
        if (++c >= ncl) break;
        const int c2 = c + 1 < ncl ? c + 1 : c;
        ra = fetch(sh.b.crec[c2], &sh.b.Wp[3 * c2][0]);
        block(c, c < 16 ? sg0 : sg1, rb, sh.b.crec[c], true);
        if (++c >= ncl) break;
      }
    }
    if (nc > CL) {
      // the spilled contacts (global scratch) with the same one-ahead ping-pong: an env that gets here is one of the heaviest of
      // the launch, i.e. the one the launch ends up waiting for, and an unprefetched global load per contact is its latency
      auto rec_of = [&](int c) { return ov.base + OVF_W_WORDS + (size_t)(c - CL) * CREC; };
      auto row_of = [&](int c) { return ov.base + (size_t)(3 * (c - CL)) * WPS; };
      CRec ra = fetch(rec_of(CL), row_of(CL));
      int c = CL;
      for (;;) {
        const int c1 = c + 1 < nc ? c + 1 : c;
        CRec rb = fetch(rec_of(c1), row_of(c1));
        block(c, c < 32 ? sg1 : sg2, ra, rec_of(c), false);
        if (++c >= nc) break;
        const int c2 = c + 1 < nc ? c + 1 : c;
        ra = fetch(rec_of(c2), row_of(c2));
        block(c, c < 32 ? sg1 : sg2, rb, rec_of(c), false);
        if (++c >= nc) break;
      }
    }
    return CHECK && !((busy >> 32) & 1ull);
  };
  const int it = pgs_iteration_loop(P.iters, P.checkstride, [&]() __attribute__((always_inline)) { return iterate(std::true_type{}); }, [&]() __attribute__((always_inline)) { return iterate(std::false_type{}); });
  w.sync();
  if (d < nc) { const real* R = d < CL ? sh.b.crec[d] : ov.base + OVF_W_WORDS + (size_t)(d - CL) * CREC; sh.r_lam[3 * d] = R[26]; sh.r_lam[3 * d + 1] = R[27]; sh.r_lam[3 * d + 2] = R[28]; }
  if (d < ND) sh.u[d] += du;
  w.sync();
  return it;
}

}  // namespace pih
