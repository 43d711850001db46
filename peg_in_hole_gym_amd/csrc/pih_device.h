// pih_device.h -- the per-env step of the MI355X-native peg-in-hole environment (ONE WAVEFRONT PER ENV), assembled:
//   pih_common.h  model tables, LDS layout, kinematics helpers, IK, reset, controller, collision, impulse responses, row build
//   pih_ikq.h     the controller / IK with one env per quad of lanes (pih_pre_kernel, pih_fly_pre_kernel, pih_ik*)
//   pih_wave.h    the gfx950 wave layer: wave context + the phases written with DPP / ds_bpermute / v_readlane (FK and velocity
//                 scans, ABA inward sweep, PGS)
//   pih_step.h    articulated-body forward dynamics and the step itself
// gfx950 only: there is no host path in the product (libpih_hip.so); the test harness in tests/emul brings its own wave layer.
#pragma once
#include "pih_common.h"
#include "pih_ikq.h"
#include "pih_wave.h"
#include "pih_step.h"
