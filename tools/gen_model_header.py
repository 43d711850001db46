#!/usr/bin/env python3
"""Emit include/pih_model.h: per-env world model constants (DATA ONLY, no algorithm).

Sources (reference paths relative to /root/reference/peg_in_hole_gym/):
  * pipe  : envs/assets/urdf/pipe.urdf  (25 links, joint origins :39-43,:76-83..., masses :16,:54,
            lateral_friction :10,:48,:765,:802; globalScaling 0.01 at envs/peg_in_hole.py:242)
  * hole  : envs/assets/urdf/hole.urdf:17-21 + obj/cylinder_base.obj (r_in .96, r_out 1.2, half-h 1;
            globalScaling 0.016 envs/peg_in_hole.py:251; pose envs/peg_in_hole.py:248-250)
  * panda : pybullet_data/franka_panda/panda.urdf -- ABSENT from the container; kinematics are the
            published Franka modified-DH chain, inertials PROVISIONAL.
  * table : pybullet_data/table/table.urdf -- ABSENT; modelled as the half-space z <= -0.05
            (envs/peg_in_hole.py:235, envs/utils.py:24-28).
  * ur5   : envs/assets/urdf/ur5.urdf:32-218,534-539 (exact).

  * free bodies: envs/assets/urdf/banana.urdf:1-32 + obj/banana_collision.obj (5 convex hulls) and
            envs/assets/urdf/Amicelli_800_tex.urdf:1-33 + obj/Amicelli_800_tex.obj (one closed mesh) -- the free-flying objects
            of the 'random-fly' task, selected by args[0] (README.md:38: args=['Banana', 1/120.]).
  * charge_board: envs/assets/urdf/charge_board.urdf:1-40 (fixed base + one hinged door, primitive cylinder) is READ by
            tools/urdf_tables.hinged_body_tables (tested) but no longer emitted: nothing of the reference loads it and no kernel
            consumed the PIH_DOOR_* tables of round 3.

Everything that exists in the reference tree is READ from it (tools/urdf_tables.py: URDF via xml.etree, fixed-joint merge,
globalScaling, OBJ / binary-STL collision meshes for the AABB box-inertia rule); only the Panda and the table, whose assets
live in the absent pybullet_data package, are a hand-entered block (marked PROVISIONAL).  tests/test_model_tables.py
regenerates the header where /root/reference exists and requires it to be byte-identical to the committed one.

The header is plain C initialiser macros so that both the fp64 oracle (oracle/) and the HIP
product (peg_in_hole_gym_amd/csrc) instantiate the same numbers in their own storage classes.
Run:  python tools/gen_model_header.py [--ref /root/reference] > include/pih_model.h
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import urdf_tables as UT  # noqa: E402

REF = sys.argv[sys.argv.index("--ref") + 1] if "--ref" in sys.argv else "/root/reference"
PIPE = UT.pipe_tables(REF)
HOLE = UT.hole_tables(REF)
UR5 = UT.ur5_tables(REF)
OBJECTS = [UT.free_body_tables(REF, f) for f in UT.FLY_OBJECT_FILES]
MARGIN = 0.001          # pybullet's default collision margin for URDF meshes [UNVERIFIED, SURVEY.md App. C]
DEFAULT_MU = 0.5        # pybullet's default lateral friction for links without a <contact> block

np.set_printoptions(precision=17)


def Rx(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def Ry(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def Rz(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def rpy(r, p, y):
    return Rz(y) @ Ry(p) @ Rx(r)


def snap(M):
    M = np.array(M, dtype=float)
    M[np.abs(M) < 1e-15] = 0.0
    for v in (1.0, -1.0):
        M[np.abs(M - v) < 1e-15] = v
    return M


def box_inertia(m, lx, ly, lz):
    return np.diag([m / 12 * (ly * ly + lz * lz), m / 12 * (lx * lx + lz * lz), m / 12 * (lx * lx + ly * ly)])


def merge(parts):
    """parts: list of (m, com[3], I[3x3] about com, axes = target frame). Returns composite."""
    M = sum(p[0] for p in parts)
    c = sum(p[0] * np.asarray(p[1]) for p in parts) / M
    I = np.zeros((3, 3))
    for m, cc, Ic in parts:
        d = np.asarray(cc) - c
        I += Ic + m * (d @ d * np.eye(3) - np.outer(d, d))
    return M, c, I


def sym6(I):
    return [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]


def fmt(x):
    x = float(x)
    r = round(x, 12)
    return repr((r if abs(r - x) < 4e-16 else x) + 0.0)     # parsed-and-scaled decimals such as 3 * 0.01 + 5.5 * 0.01 print as 0.085


def arr(xs):
    return "{" + ", ".join(fmt(x) for x in xs) + "}"


def iarr(xs):
    return "{" + ", ".join(str(int(x)) for x in xs) + "}"


REV, PRI, FLT = 0, 1, 2
H = math.pi / 2

links = []  # dicts


def add(name, parent, jtype, R, t, axis, mass, com, I, lo, hi, limited, damping, mu):
    links.append(dict(name=name, parent=parent, jtype=jtype, R=snap(R), t=np.array(t, float), axis=np.array(axis, float),
                      mass=mass, com=np.array(com, float), I=np.array(I, float), lo=lo, hi=hi, limited=limited,
                      damping=damping, mu=mu))


# ----------------------------------------------------------------------------- Panda (arm links 0..8)
# SURVEY.md App. D [UNVERIFIED recollection of pybullet_data panda.urdf]; kinematics = Franka MDH.
iso = lambda m: np.eye(3) * (0.004 * m)  # PROVISIONAL isotropic inertia (0.4 m r^2, r = 0.1 m)
add("panda_link1", -1, REV, np.eye(3), (0, 0, 0.333), (0, 0, 1), 2.7, (0, -0.04, -0.05), iso(2.7), -2.9671, 2.9671, 1, 0.0, 0.5)
add("panda_link2", 0, REV, Rx(-H), (0, 0, 0), (0, 0, 1), 2.73, (0, -0.04, 0.06), iso(2.73), -1.8326, 1.8326, 1, 0.0, 0.5)
add("panda_link3", 1, REV, Rx(H), (0, -0.316, 0), (0, 0, 1), 2.04, (0.01, 0.01, -0.05), iso(2.04), -2.9671, 2.9671, 1, 0.0, 0.5)
add("panda_link4", 2, REV, Rx(H), (0.0825, 0, 0), (0, 0, 1), 2.08, (-0.03, 0.03, 0.02), iso(2.08), -3.1416, 0.0, 1, 0.0, 0.5)
add("panda_link5", 3, REV, Rx(-H), (-0.0825, 0.384, 0), (0, 0, 1), 3.0, (0, 0.04, -0.12), iso(3.0), -2.9671, 2.9671, 1, 0.0, 0.5)
add("panda_link6", 4, REV, Rx(H), (0, 0, 0), (0, 0, 1), 1.3, (0.04, 0, 0), iso(1.3), -0.0873, 3.8223, 1, 0.0, 0.5)
# link7 + link8 (massless, fixed z 0.107) + hand (fixed rpy 0 0 -pi/4), merged into one rigid link
Rh = Rz(-math.pi / 4)
hand_I = Rh @ box_inertia(0.81, 0.04, 0.2, 0.07) @ Rh.T
m7, c7, I7 = merge([(0.2, (0, 0, 0.08), iso(0.2)), (0.81, (0, 0, 0.107 + 0.04), hand_I)])
add("panda_link7+hand", 5, REV, Rx(H), (0.088, 0, 0), (0, 0, 1), m7, c7, I7, -2.9671, 2.9671, 1, 0.0, 0.5)
# fingers: prismatic, joint frame = link7 frame * T(0,0,.107) Rz(-pi/4) T(0,0,.0584)
tf = (0, 0, 0.107 + 0.0584)
fb = box_inertia(0.1, 0.021, 0.02, 0.049)
add("panda_leftfinger", 6, PRI, Rh, tf, (0, 1, 0), 0.1, (0, 0.010, 0.0295), fb, 0.0, 0.04, 1, 0.0, 1.0)
add("panda_rightfinger", 6, PRI, Rh, tf, (0, -1, 0), 0.1, (0, -0.010, 0.0295), fb, 0.0, 0.04, 1, 0.0, 1.0)
ARM_NL = len(links)
# EE frame ("panda_grasptarget", pybullet link index 11): fixed on link7: T(0,0,.107) Rz(-pi/4) T(0,0,.105)
EE_PARENT = 6
EE_R = snap(Rh)
EE_T = (0, 0, 0.107 + 0.105)

# ----------------------------------------------------------------------------- pipe (object links 0..23), READ from pipe.urdf
S = 0.01  # globalScaling envs/peg_in_hole.py:242 (already applied by urdf_tables.pipe_tables)
# PyBullet (no URDF_USE_INERTIA_FROM_FILE) recomputes link inertia from the collision AABB [UNVERIFIED,
# SURVEY.md App. C]: cylinder r=1, len 6 scaled + 1 mm margin -> box 0.022 x 0.062 x 0.022
pipe_ext = PIPE["aabb_ext"]
pipe_box = lambda m: box_inertia(m, *pipe_ext)
mu_of = lambda f: DEFAULT_MU if f is None else f
jx = PIPE["joint_xyz"]              # 24 joints: [0] fixed link0->link1, [1..23] continuous
# root = pipe_link0 + pipe_link1 (fixed joint): one rigid link
m0, c0, I0 = merge([(PIPE["mass"][0], PIPE["com"][0], pipe_box(PIPE["mass"][0])),
                    (PIPE["mass"][1], np.array(jx[0]) + np.array(PIPE["com"][1]), pipe_box(PIPE["mass"][1]))])
add("pipe_link0+1", -1, FLT, np.eye(3), (0, 0, 0), (0, 0, 0), m0, c0, I0, 0, 0, 0, 0.0, max(mu_of(PIPE["friction"][0]), mu_of(PIPE["friction"][1])))
for k in range(2, 25):  # pipe_link2..24 ; URDF joint index k-1
    j = k - 1  # object link index
    t = np.array(jx[k - 1]) + (np.array(jx[0]) if k == 2 else 0)   # link2's joint sits on link1, which sits on link0 at jx[0]
    add("pipe_link%d" % k, ARM_NL + j - 1, REV, np.eye(3), t, PIPE["joint_axis"][k - 1], PIPE["mass"][k], PIPE["com"][k],
        pipe_box(PIPE["mass"][k]), 0, 0, 0, 0.0, mu_of(PIPE["friction"][k]))
NL = len(links)
OBJ_NL = NL - ARM_NL

# collision rope: vertices V0..V24 (sphere r = 1 cm) ; segment s on object link s from V_s to V_{s+1}.  The mesh cylinder of
# every link (collision origin y = 0.03, half length 0.03, radius 0.01 -- from the OBJ's AABB) starts at the link origin.
PIPE_R = 0.5 * (pipe_ext[0] - 2 * MARGIN)
cyl_half = 0.5 * (pipe_ext[1] - 2 * MARGIN)
col_y = PIPE["collision_xyz"][0][1]
y_first = col_y - cyl_half + PIPE_R                      # first vertex: one radius inside the first cylinder's start
y_last = col_y + cyl_half - PIPE_R                       # last vertex: one radius inside the last cylinder's end
step_y = jx[1][1]                                        # joint spacing along the chain
samples = []  # (obj_link, local_y, is_vertex)
seg_len = [jx[0][1] + step_y - y_first] + [step_y] * 22 + [y_last]
seg_y0 = [y_first] + [0.0] * 23
for s_ in range(24):
    n_int = 6 if s_ == 0 else 4
    samples.append((s_, seg_y0[s_], 1))
    for i_ in range(1, n_int + 1):
        samples.append((s_, seg_y0[s_] + seg_len[s_] * i_ / (n_int + 1), 0))
samples.append((23, seg_y0[23] + seg_len[23], 1))

print("/* GENERATED by tools/gen_model_header.py -- do not edit.  DATA ONLY (model constants).")
print(" * Panda+pipe+hole+table world model of envs/peg_in_hole.py:227-251 (see generator docstring for")
print(" * the per-number reference citations; Panda/table values are UNVERIFIED recollection, SURVEY.md App. D). */")
print("#ifndef PIH_MODEL_H\n#define PIH_MODEL_H")
print("#define PIH_JT_REVOLUTE 0\n#define PIH_JT_PRISMATIC 1\n#define PIH_JT_FLOATING 2")
print("#define PIH_ARM_NL %d   /* arm links (fixed links merged) */" % ARM_NL)
print("#define PIH_OBJ_NL %d  /* pipe links: root(link0+1) + link2..24 */" % OBJ_NL)
print("#define PIH_NL %d" % NL)
print("#define PIH_ARM_NDOF 9\n#define PIH_OBJ_NJ 23\n#define PIH_NDOF 38  /* 9 + 6 + 23 */")
print("#define PIH_LINK_PARENT " + iarr(l["parent"] for l in links))
print("#define PIH_LINK_JTYPE " + iarr(l["jtype"] for l in links))
print("#define PIH_LINK_RFIX {" + ", ".join(arr(l["R"].reshape(-1)) for l in links) + "}")
print("#define PIH_LINK_TFIX {" + ", ".join(arr(l["t"]) for l in links) + "}")
print("#define PIH_LINK_AXIS {" + ", ".join(arr(l["axis"]) for l in links) + "}")
print("#define PIH_LINK_MASS " + arr(l["mass"] for l in links))
print("#define PIH_LINK_COM {" + ", ".join(arr(l["com"]) for l in links) + "}")
print("#define PIH_LINK_INERTIA {" + ", ".join(arr(sym6(l["I"])) for l in links) + "}  /* xx yy zz xy xz yz about COM, link axes */")
print("#define PIH_LINK_LO " + arr(l["lo"] for l in links))
print("#define PIH_LINK_HI " + arr(l["hi"] for l in links))
print("#define PIH_LINK_LIMITED " + iarr(l["limited"] for l in links))
print("#define PIH_LINK_DAMPING " + arr(l["damping"] for l in links))
print("#define PIH_LINK_MU " + arr(l["mu"] for l in links))
print("/* world -> panda base: yaw -pi/2 (envs/utils.py:33) */")
print("#define PIH_ARM_BASE_R " + arr(snap(Rz(-H)).reshape(-1)))
print("#define PIH_EE_PARENT %d  /* grasptarget (pybullet link 11, envs/peg_in_hole.py:20) rides on merged link7 */" % EE_PARENT)
print("#define PIH_EE_R " + arr(EE_R.reshape(-1)))
print("#define PIH_EE_T " + arr(EE_T))
print("#define PIH_ARM_REST {0.0, -0.215, %s, -2.57, 0.0, 2.356, 2.356, 0.0, 0.0}  /* envs/peg_in_hole.py:233; fingers stay 0 (envs/utils.py:35-36) */" % fmt(-math.pi / 3))
print("/* finger pad boxes in finger-link frame (PROVISIONAL stand-in for finger.obj hull) */")
print("#define PIH_FINGER_BOX_C {{0.0, 0.010, 0.0295}, {0.0, -0.010, 0.0295}}")
print("#define PIH_FINGER_BOX_H {0.0105, 0.010, 0.0245}")
print("#define PIH_FINGER_LINK0 7")
# arm-vs-table collision spheres (PROVISIONAL stand-in for the Panda collision meshes of pybullet_data, absent): finger tips
# (the far end of the pad boxes), three hand spheres (hand frame = link7 frame * T(0,0,.107) Rz(-pi/4)), flange, and the
# origins of panda_link4/5/6.  link index, centre in the link frame, radius
hs = lambda y, z: tuple(float(v) for v in (np.array([0, 0, 0.107]) + Rh @ np.array([0, y, z])))
ARM_SPH = [(7, (0.0, 0.010, 0.0435), 0.0105), (8, (0.0, -0.010, 0.0435), 0.0105),
           (6, hs(0.0, 0.03), 0.035), (6, hs(0.065, 0.03), 0.035), (6, hs(-0.065, 0.03), 0.035), (6, (0.0, 0.0, 0.05), 0.045),
           (5, (0.0, 0.0, 0.0), 0.055), (4, (0.0, 0.0, 0.0), 0.06), (3, (0.0, 0.0, 0.0), 0.06)]
print("/* arm-vs-table collision spheres (PROVISIONAL: Panda collision meshes absent) */")
print("#define PIH_ARM_NSPH %d" % len(ARM_SPH))
print("#define PIH_ARM_SPH_LINK " + iarr(a[0] for a in ARM_SPH))
print("#define PIH_ARM_SPH_C {" + ", ".join(arr(a[1]) for a in ARM_SPH) + "}")
print("#define PIH_ARM_SPH_R " + arr(a[2] for a in ARM_SPH))
print("#define PIH_ARM_PIPE_SPH0 2   /* spheres 2.. (hand x3, flange, link 6/5/4 origins) also collide with the pipe; 0, 1 = finger tips (pad boxes do that) */")
print("/* pipe collision rope */")
print("#define PIH_PIPE_RADIUS %s" % fmt(PIPE_R))
print("#define PIH_PIPE_NSAMP %d" % len(samples))
print("#define PIH_PIPE_SAMP_LINK " + iarr(s[0] for s in samples) + "  /* object-link index */")
print("#define PIH_PIPE_SAMP_Y " + arr(s[1] for s in samples))
print("#define PIH_PIPE_SAMP_VERTEX " + iarr(s[2] for s in samples))
print("/* static geometry (env-local frame) */")
print("#define PIH_TABLE_Z (-0.05)   /* SURVEY.md App. A [UNVERIFIED]: -1.3 + 2*0.625 */")
print("#define PIH_TABLE_MU 1.0")
print("#define PIH_HOLE_POS {0.5, -0.2, 0.2}   /* envs/peg_in_hole.py:248 */")
print("#define PIH_HOLE_HALFLEN %s  /* cylinder_base.obj half height x globalScaling 0.016 (envs/peg_in_hole.py:251) */" % fmt(round(HOLE["halflen"], 7)))
print("#define PIH_HOLE_RIN %s   /* inner radius x 0.016 */" % fmt(round(HOLE["rin"], 7)))
print("#define PIH_HOLE_ROUT %s  /* outer radius x 0.016 */" % fmt(round(HOLE["rout"], 7)))
print("#define PIH_HOLE_MU 0.5")
print("#define PIH_GRAVITY_Z (-9.8)  /* envs/peg_in_hole.py:230 */")
# scripted FSM clock (envs/peg_in_hole.py:206-212,254,263): number of update_state() calls spent in each state before the
# transition fires, obtained by replaying the reference's own double-precision arithmetic `t += 1/240; if t > dur: ...`
durs = [0.25, 2, 2, 1, 1.5, 1.5, 0.5, 0.25, 0.25, 0.25]
fsm_steps = []
for dur in durs:
    t, k = 0.0, 0
    while True:
        t += 1.0 / 240
        k += 1
        if t > dur:
            break
    fsm_steps.append(k)
print("#define PIH_FSM_STEPS " + iarr(fsm_steps) + "  /* update_state() calls per FSM state (exact replay of the fp64 clock) */")
# ----------------------------------------------------------------------------- UR5 (envs/assets/urdf/ur5.urdf, READ from the file)
ur = list(zip(UR5["rpy"], UR5["xyz"], UR5["axis"]))   # shoulder_pan :32-38, shoulder_lift :60-66, elbow :88-94, wrist_1 :116-122, wrist_2 :145-151, wrist_3 :173-179
print("/* UR5 kinematic chain (ur5.urdf:32-218; world->base fixed xyz 0 0 0.1 :534-539; ee_fixed_joint :201-205), literal 3.14 / 1.6 */")
print("#define PIH_UR5_NJ 6")
print("#define PIH_UR5_RFIX {" + ", ".join(arr(rpy(*j[0]).reshape(-1)) for j in ur) + "}")
print("#define PIH_UR5_TFIX {" + ", ".join(arr(j[1]) for j in ur) + "}")
print("#define PIH_UR5_AXIS {" + ", ".join(arr(j[2]) for j in ur) + "}")
print("#define PIH_UR5_BASE_T " + arr(UR5["base_xyz"]))
print("#define PIH_UR5_EE_R " + arr(rpy(*UR5["ee_rpy"]).reshape(-1)))
print("#define PIH_UR5_EE_T " + arr(UR5["ee_xyz"]))
print("#define PIH_UR5_EFFORT " + arr(UR5["effort"]) + "   /* getJointInfo(i)[10], envs/utils.py:75-78 */")
print("#define PIH_UR5_KP 0.03   /* positionGains, envs/utils.py:82 */")
# ---- UR5 dynamics for the 'random-fly' task: masses / inertial origins / joint damping / limits from the URDF; inertia by
# pybullet's rule without URDF_USE_INERTIA_FROM_FILE (box of the collision-mesh AABB + margin, about the inertial origin, link
# axes) [UNVERIFIED rule, SURVEY.md App. C]; ee_link (fixed, 1 kg, 1 cm box :206-217) merged into wrist_3_link
ur_m, ur_c, ur_I = [], [], []
for k in range(6):
    lo, hi = np.array(UR5["aabb"][k][0]), np.array(UR5["aabb"][k][1])
    parts = [(UR5["mass"][k], UR5["com"][k], box_inertia(UR5["mass"][k], *((hi - lo) + 2 * MARGIN)))]
    if k == 5:
        elo, ehi = np.array(UR5["ee_aabb"][0]), np.array(UR5["ee_aabb"][1])
        Re = rpy(*UR5["ee_rpy"])
        parts.append((UR5["ee_mass"], np.array(UR5["ee_xyz"]) + Re @ np.array(UR5["ee_com"]), Re @ box_inertia(UR5["ee_mass"], *((ehi - elo) + 2 * MARGIN)) @ Re.T))
    m_, c_, I_ = merge(parts)
    ur_m.append(m_); ur_c.append(c_); ur_I.append(I_)
print("/* UR5 dynamics (random-fly task): link k = child of joint k; link 5 = wrist_3_link + ee_link merged */")
print("#define PIH_UR5_MASS " + arr(ur_m))
print("#define PIH_UR5_COM {" + ", ".join(arr(c) for c in ur_c) + "}")
print("#define PIH_UR5_INERTIA {" + ", ".join(arr(sym6(I)) for I in ur_I) + "}  /* xx yy zz xy xz yz about COM, link axes */")
print("#define PIH_UR5_DAMPING " + arr(UR5["damping"]) + "  /* <dynamics damping>, ur5.urdf:38,66,94,122,151,179 */")
print("#define PIH_UR5_LO " + arr(UR5["lower"]))
print("#define PIH_UR5_HI " + arr(UR5["upper"]))
print("#define PIH_UR5_MU %s" % fmt(DEFAULT_MU))
# collision capsules of the arm links, one per link, by a fixed rule from the collision-mesh AABB (BUILD-DEFINED stand-in for the
# STL convex hulls): axis = longest AABB axis through the AABB centre, radius = mean of the other two half extents, end points
# one radius inside the AABB (a sphere when the box is shorter than two radii)
caps = []
for k in range(6):
    lo, hi = np.array(UR5["aabb"][k][0]), np.array(UR5["aabb"][k][1])
    c, h = 0.5 * (lo + hi), 0.5 * (hi - lo)
    ax = int(np.argmax(h))
    r = float(np.mean([h[i_] for i_ in range(3) if i_ != ax]))
    half = max(float(h[ax]) - r, 0.0)
    e = np.zeros(3); e[ax] = half
    caps.append((c - e, c + e, r))
print("#define PIH_UR5_CAP_A {" + ", ".join(arr(np.round(c[0], 9)) for c in caps) + "}   /* capsule end points / radius in the link frame */")
print("#define PIH_UR5_CAP_B {" + ", ".join(arr(np.round(c[1], 9)) for c in caps) + "}")
print("#define PIH_UR5_CAP_R " + arr(round(c[2], 9) for c in caps))
print("#define PIH_UR5_REST {0.0, %s, %s, %s, %s, 0.0}   /* BUILD-DEFINED rest pose (the reference passes ur_orn from a task that is not in the snapshot) */" % (
    fmt(-math.pi / 2), fmt(math.pi / 2), fmt(-math.pi / 2), fmt(-math.pi / 2)))
# ----------------------------------------------------------------------------- free-flying objects of the random-fly task
MAXSPH = max(len(o["sphere_r"]) for o in OBJECTS)
print("/* free-flying objects of the random-fly task, object_id = index (pih_config.object_id; args[0] of README.md:38 names one): every")
print(" * single-link free body under envs/assets/urdf -- " + ", ".join("%s (%s, %s: %s hull%s with %s vertices)" % (
    UT.object_name(o["urdf"]), o["urdf"], o["mesh"], len(o["hull_nvert"]), "" if len(o["hull_nvert"]) == 1 else "s", "/".join(str(n) for n in o["hull_nvert"])) for o in OBJECTS) + ".")
print(" * mass / lateral_friction / contact_erp from the URDF; inertia = AABB box of the collision mesh (pybullet's rule without")
print(" * URDF_USE_INERTIA_FROM_FILE); collision = spheres by tools/urdf_tables.py cover_with_spheres (BUILD-DEFINED stand-in for the hulls:")
print(" * one sphere per hull of a multi-hull file, a row of spheres along the longest axis of a single-hull file), padded to MAXSPH */")
print("#define PIH_FLY_NOBJ %d" % len(OBJECTS))
print("#define PIH_FLY_OBJ_NAMES {" + ", ".join('"%s"' % UT.object_name(o["urdf"]) for o in OBJECTS) + "}")
print("#define PIH_FLY_OBJ_MAXSPH %d" % MAXSPH)
print("#define PIH_FLY_OBJ_MASS " + arr(o["mass"] for o in OBJECTS))
inert = []
for o in OBJECTS:
    lo, hi = np.array(o["aabb"][0]), np.array(o["aabb"][1])
    inert.append(np.diag(box_inertia(o["mass"], *((hi - lo) + 2 * MARGIN))))
print("#define PIH_FLY_OBJ_INERTIA {" + ", ".join(arr(i) for i in inert) + "}   /* xx yy zz about the inertial origin, body axes */")
print("#define PIH_FLY_OBJ_MU " + arr(o["friction"] for o in OBJECTS))
print("#define PIH_FLY_OBJ_CONTACT_ERP " + arr(o["contact_erp"] for o in OBJECTS))
print("#define PIH_FLY_OBJ_NSPH " + iarr(len(o["sphere_r"]) for o in OBJECTS))
print("#define PIH_FLY_OBJ_SPH_C {" + ", ".join("{" + ", ".join(arr(np.round(c, 9)) for c in (o["sphere_c"] + [[0.0, 0.0, 0.0]] * MAXSPH)[:MAXSPH]) + "}" for o in OBJECTS) + "}")
print("#define PIH_FLY_OBJ_SPH_R {" + ", ".join(arr(round(r, 9) for r in (o["sphere_r"] + [0.0] * MAXSPH)[:MAXSPH]) for o in OBJECTS) + "}")
print("#define PIH_FLY_OBJ_RGB {" + ", ".join(arr(o["rgba"][:3]) for o in OBJECTS) + "}   /* <material> colour */")
print("#endif")
