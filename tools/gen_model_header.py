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

The header is plain C initialiser macros so that both the fp64 oracle (oracle/) and the HIP
product (peg_in_hole_gym_amd/csrc) instantiate the same numbers in their own storage classes.
Run:  python tools/gen_model_header.py > include/pih_model.h
"""
import math
import numpy as np

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
    return repr(float(x))


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

# ----------------------------------------------------------------------------- pipe (object links 0..23)
S = 0.01  # globalScaling envs/peg_in_hole.py:242
# PyBullet (no URDF_USE_INERTIA_FROM_FILE) recomputes link inertia from the collision AABB [UNVERIFIED,
# SURVEY.md App. C]: cylinder r=1,len 6 scaled + 1 mm margin -> box 0.022 x 0.062 x 0.022
pipe_box = lambda m: box_inertia(m, 0.022, 0.062, 0.022)
# root = pipe_link0 (m .00111, com at origin, pipe.urdf:13-19) + pipe_link1 (fixed at y=3, m .0111,
# inertial origin y=1.5, pipe.urdf:39-43,:52-56)
m0, c0, I0 = merge([(0.00111, (0, 0, 0), pipe_box(0.00111)), (0.0111, (0, (3 + 1.5) * S, 0), pipe_box(0.0111))])
add("pipe_link0+1", -1, FLT, np.eye(3), (0, 0, 0), (0, 0, 0), m0, c0, I0, 0, 0, 0, 0.0, 100.0)
for k in range(2, 25):  # pipe_link2..24 ; joint index k-1 (pipe.urdf:76-83 ... :791-798)
    j = k - 1  # object link index
    ty = (3 + 5.5) * S if k == 2 else 5.5 * S  # link2's joint sits on link1 (y=5.5) which sits at y=3 of link0
    axis = (0, 0, 1) if k % 2 == 0 else (1, 0, 0)  # alternating z,x (pipe.urdf:81,114,...)
    com = (0, 1.5 * S, 0) if k == 24 else (0, 0, 0)  # inertial origin only on link1 & link24 (pipe.urdf:53,807)
    mu = 100.0 if k >= 23 else 0.5  # lateral_friction 100 on links 0,1,23,24 (pipe.urdf:10,48,765,802)
    add("pipe_link%d" % k, ARM_NL + j - 1, REV, np.eye(3), (0, ty, 0), axis, 0.0111, com, pipe_box(0.0111), 0, 0, 0, 0.0, mu)
NL = len(links)
OBJ_NL = NL - ARM_NL

# collision rope: vertices V0..V24 (sphere r = 1 cm) ; segment s on object link s from V_s to V_{s+1}
PIPE_R = 1.0 * S
samples = []  # (obj_link, local_y, is_vertex)
seg_len = [8.5 * S - 1.0 * S] + [5.5 * S] * 22 + [5.0 * S]
seg_y0 = [1.0 * S] + [0.0] * 23
for s in range(24):
    n_int = 6 if s == 0 else 4
    samples.append((s, seg_y0[s], 1))
    for i in range(1, n_int + 1):
        samples.append((s, seg_y0[s] + seg_len[s] * i / (n_int + 1), 0))
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
print("#define PIH_HOLE_HALFLEN %s  /* 1.0 * 0.016 */" % fmt(0.016))
print("#define PIH_HOLE_RIN %s   /* 0.96 * 0.016 */" % fmt(0.96 * 0.016))
print("#define PIH_HOLE_ROUT %s  /* 1.2 * 0.016 */" % fmt(1.2 * 0.016))
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
# ----------------------------------------------------------------------------- UR5 (envs/assets/urdf/ur5.urdf, exact)
ur = [  # (rpy, xyz, axis)  shoulder_pan :32-38, shoulder_lift :60-66, elbow :88-94, wrist_1 :116-122, wrist_2 :145-151, wrist_3 :173-179
    ((0.0, 0.0, 3.14), (0.0, 0.0, 0.089159), (0, 0, 1)),
    ((0.0, 1.6, 0.0), (0.0, 0.13585, 0.0), (0, 1, 0)),
    ((0.0, 0.0, 0.0), (0.0, -0.1197, 0.425), (0, 1, 0)),
    ((0.0, 1.57079632679, 0.0), (0.0, 0.0, 0.39225), (0, 1, 0)),
    ((0.0, 0.0, 0.0), (0.0, 0.093, 0.0), (0, 0, 1)),
    ((0.0, 0.0, 0.0), (0.0, 0.0, 0.09465), (0, 1, 0)),
]
print("/* UR5 kinematic chain (ur5.urdf:32-218; world->base fixed xyz 0 0 0.1 :534-539; ee_fixed_joint :201-205), literal 3.14 / 1.6 */")
print("#define PIH_UR5_NJ 6")
print("#define PIH_UR5_RFIX {" + ", ".join(arr(rpy(*j[0]).reshape(-1)) for j in ur) + "}")
print("#define PIH_UR5_TFIX {" + ", ".join(arr(j[1]) for j in ur) + "}")
print("#define PIH_UR5_AXIS {" + ", ".join(arr(j[2]) for j in ur) + "}")
print("#define PIH_UR5_BASE_T {0.0, 0.0, 0.1}")
print("#define PIH_UR5_EE_R " + arr(rpy(0.0, 0.0, 1.57079632679).reshape(-1)))
print("#define PIH_UR5_EE_T {0.0, 0.0823, 0.0}")
print("#define PIH_UR5_EFFORT {300.0, 300.0, 300.0, 300.0, 300.0, 300.0}   /* getJointInfo(i)[10], envs/utils.py:75-78 */")
print("#define PIH_UR5_KP 0.03   /* positionGains, envs/utils.py:82 */")
print("#endif")
