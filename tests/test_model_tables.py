"""The committed model header (include/pih_model.h) against the reference's own asset files, parsed by tools/urdf_tables.py.
Runs only where /root/reference exists (the build container); on the GPU box it is skipped."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "peg_in_hole_gym")), reason="reference assets not present")


def _macro(name):
    hdr = open(os.path.join(ROOT, "include", "pih_model.h")).read()
    body = re.search(r"#define %s (.*)" % name, hdr).group(1).split("/*")[0]
    return np.array(eval(body.replace("{", "[").replace("}", "]")), dtype=float)


@pytest.fixture(scope="module")
def T():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import urdf_tables
    return urdf_tables


def test_pipe_tables_match_pipe_urdf(T):
    p = T.pipe_tables(REF)
    tfix, axis, mass, com, mu = _macro("PIH_LINK_TFIX")[9:], _macro("PIH_LINK_AXIS")[9:], _macro("PIH_LINK_MASS")[9:], _macro("PIH_LINK_COM")[9:], _macro("PIH_LINK_MU")[9:]
    jx = np.array(p["joint_xyz"])
    # object link 1 = pipe_link2: its joint sits on link1, which sits at joint0 (fixed) of link0
    np.testing.assert_allclose(tfix[1], jx[0] + jx[1], atol=1e-15)
    np.testing.assert_allclose(tfix[2:], jx[2:], atol=1e-15)
    np.testing.assert_allclose(axis[1:], np.array(p["joint_axis"])[1:], atol=0)
    assert abs(mass[0] - (p["mass"][0] + p["mass"][1])) < 1e-15 and np.allclose(mass[1:], p["mass"][2:])
    # merged root COM = mass-weighted (link0 com, link1 com shifted by the fixed joint)
    c1 = jx[0] + np.array(p["com"][1])
    np.testing.assert_allclose(com[0], (p["mass"][0] * np.array(p["com"][0]) + p["mass"][1] * c1) / mass[0], atol=1e-15)
    np.testing.assert_allclose(com[1:], np.array(p["com"])[2:], atol=1e-15)
    assert [f if f is not None else 0.5 for f in p["friction"]][2:] == mu[1:].tolist() and mu[0] == p["friction"][0] == p["friction"][1]
    # AABB box inertia of an unmerged link, and the rope radius / cylinder placement
    np.testing.assert_allclose(p["aabb_ext"], [0.022, 0.062, 0.022], atol=1e-12)
    I = _macro("PIH_LINK_INERTIA")[10]
    np.testing.assert_allclose(I[:3], T.box_inertia_diag(0.0111, p["aabb_ext"]), rtol=1e-12)
    assert abs(_macro("PIH_PIPE_RADIUS") - (p["aabb_ext"][0] - 0.002) / 2) < 1e-15
    assert all(abs(c[1] - 0.03) < 1e-15 for c in p["collision_xyz"])        # every cylinder spans y in [0, 6 cm] of its link


def test_hole_tables_match_obj(T):
    h = T.hole_tables(REF)
    assert abs(_macro("PIH_HOLE_RIN") - h["rin"]) < 1e-6 and abs(_macro("PIH_HOLE_ROUT") - h["rout"]) < 1e-6
    assert abs(_macro("PIH_HOLE_HALFLEN") - h["halflen"]) < 1e-12


def test_ur5_tables_match_ur5_urdf(T):
    u = T.ur5_tables(REF)
    np.testing.assert_allclose(_macro("PIH_UR5_TFIX"), u["xyz"], atol=0)
    np.testing.assert_allclose(_macro("PIH_UR5_AXIS"), u["axis"], atol=0)
    np.testing.assert_allclose(_macro("PIH_UR5_BASE_T"), u["base_xyz"], atol=0)
    np.testing.assert_allclose(_macro("PIH_UR5_EE_T"), u["ee_xyz"], atol=0)
    np.testing.assert_allclose(_macro("PIH_UR5_EFFORT"), u["effort"], atol=0)
    R = _macro("PIH_UR5_RFIX")
    for i, rpy in enumerate(u["rpy"]):
        np.testing.assert_allclose(R[i].reshape(3, 3), T.rpy_matrix(*rpy), atol=1e-15)
    np.testing.assert_allclose(_macro("PIH_UR5_EE_R").reshape(3, 3), T.rpy_matrix(*u["ee_rpy"]), atol=1e-15)
    assert u["rpy"][0][2] == 3.14 and u["rpy"][1][1] == 1.6 and u["damping"] == [0.5] * 6       # the literal 3.14 / 1.6 of the file


def test_link_tree_shape_assumed_by_the_lane_parallel_sweeps():
    """pih_device.h derives joint type and parent of link L arithmetically in its lane-parallel paths (link_velocities_scan,
    response): arm 0-6 revolute chain, fingers 7/8 prismatic children of link 6, link 9 floating root, 10-32 revolute chain."""
    import re
    hdr = open(os.path.join(ROOT, "include", "pih_model.h")).read()
    def arr(name):
        m = re.search(r"#define %s \{([^}]*)\}" % name, hdr)
        return [int(x) for x in m.group(1).split(",")]
    jt, par = arr("PIH_LINK_JTYPE"), arr("PIH_LINK_PARENT")
    REV, PRI = 0, 1
    FLO = int(re.search(r"#define PIH_JT_FLOATING (\d+)", hdr).group(1))
    assert len(jt) == 33 and len(par) == 33
    assert jt[:7] == [REV] * 7 and jt[7:9] == [PRI] * 2 and jt[9] == FLO and jt[10:] == [REV] * 23
    assert par[:9] == [-1, 0, 1, 2, 3, 4, 5, 6, 6] and par[9] == -1 and par[10:] == list(range(9, 32))


def test_header_regenerates_byte_identical_from_the_reference_assets():
    """tools/gen_model_header.py READS pipe.urdf, hole's OBJ, ur5.urdf + its STL collision meshes and banana.urdf + its hull
    OBJ from the reference tree (only the Panda / table block is hand-entered, their assets live in the absent pybullet_data):
    regenerating must reproduce the committed include/pih_model.h byte for byte."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_model_header.py"), "--ref", REF], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout == open(os.path.join(ROOT, "include", "pih_model.h")).read()


def test_ur5_dynamics_and_banana_tables_match_the_files(T):
    u = T.ur5_tables(REF)
    np.testing.assert_allclose(_macro("PIH_UR5_MASS")[:5], u["mass"][:5], atol=0)
    assert abs(_macro("PIH_UR5_MASS")[5] - (u["mass"][5] + u["ee_mass"])) < 1e-15          # wrist_3_link + ee_link (fixed joint)
    np.testing.assert_allclose(_macro("PIH_UR5_COM")[:5], u["com"][:5], atol=0)
    np.testing.assert_allclose(_macro("PIH_UR5_DAMPING"), u["damping"], atol=0)
    np.testing.assert_allclose(_macro("PIH_UR5_LO"), u["lower"], atol=0); np.testing.assert_allclose(_macro("PIH_UR5_HI"), u["upper"], atol=0)
    # box-inertia rule on the STL AABB (+ 1 mm margin each side) for a link without a merged child
    lo, hi = np.array(u["aabb"][1][0]), np.array(u["aabb"][1][1])
    np.testing.assert_allclose(_macro("PIH_UR5_INERTIA")[1][:3], T.box_inertia_diag(u["mass"][1], (hi - lo) + 0.002), rtol=1e-12)
    # capsules stay inside the link's AABB
    A, B, R = _macro("PIH_UR5_CAP_A"), _macro("PIH_UR5_CAP_B"), _macro("PIH_UR5_CAP_R")
    for k in range(6):
        lo, hi = np.array(u["aabb"][k][0]), np.array(u["aabb"][k][1])
        ax = int(np.argmax(hi - lo))
        assert A[k][ax] - R[k] >= lo[ax] - 1e-6 and B[k][ax] + R[k] <= hi[ax] + 1e-6
    b = T.banana_tables(REF)
    assert _macro("PIH_FLY_OBJ_MASS")[0] == b["mass"] == 1.0 and _macro("PIH_FLY_OBJ_MU")[0] == b["friction"] == 0.0
    assert int(_macro("PIH_FLY_OBJ_NSPH")[0]) == len(b["hull_aabb"]) == 5
    ext = np.array(b["aabb"][1]) - np.array(b["aabb"][0]) + 0.002
    np.testing.assert_allclose(_macro("PIH_FLY_OBJ_INERTIA")[0], T.box_inertia_diag(1.0, ext), rtol=1e-9)


def test_every_free_body_and_the_hinged_board_of_the_reference_become_tables(T):
    """SURVEY 8f-4 (asset -> task): the generator takes ANY single-link free body under envs/assets/urdf -- banana.urdf and
    Amicelli_800_tex.urdf -- into the object table the random-fly kernel consumes (object_id = index, name = args[0] of README.md:38), and
    reads charge_board.urdf (fixed base + one hinge with limits / damping, primitive cylinder)."""
    files = sorted(f for f in os.listdir(os.path.join(REF, "peg_in_hole_gym/envs/assets/urdf")) if f.endswith(".urdf"))
    free = []
    for f in files:
        u = T.Urdf(os.path.join(REF, "peg_in_hole_gym/envs/assets/urdf", f))
        if len(u.links) == 1 and not u.joints and f != "hole.urdf":      # (hole.urdf is the peg-in-hole scene's FIXED tube, envs/peg_in_hole.py:248-251)
            free.append(f)
    assert sorted(free) == sorted(T.FLY_OBJECT_FILES)                         # nothing left out
    names = re.findall(r'"([^"]+)"', re.search(r"#define PIH_FLY_OBJ_NAMES \{([^}]*)\}", open(os.path.join(ROOT, "include", "pih_model.h")).read()).group(1))
    assert names == [T.object_name(f) for f in T.FLY_OBJECT_FILES] == ["Banana", "Amicelli"]
    am = T.free_body_tables(REF, "Amicelli_800_tex.urdf")
    assert am["mass"] == 1.0 and am["friction"] == 0.0 and am["contact_erp"] == 0.0 and am["hull_nvert"] == [400]
    assert _macro("PIH_FLY_OBJ_MASS")[1] == 1.0 and int(_macro("PIH_FLY_OBJ_NSPH")[1]) == len(am["sphere_r"]) == 2
    lo, hi = np.array(am["aabb"][0]), np.array(am["aabb"][1])
    np.testing.assert_allclose(_macro("PIH_FLY_OBJ_INERTIA")[1], T.box_inertia_diag(1.0, (hi - lo) + 0.002), rtol=1e-9)
    C, R = np.array(_macro("PIH_FLY_OBJ_SPH_C")[1]), np.array(_macro("PIH_FLY_OBJ_SPH_R")[1])
    ax = int(np.argmax(hi - lo))
    for k in range(2):                                                         # the spheres stay inside the mesh's AABB along its long axis and touch its ends
        assert lo[ax] - 1e-6 <= C[k][ax] - R[k] and C[k][ax] + R[k] <= hi[ax] + 1e-6
    assert abs((C[0][ax] - R[0]) - lo[ax]) < 1e-6 and abs((C[1][ax] + R[1]) - hi[ax]) < 1e-6 and (R[2:] == 0).all()
    np.testing.assert_allclose(_macro("PIH_FLY_OBJ_RGB")[1], [0.431, 0.185, 0.327])
    # charge_board.urdf (fixed base + one hinge with limits / damping, primitive <cylinder>): the reader handles it; its tables are not part of
    # the product header (nothing of the reference loads the file, no kernel consumed them)
    d = T.hinged_body_tables(REF)
    assert d["lower"] == -2.09439510239 and d["upper"] == 0.0 and d["damping"] == 1.0
    np.testing.assert_allclose(d["hinge_axis"], [0, 0, 1]); np.testing.assert_allclose(d["base_xyz"], [0.04, 0, 0])
    np.testing.assert_allclose(np.abs(d["cyl_axis"]), [0, 1, 0], atol=1e-9)    # <origin rpy="1.5708 0 0">: the disc's axis is the door's y
    assert d["cyl_radius"] == 0.04 and 0.5 * d["cyl_length"] == 0.005
    assert "PIH_DOOR" not in open(os.path.join(ROOT, "include", "pih_model.h")).read()
