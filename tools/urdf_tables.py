#!/usr/bin/env python3
"""URDF / OBJ -> model-table reader (offline tool).

Reads a URDF with xml.etree, merges fixed joints, applies pybullet's `globalScaling`, and derives what the kernels need:
joint origins / axes / limits / damping, link masses / inertial origins / lateral friction, and -- because the reference
loads without URDF_USE_INERTIA_FROM_FILE -- the collision AABB from the referenced OBJ mesh (for the box-inertia rule of
SURVEY.md App. C).  It is used to CHECK the committed include/pih_model.h against the reference's asset files
(tests/test_model_tables.py; only where /root/reference exists) and is the seed of a general custom-task importer.
Nothing here runs on the GPU box.
"""
import math
import os
import xml.etree.ElementTree as ET

import numpy as np


def _floats(s, n=3, default=0.0):
    if s is None:
        return [default] * n
    return [float(x) for x in s.split()]


def rpy_matrix(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]]); Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def obj_vertices(path):
    v = []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                v.append([float(x) for x in line.split()[1:4]])
    return np.array(v)


def stl_vertices(path):
    """Vertices of a binary STL (80-byte header, uint32 triangle count, 50 bytes per triangle)."""
    import struct
    b = open(path, "rb").read()
    n = struct.unpack("<I", b[80:84])[0]
    assert len(b) == 84 + 50 * n, "not a binary STL: %s" % path
    a = np.frombuffer(b[84:], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    return a["v"].reshape(-1, 3).astype(np.float64)


def obj_hulls(path):
    """Vertex arrays of the `o <name>` groups of an OBJ (banana_collision.obj: one convex hull per group)."""
    hulls, cur = [], None
    with open(path) as f:
        for line in f:
            if line.startswith("o "):
                cur = []; hulls.append(cur)
            elif line.startswith("v ") and cur is not None:
                cur.append([float(x) for x in line.split()[1:4]])
    return [np.array(h) for h in hulls]


def mesh_vertices(path):
    return stl_vertices(path) if path.lower().endswith(".stl") else obj_vertices(path)


class Urdf:
    def __init__(self, path, scale=1.0):
        self.path = path
        self.scale = scale
        root = ET.parse(path).getroot()
        self.links = {}
        for l in root.findall("link"):
            d = {"name": l.get("name"), "mass": 0.0, "com": [0, 0, 0], "friction": None, "collision": None, "rgba": None}
            vis = l.find("visual")
            if vis is not None and vis.find("material") is not None and vis.find("material").find("color") is not None:
                d["rgba"] = _floats(vis.find("material").find("color").get("rgba"), 4)
            ine = l.find("inertial")
            if ine is not None:
                m = ine.find("mass")
                d["mass"] = float(m.get("value")) if m is not None else 0.0
                origins = ine.findall("origin")
                if origins:
                    d["com"] = [x * scale for x in _floats(origins[-1].get("xyz"))]     # the last <origin> wins, as in urdfdom
            c = l.find("contact")
            if c is not None and c.find("lateral_friction") is not None:
                d["friction"] = float(c.find("lateral_friction").get("value"))
            col = l.find("collision")
            if col is not None:
                o = col.find("origin")
                mesh = col.find("geometry").find("mesh") if col.find("geometry") is not None else None
                box = col.find("geometry").find("box") if col.find("geometry") is not None else None
                d["collision"] = {"xyz": [x * scale for x in _floats(o.get("xyz") if o is not None else None)],
                                  "mesh": mesh.get("filename") if mesh is not None else None,
                                  "box": [x * scale for x in _floats(box.get("size"))] if box is not None else None}
            self.links[d["name"]] = d
        self.joints = []
        for j in root.findall("joint"):
            o = j.find("origin")
            ax = j.find("axis")
            lim = j.find("limit")
            dyn = j.find("dynamics")
            self.joints.append({
                "name": j.get("name"), "type": j.get("type"), "parent": j.find("parent").get("link"), "child": j.find("child").get("link"),
                "xyz": [x * scale for x in _floats(o.get("xyz") if o is not None else None)],
                "rpy": _floats(o.get("rpy") if o is not None else None),
                "axis": _floats(ax.get("xyz")) if ax is not None else [1, 0, 0],
                "effort": float(lim.get("effort")) if lim is not None and lim.get("effort") else None,
                "lower": lim.get("lower") if lim is not None else None, "upper": lim.get("upper") if lim is not None else None,
                "damping": float(dyn.get("damping")) if dyn is not None and dyn.get("damping") else 0.0})

    def mesh_aabb_extents(self, link, margin=0.001):
        """Scaled AABB extents (+ 2 * collision margin) of the link's collision mesh."""
        lo, hi = self.mesh_aabb(link)
        return (hi - lo) + 2 * margin

    def mesh_aabb(self, link):
        """Scaled AABB (min, max) of the link's collision geometry in the link frame (mesh file or <box>)."""
        c = self.links[link]["collision"]
        if c["box"] is not None:
            h = 0.5 * np.array(c["box"]); o = np.array(c["xyz"])
            return o - h, o + h
        p = os.path.normpath(os.path.join(os.path.dirname(self.path), c["mesh"]))
        v = mesh_vertices(p) * self.scale + np.array(c["xyz"])
        return v.min(0), v.max(0)

    def chain(self, root_link):
        """Joints in order down a serial chain starting at root_link."""
        out, cur = [], root_link
        by_parent = {j["parent"]: j for j in self.joints}
        while cur in by_parent:
            j = by_parent[cur]
            out.append(j)
            cur = j["child"]
        return out


def box_inertia_diag(m, ext):
    lx, ly, lz = ext
    return [m / 12 * (ly * ly + lz * lz), m / 12 * (lx * lx + lz * lz), m / 12 * (lx * lx + ly * ly)]


def pipe_tables(ref_root):
    """What include/pih_model.h encodes for the pipe (envs/assets/urdf/pipe.urdf, globalScaling 0.01)."""
    u = Urdf(os.path.join(ref_root, "peg_in_hole_gym/envs/assets/urdf/pipe.urdf"), scale=0.01)
    ch = u.chain("pipe_link0")
    assert [j["type"] for j in ch] == ["fixed"] + ["continuous"] * 23
    ext = u.mesh_aabb_extents("pipe_link0")
    links = ["pipe_link0"] + [j["child"] for j in ch]      # by chain order: link 16 is named "pipe_link10.0111" in the file
    return {"joint_xyz": [j["xyz"] for j in ch], "joint_axis": [j["axis"] for j in ch], "mass": [u.links[l]["mass"] for l in links],
            "com": [u.links[l]["com"] for l in links], "friction": [u.links[l]["friction"] for l in links], "aabb_ext": ext.tolist(),
            "collision_xyz": [u.links[l]["collision"]["xyz"] for l in links]}


def hole_tables(ref_root):
    v = obj_vertices(os.path.join(ref_root, "peg_in_hole_gym/envs/assets/obj/cylinder_base.obj")) * 0.016
    rad = np.sqrt(v[:, 0] ** 2 + v[:, 2] ** 2)
    return {"rin": float(rad.min()), "rout": float(rad.max()), "halflen": float(np.abs(v[:, 1]).max())}


def ur5_tables(ref_root):
    u = Urdf(os.path.join(ref_root, "peg_in_hole_gym/envs/assets/urdf/ur5.urdf"))
    ch = u.chain("world")
    rev = [j for j in ch if j["type"] == "revolute"]
    fixed = [j for j in ch if j["type"] == "fixed"]
    links = [j["child"] for j in rev]
    ee = fixed[-1]["child"]
    return {"base_xyz": fixed[0]["xyz"], "rpy": [j["rpy"] for j in rev], "xyz": [j["xyz"] for j in rev], "axis": [j["axis"] for j in rev],
            "effort": [j["effort"] for j in rev], "damping": [j["damping"] for j in rev], "ee_rpy": fixed[-1]["rpy"], "ee_xyz": fixed[-1]["xyz"],
            "lower": [float(j["lower"]) for j in rev], "upper": [float(j["upper"]) for j in rev],
            "mass": [u.links[l]["mass"] for l in links], "com": [u.links[l]["com"] for l in links],
            "aabb": [[a.tolist() for a in u.mesh_aabb(l)] for l in links],
            "ee_mass": u.links[ee]["mass"], "ee_com": u.links[ee]["com"], "ee_aabb": [a.tolist() for a in u.mesh_aabb(ee)]}


def _urdf_extras(path):
    """<contact> children and <collision><geometry><cylinder> of every link (fields the small Urdf class does not keep)"""
    root = ET.parse(path).getroot()
    out = {}
    for l in root.findall("link"):
        d = {"contact_erp": None, "cylinder": None, "inertia_diag": None}
        c = l.find("contact")
        if c is not None and c.find("contact_erp") is not None:
            d["contact_erp"] = float(c.find("contact_erp").get("value"))
        ine = l.find("inertial")
        if ine is not None and ine.find("inertia") is not None:
            i = ine.find("inertia"); d["inertia_diag"] = [float(i.get(k)) for k in ("ixx", "iyy", "izz")]
        col = l.find("collision")
        if col is not None and col.find("geometry") is not None and col.find("geometry").find("cylinder") is not None:
            cy = col.find("geometry").find("cylinder"); o = col.find("origin")
            d["cylinder"] = {"radius": float(cy.get("radius")), "length": float(cy.get("length")),
                             "xyz": _floats(o.get("xyz") if o is not None else None), "rpy": _floats(o.get("rpy") if o is not None else None)}
        out[l.get("name")] = d
    return out


def cover_with_spheres(hulls):
    """Sphere stand-in for the convex collision hulls of a free body (BUILD-DEFINED rule, the same for every object):
      * a file that comes as SEVERAL hulls (banana_collision.obj: one `o` group per hull of a convex decomposition) gets one sphere per
        hull: centre = the hull's AABB centre, radius = the mean of its three half extents;
      * a file with ONE hull (Amicelli_800_tex.obj: a single closed mesh, which pybullet wraps in its convex hull) gets a row of spheres
        along the longest AABB axis: radius r = mean of the two other half extents, count = ceil(longest half extent / r), centres evenly
        spaced so that the end spheres touch the ends of the AABB."""
    if len(hulls) > 1:
        return [(0.5 * (h.min(0) + h.max(0)), float(np.mean(0.5 * (h.max(0) - h.min(0))))) for h in hulls]
    h = hulls[0]
    c = 0.5 * (h.min(0) + h.max(0)); he = 0.5 * (h.max(0) - h.min(0))
    ax = int(np.argmax(he))
    r = float(np.mean([he[i] for i in range(3) if i != ax]))
    n = max(1, int(math.ceil(he[ax] / r)))
    span = max(he[ax] - r, 0.0)
    out = []
    for k in range(n):
        t = 0.0 if n == 1 else -span + 2 * span * k / (n - 1)
        e = np.zeros(3); e[ax] = t
        out.append((c + e, r))
    return out


def free_body_tables(ref_root, urdf_file):
    """A single-link free body of the reference (envs/assets/urdf/banana.urdf, Amicelli_800_tex.urdf): mass, inertial origin, lateral
    friction, contact_erp, colour and its collision mesh (one convex hull per `o` group of the OBJ, or the whole file as one hull)."""
    path = os.path.join(ref_root, "peg_in_hole_gym/envs/assets/urdf", urdf_file)
    u = Urdf(path)
    assert len(u.links) == 1 and not u.joints, "%s is not a single free body" % urdf_file
    name, l = next(iter(u.links.items()))
    ex = _urdf_extras(path)[name]
    mesh = os.path.normpath(os.path.join(os.path.dirname(u.path), l["collision"]["mesh"]))
    hulls = obj_hulls(mesh) or [obj_vertices(mesh)]
    allv = np.concatenate(hulls)
    sph = cover_with_spheres(hulls)
    return {"urdf": urdf_file, "link": name, "mass": l["mass"], "com": l["com"], "friction": l["friction"], "contact_erp": ex["contact_erp"], "rgba": l["rgba"],
            "mesh": os.path.basename(mesh), "aabb": [allv.min(0).tolist(), allv.max(0).tolist()],
            "hull_aabb": [[h.min(0).tolist(), h.max(0).tolist()] for h in hulls], "hull_nvert": [len(h) for h in hulls],
            "sphere_c": [c.tolist() for c, _ in sph], "sphere_r": [r for _, r in sph]}


def object_name(urdf_file):
    """args[0] of the reference's usage line (README.md:38 `args=['Banana', 1/120.]`) for an asset file: the file stem up to the first
    underscore, capitalised (banana.urdf -> 'Banana', Amicelli_800_tex.urdf -> 'Amicelli')"""
    stem = os.path.splitext(urdf_file)[0].split("_")[0]
    return stem[0].upper() + stem[1:]


FLY_OBJECT_FILES = ("banana.urdf", "Amicelli_800_tex.urdf")      # every single-link free body under envs/assets/urdf, in this order = object_id


def banana_tables(ref_root):
    """envs/assets/urdf/banana.urdf:1-32 + obj/banana_collision.obj (one convex hull per `o` group)."""
    return free_body_tables(ref_root, "banana.urdf")


def hinged_body_tables(ref_root, urdf_file="charge_board.urdf"):
    """A fixed base with ONE revolute link (envs/assets/urdf/charge_board.urdf: world -> door_base fixed, door_base -> door hinge with a
    primitive cylinder as collision shape): hinge origin / axis / limits / damping, door mass, the file's inertia and the inertia by
    pybullet's rule (box of the collision AABB), and the cylinder in the door frame."""
    path = os.path.join(ref_root, "peg_in_hole_gym/envs/assets/urdf", urdf_file)
    u = Urdf(path); ex = _urdf_extras(path)
    rev = [j for j in u.joints if j["type"] == "revolute"]; fixed = [j for j in u.joints if j["type"] == "fixed"]
    assert len(rev) == 1 and len(fixed) == 1, "%s is not a fixed base with one hinge" % urdf_file
    j = rev[0]; door = j["child"]; cy = ex[door]["cylinder"]
    R = rpy_matrix(*cy["rpy"]); axis = R @ np.array([0.0, 0.0, 1.0])          # a URDF cylinder is along its local z
    half = np.abs(axis) * 0.5 * cy["length"] + (1 - np.abs(axis)) * cy["radius"]   # AABB half extents of the (axis-aligned) cylinder
    return {"urdf": urdf_file, "base_xyz": fixed[0]["xyz"], "hinge_xyz": j["xyz"], "hinge_axis": j["axis"], "lower": float(j["lower"]), "upper": float(j["upper"]),
            "damping": j["damping"], "mass": u.links[door]["mass"], "inertia_file": ex[door]["inertia_diag"],
            "inertia_rule": box_inertia_diag(u.links[door]["mass"], 2 * half + 0.002), "cyl_radius": cy["radius"], "cyl_length": cy["length"],
            "cyl_xyz": cy["xyz"], "cyl_axis": axis.round(12).tolist()}


if __name__ == "__main__":
    import json
    import sys
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    print(json.dumps({"pipe": pipe_tables(ref), "hole": hole_tables(ref), "ur5": ur5_tables(ref), "objects": [free_body_tables(ref, f) for f in FLY_OBJECT_FILES],
                      "charge_board": hinged_body_tables(ref)}, indent=1)[:9000])
