"""Shared constructed scenarios for the parity tests (test infrastructure)."""
import numpy as np


def coil_pipe_flat(s):
    """Lay every env's pipe flat on the table as a planar spiral (only the z-axis joints are bent, tighter for higher env
    indices): all 25 vertex spheres touch the table and the inner turns touch each other, so a step sees 25..~42 contacts,
    i.e. the > 20 (global-scratch spill) and > 32 (third sign word) contact paths of the HIP PGS.  `s`: oracle state
    [n, >= 98] (modified in place and returned)."""
    n = s.shape[0]
    for e in range(n):
        s[e, 31:54] = 0
        s[e, 31:54:2] = np.linspace(1.5, 0.5, 12) * (0.8 + 0.05 * (e % 8))
        s[e, 20] = -0.04 + 1e-4
        s[e, 25:31] = 0
        s[e, 54:77] = 0
    return s


def stiff_finger_contact(contacts):
    """True if the oracle's contact list ([k,12]: linkA linkB p n depth mu key lambda_n) describes an ILL-CONDITIONED step for PGS:
    the arm (finger pad, hand or wrist sphere) presses on the pipe while a contact with the clamped friction mu = 10 is loaded
    (one of the pipe's 11-gram end links, URDF friction 100, against a finger, the hand or the table).  With pyramid friction that
    large on a body that light, 50 PGS iterations are not a contraction, so rounding differences between two implementations are
    amplified within one step (the fp64 build of the product algorithm shows the same against the fp64 oracle, and
    so do the two fp32 solver paths of the product against each other)."""
    c = np.asarray(contacts)
    if len(c) == 0:
        return False
    loaded = c[:, 11] > 0
    arm = (c[:, 1] >= 0) & (c[:, 1] <= 8) & loaded
    mu10 = (c[:, 9] >= 10.0) & loaded
    return bool(arm.any() and mu10.any())


def stiff_mask(contacts, counts):
    """Vectorised stiff_finger_contact over all envs: contacts [n, CMAX, 12], counts [n] (Oracle.debug_contacts_all) -> bool [n]."""
    c = np.asarray(contacts); n, m, _ = c.shape
    live = np.arange(m)[None, :] < np.asarray(counts)[:, None]
    loaded = live & (c[:, :, 11] > 0)
    arm = (c[:, :, 1] >= 0) & (c[:, :, 1] <= 8) & loaded
    mu10 = (c[:, :, 9] >= 10.0) & loaded
    return arm.any(1) & mu10.any(1)
