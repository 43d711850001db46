#!/usr/bin/env python3
"""Algorithmic fp32 operations of ONE env-step on the formulation the product EXECUTES (a*b+c = 2; compares / min / max / moves = 0).

Phases other than PGS are counted, not modelled: profiles/flops_latest.json (tools/count_flops.py: the algorithm headers built for the
host with real = CountedReal) -- they are the same code on the GPU.  The host build solves PGS in DOF space, which the product only does
for > 32 contacts, so the PGS phase is modelled here per env from what the step kernel reports about itself (state words 106 contact
count, 107 iterations executed, 114 solver variant), following pih_wave.h:

  row space (variants 1, 2, 4: <= 10 contacts, one row per lane; 5: 11..32 contacts, two rows per lane), n = 32 + 3 nc matrix rows:
    Jacobian rows            3 nc x 38 entries x 10              (cross product + dot per entry)
    Delassus contact columns n x 3 nc x 76                       (38 FMA per entry; motor columns come from the symmetry, 0 flops)
    Bn = [own] - dinv A      n x n x 2
    per iteration            r x (2 n + 3)                       r = 32 + L + 3 nc row updates: clamp (0), subtract (1), one FMA on each
                                                                 of the n z's (2 n), arm-joint chain (2); L = 18 limit rows, or 4 when the
                                                                 limit rows of arm joints 0..6 are provably inactive (variant 1)
    du = sum_i W_i lambda_i  n x 38 x 2
    variant 4                the solve ran twice (an arm motor row clamped)
  DOF space (variant 0, > 32 contacts or solver_path = 1):
    per iteration            50 x 60 (motor / limit rows: dot with a unit row, 29- or 9-entry response) + 3 nc x 38 x 16
                             (Jacobian entry recomputed 12, J.du 2, du += W dl 2)
The friction rows of unloaded contacts are skipped at run time (about half of the listed contacts): the model counts them, i.e. it is an
UPPER bound of the executed arithmetic by at most the friction share of the unloaded contacts."""
import numpy as np


def pgs_flops(nc, iters, variant):
    nc = np.asarray(nc, dtype=np.float64); it = np.asarray(iters, dtype=np.float64); v = np.asarray(variant)
    n = 32 + 3 * nc
    L = np.where(v == 1, 4.0, 18.0)
    build = 3 * nc * 38 * 10 + n * 3 * nc * 76 + 2 * n * n + n * 38 * 2
    per_it = (32 + L + 3 * nc) * (2 * n + 3)
    row = build + it * per_it * np.where(v == 4, 2.0, 1.0)
    dof = it * (50 * 60 + 3 * nc * 38 * 16)
    return np.where(v == 0, dof, row)


def env_step_flops(nc, iters, variant, counted):
    """mean algorithmic flop per env-step: counted phases (all but PGS) + the PGS model above; `counted` = profiles/flops_latest.json"""
    ph = counted["by_phase_per_env_step"]
    other = sum(val for k, val in ph.items() if k != "pgs")
    pg = pgs_flops(nc, iters, variant)
    return float(other + pg.mean()), float(pg.mean()), float(other)
