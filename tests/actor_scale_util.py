"""Helpers for the `actor_scale` tensor of Ant / Humanoid (csrc/core/engine.hpp AS_*): [N, NB + 3 ND] -- one factor per BODY for mass and
inertia, then one per DOF for the joint's damping, stiffness and armature -- and the model a set of factors amounts to."""
import dataclasses

import numpy as np


def factors(spec, rng, mass=(0.6, 1.7), damping=(0.5, 1.5), stiffness=(0.5, 2.0), armature=(0.5, 3.0)):
    """-> dict of per-element factor arrays: a different factor for every body and every dof"""
    return dict(mass=rng.uniform(*mass, spec.nb), damping=rng.uniform(*damping, spec.nd), stiffness=rng.uniform(*stiffness, spec.nd),
                armature=rng.uniform(*armature, spec.nd))


def row(spec, f):
    """the tensor row [NB + 3 ND] of a factor set"""
    return np.concatenate([np.broadcast_to(f["mass"], spec.nb), np.broadcast_to(f["damping"], spec.nd), np.broadcast_to(f["stiffness"], spec.nd),
                           np.broadcast_to(f["armature"], spec.nd)]).astype(np.float32)


def rescaled(spec, f, **more):
    """the ModelSpec whose constants are the model's times the factors"""
    m = np.broadcast_to(np.asarray(f["mass"], float), spec.nb)
    return dataclasses.replace(spec, mass=spec.mass * m, inertia=spec.inertia * m[:, None], dof_damping=spec.dof_damping * f["damping"],
                               dof_stiffness=spec.dof_stiffness * f["stiffness"], dof_armature=spec.dof_armature * f["armature"], **more)
