#!/usr/bin/env python3
"""the single-cloud entry point (the reference's grid_subsampling.compute) on clouds of very different densities: the sort formulation is linear in the points"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib, subsampling, synthetic
_lib.check(_lib.lib().ssdr_init(0))
for dens in (5000.0, 60000.0, 150000.0, 400000.0):
    p, c, l = synthetic.make_room(5003, density=dens)[:3]
    for dl in (0.04, 0.01):
        subsampling.compute(p, features=c, classes=l, sampleDl=dl)
        t0 = time.perf_counter(); out = subsampling.compute(p, features=c, classes=l, sampleDl=dl); dt = time.perf_counter() - t0
        print("density %7.0f: %5.1f M points, grid %.2f -> %7d voxels (%.0f per voxel): %.1f ms host to host (%.0f ps per point), device %.2f ms" % (dens, len(p) / 1e6, dl, len(out[0]), len(p) / len(out[0]), dt * 1e3, dt * 1e12 / len(p), _lib.last_gpu_ms()))
