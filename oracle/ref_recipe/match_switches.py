#!/usr/bin/env python3
"""Which setting of the ASSUMPTIONS.md [A] switches reproduces a set of REFERENCE vectors?  (test infrastructure)

The arithmetic of hydra::ProjectiveIntegrator / MeshIntegrator is not in the Khronos checkout; where the recalled upstream lineage
leaves two readings, both are implemented behind khr_config / orc_config switches (include/khronos_amd.h):

    alloc_candidate      0 block centre in the inflated frustum | 1 candidate point camera_W + offset * block_size
    color_blend_weight   0 voxel weight after the update        | 1 before it
    mesh_attr_source     0 nearer endpoint voxel                | 1 voxel that contains the vertex
    mesh_degenerate_eps  0 (= 1e-6)                             | any other epsilon given with --eps

This script takes vectors written by the real upstream code (oracle/ref_recipe/build.sh -> oracle/_ref/ref_small.npz, the keys of
tests/golden/make_golden.py), runs the CPU restatement on the same seeded frames in every setting and reports, per switch, which
value reproduces the layer that switch acts on:

    alloc_candidate      <- block_indices (the set of allocated blocks)
    color_blend_weight   <- color (given the matching block set)
    mesh_degenerate_eps  <- mesh_vertices, mesh_checksum
    mesh_attr_source     <- mesh_color_checksum, mesh_label_checksum, mesh_stamp_checksum

plus whether EVERYTHING else (distance, weight, flags, labels, stamps, motion clusters, archival counts) matches in the best setting.
The report goes to stdout and to <vectors>.switches.json; a maintainer then sets the named values in the YAML of the drop-in
(projective_integrator.alloc_candidate / color_blend_weight, mesh_integrator.attr_source / degenerate_eps).

    python oracle/ref_recipe/match_switches.py [oracle/_ref/ref_small.npz] [--eps 1e-3 ...]
"""
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

LAYER_OF = {
    "alloc_candidate": ("block_indices",),
    "color_blend_weight": ("color",),
    "mesh_degenerate_eps": ("mesh_vertices", "mesh_checksum"),
    "mesh_attr_source": ("mesh_color_checksum", "mesh_label_checksum", "mesh_stamp_checksum"),
}
OTHER = ("distance", "weight", "flags", "sem_label", "last_observed", "n_clusters", "dyn_pixels", "removed_counts")


def same(ref, got, key):
    if key not in ref.files if hasattr(ref, "files") else key not in ref:
        return None  # the vectors do not carry this key (older dump_vectors.cpp)
    a, b = np.asarray(ref[key]), np.asarray(got[key])
    if a.shape != b.shape:
        return False
    if key == "block_indices":
        return bool(np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])]))
    if a.dtype.kind == "f":
        return bool(np.allclose(a, b, rtol=0, atol=1e-6 * max(1.0, float(np.abs(a).max()) if a.size else 1.0)))
    return bool(np.array_equal(a, b))


def match(ref, eps_values=(0.0,), log=print):
    import make_golden
    runs = {}
    for alloc, blend, attr, eps in itertools.product((0, 1), (0, 1), (0, 1), eps_values):
        runs[(alloc, blend, attr, eps)] = make_golden.run(dict(alloc_candidate=alloc, color_blend_weight=blend, mesh_attr_source=attr,
                                                               mesh_degenerate_eps=eps))
    names = ("alloc_candidate", "color_blend_weight", "mesh_attr_source", "mesh_degenerate_eps")
    report = {"switches": {}, "layers_checked": {}}
    best = {}
    for pos, name in enumerate(names):
        values = sorted({k[pos] for k in runs})
        ok = {}
        for v in values:
            # a value "reproduces" its layer when SOME setting of the other switches does (the layers are independent given the ones
            # decided before: blocks first, then colour, then the mesh)
            cands = [k for k in runs if k[pos] == v and all(k[p] == best[names[p]] for p in range(pos) if names[p] in best)]
            res = [[same(ref, runs[k], key) for key in LAYER_OF[name]] for k in cands]
            known = [[r for r in row if r is not None] for row in res]
            ok[v] = None if not any(known) else any(all(row) for row in known if row)
        matching = [v for v in values if ok[v]]
        if all(x is None for x in ok.values()):
            verdict = "the vectors carry none of %s: cannot tell" % (LAYER_OF[name],)
        elif len(matching) == 1:
            verdict = "%r reproduces the reference" % (matching[0],)
            best[name] = matching[0]
        elif len(matching) > 1:
            verdict = "values %r all reproduce the reference on this sequence: indistinguishable here" % (matching,)
            best[name] = matching[0]
        else:
            verdict = "NO value reproduces the reference: the restatement of this step differs from upstream beyond the switch"
        report["switches"][name] = {"layer": list(LAYER_OF[name]), "reproduces": {str(v): ok[v] for v in values}, "verdict": verdict}
        log("%-20s %s" % (name, verdict))
    key = tuple(best.get(n, 0) for n in names)
    if key in runs:
        rest = {k: same(ref, runs[key], k) for k in OTHER}
        report["layers_checked"] = rest
        report["best_setting"] = dict(zip(names, key))
        bad = [k for k, v in rest.items() if v is False]
        log("everything else in the best setting %r: %s" % (dict(zip(names, key)), "matches" if not bad else "DIFFERS in %s" % bad))
    return report


def main(argv):
    path = os.path.join(ROOT, "oracle", "_ref", "ref_small.npz")
    eps = [0.0]
    args = list(argv)
    while args:
        a = args.pop(0)
        if a == "--eps":
            eps.append(float(args.pop(0)))
        else:
            path = a
    if not os.path.exists(path):
        raise SystemExit("no reference vectors at %s: run oracle/ref_recipe/build.sh with the upstream checkouts first" % path)
    ref = np.load(path)
    report = match(ref, tuple(eps))
    out = path + ".switches.json"
    with open(out, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1:])
