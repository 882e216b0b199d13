#!/usr/bin/env python3
"""Mint the golden set of a model beyond 64 nodes / 64 clusters by running the REFERENCE itself (build container only; see
make_golden.py, whose model and ligand generators this script uses):

    python tests/golden/make_golden_large.py

model_large110.pm: 110 hotspots spread over a 22 A box -> more than 64 model nodes and more than 64 node clusters (the sizes the
engine's 64-bit node and candidate sets of rounds 1-3 refused); set_l110_c8: 48 ligands x 8 conformers with the reference's scores."""
import make_golden as mg

model = mg.model_random(mg.SEED + 40, 110, 11.0, "SYNTHETIC LARGE-110", min_sep=1.2)
model.save(mg.HERE / "model_large110.pm")
print(f"model_large110: {len(model.nodes)} nodes, {len(model.edges)} edges, {len(model.node_clusters)} clusters")
mg.make_set("set_l110_c8", model, "model_large110", 48, 8, mg.SEED + 41, big=1)
