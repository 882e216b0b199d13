"""Shared vocabulary of the screening hot path.

The seven pharmacophore types and their default weights follow the reference
(`src/pmnet/scoring/ligand_utils.py:80-88` for the order in which a ligand's
features are listed, `src/pmnet/scoring/graph_match.py:32-40` for the weights).
The numeric ids below are this framework's wire format: they index the 7-bit
type masks of the packed ligand library and the `weights[7]` kernel argument.
"""

from __future__ import annotations

# id -> name; the order is the reference's pharmacophore_list order.
TYPE_NAMES: tuple[str, ...] = (
    "Hydrophobic",
    "Aromatic",
    "Cation",
    "Anion",
    "HBond_donor",
    "HBond_acceptor",
    "Halogen",
)
TYPE_ID: dict[str, int] = {name: i for i, name in enumerate(TYPE_NAMES)}
NUM_TYPES = 7

# graph_match.py:32-40
DEFAULT_WEIGHTS: dict[str, float] = dict(
    Cation=8,
    Anion=8,
    Aromatic=4,
    HBond_donor=4,
    HBond_acceptor=4,
    Halogen=4,
    Hydrophobic=1,
)

# Ligand cluster types (ligand.py:117-119) -> (group, subtype) of priority_fn
# (graph_match.py:43-60).
CLUSTER_PRIORITY: dict[str, tuple[int, int]] = {
    "Aromatic": (0, 0),
    "Cation": (0, 1),
    "Anion": (0, 2),
    "HBond": (1, 0),
    "Halogen": (1, 1),
    "Hydrophobic": (1, 2),
}

# graph_match.py:88  -- "MAX DEPTH: 20"
MAX_LEVELS = 20

# Device-side structural limits of this implementation (not of the reference).
MAX_MODEL_NODES = 256  # include/pmx.h PMX_MAX_MODEL_NODES (node sets are lists on the device)
MAX_MODEL_CLUSTERS = 128  # PMX_MAX_MODEL_CLUSTERS (candidate sets of a ligand cluster are two 64-bit words)
MAX_LIGAND_NODES = 64
MAX_LIGAND_CLUSTERS = 64
MAX_CONFORMERS = 64  # one wavefront lane per conformer


def weights_vector(weights: dict[str, float] | None = None) -> list[float]:
    """DEFAULT_WEIGHTS updated by `weights` (graph_match.py:81-83) as a 7-vector in TYPE_NAMES order."""
    w = dict(DEFAULT_WEIGHTS)
    if weights is not None:
        w.update(weights)  # unknown keys are carried and never read, as in the reference
    return [float(w[name]) for name in TYPE_NAMES]
