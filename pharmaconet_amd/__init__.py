"""MI355X-native screening hot path of PharmacoNet (pharmacophore <-> ligand graph matching and scoring).

Public surface mirrors `pmnet` for this path only: `PharmacophoreModel` with
`load / save / scoring_file / scoring_smiles / scoring_pbmol / _scoring`, plus the batched
`PharmacophoreModel.screen` and the packed ligand library it consumes. The scoring itself lives in
`pharmaconet_amd/csrc` (HIP, gfx950) behind the C ABI of `include/pmx.h`.
"""

from .constants import DEFAULT_WEIGHTS, TYPE_NAMES
from .library import LigandFeatures, PackedLibrary, pack_ligand
from .pharmacophore_model import PharmacophoreModel

__version__ = "0.1.0"
__all__ = ["PharmacophoreModel", "PackedLibrary", "LigandFeatures", "pack_ligand", "DEFAULT_WEIGHTS", "TYPE_NAMES"]
