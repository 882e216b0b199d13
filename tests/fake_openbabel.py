"""A stand-in for the parts of OpenBabel's Python API that ligand perception touches. TEST INFRASTRUCTURE.

OpenBabel is not installed in the build image, so neither the reference's perception
(`/root/reference/src/pmnet/scoring/ligand_utils.py:25-184`, `scoring/ligand.py:16-84`) nor this repository's
(`pharmaconet_amd/ligand.py`) can run on real molecules here. What CAN be pinned is the rule logic on top of OpenBabel's
atom predicates: this module implements `openbabel.pybel` / `openbabel.pybel.ob` for *described* molecules - a heavy-atom
graph whose chemistry-toolkit answers (`IsHbondAcceptor`, `IsHbondDonor`, `GetHyb`, ring aromaticity, rotor count) are
given, while everything that follows from the graph (`GetAtomicNum`, degrees, neighbour iteration, `GetIdx`,
`AddPolarHydrogens`, `removeh`, `clone`, `sssr`) is derived from it. `tests/golden/make_golden_perception.py` feeds the
same described molecules to the reference's `get_pharmacophore_nodes` / `Ligand` and the tests feed them to
`pharmaconet_amd.ligand`; both import this module as `openbabel`.

A molecule description is a plain dict (what the fixtures store):
    z          [n]      atomic numbers of the heavy atoms
    bonds      [m][2]   heavy-atom bonds (0-based)
    acceptor   [n]      OBAtom.IsHbondAcceptor()
    donor      [n]      the atom carries polar hydrogens: IsHbondDonor() once they are explicit
    hyb        [n]      OBAtom.GetHyb()
    rings      [[atom indices...], aromatic]   the SSSR
    rotors     int      OBMol.NumRotors()
    coords     [C][n][3] conformer coordinates
"""

from __future__ import annotations

import copy
import json
import sys
import types


class FakeAtom:
    def __init__(self, mol, idx0, z):
        self._mol = mol
        self._i = idx0
        self._z = z

    def GetAtomicNum(self):
        return self._z

    def GetIdx(self):  # OpenBabel atom indices start at 1
        return self._i + 1

    def _nbrs(self):
        return [self._mol._atoms[j] for j in self._mol._adj[self._i]]

    def IsHbondAcceptor(self):
        return bool(self._mol._acceptor[self._i]) if self._i < self._mol._n_heavy else False

    def IsHbondDonor(self):  # needs the hydrogens to be there, as in OpenBabel
        if self._i >= self._mol._n_heavy or not self._mol._donor[self._i]:
            return False
        return any(a.GetAtomicNum() == 1 for a in self._nbrs())

    def GetExplicitDegree(self):
        return len(self._mol._adj[self._i])

    def GetHvyDegree(self):
        return sum(1 for a in self._nbrs() if a.GetAtomicNum() != 1)

    def GetHyb(self):
        return int(self._mol._hyb[self._i]) if self._i < self._mol._n_heavy else 0


class FakeOBMol:
    def __init__(self, desc):
        n = len(desc["z"])
        self._n_heavy = n
        self._acceptor = list(desc["acceptor"])
        self._donor = list(desc["donor"])
        self._hyb = list(desc["hyb"])
        self._rotors = int(desc.get("rotors", 0))
        self._adj = [[] for _ in range(n)]
        for a, b in desc["bonds"]:
            self._adj[a].append(b)
            self._adj[b].append(a)
        self._atoms = [FakeAtom(self, i, int(z)) for i, z in enumerate(desc["z"])]
        # explicit hydrogens the description may carry (removeh() drops them): one per donor when desc["explicit_h"]
        if desc.get("explicit_h"):
            self.AddPolarHydrogens()

    def NumAtoms(self):
        return len(self._atoms)

    def GetAtom(self, idx1):
        return self._atoms[idx1 - 1]

    def NumRotors(self):
        return self._rotors

    def AddPolarHydrogens(self):
        for i in range(self._n_heavy):
            if self._donor[i] and not any(self._atoms[j].GetAtomicNum() == 1 for j in self._adj[i]):
                h = len(self._atoms)
                self._atoms.append(FakeAtom(self, h, 1))
                self._adj.append([i])
                self._adj[i].append(h)
        return True

    def _remove_h(self):
        keep = self._n_heavy
        self._atoms = self._atoms[:keep]
        self._adj = [[j for j in nb if j < keep] for nb in self._adj[:keep]]


class FakeRing:
    def __init__(self, path0, aromatic):
        self._path = tuple(i + 1 for i in path0)  # 1-based, as pybel's
        self._aromatic = bool(aromatic)

    def IsAromatic(self):
        return self._aromatic


class _PybelAtom:
    def __init__(self, coords):
        self.coords = tuple(float(x) for x in coords)


class Molecule:
    """pybel.Molecule of one conformer of a described molecule."""

    def __init__(self, desc, conformer=0):
        self._desc = desc
        self._conf = conformer
        self.OBMol = FakeOBMol(desc)

    @property
    def clone(self):
        m = Molecule.__new__(Molecule)
        m._desc = self._desc
        m._conf = self._conf
        m.OBMol = copy.deepcopy(self.OBMol)
        for a in m.OBMol._atoms:
            a._mol = m.OBMol
        return m

    def removeh(self):
        self.OBMol._remove_h()

    @property
    def sssr(self):
        return [FakeRing(path, arom) for path, arom in self._desc["rings"]]

    @property
    def atoms(self):
        xyz = self._desc["coords"][self._conf]
        heavy = self.OBMol._n_heavy
        # (a hydrogen sits on its heavy atom: only heavy-atom coordinates are described)
        return [_PybelAtom(xyz[a._i] if a._i < heavy else xyz[self.OBMol._adj[a._i][0]]) for a in self.OBMol._atoms]


SDF_DESCRIPTION_TAG = "pmx_description"


def write_sdf(desc, path, extra_hydrogens=0, crlf=False):
    """The described molecule as a real multi-record SD file (V2000): one record per conformer, heavy atoms in description
    order with `extra_hydrogens` explicit hydrogens mixed into every atom block, and - as a data item of every record - the
    description itself, which is all this stand-in can perceive chemistry from (`readfile` below). The atom blocks are what
    pharmaconet_amd's native SD reader takes the coordinates from. Coordinates are written with four decimals, as SD files
    hold them."""
    eol = "\r\n" if crlf else "\n"
    sym = {1: "H", 5: "B", 6: "C", 7: "N", 8: "O", 9: "F", 15: "P", 16: "S", 17: "Cl", 35: "Br", 53: "I"}
    out = []
    n = len(desc["z"])
    for c, xyz in enumerate(desc["coords"]):
        lines = [f"mol conformer {c}", "  fake_openbabel", ""]
        atoms = [(sym[int(z)], xyz[i]) for i, z in enumerate(desc["z"])]
        for h in range(extra_hydrogens):  # hydrogens anywhere in the block: removeh() / the native reader drop them
            at = (h * 7 + c) % (len(atoms) + 1)
            x, y, zc = xyz[(h * 3) % n]
            atoms.insert(at, ("H", (x + 0.6, y - 0.4, zc + 0.3)))
        bonds = desc["bonds"]
        lines.append(f"{len(atoms):3d}{0:3d}  0  0  0  0  0  0  0  0999 V2000")
        for s_, (x, y, zc) in atoms:
            lines.append(f"{x:10.4f}{y:10.4f}{zc:10.4f} {s_:<3s} 0  0  0  0  0  0  0  0  0  0  0  0")
        lines.append("M  END")
        lines.append(f">  <{SDF_DESCRIPTION_TAG}>")
        lines.append(json.dumps(desc))
        lines.append("")
        lines.append("$$$$")
        out.append(eol.join(lines) + eol)
    with open(path, "w", newline="") as f:
        f.write("".join(out))


def write_mol2(desc, path, extra_hydrogens=0, crlf=False):
    """The described molecule as a multi-record Tripos mol2 file: one @<TRIPOS>MOLECULE record per conformer, heavy atoms in
    description order with `extra_hydrogens` hydrogens mixed into every atom section, SYBYL-style atom types (`C.3`, `N.ar`,
    `Cl` ...), and the description as an @<TRIPOS>COMMENT section of every record (what `readfile` perceives chemistry from)."""
    eol = "\r\n" if crlf else "\n"
    typ = {1: "H", 5: "B", 6: "C.3", 7: "N.am", 8: "O.co2", 9: "F", 15: "P.3", 16: "S.o2", 17: "Cl", 35: "Br", 53: "I"}
    out = []
    n = len(desc["z"])
    for c, xyz in enumerate(desc["coords"]):
        atoms = [(typ[int(z)], xyz[i]) for i, z in enumerate(desc["z"])]
        for h in range(extra_hydrogens):
            at = (h * 7 + c) % (len(atoms) + 1)
            x, y, zc = xyz[(h * 3) % n]
            atoms.insert(at, ("H" if h % 2 else "H.spc", (x + 0.6, y - 0.4, zc + 0.3)))
        lines = ["# written by fake_openbabel", "@<TRIPOS>MOLECULE", f"conformer {c}", f" {len(atoms)} 0 1 0 0", "SMALL", "NO_CHARGES", "",
                 "@<TRIPOS>ATOM"]
        for k, (t, (x, y, zc)) in enumerate(atoms):
            lines.append(f"{k + 1:7d} {t.split('.')[0]}{k + 1:<6d} {x:10.4f} {y:10.4f} {zc:10.4f} {t:<7s} 1  LIG1  0.0000")
        lines += ["@<TRIPOS>BOND", "@<TRIPOS>COMMENT", f"{SDF_DESCRIPTION_TAG} " + json.dumps(desc), ""]
        out.append(eol.join(lines) + eol)
    with open(path, "w", newline="") as f:
        f.write("".join(out))


def readfile(fmt, filename):
    """A 'file' is the JSON of a description - every conformer becomes one record, as in a multi-record SDF - or a real SD
    file written by `write_sdf`, whose records carry the description as a data item (the stand-in perceives nothing from
    atom blocks; it reads the record count and the description)."""
    with open(filename) as f:
        text = f.read()
    try:
        desc = json.loads(text)
    except json.JSONDecodeError:
        if "@<TRIPOS>MOLECULE" in text:  # a mol2 file written by `write_mol2`
            records = text.replace("\r\n", "\n").split("@<TRIPOS>MOLECULE")[1:]
            desc = json.loads(records[0].split(SDF_DESCRIPTION_TAG + " ", 1)[1].split("\n", 1)[0])
            for c in range(len(records)):
                yield Molecule(desc, c)
            return
        records = [r for r in text.replace("\r\n", "\n").split("$$$$\n") if r.strip()]
        tag = f">  <{SDF_DESCRIPTION_TAG}>\n"
        desc = json.loads(records[0].split(tag, 1)[1].split("\n", 1)[0])
        for c in range(len(records)):
            yield Molecule(desc, c)
        return
    for c in range(len(desc["coords"])):
        yield Molecule(desc, c)


class _Permissive(types.ModuleType):
    """Names the stand-in does not implement resolve to a placeholder class: other modules of the reference only use
    them in annotations that are evaluated at import time (e.g. `ob.OBResidue`)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        placeholder = _Placeholder(name)
        setattr(self, name, placeholder)
        return placeholder


class _Placeholder:
    """Callable, attribute-bearing nothing (`ob.obErrorLog.StopLogging()` at import time of the reference's data package)."""

    def __init__(self, name):
        self.__name__ = name

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Placeholder(name)

    def __or__(self, other):  # `X | None` in annotations
        return self

    __ror__ = __or__


def _module():
    ob = _Permissive("openbabel.pybel.ob")
    ob.OBMolAtomIter = lambda obmol: iter(list(obmol._atoms))
    ob.OBAtomAtomIter = lambda atom: iter(atom._nbrs())
    ob.OBAtom = FakeAtom
    ob.OBMol = FakeOBMol
    pybel = _Permissive("openbabel.pybel")
    pybel.ob = ob
    pybel.Molecule = Molecule
    pybel.readfile = readfile
    pybel.__fake__ = True
    ob.__fake__ = True
    top = _Permissive("openbabel")
    top.pybel = pybel
    top.__fake__ = True
    return top, pybel, ob


def install():
    """Make `import openbabel` resolve to this stand-in (only when the real one is absent)."""
    try:
        import openbabel  # noqa: F401

        if not getattr(sys.modules["openbabel"], "__fake__", False):
            raise RuntimeError("a real OpenBabel is installed: the stand-in is for images without it")
        return sys.modules["openbabel"]
    except ImportError:
        pass
    top, pybel, ob = _module()
    sys.modules["openbabel"] = top
    sys.modules["openbabel.pybel"] = pybel
    sys.modules["openbabel.pybel.ob"] = ob
    return top


# ------------------------------------------------------------------------------------------- described molecules
def random_description(rng, n_conformers=3, explicit_h=False):
    """A random heavy-atom graph decorated so that every branch of ligand_utils.py:36-184 is reached over a few hundred
    draws: chains and rings of C / N / O / S, halogens on carbon and on heteroatoms, and the charged groups (quaternary and
    tertiary amines, sulfonium, guanidine, phosphate, sulfate, sulfonic acid, carboxylate) as attached motifs, plus
    near-misses of each (a 'guanidine' carbon with a fourth neighbour, a 'phosphate' with a carbon on it ...)."""
    import numpy as np

    z, bonds = [], []

    def atom(zz):
        z.append(int(zz))
        return len(z) - 1

    def bond(a, b):
        bonds.append([int(a), int(b)])

    rings = []
    # backbone
    n_back = int(rng.integers(3, 14))
    back = [atom(rng.choice([6, 6, 6, 6, 7, 8, 16]))]
    for _ in range(n_back - 1):
        a = atom(rng.choice([6, 6, 6, 6, 7, 8, 16]))
        bond(back[int(rng.integers(0, len(back)))], a)
        back.append(a)
    # rings (five or six members, some aromatic), fused onto backbone atoms
    for _ in range(int(rng.integers(0, 4))):
        size = int(rng.choice([5, 6, 6]))
        members = [atom(rng.choice([6, 6, 6, 7])) for _ in range(size)]
        for i in range(size):
            bond(members[i], members[(i + 1) % size])
        bond(members[0], back[int(rng.integers(0, len(back)))])
        rings.append([members, bool(rng.random() < 0.7)])
    carbons = [i for i, zz in enumerate(z) if zz == 6]

    def anchor():
        return carbons[int(rng.integers(0, len(carbons)))] if carbons and rng.random() < 0.8 else int(rng.integers(0, len(z)))

    for _ in range(int(rng.integers(0, 6))):  # charged groups and their near-misses
        kind = int(rng.integers(0, 10))
        base = anchor()
        if kind == 0:  # quaternary ammonium / or three residues only
            n = atom(7)
            bond(base, n)
            for _ in range(int(rng.choice([2, 3]))):
                bond(n, atom(6))
        elif kind == 1:  # tertiary amine (hyb 3 drawn below) with three heavy neighbours
            n = atom(7)
            bond(base, n)
            bond(n, atom(6))
            bond(n, atom(6))
        elif kind == 2:  # sulfonium or thioether
            s_ = atom(16)
            bond(base, s_)
            for _ in range(int(rng.choice([1, 2]))):
                bond(s_, atom(6))
        elif kind == 3:  # guanidine: C(N)(N)N with a terminal N; sometimes a fourth neighbour or an O among them
            c = atom(6)
            ns = [atom(7 if rng.random() < 0.9 else 8) for _ in range(3)]
            for n in ns:
                bond(c, n)
            bond(ns[0], base)
            if rng.random() < 0.3:
                bond(ns[1], atom(6))
            if rng.random() < 0.15:
                bond(c, atom(6))
        elif kind == 4:  # phosphate, or a phosphonate (carbon on P)
            p_ = atom(15)
            o = atom(8)
            bond(base, o)
            bond(o, p_)
            for _ in range(3):
                bond(p_, atom(8 if rng.random() < 0.9 else 6))
        elif kind == 5:  # sulfate (4 O) / sulfonic acid (3 O) / sulfone (2 O)
            s_ = atom(16)
            n_o = int(rng.choice([2, 3, 4]))
            if n_o == 4:
                o = atom(8)
                bond(base, o)
                bond(o, s_)
                for _ in range(3):
                    bond(s_, atom(8))
            else:
                bond(base, s_)
                for _ in range(n_o):
                    bond(s_, atom(8))
        elif kind == 6:  # carboxylate, ester-like (O bound further) or amide
            c = atom(6)
            bond(base, c)
            bond(c, atom(8))
            second = atom(8 if rng.random() < 0.8 else 7)
            bond(c, second)
            if rng.random() < 0.3:
                bond(second, atom(6))
        elif kind == 7:  # halogens on carbon and on a heteroatom
            bond(base, atom(rng.choice([9, 17, 35, 53])))
        elif kind == 8:
            n = atom(7)
            bond(base, n)
            bond(n, atom(rng.choice([9, 17])))
        else:  # hydroxyl / amine
            bond(base, atom(rng.choice([7, 8])))
    n = len(z)
    zarr = np.array(z)
    polar = np.isin(zarr, (7, 8, 16))
    acceptor = [bool(polar[i] and rng.random() < 0.7) or bool(zarr[i] in (9, 17) and rng.random() < 0.5) for i in range(n)]
    donor = [bool(zarr[i] in (7, 8) and rng.random() < 0.5) for i in range(n)]
    hyb = [int(rng.choice([1, 2, 3, 3])) for _ in range(n)]
    # coordinates: a random walk over the bond graph plus conformer noise
    base_xyz = np.zeros((n, 3))
    placed = {0}
    order = [0]
    adj = [[] for _ in range(n)]
    for a, b in bonds:
        adj[a].append(b)
        adj[b].append(a)
    while order:
        a = order.pop()
        for b in adj[a]:
            if b not in placed:
                step = rng.normal(size=3)
                base_xyz[b] = base_xyz[a] + 1.5 * step / np.linalg.norm(step)
                placed.add(b)
                order.append(b)
    coords = [(base_xyz + rng.normal(scale=0.25, size=(n, 3))).astype(np.float32).tolist() for _ in range(n_conformers)]
    return dict(z=z, bonds=bonds, acceptor=acceptor, donor=donor, hyb=hyb, rings=[[list(map(int, m)), a] for m, a in rings],
                rotors=int(rng.integers(0, 12)), coords=coords, explicit_h=bool(explicit_h))
