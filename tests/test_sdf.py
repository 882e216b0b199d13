"""The native SD / mol2 coordinate readers (`pmx_sdf_heavy_atoms`, `pmx_mol2_heavy_atoms`, csrc/pmx_sdf.cpp; `pharmaconet_amd/sdf.py`): what
`Ligand.load_from_file` (reference `src/pmnet/scoring/ligand.py:63-84`) takes from every record but the first - the heavy
atoms' coordinates - read without a chemistry toolkit. Held to a plain-Python parse of generated V2000 / V3000 files and,
through the OpenBabel stand-in, to the all-toolkit path of `load_from_file` on the described molecules of the perception
fixture."""

import gzip
import json

import numpy as np
import pytest

import fake_openbabel
from conftest import GOLDEN

SYMBOLS = {1: "H", 6: "C", 7: "N", 8: "O", 9: "F", 15: "P", 16: "S", 17: "Cl", 35: "Br", 53: "I"}


def _random_records(rng, n_records, n_atoms, with_h=True):
    z = rng.choice([6, 6, 6, 7, 8, 16, 17, 35, 9, 1 if with_h else 6], size=n_atoms)
    base = rng.normal(size=(n_atoms, 3)) * 8.0
    recs = [np.round(base + rng.normal(size=(n_atoms, 3)) * 0.4, 4) for _ in range(n_records)]
    return [int(x) for x in z], recs


def _v2000(z, recs, eol="\n", aligned=True, data_items=True):
    out = []
    for c, xyz in enumerate(recs):
        lines = [f"name {c}", "  generator", "comment", f"{len(z):3d}{max(len(z) - 1, 0):3d}  0  0  0  0  0  0  0  0999 V2000"]
        for el, (x, y, w) in zip(z, xyz):
            if aligned:
                lines.append(f"{x:10.4f}{y:10.4f}{w:10.4f} {SYMBOLS[el]:<3s} 0  0  0  0  0  0  0  0  0  0  0  0")
            else:
                lines.append(f"{x:.4f} {y:.4f} {w:.4f} {SYMBOLS[el]} 0 0")
        for b in range(len(z) - 1):
            lines.append(f"{b + 1:3d}{b + 2:3d}  1  0")
        lines.append("M  CHG  1   1   1")
        lines.append("M  END")
        if data_items:
            lines += [">  <ENERGY>", "-12.5", "", ">  <NOTE>", "$$$ not a separator", ""]
        lines.append("$$$$")
        out.append(eol.join(lines) + eol)
    return "".join(out).encode()


def _v3000(z, recs):
    out = []
    for c, xyz in enumerate(recs):
        lines = [f"name {c}", "  generator", "", "  0  0  0     0  0            999 V3000", "M  V30 BEGIN CTAB",
                 f"M  V30 COUNTS {len(z)} {max(len(z) - 1, 0)} 0 0 0", "M  V30 BEGIN ATOM"]
        for i, (el, (x, y, w)) in enumerate(zip(z, xyz)):
            lines.append(f"M  V30 {i + 1} {SYMBOLS[el]} {x:.4f} {y:.4f} {w:.4f} 0" + (" CHG=1" if i == 0 else ""))
        lines += ["M  V30 END ATOM", "M  V30 BEGIN BOND"]
        for b in range(len(z) - 1):
            lines.append(f"M  V30 {b + 1} 1 {b + 1} {b + 2}")
        lines += ["M  V30 END BOND", "M  V30 END CTAB", "M  END", "$$$$"]
        out.append("\n".join(lines) + "\n")
    return "".join(out).encode()


def _expect(z, recs):
    heavy = [i for i, el in enumerate(z) if el != 1]
    per = np.full(len(recs), len(heavy), dtype=np.int32)
    zz = np.array([z[i] for i in heavy] * len(recs), dtype=np.uint8)
    xyz = np.concatenate([np.asarray(r, dtype=np.float64)[heavy] for r in recs]).astype(np.float32)
    return per, zz, xyz


@pytest.mark.parametrize("variant", ["v2000", "v2000-crlf", "v2000-free-format", "v2000-no-data", "v3000"])
def test_reader_matches_a_plain_parse(variant):
    from pharmaconet_amd.sdf import read_heavy_atoms

    rng = np.random.default_rng(20250930)
    for trial in range(40):
        z, recs = _random_records(rng, int(rng.integers(1, 9)), int(rng.integers(1, 40)))
        if variant == "v3000":
            text = _v3000(z, recs)
        else:
            text = _v2000(z, recs, eol="\r\n" if variant == "v2000-crlf" else "\n", aligned=variant != "v2000-free-format",
                          data_items=variant != "v2000-no-data")
        per, zz, xyz = read_heavy_atoms(text)
        want = _expect(z, recs)
        assert np.array_equal(per, want[0]) and np.array_equal(zz, want[1])
        assert np.array_equal(xyz, want[2]), f"{variant} trial {trial}"  # doubles parsed, rounded to float32 once
        first = read_heavy_atoms(text, max_records=1)
        assert first[0].tolist() == [want[0][0]] and np.array_equal(first[2], want[2][: want[0][0]])


MOL2_TYPES = {1: ["H", "H.spc", "H.t3p"], 6: ["C.3", "C.2", "C.ar", "C.cat", "C.1"], 7: ["N.3", "N.am", "N.pl3", "N.ar", "N.4"], 8: ["O.3", "O.2", "O.co2"],
              9: ["F"], 15: ["P.3"], 16: ["S.3", "S.o2", "S.O"], 17: ["Cl", "CL"], 35: ["Br"], 53: ["I"]}


def _mol2(z, recs, rng, eol="\n"):
    out = []
    for c, xyz in enumerate(recs):
        lines = ["# a comment", "@<TRIPOS>MOLECULE", f"mol {c}", f" {len(z)} {max(len(z) - 1, 0)} 1 0 0", "SMALL", "GASTEIGER", "", "@<TRIPOS>ATOM"]
        for i, (el, (x, y, w)) in enumerate(zip(z, xyz)):
            t = MOL2_TYPES[el][int(rng.integers(len(MOL2_TYPES[el])))]
            lines.append(f"{i + 1:7d} {SYMBOLS[el]}{i + 1:<5d} {x:10.4f} {y:10.4f} {w:10.4f} {t:<6s} 1 LIG1 {0.01 * i:8.4f}")
        lines.append("@<TRIPOS>BOND")
        for b in range(len(z) - 1):
            lines.append(f"{b + 1:6d}{b + 1:6d}{b + 2:6d}    1")
        lines += ["@<TRIPOS>SUBSTRUCTURE", "     1 LIG1        1 TEMP              0 ****  ****    0 ROOT", ""]
        out.append(eol.join(lines) + eol)
    return "".join(out).encode()


@pytest.mark.parametrize("eol", ["\n", "\r\n"])
def test_mol2_reader_matches_a_plain_parse(eol, tmp_path):
    from pharmaconet_amd.sdf import SdfError, conformer_positions, read_heavy_atoms

    rng = np.random.default_rng(20250931)
    for trial in range(40):
        z, recs = _random_records(rng, int(rng.integers(1, 9)), int(rng.integers(1, 40)))
        text = _mol2(z, recs, rng, eol)
        per, zz, xyz = read_heavy_atoms(text, fmt="mol2")
        want = _expect(z, recs)
        assert np.array_equal(per, want[0]) and np.array_equal(zz, want[1])
        assert np.array_equal(xyz, want[2]), f"trial {trial}"
        first = read_heavy_atoms(text, max_records=1, fmt="mol2")
        assert first[0].tolist() == [want[0][0]] and np.array_equal(first[2], want[2][: want[0][0]])
    path = tmp_path / "confs.mol2"
    path.write_bytes(text)
    zz2, pos = conformer_positions(path)  # the extension picks the reader
    assert pos.shape[1] == len(recs) and zz2.tolist() == want[1][: want[0][0]].tolist()
    # atom types that name no element, a coordinate that is not a number, an atom section outside a record
    lines = text.decode().replace("\r\n", "\n").split("\n")
    k = lines.index("@<TRIPOS>ATOM") + 1
    for bad_type in ("Du", "LP", "Any", "Du.C"):
        bad = list(lines)
        f = bad[k].split()
        f[5] = bad_type
        bad[k] = " ".join(f)
        with pytest.raises(SdfError, match="record 0"):
            read_heavy_atoms("\n".join(bad).encode(), fmt="mol2")
    bad = list(lines)
    f = bad[k].split()
    f[3] = "1.2.3"
    bad[k] = " ".join(f)
    with pytest.raises(SdfError):
        read_heavy_atoms("\n".join(bad).encode(), fmt="mol2")
    with pytest.raises(SdfError):
        read_heavy_atoms(b"@<TRIPOS>ATOM\n 1 C1 0 0 0 C.3\n", fmt="mol2")
    assert read_heavy_atoms(b"", fmt="mol2")[0].size == 0
    with pytest.raises(ValueError):
        read_heavy_atoms(b"", fmt="pdb")


def test_conformers_of_one_molecule_and_what_is_refused(tmp_path):
    from pharmaconet_amd.sdf import SdfError, conformer_positions, read_heavy_atoms

    rng = np.random.default_rng(7)
    z, recs = _random_records(rng, 5, 17)
    path = tmp_path / "confs.sdf"
    path.write_bytes(_v2000(z, recs))
    zz, pos = conformer_positions(path)
    heavy = [i for i, el in enumerate(z) if el != 1]
    assert zz.tolist() == [z[i] for i in heavy] and pos.shape == (len(heavy), 5, 3)
    assert np.array_equal(pos[:, 3, :], np.asarray(recs[3])[heavy].astype(np.float32))
    # a second molecule in the file: not conformers of one molecule
    z2, recs2 = _random_records(rng, 1, 9, with_h=False)
    with pytest.raises(SdfError):
        conformer_positions(_v2000(z, recs) + _v2000(z2, recs2))
    # a truncated atom block, a counts line that is not one, a coordinate that is not a number
    text = _v2000(z, recs).decode().split("\n")
    with pytest.raises(SdfError, match="record 0"):
        read_heavy_atoms("\n".join(text[:8]).encode())
    bad = list(text)
    bad[3] = "not a counts line"
    with pytest.raises(SdfError):
        read_heavy_atoms("\n".join(bad).encode())
    bad = list(text)
    bad[5] = bad[5][:10] + "   abc.def" + bad[5][20:]
    with pytest.raises(SdfError):
        read_heavy_atoms("\n".join(bad).encode())
    assert read_heavy_atoms(b"")[0].size == 0 and read_heavy_atoms(b"\n\n")[0].size == 0


@pytest.fixture()
def _openbabel_stand_in():
    import sys

    fake_openbabel.install()
    yield
    for name in ("openbabel", "openbabel.pybel", "openbabel.pybel.ob"):
        if getattr(sys.modules.get(name), "__fake__", False) or name != "openbabel":
            sys.modules.pop(name, None)


def test_load_from_file_reads_sd_coordinates_natively(tmp_path, monkeypatch, _openbabel_stand_in):
    """`Ligand.load_from_file` on real SD files of the described molecules (hydrogens mixed into the atom blocks, LF and CRLF):
    the same packed record as the all-toolkit path on the same molecule, with the toolkit asked for ONE record only."""
    from pharmaconet_amd import ligand as ligand_mod
    from pharmaconet_amd.library import pack_ligand
    from pharmaconet_amd.ligand import Ligand

    with gzip.open(GOLDEN / "perception.json.gz", "rt") as f:
        molecules = json.load(f)["molecules"]
    asked = []
    real_readfile = fake_openbabel.readfile

    def counting_readfile(fmt, filename):
        for k, mol in enumerate(real_readfile(fmt, filename)):
            asked.append(k)
            yield mol

    monkeypatch.setattr(sys_pybel(), "readfile", counting_readfile)
    checked = 0
    for i, desc in enumerate(molecules[:120]):
        desc = dict(desc)
        desc["coords"] = np.round(np.asarray(desc["coords"], dtype=np.float64), 4).tolist()  # what an SD file can hold
        sdf = tmp_path / f"m{i}.sdf"
        fake_openbabel.write_sdf(desc, sdf, extra_hydrogens=i % 4, crlf=i % 5 == 0)
        del asked[:]
        fast = Ligand.load_from_file(sdf)
        assert asked == [0], "the toolkit parses the first record only"
        js = tmp_path / f"m{i}.json.sdf"
        js.write_text(json.dumps(desc))  # not an SD file: the reference's way (one toolkit molecule per record)
        slow = Ligand.load_from_file(js)
        assert fast.num_conformers == slow.num_conformers == len(desc["coords"])
        assert np.array_equal(fast.atom_positions, slow.atom_positions)
        assert bytes(pack_ligand(fast.features)) == bytes(pack_ligand(slow.features))
        if len(desc["coords"]) > 1:
            assert Ligand.load_from_file(sdf, num_conformers=1).num_conformers == 1
        if i % 3 == 0:  # the same molecule as a multi-record mol2 file
            mol2 = tmp_path / f"m{i}.mol2"
            fake_openbabel.write_mol2(desc, mol2, extra_hydrogens=i % 4, crlf=i % 5 == 0)
            del asked[:]
            fast2 = Ligand.load_from_file(mol2)
            assert asked == [0]
            assert np.array_equal(fast2.atom_positions, slow.atom_positions)
            assert bytes(pack_ligand(fast2.features)) == bytes(pack_ligand(slow.features))
        checked += 1
    assert checked == 120
    assert ligand_mod is not None


def sys_pybel():
    import sys

    return sys.modules["openbabel.pybel"]
