// pmx_sdf.cpp - conformer coordinates of an SD file (MDL molfile V2000 / V3000 records separated by $$$$) or a Tripos mol2 file
// (@<TRIPOS>MOLECULE records) in native code.
//
// What it stands in for: the coordinate half of Ligand.load_from_file (src/pmnet/scoring/ligand.py:63-84) - the reference has
// OpenBabel parse every record of a multi-conformer file into a molecule object and then copies `[atom.coords for atom in
// pbmol.atoms]` in a Python loop, after `removeh()`. The perceived pharmacophore features come from the first record alone
// (ligand.py:78-84: `cls(base_pbmol, atom_positions)`), so for every further record only the heavy-atom coordinates are
// needed: this reader delivers them - element and position of every atom that is not a hydrogen (H, D, T: what
// OBMol::DeleteHydrogens removes), in file order, one block per record - without building molecule objects. Coordinates are
// parsed as doubles and rounded to float32 once, as `np.stack(..., dtype=np.float32)` does with OpenBabel's doubles.
// Host code only (part of libpmx.so and of the host-only libpmx_pack.so).
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "pmx.h"

#ifdef PMX_PACK_STANDALONE
static int pmx_topk_fail(int code, const char *) { return code; } // (libpmx_pack.so keeps its message in pmx_pack.cpp)
#else
int pmx_topk_fail(int code, const char *msg); // error hook in pmx_api.hip
#endif

namespace {

const char *const kSymbols[] = {"", "H", "He", "Li", "Be", "B", "C", "N", "O", "F", "Ne", "Na", "Mg", "Al", "Si", "P", "S", "Cl", "Ar", "K", "Ca", "Sc", "Ti", "V",
                                "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn", "Ga", "Ge", "As", "Se", "Br", "Kr", "Rb", "Sr", "Y", "Zr", "Nb", "Mo", "Tc", "Ru", "Rh",
                                "Pd", "Ag", "Cd", "In", "Sn", "Sb", "Te", "I", "Xe", "Cs", "Ba", "La", "Ce", "Pr", "Nd", "Pm", "Sm", "Eu", "Gd", "Tb", "Dy", "Ho",
                                "Er", "Tm", "Yb", "Lu", "Hf", "Ta", "W", "Re", "Os", "Ir", "Pt", "Au", "Hg", "Tl", "Pb", "Bi", "Po", "At", "Rn"};

int atomic_number(const char *s, size_t n) {
    while (n && std::isspace((unsigned char)*s)) ++s, --n;
    while (n && std::isspace((unsigned char)s[n - 1])) --n;
    if (n == 0 || n > 2) return n == 0 ? -1 : 0; // (pseudo atoms - R#, A, Q, L ... longer names: element 0, kept)
    char a = (char)std::toupper((unsigned char)s[0]), b = n > 1 ? (char)std::tolower((unsigned char)s[1]) : 0;
    if (n == 1 && (a == 'D' || a == 'T')) return 1; // hydrogen isotopes
    for (int z = 1; z < (int)(sizeof(kSymbols) / sizeof(kSymbols[0])); ++z)
        if (kSymbols[z][0] == a && kSymbols[z][1] == b) return z;
    return 0;
}

struct Line {
    const char *p;
    size_t n; // without the line end
};
struct Cursor {
    const char *p, *end;
    bool next(Line &l) {
        if (p >= end) return false;
        const char *q = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
        const char *stop = q ? q : end;
        l.p = p;
        l.n = (size_t)(stop - p);
        if (l.n && l.p[l.n - 1] == '\r') --l.n;
        p = q ? q + 1 : end;
        return true;
    }
};

bool parse_double(const char *s, size_t n, double &out) {
    char buf[40];
    if (n >= sizeof(buf)) n = sizeof(buf) - 1;
    std::memcpy(buf, s, n);
    buf[n] = 0;
    char *e = nullptr;
    out = std::strtod(buf, &e);
    if (e == buf) return false;
    while (*e && std::isspace((unsigned char)*e)) ++e;
    return *e == 0;
}
bool starts_with(const Line &l, const char *s) { return l.n >= std::strlen(s) && std::memcmp(l.p, s, std::strlen(s)) == 0; }
// next whitespace-separated token of [p, e)
bool token(const char *&p, const char *e, const char *&t, size_t &tn) {
    while (p < e && std::isspace((unsigned char)*p)) ++p;
    if (p >= e) return false;
    t = p;
    while (p < e && !std::isspace((unsigned char)*p)) ++p;
    tn = (size_t)(p - t);
    return true;
}

} // namespace

// Heavy atoms of every record of an SD file held in memory. Outputs: atoms_per_record[record] (heavy atoms), atomic_num and
// xyz (float32 [atom][3]) for the records one after the other. `max_records` bounds the records read (Ligand.load_from_file's
// num_conformers; 0 = all). Sizing: with atomic_num == NULL and xyz == NULL nothing is stored and *n_records / *n_atoms
// return the counts. Returns PMX_ERR_INVALID for a record that cannot be parsed (bad counts line, short atom block, a
// coordinate that is not a number), naming nothing else: *n_records then holds the index of the offending record.
extern "C" int pmx_sdf_heavy_atoms(const char *text, uint64_t len, uint64_t max_records, uint64_t cap_records, uint64_t cap_atoms, uint64_t *n_records,
                                   uint64_t *n_atoms, int32_t *atoms_per_record, uint8_t *atomic_num, float *xyz) {
    if (!text || !n_records || !n_atoms) return pmx_topk_fail(PMX_ERR_INVALID, "pmx_sdf_heavy_atoms: null argument");
    const bool store = atomic_num != nullptr || xyz != nullptr;
    Cursor cur{text, text + len};
    uint64_t rec = 0, total = 0;
    Line l;
    auto fail_here = [&]() {
        *n_records = rec;
        *n_atoms = total;
        return pmx_topk_fail(PMX_ERR_INVALID, "pmx_sdf_heavy_atoms: malformed record");
    };
    for (;;) {
        if (max_records && rec >= max_records) break;
        // header: three lines, then the counts line; a file may end in blank lines after the last $$$$
        Line head[4];
        int got = 0;
        while (got < 4 && cur.next(head[got])) ++got;
        if (got == 0) break;
        if (got < 4) {
            bool blank = true;
            for (int i = 0; i < got; ++i)
                for (size_t k = 0; k < head[i].n; ++k) blank = blank && std::isspace((unsigned char)head[i].p[k]);
            if (blank) break;
            return fail_here();
        }
        const Line &counts = head[3];
        const bool v3000 = counts.n >= 39 && std::memcmp(counts.p + counts.n - 5, "V3000", 5) == 0;
        int32_t heavy = 0;
        auto emit = [&](int z, double x, double y, double zc) -> bool {
            if (z == 1) return true; // hydrogens are removed (Ligand.__init__: pbmol.removeh(), ligand.py:38)
            if (store) {
                if (total >= cap_atoms) return false;
                if (atomic_num) atomic_num[total] = (uint8_t)(z < 0 ? 0 : z);
                if (xyz) xyz[3 * total] = (float)x, xyz[3 * total + 1] = (float)y, xyz[3 * total + 2] = (float)zc;
            }
            ++total, ++heavy;
            return true;
        };
        if (!v3000) {
            if (counts.n < 6) return fail_here();
            char nb[4] = {0, 0, 0, 0};
            std::memcpy(nb, counts.p, 3);
            char *e = nullptr;
            const long na = std::strtol(nb, &e, 10);
            if (e == nb || na < 0) return fail_here();
            for (long a = 0; a < na; ++a) {
                if (!cur.next(l)) return fail_here();
                double x, y, z;
                int el;
                if (l.n >= 34 && parse_double(l.p, 10, x) && parse_double(l.p + 10, 10, y) && parse_double(l.p + 20, 10, z)) {
                    el = atomic_number(l.p + 31, l.n >= 34 ? 3 : l.n - 31); // columns 32-34
                } else { // not column-aligned: x y z symbol separated by blanks
                    const char *p = l.p, *e2 = l.p + l.n, *t;
                    size_t tn;
                    double v[3];
                    for (int k = 0; k < 3; ++k)
                        if (!token(p, e2, t, tn) || !parse_double(t, tn, v[k])) return fail_here();
                    if (!token(p, e2, t, tn)) return fail_here();
                    x = v[0], y = v[1], z = v[2];
                    el = atomic_number(t, tn);
                }
                if (el < 0) return fail_here();
                if (!emit(el, x, y, z)) return pmx_topk_fail(PMX_ERR_INVALID, "pmx_sdf_heavy_atoms: atom capacity too small");
            }
        } else {
            bool in_atoms = false, done = false;
            while (!done && cur.next(l)) {
                if (starts_with(l, "$$$$")) { // (no atom block at all)
                    cur.p = l.p; // let the record end below see it
                    break;
                }
                if (!starts_with(l, "M  V30 ")) {
                    if (starts_with(l, "M  END")) break;
                    continue;
                }
                const char *p = l.p + 7, *e2 = l.p + l.n, *t;
                size_t tn;
                if (!in_atoms) {
                    const char *q = p;
                    if (token(q, e2, t, tn) && tn == 5 && !std::memcmp(t, "BEGIN", 5) && token(q, e2, t, tn) && tn == 4 && !std::memcmp(t, "ATOM", 4)) in_atoms = true;
                    continue;
                }
                const char *q = p;
                if (token(q, e2, t, tn) && tn == 3 && !std::memcmp(t, "END", 3)) {
                    done = true;
                    continue;
                }
                // index type x y z aamap ...
                q = p;
                const char *ty;
                size_t tyn;
                double v[3];
                if (!token(q, e2, t, tn) || !token(q, e2, ty, tyn)) return fail_here();
                for (int k = 0; k < 3; ++k)
                    if (!token(q, e2, t, tn) || !parse_double(t, tn, v[k])) return fail_here();
                const int el = atomic_number(ty, tyn);
                if (el < 0) return fail_here();
                if (!emit(el, v[0], v[1], v[2])) return pmx_topk_fail(PMX_ERR_INVALID, "pmx_sdf_heavy_atoms: atom capacity too small");
            }
        }
        if (store && atoms_per_record) {
            if (rec >= cap_records) return pmx_topk_fail(PMX_ERR_INVALID, "pmx_sdf_heavy_atoms: record capacity too small");
            atoms_per_record[rec] = heavy;
        }
        ++rec;
        // the rest of the record: bonds, properties, data items, up to the $$$$ line (or the end of a plain .mol file)
        bool ended = false;
        while (cur.next(l))
            if (starts_with(l, "$$$$")) {
                ended = true;
                break;
            }
        if (!ended) break;
    }
    *n_records = rec;
    *n_atoms = total;
    return PMX_OK;
}

// The same for a Tripos mol2 file: every @<TRIPOS>MOLECULE record's @<TRIPOS>ATOM section (`atom_id atom_name x y z atom_type ...`,
// blank-separated), hydrogens dropped. The element is the part of the SYBYL atom type in front of the dot ("C.ar" -> C, "Cl" ->
// Cl, "H.spc" -> H). A type that names no element (Du, LP, Any, Hal, Het, Hev ...) makes the record one this reader does not
// understand - PMX_ERR_INVALID, and the caller goes the toolkit's way - rather than guess what the toolkit would make of it.
extern "C" int pmx_mol2_heavy_atoms(const char *text, uint64_t len, uint64_t max_records, uint64_t cap_records, uint64_t cap_atoms, uint64_t *n_records,
                                    uint64_t *n_atoms, int32_t *atoms_per_record, uint8_t *atomic_num, float *xyz) {
    if (!text || !n_records || !n_atoms) return pmx_topk_fail(PMX_ERR_INVALID, "pmx_mol2_heavy_atoms: null argument");
    const bool store = atomic_num != nullptr || xyz != nullptr;
    Cursor cur{text, text + len};
    uint64_t rec = 0, total = 0;
    bool open = false, in_atoms = false, limit = false;
    int32_t heavy = 0;
    Line l;
    auto fail_here = [&]() {
        *n_records = rec; // (the record in work)
        *n_atoms = total;
        return pmx_topk_fail(PMX_ERR_INVALID, "pmx_mol2_heavy_atoms: malformed record");
    };
    auto close_record = [&]() -> int {
        if (!open) return PMX_OK;
        if (store && atoms_per_record) {
            if (rec >= cap_records) return pmx_topk_fail(PMX_ERR_INVALID, "pmx_mol2_heavy_atoms: record capacity too small");
            atoms_per_record[rec] = heavy;
        }
        ++rec;
        open = false;
        return PMX_OK;
    };
    while (!limit && cur.next(l)) {
        const char *p = l.p, *e = l.p + l.n;
        while (p < e && std::isspace((unsigned char)*p)) ++p;
        if (p == e || *p == '#') continue; // blank and comment lines
        if (*p == '@') {
            const size_t n = (size_t)(e - p);
            in_atoms = false;
            if (n >= 17 && !std::memcmp(p, "@<TRIPOS>MOLECULE", 17)) {
                if (const int rc = close_record()) return rc;
                if (max_records && rec >= max_records) {
                    limit = true;
                    break;
                }
                open = true;
                heavy = 0;
            } else if (n >= 13 && !std::memcmp(p, "@<TRIPOS>ATOM", 13) && (n == 13 || std::isspace((unsigned char)p[13]))) {
                if (!open) return fail_here();
                in_atoms = true;
            }
            continue;
        }
        if (!in_atoms) continue;
        const char *t;
        size_t tn;
        double v[3];
        if (!token(p, e, t, tn) || !token(p, e, t, tn)) return fail_here(); // atom_id, atom_name
        for (int k = 0; k < 3; ++k)
            if (!token(p, e, t, tn) || !parse_double(t, tn, v[k])) return fail_here();
        if (!token(p, e, t, tn)) return fail_here(); // atom_type
        size_t en = 0;
        while (en < tn && t[en] != '.') ++en;
        const int el = atomic_number(t, en);
        if (el <= 0) return fail_here(); // no element: not for this reader
        if (el == 1) continue;           // hydrogens are removed (Ligand.__init__: pbmol.removeh(), ligand.py:38)
        if (store) {
            if (total >= cap_atoms) return pmx_topk_fail(PMX_ERR_INVALID, "pmx_mol2_heavy_atoms: atom capacity too small");
            if (atomic_num) atomic_num[total] = (uint8_t)el;
            if (xyz) xyz[3 * total] = (float)v[0], xyz[3 * total + 1] = (float)v[1], xyz[3 * total + 2] = (float)v[2];
        }
        ++total, ++heavy;
    }
    if (const int rc = close_record()) return rc;
    *n_records = rec;
    *n_atoms = total;
    return PMX_OK;
}
