// pmx_perceive.cpp - the rule half of ligand perception in native code: which atoms of a molecule become which pharmacophore
// features, in which order, with which atom / centre indices (src/pmnet/scoring/ligand_utils.py:25-184), for a batch of molecules.
//
// The reference asks OpenBabel per atom inside Python predicates, molecule after molecule. What only the chemistry toolkit can
// say - an atom's element, degrees, hybridisation, whether it is a hydrogen-bond acceptor / donor, the aromatic rings of the
// smallest set of smallest rings - comes in as flat per-atom answers (pmx_atom_batch, include/pmx.h); everything that follows
// from them is decided here, multi-threaded over molecules, and comes out in the layout pmx_pack_features consumes. Host code
// only. Pinned against the reference's own get_pharmacophore_nodes on 600 described molecules (tests/test_perception.py).
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <new>
#include <thread>
#include <vector>

#include "pmx.h"

#ifdef PMX_PACK_STANDALONE
static int pmx_topk_fail(int code, const char *) { return code; } // (libpmx_pack.so keeps its message in pmx_pack.cpp)
#else
int pmx_topk_fail(int code, const char *msg); // error hook in pmx_api.hip
#endif

namespace {

enum : uint8_t { F_HYDROPHOBIC = 0, F_AROMATIC = 1, F_CATION = 2, F_ANION = 3, F_DONOR = 4, F_ACCEPTOR = 5, F_HALOGEN = 6 }; // constants.TYPE_ID

struct Feature {
    uint8_t type, flags; // flags bit 0: atom_indices is a tuple, bit 1: center_indices is a tuple (pmx_feature_batch)
    std::vector<int32_t> atoms, centers;
};

inline bool halogen(int z) { return z == 9 || z == 17 || z == 35 || z == 53; }

struct MolView {
    int n;
    const uint8_t *z, *explicit_degree, *heavy_degree, *hyb, *h_count, *flags;
    const uint64_t *nbr_off; // [n + 1], offsets into nbr (absolute)
    const int32_t *nbr;
    uint64_t ring0, ring1;
    const uint64_t *ring_atom_off;
    const int32_t *ring_atoms;
};

// The features of one molecule in pharmacophore_list order (ligand_utils.py:80-88). Returns false on malformed input.
bool perceive_one(const MolView &m, std::vector<Feature> &out) {
    out.clear();
    const int n = m.n;
    auto nb = [&](int i) { return m.nbr + m.nbr_off[i]; };
    auto deg = [&](int i) { return (int)(m.nbr_off[i + 1] - m.nbr_off[i]); };
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < deg(i); ++k)
            if (nb(i)[k] < 0 || nb(i)[k] >= n) return false;
    auto count = [&](int i, int z) { // neighbours of element z (hydrogens are not atoms of the molecule: h_count)
        if (z == 1) return (int)m.h_count[i];
        int c = 0;
        for (int k = 0; k < deg(i); ++k) c += m.z[nb(i)[k]] == z ? 1 : 0;
        return c;
    };
    auto single = [&](uint8_t type, int i) {
        Feature f;
        f.type = type, f.flags = 0;
        f.atoms = {i}, f.centers = {i};
        out.push_back(std::move(f));
    };
    // Hydrophobic: a carbon bound to carbons and hydrogens only (:36-40)
    for (int i = 0; i < n; ++i) {
        if (m.z[i] != 6) continue;
        bool ok = true;
        for (int k = 0; k < deg(i); ++k) ok = ok && m.z[nb(i)[k]] == 6;
        if (ok) single(F_HYDROPHOBIC, i);
    }
    // Aromatic: the aromatic rings of the SSSR, each as the sorted tuple of its atoms, the rings sorted (:47-52)
    {
        std::vector<std::vector<int32_t>> rings;
        for (uint64_t r = m.ring0; r < m.ring1; ++r) {
            std::vector<int32_t> ring(m.ring_atoms + m.ring_atom_off[r], m.ring_atoms + m.ring_atom_off[r + 1]);
            for (int32_t a : ring)
                if (a < 0 || a >= n) return false;
            std::sort(ring.begin(), ring.end());
            rings.push_back(std::move(ring));
        }
        std::sort(rings.begin(), rings.end()); // (tuples compare lexicographically, a prefix before what extends it: as std::vector does)
        for (auto &ring : rings) {
            Feature f;
            f.type = F_AROMATIC, f.flags = 3;
            f.atoms = ring, f.centers = ring;
            out.push_back(std::move(f));
        }
    }
    // Cations: single charged atoms first (:54-58, :94-118) ...
    std::vector<Feature> anions;
    for (int i = 0; i < n; ++i) {
        const int z = m.z[i];
        const bool quaternary_n = z == 7 && m.explicit_degree[i] == 4 && m.h_count[i] == 0;
        const bool tertiary_n = z == 7 && m.hyb[i] == 3 && m.heavy_degree[i] == 3;
        const bool sulfonium = z == 16 && m.explicit_degree[i] == 3 && m.h_count[i] == 0;
        if (quaternary_n || tertiary_n || sulfonium) single(F_CATION, i);
    }
    // ... then charged groups (:61-76, :121-175): guanidine is a cation, the others anions
    for (int i = 0; i < n; ++i) {
        const int z = m.z[i], d = deg(i), dall = d + (int)m.h_count[i];
        auto with_neighbours = [&](int only) { // (i,) + the neighbours (of element `only`; 0 = all) in the toolkit's order
            std::vector<int32_t> v{i};
            for (int k = 0; k < d; ++k)
                if (only == 0 || m.z[nb(i)[k]] == only) v.push_back(nb(i)[k]);
            return v;
        };
        bool all_n = dall > 0 && m.h_count[i] == 0, all_o = m.h_count[i] == 0, terminal_n = false;
        for (int k = 0; k < d; ++k) {
            all_n = all_n && m.z[nb(i)[k]] == 7;
            all_o = all_o && m.z[nb(i)[k]] == 8;
            terminal_n = terminal_n || m.heavy_degree[nb(i)[k]] == 1;
        }
        const bool guanidine = z == 6 && all_n && dall == 3 && terminal_n;
        const bool phosphate = z == 15 && all_o;
        const bool sulfate = z == 16 && count(i, 8) == 4, sulfonic = z == 16 && count(i, 8) == 3;
        const bool carboxylate = z == 6 && count(i, 8) == 2 && count(i, 6) == 1;
        Feature f;
        if (guanidine) {
            f.type = F_CATION, f.flags = 1;
            f.atoms = with_neighbours(7), f.centers = {i};
            out.push_back(std::move(f));
        } else if (phosphate || sulfate) {
            f.type = F_ANION, f.flags = 1;
            f.atoms = with_neighbours(0), f.centers = {i};
            anions.push_back(std::move(f));
        } else if (sulfonic) {
            f.type = F_ANION, f.flags = 1;
            f.atoms = with_neighbours(8), f.centers = {i};
            anions.push_back(std::move(f));
        } else if (carboxylate) {
            f.type = F_ANION, f.flags = 3;
            f.atoms = with_neighbours(8);
            f.centers.assign(f.atoms.begin() + 1, f.atoms.end()); // the oxygens
            anions.push_back(std::move(f));
        }
    }
    for (auto &f : anions) out.push_back(std::move(f));
    for (int i = 0; i < n; ++i)
        if (m.flags[i] & 2) single(F_DONOR, i); // judged on the molecule with polar hydrogens (:30-34,46)
    for (int i = 0; i < n; ++i)
        if (!halogen(m.z[i]) && (m.flags[i] & 1)) single(F_ACCEPTOR, i); // (:41-45)
    for (int i = 0; i < n; ++i)
        if (halogen(m.z[i]) && count(i, 6) > 0) single(F_HALOGEN, i); // (:78, :178-184)
    return true;
}

} // namespace

extern "C" int pmx_perceive_features(const pmx_atom_batch *b, int threads, uint64_t *feat_off, uint8_t *feat_type, uint8_t *feat_flags,
                                     uint64_t *feat_atom_off, int32_t *feat_atoms, uint64_t *feat_center_off, int32_t *feat_centers,
                                     uint64_t cap_features, uint64_t cap_atoms, uint64_t cap_centers, uint64_t *n_features, uint64_t *n_feat_atoms,
                                     uint64_t *n_feat_centers, int32_t *status_out) {
    if (!b || !feat_off || !n_features || !n_feat_atoms || !n_feat_centers) return pmx_topk_fail(PMX_ERR_INVALID, "pmx_perceive_features: null argument");
    const uint64_t n = b->n_mols;
    if (n && (!b->atom_off || !b->ring_off)) return pmx_topk_fail(PMX_ERR_INVALID, "pmx_perceive_features: null offset array in the batch");
    if (n && b->atom_off[n] > b->atom_off[0] && (!b->atomic_num || !b->explicit_degree || !b->heavy_degree || !b->hyb || !b->h_count || !b->flags || !b->nbr_off))
        return pmx_topk_fail(PMX_ERR_INVALID, "pmx_perceive_features: null per-atom array in the batch");
    if (n && b->ring_off[n] > b->ring_off[0] && !b->ring_atom_off) return pmx_topk_fail(PMX_ERR_INVALID, "pmx_perceive_features: rings without ring_atom_off");
    for (uint64_t i = 0; i < n; ++i)
        if (b->atom_off[i + 1] < b->atom_off[i] || b->ring_off[i + 1] < b->ring_off[i])
            return pmx_topk_fail(PMX_ERR_INVALID, "pmx_perceive_features: offsets run backwards");
    try {
        // pass 1: perceive every molecule (kept per molecule), pass 2: lay the features out flat
        std::vector<std::vector<Feature>> all(n);
        std::vector<uint8_t> bad(n, 0);
        std::atomic<uint64_t> next{0};
        auto work = [&]() {
            for (;;) {
                const uint64_t i0 = next.fetch_add(64);
                if (i0 >= n) break;
                for (uint64_t i = i0; i < std::min(n, i0 + 64); ++i) {
                    const uint64_t a0 = b->atom_off[i];
                    MolView m;
                    m.n = (int)(b->atom_off[i + 1] - a0);
                    m.z = b->atomic_num + a0, m.explicit_degree = b->explicit_degree + a0, m.heavy_degree = b->heavy_degree + a0;
                    m.hyb = b->hyb + a0, m.h_count = b->h_count + a0, m.flags = b->flags + a0;
                    m.nbr_off = b->nbr_off + a0, m.nbr = b->nbr;
                    m.ring0 = b->ring_off[i], m.ring1 = b->ring_off[i + 1];
                    m.ring_atom_off = b->ring_atom_off, m.ring_atoms = b->ring_atoms;
                    bool ok = true;
                    for (int k = 0; k < m.n && ok; ++k) ok = m.nbr_off[k + 1] >= m.nbr_off[k];
                    // (a ring whose atom offsets run backwards would be a negative range in perceive_one; ring or neighbour lists without their arrays)
                    for (uint64_t r = m.ring0; r < m.ring1 && ok; ++r) ok = m.ring_atom_off[r + 1] >= m.ring_atom_off[r];
                    ok = ok && (m.ring0 == m.ring1 || m.ring_atom_off[m.ring1] == m.ring_atom_off[m.ring0] || m.ring_atoms != nullptr);
                    ok = ok && (m.n == 0 || m.nbr_off[m.n] == m.nbr_off[0] || m.nbr != nullptr);
                    try {
                        ok = ok && perceive_one(m, all[i]);
                    } catch (...) {
                        ok = false;
                    }
                    if (!ok) {
                        all[i].clear();
                        bad[i] = 1;
                    }
                }
            }
        };
        const int nt = std::max(1, std::min(threads, 256));
        std::vector<std::thread> pool;
        pool.reserve((size_t)nt);
        try {
            for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        } catch (...) { // a thread could not be started: the ones that run finish the work (joinable threads must not be unwound)
        }
        work();
        for (auto &t : pool) t.join();
        uint64_t nf = 0, na = 0, nc = 0;
        for (uint64_t i = 0; i < n; ++i) {
            feat_off[i] = nf;
            for (const Feature &f : all[i]) na += f.atoms.size(), nc += f.centers.size();
            nf += all[i].size();
            if (status_out) status_out[i] = bad[i] ? 2 : 0;
        }
        feat_off[n] = nf;
        *n_features = nf, *n_feat_atoms = na, *n_feat_centers = nc;
        if (!feat_type) return PMX_OK; // counting call
        if (!feat_flags || !feat_atom_off || !feat_center_off || (na && !feat_atoms) || (nc && !feat_centers))
            return pmx_topk_fail(PMX_ERR_INVALID, "pmx_perceive_features: null output");
        if (nf > cap_features || na > cap_atoms || nc > cap_centers)
            return pmx_topk_fail(PMX_ERR_INVALID, "pmx_perceive_features: output capacity too small (the counts hold the sizes needed)");
        uint64_t k = 0, ka = 0, kc = 0;
        for (uint64_t i = 0; i < n; ++i)
            for (const Feature &f : all[i]) {
                feat_type[k] = f.type, feat_flags[k] = f.flags;
                feat_atom_off[k] = ka, feat_center_off[k] = kc;
                for (int32_t a : f.atoms) feat_atoms[ka++] = a;
                for (int32_t c : f.centers) feat_centers[kc++] = c;
                ++k;
            }
        feat_atom_off[k] = ka, feat_center_off[k] = kc;
        return PMX_OK;
    } catch (const std::bad_alloc &) {
        return pmx_topk_fail(PMX_ERR_OOM, "pmx_perceive_features: out of host memory");
    } catch (...) {
        return pmx_topk_fail(PMX_ERR_INVALID, "pmx_perceive_features: internal error");
    }
}
