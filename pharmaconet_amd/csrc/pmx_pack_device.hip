// pmx_pack_device.hip - the library packer on the device: pmx_pack_features (pmx_pack.cpp) for feature batches that are in HBM.
//
// Same input, same records, byte for byte: LigandGraph's node merging, grouping and clustering (src/pmnet/scoring/ligand.py:110-259) and
// the priority sort of graph_match.py:43-60. The host packer needs 2.7 CPU-seconds per 10^6 molecules; a host that is granted 16 cores'
// worth of time packs 6 x 10^6 molecules/s sustained, a tenth of what one resident pass scores. Here the packer
// is four launches in front of the scoring call, on the caller's stream:
//
//   graph_wave_kernel  one wavefront per molecule, the wavefront as the data structure (lane f = feature f, lane i = node i, lane c =
//                      cluster c, atom sets as 128-bit masks): the reference's node-by-node loops become uniform readlane loops against all
//                      lanes at once; the hydrophobic flood and the cluster assignment stay sequential, as scalar code. Molecules of up to
//                      64 features, 128 atoms, 512 neighbour entries, 8 atoms per key. Out: a 196-byte descriptor (per packed node its type
//                      mask and the feature whose centres place it; the clusters' ends), the record's size, the status.
//   graph_kernel       the molecules the wave builder left: staged into LDS byte arrays, lane 0 walks pack_one (pmx_pack.cpp) step by step.
//   (exclusive scan of the sizes: hipcub)
//   record_kernel      one wavefront per molecule writes the record at its offset: header, type masks, cluster ends, and the node
//                      positions [node][3][C] gathered from the conformer coordinates (tuple centres: float32 sum atom after atom, one
//                      IEEE division - LigandNode.set_positions, ligand.py:293-301), stores coalesced.
//
// Fixed scratch means limits beyond the format's own: a molecule of more than 256 atoms, 255 features, 1024 neighbour entries,
// 1024 feature-atom entries or a feature of more than 16 atoms gets status 3 and a header-only record - pack such a batch with
// pmx_pack_features. (Drug-like molecules are an order of magnitude below every one of them.) One more difference, in the status
// only: the general builder ends its walk at the 65th node with status 1, where the host packer would still report 2 if that
// molecule's feature graph also made the reference's builder raise.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <mutex>

#include "pmx.h"

int pmx_topk_fail(int code, const char *msg); // error hook in pmx_api.hip

namespace {

constexpr int kMaxAtoms = 256, kMaxNbr = 1024, kMaxFeat = 255, kMaxFeatAtoms = 1024, kMaxKey = 16, kMaxNodes = PMX_MAX_LIGAND_NODES, kMaxMembers = 192;
static_assert(kMaxNodes == 64, "node indices are kept in int8 / one-wavefront arrays");

enum : unsigned { T_HYDROPHOBIC = 1u << 0, T_AROMATIC = 1u << 1, T_CATION = 1u << 2, T_ANION = 1u << 3, T_DONOR = 1u << 4, T_ACCEPTOR = 1u << 5, T_HALOGEN = 1u << 6 };
constexpr unsigned T_HBOND = T_DONOR | T_ACCEPTOR, T_ION = T_CATION | T_ANION;
enum ClusterType { C_AROMATIC, C_CATION, C_ANION, C_HBOND, C_HALOGEN, C_HYDROPHOBIC }; // CLUSTER_PRIORITY order (constants.py)

struct PackDesc {
    uint16_t n, ncl;   // n == 0xFFFF: header-only record
    uint8_t types[64]; // packed node -> type mask
    uint8_t feat[64];  // packed node -> the feature (index inside the molecule) whose centres place it
    uint8_t ends[64];  // cluster -> one past its last packed node
};

struct DevBatch { // pmx_feature_batch, device pointers
    uint64_t n_mols;
    const uint64_t *atom_off;
    const uint8_t *atomic_num;
    const uint64_t *nbr_off;
    const int32_t *nbr;
    const uint64_t *feat_off;
    const uint8_t *feat_type, *feat_flags;
    const uint64_t *feat_atom_off;
    const int32_t *feat_atoms;
    const uint64_t *feat_center_off;
    const int32_t *feat_centers;
    const int32_t *n_conf;
    const uint64_t *pos_off;
    const float *positions;
};

struct PackLds {
    uint16_t nbr_off[kMaxAtoms + 1];
    uint16_t fatom_off[kMaxFeat + 1];
    uint8_t z[kMaxAtoms];
    uint8_t nbr[kMaxNbr];
    uint8_t ftype[256], fflags[256];
    uint8_t fatoms[kMaxFeatAtoms];
    // nodes
    uint8_t atoms[kMaxNodes][kMaxKey]; // sorted, unique (the frozenset)
    uint8_t key[kMaxNodes][kMaxKey];   // atom indices as given (the dict key, with ktuple)
    uint8_t natoms[kMaxNodes], nkey[kMaxNodes], ktuple[kMaxNodes];
    uint8_t types[kMaxNodes], feat_of[kMaxNodes], ngroup[kMaxNodes];
    int8_t min_dep[kMaxNodes];
    uint8_t group[kMaxNodes][kMaxNodes]; // insertion-ordered set of node indices
    uint8_t node_of_feat[256];
    // functional groups / hydrophobic flood
    int8_t grp_first[2][kMaxAtoms], grp_last[2][kMaxAtoms], grp_next[2][kMaxNodes];
    int8_t where[kMaxAtoms];
    uint8_t ent_atom[kMaxNodes], ent_node[kMaxNodes], alive[kMaxNodes];
    uint8_t members[kMaxMembers], gindex[kMaxMembers];
    // clusters
    int8_t cl_first[kMaxNodes], cl_last[kMaxNodes], cl_next[kMaxNodes], founder[kMaxNodes];
    uint8_t cl_size[kMaxNodes], ctype[kMaxNodes], order[kMaxNodes], in_cluster[kMaxNodes];
    int bad;
};

__device__ inline bool subset(const uint8_t *a, int na, const uint8_t *b, int nb) { // sorted a within sorted b
    int j = 0;
    for (int i = 0; i < na; ++i) {
        while (j < nb && b[j] < a[i]) ++j;
        if (j == nb || b[j] != a[i]) return false;
        ++j;
    }
    return true;
}
__device__ inline void add_dep(PackLds &S, int node, int idx) {
    const int cur = S.min_dep[node];
    S.min_dep[node] = (int8_t)(cur < 0 ? idx : (idx < cur ? idx : cur));
}
__device__ inline void group_add(PackLds &S, int node, int idx) {
    const int ng = S.ngroup[node];
    for (int q = 0; q < ng; ++q)
        if (S.group[node][q] == idx) return;
    S.group[node][ng] = (uint8_t)idx; // (at most 64 distinct node indices exist)
    S.ngroup[node] = (uint8_t)(ng + 1);
}

// Lane 0: pack_one of pmx_pack.cpp on the staged molecule. Returns 0 (descriptor written), 1 (outside the format's limits),
// 2 (the reference's builder raises), 3 (outside this kernel's scratch).
__device__ int build_graph(PackLds &S, int n_feat, int n_conf, PackDesc *desc) {
    int n = 0;
    // __add_nodes (ligand.py:134-156)
    for (int f = 0; f < n_feat; ++f) {
        const int t = S.ftype[f];
        const int kt = S.fflags[f] & 1;
        const int kb = S.fatom_off[f], kl = S.fatom_off[f + 1] - kb;
        if (kl > kMaxKey) return 3;
        int found = -1;
        for (int i = 0; i < n && found < 0; ++i) { // by_key lookup: same tuple-ness, same indices in the same order
            if (S.ktuple[i] != kt || S.nkey[i] != kl) continue;
            bool eq = true;
            for (int q = 0; q < kl && eq; ++q) eq = S.key[i][q] == S.fatoms[kb + q];
            if (eq) found = i;
        }
        if (found >= 0) {
            S.types[found] |= (uint8_t)(1u << t);
            S.node_of_feat[f] = (uint8_t)found;
            continue;
        }
        if (n == kMaxNodes) return 1;
        const int ni = n;
        const unsigned nw_types = 1u << t;
        S.types[ni] = (uint8_t)nw_types;
        int na = 0;
        for (int q = 0; q < kl; ++q) {
            const uint8_t a = S.fatoms[kb + q];
            S.key[ni][q] = a;
            int p = 0;
            while (p < na && S.atoms[ni][p] < a) ++p;
            if (p < na && S.atoms[ni][p] == a) continue;
            for (int r = na; r > p; --r) S.atoms[ni][r] = S.atoms[ni][r - 1];
            S.atoms[ni][p] = a;
            ++na;
        }
        S.natoms[ni] = (uint8_t)na;
        S.nkey[ni] = (uint8_t)kl;
        S.ktuple[ni] = (uint8_t)kt;
        S.feat_of[ni] = (uint8_t)f;
        S.ngroup[ni] = 0;
        S.min_dep[ni] = -1;
        for (int oi = 0; oi < ni; ++oi) { // old.add_neighbors(new) (ligand.py:303-329); old's types as they are NOW
            const unsigned ot = S.types[oi];
            if ((ot & T_HYDROPHOBIC) && (nw_types & T_AROMATIC)) {
                if (subset(S.atoms[oi], S.natoms[oi], S.atoms[ni], na)) add_dep(S, oi, ni);
            } else if ((ot & T_AROMATIC) && (nw_types & T_HYDROPHOBIC)) {
                if (subset(S.atoms[ni], na, S.atoms[oi], S.natoms[oi])) add_dep(S, ni, oi);
            } else if ((ot & T_HBOND) && (nw_types & T_ION)) {
                if (subset(S.atoms[oi], S.natoms[oi], S.atoms[ni], na)) add_dep(S, oi, ni);
            } else if ((ot & T_ION) && (nw_types & T_HBOND)) {
                if (subset(S.atoms[ni], na, S.atoms[oi], S.natoms[oi])) add_dep(S, ni, oi);
            }
        }
        ++n;
        S.node_of_feat[f] = (uint8_t)ni;
    }
    // __group_nodes, functional groups (ligand.py:158-192): atoms bonded to the same single heavy neighbour
    for (int i = 0; i < n; ++i) {
        int kind;
        if (S.types[i] & T_HBOND) kind = 0;
        else if (S.types[i] & T_HYDROPHOBIC) kind = 1;
        else continue;
        const int atom = S.atoms[i][0];
        int count = 0, only = -1;
        for (int q = S.nbr_off[atom]; q < S.nbr_off[atom + 1]; ++q)
            if (S.z[S.nbr[q]] != 1) {
                ++count;
                only = S.nbr[q];
            }
        if (count == 1) {
            for (int other = S.grp_first[kind][only]; other >= 0; other = S.grp_next[kind][other]) {
                group_add(S, i, other);
                group_add(S, other, i);
            }
            if (S.grp_last[kind][only] >= 0) S.grp_next[kind][S.grp_last[kind][only]] = (int8_t)i;
            else S.grp_first[kind][only] = (int8_t)i;
            S.grp_last[kind][only] = (int8_t)i;
        }
    }
    // __group_nodes, hydrophobic flood over carbon-carbon bonds (ligand.py:194-213): index_to_node is a dict built from
    // node_dict["Hydrophobic"] - a repeated key keeps its first position and takes the last value; popitem() is LIFO
    {
        int n_ent = 0;
        for (int f = 0; f < n_feat; ++f) {
            if (S.ftype[f] != 0) continue;
            const int ni = S.node_of_feat[f];
            const int atom = S.atoms[ni][0];
            if (S.where[atom] < 0) {
                S.where[atom] = (int8_t)n_ent;
                S.ent_atom[n_ent] = (uint8_t)atom;
                S.ent_node[n_ent] = (uint8_t)ni;
                S.alive[n_ent] = 1;
                ++n_ent;
            } else {
                S.ent_node[S.where[atom]] = (uint8_t)ni;
            }
        }
        int last = n_ent - 1;
        for (;;) {
            while (last >= 0 && !S.alive[last]) --last;
            if (last < 0) break;
            const int start = S.ent_node[last];
            S.alive[last] = 0;
            S.where[S.ent_atom[last]] = -1;
            int n_mem = 0;
            S.members[n_mem++] = (uint8_t)start;
            for (int q = 0; q < S.ngroup[start]; ++q) S.members[n_mem++] = S.group[start][q];
            int n_gi = 0;
            for (int q = 0; q < n_mem; ++q) S.gindex[n_gi++] = S.atoms[S.members[q]][0];
            for (int gi = 0; gi < n_gi; ++gi) { // grows while iterating
                const int ga = S.gindex[gi];
                for (int q = S.nbr_off[ga]; q < S.nbr_off[ga + 1]; ++q) {
                    const int b = S.nbr[q];
                    if (S.z[b] != 6) continue;
                    const int at = S.where[b];
                    if (at < 0) continue;
                    const int reached = S.ent_node[at];
                    S.alive[at] = 0;
                    S.where[b] = -1;
                    if (n_mem == kMaxMembers) return 3;
                    S.gindex[n_gi++] = (uint8_t)b;
                    for (int mq = 0; mq < n_mem; ++mq) {
                        const int mi = S.members[mq];
                        group_add(S, mi, reached);
                        group_add(S, reached, mi);
                    }
                    S.members[n_mem++] = (uint8_t)reached;
                }
            }
        }
    }
    // __setup_cluster (ligand.py:215-259); a cluster is a list of nodes in the order they joined
    int ncl = 0;
    const int high_types[4] = {1, 2, 3, 6}; // Aromatic, Cation, Anion, Halogen
    const int high_ctype[4] = {C_AROMATIC, C_CATION, C_ANION, C_HALOGEN};
    for (int h = 0; h < 4; ++h)
        for (int f = 0; f < n_feat; ++f) {
            if (S.ftype[f] != high_types[h]) continue;
            const int ni = S.node_of_feat[f];
            if (S.in_cluster[ni]) continue;
            S.in_cluster[ni] = 1;
            S.cl_first[ncl] = S.cl_last[ncl] = (int8_t)ni;
            S.cl_size[ncl] = 1;
            S.ctype[ncl] = (uint8_t)high_ctype[h];
            S.founder[ni] = (int8_t)ncl;
            ++ncl;
        }
    const int low_types[3] = {0, 4, 5}; // Hydrophobic, HBond_donor, HBond_acceptor
    for (int l = 0; l < 3; ++l)
        for (int f = 0; f < n_feat; ++f) {
            if (S.ftype[f] != low_types[l]) continue;
            const int ni = S.node_of_feat[f];
            if (S.in_cluster[ni]) continue;
            S.in_cluster[ni] = 1;
            int join = -1;
            const int dep = S.min_dep[ni];
            if (dep >= 0) {
                if (S.founder[dep] < 0) return 2; // the reference's builder raises KeyError (ligand.py:238-241)
                join = S.founder[dep];
            } else {
                for (int q = 0; q < S.ngroup[ni] && join < 0; ++q)
                    if (S.founder[S.group[ni][q]] >= 0) join = S.founder[S.group[ni][q]];
            }
            if (join >= 0) {
                S.cl_next[S.cl_last[join]] = (int8_t)ni;
                S.cl_last[join] = (int8_t)ni;
                ++S.cl_size[join];
            } else {
                S.cl_first[ncl] = S.cl_last[ncl] = (int8_t)ni;
                S.cl_size[ncl] = 1;
                S.ctype[ncl] = (uint8_t)(l == 0 ? C_HYDROPHOBIC : C_HBOND);
                S.founder[ni] = (int8_t)ncl;
                ++ncl;
            }
        }
    // stable sort by priority_fn (graph_match.py:43-60): (group, -size, subtype, first atom of the founder)
    auto sort_key = [&](int c) {
        const int ct = S.ctype[c];
        const int grp = ct >= 3, sub = ct >= 3 ? ct - 3 : ct;
        return (grp << 24) | ((64 - (int)S.cl_size[c]) << 16) | (sub << 8) | (int)S.atoms[S.cl_first[c]][0];
    };
    for (int i = 0; i < ncl; ++i) {
        const int kx = sort_key(i);
        int j = i;
        for (; j > 0 && kx < sort_key(S.order[j - 1]); --j) S.order[j] = S.order[j - 1];
        S.order[j] = (uint8_t)i;
    }
    if (n_conf > PMX_MAX_CONFORMERS) return 1; // (n <= 64 and ncl <= n hold by construction)
    int pos = 0;
    for (int ci = 0; ci < ncl; ++ci) {
        for (int ni = S.cl_first[S.order[ci]]; ni >= 0; ni = S.cl_next[ni]) {
            desc->types[pos] = S.types[ni];
            desc->feat[pos] = S.feat_of[ni];
            ++pos;
        }
        desc->ends[ci] = (uint8_t)pos;
    }
    desc->n = (uint16_t)n;
    desc->ncl = (uint16_t)ncl;
    return 0;
}

__device__ inline uint64_t record_bytes(int n, int ncl, int C) {
    const uint64_t head = 8 + (uint64_t)n + (uint64_t)ncl;
    const uint64_t body = ((head + 3) & ~3ull) + 12ull * n * C;
    return (body + 15) & ~15ull;
}

// valid_molecule (pmx_pack.cpp): everything the builders index with, checked on the raw arrays first, all lanes at it.
// (Offsets that run backwards - the host packer fails the whole call on them - make the molecule malformed here.)
__device__ bool molecule_is_malformed(const DevBatch &b, uint64_t i, int lane, uint64_t a0, uint64_t f0, int64_t n_atoms, int64_t n_feat, int n_conf) {
    int bad = 0;
    if (n_atoms < 0 || n_feat < 0 || n_conf < 1 || n_conf > (1 << 16)) bad = 1;
    else if (b.pos_off[i + 1] < b.pos_off[i] || b.pos_off[i + 1] - b.pos_off[i] < (uint64_t)n_atoms * (uint64_t)n_conf * 3) bad = 1;
    if (!bad) {
        for (int64_t a = lane; a < n_atoms && !bad; a += 64) {
            const uint64_t q0 = b.nbr_off[a0 + a], q1 = b.nbr_off[a0 + a + 1];
            if (q1 < q0) bad = 1;
            for (uint64_t q = q0; q < q1 && !bad; ++q) {
                const int32_t v = b.nbr[q];
                if (v < 0 || v >= n_atoms) bad = 1;
            }
        }
        for (int64_t f = lane; f < n_feat && !bad; f += 64) {
            if (b.feat_type[f0 + f] >= PMX_NUM_TYPES) bad = 1;
            const uint64_t q0 = b.feat_atom_off[f0 + f], q1 = b.feat_atom_off[f0 + f + 1];
            const uint64_t c0 = b.feat_center_off[f0 + f], c1 = b.feat_center_off[f0 + f + 1];
            if (q1 <= q0 || c1 <= c0) bad = 1;
            for (uint64_t q = q0; q < q1 && !bad; ++q) {
                const int32_t v = b.feat_atoms[q];
                if (v < 0 || v >= n_atoms) bad = 1;
            }
            for (uint64_t q = c0; q < c1 && !bad; ++q) {
                const int32_t v = b.feat_centers[q];
                if (v < 0 || v >= n_atoms) bad = 1;
            }
        }
    }
    return __any(bad) != 0;
}

constexpr int kTodo = 4; // graph_wave_kernel -> graph_kernel: beyond the wave builder's shape, take the general one

__device__ inline void finish_molecule(int lane, int verdict, int n_conf, PackDesc *d, uint64_t *size, int32_t *status) {
    if (lane == 0) {
        *size = verdict == 0 ? record_bytes(d->n, d->ncl, n_conf) : 16;
        if (verdict != 0) d->n = 0xFFFF;
        *status = verdict;
    }
}

// ---------------------------------------------------------------------------------------------------- the wave builder
// The same builder with the wavefront as its data structure, for molecules of up to 64 features, 128 atoms, 512 neighbour entries and
// key lists of up to 8 atoms (what drug-like molecules are; anything else is left to graph_kernel below). Lane f holds feature f,
// lane i holds node i, lane c holds cluster c, in registers; atom sets are 128-bit masks (the frozenset: subset = and-not, first atom =
// count of trailing zeros); what the reference does node by node is a uniform loop of readlanes over one side and all lanes at once on the
// other. Only the hydrophobic flood and the cluster assignment stay sequential - as scalar code (values broadcast with readlane /
// readfirstlane, branches on SGPRs), their dicts and lists being lane registers and a few byte arrays in LDS.
struct WaveLds {
    uint16_t nbr_off[129];
    uint8_t z[128];
    uint8_t nbr[512];
    int8_t where[128];
    uint8_t members[kMaxMembers], gindex[kMaxMembers];
    uint8_t glist[64][64]; // node -> its group, in insertion order
    uint8_t cf[64];        // node -> the feature that made it
};

__device__ inline int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ inline uint64_t rl64(uint64_t v, int lane) {
    return (uint64_t)(uint32_t)rl((int)(uint32_t)v, lane) | ((uint64_t)(uint32_t)rl((int)(uint32_t)(v >> 32), lane) << 32);
}
__device__ inline int from_lane(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
__device__ inline uint64_t from_lane64(uint64_t v, int src) {
    return (uint64_t)(uint32_t)from_lane((int)(uint32_t)v, src) | ((uint64_t)(uint32_t)from_lane((int)(uint32_t)(v >> 32), src) << 32);
}
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); } // a value every lane holds alike, as a scalar

__global__ __launch_bounds__(64) void graph_wave_kernel(DevBatch b, PackDesc *desc, uint64_t *sizes, int32_t *status) {
    __shared__ WaveLds S;
    const uint64_t i = blockIdx.x;
    const int lane = threadIdx.x;
    const uint64_t a0 = b.atom_off[i], a1 = b.atom_off[i + 1], f0 = b.feat_off[i], f1 = b.feat_off[i + 1];
    const int64_t n_atoms64 = (int64_t)(a1 - a0), n_feat64 = (int64_t)(f1 - f0);
    const int n_conf = b.n_conf[i];
    if (molecule_is_malformed(b, i, lane, a0, f0, n_atoms64, n_feat64, n_conf)) {
        finish_molecule(lane, 2, n_conf, desc + i, sizes + i, status + i);
        return;
    }
    const uint64_t nb0 = b.nbr_off[a0];
    const uint64_t n_nbr64 = b.nbr_off[a1] - nb0;
    // this lane's feature: type, key (the atom indices as given, a byte each), the atoms as a set
    int t = 255, meta = -1; // meta: key length | tuple-ness << 8
    uint64_t key = 0, am_lo = 0, am_hi = 0;
    bool shape_ok = n_atoms64 <= 128 && n_feat64 <= 64 && n_nbr64 <= 512;
    if (shape_ok && lane < (int)n_feat64) {
        const uint64_t q0 = b.feat_atom_off[f0 + lane], kl = b.feat_atom_off[f0 + lane + 1] - q0;
        if (kl > 8) shape_ok = false;
        else {
            t = b.feat_type[f0 + lane];
            meta = (int)kl | ((b.feat_flags[f0 + lane] & 1) << 8);
            for (uint64_t q = 0; q < kl; ++q) {
                const uint32_t a = (uint32_t)b.feat_atoms[q0 + q];
                key |= (uint64_t)a << (8 * q);
                if (a < 64) am_lo |= 1ull << a;
                else am_hi |= 1ull << (a - 64);
            }
        }
    }
    if (!__all(shape_ok)) {
        if (lane == 0) status[i] = kTodo;
        return;
    }
    const int n_atoms = (int)n_atoms64, n_feat = (int)n_feat64, n_nbr = (int)n_nbr64;
    for (int a = lane; a <= n_atoms; a += 64) S.nbr_off[a] = (uint16_t)(b.nbr_off[a0 + a] - nb0);
    for (int a = lane; a < n_atoms; a += 64) S.z[a] = b.atomic_num[a0 + a];
    for (int q = lane; q < n_nbr; q += 64) S.nbr[q] = (uint8_t)b.nbr[nb0 + q];
    S.where[lane] = S.where[lane + 64] = -1;

    // __add_nodes (ligand.py:134-156), the by_key lookup: a feature makes a node unless an earlier feature has its key
    int first = lane;
    for (int g = 0; g < n_feat; ++g) {
        const uint64_t kg = rl64(key, g);
        const int mg = rl(meta, g);
        if (lane > g && first == lane && key == kg && meta == mg) first = g;
    }
    const bool is_new = lane < n_feat && first == lane;
    const uint64_t newmask = __ballot(is_new);
    const int n = __popcll(newmask);
    const int node_of = lane < n_feat ? __popcll(newmask & ((1ull << first) - 1)) : 0; // this lane's feature -> its node
    if (is_new) S.cf[node_of] = (uint8_t)lane;
    __syncthreads();
    // this lane's node
    const int cf = lane < n ? S.cf[lane] : 0;
    const uint64_t got_lo = from_lane64(am_lo, cf), got_hi = from_lane64(am_hi, cf); // (every lane takes part in the exchange)
    const uint64_t nam_lo = lane < n ? got_lo : 0, nam_hi = lane < n ? got_hi : 0;
    const int fatom = nam_lo ? __builtin_ctzll(nam_lo) : (nam_hi ? 64 + __builtin_ctzll(nam_hi) : 0); // atoms[0] of the sorted set
    int types = 0, min_dep = -1;
    // the features in order: a new node meets every older node with the types it has NOW (add_neighbors, ligand.py:303-329), then the type is added
    for (int f = 0; f < n_feat; ++f) {
        const int tf = rl(t, f), nf = rl(node_of, f);
        if ((newmask >> f) & 1) {
            const uint64_t new_lo = rl64(nam_lo, nf), new_hi = rl64(nam_hi, nf);
            const bool old_in_new = ((nam_lo & ~new_lo) | (nam_hi & ~new_hi)) == 0, new_in_old = ((new_lo & ~nam_lo) | (new_hi & ~nam_hi)) == 0;
            const bool older = lane < nf;
            // (the new node has ONE type, so one rule of the chain can apply)
            if (tf == 1) { // Aromatic: a hydrophobic atom of the ring depends on it
                if (older && (types & T_HYDROPHOBIC) && old_in_new && min_dep < 0) min_dep = nf;
            } else if (tf == 2 || tf == 3) { // ion: an H-bond atom of the group depends on it
                if (older && (types & T_HBOND) && old_in_new && min_dep < 0) min_dep = nf;
            } else if (tf == 0 || tf == 4 || tf == 5) { // the other way round: the new atom depends on the first ring / ion group that holds it
                const uint64_t holds = __ballot(older && (types & (tf == 0 ? T_AROMATIC : T_ION)) && new_in_old);
                if (holds && lane == nf) min_dep = __builtin_ctzll(holds);
            }
        }
        if (lane == nf) types |= 1 << tf;
    }
    // __group_nodes, functional groups (ligand.py:158-192): nodes whose atom has the same single heavy neighbour, per kind; everyone's group is
    // the others in node order
    int gkey = -1;
    if (lane < n && (types & (T_HBOND | T_HYDROPHOBIC))) {
        int count = 0, only = 0;
        for (int q = S.nbr_off[fatom]; q < S.nbr_off[fatom + 1]; ++q) {
            const int nb = S.nbr[q];
            if (S.z[nb] != 1) ++count, only = nb;
        }
        if (count == 1) gkey = ((types & T_HBOND) ? 0 : 128) + only;
    }
    uint64_t gm = 0; // the group as a set; S.glist holds its order
    for (int j = 0; j < n; ++j) {
        const int gj = rl(gkey, j);
        if (gkey >= 0 && gkey == gj && j != lane) gm |= 1ull << j;
    }
    int ng = 0;
    for (uint64_t m = gm; m; m &= m - 1) S.glist[lane][ng++] = (uint8_t)__builtin_ctzll(m);
    __syncthreads();

    // From here on: one sequence of steps for the wave. LDS is written by every lane alike (the same value to the same address).
    auto group_add = [&](int node, int idx) { // scalars
        if ((rl64(gm, node) >> idx) & 1) return;
        const int k = rl(ng, node);
        S.glist[node][k] = (uint8_t)idx;
        if (lane == node) gm |= 1ull << idx, ng = k + 1;
    };
    int verdict = 0;
    // __group_nodes, hydrophobic flood over carbon-carbon bonds (ligand.py:194-213). index_to_node: a dict built from node_dict["Hydrophobic"] - a
    // repeated key keeps its first position and takes the last value; popitem() is LIFO. Entry e lives in lane e.
    {
        uint64_t alive = 0;
        int n_ent = 0, ent_atom = 0, ent_node = 0;
        for (uint64_t m = __ballot(t == 0); m; m &= m - 1) {
            const int ni = rl(node_of, __builtin_ctzll(m));
            const int atom = rl(fatom, ni);
            const int w = uni(S.where[atom]);
            if (w < 0) {
                S.where[atom] = (int8_t)n_ent;
                if (lane == n_ent) ent_atom = atom, ent_node = ni;
                alive |= 1ull << n_ent;
                ++n_ent;
            } else if (lane == w) {
                ent_node = ni;
            }
        }
        while (alive && verdict == 0) {
            const int last = 63 - __builtin_clzll(alive);
            const int start = rl(ent_node, last);
            alive &= ~(1ull << last);
            S.where[rl(ent_atom, last)] = -1;
            int n_mem = 0;
            S.members[0] = (uint8_t)start;
            S.gindex[0] = (uint8_t)rl(fatom, start);
            n_mem = 1;
            const int ngs = rl(ng, start);
            for (int q = 0; q < ngs; ++q) {
                const int g = uni(S.glist[start][q]);
                S.members[n_mem] = (uint8_t)g;
                S.gindex[n_mem] = (uint8_t)rl(fatom, g);
                ++n_mem;
            }
            for (int gi = 0; gi < n_mem && verdict == 0; ++gi) { // grows while iterating
                const int ga = uni(S.gindex[gi]);
                const int q1 = uni(S.nbr_off[ga + 1]);
                for (int q = uni(S.nbr_off[ga]); q < q1; ++q) {
                    const int nb = uni(S.nbr[q]);
                    if (uni(S.z[nb]) != 6) continue;
                    const int at = uni(S.where[nb]);
                    if (at < 0) continue;
                    const int reached = rl(ent_node, at);
                    alive &= ~(1ull << at);
                    S.where[nb] = -1;
                    if (n_mem == kMaxMembers) {
                        verdict = 3;
                        break;
                    }
                    S.gindex[n_mem] = (uint8_t)nb;
                    for (int mq = 0; mq < n_mem; ++mq) {
                        const int mi = uni(S.members[mq]);
                        group_add(mi, reached);
                        group_add(reached, mi);
                    }
                    S.members[n_mem] = (uint8_t)reached;
                    ++n_mem;
                }
            }
        }
    }
    // __setup_cluster (ligand.py:215-259): the features type by type, in order; a node founds a cluster or joins one. Node lanes remember their
    // cluster and their place in it, cluster lanes their size, type and founder.
    int founder = -1, cl_of = 0, idx_in = 0; // node lane
    int csize = 0, ctype = 0, cfirst = 0;    // cluster lane
    int ncl = 0;
    uint64_t in_cluster = 0, founded = 0;
    auto new_cluster = [&](int ni, int type) {
        if (lane == ni) founder = ncl, cl_of = ncl, idx_in = 0;
        if (lane == ncl) csize = 1, ctype = type, cfirst = ni;
        founded |= 1ull << ni;
        ++ncl;
    };
    const int high_types[4] = {1, 2, 3, 6}; // Aromatic, Cation, Anion, Halogen
    const int high_ctype[4] = {C_AROMATIC, C_CATION, C_ANION, C_HALOGEN};
    for (int h = 0; h < 4; ++h)
        for (uint64_t m = __ballot(t == high_types[h]); m; m &= m - 1) {
            const int ni = rl(node_of, __builtin_ctzll(m));
            if ((in_cluster >> ni) & 1) continue;
            in_cluster |= 1ull << ni;
            new_cluster(ni, high_ctype[h]);
        }
    const int low_types[3] = {0, 4, 5}; // Hydrophobic, HBond_donor, HBond_acceptor
    for (int l = 0; l < 3 && verdict == 0; ++l)
        for (uint64_t m = __ballot(t == low_types[l]); m && verdict == 0; m &= m - 1) {
            const int ni = rl(node_of, __builtin_ctzll(m));
            if ((in_cluster >> ni) & 1) continue;
            in_cluster |= 1ull << ni;
            int join = -1;
            const int dep = rl(min_dep, ni);
            if (dep >= 0) {
                join = rl(founder, dep);
                if (join < 0) verdict = 2; // the reference's builder raises KeyError (ligand.py:238-241)
            } else if (rl64(gm, ni) & founded) { // the first of its group, in the group's order, that founded a cluster
                const int ngn = rl(ng, ni);
                for (int q = 0; q < ngn && join < 0; ++q) {
                    const int g = uni(S.glist[ni][q]);
                    if ((founded >> g) & 1) join = rl(founder, g);
                }
            }
            if (verdict) break;
            if (join >= 0) {
                const int sz = rl(csize, join);
                if (lane == ni) cl_of = join, idx_in = sz;
                if (lane == join) csize = sz + 1;
            } else {
                new_cluster(ni, l == 0 ? C_HYDROPHOBIC : C_HBOND);
            }
        }
    if (verdict == 0 && n_conf > PMX_MAX_CONFORMERS) verdict = 1;
    if (verdict == 0) {
        // stable sort by priority_fn (graph_match.py:43-60): (group, -size, subtype, first atom of the founder) - every cluster lane counts who
        // comes before it, and the nodes in front of its own
        const int fa = from_lane(fatom, cfirst);
        const int grp = ctype >= 3, sub = ctype >= 3 ? ctype - 3 : ctype;
        const int ckey = lane < ncl ? (grp << 24) | ((64 - csize) << 16) | (sub << 8) | fa : 0x7fffffff;
        int rank = 0, begin = 0;
        for (int d = 0; d < ncl; ++d) {
            const int kd = rl(ckey, d), sd = rl(csize, d);
            if (kd < ckey || (kd == ckey && d < lane)) ++rank, begin += sd;
        }
        PackDesc *out = desc + i;
        if (lane < ncl) out->ends[rank] = (uint8_t)(begin + csize);
        const int my_begin = from_lane(begin, cl_of);
        if (lane < n) {
            out->types[my_begin + idx_in] = (uint8_t)types;
            out->feat[my_begin + idx_in] = (uint8_t)cf;
        }
        if (lane == 0) out->n = (uint16_t)n, out->ncl = (uint16_t)ncl;
    }
    finish_molecule(lane, verdict, n_conf, desc + i, sizes + i, status + i);
}

// ------------------------------------------------------------------------------------------------- the general builder
// Molecules the wave builder left (status kTodo): staged into LDS byte arrays, built by lane 0 step by step.
__global__ __launch_bounds__(64) void graph_kernel(DevBatch b, PackDesc *desc, uint64_t *sizes, int32_t *status) {
    __shared__ PackLds S;
    const uint64_t i = blockIdx.x;
    if (status[i] != kTodo) return;
    const int lane = threadIdx.x;
    const uint64_t a0 = b.atom_off[i], a1 = b.atom_off[i + 1], f0 = b.feat_off[i], f1 = b.feat_off[i + 1];
    const int64_t n_atoms = (int64_t)(a1 - a0), n_feat = (int64_t)(f1 - f0); // (checked by graph_wave_kernel, like everything the staging reads)
    const int n_conf = b.n_conf[i];
    int verdict = 0;
    {
        const uint64_t n_nbr = b.nbr_off[a1] - b.nbr_off[a0];
        const uint64_t n_fa = n_feat ? b.feat_atom_off[f1] - b.feat_atom_off[f0] : 0;
        if (n_atoms > kMaxAtoms || n_feat > kMaxFeat || n_nbr > kMaxNbr || n_fa > kMaxFeatAtoms) verdict = 3;
    }
    if (verdict == 0) {
        const uint64_t nb0 = b.nbr_off[a0], fa0 = b.feat_atom_off[f0];
        for (int a = lane; a <= (int)n_atoms; a += 64) S.nbr_off[a] = (uint16_t)(b.nbr_off[a0 + a] - nb0);
        for (int a = lane; a < (int)n_atoms; a += 64) S.z[a] = b.atomic_num[a0 + a];
        const int n_nbr = (int)(b.nbr_off[a1] - nb0);
        for (int q = lane; q < n_nbr; q += 64) S.nbr[q] = (uint8_t)b.nbr[nb0 + q];
        for (int f = lane; f <= (int)n_feat; f += 64) S.fatom_off[f] = (uint16_t)(b.feat_atom_off[f0 + f] - fa0);
        for (int f = lane; f < (int)n_feat; f += 64) {
            S.ftype[f] = b.feat_type[f0 + f];
            S.fflags[f] = b.feat_flags[f0 + f];
        }
        const int n_fa = (int)(b.feat_atom_off[f1] - fa0);
        for (int q = lane; q < n_fa; q += 64) S.fatoms[q] = (uint8_t)b.feat_atoms[fa0 + q];
        for (int a = lane; a < kMaxAtoms; a += 64) {
            S.where[a] = -1;
            S.grp_first[0][a] = S.grp_first[1][a] = S.grp_last[0][a] = S.grp_last[1][a] = -1;
        }
        S.grp_next[0][lane] = S.grp_next[1][lane] = -1;
        S.cl_next[lane] = S.founder[lane] = -1;
        S.in_cluster[lane] = 0;
        __syncthreads();
        if (lane == 0) S.bad = build_graph(S, (int)n_feat, n_conf, desc + i);
        __syncthreads();
        verdict = S.bad;
    }
    finish_molecule(lane, verdict, n_conf, desc + i, sizes + i, status + i);
}

__global__ __launch_bounds__(64) void record_kernel(DevBatch b, const PackDesc *desc, const uint64_t *offsets, uint8_t *data) {
    const uint64_t i = blockIdx.x;
    const int lane = threadIdx.x;
    const PackDesc &d = desc[i];
    uint8_t *out = data + offsets[i];
    if (d.n == 0xFFFF) {
        if (lane < 4) reinterpret_cast<uint32_t *>(out)[lane] = 0;
        return;
    }
    const int n = d.n, ncl = d.ncl, C = b.n_conf[i];
    const int head = 8 + n + ncl, head4 = (head + 3) & ~3;
    // header, type masks, cluster ends, padding: as 32-bit words (the record is 16-byte aligned)
    for (int w = lane; w < head4 / 4; w += 64) {
        uint32_t word = 0;
        for (int k = 0; k < 4; ++k) {
            const int at = w * 4 + k;
            uint32_t byte = 0;
            if (at < 8) byte = at == 0 ? (n & 255) : at == 1 ? (n >> 8) : at == 2 ? (C & 255) : at == 3 ? (C >> 8) : at == 4 ? (ncl & 255) : at == 5 ? (ncl >> 8) : 0;
            else if (at < 8 + n) byte = d.types[at - 8];
            else if (at < head) byte = d.ends[at - 8 - n];
            word |= byte << (8 * k);
        }
        reinterpret_cast<uint32_t *>(out)[w] = word;
    }
    float *xyz = reinterpret_cast<float *>(out + head4);
    const uint64_t f0 = b.feat_off[i];
    const float *pos = b.positions + b.pos_off[i];
    // per packed node, once: where its centre atoms are listed and how many are averaged (1: the position of the first, as it is)
    __shared__ uint64_t s_first[64];
    __shared__ uint32_t s_count[64];
    if (lane < n) {
        const uint64_t f = f0 + d.feat[lane];
        const uint64_t c0 = b.feat_center_off[f];
        s_first[lane] = c0;
        s_count[lane] = ((b.feat_flags[f] >> 1) & 1) ? (uint32_t)(b.feat_center_off[f + 1] - c0) : 1u;
    }
    __syncthreads();
    const uint32_t per_node = 3u * (uint32_t)C, total = (uint32_t)n * per_node;
    const uint32_t padded = (uint32_t)((record_bytes(n, ncl, C) - head4) / 4);
    // exact division of e < 2^16 by per_node <= 192 and by C <= 64: multiply by floor(2^32 / d) + 1, keep the high word (d = 1 has no such multiplier in 32 bits)
    const uint32_t m_node = 0xFFFFFFFFu / per_node + 1u, m_conf = 0xFFFFFFFFu / (uint32_t)C + 1u;
    for (uint32_t e = lane; e < padded; e += 64) {
        float v = 0.f;
        if (e < total) {
            const uint32_t p = __umulhi(e, m_node), r = e - p * per_node, dd = C == 1 ? r : __umulhi(r, m_conf), c = r - dd * (uint32_t)C;
            const uint64_t c0 = s_first[p];
            const uint32_t cnt = s_count[p];
            const size_t at = (size_t)c * 3 + dd;
            v = pos[(size_t)b.feat_centers[c0] * per_node + at];
            if (cnt > 1) { // float32 mean over the centre atoms, atom after atom, then one division
                for (uint32_t q = 1; q < cnt; ++q) v = v + pos[(size_t)b.feat_centers[c0 + q] * per_node + at];
                v = __fdiv_rn(v, (float)cnt);
            }
        }
        xyz[e] = v;
    }
}

__global__ void close_offsets_kernel(const uint64_t *sizes, uint64_t *offsets, uint64_t n) { // offsets[n] = offsets[n - 1] + sizes[n - 1]
    if (threadIdx.x == 0 && blockIdx.x == 0) offsets[n] = n ? offsets[n - 1] + sizes[n - 1] : 0;
}

// Work buffers (descriptors, sizes, scan scratch) kept per device from call to call; a call holds the lock while it runs.
struct PackWork {
    std::mutex mu;
    int device = -1;
    void *desc = nullptr, *sizes = nullptr, *scan = nullptr, *status = nullptr;
    size_t desc_bytes = 0, sizes_bytes = 0, scan_bytes = 0, status_bytes = 0;
    hipEvent_t done = nullptr;     // behind the last call's record writer (it reads `desc` after the call has returned)
    hipStream_t last = nullptr;    // the stream that call was made on
    bool pending = false;
};
PackWork g_work;

bool grow(void **p, size_t *have, size_t need) {
    if (*have >= need) return true;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    if (hipMalloc(p, need) != hipSuccess) return false;
    *have = need;
    return true;
}

} // namespace

extern "C" int pmx_pack_features_device(const pmx_feature_batch *b, int device, void *stream_, uint64_t *offsets_out_dev, uint8_t *data_out_dev,
                                        uint64_t data_cap, uint64_t *data_bytes, int32_t *status_out_dev) {
    if (!b || !offsets_out_dev || (!data_out_dev && data_cap) || !data_bytes) return pmx_topk_fail(PMX_ERR_INVALID, "null argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (hipSetDevice(device) != hipSuccess) return pmx_topk_fail(PMX_ERR_HIP, "pmx_pack_features_device: hipSetDevice failed");
    const uint64_t n = b->n_mols;
    *data_bytes = 0;
    if (n == 0) {
        if (hipMemsetAsync(offsets_out_dev, 0, 8, stream) != hipSuccess) return pmx_topk_fail(PMX_ERR_HIP, "pmx_pack_features_device: memset failed");
        return PMX_OK;
    }
    if (n > 0x7fffffffull) return pmx_topk_fail(PMX_ERR_INVALID, "pmx_pack_features_device: more than 2^31 - 1 molecules in one call");
    std::lock_guard<std::mutex> lock(g_work.mu);
    if (g_work.device != device) { // (buffers belong to the device they were allocated on)
        if (g_work.device >= 0) {
            (void)hipSetDevice(g_work.device);
            (void)hipFree(g_work.desc), (void)hipFree(g_work.sizes), (void)hipFree(g_work.scan), (void)hipFree(g_work.status);
            (void)hipSetDevice(device);
        }
        g_work.desc = g_work.sizes = g_work.scan = g_work.status = nullptr;
        g_work.desc_bytes = g_work.sizes_bytes = g_work.scan_bytes = g_work.status_bytes = 0;
        if (g_work.done) (void)hipEventDestroy(g_work.done);
        g_work.done = nullptr;
        g_work.pending = false;
        g_work.device = device;
    }
    // The buffers are shared by all calls: one made on another stream than the last starts behind that call's record writer.
    if (!g_work.done && hipEventCreateWithFlags(&g_work.done, hipEventDisableTiming) != hipSuccess) return pmx_topk_fail(PMX_ERR_HIP, "pmx_pack_features_device: hipEventCreate failed");
    if (g_work.pending && g_work.last != stream && hipStreamWaitEvent(stream, g_work.done, 0) != hipSuccess)
        return pmx_topk_fail(PMX_ERR_HIP, "pmx_pack_features_device: hipStreamWaitEvent failed");
    size_t scan_need = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_need, (const uint64_t *)nullptr, (uint64_t *)nullptr, (int)n, stream);
    if (!grow(&g_work.desc, &g_work.desc_bytes, n * sizeof(PackDesc)) || !grow(&g_work.sizes, &g_work.sizes_bytes, n * 8) ||
        !grow(&g_work.scan, &g_work.scan_bytes, scan_need ? scan_need : 8) || (!status_out_dev && !grow(&g_work.status, &g_work.status_bytes, n * 4)))
        return pmx_topk_fail(PMX_ERR_OOM, "pmx_pack_features_device: out of device memory");
    DevBatch d{n, b->atom_off, b->atomic_num, b->nbr_off, b->nbr, b->feat_off, b->feat_type, b->feat_flags, b->feat_atom_off, b->feat_atoms,
               b->feat_center_off, b->feat_centers, b->n_conf, b->pos_off, b->positions};
    PackDesc *desc = static_cast<PackDesc *>(g_work.desc);
    uint64_t *sizes = static_cast<uint64_t *>(g_work.sizes);
    int32_t *status = status_out_dev ? status_out_dev : static_cast<int32_t *>(g_work.status);
    graph_wave_kernel<<<dim3((unsigned)n), dim3(64), 0, stream>>>(d, desc, sizes, status);
    graph_kernel<<<dim3((unsigned)n), dim3(64), 0, stream>>>(d, desc, sizes, status); // (a block whose molecule is done returns at once)
    size_t scan_bytes = g_work.scan_bytes;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(g_work.scan, scan_bytes, sizes, offsets_out_dev, (int)n, stream);
    if (e == hipSuccess) {
        close_offsets_kernel<<<1, 64, 0, stream>>>(sizes, offsets_out_dev, n);
        e = hipGetLastError();
    }
    uint64_t total = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&total, offsets_out_dev + n, 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return pmx_topk_fail(PMX_ERR_HIP, hipGetErrorString(e));
    *data_bytes = total;
    if (!data_out_dev) return PMX_OK; // sizing call: the exact size
    if (total > data_cap) return pmx_topk_fail(PMX_ERR_INVALID, "data_out too small (data_bytes holds the size needed)");
    record_kernel<<<dim3((unsigned)n), dim3(64), 0, stream>>>(d, desc, offsets_out_dev, data_out_dev);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(g_work.done, stream);
    if (e != hipSuccess) return pmx_topk_fail(PMX_ERR_HIP, hipGetErrorString(e));
    g_work.last = stream;
    g_work.pending = true;
    return PMX_OK;
}
