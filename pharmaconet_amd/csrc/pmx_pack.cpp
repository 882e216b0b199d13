// pmx_pack.cpp - the library packer in native code: LigandGraph's node merging, grouping and clustering
// (src/pmnet/scoring/ligand.py:110-259) and the priority sort of graph_match.py:43-60, from perceived pharmacophore
// features to records of the packed library format (pharmaconet_amd/library.py). Host code only; it is the step right
// in front of the scoring kernels and has to keep up with them.
//
// This restates pharmaconet_amd.library.cluster_ligand / pack_clustered_ligand step by step - including what follows from
// Python container semantics in the reference (dict insertion order and LIFO popitem, `atom_indices` keys where an int
// and a 1-tuple differ) - so that records are byte-identical to the ones extracted from the reference's own LigandGraph
// (tests/test_library.py on 708 fixture molecules).
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "pmx.h"

#ifdef PMX_PACK_STANDALONE
// libpmx_pack.so: this file alone, for host processes that pack libraries and never touch a GPU (no HIP / RCCL runtime is
// loaded with it). It brings its own error string.
#include <cstdio>
static thread_local char g_pack_err[256] = "";
static int pmx_topk_fail(int code, const char *msg) {
    std::snprintf(g_pack_err, sizeof(g_pack_err), "%s", msg);
    return code;
}
extern "C" const char *pmx_last_error(void) { return g_pack_err; }
extern "C" int pmx_version(void) { return 100; }
#else
int pmx_topk_fail(int code, const char *msg); // error hook in pmx_api.hip
#endif

namespace {

enum : unsigned { T_HYDROPHOBIC = 1u << 0, T_AROMATIC = 1u << 1, T_CATION = 1u << 2, T_ANION = 1u << 3, T_DONOR = 1u << 4, T_ACCEPTOR = 1u << 5, T_HALOGEN = 1u << 6 };
constexpr unsigned T_HBOND = T_DONOR | T_ACCEPTOR, T_ION = T_CATION | T_ANION;

// cluster types in the order of CLUSTER_PRIORITY (constants.py): (group, subtype) of priority_fn (graph_match.py:43-60)
enum ClusterType { C_AROMATIC, C_CATION, C_ANION, C_HBOND, C_HALOGEN, C_HYDROPHOBIC };
constexpr int kPriorityGroup[6] = {0, 0, 0, 1, 1, 1}, kPrioritySub[6] = {0, 1, 2, 0, 1, 2};

struct Node {
    unsigned types = 0;
    std::vector<int> atoms;   // sorted, unique (frozenset)
    bool center_is_tuple = false;
    std::vector<int> centers; // as given
    std::vector<int> group;   // insertion-ordered set of node indices (may hold the node itself)
    int min_dependence = -1;  // min(node.dependence), -1 = empty
};

inline bool subset(const std::vector<int> &a, const std::vector<int> &b) { return std::includes(b.begin(), b.end(), a.begin(), a.end()); }
inline void add_dep(Node &n, int idx) { n.min_dependence = n.min_dependence < 0 ? idx : std::min(n.min_dependence, idx); }
inline void group_add(Node &n, int idx) {
    if (std::find(n.group.begin(), n.group.end(), idx) == n.group.end()) n.group.push_back(idx);
}

struct Mol {
    int n_atoms, n_conf, n_feat;
    const uint8_t *z;
    const uint64_t *nbr_off; // [n_atoms + 1], absolute offsets into nbr
    const int32_t *nbr;
    const uint8_t *ftype, *fflags;
    const uint64_t *fatom_off, *fcenter_off; // [n_feat + 1], absolute
    const int32_t *fatoms, *fcenters;
    const float *pos; // [n_atoms][n_conf][3]
};

// One molecule -> one record at `out` (capacity `cap`); returns the record size, 0 if it does not fit the structural
// limits of include/pmx.h (the caller emits the header-only "unsupported" record), or -1 if cap is too small.
// Per-thread scratch reused from molecule to molecule: after the first few molecules packing allocates nothing (the
// packer runs on every host core at once; a general-purpose allocator would serialise them).
struct Scratch {
    std::vector<Node> pool; // node objects with their vectors' capacity kept
    std::vector<int> node_dict[PMX_NUM_TYPES];
    std::vector<std::pair<int, std::vector<int>>> keys; // (is_tuple, atom indices as given) of node i
    // Everything below stands in for the dicts / lists / sets the reference builds per molecule. Nothing is allocated per
    // molecule once the vectors have grown to the largest molecule seen: the packer runs on every host core at once, and with
    // some sixty small allocations per molecule the threads did not speed each other up at all (one to eight threads: 1.2x).
    std::vector<int> grp_first[2], grp_last[2]; // per atom: first / last node of the functional group keyed by that atom (H-bond | hydrophobic)
    std::vector<int> grp_next[2];               // per node: next member of its group list
    std::vector<std::pair<int, int>> entries;   // index_to_node: (atom, node) in insertion order
    std::vector<char> alive;
    std::vector<int> where;                     // per atom: position in `entries`, -1 = absent
    std::vector<int> members, group_index;
    std::vector<int> cl_first, cl_last, cl_size, cl_next; // clusters as linked lists of nodes (cl_next per node)
    std::vector<int> ctype, founder, order;
    std::vector<char> in_cluster;
};

int64_t pack_one(const Mol &m, uint8_t *out, uint64_t cap) {
    static thread_local Scratch S;
    if ((int)S.pool.size() < m.n_feat) S.pool.resize(m.n_feat);
    if ((int)S.keys.size() < m.n_feat) S.keys.resize(m.n_feat);
    std::vector<Node> &nodes = S.pool; // nodes [0, n)
    int n = 0;
    std::vector<int> *node_dict = S.node_dict; // nodes of each type in order of appearance (duplicates possible)
    for (int t = 0; t < PMX_NUM_TYPES; ++t) node_dict[t].clear();
    // __add_nodes (ligand.py:134-156)
    for (int f = 0; f < m.n_feat; ++f) {
        const int t = m.ftype[f];
        const bool key_tuple = m.fflags[f] & 1;
        const int32_t *kb = m.fatoms + m.fatom_off[f], *ke = m.fatoms + m.fatom_off[f + 1];
        int found = -1;
        for (int i = 0; i < n && found < 0; ++i) // by_key lookup: same tuple-ness, same indices in the same order
            if (S.keys[i].first == (int)key_tuple && S.keys[i].second.size() == (size_t)(ke - kb) && std::equal(kb, ke, S.keys[i].second.begin())) found = i;
        if (found >= 0) {
            nodes[found].types |= 1u << t;
            node_dict[t].push_back(found);
            continue;
        }
        const int ni = n;
        Node &nw = nodes[ni];
        nw.types = 1u << t;
        nw.atoms.assign(kb, ke);
        std::sort(nw.atoms.begin(), nw.atoms.end());
        nw.atoms.erase(std::unique(nw.atoms.begin(), nw.atoms.end()), nw.atoms.end());
        nw.center_is_tuple = (m.fflags[f] >> 1) & 1;
        nw.centers.assign(m.fcenters + m.fcenter_off[f], m.fcenters + m.fcenter_off[f + 1]);
        nw.group.clear();
        nw.min_dependence = -1;
        S.keys[ni].first = (int)key_tuple;
        S.keys[ni].second.assign(kb, ke);
        for (int oi = 0; oi < ni; ++oi) { // old.add_neighbors(new) (ligand.py:303-329); old's types as they are NOW
            Node &old = nodes[oi];
            if ((old.types & T_HYDROPHOBIC) && (nw.types & T_AROMATIC)) {
                if (subset(old.atoms, nw.atoms)) add_dep(old, ni);
            } else if ((old.types & T_AROMATIC) && (nw.types & T_HYDROPHOBIC)) {
                if (subset(nw.atoms, old.atoms)) add_dep(nw, oi);
            } else if ((old.types & T_HBOND) && (nw.types & T_ION)) {
                if (subset(old.atoms, nw.atoms)) add_dep(old, ni);
            } else if ((old.types & T_ION) && (nw.types & T_HBOND)) {
                if (subset(nw.atoms, old.atoms)) add_dep(nw, oi);
            }
        }
        ++n;
        node_dict[t].push_back(ni);
    }
    auto heavy_nbrs = [&](int atom) { return std::make_pair(m.nbr + m.nbr_off[atom], m.nbr + m.nbr_off[atom + 1]); };

    // __group_nodes, functional groups (ligand.py:158-192): atoms bonded to the same single heavy neighbour. (The reference keeps
    // a dict heavy neighbour -> member list per kind; only lookups by key are made, so per-atom list heads do.)
    {
        for (int k = 0; k < 2; ++k) {
            S.grp_first[k].assign((size_t)m.n_atoms, -1);
            S.grp_last[k].assign((size_t)m.n_atoms, -1);
            S.grp_next[k].assign((size_t)n, -1);
        }
        for (int i = 0; i < n; ++i) {
            Node &nd = nodes[i];
            int kind;
            if (nd.types & T_HBOND) kind = 0;
            else if (nd.types & T_HYDROPHOBIC) kind = 1;
            else continue;
            const int atom = nd.atoms[0];
            auto nb = heavy_nbrs(atom);
            int count = 0, only = -1;
            for (const int32_t *q = nb.first; q != nb.second; ++q)
                if (m.z[*q] != 1) {
                    ++count;
                    only = *q;
                }
            if (count == 1) {
                for (int other = S.grp_first[kind][only]; other >= 0; other = S.grp_next[kind][other]) {
                    group_add(nd, other);
                    group_add(nodes[other], i);
                }
                if (S.grp_last[kind][only] >= 0) S.grp_next[kind][S.grp_last[kind][only]] = i;
                else S.grp_first[kind][only] = i;
                S.grp_last[kind][only] = i;
            }
        }
    }
    // __group_nodes, hydrophobic flood over carbon-carbon bonds (ligand.py:194-213). index_to_node is a dict built from
    // node_dict["Hydrophobic"]: a repeated key keeps its first position and takes the last value; popitem() is LIFO.
    {
        std::vector<std::pair<int, int>> &entries = S.entries; // (atom, node), insertion order
        std::vector<char> &alive = S.alive;
        std::vector<int> &where = S.where;
        entries.clear();
        alive.clear();
        where.assign((size_t)m.n_atoms, -1);
        for (int ni : node_dict[0]) {
            const int atom = nodes[ni].atoms[0];
            if (where[atom] < 0) {
                where[atom] = (int)entries.size();
                entries.emplace_back(atom, ni);
                alive.push_back(1);
            } else {
                entries[where[atom]].second = ni;
            }
        }
        int last = (int)entries.size() - 1;
        std::vector<int> &members = S.members, &group_index = S.group_index;
        for (;;) {
            while (last >= 0 && !alive[last]) --last;
            if (last < 0) break;
            const int start = entries[last].second;
            alive[last] = 0;
            where[entries[last].first] = -1;
            members.clear();
            members.push_back(start);
            for (int g : nodes[start].group) members.push_back(g);
            group_index.clear();
            for (int mi : members) group_index.push_back(nodes[mi].atoms[0]);
            for (size_t gi = 0; gi < group_index.size(); ++gi) { // grows while iterating
                auto nb = heavy_nbrs(group_index[gi]);
                for (const int32_t *q = nb.first; q != nb.second; ++q) {
                    if (m.z[*q] != 6) continue;
                    const int at = where[*q];
                    if (at < 0) continue;
                    const int reached = entries[at].second;
                    alive[at] = 0;
                    where[*q] = -1;
                    group_index.push_back(*q);
                    for (int mi : members) {
                        group_add(nodes[mi], reached);
                        group_add(nodes[reached], mi);
                    }
                    members.push_back(reached);
                }
            }
        }
    }
    // __setup_cluster (ligand.py:215-259); a cluster is a list of nodes in the order they joined
    std::vector<int> &cl_first = S.cl_first, &cl_last = S.cl_last, &cl_size = S.cl_size, &cl_next = S.cl_next;
    std::vector<int> &ctype = S.ctype, &founder = S.founder; // founder: node -> cluster it founded
    std::vector<char> &in_cluster = S.in_cluster;
    cl_first.clear(), cl_last.clear(), cl_size.clear(), ctype.clear();
    cl_next.assign((size_t)n, -1);
    founder.assign((size_t)n, -1);
    in_cluster.assign((size_t)n, 0);
    auto new_cluster = [&](int ni, int type) {
        cl_first.push_back(ni), cl_last.push_back(ni), cl_size.push_back(1), ctype.push_back(type);
        founder[ni] = (int)cl_first.size() - 1;
    };
    auto join_cluster = [&](int ci, int ni) {
        cl_next[cl_last[ci]] = ni;
        cl_last[ci] = ni;
        ++cl_size[ci];
    };
    const int high_types[4] = {1, 2, 3, 6}; // Aromatic, Cation, Anion, Halogen
    const int high_ctype[4] = {C_AROMATIC, C_CATION, C_ANION, C_HALOGEN};
    for (int h = 0; h < 4; ++h)
        for (int ni : node_dict[high_types[h]]) {
            if (in_cluster[ni]) continue;
            in_cluster[ni] = 1;
            new_cluster(ni, high_ctype[h]);
        }
    const int low_types[3] = {0, 4, 5}; // Hydrophobic, HBond_donor, HBond_acceptor
    for (int l = 0; l < 3; ++l)
        for (int ni : node_dict[low_types[l]]) {
            if (in_cluster[ni]) continue;
            in_cluster[ni] = 1;
            Node &nd = nodes[ni];
            bool add_new = true;
            if (nd.min_dependence >= 0 && founder[nd.min_dependence] >= 0) { // (the reference would raise KeyError otherwise)
                join_cluster(founder[nd.min_dependence], ni);
                add_new = false;
            } else if (nd.min_dependence >= 0) {
                return -2; // a node that depends on a node without a cluster: the reference's builder raises KeyError (ligand.py:238-241)
            } else {
                for (int g : nd.group)
                    if (founder[g] >= 0) {
                        join_cluster(founder[g], ni);
                        add_new = false;
                        break;
                    }
            }
            if (add_new) new_cluster(ni, l == 0 ? C_HYDROPHOBIC : C_HBOND);
        }
    // pack_clustered_ligand: stable sort by priority_fn (graph_match.py:43-60), nodes renumbered cluster by cluster
    const int ncl = (int)cl_first.size();
    std::vector<int> &order = S.order;
    order.resize((size_t)ncl);
    for (int i = 0; i < ncl; ++i) order[i] = i;
    auto before = [&](int a, int b) {
        const int ka[4] = {kPriorityGroup[ctype[a]], -cl_size[a], kPrioritySub[ctype[a]], nodes[cl_first[a]].atoms[0]};
        const int kb[4] = {kPriorityGroup[ctype[b]], -cl_size[b], kPrioritySub[ctype[b]], nodes[cl_first[b]].atoms[0]};
        return std::lexicographical_compare(ka, ka + 4, kb, kb + 4);
    };
    for (int i = 1; i < ncl; ++i) { // stable insertion sort (std::stable_sort allocates a buffer per call)
        const int x = order[i];
        int j = i;
        for (; j > 0 && before(x, order[j - 1]); --j) order[j] = order[j - 1];
        order[j] = x;
    }
    if (n > PMX_MAX_LIGAND_NODES || ncl > PMX_MAX_LIGAND_CLUSTERS || m.n_conf < 1 || m.n_conf > PMX_MAX_CONFORMERS) return 0;
    const int C = m.n_conf;
    const uint64_t head = 8 + (uint64_t)n + (uint64_t)ncl;
    const uint64_t body = ((head + 3) & ~3ull) + 12ull * n * C;
    const uint64_t total = (body + 15) & ~15ull;
    if (total > cap) return -1;
    std::memset(out, 0, total);
    uint16_t h16[4] = {(uint16_t)n, (uint16_t)C, (uint16_t)ncl, 0};
    std::memcpy(out, h16, 8);
    uint8_t *tm = out + 8, *ends = out + 8 + n;
    float *xyz = reinterpret_cast<float *>(out + ((head + 3) & ~3ull));
    int pos = 0;
    for (int ci = 0; ci < ncl; ++ci) {
        for (int ni = cl_first[order[ci]]; ni >= 0; ni = cl_next[ni]) {
            const Node &nd = nodes[ni];
            tm[pos] = (uint8_t)nd.types;
            float *dst = xyz + (size_t)pos * 3 * C; // [3][C]
            if (!nd.center_is_tuple) { // LigandNode.set_positions (ligand.py:293-301)
                const float *src = m.pos + (size_t)nd.centers[0] * C * 3;
                for (int c = 0; c < C; ++c)
                    for (int d = 0; d < 3; ++d) dst[d * C + c] = src[c * 3 + d];
            } else { // float32 mean over the centre atoms, atom after atom, then one division
                const float cnt = (float)nd.centers.size();
                for (int c = 0; c < C; ++c)
                    for (int d = 0; d < 3; ++d) {
                        float sum = 0.f;
                        bool first = true;
                        for (int a : nd.centers) {
                            const float v = m.pos[((size_t)a * C + c) * 3 + d];
                            sum = first ? v : sum + v;
                            first = false;
                        }
                        dst[d * C + c] = sum / cnt;
                    }
            }
            ++pos;
        }
        ends[ci] = (uint8_t)pos;
    }
    return (int64_t)total;
}

} // namespace

// The raw arrays of one molecule, checked before the graph builder indexes with them: type ids, atom / centre / neighbour
// indices inside the molecule, features with at least one atom, monotonic offsets, a sane conformer count.
static bool valid_molecule(const pmx_feature_batch *b, uint64_t i, const Mol &m) {
    if (m.n_atoms < 0 || m.n_feat < 0 || m.n_conf < 1 || m.n_conf > 1 << 16) return false;
    const uint64_t need_pos = (uint64_t)m.n_atoms * (uint64_t)m.n_conf * 3;
    if (b->pos_off[i + 1] - b->pos_off[i] < need_pos) return false;
    for (int a = 0; a < m.n_atoms; ++a) {
        if (m.nbr_off[a + 1] < m.nbr_off[a]) return false;
        for (uint64_t q = m.nbr_off[a]; q < m.nbr_off[a + 1]; ++q)
            if (m.nbr[q] < 0 || m.nbr[q] >= m.n_atoms) return false;
    }
    for (int f = 0; f < m.n_feat; ++f) {
        if (m.ftype[f] >= PMX_NUM_TYPES) return false;
        if (m.fatom_off[f + 1] <= m.fatom_off[f] || m.fcenter_off[f + 1] <= m.fcenter_off[f]) return false; // at least one atom / centre
        for (uint64_t q = m.fatom_off[f]; q < m.fatom_off[f + 1]; ++q)
            if (m.fatoms[q] < 0 || m.fatoms[q] >= m.n_atoms) return false;
        for (uint64_t q = m.fcenter_off[f]; q < m.fcenter_off[f + 1]; ++q)
            if (m.fcenters[q] < 0 || m.fcenters[q] >= m.n_atoms) return false;
    }
    return true;
}

static int pack_features_impl(const pmx_feature_batch *b, int threads, uint64_t *offsets_out, uint8_t *data_out, uint64_t data_cap,
                              uint64_t *data_bytes, int32_t *status_out);

// One cached staging buffer per process (the call that finds it in use allocates its own and frees it afterwards).
namespace {
struct StagingCache {
    std::mutex mu;
    uint8_t *mem = nullptr;
    size_t bytes = 0;
    bool busy = false;
};
StagingCache g_staging;
constexpr size_t kHuge = (size_t)2 << 20;
uint8_t *huge_alloc(size_t bytes) {
    const size_t rounded = (bytes + kHuge - 1) & ~(kHuge - 1);
    void *p = nullptr;
    if (posix_memalign(&p, kHuge, rounded) != 0) return nullptr;
    (void)madvise(p, rounded, MADV_HUGEPAGE);
    return static_cast<uint8_t *>(p);
}
struct StagingArea {
    uint8_t *own = nullptr;
    bool cached = false;
    uint8_t *acquire(size_t bytes) {
        {
            std::lock_guard<std::mutex> lock(g_staging.mu);
            if (!g_staging.busy) {
                if (g_staging.bytes < bytes) {
                    std::free(g_staging.mem);
                    g_staging.mem = huge_alloc(bytes);
                    g_staging.bytes = g_staging.mem ? bytes : 0;
                }
                if (g_staging.mem) {
                    g_staging.busy = cached = true;
                    return g_staging.mem;
                }
            }
        }
        return own = huge_alloc(bytes);
    }
    ~StagingArea() {
        if (cached) {
            std::lock_guard<std::mutex> lock(g_staging.mu);
            g_staging.busy = false;
        }
        std::free(own);
    }
};
} // namespace

extern "C" int pmx_pack_features(const pmx_feature_batch *b, int threads, uint64_t *offsets_out, uint8_t *data_out, uint64_t data_cap,
                                 uint64_t *data_bytes, int32_t *status_out) {
    try { // no exception crosses the C boundary
        return pack_features_impl(b, threads, offsets_out, data_out, data_cap, data_bytes, status_out);
    } catch (const std::bad_alloc &) {
        return pmx_topk_fail(PMX_ERR_OOM, "pmx_pack_features: out of host memory");
    } catch (...) {
        return pmx_topk_fail(PMX_ERR_INVALID, "pmx_pack_features: internal error");
    }
}

static int pack_features_impl(const pmx_feature_batch *b, int threads, uint64_t *offsets_out, uint8_t *data_out, uint64_t data_cap,
                              uint64_t *data_bytes, int32_t *status_out) {
    if (!b || !offsets_out || (!data_out && data_cap) || !data_bytes) return pmx_topk_fail(PMX_ERR_INVALID, "null argument");
    for (uint64_t i = 0; i < b->n_mols; ++i) // offset arrays must not run backwards (everything else is checked per molecule)
        if (b->atom_off[i + 1] < b->atom_off[i] || b->feat_off[i + 1] < b->feat_off[i] || b->pos_off[i + 1] < b->pos_off[i])
            return pmx_topk_fail(PMX_ERR_INVALID, "pmx_pack_features: offsets run backwards");
    const uint64_t n = b->n_mols;
    // every record gets the worst-case room of its molecule first (features x conformers), then the records are compacted
    std::vector<uint64_t> room(n + 1, 0);
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t nf = b->feat_off[i + 1] - b->feat_off[i];
        const uint64_t c = (uint64_t)std::max(b->n_conf[i], 1);
        room[i + 1] = room[i] + ((8 + 2 * nf + 3 + 12 * nf * c + 15) & ~15ull) + 16;
    }
    if (!data_out) { // sizing call: the bound, without packing anything
        *data_bytes = room[n];
        return PMX_OK;
    }
    // The staging area (worst-case room per molecule; not zero-filled: pack_one clears what it writes). A million molecules take
    // gigabytes of it, and what the packer used to spend its time on was not packing but first-touch page faults of a fresh
    // allocation per call (no speed-up at all from 1 to 8 threads: they queue on the address space lock): the area is kept
    // from call to call, 2 MB aligned and advised as huge pages (512 times fewer faults where the kernel grants them).
    StagingArea staging;
    uint8_t *scratch = staging.acquire(room[n] + 16);
    if (!scratch) return pmx_topk_fail(PMX_ERR_OOM, "pmx_pack_features: out of host memory");
    std::vector<int64_t> sizes(n, 0);
    std::atomic<uint64_t> next{0}, bad_input{0};
    auto work = [&]() {
        for (;;) {
            const uint64_t i0 = next.fetch_add(256);
            if (i0 >= n) break;
            for (uint64_t i = i0; i < std::min(n, i0 + 256); ++i) {
                Mol m;
                const uint64_t a0 = b->atom_off[i];
                m.n_atoms = (int)(b->atom_off[i + 1] - a0);
                m.n_conf = b->n_conf[i];
                m.n_feat = (int)(b->feat_off[i + 1] - b->feat_off[i]);
                m.z = b->atomic_num + a0;
                m.nbr_off = b->nbr_off + a0;
                m.nbr = b->nbr;
                m.ftype = b->feat_type + b->feat_off[i];
                m.fflags = b->feat_flags + b->feat_off[i];
                m.fatom_off = b->feat_atom_off + b->feat_off[i];
                m.fcenter_off = b->feat_center_off + b->feat_off[i];
                m.fatoms = b->feat_atoms;
                m.fcenters = b->feat_centers;
                m.pos = b->positions + b->pos_off[i];
                int64_t sz = -2;
                try {
                    if (valid_molecule(b, i, m)) sz = pack_one(m, scratch + room[i], room[i + 1] - room[i]);
                } catch (...) { // (allocation failure inside the graph builder: report the molecule, never unwind a thread)
                    sz = -2;
                }
                if (sz <= 0) { // outside the structural limits, or malformed input: header-only record, reported per ligand
                    std::memset(scratch + room[i], 0, 16);
                    if (status_out) status_out[i] = sz == -2 ? 2 : 1;
                    if (sz == -2) bad_input.fetch_add(1);
                    sz = 16;
                } else if (status_out) {
                    status_out[i] = 0;
                }
                sizes[i] = sz;
            }
        }
    };
    const int nt = std::max(1, std::min(threads, 256));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        offsets_out[i] = total;
        total += (uint64_t)sizes[i];
    }
    offsets_out[n] = total;
    *data_bytes = total;
    if (total > data_cap) return pmx_topk_fail(PMX_ERR_INVALID, "data_out too small (data_bytes holds the size needed)");
    next = 0;
    auto compact = [&]() {
        for (;;) {
            const uint64_t i0 = next.fetch_add(1024);
            if (i0 >= n) break;
            for (uint64_t i = i0; i < std::min(n, i0 + 1024); ++i) std::memcpy(data_out + offsets_out[i], scratch + room[i], (size_t)sizes[i]);
        }
    };
    pool.clear();
    for (int t = 1; t < nt; ++t) pool.emplace_back(compact);
    compact();
    for (auto &t : pool) t.join();
    return PMX_OK;
}
