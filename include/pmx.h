/*
 * pmx.h - C ABI of the MI355X screening engine (libpmx.so, built from pharmaconet_amd/csrc).
 *
 * The reference (SeonghwanSeo/PharmacoNet, /root/reference) has no FFI for this path; its seams are
 * Python-level. Each entry point below names the reference interface it stands in for (paths are
 * relative to /root/reference). The Python shim that binds these symbols is
 * pharmaconet_amd/_ffi.py; the stub a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions: every function returns 0 on success and a non-zero pmx_status otherwise;
 * pmx_last_error() gives the calling thread's last message. No exceptions cross the boundary.
 * Handles are opaque; a handle is bound to the device it was created on. Every entry point may be called from
 * any thread. pmx_score / pmx_score_multi keep their work buffers per (device, stream): calls on different streams run
 * concurrently, calls on the same stream are ordered by it (a second host thread enqueuing on the same stream waits for
 * the first to finish enqueuing, not for the GPU). `stream` is a hipStream_t
 * passed as void* (NULL = the default stream). Pointers named *_dev are device pointers on that device.
 */
#ifndef PMX_H
#define PMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMX_NUM_TYPES 7          /* Hydrophobic, Aromatic, Cation, Anion, HBond_donor, HBond_acceptor, Halogen */
#define PMX_MAX_LEVELS 20        /* src/pmnet/scoring/graph_match.py:88 */
#define PMX_MAX_MODEL_NODES 256   /* node sets are lists on the device; cluster_nodes is a bit mask of ceil(n_nodes / 64) words */
#define PMX_MAX_MODEL_CLUSTERS 128 /* candidate sets of a ligand cluster are two 64-bit words */
#define PMX_MAX_LEVEL_CANDIDATES 64 /* model clusters that share a type with ONE ligand cluster (a tree level's candidates): a ligand
                                       with a cluster beyond that is reported PMX_LIGAND_UNSUPPORTED (only models of more than 64 clusters can do that) */
#define PMX_MAX_LIGAND_NODES 64
#define PMX_MAX_LIGAND_CLUSTERS 64
#define PMX_MAX_CONFORMERS 64

typedef enum {
    PMX_OK = 0,
    PMX_ERR_INVALID = 1, /* bad argument / model or library outside the structural limits above */
    PMX_ERR_HIP = 2,     /* a HIP runtime call failed */
    PMX_ERR_OOM = 3
} pmx_status;

/* Per-ligand status written by pmx_score. */
#define PMX_LIGAND_OK 0
#define PMX_LIGAND_UNSUPPORTED 1 /* record exceeds a structural limit; score is NaN */
#define PMX_LIGAND_TOO_LARGE 2   /* the ligand's score tables exceed the whole table arena (PMX_ARENA_MB); score is NaN */

typedef struct pmx_model pmx_model;
typedef struct pmx_library pmx_library;

/*
 * Flat pharmacophore model: the object graph built by PharmacophoreModel.__setstate__
 * (src/pmnet/pharmacophore_model.py:191-204) made positional. edge_mean / edge_std are the
 * float32 values of `model_node1.neighbor_edge_dict[model_node2].distance_mean / .distance_std`
 * (src/pmnet/scoring/match_utils.py:35-48) for every ordered node pair, self-loops included.
 * Clusters are listed in `model.node_clusters` order (pharmacophore_model.py:202-204);
 * cluster_typemask is the stored `node_types` set, cluster_center / cluster_size feed the
 * cluster-distance prefilter (src/pmnet/scoring/graph_match.py:263-268).
 * All pointers are host pointers; the data is copied.
 */
typedef struct {
    int32_t n_nodes;
    int32_t n_clusters;
    const uint8_t *node_type;        /* [n_nodes] type id 0..6 */
    const float *edge_mean;          /* [n_nodes * n_nodes] */
    const float *edge_std;           /* [n_nodes * n_nodes] */
    const uint64_t *cluster_nodes;   /* [n_clusters * W], W = max(1, ceil(n_nodes / 64)): bit (m % 64) of word a * W + m / 64 = node m belongs
                                        to cluster a (one word per cluster for models of up to 64 nodes) */
    const uint8_t *cluster_typemask; /* [n_clusters] */
    const double *cluster_center;    /* [n_clusters * 3] */
    const double *cluster_size;      /* [n_clusters] */
} pmx_model_desc;

/*
 * Packed ligand library (format: pharmaconet_amd/library.py): what GraphMatcher reads from each
 * ligand's LigandGraph (src/pmnet/scoring/ligand.py:110-259) - typed nodes, per-conformer node
 * positions, clusters in priority_fn order (graph_match.py:43-60). Replaces the per-file
 * `Ligand.load_from_file` objects that screening.py:46-47 hands to scoring_file one at a time.
 */
typedef struct {
    uint64_t n_ligands;
    const uint64_t *offsets; /* [n_ligands + 1] byte offsets into data, each a multiple of 16 */
    const uint8_t *data;
    int32_t on_device;       /* 0: host pointers (copied to the device); 1: device pointers (copied device-to-device); 2: device pointers, ADOPTED - the library
                                reads the caller's buffers in place (no allocation, no copy: what pmx_pack_features_device wrote, as it lies); they must
                                stay valid and unchanged until pmx_library_destroy, which does not free them. A device view must be complete when the call
                                is made (the call reads it on the default stream). */
} pmx_library_view;

typedef struct {
    uint64_t n_ligands;
    uint64_t n_bytes;
    uint64_t total_conformers;
    int32_t max_nodes, max_conformers, max_clusters;
    int32_t n_unsupported;
} pmx_library_info;

const char *pmx_last_error(void);
int pmx_version(void);

/* PharmacophoreModel.load (pharmacophore_model.py:163-176) -> device-resident tables. */
int pmx_model_create(const pmx_model_desc *desc, int device, pmx_model **out);
int pmx_model_destroy(pmx_model *model);

int pmx_library_upload(const pmx_library_view *view, int device, pmx_library **out);
int pmx_library_info_get(const pmx_library *lib, pmx_library_info *info);
int pmx_library_destroy(pmx_library *lib);

/*
 * Batched PharmacophoreModel._scoring (pharmacophore_model.py:101-106), i.e.
 * GraphMatcher(model, ligand, weights).run() (graph_match.py:94-101) for ligands
 * [first, first + count). weights[7] = DEFAULT_WEIGHTS updated by the caller's dict
 * (graph_match.py:32-40,81-83) in type-id order. scores_dev[count] receives the float32 value
 * of the float the reference returns (0 for ligands without clusters or candidates,
 * graph_match.py:95-99); status_dev[count] (may be NULL) receives PMX_LIGAND_*.
 * Everything is enqueued and the call returns: no device-to-host read, no synchronisation, no helper thread (a
 * synchronisation happens only when a cached work buffer has to grow). The work is ordered on `stream`: per chunk of ligands
 * (PMX_SUPER) the ligand kernel (score tables in per-wavefront slices + tree search within a pass budget), the same for
 * ligands with larger tables, a fixed number of task rounds for the subtrees of over-budget trees, finalize. With
 * PMX_OVERLAP set and more than one chunk, the rounds of a chunk run on a side stream that the workspace owns, beside the
 * next chunk's ligand kernel: they start behind an event recorded on `stream` and `stream` waits for their last event before
 * the call's work counts as done, so a caller sees one stream-ordered operation either way. Work buffers are kept per (device, stream) for at most PMX_MAX_WORKSPACES (default 4) streams per device - the least recently used idle workspace is
 * freed when one more stream appears. At the defaults: about 73 GB for libraries of up to 8 conformers and an 11-cluster model (64 GB table arena,
 * 2 GB task queue, <= 4 GB large slices, 0.7 GB slices, 1.8 GB path sums), about 70 GB at 32 / 64 conformer lanes (32 GB arena, 8 GB queue, <= 16 GB
 * large slices, slices that grow with the square of the model's cluster count); PMX_ARENA_MB, PMX_TASKQ_MB size one buffer set, and a second
 * set exists only while PMX_OVERLAP is in use. Without PMX_ARENA_MB the table arena takes at most a third of the device memory that is free when it is first allocated, and it shrinks when device memory is short - a smaller arena is slower, never
 * wrong - and keeps the size it got until pmx_release_workspaces. PMX_LIGAND_TOO_LARGE is reported for a ligand whose tables exceed a whole
 * arena; a ligand that merely found the arena full of other ligands' tables is taken again with the arena empty, in up to PMX_ARENA_RETRIES
 * (default 4) further passes (only models whose largest possible tables exceed a large slice have such passes).
 */
int pmx_score(const pmx_model *model, const pmx_library *lib, const float weights[PMX_NUM_TYPES], uint64_t first,
              uint64_t count, float *scores_dev, int32_t *status_dev, void *stream);

/*
 * pmx_score with the score as a float64: the value `GraphMatcher.run()` returns is `float(np.mean(...))` of float64 per-conformer maxima
 * (graph_match.py:103-109); pmx_score rounds it to float32, this entry point hands it over as it is (scores_dev = double[count]), so that a
 * CSV written from it (screening.py:70-75) carries the reference's digits as far as the tabulated pair functions allow (relative 1e-7) and
 * two ligands closer than a float32 ulp keep the order the float64 values give them. Same work, same kernels.
 */
int pmx_score_f64(const pmx_model *model, const pmx_library *lib, const float weights[PMX_NUM_TYPES], uint64_t first,
                  uint64_t count, double *scores_dev, int32_t *status_dev, void *stream);
int pmx_score_multi_f64(const pmx_model *const *models, int n_models, const pmx_library *lib,
                        const float weights[PMX_NUM_TYPES], uint64_t first, uint64_t count, double *scores_dev,
                        int32_t *status_dev, void *stream);

/* The same for several models over one library: the pockets' chunks are one sequence of one call (one pocket after the other
 * on `stream` by default; PMX_OVERLAP=2 lets a pocket's last task rounds run beside the next pocket's ligand kernel);
 * scores_dev is [n_models][count], status_dev[count] is written once. */
int pmx_score_multi(const pmx_model *const *models, int n_models, const pmx_library *lib,
                    const float weights[PMX_NUM_TYPES], uint64_t first, uint64_t count, float *scores_dev,
                    int32_t *status_dev, void *stream);

/*
 * The ranking step of screening.py:70 (`result.sort(key=score, reverse=True)`, stable): the k best
 * of scores_dev[n] in descending score order, ties in ascending index order. index_dev may be
 * NULL (element i has index base_index + i) or give each element's global index.
 * out_scores_dev[k], out_index_dev[k]; if n < k the tail is filled with -inf / UINT64_MAX. A NaN score (an unsupported ligand)
 * ranks after every real score and is reported as NaN with its index; input elements whose index is UINT64_MAX are padding
 * (they rank last). k <= 65536, n < 2^31. A radix select on the device (csrc/pmx_topk.hip): no sort of all n, no library.
 */
int pmx_topk(const float *scores_dev, const uint64_t *index_dev, uint64_t n, uint64_t base_index, int k,
             float *out_scores_dev, uint64_t *out_index_dev, int device, void *stream);

/*
 * Multi-GPU exchange (one process per GPU). Replaces the gather-and-sort of screening.py:66-70 (`pool.map` over all
 * files, then one Python sort): every rank scores a contiguous shard, takes its k best with pmx_topk (global indices:
 * base_index = first ligand of the shard) and calls pmx_topk_allgather, which all-gathers the (score, index) lists
 * over RCCL (xGMI inside a node) and merges them identically on every rank - descending score, ties by ascending
 * global index. The communicator is built from an id made on rank 0 (pmx_comm_unique_id) and handed to the other
 * ranks by the host program (file, socket, torch.distributed store ...).
 */
#define PMX_COMM_ID_BYTES 128
typedef struct pmx_comm pmx_comm;
int pmx_comm_unique_id(char id_out[PMX_COMM_ID_BYTES]);
int pmx_comm_create(const char id[PMX_COMM_ID_BYTES], int rank, int nranks, int device, pmx_comm **out);
int pmx_comm_info(pmx_comm *comm, int *rank_out, int *nranks_out, int *device_out); /* as RCCL reports them (ncclCommCount / UserRank / CuDevice) */
int pmx_comm_destroy(pmx_comm *comm);
int pmx_topk_allgather(pmx_comm *comm, const float *scores_k_dev, const uint64_t *index_k_dev, int k, float *out_scores_dev,
                       uint64_t *out_index_dev, void *stream);

/*
 * The library packer in native code (pmx_pack.cpp): what LigandGraph.__init__ (src/pmnet/scoring/ligand.py:110-259) and the
 * priority sort of graph_match.py:43-60 make of a molecule's perceived pharmacophore features, as records of the packed
 * library format. Input = what Ligand.__init__ hands to LigandGraph (ligand.py:16-61,120-132), flat over a batch of
 * molecules: per atom its atomic number and heavy-atom neighbours (CSR, in the order OBAtomAtomIter yields them); per
 * feature (`pharmacophore_list` order, ligand_utils.py:80-88) its type id, the atom indices and the centre indices, with
 * flags bit 0 / bit 1 telling whether atom_indices / center_indices was a tuple rather than an int (an int and a
 * 1-tuple are different node keys, ligand.py:137); positions float32 [n_atoms][n_conformers][3] per molecule.
 * offsets_out[n_mols + 1] and data_out receive the library; *data_bytes the bytes written. Sizing: a call with data_out =
 * NULL packs nothing and returns in *data_bytes an upper bound (every feature a node) - allocate that, pack once, keep the
 * first *data_bytes bytes. With a data_cap that turns out too small the call fails and *data_bytes holds the exact need.
 * status_out (may be NULL): 1 for a molecule outside the structural limits above, 2 for malformed input (type id above 6, an atom /
 * centre / neighbour index outside the molecule, a feature without atoms, too few positions) or a feature graph on which the
 * reference's builder raises (a dependent node whose ring has no cluster); either becomes a header-only record that pmx_score
 * reports as PMX_LIGAND_UNSUPPORTED. Offsets that run backwards fail the whole call.
 */
typedef struct {
    uint64_t n_mols;
    const uint64_t *atom_off;        /* [n_mols + 1] first atom of each molecule */
    const uint8_t *atomic_num;       /* [total atoms] */
    const uint64_t *nbr_off;         /* [total atoms + 1] */
    const int32_t *nbr;              /* neighbour atom indices, local to the molecule */
    const uint64_t *feat_off;        /* [n_mols + 1] first feature of each molecule */
    const uint8_t *feat_type;        /* [total features] type id 0..6 */
    const uint8_t *feat_flags;       /* [total features] bit 0: atom_indices is a tuple, bit 1: center_indices is a tuple */
    const uint64_t *feat_atom_off;   /* [total features + 1] */
    const int32_t *feat_atoms;
    const uint64_t *feat_center_off; /* [total features + 1] */
    const int32_t *feat_centers;
    const int32_t *n_conf;           /* [n_mols] */
    const uint64_t *pos_off;         /* [n_mols + 1] float offset of each molecule's positions */
    const float *positions;
} pmx_feature_batch;
int pmx_pack_features(const pmx_feature_batch *batch, int threads, uint64_t *offsets_out, uint8_t *data_out, uint64_t data_cap,
                      uint64_t *data_bytes, int32_t *status_out);
/*
 * The same packer on the device (pmx_pack_device.hip), for batches that are in device memory: every pointer of `batch`, offsets_out,
 * data_out and status_out are DEVICE pointers; the struct itself and data_bytes are on the host. Records are byte-identical to
 * pmx_pack_features'. The kernels run on `stream` (a hipStream_t; NULL = the default stream) and the call waits for the sizes (one
 * 8-byte read) before it enqueues the record writer and returns: *data_bytes is the exact size, offsets_out and data_out are
 * complete in stream order. A call with data_out = NULL only sizes (EXACTLY, unlike the host call's bound; the bound
 * sum(((8 + 2 nf + 3 + 12 nf c + 15) & ~15) + 16) over the molecules' feature and conformer counts needs no call at all).
 * Calls share one set of work buffers per process: they are serialised on the host, and a call made on another stream than the one before
 * it starts, on the device, behind that call's record writer.
 * status_out as above, plus 3: the molecule is outside the fixed scratch of the device builder (more than 256 atoms, 255 features,
 * 1024 neighbour entries, 1024 feature-atom entries, or a feature of more than 16 atoms) and became a header-only record - pack
 * such a batch with pmx_pack_features. Two differences in reporting, none in records: a molecule whose offsets run backwards is
 * reported 2 (the host call fails as a whole), and a molecule of more than 64 nodes is reported 1 without looking further (the host
 * packer reports 2 if the reference's builder would also raise on it).
 */
int pmx_pack_features_device(const pmx_feature_batch *batch, int device, void *stream, uint64_t *offsets_out, uint8_t *data_out,
                             uint64_t data_cap, uint64_t *data_bytes, int32_t *status_out);

/*
 * The rule half of ligand perception in native code (pmx_perceive.cpp): get_pharmacophore_nodes of
 * src/pmnet/scoring/ligand_utils.py:25-184 for a batch of molecules. The reference asks OpenBabel per atom inside Python
 * predicates, one molecule at a time; here what only the chemistry toolkit can say comes in as flat per-atom answers, taken on the
 * hydrogen-free molecule (pybel `removeh()`, ligand.py:37-38) in atom order:
 *   atomic_num        OBAtom::GetAtomicNum            explicit_degree  GetExplicitDegree        heavy_degree  GetHvyDegree
 *   hyb               GetHyb                          h_count          explicit hydrogens still bound to the atom (0 after removeh)
 *   flags             bit 0: IsHbondAcceptor; bit 1: IsHbondDonor of the same atom after AddPolarHydrogens on a copy (ligand_utils.py:30-34,46)
 *   nbr_off / nbr     heavy-atom neighbours (CSR over all atoms of the batch; indices local to the molecule) in OBAtomAtomIter order
 *   ring_off / ring_atom_off / ring_atoms   the AROMATIC rings of the SSSR (pybel `sssr`, `ring.IsAromatic()`), atoms in any order
 * Output: the molecules' feature lists in pharmacophore_list order (ligand_utils.py:80-88) in the layout of pmx_feature_batch
 * (feat_off[n_mols + 1], feat_type, feat_flags, feat_atom_off, feat_atoms, feat_center_off, feat_centers) - together with the atom arrays
 * above and the conformer coordinates that IS the input of pmx_pack_features. A call with feat_type == NULL only counts (*n_features,
 * *n_feat_atoms, *n_feat_centers and feat_off are written). status_out (may be NULL): 2 for a molecule with malformed input (a neighbour
 * or ring atom outside the molecule), which gets no features.
 */
typedef struct {
    uint64_t n_mols;
    const uint64_t *atom_off;      /* [n_mols + 1] first atom of each molecule */
    const uint8_t *atomic_num;     /* [total atoms] */
    const uint8_t *explicit_degree;
    const uint8_t *heavy_degree;
    const uint8_t *hyb;
    const uint8_t *h_count;
    const uint8_t *flags;
    const uint64_t *nbr_off;       /* [total atoms + 1] */
    const int32_t *nbr;
    const uint64_t *ring_off;      /* [n_mols + 1] first aromatic ring of each molecule */
    const uint64_t *ring_atom_off; /* [total rings + 1] */
    const int32_t *ring_atoms;     /* atom indices local to the molecule */
} pmx_atom_batch;
int pmx_perceive_features(const pmx_atom_batch *batch, int threads, uint64_t *feat_off, uint8_t *feat_type, uint8_t *feat_flags,
                          uint64_t *feat_atom_off, int32_t *feat_atoms, uint64_t *feat_center_off, int32_t *feat_centers,
                          uint64_t cap_features, uint64_t cap_atoms, uint64_t cap_centers, uint64_t *n_features, uint64_t *n_feat_atoms,
                          uint64_t *n_feat_centers, int32_t *status_out);

/*
 * Conformer coordinates of an SD file in native code: the coordinate half of Ligand.load_from_file
 * (src/pmnet/scoring/ligand.py:63-84), which has OpenBabel build a molecule object per record and copies
 * `[atom.coords for atom in pbmol.atoms]` in a Python loop after removeh(). The features come from the first record alone
 * (ligand.py:78-84), so the further records only contribute heavy-atom coordinates: element and float32 position of every
 * atom that is not a hydrogen (H, D, T), in file order, record after record (MDL molfile V2000 and V3000, records separated
 * by $$$$). `text` / `len`: the file's bytes. max_records: records to read (num_conformers; 0 = all). With atomic_num ==
 * NULL and xyz == NULL the call only counts (*n_records, *n_atoms). A record that cannot be parsed fails the call with
 * PMX_ERR_INVALID and leaves its index in *n_records.
 */
int pmx_sdf_heavy_atoms(const char *text, uint64_t len, uint64_t max_records, uint64_t cap_records, uint64_t cap_atoms,
                        uint64_t *n_records, uint64_t *n_atoms, int32_t *atoms_per_record, uint8_t *atomic_num, float *xyz);

/*
 * The same for a Tripos mol2 file (`pybel.readfile("mol2", ...)` in ligand.py:72): every @<TRIPOS>MOLECULE record's
 * @<TRIPOS>ATOM section, the element taken from the SYBYL atom type in front of its dot, hydrogens dropped. An atom type that
 * names no element (Du, LP, Any ...) fails the call with PMX_ERR_INVALID like any record it cannot parse: the caller then
 * reads the file the reference's way.
 */
int pmx_mol2_heavy_atoms(const char *text, uint64_t len, uint64_t max_records, uint64_t cap_records, uint64_t cap_atoms,
                         uint64_t *n_records, uint64_t *n_atoms, int32_t *atoms_per_record, uint8_t *atomic_num, float *xyz);

/*
 * Voxel components of hotspot density maps on the device - the search of `DensityMapGraph.__extract_pharmacophores`
 * (src/pmnet/utils/density_map.py:78-110) that `PharmacophoreModel.create` (pharmacophore_model.py:108-149) runs per hotspot in a
 * Python loop over 26 neighbours per voxel. `maps`: n_maps arrays float32 [size][size][size] (C order, mask[x][y][z]); a voxel is
 * in a component where its value is > 0; voxels are named by their linear index (x * size + y) * size + z.
 *   pmx_density_create  uploads the maps and labels the 26-connected components (label = smallest linear index of the component);
 *   pmx_density_labels  labels_out [n_maps * size^3], -1 outside the components;
 *   pmx_density_order   for n_components components (map, seed voxel, offset into members_out; sizes from the labels): the voxels in
 *                       the order the reference's breadth-first search from that seed discovers them (:93-109: the member list is the
 *                       queue, neighbours in itertools.product((-1, 0, 1), repeat=3) order) - the order of its centroid sums. The seed
 *                       of a search is `set.pop()` on a CPython set (:91-92): the caller's business (pharmaconet_amd/model_builder.py).
 */
typedef struct pmx_density pmx_density;
int pmx_density_create(const float *maps, int32_t n_maps, int32_t size, int device, pmx_density **out);
int pmx_density_labels(pmx_density *d, int32_t *labels_out);
int pmx_density_order(pmx_density *d, int32_t n_components, const int32_t *comp_map, const int32_t *comp_seed,
                      const int32_t *comp_offset, int32_t *members_out);
int pmx_density_destroy(pmx_density *d);

/* Frees the scoring workspaces libpmx keeps between calls on `device` (synchronises the device first). */
int pmx_release_workspaces(int device);

/* Diagnostics of the last pmx_score / pmx_score_multi on this thread. pmx_score_stats_get synchronises the stream the
 * call ran on (the counters live on the device); the times are HIP-event times and are filled when profiling was on
 * (pmx_set_profiling(1): three event records per super-chunk, nothing else changes). */
typedef struct {
    double ms_total;            /* the call's kernels, first to last */
    double ms_ligand, ms_tasks; /* of the last super-chunk: ligand kernels (tables + tree search within budget) | task rounds + finalize */
    uint64_t ligands_last;      /* ligands in that super-chunk */
    uint64_t n_frames;          /* tree nodes entered (tree.py:15-53); walkers of a split ligand share maxima while they run, so this
                                   count (not the scores) varies a little from run to run */
    uint64_t n_passes;          /* walker passes: evaluations of a frame's candidates (probes included) */
    uint64_t n_items;           /* table phase: (table entry, ligand node pair) evaluations per wavefront, i.e. / (64 / G) slots */
    uint64_t n_exact_cells;     /* items whose 2-sigma majority test was counted term by term (pass set not an interval), per lane */
    uint64_t n_heavy;           /* walks that ran over their budget (ligands and queued subtrees) */
    uint64_t n_tasks;           /* subtrees taken from the task queue (the empty records that pad a shard's end included) */
    uint64_t n_exported;        /* subtree records written to the task queue */
    uint64_t n_slice_overflow;  /* ligands whose tables did not fit a per-wavefront slice */
    uint64_t n_probes, n_probe_passes; /* reachability searches: subtrees below 5 matches that are handed over, or dropped by the bound test */
    uint64_t max_passes;        /* longest single walk */
    uint64_t queue_overflow;    /* 1 if a task queue shard filled up (results stay exact; raise PMX_TASKQ_MB) */
    uint64_t arena_bytes;       /* table arena in use at the end of the last super-chunk */
    uint64_t ticks_scan, ticks_tables, ticks_bounds, ticks_walk, ticks_alive; /* s_memtime ticks summed over wavefronts, by phase */
    uint64_t n_exact_values;    /* self-table items evaluated term by term because their cell of the tabulated function is not accurate relative
                                   to the function's own (tail) value there, per lane */
    uint64_t dbg[8];            /* walker counters of instrumented builds (-DPMX_COUNTERS, see csrc/pmx_screen.hip walk()); 0 otherwise */
    uint64_t n_path_bounds, n_path_drops; /* children tested against the bound their actual path gives (path_bound()), and dropped by it */
    uint64_t n_dead_entries;    /* pair-table entries settled as -1 without computing their items: more than half of their node pairs lie
                                   outside every 2-sigma window of the two model clusters, for every conformer */
    uint64_t arena_capacity;    /* bytes of the table arena this workspace obtained (PMX_ARENA_MB, or a third of the free device memory at its first
                                   allocation, halved until it fit): which ligands can get PMX_LIGAND_TOO_LARGE depends on it */
} pmx_score_stats;
int pmx_score_stats_get(pmx_score_stats *out);
int pmx_set_profiling(int enabled); /* when enabled pmx_score records HIP events around its phases (no synchronisation) */

#ifdef __cplusplus
}
#endif
#endif /* PMX_H */
