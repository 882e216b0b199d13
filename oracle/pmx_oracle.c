/*
 * pmx_oracle.c - CPU restatement of PharmacoNet's GraphMatcher.run() on the packed ligand format.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; nothing under pharmaconet_amd/ does.
 *
 * Parity status: PINNED. tests/test_oracle_golden.py checks this restatement against outputs of
 * the reference itself (tests/golden/, minted by tests/golden/make_golden.py, which imports
 * /root/reference/src/pmnet with the NumPy kernels of scoring/match_utils.py): final scores,
 * number of tree levels, tree nodes and leaves, and checksums of the pair-score tables.
 * The reference holds no tests or golden vectors of its own for this path.
 *
 * Every step cites the reference file:line it follows (paths relative to /root/reference/src/pmnet).
 * Arithmetic follows the NumPy variant (scoring/match_utils.py): float32 tables, float64 tree totals. That is the variant the
 * golden vectors were minted with (Numba is absent in the build container). `variant = 1` restates the Numba kernels of
 * scoring/match_utils_numba.py:54-86,126-151 instead (what a user with Numba installed runs, pyproject.toml:28): float64
 * accumulation of the likelihood, `sigma_sq < 4.0` for the 2-sigma test, weight sums W1 * W2. It is NOT pinned against
 * reference output (Numba cannot be run here; `fastmath=True` may further reassociate): it bounds the spread between
 * the reference's own two code paths, see tests/test_oracle_golden.py::test_numba_variant_spread.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Tables of one ligand come out of a per-thread bump arena that is reset per ligand (the comparator used to call calloc per
 * table); a request that does not fit a chunk falls back to calloc and is freed with the ligand. */
#define ARENA_BYTES ((size_t)8 << 20)
#define ARENA_SPILL 4096
static _Thread_local unsigned char *tl_arena;
static _Thread_local size_t tl_top;
static _Thread_local void *tl_spill[ARENA_SPILL];
static _Thread_local int tl_nspill;
static void *arena_calloc(size_t n, size_t size) {
    const size_t bytes = (n * size + 63) & ~(size_t)63;
    if (!tl_arena) tl_arena = (unsigned char *)malloc(ARENA_BYTES);
    if (tl_arena && tl_top + bytes <= ARENA_BYTES) {
        void *p = tl_arena + tl_top;
        tl_top += bytes;
        memset(p, 0, bytes);
        return p;
    }
    void *p = calloc(n ? n : 1, size ? size : 1);
    if (tl_nspill < ARENA_SPILL) tl_spill[tl_nspill++] = p; /* (beyond that: leaked, never seen) */
    return p;
}
static void arena_reset(void) {
    for (int i = 0; i < tl_nspill; ++i) free(tl_spill[i]);
    tl_nspill = 0;
    tl_top = 0;
}

#define MAX_LEVELS 20 /* scoring/graph_match.py:88 */
#define MAX_K 128
#define MAX_N 64
#define MAX_C 64

typedef struct {
    int32_t n_nodes, n_clusters;
    const uint8_t *node_type;        /* [Nm] */
    const float *edge_mean;          /* [Nm*Nm] */
    const float *edge_std;           /* [Nm*Nm] */
    const uint64_t *cluster_nodes;   /* [K * W], W = max(1, ceil(n_nodes / 64)): bit m % 64 of word a * W + m / 64 <=> node m in cluster a */
    const uint8_t *cluster_typemask; /* [K] */
    const double *cluster_center;    /* [K*3] */
    const double *cluster_size;      /* [K] */
} oracle_model;

typedef struct {
    double score;
    int32_t n_levels;
    int64_t n_tree, n_leaf; /* tree nodes below the root; leaves */
    double s_sum, p_sum;    /* sum of self-table entries; sum of valid pair-table entries */
    int64_t p_invalid, p_entries;
    int64_t n_terms; /* Gaussian terms evaluated per conformer (model node pairs over all ligand node pairs): work measure */
} oracle_result;

typedef struct {
    int n, C, ncl;
    const uint8_t *typemask;
    const uint8_t *cluster_end;
    const float *xyz; /* [n][3][C] */
} ligand_t;

/* one node-match entry: ligand node u with its compatible model nodes (graph_match.py:145-155) */
typedef struct {
    int u;
    int nm;
    uint8_t m[64];
    float w[64];
} node_match;

typedef struct {
    int len; /* L(i,a): entries with a non-empty model list (graph_match.py:164-171) */
    node_match *items;
} match_list;

typedef struct {
    const oracle_model *M;
    ligand_t L;
    int nl;
    int lev_cluster[MAX_LEVELS];
    int k[MAX_LEVELS];
    int cand[MAX_LEVELS][MAX_K];
    match_list nm[MAX_LEVELS][MAX_K];
    float *S[MAX_LEVELS];             /* [k_i][C] */
    float *P[MAX_LEVELS][MAX_LEVELS]; /* i<j: [k_i][k_j][C] */
    double best[MAX_C];
    int64_t n_tree, n_leaf, n_terms;
    /* DFS path */
    int sel[MAX_LEVELS];
    int variant; /* 0: NumPy kernels, 1: Numba kernels */
} ctx_t;

static float edge_distance(const ligand_t *L, int u, int v, int c) {
    /* LigandEdge.set_distances, scoring/ligand.py:349-351: np.linalg.norm(p1 - p2, axis=-1) in float32 */
    const float *pu = L->xyz + (size_t)u * 3 * L->C, *pv = L->xyz + (size_t)v * 3 * L->C;
    float dx = pu[0 * L->C + c] - pv[0 * L->C + c];
    float dy = pu[1 * L->C + c] - pv[1 * L->C + c];
    float dz = pu[2 * L->C + c] - pv[2 * L->C + c];
    float s = dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    return sqrtf(s);
}

static void cluster_center_size(const ligand_t *L, int start, int end, int c, float center[3], float *size) {
    /* LigandNodeCluster.center / .size, scoring/ligand.py:458-473 (float32 mean, max of norms) */
    float sum[3] = {0.f, 0.f, 0.f};
    for (int u = start; u < end; ++u)
        for (int d = 0; d < 3; ++d) sum[d] = sum[d] + L->xyz[((size_t)u * 3 + d) * L->C + c];
    float cnt = (float)(end - start);
    for (int d = 0; d < 3; ++d) center[d] = sum[d] / cnt;
    float mx = 0.f;
    for (int u = start; u < end; ++u) {
        float s = 0.f;
        for (int d = 0; d < 3; ++d) {
            float t = L->xyz[((size_t)u * 3 + d) * L->C + c] - center[d];
            s = s + t * t;
        }
        float r = sqrtf(s);
        if (u == start || r > mx) mx = r;
    }
    *size = mx;
}

/* scoring/match_utils.py:26-69 (pair) and :87-120 (self): one (ligand node, ligand node) term.
 * Adds the likelihood to score[c]; if fails != NULL also counts a fail per conformer. */
static void node_pair_term(ctx_t *X, const node_match *a, const node_match *b, float *score, int16_t *fails) {
    const oracle_model *M = X->M;
    const int C = X->L.C, Nm = M->n_nodes;
    int num_match = a->nm * b->nm;
    X->n_terms += num_match;
    if (X->variant == 1) { /* match_utils_numba.py:54-86 (pair) / :126-151 (self) */
        double W1 = 0.0, W2 = 0.0; /* sum(weights): the int 0 start value makes Numba accumulate in float64 */
        for (int i = 0; i < a->nm; ++i) W1 += (double)a->w[i];
        for (int j = 0; j < b->nm; ++j) W2 += (double)b->w[j];
        const double normalize_coeff = 1.0 / (W1 * W2), score_coeff = (W1 * W2) / (double)num_match; /* :64-65 */
        const int pass_threshold = (num_match + 1) / 2;                                               /* :59 */
        for (int c = 0; c < C; ++c) {
            const float d = edge_distance(&X->L, a->u, b->u, c);
            int num_pass = 0;
            double likelihood = 0.0;
            for (int i = 0; i < a->nm; ++i) {
                double row = 0.0; /* _likelihood */
                for (int j = 0; j < b->nm; ++j) {
                    const float mu = M->edge_mean[a->m[i] * Nm + b->m[j]], sd = M->edge_std[a->m[i] * Nm + b->m[j]];
                    const float z = (d - mu) / sd;
                    const float sigma_sq = z * z;                                  /* :75 float32 operands */
                    row += (double)(b->w[j] / sd) * exp(-0.5 * (double)sigma_sq);  /* :76 */
                    if ((double)sigma_sq < 4.0) ++num_pass;                        /* :77-78 */
                }
                likelihood += (double)a->w[i] * row; /* :79 */
            }
            score[c] = (float)((double)score[c] + likelihood * normalize_coeff * score_coeff); /* :81 */
            if (fails && num_pass < pass_threshold) fails[c] += 1;                             /* :82-83 */
        }
        return;
    }
    /* weights = outer(w1, w2).reshape(-1); weights_sum = sum(weights)  (builtin sum, float32 steps) */
    float weights_sum = 0.f;
    for (int i = 0; i < a->nm; ++i)
        for (int j = 0; j < b->nm; ++j) weights_sum = weights_sum + a->w[i] * b->w[j];
    float normalize_coeff = 1.0f / weights_sum;
    float score_coeff = weights_sum / (float)num_match;
    for (int c = 0; c < C; ++c) {
        float d = edge_distance(&X->L, a->u, b->u, c);
        int num_pass = 0;
        float likelihood = 0.f;
        for (int i = 0; i < a->nm; ++i)
            for (int j = 0; j < b->nm; ++j) {
                float mean = M->edge_mean[a->m[i] * Nm + b->m[j]];
                float std = M->edge_std[a->m[i] * Nm + b->m[j]];
                float z = (d - mean) / std;                       /* :55 */
                if (fabsf(z) < 2.0f) ++num_pass;                   /* :56-60 */
                float wos = (a->w[i] * b->w[j]) / std;             /* :65 weights / stds */
                likelihood = likelihood + wos * expf(-0.5f * (z * z)); /* :64-68 np.dot */
            }
        if (fails && (float)num_pass < (float)num_match * 0.5f) fails[c] += 1; /* :61 */
        score[c] = score[c] + likelihood * normalize_coeff * score_coeff;       /* :69 */
    }
}

static void build_node_matches(ctx_t *X, const float w[7]) {
    const oracle_model *M = X->M;
    const ligand_t *L = &X->L;
    const int nw = M->n_nodes > 64 ? (M->n_nodes + 63) / 64 : 1;
    for (int i = 0; i < X->nl; ++i) {
        int ci = X->lev_cluster[i];
        int start = ci ? L->cluster_end[ci - 1] : 0, end = L->cluster_end[ci];
        for (int s = 0; s < X->k[i]; ++s) {
            int a = X->cand[i][s];
            match_list *ml = &X->nm[i][s];
            ml->items = (node_match *)arena_calloc((size_t)(end - start), sizeof(node_match));
            ml->len = 0;
            for (int u = start; u < end; ++u) { /* cluster iteration order, graph_match.py:159 */
                node_match *it = &ml->items[ml->len];
                it->u = u;
                it->nm = 0;
                for (int m = 0; m < M->n_nodes; ++m) /* graph_match.py:148-150 */
                    if (((M->cluster_nodes[(size_t)a * nw + (m >> 6)] >> (m & 63)) & 1) && ((L->typemask[u] >> M->node_type[m]) & 1)) {
                        it->m[it->nm] = (uint8_t)m;
                        it->w[it->nm] = w[M->node_type[m]]; /* :151-154 */
                        it->nm++;
                    }
                if (it->nm > 0) ml->len++; /* :164-171 */
            }
        }
    }
}

static void build_tables(ctx_t *X, oracle_result *R) {
    const oracle_model *M = X->M;
    const ligand_t *L = &X->L;
    const int C = L->C;
    int16_t fails[MAX_C];
    for (int i = 0; i < X->nl; ++i) {
        /* self table, match_utils.py:77-122 */
        X->S[i] = (float *)arena_calloc((size_t)X->k[i] * C, sizeof(float));
        for (int s = 0; s < X->k[i]; ++s) {
            const match_list *ml = &X->nm[i][s];
            float *sc = X->S[i] + (size_t)s * C;
            for (int p = 0; p < ml->len; ++p)
                for (int q = p + 1; q < ml->len; ++q) node_pair_term(X, &ml->items[p], &ml->items[q], sc, NULL);
            for (int c = 0; c < C; ++c) R->s_sum += (double)sc[c];
        }
    }
    for (int i = 0; i < X->nl; ++i) {
        int ci = X->lev_cluster[i];
        int si = ci ? L->cluster_end[ci - 1] : 0, ei = L->cluster_end[ci];
        for (int j = i + 1; j < X->nl; ++j) {
            int cj = X->lev_cluster[j];
            int sj = cj ? L->cluster_end[cj - 1] : 0, ej = L->cluster_end[cj];
            float *tab = (float *)arena_calloc((size_t)X->k[i] * X->k[j] * C, sizeof(float));
            X->P[i][j] = tab;
            /* graph_match.py:240-241 */
            float ldist[MAX_C], lsize[MAX_C];
            for (int c = 0; c < C; ++c) {
                float c1[3], c2[3], s1, s2;
                cluster_center_size(L, si, ei, c, c1, &s1);
                cluster_center_size(L, sj, ej, c, c2, &s2);
                float dx = c1[0] - c2[0], dy = c1[1] - c2[1], dz = c1[2] - c2[2];
                float s = dx * dx;
                s = s + dy * dy;
                s = s + dz * dz;
                ldist[c] = sqrtf(s);
                lsize[c] = s1 + s2;
            }
            for (int sa = 0; sa < X->k[i]; ++sa)
                for (int sb = 0; sb < X->k[j]; ++sb) {
                    float *out = tab + ((size_t)sa * X->k[j] + sb) * C;
                    int a = X->cand[i][sa], b = X->cand[j][sb];
                    /* graph_match.py:263-268: cluster-distance prefilter */
                    const double *ca = M->cluster_center + 3 * a, *cb = M->cluster_center + 3 * b;
                    double mdist = sqrt((ca[0] - cb[0]) * (ca[0] - cb[0]) + (ca[1] - cb[1]) * (ca[1] - cb[1]) +
                                        (ca[2] - cb[2]) * (ca[2] - cb[2]));
                    float mdist32 = (float)mdist;
                    float msize32 = (float)(M->cluster_size[a] + M->cluster_size[b]);
                    float mn = 0.f;
                    for (int c = 0; c < C; ++c) {
                        float v = fabsf(ldist[c] - mdist32) - lsize[c];
                        if (c == 0 || v < mn) mn = v;
                    }
                    R->p_entries += C;
                    if (mn > msize32) {
                        for (int c = 0; c < C; ++c) out[c] = -1.f;
                        R->p_invalid += C;
                        continue;
                    }
                    /* match_utils.py:9-74 */
                    const match_list *l1 = &X->nm[i][sa], *l2 = &X->nm[j][sb];
                    float match_threshold = (float)(l1->len * l2->len) * 0.5f; /* :22 */
                    memset(fails, 0, sizeof(fails));
                    for (int p = 0; p < l1->len; ++p)
                        for (int q = 0; q < l2->len; ++q) node_pair_term(X, &l1->items[p], &l2->items[q], out, fails);
                    for (int c = 0; c < C; ++c) { /* :71-74 */
                        if ((float)fails[c] <= match_threshold) {
                            R->p_sum += (double)out[c];
                        } else {
                            out[c] = -1.f;
                            R->p_invalid += 1;
                        }
                    }
                }
        }
    }
}

/* ClusterMatchTree.dfs_run, scoring/tree.py:55-104, with the per-candidate filtering of :69-85
 * evaluated when the candidate is reached instead of being carried down in `match_dict` (same sets,
 * same float64 sums in the same order: accumulated pair score top-down, then parent + self + pair,
 * tree.py:38-41,78-82). `level` is the level of this node (-1 for the root). */
static int dfs(ctx_t *X, int level, int matched, int num_matches, const uint8_t *alive, const double *total) {
    const int C = X->L.C;
    if (level == X->nl - 1) { /* leaf: tree.py:103-104, graph_match.py:103-109 */
        X->n_leaf++;
        for (int c = 0; c < C; ++c)
            if (alive[c] && total[c] > X->best[c]) X->best[c] = total[c];
        return matched;
    }
    int f = level + 1;
    int max_num = 0, any_child = 0;
    uint8_t calive[MAX_C];
    double ctotal[MAX_C];
    for (int b = 0; b < X->k[f]; ++b) {
        int any = 0;
        for (int c = 0; c < C; ++c) {
            double pair = 0.0;
            int ok = alive[c];
            for (int j = 0; j < f && ok; ++j) {
                if (X->sel[j] < 0) continue;
                float p = X->P[j][f][((size_t)X->sel[j] * X->k[f] + b) * C + c];
                if (!(p > 0)) ok = 0; /* tree.py:81 */
                pair += (double)p;
            }
            calive[c] = (uint8_t)ok;
            if (ok) {
                ctotal[c] = total[c] + (double)X->S[f][(size_t)b * C + c] + pair; /* tree.py:38-41 */
                any = 1;
            }
        }
        if (!any) continue; /* tree.py:83-84 */
        any_child = 1;
        X->sel[f] = b;
        X->n_tree++;
        int r = dfs(X, f, 1, num_matches + 1, calive, ctotal);
        if (r > max_num) max_num = r;
    }
    if (!any_child || num_matches + max_num < 5) { /* tree.py:98-101 */
        X->sel[f] = -1;
        X->n_tree++;
        int r = dfs(X, f, 0, num_matches, alive, total);
        if (r > max_num) max_num = r;
    }
    return max_num + matched; /* tree.py:102 */
}

static void score_ligand(const oracle_model *M, const uint8_t *rec, const float w[7], oracle_result *R, int variant) {
    arena_reset();
    ctx_t *X = (ctx_t *)arena_calloc(1, sizeof(ctx_t));
    memset(R, 0, sizeof(*R));
    X->M = M;
    X->variant = variant;
    ligand_t *L = &X->L;
    L->n = rec[0] | (rec[1] << 8);
    L->C = rec[2] | (rec[3] << 8);
    L->ncl = rec[4] | (rec[5] << 8);
    L->typemask = rec + 8;
    L->cluster_end = rec + 8 + L->n;
    size_t off = 8 + (size_t)L->n + (size_t)L->ncl;
    off = (off + 3) & ~(size_t)3;
    L->xyz = (const float *)(rec + off);
    /* graph_match.py:95-96: no clusters -> 0 */
    /* cluster candidates, graph_match.py:124-137; order and cap, :87-88 (records are pre-sorted by priority_fn) */
    for (int ci = 0; ci < L->ncl && X->nl < MAX_LEVELS; ++ci) {
        int start = ci ? L->cluster_end[ci - 1] : 0, end = L->cluster_end[ci];
        unsigned lmask = 0;
        for (int u = start; u < end; ++u) lmask |= L->typemask[u];
        int k = 0;
        for (int a = 0; a < M->n_clusters; ++a)
            if (M->cluster_typemask[a] & lmask) X->cand[X->nl][k++] = a;
        if (k == 0) continue;
        X->lev_cluster[X->nl] = ci;
        X->k[X->nl] = k;
        X->nl++;
    }
    R->n_levels = X->nl;
    if (X->nl > 0) { /* graph_match.py:98-99 */
        build_node_matches(X, w);
        build_tables(X, R);
        uint8_t alive[MAX_C];
        double total[MAX_C];
        for (int c = 0; c < L->C; ++c) {
            alive[c] = 1;
            total[c] = 0.0;
            X->best[c] = 0.0;
        }
        dfs(X, -1, 0, 0, alive, total); /* tree.py:219-227 */
        double sum = 0.0;
        for (int c = 0; c < L->C; ++c) sum += X->best[c];
        R->score = sum / (double)L->C; /* graph_match.py:109 */
        R->n_tree = X->n_tree;
        R->n_leaf = X->n_leaf;
        R->n_terms = X->n_terms;
    }
    arena_reset();
}

/* Scores ligands [first, first+count) of a packed library. `results` may be NULL. Returns 0. */
int oracle_score_variant(const oracle_model *M, const uint64_t *offsets, const uint8_t *data, uint64_t first, uint64_t count,
                         const float weights[7], double *scores, oracle_result *results, int num_threads, int variant);
int oracle_score(const oracle_model *M, const uint64_t *offsets, const uint8_t *data, uint64_t first, uint64_t count,
                 const float weights[7], double *scores, oracle_result *results, int num_threads) {
    return oracle_score_variant(M, offsets, data, first, count, weights, scores, results, num_threads, 0);
}

int oracle_score_variant(const oracle_model *M, const uint64_t *offsets, const uint8_t *data, uint64_t first, uint64_t count,
                         const float weights[7], double *scores, oracle_result *results, int num_threads, int variant) {
    if (num_threads < 1) num_threads = 1;
    if (M->n_clusters > MAX_K || M->n_nodes > 256) return 2; /* fixed-size tables of this restatement */
    {
        const int nw = M->n_nodes > 64 ? (M->n_nodes + 63) / 64 : 1;
        for (int a = 0; a < M->n_clusters; ++a) { /* node_match holds 64 model nodes */
            int cnt = 0;
            for (int w = 0; w < nw; ++w) cnt += __builtin_popcountll(M->cluster_nodes[(size_t)a * nw + w]);
            if (cnt > 64) return 3;
        }
    }
    int64_t n = (int64_t)count;
#pragma omp parallel for schedule(dynamic, 16) num_threads(num_threads)
    for (int64_t i = 0; i < n; ++i) {
        oracle_result r;
        score_ligand(M, data + offsets[first + (uint64_t)i], weights, &r, variant);
        scores[i] = r.score;
        if (results) results[i] = r;
    }
    return 0;
}
