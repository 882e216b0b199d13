/* bound_study.c - how many tree frames a branch-and-bound walker enters under different bounds. ANALYSIS TOOL (test side):
 * it includes the CPU oracle's source for the score tables and runs its own searches on them; nothing in the product uses it.
 *   A: the engine's bound - per-candidate W[(f,b)] (csrc/pmx_screen.hip build_bounds), children visited best first
 *   modes 4-6: what the engine does at 32 / 64 conformer lanes (level bound R under >= 5 matches, candidates in index order), and
 *      the per-candidate bound in that order with and without the test on children with fewer than 5 matches
 *   B: a path-aware bound - for every deeper level the best candidate given the ACTUAL matches on the path (their pair
 *      entries instead of the levels' maxima, candidates incompatible with any of them left out)
 * Both walk the reference's tree semantics (tree.py:55-104) with the engine's dropping rule; scores must agree with the oracle. */
#include "../../oracle/pmx_oracle.c"
#include <stdio.h>

typedef struct {
    ctx_t *X;
    int C, nl;
    int mode; /* 0 = A, 1 = B, 2 = B for children with >= 5 matches (A below), 3 = B where >= 2 levels lie below the child's, 4 = R under >= 5 matches in index order, 5 = W everywhere in index order, 6 = W under >= 5 matches in index order */
    double best[MAX_C];
    double *base[MAX_LEVELS];  /* [k_l][C]  S + sum_{j<l} maxP_j */
    double *maxP[MAX_LEVELS][MAX_LEVELS]; /* [j][l]: [k_l][C] max(0, max_a P[(j,a),(l,b')]) */
    double *W[MAX_LEVELS];     /* [k_f][C] */
    double *R; /* [nl+1][C] */
    int64_t frames, bound_evals, rows;
    int sel[MAX_LEVELS];
} study_t;

static float Pval(ctx_t *X, int j, int a, int l, int b, int c) { return X->P[j][l][((size_t)a * X->k[l] + b) * X->L.C + c]; }

static void build_static(study_t *S) {
    ctx_t *X = S->X;
    const int C = S->C, nl = S->nl;
    for (int l = 0; l < nl; ++l) {
        S->base[l] = (double *)calloc((size_t)X->k[l] * C, sizeof(double));
        for (int j = 0; j < l; ++j) {
            S->maxP[j][l] = (double *)calloc((size_t)X->k[l] * C, sizeof(double));
            for (int b = 0; b < X->k[l]; ++b)
                for (int c = 0; c < C; ++c) {
                    double m = 0.0;
                    for (int a = 0; a < X->k[j]; ++a) {
                        float p = Pval(X, j, a, l, b, c);
                        if (p > m) m = p;
                    }
                    S->maxP[j][l][(size_t)b * C + c] = m;
                }
        }
        for (int b = 0; b < X->k[l]; ++b)
            for (int c = 0; c < C; ++c) {
                double v = X->S[l][(size_t)b * C + c];
                for (int j = 0; j < l; ++j) v += S->maxP[j][l][(size_t)b * C + c];
                S->base[l][(size_t)b * C + c] = v;
            }
    }
    S->R = (double *)calloc((size_t)(nl + 1) * C, sizeof(double));
    for (int l = nl - 1; l >= 0; --l)
        for (int c = 0; c < C; ++c) {
            double u = 0.0;
            for (int b = 0; b < X->k[l]; ++b) if (S->base[l][(size_t)b * C + c] > u) u = S->base[l][(size_t)b * C + c];
            S->R[(size_t)l * C + c] = S->R[(size_t)(l + 1) * C + c] + u;
        }
    for (int f = 0; f < nl; ++f) {
        S->W[f] = (double *)calloc((size_t)X->k[f] * C, sizeof(double));
        for (int b = 0; b < X->k[f]; ++b)
            for (int c = 0; c < C; ++c) {
                double acc = 0.0;
                for (int l = f + 1; l < nl; ++l) {
                    double u = 0.0;
                    for (int b1 = 0; b1 < X->k[l]; ++b1) {
                        float pb = Pval(X, f, b, l, b1, c);
                        if (!(pb > 0)) continue;
                        double val = S->base[l][(size_t)b1 * C + c] - S->maxP[f][l][(size_t)b1 * C + c] + pb;
                        if (val > u) u = val;
                    }
                    acc += u;
                }
                S->W[f][(size_t)b * C + c] = acc;
            }
    }
}

/* path-aware bound for the subtree below frame f (levels f .. nl-1 still open), conformer c, path = sel[0..f-1] */
static double path_bound(study_t *S, int f, int c) {
    ctx_t *X = S->X;
    const int C = S->C;
    double acc = 0.0;
    for (int l = f; l < S->nl; ++l) {
        double u = 0.0;
        for (int b1 = 0; b1 < X->k[l]; ++b1) {
            double v = X->S[l][(size_t)b1 * C + c];
            int ok = 1;
            for (int j = 0; j < f && ok; ++j) {
                if (S->sel[j] < 0) continue;
                float p = Pval(X, j, S->sel[j], l, b1, c);
                if (!(p > 0)) ok = 0;
                v += p;
            }
            if (!ok) continue;
            for (int j = f; j < l; ++j) v += S->maxP[j][l][(size_t)b1 * C + c];
            if (v > u) u = v;
        }
        acc += u;
    }
    return acc;
}

/* does a node with >= 5 matches exist below (validity only)? */
static int reach5(study_t *S, int level, int nm, const uint8_t *alive) {
    ctx_t *X = S->X;
    const int C = S->C;
    if (nm >= 5) return 1;
    if (level == S->nl - 1) return 0;
    int f = level + 1;
    if (nm + (S->nl - f) < 5) return 0;
    uint8_t ca[MAX_C];
    int any_child = 0, mx = 0;
    for (int b = 0; b < X->k[f]; ++b) {
        int any = 0;
        for (int c = 0; c < C; ++c) {
            int ok = alive[c];
            for (int j = 0; j < f && ok; ++j)
                if (S->sel[j] >= 0 && !(Pval(X, j, S->sel[j], f, b, c) > 0)) ok = 0;
            ca[c] = (uint8_t)ok;
            any |= ok;
        }
        if (!any) continue;
        any_child = 1;
        S->sel[f] = b;
        if (reach5(S, f, nm + 1, ca)) return 1;
        mx = 1;
    }
    (void)mx;
    if (!any_child || 1) { /* the reference's skip rule explores skip only if nm + mx < 5; a valid >= 5 assignment is found either way */
        S->sel[f] = -1;
        if (reach5(S, f, nm, alive)) return 1;
    }
    return 0;
}

static int walk(study_t *S, int level, int matched, int nm, const uint8_t *alive, const double *total) {
    ctx_t *X = S->X;
    const int C = S->C;
    S->frames++;
    if (level == S->nl - 1) {
        for (int c = 0; c < C; ++c)
            if (alive[c] && total[c] > S->best[c]) S->best[c] = total[c];
        return matched;
    }
    int f = level + 1, kf = X->k[f];
    int max_num = 0, any_child = 0;
    uint8_t ca[MAX_K][MAX_C];
    double ct[MAX_K][MAX_C];
    int exists[MAX_K], done[MAX_K];
    for (int b = 0; b < kf; ++b) {
        int any = 0;
        for (int c = 0; c < C; ++c) {
            double pair = 0.0;
            int ok = alive[c];
            for (int j = 0; j < f && ok; ++j) {
                if (S->sel[j] < 0) continue;
                float p = Pval(X, j, S->sel[j], f, b, c);
                if (!(p > 0)) ok = 0;
                pair += (double)p;
            }
            ca[b][c] = (uint8_t)ok;
            ct[b][c] = ok ? total[c] + (double)X->S[f][(size_t)b * C + c] + pair : 0.0;
            any |= ok;
        }
        exists[b] = any;
        done[b] = 0;
        if (any) any_child = 1;
    }
    S->rows += (int64_t)nm * ((kf + 7) / 8);
    for (;;) {
        /* best first: the child with the largest total + bound among those not visited */
        int pick = -1;
        double pk = -1.0;
        double bnd[MAX_C];
        for (int b = 0; b < kf; ++b) {
            if (!exists[b] || done[b]) continue;
            double key = 0.0;
            for (int c = 0; c < C; ++c)
                if (ca[b][c]) {
                    double v = ct[b][c] + S->W[f][(size_t)b * C + c];
                    if (v > key) key = v;
                }
            if (S->mode >= 4) { pick = b; break; } /* index order */
            if (key > pk) pk = key, pick = b;
        }
        if (pick < 0) break;
        done[pick] = 1;
        int b = pick, improve = 0;
        S->sel[f] = b;
        for (int c = 0; c < C; ++c) {
            if (!ca[b][c]) continue;
            bnd[c] = S->mode == 4 ? S->R[(size_t)(f + 1) * C + c] : S->W[f][(size_t)b * C + c];
            if ((ct[b][c] + bnd[c]) * (1.0 + 1e-9) > S->best[c]) improve = 1;
        }
        if ((S->mode == 4 || S->mode == 6) && nm < 4) improve = 1; /* no test on children with fewer than 5 matches */
        const int use_path = S->mode == 1 || (S->mode == 2 && nm + 1 >= 5) || (S->mode == 3 && S->nl - f >= 3);
        if (improve && use_path) {
            improve = 0;
            S->bound_evals++;
            for (int c = 0; c < C; ++c) {
                if (!ca[b][c]) continue;
                double pb = path_bound(S, f + 1, c);
                if ((ct[b][c] + pb) * (1.0 + 1e-9) > S->best[c]) improve = 1;
            }
        }
        if (!improve) { /* dropped: returns >= 1, and what it can reach if the frame still has to know */
            int r = 1;
            if (nm + 1 < 5 && nm + max_num < 5) r = reach5(S, f, nm + 1, ca[b]) ? 5 - nm : 1;
            if (r > max_num) max_num = r;
            continue;
        }
        int r = walk(S, f, 1, nm + 1, ca[b], ct[b]);
        if (r > max_num) max_num = r;
    }
    if (!any_child || nm + max_num < 5) {
        S->sel[f] = -1;
        int r = walk(S, f, 0, nm, alive, total);
        if (r > max_num) max_num = r;
    }
    return max_num + matched;
}

/* out[i][16] = {oracle tree nodes, (frames, path-bound evaluations) of modes 0 .. 6, score mismatch flags} */
int bound_study(const oracle_model *M, const uint64_t *offsets, const uint8_t *data, uint64_t first, uint64_t count, const float weights[7],
                int64_t *out) {
    for (uint64_t i = 0; i < count; ++i) {
        const uint8_t *rec = data + offsets[first + i];
        oracle_result R;
        score_ligand(M, rec, weights, &R, 0);
        /* rebuild the tables (score_ligand resets its arena at the end) */
        arena_reset();
        ctx_t *X = (ctx_t *)arena_calloc(1, sizeof(ctx_t));
        X->M = M;
        ligand_t *L = &X->L;
        L->n = rec[0] | (rec[1] << 8);
        L->C = rec[2] | (rec[3] << 8);
        L->ncl = rec[4] | (rec[5] << 8);
        L->typemask = rec + 8;
        L->cluster_end = rec + 8 + L->n;
        size_t off = (8 + (size_t)L->n + (size_t)L->ncl + 3) & ~(size_t)3;
        L->xyz = (const float *)(rec + off);
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
        int64_t *o = out + 16 * i;
        memset(o, 0, 128);
        o[0] = R.n_tree;
        if (X->nl == 0) continue;
        oracle_result R2;
        memset(&R2, 0, sizeof(R2));
        build_node_matches(X, weights);
        build_tables(X, &R2);
        for (int mode = 0; mode < 7; ++mode) {
            study_t S;
            memset(&S, 0, sizeof(S));
            S.X = X, S.C = L->C, S.nl = X->nl, S.mode = mode;
            build_static(&S);
            uint8_t alive[MAX_C];
            double total[MAX_C];
            for (int c = 0; c < L->C; ++c) alive[c] = 1, total[c] = 0.0, S.best[c] = 0.0;
            walk(&S, -1, 0, 0, alive, total);
            double sum = 0.0;
            for (int c = 0; c < L->C; ++c) sum += S.best[c];
            if (sum / L->C != R.score) o[15] |= 1 << mode;
            o[1 + 2 * mode] = S.frames;
            o[2 + 2 * mode] = S.bound_evals;
            for (int l = 0; l < S.nl; ++l) {
                free(S.base[l]), free(S.W[l]);
                for (int j = 0; j < l; ++j) free(S.maxP[j][l]);
            }
            free(S.R);
        }
        arena_reset();
    }
    return 0;
}
