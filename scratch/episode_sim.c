// Experiment (not shipped): exact "episode" speculation for the flood of one component.
//
// Round: the R smallest queued elements x_0 < ... < x_{R-1} are taken; lane r runs the EPISODE of x_r: pop x_r,
// then keep popping from its private queue of newly discovered voxels while the private minimum sorts below
// x_{r+1} (a new element is younger than every queued one, so that is `value < value(x_{r+1})`).  Discoveries are
// claims with priority = rank (atomicMax in the kernel).  Episode s commits iff it lost no claim to a lower rank
// after acting on it, and everything it popped sorts below every residual (unpopped) element of the episodes
// before it.  The committed prefix is exactly the sequential pop sequence.  The simulator runs the lanes in
// lock-step sub-rounds like a wave would and checks the final labels against the sequential result.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { float v; int64_t age; int64_t idx; } el;
static inline int less(const el *a, const el *b)
{
    if (a->v < b->v) return 1;
    if (a->v > b->v) return 0;
    if (a->age != b->age) return a->age < b->age;
    return a->idx < b->idx;
}
static el *h; static int64_t hn, hcap;
static void hpush(el e)
{
    if (hn == hcap) { hcap *= 2; h = realloc(h, sizeof(el) * hcap); }
    int64_t c = hn++;
    while (c > 0) { int64_t p = (c - 1) >> 1; if (!less(&e, &h[p])) break; h[c] = h[p]; c = p; }
    h[c] = e;
}
static el hpop(void)
{
    el top = h[0], last = h[--hn];
    if (hn > 0) {
        int64_t c = 0;
        for (;;) { int64_t l = 2 * c + 1, r = l + 1, m; if (l >= hn) break; m = (r < hn && less(&h[r], &h[l])) ? r : l; if (!less(&h[m], &last)) break; h[c] = h[m]; c = m; }
        h[c] = last;
    }
    return top;
}
#define MAXR 1024
#define MAXP 512
typedef struct { int64_t idx; float v; int popped; } pushrec;
typedef struct { int live, npush, npop, lost, trunc; float maxnew; int64_t cur; pushrec p[MAXP]; } episode;
static episode ep[MAXR];

// stats[0]=rounds [1]=pops [2]=committed episodes [3]=sub-rounds [4]=cut by lost claim [5]=cut by F-rule (not truncated)
// [6]=cut by truncation [7]=wasted pops [8]=pushes committed [9]=residual pushes [10..] hist of episode length
int64_t episode_sim(const float *aff, const int64_t *seeds, int64_t ns, const uint8_t *mask, uint32_t *out,
                    const int64_t *dims, int R, int K, int LCAP, int64_t *stats)
{
    int64_t N = dims[0] * dims[1] * dims[2], SZ = dims[1] * dims[2], SY = dims[2];
    const int64_t noff[6] = {-SZ, -SY, -1, 1, SY, SZ};
    const int ax[6] = {0, 1, 2, 2, 1, 0};
    uint32_t *c_epoch = calloc(N, 4); uint16_t *c_rank = calloc(N, 2); uint8_t *c_sub = calloc(N, 1);
    hcap = ns + 1024; hn = 0; h = malloc(sizeof(el) * hcap);
    for (int64_t i = 0; i < ns; ++i) out[seeds[i]] = (uint32_t)(i + 1);
    for (int64_t i = 0; i < ns; ++i) { el e = {0.f, 0, seeds[i]}; hpush(e); }
    int64_t gage = 0; uint32_t epoch = 0;
    el cand[MAXR]; float bound[MAXR];
    while (hn > 0) {
        ++epoch; ++stats[0];
        int n = 0;
        while (n < R && hn > 0) cand[n++] = hpop();
        for (int r = 0; r < n; ++r) bound[r] = r + 1 < n ? cand[r + 1].v : (hn > 0 ? h[0].v : INFINITY);
        for (int r = 0; r < n; ++r) { ep[r].live = 1; ep[r].npush = 0; ep[r].npop = 0; ep[r].lost = 0; ep[r].trunc = 0; ep[r].maxnew = -INFINITY; ep[r].cur = cand[r].idx; }
        int k;
        for (k = 0; k < K; ++k) {
            int any = 0;
            for (int r = 0; r < n; ++r) any |= ep[r].live;
            if (!any) break;
            // phase 1: claims (atomicMax by priority; order of arrival within a sub-round is irrelevant)
            for (int r = 0; r < n; ++r) {
                if (!ep[r].live) continue;
                for (int d = 0; d < 6; ++d) {
                    int64_t nb = ep[r].cur + noff[d];
                    if (!mask[nb] || out[nb]) continue;
                    if (c_epoch[nb] == epoch) {
                        if (c_rank[nb] <= r) continue;              // held by a lower rank (or by me)
                        if (c_sub[nb] < k) ep[c_rank[nb]].lost = 1;   // the holder already acted on it
                    }
                    c_epoch[nb] = epoch; c_rank[nb] = (uint16_t)r; c_sub[nb] = (uint8_t)k;
                }
            }
            // phase 2: read back, push, choose the next pop
            for (int r = 0; r < n; ++r) {
                episode *e = &ep[r];
                if (!e->live) continue;
                ++e->npop;
                for (int d = 0; d < 6; ++d) {
                    int64_t nb = e->cur + noff[d];
                    if (!mask[nb] || out[nb]) continue;
                    if (!(c_epoch[nb] == epoch && c_rank[nb] == r && c_sub[nb] == k)) continue;
                    int dup = 0;            // the same voxel reached twice from one pop cannot happen; from two pops it is "labelled"
                    for (int j = 0; j < e->npush; ++j) if (e->p[j].idx == nb) dup = 1;
                    if (dup) continue;
                    if (e->npush == MAXP) { e->trunc = 1; break; }
                    float v = aff[(int64_t)ax[d] * N + (d < 3 ? e->cur : nb)] + 0.0f;
                    e->p[e->npush].idx = nb; e->p[e->npush].v = v; e->p[e->npush].popped = 0; ++e->npush;
                }
                int best = -1, pending = 0;
                for (int j = 0; j < e->npush; ++j) if (!e->p[j].popped) { ++pending; if (best < 0 || e->p[j].v < e->p[best].v) best = j; }
                if (pending > LCAP) { e->trunc = 1; e->live = 0; continue; }
                if (best >= 0 && e->p[best].v < bound[r] && !e->trunc) {
                    if (k + 1 == K) { e->trunc = 1; e->live = 0; continue; }
                    e->p[best].popped = 1; e->cur = e->p[best].idx; if (e->p[best].v > e->maxnew) e->maxnew = e->p[best].v;
                } else e->live = 0;
            }
        }
        stats[3] += k;
        // own claims made in an earlier sub-round of this episode must not block it: handled by dup check above (c_rank==r, c_sub<k)
        float F = INFINITY; int nc = n, reason = 0;
        for (int s = 0; s < n; ++s) {
            int ok = !ep[s].lost && cand[s].v <= F && ep[s].maxnew < F;
            if (!ok) { nc = s; reason = ep[s].lost ? 1 : 2; break; }
            for (int j = 0; j < ep[s].npush; ++j) if (!ep[s].p[j].popped && ep[s].p[j].v < F) F = ep[s].p[j].v;
        }
        if (nc < n) {
            if (reason == 1) ++stats[4];
            else { int tr = 0; for (int s = 0; s < nc; ++s) tr |= ep[s].trunc; if (tr) ++stats[6]; else ++stats[5]; }
        }
        for (int s = 0; s < n; ++s) {
            if (s < nc) {
                uint32_t lab = out[cand[s].idx];
                stats[1] += ep[s].npop; ++stats[2];
                int b = 0, L = ep[s].npop; while (L > 1 && b < 15) { L >>= 1; ++b; }
                ++stats[10 + b];
                for (int j = 0; j < ep[s].npush; ++j) {
                    ++gage; out[ep[s].p[j].idx] = lab; ++stats[8];
                    if (!ep[s].p[j].popped) { el e = {ep[s].p[j].v, gage, ep[s].p[j].idx}; hpush(e); ++stats[9]; }
                }
            } else { stats[7] += ep[s].npop; hpush(cand[s]); }
        }
    }
    free(h); free(c_epoch); free(c_rank); free(c_sub);
    return stats[0];
}
