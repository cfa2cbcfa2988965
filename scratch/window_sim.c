// Experiment (not shipped): ordered speculative window for the flood of one component (see episode_sim.c for the
// episode idea).  W episodes are in flight, sorted by the key of their root (the queued element they start from);
// every sub-round each non-idle episode makes one pop; the front episode commits when it is idle; claims carry the
// root key as priority (atomicMin in a kernel).  A lost claim, or an episode that popped past a bound that later
// dropped, squashes that episode and everything behind it.  Measures committed pops per sub-round (= per dependent
// memory trip of a wave).
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
#define MAXW 1024
#define MAXP 256
typedef struct { int64_t idx; float v; int popped; } pushrec;
typedef struct { el root; int64_t id; int started, lost, over, npush, npop; float maxnew; int64_t cur; int have_cur; pushrec p[MAXP]; } episode;
static episode *win[MAXW]; static int nw;
static int64_t *c_owner; static int32_t *c_sub; static el *c_key;   // claim: owner id (0 = free), sub-round, priority key

static void ep_reset(episode *e) { e->started = 0; e->lost = 0; e->over = 0; e->npush = 0; e->npop = 0; e->maxnew = -INFINITY; e->cur = e->root.idx; e->have_cur = 1; }
static void ep_release(episode *e) { for (int j = 0; j < e->npush; ++j) if (c_owner[e->p[j].idx] == e->id) c_owner[e->p[j].idx] = 0; }
static episode *find_ep(int64_t id) { for (int i = 0; i < nw; ++i) if (win[i]->id == id) return win[i]; return NULL; }

// stats: [0] sub-rounds [1] committed pops [2] committed episodes [3] squashed episodes [4] squashed pops [5] squash events (lost)
// [6] squash events (bound) [7] evictions [8] mid-window insertions [9] pop steps executed
int64_t window_sim(const float *aff, const int64_t *seeds, int64_t ns, const uint8_t *mask, uint32_t *out,
                   const int64_t *dims, int W, int LCAP, int detect_at_commit, int64_t *stats, int LANES, int T)
{
    int64_t N = dims[0] * dims[1] * dims[2], SZ = dims[1] * dims[2], SY = dims[2];
    const int64_t noff[6] = {-SZ, -SY, -1, 1, SY, SZ};
    const int ax[6] = {0, 1, 2, 2, 1, 0};
    c_owner = calloc(N, 8); c_sub = calloc(N, 4); c_key = calloc(N, sizeof(el));
    hcap = ns + 1024; hn = 0; h = malloc(sizeof(el) * hcap);
    for (int64_t i = 0; i < ns; ++i) out[seeds[i]] = (uint32_t)(i + 1);
    for (int64_t i = 0; i < ns; ++i) { el e = {0.f, 0, seeds[i]}; hpush(e); }
    int64_t gage = 0, next_id = 1; int32_t sub = 0;
    nw = 0;
    episode *pool = malloc(sizeof(episode) * (MAXW + 1)); episode *freel[MAXW + 1]; int nfree = 0;
    for (int i = 0; i <= MAXW; ++i) freel[nfree++] = &pool[i];
    while (hn > 0 || nw > 0) {
        ++sub; ++stats[0];
        // ---- A: the window holds the W smallest roots --------------------------------------------------
        while (hn > 0 && (nw < W || less(&h[0], &win[nw - 1]->root))) {
            if (nw == W) {                      // evict the last root
                episode *e = win[--nw]; ep_release(e); stats[3] += e->started; stats[4] += e->npop; ++stats[7];
                el back = e->root; freel[nfree++] = e;
                el in = hpop(); hpush(back); hpush(in);      // keep `in` the minimum
            }
            el r = hpop();
            episode *e = freel[--nfree]; e->root = r; e->id = next_id++; ep_reset(e);
            int p = nw; while (p > 0 && less(&r, &win[p - 1]->root)) { win[p] = win[p - 1]; --p; }
            if (p < nw) ++stats[8];
            win[p] = e; ++nw;
        }
        // bounds; an episode that popped past its (new) bound is wrong from there on
        float qmin = hn > 0 ? h[0].v : INFINITY;
        int squash_from = -1, why = 0;
        for (int s = 0; s < nw; ++s) {
            float B = s + 1 < nw ? win[s + 1]->root.v : INFINITY; if (qmin < B) B = qmin;
            if (win[s]->maxnew >= B) { win[s]->over = 1; }
            if ((win[s]->over || (win[s]->lost && !detect_at_commit)) && squash_from < 0) { squash_from = s; why = win[s]->over ? 2 : 1; }
        }
        if (squash_from >= 0) {
            ++stats[4 + why];
            for (int s = squash_from; s < nw; ++s) { episode *e = win[s]; stats[3] += e->started; stats[4] += e->npop; ep_release(e); ep_reset(e); }
        }
        // ---- B: one pop per non-idle episode --------------------------------------------------------------
        // phase 1: claims
        static unsigned char act[MAXW]; int nact = 0;
        for (int s = 0; s < nw; ++s) { act[s] = win[s]->have_cur && nact < LANES && (win[s]->npush + 6 <= T || !win[s]->started); nact += act[s]; }
        for (int s = 0; s < nw; ++s) {
            episode *e = win[s];
            if (!act[s]) continue;
            for (int d = 0; d < 6; ++d) {
                int64_t nb = e->cur + noff[d];
                if (!mask[nb] || out[nb]) continue;
                if (c_owner[nb]) {
                    if (c_owner[nb] == e->id) continue;
                    if (!less(&e->root, &c_key[nb])) continue;            // held by a higher priority
                    if (c_sub[nb] < sub) { episode *o = find_ep(c_owner[nb]); if (o) o->lost = 1; }
                }
                c_owner[nb] = e->id; c_key[nb] = e->root; c_sub[nb] = sub;
            }
        }
        // phase 2: read back, push, choose next
        for (int s = 0; s < nw; ++s) {
            episode *e = win[s];
            float B = s + 1 < nw ? win[s + 1]->root.v : INFINITY; if (qmin < B) B = qmin;
            if (act[s]) {
                e->started = 1; ++e->npop; ++stats[9];
                for (int d = 0; d < 6; ++d) {
                    int64_t nb = e->cur + noff[d];
                    if (!mask[nb] || out[nb]) continue;
                    if (!(c_owner[nb] == e->id && c_sub[nb] == sub)) continue;
                    if (e->npush == MAXP) { fprintf(stderr, "MAXP\n"); exit(1); }
                    float v = aff[(int64_t)ax[d] * N + (d < 3 ? e->cur : nb)] + 0.0f;
                    e->p[e->npush].idx = nb; e->p[e->npush].v = v; e->p[e->npush].popped = 0; ++e->npush;
                }
                e->have_cur = 0;
            }
            if (e->have_cur) continue;            // waiting for a lane or for room in its list
            int best = -1, pending = 0;
            for (int j = 0; j < e->npush; ++j) if (!e->p[j].popped) { ++pending; if (best < 0 || e->p[j].v < e->p[best].v) best = j; }
            if (best >= 0 && e->p[best].v < B) {     // a full private queue stalls until it is the front
                e->p[best].popped = 1; e->cur = e->p[best].idx; e->have_cur = 1; if (e->p[best].v > e->maxnew) e->maxnew = e->p[best].v;
            }
        }
        // ---- C: commit idle episodes from the front ---------------------------------------------------------
        while (nw > 0) {
            episode *e = win[0];
            int force = e->npush + 6 > T && e->started;
            if (force && e->have_cur) {           // partial commit: the chosen-but-unpopped element goes back to the queue
                for (int j = 0; j < e->npush; ++j) if (e->p[j].popped && e->p[j].idx == e->cur) e->p[j].popped = 0;
                e->have_cur = 0;
                e->maxnew = -INFINITY; for (int j = 0; j < e->npush; ++j) if (e->p[j].popped && e->p[j].v > e->maxnew) e->maxnew = e->p[j].v;
            }
            if (e->have_cur || !e->started) break;
            float q = hn > 0 ? h[0].v : INFINITY;
            if (hn > 0 && less(&h[0], &e->root)) break;              // a residual sorts before this root: insert first
            if (e->maxnew >= q) break;                                // next sub-round's bound check squashes it
            float B = 1 < nw ? win[1]->root.v : INFINITY; if (q < B) B = q;
            int idle = 1;
            for (int j = 0; j < e->npush; ++j) if (!e->p[j].popped && e->p[j].v < B) idle = 0;
            if (!idle && !force) break;                                         // continues next sub-round under the updated bound
            if (e->lost) {
                ++stats[5];
                for (int s = 0; s < nw; ++s) { episode *x = win[s]; stats[3] += x->started; stats[4] += x->npop; ep_release(x); ep_reset(x); }
                break;
            }
            uint32_t lab = out[e->root.idx];
            stats[1] += e->npop; ++stats[2];
            for (int j = 0; j < e->npush; ++j) {
                ++gage; out[e->p[j].idx] = lab; c_owner[e->p[j].idx] = 0;
                if (!e->p[j].popped) { el r = {e->p[j].v, gage, e->p[j].idx}; hpush(r); }
            }
            freel[nfree++] = e;
            for (int s = 1; s < nw; ++s) win[s - 1] = win[s];
            --nw;
        }
    }
    free(h); free(c_owner); free(c_sub); free(c_key); free(pool);
    return stats[0];
}
