// CPU model of flood_group_kernel (iterseg_amd/csrc/ws_flood_group.h): the ordered speculative window of
// ws_flood_window.h re-cast for one WORK-GROUP per component -- one episode per lane (state in registers), discovery
// lists in a chunk pool, candidates (unpopped discoveries below the bound) in registers, commit bursts as wide as the
// window, residuals merged into the run as a batch.  Every phase below is a loop over lanes; the kernel runs the same
// phases SPMD with a barrier where this file has a loop boundary.  Claims arrive in a random lane order (`seed`) to model
// the unordered arrival of atomics from eight waves.  Checked against the sequential oracle by scratch/group_sim.py.
//
// Simplification: the cold tier (HBM buckets) is an exact priority queue here with a coarse bucket function, so the
// tier rule and the eviction / drain paths are exercised; the kernel reuses cold_append / cold_drain of ws.hip.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef uint64_t u64;
typedef uint32_t u32;
#define NONE 0xFFFFFFFFu
#define POPPED 0xFFFFFFFFu
#define ST_FREE (~0ull)
#define WMAX 512
#ifndef CMAX
#define CMAX 6
#endif
#define CH 8
#ifndef NCH
#define NCH 1024
#endif
#define RCAP 2048
#define PENDCAP 1024
typedef struct { u64 key; u32 vox, lab; } ent;

static u32 ford(float f) { u32 b; memcpy(&b, &f, 4); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }

// ---- cold tier model: binary heap on key ------------------------------------------------------------------
static ent *ch_; static int64_t chn, chcap;
static void cpush(ent e) { if (chn == chcap) { chcap = chcap ? 2 * chcap : 1024; ch_ = realloc(ch_, sizeof(ent) * chcap); } int64_t c = chn++; while (c > 0) { int64_t p = (c - 1) >> 1; if (!(e.key < ch_[p].key)) break; ch_[c] = ch_[p]; c = p; } ch_[c] = e; }
static ent cpop(void) { ent top = ch_[0], last = ch_[--chn]; if (chn > 0) { int64_t c = 0; for (;;) { int64_t l = 2 * c + 1, r = l + 1, m; if (l >= chn) break; m = (r < chn && ch_[r].key < ch_[l].key) ? r : l; if (!(ch_[m].key < last.key)) break; ch_[c] = ch_[m]; c = m; } ch_[c] = last; } return top; }
static u32 bucket(u64 key) { return (u32)(key >> 44); }          // coarse: 2^12 values per bucket

typedef struct {
    int has, started;
    u64 key; u32 root, lab;
    u32 cur, curv; int cur_addr, cur_li;
    int np, npop; u32 mx, resmin;
    int ncand; u32 cv[CMAX], cx[CMAX]; int cli[CMAX], cad[CMAX];
    int ndef;
    int first, last, spare;
} slot_t;

static slot_t S[WMAX];
static u32 pvox[NCH * CH], pval[NCH * CH]; static int cnext[NCH], freech[NCH], nfreech;
static u64 okey[WMAX]; static int oslot[WMAX]; static int oh, nw; static u64 succ[WMAX];
static int freeslot[WMAX], nfreeslot;
static int resch[64], nres;
static u64 ksq_pool = ~0ull;
static ent run[RCAP]; static int shead, scount;
static ent pend[PENDCAP];
static int64_t stats[32];

static int ord_at(int p) { return (oh + p) % WMAX; }
static int last_slot(void) { return nw > 0 ? oslot[ord_at(nw - 1)] : -1; }

static void ep_fresh(slot_t *e) { e->started = 0; e->cur = e->root; e->curv = 0; e->cur_addr = -1; e->cur_li = -1; e->np = e->npop = 0; e->mx = 0; e->resmin = 0xFFFFFFFFu; e->ncand = e->ndef = 0; e->first = e->last = e->spare = -1; }
static void free_chunks(slot_t *e) { for (int c = e->first; c >= 0;) { int n = cnext[c]; freech[nfreech++] = c; c = n; } if (e->spare >= 0) freech[nfreech++] = e->spare; e->first = e->last = e->spare = -1; }
static int entry_addr(const slot_t *e, int i) { int c = e->first; for (int k = 0; k < i / CH; ++k) c = cnext[c]; return c * CH + i % CH; }
static void release_claims(slot_t *e, u64 *state) { for (int i = 0; i < e->np; ++i) { int a = entry_addr(e, i); if (state[pvox[a]] == e->key) state[pvox[a]] = ST_FREE; } }
static void squash_one(slot_t *e, u64 *state) { if (e->started) { ++stats[12]; stats[4] += e->npop; release_claims(e, state); } free_chunks(e); ep_fresh(e); }

// candidate order: (value, discovery index)
static int cand_less(u32 v0, int i0, u32 v1, int i1) { return v0 < v1 || (v0 == v1 && i0 < i1); }
static void cand_add(slot_t *e, u32 v, u32 x, int li, int ad)
{
    // invariant: every deferred entry (below the bound, unpopped, only in the list) sorts after every register candidate
    int mk = -1;
    for (int k = 0; k < e->ncand; ++k) if (mk < 0 || cand_less(e->cv[mk], e->cli[mk], e->cv[k], e->cli[k])) mk = k;
    const int below_max = mk >= 0 && cand_less(v, li, e->cv[mk], e->cli[mk]);
    if (e->ncand < CMAX && (e->ndef == 0 || below_max)) { int k = e->ncand++; e->cv[k] = v; e->cx[k] = x; e->cli[k] = li; e->cad[k] = ad; return; }
    if (e->ncand == CMAX && below_max) { e->cv[mk] = v; e->cx[mk] = x; e->cli[mk] = li; e->cad[mk] = ad; }     // the old maximum is deferred
    ++e->ndef;
}
// rebuild candidates / deferred count / resmin from the list (bound = ordered value)
static void reload(slot_t *e, u32 bound)
{
    e->ncand = 0; e->ndef = 0; e->resmin = 0xFFFFFFFFu;
    ++stats[16];
    for (int i = 0; i < e->np; ++i) {
        int a = entry_addr(e, i);
        u32 v = pval[a];
        if (v == POPPED) continue;
        if (e->cur != NONE && a == e->cur_addr) continue;       // in hand
        if (v < bound) cand_add(e, v, pvox[a], i, a);
        else if (v < e->resmin) e->resmin = v;
    }
}
// next pop: the smallest unpopped discovery if it sorts below the bound
static void select_next(slot_t *e, u32 bound)
{
    if (e->ncand == 0 && e->ndef > 0) reload(e, bound);
    int mk = -1;
    for (int k = 0; k < e->ncand; ++k) if (mk < 0 || cand_less(e->cv[k], e->cli[k], e->cv[mk], e->cli[mk])) mk = k;
    if (mk >= 0 && e->cv[mk] < bound) {
        e->cur = e->cx[mk]; e->curv = e->cv[mk]; e->cur_addr = e->cad[mk]; e->cur_li = e->cli[mk];
        e->cv[mk] = e->cv[e->ncand - 1]; e->cx[mk] = e->cx[e->ncand - 1]; e->cli[mk] = e->cli[e->ncand - 1]; e->cad[mk] = e->cad[e->ncand - 1];
        --e->ncand;
        return;
    }
    // idle: whatever is left sorts at or above the bound -> residuals
    for (int k = 0; k < e->ncand; ++k) if (e->cv[k] < e->resmin) e->resmin = e->cv[k];
    e->ncand = 0;
    if (e->ndef > 0) reload(e, bound);              // deferred ones are >= every candidate, i.e. >= bound: fold them in
    e->ncand = 0; e->ndef = 0;                       // (reload put nothing below the bound: all of them are >= bound)
    e->cur = NONE;
}

// ---- queue pieces ------------------------------------------------------------------------------------------
static int cold_on = 1; static u32 cold_low = 0xFFFFFFFFu;
static void cold_put(ent e) { cpush(e); if (bucket(e.key) < cold_low) cold_low = bucket(e.key); ++stats[6]; }
static void run_evict(int keep) { while (scount > keep) { cold_put(run[(shead + scount - 1) % RCAP]); --scount; ++stats[7]; } }
static int run_pos(u64 key) { int lo = 0, hi = scount; while (lo < hi) { int m = (lo + hi) / 2; if (run[(shead + m) % RCAP].key < key) lo = m + 1; else hi = m; } return lo; }

static void win_set_succ(void)
{   // recompute succ[] (the kernel updates two entries per insertion; the model recomputes)
    for (int p = 0; p < nw; ++p) succ[oslot[ord_at(p)]] = p + 1 < nw ? okey[ord_at(p + 1)] : ~0ull;
}
static void win_insert(ent r, u64 *state, int wcap)
{
    int pos = 0;
    while (pos < nw && okey[ord_at(pos)] < r.key) ++pos;
    if (nw == wcap || nfreeslot == 0) {
        // full: the last root goes back to the front of the run
        int ls = last_slot(); slot_t *e = &S[ls];
        ent back = {e->key, e->root, e->lab};
        squash_one(e, state); e->has = 0; freeslot[nfreeslot++] = ls; --nw;
        shead = (shead + RCAP - 1) % RCAP; run[shead] = back; ++scount; ++stats[8];
        if (pos > nw) pos = nw;
    }
    int s = freeslot[--nfreeslot];
    for (int p = nw; p > pos; --p) { okey[ord_at(p)] = okey[ord_at(p - 1)]; oslot[ord_at(p)] = oslot[ord_at(p - 1)]; }
    okey[ord_at(pos)] = r.key; oslot[ord_at(pos)] = s; ++nw;
    slot_t *e = &S[s]; e->has = 1; e->key = r.key; e->root = r.vox; e->lab = r.lab; ep_fresh(e);
    if (pos < nw - 1) ++stats[9];
    win_set_succ();
}

static void enqueue_batch(int P, u64 *state, int wcap, int run_limit)
{
    if (P == 0) return;
    // capacity first: the eviction moves the hot maximum, the batch is classified against the tiers AFTER it
    if (scount + P > run_limit) { run_evict(run_limit / 2 < RCAP - PENDCAP ? run_limit / 2 : RCAP - PENDCAP); }
    int hot_ne = nw > 0 || scount > 0;
    u64 hot_max = scount > 0 ? run[(shead + scount - 1) % RCAP].key : (nw > 0 ? okey[ord_at(nw - 1)] : 0);
    static ent hot[PENDCAP]; int H = 0;
    for (int i = 0; i < P; ++i) {
        ent r = pend[i];
        int tocold = cold_on && chn > 0 && !(hot_ne && r.key < hot_max) && bucket(r.key) >= cold_low;
        if (tocold) { cold_put(r); continue; }
        hot[H++] = r;
    }
    // window-bound ones first, one at a time against the CURRENT last root
    static ent rb[PENDCAP]; int R = 0;
    for (int i = 0; i < H; ++i) {
        if (nw > 0 && hot[i].key < okey[ord_at(nw - 1)]) {
            if ((nw == wcap || nfreeslot == 0) && scount >= RCAP - 2) run_evict(run_limit / 2);
            win_insert(hot[i], state, wcap);
        } else rb[R++] = hot[i];
    }
    // the rest: one merge into the sorted run
    if (scount + R > RCAP) { fprintf(stderr, "run overflow\n"); exit(1); }
    for (int i = 0; i < R; ++i) {            // (model: insertion sort; the kernel ranks and scatters)
        int pos = run_pos(rb[i].key);
        for (int p = scount; p > pos; --p) run[(shead + p) % RCAP] = run[(shead + p - 1) % RCAP];
        run[(shead + pos) % RCAP] = rb[i]; ++scount; ++stats[3];
    }
}

static void refill(int wcap, int drain_n)
{
    for (;;) {
        if (scount == 0 && chn > 0) {
            // drain: the lowest bucket (all of it, at most drain_n ... the model takes the drain_n smallest, which is a
            // prefix of the lowest buckets -- still "every cold element sorts after every hot one")
            int n = 0; shead = 0;
            while (chn > 0 && n < drain_n) run[n++] = cpop();
            scount = n; cold_low = chn > 0 ? bucket(ch_[0].key) : 0xFFFFFFFFu; ++stats[5];
        }
        if (!(nw < wcap && scount > 0 && nfreeslot > 0)) break;
        ent r = run[shead]; shead = (shead + 1) % RCAP; --scount;
        int s = freeslot[--nfreeslot];
        okey[ord_at(nw)] = r.key; oslot[ord_at(nw)] = s; ++nw;
        slot_t *e = &S[s]; e->has = 1; e->key = r.key; e->root = r.vox; e->lab = r.lab; ep_fresh(e);
    }
    win_set_succ();
}

int64_t group_sim(const float *aff, const int64_t *seeds, int64_t ns, const uint8_t *mask, u32 *out, const int64_t *dims,
                  int wcap, int tcap, int run_limit, int drain_n, int max_bursts, unsigned seed, int64_t *st_out, int kmax)
{
    const int64_t N = dims[0] * dims[1] * dims[2], SZ = dims[1] * dims[2], SY = dims[2];
    const int64_t noff[6] = {-SZ, -SY, -1, 1, SY, SZ};
    const int ax[6] = {0, 1, 2, 2, 1, 0};
    const u64 ZKEY = (u64)0x80000000u << 32;
    memset(stats, 0, sizeof(stats));
    u64 *state = malloc(sizeof(u64) * N);
    for (int64_t i = 0; i < ns; ++i) out[seeds[i]] = (u32)(i + 1);
    for (int64_t i = 0; i < N; ++i) state[i] = mask[i] ? (out[i] ? (u64)out[i] : ST_FREE) : 0ull;
    chn = 0; cold_low = 0xFFFFFFFFu;
    nfreech = 0; for (int c = NCH - 1; c >= 0; --c) freech[nfreech++] = c;
    const int reserve = (tcap + CH - 1) / CH + 1;
    nres = 0; for (int i = 0; i < reserve; ++i) resch[nres++] = freech[--nfreech];
    ksq_pool = ~0ull;
    nfreeslot = 0; for (int s = WMAX - 1; s >= 0; --s) { S[s].has = 0; freeslot[nfreeslot++] = s; }
    oh = 0; nw = 0; shead = 0; scount = 0;
    if (wcap > WMAX) wcap = WMAX;
    u32 agebase = (u32)ns - 1;
    // seeds: value 0, age = rank in INDEX order (the reference pushes them all with age 0: ties fall to the index)
    int64_t *sorted = malloc(sizeof(int64_t) * ns);
    memcpy(sorted, seeds, sizeof(int64_t) * ns);
    for (int64_t i = 1; i < ns; ++i) { int64_t v = sorted[i], j = i; while (j > 0 && sorted[j - 1] > v) { sorted[j] = sorted[j - 1]; --j; } sorted[j] = v; }
    for (int64_t j = 0; j < ns; ++j) {
        ent e = {ZKEY | (u64)j, (u32)sorted[j], out[sorted[j]]};
        if (j < wcap) { int s = freeslot[--nfreeslot]; okey[nw] = e.key; oslot[nw] = s; ++nw; S[s].has = 1; S[s].key = e.key; S[s].root = e.vox; S[s].lab = e.lab; ep_fresh(&S[s]); }
        else if (scount < run_limit) run[scount++] = e;
        else cold_put(e);
    }
    win_set_succ();
    srand(seed);
    int order[WMAX];
    while (nw > 0) {
        ++stats[0];
        if (scount > run_limit) run_evict(run_limit / 2);
        const u64 qhead = scount > 0 ? run[shead].key : ~0ull;
        const int ls = last_slot();
        // ---- A: bounds.  An episode that popped (or holds) something at or above its bound is wrong ----------------
        u64 ksq = ~0ull;
        for (int s = 0; s < WMAX; ++s) {
            slot_t *e = &S[s];
            if (!e->has || !e->started) continue;
            const u64 nk = s == ls ? qhead : succ[s];
            const u32 bound = (u32)(nk >> 32);
            u32 mx = e->mx;
            if (e->cur != NONE && e->cur_li >= 0 && e->curv > mx) mx = e->curv;
            if (mx != 0 && mx >= bound && e->key < ksq) ksq = e->key;
            // candidates that the (lower) bound turned into residuals
            for (int k = 0; k < e->ncand;) {
                if (e->cv[k] >= bound) { if (e->cv[k] < e->resmin) e->resmin = e->cv[k]; --e->ncand; e->cv[k] = e->cv[e->ncand]; e->cx[k] = e->cx[e->ncand]; e->cli[k] = e->cli[e->ncand]; e->cad[k] = e->cad[e->ncand]; }
                else ++k;
            }
        }
        if (ksq != ~0ull) { ++stats[14]; for (int s = 0; s < WMAX; ++s) if (S[s].has && S[s].key >= ksq) squash_one(&S[s], state); }
        // ---- C: free-running steps.  Every wave (64 slots) pops on its own -- no barrier between its steps, claims of the
        //         waves interleave arbitrarily (here: waves in random order each step) -- until the front episode is idle or
        //         stalled, or the wave has nothing left to pop.  A claim is won iff the voxel was free or held by a later
        //         episode; its previous holder (with everything behind it) is squashed when the waves are back at the barrier.
        int front_forced = 0, front_ready = 0;
        const int fs = oslot[ord_at(0)];
        int wave_done[WMAX / 64]; for (int w = 0; w < WMAX / 64; ++w) wave_done[w] = 0;
        if (S[fs].has && S[fs].started && S[fs].cur == NONE) front_ready = 1;
        for (int step = 0; step < kmax; ++step) {
            int worder[WMAX / 64];
            for (int w = 0; w < WMAX / 64; ++w) worder[w] = w;
            for (int w = WMAX / 64 - 1; w > 0; --w) { int jj = rand() % (w + 1), t = worder[w]; worder[w] = worder[jj]; worder[jj] = t; }
            int any_wave = 0;
            for (int wi = 0; wi < WMAX / 64; ++wi) {
                const int w = worder[wi];
                if (wave_done[w]) continue;
                if (step > 0 && front_ready) { wave_done[w] = 1; continue; }
                int any = 0;
                for (int s = 64 * w; s < 64 * w + 64; ++s) {
                    slot_t *e = &S[s];
                    if (!e->has || e->cur == NONE) continue;
                    int room = 1;
                    if (e->started) {
                        room = e->np + 6 <= tcap;
                        const int used = e->np % CH, left = (e->np == 0 || used == 0) ? 0 : CH - used;
                        if (room && left < 6 && e->spare < 0) { if (nfreech > 0) e->spare = freech[--nfreech]; else if (s == fs && nres > 0) e->spare = resch[--nres]; else room = 0; }
                    } else if (e->spare < 0) { if (nfreech > 0) e->spare = freech[--nfreech]; else if (s == fs && nres > 0) e->spare = resch[--nres]; else room = 0; }
                    if (!room) {
                        if (s == fs) { if (e->started) front_forced = 1; else if (s != ls && succ[s] < ksq_pool) ksq_pool = succ[s]; front_ready = 1; }
                        continue;
                    }
                    any = 1;
                    ++stats[2];
                    const u64 nk = s == ls ? qhead : succ[s];
                    const u32 bound = (u32)(nk >> 32);
                    if (e->started && e->cur_li >= 0) { pval[e->cur_addr] = POPPED; ++e->npop; if (e->curv > e->mx) e->mx = e->curv; }
                    e->started = 1;
                    const u32 cv = e->cur;
                    e->cur = NONE;
                    for (int k = 0; k < 6; ++k) {
                        const int64_t nbi = (int64_t)cv + noff[k];
                        const u64 oldv = state[nbi];
                        if (!(oldv > e->key)) continue;                  // not mine: off-mask (0), labelled, or an earlier episode's
                        state[nbi] = e->key;                             // atomicMin
                        // (its previous holder finds out when it commits: a lost discovery it has not popped is dropped there,
                        //  a lost discovery it HAS popped makes it -- and everything behind it -- wrong)
                        const u32 nb = (u32)nbi;
                        const u32 v = ford(aff[(int64_t)ax[k] * N + (k < 3 ? cv : nb)] + 0.0f);
                        if (e->np % CH == 0) {
                            const int c = e->spare; e->spare = -1; cnext[c] = -1;
                            if (e->first < 0) e->first = c; else cnext[e->last] = c;
                            e->last = c;
                        }
                        const int a = e->last * CH + e->np % CH;
                        pvox[a] = nb; pval[a] = v;
                        if (v < bound) cand_add(e, v, nb, e->np, a); else if (v < e->resmin) e->resmin = v;
                        ++e->np;
                    }
                    select_next(e, bound);
                    if (s == fs && e->cur == NONE) front_ready = 1;
                }
                if (!any) wave_done[w] = 1; else any_wave = 1;
            }
            if (!any_wave) break;
        }
        if (ksq_pool != ~0ull) { ++stats[11]; for (int s = 0; s < WMAX; ++s) if (S[s].has && S[s].key >= ksq_pool) squash_one(&S[s], state); ksq_pool = ~0ull; }
        // an idle episode appends nothing more: its spare chunk goes back (frees never run beside the allocations above)
        for (int s = 0; s < WMAX; ++s) if (S[s].has && S[s].started && S[s].cur == NONE && S[s].spare >= 0) { freech[nfreech++] = S[s].spare; S[s].spare = -1; }
        // ---- E: commit the idle episodes at the front, in order ------------------------------------------------
        for (int burst = 0; burst < max_bursts && nw > 0; ++burst) {
            const u64 qh = scount > 0 ? run[shead].key : ~0ull;
            const int cons = nw;            // all window positions take part
            u32 resmin[WMAX], pc[WMAX]; int idle[WMAX], forced0 = 0;
            int m = 0;
            for (int j = 0; j < cons; ++j) {
                slot_t *e = &S[oslot[ord_at(j)]];
                idle[j] = e->started && e->cur == NONE;
                resmin[j] = e->resmin; pc[j] = (u32)(e->np - e->npop);
                if (burst == 0 && j == 0 && front_forced && e->started && e->cur != NONE) {
                    // the front episode with a full list commits what it has done: the element in hand and every candidate go back
                    forced0 = 1; idle[j] = 1;
                    u32 rm = 0xFFFFFFFFu;
                    for (int i = 0; i < e->np; ++i) { u32 v = pval[entry_addr(e, i)]; if (v != POPPED && v < rm) rm = v; }
                    resmin[j] = rm; pc[j] = (u32)(e->np - e->npop);
                }
            }
            while (m < cons && idle[m]) ++m;
            int ncommit = 0; u32 F = (u32)(qh >> 32), pendn = 0;
            for (int j = 0; j < m; ++j) {
                slot_t *e = &S[oslot[ord_at(j)]];
                const u32 succv = j + 1 < nw ? (u32)(okey[ord_at(j + 1)] >> 32) : 0xFFFFFFFFu;
                const u32 maxnew = e->mx;
                const int ok = (u32)(e->key >> 32) <= F && (maxnew == 0 || (maxnew < F && maxnew < succv));
                if (!ok || pendn + pc[j] > PENDCAP) break;
                pendn += pc[j];
                if (resmin[j] < F) F = resmin[j];
                ++ncommit;
            }
            if (ncommit == 0) break;
            // verification: every POPPED discovery of a committing episode must still be its own; an unpopped one that an
            // earlier episode took in the meantime is dropped (sequentially this episode would never have discovered it)
            {
                u64 kinv = ~0ull; int cut = ncommit;
                for (int j = 0; j < ncommit && kinv == ~0ull; ++j) {
                    slot_t *e = &S[oslot[ord_at(j)]];
                    for (int i = 0; i < e->np; ++i) {
                        const int a = entry_addr(e, i);
                        if (state[pvox[a]] != e->key && pval[a] == POPPED) { kinv = e->key; cut = j; break; }
                    }
                }
                if (kinv != ~0ull) {
                    ++stats[19];
                    for (int s2 = 0; s2 < WMAX; ++s2) if (S[s2].has && S[s2].key >= kinv) squash_one(&S[s2], state);
                    ncommit = cut;
                }
            }
            if (ncommit == 0) break;
            if (forced0) ++stats[15];
            u32 po = 0;
            for (int j = 0; j < ncommit; ++j) {
                const int s = oslot[ord_at(j)]; slot_t *e = &S[s];
                stats[1] += e->npop + 1;
                for (int i = 0; i < e->np; ++i) {
                    const int a = entry_addr(e, i);
                    if (state[pvox[a]] != e->key) { ++stats[17]; continue; }           // lost, unpopped: dropped
                    out[pvox[a]] = e->lab; state[pvox[a]] = (u64)e->lab;
                    if (pval[a] != POPPED) { ent r = {((u64)pval[a] << 32) | (u64)(agebase + (u32)i + 1u), pvox[a], e->lab}; pend[po++] = r; }
                }
                agebase += (u32)e->np;
                free_chunks(e); e->has = 0; freeslot[nfreeslot++] = s;
            }
            oh = ord_at(ncommit); nw -= ncommit;
            win_set_succ();
            enqueue_batch((int)po, state, wcap, run_limit);
            if (ncommit < cons) break;
        }
        // ---- G: refill -----------------------------------------------------------------------------------------
        refill(wcap, drain_n);
        while (nres < reserve && nfreech > 0) resch[nres++] = freech[--nfreech];
        if (stats[0] > 64 * N + 4096) { fprintf(stderr, "watchdog\n"); break; }
    }
    free(state);
    if (st_out) memcpy(st_out, stats, sizeof(stats));
    return stats[0];
}
