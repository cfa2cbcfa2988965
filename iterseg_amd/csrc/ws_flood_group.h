// ws_flood_group.h -- the flood of one LARGE mask component by one WORK-GROUP of eight waves (512 lanes).
// Included by ws.hip inside namespace isx, after ws_flood_window.h (same rules, same queue tiers; scratch/group_sim.c is
// the CPU model of exactly these phases and is checked against the sequential oracle by scratch/group_sim.py).
//
// Replaces raveled_affinity_watershed (watershed.py:95-159) for components of >= group_min voxels, bit-exactly.
//
// The single-wave window kernel keeps 192 episodes in LDS and walks them with one wave: every sub-round it compacts an
// active list, loads and stores the episode words, scans the discovery lists for the next pop, and one lane at a time
// inserts residuals into the window / the run; the wave owns a whole CU's LDS while three of its four SIMDs idle.  Here
//   * ONE EPISODE PER LANE: slot = thread id, the episode's words live in registers, no active list, no slot loads;
//   * the window is 512 episodes wide and a discovery list may grow to 192 entries (chunk pool in LDS, 8 entries per chunk):
//     the model gives 36.8 committed pops per sub-round on the random-weight U-Net output against 17.3 (W = 192, T = 48),
//     and a tenth of the window insertions / evictions, which were mostly forced commits of long episodes;
//   * the unpopped discoveries BELOW the bound ("candidates") sit in six registers -- the next pop is a register minimum,
//     the list is append-only (commit, squash and the rare overflow re-read it);
//   * a squash is "every episode with key >= K" -- each lane compares its own key;
//   * a commit burst is as wide as the window (prefix minimum / prefix sums over 512 lanes), the committing lanes write
//     their own labels and residuals, and the residuals of a burst enter the run in ONE ranked merge;
//   * the cold tier (HBM buckets: cold_append / cold_drain of ws.hip) is driven by wave 0.
// Phases are separated by work-group barriers; wave-uniform queue state lives in LDS (`sh`).
//
// Correctness argument: as ws_flood_window.h (induction over window positions; the front episode is never wrong).  What is
// new is only WHO does the work.  Claims of one sub-round arrive in any order from the eight waves: after the barrier that
// ends the claim phase the holder of a voxel is the smallest claiming key, every lane reads its claims back, and a
// take-over that hit a claim of an EARLIER sub-round (take-overs != fresh losses, then matched entry by entry) squashes
// from the victim's key.

#ifndef ISX_GRP_W
#define ISX_GRP_W 512
#endif
static constexpr int GRP_W = ISX_GRP_W;           // lanes = episode slots
static constexpr int GRP_NWV = GRP_W / 64;        // waves
static constexpr int GRP_T = GRP_W > 256 ? 128 : 192;   // discoveries one episode may hold (LDS: the chunk table is GRP_W x GRP_T / 8)
static constexpr int GRP_CH = 8;                  // entries per chunk
static constexpr int GRP_NCH = GRP_W > 256 ? 640 : GRP_W + GRP_W / 2;   // chunks in the pool
static constexpr int GRP_CMAX = 6;                // register candidates
static constexpr int GRP_PEND = 2 * GRP_W;        // residuals of one commit burst (enqueued in slices of one per lane)
static constexpr int GRP_STAGE = GRP_W + 64;      // staged cold-bound elements (flushed down to < 64 after every slice)
static constexpr int GRP_TL = 256;                // take-over / fresh-loss records of one sub-round
static constexpr uint32_t GRP_NONE = 0xFFFFFFFFu;
static constexpr unsigned long long GROUP_MIN_SIZE = 65536;    // components from this size flood on the work-group kernel (ISX_FLOOD_GROUP_MIN; 0 = never).  Round 5, after round 4's leaner step: structured 512^3 seeds 0 / 4 74 -> 58 / 79 -> 62 ms against 262144 (their largest components, 100-180 k voxels, were the whole stage on the one-wave kernel); 32768 and 16384 measure the same.  Measured r3 (at 262144):
                                                               // config 2 (one 8.3 M chain-like component) 3.01 -> 1.55 s, blobs 256^3 (one 5.4 M) 0.95 -> 0.70 s, structured 512^3 +3..6 %

enum { GI_OH, GI_NW, GI_SHEAD, GI_SCOUNT, GI_NSTAGE, GI_NFREECH, GI_NFREESLOT, GI_SERIAL, GI_FORCED, GI_NTOOK, GI_NLOST, GI_NT, GI_NL,
       GI_DIRTY, GI_COLD_COUNT, GI_COLD_LOW, GI_M, GI_NCOMMIT, GI_PENDN, GI_NHOT, GI_NWB, GI_AGEBASE, GI_NPTOT, GI_ANY, GI_TAKE, GI_POS,
       GI_EVICT, GI_NSPILL, GI_REQ, GI_NRES, GI_WB0, GI_FRONT, GI_LAST, GI_NRELOAD, GI_READY, GI_SORTN, GI_PEND0,
       GI_N };

// A control word in LDS that steers a wave-uniform branch.  Other waves may write such a word while this wave runs between two
// barriers, so it is read through ONE helper: a volatile load (the compiler may not cache it across iterations, hoist it, or
// duplicate it into the arms of a divergent branch) whose first active lane's value is taken for the whole wave (readfirstlane is
// convergent).  Round 4: a plain read of GI_READY in the free-running step loop split a wave, and the DPP scans of coop_reload then
// ran under a partial EXEC mask.  Every read of a GI_* word that decides control flow goes through this.
__device__ __forceinline__ int grp_flag(const int *shI, int i)
{
    return __builtin_amdgcn_readfirstlane(*(const volatile int *)(shI + i));
}
enum { GQ_KSQ, GQ_TMIN, GQ_QHEAD, GQ_KINV, GQ_N };
enum { GS_SUB, GS_POPS, GS_EXEC, GS_SQ_EPS, GS_SQ_BOUND, GS_SQ_LOST, GS_SLOW, GS_FORCED, GS_RELOAD, GS_RUNINS, GS_WININS, GS_WINEV, GS_COLD,
       GS_DRAIN, GS_BURST, GS_C_NOTREADY, GS_C_NOTSTARTED, GS_C_RUNNING, GS_C_BOUND, GS_C_CAP, GS_C_ALL, GS_NWSUM, GS_F_BULK, GS_F_CALLS, GS_F_FLUSHES, GS_N };

// ---- the cold tier, driven by wave 0, OUT OF LINE: cold_append / cold_drain are large, and inlined into the kernel their
// registers pushed the episode words of the hot loop into scratch memory (a scratch round trip per phase).  The tier's
// state (`Cold`) lives in LDS between calls; count / low are published in shI for the other waves. --------------------------------
__device__ __forceinline__ void grp_publish_cold(const Cold &cold, int *shI, int lane)
{
    if (lane == 0) { shI[GI_COLD_COUNT] = (int)cold.count; shI[GI_COLD_LOW] = (int)cold.low; }
}

// append stage[0, nstage) to the buckets, 64 at a time
__device__ __noinline__ void grp_w0_flush_stage(Cold *cs, HeapEnt *stage, int *shI, uint32_t *lstat, int lane, HeapEnt *tmp)
{
    Cold cold = *cs;
    int nstage = shI[GI_NSTAGE];
    int done = 0;
    if (nstage > 64 && tmp) {
        // The elements of the first one's bucket (on a plateau: nearly all of them) leave in ONE append (cold_append_bucket: one
        // state update, at most one new segment); the others are closed up in place, in order, for the 64-at-a-time path.
        // `tmp`: GRP_W free entries of LDS (the run-bound batch buffer, empty by now).
        const uint32_t b0 = cold_bucket(cold, stage[0].key);
        int nm = 0, nn = 0;
        for (int j0 = 0; j0 < nstage; j0 += 64) {
            const int j = j0 + lane;
            HeapEnt e; e.key = 0; e.idx = 0; e.pad = 0;
            const bool hs = j < nstage;
            if (hs) e = stage[j];
            const bool mt = hs && cold_bucket(cold, e.key) == b0;
            const unsigned long long mb = __ballot(mt), ob = __ballot(hs && !mt);
            const unsigned long long below = (1ull << lane) - 1ull;
            const int room = GRP_W - nm;                        // the batch buffer holds GRP_W entries; what does not fit stays behind
            const int mr = __popcll(mb & below);
            const bool take = mt && mr < room;
            const bool keep = hs && !take;
            const unsigned long long kb = __ballot(keep);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the whole chunk is in registers before the in-place writes)
            if (take) tmp[nm + mr] = e;
            if (keep) stage[nn + __popcll(kb & below)] = e;
            nm += __popcll(mb) < room ? __popcll(mb) : room;
            nn += __popcll(kb);
            (void)ob;
        }
        if (nm > 0) {
            bool asc = true;
            for (int j = lane; j < nm; j += 64) if (j > 0) asc = asc && tmp[j - 1].key <= tmp[j].key;
            cold_append_bucket(cold, tmp, nm, b0, __all(asc) != 0, tmp[0].key, tmp[nm - 1].key, lane);
#ifdef ISX_FLOOD_PROF
            if (lane == 0 && !cold.failed) atomicAdd(&lstat[GS_F_BULK], 1u);
#endif
        }
        if (cold.failed) {
            // nothing was written: put the batch back in front of the rest (GRP_STAGE holds both) for the serial continuation
            for (int j0 = ((nn - 1) / 64) * 64; j0 >= 0 && nn > 0; j0 -= 64) {
                HeapEnt e; e.key = 0; e.idx = 0; e.pad = 0;
                if (j0 + lane < nn) e = stage[j0 + lane];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (j0 + lane < nn) stage[j0 + lane + nm] = e;
            }
            for (int j = lane; j < nm; j += 64) stage[j] = tmp[j];
            nn += nm;
        } else if (lane == 0 && nm > 0) atomicAdd(&lstat[GS_COLD], (uint32_t)nm);
        nstage = nn;
    }
#ifdef ISX_FLOOD_PROF
    if (lane == 0) { atomicAdd(&lstat[GS_F_FLUSHES], 1u); atomicAdd(&lstat[GS_F_CALLS], (uint32_t)((nstage - done + 63) / 64)); }
#endif
    while (done < nstage && !cold.failed) {
        const int m = nstage - done < 64 ? nstage - done : 64;
        const bool hs = lane < m;
        HeapEnt e; e.key = 0; e.idx = 0; e.pad = 0;
        if (hs) e = stage[done + lane];
        cold_append(cold, hs, e.key, e.idx, e.pad, lane);
        if (cold.failed) break;                      // nothing of this batch was written
        done += m;
    }
    if (lane == 0) atomicAdd(&lstat[GS_COLD], (uint32_t)done);
    if (cold.failed) {
        // keep what is left at the front of `stage` for the serial continuation
        for (int j0 = done; j0 < nstage; j0 += 64) {
            HeapEnt e; e.key = 0; e.idx = 0; e.pad = 0;
            if (j0 + lane < nstage) e = stage[j0 + lane];
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (j0 + lane < nstage) stage[j0 + lane - done] = e;
        }
        if (lane == 0) { shI[GI_NSTAGE] = nstage - done; shI[GI_SERIAL] = 1; }
    } else if (lane == 0) shI[GI_NSTAGE] = 0;
    grp_publish_cold(cold, shI, lane);
    *cs = cold;
}


// the largest run elements go to their buckets (ascending, so that a bucket's list stays FIFO)
__device__ __noinline__ void grp_w0_run_evict(Cold *cs, HeapEnt *srun, int *shI, uint32_t *lstat, unsigned long long *stats, int keep, int lane)
{
    constexpr int SMASK = COOP_SCAP - 1;
    const int shead = shI[GI_SHEAD], scount = shI[GI_SCOUNT];
    if (scount <= keep) return;
    Cold cold = *cs;
    if (!cold.on) { if (lane == 0) shI[GI_SERIAL] = 1; return; }
    const int ev = scount - keep;
    int done = 0;
    while (done < ev) {
        const int m = ev - done < 64 ? ev - done : 64;
        const bool hs = lane < m;
        HeapEnt e; e.key = 0; e.idx = 0; e.pad = 0;
        if (hs) e = srun[(shead + keep + done + lane) & SMASK];
        cold_append(cold, hs, e.key, e.idx, e.pad, lane);
        if (cold.failed) break;
        done += m;
    }
    if (cold.failed) {
        for (int j0 = keep + done; j0 < scount; j0 += 64) {      // close the gap
            const bool mv = j0 + lane < scount;
            HeapEnt e; e.key = 0; e.idx = 0; e.pad = 0;
            if (mv) e = srun[(shead + j0 + lane) & SMASK];
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (mv) srun[(shead + j0 + lane - done) & SMASK] = e;
        }
        if (lane == 0) { shI[GI_SCOUNT] = scount - done; shI[GI_SERIAL] = 1; }
    } else {
        if (lane == 0) { shI[GI_SCOUNT] = keep; atomicAdd(&lstat[GS_COLD], (uint32_t)ev); }
        // always-on tier assertion (cheap form): the hot maximum may not lie in a bucket above the lowest cold one
        if (lane == 0 && keep > 0 && stats) {
            const unsigned long long hm = srun[(shead + keep - 1) & SMASK].key;
            if (cold.count > 0 && cold_bucket(cold, hm) > cold.low) atomicAdd(&stats[16], 1ull);
        }
    }
    grp_publish_cold(cold, shI, lane);
    *cs = cold;
}

// the run is empty: staged elements to their buckets, then the lowest bucket into the run
__device__ __noinline__ void grp_w0_drain(Cold *cs, HeapEnt *srun, HeapEnt *stage, int *hist, int *shI, uint32_t *lstat, int drain_n, int lane)
{
    if (shI[GI_NSTAGE] > 0) grp_w0_flush_stage(cs, stage, shI, lstat, lane, nullptr);
    Cold cold = *cs;
    if (!cold.failed && cold.count > 0) {
        bool sorted = true;
        const int got = cold_drain(cold, srun, hist, lane, drain_n, &sorted);
        if (lane == 0) { shI[GI_SCOUNT] = got; shI[GI_SHEAD] = 0; shI[GI_SORTN] = sorted ? 0 : got; atomicAdd(&lstat[GS_DRAIN], 1u); if (!sorted) atomicAdd(&lstat[GS_WINEV + 0], 0u); }
    }
    grp_publish_cold(cold, shI, lane);
    *cs = cold;
}

// seeds beyond the hot tiers start cold
__device__ __noinline__ void grp_w0_seed_append(Cold *cs, const unsigned long long *keys, uint32_t first, uint32_t from, uint32_t ns,
                                                const uint32_t *out, int *shI, int lane)
{
    Cold cold = *cs;
    const unsigned long long ZKEY = (unsigned long long)0x80000000u << 32;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t j0 = from; j0 < ns; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool hs = j < ns;
        const uint32_t vi = hs ? (uint32_t)(keys[first + j] & 0xFFFFFFFFull) : 0u;
        cold_append(cold, hs, ZKEY | (unsigned long long)j, vi, hs ? out[vi] : 0u, lane);
        if (cold.failed) { if (lane == 0) shI[GI_SERIAL] = 2; break; }
    }
    grp_publish_cold(cold, shI, lane);
    *cs = cold;
}

// serial continuation: every bucket's elements into the heap
__device__ __noinline__ uint32_t grp_w0_collect_cold(Cold *cs, HeapEnt *heap, uint32_t hn, int lane)
{
    Cold cold = *cs;
    if (cold.on) {
        while (cold.count > 0) {
            const uint32_t b = cold_lowest(cold, lane);
            uint32_t seg = cld32(&cold.bhead()[b]);
            uint32_t off = cld32(&cold.bmeta()[b]) & ~COLD_UNSORTED;
            uint32_t got = 0;
            while (seg != COLD_NONE) {
                const HeapEnt h = cld_ent(&cold.pool()[seg]);
                const uint32_t n = (uint32_t)(h.key >> 32);
                if (lane == 0)
                    for (uint32_t j = off; j < n; ++j) { HeapEnt e = cld_ent(&cold.pool()[seg + 1u + j]); e.pad = 0; heap_push(heap, hn, e); }
                got += n - off;
                off = 0;
                seg = (uint32_t)h.key;
            }
            cold_mark_empty(cold, b, lane);
            cold.count -= got;
        }
    }
    *cs = cold;
    return hn;
}

__global__ __launch_bounds__(GRP_W) void flood_group_kernel(
    const unsigned long long *__restrict__ order,
    const float *__restrict__ aff, const uint8_t *__restrict__ mask, uint32_t *__restrict__ out,
    unsigned long long *__restrict__ state, const unsigned long long *__restrict__ keys, uint32_t n_seeds,
    const uint32_t *__restrict__ comp_first, const unsigned long long *__restrict__ comp_off, uint32_t n_comp,
    HeapEnt *__restrict__ heap_pool, const unsigned long long *__restrict__ arena_off, HeapEnt *__restrict__ arena,
    int64_t PY, int64_t PX, int64_t PN, float s0, float s1, float s2, unsigned long long group_min, int run_limit,
    int drain_n, int wcap, int tcap, int max_bursts, float bk_lo, float bk_inv, unsigned long long *stats, int shard_rank, int shard_world, int kmax)
{
    __shared__ HeapEnt srun[COOP_SCAP];
    __shared__ HeapEnt stage[GRP_STAGE];
    __shared__ HeapEnt pend[GRP_PEND];
    __shared__ HeapEnt hotb[GRP_W];
    __shared__ uint32_t pvox[GRP_NCH * GRP_CH], pval[GRP_NCH * GRP_CH];
    __shared__ unsigned short freech[GRP_NCH];
    __shared__ unsigned short chtab[GRP_W][GRP_T / GRP_CH];             // chunk c of the list of slot s (entry i lives in chunk i / 8)
    __shared__ unsigned short resch[GRP_T / GRP_CH + 2];                 // the front episode's reserve (one full list)
    __shared__ unsigned long long okey[GRP_W], succ[GRP_W];
    __shared__ unsigned short oslot[GRP_W], freeslot[GRP_W];
    __shared__ uint32_t pub_res[GRP_W], pub_mx[GRP_W], pub_info[GRP_W];   // info: np | npop << 8 | started << 16 | idle << 17
    __shared__ uint32_t ci_age[GRP_W], ci_off[GRP_W];                    // commit orders of a burst (ci_off: GRP_NONE = not committing)
    __shared__ u32x4 newroot[GRP_W];                                     // {key.lo, key.hi, voxel, label}
    __shared__ uint32_t newflag[GRP_W];
    __shared__ int wtmp[4][GRP_NWV];
    __shared__ int hist[64];
    __shared__ int shI[GI_N];
    __shared__ unsigned long long shQ[GQ_N];
    __shared__ uint32_t lstat[GS_N];
    __shared__ Cold scold;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x >= n_comp) return;
    const unsigned long long okey_comp = order[blockIdx.x];
    if (okey_comp == ~0ull) return;
    const uint32_t c = (uint32_t)okey_comp;
    if (!comp_owned(comp_off, c, n_comp, shard_rank, shard_world)) return;
    const uint32_t first = comp_first[c];
    const uint32_t end = (c + 1 < n_comp) ? comp_first[c + 1] : n_seeds;
    if ((uint32_t)(keys[first] >> 32) == NO_PARENT) return;
    const unsigned long long csize = comp_off[c + 1] - comp_off[c];
    if (csize < group_min) return;
    HeapEnt *heap = heap_pool + comp_off[c];
    const uint32_t ns = end - first;

    const int noff32[6] = {-(int)(PY * PX), -(int)PX, -1, 1, (int)PX, (int)(PY * PX)};
    const int ax[6] = {0, 1, 2, 2, 1, 0};
    const float sc[3] = {s0, s1, s2};
    const unsigned long long ZKEY = (unsigned long long)0x80000000u << 32;
    constexpr int SMASK = COOP_SCAP - 1;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    if (wcap > GRP_W) wcap = GRP_W;
    if (tcap > GRP_T) tcap = GRP_T;
    const int reserve = (tcap + GRP_CH - 1) / GRP_CH + 1;

    // ---- cold tier: owned by wave 0 (its lanes hold `cold`); count / low are published in shI for the others ---------------
    Cold cold;
    cold.on = arena_off && (arena_off[c + 1] > arena_off[c]);
    cold.failed = false; cold.bump = 0; cold.count = 0; cold.low = COLD_NBK; cold.fifo_drains = 0;
#ifdef ISX_FLOOD_PROF
    for (int i = 0; i < 8; ++i) cold.tk[i] = 0;
    cold.n_groups = 0;
#endif
    cold.cb = COLD_NONE; cold.pool_cap = 0; cold.vlo = bk_lo; cold.vinv = bk_inv; cold.base = nullptr;
    if (cold.on) {
        HeapEnt *base = arena + arena_off[c];
        cold.base = (uint32_t *)base;
        cold.pool_cap = (uint32_t)(arena_off[c + 1] - arena_off[c] - COLD_META_ENTS);
        for (int i = tid; i < COLD_NBK; i += GRP_W) { cold.bhead()[i] = COLD_NONE; cold.btail()[i] = COLD_NONE; cold.bcount()[i] = 0; cold.bmeta()[i] = 0; }
        for (int i = tid; i < COLD_NBK / 64; i += GRP_W) cold.bm0()[i] = 0ull;
        if (tid < COLD_NBK / 4096) cold.bm1()[tid] = 0ull;
    }
    if (tid == 0) scold = cold;
    const bool cold_on = cold.on;
    const int kmin = ((kmax >> 8) & 0xFF) > 1 ? ((kmax >> 8) & 0xFF) : 1;   // steps a wave makes before it looks at the front's "ready"
    for (int i = tid; i < GI_N; i += GRP_W) shI[i] = i == GI_COLD_LOW ? COLD_NBK : 0;
    if (tid < GQ_N) shQ[tid] = ~0ull;
    for (int i = tid; i < GS_N; i += GRP_W) lstat[i] = 0u;
    for (int i = tid; i < GRP_NCH; i += GRP_W) freech[i] = (unsigned short)(GRP_NCH - 1 - i);
    freeslot[tid] = (unsigned short)(GRP_W - 1 - tid);
    newflag[tid] = 0u;
    ci_off[tid] = GRP_NONE;
    __syncthreads();

    // ---- episode of this lane (slot = tid) -----------------------------------------------------------------------------------------
    bool has = false, started = false;
    unsigned long long key = 0ull;
    uint32_t root = 0, lab = 0, cur = GRP_NONE, curv = 0;
    int cur_addr = -1, cur_li = -1, np = 0, npop = 0;
    uint32_t mx = 0, resmin = 0xFFFFFFFFu;
    int ncand = 0, ndef = 0;
    // register candidates: value << 32 | discovery index << 16 | pool address -- one 64-bit compare orders them like the
    // reference orders its heap entries of one episode (value, then push order).  Kept SORTED ascending, free registers hold
    // ~0 (sorts last): the next pop is ck[0], an insertion is one compare per register and a shift by selects, the largest
    // candidate is `cmax` (= ck[ncand - 1]; 0 when there is none).  (Unsorted registers with a minimum / maximum search per
    // pop / discovery cost 205 + 6 x 71 instructions of the ~850 a step executes: the step loop is instruction-bound, LOG.md.)
    unsigned long long ck[GRP_CMAX], cmax = 0ull;
    int chlast = -1, spare = -1;
    auto cand_clear = [&]() {
#pragma unroll
        for (int k = 0; k < GRP_CMAX; ++k) ck[k] = ~0ull;
        ncand = 0; ndef = 0; cmax = 0ull;
    };
    cand_clear();

    auto ep_fresh = [&]() {
        started = false; cur = root; curv = 0; cur_addr = -1; cur_li = -1; np = 0; npop = 0; mx = 0; resmin = 0xFFFFFFFFu;
        cand_clear(); chlast = -1; spare = -1;
    };
    auto free_chunks = [&]() {                       // pushes only: never in a phase that allocates (uses np: call before ep_fresh)
        const int nc = (np + GRP_CH - 1) / GRP_CH;
        for (int c2 = 0; c2 < nc; ++c2) freech[atomicAdd(&shI[GI_NFREECH], 1)] = chtab[tid][c2];
        if (spare >= 0) freech[atomicAdd(&shI[GI_NFREECH], 1)] = (unsigned short)spare;
        chlast = spare = -1;
    };
    auto squash_me = [&]() {
        if (started) {
            atomicAdd(&lstat[GS_SQ_EPS], 1u);
            int ch = 0;
            for (int i = 0; i < np; ++i) {
                if ((i & (GRP_CH - 1)) == 0) ch = (int)chtab[tid][i >> 3];
                const uint32_t vq = pvox[ch * GRP_CH + (i & (GRP_CH - 1))];
                if (vq < (uint32_t)PN) atomicCAS(&state[vq], key, ST_FREE);
                else if (vq != GRP_NONE && stats) { atomicAdd(&stats[20], 1ull); stats[21] = 1ull; }     // (GRP_NONE: dropped at a verification)
            }
        }
        free_chunks();
        ep_fresh();
        pub_info[tid] = 0u; pub_res[tid] = 0xFFFFFFFFu; pub_mx[tid] = 0u;      // "not started" is what the commit phase must see
    };
    // every deferred entry (below the bound, unpopped, only in the list) sorts after every register candidate: a discovery enters
    // the registers if it sorts below the largest candidate, or if a register is free and nothing is deferred; what falls off
    // the end of a full set becomes a deferred entry.  (Selects only: with an early-return form the compiler merged `++ncand` /
    // `++ndef` into one store through a selected pointer and kept both counters in scratch memory.)
    auto cand_add = [&](uint32_t v, int li, int ad) {
        const unsigned long long nk_ = ((unsigned long long)v << 32) | ((unsigned long long)(uint32_t)li << 16) | (unsigned long long)(uint32_t)ad;
        const bool full = ncand == GRP_CMAX;
        const bool ins = (ncand > 0 && nk_ < cmax) || (!full && ndef == 0);
        bool lt[GRP_CMAX];
#pragma unroll
        for (int k = 0; k < GRP_CMAX; ++k) lt[k] = ins && nk_ < ck[k];
#pragma unroll
        for (int k = GRP_CMAX - 1; k >= 0; --k) {
            const unsigned long long below = k > 0 ? ck[k - 1] : 0ull;
            const bool shift = k > 0 && lt[k - 1];                 // the new key sorts before ck[k - 1]: that one moves up
            ck[k] = shift ? below : (lt[k] ? nk_ : ck[k]);
        }
        const bool grew = ins && !full;
        cmax = ins ? (full ? ck[GRP_CMAX - 1] : (nk_ > cmax ? nk_ : cmax)) : cmax;
        ncand += grew ? 1 : 0;
        ndef += grew ? 0 : 1;
    };
    // The candidates of lane L's episode rebuilt from its list by the WHOLE wave (three entries per lane): candidates,
    // deferred count and the minimum of the residuals.  Runs when the six registers are used up and discoveries below
    // the bound remain in the list only -- a long descent does that every few pops, and one lane walking 190 entries
    // (with every other lane of the work-group waiting at the next barrier) was the single most expensive step.
    auto coop_reload = [&](int L, uint32_t bnd) {
        // guard site 4 (always on): the wave reductions below are DPP scans and need every lane -- a partial EXEC mask here means the
        // wave has split somewhere above (round 4), and the candidates rebuilt from such scans are garbage: report, rebuild nothing
        if (__builtin_amdgcn_read_exec() != ~0ull) {
            if (stats && atomicAdd(&stats[20], 1ull) == 0ull) stats[21] = 4ull;
            return;
        }
        const int slot = (wave << 6) + L;
        const int npL = (int)rdlane((uint32_t)np, L);
        unsigned long long k3[3];
        uint32_t rm = 0xFFFFFFFFu, cnt = 0;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = lane + 64 * u;
            k3[u] = ~0ull;
            if (i < npL) {
                const int a = (int)chtab[slot][i >> 3] * GRP_CH + (i & (GRP_CH - 1));
                const uint32_t v = pval[a];
                if (v != WIN_POPPED) {
                    if (v < bnd) { k3[u] = ((unsigned long long)v << 32) | ((unsigned long long)(uint32_t)i << 16) | (unsigned long long)(uint32_t)a; ++cnt; }
                    else if (v < rm) rm = v;
                }
            }
        }
        const uint32_t total = rdlane(wave_incl_sum(cnt), 63);
        const uint32_t rmin = rdlane(wave_prefix_min(rm), 63);
#pragma unroll
        for (int r = 0; r < GRP_CMAX; ++r) {
            unsigned long long m = k3[0] < k3[1] ? k3[0] : k3[1];
            m = k3[2] < m ? k3[2] : m;
            const uint32_t vmin = rdlane(wave_prefix_min((uint32_t)(m >> 32)), 63);
            const uint32_t lmin = rdlane(wave_prefix_min((uint32_t)(m >> 32) == vmin ? (uint32_t)m : 0xFFFFFFFFu), 63);
            const unsigned long long best = ((unsigned long long)vmin << 32) | lmin;
            if (lane == L) { ck[r] = vmin != 0xFFFFFFFFu ? best : ~0ull; if (r == 0) cmax = 0ull; if (vmin != 0xFFFFFFFFu) cmax = best; }      // ascending
            if (vmin != 0xFFFFFFFFu) {
#pragma unroll
                for (int u = 0; u < 3; ++u) if (k3[u] == best) k3[u] = ~0ull;
            }
        }
        if (lane == L) { ncand = (int)(total < (uint32_t)GRP_CMAX ? total : (uint32_t)GRP_CMAX); ndef = (int)total - ncand; resmin = rmin; }
    };
    // next pop: the smallest unpopped discovery if it sorts below the bound (never called with ncand == 0 && ndef > 0)
    auto select_next = [&](uint32_t bound) {
        const unsigned long long mv = ck[0];                        // the smallest candidate (~0: none)
        if (ncand > 0 && (uint32_t)(mv >> 32) < bound) {
            curv = (uint32_t)(mv >> 32); cur_li = (int)((mv >> 16) & 0xFFFFull); cur_addr = (int)(mv & 0xFFFFull);
            cur = pvox[cur_addr];
#pragma unroll
            for (int k = 0; k + 1 < GRP_CMAX; ++k) ck[k] = ck[k + 1];
            ck[GRP_CMAX - 1] = ~0ull;
            --ncand;
            if (ncand == 0) cmax = 0ull;
            return;
        }
        // idle: what is left in the registers sorts at or above the bound, and every deferred entry sorts after every
        // register candidate -- the smallest candidate is the smallest of all of them
        if (ncand > 0 && (uint32_t)(mv >> 32) < resmin) resmin = (uint32_t)(mv >> 32);
        cand_clear();
        cur = GRP_NONE;
    };
    auto adopt = [&]() {
        if (newflag[tid]) {
            const u32x4 r = newroot[tid];
            newflag[tid] = 0u;
            has = true; key = ((unsigned long long)r.y << 32) | r.x; root = r.z; lab = r.w;
            ep_fresh();
            // what the slot's previous episode published is void: a later burst of this sub-round must see "not started"
            pub_info[tid] = 0u; pub_res[tid] = 0xFFFFFFFFu; pub_mx[tid] = 0u;
        }
    };
    uint32_t st_sub = 0;
    auto ord_at = [&](int p) -> int { return (shI[GI_OH] + p) & (GRP_W - 1); };
    // lane 0, end of a sub-round: what every lane needs at the top of the next one (front / last slot, head of the run)
    auto publish_heads = [&]() {
        const int nwp = shI[GI_NW];
        if (nwp > 0) { shI[GI_FRONT] = (int)oslot[ord_at(0)]; shI[GI_LAST] = (int)oslot[ord_at(nwp - 1)]; }
        shQ[GQ_QHEAD] = shI[GI_SCOUNT] > 0 ? srun[shI[GI_SHEAD]].key : ~0ull;
    };
    // development (ISX_FLOOD_BURSTS bit 0x400): window sorted, window <= run head, run sorted -- the first violation is
    // reported with its tag and the component stops (all lanes call it, between barriers)
    auto check_order = [&](int tag) {
        if (!(max_bursts & 0x400)) return;
        __syncthreads();
        const int nwc = shI[GI_NW], scc = shI[GI_SCOUNT], shc = shI[GI_SHEAD];
        bool bad = false; unsigned long long a_ = 0, b_ = 0; int where = 0;
        if (tid < nwc) {
            a_ = okey[ord_at(tid)];
            b_ = tid + 1 < nwc ? okey[ord_at(tid + 1)] : (scc > 0 ? srun[shc].key : ~0ull);
            if (!(a_ < b_)) { bad = true; where = tid + 1 < nwc ? 1 : 2; }
        }
        for (int p0 = tid; p0 + 1 < scc && !bad; p0 += GRP_W) {
            a_ = srun[(shc + p0) & SMASK].key; b_ = srun[(shc + p0 + 1) & SMASK].key;
            if (!(a_ < b_)) { bad = true; where = 3; }
        }
        if (bad && stats && atomicAdd(&stats[20], 1ull) == 0ull) {
            stats[21] = 300ull + (unsigned long long)tag; stats[10] = (unsigned long long)st_sub | ((unsigned long long)where << 32) | ((unsigned long long)tid << 40);
            stats[40] = a_; stats[41] = b_; stats[42] = (unsigned long long)nwc | ((unsigned long long)scc << 32);
            shI[GI_SERIAL] = 1;
        }
        __syncthreads();
    };

    // ---- wave-0 helpers on the queue tiers (the other waves wait at the next barrier) -----------------------------------------------
    auto publish_cold = [&]() { };                   // (the out-of-line routines publish count / low themselves)
    auto flush_stage_w0 = [&]() { grp_w0_flush_stage(&scold, stage, shI, lstat, lane, hotb); };
    auto run_evict_w0 = [&](int keep) { grp_w0_run_evict(&scold, srun, shI, lstat, stats, keep, lane); };

    // ---- start: seeds (value 0, age = rank in index order) ---------------------------------------------------------------------------------
    {
        const uint32_t in_win = ns < (uint32_t)wcap ? ns : (uint32_t)wcap;
        if ((uint32_t)tid < in_win) {
            const uint32_t vi = (uint32_t)(keys[first + tid] & 0xFFFFFFFFull);
            const unsigned long long k = ZKEY | (unsigned long long)tid;
            okey[tid] = k; oslot[tid] = (unsigned short)tid;
            succ[tid] = (uint32_t)tid + 1u < in_win ? (ZKEY | (unsigned long long)(tid + 1)) : ~0ull;
            has = true; key = k; root = vi; lab = out[vi];
            ep_fresh();
        }
        // slots [0, in_win) are taken: the free list holds the others (top of the stack = lowest free slot)
        if (tid == 0) {
            shI[GI_NW] = (int)in_win; shI[GI_NFREESLOT] = GRP_W - (int)in_win; shI[GI_AGEBASE] = (int)(ns - 1u);
            int nf = GRP_NCH;
            for (int i = 0; i < reserve; ++i) resch[i] = freech[--nf];
            shI[GI_NRES] = reserve; shI[GI_NFREECH] = nf;
        }
        const uint32_t hot_cap = in_win + (uint32_t)run_limit;
        const uint32_t in_hot = ns > hot_cap ? hot_cap : ns;
        for (uint32_t j = in_win + (uint32_t)tid; j < in_hot; j += GRP_W) {
            HeapEnt e;
            e.key = ZKEY | (unsigned long long)j;
            e.idx = (uint32_t)(keys[first + j] & 0xFFFFFFFFull);
            e.pad = out[e.idx];
            srun[j - in_win] = e;
        }
        if (tid == 0) shI[GI_SCOUNT] = in_hot > in_win ? (int)(in_hot - in_win) : 0;
        __syncthreads();
        if (ns > in_hot) {
            if (!cold_on) { if (tid == 0) shI[GI_SERIAL] = 2; }            // 2: restart from the seeds on the serial heap
            else if (wave == 0) grp_w0_seed_append(&scold, keys, first, in_hot, ns, out, shI, lane);
        }
        __syncthreads();
        if (tid == 0) publish_heads();
    }

#ifdef ISX_FLOOD_PROF
    unsigned long long tk_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk_prev = __builtin_readcyclecounter();
#define GTICK(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); tk_acc[i] += t_ - tk_prev; tk_prev = t_; } while (0)
#else
#define GTICK(i) do { } while (0)
#endif
    // ---- enqueue of a committed batch: pend[0, GI_PENDN) in age order -> cold stage / window / run (a lambda: the commit bursts
    //      and the plateau round share it).  Returns false when the component has gone to the serial continuation. ---------------
    auto enqueue_batch = [&]() -> bool {
        const int Ptot = shI[GI_PENDN];
        for (int p0 = 0; p0 < Ptot; p0 += GRP_W) {
            const int P = Ptot - p0 < GRP_W ? Ptot - p0 : GRP_W;
            // capacity first: an eviction moves the hot maximum, so the batch is judged against the tiers AFTER it
            if (shI[GI_SCOUNT] + P > run_limit || shI[GI_SCOUNT] + P > COOP_SCAP - 8) {
                if (wave == 0) { int keep = run_limit / 2; if (keep > COOP_SCAP - GRP_W - 8) keep = COOP_SCAP - GRP_W - 8; run_evict_w0(keep); publish_cold(); }
                __syncthreads();
                if (grp_flag(shI, GI_SERIAL)) break;                                  // `pend` keeps the batch for the serial continuation
            }
            const int nw2 = shI[GI_NW], sc2 = shI[GI_SCOUNT], sh2 = shI[GI_SHEAD];
            const bool hot_ne = nw2 > 0 || sc2 > 0;
            const unsigned long long hot_max = sc2 > 0 ? srun[(sh2 + sc2 - 1) & SMASK].key : (nw2 > 0 ? okey[ord_at(nw2 - 1)] : 0ull);
            const unsigned long long lastwin = nw2 > 0 ? okey[ord_at(nw2 - 1)] : 0ull;
            const bool cold_ne = cold_on && (shI[GI_COLD_COUNT] > 0 || shI[GI_NSTAGE] > 0);
            HeapEnt r; r.key = 0; r.idx = 0; r.pad = 0;
            bool hasr = tid < P;
            if (hasr) r = pend[p0 + tid];
            hasr = hasr && r.key != ~0ull;                                  // (a slot of a dropped residual)
            const bool tocold = hasr && cold_ne && !(hot_ne && r.key < hot_max) && cold_bucket(cold, r.key) >= (uint32_t)shI[GI_COLD_LOW];
            const bool towin = hasr && !tocold && nw2 > 0 && r.key < lastwin;
            const bool torun = hasr && !tocold && !towin;
            // three compactions (order inside a class is the age order of `pend`)
            const unsigned long long bc = __ballot(tocold), bw = __ballot(towin), br = __ballot(torun);
            if (lane == 0) { wtmp[0][wave] = __popcll(bc); wtmp[1][wave] = __popcll(bw); wtmp[2][wave] = __popcll(br); }
            __syncthreads();
            int oc = shI[GI_NSTAGE], ow = 0, orr = 0;
#pragma unroll
            for (int w = 0; w < GRP_NWV; ++w) if (w < wave) { oc += wtmp[0][w]; ow += wtmp[1][w]; orr += wtmp[2][w]; }
            if (tocold) stage[oc + __popcll(bc & lt_mask)] = r;
            // window-bound ones go back into `pend` (every lane has its element in a register by now), run-bound ones into
            // `hotb`, which also takes the window-bound ones that turn out run-bound below: together at most P
            if (towin) pend[ow + __popcll(bw & lt_mask)] = r;
            if (torun) hotb[orr + __popcll(br & lt_mask)] = r;
            __syncthreads();
            if (tid == 0) {
                int tc = 0, tw = 0, tr = 0;
                for (int w = 0; w < GRP_NWV; ++w) { tc += wtmp[0][w]; tw += wtmp[1][w]; tr += wtmp[2][w]; }
                shI[GI_NSTAGE] += tc; shI[GI_NWB] = tw; shI[GI_NHOT] = tr;
                if (p0 + P >= Ptot) { shI[GI_PENDN] = 0; shI[GI_PEND0] = 0; } else shI[GI_PEND0] = p0 + P;
            }
            __syncthreads();
            // Window-bound residuals (keys below the last root) become roots at their sorted positions -- ALL of them in one
            // ranked merge with the window: an old root moves up by the number of new keys below it, a new one lands at
            // (old roots below it) + (its rank among the new ones); whatever ends beyond the window's capacity is the top
            // of the merged order and goes back to the FRONT of the run, in order (a root evicted this way is squashed by
            // its own lane).  One element at a time this was ~5 barriers per residual: 250 of them per burst on plateaus.
            const int T = shI[GI_NWB];
            if (T > 0) {                                                   // uniform
                const int nw3 = shI[GI_NW];
                int E = nw3 + T - wcap;
                if (E < 0) E = 0;
                if (E > 0 && shI[GI_SCOUNT] + E > COOP_SCAP - 2) {          // room at the front of the run (rare)
                    __syncthreads();
                    if (wave == 0) { run_evict_w0(run_limit / 2); publish_cold(); }
                    __syncthreads();
                    if (grp_flag(shI, GI_SERIAL)) break;                              // pend[0, NWB) are still to be queued (WB0 = 0)
                }
                const int sh3w = shI[GI_SHEAD];
                unsigned long long ok_ = 0ull; int os_ = 0, np_old = -1;
                if (tid < nw3) {
                    ok_ = okey[ord_at(tid)]; os_ = (int)oslot[ord_at(tid)];
                    int c2 = 0;
                    for (int q = 0; q < T; ++q) if (pend[q].key < ok_) ++c2;
                    np_old = tid + c2;
                }
                HeapEnt ne; ne.key = 0; ne.idx = 0; ne.pad = 0;
                int np_new = -1, rk_new = 0;
                if (tid < T) {
                    ne = pend[tid];
                    for (int q = 0; q < T; ++q) if (pend[q].key < ne.key) ++rk_new;
                    int lo = 0, hi = nw3;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (okey[ord_at(mid)] < ne.key) lo = mid + 1; else hi = mid; }
                    np_new = lo + rk_new;
                }
                {
                    const unsigned long long kb = __ballot(np_new >= 0 && np_new < wcap);
                    if (lane == 0) wtmp[0][wave] = __popcll(kb);
                }
                __syncthreads();                                            // (W1) everything read
                // evictions: an old root beyond the capacity tells its owner where in the run it goes; a new element goes itself
                if (np_old >= wcap) ci_off[os_] = (uint32_t)((sh3w - E + (np_old - wcap)) & SMASK);
                if (np_new >= wcap) srun[(sh3w - E + (np_new - wcap)) & SMASK] = ne;
                __syncthreads();                                            // (W2)
                if (has && ci_off[tid] != GRP_NONE) {
                    HeapEnt b; b.key = key; b.idx = root; b.pad = lab;
                    srun[ci_off[tid]] = b;
                    ci_off[tid] = GRP_NONE;
                    squash_me();
                    has = false;
                    freeslot[atomicAdd(&shI[GI_NFREESLOT], 1)] = (unsigned short)tid;
                    atomicAdd(&lstat[GS_WINEV], 1u);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();                                            // (W3) slots of the evicted are back
                const int nf_mid = shI[GI_NFREESLOT];
                if (np_old >= 0 && np_old < wcap) { okey[ord_at(np_old)] = ok_; oslot[ord_at(np_old)] = (unsigned short)os_; }
                if (np_new >= 0 && np_new < wcap) {
                    const int sl = (int)freeslot[nf_mid - 1 - rk_new];     // the kept new ones are a prefix of the new ones in key order
                    okey[ord_at(np_new)] = ne.key; oslot[ord_at(np_new)] = (unsigned short)sl;
                    u32x4 nr = {(uint32_t)ne.key, (uint32_t)(ne.key >> 32), ne.idx, ne.pad};
                    newroot[sl] = nr; newflag[sl] = 1u;
                }
                __syncthreads();                                            // (W4) the new order is written
                int tin = 0;
#pragma unroll
                for (int w = 0; w < GRP_NWV; ++w) tin += wtmp[0][w];
                const int nw_new = nw3 + T - E;
                if (tid < nw_new) succ[oslot[ord_at(tid)]] = tid + 1 < nw_new ? okey[ord_at(tid + 1)] : ~0ull;
                if (tid == 0) {
                    shI[GI_NW] = nw_new; shI[GI_NFREESLOT] = nf_mid - tin; shI[GI_DIRTY] = 1;
                    shI[GI_SHEAD] = (sh3w - E) & SMASK; shI[GI_SCOUNT] += E;
                    atomicAdd(&lstat[GS_WININS], (uint32_t)tin);
                }
                __syncthreads();                                            // (W5)
                adopt();
            }
            if (tid == 0) { shI[GI_NWB] = 0; shI[GI_WB0] = 0; }            // all of them are roots (or run-bound) now
            // cold-bound ones: wave 0 appends them; the other waves merge the run-bound ones meanwhile
            const int R = shI[GI_NHOT];
            const int sc3 = shI[GI_SCOUNT], sh3 = shI[GI_SHEAD];
            // a new element: rank among the new ones + number of run elements below it (against the run as it stands)
            int mypos = -1; HeapEnt me; me.key = 0; me.idx = 0; me.pad = 0;
            if (tid < R) {
                me = hotb[tid];
                int rk = 0;
                for (int q = 0; q < R; ++q) if (hotb[q].key < me.key) ++rk;
                int lo = 0, hi = sc3;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (srun[(sh3 + mid) & SMASK].key < me.key) lo = mid + 1; else hi = mid; }
                mypos = lo + rk;
            }
            // the run's own entries move up by the number of new keys below them, from the top of the run down, one entry
            // per lane and round: an entry lands at or above its place, so a round never overwrites what a later round reads
            if (R > 0 && sc3 > 0) {
                for (int base = ((sc3 - 1) / GRP_W) * GRP_W; base >= 0; base -= GRP_W) {
                    const int pp = base + tid;
                    const bool mv = pp < sc3;
                    u32x4 e = {0u, 0u, 0u, 0u};
                    int np2 = pp;
                    if (mv) {
                        e = run_ld(srun, (sh3 + pp) & SMASK);
                        const unsigned long long k0 = ((unsigned long long)e.y << 32) | e.x;
                        for (int q = 0; q < R; ++q) if (hotb[q].key < k0) ++np2;
                    }
                    __syncthreads();                                        // this round's entries (and, first round, the searches) are read
                    if (mv && np2 != pp) run_st(srun, (sh3 + np2) & SMASK, e);
                }
            } else __syncthreads();
            if (mypos >= 0) srun[(sh3 + mypos) & SMASK] = me;
            GTICK(6);
            if (wave == 0 && shI[GI_NSTAGE] >= 64) { flush_stage_w0(); publish_cold(); }
            GTICK(15);
            if (tid == 0 && R > 0) { shI[GI_SCOUNT] = sc3 + R; shI[GI_NHOT] = 0; shI[GI_DIRTY] = 1; atomicAdd(&lstat[GS_RUNINS], (uint32_t)R); }
            __syncthreads();
            if (grp_flag(shI, GI_SERIAL)) break;
        }
        return !grp_flag(shI, GI_SERIAL);
    };

    // ================================================= sub-rounds ==================================================================
    // watchdog: sub-rounds (never expected to be reached; ISX_FLOOD_GWATCHDOG sets a tiny limit to test the hand-over it leads to)
    const uint32_t sub_limit = (max_bursts >> 16) ? (uint32_t)(max_bursts >> 16) : (csize > 60000000ull ? 0xFFFFFFFFu : 4096u + 64u * (uint32_t)csize);
    for (;;) {
        __syncthreads();                                                        // (S0) the previous sub-round is complete
        GTICK(8);
        if (grp_flag(shI, GI_SERIAL) || shI[GI_NW] == 0) break;
        if (++st_sub > sub_limit) {                                             // watchdog (never expected): finish on the serial heap
            if (tid < 8) wtmp[tid >> 2][tid & 3] = 0;
            __syncthreads();
            if (has) {
                atomicAdd(&wtmp[0][0], 1);
                if (started) atomicAdd(&wtmp[0][1], 1);
                if (started && cur == GRP_NONE) atomicAdd(&wtmp[0][2], 1);
                if (spare >= 0) atomicAdd(&wtmp[0][3], 1);
                atomicAdd(&wtmp[1][0], (np + GRP_CH - 1) / GRP_CH);
                atomicAdd(&wtmp[1][1], np);
            }
            __syncthreads();
            if (tid == 0 && stats && !(max_bursts >> 16)) {
                const int fs_ = (int)oslot[ord_at(0)];
                atomicAdd(&stats[9], 1ull);
                stats[24] = (unsigned long long)shI[GI_NW]; stats[25] = (unsigned long long)shI[GI_SCOUNT];
                stats[26] = (unsigned long long)(uint32_t)shI[GI_NFREECH] | ((unsigned long long)(uint32_t)shI[GI_NRES] << 16) | ((unsigned long long)(uint32_t)shI[GI_NFREESLOT] << 32) | ((unsigned long long)(uint32_t)shI[GI_DIRTY] << 48) | ((unsigned long long)(uint32_t)grp_flag(shI, GI_FORCED) << 56);
                stats[27] = (unsigned long long)pub_info[fs_] | ((unsigned long long)pub_mx[fs_] << 32);
                stats[28] = okey[ord_at(0)];
                stats[29] = shI[GI_SCOUNT] > 0 ? srun[shI[GI_SHEAD]].key : ~0ull;
                stats[30] = (unsigned long long)(uint32_t)wtmp[0][0] | ((unsigned long long)(uint32_t)wtmp[0][1] << 16) | ((unsigned long long)(uint32_t)wtmp[0][2] << 32) | ((unsigned long long)(uint32_t)wtmp[0][3] << 48);
                stats[31] = (unsigned long long)lstat[GS_POPS] | ((unsigned long long)(uint32_t)wtmp[1][0] << 32) | ((unsigned long long)(uint32_t)wtmp[1][1] << 48);
                for (int q = 0; q < 4; ++q) {
                    stats[40 + q] = q < shI[GI_NW] ? okey[ord_at(q)] : 0ull;
                    stats[44 + q] = q < shI[GI_SCOUNT] ? srun[(shI[GI_SHEAD] + q) & SMASK].key : 0ull;
                }
                stats[10] = (unsigned long long)pub_res[fs_] | ((unsigned long long)fs_ << 32) | ((unsigned long long)shI[GI_COLD_COUNT] << 44);
            }
            __syncthreads();
            if (tid == 0) shI[GI_SERIAL] = 1;
            continue;
        }
        adopt();
        if (shI[GI_SCOUNT] > run_limit) {                                       // uniform: the run outgrew its limit (tiny stress limits)
            __syncthreads();
            if (wave == 0) { run_evict_w0(run_limit / 2); publish_cold(); }
            continue;
        }
        const int nw = shI[GI_NW];
        // invariant of the bounds: the run is empty only if nothing at all is queued behind the window
        if (tid == 0 && stats && shI[GI_SCOUNT] == 0 && cold_on && (shI[GI_COLD_COUNT] > 0 || shI[GI_NSTAGE] > 0)) {
            if (atomicAdd(&stats[20], 1ull) == 0ull) { stats[21] = 200ull; stats[10] = (unsigned long long)st_sub | ((unsigned long long)shI[GI_COLD_COUNT] << 32); }
        }
        bool did_commit = false;
        {
        const unsigned long long qhead = shQ[GQ_QHEAD];                         // published by lane 0 at the end of the last sub-round
        const int lastslot = shI[GI_LAST], frontslot = shI[GI_FRONT];
        const bool dirty = shI[GI_DIRTY] != 0;
        // ---- A: an episode that popped (or holds) something at or above its bound is wrong; a bound only drops after an
        //         insertion right behind the episode, so the check runs after insertions only -------------------------------------------
        unsigned long long nk = ~0ull;
        if (has) nk = tid == lastslot ? qhead : succ[tid];
        uint32_t bound = (uint32_t)(nk >> 32);
        if (dirty) {
            if (has && started) {
                uint32_t m2 = mx;
                if (cur != GRP_NONE && cur_li >= 0 && curv > m2) m2 = curv;
                if (m2 != 0u && m2 >= bound) atomicMin(&shQ[GQ_KSQ], key);
                // candidates that the lower bound turned into residuals: a suffix of the sorted registers
#pragma unroll
                for (int k = GRP_CMAX - 1; k >= 0; --k) {
                    if (k < ncand && (uint32_t)(ck[k] >> 32) >= bound) {
                        if ((uint32_t)(ck[k] >> 32) < resmin) resmin = (uint32_t)(ck[k] >> 32);
                        ck[k] = ~0ull;
                        --ncand;
                    }
                }
                cmax = 0ull;
#pragma unroll
                for (int k = 0; k < GRP_CMAX; ++k) if (k < ncand) cmax = ck[k];
            }
            __syncthreads();                                                    // (S1)
            const unsigned long long ksq = shQ[GQ_KSQ];
            if (ksq != ~0ull) {
                if (has && key >= ksq) squash_me();
                if (tid == 0) atomicAdd(&lstat[GS_SQ_BOUND], 1u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();                                                    // (S2) releases performed, chunks back
            if (tid == 0) { shQ[GQ_KSQ] = ~0ull; shI[GI_DIRTY] = 0; }
        }
        // ---- C: free-running steps.  Every wave pops on its own -- no work-group barrier between its steps, the claims of the
        //         eight waves interleave as they come -- until the front episode is idle or stalled, or the wave has nothing left to
        //         pop.  (A barrier per pop made every wave wait for the slowest of ~500 scattered memory requests, every step.)
        //         A claim is WON iff the voxel was free or held by a later episode; what that costs the holder is settled when it
        //         commits (E).  scratch/group_sim.c models this with the waves stepping in random order (kmax). -------------------------
        if (tid == frontslot && has && started && cur == GRP_NONE) shI[GI_READY] = 1;
        GTICK(0);
        for (int step = 0; step < (kmax & 0xFF); ++step) {
            // The exit must be taken by the WHOLE wave: other waves set GI_READY while this loop runs, and a plain per-lane read is
            // only uniform if every lane reads it in the same instruction -- the compiler is free to duplicate the read into the
            // two arms of `if (active)` below, lanes then see the flag at different times, some leave and the rest go on under a
            // partial EXEC mask, where the DPP scans of coop_reload read lane 63's stale register (round 4: the group-* stress
            // modes tripped guard site 3 in 3 of 8 runs with one code generation, 0 of 24 with others).  One lane's value for all:
            if (step >= kmin && grp_flag(shI, GI_READY)) break;
            bool active = false;
            if (has && cur != GRP_NONE) {
                bool room = true, need = false;
                if (started) {
                    room = np + 6 <= tcap;
                    const int used = np & (GRP_CH - 1), left = (np == 0 || used == 0) ? 0 : GRP_CH - used;
                    need = room && left < 6 && spare < 0;
                } else need = spare < 0;
                if (need) {
                    // pops only (nothing is freed while the waves run free); a request that finds the pool empty undoes its
                    // decrement -- harmless, nobody can pop "below empty".  The front episode falls back on its reserve.
                    const int idx = atomicSub(&shI[GI_NFREECH], 1) - 1;
                    if (idx >= 0) spare = (int)freech[idx];
                    else {
                        atomicAdd(&shI[GI_NFREECH], 1);
                        if (tid == frontslot && shI[GI_NRES] > 0) spare = (int)resch[--shI[GI_NRES]];
                        else room = false;
                    }
                }
                active = room;
                if (!room && tid == frontslot) {
                    if (started) shI[GI_FORCED] = 1;  // commits what it has done; that returns its chunks
                    // A front episode that has not even popped its root and finds pool AND reserve empty: every chunk is held by
                    // episodes behind it.  Nothing can commit before the front, so everything behind it is squashed -- always
                    // legal -- and its chunks come back.
                    else atomicMin(&shQ[GQ_KSQ], tid == lastslot ? ~0ull : succ[tid]);
                    shI[GI_READY] = 1;
                }
            }
            if (!__any(active)) break;
            unsigned long long old[6];
            float val[6];
            uint32_t nbv[6];
            uint32_t wonm = 0;
            GTICK(11);
            if (active) {
                // guard site 3 (always on): the voxel in hand must be a voxel -- checked once, before the six claims it would index
                const bool cur_ok = cur < (uint32_t)PN;
                if (!cur_ok && stats) {
                    if (atomicAdd(&stats[20], 1ull) == 0ull) {
                        stats[21] = 3ull;
                    }
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    nbv[k] = cur + (uint32_t)noff32[k];
                    old[k] = 0ull; val[k] = 0.0f;
                    if (nbv[k] < (uint32_t)PN && cur_ok) {
                        old[k] = __hip_atomic_fetch_min(&state[nbv[k]], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        val[k] = (aff + (int64_t)ax[k] * PN)[k < 3 ? cur : nbv[k]];
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                GTICK(14);
                // a voxel taken from a later episode: that episode finds out when it commits -- a lost discovery it has not popped
                // is dropped there (sequentially it would never have discovered it), a lost discovery it HAS popped makes it and
                // everything behind it wrong.  (Squashing the holder at once cost 95 % of the work on tie plateaus, where the
                // episodes of a sub-round claim the same free voxels all the time.)
#pragma unroll
                for (int k = 0; k < 6; ++k) if (old[k] > key) wonm |= 1u << k;
                // discoveries (direction order = the reference's push order, watershed.py:146-158)
                if (started && cur_li >= 0) { pval[cur_addr] = WIN_POPPED; ++npop; if (curv > mx) mx = curv; }
                started = true;
                cur = GRP_NONE;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    if (wonm & (1u << k)) {
                        const uint32_t v = ford(val[k] * sc[ax[k]] + 0.0f);
                        if ((np & (GRP_CH - 1)) == 0) {                          // next chunk (reserved before the pop)
                            chlast = spare; spare = -1;
                            chtab[tid][np >> 3] = (unsigned short)chlast;
                        }
                        const int a = chlast * GRP_CH + (np & (GRP_CH - 1));
                        pvox[a] = nbv[k]; pval[a] = v;
                        if (v < bound) cand_add(v, np, a); else if (v < resmin) resmin = v;
                        ++np;
                    }
                }
            }
            {   // register candidates used up while the list still holds discoveries below the bound: rebuilt by the wave
                unsigned long long need = __ballot(active && ncand == 0 && ndef > 0);
                if (need && lane == 0) atomicAdd(&lstat[GS_RELOAD], (uint32_t)__popcll(need));
                while (need) {
                    const int L = __ffsll((long long)need) - 1;
                    coop_reload(L, rdlane(bound, L));
                    need &= need - 1ull;
                }
            }
            if (active) {
                select_next(bound);
                pub_res[tid] = resmin; pub_mx[tid] = mx;
                pub_info[tid] = (uint32_t)np | ((uint32_t)npop << 8) | (1u << 16) | ((cur == GRP_NONE ? 1u : 0u) << 17);
                if (tid == frontslot && cur == GRP_NONE) shI[GI_READY] = 1;
            }
            {
                const unsigned long long ab = __ballot(active);
                if (ab && lane == 0) atomicAdd(&lstat[GS_EXEC], (uint32_t)__popcll(ab));
            }
        }
        GTICK(11);
        __syncthreads();                                                        // (S5) the waves are back: lists written and published
        GTICK(2);
        {
            const unsigned long long ksq = shQ[GQ_KSQ];
            if (ksq != ~0ull) {                                                 // uniform
                if (has && key >= ksq) squash_me();                             // (publishes "not started")
                if (tid == 0) atomicAdd(&lstat[GS_SQ_LOST], 1u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            // an idle episode appends nothing more: its spare chunk goes back (frees only from here to the end of the sub-round)
            if (has && started && cur == GRP_NONE && spare >= 0) { freech[atomicAdd(&shI[GI_NFREECH], 1)] = (unsigned short)spare; spare = -1; }
        }
        GTICK(3);
        // ---- E: commit the idle episodes at the front, in order; bursts as wide as the window ------------------------------------------------
        bool front_ready = grp_flag(shI, GI_FORCED) != 0 || ((pub_info[frontslot] >> 17) & 1u);
        const bool forced_front = grp_flag(shI, GI_FORCED) != 0;
#ifdef ISX_FLOOD_PROF
        if (tid == 0) { if (!front_ready) atomicAdd(&lstat[GS_C_NOTREADY], 1u); atomicAdd(&lstat[GS_NWSUM], (uint32_t)shI[GI_NW]); }
#endif
        for (int burst = 0; front_ready && burst < (max_bursts & 0xFF) && shI[GI_NW] > 0 && !grp_flag(shI, GI_SERIAL); ++burst) {
            const int nwb = shI[GI_NW];
            const unsigned long long qh = shI[GI_SCOUNT] > 0 ? srun[shI[GI_SHEAD]].key : ~0ull;
            // the front episode with a full list (or an empty pool) commits what it has done: the element in hand and every
            // candidate go back to the queue as residuals -- its owner rescans its list
            const int fslot = (int)oslot[ord_at(0)];
            if (burst == 0 && forced_front && tid == fslot && has && started && cur != GRP_NONE) {
                uint32_t rm = 0xFFFFFFFFu;
                int ch = 0;
                for (int i = 0; i < np; ++i) {
                    if ((i & (GRP_CH - 1)) == 0) ch = (int)chtab[tid][i >> 3];
                    const uint32_t v = pval[ch * GRP_CH + (i & (GRP_CH - 1))];
                    if (v != WIN_POPPED && v < rm) rm = v;
                }
                cur = GRP_NONE; cand_clear(); resmin = rm;
                pub_res[tid] = rm; pub_info[tid] = (uint32_t)np | ((uint32_t)npop << 8) | (1u << 16) | (1u << 17);
                atomicAdd(&lstat[GS_FORCED], 1u);
            }
            __syncthreads();                                                    // (E0)
            GTICK(12);
            // lane j = window position j
            const int j = tid;
            const bool valid = j < nwb;
            const int s = valid ? (int)oslot[ord_at(j)] : 0;
            const uint32_t info = valid ? pub_info[s] : 0u;
            const bool idle = valid && ((info >> 17) & 1u);
            const uint32_t e_np = info & 0xFFu, e_npop = (info >> 8) & 0xFFu;
            const uint32_t e_res = valid ? pub_res[s] : 0xFFFFFFFFu, e_mx = valid ? pub_mx[s] : 0u;
            const unsigned long long ek = valid ? okey[ord_at(j)] : ~0ull;
            {
                const unsigned long long nb_ = ~__ballot(idle);
                if (lane == 0) wtmp[0][wave] = nb_ ? wave * 64 + __ffsll((long long)nb_) - 1 : (wave + 1) * 64;
            }
            __syncthreads();                                                    // (E1)
            int m = GRP_W;
#pragma unroll
            for (int w = GRP_NWV - 1; w >= 0; --w) if (wtmp[0][w] < (w + 1) * 64) m = wtmp[0][w];
            const bool inm = j < m;
            const uint32_t pc = inm ? e_np - e_npop : 0u;
            const uint32_t incl_min = wave_prefix_min(inm ? e_res : 0xFFFFFFFFu);
            const uint32_t incl_pc = wave_incl_sum(pc), incl_np = wave_incl_sum(inm ? e_np : 0u);
            if (lane == 63) { wtmp[1][wave] = (int)incl_min; wtmp[2][wave] = (int)incl_pc; wtmp[3][wave] = (int)incl_np; }
            __syncthreads();                                                    // (E2)
            uint32_t F = wave_shr1(incl_min, 0xFFFFFFFFu), poff = incl_pc - pc, nbase = incl_np - (inm ? e_np : 0u);
#pragma unroll
            for (int w = 0; w < GRP_NWV; ++w)
                if (w < wave) { const uint32_t t = (uint32_t)wtmp[1][w]; F = t < F ? t : F; poff += (uint32_t)wtmp[2][w]; nbase += (uint32_t)wtmp[3][w]; }
            const uint32_t qv = (uint32_t)(qh >> 32);
            if (qv < F) F = qv;
            uint32_t succv = 0xFFFFFFFFu;
            if (valid && j + 1 < nwb) succv = (uint32_t)(okey[ord_at(j + 1)] >> 32);
            const bool ok = inm && (uint32_t)(ek >> 32) <= F && (e_mx == 0u || (e_mx < F && e_mx < succv)) && poff + pc <= (uint32_t)GRP_PEND;
            {
                const unsigned long long nok = ~__ballot(ok);
                if (lane == 0) wtmp[0][wave] = nok ? wave * 64 + __ffsll((long long)nok) - 1 : (wave + 1) * 64;
            }
            __syncthreads();                                                    // (E3)
            int ncommit = GRP_W;
#pragma unroll
            for (int w = GRP_NWV - 1; w >= 0; --w) if (wtmp[0][w] < (w + 1) * 64) ncommit = wtmp[0][w];
            if (ncommit > nwb) ncommit = nwb;
#ifdef ISX_FLOOD_PROF
            if (ncommit == nwb) { if (tid == 0) atomicAdd(&lstat[GS_C_ALL], 1u); }
            else if (j == ncommit) {
                if (!inm) atomicAdd(&lstat[((info >> 16) & 1u) ? GS_C_RUNNING : GS_C_NOTSTARTED], 1u);
                else if (!((uint32_t)(ek >> 32) <= F && (e_mx == 0u || (e_mx < F && e_mx < succv)))) atomicAdd(&lstat[GS_C_BOUND], 1u);
                else atomicAdd(&lstat[GS_C_CAP], 1u);
            }
#endif
            if (ncommit == 0) break;                                            // uniform
            const uint32_t agebase = (uint32_t)shI[GI_AGEBASE];
            if (j < ncommit) { ci_age[s] = agebase + nbase; ci_off[s] = poff; }
            const bool cap_cut = j == ncommit && j < nwb && inm && (uint32_t)(ek >> 32) <= F && (e_mx == 0u || (e_mx < F && e_mx < succv));
            __syncthreads();                                                    // (E3a) tentative orders are out
            GTICK(1);
            // ---- verification: every POPPED discovery of a committing episode must still be its own; an unpopped one that an
            //      earlier episode took in the meantime is dropped (voxel slot <- GRP_NONE).  Short lists by their own lanes, four
            //      loads in flight; long ones by the whole wave, 64 per trip.
            {
                const bool mine = has && ci_off[tid] != GRP_NONE;
                bool wrong = false;
                unsigned long long longb = __ballot(mine && np > 12);
                while (longb) {
                    const int L = __ffsll((long long)longb) - 1;
                    const int slot = (wave << 6) + L;
                    const int npL = (int)rdlane((uint32_t)np, L);
                    const unsigned long long keyL = rdlane64(key, L);
                    bool bad = false;
                    for (int b0 = 0; b0 < npL; b0 += 64) {
                        const int i = b0 + lane;
                        if (i < npL) {
                            const int a = (int)chtab[slot][i >> 3] * GRP_CH + (i & (GRP_CH - 1));
                            const uint32_t vx = pvox[a];
                            const unsigned long long stv = vx < (uint32_t)PN ? __hip_atomic_load(&state[vx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : keyL;
                            if (stv != keyL) { if (pval[a] == WIN_POPPED) bad = true; else pvox[a] = GRP_NONE; }
                        }
                    }
                    if (__any(bad) && lane == L) wrong = true;
                    longb &= longb - 1ull;
                }
                if (mine && np <= 12) {
                    for (int i0 = 0; i0 < np; i0 += 4) {
                        unsigned long long st4[4]; int a4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int i = i0 + u;
                            st4[u] = key; a4[u] = -1;
                            if (i < np) {
                                a4[u] = (int)chtab[tid][i >> 3] * GRP_CH + (i & (GRP_CH - 1));
                                const uint32_t vx = pvox[a4[u]];
                                if (vx < (uint32_t)PN) st4[u] = __hip_atomic_load(&state[vx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (a4[u] >= 0 && st4[u] != key) { if (pval[a4[u]] == WIN_POPPED) wrong = true; else pvox[a4[u]] = GRP_NONE; }
                    }
                }
                if (wrong) atomicMin(&shQ[GQ_KINV], key);
            }
            __syncthreads();                                                    // (E3b)
            const unsigned long long kinv = shQ[GQ_KINV];
            if (kinv != ~0ull) {                                                // uniform: a committing episode is wrong -- it and everything behind it restart
                const unsigned long long below = __ballot(j < ncommit && ek < kinv);
                if (lane == 0) wtmp[0][wave] = __popcll(below);
                if (has && key >= kinv) { ci_off[tid] = GRP_NONE; squash_me(); }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                                                // (E3c)
                int nc2 = 0;
#pragma unroll
                for (int w = 0; w < GRP_NWV; ++w) nc2 += wtmp[0][w];
                ncommit = nc2;
                if (tid == 0) { shQ[GQ_KINV] = ~0ull; atomicAdd(&lstat[GS_SQ_LOST], 1u); }
                if (ncommit == 0) break;                                        // uniform
            }
            did_commit = true;
            // cut only because `pend` is full?  then another burst of this sub-round takes the rest
            if (kinv == ~0ull && j == ncommit && j < nwb) shI[GI_ANY] = cap_cut ? 1 : 0;
            if (j == ncommit - 1) { shI[GI_NPTOT] = (int)(nbase + e_np); shI[GI_PENDN] = (int)(poff + pc); shI[GI_PEND0] = 0; }
            __syncthreads();                                                    // (E4) orders are out
            GTICK(4);
            // the committing episodes: labels (watershed.py:153), residuals in age order, chunks and slot back.  A LONG list (the
            // front's descent: up to 190 entries while everybody else waits) is written by its whole wave, 64 entries per pass
            // with the residuals ranked by ballot; the short ones by their own lanes in parallel.
            {
                const bool mine = has && ci_off[tid] != GRP_NONE;
                const uint32_t my_po = mine ? ci_off[tid] : 0u, my_base = mine ? ci_age[tid] : 0u;
                unsigned long long longb = __ballot(mine && np > 12);
                while (longb) {
                    const int L = __ffsll((long long)longb) - 1;
                    const int slot = (wave << 6) + L;
                    const int npL = (int)rdlane((uint32_t)np, L);
                    const uint32_t labL = rdlane(lab, L), baseL = rdlane(my_base, L);
                    uint32_t poL = rdlane(my_po, L);
                    for (int b0 = 0; b0 < npL; b0 += 64) {
                        const int i = b0 + lane;
                        const bool v_ = i < npL;
                        uint32_t vx = 0u, f = WIN_POPPED;
                        if (v_) { const int a = (int)chtab[slot][i >> 3] * GRP_CH + (i & (GRP_CH - 1)); vx = pvox[a]; f = pval[a]; }
                        const bool okv = v_ && vx < (uint32_t)PN;            // (a dropped entry carries GRP_NONE)
                        if (okv) { __hip_atomic_store(&state[vx], (unsigned long long)labL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); out[vx] = labL; }
                        const bool isres = okv && f != WIN_POPPED;
                        const unsigned long long rb = __ballot(isres);
                        if (isres) {
                            HeapEnt r;
                            r.key = ((unsigned long long)f << 32) | (unsigned long long)(baseL + (uint32_t)i + 1u);
                            r.idx = vx; r.pad = labL;
                            pend[poL + (uint32_t)__popcll(rb & lt_mask)] = r;
                        }
                        poL += (uint32_t)__popcll(rb);
                    }
                    // the slots reserved for residuals that were dropped stay empty
                    {
                        const uint32_t endL = rdlane(my_po, L) + rdlane((uint32_t)np - (uint32_t)npop, L);
                        for (uint32_t q = poL + (uint32_t)lane; q < endL; q += 64u) { HeapEnt r; r.key = ~0ull; r.idx = 0; r.pad = 0; pend[q] = r; }
                    }
                    longb &= longb - 1ull;
                }
                if (mine) {
                    if (np <= 12) {
                        uint32_t po = my_po;
                        int ch = 0;
                        for (int i = 0; i < np; ++i) {
                            if ((i & (GRP_CH - 1)) == 0) ch = (int)chtab[tid][i >> 3];
                            const int a = ch * GRP_CH + (i & (GRP_CH - 1));
                            const uint32_t vx = pvox[a], f = pval[a];
                            if (vx >= (uint32_t)PN) continue;                 // dropped at the verification (GRP_NONE)
                            __hip_atomic_store(&state[vx], (unsigned long long)lab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            out[vx] = lab;
                            if (f != WIN_POPPED) {
                                HeapEnt r;
                                r.key = ((unsigned long long)f << 32) | (unsigned long long)(my_base + (uint32_t)i + 1u);
                                r.idx = vx; r.pad = lab;
                                pend[po++] = r;
                            }
                        }
                        for (const uint32_t endp = my_po + (uint32_t)np - (uint32_t)npop; po < endp; ++po) { HeapEnt r; r.key = ~0ull; r.idx = 0; r.pad = 0; pend[po] = r; }
                    }
                    free_chunks();
                    has = false;
                    ci_off[tid] = GRP_NONE;
                    freeslot[atomicAdd(&shI[GI_NFREESLOT], 1)] = (unsigned short)tid;
                }
                // committed pops of this wave: one counter update
                const uint32_t pops_w = rdlane(wave_incl_sum(mine ? (uint32_t)npop + 1u : 0u), 63);
                if (lane == 0 && pops_w) atomicAdd(&lstat[GS_POPS], pops_w);
            }
            __syncthreads();                                                    // (E5)
            if (tid == 0) {
                shI[GI_AGEBASE] = (int)(agebase + (uint32_t)shI[GI_NPTOT]);
                shI[GI_OH] = (shI[GI_OH] + ncommit) & (GRP_W - 1);
                shI[GI_NW] = nwb - ncommit;
                atomicAdd(&lstat[GS_BURST], 1u);
            }
            __syncthreads();                                                    // (E6)
            GTICK(5);
            // ---- enqueue the burst's residuals --------------------------------------------------------------------------------------------
            // (slices of one residual per lane; pend[GI_PEND0, GI_PENDN) is what the serial continuation would still have to queue)
            if (!enqueue_batch()) break;                                        // uniform (set before a barrier every thread has passed)
            GTICK(6);
            check_order(10 + burst);
            if (ncommit < nwb && !grp_flag(shI, GI_ANY)) break;                           // uniform (GI_ANY: written before E4)
            __syncthreads();
            if (tid == 0) shI[GI_ANY] = 0;
        }
        }
        __syncthreads();                                                        // everybody has read this sub-round's verdicts
        if (tid == 0) {
            shI[GI_FORCED] = 0; shI[GI_ANY] = 0;
            shQ[GQ_KSQ] = ~0ull; shQ[GQ_KINV] = ~0ull; shI[GI_READY] = 0;
        }
        if (grp_flag(shI, GI_SERIAL)) continue;
        GTICK(13);
        // ---- G: refill the window from the run (from the buckets when the run is empty) ------------------------------------------------------
        if (did_commit || shI[GI_NW] < wcap) {
            for (;;) {
                __syncthreads();
                const bool need_drain = shI[GI_SCOUNT] == 0 && cold_on && (shI[GI_COLD_COUNT] > 0 || shI[GI_NSTAGE] > 0);
                if (need_drain) {
                    GTICK(7);
                    if (wave == 0) grp_w0_drain(&scold, srun, stage, hist, shI, lstat, drain_n, lane);
                    __syncthreads();
                    GTICK(9);
                    if (grp_flag(shI, GI_SERIAL)) break;
                    // the drained bucket comes unsorted unless its list was in key order: a bitonic sort over ALL waves (one
                    // wave sorting 1024 entries was two thirds of a drain)
                    const int sn = shI[GI_SORTN];
                    if (sn > 1) {
                        int np2 = 2;
                        while (np2 < sn) np2 <<= 1;
                        for (int q = sn + tid; q < np2; q += GRP_W) { HeapEnt pad_; pad_.key = ~0ull; pad_.idx = 0; pad_.pad = 0; srun[q] = pad_; }
                        __syncthreads();
                        // Pair t of a stage with distance j2 <= 64 lies in the 128-entry block [128 (t >> 6), +128): a wave's pairs
                        // (t = tid, tid + 512, ...) stay in that wave's own blocks, so consecutive stages with j2 <= 64 need no
                        // work-group barrier -- the LDS executes one wave's accesses in order.  Only the stages with j2 >= 128
                        // exchange entries between waves: 9 barriers for 1 024 entries instead of 55 (a barrier with eight waves
                        // is 300-400 cycles: the sort was 25 k cycles of a 43 k-cycle drain).
                        bool prev_wide = false;                                  // (the barrier above covers the padding and the drain)
                        for (int k2 = 2; k2 <= np2; k2 <<= 1)
                            for (int j2 = k2 >> 1, lj = 31 - __clz(j2); j2 > 0; j2 >>= 1, --lj) {
                                const bool wide = j2 > 64;
                                if (wide || prev_wide) __syncthreads();
                                else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
                                prev_wide = wide;
                                for (int t = tid; t < np2 / 2; t += GRP_W) {
                                    const int i2 = ((t >> lj) << (lj + 1)) | (t & (j2 - 1));
                                    const u32x4 a2 = run_ld(srun, i2), b2 = run_ld(srun, i2 + j2);
                                    const unsigned long long ka = ((unsigned long long)a2.y << 32) | a2.x, kb = ((unsigned long long)b2.y << 32) | b2.x;
                                    const bool up = (i2 & k2) == 0;
                                    if ((ka > kb) == up) { run_st(srun, i2, b2); run_st(srun, i2 + j2, a2); }
                                }
                            }
                        __syncthreads();
                    }
                    if (tid == 0) shI[GI_SORTN] = 0;
                    GTICK(10);
                }
                const int nw5 = shI[GI_NW], sc5 = shI[GI_SCOUNT], sh5 = shI[GI_SHEAD], nf5 = shI[GI_NFREESLOT];
                int take = wcap - nw5;
                if (take > sc5) take = sc5;
                if (take > nf5) take = nf5;
                if (take <= 0) break;
                if (tid < take) {
                    const u32x4 e = run_ld(srun, (sh5 + tid) & SMASK);
                    const int sl = (int)freeslot[nf5 - 1 - tid];
                    okey[ord_at(nw5 + tid)] = ((unsigned long long)e.y << 32) | e.x;
                    oslot[ord_at(nw5 + tid)] = (unsigned short)sl;
                    newroot[sl] = e; newflag[sl] = 1u;
                    unsigned long long nx = ~0ull;
                    if (tid + 1 < take) { const u32x4 e2 = run_ld(srun, (sh5 + tid + 1) & SMASK); nx = ((unsigned long long)e2.y << 32) | e2.x; }
                    succ[sl] = nx;
                    if (tid == 0 && nw5 > 0) succ[oslot[ord_at(nw5 - 1)]] = ((unsigned long long)e.y << 32) | e.x;
                }
                __syncthreads();
                if (tid == 0) { shI[GI_NW] = nw5 + take; shI[GI_SCOUNT] = sc5 - take; shI[GI_SHEAD] = (sh5 + take) & SMASK; shI[GI_NFREESLOT] = nf5 - take; }
            }
        }
        GTICK(7);
        check_order(30);
        // the front's reserve is topped up from the pool (nobody allocates or frees between here and the next (S0))
        if (tid == 0) {
            while (shI[GI_NRES] < reserve && shI[GI_NFREECH] > 0) resch[shI[GI_NRES]++] = freech[--shI[GI_NFREECH]];
            publish_heads();
        }
    }
    __syncthreads();

#ifdef ISX_FLOOD_PROF
    // phase timers of the work-group's first lane: 0 bounds (A), 1 claims up to (S3), 2 read-back + lists + next pop, 3 squash + publish,
    // 4 commit decision, 5 labels + residuals, 6 enqueue, 7 refill + drains, 8 loop end
#ifndef ISX_FLOOD_PROF_WAVE
#define ISX_FLOOD_PROF_WAVE 0
#endif
    if (stats && tid == 64 * ISX_FLOOD_PROF_WAVE) for (int i = 0; i < 16; ++i) atomicAdd(&stats[32 + i], tk_acc[i]);        // (development: whose wave's phase timers are reported)
#endif
#undef GTICK
    if (stats && tid == 0) {
        atomicAdd(&stats[0], (unsigned long long)st_sub);
        atomicAdd(&stats[1], (unsigned long long)lstat[GS_POPS]);
        atomicAdd(&stats[2], (unsigned long long)lstat[GS_EXEC]);
        atomicAdd(&stats[3], (unsigned long long)lstat[GS_RUNINS]);
        atomicAdd(&stats[4], (unsigned long long)lstat[GS_COLD]);
        atomicAdd(&stats[5], (unsigned long long)lstat[GS_DRAIN]);
        atomicAdd(&stats[6], (unsigned long long)lstat[GS_WINEV]);
        atomicAdd(&stats[7], (unsigned long long)lstat[GS_WININS]);
        atomicAdd(&stats[11], (unsigned long long)lstat[GS_SQ_LOST]);
        atomicAdd(&stats[12], (unsigned long long)lstat[GS_SQ_EPS]);
        atomicAdd(&stats[14], (unsigned long long)lstat[GS_SQ_BOUND]);
        atomicAdd(&stats[15], (unsigned long long)lstat[GS_FORCED]);
        atomicAdd(&stats[19], (unsigned long long)lstat[GS_SLOW]);
        atomicAdd(&stats[60], (unsigned long long)lstat[GS_DRAIN]);        // (60..63: the watchdog dumps own 20..31)
        atomicAdd(&stats[61], (unsigned long long)scold.fifo_drains);
        atomicAdd(&stats[62], (unsigned long long)lstat[GS_POPS]);
        atomicAdd(&stats[63], (unsigned long long)st_sub);
        atomicAdd(&stats[17], (unsigned long long)lstat[GS_BURST]);
        atomicAdd(&stats[18], (unsigned long long)lstat[GS_RELOAD]);
        if (grp_flag(shI, GI_SERIAL)) atomicAdd(&stats[8], 1ull);
#ifdef ISX_FLOOD_PROF
        for (int i = 0; i < 10; ++i) atomicAdd(&stats[48 + i], (unsigned long long)lstat[GS_C_NOTREADY + i]);
#endif
    }
    if (grp_flag(shI, GI_SERIAL)) {
        // Exact continuation on one lane: every queued element (window roots -- their speculative work is dropped, `out`
        // only ever holds committed labels --, run, staged, pending residuals, buckets) goes into the binary heap.
        const int mode = grp_flag(shI, GI_SERIAL);
        if (max_bursts & 0x200) {            // development: no continuation, only report that (and where) it was asked for
            if (tid == 0 && stats) { atomicAdd(&stats[22], 1ull); stats[23] = (unsigned long long)mode | ((unsigned long long)st_sub << 8) | ((unsigned long long)csize << 32); }
            return;
        }
        adopt();                                                                // a root handed to this slot but not picked up yet
        if (has) { u32x4 r = {(uint32_t)key, (uint32_t)(key >> 32), root, lab}; newroot[tid] = r; }
        newflag[tid] = has ? 1u : 0u;
        __syncthreads();
        if (wave != 0) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t hn = 0;
        if (mode == 2) {
            if (lane == 0)
                for (uint32_t j = 0; j < ns; ++j) {
                    HeapEnt e; e.key = ZKEY | (unsigned long long)j; e.idx = (uint32_t)(keys[first + j] & 0xFFFFFFFFull); e.pad = 0;
                    heap_push(heap, hn, e);
                }
        } else {
            if (lane == 0) {
                for (int s = 0; s < GRP_W; ++s)
                    if (newflag[s]) { const u32x4 r = newroot[s]; HeapEnt e; e.key = ((unsigned long long)r.y << 32) | r.x; e.idx = r.z; e.pad = 0; heap_push(heap, hn, e); }
                const int sc = shI[GI_SCOUNT], sh_ = shI[GI_SHEAD];
                for (int j = 0; j < sc; ++j) { HeapEnt e = srun[(sh_ + j) & SMASK]; e.pad = 0; heap_push(heap, hn, e); }
                for (int j = 0; j < shI[GI_NSTAGE]; ++j) { HeapEnt e = stage[j]; e.pad = 0; heap_push(heap, hn, e); }
                for (int j = shI[GI_PEND0]; j < shI[GI_PENDN]; ++j) { HeapEnt e = pend[j]; e.pad = 0; if (e.key != ~0ull) heap_push(heap, hn, e); }
                for (int j = 0; j < shI[GI_NHOT]; ++j) { HeapEnt e = hotb[j]; e.pad = 0; heap_push(heap, hn, e); }
                for (int j = shI[GI_WB0]; j < shI[GI_NWB]; ++j) { HeapEnt e = pend[j]; e.pad = 0; heap_push(heap, hn, e); }
            }
            hn = grp_w0_collect_cold(&scold, heap, hn, lane);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        // sanity of what the continuation was handed (an element outside the volume or more elements than the component has
        // voxels would take the heap or the flood out of bounds): reported, the component is left as it is
        {
            const uint32_t hn0 = rdlane(hn, 0);
            bool bad = (unsigned long long)hn0 > csize;
            for (uint32_t j = lane; j < hn0 && j < (uint32_t)csize; j += 64) if (cld_ent(&heap[j]).idx >= (uint32_t)PN) bad = true;
            if (__any(bad)) {
                if (lane == 0 && stats) { atomicAdd(&stats[20], 1ull); stats[21] = 100ull + (unsigned long long)mode; }
                return;
            }
        }
        if (lane == 0) serial_flood(heap, hn, (uint32_t)shI[GI_AGEBASE], aff, mask, out, PY, PX, PN, s0, s1, s2);
    }
}
