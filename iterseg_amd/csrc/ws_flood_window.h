// ws_flood_window.h -- the flood of one large mask component as an ORDERED SPECULATIVE WINDOW (one wave per component).
// Included by ws.hip inside namespace isx (uses HeapEnt, Cold, cold_append/cold_drain, serial_flood, ...).
//
// Replaces raveled_affinity_watershed (watershed.py:95-159) for components of >= COOP_MIN_SIZE voxels, bit-exactly.
//
// The reference pops the minimum (value, age) element, labels + pushes its unlabelled mask neighbours, repeats.  Call
// the EPISODE of a queued element x the pops the sequential loop makes from popping x until it pops the next element
// that was already queued: x itself, then every newly discovered voxel that sorts below the next queued key (a new
// element is younger than every queued one, so that is `value < value(next)`), in (value, discovery order) order.  An
// episode is a small local flood around x; the sequential pop sequence is the concatenation of the episodes of the
// queued elements in key order.  This kernel keeps the W smallest queued elements in a window (LDS, sorted by key) and
// runs their episodes concurrently, one pop per episode per sub-round, one dependent memory trip per sub-round:
//
//   * discovery = claim: one returning 64-bit atomicMin per neighbour on state[nb] with the episode's ROOT KEY as
//     priority.  state[v]: 0 = not in the mask, 1..2^32-1 = final label, ~0 = free, otherwise the root key of the
//     speculative episode holding the voxel (keys are >= 2^32).  A smaller key (earlier episode) takes a voxel over;
//   * commit is in window order: the front episode commits when it is idle (nothing left below its bound): labels are
//     written, its unpopped discoveries ("residuals") get their ages and enter the queue;
//   * an episode is WRONG -- squashed together with everything behind it, claims released -- when (a) an earlier
//     episode took over a voxel it had already used, or (b) its bound dropped below something it popped (a residual
//     or re-inserted root landed between its root and its largest pop).  Everything else stays valid: residuals that
//     land inside the window's key range simply become new roots at their sorted position.
// By induction over window positions every committed episode made exactly the sequential pops (the front episode is
// never wrong: it has the highest priority and its bound is the true next key).  scratch/window_sim.c is the model
// of these rules on the CPU; it reproduces the sequential labels on every test input.
//
// Queue = window (roots) + sorted run in LDS + the cold bucket arena in HBM (unchanged: cold_append / cold_drain).

// Two instantiations (flood_impl): the large tier <WIN_W, WIN_T, WIN_RCAP, WIN_PEND> (130 KB of LDS: one component per CU) and the
// small tier <WIN_SW, WIN_ST, WIN_SRCAP, WIN_SPEND> (~30 KB: five mid-size components per CU) for components below WIN_SMALL_MAX voxels.
static constexpr int WIN_W = 192;                 // episodes in flight (physical slots)
static constexpr int WIN_T = 48;                  // discoveries one episode holds before it must become the front
static constexpr int WIN_RCAP = 2048;             // LDS run capacity
static constexpr int WIN_STAGE = 128;             // staged cold-bound elements (flushed 64 at a time)
static constexpr int WIN_PEND = 512;              // residuals of one commit burst, in age order
static constexpr int WIN_SW = 48, WIN_ST = 24, WIN_SRCAP = 512, WIN_SPEND = 128;      // the small tier
static constexpr unsigned long long WIN_SMALL_MAX = 8192;       // components below this many voxels use it (ISX_FLOOD_SMALL_MAX; 0 = none).  Measured r3,
                                                                // structured 512^3: seed 3 (2 939 components >= 512 voxels) 102.8 -> 96.5 ms, seed 0 77.2 -> 77.7;
                                                                // at 32 768 the mid-size components flood slower on the 48-episode window than they wait: seed 0 81.8 ms
static constexpr uint32_t WIN_NONE = 0xFFFFFFFFu;
static constexpr unsigned long long ST_FREE = ~0ull;
static constexpr uint32_t WIN_POPPED = 0xFFFFFFFFu;     // ford() of no real value

// One wave works on a component and the LDS executes a wave's DS instructions in issue order: a word written by one lane is
// seen by every later read of the same wave without waiting for the write to retire.  The fence therefore only has to stop
// the COMPILER from moving LDS accesses across it; the compiler inserts the waits that loaded values need on its own.
// (It used to be an `s_waitcnt lgkmcnt(0)`: SQ counters showed half of the kernel's cycles in waits, most of them here.)
#define WIN_FENCE() asm volatile("" ::: "memory")

__global__ void flood_state_init_kernel(const uint8_t *__restrict__ mask, const uint32_t *__restrict__ out,
                                        unsigned long long *__restrict__ state, int64_t PN)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PN) return;
    const uint32_t l = out[i];
    state[i] = mask[i] ? (l ? (unsigned long long)l : ST_FREE) : 0ull;
}

// inclusive prefix sum over the 64 lanes on the DPP path (row_shr 1/2/4/8, then row_bcast 15 and 31): a few cycles per
// step where a ds_bpermute shuffle costs an LDS round trip.  Lane 63 holds the total.
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v)
{
#define ISX_SUM_STEP(ctrl, rmask) { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false); v += t; }
    ISX_SUM_STEP(0x111, 0xf)
    ISX_SUM_STEP(0x112, 0xf)
    ISX_SUM_STEP(0x114, 0xf)
    ISX_SUM_STEP(0x118, 0xf)
    ISX_SUM_STEP(0x142, 0xa)
    ISX_SUM_STEP(0x143, 0xc)
#undef ISX_SUM_STEP
    return v;
}

template <int WW, int WT, int WRCAP, int WPEND>
__global__ __launch_bounds__(64) void flood_window_kernel(
    const unsigned long long *__restrict__ order,
    const float *__restrict__ aff, const uint8_t *__restrict__ mask, uint32_t *__restrict__ out,
    unsigned long long *__restrict__ state, const unsigned long long *__restrict__ keys, uint32_t n_seeds,
    const uint32_t *__restrict__ comp_first, const unsigned long long *__restrict__ comp_off, uint32_t n_comp,
    HeapEnt *__restrict__ heap_pool, const unsigned long long *__restrict__ arena_off, HeapEnt *__restrict__ arena,
    int64_t PY, int64_t PX, int64_t PN, float s0, float s1, float s2, unsigned long long coop_min_size, int run_limit,
    int drain_n, int wcap, int tcap, int max_bursts, float bk_lo, float bk_inv, unsigned long long *stats, int shard_rank, int shard_world,
    uint32_t *dbg_log, unsigned long long coop_max_size)
{
    __shared__ HeapEnt srun[WRCAP];            // .pad carries the element's label
    __shared__ HeapEnt stage[WIN_STAGE];
    __shared__ HeapEnt pend[WPEND];            // residuals of the episodes committing now, in (episode, discovery) = age order
    __shared__ u32x4 winq0[WW];                // {key.lo, key.hi, root voxel, label}
    __shared__ u32x4 winq1[WW];                // {cur voxel | WIN_NONE, meta, max popped value excl. cur, value of cur}; meta = npush | started << 8 | plist << 16 | cur entry << 24
    __shared__ __attribute__((aligned(16))) uint32_t pidx[WW * WT];   // discoveries of an episode: voxel ...
    __shared__ __attribute__((aligned(16))) uint32_t pval[WW * WT];   // ... and ordered value (WIN_POPPED once popped)
    __shared__ unsigned short pfree[WW];
    __shared__ unsigned short actpos[WW];
    __shared__ HeapEnt ovl[6 * 64];               // take-overs of this row (at most one per lane and direction): key = old holder, idx = voxel
    __shared__ int hist[64];
    __shared__ uint32_t lstat[16];
#ifdef ISX_FLOOD_CHECK
    __shared__ int dbg_seen[WW];
#endif
    const int lane = threadIdx.x;
    if (blockIdx.x >= n_comp) return;
    const unsigned long long okey = order[blockIdx.x];         // components by decreasing size (comp_order_key_kernel)
    if (okey == ~0ull) return;
    const uint32_t c = (uint32_t)okey;
    if (!comp_owned(comp_off, c, n_comp, shard_rank, shard_world)) return;
    const uint32_t first = comp_first[c];
    const uint32_t end = (c + 1 < n_comp) ? comp_first[c + 1] : n_seeds;
    if ((uint32_t)(keys[first] >> 32) == NO_PARENT) return;
    const unsigned long long csize = comp_off[c + 1] - comp_off[c];
    if (csize < coop_min_size || csize >= coop_max_size) return;     // larger ones: flood_group_kernel (ws_flood_group.h)
    HeapEnt *heap = heap_pool + comp_off[c];
    const uint32_t ns = end - first;

    const int noff32[6] = {-(int)(PY * PX), -(int)PX, -1, 1, (int)PX, (int)(PY * PX)};    // voxel indices are 32-bit (host-checked)
    const int ax[6] = {0, 1, 2, 2, 1, 0};
    const float sc[3] = {s0, s1, s2};
    const unsigned long long ZKEY = (unsigned long long)0x80000000u << 32;
    constexpr int SMASK = WRCAP - 1;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // ---- cold tier (large components only): arena views + clean metadata (as flood_wave_kernel) ---------------
    Cold cold;
    cold.on = arena_off && (arena_off[c + 1] > arena_off[c]);
    cold.failed = false;
    cold.bump = 0;
    cold.count = 0;
    cold.low = COLD_NBK;
    cold.fifo_drains = 0;
#ifdef ISX_FLOOD_PROF
    for (int i = 0; i < 8; ++i) cold.tk[i] = 0;
    cold.n_groups = 0;
#endif
    cold.cb = COLD_NONE;
    cold.pool_cap = 0;
    cold.vlo = bk_lo;
    cold.vinv = bk_inv;
    if (cold.on) {
        HeapEnt *base = arena + arena_off[c];
        cold.base = (uint32_t *)base;
        cold.pool_cap = (uint32_t)(arena_off[c + 1] - arena_off[c] - COLD_META_ENTS);
        for (int i = lane; i < COLD_NBK; i += 64) { cold.bhead()[i] = COLD_NONE; cold.btail()[i] = COLD_NONE; cold.bcount()[i] = 0; cold.bmeta()[i] = 0; }
        for (int i = lane; i < COLD_NBK / 64; i += 64) cold.bm0()[i] = 0ull;
        if (lane < COLD_NBK / 4096) cold.bm1()[lane] = 0ull;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- queue state (wave-uniform) ---------------------------------------------------------------------------
    int wh = 0, nw = 0, npfree = 0;               // window: circular over WW slots, at most wcap used
    int shead = 0, scount = 0, nstage = 0;
    uint32_t agebase = ns - 1;                    // seeds carry ages 0..ns-1 in index order; pushes continue
    uint32_t nspill = 0;                          // elements that could not be queued (kept for the serial continuation)
    bool serial = false, started_any = false;
    bool dirty = false;
    bool list_valid = false;                      // actpos[0, nact_carry) is the active list of the coming sub-round (window untouched since)
    int nact_carry = 0;
#ifdef ISX_FLOOD_CHECK
    unsigned long long dbg_key = 0, dbg_hmax = 0; int dbg_reason = 0, dbg_nw = 0, dbg_sc = 0; uint32_t dbg_count = 0, dbg_ecount = 0;
#endif                           // a key landed right behind an episode since the last bound check
    // statistics (ISX_FLOOD_STATS) live in LDS, bumped by lane 0: every live scalar costs SGPRs this kernel is short of
    enum { ST_SUB, ST_POPS, ST_EXEC, ST_HOT, ST_COLD, ST_DRAIN, ST_EVICT, ST_MID, ST_SQ_EV, ST_SQ_EPS, ST_ROWS, ST_OVER, ST_FORCED, ST_SLOW, ST_N };
    for (int i = lane; i < ST_N; i += 64) lstat[i] = 0u;
#define WSTAT(i, n) do { const uint32_t n_ = (uint32_t)(n); if (lane == 0) lstat[i] += n_; } while (0)
    uint32_t st_sub = 0;                          // the watchdog's counter stays in a register
#ifdef ISX_FLOOD_PROF
    unsigned long long tk_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk_prev = __builtin_readcyclecounter();
#define WTICK(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_readcyclecounter(); tk_acc[i] += t_ - tk_prev; tk_prev = t_; } while (0)
#else
#define WTICK(i) do { } while (0)
#endif

    auto slot_of = [&](int p) -> int { const int s = wh + p; return s >= WW ? s - WW : s; };
    auto key_at = [&](int s) -> unsigned long long { const u32x4 q = winq0[s]; return ((unsigned long long)q.y << 32) | q.x; };
    auto fresh_q1 = [&](uint32_t root_idx, int pl) -> u32x4 { u32x4 q = {root_idx, ((uint32_t)pl << 16) | (0xFFu << 24), 0u, 0u}; return q; };   // root not popped yet

    // elements that can no longer be queued go to the far end of this component's heap area (the serial
    // continuation picks them up); the queue never holds more than csize elements, so the two ends cannot meet
    auto spill64 = [&](bool has, unsigned long long key, uint32_t idx) {
        const unsigned long long b = __ballot(has);
        if (has) { HeapEnt e; e.key = key; e.idx = idx; e.pad = 0; heap[csize - 1ull - (nspill + (uint32_t)__popcll(b & lt_mask))] = e; }
        nspill += (uint32_t)__popcll(b);
    };

    auto flush_stage = [&](int upto) {            // append stage[0, upto) to the buckets, 64 at a time; keeps the rest
        int done = 0;
        while (done < upto && !serial) {
            const int m = upto - done < 64 ? upto - done : 64;
            const bool has = lane < m;
            HeapEnt e; e.key = 0; e.idx = 0; e.pad = 0;
            if (has) e = stage[done + lane];
            cold_append(cold, has, e.key, e.idx, e.pad, lane);
            if (cold.failed) { serial = true; break; }       // nothing of this batch was written: it stays staged
            done += m;
        }
        // compact what is left
        const int rest = nstage - done;
        if (done > 0 && rest > 0) {
            HeapEnt t0, t1; t0.key = 0; t0.idx = 0; t0.pad = 0; t1 = t0;
            if (lane < rest) t0 = stage[done + lane];
            if (lane + 64 < rest) t1 = stage[done + lane + 64];
            WIN_FENCE();
            if (lane < rest) stage[lane] = t0;
            if (lane + 64 < rest) stage[lane + 64] = t1;
            WIN_FENCE();
        }
        nstage = rest;
    };

    // the run is full: its largest elements go to their buckets (ascending, so that a bucket's list stays FIFO)
    auto run_evict = [&]() {
        if (!cold.on) { serial = true; return; }
        const int keep = run_limit / 2, ev = scount - keep;
        if (ev <= 0) return;
        WSTAT(ST_EVICT, (uint32_t)ev);
        int done = 0;
        while (done < ev) {
            const int m = ev - done < 64 ? ev - done : 64;
            const bool has = lane < m;
            HeapEnt e; e.key = 0; e.idx = 0; e.pad = 0;
            if (has) e = srun[(shead + keep + done + lane) & SMASK];
            cold_append(cold, has, e.key, e.idx, e.pad, lane);
            if (cold.failed) break;
            done += m;
        }
        if (cold.failed) {
            for (int j0 = keep + done; j0 < scount; j0 += 64) {      // close the gap [keep, keep + done)
                const bool mv = j0 + lane < scount;
                u32x4 e = {0u, 0u, 0u, 0u};
                if (mv) e = run_ld(srun, (shead + j0 + lane) & SMASK);
                WIN_FENCE();
                if (mv) run_st(srun, (shead + j0 + lane - done) & SMASK, e);
                WIN_FENCE();
            }
            scount -= done;
            serial = true;
            return;
        }
        scount = keep;
        // always-on tier assertion (the cheap form of the development build's check_tiers): the hot maximum may not lie in a
        // bucket above the lowest non-empty cold one; reported through stats[16] like the watchdog (host: ISX_ERR_INTERNAL)
        if (stats && keep > 0 && cold.count > 0 && cold_bucket(cold, srun[(shead + keep - 1) & SMASK].key) > cold.low && lane == 0)
            atomicAdd(&stats[16], 1ull);
    };

    // sorted insert of one (wave-uniform) element into the circular LDS run
    auto run_insert = [&](unsigned long long ek, uint32_t ei, uint32_t el) {
        int pos;
        {
            const int stride = (scount + 63) >> 6;
            int e = lane * stride + stride - 1;
            if (e > scount - 1) e = scount - 1;
            const bool seg_below = scount > 0 && lane * stride < scount && srun[(shead + e) & SMASK].key < ek;
            const int base = __popcll(__ballot(seg_below)) * stride;
            const int j = base + lane;
            const bool in_below = lane < stride && j < scount && srun[(shead + j) & SMASK].key < ek;
            pos = base + __popcll(__ballot(in_below));
            if (pos > scount) pos = scount;
        }
        if (pos >= scount - pos) {
            for (int base = scount - 64; base > pos - 64; base -= 64) {
                const int j = base + lane;
                const bool mv = j >= pos && j >= 0 && j < scount;
                u32x4 e = {0u, 0u, 0u, 0u};
                if (mv) e = run_ld(srun, (shead + j) & SMASK);
                WIN_FENCE();
                if (mv) run_st(srun, (shead + j + 1) & SMASK, e);
            }
        } else {
            for (int base = 0; base < pos; base += 64) {
                const int j = base + lane;
                const bool mv = j < pos;
                u32x4 e = {0u, 0u, 0u, 0u};
                if (mv) e = run_ld(srun, (shead + j) & SMASK);
                WIN_FENCE();
                if (mv) run_st(srun, (shead + j - 1) & SMASK, e);
            }
            shead = (shead - 1) & SMASK;
        }
        WIN_FENCE();
        if (lane == 0) { HeapEnt e; e.key = ek; e.idx = ei; e.pad = el; srun[(shead + pos) & SMASK] = e; }
        ++scount;
        if (pos == 0) dirty = true;                // new head of the run: the last window episode has a new, lower bound
        WIN_FENCE();
    };

    // give the claims of one episode back (lane-parallel over its discoveries)
    auto release_one = [&](unsigned long long key, int pl, int np) {
        for (int i = lane; i < np; i += 64) {
            const uint32_t v = pidx[pl * WT + i];
            atomicCAS(&state[v], key, ST_FREE);
        }
    };

    // squash every episode from window position sp on: claims released, back to "root not popped yet"
    auto squash_from = [&](int sp) {
        WSTAT(ST_SQ_EV, 1);
        list_valid = false;
        for (int base = sp; base < nw; base += 64) {
            const int p = base + lane;
            const bool valid = p < nw;
            const int s = valid ? slot_of(p) : 0;
            u32x4 q0 = {0u, 0u, 0u, 0u}, q1 = {0u, 0u, 0u, 0u};
            if (valid) { q0 = winq0[s]; q1 = winq1[s]; }
            const bool st = valid && ((q1.y >> 8) & 1u);
            const int np = st ? (int)(q1.y & 0xFFu) : 0, pl = (int)((q1.y >> 16) & 0xFFu);
            const unsigned long long key = ((unsigned long long)q0.y << 32) | q0.x;
            WSTAT(ST_SQ_EPS, (uint32_t)__popcll(__ballot(st)));
            for (int i = 0; __any(i < np); ++i)
                if (i < np) { const uint32_t v = pidx[pl * WT + i]; atomicCAS(&state[v], key, ST_FREE); }
            if (valid) winq1[s] = fresh_q1(q0.z, pl);
        }
        WIN_FENCE();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // a new root at its sorted position in the window (wave-uniform element, ek sorts below the last root or the
    // window has room at its end)
    auto win_insert = [&](unsigned long long ek, uint32_t ei, uint32_t el) {
        list_valid = false;
        int pos = 0;
        for (int base = 0; base < nw; base += 64) {
            const int p = base + lane;
            const bool lt = p < nw && key_at(slot_of(p)) < ek;
            pos += __popcll(__ballot(lt));
        }
        int pl;
        if (nw == wcap || npfree == 0) {
            // full: the last root goes back to the front of the run (it sorts below everything there)
            const int s = slot_of(nw - 1);
            const u32x4 q0 = winq0[s], q1 = winq1[s];
            pl = (int)((q1.y >> 16) & 0xFFu);
            if ((q1.y >> 8) & 1u) {
                release_one(((unsigned long long)q0.y << 32) | q0.x, pl, (int)(q1.y & 0xFFu));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                WSTAT(ST_SQ_EPS, 1);
            }
            shead = (shead - 1) & SMASK;
            if (lane == 0) { u32x4 e = {q0.x, q0.y, q0.z, q0.w}; run_st(srun, shead, e); }
            ++scount;
            --nw;
            WSTAT(ST_EVICT, 1);
            if (pos > nw) pos = nw;
        } else {
            pl = (int)pfree[--npfree];
        }
        if (pos < nw) {
            WSTAT(ST_MID, 1);
            u32x4 a[3], b[3];
            bool mv[3];
#pragma unroll
            for (int cidx = 0; cidx < 3; ++cidx) {
                const int p = pos + cidx * 64 + lane;
                mv[cidx] = p < nw;
                a[cidx] = b[cidx] = u32x4{0u, 0u, 0u, 0u};
                if (mv[cidx]) { const int s = slot_of(p); a[cidx] = winq0[s]; b[cidx] = winq1[s]; }
            }
            WIN_FENCE();
#pragma unroll
            for (int cidx = 0; cidx < 3; ++cidx)
                if (mv[cidx]) { const int s = slot_of(pos + cidx * 64 + lane + 1); winq0[s] = a[cidx]; winq1[s] = b[cidx]; }
            WIN_FENCE();
        }
        dirty = true;                              // the episode before `pos` has a new, lower bound
        if (lane == 0) {
            const int s = slot_of(pos);
            u32x4 q0 = {(uint32_t)ek, (uint32_t)(ek >> 32), ei, el};
            winq0[s] = q0;
            winq1[s] = fresh_q1(ei, pl);
        }
        ++nw;
        WIN_FENCE();
    };

#ifdef ISX_FLOOD_CHECK
    // development build: window sorted, window <= run head, run sorted
    auto check_order = [&](int tag) {
        bool bad = false;
        unsigned long long bk = 0, bnk = 0; int bp = -1, where = 0;
        for (int base = 0; base < nw; base += 64) {
            const int p = base + lane;
            if (p < nw) {
                const unsigned long long k = key_at(slot_of(p));
                const unsigned long long nk = p + 1 < nw ? key_at(slot_of(p + 1)) : (scount > 0 ? srun[shead].key : ~0ull);
                if (!(k < nk) && !bad) { bad = true; bk = k; bnk = nk; bp = p; where = p + 1 < nw ? 1 : 2; }
            }
        }
        for (int base = 0; base + 1 < scount; base += 64) {
            const int p = base + lane;
            if (p + 1 < scount && !(srun[(shead + p) & SMASK].key < srun[(shead + p + 1) & SMASK].key) && !bad) {
                bad = true; bk = srun[(shead + p) & SMASK].key; bnk = srun[(shead + p + 1) & SMASK].key; bp = p; where = 3;
            }
        }
        // every window episode owns its own discovery list; free list and owned lists partition the ids
        for (int i = lane; i < WW; i += 64) dbg_seen[i] = 0;
        WIN_FENCE();
        for (int base = 0; base < nw; base += 64) {
            const int p = base + lane;
            if (p < nw) atomicAdd(&dbg_seen[(winq1[slot_of(p)].y >> 16) & 0xFFu], 1);
        }
        for (int i = lane; i < npfree; i += 64) atomicAdd(&dbg_seen[pfree[i]], 1);
        WIN_FENCE();
        for (int i = lane; i < WW; i += 64)
            if (dbg_seen[i] != 1 && !bad) { bad = true; bk = (unsigned long long)i; bnk = (unsigned long long)dbg_seen[i]; bp = npfree; where = 5; }
        const unsigned long long bb = __ballot(bad);
        if (bb) serial = true;             // stop here: the serial continuation finishes the component
        if (bb && stats && lane == __ffsll((long long)bb) - 1 && stats[40] == 0ull) {
            stats[40] = (unsigned long long)tag; stats[41] = (unsigned long long)st_sub; stats[42] = (unsigned long long)nw; stats[43] = (unsigned long long)scount;
            stats[44] = bk; stats[45] = bnk; stats[46] = (unsigned long long)bp; stats[47] = (unsigned long long)where;
            for (int q = 0; q < 6; ++q) stats[10 + q] = 0;
            for (int q = 0; q < 4 && q < nw; ++q) stats[20 + q] = key_at(slot_of(nw - 1 - q));
            for (int q = 0; q < 4 && q < scount; ++q) stats[24 + q] = srun[(shead + q) & SMASK].key;
            stats[28] = (unsigned long long)cold.count; stats[29] = (unsigned long long)cold.fifo_drains; stats[30] = (unsigned long long)ns; stats[31] = (unsigned long long)nstage;
        }
    };
    // every cold element (arena's lowest bucket, staged) must sort after every hot element
    auto check_tiers = [&](int tag) {
        if (!cold.on) return;
        unsigned long long cmin = ~0ull;
        for (int i = lane; i < nstage; i += 64) { const unsigned long long k = stage[i].key; if (k < cmin) cmin = k; }
        if (cold.count > 0) {
            const uint32_t b = cold_lowest(cold, lane);
            uint32_t seg = cld32(&cold.bhead()[b]);
            uint32_t off = cld32(&cold.bmeta()[b]) & ~COLD_UNSORTED;
            while (seg != COLD_NONE) {
                const HeapEnt h = cld_ent(&cold.pool()[seg]);
                const uint32_t n = (uint32_t)(h.key >> 32);
                for (uint32_t j = off + lane; j < n; j += 64) { const unsigned long long k = cld64(&cold.pool()[seg + 1u + j].key); if (k < cmin) cmin = k; }
                off = 0;
                seg = (uint32_t)h.key;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = shfl64(cmin, lane ^ o); if (t < cmin) cmin = t; }
        unsigned long long hmax = 0ull;
        if (scount > 0) hmax = srun[(shead + scount - 1) & SMASK].key;
        else if (nw > 0) hmax = key_at(slot_of(nw - 1));
        if (cmin < hmax) {
            serial = true;
            if (stats && lane == 0 && stats[40] == 0ull) {
                stats[40] = (unsigned long long)tag; stats[41] = (unsigned long long)st_sub; stats[42] = (unsigned long long)nw; stats[43] = (unsigned long long)scount;
                stats[44] = hmax; stats[45] = cmin; stats[46] = (unsigned long long)cold.low; stats[47] = 7ull;
                stats[28] = (unsigned long long)cold.count; stats[29] = (unsigned long long)cold.fifo_drains; stats[30] = (unsigned long long)ns; stats[31] = (unsigned long long)nstage;
                stats[20] = dbg_key; stats[21] = dbg_hmax; stats[22] = (unsigned long long)dbg_reason; stats[23] = (unsigned long long)dbg_nw; stats[24] = (unsigned long long)dbg_sc;
            }
        }
    };
#define WCHECK(tag) do { check_order(tag); if (!serial) check_tiers(100 + (tag)); } while (0)
#else
#define WCHECK(tag) do { } while (0)
#endif
    // One (wave-uniform) element that the tier rule keeps hot: window if it sorts into it, else the run.
    auto hot_insert = [&](unsigned long long ek, uint32_t ei, uint32_t el) {
        WSTAT(ST_HOT, 1);
#ifdef ISX_FLOOD_CHECK
        if (cold.on && cold.count > 0) {
            const uint32_t truth = cold_lowest(cold, lane);
            if (truth != cold.low && stats && lane == 0 && stats[40] == 0ull) {
                stats[40] = 99ull; stats[41] = (unsigned long long)st_sub; stats[42] = (unsigned long long)truth; stats[43] = (unsigned long long)cold.low;
                stats[44] = ek; stats[45] = (unsigned long long)cold.count; stats[46] = (unsigned long long)nstage; stats[47] = (unsigned long long)cold_bucket(cold, ek);
            }
        }
#endif
        if (nw > 0 && ek < key_at(slot_of(nw - 1))) {
            if ((nw == wcap || npfree == 0) && scount >= WRCAP - 2) { run_evict(); if (serial) { spill64(lane == 0, ek, ei); return; } }
            win_insert(ek, ei, el);
            return;
        }
        if (scount == 0 && nw < wcap && npfree > 0) { win_insert(ek, ei, el); return; }
        if (scount >= WRCAP - 2) {
            run_evict();
            if (serial) { spill64(lane == 0, ek, ei); return; }
            // the eviction lowered the hot maximum (and possibly cold.low): the element was judged against the old tiers
            // and may belong to the cold tier now (found by the tier check of the development build: a full run in the
            // middle of a commit burst put 0.44 behind zeros that had just been evicted)
            bool below_hot_max = false;
            if (scount > 0) below_hot_max = ek < srun[(shead + scount - 1) & SMASK].key;
            else if (nw > 0) below_hot_max = ek < key_at(slot_of(nw - 1));
            const bool cold_free = (cold.count == 0 && nstage == 0) || cold_bucket(cold, ek) < cold.low;
            if (!below_hot_max && !cold_free) {
                if (lane == 0) { HeapEnt e; e.key = ek; e.idx = ei; e.pad = el; stage[nstage] = e; }
                ++nstage;
                WSTAT(ST_COLD, 1);
                WIN_FENCE();
                if (nstage >= WIN_STAGE - 64) flush_stage(64);
                return;
            }
        }
        run_insert(ek, ei, el);
    };

    // Up to one new queue element per lane.  Tier rule (keeps "every cold element sorts after every hot element"): an
    // element stays hot if it sorts below the hot maximum, or if nothing cold can precede it (cold tier empty, or its
    // bucket lies below the lowest non-empty bucket); otherwise it is staged for its bucket.
    auto enqueue64 = [&](bool has, unsigned long long key, uint32_t idx, uint32_t lab) {
        if (serial) { spill64(has, key, idx); return; }
        bool tocold = false;
        if (cold.on && (cold.count > 0 || nstage > 0)) {
            unsigned long long hot_max = 0ull;
            const bool hot_ne = nw > 0 || scount > 0;
            if (scount > 0) hot_max = srun[(shead + scount - 1) & SMASK].key;
            else if (nw > 0) hot_max = key_at(slot_of(nw - 1));
            tocold = has && !(hot_ne && key < hot_max) && cold_bucket(cold, key) >= cold.low;
        }
        {
            const unsigned long long b = __ballot(tocold);
            if (tocold) { HeapEnt e; e.key = key; e.idx = idx; e.pad = lab; stage[nstage + __popcll(b & lt_mask)] = e; }
            nstage += __popcll(b);
            WSTAT(ST_COLD, (uint32_t)__popcll(b));
        }
        WIN_FENCE();
        for (unsigned long long todo = __ballot(has && !tocold); todo; todo &= todo - 1ull) {
            const int l = __ffsll((long long)todo) - 1;
            const unsigned long long ek = rdlane64(key, l);
            const uint32_t ei = rdlane(idx, l), el = rdlane(lab, l);
            if (serial) { spill64(lane == 0, ek, ei); continue; }
            // re-evaluate against the current tiers (an eviction in this loop may have moved them)
            bool below_hot_max = false;
            if (scount > 0) below_hot_max = ek < srun[(shead + scount - 1) & SMASK].key;
            else if (nw > 0) below_hot_max = ek < key_at(slot_of(nw - 1));
            const bool cold_free = !cold.on || (cold.count == 0 && nstage == 0) || cold_bucket(cold, ek) < cold.low;
            if (!below_hot_max && !cold_free) {
                if (lane == 0) { HeapEnt e; e.key = ek; e.idx = ei; e.pad = el; stage[nstage] = e; }
                ++nstage; WSTAT(ST_COLD, 1);
                WIN_FENCE();
                if (nstage >= WIN_STAGE - 64) flush_stage(64);
                continue;
            }
#ifdef ISX_FLOOD_CHECK
            dbg_key = ek; dbg_reason = (below_hot_max ? 1 : 0) | (cold_free ? 2 : 0); dbg_nw = nw; dbg_sc = scount;
            dbg_hmax = scount > 0 ? srun[(shead + scount - 1) & SMASK].key : (nw > 0 ? key_at(slot_of(nw - 1)) : 0ull);
#endif
            hot_insert(ek, ei, el);
#ifdef ISX_FLOOD_CHECK
            if (!serial) check_tiers(700);
#endif
        }
        if (nstage >= 64 && !serial) flush_stage(64);
    };

    // ---- start: seeds (value 0, age = rank in index order) -------------------------------------------------------
    if (wcap > WW) wcap = WW;
    if (tcap > WT) tcap = WT;
    for (int i = lane; i < WW; i += 64) pfree[i] = (unsigned short)(WW - 1 - i);
    npfree = WW;
    WIN_FENCE();
    {
        started_any = true;
        const uint32_t in_win = ns < (uint32_t)wcap ? ns : (uint32_t)wcap;
        for (uint32_t j = lane; j < in_win; j += 64) {
            const uint32_t vi = (uint32_t)(keys[first + j] & 0xFFFFFFFFull);
            const unsigned long long k = ZKEY | (unsigned long long)j;
            const int pl = (int)pfree[npfree - 1 - (int)j];
            u32x4 q0 = {(uint32_t)k, (uint32_t)(k >> 32), vi, out[vi]};
            winq0[j] = q0;
            winq1[j] = fresh_q1(vi, pl);
        }
        nw = (int)in_win;
        npfree -= (int)in_win;
        const uint32_t hot_cap = (uint32_t)wcap + (uint32_t)run_limit;
        const uint32_t in_hot = ns > hot_cap ? hot_cap : ns;
        for (uint32_t j = in_win + lane; j < in_hot; j += 64) {
            HeapEnt e;
            e.key = ZKEY | (unsigned long long)j;
            e.idx = (uint32_t)(keys[first + j] & 0xFFFFFFFFull);
            e.pad = out[e.idx];
            srun[j - in_win] = e;
        }
        scount = in_hot > in_win ? (int)(in_hot - in_win) : 0;
        WIN_FENCE();
        if (ns > in_hot && !cold.on) { serial = true; started_any = false; }
        for (uint32_t j0 = in_hot; j0 < ns && !serial; j0 += 64) {       // seeds beyond the hot tiers start cold
            const uint32_t j = j0 + lane;
            const bool has = j < ns;
            const uint32_t vi = has ? (uint32_t)(keys[first + j] & 0xFFFFFFFFull) : 0u;
            cold_append(cold, has, ZKEY | (unsigned long long)j, vi, has ? out[vi] : 0u, lane);
            if (cold.failed) { serial = true; started_any = false; }    // restart from the seeds on the serial heap
        }
    }

    // ================================================= sub-rounds ==================================================
    // Watchdog (never expected to trip): the loop below commits at least the front episode every few sub-rounds, so far
    // fewer than this many are ever needed.  A trip is reported to the host as an error (stats[9]) with the state of
    // the front of the window; the component is then finished by the serial heap so that the grid always drains.
    const uint32_t sub_limit = csize > 60000000ull ? 0xFFFFFFFFu : 4096u + 64u * (uint32_t)csize;
    while (!serial && nw > 0) {
        ++st_sub;
        if (st_sub > sub_limit) {
            if (stats && lane == 0) {
                const u32x4 f0 = winq0[slot_of(0)], f1 = winq1[slot_of(0)];
                atomicAdd(&stats[9], 1ull);
                stats[24] = (unsigned long long)nw; stats[25] = (unsigned long long)scount; stats[26] = (unsigned long long)cold.count;
                stats[27] = ((unsigned long long)f1.x << 32) | f1.y; stats[28] = ((unsigned long long)f0.y << 32) | f0.x;
                stats[29] = scount > 0 ? srun[shead].key : ~0ull; stats[30] = ((unsigned long long)f1.z << 32) | (uint32_t)nstage;
                stats[31] = (unsigned long long)lstat[ST_POPS];
                const u32x4 g0 = winq0[slot_of(1)], g1 = winq1[slot_of(1)];
                stats[20] = ((unsigned long long)g0.y << 32) | g0.x; stats[21] = ((unsigned long long)g1.x << 32) | g1.y; stats[22] = (unsigned long long)wh;
                stats[23] = (unsigned long long)npfree;
            }
            serial = true;
            break;
        }
        WTICK(0);
        WCHECK(1);
        if (scount > run_limit) { run_evict(); if (serial) break; }
        WCHECK(2);
        // ---- A: an episode that popped past its bound (bound = value of the next larger queued key) is wrong.  A bound
        //         only ever drops when a new key lands right behind an episode, so the scan runs after such insertions only
        const unsigned long long qhead = scount > 0 ? srun[shead].key : ~0ull;
        if (dirty || (max_bursts & 0x100)) {
            dirty = false;
            int sp = -1;
            for (int base = 0; base < nw && sp < 0; base += 64) {
                const int p = base + lane;
                bool over = false;
                if (p < nw) {
                    const u32x4 q1 = winq1[slot_of(p)];
                    const unsigned long long nk = p + 1 < nw ? key_at(slot_of(p + 1)) : qhead;
                    uint32_t mx = q1.z;
                    if ((q1.y >> 24) != 0xFFu && q1.w > mx) mx = q1.w;          // the element it is about to pop counts
                    over = mx != 0u && mx >= (uint32_t)(nk >> 32);
                }
                const unsigned long long ob = __ballot(over);
                if (ob) sp = base + __ffsll((long long)ob) - 1;
            }
            if (sp >= 0) { WSTAT(ST_OVER, 1); squash_from(sp); }
        }
        // ---- B: episodes that want to pop ------------------------------------------------------------------------
        int nact = 0;
        bool front_forced = false;
        if (list_valid) {
            // nothing touched the window since the last sub-round (no commit, insertion, refill or squash -- three sub-rounds
            // in four while a long descent sits at the front): the episodes that still want to pop were compacted by C
            nact = nact_carry;
            const u32x4 f1 = winq1[slot_of(0)];
            const bool st0 = (f1.y >> 8) & 1u;
            front_forced = rdlane((uint32_t)(f1.x != WIN_NONE && st0 && (int)(f1.y & 0xFFu) + 6 > tcap), 0) != 0u;
        } else
        for (int base = 0; base < nw; base += 64) {
            const int p = base + lane;
            bool act = false, stalled = false;
            if (p < nw) {
                const u32x4 q1 = winq1[slot_of(p)];
                const bool st = (q1.y >> 8) & 1u;
                const bool wants = q1.x != WIN_NONE;
                const bool room = !st || (int)(q1.y & 0xFFu) + 6 <= tcap;
                act = wants && room;
                stalled = wants && !room;
            }
            if (base == 0) front_forced = rdlane((uint32_t)stalled, 0) != 0u;
            const unsigned long long ab = __ballot(act);
            if (act) actpos[nact + __popcll(ab & lt_mask)] = (unsigned short)p;
            nact += __popcll(ab);
        }
        WIN_FENCE();
        WTICK(1);
        // ---- C: one pop per active episode, 64 episodes per row -----------------------------------------------------
        unsigned long long squash_key = ~0ull;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // label stores / releases of earlier sub-rounds are performed
        int nnext = 0;                                            // active list of the next sub-round, compacted in place
        // While the front episode keeps descending nothing can commit, nothing is inserted and no bound moves: the next
        // sub-round is this phase again with the compacted list, and the loop below repeats it without going round the
        // other phases (three sub-rounds in four on config 2, where those phases cost 40 % of such a sub-round).
        for (;;) {
        nnext = 0;
        for (int r0 = 0; r0 < nact; r0 += 64) {
            WSTAT(ST_ROWS, 1);
            const bool act = r0 + lane < nact;
            const int p = act ? (int)actpos[r0 + lane] : 0;
            const int s = slot_of(p);
            const u32x4 q0 = winq0[s];
            u32x4 q1 = winq1[s];
            const unsigned long long mykey = ((unsigned long long)q0.y << 32) | q0.x;
            const uint32_t cv = q1.x;                             // the voxel popped now (idle lanes: front root, unused)
            const int pl = (int)((q1.y >> 16) & 0xFFu);
            const bool was_started = (q1.y >> 8) & 1u;
            int np = was_started ? (int)(q1.y & 0xFFu) : 0;
            uint32_t npop = was_started ? ((q1.y >> 9) & 0x7Fu) : 0u;       // discoveries popped so far (marked WIN_POPPED)
            uint32_t mx = q1.z;
            if (was_started && q1.w > mx) mx = q1.w;              // the popped element joins "everything this episode popped"
            unsigned long long old[6];
            float val[6];
            uint32_t nbv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                nbv[k] = cv + (uint32_t)noff32[k];
                old[k] = 0ull;
                val[k] = 0.0f;
                if (act && nbv[k] < (uint32_t)PN) {           // always true with the padded border; never fault
                    old[k] = __hip_atomic_fetch_min(&state[nbv[k]], mykey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    val[k] = (aff + (int64_t)ax[k] * PN)[k < 3 ? cv : nbv[k]];
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WTICK(2);
            // old > mykey: the voxel was free or held by a later episode -- it is mine now
            uint32_t won = 0, tookmask = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (act && old[k] > mykey) {
                    won |= 1u << k;
                    if (old[k] != ST_FREE) tookmask |= 1u << k;
                }
            }
            // Take-overs.  A lane whose claim of THIS row was taken over again (a later arrival of an earlier episode) simply
            // loses it: it finds out by reading the voxel back.  A take-over from an episode that had claimed the voxel in an
            // earlier sub-round makes that episode wrong.  Every fresh loss corresponds to exactly one take-over, so the
            // counts differ iff an older claim was hit; only then is the list of take-overs matched entry by entry.
            if (__ballot(tookmask != 0u)) {
                unsigned long long now[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    now[k] = mykey;
                    if (won & (1u << k)) now[k] = __hip_atomic_load(&state[nbv[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                uint32_t lost = 0;
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if ((won & (1u << k)) && now[k] != mykey) lost |= 1u << k;
                const uint32_t took_incl = wave_incl_sum((uint32_t)__popc(tookmask));
                const uint32_t n_took = rdlane(took_incl, 63), n_lost = rdlane(wave_incl_sum((uint32_t)__popc(lost)), 63);
                if (n_took != n_lost) {
                    WSTAT(ST_SLOW, 1);
                    int slot = (int)(took_incl - (uint32_t)__popc(tookmask));
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        if (tookmask & (1u << k)) { HeapEnt e; e.key = old[k]; e.idx = nbv[k]; e.pad = 0; ovl[slot++] = e; }
                    WIN_FENCE();
                    for (int i = 0; i < (int)n_took; ++i) {
                        const HeapEnt e = ovl[i];
                        bool mine = false;
#pragma unroll
                        for (int k = 0; k < 6; ++k)
                            if ((won & (1u << k)) && nbv[k] == e.idx && mykey == e.key) mine = true;
                        if (!__any(mine) && e.key < squash_key) squash_key = e.key;
                    }
                }
                won &= ~lost;
            }
            // discoveries in direction order = the reference's push order (watershed.py:146-158)
            if (act) {
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (won & (1u << k)) {
                        const float v = val[k] * sc[ax[k]] + 0.0f;              // -0.0 == +0.0 in the order
                        pidx[pl * WT + np] = nbv[k];
                        pval[pl * WT + np] = ford(v);
                        ++np;
                    }
            }
            WIN_FENCE();
            // next pop: the smallest unpopped discovery (ties: the earliest), if it sorts below the next queued key
            uint32_t bf = WIN_POPPED;
            int bi = -1;
            {
                // sixteen entries per trip: the four reads of a group are issued together (the kernel is bound by dependent
                // LDS trips, not by LDS bandwidth); entries past `np` are stale but in bounds (WT = 48) and masked
                const u32x4 *pv4 = (const u32x4 *)&pval[pl * WT];
                for (int i4 = 0; __any(act && 4 * i4 < np); i4 += 4) {
                    u32x4 g[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) g[k] = pv4[i4 + k < WT / 4 ? i4 + k : WT / 4 - 1];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = 4 * (i4 + k);
                        if (act && i < np && g[k].x < bf) { bf = g[k].x; bi = i; }
                        if (act && i + 1 < np && g[k].y < bf) { bf = g[k].y; bi = i + 1; }
                        if (act && i + 2 < np && g[k].z < bf) { bf = g[k].z; bi = i + 2; }
                        if (act && i + 3 < np && g[k].w < bf) { bf = g[k].w; bi = i + 3; }
                    }
                }
            }
            if (act) {
                const unsigned long long nk = p + 1 < nw ? key_at(slot_of(p + 1)) : qhead;
                const uint32_t bound = (uint32_t)(nk >> 32);
                // an episode that goes idle keeps the smallest of its residuals in .w (no element in hand any more): the
                // commit rule needs it every sub-round until the episode commits, and the list does not change any more
                uint32_t cur = WIN_NONE, curi = 0xFFu, curv = bf;
                if (bi >= 0 && bf < bound) {
                    cur = pidx[pl * WT + bi];
                    curi = (uint32_t)bi;
                    pval[pl * WT + bi] = WIN_POPPED;
                    ++npop;
                }
                q1.x = cur;
                q1.y = (uint32_t)np | (1u << 8) | (npop << 9) | ((uint32_t)pl << 16) | (curi << 24);
                q1.z = mx;
                q1.w = curv;
                winq1[s] = q1;
            }
            {
                // still wants to pop and has room for six more discoveries (B's rule): stays in the list; positions keep
                // their order, and the write index never passes the row being read
                const bool keep = act && q1.x != WIN_NONE && np + 6 <= tcap;
                const unsigned long long kb = __ballot(keep);
                if (keep) actpos[nnext + __popcll(kb & lt_mask)] = (unsigned short)p;
                nnext += __popcll(kb);
            }
            WSTAT(ST_EXEC, (uint32_t)__popcll(__ballot(act)));
            WIN_FENCE();
            WTICK(3);
        }
        // again at once?  needs: nobody lost a claim (D), the front episode still in the list (it is the first entry: the list
        // is in window order) -- otherwise it went idle (E may commit) or its list is full (B flags the forced commit)
        if (squash_key != ~0ull || nnext == 0 || st_sub >= sub_limit) break;
        if (rdlane((uint32_t)actpos[0], 0) != 0u) break;
        ++st_sub;
        nact = nnext;
        }
        nact_carry = nnext;
        list_valid = true;                                       // until something below touches the window
        // ---- D: squash from the first episode that lost a discovery it had already used --------------------------------
        if (squash_key != ~0ull) {
            int sp = 0;
            for (int base = 0; base < nw; base += 64) {
                const int p = base + lane;
                const bool lt = p < nw && key_at(slot_of(p)) < squash_key;
                sp += __popcll(__ballot(lt));
            }
            if (sp < nw) squash_from(sp);
        }
        WCHECK(4);
        WTICK(4);
        // ---- E: commit the idle episodes at the front, in order (up to three bursts of 64 per sub-round: on tie plateaus
        //         every episode is a single pop and the whole window turns over) ---------------------------------------------
        // Three sub-rounds in four (config 2) have a front episode that is still descending: nothing can commit, and the
        // whole commit phase is skipped on the strength of one word of the front slot.
        bool front_ready = front_forced;
        if (!front_ready && nw > 0) {
            const u32x4 f1 = winq1[slot_of(0)];
            front_ready = rdlane((uint32_t)(((f1.y >> 8) & 1u) != 0u && f1.x == WIN_NONE), 0) != 0u;
        }
        for (int burst = 0; front_ready && burst < (max_bursts & 0xFF) && nw > 0 && !serial; ++burst) {
            const unsigned long long qh = burst == 0 ? qhead : (scount > 0 ? srun[shead].key : ~0ull);   // enqueues moved the run
            const int j = lane;
            const bool valid = j < nw;
            const int s = valid ? slot_of(j) : 0;
            u32x4 q0 = {0u, 0u, 0u, 0u}, q1 = {0u, 0u, 0u, 0u};
            if (valid) { q0 = winq0[s]; q1 = winq1[s]; }
            const bool st = valid && ((q1.y >> 8) & 1u);
            const int np = st ? (int)(q1.y & 0xFFu) : 0, pl = (int)((q1.y >> 16) & 0xFFu);
            bool idle = st && q1.x == WIN_NONE;
            // the front episode with a full list commits what it has done: the element it was about to pop goes back to
            // the queue like any other residual
            const bool forced = burst == 0 && front_forced && j == 0 && st && q1.x != WIN_NONE && np + 6 > tcap;
            if (forced) { pval[pl * WT + (int)(q1.y >> 24)] = q1.w; idle = true; }
            WIN_FENCE();
            const unsigned long long nb_ = ~__ballot(idle);
            const int m = nb_ ? __ffsll((long long)nb_) - 1 : 64;       // leading idle episodes
            // smallest residual, number of residuals: cached when the episode went idle (C); only the forced front episode,
            // whose element in hand went back into its list, is scanned
            uint32_t resmin = 0xFFFFFFFFu, pc = 0u;
            if (j < m && !forced) { resmin = q1.w; pc = (uint32_t)np - ((q1.y >> 9) & 0x7Fu); }
#ifdef ISX_FLOOD_CHECK
            const uint32_t c_resmin = resmin, c_pc = pc;
            resmin = 0xFFFFFFFFu; pc = 0u;
            const bool scan_me = j < m;
#else
            const bool scan_me = j < m && forced;
#endif
            {
                const u32x4 *pv4 = (const u32x4 *)&pval[pl * WT];
                for (int i4 = 0; __any(scan_me && 4 * i4 < np); ++i4) {
                    if (scan_me && 4 * i4 < np) {
                        const u32x4 v4 = pv4[i4];
                        const int i = 4 * i4;
                        if (v4.x != WIN_POPPED) { ++pc; if (v4.x < resmin) resmin = v4.x; }
                        if (i + 1 < np && v4.y != WIN_POPPED) { ++pc; if (v4.y < resmin) resmin = v4.y; }
                        if (i + 2 < np && v4.z != WIN_POPPED) { ++pc; if (v4.z < resmin) resmin = v4.z; }
                        if (i + 3 < np && v4.w != WIN_POPPED) { ++pc; if (v4.w < resmin) resmin = v4.w; }
                    }
                }
            }
#ifdef ISX_FLOOD_CHECK
            if (j < m && !forced && (c_resmin != resmin || c_pc != pc) && stats && stats[40] == 0ull) {
                stats[40] = 300ull; stats[41] = (unsigned long long)st_sub; stats[42] = (unsigned long long)nw; stats[43] = (unsigned long long)j;
                stats[44] = c_resmin; stats[45] = resmin; stats[46] = c_pc; stats[47] = pc;
            }
#endif
            const uint32_t maxnew = q1.z;                               // an idle episode has no element in hand; forced: excluded
            // residuals of the episodes committing before me must all sort after everything I popped (they are older
            // than my discoveries: strict for those; my root is older than any residual: ties go to the root)
            uint32_t F = wave_shr1(wave_prefix_min(j < m ? resmin : 0xFFFFFFFFu), 0xFFFFFFFFu);
            const uint32_t qv = (uint32_t)(qh >> 32);
            if (qv < F) F = qv;
            // ... and after the root behind me, which may be a residual that an earlier burst of this sub-round put there
            // (the bound check at the top of the sub-round has not seen it yet)
            uint32_t succv = 0xFFFFFFFFu;
            if (valid && j + 1 < nw) succv = winq0[slot_of(j + 1)].y;
            const bool ok = j < m && q0.y <= F && (maxnew == 0u || (maxnew < F && maxnew < succv));
            const unsigned long long nok = ~__ballot(ok);
            int ncommit = nok ? __ffsll((long long)nok) - 1 : 64;
            WTICK(5);
            if (ncommit > 0) {
                // residuals enter the queue in age order (a bucket whose appends arrive in key order drains as a FIFO):
                // they are gathered in `pend` first, episode by episode; the burst is cut where `pend` would overflow
                if (j >= ncommit) pc = 0u;
                uint32_t poff = wave_incl_sum(pc) - pc;
                {
                    const unsigned long long fits = __ballot(j >= ncommit || poff + pc <= (uint32_t)WPEND);
                    const int nf = (~fits) ? __ffsll((long long)~fits) - 1 : 64;
                    if (nf < ncommit) ncommit = nf;                    // >= 1: one list always fits
                }
                const bool cm = j < ncommit;
                if (!cm) pc = 0u;
                if (rdlane((uint32_t)forced, 0)) WSTAT(ST_FORCED, 1);
                const uint32_t np_incl = wave_incl_sum(cm ? (uint32_t)np : 0u);
                const uint32_t mybase = agebase + np_incl - (cm ? (uint32_t)np : 0u);
                const uint32_t tot = rdlane(np_incl, 63), ptotal = rdlane(wave_incl_sum(pc), 63);
                agebase += tot;
                WSTAT(ST_POPS, tot - ptotal + (uint32_t)ncommit);           // popped discoveries + the roots
                const uint32_t lab = q0.w;
#ifdef ISX_FLOOD_CHECK
                if (dbg_log) {      // commit order of the roots of this component (development build)
                    uint32_t *dl = dbg_log + comp_off[c];
                    const unsigned long long tot_ = comp_off[n_comp] + 1ull;
                    if (cm && dbg_count + (uint32_t)j + 1u < (uint32_t)csize) {
                        dl[dbg_count + (uint32_t)j] = q0.z;
                        dl[2ull * tot_ + dbg_count + (uint32_t)j] = (uint32_t)np | ((uint32_t)st_sub << 8);
                        for (int i = 0; i < np; ++i) {
                            const uint32_t o_ = dbg_ecount + np_incl - (uint32_t)np + (uint32_t)i;
                            if (o_ < (uint32_t)csize) dl[tot_ + o_] = pidx[pl * WT + i] | (pval[pl * WT + i] == WIN_POPPED ? 0x80000000u : 0u);
                        }
                    }
                    dbg_ecount += tot;
                    dbg_count += (uint32_t)ncommit;
                    if (lane == 0) dl[csize - 1ull] = dbg_count;
                }
#endif
                {
                    const u32x4 *pi4 = (const u32x4 *)&pidx[pl * WT], *pv4 = (const u32x4 *)&pval[pl * WT];
                    for (int i4 = 0; __any(cm && 4 * i4 < np); ++i4) {
                        if (cm && 4 * i4 < np) {
                            const u32x4 v4 = pi4[i4], f4 = pv4[i4];
                            const uint32_t vv[4] = {v4.x, v4.y, v4.z, v4.w}, ff[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int i = 4 * i4 + e;
                                if (i < np) {
#ifdef ISX_FLOOD_CHECK
                                    {   // a committing episode must still hold every voxel it discovered
                                        const unsigned long long mykey_ = ((unsigned long long)q0.y << 32) | q0.x;
                                        const unsigned long long cur_ = __hip_atomic_load(&state[vv[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        if (cur_ != mykey_ && stats && stats[40] == 0ull) {
                                            stats[40] = 200ull; stats[41] = (unsigned long long)st_sub; stats[42] = (unsigned long long)nw; stats[43] = (unsigned long long)j;
                                            stats[44] = mykey_; stats[45] = cur_; stats[46] = (unsigned long long)vv[e]; stats[47] = (unsigned long long)i;
                                        }
                                    }
#endif
                                    __hip_atomic_store(&state[vv[e]], (unsigned long long)lab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    out[vv[e]] = lab;                    // label at discovery (watershed.py:153)
                                    if (ff[e] != WIN_POPPED) {
                                        HeapEnt r;
                                        r.key = ((unsigned long long)ff[e] << 32) | (unsigned long long)(mybase + (uint32_t)i + 1u);
                                        r.idx = vv[e];
                                        r.pad = lab;
                                        pend[poff++] = r;
                                    }
                                }
                            }
                        }
                    }
                }
                list_valid = false;
                if (cm) pfree[npfree + j] = (unsigned short)pl;
                npfree += ncommit;
                wh = slot_of(ncommit);
                nw -= ncommit;
                WIN_FENCE();
                WTICK(6);
                for (uint32_t b0 = 0; b0 < ptotal; b0 += 64u) {
                    const bool has = b0 + (uint32_t)lane < ptotal;
                    HeapEnt r; r.key = 0; r.idx = 0; r.pad = 0;
                    if (has) r = pend[b0 + lane];
                    enqueue64(has, r.key, r.idx, r.pad);         // once `serial` is set this spills: every residual must survive
                    if (!serial) WCHECK(50 + burst);
                }
            }
            WCHECK(10 + burst);
            if (serial) break;
            if (ncommit < 64) break;
        }
        WTICK(7);
        if (serial) break;
        // ---- G: refill the window from the run (from the buckets when the run is empty) ----------------------------------
        // invariant for the bounds above: the run is empty only if nothing is queued behind the window
        {
            if (scount == 0 && cold.on && (cold.count > 0 || nstage > 0)) {
                if (nstage > 0) { flush_stage(nstage); if (serial) break; }
                if (cold.count > 0) { scount = cold_drain(cold, srun, hist, lane, drain_n); shead = 0; WSTAT(ST_DRAIN, 1); WCHECK(40); }
            }
            while (nw < wcap && scount > 0) {
                int take = wcap - nw;
                if (take > scount) take = scount;
                if (take > 64) take = 64;
                if (take > npfree) take = npfree;
                if (take <= 0) break;
                if (lane < take) {
                    const u32x4 e = run_ld(srun, (shead + lane) & SMASK);
                    const int pl = (int)pfree[npfree - 1 - lane];
                    const int s = slot_of(nw + lane);
                    winq0[s] = e;
                    winq1[s] = fresh_q1(e.z, pl);
                }
                list_valid = false;
                npfree -= take;
                nw += take;
                shead = (shead + take) & SMASK;
                scount -= take;
                WIN_FENCE();
                if (scount == 0 && cold.on && (cold.count > 0 || nstage > 0)) {
                    if (nstage > 0) { flush_stage(nstage); if (serial) break; }
                    if (cold.count > 0) { scount = cold_drain(cold, srun, hist, lane, drain_n); shead = 0; WSTAT(ST_DRAIN, 1); WCHECK(41); }
                }
            }
            if (serial) break;
        }
        WCHECK(30);
        WTICK(8);
    }

#if defined(ISX_FLOOD_PROF) && !defined(ISX_FLOOD_PROF_GROUP_ONLY)      // (the work-group kernel's timers share these slots: -DISX_FLOOD_PROF_GROUP_ONLY for a volume that uses both)
    if (stats && lane == 0)
        for (int i = 0; i < 10; ++i) { atomicAdd(&stats[32 + i], tk_acc[i]); if (i < 4) atomicAdd(&stats[44 + i], cold.tk[i]); }
#endif
    if (stats && lane == 0) {
        atomicAdd(&stats[0], (unsigned long long)st_sub);
        const int gidx[ST_N] = {0, 1, 2, 3, 4, 5, 6, 7, 11, 12, 13, 14, 15, 19};
        for (int i = 1; i < ST_N; ++i) atomicAdd(&stats[gidx[i]], (unsigned long long)lstat[i]);
    }
    if (serial) {
        // Exact continuation on one lane: every queued element (window roots -- their speculative work is dropped,
        // `out` only ever holds committed labels --, run, staged, buckets, spilled) goes into the binary heap.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t hn = 0;
        if (!started_any) {
            if (lane == 0)
                for (uint32_t j = 0; j < ns; ++j) {
                    HeapEnt e; e.key = ZKEY | (unsigned long long)j; e.idx = (uint32_t)(keys[first + j] & 0xFFFFFFFFull); e.pad = 0;
                    heap_push(heap, hn, e);
                }
        } else {
            if (lane == 0) {
                for (int p = 0; p < nw; ++p) {
                    const u32x4 q0 = winq0[slot_of(p)];
                    HeapEnt e; e.key = ((unsigned long long)q0.y << 32) | q0.x; e.idx = q0.z; e.pad = 0;
                    heap_push(heap, hn, e);
                }
                for (int j = 0; j < scount; ++j) { HeapEnt e = srun[(shead + j) & SMASK]; e.pad = 0; heap_push(heap, hn, e); }
                for (int j = 0; j < nstage; ++j) { HeapEnt e = stage[j]; e.pad = 0; heap_push(heap, hn, e); }
            }
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                for (uint32_t j = 0; j < nspill; ++j) {            // lowest address first: the growing heap may only reach slots already read
                    HeapEnt e = cld_ent(&heap[csize - (unsigned long long)nspill + j]); e.pad = 0;
                    heap_push(heap, hn, e);
                }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (lane == 0) serial_flood(heap, hn, agebase, aff, mask, out, PY, PX, PN, s0, s1, s2);
    }
}
