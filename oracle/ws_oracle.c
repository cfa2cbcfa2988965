/*
 * ws_oracle.c -- CPU restatement of the reference's watershed post-processor (W0-W4).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under iterseg_amd/ may call, link or import this;
 * it is the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Each function cites the reference (paths relative to /root/reference) or, where the
 * reference delegates to third-party code that is not vendored there, the library
 * routine whose published algorithm is restated (scikit-image 0.18.3 / scipy 1.7.1 /
 * numpy 1.26.4 -- the versions the golden vectors under tests/golden/ were produced
 * with; see DESIGN.md "Oracle").  Pinned against every fixture in tests/golden/ws_*.npz
 * by tests/test_oracle_watershed.py.
 *
 * Plain C99, scalar, single-threaded.  Build: see oracle/Makefile (-ffp-contract=off so
 * that double arithmetic rounds exactly like the generic x86-64 builds of scipy/numpy).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * Gaussian kernels.  scipy.ndimage._gaussian_kernel1d(sigma, 0, radius) evaluated with
 * numpy 1.26.4 (np.exp is not bit-stable across numpy releases, so the table is data,
 * not code).  w[0] is the centre tap, w[i] the taps at distance i.
 * skimage.filters.gaussian -> ndi.gaussian_filter(mode='nearest', truncate=4.0):
 * radius = int(4*sigma + 0.5)  (reference call sites: watershed.py:227, :234).
 * ---------------------------------------------------------------------------------- */
static const double GK_SIGMA2[9] = {
    0x1.98862a07ae7b4p-3, 0x1.68856f9ab1982p-3, 0x1.ef9093fc46e5ap-4, 0x1.0941b71ceef37p-4,
    0x1.ba4d4125ffd29p-6, 0x1.1f30504e20207p-7, 0x1.227362b5fc92ep-9, 0x1.c98b8c5d0dda5p-12,
    0x1.18aad19e4159bp-14};
static const double GK_SIGMA1[5] = {
    0x1.9884a307594fbp-2, 0x1.ef8eb9ad499bap-3, 0x1.ba4b99d1799a9p-5, 0x1.22724cb7eb26ap-8,
    0x1.18a9c4fd536c6p-13};

const double *orc_gauss_weights(int sigma_int, int *radius)
{
    if (sigma_int == 2) { *radius = 8; return GK_SIGMA2; }
    *radius = 4;
    return GK_SIGMA1;
}

/* One separable pass along `axis` of a C-contiguous (d0,d1,d2) float32 array.
 * Restates scipy NI_Correlate1D's symmetric branch: lines are converted to double,
 *   tmp = line[l]*w0;  for i = R..1:  tmp += (line[l-i] + line[l+i]) * w[i];
 * (farthest taps first), then the double is stored as float32.  Border: 'nearest'.
 * in == out is allowed (a line buffer is used, as scipy does). */
void orc_gaussian_pass(const float *in, float *out, const int64_t dims[3], int axis,
                       const double *w, int radius)
{
    int64_t len = dims[axis];
    int64_t stride = 1;
    for (int a = 2; a > axis; --a) stride *= dims[a];
    int64_t nlines = dims[0] * dims[1] * dims[2] / len;
    double *line = (double *)malloc(sizeof(double) * (size_t)(len + 2 * radius));
    for (int64_t ln = 0; ln < nlines; ++ln) {
        /* base offset of line `ln`: enumerate all index tuples with axis removed */
        int64_t inner = ln % stride, outer = ln / stride;
        int64_t base = outer * stride * len + inner;
        for (int64_t i = 0; i < len; ++i) line[radius + i] = (double)in[base + i * stride];
        for (int i = 0; i < radius; ++i) {
            line[i] = line[radius];
            line[radius + len + i] = line[radius + len - 1];
        }
        for (int64_t l = 0; l < len; ++l) {
            const double *c = line + radius + l;
            double tmp = c[0] * w[0];
            for (int i = radius; i >= 1; --i) tmp += (c[-i] + c[i]) * w[i];
            out[base + l * stride] = (float)tmp;
        }
    }
    free(line);
}

/* skimage.filters.gaussian(img, sigma) for float32 input: passes over the axes with
 * sigma > 0 in axis order, each pass stored as float32 (ndi.gaussian_filter). */
void orc_gaussian(const float *in, float *out, const int64_t dims[3], const int sigma_int[3])
{
    const float *src = in;
    int did = 0;
    for (int a = 0; a < 3; ++a) {
        if (sigma_int[a] <= 0) continue;
        int r;
        const double *w = orc_gauss_weights(sigma_int[a], &r);
        orc_gaussian_pass(src, out, dims, a, w, r);
        src = out;
        did = 1;
    }
    if (!did) memcpy(out, in, sizeof(float) * (size_t)(dims[0] * dims[1] * dims[2]));
}

/* ------------------------------------------------------------------------------------
 * Otsu threshold: skimage.filters.threshold_otsu (thresholding.py:326-350) on a float32
 * image = np.histogram(256 bins over [min,max]) + between-class variance argmax.
 * numpy.lib.histograms (1.26.4) :: uniform-bin fast path restated literally:
 *   edges  = float32( float64 linspace(min, max, 257) )
 *   f_idx  = float32((a - min) / (max - min)) * 256 -> trunc; 256 -> 255;
 *            then one decrement if a < edges[idx], one increment if a >= edges[idx+1].
 * ---------------------------------------------------------------------------------- */
void orc_minmax(const float *a, int64_t n, float *mn, float *mx)
{
    float lo = a[0], hi = a[0];
    for (int64_t i = 1; i < n; ++i) {
        if (a[i] < lo) lo = a[i];
        if (a[i] > hi) hi = a[i];
    }
    *mn = lo;
    *mx = hi;
}

void orc_hist_edges(float mn, float mx, float edges[257])
{
    double start = (double)mn, stop = (double)mx;
    double step = (stop - start) / 256.0;
    for (int i = 0; i < 257; ++i) {
        double y = (double)i * step;
        y = y + start;
        edges[i] = (float)y;
    }
    edges[256] = mx;
}

void orc_histogram256(const float *a, int64_t n, float mn, float mx, const float edges[257],
                      int64_t counts[256])
{
    memset(counts, 0, sizeof(int64_t) * 256);
    float denom = mx - mn;
    for (int64_t i = 0; i < n; ++i) {
        float v = a[i];
        float f = (v - mn) / denom;
        f = f * 256.0f;
        int64_t idx = (int64_t)f;
        if (idx == 256) idx = 255;
        if (v < edges[idx]) idx -= 1;
        if (v >= edges[idx + 1] && idx != 255) idx += 1;
        counts[idx] += 1;
    }
}

float orc_otsu_from_hist(const int64_t counts[256], const float edges[257])
{
    float centers[256];
    double cw[256], w1[256], w2[256], m1[256], m2[256], t[256];
    for (int i = 0; i < 256; ++i) {
        float s = edges[i] + edges[i + 1];
        centers[i] = s / 2.0f;
        cw[i] = (double)counts[i] * (double)centers[i];
    }
    double acc = 0.0;
    for (int i = 0; i < 256; ++i) { acc += (double)counts[i]; w1[i] = acc; }
    acc = 0.0;
    for (int i = 255; i >= 0; --i) { acc += (double)counts[i]; w2[i] = acc; }
    acc = 0.0;
    for (int i = 0; i < 256; ++i) { acc += cw[i]; m1[i] = acc / w1[i]; }
    acc = 0.0;
    for (int i = 255; i >= 0; --i) { acc += cw[i]; t[i] = acc; }
    for (int i = 0; i < 256; ++i) m2[i] = t[i] / w2[i];
    int best = 0;
    double bestv = -1.0;
    for (int i = 0; i < 255; ++i) {
        double d = m1[i] - m2[i + 1];
        double v = (w1[i] * w2[i + 1]) * (d * d);
        if (i == 0 || v > bestv) { bestv = v; best = i; }   /* np.argmax: first maximum */
    }
    return centers[best];
}

/* threshold_otsu incl. the single-intensity early-out (thresholding.py:328-331). */
float orc_threshold_otsu(const float *a, int64_t n)
{
    float mn, mx;
    orc_minmax(a, n, &mn, &mx);
    if (mn == mx) return a[0];
    float edges[257];
    int64_t counts[256];
    orc_hist_edges(mn, mx, edges);
    orc_histogram256(a, n, mn, mx, edges, counts);
    return orc_otsu_from_hist(counts, edges);
}

/* _get_mask (watershed.py:226-229): threshold from the smoothed image, applied to the raw one. */
float orc_get_mask(const float *img, const int64_t dims[3], uint8_t *mask, float *smooth_out)
{
    int64_t n = dims[0] * dims[1] * dims[2];
    float *sm = smooth_out ? smooth_out : (float *)malloc(sizeof(float) * (size_t)n);
    const int sig[3] = {2, 2, 2};
    orc_gaussian(img, sm, dims, sig);
    float th = orc_threshold_otsu(sm, n);
    for (int64_t i = 0; i < n; ++i) mask[i] = img[i] > th;
    if (!smooth_out) free(sm);
    return th;
}

/* ------------------------------------------------------------------------------------
 * _get_centroids (watershed.py:232-236): gaussian(sigma=(0,1,1)) then
 * skimage.feature.peak_local_max(threshold_abs=0.04) with defaults (peak.py 0.18.3):
 *   peak  = (g == maximum_filter(g, 3x3x3, mode='constant', cval=0)) ; all-equal image -> none
 *   peak &= g > float32(0.04) ; a 1-voxel border on every axis is cleared
 *   order = descending intensity; ties are implementation-defined upstream (unstable
 *           argsort) -- here: ascending raveled index.  ensure_spacing(1) is a no-op.
 * Returns the number of peaks; coords are raveled indices into `g`.
 * ---------------------------------------------------------------------------------- */
typedef struct { float v; int64_t idx; } orc_peak_t;
static int peak_cmp(const void *pa, const void *pb)
{
    const orc_peak_t *a = (const orc_peak_t *)pa, *b = (const orc_peak_t *)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

int64_t orc_peak_local_max(const float *g, const int64_t dims[3], float threshold,
                           int64_t *out_idx, int64_t cap)
{
    int64_t Z = dims[0], Y = dims[1], X = dims[2], n = Z * Y * X;
    uint8_t *pk = (uint8_t *)malloc((size_t)n);
    int64_t all = 1;
    for (int64_t z = 0; z < Z; ++z)
        for (int64_t y = 0; y < Y; ++y)
            for (int64_t x = 0; x < X; ++x) {
                float m = -INFINITY;
                for (int dz = -1; dz <= 1; ++dz)
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            int64_t zz = z + dz, yy = y + dy, xx = x + dx;
                            float v = 0.0f; /* cval */
                            if (zz >= 0 && zz < Z && yy >= 0 && yy < Y && xx >= 0 && xx < X)
                                v = g[(zz * Y + yy) * X + xx];
                            if (v > m) m = v;
                        }
                int64_t i = (z * Y + y) * X + x;
                pk[i] = (g[i] == m);
                all &= pk[i];
            }
    int64_t cnt = 0;
    orc_peak_t *lst = (orc_peak_t *)malloc(sizeof(orc_peak_t) * (size_t)(n > 0 ? n : 1));
    if (!all) {
        for (int64_t z = 1; z < Z - 1; ++z)
            for (int64_t y = 1; y < Y - 1; ++y)
                for (int64_t x = 1; x < X - 1; ++x) {
                    int64_t i = (z * Y + y) * X + x;
                    if (pk[i] && g[i] > threshold) { lst[cnt].v = g[i]; lst[cnt].idx = i; ++cnt; }
                }
    }
    qsort(lst, (size_t)cnt, sizeof(orc_peak_t), peak_cmp);
    int64_t m = cnt < cap ? cnt : cap;
    for (int64_t i = 0; i < m; ++i) out_idx[i] = lst[i].idx;
    free(lst);
    free(pk);
    return cnt;
}

/* ------------------------------------------------------------------------------------
 * _remove_unwanted_objects (watershed.py:239-251): ndi.label (6-connectivity) ->
 * remove_small_objects(min) -> XOR with remove_small_objects(max): keep components with
 * min_area <= size < max_area.  comp (optional) receives scan-order component ids.
 * ---------------------------------------------------------------------------------- */
int64_t orc_label_filter(const uint8_t *mask, const int64_t dims[3], int64_t min_area,
                         int64_t max_area, uint8_t *mask_out, int32_t *comp)
{
    int64_t Z = dims[0], Y = dims[1], X = dims[2], n = Z * Y * X;
    int32_t *lab = comp ? comp : (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int64_t *queue = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    memset(lab, 0, sizeof(int32_t) * (size_t)n);
    memset(mask_out, 0, (size_t)n);
    const int64_t off[6] = {-Y * X, -X, -1, 1, X, Y * X};
    int32_t ncomp = 0;
    for (int64_t s = 0; s < n; ++s) {
        if (!mask[s] || lab[s]) continue;
        ++ncomp;
        int64_t head = 0, tail = 0;
        queue[tail++] = s;
        lab[s] = ncomp;
        while (head < tail) {
            int64_t p = queue[head++];
            int64_t z = p / (Y * X), y = (p / X) % Y, x = p % X;
            const int ok[6] = {z > 0, y > 0, x > 0, x < X - 1, y < Y - 1, z < Z - 1};
            for (int k = 0; k < 6; ++k) {
                if (!ok[k]) continue;
                int64_t q = p + off[k];
                if (mask[q] && !lab[q]) { lab[q] = ncomp; queue[tail++] = q; }
            }
        }
        if (tail >= min_area && tail < max_area)
            for (int64_t i = 0; i < tail; ++i) mask_out[queue[i]] = 1;
    }
    free(queue);
    if (!comp) free(lab);
    return ncomp;
}

/* ------------------------------------------------------------------------------------
 * raveled_affinity_watershed (watershed.py:95-159) with _prep_data's seeding (:61-62).
 * Heap of (value f32, age, index); Python tuple order = value, then age, then index
 * (source never decides: indices are unique).  Neighbour table (watershed.py:84-92 +
 * skimage _offsets_to_raveled_neighbors): [(-SZ, z), (-SY, y), (-1, x), (+1, x),
 * (+SY, y), (+SZ, z)]; the edge value is aff[axis][idx] for the three negative
 * directions and aff[axis][neighbour] for the positive ones (:119-120, :154-155).
 * A voxel is labelled when first discovered (:153).  No bounds checks: the mask's
 * zero border keeps the flood inside (:213).
 * aff: (3, N) float32, N = dims product.  out: N uint32 (int32 callers share the bits).
 * Returns pops performed, or -1 when there are no seeds (the reference raises).
 * ---------------------------------------------------------------------------------- */
typedef struct { float v; int64_t age; int64_t idx; } orc_elem_t;
static inline int elem_less(const orc_elem_t *a, const orc_elem_t *b)
{
    if (a->v < b->v) return 1;
    if (a->v > b->v) return 0;
    if (a->age != b->age) return a->age < b->age;
    return a->idx < b->idx;
}

/* the heap loop (watershed.py:127-158) over the given seeds, whose labels are already in `out` */
static int64_t *g_trace = NULL; static int64_t g_trace_cap = 0, g_trace_n = 0;
void orc_set_trace(int64_t *buf, int64_t cap) { g_trace = buf; g_trace_cap = cap; g_trace_n = 0; }
int64_t orc_trace_count(void) { return g_trace_n; }

static int64_t flood_core(const float *aff, const int64_t *seeds, int64_t nseeds,
                          const uint8_t *mask, uint32_t *out, const int64_t dims[3])
{
    int64_t N = dims[0] * dims[1] * dims[2];
    int64_t SZ = dims[1] * dims[2], SY = dims[2];
    const int64_t noff[6] = {-SZ, -SY, -1, 1, SY, SZ};
    const int ax[6] = {0, 1, 2, 2, 1, 0};
    int64_t cap = nseeds + 16, hn = 0;
    orc_elem_t *heap = (orc_elem_t *)malloc(sizeof(orc_elem_t) * (size_t)cap);
#define HPUSH(E)                                                                   \
    do {                                                                           \
        if (hn == cap) { cap *= 2; heap = (orc_elem_t *)realloc(heap, sizeof(orc_elem_t) * (size_t)cap); } \
        int64_t c_ = hn++;                                                         \
        while (c_ > 0) {                                                           \
            int64_t p_ = (c_ - 1) >> 1;                                            \
            if (!elem_less(&(E), &heap[p_])) break;                                \
            heap[c_] = heap[p_];                                                   \
            c_ = p_;                                                               \
        }                                                                          \
        heap[c_] = (E);                                                            \
    } while (0)
    int64_t age = 0, pops = 0;
    for (int64_t i = 0; i < nseeds; ++i) {
        orc_elem_t e = {0.0f, 0, seeds[i]};
        HPUSH(e);
    }
    while (hn > 0) {
        orc_elem_t top = heap[0];
        orc_elem_t last = heap[--hn];
        if (hn > 0) {
            int64_t c = 0;
            for (;;) {
                int64_t l = 2 * c + 1, r = l + 1, m;
                if (l >= hn) break;
                m = (r < hn && elem_less(&heap[r], &heap[l])) ? r : l;
                if (!elem_less(&heap[m], &last)) break;
                heap[c] = heap[m];
                c = m;
            }
            heap[c] = last;
        }
        ++pops;
        if (g_trace && g_trace_n < g_trace_cap) g_trace[g_trace_n++] = top.idx;      /* test hook: pop order (single-threaded runs only) */
        for (int k = 0; k < 6; ++k) {
            int64_t nb = top.idx + noff[k];
            if (!mask[nb] || out[nb]) continue;
            out[nb] = out[top.idx];
            int64_t src = (k < 3) ? top.idx : nb;
            orc_elem_t e = {aff[(int64_t)ax[k] * N + src], ++age, nb};
            HPUSH(e);
        }
    }
#undef HPUSH
    free(heap);
    return pops;
}

int64_t orc_affinity_watershed(const float *aff, const int64_t *seeds, int64_t nseeds,
                               const uint8_t *mask, uint32_t *out, const int64_t dims[3])
{
    if (nseeds <= 0) return -1;
    for (int64_t i = 0; i < nseeds; ++i) out[seeds[i]] = (uint32_t)(i + 1);      /* _prep_data, watershed.py:61-62 */
    return flood_core(aff, seeds, nseeds, mask, out, dims);
}

/* ------------------------------------------------------------------------------------
 * The same flood, one 6-connected mask component at a time, components spread over
 * `nthreads` OpenMP threads.  Exact: the pop order inside a component does not depend on
 * the other components (pops are ordered by (value, age), ages are monotone in push order,
 * floods never cross mask == 0; SURVEY.md F4, checked by
 * tests/test_oracle_watershed.py::test_flood_decomposes_over_components), and seeds keep
 * their list order inside a component.  This is the CPU baseline bench.py times beside
 * the GPU path ("W4 single-threaded per component + component-parallel").
 * ---------------------------------------------------------------------------------- */
typedef struct { int64_t comp, order; } seed_key_t;
static int seed_key_cmp(const void *a, const void *b)
{
    const seed_key_t *x = (const seed_key_t *)a, *y = (const seed_key_t *)b;
    if (x->comp != y->comp) return x->comp < y->comp ? -1 : 1;
    return x->order < y->order ? -1 : (x->order > y->order ? 1 : 0);
}

int64_t orc_affinity_watershed_mt(const float *aff, const int64_t *seeds, int64_t nseeds,
                                  const uint8_t *mask, uint32_t *out, const int64_t dims[3], int nthreads)
{
    if (nseeds <= 0) return -1;
    /* A seed OFF the mask is still popped by the reference loop and labels its in-mask neighbours (watershed.py:139-153),
     * and it can join two components: such marker lists do not decompose, they take the single heap.  (The plugin path
     * never produces one: _remove_unwanted_objects keeps only seeds inside surviving components, watershed.py:247-250.) */
    for (int64_t i = 0; i < nseeds; ++i)
        if (!mask[seeds[i]]) return orc_affinity_watershed(aff, seeds, nseeds, mask, out, dims);
    int64_t N = dims[0] * dims[1] * dims[2];
    int64_t SZ = dims[1] * dims[2], SY = dims[2];
    const int64_t noff[6] = {-SZ, -SY, -1, 1, SY, SZ};
    for (int64_t i = 0; i < nseeds; ++i) out[seeds[i]] = (uint32_t)(i + 1);
    int64_t *saved_trace = g_trace;
    g_trace = NULL;                      /* the trace hook is a process-global: not from inside the parallel region */
    /* component id of every seed: flood fill of the mask from each not yet visited seed */
    int32_t *comp = (int32_t *)calloc((size_t)N, sizeof(int32_t));
    int64_t *stack = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    seed_key_t *keys = (seed_key_t *)malloc(sizeof(seed_key_t) * (size_t)nseeds);
    int32_t ncomp = 0;
    for (int64_t i = 0; i < nseeds; ++i) {
        int64_t s0 = seeds[i];
        if (!comp[s0]) {
            ++ncomp;
            int64_t sp = 0;
            stack[sp++] = s0;
            comp[s0] = ncomp;
            while (sp > 0) {
                int64_t v = stack[--sp];
                for (int k = 0; k < 6; ++k) {
                    int64_t nb = v + noff[k];
                    if (nb < 0 || nb >= N || !mask[nb] || comp[nb]) continue;
                    comp[nb] = ncomp;
                    stack[sp++] = nb;
                }
            }
        }
        keys[i].comp = comp[s0];
        keys[i].order = i;
    }
    free(stack);
    free(comp);
    qsort(keys, (size_t)nseeds, sizeof(seed_key_t), seed_key_cmp);
    /* runs of equal component id */
    int64_t *first = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nseeds + 1));
    int64_t *sorted = (int64_t *)malloc(sizeof(int64_t) * (size_t)nseeds);
    int64_t nrun = 0;
    for (int64_t i = 0; i < nseeds; ++i) {
        sorted[i] = seeds[keys[i].order];
        if (i == 0 || keys[i].comp != keys[i - 1].comp) first[nrun++] = i;
    }
    first[nrun] = nseeds;
    int64_t pops = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : pops)
#endif
    for (int64_t r = 0; r < nrun; ++r) {
        pops += flood_core(aff, sorted + first[r], first[r + 1] - first[r], mask, out, dims);
    }
    (void)nthreads;
    g_trace = saved_trace;
    free(first);
    free(sorted);
    free(keys);
    return pops;
}

/* ------------------------------------------------------------------------------------
 * segment_output_image (watershed.py:165-223), plugin-path argument set
 * (segmentation.py:190-194): fm is (5, Z, Y, X); channels 0..2 affinities, 3 mask, 4 centres.
 * labels_padded: (Z+2, Y+2, X+2) uint32, must be zero on entry (segment_single_volume
 * allocates it so, segmentation.py:890-895).  Optional outputs may be NULL.
 * Returns number of surviving seeds (0 => reference would raise; labels stay zero).
 * ---------------------------------------------------------------------------------- */
void orc_normalise_pad(const float *fm3, const int64_t dims[3], float *aff_padded)
{
    int64_t Z = dims[0], Y = dims[1], X = dims[2], n = Z * Y * X;
    int64_t PY = Y + 2, PX = X + 2, PN = (Z + 2) * PY * PX;
    memset(aff_padded, 0, sizeof(float) * (size_t)(3 * PN));
    for (int c = 0; c < 3; ++c) {
        const float *src = fm3 + c * n;
        float mx = src[0];
        for (int64_t i = 1; i < n; ++i) if (src[i] > mx) mx = src[i];
        for (int64_t z = 0; z < Z; ++z)
            for (int64_t y = 0; y < Y; ++y)
                for (int64_t x = 0; x < X; ++x)
                    aff_padded[c * PN + ((z + 1) * PY + (y + 1)) * PX + (x + 1)] =
                        src[(z * Y + y) * X + x] / mx;
    }
}

int64_t orc_segment_output_image_mt(const float *fm, const int64_t dims[3], uint32_t *labels_padded,
                                    float *thresh_out, int64_t *seeds_out, int64_t seeds_cap,
                                    uint8_t *mask_out_padded, int64_t min_area, int64_t max_area, int nthreads);

int64_t orc_segment_output_image(const float *fm, const int64_t dims[3], uint32_t *labels_padded,
                                 float *thresh_out, int64_t *seeds_out, int64_t seeds_cap,
                                 uint8_t *mask_out_padded, int64_t min_area, int64_t max_area)
{
    return orc_segment_output_image_mt(fm, dims, labels_padded, thresh_out, seeds_out, seeds_cap, mask_out_padded,
                                       min_area, max_area, 0);
}

/* nthreads = 0: the flood as one heap over the whole volume (the reference's loop, verbatim);
 * nthreads >= 1: the same flood component by component on that many threads */
int64_t orc_segment_output_image_mt(const float *fm, const int64_t dims[3], uint32_t *labels_padded,
                                    float *thresh_out, int64_t *seeds_out, int64_t seeds_cap,
                                    uint8_t *mask_out_padded, int64_t min_area, int64_t max_area, int nthreads)
{
    int64_t Z = dims[0], Y = dims[1], X = dims[2], n = Z * Y * X;
    int64_t pd[3] = {Z + 2, Y + 2, X + 2};
    int64_t PN = pd[0] * pd[1] * pd[2];
    float *aff = (float *)malloc(sizeof(float) * (size_t)(3 * PN));
    orc_normalise_pad(fm, dims, aff);
    /* seeds */
    float *g = (float *)malloc(sizeof(float) * (size_t)n);
    const int sig[3] = {0, 1, 1};
    orc_gaussian(fm + 4 * n, g, dims, sig);
    int64_t *pidx = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t npk = orc_peak_local_max(g, dims, 0.04f, pidx, n);
    free(g);
    /* mask */
    uint8_t *m = (uint8_t *)malloc((size_t)n);
    float th = orc_get_mask(fm + 3 * n, dims, m, NULL);
    if (thresh_out) *thresh_out = th;
    uint8_t *mp = (uint8_t *)calloc((size_t)PN, 1);
    for (int64_t z = 0; z < Z; ++z)
        for (int64_t y = 0; y < Y; ++y)
            memcpy(mp + ((z + 1) * pd[1] + (y + 1)) * pd[2] + 1, m + (z * Y + y) * X, (size_t)X);
    free(m);
    uint8_t *mf = mask_out_padded ? mask_out_padded : (uint8_t *)malloc((size_t)PN);
    orc_label_filter(mp, pd, min_area, max_area, mf, NULL);
    free(mp);
    /* keep seeds inside surviving components, order preserved, padded raveled index */
    int64_t *sd = (int64_t *)malloc(sizeof(int64_t) * (size_t)(npk > 0 ? npk : 1));
    int64_t ns = 0;
    for (int64_t i = 0; i < npk; ++i) {
        int64_t z = pidx[i] / (Y * X), y = (pidx[i] / X) % Y, x = pidx[i] % X;
        int64_t p = ((z + 1) * pd[1] + (y + 1)) * pd[2] + (x + 1);
        if (mf[p]) sd[ns++] = p;
    }
    free(pidx);
    if (seeds_out) for (int64_t i = 0; i < ns && i < seeds_cap; ++i) seeds_out[i] = sd[i];
    if (ns > 0) {
        if (nthreads > 0) orc_affinity_watershed_mt(aff, sd, ns, mf, labels_padded, pd, nthreads);
        else orc_affinity_watershed(aff, sd, ns, mf, labels_padded, pd);
    }
    free(sd);
    free(aff);
    if (!mask_out_padded) free(mf);
    return ns;
}
