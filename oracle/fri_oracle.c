/*
 * fri_oracle.c -- CPU restatement of libfri's hot path (address map, residue transform,
 * quantiser, neighbour gather, bucket/prediction, ANS symbol histogram, inverse transform,
 * symbol order).
 *
 * TEST INFRASTRUCTURE ONLY -- see fri_oracle.h.  PARITY UNPINNED (no reference golden vectors,
 * reference not buildable here); pinned by source citations + SURVEY.md section 8c KATs.
 *
 * The data structures deliberately keep the reference's shape (a lattice of Fractal objects with
 * per-level position maps and a global position map of hash tables) rather than the dense tables
 * the HIP product uses, so that the two implementations are independent readings.
 *
 * Citations are relative to /root/reference/crates/libfri/src/.
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (see Makefile).
 */
#include "fri_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t re, im;
} cpx;

static inline cpx cadd(cpx a, cpx b) { return (cpx){a.re + b.re, a.im + b.im}; }
static inline cpx csub(cpx a, cpx b) { return (cpx){a.re - b.re, a.im - b.im}; }
static inline cpx cneg(cpx a) { return (cpx){-a.re, -a.im}; }
static inline int ceq(cpx a, cpx b) { return a.re == b.re && a.im == b.im; }

/* fractal.rs:51-86 -- the normative digit table of the complex-base address map. */
static const cpx LITERALS[30] = {
    {0, 1},        {-1, 1},       {2, 0},       {-3, -1},      {5, -1},        {1, 3},
    {-11, -1},     {9, -5},       {13, 7},      {-31, 3},      {5, -17},       {57, 11},
    {-67, 23},     {-47, -45},    {181, -1},    {-87, 91},     {-275, -89},    {449, -93},
    {101, 271},    {-999, -85},   {797, -457},  {1201, 627},   {-2795, 287},   {393, -1541},
    {5197, 967},   {-5983, 2115}, {-4411, -4049}, {16377, -181}, {-7555, 8279}, {-25199, -7917}};

#define BASE_FRAC_DEPTH 9 /* wavelet_transform.rs:39 */
#define NODES 512         /* 1 << depth */
#define CONTEXT_AMOUNT 10 /* prediction.rs:15 */
#define ALPHABET_SIZE 1024 /* entropy_coding.rs:25 */

/* ------------------------------------------------------------------ */
/* HashMap<Complex<i32>, V> stand-in: open addressing, V = 2 x int32.  */
/* ------------------------------------------------------------------ */
typedef struct {
    cpx *keys;
    cpx *vals;
    uint8_t *used;
    size_t cap; /* power of two */
    size_t len;
} cmap;

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
static inline size_t chash(cpx k) { return (size_t)mix64(((uint64_t)(uint32_t)k.re << 32) | (uint32_t)k.im); }

static void cmap_init(cmap *m, size_t cap_hint) {
    size_t cap = 8;
    while (cap < cap_hint * 2) cap <<= 1;
    m->keys = (cpx *)malloc(cap * sizeof(cpx));
    m->vals = (cpx *)malloc(cap * sizeof(cpx));
    m->used = (uint8_t *)calloc(cap, 1);
    m->cap = cap;
    m->len = 0;
}
static void cmap_free(cmap *m) {
    free(m->keys);
    free(m->vals);
    free(m->used);
    memset(m, 0, sizeof(*m));
}
static void cmap_insert(cmap *m, cpx k, cpx v);
static void cmap_grow(cmap *m) {
    cmap n;
    cmap_init(&n, m->cap); /* doubles */
    for (size_t i = 0; i < m->cap; i++)
        if (m->used[i]) cmap_insert(&n, m->keys[i], m->vals[i]);
    cmap_free(m);
    *m = n;
}
static void cmap_insert(cmap *m, cpx k, cpx v) {
    if ((m->len + 1) * 2 > m->cap) cmap_grow(m);
    size_t i = chash(k) & (m->cap - 1);
    while (m->used[i]) {
        if (ceq(m->keys[i], k)) {
            m->vals[i] = v; /* HashMap::insert overwrites */
            return;
        }
        i = (i + 1) & (m->cap - 1);
    }
    m->used[i] = 1;
    m->keys[i] = k;
    m->vals[i] = v;
    m->len++;
}
static const cpx *cmap_get(const cmap *m, cpx k) {
    if (!m->cap) return NULL;
    size_t i = chash(k) & (m->cap - 1);
    while (m->used[i]) {
        if (ceq(m->keys[i], k)) return &m->vals[i];
        i = (i + 1) & (m->cap - 1);
    }
    return NULL;
}
static inline int cmap_contains(const cmap *m, cpx k) { return cmap_get(m, k) != NULL; }

/* ------------------------------------------------------------------ */
/* Option<i32>                                                         */
/* ------------------------------------------------------------------ */
typedef struct {
    int32_t some;
    int32_t v;
} opt_i32;
static const opt_i32 NONE = {0, 0};
static inline opt_i32 some_i32(int32_t v) { return (opt_i32){1, v}; }

/* wavelet_transform.rs:14-26 */
typedef int32_t (*binop)(int32_t, int32_t);
static opt_i32 try_apply(opt_i32 first, opt_i32 second, binop operation, int32_t dflt) {
    if (first.some && second.some) return some_i32(operation(first.v, second.v));
    if (first.some) return some_i32(operation(first.v, dflt));
    if (second.some) return some_i32(operation(dflt, second.v));
    return NONE;
}
static int32_t op_diff(int32_t l, int32_t r) { return l - r; }        /* :212  |l, r| (l - r)     */
static int32_t op_lowpass(int32_t l, int32_t r) { return l + r / 2; } /* :216  |l, r| (l + r / 2) ; C '/' truncates like Rust */

/* ------------------------------------------------------------------ */
/* Fractal (wavelet_transform.rs:29-37)                                */
/* ------------------------------------------------------------------ */
typedef struct {
    uint8_t depth;
    cpx center;
    opt_i32 coefficients[3][NODES];
    uint8_t pred_bucket[3][NODES]; /* parameter_predictors.0 */
    int32_t pred_value[3][NODES];  /* parameter_predictors.1 */
    cmap position_map[BASE_FRAC_DEPTH];
    cpx image_positions[2 * NODES];
    int retained;
} fractal;

struct fri_oracle_wavelet {
    uint32_t height, width, channels;
    fractal **cells; /* all BFS cells */
    size_t n_cells, cap_cells;
    cmap fractal_lattice;                      /* centre -> (index into cells, 0); only retained after retain() */
    cmap global_position_map[BASE_FRAC_DEPTH]; /* position -> centre */
    uint32_t *order;                           /* retained cell indices in canonical order */
    uint32_t n_retained;
    const uint8_t *data; /* borrowed during from_raster only */
};

/* Fractal::new, wavelet_transform.rs:42-69 */
static fractal *fractal_new(uint8_t depth, cpx center) {
    fractal *f = (fractal *)calloc(1, sizeof(fractal));
    f->depth = depth;
    f->center = center;
    for (int l = 0; l < depth; l++) cmap_init(&f->position_map[l], (size_t)1 << l);
    f->image_positions[0] = center;
    f->image_positions[1] = center;
    for (int level = 0; level < depth; level++) {
        for (int pos = 1 << level; pos < 1 << (level + 1); pos++) {
            cmap_insert(&f->position_map[level], f->image_positions[pos], (cpx){pos, 0});
            f->image_positions[2 * pos] = f->image_positions[pos];
            f->image_positions[2 * pos + 1] = cadd(f->image_positions[pos], LITERALS[depth - level - 1]);
        }
    }
    /* coefficients start empty (None); parameter_predictors start (0,0) (:60-64) */
    return f;
}
static void fractal_free(fractal *f) {
    for (int l = 0; l < BASE_FRAC_DEPTH; l++) cmap_free(&f->position_map[l]);
    free(f);
}

/* Fractal::get_nearby_vectors, wavelet_transform.rs:71-90 */
static void get_nearby_vectors(uint8_t depth, cpx out[6]) {
    cpx zl, zmd;
    if (depth == 1) {
        zl = (cpx){-1, 1};
        zmd = (cpx){0, 2};
    } else if (depth == 2) {
        zl = (cpx){-2, 0};
        zmd = (cpx){0, -2};
    } else if (depth == 3) {
        zl = (cpx){-3, -1};
        zmd = (cpx){-1, -3};
    } else {
        zl = LITERALS[depth];
        zmd = cadd(LITERALS[depth + 1], zl);
    }
    out[0] = zl;
    out[1] = csub(zl, zmd);
    out[2] = cneg(zmd);
    out[3] = cneg(zl);
    out[4] = csub(zmd, zl);
    out[5] = zmd;
}

/* wavelet_transform.rs:97-177. gpm may be NULL when depth != 2 never indexes it (the reference
 * passes an empty Vec from get_lf_context_bucket, prediction.rs:95). */
static cpx get_left(cpx c, uint8_t depth, const cmap *gpm) {
    (void)gpm;
    cpx v[6];
    get_nearby_vectors(depth, v);
    return cadd(c, v[4]);
}
static cpx get_right(cpx c, uint8_t depth, const cmap *gpm) {
    (void)gpm;
    cpx v[6];
    get_nearby_vectors(depth, v);
    return cadd(c, v[1]);
}
static cpx get_down_left(cpx c, uint8_t depth, const cmap *gpm) {
    cpx v[6];
    get_nearby_vectors(depth, v);
    /* NB: indexes the global map by `depth`, i.e. level 2's key set (:121-123) */
    if (depth == 2 && !cmap_contains(&gpm[depth], cadd(c, v[3])) && cmap_contains(&gpm[depth], cadd(c, (cpx){1, 1})))
        return cadd(c, (cpx){1, 1});
    return cadd(c, v[3]);
}
static cpx get_down_right(cpx c, uint8_t depth, const cmap *gpm) {
    cpx v[6];
    get_nearby_vectors(depth, v);
    if (depth == 2 && !cmap_contains(&gpm[depth], cadd(c, v[3])) && cmap_contains(&gpm[depth], cadd(c, (cpx){1, 1})))
        return cadd(cadd(c, (cpx){1, 1}), v[1]);
    return cadd(c, v[2]);
}
static cpx get_up_right(cpx c, uint8_t depth, const cmap *gpm) {
    cpx v[6];
    get_nearby_vectors(depth, v);
    if (depth == 2 && !cmap_contains(&gpm[depth], cadd(c, v[0])) && cmap_contains(&gpm[depth], cadd(c, (cpx){-1, -1})))
        return cadd(c, (cpx){-1, -1});
    return cadd(c, v[0]);
}
static cpx get_up_left(cpx c, uint8_t depth, const cmap *gpm) {
    cpx v[6];
    get_nearby_vectors(depth, v);
    if (depth == 2 && !cmap_contains(&gpm[depth], cadd(c, v[0])) && cmap_contains(&gpm[depth], cadd(c, (cpx){-1, -1})))
        return cadd(cadd(c, (cpx){-1, -1}), v[4]);
    return cadd(c, v[5]);
}

/* RasterImage::get_pixel, images.rs:89-100 */
static opt_i32 get_pixel(const fri_oracle_wavelet *w, int32_t x, int32_t y, uint32_t channel) {
    if (x >= 0 && y >= 0 && x < (int32_t)w->width && y < (int32_t)w->height) {
        uint32_t position = (((uint32_t)y * w->width + (uint32_t)x) * w->channels + channel);
        return some_i32((int32_t)w->data[position]);
    }
    return NONE;
}

/* Fractal::extract_coefficients, wavelet_transform.rs:179-225 */
static void extract_coefficients(fractal *f, const fri_oracle_wavelet *w, uint8_t depth) {
    opt_i32 low_pass_values[3][NODES];
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < NODES; i++) {
            f->coefficients[c][i] = NONE;
            low_pass_values[c][i] = NONE;
        }
    for (uint32_t channel = 0; channel < w->channels; channel++) {
        for (int level = depth - 1; level >= 0; level--) {
            for (int pos = 1 << level; pos < 1 << (level + 1); pos++) {
                opt_i32 left_coef, right_coef;
                if (level == depth - 1) {
                    left_coef = get_pixel(w, f->image_positions[2 * pos].re, f->image_positions[2 * pos].im, channel);
                    right_coef = get_pixel(w, f->image_positions[2 * pos + 1].re, f->image_positions[2 * pos + 1].im, channel);
                } else {
                    left_coef = low_pass_values[channel][2 * pos];
                    right_coef = low_pass_values[channel][2 * pos + 1];
                }
                f->coefficients[channel][pos] = try_apply(left_coef, right_coef, op_diff, 0);
                low_pass_values[channel][pos] = try_apply(right_coef, f->coefficients[channel][pos], op_lowpass, 0);
            }
        }
        f->coefficients[channel][0] = low_pass_values[channel][1];
    }
    /* self.values = low_pass_values (:223) is write-only in the reference; not kept. */
}

/* utils.rs:17-32 (order_complex): ascending im, then re */
static _Thread_local const fri_oracle_wavelet *g_sort_ctx; /* qsort has no context argument; thread-local so that independent images can be transformed on independent threads (bench.py's all-core cpu_baseline) */
static int cmp_cells(const void *a, const void *b) {
    const fractal *fa = g_sort_ctx->cells[*(const uint32_t *)a];
    const fractal *fb = g_sort_ctx->cells[*(const uint32_t *)b];
    if (fa->center.im != fb->center.im) return fa->center.im < fb->center.im ? -1 : 1;
    if (fa->center.re != fb->center.re) return fa->center.re < fb->center.re ? -1 : 1;
    return 0;
}

static void push_cell(fri_oracle_wavelet *w, fractal *f) {
    if (w->n_cells == w->cap_cells) {
        w->cap_cells = w->cap_cells ? w->cap_cells * 2 : 64;
        w->cells = (fractal **)realloc(w->cells, w->cap_cells * sizeof(fractal *));
    }
    w->cells[w->n_cells] = f;
    cmap_insert(&w->fractal_lattice, f->center, (cpx){(int32_t)w->n_cells, 0});
    w->n_cells++;
}

/* WaveletImage::fractal_divide, wavelet_transform.rs:450-484.
 * `to_add.contains()` (:470, a linear VecDeque scan) is answered from a membership set kept in
 * lock-step with the queue: same truth value, O(1). A boundary centre can be queued twice in the
 * reference (it is in neither map when re-discovered); the second HashMap insert (:480) is a
 * no-op overwrite, reproduced here by the lattice lookup. */
static void fractal_divide(fri_oracle_wavelet *w, uint32_t width, uint32_t height, uint8_t depth) {
    cpx center = {(int32_t)width / 2, (int32_t)height / 2};
    size_t qcap = 1024, qhead = 0, qtail = 0;
    cpx *to_add = (cpx *)malloc(qcap * sizeof(cpx));
    cmap in_queue;
    cmap_init(&in_queue, 1024);
    size_t bcap = 256, blen = 0;
    cpx *boundary = (cpx *)malloc(bcap * sizeof(cpx));

    to_add[qtail++] = center;
    cmap_insert(&in_queue, center, (cpx){1, 0});
    while (qhead < qtail) {
        cpx position = to_add[qhead++];
        cmap_insert(&in_queue, position, (cpx){0, 0}); /* popped */
        if (position.re < 0 || position.im < 0 || position.re > (int32_t)width || position.im > (int32_t)height) {
            if (blen == bcap) boundary = (cpx *)realloc(boundary, (bcap *= 2) * sizeof(cpx));
            boundary[blen++] = position;
            continue;
        }
        fractal *f = fractal_new(depth, position);
        cpx v[6];
        get_nearby_vectors(f->depth, v); /* get_neighbour_locations :92-95 */
        for (int i = 0; i < 6; i++) {
            cpx neighbour = cadd(f->center, v[i]);
            const cpx *q = cmap_get(&in_queue, neighbour);
            int queued = q && q->re == 1;
            if (!cmap_contains(&w->fractal_lattice, neighbour) && !queued) {
                if (qtail == qcap) to_add = (cpx *)realloc(to_add, (qcap *= 2) * sizeof(cpx));
                to_add[qtail++] = neighbour;
                cmap_insert(&in_queue, neighbour, (cpx){1, 0});
            }
        }
        if (cmap_contains(&w->fractal_lattice, position))
            fractal_free(f); /* cannot happen for in-bounds cells; kept for symmetry */
        else
            push_cell(w, f);
    }
    for (size_t i = 0; i < blen; i++) {
        if (cmap_contains(&w->fractal_lattice, boundary[i])) continue; /* duplicate insert == overwrite */
        push_cell(w, fractal_new(depth, boundary[i]));
    }
    free(to_add);
    free(boundary);
    cmap_free(&in_queue);
}

/* WaveletImage::get_global_position_map, wavelet_transform.rs:434-448 */
static void build_global_position_map(fri_oracle_wavelet *w) {
    for (int l = 0; l < BASE_FRAC_DEPTH; l++) cmap_init(&w->global_position_map[l], ((size_t)w->n_retained << l) + 8);
    for (uint32_t k = 0; k < w->n_retained; k++) {
        const fractal *frac = w->cells[w->order[k]];
        for (int level = 0; level < BASE_FRAC_DEPTH; level++)
            for (int p = 1 << level; p < (1 << (level + 1)); p++) /* `1 << level + 1` == 1 << (level+1) */
                cmap_insert(&w->global_position_map[level], frac->image_positions[p], frac->center);
    }
}

/* The part of from_raster that does not depend on which cells the BFS found: coefficients, retain(), canonical order, position map. */
static void finish_from_raster(fri_oracle_wavelet *w) {
    for (size_t i = 0; i < w->n_cells; i++) extract_coefficients(w->cells[i], w, w->cells[i]->depth); /* :412-414 */

    /* retain(): keep cells whose DC is Some in all three channel slots (:415-416).
     * For channels == 1 the reference leaves slots 1,2 None, drops every cell and then panics in
     * sort_lattice (:664); the C ABI defines a luma plane by the channel-0 rule instead
     *  and this restatement follows the ABI there. */
    cmap kept;
    cmap_init(&kept, w->n_cells + 8);
    w->order = (uint32_t *)malloc((w->n_cells + 1) * sizeof(uint32_t));
    for (size_t i = 0; i < w->n_cells; i++) {
        fractal *f = w->cells[i];
        int all = 1;
        for (uint32_t c = 0; c < (w->channels == 3 ? 3u : 1u); c++) all &= f->coefficients[c][0].some;
        f->retained = all;
        if (all) {
            w->order[w->n_retained++] = (uint32_t)i;
            cmap_insert(&kept, f->center, (cpx){(int32_t)i, 0});
        }
    }
    cmap_free(&w->fractal_lattice);
    w->fractal_lattice = kept;
    g_sort_ctx = w;
    qsort(w->order, w->n_retained, sizeof(uint32_t), cmp_cells);
    build_global_position_map(w); /* :418 */
    w->data = NULL;
}

fri_oracle_wavelet *fri_oracle_from_raster(const uint8_t *data, uint32_t height, uint32_t width, uint32_t channels) {
    if (!data || (channels != 1 && channels != 3) || !height || !width) return NULL;
    fri_oracle_wavelet *w = (fri_oracle_wavelet *)calloc(1, sizeof(*w));
    w->height = height;
    w->width = width;
    w->channels = channels;
    w->data = data;
    cmap_init(&w->fractal_lattice, 1024);

    fractal_divide(w, width, height, BASE_FRAC_DEPTH); /* :406-410 */
    finish_from_raster(w);                             /* :412-418 */
    return w;
}

/* from_raster over a GIVEN set of cell centres instead of fractal_divide's BFS (config 5's sampled checks: the whole 16384^2 lattice
 * would take this hash-map-shaped restatement ~40 GB). Everything per cell is the reference's (Fractal::new :42-69, extract_coefficients
 * :179-225, the retain rule :415-416, the position map :434-448); a node's context (fri_oracle_context_at) is the full image's as long
 * as every cell its neighbour positions fall into is among the centres - the caller passes a cell together with its lattice
 * neighbourhood. Duplicate centres are taken once. */
fri_oracle_wavelet *fri_oracle_from_raster_cells(const uint8_t *data, uint32_t height, uint32_t width, uint32_t channels, const int32_t *centers, uint32_t n_centers) {
    if (!data || !centers || !n_centers || (channels != 1 && channels != 3) || !height || !width) return NULL;
    fri_oracle_wavelet *w = (fri_oracle_wavelet *)calloc(1, sizeof(*w));
    w->height = height;
    w->width = width;
    w->channels = channels;
    w->data = data;
    cmap_init(&w->fractal_lattice, 2 * (size_t)n_centers + 8);
    for (uint32_t i = 0; i < n_centers; i++) {
        const cpx c = {centers[2 * i], centers[2 * i + 1]};
        if (!cmap_contains(&w->fractal_lattice, c)) push_cell(w, fractal_new(BASE_FRAC_DEPTH, c));
    }
    finish_from_raster(w);
    return w;
}

/* One cell on its own: Fractal::new(depth 9, centre) + extract_coefficients (wavelet_transform.rs:42-69, :179-225) - the transform is
 * per-cell independent. out[channels][512] in heap order, None = FRI_ORACLE_NONE. Returns 1 if the retain rule (:415-416, the C ABI's
 * channel-0 rule for planes) keeps the cell, 0 if not, -1 on bad arguments. */
int fri_oracle_cell(const uint8_t *data, uint32_t height, uint32_t width, uint32_t channels, int32_t center_re, int32_t center_im, int32_t *out) {
    if (!data || !out || (channels != 1 && channels != 3) || !height || !width) return -1;
    fri_oracle_wavelet w;
    memset(&w, 0, sizeof w);
    w.height = height, w.width = width, w.channels = channels, w.data = data;
    fractal *f = fractal_new(BASE_FRAC_DEPTH, (cpx){center_re, center_im});
    extract_coefficients(f, &w, f->depth);
    int all = 1;
    for (uint32_t c = 0; c < channels; c++) {
        all &= f->coefficients[c][0].some;
        for (int i = 0; i < NODES; i++) out[(size_t)c * NODES + i] = f->coefficients[c][i].some ? f->coefficients[c][i].v : FRI_ORACLE_NONE;
    }
    fractal_free(f);
    return all;
}

void fri_oracle_free(fri_oracle_wavelet *w) {
    if (!w) return;
    for (size_t i = 0; i < w->n_cells; i++) fractal_free(w->cells[i]);
    free(w->cells);
    free(w->order);
    cmap_free(&w->fractal_lattice);
    for (int l = 0; l < BASE_FRAC_DEPTH; l++) cmap_free(&w->global_position_map[l]);
    free(w);
}

uint32_t fri_oracle_num_cells(const fri_oracle_wavelet *w) { return w->n_retained; }
uint32_t fri_oracle_num_bfs_cells(const fri_oracle_wavelet *w) { return (uint32_t)w->n_cells; }
uint32_t fri_oracle_channels(const fri_oracle_wavelet *w) { return w->channels; }

void fri_oracle_centers(const fri_oracle_wavelet *w, int32_t *out) {
    for (uint32_t k = 0; k < w->n_retained; k++) {
        out[2 * k] = w->cells[w->order[k]]->center.re;
        out[2 * k + 1] = w->cells[w->order[k]]->center.im;
    }
}

void fri_oracle_coefficients(const fri_oracle_wavelet *w, int32_t *out) {
    size_t F = w->n_retained;
    for (uint32_t c = 0; c < w->channels; c++)
        for (size_t k = 0; k < F; k++) {
            const fractal *f = w->cells[w->order[k]];
            int32_t *o = out + ((size_t)c * F + k) * NODES;
            for (int i = 0; i < NODES; i++) o[i] = f->coefficients[c][i].some ? f->coefficients[c][i].v : FRI_ORACLE_NONE;
        }
}

void fri_oracle_set_coefficients(fri_oracle_wavelet *w, const int32_t *in) {
    size_t F = w->n_retained;
    for (uint32_t c = 0; c < w->channels; c++)
        for (size_t k = 0; k < F; k++) {
            fractal *f = w->cells[w->order[k]];
            const int32_t *o = in + ((size_t)c * F + k) * NODES;
            for (int i = 0; i < NODES; i++) f->coefficients[c][i] = (o[i] == FRI_ORACLE_NONE) ? NONE : some_i32(o[i]);
        }
}

/* utils.rs:5-14 */
static size_t get_prev_power_two(size_t x) {
    size_t num = x;
    num |= num >> 1;
    num |= num >> 2;
    num |= num >> 4;
    num |= num >> 8;
    num |= num >> 16;
    return num ^ (num >> 1);
}
static uint32_t trailing_zeros(size_t x) {
    if (!x) return 64; /* usize::trailing_zeros(0) */
    uint32_t n = 0;
    while (!(x & 1)) {
        x >>= 1;
        n++;
    }
    return n;
}
uint32_t fri_oracle_quant_layer(uint32_t i) { return trailing_zeros(get_prev_power_two((size_t)i + 1)); }

/* quantization::encode, stages/quantization.rs:7-25 */
int fri_oracle_quantize(fri_oracle_wavelet *w, const int32_t qmatrix[32]) {
    for (uint32_t k = 0; k < w->n_retained; k++) {
        fractal *f = w->cells[w->order[k]];
        for (int channel = 0; channel < 3; channel++)
            for (int i = 0; i < NODES; i++)
                if (f->coefficients[channel][i].some) {
                    uint32_t layer = trailing_zeros(get_prev_power_two((size_t)i + 1));
                    if (qmatrix[layer] == 0) return -1; /* Rust: panic "attempt to divide by zero" */
                    f->coefficients[channel][i].v /= qmatrix[layer];
                }
    }
    return 0;
}

/* utils.rs:34-48. Release-build (wrapping) arithmetic for out-of-range inputs. */
uint32_t fri_oracle_pack_signed(int32_t k) {
    if (k >= 0) return 2u * (uint32_t)k;
    return (uint32_t)(-2 * (int64_t)k - 1);
}
int32_t fri_oracle_unpack_signed(uint32_t k) {
    if (k % 2 == 0) return (int32_t)(k / 2);
    return (int32_t)(k + 1) / -2;
}

/* Rust `f32 as u32` / `f32 as i32`: saturating, NaN -> 0, truncation toward zero. */
static uint32_t f32_as_u32(float x) {
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 4294967296.0f) return UINT32_MAX;
    return (uint32_t)x;
}
static int32_t f32_as_i32(float x) {
    if (!(x == x)) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}

/* prediction.rs:55-68 */
uint32_t fri_oracle_assign_bucket(float width) {
    uint32_t w = f32_as_u32(width);
    if (w < 3) return 0;
    if (w < 5) return 1;
    if (w < 6) return 2;
    if (w < 8) return 3;
    if (w < 12) return 4;
    if (w < 16) return 5;
    if (w < 20) return 6;
    if (w < 25) return 7;
    if (w < 30) return 8;
    return 9;
}

static inline const fractal *lattice_get(const fri_oracle_wavelet *w, cpx center) {
    const cpx *v = cmap_get(&w->fractal_lattice, center);
    return v ? w->cells[v->re] : NULL;
}

/* prediction.rs:39-53 */
static int get_containing_fractal(const fri_oracle_wavelet *w, cpx pos, size_t level, const fractal *f, cpx *out) {
    cpx v[6];
    get_nearby_vectors(f->depth, v);
    for (int i = 0; i < 6; i++) {
        cpx location = cadd(f->center, v[i]);
        const fractal *neighbour = lattice_get(w, location);
        if (neighbour && cmap_contains(&neighbour->position_map[level], pos)) {
            *out = location;
            return 1;
        }
    }
    return 0;
}

static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
static inline int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
static inline int32_t iabs_wrapping(int32_t a) { return a < 0 ? (int32_t)(0u - (uint32_t)a) : a; }
static inline int32_t sub_wrapping(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t add_wrapping(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }

/* get_lf_context_bucket, prediction.rs:86-149 */
static void get_lf_context_bucket(const fri_oracle_wavelet *w, size_t position, uint8_t current_depth, cpx parent_fractal_pos,
                                  uint32_t channel, uint32_t *bucket_out, int32_t *prediction_out) {
    const fractal *f = lattice_get(w, parent_fractal_pos);
    cpx position_in_image = f->image_positions[position];
    cpx neighbours[3] = {
        get_left(position_in_image, (uint8_t)(f->depth - current_depth), NULL),
        get_up_left(position_in_image, (uint8_t)(f->depth - current_depth), NULL),
        get_up_right(position_in_image, (uint8_t)(f->depth - current_depth), NULL),
    };
    size_t level = current_depth;
    int32_t values[3];
    for (int n = 0; n < 3; n++) {
        cpx pos = neighbours[n];
        const cpx *own = cmap_get(&f->position_map[level], pos);
        if (!own) {
            cpx nposition;
            if (get_containing_fractal(w, pos, level, f, &nposition)) {
                const fractal *containing = lattice_get(w, nposition);
                values[n] = containing->coefficients[channel][position].some ? containing->coefficients[channel][position].v : 0;
            } else {
                values[n] = 0;
            }
        } else {
            size_t loc = (size_t)own->re;
            values[n] = f->coefficients[channel][loc].some ? f->coefficients[channel][loc].v : 0;
        }
    }
    uint32_t width = (uint32_t)iabs_wrapping(sub_wrapping(values[0], values[2]));
    *bucket_out = fri_oracle_assign_bucket((float)width);
    int32_t prediction;
    if (values[1] >= imax(values[0], values[2]))
        prediction = imax(values[0], values[2]);
    else if (values[1] <= imin(values[0], values[2]))
        prediction = imin(values[0], values[2]);
    else
        prediction = sub_wrapping(add_wrapping(values[0], values[2]), values[1]);
    *prediction_out = prediction;
}

/* ContextModeler::get_neighbour_values, context_modeling.rs:25-77 */
static void get_neighbour_values(const fri_oracle_wavelet *w, cpx image_position, uint8_t current_depth, cpx parent_fractal_pos,
                                 uint32_t channel, int32_t out[6]) {
    size_t level = current_depth;
    const fractal *f = lattice_get(w, parent_fractal_pos);
    const cmap *gpm = w->global_position_map;
    uint8_t d = (uint8_t)(f->depth - (uint8_t)level);
    cpx same_level[3] = {get_left(image_position, d, gpm), get_up_left(image_position, d, gpm), get_up_right(image_position, d, gpm)};
    for (int n = 0; n < 3; n++) {
        const cpx *parent_fractal_loc = cmap_get(&gpm[level], same_level[n]);
        if (parent_fractal_loc) {
            const fractal *containing = lattice_get(w, *parent_fractal_loc);
            size_t haar_pos = (size_t)cmap_get(&containing->position_map[level], same_level[n])->re;
            out[n] = containing->coefficients[channel][haar_pos].some ? containing->coefficients[channel][haar_pos].v : 0;
        } else {
            out[n] = 0;
        }
    }
    cpx above_level[3] = {get_right(image_position, d, gpm), get_down_left(image_position, d, gpm), get_down_right(image_position, d, gpm)};
    for (int n = 0; n < 3; n++) {
        const cpx *parent_fractal_loc = cmap_get(&gpm[level], above_level[n]);
        if (parent_fractal_loc) {
            const fractal *containing = lattice_get(w, *parent_fractal_loc);
            size_t haar_pos = (size_t)cmap_get(&containing->position_map[level], above_level[n])->re;
            out[3 + n] = containing->coefficients[channel][haar_pos / 2].some ? containing->coefficients[channel][haar_pos / 2].v : 0;
        } else {
            out[3 + n] = 0;
        }
    }
}

/* get_hf_context_bucket, prediction.rs:151-207. f32, left-to-right, one rounding per op
 * (compiled with -ffp-contract=off; volatile-free because SSE2 has no excess precision). */
static void get_hf_context_bucket(const fri_oracle_wavelet *w, cpx image_position, uint8_t current_depth, cpx parent_fractal_pos,
                                  const float value_prediction_params[3][6], const float width_prediction_params[3][6],
                                  uint32_t channel, uint32_t *bucket_out, int32_t *prediction_out) {
    uint8_t depth = lattice_get(w, parent_fractal_pos)->depth;
    const float *vp, *wp;
    if (current_depth < depth - 2) {
        vp = value_prediction_params[2];
        wp = width_prediction_params[2];
    } else if (current_depth == depth - 2) {
        vp = value_prediction_params[1];
        wp = width_prediction_params[1];
    } else {
        vp = value_prediction_params[0];
        wp = width_prediction_params[0];
    }
    int32_t values[6];
    get_neighbour_values(w, image_position, current_depth, parent_fractal_pos, channel, values);

    float width = wp[0];
    width = width + wp[1] * (float)iabs_wrapping(sub_wrapping(values[0], values[3]));
    width = width + wp[2] * (float)iabs_wrapping(sub_wrapping(values[1], values[2]));
    width = width + wp[3] * (float)iabs_wrapping(sub_wrapping(values[4], values[5]));
    width = width + wp[4] * (float)iabs_wrapping(sub_wrapping(values[1], values[5]));
    width = width + wp[5] * (float)iabs_wrapping(sub_wrapping(values[2], values[4]));
    *bucket_out = fri_oracle_assign_bucket(width);

    float prediction = (float)values[0] * vp[0];
    prediction = prediction + (float)values[1] * vp[1];
    prediction = prediction + (float)values[2] * vp[2];
    prediction = prediction + (float)values[3] * vp[3];
    prediction = prediction + (float)values[4] * vp[4];
    prediction = prediction + (float)values[5] * vp[5];
    *prediction_out = f32_as_i32(prediction);
}

static void bump(uint32_t *hist, uint32_t bucket, int32_t value, int32_t prediction, uint64_t *n_oob) {
    int32_t residual = sub_wrapping(value, prediction);
    uint32_t sym = fri_oracle_pack_signed(residual);
    if (sym >= ALPHABET_SIZE) { /* entropy_coding.rs:99 would panic (index out of bounds) */
        (*n_oob)++;
        return;
    }
    hist[bucket * ALPHABET_SIZE + sym] += 1; /* bump_freq, entropy_coding.rs:98-100 */
}

/* prediction::encode loop, prediction.rs:237-298, for one channel with given parameters.
 * The reference walks sorted_lattice[level]; every key of global_position_map[level] appears in it
 * exactly once (asserted at wavelet_transform.rs:701) and nothing here depends on the visiting
 * order, so the walk below goes over cells and heap indices directly. */
int fri_oracle_predict(fri_oracle_wavelet *w, uint32_t channel, const float value_params[3][6], const float width_params[3][6],
                       uint32_t *hist, uint64_t *n_out_of_alphabet) {
    if (channel >= w->channels) return -1;
    uint64_t oob = 0;
    /* first scan: DC (:241-254) */
    for (uint32_t k = 0; k < w->n_retained; k++) {
        fractal *f = w->cells[w->order[k]];
        if (f->coefficients[channel][0].some) {
            uint32_t bucket;
            int32_t prediction;
            get_lf_context_bucket(w, 0, 0, f->center, channel, &bucket, &prediction);
            f->pred_bucket[channel][0] = (uint8_t)bucket;
            f->pred_value[channel][0] = prediction;
            bump(hist, bucket, f->coefficients[channel][0].v, prediction, &oob);
        }
    }
    /* second scan: high-frequency root (:257-270) */
    for (uint32_t k = 0; k < w->n_retained; k++) {
        fractal *f = w->cells[w->order[k]];
        if (f->coefficients[channel][1].some) {
            uint32_t bucket;
            int32_t prediction;
            get_lf_context_bucket(w, 1, 0, f->center, channel, &bucket, &prediction);
            f->pred_bucket[channel][1] = (uint8_t)bucket;
            f->pred_value[channel][1] = prediction;
            bump(hist, bucket, f->coefficients[channel][1].v, prediction, &oob);
        }
    }
    /* levels depth-1 .. 1 (:272-298) */
    for (int level = BASE_FRAC_DEPTH - 1; level >= 1; level--) {
        for (uint32_t k = 0; k < w->n_retained; k++) {
            fractal *f = w->cells[w->order[k]];
            for (int p = 1 << level; p < (1 << (level + 1)); p++) {
                cpx image_pos = f->image_positions[p];
                cpx parent_pos = *cmap_get(&w->global_position_map[level], image_pos);
                fractal *pf = (fractal *)lattice_get(w, parent_pos);
                size_t haar_tree_pos = (size_t)cmap_get(&pf->position_map[level], image_pos)->re;
                if (pf->coefficients[channel][haar_tree_pos].some) {
                    uint32_t bucket;
                    int32_t prediction;
                    get_hf_context_bucket(w, image_pos, (uint8_t)level, parent_pos, value_params, width_params, channel, &bucket, &prediction);
                    bump(hist, bucket, pf->coefficients[channel][haar_tree_pos].v, prediction, &oob);
                    pf->pred_bucket[channel][haar_tree_pos] = (uint8_t)bucket;
                    pf->pred_value[channel][haar_tree_pos] = prediction;
                }
            }
        }
    }
    if (n_out_of_alphabet) *n_out_of_alphabet = oob;
    return 0;
}

/* decode_symbol's context, entropy_coding.rs:205-236: (bucket, prediction) of ONE node from the coefficients as they are now.
 * cell = index in canonical order, heap = heap index (0 = DC, 1 = root: the low-frequency predictor). Returns -1 for a None node. */
int fri_oracle_context_at(const fri_oracle_wavelet *w, uint32_t channel, uint32_t cell, uint32_t heap, const float value_params[3][6],
                          const float width_params[3][6], uint32_t *bucket, int32_t *prediction) {
    if (channel >= w->channels || cell >= w->n_retained || heap >= NODES) return -2;
    const fractal *f = w->cells[w->order[cell]];
    if (!f->coefficients[channel][heap].some) return -1;
    if (heap < 2) {
        get_lf_context_bucket(w, heap, 0, f->center, channel, bucket, prediction);
    } else {
        int level = 0;
        while ((2u << level) <= heap) level++;
        get_hf_context_bucket(w, f->image_positions[heap], (uint8_t)level, f->center, value_params, width_params, channel, bucket, prediction);
    }
    return 0;
}

/* fractal.coefficients[channel][haar_tree_pos] = Some(symbol), entropy_coding.rs:387, :409, :440 */
int fri_oracle_set_coefficient(fri_oracle_wavelet *w, uint32_t channel, uint32_t cell, uint32_t heap, int32_t value) {
    if (channel >= w->channels || cell >= w->n_retained || heap >= NODES) return -2;
    w->cells[w->order[cell]]->coefficients[channel][heap] = some_i32(value);
    return 0;
}

void fri_oracle_predictors(const fri_oracle_wavelet *w, uint32_t channel, uint8_t *bucket, int32_t *prediction) {
    for (uint32_t k = 0; k < w->n_retained; k++) {
        const fractal *f = w->cells[w->order[k]];
        memcpy(bucket + (size_t)k * NODES, f->pred_bucket[channel], NODES);
        memcpy(prediction + (size_t)k * NODES, f->pred_value[channel], NODES * sizeof(int32_t));
    }
}

void fri_oracle_neighbour_values(const fri_oracle_wavelet *w, uint32_t channel, int32_t *out) {
    memset(out, 0, (size_t)w->n_retained * NODES * 6 * sizeof(int32_t));
    for (uint32_t k = 0; k < w->n_retained; k++) {
        const fractal *f = w->cells[w->order[k]];
        for (int level = 1; level < BASE_FRAC_DEPTH; level++)
            for (int p = 1 << level; p < (1 << (level + 1)); p++)
                get_neighbour_values(w, f->image_positions[p], (uint8_t)level, f->center, channel, out + ((size_t)k * NODES + p) * 6);
    }
}

/* RasterImage::set_pixel, images.rs:103-111 */
static void set_pixel(const fri_oracle_wavelet *w, uint8_t *data, int32_t x, int32_t y, int32_t value, uint32_t channel) {
    if (x >= 0 && y >= 0 && x < (int32_t)w->width && y < (int32_t)w->height) {
        size_t position = (((size_t)y * w->width + (size_t)x) * w->channels + channel);
        data[position] = (uint8_t)(value < 0 ? 0 : value > 255 ? 255 : value);
    }
}

/* RasterImage::from_wavelet + extract_values, wavelet_transform.rs:308-381 */
void fri_oracle_to_raster(const fri_oracle_wavelet *w, uint8_t *out) {
    memset(out, 0, (size_t)w->height * w->width * w->channels);
    for (uint32_t k = 0; k < w->n_retained; k++) {
        const fractal *f = w->cells[w->order[k]];
        for (uint32_t channel = 0; channel < w->channels; channel++) {
            int32_t low_pass_values[NODES];
            memset(low_pass_values, 0, sizeof(low_pass_values));
            low_pass_values[1] = f->coefficients[channel][0].v; /* .unwrap(): retained cells always have a DC */
            for (int level = 0; level < f->depth; level++) {
                for (int pos = 1 << level; pos < 1 << (level + 1); pos++) {
                    if (f->coefficients[channel][pos].some) {
                        int32_t dif = f->coefficients[channel][pos].v;
                        int32_t right_subtree = low_pass_values[pos] - dif / 2;
                        int32_t left_subtree = dif + right_subtree;
                        if (level == f->depth - 1) {
                            cpx lp = f->image_positions[2 * pos], rp = f->image_positions[2 * pos + 1];
                            set_pixel(w, out, lp.re, lp.im, left_subtree, channel);
                            set_pixel(w, out, rp.re, rp.im, right_subtree, channel);
                        } else {
                            low_pass_values[2 * pos] = left_subtree;
                            low_pass_values[2 * pos + 1] = right_subtree;
                        }
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* sort_lattice / scan_level, wavelet_transform.rs:490-705             */
/* ------------------------------------------------------------------ */
typedef struct {
    cpx *v;
    size_t len, cap;
} cvec;
static void cvec_push(cvec *p, cpx c) {
    if (p->len == p->cap) p->v = (cpx *)realloc(p->v, (p->cap = p->cap ? p->cap * 2 : 256) * sizeof(cpx));
    p->v[p->len++] = c;
}

static int is_pos_in_row_boundary(cpx pos, cpx row_dir, int32_t min_real, int32_t max_real, int32_t min_imag, int32_t max_imag) {
    if (abs(row_dir.re) > abs(row_dir.im)) return pos.im >= min_imag && pos.im <= max_imag;
    return pos.re >= min_real && pos.re <= max_real;
}

static cvec scan_level(uint8_t level, uint8_t depth, cpx center, const cmap *gpm, int32_t min_real, int32_t max_real, int32_t min_imag,
                       int32_t max_imag, size_t expected) {
    cpx nv[6];
    get_nearby_vectors((uint8_t)(BASE_FRAC_DEPTH - level), nv);
    cpx row_dir = nv[3], rev_row_dir = nv[0], col_dir = nv[1], rev_col_dir = nv[4];
    const cpx m11 = {-1, -1}, p11 = {1, 1};
    cvec plane = {0};

    cpx first = center;
    int layer_seven_mod = 0;
    if (!cmap_contains(gpm, cadd(center, rev_row_dir)) && cmap_contains(gpm, cadd(center, m11))) layer_seven_mod = 1;
    cpx last_seen = first;

#define STEP_BACK()                                                                  \
    do {                                                                             \
        if (depth - level != 2) {                                                    \
            first = cadd(first, rev_row_dir);                                        \
        } else {                                                                     \
            first = cadd(first, (layer_seven_mod % 2 == 0) ? rev_row_dir : m11);     \
            layer_seven_mod += 1;                                                    \
        }                                                                            \
    } while (0)
#define IN_BBOX(p) ((p).im <= max_imag && (p).im >= min_imag && (p).re <= max_real && (p).re >= min_real)

    while (cmap_contains(gpm, first)) { /* :533-545 */
        last_seen = first;
        STEP_BACK();
    }
    for (;;) { /* find first row, :548-585 */
        cpx column_forward = first, column_backward = first;
        int empty_column = 1;
        while ((column_forward.im <= max_imag && column_forward.im >= min_imag) || (column_backward.im <= max_imag && column_backward.im >= min_imag) ||
               (column_forward.re <= max_real && column_forward.re >= min_real) || (column_backward.re <= max_real && column_backward.re >= min_real)) {
            column_forward = cadd(column_forward, col_dir);
            column_backward = cadd(column_backward, rev_col_dir);
            if (cmap_contains(gpm, column_forward)) {
                last_seen = column_forward;
                empty_column = 0;
                break;
            }
            if (cmap_contains(gpm, column_backward)) {
                last_seen = column_backward;
                empty_column = 0;
                break;
            }
        }
        if (empty_column) {
            first = last_seen;
            break;
        }
        STEP_BACK();
    }
    while (IN_BBOX(first)) { /* scanning backwards find first column, :588-597 */
        first = cadd(first, rev_col_dir);
        if (cmap_contains(gpm, first)) last_seen = first;
    }
    first = last_seen;
    layer_seven_mod = 1;

    for (;;) { /* fill plane in sorted order, :603-652 */
        cpx scan = first;
        for (;;) {
            if (cmap_contains(gpm, scan)) cvec_push(&plane, scan);
            if ((scan.im > max_imag || scan.im < min_imag) || (col_dir.im == 0 && (scan.re > max_real || scan.re < min_real))) break;
            scan = cadd(scan, col_dir);
        }
        if (plane.len > expected * 4 + 1024) break; /* runaway guard (never hit when the reading is right) */
        if (depth - level != 2) {
            first = cadd(first, row_dir);
        } else {
            first = cadd(first, (layer_seven_mod % 2 == 0) ? p11 : row_dir);
            layer_seven_mod += 1;
        }
        int out = 0;
        while (!cmap_contains(gpm, first)) {
            first = cadd(first, col_dir);
            if (!is_pos_in_row_boundary(first, row_dir, min_real, max_real, min_imag, max_imag)) {
                out = 1;
                break;
            }
        }
        if (out) break; /* break 'outer */
        if (cmap_contains(gpm, first)) {
            last_seen = first;
            while (IN_BBOX(first)) {
                first = cadd(first, rev_col_dir);
                if (cmap_contains(gpm, first)) last_seen = first;
            }
            first = last_seen;
        }
    }
#undef STEP_BACK
#undef IN_BBOX
    return plane;
}

int64_t fri_oracle_sorted_level(const fri_oracle_wavelet *w, uint32_t level, int32_t *out) {
    if (level >= BASE_FRAC_DEPTH || !w->n_retained) return -1;
    /* bbox of the level-8 node positions of all retained cells, :666-685 */
    const cmap *g8 = &w->global_position_map[BASE_FRAC_DEPTH - 1];
    int32_t min_real = INT32_MAX, max_real = INT32_MIN, min_imag = INT32_MAX, max_imag = INT32_MIN;
    for (size_t i = 0; i < g8->cap; i++)
        if (g8->used[i]) {
            cpx k = g8->keys[i];
            if (k.re < min_real) min_real = k.re;
            if (k.re > max_real) max_real = k.re;
            if (k.im < min_imag) min_imag = k.im;
            if (k.im > max_imag) max_imag = k.im;
        }
    cpx center = {(int32_t)w->width / 2, (int32_t)w->height / 2}; /* :688 */
    size_t expected = (size_t)w->n_retained << level;
    cvec plane = scan_level((uint8_t)level, BASE_FRAC_DEPTH, center, &w->global_position_map[level], min_real, max_real, min_imag, max_imag, expected);
    int64_t n = (int64_t)plane.len;
    if (out)
        for (size_t i = 0; i < plane.len; i++) {
            out[2 * i] = plane.v[i].re;
            out[2 * i + 1] = plane.v[i].im;
        }
    free(plane.v);
    return n;
}

/* ------------------------------------------------------------------ */
/* small exports for KATs                                              */
/* ------------------------------------------------------------------ */
int fri_oracle_pair(int l_some, int32_t l, int r_some, int32_t r, int32_t *d, int32_t *s) {
    opt_i32 lo = l_some ? some_i32(l) : NONE, ro = r_some ? some_i32(r) : NONE;
    opt_i32 c = try_apply(lo, ro, op_diff, 0);
    opt_i32 lp = try_apply(ro, c, op_lowpass, 0);
    if (c.some) *d = c.v;
    if (lp.some) *s = lp.v;
    return c.some;
}
void fri_oracle_nearby_vectors(uint32_t depth, int32_t out[6][2]) {
    cpx v[6];
    get_nearby_vectors((uint8_t)depth, v);
    for (int i = 0; i < 6; i++) {
        out[i][0] = v[i].re;
        out[i][1] = v[i].im;
    }
}
void fri_oracle_literal(uint32_t i, int32_t out[2]) {
    out[0] = LITERALS[i].re;
    out[1] = LITERALS[i].im;
}
