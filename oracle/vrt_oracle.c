/* vrt_oracle.c -- CPU restatement of the reference's per-pixel trace path (see vrt_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: never imported, linked or executed by the shipped package.
 * Scalar binary64, evaluation order exactly as the reference writes it; build with
 * -ffp-contract=off (oracle/Makefile).  Every function cites the reference lines it restates.
 */
#include "vrt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "vrt_math.h" /* the product's portable sin/cos/pow: libm mode ORC_LIBM_PORTABLE */

/* ------------------------------------------------------------------------------------------
 * CPython float helpers
 * ---------------------------------------------------------------------------------------- */
/* float.__mod__ (CPython Objects/floatobject.c float_rem): result takes the sign of the divisor */
static double py_mod(double v, double w) {
    double m = fmod(v, w);
    if (m != 0.0) {
        if ((w < 0) != (m < 0)) m += w;
    } else {
        m = copysign(0.0, w);
    }
    return m;
}

/* float.__floordiv__ (CPython float_divmod / float_floor_div) */
static double py_floordiv(double v, double w) {
    double m = fmod(v, w);
    double d = (v - m) / w;
    if (m != 0.0 && ((w < 0) != (m < 0))) d -= 1.0;
    if (d != 0.0) {
        double f = floor(d);
        if (d - f > 0.5) f += 1.0;
        return f;
    }
    return copysign(0.0, v / w);
}

/* int.__floordiv__ */
static int64_t ifloordiv(int64_t a, int64_t b) {
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) q -= 1;
    return q;
}

/* builtin round(float) -> int: round-half-even */
static double py_round(double v) { return rint(v); }

/* ------------------------------------------------------------------------------------------
 * MT19937 as used by CPython's random module (Modules/_randommodule.c)
 * ---------------------------------------------------------------------------------------- */
#define MT_N 624
#define MT_M 397
typedef struct {
    uint32_t mt[MT_N];
    int idx;
} mt_state;

static void mt_init_genrand(mt_state* s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < MT_N; i++) s->mt[i] = 1812433253U * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = MT_N;
}

static void mt_init_by_array(mt_state* s, const uint32_t* key, int len) {
    mt_init_genrand(s, 19650218U);
    uint32_t* mt = s->mt;
    int i = 1, j = 0;
    int k = MT_N > len ? MT_N : len;
    for (; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525U)) + key[j] + (uint32_t)j;
        i++;
        j++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
        if (j >= len) j = 0;
    }
    for (k = MT_N - 1; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
        i++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
    }
    mt[0] = 0x80000000U;
}

/* random.seed(n) for a non-negative int n < 2^128: key = little-endian 32-bit digits (random_seed) */
static void mt_seed_u128(mt_state* s, uint64_t lo, uint64_t hi) {
    uint32_t key[4] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
    int len = 4;
    while (len > 1 && key[len - 1] == 0) len--;
    mt_init_by_array(s, key, len);
}

static uint32_t mt_genrand(mt_state* s) {
    static const uint32_t mag01[2] = {0x0U, 0x9908b0dfU};
    uint32_t y;
    uint32_t* mt = s->mt;
    if (s->idx >= MT_N) {
        int kk;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1U];
        }
        for (; kk < MT_N - 1; kk++) {
            y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1U];
        }
        y = (mt[MT_N - 1] & 0x80000000U) | (mt[0] & 0x7fffffffU);
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1U];
        s->idx = 0;
    }
    y = mt[s->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
}

/* random.random(): genrand_res53 */
static double mt_random(mt_state* s) {
    uint32_t a = mt_genrand(s) >> 5, b = mt_genrand(s) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

void orc_rng_draws(uint64_t seed_lo, uint64_t seed_hi, int n, double* out) {
    mt_state s;
    mt_seed_u128(&s, seed_lo, seed_hi);
    for (int i = 0; i < n; i++) out[i] = mt_random(&s);
}

/* ------------------------------------------------------------------------------------------
 * libm dispatch
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double (*sin_)(double);
    double (*cos_)(double);
    double (*pow_)(double, double);
    int portable;
} libm_t;

static double p_sin(double x) { return vrt_sin(x); }
static double p_cos(double x) { return vrt_cos(x); }
static double p_pow(double x, double y) { return vrt_pow(x, y); }

static libm_t libm_get(int mode) {
    libm_t m;
    if (mode == ORC_LIBM_PORTABLE) {
        m.sin_ = p_sin; m.cos_ = p_cos; m.pow_ = p_pow; m.portable = 1;
    } else {
        m.sin_ = sin; m.cos_ = cos; m.pow_ = pow; m.portable = 0;
    }
    return m;
}
double orc_sin(int mode, double x) { return libm_get(mode).sin_(x); }
double orc_cos(int mode, double x) { return libm_get(mode).cos_(x); }
double orc_pow(int mode, double x, double y) { return libm_get(mode).pow_(x, y); }
/* vrt_math.h's joint form of the lens quaternion's four values (what the HIP ray generation calls): out = sin a, cos a,
 * sin b, cos b; returns whether the straight-line fast block produced them.  tests/test_math.py holds it to vrt_sin /
 * vrt_cos bit for bit. */
int orc_sincos2(double a, double b, double* out) {
    int fast = vrt_sincos2_fast(a, b, out);
    vrt_sincos2(a, b, out);
    return fast;
}

/* float ** 2 (lib.py:375): CPython float_pow -> C pow(v, 2.0).  The portable mode uses v*v, which is the
 * correctly rounded square (vrt_pow does not take negative bases). */
static double sq(const libm_t* lm, double v) { return lm->portable ? v * v : lm->pow_(v, 2.0); }

/* ------------------------------------------------------------------------------------------
 * Scene access
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const orc_scene* sc;
    const orc_settings* st;
    const orc_camera* cam;
    libm_t lm;
    int64_t cs;
} ctx_t;

/* self.chunks[post] if post in self.chunks else None (init.py:71, init.py:30-33).
 * cmin = snapped position (integral doubles).  Returns chunk index or -1. */
static int64_t chunk_index(const ctx_t* c, const double cmin[3]) {
    int64_t ci[3];
    for (int a = 0; a < 3; a++) {
        double rel = cmin[a] - (double)c->sc->origin[a];
        if (rel < 0 || rel >= (double)(c->sc->dims[a] * c->cs)) return -1;
        ci[a] = (int64_t)rel / c->cs;
    }
    int64_t idx = (ci[0] * c->sc->dims[1] + ci[1]) * c->sc->dims[2] + ci[2];
    return c->sc->present[idx] ? idx : -1;
}

/* Frame.get_voxel (data.py:136-145) for a camera chunk Frame produced by Window.chunk_update
 * (init.py:441-444): the Frame of chunk `cmin` with resolution r holds the voxels whose world
 * coordinates are multiples of r inside [cmin, cmin + cs), keyed by p // r.  get_voxel(fp) looks up
 * key fp // r, i.e. world cell (fp // r) * r, which exists only inside the chunk's own box.
 * Returns material id (0 = None). */
static int orc_lookup(const ctx_t* c, int64_t chunk_idx, const double cmin[3], const int64_t fp[3]) {
    int64_t r = c->sc->res[chunk_idx];
    int64_t g[3];
    for (int a = 0; a < 3; a++) {
        int64_t w = r > 1 ? ifloordiv(fp[a], r) * r : fp[a];
        int64_t lo = (int64_t)cmin[a];
        if (w < lo || w >= lo + c->cs) return 0;
        g[a] = w - c->sc->origin[a];
    }
    int64_t Y = c->sc->dims[1] * c->cs, Z = c->sc->dims[2] * c->cs;
    return c->sc->grid[(g[0] * Y + g[1]) * Z + g[2]];
}

/* ------------------------------------------------------------------------------------------
 * Ray state and shaders
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int64_t color[3];
    double energy, step, life, bounces;
    double pos[3], vel[3];
    /* traversed: list of chunk tuples in first-visit order (init.py:58, 72-73) */
    int64_t (*trav)[3];
    int ntrav, trav_cap;
    int32_t cnt[ORC_NCOUNTERS];
} ray_t;

static double rnd(mt_state* rng, ray_t* ray) {
    ray->cnt[ORC_C_DRAW]++;
    return mt_random(rng);
}

/* lib.rand (lib.py:431-434): no draw when amp is falsy */
static double py_rand(mt_state* rng, ray_t* ray, double amp) {
    if (amp == 0.0) return 0.0;
    return (-1 + rnd(rng, ray) * 2) * amp;
}

/* lib.mix (lib.py:437-439) */
static double py_mix(double v1, double v2, double b1) {
    double b2 = 1 - b1;
    return v1 * b2 + v2 * b1;
}

/* rgb.mix (lib.py:393-395): per channel round(self * bias2 + col * bias1) */
static void rgb_mix(int64_t col[3], const double other[3], double b1) {
    double b2 = 1 - b1;
    for (int i = 0; i < 3; i++) col[i] = (int64_t)py_round((double)col[i] * b2 + other[i] * b1);
}

/* lib.material (lib.py:448-460); returns the bounce amount */
static double shade_material(const ctx_t* c, ray_t* ray, const double* mat, mt_state* rng) {
    double absorption = mat[4] / c->lm.pow_(1 + ray->bounces, 1 + c->st->falloff);
    if (!(absorption < 1)) absorption = 1; /* min(1, x) */
    rgb_mix(ray->color, mat, absorption);
    ray->energy = py_mix(ray->energy, mat[6], absorption);
    ray->life *= 1 - (mat[3] * absorption);
    double jx = py_rand(rng, ray, mat[3]);
    double jy = py_rand(rng, ray, mat[3]);
    double jz = py_rand(rng, ray, mat[3]);
    ray->vel[0] += jx;
    ray->vel[1] += jy;
    ray->vel[2] += jz;
    return mat[4];
}

/* lib.material_background (lib.py:463-476) */
static void shade_background(const ctx_t* c, ray_t* ray) {
    double absorption = 1 / c->lm.pow_(1 + ray->bounces, 1 + c->st->falloff);
    if (!(absorption < 1)) absorption = 1;
    double up = ray->vel[1] > 0 ? ray->vel[1] : 0; /* max(0, +vel.y) */
    double sky[3] = {127, 127 + up * 64, 127 + up * 128};
    double energy = 1 + up;
    rgb_mix(ray->color, sky, absorption);
    ray->energy = py_mix(ray->energy, energy, absorption);
    for (int i = 0; i < 3; i++) {
        double v = py_round((double)ray->color[i] * ray->energy);
        ray->color[i] = v < 255 ? (int64_t)v : 255; /* min(255, round(...)) */
    }
}

static void trav_append(ray_t* ray, const double cmin[3]) {
    int64_t p[3] = {(int64_t)cmin[0], (int64_t)cmin[1], (int64_t)cmin[2]};
    for (int i = 0; i < ray->ntrav; i++)
        if (ray->trav[i][0] == p[0] && ray->trav[i][1] == p[1] && ray->trav[i][2] == p[2]) return;
    if (ray->ntrav == ray->trav_cap) {
        ray->trav_cap = ray->trav_cap ? ray->trav_cap * 2 : 32;
        ray->trav = realloc(ray->trav, sizeof(int64_t[3]) * (size_t)ray->trav_cap);
    }
    memcpy(ray->trav[ray->ntrav++], p, sizeof p);
}

/* vec3 >= / <= (lib.py:228-244): all components */
static int ge3(const double a[3], const double b[3]) { return a[0] >= b[0] && a[1] >= b[1] && a[2] >= b[2]; }
static int le3(const double a[3], const double b[3]) { return a[0] <= b[0] && a[1] <= b[1] && a[2] <= b[2]; }

/* vec3.snapped(unit) (lib.py:316-320): (v // unit) * unit */
static void snapped(const double p[3], double unit, double out[3]) {
    for (int a = 0; a < 3; a++) out[a] = py_floordiv(p[a], unit) * unit;
}

/* Camera.trace (init.py:37-121) */
static void trace(const ctx_t* c, double dir_x, double dir_y, double detail, mt_state* rng, ray_t* ray) {
    const orc_settings* st = c->st;
    const libm_t* lm = &c->lm;
    /* init.py:41-45: lens angles, quaternion, forward vector */
    double lens_x = (dir_x / st->proportions) * c->cam->lens + py_rand(rng, ray, st->dof);
    double lens_y = (dir_y * st->proportions) * c->cam->lens + py_rand(rng, ray, st->dof);
    double ex = 0.0, ey = -lens_x, ez = +lens_y;
    /* vec3.quaternion (lib.py:322-338); math.radians(x) = x * (pi / 180) */
    const double deg2rad = 3.141592653589793 / 180.0;
    double rad_x = ex * deg2rad, rad_y = ey * deg2rad, rad_z = ez * deg2rad;
    double sin_x = lm->sin_(rad_x / 2), cos_x = lm->cos_(rad_x / 2);
    double sin_y = lm->sin_(rad_y / 2), cos_y = lm->cos_(rad_y / 2);
    double sin_z = lm->sin_(rad_z / 2), cos_z = lm->cos_(rad_z / 2);
    double ox = sin_x * cos_y * cos_z - cos_x * sin_y * sin_z;
    double oy = cos_x * sin_y * cos_z - sin_x * cos_y * sin_z;
    double oz = cos_x * cos_y * sin_z + sin_x * sin_y * cos_z;
    double ow = cos_x * cos_y * cos_z + sin_x * sin_y * sin_z;
    /* quaternion.multiply (lib.py:353-358): self = camera rotation, other = lens quaternion */
    const double* q = c->cam->rot; /* x y z w */
    double rx = q[3] * ox + q[2] * oy - q[1] * oz + q[0] * ow;
    double ry = q[2] * ox + q[3] * oy + q[0] * oz + q[1] * ow;
    double rz = q[1] * ox - q[0] * oy + q[3] * oz + q[2] * ow;
    double rw = q[0] * ox - q[1] * oy - q[2] * oz + q[3] * ow;
    /* quaternion.vec_forward (lib.py:372-376) */
    double dir[3];
    dir[0] = 2 * (rz * rx + rw * ry);
    dir[1] = 2 * (ry * rx - rw * rz);
    dir[2] = 1 - 2 * (sq(lm, rz) + sq(lm, ry));

    double chunk_min[3] = {0, 0, 0}, chunk_max[3] = {0, 0, 0};
    int64_t chunk = -1;

    /* init.py:50-59 */
    ray->color[0] = ray->color[1] = ray->color[2] = 0;
    ray->energy = 0;
    for (int a = 0; a < 3; a++) {
        ray->pos[a] = c->cam->pos[a] + dir[a] * st->dist_min;
        ray->vel[a] = dir[a];
    }
    ray->step = 0;
    ray->life = (st->dist_max - st->dist_min) * detail;
    ray->bounces = 0;
    ray->ntrav = 0;
    double cs = (double)st->chunk_size;
    int pending_hit = 0;

    while (ray->step < ray->life) { /* init.py:66 */
        if (!ge3(ray->pos, chunk_min) || !le3(ray->pos, chunk_max)) { /* init.py:67-73 */
            snapped(ray->pos, cs, chunk_min);
            for (int a = 0; a < 3; a++) chunk_max[a] = chunk_min[a] + cs;
            chunk = chunk_index(c, chunk_min);
            trav_append(ray, chunk_min);
            ray->cnt[ORC_C_RESNAP]++;
        }
        if (chunk >= 0) { /* init.py:75-77 */
            int64_t fp[3] = {(int64_t)floor(ray->pos[0]), (int64_t)floor(ray->pos[1]), (int64_t)floor(ray->pos[2])};
            int id = orc_lookup(c, chunk, chunk_min, fp);
            ray->cnt[ORC_C_LOOKUP]++;
            if (id) {
                const double* mat = c->sc->materials + (size_t)(id - 1) * 7;
                /* init.py:81-86 */
                double bounce = shade_material(c, ray, mat, rng);
                ray->cnt[ORC_C_HIT]++;
                pending_hit = 1;
                ray->bounces += bounce;
                ray->life /= (double)c->sc->res[chunk] + bounce * st->lod_bounces;
                { /* vec3.normalize (lib.py:310-314) */
                    double ref = fmax(fmax(fabs(ray->vel[0]), fabs(ray->vel[1])), fabs(ray->vel[2]));
                    if (ref != 0.0 && ref != 1.0)
                        for (int a = 0; a < 3; a++) ray->vel[a] = ray->vel[a] / ref;
                }
                if (ray->step >= ray->life || ray->energy >= st->max_light || ray->bounces >= st->max_bounces + 1)
                    break;
                double ior = mat[5];
                if (ior != 0.0) { /* init.py:92-111 */
                    double direction = (ior - 0.5) * 2;
                    int solid[3];
                    for (int a = 0; a < 3; a++) {
                        double np[3] = {ray->pos[0] + 0, ray->pos[1] + 0, ray->pos[2] + 0};
                        if (ray->vel[a] < direction) np[a] = ray->pos[a] + 1; else np[a] = ray->pos[a] - 1;
                        /* unchanged components are pos + 0 / pos - 0 in the reference: value-preserving */
                        int64_t nfp[3] = {(int64_t)floor(np[0]), (int64_t)floor(np[1]), (int64_t)floor(np[2])};
                        int64_t nchunk;
                        double ncmin[3];
                        if (ge3(np, chunk_min) && le3(np, chunk_max)) {
                            nchunk = chunk;
                            memcpy(ncmin, chunk_min, sizeof ncmin);
                        } else { /* Camera.chunk_get (init.py:28-33) */
                            snapped(np, cs, ncmin);
                            nchunk = chunk_index(c, ncmin);
                            ray->cnt[ORC_C_CHUNK_GET]++;
                        }
                        int nid = 0;
                        if (nchunk >= 0) {
                            nid = orc_lookup(c, nchunk, ncmin, nfp);
                            ray->cnt[ORC_C_NBR]++;
                        }
                        solid[a] = nid && c->sc->materials[(size_t)(nid - 1) * 7 + 5] == ior;
                    }
                    for (int a = 0; a < 3; a++)
                        if (!solid[a]) ray->vel[a] -= ray->vel[a] * ior * 2;
                }
            }
        }
        /* init.py:114-116 */
        double stepsize;
        if (chunk >= 0) {
            stepsize = (double)c->sc->res[chunk];
        } else {
            double mn = fmin(fmin(ray->pos[0], ray->pos[1]), ray->pos[2]);
            stepsize = 1 + fabs((double)st->chunk_radius - py_mod(mn + (double)st->chunk_radius, cs));
        }
        ray->step += stepsize;
        for (int a = 0; a < 3; a++) ray->pos[a] += ray->vel[a] * stepsize;
        ray->cnt[ORC_C_ADV]++;
        pending_hit = 0;
    }
    ray->cnt[ORC_C_BROKE] = pending_hit;
    if (st->has_background) shade_background(c, ray); /* init.py:119-120 */
}

/* ------------------------------------------------------------------------------------------
 * Camera.tile (init.py:126-150)
 * ---------------------------------------------------------------------------------------- */
static void pixel_dir(const orc_settings* st, int32_t x, int32_t y, double* dir_x, double* dir_y, double* detail,
                      int32_t* n) {
    *dir_x = -1 + ((double)x / (double)st->width) * 2;
    *dir_y = -1 + ((double)y / (double)st->height) * 2;
    *detail = 1 - fabs(*dir_x * *dir_y) * st->lod_edge;
    double r = py_round((double)st->samples * *detail);
    *n = r > 1 ? (int32_t)r : 1;
}

int32_t orc_pixel_samples(const orc_settings* st, int32_t x, int32_t y) {
    double a, b, d;
    int32_t n;
    pixel_dir(st, x, y, &a, &b, &d, &n);
    return n;
}

typedef struct {
    int64_t key_ray, key_seq; /* first visit: global ray index, position in that ray's list */
    int64_t pos[3];
} trav_rec;

typedef struct {
    ctx_t ctx;
    const int32_t* pixels;
    int64_t n_px;
    const int64_t* ray_offset; /* [n_px] first global ray index of each pixel */
    int tid, nthreads;
    double* pix_mean;
    uint8_t* pix_rgba8;
    orc_ray* rays;
    int64_t counters[ORC_NCOUNTERS];
    trav_rec* trav;
    int64_t ntrav, trav_cap;
    int want_trav;
} worker_t;

static void worker_trav_merge(worker_t* w, const ray_t* ray, int64_t ray_index) {
    for (int i = 0; i < ray->ntrav; i++) {
        int found = 0;
        for (int64_t k = 0; k < w->ntrav; k++) {
            if (w->trav[k].pos[0] == ray->trav[i][0] && w->trav[k].pos[1] == ray->trav[i][1] &&
                w->trav[k].pos[2] == ray->trav[i][2]) { found = 1; break; }
        }
        if (found) continue;
        if (w->ntrav == w->trav_cap) {
            w->trav_cap = w->trav_cap ? w->trav_cap * 2 : 256;
            w->trav = realloc(w->trav, sizeof(trav_rec) * (size_t)w->trav_cap);
        }
        trav_rec* t = &w->trav[w->ntrav++];
        t->key_ray = ray_index;
        t->key_seq = i;
        memcpy(t->pos, ray->trav[i], sizeof t->pos);
    }
}

static void* worker_main(void* arg) {
    worker_t* w = (worker_t*)arg;
    const orc_settings* st = w->ctx.st;
    mt_state rng;
    ray_t ray;
    memset(&ray, 0, sizeof ray);
    for (int64_t i = w->tid; i < w->n_px; i += w->nthreads) {
        int32_t x = w->pixels[2 * i], y = w->pixels[2 * i + 1];
        double dir_x, dir_y, detail;
        int32_t n;
        pixel_dir(st, x, y, &dir_x, &dir_y, &detail, &n);
        double sum[4] = {0, 0, 0, 0};
        for (int32_t s = 0; s < n; s++) {
            /* static: random.seed((1 + x) * (1 + y) * (1 + sample)) (init.py:137).  Non-static (seed_nonce != 0): the
             * reference seeds nothing per sample and re-seeds from OS entropy after every pixel (init.py:136-147), so
             * there is nothing to reproduce; the build gives every ray slot a stream of its own,
             * (y * width + x) * max_samples + sample + nonce, and this restates that rule. */
            unsigned __int128 seed = (unsigned __int128)(1 + (uint64_t)x) * (1 + (uint64_t)y);
            seed *= (1 + (uint64_t)s);
            uint64_t lo = (uint64_t)seed, hi = (uint64_t)(seed >> 64);
            if (st->seed_nonce) {
                double dmax = st->lod_edge < 0 ? 1 - st->lod_edge : 1;
                double rm = py_round((double)st->samples * dmax);
                uint64_t smax = rm > 1 ? (uint64_t)rm : 1;
                lo = ((uint64_t)y * (uint64_t)st->width + (uint64_t)x) * smax + (uint64_t)s + st->seed_nonce;
                hi = 0;
            }
            mt_seed_u128(&rng, lo, hi);
            memset(ray.cnt, 0, sizeof ray.cnt);
            /* init.py:139 */
            double ray_detail = detail / (1 + s * st->lod_samples) * (1 - st->lod_random * rnd(&rng, &ray));
            trace(&w->ctx, dir_x, dir_y, ray_detail, &rng, &ray);
            /* init.py:141-142 */
            double e = ray.energy + st->shutter;
            if (!(e < 1)) e = 1;
            int32_t alpha = (int32_t)py_round(e * 255);
            /* average() accumulates in list order (lib.py:419-428); ints, so exact */
            sum[0] += (double)ray.color[0];
            sum[1] += (double)ray.color[1];
            sum[2] += (double)ray.color[2];
            sum[3] += (double)alpha;
            for (int k = 0; k < ORC_NCOUNTERS; k++) w->counters[k] += ray.cnt[k];
            int64_t gi = w->ray_offset[i] + s;
            if (w->rays) {
                orc_ray* o = &w->rays[gi];
                o->x = x; o->y = y; o->s = s;
                for (int k = 0; k < 3; k++) { o->color[k] = (int32_t)ray.color[k]; o->pos[k] = ray.pos[k]; o->vel[k] = ray.vel[k]; }
                o->alpha = alpha;
                o->ntrav = ray.ntrav;
                memcpy(o->counters, ray.cnt, sizeof ray.cnt);
                o->detail = ray_detail; o->energy = ray.energy; o->step = ray.step; o->life = ray.life;
                o->bounces = ray.bounces;
            }
            if (w->want_trav) worker_trav_merge(w, &ray, gi);
        }
        for (int k = 0; k < 4; k++) {
            double m = sum[k] / (double)n;
            w->pix_mean[4 * i + k] = m;
            /* Surface.set_at float -> u8: truncation assumed (pygame unavailable; exact when spp == 1) */
            if (w->pix_rgba8) w->pix_rgba8[4 * i + k] = (uint8_t)(m < 0 ? 0 : (m > 255 ? 255 : (int)m));
        }
    }
    free(ray.trav);
    return NULL;
}

static int trav_cmp(const void* a, const void* b) {
    const trav_rec* x = (const trav_rec*)a;
    const trav_rec* y = (const trav_rec*)b;
    if (x->key_ray != y->key_ray) return x->key_ray < y->key_ray ? -1 : 1;
    if (x->key_seq != y->key_seq) return x->key_seq < y->key_seq ? -1 : 1;
    return 0;
}

int orc_render(const orc_scene* scene, const orc_settings* st, const orc_camera* cam, const int32_t* pixels_xy,
               int64_t n_px, int libm_mode, int n_threads, double* pix_mean, uint8_t* pix_rgba8, orc_ray* rays,
               int64_t rays_cap, int64_t* n_rays, int64_t* counters, int64_t* traversed, int64_t trav_cap,
               int64_t* n_trav) {
    if (!scene || !st || !cam || !pix_mean || n_px < 0 || st->chunk_size <= 0) return -1;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    int64_t* off = malloc(sizeof(int64_t) * (size_t)(n_px + 1));
    int64_t total = 0;
    for (int64_t i = 0; i < n_px; i++) {
        off[i] = total;
        total += orc_pixel_samples(st, pixels_xy[2 * i], pixels_xy[2 * i + 1]);
    }
    if (n_rays) *n_rays = total;
    if (rays && total > rays_cap) { free(off); return -2; }
    worker_t* ws = calloc((size_t)n_threads, sizeof(worker_t));
    pthread_t* th = malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int t = 0; t < n_threads; t++) {
        worker_t* w = &ws[t];
        w->ctx.sc = scene; w->ctx.st = st; w->ctx.cam = cam; w->ctx.lm = libm_get(libm_mode);
        w->ctx.cs = st->chunk_size;
        w->pixels = pixels_xy; w->n_px = n_px; w->ray_offset = off; w->tid = t; w->nthreads = n_threads;
        w->pix_mean = pix_mean; w->pix_rgba8 = pix_rgba8; w->rays = rays; w->want_trav = traversed != NULL;
        if (n_threads > 1) pthread_create(&th[t], NULL, worker_main, w); else worker_main(w);
    }
    int rc = 0;
    int64_t nt = 0;
    trav_rec* all = NULL;
    if (counters) memset(counters, 0, sizeof(int64_t) * ORC_NCOUNTERS);
    for (int t = 0; t < n_threads; t++) {
        if (n_threads > 1) pthread_join(th[t], NULL);
        if (counters) for (int k = 0; k < ORC_NCOUNTERS; k++) counters[k] += ws[t].counters[k];
        if (ws[t].ntrav) {
            all = realloc(all, sizeof(trav_rec) * (size_t)(nt + ws[t].ntrav));
            memcpy(all + nt, ws[t].trav, sizeof(trav_rec) * (size_t)ws[t].ntrav);
            nt += ws[t].ntrav;
        }
        free(ws[t].trav);
    }
    if (traversed) {
        /* order-preserving union (lib.merge, lib.py:404-409): sort by first visit, keep first of each chunk */
        qsort(all, (size_t)nt, sizeof(trav_rec), trav_cmp);
        int64_t out = 0;
        for (int64_t i = 0; i < nt; i++) {
            int dup = 0;
            for (int64_t k = 0; k < out; k++)
                if (traversed[3 * k] == all[i].pos[0] && traversed[3 * k + 1] == all[i].pos[1] &&
                    traversed[3 * k + 2] == all[i].pos[2]) { dup = 1; break; }
            if (dup) continue;
            if (out == trav_cap) { rc = -2; break; }
            memcpy(&traversed[3 * out], all[i].pos, sizeof all[i].pos);
            out++;
        }
        if (n_trav) *n_trav = out;
    }
    free(all);
    free(ws);
    free(th);
    free(off);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Window.chunk_update selection loop (init.py:447-452)
 * ---------------------------------------------------------------------------------------- */
void orc_select_chunks(const int64_t* origin, const int64_t* dims, int32_t cs, int32_t chunk_radius,
                       const uint8_t* world_present, const double* cam_pos, double dist_max, int32_t chunk_lod,
                       int32_t culling, const int64_t* traversed, int64_t n_trav, uint8_t* out_present, uint8_t* out_res) {
    for (int64_t cx = 0; cx < dims[0]; cx++)
        for (int64_t cy = 0; cy < dims[1]; cy++)
            for (int64_t cz = 0; cz < dims[2]; cz++) {
                int64_t idx = (cx * dims[1] + cy) * dims[2] + cz;
                out_present[idx] = 0;
                out_res[idx] = 0;
                if (!world_present[idx]) continue;
                int64_t post[3] = {origin[0] + cx * cs, origin[1] + cy * cs, origin[2] + cz * cs};
                int in_trav = 0;
                for (int64_t k = 0; k < n_trav && !in_trav; k++)
                    in_trav = traversed[3 * k] == post[0] && traversed[3 * k + 1] == post[1] && traversed[3 * k + 2] == post[2];
                if (culling && !in_trav) continue; /* init.py:447 */
                /* init.py:448-449 */
                double dx = (double)(post[0] + chunk_radius) - cam_pos[0];
                double dy = (double)(post[1] + chunk_radius) - cam_pos[1];
                double dz = (double)(post[2] + chunk_radius) - cam_pos[2];
                double dist = sqrt(dx * dx + dy * dy + dz * dz);
                double q = trunc(dist / (dist_max / (double)(1 + chunk_lod)));
                int lod = q < (double)chunk_lod ? (int)q : chunk_lod;
                out_present[idx] = 1;
                out_res[idx] = (uint8_t)(lod + 1);
            }
}

/* ------------------------------------------------------------------------------------------
 * Window.draw_tile (init.py:185-190): canvas.blit(tile) of two SRCALPHA surfaces.
 * PARITY UNPINNED: pygame is not installable in the build environment; this restates the ALPHA_BLEND macro of
 * pygame 2's src_c/surface.h as published.
 * ---------------------------------------------------------------------------------------- */
void orc_canvas_blit(uint8_t* canvas, const uint8_t* tile, int64_t n_pixels) {
    for (int64_t i = 0; i < n_pixels; i++) {
        uint8_t* d = canvas + 4 * i;
        const uint8_t* s = tile + 4 * i;
        if (d[3] == 0) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3]; continue; }
        int sa = s[3];
        for (int c = 0; c < 3; c++) d[c] = (uint8_t)((((int)d[c] << 8) + ((int)s[c] - (int)d[c]) * sa + (int)s[c]) >> 8);
        d[3] = (uint8_t)(sa + (int)d[3] - (sa * (int)d[3]) / 255);
    }
}
