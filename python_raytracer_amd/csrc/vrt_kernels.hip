// vrt_kernels.hip -- MI355X (gfx950) voxel ray-march: the reference's Camera.tile -> Camera.trace ->
// Frame.get_voxel -> lib.material / lib.material_background path (reference init.py:37-150,
// data.py:136-145, lib.py:448-476) as hand-written HIP behind the C ABI of include/vrt.h.
//
// Three kernels per batch of rays:
//   rng_tile_kernel   MT19937 init_by_array + first D draws per (pixel, sample), all in registers
//                     (CPython random.seed(int) / random.random(), reference init.py:137,139; lib.py:434)
//   march_kernel      one lane per ray: ray generation, chunk/voxel march, PBR shading, reflection, sky
//   resolve_kernel    per-pixel mean of the samples (lib.average, init.py:145) -> fp32 RGBA + RGBA8
//
// Arithmetic is binary64 in the reference's evaluation order; build with -ffp-contract=off.
// gfx950 only: 64-wide waves are assumed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/vrt.h"
#include "vrt_math.h"

#pragma clang fp contract(off)

#define VRT_WAVE 64
#define VRT_BLOCK 256

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local int g_last_hip_error = 0;
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e__ = (expr);                        \
        if (e__ != hipSuccess) {                        \
            g_last_hip_error = (int)e__;                \
            return VRT_ERR_HIP;                         \
        }                                               \
    } while (0)

// optional per-kernel timing (vrt_profile_begin / vrt_profile_end): HIP events on the launch stream
#include <vector>
struct ProfEvent { hipEvent_t a, b; int kind; };
static bool g_prof_on = false;
static std::vector<ProfEvent> g_prof;
struct ProfScope {
    hipStream_t s;
    bool on;
    ProfEvent e;
    ProfScope(hipStream_t stream, int kind) : s(stream), on(g_prof_on) {
        if (!on) return;
        e.kind = kind;
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(e.a, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(e.b, s);
        g_prof.push_back(e);
    }
};

// ---------------------------------------------------------------------------------------------
// shared device helpers
// ---------------------------------------------------------------------------------------------
__host__ __device__ static inline int64_t voxel_offset(int cs, int lx, int ly, int lz) {
    int nb = cs >> 3;
    int brick = (((lx >> 3) * nb) + (ly >> 3)) * nb + (lz >> 3);
    int micro = ((((lx >> 2) & 1) * 2) + ((ly >> 2) & 1)) * 2 + ((lz >> 2) & 1);
    int vox = (((lx & 3) * 4) + (ly & 3)) * 4 + (lz & 3);
    return (int64_t)brick * 512 + micro * 64 + vox;
}

// tile(): direction, detail and sample count of pixel (x, y)  (init.py:131-134)
__host__ __device__ static inline void pixel_setup(const vrt_settings& st, int x, int y, double& dir_x, double& dir_y,
                                                   double& detail, int& n) {
    dir_x = -1 + ((double)x / (double)st.width) * 2;
    dir_y = -1 + ((double)y / (double)st.height) * 2;
    detail = 1 - __builtin_fabs(dir_x * dir_y) * st.lod_edge;
    double r = __builtin_rint((double)st.samples * detail);
    n = r > 1 ? (int)r : 1;
}

// ---------------------------------------------------------------------------------------------
// MT19937 seeding + first draws, register-resident
// ---------------------------------------------------------------------------------------------
struct MtInit { uint32_t v[624]; };
static constexpr MtInit make_mt_init() {  // init_genrand(19650218)
    MtInit t{};
    t.v[0] = 19650218u;
    for (int i = 1; i < 624; i++) t.v[i] = 1812433253u * (t.v[i - 1] ^ (t.v[i - 1] >> 30)) + (uint32_t)i;
    return t;
}
__constant__ MtInit c_mt_init = make_mt_init();

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
// genrand output k < 227 from the seeded words: new mt[k] = s[k+397] ^ twist(s[k], s[k+1])
__device__ __forceinline__ uint32_t mt_out(uint32_t sk, uint32_t sk1, uint32_t sk397) {
    uint32_t y = (sk & 0x80000000u) | (sk1 & 0x7fffffffu);
    uint32_t v = sk397 ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    return mt_temper(v);
}
__device__ __forceinline__ double mt_res53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// Seeds with the 64-bit integer `seed` exactly like random.seed(seed) and writes the first D results of
// random.random() to out[d * stride].  init_by_array's two dependent sweeps over the 624-word state are
// evaluated as three register-only chain passes (the second sweep re-derives the first sweep's words on
// the fly), and only the seeded words the first 2*D outputs depend on are kept.
template <int D>
__device__ __forceinline__ void mt_seed_draws(uint64_t seed, double* out, int64_t stride) {
    constexpr int K = 2 * D;
    static_assert(K <= 226, "first-227 outputs only");
    const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
    const uint32_t addA = key0;                          // j = 0: key[0] + 0
    const uint32_t addB = key1 ? key1 + 1u : key0;       // j = 1 (len 2): key[1] + 1, else j stays 0
    const uint32_t* init = c_mt_init.v;
    // ---- pass A: first sweep (multiplier 1664525), t-th step writes index i = t + 1 ----
    uint32_t m = init[0];
    m = (init[1] ^ ((m ^ (m >> 30)) * 1664525u)) + addA;
    const uint32_t first1 = m;  // mt[1] after the first sweep's first step
    for (int i = 2; i < 624; i++) m = (init[i] ^ ((m ^ (m >> 30)) * 1664525u)) + (((i - 1) & 1) ? addB : addA);
    // step 624: i wrapped to 1, mt[0] = mt[623]
    const uint32_t mt1 = (first1 ^ ((m ^ (m >> 30)) * 1664525u)) + ((623 & 1) ? addB : addA);
    // ---- pass B: second sweep (multiplier 1566083941) from i = 2, re-deriving the first sweep ----
    uint32_t p = first1;  // first-sweep word i-1
    uint32_t q = mt1;     // second-sweep word i-1
    uint32_t keep[K + 1];  // seeded words s[2..K]
#define VRT_MT_STEP(i)                                                                  \
    p = (init[i] ^ ((p ^ (p >> 30)) * 1664525u)) + ((((i) - 1) & 1) ? addB : addA);      \
    q = (p ^ ((q ^ (q >> 30)) * 1566083941u)) - (uint32_t)(i);
#pragma unroll
    for (int i = 2; i <= K; i++) {
        VRT_MT_STEP(i)
        keep[i] = q;
    }
    for (int i = K + 1; i < 397; i++) { VRT_MT_STEP(i) }
    uint32_t s397 = 0, s398 = 0, prev = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int i = 397 + k;
        VRT_MT_STEP(i)
        if (k == 0) s397 = q;
        if (k == 1) s398 = q;
        if (k >= 2) {
            uint32_t o = mt_out(keep[k], k + 1 <= K ? keep[k + 1] : 0u, q);
            if (k & 1) out[(int64_t)(k >> 1) * stride] = mt_res53(prev, o);
            else prev = o;
        }
    }
    for (int i = 397 + K; i < 624; i++) { VRT_MT_STEP(i) }
#undef VRT_MT_STEP
    // wrap: mt[0] = mt[623]; final step writes index 1; then mt[0] = 0x80000000
    const uint32_t s1 = (mt1 ^ ((q ^ (q >> 30)) * 1566083941u)) - 1u;
    const uint32_t o0 = mt_out(0x80000000u, s1, s397);
    const uint32_t o1 = mt_out(s1, keep[2], s398);
    out[0] = mt_res53(o0, o1);
}

struct TileGeom {
    const int32_t* pixels;  // [n_px][2]
    int64_t n_px;
    int32_t smax;           // sample slots per pixel
};

// rays [ray0, ray0 + n) of the tile; table[d * tstride + (ray - ray0)]
template <int D>
__global__ void __launch_bounds__(VRT_BLOCK) rng_tile_kernel(vrt_settings st, TileGeom g, int64_t ray0, int64_t n,
                                                             double* table, int64_t tstride) {
    int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    int64_t ray = ray0 + i;
    int64_t p = ray / g.smax;
    int s = (int)(ray - p * g.smax);
    int x = g.pixels[2 * p], y = g.pixels[2 * p + 1];
    double dx, dy, det;
    int ns;
    pixel_setup(st, x, y, dx, dy, det, ns);
    if (s >= ns) return;
    uint64_t seed = (uint64_t)(1 + (int64_t)x) * (uint64_t)(1 + (int64_t)y) * (uint64_t)(1 + s) + st.seed_nonce;
    mt_seed_draws<D>(seed, table + i, tstride);
}

// retrace list variant: list[k] = ray offset inside the batch
template <int D>
__global__ void __launch_bounds__(VRT_BLOCK) rng_list_kernel(vrt_settings st, TileGeom g, int64_t ray0,
                                                             const uint32_t* list, const uint32_t* count,
                                                             double* table, int64_t tstride) {
    uint32_t n = *count;
    for (uint32_t k = blockIdx.x * VRT_BLOCK + threadIdx.x; k < n; k += gridDim.x * VRT_BLOCK) {
        int64_t ray = ray0 + list[k];
        int64_t p = ray / g.smax;
        int s = (int)(ray - p * g.smax);
        int x = g.pixels[2 * p], y = g.pixels[2 * p + 1];
        uint64_t seed = (uint64_t)(1 + (int64_t)x) * (uint64_t)(1 + (int64_t)y) * (uint64_t)(1 + s) + st.seed_nonce;
        mt_seed_draws<D>(seed, table + k, tstride);
    }
}

template <int D>
__global__ void __launch_bounds__(VRT_BLOCK) rng_seeds_kernel(const uint64_t* seeds, int64_t n, double* out) {
    int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    mt_seed_draws<D>(seeds[i], out + i, n);
}

// ---------------------------------------------------------------------------------------------
// march
// ---------------------------------------------------------------------------------------------
struct MarchParams {
    vrt_settings st;
    vrt_camera cam;
    // scene
    int64_t origin[3];
    int32_t dims[3];
    int32_t cs, cs_shift;
    int32_t n_materials;
    const uint32_t* chunk_table;
    const uint8_t* voxels;
    const double* materials;
    // traversed
    int64_t t_origin[3];
    int32_t t_dims[3];
    uint64_t* t_keys;
    // rays
    TileGeom g;              // tile mode
    const double* dir_x;     // explicit mode
    const double* dir_y;
    const double* detail;
    int64_t ray0, n;         // batch [ray0, ray0 + n)
    const uint32_t* list;    // optional retrace list (ray offsets in batch) and its device-side count
    const uint32_t* list_count;
    const double* draws;     // [n_draws][dstride]
    int64_t dstride;
    int32_t n_draws;
    // outputs
    uint32_t* ray_rgba;      // [rays] packed result (tile mode)
    vrt_ray* rays;           // debug records (may be NULL)
    uint64_t* stats;
    uint32_t* retrace_list;  // rays whose draws ran out are appended here (may be NULL)
    uint32_t* retrace_count;
};

struct Chunk {
    double mnx, mny, mnz;  // chunk_min (init.py:68); chunk_max = chunk_min + cs
    uint32_t entry;        // chunk table entry, 0 = None
};

struct RayCtx {
    const MarchParams* P;
    const double* mats;  // LDS copy of the material table
    const double* draws;
    int64_t draw_idx;    // column of this ray in the draw table
    int ndraw;
    bool exhausted;
    int32_t cnt[VRT_NCOUNTERS];
    volatile unsigned long long* pw_keys;  // LDS cache of pow(x, 1 + falloff): insert-only, per workgroup
    volatile unsigned long long* pw_vals;
};

// (1 + bounces) ** (1 + falloff) (lib.py:450, 465).  The exponent is fixed for a launch and the bases are sums of
// material absorptions -- a handful of distinct values -- so each workgroup memoises vrt_pow in LDS.  The table is
// insert-only (key: EMPTY -> bits once, by CAS; value: NOT_READY -> result once), hence a reader that sees its key
// and a ready value has the value vrt_pow returns for that key: results are identical to calling vrt_pow.
#define VRT_PW_SLOTS 256
#define VRT_PW_NOT_READY 0x7ff8dead0000beefull
__device__ __forceinline__ double pow_cached(RayCtx& c, double x, double y) {
    if (x == 1.0) return 1.0;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    unsigned h = (unsigned)(bits >> 32) * 0x9e3779b1u + (unsigned)bits * 0x85ebca6bu;
    h >>= 24;
    int free_slot = -1;
    unsigned long long found = VRT_PW_NOT_READY;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int sidx = (int)((h + i) & (VRT_PW_SLOTS - 1));
        const unsigned long long k = c.pw_keys[sidx];
        if (k == bits) {
            found = c.pw_vals[sidx];
            break;
        }
        if (k == 0ull) {
            free_slot = sidx;
            break;
        }
    }
    if (found != VRT_PW_NOT_READY) return __longlong_as_double((long long)found);
    const double v = vrt_pow(x, y);
    if (free_slot >= 0) {
        const unsigned long long old = atomicCAS((unsigned long long*)&c.pw_keys[free_slot], 0ull, bits);
        if (old == 0ull || old == bits) c.pw_vals[free_slot] = (unsigned long long)__double_as_longlong(v);
    }
    return v;
}

__device__ __forceinline__ double next_draw(RayCtx& c) {
    int k = c.ndraw++;
    if (k >= c.P->n_draws) {
        c.exhausted = true;
        return 0.5;
    }
    return c.draws[(int64_t)k * c.P->dstride + c.draw_idx];
}
// lib.rand (lib.py:431-434)
__device__ __forceinline__ double py_rand(RayCtx& c, double amp) {
    if (amp == 0.0) return 0.0;
    return (-1 + next_draw(c) * 2) * amp;
}

// chunks.get(snapped(pos)) (init.py:68-71 / 28-33): cmin are integral doubles
__device__ __forceinline__ uint32_t chunk_entry(const MarchParams& P, double mnx, double mny, double mnz) {
    int64_t cx = ((int64_t)mnx - P.origin[0]) >> P.cs_shift;
    int64_t cy = ((int64_t)mny - P.origin[1]) >> P.cs_shift;
    int64_t cz = ((int64_t)mnz - P.origin[2]) >> P.cs_shift;
    if ((uint64_t)cx >= (uint64_t)P.dims[0] || (uint64_t)cy >= (uint64_t)P.dims[1] ||
        (uint64_t)cz >= (uint64_t)P.dims[2])
        return 0;
    return P.chunk_table[(cx * P.dims[1] + cy) * P.dims[2] + cz];
}

__device__ __forceinline__ int snap_res(int f, int res) {
    if (res == 1) return f;
    if (res == 2) return f & ~1;
    return (int)__builtin_floor((double)f / (double)res) * res;  // int // int, exact (|f| < 2^31, res <= 255)
}

// Frame.get_voxel(floor(pos)) (data.py:136-145) on the packed chunk block: cell (fp // res) * res, which only
// exists inside the chunk's own half-open box
__device__ __forceinline__ int lookup(const MarchParams& P, const Chunk& ch, int fx, int fy, int fz) {
    int res = (int)(ch.entry >> 24);
    int lx = snap_res(fx, res) - (int)ch.mnx;
    int ly = snap_res(fy, res) - (int)ch.mny;
    int lz = snap_res(fz, res) - (int)ch.mnz;
    if ((unsigned)lx >= (unsigned)P.cs || (unsigned)ly >= (unsigned)P.cs || (unsigned)lz >= (unsigned)P.cs) return 0;
    int64_t slot = (int64_t)(ch.entry & 0xffffffu) - 1;
    return P.voxels[slot * ((int64_t)P.cs * P.cs * P.cs) + voxel_offset(P.cs, lx, ly, lz)];
}

__device__ __forceinline__ void trav_visit(const MarchParams& P, RayCtx& c, double mnx, double mny, double mnz,
                                           uint64_t key) {
    if (!P.t_keys) return;
    int64_t cx = ((int64_t)mnx - P.t_origin[0]) >> P.cs_shift;
    int64_t cy = ((int64_t)mny - P.t_origin[1]) >> P.cs_shift;
    int64_t cz = ((int64_t)mnz - P.t_origin[2]) >> P.cs_shift;
    if ((uint64_t)cx >= (uint64_t)P.t_dims[0] || (uint64_t)cy >= (uint64_t)P.t_dims[1] ||
        (uint64_t)cz >= (uint64_t)P.t_dims[2]) {
        atomicAdd((unsigned long long*)&P.stats[VRT_S_TRAV_OUTSIDE], 1ull);
        return;
    }
    uint64_t* slot = &P.t_keys[(cx * P.t_dims[1] + cy) * P.t_dims[2] + cz];
    // keys only decrease, so a stale (larger) value read here can only cause a redundant atomic
    if (key < *slot) atomicMin((unsigned long long*)slot, (unsigned long long)key);
}

struct RayOut {
    int cr, cg, cb;
    double energy, step, life, bounces;
    double px, py, pz, vx, vy, vz;
    int ntrav;
};

// Camera.trace (init.py:37-121) + the background call; ray_order = index in the reference's call order
template <bool RECORD>
__device__ __forceinline__ void trace_ray(RayCtx& c, double dir_x, double dir_y, double detail, uint64_t ray_order,
                                          RayOut& o) {
    const MarchParams& P = *c.P;
    const vrt_settings& st = P.st;
    // ---- ray generation (init.py:41-45; lib.py:322-338, 353-358, 372-376) ----
    const double lens_x = (dir_x / st.proportions) * P.cam.lens + py_rand(c, st.dof);
    const double lens_y = (dir_y * st.proportions) * P.cam.lens + py_rand(c, st.dof);
    const double deg2rad = 3.141592653589793 / 180.0;  // math.radians
    const double rad_y = (-lens_x) * deg2rad, rad_z = lens_y * deg2rad;
    const double sin_x = 0.0, cos_x = 1.0;  // sin(0.0 / 2), cos(0.0 / 2)
    const double sin_y = vrt_sin(rad_y / 2), cos_y = vrt_cos(rad_y / 2);
    const double sin_z = vrt_sin(rad_z / 2), cos_z = vrt_cos(rad_z / 2);
    const double ox = sin_x * cos_y * cos_z - cos_x * sin_y * sin_z;
    const double oy = cos_x * sin_y * cos_z - sin_x * cos_y * sin_z;
    const double oz = cos_x * cos_y * sin_z + sin_x * sin_y * cos_z;
    const double ow = cos_x * cos_y * cos_z + sin_x * sin_y * sin_z;
    const double qx = P.cam.rot[0], qy = P.cam.rot[1], qz = P.cam.rot[2], qw = P.cam.rot[3];
    const double rx = qw * ox + qz * oy - qy * oz + qx * ow;
    const double ry = qz * ox + qw * oy + qx * oz + qy * ow;
    const double rz = qy * ox - qx * oy + qw * oz + qz * ow;
    const double rw = qx * ox - qy * oy - qz * oz + qw * ow;
    (void)rw;
    double vx = 2 * (rz * rx + rw * ry);
    double vy = 2 * (ry * rx - rw * rz);
    double vz = 1 - 2 * (rz * rz + ry * ry);  // z ** 2 + y ** 2, correctly rounded squares

    // ---- ray store (init.py:50-59) ----
    double px = P.cam.pos[0] + vx * st.dist_min;
    double py = P.cam.pos[1] + vy * st.dist_min;
    double pz = P.cam.pos[2] + vz * st.dist_min;
    int cr = 0, cg = 0, cb = 0;
    double energy = 0, step = 0, bounces = 0;
    double life = (st.dist_max - st.dist_min) * detail;
    const double cs = (double)P.cs;
    const double inv_cs = 1.0 / cs;  // cs is a power of two: x * inv_cs == x / cs exactly
    Chunk ch;
    ch.mnx = ch.mny = ch.mnz = 0;
    ch.entry = 0;
    double mxx = 0, mxy = 0, mxz = 0;  // chunk_max
    int resnaps = 0;
    bool pending_hit = false;
    // RECORD: the ray's own traversed list, to report its length (init.py:72-73)
    int64_t seen[RECORD ? 48 : 1];
    int nseen = 0;

    // The reference's single loop (init.py:66-116) is split into two phases so that a wave does not pay the shading
    // path on every step: phase A marches (lookup + advance only) until THIS ray finds a material or runs out of
    // life; phase B shades, tests termination, reflects and advances once.  Per-ray semantics are unchanged.
    for (;;) {
        int id = 0;
        // ---------------- phase A: init.py:66-77, 114-116 for steps without a material ----------------
        while (step < life) {
            if (!(px >= ch.mnx && py >= ch.mny && pz >= ch.mnz) || !(px <= mxx && py <= mxy && pz <= mxz)) {
                // snapped(): (v // cs) * cs, exact for a power-of-two cs (init.py:68-73)
                ch.mnx = __builtin_floor(px * inv_cs) * cs;
                ch.mny = __builtin_floor(py * inv_cs) * cs;
                ch.mnz = __builtin_floor(pz * inv_cs) * cs;
                mxx = ch.mnx + cs;
                mxy = ch.mny + cs;
                mxz = ch.mnz + cs;
                ch.entry = chunk_entry(P, ch.mnx, ch.mny, ch.mnz);
                trav_visit(P, c, ch.mnx, ch.mny, ch.mnz, (ray_order << 12) | (uint64_t)(resnaps < 4095 ? resnaps : 4095));
                resnaps++;
                c.cnt[VRT_C_RESNAP]++;
                if (RECORD) {
                    int64_t cid = (((int64_t)ch.mnx >> P.cs_shift) * 2097152 + ((int64_t)ch.mny >> P.cs_shift)) * 2097152 +
                                  ((int64_t)ch.mnz >> P.cs_shift);
                    bool dup = false;
                    for (int k = 0; k < nseen && k < 48; k++) dup |= (seen[k] == cid);
                    if (!dup) {
                        if (nseen < 48) seen[nseen] = cid;
                        nseen++;
                    }
                }
            }
            double stepsize;
            if (ch.entry) {  // init.py:75-77
                id = lookup(P, ch, (int)__builtin_floor(px), (int)__builtin_floor(py), (int)__builtin_floor(pz));
                c.cnt[VRT_C_LOOKUP]++;
                if (id) break;
                stepsize = (double)(ch.entry >> 24);
            } else {  // void skip (init.py:114)
                const double mn = __builtin_fmin(__builtin_fmin(px, py), pz);
                const double t = mn + (double)st.chunk_radius;
                const double md = t - __builtin_floor(t * inv_cs) * cs;  // float % for a power-of-two divisor: exact
                stepsize = 1 + __builtin_fabs((double)st.chunk_radius - md);
            }
            step += stepsize;
            px += vx * stepsize;
            py += vy * stepsize;
            pz += vz * stepsize;
            c.cnt[VRT_C_ADV]++;
        }
        if (!id) break;  // while condition failed: the ray's life ran out
        // ---------------- phase B: a material was found (init.py:78-116) ----------------
        {
            const double* mat = c.mats + (id - 1) * 8;
            const double m_rough = mat[3], m_absorb = mat[4], m_ior = mat[5], m_energy = mat[6];
            // ---- lib.material (lib.py:448-460) ----
            double a = m_absorb / pow_cached(c, 1 + bounces, 1 + st.falloff);
            if (!(a < 1)) a = 1;
            const double b2 = 1 - a;
            cr = (int)__builtin_rint((double)cr * b2 + mat[0] * a);
            cg = (int)__builtin_rint((double)cg * b2 + mat[1] * a);
            cb = (int)__builtin_rint((double)cb * b2 + mat[2] * a);
            energy = energy * b2 + m_energy * a;
            life *= 1 - (m_rough * a);
            const double jx = py_rand(c, m_rough);
            const double jy = py_rand(c, m_rough);
            const double jz = py_rand(c, m_rough);
            vx += jx;
            vy += jy;
            vz += jz;
            c.cnt[VRT_C_HIT]++;
            // ---- init.py:82-86 ----
            bounces += m_absorb;
            life /= (double)(ch.entry >> 24) + m_absorb * st.lod_bounces;
            const double ref = __builtin_fmax(__builtin_fmax(__builtin_fabs(vx), __builtin_fabs(vy)), __builtin_fabs(vz));
            if (ref != 0.0 && ref != 1.0) {
                vx = vx / ref;
                vy = vy / ref;
                vz = vz / ref;
            }
            if (step >= life || energy >= st.max_light || bounces >= st.max_bounces + 1) {
                pending_hit = true;  // left through the reference's `break` (init.py:86)
                break;
            }
            // ---- reflection from the three neighbours (init.py:92-111) ----
            if (m_ior != 0.0) {
                const double direction = (m_ior - 0.5) * 2;
                bool solid[3];
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    const double v = ax == 0 ? vx : (ax == 1 ? vy : vz);
                    const double d = v < direction ? 1.0 : -1.0;
                    const double nx = ax == 0 ? px + d : px;
                    const double ny = ax == 1 ? py + d : py;
                    const double nz = ax == 2 ? pz + d : pz;
                    Chunk nc = ch;
                    if (!((nx >= ch.mnx && ny >= ch.mny && nz >= ch.mnz) && (nx <= mxx && ny <= mxy && nz <= mxz))) {
                        nc.mnx = __builtin_floor(nx * inv_cs) * cs;  // Camera.chunk_get (init.py:28-33)
                        nc.mny = __builtin_floor(ny * inv_cs) * cs;
                        nc.mnz = __builtin_floor(nz * inv_cs) * cs;
                        nc.entry = chunk_entry(P, nc.mnx, nc.mny, nc.mnz);
                        c.cnt[VRT_C_CHUNK_GET]++;
                    }
                    int nid = 0;
                    if (nc.entry) {
                        nid = lookup(P, nc, (int)__builtin_floor(nx), (int)__builtin_floor(ny), (int)__builtin_floor(nz));
                        c.cnt[VRT_C_NBR]++;
                    }
                    solid[ax] = nid != 0 && c.mats[(nid - 1) * 8 + 5] == m_ior;
                }
                if (!solid[0]) vx -= vx * m_ior * 2;
                if (!solid[1]) vy -= vy * m_ior * 2;
                if (!solid[2]) vz -= vz * m_ior * 2;
            }
            // ---- advance inside a present chunk (init.py:114-116) ----
            const double stepsize = (double)(ch.entry >> 24);
            step += stepsize;
            px += vx * stepsize;
            py += vy * stepsize;
            pz += vz * stepsize;
            c.cnt[VRT_C_ADV]++;
            if (c.exhausted) break;  // result is discarded and the ray re-traced with a longer draw table
        }
    }
    c.cnt[VRT_C_BROKE] = pending_hit ? 1 : 0;
    // ---- lib.material_background (lib.py:463-476) ----
    if (st.has_background) {
        double a = 1 / pow_cached(c, 1 + bounces, 1 + st.falloff);
        if (!(a < 1)) a = 1;
        const double up = vy > 0 ? vy : 0;
        const double b2 = 1 - a;
        cr = (int)__builtin_rint((double)cr * b2 + 127.0 * a);
        cg = (int)__builtin_rint((double)cg * b2 + (127 + up * 64) * a);
        cb = (int)__builtin_rint((double)cb * b2 + (127 + up * 128) * a);
        energy = energy * b2 + (1 + up) * a;
        double t;
        t = __builtin_rint((double)cr * energy); cr = t < 255 ? (int)t : 255;
        t = __builtin_rint((double)cg * energy); cg = t < 255 ? (int)t : 255;
        t = __builtin_rint((double)cb * energy); cb = t < 255 ? (int)t : 255;
    }
    o.cr = cr; o.cg = cg; o.cb = cb;
    o.energy = energy; o.step = step; o.life = life; o.bounces = bounces;
    o.px = px; o.py = py; o.pz = pz; o.vx = vx; o.vy = vy; o.vz = vz;
    o.ntrav = nseen;
}

template <bool TILE, bool RECORD, bool LIST>
__global__ void __launch_bounds__(VRT_BLOCK) march_kernel(MarchParams P) {
    __shared__ double s_mats[256 * 8];
    __shared__ unsigned long long s_stats[VRT_NSTATS];
    __shared__ unsigned long long s_pw_keys[VRT_PW_SLOTS];
    __shared__ unsigned long long s_pw_vals[VRT_PW_SLOTS];
    for (int i = threadIdx.x; i < P.n_materials * 8; i += VRT_BLOCK) s_mats[i] = P.materials[i];
    for (int i = threadIdx.x; i < VRT_PW_SLOTS; i += VRT_BLOCK) {
        s_pw_keys[i] = 0ull;
        s_pw_vals[i] = VRT_PW_NOT_READY;
    }
    if (threadIdx.x < VRT_NSTATS) s_stats[threadIdx.x] = 0;
    __syncthreads();

    const int64_t count = LIST ? (int64_t)*P.list_count : P.n;
    for (int64_t k = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x; k < count; k += (int64_t)gridDim.x * VRT_BLOCK) {
        const int64_t off = LIST ? (int64_t)P.list[k] : k;  // offset of the ray inside the batch
        const int64_t ray = P.ray0 + off;
        double dir_x, dir_y, detail;
        int x = 0, y = 0, s = 0;
        RayCtx c;
        c.P = &P;
        c.mats = s_mats;
        c.draws = P.draws;
        c.draw_idx = LIST ? k : off;
        c.ndraw = 0;
        c.exhausted = false;
        c.pw_keys = s_pw_keys;
        c.pw_vals = s_pw_vals;
#pragma unroll
        for (int j = 0; j < VRT_NCOUNTERS; j++) c.cnt[j] = 0;
        if (TILE) {
            const int64_t p = ray / P.g.smax;
            s = (int)(ray - p * P.g.smax);
            x = P.g.pixels[2 * p];
            y = P.g.pixels[2 * p + 1];
            int ns;
            pixel_setup(P.st, x, y, dir_x, dir_y, detail, ns);
            if (s >= ns) {
                if (P.ray_rgba) P.ray_rgba[ray] = 0;
                if (P.rays) P.rays[ray].s = -1;
                continue;
            }
            // init.py:139
            detail = detail / (1 + s * P.st.lod_samples) * (1 - P.st.lod_random * next_draw(c));
        } else {
            dir_x = P.dir_x[ray];
            dir_y = P.dir_y[ray];
            detail = P.detail[ray];
        }
        RayOut o;
        trace_ray<RECORD>(c, dir_x, dir_y, detail, (uint64_t)ray, o);
        c.cnt[VRT_C_DRAW] = c.ndraw;
        if (c.exhausted) {
            if (P.retrace_list) {
                uint32_t slot = atomicAdd(P.retrace_count, 1u);
                P.retrace_list[slot] = (uint32_t)off;
            } else {
                atomicAdd(&s_stats[VRT_S_RNG_EXHAUSTED], 1ull);
            }
            continue;
        }
        // init.py:141
        double e = o.energy + P.st.shutter;
        if (!(e < 1)) e = 1;
        const int alpha = (int)__builtin_rint(e * 255);
        if (P.ray_rgba) P.ray_rgba[ray] = (uint32_t)o.cr | ((uint32_t)o.cg << 8) | ((uint32_t)o.cb << 16) | ((uint32_t)alpha << 24);
        if (RECORD && P.rays) {
            vrt_ray& r = P.rays[ray];
            r.x = x; r.y = y; r.s = s;
            r.color[0] = o.cr; r.color[1] = o.cg; r.color[2] = o.cb;
            r.alpha = alpha;
            r.ntrav = o.ntrav;
#pragma unroll
            for (int j = 0; j < VRT_NCOUNTERS; j++) r.counters[j] = c.cnt[j];
            r.detail = detail; r.energy = o.energy; r.step = o.step; r.life = o.life; r.bounces = o.bounces;
            r.pos[0] = o.px; r.pos[1] = o.py; r.pos[2] = o.pz;
            r.vel[0] = o.vx; r.vel[1] = o.vy; r.vel[2] = o.vz;
        }
#pragma unroll
        for (int j = 0; j < VRT_NCOUNTERS; j++)
            if (c.cnt[j]) atomicAdd(&s_stats[j], (unsigned long long)c.cnt[j]);
        atomicAdd(&s_stats[VRT_S_RAYS], 1ull);
        if (LIST) atomicAdd(&s_stats[VRT_S_RNG_RETRACED], 1ull);
    }
    __syncthreads();
    if (threadIdx.x < VRT_NSTATS && s_stats[threadIdx.x])
        atomicAdd((unsigned long long*)&P.stats[threadIdx.x], s_stats[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// resolve: lib.average over the samples of each pixel (init.py:142-146)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(VRT_BLOCK) resolve_kernel(vrt_settings st, TileGeom g, const uint32_t* ray_rgba,
                                                            float* rgba_f32, uint8_t* image_u8) {
    int64_t p = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (p >= g.n_px) return;
    int x = g.pixels[2 * p], y = g.pixels[2 * p + 1];
    double dx, dy, det;
    int ns;
    pixel_setup(st, x, y, dx, dy, det, ns);
    uint32_t sr = 0, sg = 0, sb = 0, sa = 0;
    for (int s = 0; s < ns; s++) {
        uint32_t v = ray_rgba[p * g.smax + s];
        sr += v & 255u;
        sg += (v >> 8) & 255u;
        sb += (v >> 16) & 255u;
        sa += v >> 24;
    }
    const double n = (double)ns;
    const double mr = (double)sr / n, mg = (double)sg / n, mb = (double)sb / n, ma = (double)sa / n;
    if (rgba_f32) {
        float4 o = make_float4((float)mr, (float)mg, (float)mb, (float)ma);
        reinterpret_cast<float4*>(rgba_f32)[p] = o;
    }
    if (image_u8) {
        // Surface.set_at float -> u8 (init.py:146): truncation (exact for samples == 1)
        uchar4 o = make_uchar4((unsigned char)(int)mr, (unsigned char)(int)mg, (unsigned char)(int)mb, (unsigned char)(int)ma);
        reinterpret_cast<uchar4*>(image_u8)[(int64_t)y * st.width + x] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// synthetic volume (BASELINE config 5 generator)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
__global__ void __launch_bounds__(VRT_BLOCK) synth_kernel(int n, int cs, uint8_t* voxels) {
    // one thread per 4-voxel z-run of a micro-brick row: out index is the packed byte offset / 4
    const int64_t total4 = (int64_t)n * n * n / 4;
    const int ncs = n / cs;
    const int64_t block_bytes = (int64_t)cs * cs * cs;
    for (int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x; i < total4; i += (int64_t)gridDim.x * VRT_BLOCK) {
        int64_t byte = i * 4;
        int64_t slot = byte / block_bytes;
        int rem = (int)(byte - slot * block_bytes);
        int brick = rem >> 9, micro = (rem >> 6) & 7, vox = rem & 63;
        int nb = cs >> 3;
        int bz = brick % nb, by = (brick / nb) % nb, bx = brick / (nb * nb);
        int lx = bx * 8 + ((micro >> 2) & 1) * 4 + (vox >> 4);
        int ly = by * 8 + ((micro >> 1) & 1) * 4 + ((vox >> 2) & 3);
        int lz0 = bz * 8 + (micro & 1) * 4;
        int cz = (int)(slot % ncs), cy = (int)((slot / ncs) % ncs), cx = (int)(slot / ((int64_t)ncs * ncs));
        uint32_t X = (uint32_t)(cx * cs + lx), Y = (uint32_t)(cy * cs + ly);
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t Z = (uint32_t)(cz * cs + lz0 + k);
            uint32_t h = fmix32((X + (uint32_t)n * (Y + (uint32_t)n * Z)) ^ 0x5eed5eedu);
            uint32_t id = ((h & 0xffffu) >= 1311u) ? 0u : 1u + ((h >> 16) % 13u);
            packed |= id << (8 * k);
        }
        reinterpret_cast<uint32_t*>(voxels)[i] = packed;
    }
}
__global__ void synth_table_kernel(int64_t n_chunks, uint32_t* table) {
    int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (i < n_chunks) table[i] = (uint32_t)(i + 1) | (1u << 24);
}

// ---------------------------------------------------------------------------------------------
// host side of the C ABI
// ---------------------------------------------------------------------------------------------
static constexpr int D_FAST = 32;    // draws per ray in the first-pass table
static constexpr int D_SLOW = 113;   // draws in the retrace table (all outputs that need no state twist)
static constexpr int64_t BATCH_RAYS = 1 << 22;

extern "C" {

int vrt_abi_version(void) { return VRT_ABI_VERSION; }

const char* vrt_status_string(int s) {
    switch (s) {
        case VRT_OK: return "ok";
        case VRT_ERR_ARG: return "invalid argument or unsupported setting";
        case VRT_ERR_HIP: return "HIP runtime error";
        case VRT_ERR_WORKSPACE: return "workspace too small";
        case VRT_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown status";
    }
}

int vrt_last_hip_error(void) { return g_last_hip_error; }

int vrt_device_count(int* count) {
    if (!count) return VRT_ERR_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        *count = 0;
        return VRT_ERR_NO_DEVICE;
    }
    *count = n;
    return VRT_OK;
}

int64_t vrt_voxel_offset(int32_t cs, int32_t lx, int32_t ly, int32_t lz) { return voxel_offset(cs, lx, ly, lz); }

static int check_settings(const vrt_settings* st) {
    if (!st) return VRT_ERR_ARG;
    if (st->width <= 0 || st->height <= 0 || st->samples <= 0) return VRT_ERR_ARG;
    int cs = st->chunk_size;
    if (cs < 8 || (cs & (cs - 1)) || cs > 256) return VRT_ERR_ARG;
    if (!(st->proportions > 0)) return VRT_ERR_ARG;
    return VRT_OK;
}

int32_t vrt_max_samples(const vrt_settings* st) {
    if (!st) return 0;
    // detail = 1 - |dx*dy|*lod_edge with |dx*dy| <= 1
    double dmax = st->lod_edge < 0 ? 1 - st->lod_edge : 1;
    double r = __builtin_rint((double)st->samples * dmax);
    return r > 1 ? (int32_t)r : 1;
}

struct WsLayout {
    int64_t batch;        // rays per batch
    int64_t off_fast, off_slow, off_rgba, off_list, off_count, total;
};
static WsLayout ws_layout(const vrt_settings* st, int64_t n_px) {
    WsLayout w;
    int64_t rays = n_px * vrt_max_samples(st);
    w.batch = rays < BATCH_RAYS ? rays : BATCH_RAYS;
    if (w.batch < 1) w.batch = 1;
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o += (bytes + 255) & ~(int64_t)255; return r; };
    w.off_fast = take(w.batch * D_FAST * 8);
    w.off_slow = take(w.batch * D_SLOW * 8);
    w.off_rgba = take(rays * 4);
    w.off_list = take(w.batch * 4);
    w.off_count = take(256);
    w.total = o;
    return w;
}

int vrt_workspace_bytes(const vrt_settings* st, int64_t n_px, int64_t* bytes) {
    if (check_settings(st) != VRT_OK || n_px < 0 || !bytes) return VRT_ERR_ARG;
    *bytes = ws_layout(st, n_px).total;
    return VRT_OK;
}

static int fill_params(MarchParams& P, const vrt_scene* sc, const vrt_settings* st, const vrt_camera* cam,
                       const vrt_traversed* trav, uint64_t* d_stats) {
    if (!sc || !cam || check_settings(st) != VRT_OK) return VRT_ERR_ARG;
    if (sc->chunk_size != st->chunk_size || sc->n_materials < 0 || sc->n_materials > 255) return VRT_ERR_ARG;
    if (!sc->d_chunk_table || (sc->n_slots > 0 && !sc->d_voxels) || (sc->n_materials > 0 && !sc->d_materials)) return VRT_ERR_ARG;
    if (!d_stats) return VRT_ERR_ARG;
    P.st = *st;
    P.cam = *cam;
    int shift = 0;
    while ((1 << shift) < st->chunk_size) shift++;
    P.cs = st->chunk_size;
    P.cs_shift = shift;
    for (int a = 0; a < 3; a++) {
        if (sc->dims[a] <= 0 || (sc->origin[a] % st->chunk_size) != 0) return VRT_ERR_ARG;
        P.origin[a] = sc->origin[a];
        P.dims[a] = sc->dims[a];
    }
    P.n_materials = sc->n_materials;
    P.chunk_table = sc->d_chunk_table;
    P.voxels = sc->d_voxels;
    P.materials = sc->d_materials;
    P.t_keys = nullptr;
    for (int a = 0; a < 3; a++) { P.t_origin[a] = 0; P.t_dims[a] = 0; }
    if (trav && trav->d_keys) {
        for (int a = 0; a < 3; a++) {
            if (trav->dims[a] <= 0 || (trav->origin[a] % st->chunk_size) != 0) return VRT_ERR_ARG;
            P.t_origin[a] = trav->origin[a];
            P.t_dims[a] = trav->dims[a];
        }
        P.t_keys = trav->d_keys;
    }
    P.stats = d_stats;
    P.dir_x = P.dir_y = P.detail = nullptr;
    P.list = nullptr;
    P.list_count = nullptr;
    P.ray_rgba = nullptr;
    P.rays = nullptr;
    P.retrace_list = nullptr;
    P.retrace_count = nullptr;
    return VRT_OK;
}

static inline int grid_for(int64_t n) { return (int)((n + VRT_BLOCK - 1) / VRT_BLOCK); }
// march grid: bounded so that a workgroup lives long enough to amortise its LDS tables (grid-stride over rays)
static int march_grid(int64_t n) {
    static int cap = -1;
    if (cap < 0) {
        const char* e = getenv("VRT_MARCH_GRID");
        cap = e ? atoi(e) : 4096;
        if (cap < 1) cap = 1;
    }
    int g = grid_for(n);
    return g < cap ? g : cap;
}

int vrt_render_tile(const vrt_scene* scene, const vrt_settings* st, const vrt_camera* cam, const int32_t* d_pixels_xy,
                    int64_t n_px, void* d_workspace, int64_t workspace_bytes, float* d_rgba_f32, uint8_t* d_image_u8,
                    uint32_t* d_ray_rgba, vrt_ray* d_rays, uint64_t* d_stats, const vrt_traversed* trav, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MarchParams P;
    int rc = fill_params(P, scene, st, cam, trav, d_stats);
    if (rc != VRT_OK) return rc;
    if (n_px < 0 || (n_px > 0 && !d_pixels_xy) || !d_workspace) return VRT_ERR_ARG;
    const int smax = vrt_max_samples(st);
    const int64_t rays = n_px * smax;
    if (rays >= ((int64_t)1 << 51)) return VRT_ERR_ARG;
    WsLayout w = ws_layout(st, n_px);
    if (workspace_bytes < w.total) return VRT_ERR_WORKSPACE;
    char* ws = (char*)d_workspace;
    double* t_fast = (double*)(ws + w.off_fast);
    double* t_slow = (double*)(ws + w.off_slow);
    uint32_t* rgba = d_ray_rgba ? d_ray_rgba : (uint32_t*)(ws + w.off_rgba);
    uint32_t* list = (uint32_t*)(ws + w.off_list);
    uint32_t* count = (uint32_t*)(ws + w.off_count);
    HIP_TRY(hipMemsetAsync(d_stats, 0, sizeof(uint64_t) * VRT_NSTATS, stream));
    if (n_px == 0) return VRT_OK;
    TileGeom g;
    g.pixels = d_pixels_xy;
    g.n_px = n_px;
    g.smax = smax;
    P.g = g;
    P.ray_rgba = rgba;
    P.rays = d_rays;
    for (int64_t ray0 = 0; ray0 < rays; ray0 += w.batch) {
        const int64_t n = (rays - ray0) < w.batch ? (rays - ray0) : w.batch;
        HIP_TRY(hipMemsetAsync(count, 0, 4, stream));
        {
            ProfScope ps(stream, 0);
            hipLaunchKernelGGL(rng_tile_kernel<D_FAST>, dim3(grid_for(n)), dim3(VRT_BLOCK), 0, stream, *st, g, ray0, n, t_fast, n);
        }
        P.ray0 = ray0;
        P.n = n;
        P.list = nullptr;
        P.list_count = nullptr;
        P.draws = t_fast;
        P.dstride = n;
        P.n_draws = D_FAST;
        P.retrace_list = list;
        P.retrace_count = count;
        {
            ProfScope ps(stream, 1);
            if (d_rays) hipLaunchKernelGGL((march_kernel<true, true, false>), dim3(march_grid(n)), dim3(VRT_BLOCK), 0, stream, P);
            else hipLaunchKernelGGL((march_kernel<true, false, false>), dim3(march_grid(n)), dim3(VRT_BLOCK), 0, stream, P);
        }
        // rays that ran out of draws: longer table, device-side count (no host sync)
        ProfScope ps(stream, 2);
        const int rgrid = 1024;
        hipLaunchKernelGGL(rng_list_kernel<D_SLOW>, dim3(rgrid), dim3(VRT_BLOCK), 0, stream, *st, g, ray0, list, count, t_slow, n);
        P.list = list;
        P.list_count = count;
        P.draws = t_slow;
        P.dstride = n;
        P.n_draws = D_SLOW;
        P.retrace_list = nullptr;
        P.retrace_count = nullptr;
        if (d_rays) hipLaunchKernelGGL((march_kernel<true, true, true>), dim3(rgrid), dim3(VRT_BLOCK), 0, stream, P);
        else hipLaunchKernelGGL((march_kernel<true, false, true>), dim3(rgrid), dim3(VRT_BLOCK), 0, stream, P);
    }
    if (d_rgba_f32 || d_image_u8) {
        ProfScope ps(stream, 3);
        hipLaunchKernelGGL(resolve_kernel, dim3(grid_for(n_px)), dim3(VRT_BLOCK), 0, stream, *st, g, rgba, d_rgba_f32, d_image_u8);
    }
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int vrt_trace_rays(const vrt_scene* scene, const vrt_settings* st, const vrt_camera* cam, const double* d_dir_x,
                   const double* d_dir_y, const double* d_detail, const double* d_draws, int32_t n_draws, int64_t n_rays,
                   vrt_ray* d_rays, uint64_t* d_stats, const vrt_traversed* trav, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MarchParams P;
    int rc = fill_params(P, scene, st, cam, trav, d_stats);
    if (rc != VRT_OK) return rc;
    if (n_rays < 0 || n_draws < 0 || !d_rays) return VRT_ERR_ARG;
    if (n_rays > 0 && (!d_dir_x || !d_dir_y || !d_detail || (n_draws > 0 && !d_draws))) return VRT_ERR_ARG;
    HIP_TRY(hipMemsetAsync(d_stats, 0, sizeof(uint64_t) * VRT_NSTATS, stream));
    if (n_rays == 0) return VRT_OK;
    P.g.pixels = nullptr;
    P.g.n_px = 0;
    P.g.smax = 1;
    P.dir_x = d_dir_x;
    P.dir_y = d_dir_y;
    P.detail = d_detail;
    P.ray0 = 0;
    P.n = n_rays;
    P.draws = d_draws;
    P.dstride = n_rays;
    P.n_draws = n_draws;
    P.rays = d_rays;
    hipLaunchKernelGGL((march_kernel<false, true, false>), dim3(grid_for(n_rays)), dim3(VRT_BLOCK), 0, stream, P);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int vrt_rng_draws(const uint64_t* d_seeds, int64_t n_seeds, int32_t n_draws, double* d_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_seeds < 0 || n_draws < 1 || n_draws > D_SLOW || (n_seeds > 0 && (!d_seeds || !d_out))) return VRT_ERR_ARG;
    if (n_seeds == 0) return VRT_OK;
    // the kernels are instantiated for fixed draw counts; produce the next size up into the caller's rows
    if (n_draws != D_FAST && n_draws != D_SLOW && n_draws != 8) return VRT_ERR_ARG;
    dim3 grid(grid_for(n_seeds)), block(VRT_BLOCK);
    if (n_draws == 8) hipLaunchKernelGGL(rng_seeds_kernel<8>, grid, block, 0, stream, d_seeds, n_seeds, d_out);
    else if (n_draws == D_FAST) hipLaunchKernelGGL(rng_seeds_kernel<D_FAST>, grid, block, 0, stream, d_seeds, n_seeds, d_out);
    else hipLaunchKernelGGL(rng_seeds_kernel<D_SLOW>, grid, block, 0, stream, d_seeds, n_seeds, d_out);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int vrt_profile_begin(void) {
    for (auto& e : g_prof) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    g_prof.clear();
    g_prof_on = true;
    return VRT_OK;
}

int vrt_profile_end(double* ms, int64_t* launches) {
    g_prof_on = false;
    if (!ms || !launches) return VRT_ERR_ARG;
    for (int k = 0; k < VRT_NPROF; k++) { ms[k] = 0; launches[k] = 0; }
    int rc = VRT_OK;
    for (auto& e : g_prof) {
        float t = 0;
        if (hipEventSynchronize(e.b) != hipSuccess || hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) rc = VRT_ERR_HIP;
        ms[e.kind] += t;
        launches[e.kind]++;
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    g_prof.clear();
    return rc;
}

int vrt_synth_volume(int32_t n, int32_t cs, uint32_t* d_chunk_table, uint8_t* d_voxels, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (cs < 8 || (cs & (cs - 1)) || n <= 0 || (n % cs) || !d_chunk_table || !d_voxels || n > 2048) return VRT_ERR_ARG;
    int64_t nc = (int64_t)(n / cs) * (n / cs) * (n / cs);
    hipLaunchKernelGGL(synth_table_kernel, dim3(grid_for(nc)), dim3(VRT_BLOCK), 0, stream, nc, d_chunk_table);
    hipLaunchKernelGGL(synth_kernel, dim3(8192), dim3(VRT_BLOCK), 0, stream, (int)n, (int)cs, d_voxels);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

}  // extern "C"
