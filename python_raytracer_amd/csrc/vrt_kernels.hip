// vrt_kernels.hip -- MI355X (gfx950) voxel ray-march: the reference's Camera.tile -> Camera.trace ->
// Frame.get_voxel -> lib.material / lib.material_background path (reference init.py:37-150,
// data.py:136-145, lib.py:448-476) as hand-written HIP behind the C ABI of include/vrt.h.
//
// Per frame (vrt_render_tile):
//   rng_plan_kernel   MT19937 init_by_array + first 32 draws for every DISTINCT seed of the tile plan, state in
//                     registers only (CPython random.seed(int) / random.random(); init.py:137,139; lib.py:434)
//   raygen_kernel     per ray: tile()'s detail LOD and trace()'s lens / quaternion / forward vector
//                     (init.py:131-139, 41-45) -> a 32-byte ray record; uniform work, full lanes
//   march_kernel      persistent waves: each lane marches a ray through the chunk/voxel grid (init.py:66-116),
//                     shades with the default PBR material + sky (lib.py:448-476), and refills itself with the
//                     next ray of the wave's range when it finishes
//   resolve_kernel    per-pixel mean of the samples (lib.average, init.py:145) -> fp32 RGBA + RGBA8
// plus the plan kernels (static seed index, built once per pixel list) and a retrace pass for rays that need
// more than 32 random draws.
//
// Arithmetic is binary64 in the reference's evaluation order; build with -ffp-contract=off.
// gfx950 only: 64-wide waves are assumed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "../../include/vrt.h"
#include "vrt_math.h"

#pragma clang fp contract(off)

#define VRT_WAVE 64
#define VRT_BLOCK 256

// ---------------------------------------------------------------------------------------------
// error plumbing + optional per-kernel timing
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

// (one process-wide collector, switched on by vrt_profile_begin for bench.py; guarded so that concurrent callers
// on different threads cannot corrupt it)
struct ProfEvent { hipEvent_t a, b; int kind; };
static bool g_prof_on = false;
static std::vector<ProfEvent> g_prof;
static std::mutex g_prof_mu;
struct ProfScope {
    hipStream_t s;
    bool on;
    ProfEvent e;
    ProfScope(hipStream_t stream, int kind) : s(stream), on(g_prof_on) {
        if (!on) return;
        e.kind = kind;
        if (hipEventCreate(&e.a) != hipSuccess) { on = false; return; }
        if (hipEventCreate(&e.b) != hipSuccess) { (void)hipEventDestroy(e.a); on = false; return; }
        (void)hipEventRecord(e.a, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(e.b, s);
        std::lock_guard<std::mutex> lock(g_prof_mu);
        g_prof.push_back(e);
    }
};

// ---------------------------------------------------------------------------------------------
// shared helpers
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

struct TileGeom {
    const int32_t* pixels;  // [n_px][2]
    int64_t n_px;
    int32_t smax;           // sample slots per pixel
};

// ---------------------------------------------------------------------------------------------
// tile plan: distinct seeds of a pixel list (see vrt.h)
// ---------------------------------------------------------------------------------------------
#define VRT_PLAN_MAGIC 0x5652544e414c5032ull
struct PlanHeader {
    uint64_t magic, n_px, n_slots, n_distinct, settings_hash, n_words, pad[2];
};
static_assert(sizeof(PlanHeader) == 64, "plan header is 64 bytes");

static uint64_t plan_hash(const vrt_settings* st, int64_t n_px) {
    uint64_t h = 0xcbf29ce484222325ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 0x100000001b3ull; };
    union { double d; uint64_t u; } c;
    c.d = st->lod_edge;
    mix((uint64_t)st->width);
    mix((uint64_t)st->height);
    mix((uint64_t)st->samples);
    mix(c.u);
    mix((uint64_t)n_px);
    return h;
}

#define VRT_SCAN_WORDS 1024  // bitmap words per scan block (256 threads x 4)

__global__ void __launch_bounds__(VRT_BLOCK) plan_mark_kernel(vrt_settings st, TileGeom g, uint32_t* bitmap) {
    int64_t p = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (p >= g.n_px) return;
    int x = g.pixels[2 * p], y = g.pixels[2 * p + 1];
    double dx, dy, det;
    int ns;
    pixel_setup(st, x, y, dx, dy, det, ns);
    const uint32_t base = (uint32_t)(1 + x) * (uint32_t)(1 + y);
    for (int s = 0; s < ns; s++) {
        const uint32_t seed = base * (uint32_t)(1 + s);
        const uint32_t bit = 1u << (seed & 31);
        // bits only get set: a stale read can only cause a redundant atomic
        if (!(bitmap[seed >> 5] & bit)) atomicOr(&bitmap[seed >> 5], bit);
    }
}

__global__ void __launch_bounds__(VRT_BLOCK) plan_blocksum_kernel(const uint32_t* bitmap, int64_t n_words,
                                                                   uint32_t* block_sums) {
    __shared__ uint32_t s_sum[VRT_BLOCK / VRT_WAVE];
    int64_t w0 = (int64_t)blockIdx.x * VRT_SCAN_WORDS + threadIdx.x * 4;
    uint32_t c = 0;
    for (int k = 0; k < 4; k++)
        if (w0 + k < n_words) c += __popc(bitmap[w0 + k]);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// exclusive scan of the block sums in place (one workgroup) and the total into the header
__global__ void __launch_bounds__(VRT_BLOCK) plan_scan_sums_kernel(uint32_t* block_sums, int64_t n_blocks,
                                                                    PlanHeader* hdr) {
    __shared__ uint32_t s_part[VRT_BLOCK];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_blocks; base += VRT_BLOCK) {
        int64_t i = base + threadIdx.x;
        uint32_t v = i < n_blocks ? block_sums[i] : 0;
        s_part[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < VRT_BLOCK; o <<= 1) {  // Hillis-Steele inclusive scan
            uint32_t t = threadIdx.x >= (unsigned)o ? s_part[threadIdx.x - o] : 0;
            __syncthreads();
            s_part[threadIdx.x] += t;
            __syncthreads();
        }
        uint32_t incl = s_part[threadIdx.x];
        uint32_t carry = s_carry;
        if (i < n_blocks) block_sums[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == VRT_BLOCK - 1) s_carry = carry + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) hdr->n_distinct = s_carry;
}

// per word: exclusive prefix of set bits (into `prefix`) and the seed values into the compact list
__global__ void __launch_bounds__(VRT_BLOCK) plan_compact_kernel(const uint32_t* bitmap, int64_t n_words,
                                                                  const uint32_t* block_sums, uint32_t* prefix,
                                                                  uint32_t* seed_list) {
    __shared__ uint32_t s_part[VRT_BLOCK];
    int64_t w0 = (int64_t)blockIdx.x * VRT_SCAN_WORDS + threadIdx.x * 4;
    uint32_t words[4], c = 0;
    for (int k = 0; k < 4; k++) {
        words[k] = (w0 + k < n_words) ? bitmap[w0 + k] : 0;
        c += __popc(words[k]);
    }
    s_part[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < VRT_BLOCK; o <<= 1) {
        uint32_t t = threadIdx.x >= (unsigned)o ? s_part[threadIdx.x - o] : 0;
        __syncthreads();
        s_part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t pos = block_sums[blockIdx.x] + s_part[threadIdx.x] - c;
    for (int k = 0; k < 4; k++) {
        if (w0 + k >= n_words) break;
        prefix[w0 + k] = pos;
        uint32_t w = words[k];
        while (w) {
            int b = __ffs(w) - 1;
            w &= w - 1;
            seed_list[pos++] = (uint32_t)((w0 + k) * 32 + b);
        }
    }
}

__global__ void __launch_bounds__(VRT_BLOCK) plan_index_kernel(vrt_settings st, TileGeom g, const uint32_t* bitmap,
                                                                const uint32_t* prefix, uint32_t* ray_seedidx) {
    int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (i >= g.n_px * g.smax) return;
    int64_t p = i / g.smax;
    int s = (int)(i - p * g.smax);
    int x = g.pixels[2 * p], y = g.pixels[2 * p + 1];
    double dx, dy, det;
    int ns;
    pixel_setup(st, x, y, dx, dy, det, ns);
    uint32_t out = 0xffffffffu;
    if (s < ns) {
        const uint32_t seed = (uint32_t)(1 + x) * (uint32_t)(1 + y) * (uint32_t)(1 + s);
        const uint32_t w = bitmap[seed >> 5];
        out = prefix[seed >> 5] + __popc(w & ((1u << (seed & 31)) - 1u));
    }
    ray_seedidx[i] = out;
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

// Seeds with the 64-bit integer `seed` exactly like random.seed(seed) and writes the first D (<= 113) results of
// random.random() to out[0..D) (a row).  init_by_array's two dependent sweeps over the 624-word state are evaluated
// as register-only chains -- no state array anywhere:
//   pass A   first sweep (multiplier 1664525) end to end, to learn the word the second sweep starts from;
//   pass B   second sweep (multiplier 1566083941) with the first sweep re-derived in lockstep (two chains);
//   replay   output k < 227 needs seeded words s[k], s[k+1], s[k+397]: while pass B produces s[397 + k], a second
//            copy of the pass-B chains re-derives s[k + 1] from the start, so nothing has to be kept.
// PAIR: `out` is 16-byte aligned and draws are stored two at a time.
template <bool PAIR>
__device__ __forceinline__ void mt_seed_draws(uint64_t seed, int D, double* out) {
    const int K = 2 * D;  // <= 226
    const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
    const uint32_t addA = key0;                          // j = 0: key[0] + 0
    const uint32_t addB = key1 ? key1 + 1u : key0;       // j = 1 (len 2): key[1] + 1, else j stays 0
    const uint32_t* init = c_mt_init.v;
    // ---- pass A: t-th step writes index i = t + 1 ----
    uint32_t m = init[0];
    m = (init[1] ^ ((m ^ (m >> 30)) * 1664525u)) + addA;
    const uint32_t first1 = m;  // mt[1] after the first sweep's first step
#pragma unroll 2
    for (int i = 2; i < 624; i++) m = (init[i] ^ ((m ^ (m >> 30)) * 1664525u)) + (((i - 1) & 1) ? addB : addA);
    // step 624: i wrapped to 1, mt[0] = mt[623]
    const uint32_t mt1 = (first1 ^ ((m ^ (m >> 30)) * 1664525u)) + ((623 & 1) ? addB : addA);
#define VRT_MT_STEP(p, q, i)                                                            \
    p = (init[i] ^ ((p ^ (p >> 30)) * 1664525u)) + ((((i) - 1) & 1) ? addB : addA);      \
    q = (p ^ ((q ^ (q >> 30)) * 1566083941u)) - (uint32_t)(i);
    // ---- pass B up to the first word an output needs ----
    uint32_t p = first1, q = mt1;  // first / second sweep word i - 1
#pragma unroll 2
    for (int i = 2; i < 397; i++) { VRT_MT_STEP(p, q, i) }
    // ---- outputs: main chains at i = 397 + k, replay chains at j = k + 1 ----
    uint32_t rp = first1, rq = mt1;  // replay: word j - 1
    uint32_t s397, s398, s2;
    { VRT_MT_STEP(p, q, 397) }
    s397 = q;
    { VRT_MT_STEP(p, q, 398) }
    s398 = q;
    { VRT_MT_STEP(rp, rq, 2) }
    s2 = rq;
    uint32_t sk = s2;  // s[k]
    uint32_t prev = 0;
    double d1 = 0, dprev = 0;
    for (int k = 2; k < K; k++) {
        const int i = 397 + k, j = k + 1;
        VRT_MT_STEP(p, q, i)
        VRT_MT_STEP(rp, rq, j)
        const uint32_t o = mt_out(sk, rq, q);
        sk = rq;
        if (k & 1) {
            const int d = k >> 1;
            const double v = mt_res53(prev, o);
            if (d == 1) d1 = v;
            else if (!PAIR) out[d] = v;
            else if (d & 1) *reinterpret_cast<double2*>(out + d - 1) = make_double2(dprev, v);
            else if (d == D - 1) out[d] = v;  // odd D: the last draw has no partner
            else dprev = v;
        } else {
            prev = o;
        }
    }
#pragma unroll 2
    for (int i = 397 + K; i < 624; i++) { VRT_MT_STEP(p, q, i) }
#undef VRT_MT_STEP
    // wrap: mt[0] = mt[623]; the final step writes index 1; then mt[0] = 0x80000000
    const uint32_t s1 = (mt1 ^ ((q ^ (q >> 30)) * 1566083941u)) - 1u;
    const uint32_t o0 = mt_out(0x80000000u, s1, s397);
    const uint32_t o1 = mt_out(s1, s2, s398);
    if (PAIR) {
        *reinterpret_cast<double2*>(out) = make_double2(mt_res53(o0, o1), d1);
    } else {
        out[0] = mt_res53(o0, o1);
        out[1] = d1;
    }
}

// one lane per distinct seed of the plan: table[idx * D + k]
__global__ void __launch_bounds__(VRT_BLOCK) rng_plan_kernel(const uint32_t* seed_list, int64_t n, uint64_t nonce, int D,
                                                             double* table) {
    int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    mt_seed_draws<true>((uint64_t)seed_list[i] + nonce, D, table + i * D);
}

// retrace list: list[k] = ray offset inside the batch; table[k * STRIDE + d] (STRIDE even: rows 16-byte aligned)
#define VRT_SLOW_STRIDE 114
#define D_SLOW_DEV 113
__global__ void __launch_bounds__(VRT_BLOCK) rng_list_kernel(vrt_settings st, TileGeom g, int64_t ray0,
                                                             const uint32_t* list, const uint32_t* count, uint32_t cap,
                                                             double* table) {
    uint32_t n = *count < cap ? *count : cap;
    for (uint32_t k = blockIdx.x * VRT_BLOCK + threadIdx.x; k < n; k += gridDim.x * VRT_BLOCK) {
        int64_t ray = ray0 + list[k];
        int64_t p = ray / g.smax;
        int s = (int)(ray - p * g.smax);
        int x = g.pixels[2 * p], y = g.pixels[2 * p + 1];
        uint64_t seed = (uint64_t)((uint32_t)(1 + x) * (uint32_t)(1 + y) * (uint32_t)(1 + s)) + st.seed_nonce;
        mt_seed_draws<true>(seed, D_SLOW_DEV, table + (int64_t)k * VRT_SLOW_STRIDE);
    }
}

// Full-state MT19937 (CPython _randommodule.c: init_by_array, genrand_uint32 with the in-place twist): any number of
// draws.  The 624-word state is a per-lane private array (scratch memory), so this is slow and only used for the
// few rays that outrun the 113-draw rows, and for vrt_rng_draws beyond 113.
#define D_FULL_DEV 1024
__device__ __noinline__ void mt_full_draws(uint64_t seed, int D, double* out) {
    uint32_t mt[624];
    for (int i = 0; i < 624; i++) mt[i] = c_mt_init.v[i];
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    const int len = key[1] ? 2 : 1;
    int i = 1, j = 0;
    for (int k = 624; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++;
        j++;
        if (i >= 624) {
            mt[0] = mt[623];
            i = 1;
        }
        if (j >= len) j = 0;
    }
    for (int k = 623; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) {
            mt[0] = mt[623];
            i = 1;
        }
    }
    mt[0] = 0x80000000u;
    int pos = 624;
    uint32_t prev = 0;
    for (int n = 0; n < 2 * D; n++) {
        if (pos >= 624) {  // regenerate the whole state
            for (int k = 0; k < 624; k++) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            pos = 0;
        }
        const uint32_t o = mt_temper(mt[pos++]);
        if (n & 1) out[n >> 1] = mt_res53(prev, o);
        else prev = o;
    }
}

// third tier: rays that outran even their 113-draw row get D_FULL_DEV draws
__global__ void __launch_bounds__(64) rng_list_full_kernel(vrt_settings st, TileGeom g, int64_t ray0, const uint32_t* list,
                                                           const uint32_t* count, uint32_t cap, double* table) {
    uint32_t n = *count < cap ? *count : cap;
    for (uint32_t k = blockIdx.x * 64 + threadIdx.x; k < n; k += gridDim.x * 64) {
        int64_t ray = ray0 + list[k];
        int64_t p = ray / g.smax;
        int s = (int)(ray - p * g.smax);
        int x = g.pixels[2 * p], y = g.pixels[2 * p + 1];
        uint64_t seed = (uint64_t)((uint32_t)(1 + x) * (uint32_t)(1 + y) * (uint32_t)(1 + s)) + st.seed_nonce;
        mt_full_draws(seed, D_FULL_DEV, table + (int64_t)k * D_FULL_DEV);
    }
}

__global__ void __launch_bounds__(64) rng_seeds_full_kernel(const uint64_t* seeds, int64_t n, int D, double* out) {
    int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    mt_full_draws(seeds[i], D, out + i * D);
}

__global__ void __launch_bounds__(VRT_BLOCK) rng_seeds_kernel(const uint64_t* seeds, int64_t n, int D, double* out) {
    int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    mt_seed_draws<false>(seeds[i], D, out + i * D);
}

// ---------------------------------------------------------------------------------------------
// ray generation (init.py:131-139 and 41-56): one lane per ray, no divergence
// ---------------------------------------------------------------------------------------------
struct RayRec {  // SoA, stride = batch size
    double* vx;
    double* vy;
    double* vz;
    double* life;
};

__device__ __forceinline__ void gen_direction(const vrt_settings& st, const vrt_camera& cam, double dir_x, double dir_y,
                                              double jx, double jy, double& vx, double& vy, double& vz) {
    // init.py:41-45; lib.py:322-338 (vec3.quaternion), 353-358 (multiply), 372-376 (vec_forward)
    const double lens_x = (dir_x / st.proportions) * cam.lens + jx;
    const double lens_y = (dir_y * st.proportions) * cam.lens + jy;
    const double deg2rad = 3.141592653589793 / 180.0;  // math.radians
    const double rad_y = (-lens_x) * deg2rad, rad_z = lens_y * deg2rad;
    const double sin_x = 0.0, cos_x = 1.0;  // sin(0.0 / 2), cos(0.0 / 2)
    const double sin_y = vrt_sin(rad_y / 2), cos_y = vrt_cos(rad_y / 2);
    const double sin_z = vrt_sin(rad_z / 2), cos_z = vrt_cos(rad_z / 2);
    const double ox = sin_x * cos_y * cos_z - cos_x * sin_y * sin_z;
    const double oy = cos_x * sin_y * cos_z - sin_x * cos_y * sin_z;
    const double oz = cos_x * cos_y * sin_z + sin_x * sin_y * cos_z;
    const double ow = cos_x * cos_y * cos_z + sin_x * sin_y * sin_z;
    const double qx = cam.rot[0], qy = cam.rot[1], qz = cam.rot[2], qw = cam.rot[3];
    const double rx = qw * ox + qz * oy - qy * oz + qx * ow;
    const double ry = qz * ox + qw * oy + qx * oz + qy * ow;
    const double rz = qy * ox - qx * oy + qw * oz + qz * ow;
    const double rw = qx * ox - qy * oy - qz * oz + qw * ow;
    vx = 2 * (rz * rx + rw * ry);
    vy = 2 * (ry * rx - rw * rz);
    vz = 1 - 2 * (rz * rz + ry * ry);  // z ** 2 + y ** 2: correctly rounded squares
}

// lib.rand (lib.py:431-434) on a known draw
__device__ __forceinline__ double rand_amp(double draw, double amp) { return (-1 + draw * 2) * amp; }

// tile rays [ray0, ray0 + n): draws come from the plan's table rows
__global__ void __launch_bounds__(VRT_BLOCK) raygen_tile_kernel(vrt_settings st, vrt_camera cam, TileGeom g,
                                                                const uint32_t* ray_seedidx, const double* table,
                                                                int n_draws, int64_t ray0, int64_t n, RayRec rec) {
    int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t ray = ray0 + i;
    const uint32_t sidx = ray_seedidx[ray];
    if (sidx == 0xffffffffu) {
        rec.life[i] = -1.0;
        return;
    }
    const int64_t p = ray / g.smax;
    const int s = (int)(ray - p * g.smax);
    const int x = g.pixels[2 * p], y = g.pixels[2 * p + 1];
    double dir_x, dir_y, detail;
    int ns;
    pixel_setup(st, x, y, dir_x, dir_y, detail, ns);
    const double* row = table + (int64_t)sidx * n_draws;
    // init.py:139
    detail = detail / (1 + s * st.lod_samples) * (1 - st.lod_random * row[0]);
    double jx = 0, jy = 0;
    if (st.dof != 0.0) {
        jx = rand_amp(row[1], st.dof);
        jy = rand_amp(row[2], st.dof);
    }
    double vx, vy, vz;
    gen_direction(st, cam, dir_x, dir_y, jx, jy, vx, vy, vz);
    rec.vx[i] = vx;
    rec.vy[i] = vy;
    rec.vz[i] = vz;
    rec.life[i] = (st.dist_max - st.dist_min) * detail;  // init.py:56
}

// explicit rays (vrt_trace_rays): draws[i * n_draws + k]
__global__ void __launch_bounds__(VRT_BLOCK) raygen_explicit_kernel(vrt_settings st, vrt_camera cam, const double* dir_x,
                                                                    const double* dir_y, const double* detail,
                                                                    const double* draws, int n_draws, int64_t n,
                                                                    RayRec rec) {
    int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    double jx = 0, jy = 0;
    if (st.dof != 0.0 && n_draws >= 2) {
        jx = rand_amp(draws[i * n_draws + 0], st.dof);
        jy = rand_amp(draws[i * n_draws + 1], st.dof);
    }
    double vx, vy, vz;
    gen_direction(st, cam, dir_x[i], dir_y[i], jx, jy, vx, vy, vz);
    rec.vx[i] = vx;
    rec.vy[i] = vy;
    rec.vz[i] = vz;
    rec.life[i] = (st.dist_max - st.dist_min) * detail[i];
}

// ---------------------------------------------------------------------------------------------
// march
// ---------------------------------------------------------------------------------------------
#define VRT_PW_SLOTS 256
#define VRT_CHUNK 512
#define VRT_SPEC 4        // reference iterations fetched together per march pass ...
#ifndef VRT_SPEC_DEEP
#define VRT_SPEC_DEEP 8   // ... and for scenes far larger than the caches, where more loads in flight pay (config 5)
#endif
struct MarchParams {
    vrt_settings st;
    vrt_camera cam;
    // scene
    int64_t origin[3];
    int32_t origin32[3], t_origin32[3];
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
    // rays of this launch: batch [ray0, ray0 + n), records indexed by offset in the batch
    TileGeom g;                  // tile mode (g.pixels != NULL)
    const uint32_t* ray_seedidx; // tile mode: draw-table row of every ray slot of the tile
    const double* expl_detail;   // explicit mode
    RayRec rec;
    int64_t ray0, n;
    const uint32_t* list;        // LIST: ray offsets to re-trace and their device-side count
    const uint32_t* list_count;
    const double* draws;         // rows of draw_stride doubles, n_draws of them valid
    int32_t n_draws, draw_stride;
    int32_t first_draw;          // draws already consumed by ray generation
    int32_t threshold;           // lanes waiting for a slow body before the wave leaves the march loop
    // outputs
    uint32_t* ray_rgba;          // [rays of the tile] packed result (tile mode)
    vrt_ray* rays;               // debug records (may be NULL)
    uint64_t* stats;
    uint32_t* retrace_list;      // rays whose draws ran out are appended here (may be NULL)
    uint32_t* retrace_count;
    unsigned long long* pow_global;  // [2 * VRT_PW_SLOTS]: keys then values, shared by every workgroup (device_pow_memo)
    unsigned long long* queue_head;  // launch-wide ray counter (zeroed before every launch)
    uint32_t retrace_cap;            // capacity of retrace_list
    uint32_t list_cap;               // LIST: capacity of `list` (its count may have run past it)
    int32_t chunk;                   // rays per hand-out from queue_head; 0 = static range per wave
    int32_t end_period;              // ENDED lanes are served every end_period-th pass
};

// local cell of world cell (f // res) * res for res >= 3 (int // int, exact: |f| < 2^31, res <= 255); rare
__device__ __noinline__ int3 snap_generic3(int res, int imx, int imy, int imz, int lx, int ly, int lz) {
    int3 o;
    o.x = (int)__builtin_floor((double)(lx + imx) / (double)res) * res - imx;
    o.y = (int)__builtin_floor((double)(ly + imy) / (double)res) * res - imy;
    o.z = (int)__builtin_floor((double)(lz + imz) / (double)res) * res - imz;
    return o;
}

// Frame.get_voxel(floor(pos)) (data.py:136-145) on the packed chunk block: cell (fp // res) * res, which only
// exists inside the chunk's own half-open box.  (lx, ly, lz) = floor(pos) - chunk_min; tab[3][256]: per-axis parts
// of vrt_voxel_offset (disjoint bits); base: the chunk's voxel block; entry != 0.
__device__ __forceinline__ int lookup(const MarchParams& P, const uint32_t* tab, const uint8_t* base, uint32_t entry,
                                      int imx, int imy, int imz, int lx, int ly, int lz) {
    if (entry >= (2u << 24)) {  // resolution > 1
        if (entry < (3u << 24)) {  // 2: chunk_min is even, (f & ~1) - chunk_min == (f - chunk_min) & ~1
            lx &= ~1;
            ly &= ~1;
            lz &= ~1;
        } else {
            const int3 o = snap_generic3((int)(entry >> 24), imx, imy, imz, lx, ly, lz);
            lx = o.x;
            ly = o.y;
            lz = o.z;
        }
    }
    if ((unsigned)(lx | ly | lz) >= (unsigned)P.cs) return 0;  // cs is a power of two: some coordinate out of [0, cs)
    return base[tab[lx] | tab[256 + ly] | tab[512 + lz]];
}

// address form of lookup(): the voxel byte to read, or nullptr when the cell lies outside the chunk's block
__device__ __forceinline__ const uint8_t* voxel_addr(const MarchParams& P, const uint32_t* tab, const uint8_t* base,
                                                     uint32_t entry, int imx, int imy, int imz, int lx, int ly, int lz) {
    if (entry >= (2u << 24)) {
        if (entry < (3u << 24)) {
            lx &= ~1;
            ly &= ~1;
            lz &= ~1;
        } else {
            const int3 o = snap_generic3((int)(entry >> 24), imx, imy, imz, lx, ly, lz);
            lx = o.x;
            ly = o.y;
            lz = o.z;
        }
    }
    if ((unsigned)(lx | ly | lz) >= (unsigned)P.cs) return nullptr;
    return base + (tab[lx] | tab[256 + ly] | tab[512 + lz]);
}

// chunk table entry of chunk cell (cx, cy, cz), 0 outside the scene box
__device__ __forceinline__ uint32_t chunk_entry_i(const MarchParams& P, int cx, int cy, int cz) {
    if ((unsigned)cx >= (unsigned)P.dims[0] || (unsigned)cy >= (unsigned)P.dims[1] || (unsigned)cz >= (unsigned)P.dims[2])
        return 0;
    return P.chunk_table[(cx * P.dims[1] + cy) * P.dims[2] + cz];
}
__device__ __forceinline__ const uint8_t* chunk_base(const MarchParams& P, uint32_t entry) {
    return P.voxels + ((int64_t)((entry & 0xffffffu) - 1u) << (3 * P.cs_shift));
}

// record a visited chunk (world chunk_min as integers) for the `traversed` list (init.py:72-73)
__device__ __forceinline__ void trav_visit(const MarchParams& P, int imx, int imy, int imz, uint64_t key) {
    if (!P.t_keys) return;
    const int cx = (imx - P.t_origin32[0]) >> P.cs_shift;
    const int cy = (imy - P.t_origin32[1]) >> P.cs_shift;
    const int cz = (imz - P.t_origin32[2]) >> P.cs_shift;
    if ((unsigned)cx >= (unsigned)P.t_dims[0] || (unsigned)cy >= (unsigned)P.t_dims[1] ||
        (unsigned)cz >= (unsigned)P.t_dims[2]) {
        atomicAdd((unsigned long long*)&P.stats[VRT_S_TRAV_OUTSIDE], 1ull);
        return;
    }
    uint64_t* slot = &P.t_keys[((int64_t)cx * P.t_dims[1] + cy) * P.t_dims[2] + cz];
    // keys only decrease, so a stale (larger) value read here can only cause a redundant atomic
    if (key < *slot) atomicMin((unsigned long long*)slot, (unsigned long long)key);
}

// (1 + bounces) ** (1 + falloff) (lib.py:450, 465).  The exponent is fixed for a frame and the bases are sums of
// material absorptions -- a handful of distinct values -- so vrt_pow is memoised: an LDS table per workgroup,
// preloaded from / published to a global table that lives for the frame.  Both tables are insert-only (key:
// EMPTY(0) -> bits once, by CAS; value: NOT_READY(0) -> result once), hence a reader that sees its key and a
// ready value has exactly the value vrt_pow returns for that key: results are identical to calling vrt_pow.
struct PowCache {
    unsigned long long* keys;  // LDS (address space inferred after inlining)
    unsigned long long* vals;
    unsigned long long* gkeys; // global
    unsigned long long* gvals;
};
__device__ __forceinline__ unsigned pow_hash(unsigned long long bits) {
    unsigned h = (unsigned)(bits >> 32) * 0x9e3779b1u + (unsigned)bits * 0x85ebca6bu;
    return h >> 24;
}
#define VRT_PW_PROBES 4
__device__ __noinline__ double pow_miss(unsigned long long* keys, unsigned long long* vals, unsigned long long* gkeys,
                                        unsigned long long* gvals, double x, double y) {
    const double v = vrt_pow(x, y);
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
    const unsigned h = pow_hash(bits);
    for (int i = 0; i < VRT_PW_PROBES; i++) {
        const int s = (int)((h + i) & (VRT_PW_SLOTS - 1));
        const unsigned long long old = atomicCAS(&keys[s], 0ull, bits);
        if (old == 0ull || old == bits) {
            __hip_atomic_store(&vals[s], vb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
    }
    if (gkeys) {  // publish for later workgroups / launches of this frame
        for (int i = 0; i < VRT_PW_PROBES; i++) {
            const int s = (int)((h + i) & (VRT_PW_SLOTS - 1));
            const unsigned long long old = atomicCAS(&gkeys[s], 0ull, bits);
            if (old == 0ull || old == bits) {
                __hip_atomic_store(&gvals[s], vb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    return v;
}
// needs_pow: lanes whose base is not 1.0; returns 0 bits when the value is not in the LDS table
__device__ __forceinline__ unsigned long long pow_probe(const PowCache& pc, unsigned long long bits) {
    const unsigned h = pow_hash(bits);
    unsigned long long found = 0ull;
#pragma unroll
    for (int i = 0; i < VRT_PW_PROBES; i++) {
        const int sidx = (int)((h + i) & (VRT_PW_SLOTS - 1));
        const unsigned long long k = __hip_atomic_load(&pc.keys[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == bits) found = __hip_atomic_load(&pc.vals[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == bits || k == 0ull) break;
    }
    return found;
}
__device__ __forceinline__ double pow_cached(const PowCache& pc, double x, double y) {
    if (x == 1.0) return 1.0;
    const unsigned long long found = pow_probe(pc, (unsigned long long)__double_as_longlong(x));
    if (found != 0ull) return __longlong_as_double((long long)found);
    return pow_miss(pc.keys, pc.vals, pc.gkeys, pc.gvals, x, y);
}

// state of the ray a lane is marching (the `ray` store of init.py:50-59 plus the chunk cursor of init.py:46-47)
struct Ray {
    double px, py, pz, vx, vy, vz;
    double step, life, bounces, energy;
    int imx, imy, imz;      // chunk_min (init.py:68) as integers; chunk_max = chunk_min + cs
    uint32_t entry;         // chunk table entry of the current chunk (0 = None)
    const uint8_t* base;    // its voxel block
    double stepd;           // its resolution = the step inside it (init.py:114)
    int cr, cg, cb;
    int ndraw, resnaps;
    int id;                 // material found by the march, waiting to be shaded
    int64_t off;            // offset of the ray in the batch
    const double* row;      // its draw-table row
};

enum { LANE_IDLE = 0, LANE_MARCH = 1, LANE_HIT = 2, LANE_ENDED = 3 };

// Persistent waves.  Every lane is a small state machine: MARCH (phase A of the reference loop: snap chunk, look up
// the voxel, advance -- init.py:66-77, 114-116), HIT (phase B: shade, test termination, reflect, advance --
// init.py:78-116), ENDED (background + outputs -- init.py:119-120, 141-142), IDLE (take the next ray of the wave's
// range).  The cheap MARCH step runs every iteration; the expensive HIT / ENDED / refill bodies run only once
// `threshold` lanes are waiting for them (or nothing is marching), so they execute with many lanes active.
// Per-ray semantics are exactly the reference's single loop.
template <bool RECORD, bool LIST, int SPEC = VRT_SPEC>
__global__ void __launch_bounds__(VRT_BLOCK, 4) march_kernel(MarchParams P) {
    __shared__ double s_mats[256 * 8];
    __shared__ unsigned long long s_stats[VRT_NSTATS];
    __shared__ unsigned long long s_pw_keys[VRT_PW_SLOTS];
    __shared__ unsigned long long s_pw_vals[VRT_PW_SLOTS];
    __shared__ uint32_t s_tab[3 * 256];
    if (LIST && *P.list_count == 0) return;  // the usual case: no ray ran out of draws
    for (int i = threadIdx.x; i < P.n_materials * 8; i += VRT_BLOCK) s_mats[i] = P.materials[i];
    for (int i = threadIdx.x; i < VRT_PW_SLOTS; i += VRT_BLOCK) {
        unsigned long long k = 0, v = 0;
        if (P.pow_global) {
            k = __hip_atomic_load(&P.pow_global[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = __hip_atomic_load(&P.pow_global[VRT_PW_SLOTS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_pw_keys[i] = k;
        s_pw_vals[i] = k ? v : 0ull;
    }
    for (int i = threadIdx.x; i < P.cs; i += VRT_BLOCK) {
        s_tab[i] = (uint32_t)voxel_offset(P.cs, i, 0, 0);  // fixed stride 256: constant LDS offsets in the lookups
        s_tab[256 + i] = (uint32_t)voxel_offset(P.cs, 0, i, 0);
        s_tab[512 + i] = (uint32_t)voxel_offset(P.cs, 0, 0, i);
    }
    if (threadIdx.x < VRT_NSTATS) s_stats[threadIdx.x] = 0;
    __syncthreads();

    const vrt_settings& st = P.st;
    PowCache pc;
    pc.keys = s_pw_keys;
    pc.vals = s_pw_vals;
    pc.gkeys = P.pow_global;
    pc.gvals = P.pow_global ? P.pow_global + VRT_PW_SLOTS : nullptr;
    const double cs = (double)P.cs;
    const double inv_cs = 1.0 / cs;  // cs is a power of two: x * inv_cs == x / cs exactly
    const bool tile = P.g.pixels != nullptr;
    const int threshold = P.threshold;

    // rays are handed out in chunks of VRT_CHUNK consecutive rays from a launch-wide counter: coherent lanes,
    // balanced waves.  `next`/`range_end` are wave-uniform.
    const int64_t count = LIST ? (int64_t)(*P.list_count < P.list_cap ? *P.list_count : P.list_cap) : P.n;
    const int64_t chunk = P.chunk;
    int64_t next = 0, range_end = 0;
    bool more = true;  // the launch-wide counter may still have rays
    if (chunk == 0) {  // static contiguous range per wave
        const int64_t n_waves = (int64_t)gridDim.x * (VRT_BLOCK / VRT_WAVE);
        const int64_t wave = (int64_t)blockIdx.x * (VRT_BLOCK / VRT_WAVE) + (threadIdx.x >> 6);
        int64_t per = (count + n_waves - 1) / n_waves;
        per = (per + 7) & ~(int64_t)7;
        next = wave * per;
        range_end = next + per < count ? next + per : count;
        more = false;
    }

    Ray r;
    r.off = 0;
    r.row = nullptr;
    r.px = r.py = r.pz = r.vx = r.vy = r.vz = 0;
    r.step = r.life = r.bounces = r.energy = 0;
    r.imx = r.imy = r.imz = 0;
    r.entry = 0;
    r.base = nullptr;
    r.stepd = 1;
    r.cr = r.cg = r.cb = 0;
    r.ndraw = r.resnaps = r.id = 0;
    int state = LANE_IDLE;
    bool exhausted = false, broke = false;
    int32_t cnt[VRT_NCOUNTERS];   // events of the current ray
    int32_t tot[VRT_NCOUNTERS];   // summed over the rays this lane completed
    unsigned n_done = 0, n_retraced = 0, n_exhausted = 0;
#pragma unroll
    for (int j = 0; j < VRT_NCOUNTERS; j++) cnt[j] = tot[j] = 0;
    int64_t seen[RECORD ? 48 : 1];  // RECORD: the ray's own traversed list, to report its length (init.py:72-73)
    int nseen = 0;
    int pass = 0;
#ifdef VRT_DIAG
    unsigned long long dg_inner = 0, dg_march_lanes = 0, dg_outer = 0, dg_hit_lanes = 0, dg_end_lanes = 0, dg_refill_lanes = 0;
    unsigned long long dg_cyc[4] = {0, 0, 0, 0};
    unsigned long long dg_tend = 0;
#endif

    for (;;) {
#ifdef VRT_DIAG
        dg_outer++;
        dg_refill_lanes += __popcll(__ballot(state == LANE_IDLE));
        unsigned long long dg_t0 = clock64();
#endif
        // ------------------------------------------------------------------ refill idle lanes
        unsigned long long idle_mask = __ballot(state == LANE_IDLE);
        while (idle_mask != 0ull && (next < range_end || more)) {
            if (next >= range_end) {  // take the next chunk (one atomic per wave per chunk)
                unsigned long long base = 0;
                if ((threadIdx.x & 63) == 0) base = atomicAdd(P.queue_head, (unsigned long long)chunk);
                base = (unsigned long long)__shfl((long long)base, 0);
                if ((int64_t)base >= count) {
                    more = false;
                    break;
                }
                next = (int64_t)base;
                range_end = next + chunk < count ? next + chunk : count;
            }
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle_mask >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((unsigned)idle_mask, 0u));
            const int64_t k = next + rank;
            next += __popcll(idle_mask);
            if (state == LANE_IDLE && k < range_end) {
                const int64_t off = LIST ? (int64_t)P.list[k] : k;
                // the whole record is fetched at once (one memory round trip), then inspected
                const double life = P.rec.life[off];
                const double rvx = P.rec.vx[off], rvy = P.rec.vy[off], rvz = P.rec.vz[off];
                const int64_t rowi = LIST ? k : (tile ? (int64_t)P.ray_seedidx[P.ray0 + off] : off);
                if (life < 0.0) {  // unused sample slot of the tile
                    if (P.ray_rgba) P.ray_rgba[P.ray0 + off] = 0;
                    if (RECORD && P.rays) P.rays[P.ray0 + off].s = -1;
                } else {
                    r.off = off;
                    r.vx = rvx;
                    r.vy = rvy;
                    r.vz = rvz;
                    r.life = life;
                    // init.py:50-59
                    r.px = P.cam.pos[0] + r.vx * st.dist_min;
                    r.py = P.cam.pos[1] + r.vy * st.dist_min;
                    r.pz = P.cam.pos[2] + r.vz * st.dist_min;
                    r.step = 0;
                    r.bounces = 0;
                    r.energy = 0;
                    r.cr = r.cg = r.cb = 0;
                    // chunk_min = chunk_max = vec3(0, 0, 0), chunk = None (init.py:46-47): the sentinel makes the
                    // fast in-chunk test fail until the first snap (resnaps == 0 selects the reference's test)
                    r.imx = r.imy = r.imz = 0x20000000;
                    r.entry = 0;
                    r.resnaps = 0;
                    r.ndraw = P.first_draw;
                    r.row = P.draws + rowi * P.draw_stride;
                    exhausted = false;
                    broke = false;
                    nseen = 0;
#pragma unroll
                    for (int j = 0; j < VRT_NCOUNTERS; j++) cnt[j] = 0;
                    state = LANE_MARCH;
                }
            }
            idle_mask = __ballot(state == LANE_IDLE);
        }
        if (__ballot(state != LANE_IDLE) == 0ull) break;  // range exhausted and every lane finished
#ifdef VRT_DIAG
        unsigned long long dg_t1 = clock64();
        dg_cyc[0] += dg_t1 - dg_t0;
#endif

        // ------------------------------------------------------------------ MARCH steps (phase A)
        // lanes that can make progress in this pass: marching ones plus those waiting for a slow body (HIT / ENDED /
        // refillable IDLE).  The march loop runs until `threshold` of them wait, i.e. until at most `limit` march.
        // The loop collects `threshold` NEW events per pass: it runs until at most `limit` of the lanes that were
        // marching at its start still march (lanes carried over in ENDED state do not shorten it).
        pass++;
        const int m0 = __popcll(__ballot(state == LANE_MARCH));
        const int limit = m0 > threshold ? m0 - threshold : 0;
        for (;;) {
            const unsigned long long marching = __ballot(state == LANE_MARCH);
            if ((int)__popcll(marching) <= limit) break;
#ifdef VRT_DIAG
            dg_inner++;
            dg_march_lanes += __popcll(marching);
#endif
            if (state == LANE_MARCH) {
                if (!(r.step < r.life)) {  // init.py:66: the ray's life ran out
                    state = LANE_ENDED;
                } else {
                    const int fx = (int)__builtin_floor(r.px), fy = (int)__builtin_floor(r.py), fz = (int)__builtin_floor(r.pz);
                    int lx = fx - r.imx, ly = fy - r.imy, lz = fz - r.imz;
                    // strictly inside the half-open chunk box => inside the reference's inclusive box (init.py:67)
                    if ((unsigned)(lx | ly | lz) >= (unsigned)P.cs) {
                        bool outside;
                        if (r.resnaps == 0) {  // chunk_min == chunk_max == (0, 0, 0) (init.py:46)
                            outside = !(r.px >= 0.0 && r.py >= 0.0 && r.pz >= 0.0) || !(r.px <= 0.0 && r.py <= 0.0 && r.pz <= 0.0);
                        } else {
                            const double mnx = (double)r.imx, mny = (double)r.imy, mnz = (double)r.imz;
                            outside = !(r.px >= mnx && r.py >= mny && r.pz >= mnz) ||
                                      !(r.px <= mnx + cs && r.py <= mny + cs && r.pz <= mnz + cs);
                        }
                        if (outside) {
                            // snapped(): (v // cs) * cs (init.py:68-73); floor(p / cs) * cs == (floor(p) >> shift) << shift
                            r.imx = (fx >> P.cs_shift) << P.cs_shift;
                            r.imy = (fy >> P.cs_shift) << P.cs_shift;
                            r.imz = (fz >> P.cs_shift) << P.cs_shift;
                            lx = fx - r.imx;
                            ly = fy - r.imy;
                            lz = fz - r.imz;
                            r.entry = chunk_entry_i(P, (r.imx - P.origin32[0]) >> P.cs_shift, (r.imy - P.origin32[1]) >> P.cs_shift,
                                                    (r.imz - P.origin32[2]) >> P.cs_shift);
                            r.base = chunk_base(P, r.entry);
                            r.stepd = (double)((r.entry >> 24) ? (r.entry >> 24) : 1u);  // a zero resolution must not stall the march
                            trav_visit(P, r.imx, r.imy, r.imz,
                                       ((uint64_t)(P.ray0 + r.off) << 12) | (uint64_t)(r.resnaps < 4095 ? r.resnaps : 4095));
                            r.resnaps++;
                            cnt[VRT_C_RESNAP]++;
                            if (RECORD) {
                                int64_t cid = (((int64_t)r.imx >> P.cs_shift) * 2097152 + ((int64_t)r.imy >> P.cs_shift)) * 2097152 +
                                              ((int64_t)r.imz >> P.cs_shift);
                                bool dup = false;
                                for (int k = 0; k < nseen && k < 48; k++) dup |= (seen[k] == cid);
                                if (!dup) {
                                    if (nseen < 48) seen[nseen] = cid;
                                    nseen++;
                                }
                            }
                        }
                    }
                    if (r.entry) {  // init.py:75-77
                        // SPEC reference iterations per pass: the voxel of this position and, speculatively, of the
                        // next ones (pos + vel * step added repeatedly, the values the reference computes at init.py:116)
                        // are fetched together, so empty voxels cost one memory round trip per SPEC steps.  A
                        // speculative step is only taken when the reference would take it unchanged: loop condition true
                        // (init.py:66), still strictly inside the same chunk (no re-snap at init.py:67), every earlier
                        // voxel empty.  vel * step is the same rounded product in every one of these iterations.
                        const double sd = r.stepd;
                        const double dvx = r.vx * sd, dvy = r.vy * sd, dvz = r.vz * sd;
                        const uint8_t* addr[SPEC];
                        addr[0] = voxel_addr(P, s_tab, r.base, r.entry, r.imx, r.imy, r.imz, lx, ly, lz);
                        int n_valid = 1;  // positions whose voxel the reference would look up, if all before are empty
                        {
                            double qx = r.px, qy = r.py, qz = r.pz, qs = r.step;
                            bool ok = true;
#pragma unroll
                            for (int k = 1; k < SPEC; k++) {
                                qx += dvx;
                                qy += dvy;
                                qz += dvz;
                                qs += sd;
                                const int kx = (int)__builtin_floor(qx) - r.imx, ky = (int)__builtin_floor(qy) - r.imy,
                                          kz = (int)__builtin_floor(qz) - r.imz;
                                ok = ok && (qs < r.life) && ((unsigned)(kx | ky | kz) < (unsigned)P.cs);
                                addr[k] = ok ? voxel_addr(P, s_tab, r.base, r.entry, r.imx, r.imy, r.imz, kx, ky, kz) : nullptr;
                                n_valid += ok ? 1 : 0;
                            }
                        }
                        int ids[SPEC];
#pragma unroll
                        for (int k = 0; k < SPEC; k++) {
                            ids[k] = 0;
                            if (addr[k]) ids[k] = *addr[k];
                        }
                        // first occupied voxel among the valid positions
                        int h = n_valid, id = 0;
#pragma unroll
                        for (int k = SPEC - 1; k >= 0; k--) {
                            if (k < n_valid && ids[k] != 0) {
                                h = k;
                                id = ids[k];
                            }
                        }
                        // h advances were made before the hit (or n_valid advances and no hit)
                        cnt[VRT_C_LOOKUP] += id ? h + 1 : n_valid;
                        cnt[VRT_C_ADV] += h;
#pragma unroll
                        for (int k = 0; k < SPEC; k++) {
                            if (k < h) {
                                r.step += sd;
                                r.px += dvx;
                                r.py += dvy;
                                r.pz += dvz;
                            }
                        }
                        if (id) {
                            r.id = id;
                            state = LANE_HIT;
                        }
                    } else {  // void skip (init.py:114)
                        const double mn = __builtin_fmin(__builtin_fmin(r.px, r.py), r.pz);
                        const double t = mn + (double)st.chunk_radius;
                        const double md = t - __builtin_floor(t * inv_cs) * cs;  // float % for a power-of-two divisor: exact
                        const double stepsize = 1 + __builtin_fabs((double)st.chunk_radius - md);
                        r.step += stepsize;
                        r.px += r.vx * stepsize;
                        r.py += r.vy * stepsize;
                        r.pz += r.vz * stepsize;
                        cnt[VRT_C_ADV]++;
                    }
                }
            }
        }

        // ------------------------------------------------------------------ HIT (phase B: init.py:78-116)
#ifdef VRT_DIAG
        dg_hit_lanes += __popcll(__ballot(state == LANE_HIT));
        dg_end_lanes += __popcll(__ballot(state == LANE_ENDED));
        unsigned long long dg_t2 = clock64();
        dg_cyc[1] += dg_t2 - dg_t1;
#endif
        if (state == LANE_HIT) {
            const double* mat = s_mats + (r.id - 1) * 8;
            const double m_rough = mat[3], m_absorb = mat[4], m_ior = mat[5], m_energy = mat[6];
            // the three draws of a rough material are requested first so that they arrive under the shading math
            const bool have_draws = r.ndraw + 3 <= P.n_draws;
            double d0 = 0.5, d1 = 0.5, d2 = 0.5;
            if (m_rough != 0.0 && have_draws) {
                d0 = r.row[r.ndraw];
                d1 = r.row[r.ndraw + 1];
                d2 = r.row[r.ndraw + 2];
            }
            // ---- lib.material (lib.py:448-460) ----
            double a = m_absorb / pow_cached(pc, 1 + r.bounces, 1 + st.falloff);
            if (!(a < 1)) a = 1;
            const double b2 = 1 - a;
            r.cr = (int)__builtin_rint((double)r.cr * b2 + mat[0] * a);
            r.cg = (int)__builtin_rint((double)r.cg * b2 + mat[1] * a);
            r.cb = (int)__builtin_rint((double)r.cb * b2 + mat[2] * a);
            r.energy = r.energy * b2 + m_energy * a;
            r.life *= 1 - (m_rough * a);
            if (m_rough != 0.0) {  // lib.rand draws nothing for amplitude 0 (lib.py:431-434)
                if (have_draws) {
                    r.vx += rand_amp(d0, m_rough);
                    r.vy += rand_amp(d1, m_rough);
                    r.vz += rand_amp(d2, m_rough);
                } else {
                    exhausted = true;
                }
                r.ndraw += 3;
            }
            cnt[VRT_C_HIT]++;
            // ---- init.py:82-86 ----
            r.bounces += m_absorb;
            r.life /= r.stepd + m_absorb * st.lod_bounces;
            const double ref = __builtin_fmax(__builtin_fmax(__builtin_fabs(r.vx), __builtin_fabs(r.vy)), __builtin_fabs(r.vz));
            if (ref != 0.0 && ref != 1.0) {
                r.vx = r.vx / ref;
                r.vy = r.vy / ref;
                r.vz = r.vz / ref;
            }
            if (r.step >= r.life || r.energy >= st.max_light || r.bounces >= st.max_bounces + 1) {
                state = LANE_ENDED;  // left through the reference's `break` (init.py:86)
                broke = true;
            } else if (exhausted) {
                state = LANE_ENDED;  // result is discarded and the ray re-traced with a longer draw table
            } else {
                // ---- reflection from the three neighbours (init.py:92-111) ----
                if (m_ior != 0.0) {
                    const double direction = (m_ior - 0.5) * 2;
                    const int lx = (int)__builtin_floor(r.px) - r.imx, ly = (int)__builtin_floor(r.py) - r.imy,
                              lz = (int)__builtin_floor(r.pz) - r.imz;
                    // Three independent neighbour lookups, done in phases so that their memory accesses overlap:
                    // (1) which chunk each neighbour point belongs to, (2) its voxel address, (3) the three reads.
                    uint32_t nentry[3];
                    int nl[3][3], nm[3][3];
                    bool foreign[3];
#pragma unroll
                    for (int ax = 0; ax < 3; ax++) {
                        const double v = ax == 0 ? r.vx : (ax == 1 ? r.vy : r.vz);
                        const double p = ax == 0 ? r.px : (ax == 1 ? r.py : r.pz);
                        const int im = ax == 0 ? r.imx : (ax == 1 ? r.imy : r.imz);
                        const int di = v < direction ? 1 : -1;
                        const double np = p + (double)di;  // ray.pos + / - unit vector (init.py:94-96)
                        // floor(p + d) == floor(p) + d for |p| < 2^52
                        nl[ax][0] = ax == 0 ? lx + di : lx;
                        nl[ax][1] = ax == 1 ? ly + di : ly;
                        nl[ax][2] = ax == 2 ? lz + di : lz;
                        nm[ax][0] = r.imx;
                        nm[ax][1] = r.imy;
                        nm[ax][2] = r.imz;
                        nentry[ax] = r.entry;
                        // init.py:100-102: the point stays in the current chunk when it is inside its inclusive box
                        // (the other two coordinates are the ray's own, already inside); else Camera.chunk_get
                        // (init.py:28-33) snaps every coordinate of the point
                        foreign[ax] = !(np >= (double)im && np <= (double)im + cs);
                        if (foreign[ax]) {
                            const int nfx = nl[ax][0] + r.imx, nfy = nl[ax][1] + r.imy, nfz = nl[ax][2] + r.imz;
                            nm[ax][0] = (nfx >> P.cs_shift) << P.cs_shift;
                            nm[ax][1] = (nfy >> P.cs_shift) << P.cs_shift;
                            nm[ax][2] = (nfz >> P.cs_shift) << P.cs_shift;
                            nl[ax][0] = nfx - nm[ax][0];
                            nl[ax][1] = nfy - nm[ax][1];
                            nl[ax][2] = nfz - nm[ax][2];
                            nentry[ax] = chunk_entry_i(P, (nm[ax][0] - P.origin32[0]) >> P.cs_shift,
                                                       (nm[ax][1] - P.origin32[1]) >> P.cs_shift,
                                                       (nm[ax][2] - P.origin32[2]) >> P.cs_shift);
                            cnt[VRT_C_CHUNK_GET]++;
                        }
                    }
                    const uint8_t* naddr[3];
#pragma unroll
                    for (int ax = 0; ax < 3; ax++) {
                        naddr[ax] = nullptr;
                        if (nentry[ax]) {
                            const uint8_t* nbase = foreign[ax] ? chunk_base(P, nentry[ax]) : r.base;
                            naddr[ax] = voxel_addr(P, s_tab, nbase, nentry[ax], nm[ax][0], nm[ax][1], nm[ax][2], nl[ax][0],
                                                   nl[ax][1], nl[ax][2]);
                            cnt[VRT_C_NBR]++;
                        }
                    }
                    int nid[3] = {0, 0, 0};
#pragma unroll
                    for (int ax = 0; ax < 3; ax++)
                        if (naddr[ax]) nid[ax] = *naddr[ax];
                    bool solid[3];
#pragma unroll
                    for (int ax = 0; ax < 3; ax++) solid[ax] = nid[ax] != 0 && s_mats[(nid[ax] - 1) * 8 + 5] == m_ior;
                    if (!solid[0]) r.vx -= r.vx * m_ior * 2;
                    if (!solid[1]) r.vy -= r.vy * m_ior * 2;
                    if (!solid[2]) r.vz -= r.vz * m_ior * 2;
                }
                // ---- advance inside a present chunk (init.py:114-116) ----
                const double stepsize = r.stepd;
                r.step += stepsize;
                r.px += r.vx * stepsize;
                r.py += r.vy * stepsize;
                r.pz += r.vz * stepsize;
                cnt[VRT_C_ADV]++;
                state = LANE_MARCH;
            }
        }

        // ------------------------------------------------------------------ ENDED: background, outputs
#ifdef VRT_DIAG
        unsigned long long dg_t3 = clock64();
        dg_cyc[2] += dg_t3 - dg_t2;
#endif
        // ENDED (and with it the refill) is served every `end_period`-th pass, so that it runs with about that many
        // times more lanes; always when nothing is left to march
        const bool serve_ended = (pass % P.end_period) == 0 || __ballot(state == LANE_MARCH) == 0ull;
        if (serve_ended && state == LANE_ENDED) {
            state = LANE_IDLE;
            const int64_t ray = P.ray0 + r.off;
            if (exhausted) {
                bool queued = false;
                if (P.retrace_list) {
                    const uint32_t slot = atomicAdd(P.retrace_count, 1u);
                    if (slot < P.retrace_cap) {
                        P.retrace_list[slot] = (uint32_t)r.off;
                        queued = true;
                    }
                }
                if (!queued) n_exhausted++;
            } else {
                // ---- lib.material_background (lib.py:463-476) ----
                int cr = r.cr, cg = r.cg, cb = r.cb;
                double energy = r.energy;
                if (st.has_background) {
                    double a = 1 / pow_cached(pc, 1 + r.bounces, 1 + st.falloff);
                    if (!(a < 1)) a = 1;
                    const double up = r.vy > 0 ? r.vy : 0;
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
                // init.py:141
                double e = energy + st.shutter;
                if (!(e < 1)) e = 1;
                const int alpha = (int)__builtin_rint(e * 255);
                if (P.ray_rgba) P.ray_rgba[ray] = (uint32_t)cr | ((uint32_t)cg << 8) | ((uint32_t)cb << 16) | ((uint32_t)alpha << 24);
                cnt[VRT_C_BROKE] = broke ? 1 : 0;
                cnt[VRT_C_DRAW] = r.ndraw;
                if (RECORD && P.rays) {
                    vrt_ray& o = P.rays[ray];
                    int x = 0, y = 0, s = 0;
                    double detail;
                    if (tile) {
                        const int64_t p = ray / P.g.smax;
                        s = (int)(ray - p * P.g.smax);
                        x = P.g.pixels[2 * p];
                        y = P.g.pixels[2 * p + 1];
                        double dx, dy;
                        int ns;
                        pixel_setup(st, x, y, dx, dy, detail, ns);
                        detail = detail / (1 + s * st.lod_samples) * (1 - st.lod_random * r.row[0]);
                    } else {
                        detail = P.expl_detail[ray];
                    }
                    o.x = x; o.y = y; o.s = s;
                    o.color[0] = cr; o.color[1] = cg; o.color[2] = cb;
                    o.alpha = alpha;
                    o.ntrav = nseen;
#pragma unroll
                    for (int j = 0; j < VRT_NCOUNTERS; j++) o.counters[j] = cnt[j];
                    o.detail = detail; o.energy = energy; o.step = r.step; o.life = r.life; o.bounces = r.bounces;
                    o.pos[0] = r.px; o.pos[1] = r.py; o.pos[2] = r.pz;
                    o.vel[0] = r.vx; o.vel[1] = r.vy; o.vel[2] = r.vz;
                }
#pragma unroll
                for (int j = 0; j < VRT_NCOUNTERS; j++) tot[j] += cnt[j];
                n_done++;
                if (LIST) n_retraced++;
            }
        }
#ifdef VRT_DIAG
        dg_cyc[3] += clock64() - dg_t3;
#endif
    }

    // ------------------------------------------------------------------ statistics
#pragma unroll
    for (int j = 0; j < VRT_NCOUNTERS; j++) {
        if (tot[j]) atomicAdd(&s_stats[j], (unsigned long long)tot[j]);
    }
    if (n_done) atomicAdd(&s_stats[VRT_S_RAYS], (unsigned long long)n_done);
    if (n_retraced) atomicAdd(&s_stats[VRT_S_RNG_RETRACED], (unsigned long long)n_retraced);
    if (n_exhausted) atomicAdd(&s_stats[VRT_S_RNG_EXHAUSTED], (unsigned long long)n_exhausted);
#ifdef VRT_DIAG
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_stats[12], dg_inner);
        atomicAdd(&s_stats[13], dg_march_lanes);
        atomicAdd(&s_stats[14], dg_outer);
        atomicAdd(&s_stats[15], dg_hit_lanes);
        atomicAdd(&s_stats[9], dg_cyc[0]);
        atomicAdd(&s_stats[10], dg_cyc[1]);
        atomicAdd(&s_stats[11], dg_cyc[2]);
        atomicAdd(&s_stats[8], dg_cyc[3] << 32);
    }
#endif
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
// world voxelisation: the object loop of Window.chunk_update (init.py:398-444)
// ---------------------------------------------------------------------------------------------
// Sprite.pos_rotated (data.py:338-371): where a model rotated in quarter turns is read for local position (x, y, z)
__device__ __forceinline__ void rotate_index(const vrt_object& o, int& x, int& y, int& z) {
    const int ex = o.size[0] - 1, ey = o.size[1] - 1, ez = o.size[2] - 1;
    const int ax = o.turns[0], ay = o.turns[1], az = o.turns[2];
    int a, b, c;
    if (ax && o.size[1] == o.size[2]) {
        a = x;
        b = ax == 1 ? ez - z : (ax == 2 ? ey - y : z);
        c = ax == 1 ? y : (ax == 2 ? ez - z : ey - y);
        x = a, y = b, z = c;
    }
    if (ay && o.size[0] == o.size[2]) {
        a = ay == 1 ? z : (ay == 2 ? ex - x : ez - z);
        b = y;
        c = ay == 1 ? ex - x : (ay == 2 ? ez - z : x);
        x = a, y = b, z = c;
    }
    if (az && o.size[0] == o.size[1]) {
        a = az == 1 ? ey - y : (az == 2 ? ex - x : y);
        b = az == 1 ? x : (az == 2 ? ey - y : ex - x);
        c = z;
        x = a, y = b, z = c;
    }
}

// one workgroup per world chunk: every voxel asks the objects in order for their voxel at its position (the later
// object wins, the dict union of init.py:437-439) and the chunk's table entry says whether anything is there
__global__ void __launch_bounds__(VRT_BLOCK) voxelize_kernel(const vrt_object* objects, int n_objects, const uint8_t* models,
                                                             const uint8_t* remap, int ox, int oy, int oz, int dy, int dz,
                                                             int cs, uint32_t* table, uint8_t* voxels) {
    const int chunk = blockIdx.x;
    const int cx = chunk / (dy * dz), cy = (chunk / dz) % dy, cz = chunk % dz;
    const int wx0 = ox + cx * cs, wy0 = oy + cy * cs, wz0 = oz + cz * cs;
    uint8_t* block = voxels + (int64_t)chunk * cs * cs * cs;
    int any = 0;
    for (int i = threadIdx.x; i < cs * cs * cs; i += VRT_BLOCK) {
        const int lx = i / (cs * cs), ly = (i / cs) % cs, lz = i % cs;
        const int wx = wx0 + lx, wy = wy0 + ly, wz = wz0 + lz;
        int id = 0;
        for (int k = 0; k < n_objects; k++) {
            const vrt_object& o = objects[k];
            if (wx < o.mins[0] || wy < o.mins[1] || wz < o.mins[2] || wx >= o.maxs[0] || wy >= o.maxs[1] || wz >= o.maxs[2])
                continue;
            int x = wx - o.mins[0], y = wy - o.mins[1], z = wz - o.mins[2];
            rotate_index(o, x, y, z);
            if (x < 0 || y < 0 || z < 0 || x >= o.size[0] || y >= o.size[1] || z >= o.size[2]) continue;
            const int local = models[o.model + ((int64_t)x * o.size[1] + y) * o.size[2] + z];
            if (local) id = remap[o.remap + local];
        }
        block[voxel_offset(cs, lx, ly, lz)] = (uint8_t)id;
        any |= id;
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) table[chunk] = any ? ((uint32_t)(chunk + 1) | (1u << 24)) : 0u;
}

// ---------------------------------------------------------------------------------------------
// camera chunk selection: Window.chunk_update's loop over chunks (init.py:447-452)
// ---------------------------------------------------------------------------------------------
struct SelectParams {
    int32_t origin[3], dims[3];
    int32_t cs, cs_shift, chunk_radius, chunk_lod, culling;
    double cam[3], dist_max;
    int32_t t_origin[3], t_dims[3];
    const unsigned long long* t_keys;
};
__global__ void __launch_bounds__(VRT_BLOCK) select_chunks_kernel(SelectParams S, const uint32_t* world, uint32_t* out) {
    const int64_t n = (int64_t)S.dims[0] * S.dims[1] * S.dims[2];
    const int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = world[i] & 0xffffffu;
    uint32_t e = 0;
    if (slot) {
        const int cz = (int)(i % S.dims[2]), cy = (int)((i / S.dims[2]) % S.dims[1]), cx = (int)(i / ((int64_t)S.dims[2] * S.dims[1]));
        const int px = S.origin[0] + cx * S.cs, py = S.origin[1] + cy * S.cs, pz = S.origin[2] + cz * S.cs;
        bool keep = !S.culling;
        if (!keep && S.t_keys) {  // post_chunk in traversed (init.py:447)
            const int tx = (px - S.t_origin[0]) >> S.cs_shift, ty = (py - S.t_origin[1]) >> S.cs_shift,
                      tz = (pz - S.t_origin[2]) >> S.cs_shift;
            if ((unsigned)tx < (unsigned)S.t_dims[0] && (unsigned)ty < (unsigned)S.t_dims[1] && (unsigned)tz < (unsigned)S.t_dims[2])
                keep = S.t_keys[((int64_t)tx * S.t_dims[1] + ty) * S.t_dims[2] + tz] != ~0ull;
        }
        if (keep) {  // init.py:448-449
            const double dx = (double)(px + S.chunk_radius) - S.cam[0], dy = (double)(py + S.chunk_radius) - S.cam[1],
                         dz = (double)(pz + S.chunk_radius) - S.cam[2];
            const double dist = __builtin_sqrt(dx * dx + dy * dy + dz * dz);
            const double q = __builtin_trunc(dist / (S.dist_max / (double)(1 + S.chunk_lod)));
            const int lod = q < (double)S.chunk_lod ? (int)q : S.chunk_lod;
            e = slot | ((uint32_t)(lod + 1) << 24);
        }
    }
    out[i] = e;
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
static constexpr int D_SLOW = 113;   // draws in the retrace table (all outputs that need no state twist)
static constexpr int64_t SLOW_CAP_MIN = 1 << 21;  // rays per launch that may be re-traced with a 113-draw row:
static constexpr int64_t SLOW_CAP_MAX = 1 << 23;  // 1/8 of the launch, within these bounds
static constexpr int64_t FULL_CAP = 1 << 12;      // of those, rays that may be re-traced again with D_FULL_DEV draws

// ray slots per march launch: every launch ends with a drain phase in which the last, longest rays finish in
// mostly empty waves, so fewer and larger launches are better (VRT_BATCH_LOG2 overrides, 12..30)
static int64_t batch_rays() {
    static int64_t b = 0;
    if (!b) {
        const char* e = getenv("VRT_BATCH_LOG2");
        int l = e ? atoi(e) : 28;
        if (l < 12) l = 12;
        if (l > 30) l = 30;
        b = (int64_t)1 << l;
    }
    return b;
}

static inline int grid_for(int64_t n) { return (int)((n + VRT_BLOCK - 1) / VRT_BLOCK); }

// Small per-frame buffers (statistics, counters, the per-frame pow memo) are cleared by a kernel, not by
// hipMemsetAsync: a captured hipMemsetAsync node of such a buffer was seen to fill it with a stale 16-byte pattern
// when the graph is replayed (ROCm 7.2, tests/test_gpu_parity.py::test_frame_is_graph_capturable).
__global__ void __launch_bounds__(VRT_BLOCK) clear_words_kernel(uint32_t* p, int n) {
    for (int i = blockIdx.x * VRT_BLOCK + threadIdx.x; i < n; i += gridDim.x * VRT_BLOCK) p[i] = 0u;
}
static inline void clear_words(void* p, int64_t bytes, hipStream_t stream) {
    hipLaunchKernelGGL(clear_words_kernel, dim3(1), dim3(VRT_BLOCK), 0, stream, (uint32_t*)p, (int)(bytes / 4));
}
static inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

// march grid: persistent workgroups; each wave owns a contiguous range of the launch's rays
static int march_grid(int64_t n) {
    static int cap = -1;
    if (cap < 0) {
        const char* e = getenv("VRT_MARCH_GRID");
        cap = e ? atoi(e) : 1024;
        if (cap < 1) cap = 1;
    }
    int64_t g = (n + VRT_BLOCK * 4 - 1) / (VRT_BLOCK * 4);  // at least ~4 rays per lane
    if (g < 1) g = 1;
    return (int)(g < cap ? g : cap);
}

// The pow memo (see PowCache) caches a pure function, x -> vrt_pow(x, y) for one exponent y, so it may outlive the
// frame: one 4-KB table per (device, exponent), owned by the library, zeroed once when it is created and only ever
// inserted into afterwards.  Without it every workgroup of every frame starts cold and recomputes the same few
// dozen powers (0.3-0.4 ms per frame on MI355X, whatever the frame size).  VRT_POW_MEMO=frame restores the
// per-frame table in the workspace; more than 16 (device, exponent) pairs fall back to it as well.
struct PowMemo {
    int dev;
    double y;
    unsigned long long* buf;
};
static std::mutex g_memo_mu;
static PowMemo g_memos[16];
static int g_n_memos = 0;
static unsigned long long* device_pow_memo(double y) {
    static int per_frame = -1;
    if (per_frame < 0) {
        const char* e = getenv("VRT_POW_MEMO");
        per_frame = (e && e[0] == 'f') ? 1 : 0;
    }
    if (per_frame) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_memo_mu);
    for (int i = 0; i < g_n_memos; i++)
        if (g_memos[i].dev == dev && g_memos[i].y == y) return g_memos[i].buf;
    if (g_n_memos >= 16) return nullptr;
    unsigned long long* buf = nullptr;
    if (hipMalloc((void**)&buf, 2 * VRT_PW_SLOTS * 8) != hipSuccess) return nullptr;
    if (hipMemset(buf, 0, 2 * VRT_PW_SLOTS * 8) != hipSuccess) {  // synchronous: visible to every stream
        (void)hipFree(buf);
        return nullptr;
    }
    g_memos[g_n_memos++] = PowMemo{dev, y, buf};
    return buf;
}

// rays per hand-out for a launch of n rays: 512, but 128 for small launches (about one 512-ray chunk per wave would
// leave nothing to balance: config 2's 2 M rays march in 0.62 instead of 0.73 ms); VRT_CHUNK overrides
static int march_chunk(int64_t n) {
    static int c = -2;
    if (c == -2) {
        const char* e = getenv("VRT_CHUNK");
        c = e ? atoi(e) : -1;
        if (e && c < 0) c = 0;
    }
    if (c >= 0) return c;
    return n <= ((int64_t)1 << 22) ? 128 : VRT_CHUNK;
}

// deeper speculation when the voxel data is far larger than L2 + Infinity Cache (VRT_SPEC_DEEP=0/1 forces it)
static bool march_deep(const vrt_scene* sc) {
    static int c = -2;
    if (c == -2) {
        const char* e = getenv("VRT_SPEC_DEEP");
        c = e ? atoi(e) : -1;
    }
    if (c >= 0) return c != 0;
    const int64_t bytes = (int64_t)sc->n_slots * sc->chunk_size * sc->chunk_size * sc->chunk_size;
    return bytes > ((int64_t)512 << 20);
}

static int march_end_period() {
    static int c = -1;
    if (c < 0) {
        const char* e = getenv("VRT_END_PERIOD");
        c = e ? atoi(e) : 2;
        if (c < 1) c = 1;
    }
    return c;
}

// new events per march pass before the slow bodies run: 36, and 24 with the 8-step march (more steps per pass bring
// more events per pass; config 5: 353 instead of 376 ms); VRT_MARCH_T overrides
static int march_threshold(bool deep) {
    static int t = -1;
    if (t < 0) {
        const char* e = getenv("VRT_MARCH_T");
        t = e ? atoi(e) : 0;
        if (t < 0) t = 0;
        if (t > 64) t = 64;
    }
    return t ? t : (deep ? 24 : 36);
}

extern "C" {

int vrt_abi_version(void) { return VRT_ABI_VERSION; }

const char* vrt_status_string(int s) {
    switch (s) {
        case VRT_OK: return "ok";
        case VRT_ERR_ARG: return "invalid argument or unsupported setting";
        case VRT_ERR_HIP: return "HIP runtime error";
        case VRT_ERR_WORKSPACE: return "workspace or plan buffer too small";
        case VRT_ERR_NO_DEVICE: return "no HIP device";
        case VRT_ERR_PLAN: return "tile plan does not match the pixel list / settings";
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

// ---- plan ----
static int64_t plan_words(const vrt_settings* st) {
    // seeds are (1+x)(1+y)(1+s) <= width * height * smax
    int64_t max_seed = (int64_t)st->width * st->height * vrt_max_samples(st);
    return max_seed / 32 + 1;
}
static int plan_supported(const vrt_settings* st) {
    double m = (double)st->width * (double)st->height * (double)vrt_max_samples(st);
    return m < 4294967296.0;
}

int vrt_plan_bytes(const vrt_settings* st, int64_t n_px, int64_t* plan_bytes, int64_t* scratch_bytes) {
    if (check_settings(st) != VRT_OK || n_px < 0 || !plan_bytes || !scratch_bytes) return VRT_ERR_ARG;
    if (!plan_supported(st)) return VRT_ERR_ARG;
    int64_t slots = n_px * vrt_max_samples(st);
    if (slots >= 4294967295ll) return VRT_ERR_ARG;
    int64_t words = plan_words(st);
    int64_t blocks = (words + VRT_SCAN_WORDS - 1) / VRT_SCAN_WORDS;
    *plan_bytes = 64 + align256(slots * 4) + align256(slots * 4) + 256;
    *scratch_bytes = align256(words * 4) + align256(words * 4) + align256(blocks * 4) + 256;
    return VRT_OK;
}

int vrt_plan_build(const vrt_settings* st, const int32_t* d_pixels_xy, int64_t n_px, void* d_plan, int64_t plan_bytes,
                   void* d_scratch, int64_t scratch_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int64_t need_plan = 0, need_scratch = 0;
    int rc = vrt_plan_bytes(st, n_px, &need_plan, &need_scratch);
    if (rc != VRT_OK) return rc;
    if (!d_plan || !d_scratch || (n_px > 0 && !d_pixels_xy)) return VRT_ERR_ARG;
    if (plan_bytes < need_plan || scratch_bytes < need_scratch) return VRT_ERR_WORKSPACE;
    const int smax = vrt_max_samples(st);
    const int64_t slots = n_px * smax;
    const int64_t words = plan_words(st);
    const int64_t blocks = (words + VRT_SCAN_WORDS - 1) / VRT_SCAN_WORDS;
    char* pl = (char*)d_plan;
    PlanHeader* hdr = (PlanHeader*)pl;
    uint32_t* seed_list = (uint32_t*)(pl + 64);
    uint32_t* ray_seedidx = (uint32_t*)(pl + 64 + align256(slots * 4));
    char* sc = (char*)d_scratch;
    uint32_t* bitmap = (uint32_t*)sc;
    uint32_t* prefix = (uint32_t*)(sc + align256(words * 4));
    uint32_t* block_sums = (uint32_t*)(sc + 2 * align256(words * 4));
    PlanHeader h;
    h.magic = VRT_PLAN_MAGIC;
    h.n_px = (uint64_t)n_px;
    h.n_slots = (uint64_t)slots;
    h.n_distinct = 0;
    h.settings_hash = plan_hash(st, n_px);
    h.n_words = (uint64_t)words;
    h.pad[0] = h.pad[1] = 0;
    HIP_TRY(hipMemcpyAsync(hdr, &h, sizeof h, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemsetAsync(bitmap, 0, (size_t)words * 4, stream));
    TileGeom g;
    g.pixels = d_pixels_xy;
    g.n_px = n_px;
    g.smax = smax;
    if (n_px > 0) hipLaunchKernelGGL(plan_mark_kernel, dim3(grid_for(n_px)), dim3(VRT_BLOCK), 0, stream, *st, g, bitmap);
    hipLaunchKernelGGL(plan_blocksum_kernel, dim3((unsigned)blocks), dim3(VRT_BLOCK), 0, stream, bitmap, words, block_sums);
    hipLaunchKernelGGL(plan_scan_sums_kernel, dim3(1), dim3(VRT_BLOCK), 0, stream, block_sums, blocks, hdr);
    hipLaunchKernelGGL(plan_compact_kernel, dim3((unsigned)blocks), dim3(VRT_BLOCK), 0, stream, bitmap, words, block_sums,
                       prefix, seed_list);
    if (slots > 0)
        hipLaunchKernelGGL(plan_index_kernel, dim3(grid_for(slots)), dim3(VRT_BLOCK), 0, stream, *st, g, bitmap, prefix,
                           ray_seedidx);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

// ---- workspace ----
struct WsLayout {
    int64_t batch;  // rays per march launch
    int64_t slow_cap;
    int64_t full_cap;  // rays per launch that may be re-traced a second time, with a D_FULL_DEV-draw row
    int64_t off_table, off_slow, off_full, off_rec, off_rgba, off_list, off_list_full, off_count, off_pow, total;
};
static WsLayout ws_layout(const vrt_settings* st, int64_t n_px, int64_t n_distinct, int fast_draws) {
    WsLayout w;
    int64_t rays = n_px * vrt_max_samples(st);
    w.batch = rays < batch_rays() ? rays : batch_rays();
    if (w.batch < 1) w.batch = 1;
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o += align256(bytes); return r; };
    w.off_table = take((n_distinct > 0 ? n_distinct : 1) * (int64_t)fast_draws * 8);
    w.slow_cap = w.batch / 8;
    if (w.slow_cap < SLOW_CAP_MIN) w.slow_cap = w.batch < SLOW_CAP_MIN ? w.batch : SLOW_CAP_MIN;
    if (w.slow_cap > SLOW_CAP_MAX) w.slow_cap = SLOW_CAP_MAX;
    w.off_slow = take(w.slow_cap * VRT_SLOW_STRIDE * 8);
    w.full_cap = w.slow_cap < FULL_CAP ? w.slow_cap : FULL_CAP;
    w.off_full = take(w.full_cap * D_FULL_DEV * 8);
    w.off_rec = take(w.batch * 8 * 4);
    w.off_rgba = take(rays * 4);
    w.off_list = take(w.slow_cap * 4);
    w.off_list_full = take(w.full_cap * 4);
    w.off_count = take(256);
    w.off_pow = take(2 * VRT_PW_SLOTS * 8);
    w.total = o;
    return w;
}

static int fast_draws_ok(int32_t d) { return d == 32 || d == 64; }

int vrt_workspace_bytes(const vrt_settings* st, int64_t n_px, int64_t n_distinct, int32_t fast_draws, int64_t* bytes) {
    if (check_settings(st) != VRT_OK || n_px < 0 || n_distinct < 0 || !bytes || !fast_draws_ok(fast_draws)) return VRT_ERR_ARG;
    *bytes = ws_layout(st, n_px, n_distinct, fast_draws).total;
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
        if (sc->origin[a] < -(1ll << 28) || sc->origin[a] + (int64_t)sc->dims[a] * st->chunk_size > (1ll << 28)) return VRT_ERR_ARG;
        P.origin[a] = sc->origin[a];
        P.origin32[a] = (int32_t)sc->origin[a];
        P.t_origin32[a] = 0;
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
            if (trav->origin[a] < -(1ll << 28) || trav->origin[a] + (int64_t)trav->dims[a] * st->chunk_size > (1ll << 28)) return VRT_ERR_ARG;
            P.t_origin[a] = trav->origin[a];
            P.t_origin32[a] = (int32_t)trav->origin[a];
            P.t_dims[a] = trav->dims[a];
        }
        P.t_keys = trav->d_keys;
    }
    P.stats = d_stats;
    P.g.pixels = nullptr;
    P.g.n_px = 0;
    P.g.smax = 1;
    P.ray_seedidx = nullptr;
    P.expl_detail = nullptr;
    P.list = nullptr;
    P.list_count = nullptr;
    P.ray_rgba = nullptr;
    P.rays = nullptr;
    P.retrace_list = nullptr;
    P.retrace_count = nullptr;
    P.pow_global = nullptr;
    P.queue_head = nullptr;
    P.retrace_cap = 0;
    P.list_cap = 0;
    P.chunk = march_chunk(0);  // the launch sites set it for their ray count
    P.end_period = march_end_period();
    P.first_draw = 0;
    P.threshold = march_threshold(march_deep(sc));
    return VRT_OK;
}

int vrt_draw_table_bytes(int64_t n_distinct, int32_t fast_draws, int64_t* bytes) {
    if (n_distinct < 0 || !bytes || !fast_draws_ok(fast_draws)) return VRT_ERR_ARG;
    *bytes = align256((n_distinct > 0 ? n_distinct : 1) * (int64_t)fast_draws * 8);
    return VRT_OK;
}

int vrt_draw_table_build(const vrt_settings* st, int64_t n_px, const void* d_plan, int64_t n_distinct, int32_t fast_draws,
                         double* d_table, int64_t table_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (check_settings(st) != VRT_OK || !plan_supported(st) || n_px < 0 || !d_plan || !d_table) return VRT_ERR_ARG;
    const int64_t rays = n_px * vrt_max_samples(st);
    int64_t need = 0;
    if (n_distinct < 0 || n_distinct > rays || vrt_draw_table_bytes(n_distinct, fast_draws, &need) != VRT_OK) return VRT_ERR_ARG;
    if (table_bytes < need) return VRT_ERR_WORKSPACE;
    if (n_distinct == 0) return VRT_OK;
    const uint32_t* seed_list = (const uint32_t*)((const char*)d_plan + 64);
    ProfScope ps(stream, VRT_PROF_RNG);
    hipLaunchKernelGGL(rng_plan_kernel, dim3(grid_for(n_distinct)), dim3(VRT_BLOCK), 0, stream, seed_list, n_distinct,
                       st->seed_nonce, (int)fast_draws, d_table);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int vrt_render_tile(const vrt_scene* scene, const vrt_settings* st, const vrt_camera* cam, const int32_t* d_pixels_xy,
                    int64_t n_px, const void* d_plan, int64_t n_distinct, int32_t fast_draws,
                    const double* d_draw_table, void* d_workspace, int64_t workspace_bytes, float* d_rgba_f32,
                    uint8_t* d_image_u8, uint32_t* d_ray_rgba, vrt_ray* d_rays, uint64_t* d_stats,
                    const vrt_traversed* trav, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MarchParams P;
    int rc = fill_params(P, scene, st, cam, trav, d_stats);
    if (rc != VRT_OK) return rc;
    if (n_px < 0 || (n_px > 0 && !d_pixels_xy) || !d_workspace || !d_plan) return VRT_ERR_ARG;
    if (!plan_supported(st)) return VRT_ERR_ARG;
    const int smax = vrt_max_samples(st);
    const int64_t rays = n_px * smax;
    if (rays >= 4294967295ll) return VRT_ERR_ARG;
    // the plan header (n_distinct, settings hash) is read back and validated by the caller once, after
    // vrt_plan_build; no host synchronisation happens here
    const char* pl = (const char*)d_plan;
    const uint32_t* seed_list = (const uint32_t*)(pl + 64);
    const uint32_t* ray_seedidx = (const uint32_t*)(pl + 64 + align256(rays * 4));
    if (n_distinct < 0 || n_distinct > rays || !fast_draws_ok(fast_draws)) return VRT_ERR_ARG;
    WsLayout w = ws_layout(st, n_px, n_distinct, fast_draws);
    if (workspace_bytes < w.total) return VRT_ERR_WORKSPACE;
    char* ws = (char*)d_workspace;
    const double* table = d_draw_table ? d_draw_table : (const double*)(ws + w.off_table);
    double* t_slow = (double*)(ws + w.off_slow);
    double* recbuf = (double*)(ws + w.off_rec);
    uint32_t* rgba = d_ray_rgba ? d_ray_rgba : (uint32_t*)(ws + w.off_rgba);
    uint32_t* list = (uint32_t*)(ws + w.off_list);
    uint32_t* count = (uint32_t*)(ws + w.off_count);
    unsigned long long* pow_global = device_pow_memo(1 + st->falloff);
    const bool frame_memo = pow_global == nullptr;
    if (frame_memo) pow_global = (unsigned long long*)(ws + w.off_pow);
    clear_words(d_stats, sizeof(uint64_t) * VRT_NSTATS, stream);
    if (n_px == 0) return VRT_OK;
    if (frame_memo) clear_words(pow_global, 2 * VRT_PW_SLOTS * 8, stream);
    TileGeom g;
    g.pixels = d_pixels_xy;
    g.n_px = n_px;
    g.smax = smax;
    if (!d_draw_table) {  // no table from vrt_draw_table_build: seed this frame's draws into the workspace
        ProfScope ps(stream, VRT_PROF_RNG);
        hipLaunchKernelGGL(rng_plan_kernel, dim3(grid_for(n_distinct)), dim3(VRT_BLOCK), 0, stream, seed_list, n_distinct,
                           st->seed_nonce, (int)fast_draws, (double*)(ws + w.off_table));
    }
    P.g = g;
    P.ray_seedidx = ray_seedidx;
    P.ray_rgba = rgba;
    P.rays = d_rays;
    P.pow_global = pow_global;
    P.first_draw = 1 + (st->dof != 0.0 ? 2 : 0);
    for (int64_t ray0 = 0; ray0 < rays; ray0 += w.batch) {
        const int64_t n = (rays - ray0) < w.batch ? (rays - ray0) : w.batch;
        RayRec rec;
        rec.vx = recbuf;
        rec.vy = recbuf + w.batch;
        rec.vz = recbuf + 2 * w.batch;
        rec.life = recbuf + 3 * w.batch;
        clear_words(count, 256, stream);  // retrace counts + the launch-wide ray counters
        {
            ProfScope ps(stream, VRT_PROF_RAYGEN);
            hipLaunchKernelGGL(raygen_tile_kernel, dim3(grid_for(n)), dim3(VRT_BLOCK), 0, stream, *st, *cam, g, ray_seedidx,
                               table, (int)fast_draws, ray0, n, rec);
        }
        P.rec = rec;
        P.ray0 = ray0;
        P.n = n;
        P.chunk = march_chunk(n);
        P.list = nullptr;
        P.list_count = nullptr;
        P.draws = table;
        P.n_draws = fast_draws;
        P.draw_stride = fast_draws;
        P.retrace_list = list;
        P.retrace_count = count;
        P.retrace_cap = (uint32_t)w.slow_cap;
        P.queue_head = (unsigned long long*)(count + 2);
        {
            ProfScope ps(stream, VRT_PROF_MARCH);
            if (d_rays) hipLaunchKernelGGL((march_kernel<true, false>), dim3(march_grid(n)), dim3(VRT_BLOCK), 0, stream, P);
            else if (march_deep(scene))
                hipLaunchKernelGGL((march_kernel<false, false, VRT_SPEC_DEEP>), dim3(march_grid(n)), dim3(VRT_BLOCK), 0, stream, P);
            else hipLaunchKernelGGL((march_kernel<false, false>), dim3(march_grid(n)), dim3(VRT_BLOCK), 0, stream, P);
        }
        // rays that ran out of draws: per-ray 113-draw rows, device-side count (no host sync)
        ProfScope ps(stream, VRT_PROF_RETRACE);
        const int rgrid = 256;
        hipLaunchKernelGGL(rng_list_kernel, dim3(rgrid), dim3(VRT_BLOCK), 0, stream, *st, g, ray0, list, count,
                           (uint32_t)w.slow_cap, t_slow);
        uint32_t* list_full = (uint32_t*)(ws + w.off_list_full);
        double* t_full = (double*)(ws + w.off_full);
        P.list = list;
        P.list_count = count;
        P.draws = t_slow;
        P.n_draws = D_SLOW;
        P.draw_stride = VRT_SLOW_STRIDE;
        P.list_cap = (uint32_t)w.slow_cap;
        P.retrace_list = list_full;  // rays that outrun even 113 draws: third tier below
        P.retrace_count = count + 8;
        P.retrace_cap = (uint32_t)w.full_cap;
        P.queue_head = (unsigned long long*)(count + 4);
        if (d_rays) hipLaunchKernelGGL((march_kernel<true, true>), dim3(rgrid), dim3(VRT_BLOCK), 0, stream, P);
        else hipLaunchKernelGGL((march_kernel<false, true>), dim3(rgrid), dim3(VRT_BLOCK), 0, stream, P);
        // third tier: full-state MT19937, D_FULL_DEV draws per ray; usually empty (both kernels return at once)
        hipLaunchKernelGGL(rng_list_full_kernel, dim3(64), dim3(64), 0, stream, *st, g, ray0, list_full, count + 8,
                           (uint32_t)w.full_cap, t_full);
        P.list = list_full;
        P.list_count = count + 8;
        P.draws = t_full;
        P.n_draws = D_FULL_DEV;
        P.draw_stride = D_FULL_DEV;
        P.list_cap = (uint32_t)w.full_cap;
        P.retrace_list = nullptr;
        P.retrace_count = nullptr;
        P.retrace_cap = 0;
        P.queue_head = (unsigned long long*)(count + 10);
        if (d_rays) hipLaunchKernelGGL((march_kernel<true, true>), dim3(64), dim3(VRT_BLOCK), 0, stream, P);
        else hipLaunchKernelGGL((march_kernel<false, true>), dim3(64), dim3(VRT_BLOCK), 0, stream, P);
    }
    if (d_rgba_f32 || d_image_u8) {
        ProfScope ps(stream, VRT_PROF_RESOLVE);
        hipLaunchKernelGGL(resolve_kernel, dim3(grid_for(n_px)), dim3(VRT_BLOCK), 0, stream, *st, g, rgba, d_rgba_f32, d_image_u8);
    }
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int vrt_trace_workspace_bytes(int64_t n_rays, int64_t* bytes) {
    if (n_rays < 0 || !bytes) return VRT_ERR_ARG;
    *bytes = align256((n_rays > 0 ? n_rays : 1) * 8 * 4) + 512;
    return VRT_OK;
}

int vrt_trace_rays(const vrt_scene* scene, const vrt_settings* st, const vrt_camera* cam, const double* d_dir_x,
                   const double* d_dir_y, const double* d_detail, const double* d_draws, int32_t n_draws, int64_t n_rays,
                   void* d_workspace, int64_t workspace_bytes, vrt_ray* d_rays, uint64_t* d_stats,
                   const vrt_traversed* trav, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MarchParams P;
    int rc = fill_params(P, scene, st, cam, trav, d_stats);
    if (rc != VRT_OK) return rc;
    if (n_rays < 0 || n_draws < 0 || !d_rays || !d_workspace) return VRT_ERR_ARG;
    if (n_rays > 0 && (!d_dir_x || !d_dir_y || !d_detail || (n_draws > 0 && !d_draws))) return VRT_ERR_ARG;
    if (st->dof != 0.0 && n_draws < 2) return VRT_ERR_ARG;  // the lens jitter alone takes two draws (init.py:41-42)
    int64_t need = 0;
    vrt_trace_workspace_bytes(n_rays, &need);
    if (workspace_bytes < need) return VRT_ERR_WORKSPACE;
    clear_words(d_stats, sizeof(uint64_t) * VRT_NSTATS, stream);
    if (n_rays == 0) return VRT_OK;
    double* recbuf = (double*)d_workspace;
    RayRec rec;
    rec.vx = recbuf;
    rec.vy = recbuf + n_rays;
    rec.vz = recbuf + 2 * n_rays;
    rec.life = recbuf + 3 * n_rays;
    hipLaunchKernelGGL(raygen_explicit_kernel, dim3(grid_for(n_rays)), dim3(VRT_BLOCK), 0, stream, *st, *cam, d_dir_x, d_dir_y,
                       d_detail, d_draws, (int)n_draws, n_rays, rec);
    unsigned long long* qh = (unsigned long long*)((char*)d_workspace + align256(n_rays * 8 * 4));
    clear_words(qh, 256, stream);
    P.queue_head = qh;
    P.pow_global = device_pow_memo(1 + st->falloff);
    P.retrace_cap = 0;
    P.expl_detail = d_detail;
    P.rec = rec;
    P.ray0 = 0;
    P.n = n_rays;
    P.chunk = march_chunk(n_rays);
    P.draws = d_draws;
    P.n_draws = n_draws;
    P.draw_stride = n_draws;
    P.first_draw = (st->dof != 0.0) ? 2 : 0;
    P.rays = d_rays;
    hipLaunchKernelGGL((march_kernel<true, false>), dim3(march_grid(n_rays)), dim3(VRT_BLOCK), 0, stream, P);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int vrt_rng_draws(const uint64_t* d_seeds, int64_t n_seeds, int32_t n_draws, double* d_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_seeds < 0 || (n_seeds > 0 && (!d_seeds || !d_out))) return VRT_ERR_ARG;
    if (n_draws < 2 || n_draws > 4096) return VRT_ERR_ARG;
    if (n_seeds == 0) return VRT_OK;
    if (n_draws > D_SLOW)  // beyond the outputs that need no state twist: full-state generator
        hipLaunchKernelGGL(rng_seeds_full_kernel, dim3((unsigned)((n_seeds + 63) / 64)), dim3(64), 0, stream, d_seeds, n_seeds,
                           (int)n_draws, d_out);
    else
        hipLaunchKernelGGL(rng_seeds_kernel, dim3(grid_for(n_seeds)), dim3(VRT_BLOCK), 0, stream, d_seeds, n_seeds,
                           (int)n_draws, d_out);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int vrt_select_chunks(const uint32_t* d_world_table, const int64_t* origin, const int32_t* dims, int32_t cs,
                      const double* cam_pos, double dist_max, int32_t chunk_lod, int32_t culling, const vrt_traversed* prev,
                      uint32_t* d_camera_table, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_world_table || !origin || !dims || !cam_pos || !d_camera_table) return VRT_ERR_ARG;
    if (cs < 8 || (cs & (cs - 1)) || cs > 256 || chunk_lod < 0 || chunk_lod > 254 || !(dist_max > 0)) return VRT_ERR_ARG;
    SelectParams S;
    int shift = 0;
    while ((1 << shift) < cs) shift++;
    S.cs = cs;
    S.cs_shift = shift;
    S.chunk_radius = (int)__builtin_rint((double)cs / 2);
    S.chunk_lod = chunk_lod;
    S.culling = culling ? 1 : 0;
    S.dist_max = dist_max;
    S.t_keys = nullptr;
    for (int a = 0; a < 3; a++) {
        if (dims[a] <= 0 || (origin[a] % cs) != 0 || origin[a] < -(1ll << 28) || origin[a] + (int64_t)dims[a] * cs > (1ll << 28))
            return VRT_ERR_ARG;
        S.origin[a] = (int32_t)origin[a];
        S.dims[a] = dims[a];
        S.cam[a] = cam_pos[a];
        S.t_origin[a] = 0;
        S.t_dims[a] = 0;
    }
    if (prev && prev->d_keys) {
        for (int a = 0; a < 3; a++) {
            if (prev->dims[a] <= 0 || (prev->origin[a] % cs) != 0 || prev->origin[a] < -(1ll << 28) ||
                prev->origin[a] + (int64_t)prev->dims[a] * cs > (1ll << 28))
                return VRT_ERR_ARG;
            S.t_origin[a] = (int32_t)prev->origin[a];
            S.t_dims[a] = prev->dims[a];
        }
        S.t_keys = (const unsigned long long*)prev->d_keys;
    }
    const int64_t n = (int64_t)dims[0] * dims[1] * dims[2];
    hipLaunchKernelGGL(select_chunks_kernel, dim3(grid_for(n)), dim3(VRT_BLOCK), 0, stream, S, d_world_table, d_camera_table);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int vrt_release_caches(void) {
    std::lock_guard<std::mutex> lock(g_memo_mu);
    int rc = VRT_OK;
    for (int i = 0; i < g_n_memos; i++) {
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess || hipSetDevice(g_memos[i].dev) != hipSuccess ||
            hipFree(g_memos[i].buf) != hipSuccess)
            rc = VRT_ERR_HIP;
        (void)hipSetDevice(cur);
    }
    g_n_memos = 0;
    return rc;
}

int vrt_profile_begin(void) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (auto& e : g_prof) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    g_prof.clear();
    g_prof_on = true;
    return VRT_OK;
}

int vrt_profile_end(double* ms, int64_t* launches) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
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

int vrt_voxelize(const vrt_object* d_objects, int32_t n_objects, const uint8_t* d_models, const uint8_t* d_remap,
                 const int64_t* origin, const int32_t* dims, int32_t cs, uint32_t* d_world_table, uint8_t* d_voxels,
                 void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_objects < 0 || !origin || !dims || !d_world_table || !d_voxels) return VRT_ERR_ARG;
    if (n_objects > 0 && (!d_objects || !d_models || !d_remap)) return VRT_ERR_ARG;
    if (cs < 8 || cs > 256 || (cs & (cs - 1))) return VRT_ERR_ARG;
    const int64_t n = (int64_t)dims[0] * dims[1] * dims[2];
    if (dims[0] < 1 || dims[1] < 1 || dims[2] < 1 || n > (1 << 24) - 2) return VRT_ERR_ARG;
    for (int a = 0; a < 3; a++) {
        if (origin[a] % cs) return VRT_ERR_ARG;
        if (origin[a] < -(1ll << 30) || origin[a] + (int64_t)dims[a] * cs > (1ll << 30)) return VRT_ERR_ARG;
    }
    hipLaunchKernelGGL(voxelize_kernel, dim3((unsigned)n), dim3(VRT_BLOCK), 0, stream, d_objects, (int)n_objects, d_models,
                       d_remap, (int)origin[0], (int)origin[1], (int)origin[2], (int)dims[1], (int)dims[2], (int)cs,
                       d_world_table, d_voxels);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
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
