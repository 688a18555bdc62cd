// vrt_kernels.hip -- MI355X (gfx950) voxel ray-march: the reference's Camera.tile -> Camera.trace ->
// Frame.get_voxel -> lib.material / lib.material_background path (reference init.py:37-150,
// data.py:136-145, lib.py:448-476) as hand-written HIP behind the C ABI of include/vrt.h.
//
// Per frame (vrt_render_tile):
//   rng_plan_kernel   MT19937 init_by_array + first 32 draws for every DISTINCT seed of the tile plan, state in
//                     registers only (CPython random.seed(int) / random.random(); init.py:137,139; lib.py:434)
//   raygen_tile_kernel  per ray: tile()'s detail LOD and trace()'s lens quaternion (init.py:131-139, 41-43) -> the ray
//                     table (one 64-byte record per ray slot); uniform work, full lanes.  With static seeds both tables are
//                     frame-invariant and are built once (vrt_draw_table_build, vrt_ray_table_build); a frame without a
//                     cached ray table writes none -- the march's lanes make their records (take_ray, PERPIX 3)
//   frame_begin_kernel  the frame's statistics, counters and (vrt_traversed.reset) traversed keys, in one launch
//   march_pool_kernel / march_kernel   persistent waves: each lane marches a ray through the chunk/voxel grid
//                     (init.py:66-116), shades with the default PBR material + sky (lib.py:448-476), and takes the
//                     next ray of the wave's range when it finishes.  Frames of 5 Mi rays and more run the pool
//                     variant, which regroups rays between the lanes of a wave through LDS so that a body executes
//                     with nearly all lanes active; smaller launches, records and re-traces run one ray per lane
//   resolve_kernel    per-pixel mean of the samples (lib.average, init.py:145) -> fp32 RGBA + RGBA8
// plus the plan kernels (static seed index, built once per pixel list) and two re-trace launches for rays that need
// more than 32 (then more than 113) random draws; the lane that takes such a ray seeds its draw row itself.
//
// Arithmetic is binary64 in the reference's evaluation order; build with -ffp-contract=off.
// gfx950 only: 64-wide waves are assumed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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
static unsigned g_prof_kinds = ~0u;  // bit k: kind k is timed (vrt_profile_begin_kinds)
static std::vector<ProfEvent> g_prof;
static std::mutex g_prof_mu;
struct ProfScope {
    hipStream_t s;
    bool on;
    ProfEvent e;
    ProfScope(hipStream_t stream, int kind) : s(stream), on(g_prof_on && ((g_prof_kinds >> kind) & 1u)) {
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
// (ST: vrt_settings in whatever address space the caller holds it)
template <class ST>
__host__ __device__ static inline void pixel_setup(const ST& st, int x, int y, double& dir_x, double& dir_y,
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
// per-ray seeds
// ---------------------------------------------------------------------------------------------
// static (nonce 0): random.seed((1 + x) * (1 + y) * (1 + sample)) (init.py:137), shared by every ray with that product.
// non-static: the reference seeds nothing per sample and re-seeds from OS entropy after every pixel (init.py:136-147),
// so every ray has noise of its own; here each ray slot gets its own stream, (y * width + x) * max_samples + s + nonce.
template <class ST>
__host__ __device__ static inline uint64_t ray_seed(const ST& st, int smax, int x, int y, int s) {
    if (st.seed_nonce == 0) return (uint64_t)((uint32_t)(1 + x) * (uint32_t)(1 + y) * (uint32_t)(1 + s));
    return ((uint64_t)y * (uint64_t)st.width + (uint64_t)x) * (uint64_t)smax + (uint64_t)s + st.seed_nonce;
}

// ---------------------------------------------------------------------------------------------
// tile plan: distinct seeds of a pixel list (see vrt.h)
// ---------------------------------------------------------------------------------------------
#define VRT_PLAN_MAGIC 0x5652544e414c5032ull
struct PlanHeader {
    uint64_t magic, n_px, n_slots, n_distinct, settings_hash, n_words;
    uint64_t full_frame;  // 1: the pixel list is the whole window in the reference's x-major order (pixel p = (p / H, p % H))
    uint64_t pad;
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

__global__ void __launch_bounds__(VRT_BLOCK) plan_mark_kernel(vrt_settings st, TileGeom g, uint32_t* bitmap, PlanHeader* hdr) {
    int64_t p = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (p >= g.n_px) return;
    int x = g.pixels[2 * p], y = g.pixels[2 * p + 1];
    // (the header starts with full_frame = 1 when the count fits; any pixel out of place clears it)
    if ((int64_t)x * st.height + y != p) hdr->full_frame = 0;
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

// The same for exactly 32 draws (the frame tables' default width): the second sweep's words s[2..64] -- what outputs
// 2..63 need beside s[k + 397] -- stay in 63 registers from the moment the main chains produce them, instead of a second
// pair of chains re-deriving them while the outputs are made (6 % fewer chain steps; the loops are fully unrolled, so the
// array never exists in memory).
template <bool PAIR>
__device__ __forceinline__ void mt_seed_draws32(uint64_t seed, double* out) {
    constexpr int D = 32, K = 2 * D;
    const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
    const uint32_t addA = key0;
    const uint32_t addB = key1 ? key1 + 1u : key0;
    const uint32_t* init = c_mt_init.v;
    uint32_t m = init[0];
    m = (init[1] ^ ((m ^ (m >> 30)) * 1664525u)) + addA;
    const uint32_t first1 = m;
#pragma unroll 2
    for (int i = 2; i < 624; i++) m = (init[i] ^ ((m ^ (m >> 30)) * 1664525u)) + (((i - 1) & 1) ? addB : addA);
    const uint32_t mt1 = (first1 ^ ((m ^ (m >> 30)) * 1664525u)) + ((623 & 1) ? addB : addA);
#define VRT_MT_STEP(p, q, i)                                                            \
    p = (init[i] ^ ((p ^ (p >> 30)) * 1664525u)) + ((((i) - 1) & 1) ? addB : addA);      \
    q = (p ^ ((q ^ (q >> 30)) * 1566083941u)) - (uint32_t)(i);
    uint32_t p = first1, q = mt1;
    uint32_t keep[K - 1];  // keep[j] = s[2 + j]
#pragma unroll
    for (int i = 2; i <= K; i++) {
        VRT_MT_STEP(p, q, i)
        keep[i - 2] = q;
    }
#pragma unroll 2
    for (int i = K + 1; i < 397; i++) { VRT_MT_STEP(p, q, i) }
    { VRT_MT_STEP(p, q, 397) }
    const uint32_t s397 = q;
    { VRT_MT_STEP(p, q, 398) }
    const uint32_t s398 = q;
    uint32_t prev = 0;
    double d1 = 0, dprev = 0;
#pragma unroll
    for (int k = 2; k < K; k++) {
        VRT_MT_STEP(p, q, 397 + k)
        const uint32_t o = mt_out(keep[k - 2], keep[k - 1], q);
        if (k & 1) {
            const int d = k >> 1;
            const double v = mt_res53(prev, o);
            if (d == 1) d1 = v;
            else if (!PAIR) out[d] = v;
            else if (d & 1) *reinterpret_cast<double2*>(out + d - 1) = make_double2(dprev, v);
            else dprev = v;
        } else {
            prev = o;
        }
    }
#pragma unroll 2
    for (int i = 397 + K; i < 624; i++) { VRT_MT_STEP(p, q, i) }
#undef VRT_MT_STEP
    const uint32_t s1 = (mt1 ^ ((q ^ (q >> 30)) * 1566083941u)) - 1u;
    const uint32_t o0 = mt_out(0x80000000u, s1, s397);
    const uint32_t o1 = mt_out(s1, keep[0], s398);
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
    if (D == 32) mt_seed_draws32<true>((uint64_t)seed_list[i] + nonce, table + i * 32);
    else mt_seed_draws<true>((uint64_t)seed_list[i] + nonce, D, table + i * D);
}

// re-trace rows: table[k * STRIDE + d] for the k-th listed ray (STRIDE even: rows 16-byte aligned); seeded by the lane that
// takes the ray (take_ray, SEED)
#define VRT_SLOW_STRIDE 114
#define D_SLOW_DEV 113

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

// non-static frames: one row per ray slot
__global__ void __launch_bounds__(VRT_BLOCK) rng_slots_kernel(vrt_settings st, TileGeom g, int D, double* table) {
    int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (i >= g.n_px * g.smax) return;
    const int64_t p = i / g.smax;
    const int s = (int)(i - p * g.smax);
    const int x = g.pixels[2 * p], y = g.pixels[2 * p + 1];
    double dx, dy, det;
    int ns;
    pixel_setup(st, x, y, dx, dy, det, ns);
    if (s >= ns) return;
    mt_seed_draws<true>(ray_seed(st, g.smax, x, y, s), D, table + i * D);
}

// ---------------------------------------------------------------------------------------------
// ray table: lens quaternion + life of every ray slot (init.py:131-139 and 41-43, 56): one lane per ray, no divergence
// ---------------------------------------------------------------------------------------------
#define VRT_RAY_WORDS 8  // doubles per ray slot: ox, oy, oz, ow, life, and the three draws of the ray's first rough hit
struct __align__(16) RayRecord {  // 64 bytes: one ray slot of the ray table
    double ox, oy, oz, ow;  // lens quaternion
    double life;            // < 0: unused sample slot
    double d0, d1, d2;      // the draws a first rough hit takes
};
static_assert(sizeof(RayRecord) == 8 * VRT_RAY_WORDS, "ray record is 64 bytes");
struct RayTab {  // the ray slots of the tile (or the explicit rays)
    RayRecord* rec;
};
static inline RayTab ray_tab_at(double* base, int64_t) {
    RayTab t;
    t.rec = reinterpret_cast<RayRecord*>(base);
    return t;
}

// the general case of vrt_sincos2 (an argument outside (-pi/4, pi/4), a zero, or a result too close to a rounding
// boundary for the fast kernels): out of line, so that the march kernels that make their own ray records carry one copy
struct Trig4 { double sa, ca, sb, cb; };
__device__ __noinline__ Trig4 lens_trig_slow(double a, double b) {
    Trig4 t;
    t.sa = vrt_sin(a);
    t.ca = vrt_cos(a);
    t.sb = vrt_sin(b);
    t.cb = vrt_cos(b);
    return t;
}
// init.py:41-43; lib.py:322-338 (vec3.quaternion of vec3(0, -lens_x, +lens_y))
template <class ST>
__device__ __forceinline__ void lens_quaternion(const ST& st, double lens, double dir_x, double dir_y, double jx,
                                                double jy, double& ox, double& oy, double& oz, double& ow) {
    const double lens_x = (dir_x / st.proportions) * lens + jx;
    const double lens_y = (dir_y * st.proportions) * lens + jy;
    const double deg2rad = 3.141592653589793 / 180.0;  // math.radians
    const double rad_y = (-lens_x) * deg2rad, rad_z = lens_y * deg2rad;
    const double sin_x = 0.0, cos_x = 1.0;  // sin(0.0 / 2), cos(0.0 / 2)
    double t[4];  // sin, cos of rad_y / 2 and of rad_z / 2
    if (!vrt_sincos2_fast(rad_y / 2, rad_z / 2, t)) {
        const Trig4 g = lens_trig_slow(rad_y / 2, rad_z / 2);
        t[0] = g.sa; t[1] = g.ca; t[2] = g.sb; t[3] = g.cb;
    }
    const double sin_y = t[0], cos_y = t[1], sin_z = t[2], cos_z = t[3];
    ox = sin_x * cos_y * cos_z - cos_x * sin_y * sin_z;
    oy = cos_x * sin_y * cos_z - sin_x * cos_y * sin_z;
    oz = cos_x * cos_y * sin_z + sin_x * sin_y * cos_z;
    ow = cos_x * cos_y * cos_z + sin_x * sin_y * cos_z;
}
// init.py:44-45; lib.py:353-358 (rot.multiply(o)), 372-376 (vec_forward)
__device__ __forceinline__ void camera_forward(double qx, double qy, double qz, double qw, double ox, double oy, double oz,
                                               double ow, double& vx, double& vy, double& vz) {
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

// With dof == 0, lod_random == 0 and lod_samples == 0 neither the lens quaternion nor the life of a ray depends on its
// sample (init.py:41-43, 139: no lens jitter, detail / (1 + s * 0) * (1 - 0 * draw) == detail exactly): the ray table then
// holds one record per PIXEL (d0 = its sample count), 16 x less memory at config 5, and the march fetches a ray's
// first-hit draws from the draw table itself.
__host__ __device__ static inline bool ray_table_per_pixel(const vrt_settings& st) {
    return st.dof == 0.0 && st.lod_random == 0.0 && st.lod_samples == 0.0;
}

// The record of sample slot s of pixel (x, y) from the draw row of its seed (init.py:131-139, 41-43, 56): life < 0 for a
// slot the pixel does not use (its row is not touched then).
template <class ST>
__device__ __forceinline__ void ray_record_from_draws(const ST& st, double lens, int x, int y, int s, const double* row,
                                                      RayRecord& rec) {
    double dir_x, dir_y, detail;
    int ns;
    pixel_setup(st, x, y, dir_x, dir_y, detail, ns);
    if (s >= ns) {
        rec.ox = rec.oy = rec.oz = rec.ow = 0.0;
        rec.d0 = rec.d1 = rec.d2 = 0.5;
        rec.life = -1.0;
        return;
    }
    // the draws a first rough hit will take (lib.py:457) travel with the ray, so that the hit needs no second look
    // into the draw table
    const int fd = 1 + (st.dof != 0.0 ? 2 : 0);
    const double r0 = row[0], r1 = row[1], r2 = row[2];
    rec.d0 = row[fd];
    rec.d1 = row[fd + 1];
    rec.d2 = row[fd + 2];
    // init.py:139
    detail = detail / (1 + s * st.lod_samples) * (1 - st.lod_random * r0);
    double jx = 0, jy = 0;
    if (st.dof != 0.0) {
        jx = rand_amp(r1, st.dof);
        jy = rand_amp(r2, st.dof);
    }
    lens_quaternion(st, lens, dir_x, dir_y, jx, jy, rec.ox, rec.oy, rec.oz, rec.ow);
    rec.life = (st.dist_max - st.dist_min) * detail;  // init.py:56
}

// tile rays: draws come from the draw table (row of the ray's seed, or of the ray itself in a non-static frame)
__global__ void __launch_bounds__(VRT_BLOCK) raygen_tile_kernel(vrt_settings st, double lens, TileGeom g,
                                                                const uint32_t* ray_seedidx, const double* table,
                                                                int n_draws, RayTab tab, int per_pixel) {
    const int64_t ray = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (per_pixel) {  // one record per pixel
        if (ray >= g.n_px) return;
        const int x = g.pixels[2 * ray], y = g.pixels[2 * ray + 1];
        double dir_x, dir_y, detail;
        int ns;
        pixel_setup(st, x, y, dir_x, dir_y, detail, ns);
        RayRecord rec;
        lens_quaternion(st, lens, dir_x, dir_y, 0.0, 0.0, rec.ox, rec.oy, rec.oz, rec.ow);
        rec.life = (st.dist_max - st.dist_min) * detail;  // init.py:56
        rec.d0 = (double)ns;
        rec.d1 = rec.d2 = 0.0;
        tab.rec[ray] = rec;
        return;
    }
    if (ray >= g.n_px * g.smax) return;
    const int64_t p = ray / g.smax;
    const int s = (int)(ray - p * g.smax);
    const int64_t rowi = st.seed_nonce ? ray : (int64_t)ray_seedidx[ray];  // (0xFFFFFFFF for an unused slot: never read)
    RayRecord rec;
    ray_record_from_draws(st, lens, g.pixels[2 * p], g.pixels[2 * p + 1], s, table + rowi * n_draws, rec);
    if (rec.life < 0.0) tab.rec[ray].life = -1.0;
    else tab.rec[ray] = rec;
}

// explicit rays (vrt_trace_rays): draws[i * n_draws + k]
__global__ void __launch_bounds__(VRT_BLOCK) raygen_explicit_kernel(vrt_settings st, double lens, const double* dir_x,
                                                                    const double* dir_y, const double* detail,
                                                                    const double* draws, int n_draws, int64_t n, RayTab tab) {
    int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    double jx = 0, jy = 0;
    if (st.dof != 0.0 && n_draws >= 2) {
        jx = rand_amp(draws[i * n_draws + 0], st.dof);
        jy = rand_amp(draws[i * n_draws + 1], st.dof);
    }
    double ox, oy, oz, ow;
    lens_quaternion(st, lens, dir_x[i], dir_y[i], jx, jy, ox, oy, oz, ow);
    const int fd = st.dof != 0.0 ? 2 : 0;
    const bool have = fd + 3 <= n_draws;
    RayRecord rec;
    rec.ox = ox;
    rec.oy = oy;
    rec.oz = oz;
    rec.ow = ow;
    rec.life = (st.dist_max - st.dist_min) * detail[i];
    rec.d0 = have ? draws[i * n_draws + fd] : 0.5;
    rec.d1 = have ? draws[i * n_draws + fd + 1] : 0.5;
    rec.d2 = have ? draws[i * n_draws + fd + 2] : 0.5;
    tab.rec[i] = rec;
}

// ---------------------------------------------------------------------------------------------
// occupancy words: bit b of word w = (voxels[64 w + b] != 0)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(VRT_BLOCK) occupancy_kernel(const uint8_t* voxels, int64_t n_words, uint64_t* occ) {
    // one lane per 16 bytes: four lanes build one word
    const int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (i >= n_words * 4) return;
    const uint4 v = reinterpret_cast<const uint4*>(voxels)[i];
    uint32_t bits = 0;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int b = 0; b < 4; b++) bits |= (((w[k] >> (8 * b)) & 255u) ? 1u : 0u) << (4 * k + b);
    }
    // combine the four 16-bit parts of a word (lanes 4j .. 4j + 3 are in one wave: 256 threads, aligned)
    uint64_t part = (uint64_t)bits << (16 * (threadIdx.x & 3));
    part |= (uint64_t)__shfl_xor((long long)part, 1);
    part |= (uint64_t)__shfl_xor((long long)part, 2);
    if ((threadIdx.x & 3) == 0) occ[i >> 2] = part;
}

#define VRT_WT_GUARD 32          // guard cells either side of the world in the world-axis offset tables (march_step_w)
#define VRT_WT_DVMAX 4.0         // ... and the largest step per axis a ray may take to look ahead with them: 8 x 4 <= 32
#define VRT_WT_LDS_MAX 16384     // ... and the most LDS the three tables may take
#ifndef VRT_WADDR_DEFAULT
#define VRT_WADDR_DEFAULT 0
#endif
// ---------------------------------------------------------------------------------------------
// world-axis offset tables (vrt_world_tables_build; read by march_step_w)
// ---------------------------------------------------------------------------------------------
// With the voxel blocks in table order (VRT_SCENE_LAYOUT_DENSE) the voxel of world cell (x, y, z) sits at X[x] + Y[y] + Z[z]:
// each term is the axis' share of the block number times the block size plus the axis' part of vrt_voxel_offset.  The
// VRT_WT_GUARD cells either side of the world read 2^30: any sum that holds one lies beyond the voxel buffer (at most
// 2^30 bytes for such scenes), reads as empty and never has the block number of a real cell.
__global__ void __launch_bounds__(VRT_BLOCK) world_tables_kernel(int d0, int d1, int d2, int cs, int cs_shift, uint32_t* out) {
    const int dims[3] = {d0, d1, d2};
    int i = (int)(blockIdx.x * VRT_BLOCK + threadIdx.x);
    for (int a = 0; a < 3; a++) {
        const int cells = dims[a] * cs, n = cells + 2 * VRT_WT_GUARD;
        if (i < n) {
            const int c = i - VRT_WT_GUARD;
            uint32_t v = 0x40000000u;
            if (c >= 0 && c < cells) {
                const unsigned stride = (a == 0 ? (unsigned)(d1 * d2) : (a == 1 ? (unsigned)d2 : 1u)) << (3 * cs_shift);
                const int l = c & (cs - 1);
                v = (uint32_t)(c >> cs_shift) * stride + (uint32_t)voxel_offset(cs, a == 0 ? l : 0, a == 1 ? l : 0, a == 2 ? l : 0);
            }
            out[i] = v;
            return;
        }
        i -= n;
        out += n;
    }
}

// ---------------------------------------------------------------------------------------------
// march
// ---------------------------------------------------------------------------------------------
#define VRT_PW_SLOTS 256   // (128: the memo thrashes -- config 3 226 ms, config 5 2 036 ms)
#define VRT_TOT_COLS 16    // columns of the workgroup's event totals: lane l adds into column l % 16 (1.7 KiB of LDS less than
                           // one column per lane: what the settled bitmap lacked beside the ray pools at config 3)
#define VRT_CHUNK 512
#define VRT_SPEC 4        // reference iterations fetched together per march pass in the generic-resolution kernel ...
#ifndef VRT_TAB_PIPELINE
#define VRT_TAB_PIPELINE 1
#endif
#ifndef VRT_HIT_BATCH
#define VRT_HIT_BATCH 1
#endif
#ifndef VRT_SPEC_DEEP
#define VRT_SPEC_DEEP 8   // ... and in the resolution <= 2 kernels and for scenes far larger than the caches (march_deep)
#endif
#ifndef VRT_WAVES_PER_SIMD
#define VRT_WAVES_PER_SIMD 4   // march_kernel's occupancy target: workgroups per CU = waves per SIMD (256-thread workgroups)
#endif
#define VRT_CT_LDS_MAX 4096      // chunk tables up to this many cells are copied to LDS (16 KiB)
#define VRT_TRAV_LDS_MAX 65536   // traversed boxes up to this many cells get a "settled" bitmap in LDS (8 KiB per workgroup)
struct MarchParams {
    vrt_settings st;
    vrt_camera cam;
    // scene
    int32_t origin_c[3], t_origin_c[3];  // scene box / traversed box origin in chunks (both are multiples of the chunk size)
    int32_t dims[3];
    int32_t cs, cs_shift;
    int32_t n_materials;
    const uint32_t* chunk_table;
    const uint8_t* voxels;
    const uint64_t* occ;         // occupancy words (lookup variants 1 and 2 only)
    uint32_t vox_bytes;          // n_slots * cs^3 (< 2^32: the march reads the voxels through a raw buffer resource)
    const double* materials;
    // traversed
    int32_t t_dims[3];
    uint64_t* t_keys;
    // rays of this launch: ray slots [ray0, ray0 + n) of the table
    TileGeom g;                  // tile mode (g.pixels != NULL)
    const uint32_t* ray_seedidx; // tile mode, static seeds: draw-table row of every ray slot (NULL: row = ray slot)
    const double* expl_detail;   // explicit mode
    RayTab tab;
    int64_t ray0, n;
    const uint32_t* list;        // LIST: ray offsets to re-trace and their device-side count
    const uint32_t* list_count;
    const double* draws;         // rows of draw_stride doubles, n_draws of them valid
    int32_t n_draws, draw_stride;
    int32_t first_draw;          // draws already consumed by ray generation
    int32_t per_pixel;           // tile mode: 1 = the ray table holds one record per pixel (ray_table_per_pixel); 2 = there
                                 // is no ray table, a lane that takes a ray derives its record from the draw row (take_ray)
    double lens;                 // ... the camera's lens for that (vrt_camera.lens)
    // scheduling (never changes a result)
    int32_t t_hit, t_end;        // lanes waiting for the HIT / ENDED body before the wave leaves the march loop for it
    int32_t max_iters;           // march iterations per pass at most, while anything waits
    int32_t chunk;               // rays per hand-out from queue_head; 0 = static range per wave
    int32_t brick_lds_off;       // lookup variant 2: byte offset of the brick slots in the dynamic LDS
    int32_t pool_lds_off;        // march_pool_kernel: byte offset of the waves' ray pools in the dynamic LDS
    int32_t pool_swap_min;       // ... rays a pass must be able to bring into the lanes before it exchanges any
    int32_t pool_refill_min;     // ... idle lanes before a marching wave stops to fetch new rays
    int32_t pool_keep;           // ... lanes that must still march for a pass to take another MARCH step at once
    int32_t list_seed;           // LIST: the lane that takes a listed ray seeds its draw row first (take_ray, SEED): 1 = the 113
                                 // draws that need no state twist, 2 = D_FULL_DEV draws from a full-state MT19937
    int32_t prefix_draws;        // LIST: > 0 = the frame's march counted a re-traced ray's events up to the hit at which
                                 // a row of this many draws ran out; the re-trace takes them off again (hit_body)
    int32_t ct_cells;            // > 0: the chunk table (that many cells) is copied to LDS
    int32_t ct_identity;         // the chunk table is (i + 1) | 1 << 24 (VRT_SCENE_TABLE_IS_IDENTITY): computed, not read
    int32_t trav_words;          // > 0: per-wave settled bitmaps of that many 32-bit words in LDS
    int32_t snap_flags;          // CF_* bits (march_lds sets them: the last word on what a launch keeps in LDS)
    int32_t bm_window;           // a box too large for a settled bitmap gets one over the 32^3 cells around the camera, where
                                 // nearly all visits fall (CF_BM_WINDOW): the window's first cell, 10 bits per axis
    int32_t wt_on;               // the scene's blocks lie in table order (VRT_SCENE_LAYOUT_DENSE) and its world-axis offset
                                 // tables fit LDS: the march looks ahead across chunk borders (march_step_w)
    int32_t wt_lds_off;          // ... byte offset of the three tables in the dynamic LDS
    int32_t wt_cells[3];         // ... the world's extent in cells
    int32_t wt_min[3];           // ... and its first cell (the scene's origin)
    int32_t wt_words;            // ... words of the three tables together
    const uint32_t* wt_table;    // ... the tables (vrt_scene.d_world_tables)
    // outputs
    uint32_t* ray_rgba;          // [rays of the tile] packed result (tile mode)
    vrt_ray* rays;               // debug records (may be NULL)
    uint64_t* stats;
    uint32_t* retrace_list;      // rays whose draws ran out are appended here (may be NULL)
    uint32_t* retrace_count;
    unsigned long long* pow_global;  // [2 * VRT_PW_SLOTS]: keys then values, shared by every workgroup
    unsigned long long* queue_head;  // launch-wide ray counter (zeroed before every launch)
    // tiled hand-out (march_pool_kernel, scenes beyond the caches; see tile_ticket): the whole window's rays are handed out as
    // square pixel tiles in Morton order, one eighth of that order per XCD (each has its own L2) from a head of its own
    const PlanHeader* plan_hdr;      // ... the plan's header: only a list that is the whole window in x-major order is tiled
    uint32_t* tile_heads;            // ... 8 heads, 128 bytes apart (zeroed before the launch); NULL: no tiling
    int32_t tile_px_log2;            // ... log2 of the pixels of a column per hand-out (chunk / sample slots) = the tile's edge
    int32_t tile_log2;               // ... the padded tile grid is 2^tile_log2 squared
    int32_t tile_ntx, tile_nty;      // ... tiles across and down the window
    uint32_t retrace_cap;            // capacity of retrace_list
    uint32_t list_cap;               // LIST: capacity of `list` (its count may have run past it)
};

// local cell of world cell (f // res) * res for res >= 3 (int // int, exact: |f| < 2^31, res <= 255); rare
__device__ __noinline__ int3 snap_generic3(int res, int imx, int imy, int imz, int lx, int ly, int lz) {
    int3 o;
    o.x = (int)__builtin_floor((double)(lx + imx) / (double)res) * res - imx;
    o.y = (int)__builtin_floor((double)(ly + imy) / (double)res) * res - imy;
    o.z = (int)__builtin_floor((double)(lz + imz) / (double)res) * res - imz;
    return o;
}

// chunk table entry of chunk cell (cx, cy, cz), 0 outside the scene box; ct: the table's copy in LDS (used if P.ct_cells)
// (d0, d1, d2: the table's dimensions, P.dims -- the march's re-snap passes per-lane copies, see MarchCtx::dm)
template <class PT>
__device__ __forceinline__ uint32_t chunk_entry_i(const PT& P, const __attribute__((address_space(3))) uint32_t* ct, int cx, int cy,
                                                  int cz, int d0, int d1, int d2, bool lds, bool identity) {
    if ((unsigned)cx >= (unsigned)d0 || (unsigned)cy >= (unsigned)d1 || (unsigned)cz >= (unsigned)d2) return 0;
    const int i = (cx * d1 + cy) * d2 + cz;
    // (two loads in two address spaces, never one load through a generic pointer: a flat load waits on both counters)
    if (lds) return ct[i];
    if (identity) return (uint32_t)(i + 1) | (1u << 24);  // (VRT_SCENE_TABLE_IS_IDENTITY: nothing to read)
    // (a table in memory -- too large for LDS and not the identity -- is waited for here, inside its own branch: a load
    // still in flight where the branches join makes the compiler wait for EVERY load there, the traversed key included,
    // which the march step wants to leave in flight until its voxel reads are out: resnap_commit<DEFER>)
    uint32_t e = P.chunk_table[i];
    asm volatile("" : "+v"(e));
    return e;
}
__device__ __forceinline__ const uint8_t* chunk_base(const MarchParams& P, uint32_t entry) {
    return P.voxels + ((int64_t)((entry & 0xffffffu) - 1u) << (3 * P.cs_shift));
}

// Recording a visited chunk (world chunk_min as integers) for the `traversed` list (init.py:72-73): the cell keeps the
// smallest key (ray index << 12 | re-snap index).  bm: the workgroup's "settled" bitmap in LDS, or nullptr.  A cell is
// settled once a wave has seen there a key smaller than wmin_key, the smallest key any ray the workgroup holds or
// will ever hold can produce (rays are handed out in increasing order; wmin_key is the minimum over the waves' own
// minima, each published as a value that only grows, so a stale read is a lower bound): no later visit by this
// workgroup can lower the cell, so it skips the global read.  A stale (larger) value read from the cell only causes
// a redundant atomic or delays the settling.
// cell of the traversed box for chunk_min (ccx, ccy, ccz) * chunk size: its index, -1 when nothing is recorded, -2 outside the box
// td: the box's dimensions (P.t_dims).  The march keeps them in vector registers (MarchCtx::td): as kernel arguments they
// do not fit the scalar file next to everything else, and the compiler re-reads them from memory inside every re-snap --
// two scalar-load round trips in a row, with nothing to do in between.
__device__ __forceinline__ int trav_cell(bool has_keys, int ccx, int ccy, int ccz, int to0, int to1, int to2, int td0, int td1,
                                         int td2) {  // chunk_min / chunk size; to*: P.t_origin_c; has_keys: P.t_keys != NULL
    if (!has_keys) return -1;
    const int cx = ccx - to0;
    const int cy = ccy - to1;
    const int cz = ccz - to2;
    if ((unsigned)cx >= (unsigned)td0 || (unsigned)cy >= (unsigned)td1 || (unsigned)cz >= (unsigned)td2) return -2;
    return (cx * td1 + cy) * td2 + cz;
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
#ifndef VRT_POW_INLINE
#define VRT_POW_INLINE __noinline__
#endif
__device__ VRT_POW_INLINE double pow_miss(unsigned long long* keys, unsigned long long* vals, unsigned long long* gkeys,
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
// slow path of pow_cached: the remaining probes, then the computation
__device__ VRT_POW_INLINE double pow_slow(unsigned long long* keys, unsigned long long* vals, unsigned long long* gkeys,
                                        unsigned long long* gvals, double x, double y) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const unsigned h = pow_hash(bits);
    for (int i = 1; i < VRT_PW_PROBES; i++) {
        const int sidx = (int)((h + i) & (VRT_PW_SLOTS - 1));
        const unsigned long long k = __hip_atomic_load(&keys[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == bits) {
            const unsigned long long v = __hip_atomic_load(&vals[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v != 0ull) return __longlong_as_double((long long)v);
            break;
        }
        if (k == 0ull) break;
    }
    return pow_miss(keys, vals, gkeys, gvals, x, y);
}
__device__ __forceinline__ double pow_cached(const PowCache& pc, double x, double y) {
    if (x == 1.0) return 1.0;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const int sidx = (int)pow_hash(bits);
    // first probe inline (nearly always the answer); key and value are read together
    const unsigned long long k = __hip_atomic_load(&pc.keys[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned long long v = __hip_atomic_load(&pc.vals[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (k == bits && v != 0ull) return __longlong_as_double((long long)v);
    return pow_slow(pc.keys, pc.vals, pc.gkeys, pc.gvals, x, y);
}

// vec3.normalize (lib.py:310-314): three binary64 divisions by the same positive divisor.  For operands in a range
// where v_div_scale_f64 does not rescale (no denormals, no extreme exponents) the compiler's division is
//   y = rcp(d) refined by two Newton steps; q = n * y; r = fma(-d, q, n); result = fma(r, y, q)
// -- the same y for every numerator, so it is computed once.  Outside that range (never seen in practice) each
// quotient is a plain division.  Zero numerators keep their sign.
__device__ __forceinline__ bool div_safe(double a) {
    const double m = __builtin_fabs(a);
    return m >= 0x1p-500 && m <= 0x1p500;
}
__device__ __forceinline__ double with_sign_of(double mag, double sgn) {  // |mag| with the sign bit of sgn (one v_bfi_b32)
    const unsigned hi = ((unsigned)__double2hiint(mag) & 0x7fffffffu) | ((unsigned)__double2hiint(sgn) & 0x80000000u);
    return __hiloint2double((int)hi, __double2loint(mag));
}
__device__ __forceinline__ void div3_same_divisor(double& a, double& b, double& c, double d) {
    // d is the largest of |a|, |b|, |c| (the only caller): the numerators need no upper bound of their own
    const bool fast = div_safe(d) & ((a == 0.0) | (__builtin_fabs(a) >= 0x1p-500)) & ((b == 0.0) | (__builtin_fabs(b) >= 0x1p-500)) &
                      ((c == 0.0) | (__builtin_fabs(c) >= 0x1p-500));
    if (fast) {
        double y = __builtin_amdgcn_rcp(d);
        double e = __builtin_fma(-d, y, 1.0);
        y = __builtin_fma(y, e, y);
        e = __builtin_fma(-d, y, 1.0);
        y = __builtin_fma(y, e, y);
        // d > 0, so a quotient has its numerator's sign; a zero numerator gives a zero of either sign from the last fma
        double q, r;
        q = a * y; r = __builtin_fma(-d, q, a); a = with_sign_of(__builtin_fma(r, y, q), a);
        q = b * y; r = __builtin_fma(-d, q, b); b = with_sign_of(__builtin_fma(r, y, q), b);
        q = c * y; r = __builtin_fma(-d, q, c); c = with_sign_of(__builtin_fma(r, y, q), c);
    } else {
        a = a / d;
        b = b / d;
        c = c / d;
    }
}

// state of the ray a lane is marching (the `ray` store of init.py:50-59 plus the chunk cursor of init.py:46-47)
struct Ray {
    double px, py, pz, vx, vy, vz;
    double step, life, bounces, energy;
    int nm4x, nm4y, nm4z;   // -4 * chunk_min (init.py:68) as integers: 4 * floor(pos) + nm4 is the LDS table byte index of the
                            // local coordinate (one shift-add)
    uint32_t entry;         // chunk table entry of the current chunk (0 = None)
    uint32_t boff;          // byte offset of its voxel block in the voxel buffer
    uint32_t color;         // r | g << 8 | b << 16 (init.py:51; channels stay in 0..255: lib.py:393-395 mixes towards albedo);
                            // bits 24..31: the material the march found, until the HIT body has shaded it
    int ndraw, resnaps;
    uint32_t off;           // offset of the ray in the launch
    uint32_t rowi;          // its draw-table row
    double d0, d1, d2;      // the draws of the ray's next rough hit (from the ray table, then requested after each rough hit)
};

// lane 0's value in scalar registers (all lanes active)
__device__ __forceinline__ unsigned long long wave_first_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// what a lane's ray waits for.  The three ENDED states say how the ray left the loop: its life ran out (init.py:66), the
// reference's `break` (init.py:86), or its draws ran out (the result is discarded and the ray re-traced with a longer row)
enum { LANE_IDLE = 0, LANE_MARCH = 1, LANE_HIT = 2, LANE_ENDED = 3, LANE_ENDED_BROKE = 4, LANE_ENDED_EXHAUSTED = 5 };
__device__ __forceinline__ int opaque_zero() {
    int z = 0;
    asm volatile("" : "+v"(z));
    return z;
}
enum { COLD_POS = 0, COLD_ROT = 3, COLD_DIST_MIN = 7, COLD_POW_Y, COLD_LOD_BOUNCES, COLD_MAX_LIGHT, COLD_MAX_BOUNCES1,
       COLD_SHUTTER, COLD_CS, COLD_INV_CS, COLD_N };  // (COLD_CS: the chunk size, a power of two: x * COLD_INV_CS == x / cs exactly)
enum { C_LOOKUP = 0, C_NBR, C_CGET, C_HIT, C_ADV, C_NLOCAL };  // event counters kept in registers

// LDS is addressed through address-space-3 pointers everywhere in the march: a generic pointer that the compiler cannot
// trace back to LDS becomes a flat load, which waits on both the LDS and the memory counter
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) char lds_char;

// what the bodies of the march read besides the kernel arguments: the workgroup's tables in LDS and a few constants
struct MarchCtx {
    const lds_char* tab;   // per-axis parts of vrt_voxel_offset: axis a's table begins tl[a] bytes behind it (tab_at)
    int tl[3];
    const lds_u32* ct;     // the chunk table's copy (if P.ct_cells)
    lds_u32* bm;           // "settled" bitmap of the traversed box (if has_bm)
    const lds_f64* mats;   // material records
    const lds_f64* cold;   // scalars only the slow bodies read (camera, shader settings): LDS, not SGPRs -- the kernel
                           // arguments alone would otherwise overflow the scalar register file.  Read through COLD(): an
                           // opaque zero in the index keeps every read a ds_read at its use (plain reads would be hoisted
                           // out of the loop into registers that live across it)
    lds_u32* tot;          // [VRT_NCOUNTERS + 1][64] totals over completed rays (+ their number), one column per lane index
    PowCache pc;
    __amdgpu_buffer_rsrc_t vox;  // the voxel bytes as a raw buffer: 32-bit offsets, out-of-range (~0) reads return 0
    unsigned cs4;
    double cs, inv_cs;     // chunk size (a power of two: x * inv_cs == x / cs exactly); kept_cs says whether these are to be
    bool kept_cs;          // used (registers) or COLD_CS / COLD_INV_CS (LDS: the look-ahead variant's register diet)
    bool has_bm, tile;

    int td[3];             // P.t_dims, one copy per lane (see trav_cell)
    int toc[3], oc[3], dm[3];  // ... and P.t_origin_c, P.origin_c, P.dims as the re-snap reads them (scalar registers: vector copies
                               // of these nine, tried in round 4, raised the spill count -- profiles/r04_experiments.md)
    const lds_char* wt;    // world-axis offset tables (P.wt_on): the entry of world cell g on axis a sits at byte
    int wb[3];             // (g << 2) + wb[a], VRT_WT_GUARD guard entries either side of the world included
    unsigned cs3;          // bytes of a chunk's voxel block
};
#define COLD(i) C.cold[(i) + opaque_zero()]
// MarchParams::snap_flags: the wave-uniform switches of the re-snap.  Every caller re-reads the word from the kernel
// arguments (one scalar load, asked for where its body begins): a uniform boolean that is hoisted out of the march loop lives
// there as a 64-bit lane mask, and the scalar file has none to spare -- each of them took two spill lanes.
enum { CF_HAS_BM = 1, CF_CT_LDS = 2, CF_CT_IDENTITY = 4, CF_HAS_KEYS = 8, CF_BM_WINDOW = 16 };

#ifdef VRT_DIAG
// diagnostic build only (tools/diag_march.py): per-phase cycles and lane counts summed over the waves of a launch
enum { DG_PASSES = 0, DG_CYC_REFILL, DG_CYC_MARCH, DG_CYC_HIT, DG_CYC_END, DG_ITERS, DG_MARCH_LANES, DG_HIT_EXEC,
       DG_HIT_LANES, DG_END_EXEC, DG_END_LANES, DG_REFILL_EXEC, DG_REFILL_LANES, DG_WAVE_CYCLES, DG_SNAP_ITERS,
       DG_SNAP_LANES, DG_BRICK_VISITS, DG_SWAPS, DG_SWAP_LANES, DG_EVICT_LANES, DG_CYC_SWAP, DG_VOID_LANES,
#ifdef VRT_DIAG_HIST  // NV / H: lanes whose speculation stayed valid for >= k positions / that advanced >= k
       DG_NV1, DG_H1 = DG_NV1 + 8, DG_N = DG_H1 + 8  // (8 rows each: the frame kernels look 8 positions ahead; deeper
                                                     // variants only fill the first 8 -- the rows are LDS the ray pools need)
#else
       DG_N
#endif
};
__device__ unsigned long long g_diag[DG_N];
// launch timeline in s_memrealtime ticks (100 MHz): [0] ~(first wave start), [1] ~(first time a wave found the ray queue
// empty), [2] last wave exit, [3] sum of the waves' exit times, [4] waves -- [0], [1] kept as maxima of the complement
__device__ unsigned long long g_diag_t[5];
// (wave-uniform sums kept in LDS, one row per wave, added by the first active lane: as per-lane register arrays they
// pushed the instrumented kernels into 512 bytes of scratch per lane and made them six times slower than the shipped ones)
#define DG_ADD(idx_, val_)                                                                                      \
    do {                                                                                                        \
        const unsigned long long v_ = (unsigned long long)(val_);  /* (evaluated by every active lane: ballots) */ \
        const unsigned long long e_ = __ballot(1);                                                              \
        if ((int)(threadIdx.x & 63) == __ffsll((long long)e_) - 1) dg.acc[idx_] += v_;                           \
    } while (0)
#define DG_TIME() __builtin_amdgcn_s_memtime()
struct DgLane {
    __attribute__((address_space(3))) unsigned long long* acc;
    // SURVEY.md 8d's layout-aware figure: how often a ray's 8^3 brick (floor(pos) >> 3) changes, plus one per ray
    unsigned brick;
    unsigned long long bv;
};
#define DG_BRICK()                                                                                                     \
    do {                                                                                                               \
        const unsigned key_ = (((unsigned)(int)__builtin_floor(r.px) >> 3) & 1023u) |                                  \
                              ((((unsigned)(int)__builtin_floor(r.py) >> 3) & 1023u) << 10) |                          \
                              ((((unsigned)(int)__builtin_floor(r.pz) >> 3) & 1023u) << 20);                           \
        dg.bv += key_ != dg.brick ? 1 : 0;                                                                             \
        dg.brick = key_;                                                                                               \
    } while (0)
#else
struct DgLane {};
#define DG_ADD(i, v)
#define DG_BRICK()
#endif
// -DVRT_ISA_MARK: comments in the assembly at the borders of the bodies (tools/isa_regions.py counts between them)
#ifdef VRT_ISA_MARK
#define VRT_MARK(name) asm volatile("; @@" name)
#else
#define VRT_MARK(name)
#endif

// (int)floor(x), (int)floor(y), (int)floor(z) for |x|, |y|, |z| < 2^31 in three VALU instructions instead of six: with the
// binary64 rounding mode set to round-down, x + 1.5 * 2^52 is exactly floor(x) + 1.5 * 2^52 (the sum's ulp is 1), whose
// low mantissa word is floor(x) in two's complement.  The mode bits (MODE[3:2]) are switched and restored inside one
// asm statement, so no other arithmetic can be scheduled under the changed mode.
__device__ __forceinline__ void floor3_i32(double x, double y, double z, int& fx, int& fy, int& fz) {
    double a, b, c;
    const double magic = 6755399441055744.0;
    asm volatile(
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
        "v_add_f64 %0, %3, %6\n\t"
        "v_add_f64 %1, %4, %6\n\t"
        "v_add_f64 %2, %5, %6\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
        : "=&v"(a), "=&v"(b), "=&v"(c)
        : "v"(x), "v"(y), "v"(z), "s"(magic));
    fx = __double2loint(a);
    fy = __double2loint(b);
    fz = __double2loint(c);
}

// min(a, b, c) and max(|a|, |b|, |c|) in two instructions each: the compiler's fmin / fmax first canonicalise every
// operand (v_max_f64 x, x -- a signalling-NaN matter; positions and velocities are never NaN)
__device__ __forceinline__ double min3_f64(double a, double b, double c) {
    double t;
    asm("v_min_f64 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
    asm("v_min_f64 %0, %1, %2" : "=v"(t) : "v"(t), "v"(c));
    return t;
}
__device__ __forceinline__ double absmax3_f64(double a, double b, double c) {
    double t;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(t) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, |%2|" : "=v"(t) : "v"(t), "v"(c));
    return t;
}

// The same with a per-lane magic constant: 1.5 * 2^52 gives floor(x); 1.5 * 2^53 (sum's ulp 2) gives floor(x / 2), the
// coordinate a resolution-2 chunk snaps to, halved.
__device__ __forceinline__ void floor3_i32_lane(double x, double y, double z, double magic, int& fx, int& fy, int& fz) {
    double a, b, c;
    asm volatile(
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
        "v_add_f64 %0, %3, %6\n\t"
        "v_add_f64 %1, %4, %6\n\t"
        "v_add_f64 %2, %5, %6\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
        : "=&v"(a), "=&v"(b), "=&v"(c)
        : "v"(x), "v"(y), "v"(z), "v"(magic));
    fx = __double2loint(a);
    fy = __double2loint(b);
    fz = __double2loint(c);
}

// per-axis parts of vrt_voxel_offset, read with a byte index (local coordinate * 4, masked): tab + axis * 1024
// (MarchCtx::tl: where each axis' table begins -- three arrays of 256 words, or, in the kernels that hold world-axis tables,
// the entries of the world's first chunk in those: its share of the block number is 0)
__device__ __forceinline__ uint32_t tab_at(const MarchCtx& C, int axis, unsigned byte_index) {
    return *reinterpret_cast<const lds_u32*>(C.tab + C.tl[axis] + (int)byte_index);
}

// Frame.get_voxel(floor(pos)) (data.py:136-145) on the packed chunk block looks at cell (fp // res) * res, which only
// exists inside the chunk's own half-open box.  cell_offset returns the byte offset of that cell in the voxel buffer
// (`base`: the chunk block's offset) for l4 = 4 * (floor(pos) - chunk_min), or ~0 when the cell lies outside the block.  RESMODE 0: every chunk has
// resolution 1; 1: resolutions 1 and 2 (chunk_min is even, so (f & ~1) - chunk_min == (f - chunk_min) & ~1: the snap is
// the table-index mask m4, and a position inside the block stays inside); 2: any resolution.
// `inside`: the position must be looked up and lies strictly inside the block; `valid`: it must be looked up (only
// differs from `inside` for the ray's own position on the block's upper faces, whose cell a resolution >= 3 can snap
// back into the block).
template <int RESMODE>
__device__ __forceinline__ unsigned cell_offset(const MarchCtx& C, uint32_t entry, unsigned base, unsigned m4, unsigned cs4, int nm4x,
                                                int nm4y, int nm4z, int l4x, int l4y, int l4z, bool inside, bool valid) {
    if (RESMODE == 2 && entry >= (3u << 24)) {  // rare: the reference's floor division (nm4 = -4 * chunk_min)
        const int3 o = snap_generic3((int)(entry >> 24), -(nm4x >> 2), -(nm4y >> 2), -(nm4z >> 2), l4x >> 2, l4y >> 2, l4z >> 2);
        l4x = o.x << 2;
        l4y = o.y << 2;
        l4z = o.z << 2;
        inside = valid && (unsigned)(l4x | l4y | l4z) < cs4;
        m4 = 0x3fcu;
    }
    // RESMODE 0 needs no snap mask, and no range mask either: a local coordinate outside the block reads LDS beyond the
    // table (or beyond the allocation, which returns 0), and the value is discarded by `inside`
    if (RESMODE == 0) m4 = ~0u;
    const unsigned t = tab_at(C, 0, (unsigned)l4x & m4) | tab_at(C, 1, (unsigned)l4y & m4) | tab_at(C, 2, (unsigned)l4z & m4);
    return inside ? base + t : ~0u;
}

// lib.material_background + tile()'s alpha (lib.py:463-476, init.py:141) for a finished ray; `energy` leaves as the
// ray's final energy
__device__ __forceinline__ uint32_t finish_color(const PowCache& pc, bool has_background, double pow_y, double shutter, uint32_t color,
                                                 double& energy, double bounces, double vy) {
    int cr = (int)(color & 255u), cg = (int)((color >> 8) & 255u), cb = (int)((color >> 16) & 255u);
    if (has_background) {
        double a = 1 / pow_cached(pc, 1 + bounces, pow_y);
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
    // init.py:141
    double e = energy + shutter;
    if (!(e < 1)) e = 1;
    const int alpha = (int)__builtin_rint(e * 255);
    return (uint32_t)cr | ((uint32_t)cg << 8) | ((uint32_t)cb << 16) | ((uint32_t)alpha << 24);
}

// The kernel arguments as a body sees them: re-read from the kernel-argument segment (scalar loads) where the body
// needs them, instead of held in scalar registers for the whole kernel.  A march kernel keeps ~100 wave-uniform values
// (arguments, LDS offsets, masks); what does not fit the scalar file is spilled to VGPR lanes, and every later use is a
// v_readlane -- a VALU instruction, in a kernel that is bound by VALU issue.  The empty asm makes the pointer opaque, so
// loads through it cannot be hoisted out of the body; the address space keeps them scalar (s_load_dword).
typedef const __attribute__((address_space(4))) MarchParams* kernarg_ptr;
__device__ __forceinline__ const __attribute__((address_space(4))) MarchParams& fresh_args(const MarchParams&) {
    // (MarchParams is the kernels' only argument: it starts the kernel-argument segment.  Taking the argument's own
    // address instead would make the compiler copy all of it to scratch memory first.)
    auto p = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *p;
}

// ---- the bodies of the march: what one lane does in each state.  march_kernel and march_pool_kernel differ only in
// ---- how they decide which body a wave runs next and with which rays in its lanes.

// lookup variants 1 and 2: the occupancy word / brick a lane holds
struct LkState {
    uint32_t okey;         // LK 1: index of the occupancy word the lane holds; LK 2: of the staged brick
    uint64_t oword;        // LK 1: its bits
    uint64_t* brick_slot;  // LK 2: the lane's brick slot in LDS, 8 words + 1 of padding (9 x 8 bytes: lanes fall into different banks)
};
// RECORD: the ray's own traversed list, to report its length (init.py:72-73)
template <bool RECORD>
struct SeenList {
    int64_t seen[RECORD ? 48 : 1];
    int n;
};

// A traversed visit whose key comparison has been put off (resnap_commit<.., DEFER>): the cell, the visit's key and the
// cell's key as asked for -- not yet waited for -- when the re-snap was made
struct PendingVisit {
    int tci;        // < 0: nothing pending
    int bi;         // the cell's bit in the settled bitmap, < 0: it has none
    uint64_t tkey, tcur;
};

// Camera.trace's re-snap (init.py:68-73) for a ray whose position has the floor (fx, fy, fz): the chunk cursor moves to the
// chunk of that cell -- chunk_min = snapped(), chunk = chunks.get() -- and the visit joins `traversed`.  (The caller has
// decided that the position lies outside the current chunk's inclusive box.)
// TDQ: the traversed box's dimensions come from the arguments as the caller sees them (re-read ones), not from MarchCtx::td
// DEFER (with `pend`): when the traversed box has no settled bitmap -- every visit reads its cell's key from memory -- the
// key is asked for here but compared by the caller (resnap_finish), after it has issued its voxel reads: the step then
// waits for one round trip, not for two in a row.  (With a bitmap nearly every visit skips the read, and the earlier a
// visit lowers its cell's key the sooner the cell settles for everyone: those are not put off.)
// (DEFER 2: such a launch may have a bitmap over the cells around the camera -- march_pool_kernel's instances; 1: never)
template <bool RECORD, bool TDQ = false, int DEFER = 0, class PT>
__device__ __forceinline__ void resnap_commit(const PT& Q, const MarchCtx& C, Ray& r, int fx, int fy, int fz, uint64_t wmin_key,
                                              SeenList<RECORD>& sl, int fl, PendingVisit* pend = nullptr) {
    (void)sl; (void)pend;
    // snapped(): (v // cs) * cs; floor(p / cs) == floor(p) >> shift: the chunk's coordinates in chunks
    const int ccx = fx >> Q.cs_shift, ccy = fy >> Q.cs_shift, ccz = fz >> Q.cs_shift;
    r.nm4x = -(ccx << (Q.cs_shift + 2));
    r.nm4y = -(ccy << (Q.cs_shift + 2));
    r.nm4z = -(ccz << (Q.cs_shift + 2));
    // the chunk's table entry and the traversed cell's current key are fetched together (two
    // independent reads, one round trip), then used
    const uint64_t tkey = ((uint64_t)(Q.ray0 + r.off) << 12) | (uint64_t)(r.resnaps < 4095 ? r.resnaps : 4095);
    // -1: not recorded, -2: outside the box
    const bool has_keys = (fl & CF_HAS_KEYS) != 0, has_bm = (fl & CF_HAS_BM) != 0;
    const int tci = TDQ ? trav_cell(has_keys, ccx, ccy, ccz, Q.t_origin_c[0], Q.t_origin_c[1], Q.t_origin_c[2], Q.t_dims[0], Q.t_dims[1], Q.t_dims[2])
                        : trav_cell(has_keys, ccx, ccy, ccz, C.toc[0], C.toc[1], C.toc[2], C.td[0], C.td[1], C.td[2]);
    // (the table entry is asked for before the settled bit is looked at: both LDS reads travel together)
    r.entry = TDQ ? chunk_entry_i(Q, C.ct, ccx - Q.origin_c[0], ccy - Q.origin_c[1], ccz - Q.origin_c[2], Q.dims[0], Q.dims[1], Q.dims[2],
                                  (fl & CF_CT_LDS) != 0, (fl & CF_CT_IDENTITY) != 0)
                  : chunk_entry_i(Q, C.ct, ccx - C.oc[0], ccy - C.oc[1], ccz - C.oc[2], C.dm[0], C.dm[1], C.dm[2], (fl & CF_CT_LDS) != 0,
                                  (fl & CF_CT_IDENTITY) != 0);
    // the cell's bit in the settled bitmap: the cell's index -- or, for a box too large for a bitmap of its own, its index
    // inside the 32^3 cells around the camera that have one (cells outside that window never count as settled)
    int bi = tci;
    bool in_bm = has_bm;
    if (DEFER != 1 && (fl & CF_BM_WINDOW) != 0) {
        const int w = Q.bm_window;
        const int wx = ccx - (TDQ ? Q.t_origin_c[0] : C.toc[0]) - (w & 1023), wy = ccy - (TDQ ? Q.t_origin_c[1] : C.toc[1]) - ((w >> 10) & 1023),
                  wz = ccz - (TDQ ? Q.t_origin_c[2] : C.toc[2]) - (w >> 20);
        in_bm = (unsigned)(wx | wy | wz) < 32u;
        bi = in_bm ? (wx << 10) | (wy << 5) | wz : 0;
    }
    const bool settled = tci >= 0 && in_bm && ((C.bm[bi >> 5] >> (bi & 31)) & 1u);
    uint64_t tcur = 0;
    if (tci >= 0 && !settled) tcur = Q.t_keys[tci];
#ifdef VRT_COUNT_UNSETTLED  // (measurement build: key reads into stats[14], visits outside the bitmap's window into stats[15])
    if (tci >= 0 && !settled) atomicAdd((unsigned long long*)&Q.stats[14], 1ull);
    if (tci >= 0 && !in_bm) atomicAdd((unsigned long long*)&Q.stats[15], 1ull);
#endif
    r.boff = ((r.entry & 0xffffffu) - 1u) << (3 * Q.cs_shift);
    if (DEFER) {  // (such launches have no bitmap, or one over the cells around the camera only: launch_march)
        pend->tci = settled ? -1 : tci;  // (-1 / -2 included: resnap_finish only acts on cells)
        pend->bi = in_bm ? bi : -1;
        pend->tkey = tkey;
        pend->tcur = tcur;
        if (tci == -2) atomicAdd((unsigned long long*)&Q.stats[VRT_S_TRAV_OUTSIDE], 1ull);
    } else if (tci >= 0 && !settled) {
        if (tkey < tcur) atomicMin((unsigned long long*)&Q.t_keys[tci], (unsigned long long)tkey);
        if (in_bm && tcur < wmin_key)
            __hip_atomic_fetch_or(&C.bm[bi >> 5], 1u << (bi & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (tci == -2) {
        atomicAdd((unsigned long long*)&Q.stats[VRT_S_TRAV_OUTSIDE], 1ull);
    }
    r.resnaps++;
    if (RECORD) {
        int64_t cid = ((int64_t)ccx * 2097152 + (int64_t)ccy) * 2097152 + (int64_t)ccz;
        bool dup = false;
        for (int k = 0; k < sl.n && k < 48; k++) dup |= (sl.seen[k] == cid);
        if (!dup) {
            if (sl.n < 48) sl.seen[sl.n] = cid;
            sl.n++;
        }
    }
}

// `after`: a value that only exists once the caller's voxel reads are back.  The key passes through an empty statement that
// names it, so the comparison (and the wait for the key) cannot be scheduled ahead of those reads
template <bool BM, class PT>
__device__ __forceinline__ void resnap_finish(const PT& Q, const MarchCtx& C, const PendingVisit& pend, unsigned after, uint64_t wmin_key) {
    unsigned lo = (unsigned)pend.tcur, hi = (unsigned)(pend.tcur >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi) : "v"(after));
    const uint64_t tcur = ((uint64_t)hi << 32) | lo;
    if (pend.tci >= 0) {
        if (pend.tkey < tcur) atomicMin((unsigned long long*)&Q.t_keys[pend.tci], (unsigned long long)pend.tkey);
        if (BM && pend.bi >= 0 && tcur < wmin_key)  // (see resnap_commit: the cell is settled for this workgroup)
            __hip_atomic_fetch_or(&C.bm[pend.bi >> 5], 1u << (pend.bi & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// IDLE -> MARCH: the lane takes ray k of the launch (init.py:41-59 with the lens quaternion and the life from the ray
// table).  False for an unused sample slot of the tile.
// PERPIX: the ray table holds one record per pixel (ray_table_per_pixel) -- 0 no, 1 yes, 2 ask P.per_pixel at run time
// (the record-keeping and re-trace kernels).  The frame kernels are compiled for either layout: a run-time branch
// around the record load cost config 3 2-6 % although it never took the other arm.
// PERPIX 4: like 2, for launches that may also have no ray table (the record-keeping and re-trace kernels).
// PERPIX 3 (P.per_pixel == 2): no ray table at all -- the frame's draw rows were seeded just now and are used once, so
// the lane works out what raygen_tile_kernel would have written (ray_record_from_draws) instead of that kernel writing
// 64 bytes per ray slot for this one to read back.
// SNAP: the re-snap of the ray's first iteration (init.py:66-73) is made here, where nearly every lane of the wave takes a
// ray, instead of by the few lanes of a march step that hold fresh rays (march_step_w's one-at-a-time path).
// SEED (re-trace launches): the draw row of a listed ray is seeded by the lane that takes the ray, here, instead of by a
// launch of its own before the march (nobody else reads that row): 1 = the D_SLOW_DEV draws that need no state twist
// (registers only), 2 = D_FULL_DEV draws from a full-state generator (a 624-word private array).
template <bool RECORD, bool LIST, int PERPIX, bool SNAP = false, int SEED = 0>
__device__ __forceinline__ bool take_ray(const MarchParams& P, const MarchCtx& C, int64_t k, Ray& r, DgLane& dg, uint64_t wmin_key,
                                         SeenList<RECORD>& sl) {
    (void)dg; (void)wmin_key; (void)sl;
    const auto& Q = fresh_args(P);  // (see fresh_args)
    const int64_t off = LIST ? (int64_t)Q.list[k] : k;
    const int64_t ray = Q.ray0 + off;
    const int64_t rowi = LIST ? k : ((C.tile && Q.ray_seedidx) ? (int64_t)Q.ray_seedidx[ray] : ray);
    if (LIST && SEED != 0) {
        const int64_t p = ray / Q.g.smax;
        const uint64_t seed = ray_seed(Q.st, Q.g.smax, Q.g.pixels[2 * p], Q.g.pixels[2 * p + 1], (int)(ray - p * Q.g.smax));
        double* row = const_cast<double*>(Q.draws) + rowi * Q.draw_stride;
        if (SEED == 1) mt_seed_draws<true>(seed, D_SLOW_DEV, row);
        else mt_full_draws(seed, D_FULL_DEV, row);
        __threadfence();  // (the row is read back below and by the ray's hits)
    }
    double life, ox, oy, oz, ow, t0, t1, t2;
    if (PERPIX == 3 || (PERPIX == 4 && Q.per_pixel == 2)) {
        const uint32_t px = (uint32_t)ray / (uint32_t)Q.g.smax;
        const int2 xy = reinterpret_cast<const int2*>(Q.g.pixels)[px];
        RayRecord rec;
        ray_record_from_draws(Q.st, Q.lens, xy.x, xy.y, (int)((uint32_t)ray - px * (uint32_t)Q.g.smax),
                              Q.draws + rowi * Q.draw_stride, rec);
        life = rec.life; ox = rec.ox; oy = rec.oy; oz = rec.oz; ow = rec.ow;
        t0 = rec.d0; t1 = rec.d1; t2 = rec.d2;
    } else if (PERPIX != 0 && (PERPIX == 1 || Q.per_pixel)) {  // (2, 4: asked at run time)
        // the pixel's record (d0 = its sample count); the ray's first-hit draws come from the draw table
        const uint32_t px = (uint32_t)ray / (uint32_t)Q.g.smax;
        const RayRecord rec = Q.tab.rec[px];
        ox = rec.ox; oy = rec.oy; oz = rec.oz; ow = rec.ow;
        life = (int)((uint32_t)ray - px * (uint32_t)Q.g.smax) < (int)rec.d0 ? rec.life : -1.0;
        t0 = t1 = t2 = 0.5;
        if (life >= 0.0) {  // (an unused sample slot has no draw row: its index in the plan is 0xFFFFFFFF)
            const double* row = Q.draws + rowi * Q.draw_stride + Q.first_draw;
            t0 = row[0];
            t1 = row[1];
            t2 = row[2];
        }
    } else {
        // the whole record is fetched at once (one memory round trip), then inspected
        const RayRecord rec = Q.tab.rec[ray];
        life = rec.life; ox = rec.ox; oy = rec.oy; oz = rec.oz; ow = rec.ow;
        t0 = rec.d0; t1 = rec.d1; t2 = rec.d2;
    }
    if (life < 0.0) {  // unused sample slot of the tile
        if (Q.ray_rgba) Q.ray_rgba[ray] = 0;
        if (RECORD && Q.rays) Q.rays[ray].s = -1;
        return false;
    }
    r.off = (uint32_t)off;
    // (the camera's eight scalars are read from LDS in one go: read where they are used, each of the last four waited for
    // its own round trip)
    const double cq0 = COLD(COLD_ROT), cq1 = COLD(COLD_ROT + 1), cq2 = COLD(COLD_ROT + 2), cq3 = COLD(COLD_ROT + 3);
    const double dist_min = COLD(COLD_DIST_MIN), cpx = COLD(COLD_POS), cpy = COLD(COLD_POS + 1), cpz = COLD(COLD_POS + 2);
    camera_forward(cq0, cq1, cq2, cq3, ox, oy, oz, ow, r.vx, r.vy, r.vz);  // init.py:44-45
    r.life = life;
    // init.py:50-59
    r.px = cpx + r.vx * dist_min;
    r.py = cpy + r.vy * dist_min;
    r.pz = cpz + r.vz * dist_min;
    r.step = 0;
    r.bounces = 0;
    r.energy = 0;
    r.color = 0;
#ifdef VRT_DIAG
    dg.brick = ~0u;
    DG_BRICK();  // the brick the ray starts in
#endif
    // chunk_min = chunk_max = vec3(0, 0, 0), chunk = None (init.py:46-47): the sentinel -4 * chunk_min = 2^31
    // makes the in-chunk test fail until the first snap (the march handles p == (0, 0, 0) itself)
    r.nm4x = r.nm4y = r.nm4z = (int)0x80000000u;
    r.entry = 0;
    r.boff = 0;
    r.resnaps = 0;
    r.ndraw = Q.first_draw;
    r.rowi = (uint32_t)rowi;
    r.d0 = t0;
    r.d1 = t1;
    r.d2 = t2;
    if (SNAP) {
        // before the first iteration chunk_min == chunk_max == (0, 0, 0) (init.py:46): a ray whose loop runs at all
        // (0 < life, init.py:66) snaps there unless it starts at the origin exactly (march_step keeps that case)
        const bool at_origin = r.px == 0.0 && r.py == 0.0 && r.pz == 0.0;
        if (0.0 < life && !at_origin) {
            int fx, fy, fz;
            floor3_i32(r.px, r.py, r.pz, fx, fy, fz);
            resnap_commit<RECORD, true>(Q, C, r, fx, fy, fz, wmin_key, sl, Q.snap_flags);
        }
    }
    return true;
}

// MARCH: phase A of the reference loop (init.py:66-77, 114-116) for a lane in LANE_MARCH -- loop condition, chunk
// re-snap, the voxel of this position and speculatively of the next SPEC - 1, advance.  Leaves the lane in LANE_MARCH,
// LANE_HIT (a voxel was found: its material sits in the colour word's top byte) or LANE_ENDED.
#ifndef VRT_FRESH_MARCH
#define VRT_FRESH_MARCH 0
#endif
// ... in march_pool_kernel, whose scheduler keeps many more wave-uniform values alive across the step: with the step's
// arguments re-read, 6 scalar registers are spilled to vector lanes instead of 16, at the same speed (config 3 5.244 against
// 5.241 ms over four alternating runs, config 5 231.4 against 231.9 ms); march_kernel is 2.5 % slower that way (config 2)
#ifndef VRT_FRESH_MARCH_POOL
#define VRT_FRESH_MARCH_POOL 1
#endif
#ifndef VRT_PASS_COUNTS
#define VRT_PASS_COUNTS 0
#endif
// the march step's view of the kernel arguments: held in scalar registers (the step runs every pass and needs them at
// once), or re-read like the slow bodies do (march_step_w's one-at-a-time path; -DVRT_FRESH_MARCH=1)
template <bool FRESH>
__device__ __forceinline__ decltype(auto) march_args(const MarchParams& P) {
    if constexpr (FRESH) return fresh_args(P);
    else return (P);
}
// DEFER: the launch's traversed box has no settled bitmap -- the key reads of its re-snaps are compared after the voxel
// reads have been issued (resnap_commit<DEFER>)
template <int SPEC, int RESMODE, bool RECORD, int LK, bool FRESH = VRT_FRESH_MARCH, int DEFER = 0>
__device__ __forceinline__ void march_step(const MarchParams& P, const MarchCtx& C, Ray& r, int& state, int32_t (&cnt)[C_NLOCAL],
                                           uint64_t wmin_key, LkState& lk, SeenList<RECORD>& sl, DgLane& dg) {
    (void)lk; (void)sl; (void)dg;
    const auto& Q = march_args<FRESH>(P);
    const int fl = fresh_args(P).snap_flags;  // (see CF_*: asked for here, used by the re-snap)
    const auto& st = Q.st;
    const unsigned cs4 = C.cs4;
    if (!(r.step < r.life)) {  // init.py:66: the ray's life ran out
        state = LANE_ENDED;
        return;
    }
    PendingVisit pend;
    pend.tci = pend.bi = -1;
    pend.tkey = pend.tcur = 0;
    int fx, fy, fz;
    floor3_i32(r.px, r.py, r.pz, fx, fy, fz);
    // 4 * (floor(pos) - chunk_min), in wrap-around arithmetic (|floor(pos)|, |chunk_min| < 2^28)
    int l4x = (int)(((unsigned)fx << 2) + (unsigned)r.nm4x), l4y = (int)(((unsigned)fy << 2) + (unsigned)r.nm4y),
        l4z = (int)(((unsigned)fz << 2) + (unsigned)r.nm4z);
    // strictly inside the half-open chunk box => inside the reference's inclusive box (init.py:67)
    const unsigned l4or = (unsigned)(l4x | l4y | l4z);
    bool inside = l4or < cs4;
#ifdef VRT_DIAG
    if (__ballot(!inside)) { DG_ADD(DG_SNAP_ITERS, 1); DG_ADD(DG_SNAP_LANES, __popcll(__ballot(!inside))); }
#endif
    if (!inside) {
        VRT_MARK("m_resnap");
        // the reference's inclusive box test (init.py:67) in integers: chunk_min <= p <= chunk_min + cs on
        // an axis <=> floor(p) - chunk_min in [0, cs), or == cs with p itself an integer
        const unsigned ux = (unsigned)l4x, uy = (unsigned)l4y, uz = (unsigned)l4z;
        const unsigned umax = ux > uy ? (ux > uz ? ux : uz) : (uy > uz ? uy : uz);
        bool outside = (umax > cs4) | ((ux == cs4) & (r.px != (double)fx)) | ((uy == cs4) & (r.py != (double)fy)) |
                       ((uz == cs4) & (r.pz != (double)fz));
        // before the first snap chunk_min == chunk_max == (0, 0, 0) (init.py:46) and nm4 is the sentinel 2^31,
        // which the test above calls outside; the reference's answer differs for p == (0, 0, 0) only, and
        // l4or == 2^31 exactly then (4 * |floor(p) - chunk_min| < 2^31 after a snap)
        // (a wave-level branch: the compiler otherwise evaluates the three compares in every iteration)
        if (__ballot(l4or == 0x80000000u) != 0ull) {
            if (l4or == 0x80000000u) outside = !(r.px == 0.0 && r.py == 0.0 && r.pz == 0.0);
        }
        if (outside) {
            resnap_commit<RECORD, FRESH, DEFER>(Q, C, r, fx, fy, fz, wmin_key, sl, fl, &pend);
            l4x = (int)(((unsigned)fx << 2) + (unsigned)r.nm4x);
            l4y = (int)(((unsigned)fy << 2) + (unsigned)r.nm4y);
            l4z = (int)(((unsigned)fz << 2) + (unsigned)r.nm4z);
            inside = true;
        }
    }
    VRT_MARK("m_entry");
    if (r.entry) {  // init.py:75-77
        VRT_MARK("m_spec");
        // SPEC reference iterations per pass: the voxel of this position and, speculatively, of the
        // next ones (pos + vel * step added repeatedly, the values the reference computes at init.py:116)
        // are fetched together.  A speculative step is only taken when the reference would take it
        // unchanged: loop condition true (init.py:66), still strictly inside the same chunk (no re-snap
        // at init.py:67), every earlier voxel empty.  vel * step is the same rounded product in every
        // one of these iterations.
        const unsigned res = r.entry >> 24;
        // Frame.resolution (init.py:114); a zero must not stall the march.  RESMODE 0: every chunk has
        // resolution 1, and v * 1.0 == v
        const double sd = RESMODE == 0 ? 1.0 : (double)(res ? res : 1u);
        const bool res2 = RESMODE != 0 && res == 2u;
        const unsigned m4 = res2 ? 0x3f8u : 0x3fcu;
        // the speculative positions of a resolution-2 chunk are floored to even coordinates directly
        // (floor3_i32_lane), which is that chunk's snap (chunk_min is even): no mask
        const double magic_r = __hiloint2double(res2 ? 0x43480000 : 0x43380000, 0);
        const unsigned sh_r = res2 ? 3u : 2u;
        const double dvx = RESMODE == 0 ? r.vx : r.vx * sd, dvy = RESMODE == 0 ? r.vy : r.vy * sd,
                     dvz = RESMODE == 0 ? r.vz : r.vz * sd;
        unsigned o[SPEC];  // voxel-buffer offset of each position's cell (block offset included), ~0 = nothing to read
        // RESMODE 0, 1 (VRT_TAB_PIPELINE): the three table words of a position are read from LDS while the next position is
        // being computed, and combined after that one's reads have been issued -- the rounding-mode statements of the floors
        // are barriers to the compiler's scheduler, which otherwise waits for every position's words right where it asked for them
        constexpr bool PIPE = VRT_TAB_PIPELINE && RESMODE != 2;
        uint32_t pt0 = 0, pt1 = 0, pt2 = 0;
        bool pin = inside;
        if (PIPE) {
            const unsigned m0 = RESMODE == 0 ? ~0u : m4;
            pt0 = tab_at(C, 0, (unsigned)l4x & m0);
            pt1 = tab_at(C, 1, (unsigned)l4y & m0);
            pt2 = tab_at(C, 2, (unsigned)l4z & m0);
        } else {
            o[0] = cell_offset<RESMODE>(C, r.entry, r.boff, m4, cs4, r.nm4x, r.nm4y, r.nm4z, l4x, l4y, l4z, inside, true);
        }
        int n_valid = 1;  // positions whose voxel the reference would look up, if all before are empty
        // position CKPT_AT of the sequence is kept (where registers allow): the advance below then starts
        // from it when the ray gets that far, and re-adds at most SPEC - CKPT_AT steps instead of SPEC
        constexpr bool CKPT = SPEC == 8 && RESMODE != 2 && LK == 0;
        constexpr int CKPT_AT = 4;
        double cqx = 0, cqy = 0, cqz = 0, cqs = 0;
        {
            double qx = r.px, qy = r.py, qz = r.pz, qs = r.step;
            bool ok = true;
#pragma unroll
            for (int k = 1; k < SPEC; k++) {
                qx += dvx;
                qy += dvy;
                qz += dvz;
                qs += sd;
                int gx, gy, gz, kx, ky, kz;
                if (RESMODE == 0) {
                    floor3_i32(qx, qy, qz, gx, gy, gz);
                    kx = (int)(((unsigned)gx << 2) + (unsigned)r.nm4x);
                    ky = (int)(((unsigned)gy << 2) + (unsigned)r.nm4y);
                    kz = (int)(((unsigned)gz << 2) + (unsigned)r.nm4z);
                } else {
                    floor3_i32_lane(qx, qy, qz, magic_r, gx, gy, gz);
                    kx = (int)(((unsigned)gx << sh_r) + (unsigned)r.nm4x);
                    ky = (int)(((unsigned)gy << sh_r) + (unsigned)r.nm4y);
                    kz = (int)(((unsigned)gz << sh_r) + (unsigned)r.nm4z);
                }
                ok = ok && (qs < r.life) && ((unsigned)(kx | ky | kz) < cs4);
                n_valid += ok ? 1 : 0;
                if (CKPT && k == CKPT_AT) {
                    cqx = qx;
                    cqy = qy;
                    cqz = qz;
                    cqs = qs;
                }
                if (PIPE) {
                    const uint32_t t0 = tab_at(C, 0, (unsigned)kx), t1 = tab_at(C, 1, (unsigned)ky),
                                   t2 = tab_at(C, 2, (unsigned)kz);
                    o[k - 1] = pin ? r.boff + (pt0 | pt1 | pt2) : ~0u;
                    pt0 = t0;
                    pt1 = t1;
                    pt2 = t2;
                    pin = ok;
                } else {
                    o[k] = cell_offset<RESMODE>(C, r.entry, r.boff, ~0u, cs4, r.nm4x, r.nm4y, r.nm4z, kx, ky, kz, ok, ok);
                }
            }
            if (PIPE) o[SPEC - 1] = pin ? r.boff + (pt0 | pt1 | pt2) : ~0u;
        }
        bool found;
        int h = n_valid;  // advances made before the hit (or all of them, and no hit)
        if (LK == 0) {
            VRT_MARK("m_load");
            unsigned ids[SPEC];
#pragma unroll
            for (int k = 0; k < SPEC; k++)
                ids[k] = __builtin_amdgcn_raw_buffer_load_b8(C.vox, o[k], 0, 0);  // (~0 is out of range: reads 0)
            // first occupied voxel among the positions (a position that was not read is 0): the bytes are
            // packed four to a word
            unsigned w[(SPEC + 3) / 4];
#pragma unroll
            for (int g = 0; g < (SPEC + 3) / 4; g++) {
                w[g] = 0;
#pragma unroll
                for (int k = 4 * g; k < SPEC && k < 4 * g + 4; k++) w[g] |= ids[k] << (8 * (k - 4 * g));
            }
            unsigned wsel = w[(SPEC + 3) / 4 - 1];
            int wbase = 4 * ((SPEC + 3) / 4 - 1);
#pragma unroll
            for (int g = (SPEC + 3) / 4 - 2; g >= 0; g--) {
                wbase = w[g] ? 4 * g : wbase;
                wsel = w[g] ? w[g] : wsel;
            }
            found = wsel != 0u;
            if (found) {
                const int byte = (__ffs(wsel) - 1) >> 3;
                h = wbase + byte;
                r.color |= ((wsel >> (byte << 3)) & 255u) << 24;
            }
        } else {
            unsigned hitmask = 0;
            if (LK == 1) {
                // occupancy words: one load per NEW micro-brick along the positions, all in flight together
                uint32_t key[SPEC];
                bool need[SPEC];
                uint32_t pk = lk.okey;
#pragma unroll
                for (int k = 0; k < SPEC; k++) {
                    key[k] = o[k] != ~0u ? o[k] >> 6 : pk;
                    need[k] = key[k] != pk;
                    pk = key[k];
                }
                lk.okey = pk;
                uint64_t w[SPEC];
#pragma unroll
                for (int k = 0; k < SPEC; k++) {
                    w[k] = 0;
                    if (need[k]) w[k] = Q.occ[key[k]];
                }
                // (opaque to the compiler: without this it folds each load into the select chain below
                // and waits for load k before it issues load k + 1)
#pragma unroll
                for (int k = 0; k < SPEC; k++) asm volatile("" : "+v"(w[k]));
                uint64_t cw = lk.oword;
#pragma unroll
                for (int k = 0; k < SPEC; k++) {
                    cw = need[k] ? w[k] : cw;
                    const unsigned bit = (unsigned)(cw >> (o[k] & 63u)) & 1u;
                    hitmask |= (o[k] != ~0u ? bit : 0u) << k;
                }
                lk.oword = cw;
            } else {
                // the 8^3 brick of occupancy bits (8 words = one 64-byte line) of each position is staged
                // in the lane's LDS slot when it is not the one already there, then its bit is read
#pragma unroll
                for (int k = 0; k < SPEC; k++) {
                    if (o[k] != ~0u) {
                        const uint32_t widx = o[k] >> 6;
                        if ((widx >> 3) != lk.okey) {
                            lk.okey = widx >> 3;
                            const ulonglong2* src = reinterpret_cast<const ulonglong2*>(Q.occ + ((size_t)lk.okey << 3));
                            const ulonglong2 a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
                            uint64_t* bs = lk.brick_slot;
                            bs[0] = a0.x; bs[1] = a0.y; bs[2] = a1.x; bs[3] = a1.y;
                            bs[4] = a2.x; bs[5] = a2.y; bs[6] = a3.x; bs[7] = a3.y;
                        }
                        const uint64_t cw = lk.brick_slot[widx & 7u];
                        hitmask |= ((unsigned)(cw >> (o[k] & 63u)) & 1u) << k;
                    }
                }
            }
            found = hitmask != 0u;
            if (found) {
                h = __ffs(hitmask) - 1;
                unsigned hoff = o[0];
#pragma unroll
                for (int k = 1; k < SPEC; k++) hoff = (h == k) ? o[k] : hoff;
                r.color |= (unsigned)__builtin_amdgcn_raw_buffer_load_b8(C.vox, hoff, 0, 0) << 24;
            }
        }
        VRT_MARK("m_adv");
        if (DEFER) resnap_finish<DEFER == 2>(Q, C, pend, (unsigned)h, wmin_key);  // (the cell's key has come back with the voxels)
#ifdef VRT_DIAG_HIST
#pragma unroll
        for (int k = 1; k <= SPEC && k <= 8; k++) {
            DG_ADD(DG_NV1 + k - 1, __popcll(__ballot(n_valid >= k)));
            DG_ADD(DG_H1 + k - 1, __popcll(__ballot(h >= k)));
        }
#endif
        cnt[C_LOOKUP] += h + (found ? 1 : 0);
        cnt[C_ADV] += h;
        int rem = h;  // advances still to add
        if (CKPT) {
#ifndef VRT_DIAG
            if (h >= CKPT_AT) {  // (the values the sequential adds produce: they are those adds)
                r.px = cqx;
                r.py = cqy;
                r.pz = cqz;
                r.step = cqs;
                rem = h - CKPT_AT;
            }
#endif
        }
#pragma unroll
        for (int k = 0; k < SPEC; k++) {
#ifndef VRT_DIAG
            if (CKPT && k >= (CKPT_AT - 1 > SPEC - CKPT_AT ? CKPT_AT - 1 : SPEC - CKPT_AT)) break;  // rem is at most that
#endif
            if (k < rem) {
                r.step += sd;
                r.px += dvx;
                r.py += dvy;
                r.pz += dvz;
                DG_BRICK();
            }
        }
        if (found) state = LANE_HIT;
    } else {  // void skip (init.py:114)
        VRT_MARK("m_void");
        if (DEFER) resnap_finish<DEFER == 2>(Q, C, pend, 0u, wmin_key);
#ifdef VRT_DIAG
        DG_ADD(DG_VOID_LANES, __popcll(__ballot(1)));
#endif
        const double mn = min3_f64(r.px, r.py, r.pz);
        const double t = mn + (double)st.chunk_radius;
        const double inv_cs = C.kept_cs ? C.inv_cs : COLD(COLD_INV_CS), csd = C.kept_cs ? C.cs : COLD(COLD_CS);
        const double md = t - __builtin_floor(t * inv_cs) * csd;  // float % for a power-of-two divisor: exact
        const double stepsize = 1 + __builtin_fabs((double)st.chunk_radius - md);
        r.step += stepsize;
        r.px += r.vx * stepsize;
        r.py += r.vy * stepsize;
        r.pz += r.vz * stepsize;
        DG_BRICK();
        cnt[C_ADV]++;
    }
}

// MARCH for scenes whose voxel blocks lie in table order (P.wt_on): the look-ahead does not stop at chunk borders.
//
// march_step drops a speculative position as soon as it leaves the ray's chunk, because the position's voxel address
// needs the block of ITS chunk, and that comes from the chunk table.  With the blocks in table order the address of a
// world cell is X[x] + Y[y] + Z[z] (three LDS words, MarchCtx::wt) whatever chunk it lies in, so all eight positions are
// read regardless of borders, and the block number in an address says which positions lie in another chunk than their
// predecessor (bit (7 - k) of `xm`).  What the reference does at a border -- the re-snap of init.py:67-73: inclusive box
// test, chunk cursor, `traversed` -- is replayed after the hit search, for the first border among the iterations that
// really ran: the ray is advanced to that position (the same sequential sums), the reference's own test is made on
// it, and the look-ahead beyond it stands if the chunk entered is present at the resolution the positions were
// computed for.  Everything else ends the step AT a position whose iteration has not begun -- a second border, a
// missing chunk or another resolution behind the first, a position exactly on the current chunk's upper faces (inside the
// reference's inclusive box: no re-snap, and the cell read is none of the current chunk's) -- and the next step starts
// there: a step may always stop early, the voxels read beyond are simply discarded.
// Lanes that cannot look ahead like this run the reference's iterations one at a time (march_step<1>): no chunk
// (void skipping, a ray that has not snapped yet), or a step of more than VRT_WT_DVMAX cells on an axis (the tables carry
// VRT_WT_GUARD guard cells either side of the world: eight positions stay inside them).
template <int RESMODE>
__device__ __forceinline__ void march_step_w(const MarchParams& P, const MarchCtx& C, Ray& r, int& state, int32_t (&cnt)[C_NLOCAL],
                                             uint64_t wmin_key, LkState& lk, SeenList<false>& sl, DgLane& dg) {
    (void)dg;
    static_assert(RESMODE == 0 || RESMODE == 1, "resolutions 1 and 2");
    constexpr int SPEC = 8;
    if (!(r.step < r.life)) {  // init.py:66: the ray's life ran out
        state = LANE_ENDED;
        return;
    }
    const unsigned res = r.entry >> 24;
    const double sd = RESMODE == 0 ? 1.0 : (double)(res ? res : 1u);  // Frame.resolution (init.py:114)
    const bool res2 = RESMODE != 0 && res == 2u;
    const double dvx = RESMODE == 0 ? r.vx : r.vx * sd, dvy = RESMODE == 0 ? r.vy : r.vy * sd, dvz = RESMODE == 0 ? r.vz : r.vz * sd;
    if ((r.entry == 0u) | !(absmax3_f64(dvx, dvy, dvz) <= VRT_WT_DVMAX)) {
        march_step<1, RESMODE, false, 0, true>(P, C, r, state, cnt, wmin_key, lk, sl, dg);
        return;
    }
    VRT_MARK("w_spec");
    if (RESMODE != 0) {
        // The velocity itself is not needed until the step is over: it is taken back from the step vector there (the
        // resolution is 1 or 2 and |step| <= VRT_WT_DVMAX, so v * sd * (1 / sd) == v exactly), which frees its registers
        // for the look-ahead.
        const double inv = res2 ? 0.5 : 1.0;
        r.vx = dvx * inv;
        r.vy = dvy * inv;
        r.vz = dvz * inv;
    }
    const double magic_r = __hiloint2double(res2 ? 0x43480000 : 0x43380000, 0);  // (see floor3_i32_lane)
    const unsigned sh_r = res2 ? 3u : 2u;
    unsigned o[SPEC];  // voxel-buffer offset of each position's cell
    unsigned xm = 0;   // bit (7 - k): position k lies in another chunk than position k - 1 (position -1: the ray's chunk)
    int n_life = 1;    // leading positions whose iteration the loop condition allows (init.py:66); position 0 is one
    double cqx = 0, cqy = 0, cqz = 0, cqs = 0;  // position CKPT_AT of the sequence (see march_step)
    constexpr int CKPT_AT = 4;
    const unsigned cs3 = C.cs3;
    {
        double qx = r.px, qy = r.py, qz = r.pz, qs = r.step;
        uint32_t pt0, pt1, pt2;
        unsigned prev = r.boff;
        auto words = [&](double x, double y, double z, uint32_t& t0, uint32_t& t1, uint32_t& t2) {
            int gx, gy, gz;
            if (RESMODE == 0) floor3_i32(x, y, z, gx, gy, gz);
            else floor3_i32_lane(x, y, z, magic_r, gx, gy, gz);
            const unsigned sh = RESMODE == 0 ? 2u : sh_r;
            t0 = *reinterpret_cast<const lds_u32*>(C.wt + (int)(((unsigned)gx << sh) + (unsigned)C.wb[0]));
            t1 = *reinterpret_cast<const lds_u32*>(C.wt + (int)(((unsigned)gy << sh) + (unsigned)C.wb[1]));
            t2 = *reinterpret_cast<const lds_u32*>(C.wt + (int)(((unsigned)gz << sh) + (unsigned)C.wb[2]));
        };
        words(qx, qy, qz, pt0, pt1, pt2);
#pragma unroll
        for (int k = 1; k < SPEC; k++) {
            qx += dvx;
            qy += dvy;
            qz += dvz;
            qs += sd;
            n_life += (qs < r.life) ? 1 : 0;  // (the sums only grow: the positions that pass are the leading ones)
            if (k == CKPT_AT) {
                cqx = qx;
                cqy = qy;
                cqz = qz;
                cqs = qs;
            }
            // (the table words of a position are read while the next one is computed: see VRT_TAB_PIPELINE)
            uint32_t t0, t1, t2;
            words(qx, qy, qz, t0, t1, t2);
            o[k - 1] = pt0 + pt1 + pt2;
            xm = xm + xm + (((o[k - 1] ^ prev) >= cs3) ? 1u : 0u);
            prev = o[k - 1];
            pt0 = t0;
            pt1 = t1;
            pt2 = t2;
        }
        o[SPEC - 1] = pt0 + pt1 + pt2;
        xm = xm + xm + (((o[SPEC - 1] ^ prev) >= cs3) ? 1u : 0u);
    }
    VRT_MARK("w_load");
    unsigned ids[SPEC];
#pragma unroll
    for (int k = 0; k < SPEC; k++) ids[k] = __builtin_amdgcn_raw_buffer_load_b8(C.vox, o[k], 0, 0);  // (beyond the buffer: 0)
    const unsigned w0 = ids[0] | (ids[1] << 8) | (ids[2] << 16) | (ids[3] << 24);
    const unsigned w1 = ids[4] | (ids[5] << 8) | (ids[6] << 16) | (ids[7] << 24);
    const unsigned wsel = w0 ? w0 : w1;
    int h = SPEC;  // advances made before the hit (or all of them, and no hit)
    unsigned id = 0;
    if (wsel != 0u) {
        const int byte = (__ffs(wsel) - 1) >> 3;
        h = (w0 ? 0 : 4) + byte;
        id = (wsel >> (byte << 3)) & 255u;
    }
    bool found = h < n_life;  // (a voxel behind the end of the ray's life is never looked up)
    h = found ? h : n_life;
    VRT_MARK("w_cross");
    // chunk borders among the iterations that ran: positions 0 .. L (the hit's iteration began; the one after the last
    // advance did not).  Bit j of xr: position L - j.
    const int L = h - (found ? 0 : 1);
    const unsigned xr = xm >> (7 - L);
    int cur = 0;  // the position r.px/py/pz/step stand at
    auto advance_to = [&](int tgt) {  // the reference's own sums (init.py:115-116), from the kept position where that is nearer
        int rem = tgt - cur;
#ifndef VRT_DIAG
        if (cur < CKPT_AT && tgt >= CKPT_AT) {
            r.px = cqx;
            r.py = cqy;
            r.pz = cqz;
            r.step = cqs;
            rem = tgt - CKPT_AT;
        }
#endif
#pragma unroll
        for (int k = 0; k < SPEC; k++) {
#ifndef VRT_DIAG
            if (k >= CKPT_AT) break;  // rem is at most that
#endif
            if (k < rem) {
                r.step += sd;
                r.px += dvx;
                r.py += dvy;
                r.pz += dvz;
                DG_BRICK();
            }
        }
        cur = tgt;
    };
#ifdef VRT_DIAG
    if (__ballot(xr != 0u)) { DG_ADD(DG_SNAP_ITERS, 1); DG_ADD(DG_SNAP_LANES, __popcll(__ballot(xr != 0u))); }
#endif
    if (xr != 0u) {
        VRT_MARK("w_replay");
        const auto& Q = fresh_args(P);  // (see fresh_args)
        const int msb = 31 - __clz((int)xr);
        const int kc = L - msb;  // the first border
        const unsigned xr2 = xr & ~(1u << msb);
        if (xr2 != 0u) {  // a second border: the step ends in front of it
            h = L - (31 - __clz((int)xr2));
            found = false;
        }
        advance_to(kc);
        int fx, fy, fz;
        floor3_i32(r.px, r.py, r.pz, fx, fy, fz);
        // the reference's inclusive box test (init.py:67) in integers, as in march_step (the ray has snapped before: it has a chunk)
        const unsigned cs4 = C.cs4;
        const unsigned ux = ((unsigned)fx << 2) + (unsigned)r.nm4x, uy = ((unsigned)fy << 2) + (unsigned)r.nm4y,
                       uz = ((unsigned)fz << 2) + (unsigned)r.nm4z;
        const unsigned umax = ux > uy ? (ux > uz ? ux : uz) : (uy > uz ? uy : uz);
        const bool outside = (umax > cs4) | ((ux == cs4) & (r.px != (double)fx)) | ((uy == cs4) & (r.py != (double)fy)) |
                             ((uz == cs4) & (r.pz != (double)fz));
        if (outside) {
            resnap_commit<false, true>(Q, C, r, fx, fy, fz, wmin_key, sl, Q.snap_flags);
            // the positions behind the border were computed for this step size and read from the block of the cell's own
            // chunk: they stand if that chunk is there at the same resolution
            const bool same = r.entry != 0u && (RESMODE == 0 || (r.entry >> 24) == res);
            if (!same) {
                h = kc;
                found = false;
            }
        } else {
            // on the upper faces of the current chunk: the reference keeps the chunk, finds nothing in it at this cell
            // (Frame.get_voxel outside the chunk's own box) and moves on.  That one iteration is all this step does.
            h = kc == 0 ? 1 : kc;
            found = false;
        }
    }
    VRT_MARK("w_adv");
#ifdef VRT_DIAG_HIST
#pragma unroll
    for (int k = 1; k <= SPEC; k++) {
        DG_ADD(DG_NV1 + k - 1, __popcll(__ballot(n_life >= k)));
        DG_ADD(DG_H1 + k - 1, __popcll(__ballot(h >= k)));
    }
#endif
    advance_to(h);
    cnt[C_LOOKUP] += h + (found ? 1 : 0);
    cnt[C_ADV] += h;
    if (found) {
        r.color |= id << 24;
        state = LANE_HIT;
    }
}

// HIT: phase B of the reference loop (init.py:78-116) for a lane in LANE_HIT -- lib.material, the termination tests,
// the reflection from the three neighbour voxels, the advance.  Leaves the lane in LANE_MARCH or one of the ENDED states.
// LIST (a re-trace launch) with P.prefix_draws > 0: the frame's march counted this ray's events into its lanes' totals
// up to the hit at which a P.prefix_draws-wide row ran out (march_pool_kernel keeps no per-ray counts); the re-trace
// walks through the same hit with the same counts and takes them off again there.
template <int RESMODE, bool LIST>
__device__ __forceinline__ void hit_body(const MarchParams& P, const MarchCtx& C, Ray& r, int& state, int32_t (&cnt)[C_NLOCAL],
                                         DgLane& dg) {
    (void)dg;
    const auto& Q = fresh_args(P);  // (see fresh_args)
    const double cs = C.kept_cs ? C.cs : COLD(COLD_CS);
    const unsigned cs4 = C.cs4;
    const lds_f64* mat = C.mats + ((int)(r.color >> 24) - 1) * 8;
    const double m_rough = mat[3], m_absorb = mat[4], m_ior = mat[5], m_energy = mat[6];
    const bool have_draws = r.ndraw + 3 <= Q.n_draws;
    bool exhausted = false;
    // ---- lib.material (lib.py:448-460) ----
    double a = m_absorb / pow_cached(C.pc, 1 + r.bounces, COLD(COLD_POW_Y));
    if (!(a < 1)) a = 1;
    const double b2 = 1 - a;
    {
        const int cr = (int)__builtin_rint((double)(r.color & 255u) * b2 + mat[0] * a);
        const int cg = (int)__builtin_rint((double)((r.color >> 8) & 255u) * b2 + mat[1] * a);
        const int cb = (int)__builtin_rint((double)((r.color >> 16) & 255u) * b2 + mat[2] * a);
        r.color = (uint32_t)cr | ((uint32_t)cg << 8) | ((uint32_t)cb << 16);
    }
    r.energy = r.energy * b2 + m_energy * a;
    r.life *= 1 - (m_rough * a);
    cnt[C_HIT]++;
    if (m_rough != 0.0) {  // lib.rand draws nothing for amplitude 0 (lib.py:431-434)
        if (have_draws) {
            r.vx += rand_amp(r.d0, m_rough);
            r.vy += rand_amp(r.d1, m_rough);
            r.vz += rand_amp(r.d2, m_rough);
        } else {
            exhausted = true;
        }
        if (LIST && Q.prefix_draws > 0 && r.ndraw <= Q.prefix_draws && r.ndraw + 3 > Q.prefix_draws) {
            lds_u32* col = C.tot + (threadIdx.x & (VRT_TOT_COLS - 1));
            __hip_atomic_fetch_add(col + VRT_C_LOOKUP * VRT_TOT_COLS, (uint32_t)-cnt[C_LOOKUP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(col + VRT_C_NBR * VRT_TOT_COLS, (uint32_t)-cnt[C_NBR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(col + VRT_C_CHUNK_GET * VRT_TOT_COLS, (uint32_t)-cnt[C_CGET], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(col + VRT_C_HIT * VRT_TOT_COLS, (uint32_t)-cnt[C_HIT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(col + VRT_C_ADV * VRT_TOT_COLS, (uint32_t)-cnt[C_ADV], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        r.ndraw += 3;
    }
    VRT_MARK("h_tests");
    // ---- init.py:82-86 ----
    const unsigned res = r.entry >> 24;
    const double stepd = RESMODE == 0 ? 1.0 : (double)(res ? res : 1u);
    r.bounces += m_absorb;
    r.life /= stepd + m_absorb * COLD(COLD_LOD_BOUNCES);
    const double ref = absmax3_f64(r.vx, r.vy, r.vz);
    if (ref != 0.0 && ref != 1.0) div3_same_divisor(r.vx, r.vy, r.vz, ref);
    // (one flag from three compares, no short-circuit branches: every level of such a branch nest re-copies the ray's
    // registers at its join)
    const bool stop = ((int)(r.step >= r.life) | (int)(r.energy >= COLD(COLD_MAX_LIGHT)) | (int)(r.bounces >= COLD(COLD_MAX_BOUNCES1))) != 0;
    // ENDED_EXHAUSTED: the result is discarded and the ray re-traced with a longer draw table; ENDED_BROKE: left through
    // the reference's `break` (init.py:86)
    state = exhausted ? LANE_ENDED_EXHAUSTED : LANE_ENDED_BROKE;
    if (!(exhausted | stop)) {
        VRT_MARK("h_reflect");
        // ---- reflection from the three neighbours (init.py:92-111) ----
        if (m_ior != 0.0) {
            const double direction = (m_ior - 0.5) * 2;
            // Three independent neighbour lookups, done in phases so that their memory accesses overlap:
            // (1) which chunk each neighbour point belongs to, (2) its voxel offset, (3) the three reads.
            // The point of axis `ax` is ray.pos with +/- 1 added to that coordinate (init.py:94-96); its cell is
            // floor() of it (data.py:136), taken from the sum itself.
            int fl[3];
            floor3_i32(r.px, r.py, r.pz, fl[0], fl[1], fl[2]);
            const int nm4[3] = {r.nm4x, r.nm4y, r.nm4z};
            double np[3];
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                const double v = ax == 0 ? r.vx : (ax == 1 ? r.vy : r.vz);
                const double p = ax == 0 ? r.px : (ax == 1 ? r.py : r.pz);
                np[ax] = p + (v < direction ? 1.0 : -1.0);
            }
            int gl[3];
            floor3_i32(np[0], np[1], np[2], gl[0], gl[1], gl[2]);
            uint32_t nentry[3];
            int n4[3][3], nnm4[3][3];  // per point: 4 * local coordinates, -4 * chunk_min of its chunk
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                int nf[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    nf[c] = c == ax ? gl[c] : fl[c];
                    nnm4[ax][c] = nm4[c];
                    n4[ax][c] = (int)(((unsigned)nf[c] << 2) + (unsigned)nm4[c]);
                }
                nentry[ax] = r.entry;
                // init.py:100-102: the point stays in the current chunk when it is inside its inclusive box
                // (the other two coordinates are the ray's own, already inside); else Camera.chunk_get
                // (init.py:28-33) snaps every coordinate of the point
                const double mn = (double)(-(nm4[ax] >> 2));
                const bool foreign = !((np[ax] >= mn) & (np[ax] <= mn + cs));
                if (foreign) {
                    int cc[3];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        cc[c] = nf[c] >> Q.cs_shift;
                        nnm4[ax][c] = -(cc[c] << (Q.cs_shift + 2));
                        n4[ax][c] = (int)(((unsigned)nf[c] << 2) + (unsigned)nnm4[ax][c]);
                    }
                    nentry[ax] = chunk_entry_i(Q, C.ct, cc[0] - Q.origin_c[0], cc[1] - Q.origin_c[1], cc[2] - Q.origin_c[2], Q.dims[0], Q.dims[1], Q.dims[2],
                                               Q.ct_cells != 0, Q.ct_identity != 0);
                    cnt[C_CGET]++;
                }
            }
            unsigned noff[3];
#if VRT_HIT_BATCH
            if (RESMODE != 2) {
                // (the nine table words of the three cells are read together, then combined: one LDS round trip, not three)
                unsigned tw[3][3];
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    const unsigned m4n = RESMODE == 0 ? ~0u : ((nentry[ax] >> 24) == 2u ? 0x3f8u : 0x3fcu);
#pragma unroll
                    for (int c = 0; c < 3; c++) tw[ax][c] = tab_at(C, c, (unsigned)n4[ax][c] & m4n);
                }
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    const unsigned nb = ((nentry[ax] & 0xffffffu) - 1u) << (3 * Q.cs_shift);
                    const bool in = (unsigned)(n4[ax][0] | n4[ax][1] | n4[ax][2]) < cs4;
                    noff[ax] = (nentry[ax] != 0u && in) ? nb + (tw[ax][0] | tw[ax][1] | tw[ax][2]) : ~0u;
                    cnt[C_NBR] += nentry[ax] != 0u ? 1 : 0;
                }
            } else
#endif
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                const unsigned nres = nentry[ax] >> 24;
                const unsigned m4n = (RESMODE != 0 && nres == 2u) ? 0x3f8u : 0x3fcu;
                const unsigned nb = ((nentry[ax] & 0xffffffu) - 1u) << (3 * Q.cs_shift);
                const unsigned t = cell_offset<RESMODE>(C, nentry[ax], nb, m4n, cs4, nnm4[ax][0], nnm4[ax][1], nnm4[ax][2],
                                                        n4[ax][0], n4[ax][1], n4[ax][2],
                                                        (unsigned)(n4[ax][0] | n4[ax][1] | n4[ax][2]) < cs4, true);
                noff[ax] = nentry[ax] != 0u ? t : ~0u;
                cnt[C_NBR] += nentry[ax] != 0u ? 1 : 0;
            }
            unsigned nid[3];
#pragma unroll
            for (int ax = 0; ax < 3; ax++) nid[ax] = __builtin_amdgcn_raw_buffer_load_b8(C.vox, noff[ax], 0, 0);
            bool solid[3];
#if VRT_HIT_BATCH
            {   // (no short-circuit: the three material reads leave together once the three bytes are there)
                double nior[3];
#pragma unroll
                for (int ax = 0; ax < 3; ax++) nior[ax] = C.mats[(nid[ax] ? (int)nid[ax] - 1 : 0) * 8 + 5];
#pragma unroll
                for (int ax = 0; ax < 3; ax++) solid[ax] = (nid[ax] != 0u) & (nior[ax] == m_ior);
            }
#else
#pragma unroll
            for (int ax = 0; ax < 3; ax++) solid[ax] = nid[ax] != 0u && C.mats[((int)nid[ax] - 1) * 8 + 5] == m_ior;
#endif
            if (!solid[0]) r.vx -= r.vx * m_ior * 2;
            if (!solid[1]) r.vy -= r.vy * m_ior * 2;
            if (!solid[2]) r.vz -= r.vz * m_ior * 2;
        }
        VRT_MARK("h_draws");
        // The draws of the ray's NEXT rough hit are requested now (only a rough hit consumed the ones held): they
        // come from HBM, and the wave's next wait on memory is the refill's (or the first march step's), which
        // then covers both.
        if (m_rough != 0.0 && r.ndraw + 3 <= Q.n_draws) {
            const double* row = Q.draws + (int64_t)r.rowi * Q.draw_stride + r.ndraw;
            r.d0 = row[0];
            r.d1 = row[1];
            r.d2 = row[2];
        }
        // ---- advance inside a present chunk (init.py:114-116) ----
        r.step += stepd;
        r.px += r.vx * stepd;
        r.py += r.vy * stepd;
        r.pz += r.vz * stepd;
        DG_BRICK();
        cnt[C_ADV]++;
        state = LANE_MARCH;
    }
}

// ENDED: lib.material_background (lib.py:463-476), tile()'s alpha (init.py:141), the ray's outputs.  PERRAY: `cnt` holds
// this ray's events and joins the totals here; else the lane counts events of all its rays in `cnt` itself and only the
// per-ray words (re-snaps, draws, broke) are added.  A ray whose draws ran out is queued for the re-trace instead.
template <bool RECORD, bool PERRAY>
__device__ __forceinline__ void ended_body(const MarchParams& P, const MarchCtx& C, const Ray& r, int state, const int32_t (&cnt)[C_NLOCAL],
                                           int nseen, unsigned long long* s_stats) {
    (void)nseen;
    const auto& Q = fresh_args(P);  // (see fresh_args)
    // (the arguments the usual path needs are asked for together: one scalar-memory round trip)
    const int64_t ray0 = Q.ray0;
    uint32_t* const out_rgba = Q.ray_rgba;
    const bool has_bg = Q.st.has_background != 0;
    const int64_t ray = ray0 + r.off;
    if (state == LANE_ENDED_EXHAUSTED) {
        bool queued = false;
        if (Q.retrace_list) {
            const uint32_t slot = atomicAdd(Q.retrace_count, 1u);
            if (slot < Q.retrace_cap) {
                Q.retrace_list[slot] = r.off;
                queued = true;
            }
        }
        if (!queued) atomicAdd(&s_stats[VRT_S_RNG_EXHAUSTED], 1ull);
        return;
    }
    double energy = r.energy;
    const uint32_t rgba = finish_color(C.pc, has_bg, COLD(COLD_POW_Y), COLD(COLD_SHUTTER), r.color, energy, r.bounces, r.vy);
    if (out_rgba) out_rgba[ray] = rgba;
    const int broke = state == LANE_ENDED_BROKE ? 1 : 0;
    if (RECORD && Q.rays) {
        vrt_ray& o = Q.rays[ray];
        int x = 0, y = 0, s = 0;
        double detail;
        if (C.tile) {
            const int64_t p = ray / Q.g.smax;
            s = (int)(ray - p * Q.g.smax);
            x = Q.g.pixels[2 * p];
            y = Q.g.pixels[2 * p + 1];
            double dx, dy;
            int ns;
            pixel_setup(P.st, x, y, dx, dy, detail, ns);
            detail = detail / (1 + s * P.st.lod_samples) * (1 - P.st.lod_random * P.draws[(int64_t)r.rowi * P.draw_stride]);
        } else {
            detail = Q.expl_detail[ray];
        }
        o.x = x; o.y = y; o.s = s;
        o.color[0] = (int)(rgba & 255u); o.color[1] = (int)((rgba >> 8) & 255u); o.color[2] = (int)((rgba >> 16) & 255u);
        o.alpha = (int)(rgba >> 24);
        o.ntrav = nseen;
        o.counters[VRT_C_LOOKUP] = cnt[C_LOOKUP];
        o.counters[VRT_C_NBR] = cnt[C_NBR];
        o.counters[VRT_C_RESNAP] = r.resnaps;
        o.counters[VRT_C_CHUNK_GET] = cnt[C_CGET];
        o.counters[VRT_C_HIT] = cnt[C_HIT];
        o.counters[VRT_C_DRAW] = r.ndraw;
        o.counters[VRT_C_ADV] = cnt[C_ADV];
        o.counters[VRT_C_BROKE] = broke;
        o.detail = detail; o.energy = energy; o.step = r.step; o.life = r.life; o.bounces = r.bounces;
        o.pos[0] = r.px; o.pos[1] = r.py; o.pos[2] = r.pz;
        o.vel[0] = r.vx; o.vel[1] = r.vy; o.vel[2] = r.vz;
    }
    lds_u32* col = C.tot + (threadIdx.x & (VRT_TOT_COLS - 1));
#define VRT_TOT_ADD(j, v) __hip_atomic_fetch_add(col + (j) * VRT_TOT_COLS, (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
    if (PERRAY) {
        VRT_TOT_ADD(VRT_C_LOOKUP, cnt[C_LOOKUP]);
        VRT_TOT_ADD(VRT_C_NBR, cnt[C_NBR]);
        VRT_TOT_ADD(VRT_C_CHUNK_GET, cnt[C_CGET]);
        VRT_TOT_ADD(VRT_C_HIT, cnt[C_HIT]);
        VRT_TOT_ADD(VRT_C_ADV, cnt[C_ADV]);
    }
    VRT_TOT_ADD(VRT_C_RESNAP, r.resnaps);
    VRT_TOT_ADD(VRT_C_DRAW, r.ndraw);
    VRT_TOT_ADD(VRT_C_BROKE, broke);
    VRT_TOT_ADD(VRT_NCOUNTERS, 1);
#undef VRT_TOT_ADD
}

// ---- what both kernels do before and after their loops ---------------------------------------------------------------
template <bool W>
struct MarchSharedT {  // static LDS of a march workgroup (W: the per-axis offset tables are part of the world-axis tables)
    unsigned long long stats[VRT_NSTATS];
    unsigned long long pw_keys[VRT_PW_SLOTS];
    unsigned long long pw_vals[VRT_PW_SLOTS];
    uint32_t tab[W ? 4 : 3 * 256];
    uint32_t tot[VRT_NCOUNTERS + 1][VRT_TOT_COLS];
    uint32_t wmin[VRT_BLOCK / VRT_WAVE], wtmp[VRT_BLOCK / VRT_WAVE];
    double cold[COLD_N];
};
typedef MarchSharedT<false> MarchShared;
// dynamic LDS: materials | chunk table | settled bitmap [| brick slots of lookup variant 2 | ray pool]
template <bool W = false>
__device__ __forceinline__ void march_prologue(const MarchParams& P, MarchSharedT<W>& S, unsigned char* s_dyn, MarchCtx& C) {
    double* s_mats = reinterpret_cast<double*>(s_dyn);
    uint32_t* s_ct = reinterpret_cast<uint32_t*>(s_dyn + (size_t)P.n_materials * 64);
    uint32_t* s_trav = s_ct + P.ct_cells;
    for (int i = threadIdx.x; i < P.n_materials * 8; i += VRT_BLOCK) s_mats[i] = P.materials[i];
    for (int i = threadIdx.x; i < P.ct_cells; i += VRT_BLOCK) s_ct[i] = P.chunk_table[i];
    for (int i = threadIdx.x; i < P.trav_words; i += VRT_BLOCK) s_trav[i] = 0u;
    if (threadIdx.x < VRT_BLOCK / VRT_WAVE) S.wmin[threadIdx.x] = 0u;
    for (int i = threadIdx.x; i < VRT_PW_SLOTS; i += VRT_BLOCK) {
        unsigned long long k = 0, v = 0;
        if (P.pow_global) {
            k = __hip_atomic_load(&P.pow_global[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = __hip_atomic_load(&P.pow_global[VRT_PW_SLOTS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        S.pw_keys[i] = k;
        S.pw_vals[i] = k ? v : 0ull;
    }
    for (int i = threadIdx.x; !W && i < 256; i += VRT_BLOCK) {
        const int c = i < P.cs ? i : 0;  // (entries beyond the chunk are never selected)
        S.tab[i] = (uint32_t)voxel_offset(P.cs, c, 0, 0);  // fixed stride 256: constant LDS offsets in the lookups
        S.tab[256 + i] = (uint32_t)voxel_offset(P.cs, 0, c, 0);
        S.tab[512 + i] = (uint32_t)voxel_offset(P.cs, 0, 0, c);
    }
    if (threadIdx.x < VRT_TOT_COLS) {
#pragma unroll
        for (int j = 0; j <= VRT_NCOUNTERS; j++) S.tot[j][threadIdx.x] = 0u;
    }
    if (threadIdx.x < VRT_NSTATS) S.stats[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        for (int a = 0; a < 3; a++) S.cold[COLD_POS + a] = P.cam.pos[a];
        for (int a = 0; a < 4; a++) S.cold[COLD_ROT + a] = P.cam.rot[a];
        S.cold[COLD_DIST_MIN] = P.st.dist_min;
        S.cold[COLD_POW_Y] = 1 + P.st.falloff;
        S.cold[COLD_LOD_BOUNCES] = P.st.lod_bounces;
        S.cold[COLD_MAX_LIGHT] = P.st.max_light;
        S.cold[COLD_MAX_BOUNCES1] = P.st.max_bounces + 1;
        S.cold[COLD_SHUTTER] = P.st.shutter;
        S.cold[COLD_CS] = (double)P.cs;
        S.cold[COLD_INV_CS] = 1.0 / (double)P.cs;
    }
    // world-axis offset tables (P.wt_on; world_tables_kernel): copied to LDS like the chunk table
    C.wt = (const lds_char*)(s_dyn + P.wt_lds_off);
    C.wb[0] = C.wb[1] = C.wb[2] = 0;
    if (W) {
        uint32_t* wt = reinterpret_cast<uint32_t*>(s_dyn + P.wt_lds_off);
        for (int i = threadIdx.x; i < P.wt_words; i += VRT_BLOCK) wt[i] = P.wt_table[i];
        int first = 0;
        for (int a = 0; a < 3; a++) {
            C.wb[a] = 4 * (first + VRT_WT_GUARD - P.wt_min[a]);
            C.tl[a] = 4 * (first + VRT_WT_GUARD);
            first += P.wt_cells[a] + 2 * VRT_WT_GUARD;
        }
        C.tab = C.wt;
    } else {
        C.tab = (const lds_char*)S.tab;
        C.tl[0] = 0;
        C.tl[1] = 1024;
        C.tl[2] = 2048;
    }
    C.cs3 = 1u << (3 * P.cs_shift);
    C.ct = (const lds_u32*)s_ct;
    C.bm = (lds_u32*)s_trav;
    C.mats = (const lds_f64*)s_mats;
    C.cold = (const lds_f64*)S.cold;
    C.tot = (lds_u32*)&S.tot[0][0];
    C.pc.keys = S.pw_keys;
    C.pc.vals = S.pw_vals;
    C.pc.gkeys = P.pow_global;
    C.pc.gvals = P.pow_global ? P.pow_global + VRT_PW_SLOTS : nullptr;
    C.vox = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(P.voxels), 0, (int)P.vox_bytes, 0x00020000);
    C.cs4 = (unsigned)P.cs << 2;
    C.cs = (double)P.cs;
    C.inv_cs = 1.0 / C.cs;
    C.kept_cs = !W;
    C.has_bm = P.trav_words != 0;

    C.tile = P.g.pixels != nullptr;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        C.td[a] = P.t_dims[a];
        C.toc[a] = P.t_origin_c[a];
        C.oc[a] = P.origin_c[a];
        C.dm[a] = P.dims[a];
        if (!W) asm volatile("" : "+v"(C.td[a]));  // (W: every re-snap reads the dimensions with its other arguments)
    }
}
// the workgroup's totals -> the launch's statistics.  Columns are added as signed 32-bit sums: a re-trace launch takes
// the prefix of a re-traced ray off again (hit_body), which may leave a column below zero until the ray's full counts
// join the frame's 64-bit statistics
template <bool LIST, class ST>
__device__ __forceinline__ void march_epilogue(const MarchParams& P, ST& S) {
    __syncthreads();
    if (threadIdx.x < VRT_TOT_COLS) {
#pragma unroll
        for (int j = 0; j < VRT_NCOUNTERS; j++) {
            const uint32_t t = S.tot[j][threadIdx.x];
            if (t) atomicAdd(&S.stats[j], LIST ? (unsigned long long)(long long)(int32_t)t : (unsigned long long)t);
        }
        const uint32_t done = S.tot[VRT_NCOUNTERS][threadIdx.x];
        if (done) {
            atomicAdd(&S.stats[VRT_S_RAYS], (unsigned long long)done);
            if (LIST) atomicAdd(&S.stats[VRT_S_RNG_RETRACED], (unsigned long long)done);
        }
    }
    __syncthreads();
    if (threadIdx.x < VRT_NSTATS && S.stats[threadIdx.x])
        atomicAdd((unsigned long long*)&P.stats[threadIdx.x], S.stats[threadIdx.x]);
}
// smallest key any ray of this workgroup can still produce (see trav_cell): `mine` = smallest ray index this wave holds,
// reduced through a scratch word, then published in one store, so that the other waves of the workgroup only ever read
// a lower bound (a wave's value never decreases)
template <class ST>
__device__ __forceinline__ uint64_t publish_wave_min(ST& S, int wave_in_block, uint32_t mine) {
    if ((threadIdx.x & 63) == 0) S.wmin[wave_in_block] = mine;
    // (the wave's own word is `mine` by now: LDS operations of one wave complete in order)
    uint32_t m = S.wmin[0];
#pragma unroll
    for (int w = 1; w < VRT_BLOCK / VRT_WAVE; w++) {
        const uint32_t o = S.wmin[w];
        m = o < m ? o : m;
    }
    return (uint64_t)m << 12;
}
#ifdef VRT_DIAG
__device__ __forceinline__ void diag_flush(DgLane& dg, unsigned long long dg_start, unsigned long long dg_t_start,
                                           unsigned long long dg_t_empty) {
    DG_ADD(DG_WAVE_CYCLES, DG_TIME() - dg_start);
    if (dg.bv) atomicAdd(&g_diag[DG_BRICK_VISITS], dg.bv);
    if ((threadIdx.x & 63) == 0) {
        for (int j = 0; j < DG_N; j++) atomicAdd(&g_diag[j], (unsigned long long)dg.acc[j]);
        const unsigned long long t_exit = __builtin_amdgcn_s_memrealtime();
        atomicMax(&g_diag_t[0], ~dg_t_start);
        if (dg_t_empty) atomicMax(&g_diag_t[1], ~dg_t_empty);
        atomicMax(&g_diag_t[2], t_exit);
        atomicAdd(&g_diag_t[3], t_exit);
        atomicAdd(&g_diag_t[4], 1ull);
    }
}
#endif

// Persistent waves, one ray per lane.  Every lane is a small state machine: MARCH (phase A of the reference loop: snap
// chunk, look up the voxel, advance -- init.py:66-77, 114-116), HIT (phase B: shade, test termination, reflect, advance
// -- init.py:78-116), ENDED (background + outputs -- init.py:119-120, 141-142), IDLE (take the next ray of the wave's
// range).  The cheap MARCH step runs every iteration; the expensive HIT and ENDED (+ refill) bodies run once t_hit /
// t_end lanes wait for them (or nothing marches, or max_iters iterations have passed), so they execute with many
// lanes active.  Per-ray semantics are exactly the reference's single loop.
//
// The kernel is bound by instruction issue, so the march step is written for few instructions: local coordinates are
// kept times 4 (they index the per-axis offset tables in LDS directly), voxel bytes are read with raw buffer loads
// whose offset is ~0 for a position that must not be looked up (the hardware's range check returns 0 = empty, no
// predicate or branch), and the SPEC bytes of an iteration are combined into one word whose first non-zero byte is
// the hit.  The chunk table (if it has at most VRT_CT_LDS_MAX cells) and a per-wave bitmap of settled traversed cells
// live in LDS, so crossing into another chunk costs no global round trip.
//
// LK selects how a march step learns whether its cell is occupied (the results are identical):
//   0  the voxel's material byte itself (shipped: fewest instructions, and the kernel is bound by instruction issue)
//   1  one bit of the 64-bit occupancy word of the cell's 4^3 micro-brick, kept in registers while the ray stays in
//      that micro-brick (3.5 x fewer L1 requests, 40 % more instructions per step)
//   2  one bit of the cell's 8^3 brick of occupancy bits (64 bytes), staged in a per-lane LDS slot (the "LDS-staged 8^3
//      bricks" of BASELINE.json's north star)
// 1 and 2 are kept for measurement (VRT_LOOKUP=1|2, profiles/r02_v7_lookup_variants.md); a hit reads the byte in both.
// W: the scene's blocks lie in table order and the march step looks ahead across chunk borders (march_step_w)
// DEFER: see march_step
template <int SPEC, int RESMODE, bool RECORD, bool LIST, int LK = 0, int PERPIX = (RECORD || LIST) ? 4 : 0, bool W = false, bool DEFER = false, int SEED = 0>
__global__ void __launch_bounds__(VRT_BLOCK, VRT_WAVES_PER_SIMD) march_kernel(MarchParams P) {
    static_assert(SPEC >= 4 && SPEC <= 16, "speculation depth");
    static_assert(!W || (!RECORD && !LIST && LK == 0 && RESMODE != 2 && SPEC == 8), "march_step_w");
    static_assert(!DEFER || (!W && !RECORD && !LIST && LK == 0), "deferred key comparison: frame kernels, byte lookup");
    __shared__ MarchSharedT<W> S;
    extern __shared__ __align__(16) unsigned char s_dyn[];
    if (LIST && *P.list_count == 0) return;  // the usual case: no ray ran out of draws
    MarchCtx C;
    march_prologue<W>(P, S, s_dyn, C);
    __syncthreads();

    const int wave_in_block = threadIdx.x >> 6;
    // rays are handed out in chunks of VRT_CHUNK consecutive rays from a launch-wide counter: coherent lanes,
    // balanced waves.  `next`/`range_end` are wave-uniform.
    const int64_t count = LIST ? (int64_t)(*P.list_count < P.list_cap ? *P.list_count : P.list_cap) : P.n;
    const int64_t chunk = P.chunk;
    int64_t next = 0, range_end = 0;
    bool more = true;  // the launch-wide counter may still have rays
    if (chunk == 0) {  // static contiguous range per wave
        const int64_t n_waves = (int64_t)gridDim.x * (VRT_BLOCK / VRT_WAVE);
        const int64_t wave = (int64_t)blockIdx.x * (VRT_BLOCK / VRT_WAVE) + wave_in_block;
        int64_t per = (count + n_waves - 1) / n_waves;
        per = (per + 7) & ~(int64_t)7;
        next = wave * per;
        range_end = next + per < count ? next + per : count;
        more = false;
    }

    Ray r;
    r.px = r.py = r.pz = r.vx = r.vy = r.vz = 0;
    r.step = r.life = r.bounces = r.energy = 0;
    r.nm4x = r.nm4y = r.nm4z = 0;
    r.entry = 0;
    r.boff = 0;
    r.color = 0;
    r.ndraw = r.resnaps = 0;
    r.off = 0;
    r.rowi = 0;
    r.d0 = r.d1 = r.d2 = 0.5;
    int state = LANE_IDLE;
    int32_t cnt[C_NLOCAL];   // events of the current ray
#pragma unroll
    for (int j = 0; j < C_NLOCAL; j++) cnt[j] = 0;
    SeenList<RECORD> sl;
    sl.n = 0;
    uint64_t wmin_key = 0;
    LkState lk;
    lk.okey = ~0u;
    lk.oword = 0;
    lk.brick_slot = LK == 2 ? reinterpret_cast<uint64_t*>(s_dyn + P.brick_lds_off) + 9 * threadIdx.x : nullptr;
    DgLane dg;
#ifdef VRT_DIAG
    __shared__ unsigned long long s_dg[VRT_BLOCK / VRT_WAVE][DG_N];
    dg.acc = (__attribute__((address_space(3))) unsigned long long*)&s_dg[threadIdx.x >> 6][0];
    if ((threadIdx.x & 63) == 0)
        for (int j = 0; j < DG_N; j++) dg.acc[j] = 0;
    dg.brick = ~0u;
    dg.bv = 0;
    const unsigned long long dg_t_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long dg_t_empty = 0;
    const unsigned long long dg_start = DG_TIME();
#endif

    for (;;) {
#ifdef VRT_DIAG
        DG_ADD(DG_PASSES, 1);
        unsigned long long dg_t0 = DG_TIME();
        {
            const int nidle = __popcll(__ballot(state == LANE_IDLE));
            if (nidle && (next < range_end || more)) { DG_ADD(DG_REFILL_EXEC, 1); DG_ADD(DG_REFILL_LANES, nidle); }
        }
#endif
        // (W: fresh rays snap to their first chunk in the refill, so the bound of the wave's future keys -- see trav_cell --
        // is taken before it and covers the rays it may take; with the wave's range used up their indices are not known
        // yet and the older, smaller bound stays)
        if (W && C.has_bm && (!(next < range_end || more) || next < range_end)) {
            if ((threadIdx.x & 63) == 0) S.wtmp[wave_in_block] = (next < range_end || more) ? (uint32_t)(P.ray0 + next) : 0xffffffffu;
            if (state != LANE_IDLE) atomicMin(&S.wtmp[wave_in_block], (uint32_t)(P.ray0 + r.off));
            wmin_key = publish_wave_min(S, wave_in_block, S.wtmp[wave_in_block]);
        }
        // ------------------------------------------------------------------ refill idle lanes
        unsigned long long idle_mask = __ballot(state == LANE_IDLE);
        while (idle_mask != 0ull && (next < range_end || more)) {
            if (next >= range_end) {  // take the next chunk (one atomic per wave per chunk)
                unsigned long long base = 0;
                if ((threadIdx.x & 63) == 0) base = atomicAdd(P.queue_head, (unsigned long long)chunk);
                base = wave_first_u64(base);  // (into scalar registers: what is derived from it stays wave-uniform, scalar code)
                if ((int64_t)base >= count) {
                    more = false;
#ifdef VRT_DIAG
                    if (!dg_t_empty) dg_t_empty = __builtin_amdgcn_s_memrealtime();
#endif
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
                sl.n = 0;
                if (take_ray<RECORD, LIST, PERPIX, W, SEED>(P, C, k, r, dg, wmin_key, sl)) {
#pragma unroll
                    for (int j = 0; j < C_NLOCAL; j++) cnt[j] = 0;
                    state = LANE_MARCH;
                }
            }
            idle_mask = __ballot(state == LANE_IDLE);
        }
        if (__ballot(state != LANE_IDLE) == 0ull) break;  // range exhausted and every lane finished
        if (!W && C.has_bm) {
            if ((threadIdx.x & 63) == 0) S.wtmp[wave_in_block] = 0xffffffffu;
            if (state != LANE_IDLE) atomicMin(&S.wtmp[wave_in_block], (uint32_t)(P.ray0 + r.off));
            wmin_key = publish_wave_min(S, wave_in_block, S.wtmp[wave_in_block]);
        }
#ifdef VRT_DIAG
        unsigned long long dg_t1 = DG_TIME();
        DG_ADD(DG_CYC_REFILL, dg_t1 - dg_t0);
#endif

        // ------------------------------------------------------------------ MARCH steps (phase A)
        int iters = 0;
        for (;;) {
            const int n_march = (int)__popcll(__ballot(state == LANE_MARCH));
            const int n_hit = (int)__popcll(__ballot(state == LANE_HIT));
            const int n_end = (int)__popcll(__ballot(state >= LANE_ENDED));
            if (n_march == 0 || n_hit >= P.t_hit || n_end >= P.t_end) break;
            if (iters >= P.max_iters && n_hit + n_end > 0) break;
            iters++;
            DG_ADD(DG_ITERS, 1);
            DG_ADD(DG_MARCH_LANES, n_march);
            if (state == LANE_MARCH) {
                if constexpr (W) march_step_w<RESMODE>(P, C, r, state, cnt, wmin_key, lk, sl, dg);
                else march_step<SPEC, RESMODE, RECORD, LK, VRT_FRESH_MARCH, DEFER ? 1 : 0>(P, C, r, state, cnt, wmin_key, lk, sl, dg);
            }
        }
        const bool none_marching = __ballot(state == LANE_MARCH) == 0ull;
        const bool capped = iters >= P.max_iters;
#ifdef VRT_DIAG
        unsigned long long dg_t2 = DG_TIME();
        DG_ADD(DG_CYC_MARCH, dg_t2 - dg_t1);
#endif

        // ------------------------------------------------------------------ HIT (phase B: init.py:78-116)
        const bool serve_hit = none_marching || capped || (int)__popcll(__ballot(state == LANE_HIT)) >= P.t_hit;
#ifdef VRT_DIAG
        if (serve_hit && __ballot(state == LANE_HIT)) { DG_ADD(DG_HIT_EXEC, 1); DG_ADD(DG_HIT_LANES, __popcll(__ballot(state == LANE_HIT))); }
#endif
        if (serve_hit && state == LANE_HIT) hit_body<RESMODE, LIST>(P, C, r, state, cnt, dg);

        // ------------------------------------------------------------------ ENDED: background, outputs
        const bool serve_ended = none_marching || capped || (int)__popcll(__ballot(state >= LANE_ENDED)) >= P.t_end ||
                                 __ballot(state == LANE_MARCH) == 0ull;
#ifdef VRT_DIAG
        unsigned long long dg_t3 = DG_TIME();
        DG_ADD(DG_CYC_HIT, dg_t3 - dg_t2);
        if (serve_ended && __ballot(state >= LANE_ENDED)) { DG_ADD(DG_END_EXEC, 1); DG_ADD(DG_END_LANES, __popcll(__ballot(state >= LANE_ENDED))); }
#endif
        if (serve_ended && state >= LANE_ENDED) {
            ended_body<RECORD, true>(P, C, r, state, cnt, sl.n, S.stats);
            state = LANE_IDLE;
        }
#ifdef VRT_DIAG
        DG_ADD(DG_CYC_END, DG_TIME() - dg_t3);
#endif
    }
#ifdef VRT_DIAG
    diag_flush(dg, dg_start, dg_t_start, dg_t_empty);
#endif
    if (W && threadIdx.x == 0) atomicAdd(&S.stats[VRT_S_LOOKAHEAD_GROUPS], 1ull);
    if (!LIST && threadIdx.x == 0 && PERPIX == 3) atomicAdd(&S.stats[VRT_S_RAYGEN_GROUPS], 1ull);
    march_epilogue<LIST>(P, S);
}

// ---- march_pool_kernel: the same bodies, rays regrouped between the lanes of a wave through LDS ------------------------
// march_kernel executes a body for the lanes that wait for it while the others idle: per VALU instruction 49 % of the
// lanes are active at config 3 (profiles/r02_v7_sq_c3_summary.txt), and the kernel is bound by VALU issue.  Here every
// wave owns VRT_POOL_SLOTS parked rays in LDS besides the 64 in its lanes.  A pass picks ONE body -- HIT once t_hit rays
// wait for it anywhere in the wave's pool, ENDED (+ refill) once t_end do, else MARCH -- and first brings rays of that
// state into the lanes that hold something else: such a lane exchanges its whole ray (18 eight-byte words and its state)
// with a parked one by ds_wrxchg, or, when the pool has room and the launch has rays left, parks its waiting ray in a free
// slot and takes a fresh ray instead.  Nothing is shared between waves: no flags, no atomics on the pool, no waiting.
//
// A parked ray has no lane to count its events in, so lanes count the events of whatever rays they run (`tot`, flushed
// once at the end) and only the per-ray words (re-snaps, draws, broke) join the totals when a ray finishes.  A ray whose
// draws ran out has already been counted up to that hit; the re-trace launch takes exactly that prefix off again
// (hit_body, P.prefix_draws).
#ifndef VRT_POOL_SLOTS
#define VRT_POOL_SLOTS 48
#endif
#define VRT_POOL_WORDS 18  // 8-byte words per parked ray
#define VRT_POOL_WAVE_BYTES (VRT_POOL_WORDS * 8 * VRT_POOL_SLOTS)
static_assert(VRT_POOL_SLOTS >= 8 && VRT_POOL_SLOTS <= 63, "one lane looks after one slot; lane 63 stands for no slot");

__device__ __forceinline__ double lds_xchg_f64(lds_u64* p, double v) {
    const unsigned long long o = __hip_atomic_exchange(p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WAVEFRONT);
    return __longlong_as_double((long long)o);
}
__device__ __forceinline__ uint32_t lds_xchg_u32(lds_u32* p, uint32_t v) {
    return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
// mask bit of this lane ? if_set : if_clear, for a wave-uniform mask: one v_cndmask with the mask as its condition (the
// compiler's own select on a lane's bit of a 64-bit value shifts, masks and compares first)
__device__ __forceinline__ int sel_mask(unsigned long long mask, int if_set, int if_clear) {
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}
// the lane's ray and state <-> slot s of the wave's pool.  Field-major layout: word w of slot s at (w * SLOTS + s) * 8,
// so lanes that exchange with different slots touch different banks.  Word 17 = draw row | state << 32.
__device__ __forceinline__ void pool_swap(lds_u64* pool, int s, int cs_shift, Ray& r, int& state) {
    lds_u64* q = pool + s;
#define VRT_X64(w, f) f = lds_xchg_f64(q + (w) * VRT_POOL_SLOTS, f)
    VRT_X64(0, r.px); VRT_X64(1, r.py); VRT_X64(2, r.pz);
    VRT_X64(3, r.vx); VRT_X64(4, r.vy); VRT_X64(5, r.vz);
    VRT_X64(6, r.step); VRT_X64(7, r.life); VRT_X64(8, r.bounces); VRT_X64(9, r.energy);
    VRT_X64(10, r.d0); VRT_X64(11, r.d1); VRT_X64(12, r.d2);
#undef VRT_X64
    lds_u32* u = (lds_u32*)(pool + 13 * VRT_POOL_SLOTS) + s;
#define VRT_X32(w, f) f = lds_xchg_u32(u + (w) * VRT_POOL_SLOTS, f)
    uint32_t a;
    a = (uint32_t)r.nm4x; VRT_X32(0, a); r.nm4x = (int)a;
    a = (uint32_t)r.nm4y; VRT_X32(1, a); r.nm4y = (int)a;
    a = (uint32_t)r.nm4z; VRT_X32(2, a); r.nm4z = (int)a;
    VRT_X32(3, r.entry);
    VRT_X32(4, r.color);
    a = (uint32_t)r.ndraw; VRT_X32(5, a); r.ndraw = (int)a;
    a = (uint32_t)r.resnaps; VRT_X32(6, a); r.resnaps = (int)a;
    VRT_X32(7, r.off);
    VRT_X32(8, r.rowi);
    a = (uint32_t)state; VRT_X32(9, a); state = (int)a;
#undef VRT_X32
    r.boff = ((r.entry & 0xffffffu) - 1u) << (3 * cs_shift);
}

// Tiled hand-out: ticket g of the launch -> the rays of `tile_px` pixels of one column.  Tickets are numbered tile by tile
// (tile_px columns each) along the Morton curve of the padded tile grid, so that the 2^(2 log2 - 3) tiles a head hands out
// form one compact eighth of the window and consecutive tickets stay inside one square tile: the rays a wave (and the
// waves of its XCD) march at a time start close together and, after their first hits, scatter around one patch of the
// scene -- what a 4 MiB L2 can hold (config 5).  In plain list order the 4 096 waves work on a 16-column, full-height strip.
// Returns false for a ticket of the padding.
__device__ __forceinline__ uint32_t morton_even_bits(uint32_t v) {  // bits 0, 2, 4, ... of v, packed
    v &= 0x55555555u;
    v = (v | (v >> 1)) & 0x33333333u;
    v = (v | (v >> 2)) & 0x0f0f0f0fu;
    v = (v | (v >> 4)) & 0x00ff00ffu;
    v = (v | (v >> 8)) & 0x0000ffffu;
    return v;
}
// (tile_px is a power of two, tile_px_log2 its logarithm; tickets are 32-bit: launch_march's caller checks)
template <class PT>
__device__ __forceinline__ bool tile_ticket(const PT& Q, uint32_t g, int64_t& first_ray) {
    const uint32_t t = g >> Q.tile_px_log2, c = g & ((1u << Q.tile_px_log2) - 1u);
    const uint32_t tx = morton_even_bits(t), ty = morton_even_bits(t >> 1);
    const uint32_t x = (tx << Q.tile_px_log2) + c;
    first_ray = ((int64_t)x * Q.st.height + (int64_t)(ty << Q.tile_px_log2)) * Q.g.smax;
    return x < (uint32_t)Q.st.width && ty < (uint32_t)Q.tile_nty;
}
// The wave's next ticket: from the head its state names, else from the next one that still has tickets (every head is tried
// once; tickets of the tile grid's padding are skipped).  Scalar code only -- no ray register is touched.  Returns the new
// state word (bits 0-2 the head, 3-6 heads found dry, 7 "few tickets left"); first_ray < 0: every head has run dry.
__device__ __forceinline__ int tile_take(const MarchParams& P, int ts, int lane, int64_t chunk, int64_t& first_ray) {
    const auto& Q = fresh_args(P);
    const uint32_t per_head = 1u << (Q.tile_px_log2 + 2 * Q.tile_log2 - 3);
    first_ray = -1;
    while ((ts & 0x78) != 0x40) {
        uint32_t k = 0;
        if (lane == 0) k = atomicAdd(Q.tile_heads + (ts & 7) * 32, 1u);
        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
        if (k >= per_head) {  // this head has run dry: the next one
            ts = ((ts + 1) & 7) | ((ts & 0x78) + 8) | 0x80;
            continue;
        }
        // (as in list order: nothing is parked any more once fewer than two rays per lane of the XCD's waves are left)
        if ((int64_t)(per_head - k) * chunk < (int64_t)(gridDim.x >> 3) * (2 * VRT_BLOCK)) ts |= 0x80;
        int64_t first;
        if (tile_ticket(Q, (uint32_t)(ts & 7) * per_head + k, first)) {
            first_ray = first;
            break;
        }
    }
    return ts;
}
#define VRT_POOL_STATE_WORD (13 * 2 * VRT_POOL_SLOTS + 9 * VRT_POOL_SLOTS)  // index (in 32-bit words) of slot 0's state
#define VRT_POOL_OFF_WORD (13 * 2 * VRT_POOL_SLOTS + 7 * VRT_POOL_SLOTS)    // ... and of its ray offset

template <int SPEC, int RESMODE, int PERPIX = 0, bool W = false, bool DEFER = false, bool TILE = false>
__global__ void __launch_bounds__(VRT_BLOCK, VRT_WAVES_PER_SIMD) march_pool_kernel(MarchParams P) {
    static_assert(!TILE || DEFER, "tiled hand-out: the instances for scenes beyond the caches");
    static_assert(!W || (RESMODE != 2 && SPEC == 8), "march_step_w");
    static_assert(!(W && DEFER), "deferred key comparison: the shipped march step");
    __shared__ MarchSharedT<W> S;
    extern __shared__ __align__(16) unsigned char s_dyn[];
    MarchCtx C;
    march_prologue<W>(P, S, s_dyn, C);
    const int wave_in_block = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    lds_u64* const pool = (lds_u64*)(s_dyn + P.pool_lds_off + wave_in_block * VRT_POOL_WAVE_BYTES);
    lds_u32* const pool_state = (lds_u32*)pool + VRT_POOL_STATE_WORD;
    if (lane < VRT_POOL_SLOTS) pool_state[lane] = (uint32_t)LANE_IDLE;
    __syncthreads();

    const int64_t count = P.n;
    const int64_t chunk = P.chunk > 0 ? P.chunk : VRT_CHUNK;
    const int64_t tail_start = count - (int64_t)gridDim.x * (2 * VRT_BLOCK);
    int64_t next = 0, range_end = 0;
    bool more = true;  // the launch-wide counter may still have rays
    // tiled hand-out (tile_ticket): the head this wave takes from -- its XCD's until that runs dry, then the others' in turn
    // (instances of their own, a measured variant: VRT_TILED=1, launch_march.  One word of state: bits 0-2 the head, 3-6
    // heads found dry, 7 "few tickets left where this wave takes them" -- like range_end >= tail_start)
    bool tiled = false;
    int ts = 0;
    if constexpr (TILE) {
        tiled = P.tile_heads != nullptr && P.plan_hdr->full_frame != 0;
        unsigned xcc = 0;
        xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // hwreg(HW_REG_XCC_ID, 0, 4)
        ts = (int)(xcc & 7u);
    }

    Ray r;
    r.px = r.py = r.pz = r.vx = r.vy = r.vz = 0;
    r.step = r.life = r.bounces = r.energy = 0;
    r.nm4x = r.nm4y = r.nm4z = 0;
    r.entry = 0;
    r.boff = 0;
    r.color = 0;
    r.ndraw = r.resnaps = 0;
    r.off = 0;
    r.rowi = 0;
    r.d0 = r.d1 = r.d2 = 0.5;
    int state = LANE_IDLE;
    SeenList<false> sl;
    sl.n = 0;
    LkState lk;
    lk.okey = ~0u;
    lk.oword = 0;
    lk.brick_slot = nullptr;
    uint64_t wmin_key = 0;
    DgLane dg;
#ifdef VRT_DIAG
    __shared__ unsigned long long s_dg[VRT_BLOCK / VRT_WAVE][DG_N];
    dg.acc = (__attribute__((address_space(3))) unsigned long long*)&s_dg[threadIdx.x >> 6][0];
    if ((threadIdx.x & 63) == 0)
        for (int j = 0; j < DG_N; j++) dg.acc[j] = 0;
    dg.brick = ~0u;
    dg.bv = 0;
    const unsigned long long dg_t_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long dg_t_empty = 0;
    const unsigned long long dg_start = DG_TIME();
#endif

    // refill: idle lanes take the next rays of the wave's range (wave-uniform; `lane` etc. from the kernel)
    auto refill = [&]() {
        unsigned long long idle_mask = __ballot(state == LANE_IDLE);
#ifdef VRT_DIAG
        DG_ADD(DG_REFILL_EXEC, 1);
        DG_ADD(DG_REFILL_LANES, __popcll(idle_mask));
#endif
        while (idle_mask != 0ull && (next < range_end || more)) {
            if (TILE && next >= range_end && tiled) {  // take the next ticket of a head (one atomic per wave per ticket)
                int64_t first = -1;
                ts = tile_take(P, ts, lane, chunk, first);
                if (first < 0) {  // every head has run dry
                    more = false;
#ifdef VRT_DIAG
                    if (!dg_t_empty) dg_t_empty = __builtin_amdgcn_s_memrealtime();
#endif
                    break;
                }
                next = first;
                range_end = first + chunk;
            } else if (next >= range_end) {  // take the next chunk (one atomic per wave per chunk)
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(P.queue_head, (unsigned long long)chunk);
                base = wave_first_u64(base);  // (into scalar registers: the pass's decisions stay scalar code)
                if ((int64_t)base >= count) {
                    more = false;
#ifdef VRT_DIAG
                    if (!dg_t_empty) dg_t_empty = __builtin_amdgcn_s_memrealtime();
#endif
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
                SeenList<false> none;
                none.n = 0;
                if (take_ray<false, false, PERPIX, W>(P, C, k, r, dg, W ? wmin_key : 0ull, none)) state = LANE_MARCH;
            }
            idle_mask = __ballot(state == LANE_IDLE);
        }
    };
    // events of the rays this lane runs.  PASS_COUNTS: they join the lane's column of the workgroup's totals at the end of
    // every pass (LDS adds) instead of living in registers for the whole kernel -- what the look-ahead variant needs to
    // stay inside its registers; the shipped kernels have the five registers (-DVRT_PASS_COUNTS=1 gives them up too)
    constexpr bool PASS_COUNTS = W || VRT_PASS_COUNTS;
    int32_t tot[C_NLOCAL];
#pragma unroll
    for (int j = 0; j < C_NLOCAL; j++) tot[j] = 0;
    unsigned pass = 0;
    int stalled = 0;  // MARCH passes that found nothing to march
    for (;; pass++) {
#ifdef VRT_DIAG
        DG_ADD(DG_PASSES, 1);
#endif
        VRT_MARK("pass");
        if (PASS_COUNTS) {
#pragma unroll
            for (int j = 0; j < C_NLOCAL; j++) tot[j] = 0;
        }
        // ------------------------------------------------------------------ what waits where
        const int sstate = lane < VRT_POOL_SLOTS ? (int)__hip_atomic_load(pool_state + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : -1;
        const bool lane_i = state == LANE_IDLE;
        const unsigned long long l_m = __ballot(state == LANE_MARCH), l_h = __ballot(state == LANE_HIT),
                                 l_e = __ballot(state >= LANE_ENDED), l_i = __ballot(lane_i);
        const unsigned long long s_m = __ballot(sstate == LANE_MARCH), s_h = __ballot(sstate == LANE_HIT),
                                 s_e = __ballot(sstate >= LANE_ENDED), s_f = __ballot(sstate == LANE_IDLE);
        const int n_m = (int)(__popcll(l_m) + __popcll(s_m)), n_h = (int)(__popcll(l_h) + __popcll(s_h)),
                  n_e = (int)(__popcll(l_e) + __popcll(s_e));
        const bool rays_left = next < range_end || more;
        // A MARCH pass with nothing to march happens when it was chosen for the new rays it could take and the launch had
        // none left: once per wave.  (HIT and ENDED passes always find a lane: they are chosen only when a ray waits for
        // them, and the exchange brings one into a lane if no lane holds one.)  No state of the scheduler may leave a wave
        // spinning on the GPU: a wave that has met it 4 096 times gives up and says so (VRT_S_STALLED: the frame is
        // invalid, the wrapper raises).
        if (stalled > 4096) {
            if (lane == 0) atomicAdd(&S.stats[VRT_S_STALLED], 1ull);
            break;
        }
        // new rays can enter through idle lanes, or through lanes whose waiting ray goes to a free slot -- but nothing is
        // parked any more near the end of the launch (fewer than two rays per lane of the grid left to hand out): what a
        // wave parks then it must finish alone after the others have run dry
        const bool evict_ok = rays_left && ((TILE && tiled) ? !(ts & 0x80) : range_end < tail_start);
        const bool can_add = rays_left && (l_i != 0ull || (evict_ok && s_f != 0ull));
        int target;
        if (rays_left) {  // steady state: the slow bodies once enough rays wait for them, else march (with fresh rays if need be)
            if (n_h >= P.t_hit) target = LANE_HIT;
            else if (n_e >= P.t_end) target = LANE_ENDED;
            else if (n_m > 0 || can_add) target = LANE_MARCH;
            else target = n_h >= n_e ? LANE_HIT : LANE_ENDED;  // (the pool is full of waiting rays)
        } else {
            // The launch has no rays left.  What counts now is how many passes the wave's last rays still need, not how
            // many lanes a body finds (the GPU is running empty): every pass runs all three bodies, each for the lanes
            // that wait for it, and idle lanes take parked rays of any state.
            if (n_m + n_h + n_e == 0) break;
            target = LANE_MARCH;
        }
        const bool tail = !rays_left;
        // The smallest ray index the wave holds or may still take bounds its future keys from below (see trav_cell).  It only
        // grows, so it is refreshed every 8th pass only; the rays this pass may still take count too (the refill comes
        // later in the pass), and when the wave's range is used up their indices are not known yet: no refresh then.
        if (C.has_bm && (pass & 7u) == 0u && (!rays_left || next < range_end)) {
            if (lane == 0) S.wtmp[wave_in_block] = rays_left ? (uint32_t)(P.ray0 + next) : 0xffffffffu;
            if (!lane_i) atomicMin(&S.wtmp[wave_in_block], (uint32_t)(P.ray0 + r.off));
            if (sstate > LANE_IDLE)
                atomicMin(&S.wtmp[wave_in_block],
                          (uint32_t)(P.ray0 + __hip_atomic_load((lds_u32*)pool + VRT_POOL_OFF_WORD + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)));
            wmin_key = publish_wave_min(S, wave_in_block, S.wtmp[wave_in_block]);
        }

        VRT_MARK("exchange");
        // ------------------------------------------------------------------ bring rays of the target state into the lanes
        {
            const bool t_m = target == LANE_MARCH, t_h = target == LANE_HIT;
            // (tail: every lane that holds a ray is where it should be, every parked ray is wanted)
            const unsigned long long l_t = tail ? ~l_i : (t_m ? l_m : (t_h ? l_h : l_e));
            const unsigned long long c1 = tail ? (s_m | s_h | s_e) : (t_m ? s_m : (t_h ? s_h : s_e));  // slots to take from
            // MARCH, and parking is allowed: a lane's waiting ray may also go to a free slot, the lane takes a fresh ray
            const unsigned long long c2 = (t_m && evict_ok) ? s_f : 0ull;
            const unsigned long long l_a = ~(l_t | l_i);  // lanes that hold a ray of another state
            const int n1 = (int)__popcll(c1), n2 = (int)__popcll(c2), n_a = (int)__popcll(l_a), n_b = (int)__popcll(l_i);
            // the j-th such lane (ray holders first, then idle lanes) is paired with the j-th such slot (rays first, then free slots);
            // an idle lane has no use for a free slot
            const int take = n_a + n_b < n1 ? n_a + n_b : n1;
            const int evict = n_a > n1 ? (n_a - n1 < n2 ? n_a - n1 : n2) : 0;
            if (take + evict >= (tail ? 1 : P.pool_swap_min) || (take + evict > 0 && l_t == 0ull)) {
#ifdef VRT_DIAG
                unsigned long long dg_ts = DG_TIME();
                DG_ADD(DG_SWAPS, 1);
                DG_ADD(DG_SWAP_LANES, take + evict);
                DG_ADD(DG_EVICT_LANES, evict);
#endif
                auto rank_in = [&](unsigned long long m) {
                    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                };
                // (lane predicates straight from the wave-uniform masks: one v_cndmask each)
                // slot side: the slot a lane looks after sends its number to the lane of its rank (slots that take no part
                // all send to lane 63, which no rank reaches: at most VRT_POOL_SLOTS <= 63 ranks)
                int crank = 63;
                if (n2 > 0) crank = sel_mask(c2, n1 + rank_in(c2), crank);
                crank = sel_mask(c1, rank_in(c1), crank);
                const int slot_of_rank = __builtin_amdgcn_ds_permute(crank << 2, lane);
                // lane side: ray holders first, then idle lanes; a lane already in the target state takes no part
                const int brank = sel_mask(l_a, rank_in(l_a), n_a + rank_in(l_i));
                const int my_slot = __builtin_amdgcn_ds_bpermute(brank << 2, slot_of_rank);
                const int limit = sel_mask(l_t, 0, sel_mask(l_a, n1 + n2, n1));
                if (brank < limit) pool_swap(pool, my_slot, P.cs_shift, r, state);
#ifdef VRT_DIAG
                DG_ADD(DG_CYC_SWAP, DG_TIME() - dg_ts);
#endif
            }
        }

        VRT_MARK("select");
        // The bodies follow one another, each for the lanes whose state AND the pass's target call for it -- lane predicates
        // (exec masks), not an if / else chain on the wave-uniform target: alternatives of a uniform branch need their ray
        // registers reconciled where they join (~250 more register copies in this kernel), masked bodies write them in place.
        int tv = tail ? 7 : (target == LANE_ENDED ? 1 : (target == LANE_MARCH ? 2 : 4));  // bodies to run: ENDED 1, MARCH 2, HIT 4
        asm volatile("" : "+v"(tv));  // (a per-lane copy the compiler cannot prove uniform)
        // ------------------------------------------------------------------ ENDED: outputs; the lanes take new rays at once
#ifdef VRT_DIAG
        unsigned long long dg_t1 = DG_TIME();
        if ((target == LANE_ENDED || tail) && __ballot(state >= LANE_ENDED)) { DG_ADD(DG_END_EXEC, 1); DG_ADD(DG_END_LANES, __popcll(__ballot(state >= LANE_ENDED))); }
#endif
        VRT_MARK("ended");
        if ((tv & 1) && state >= LANE_ENDED) {
            ended_body<false, false>(P, C, r, state, tot, 0, S.stats);
            state = LANE_IDLE;
        }
        VRT_MARK("refill");
#ifdef VRT_DIAG
        unsigned long long dg_t2 = DG_TIME();
        DG_ADD(DG_CYC_END, dg_t2 - dg_t1);
#endif
        // ------------------------------------------------------------------ refill: after ENDED; before MARCH once enough lanes idle
        if (target != LANE_HIT && (next < range_end || more)) {
            const unsigned long long idle_mask = __ballot(state == LANE_IDLE);
            if (idle_mask != 0ull && (target == LANE_ENDED || (int)__popcll(idle_mask) >= P.pool_refill_min ||
                                      __ballot(state == LANE_MARCH) == 0ull))
                refill();
        }
#ifdef VRT_DIAG
        unsigned long long dg_t3 = DG_TIME();
        DG_ADD(DG_CYC_REFILL, dg_t3 - dg_t2);
#endif
        VRT_MARK("march");
        // ------------------------------------------------------------------ MARCH steps: further ones at once while most lanes
        // still march (a pass costs ~40 instructions before its body starts)
        for (int it = 0; target == LANE_MARCH && it < (tail ? 1 : P.max_iters); it++) {
            const int marching = (int)__popcll(__ballot(state == LANE_MARCH));
            stalled += (it == 0 && marching == 0 && !tail) ? 1 : 0;  // (a MARCH pass with nothing to march: see the top of the loop)
            if (marching == 0 || (it > 0 && marching < P.pool_keep)) break;
#ifdef VRT_DIAG
            DG_ADD(DG_ITERS, 1);
            DG_ADD(DG_MARCH_LANES, marching);
#endif
            if (state == LANE_MARCH) {
                if constexpr (W) march_step_w<RESMODE>(P, C, r, state, tot, wmin_key, lk, sl, dg);
                else march_step<SPEC, RESMODE, false, 0, VRT_FRESH_MARCH_POOL, DEFER ? 2 : 0>(P, C, r, state, tot, wmin_key, lk, sl, dg);
            }
        }
#ifdef VRT_DIAG
        unsigned long long dg_t4 = DG_TIME();
        DG_ADD(DG_CYC_MARCH, dg_t4 - dg_t3);
        if ((target == LANE_HIT || tail) && __ballot(state == LANE_HIT)) { DG_ADD(DG_HIT_EXEC, 1); DG_ADD(DG_HIT_LANES, __popcll(__ballot(state == LANE_HIT))); }
#endif
        VRT_MARK("hit");
        // ------------------------------------------------------------------ HIT
        if ((tv & 4) && state == LANE_HIT) hit_body<RESMODE, false>(P, C, r, state, tot, dg);
        VRT_MARK("pass_end");
        // the events this lane counted in this pass (HIT passes: the shader's; MARCH passes: lookups and advances)
        if (PASS_COUNTS) {
            lds_u32* col = C.tot + (lane & (VRT_TOT_COLS - 1));
            const bool ran_march = tail || target == LANE_MARCH, ran_hit = tail || target == LANE_HIT;  // (wave-uniform)
            if (ran_hit) {
                __hip_atomic_fetch_add(col + VRT_C_NBR * VRT_TOT_COLS, (uint32_t)tot[C_NBR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(col + VRT_C_CHUNK_GET * VRT_TOT_COLS, (uint32_t)tot[C_CGET], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(col + VRT_C_HIT * VRT_TOT_COLS, (uint32_t)tot[C_HIT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (ran_march) __hip_atomic_fetch_add(col + VRT_C_LOOKUP * VRT_TOT_COLS, (uint32_t)tot[C_LOOKUP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ran_march || ran_hit) __hip_atomic_fetch_add(col + VRT_C_ADV * VRT_TOT_COLS, (uint32_t)tot[C_ADV], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#ifdef VRT_DIAG
        DG_ADD(DG_CYC_HIT, DG_TIME() - dg_t4);
#endif
    }
    if (!PASS_COUNTS) {  // the events this lane counted
        lds_u32* col = C.tot + (lane & (VRT_TOT_COLS - 1));
        __hip_atomic_fetch_add(col + VRT_C_LOOKUP * VRT_TOT_COLS, (uint32_t)tot[C_LOOKUP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(col + VRT_C_NBR * VRT_TOT_COLS, (uint32_t)tot[C_NBR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(col + VRT_C_CHUNK_GET * VRT_TOT_COLS, (uint32_t)tot[C_CGET], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(col + VRT_C_HIT * VRT_TOT_COLS, (uint32_t)tot[C_HIT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(col + VRT_C_ADV * VRT_TOT_COLS, (uint32_t)tot[C_ADV], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#ifdef VRT_DIAG
    diag_flush(dg, dg_start, dg_t_start, dg_t_empty);
#endif
    if (threadIdx.x == 0) atomicAdd(&S.stats[VRT_S_POOL_GROUPS], (TILE && tiled) ? (1ull << 32) + 1ull : 1ull);  // (see vrt.h)
    if (W && threadIdx.x == 0) atomicAdd(&S.stats[VRT_S_LOOKAHEAD_GROUPS], 1ull);
    if (PERPIX == 3 && threadIdx.x == 0) atomicAdd(&S.stats[VRT_S_RAYGEN_GROUPS], 1ull);
    march_epilogue<false>(P, S);
}

// ---------------------------------------------------------------------------------------------
// resolve: lib.average over the samples of each pixel (init.py:142-146)
// ---------------------------------------------------------------------------------------------
// mean of a pixel's samples: fp32 RGBA, and RGBA8 as one word (Surface.set_at float -> u8, init.py:146: truncation, exact
// for samples == 1)
__device__ __forceinline__ uint32_t resolve_pixel(const vrt_settings& st, int smax, const uint32_t* ray_rgba, int64_t p, int x, int y,
                                                  float4& mean) {
    double dx, dy, det;
    int ns;
    pixel_setup(st, x, y, dx, dy, det, ns);
    uint32_t sr = 0, sg = 0, sb = 0, sa = 0;
    for (int s = 0; s < ns; s++) {
        uint32_t v = ray_rgba[p * smax + s];
        sr += v & 255u;
        sg += (v >> 8) & 255u;
        sb += (v >> 16) & 255u;
        sa += v >> 24;
    }
    const double n = (double)ns;
    const double mr = (double)sr / n, mg = (double)sg / n, mb = (double)sb / n, ma = (double)sa / n;
    mean = make_float4((float)mr, (float)mg, (float)mb, (float)ma);
    return (uint32_t)(unsigned char)(int)mr | ((uint32_t)(unsigned char)(int)mg << 8) | ((uint32_t)(unsigned char)(int)mb << 16) |
           ((uint32_t)(unsigned char)(int)ma << 24);
}
// One thread per pixel of the list.  The list is x-major (the reference's settings.pixels order), the image row-major:
// consecutive threads write 4 bytes W * 4 bytes apart.  When the list is the whole window in that order (the plan
// knows: PlanHeader.full_frame) a workgroup takes a 16 x 16 pixel tile instead, reads its 16 columns' samples, and
// writes the image rows from an LDS tile, 64 contiguous bytes at a time (config 3: 0.18 -> 0.1 ms).
__global__ void __launch_bounds__(VRT_BLOCK) resolve_kernel(vrt_settings st, TileGeom g, const PlanHeader* hdr,
                                                            const uint32_t* ray_rgba, float* rgba_f32, uint8_t* image_u8) {
    __shared__ uint32_t s_tile[16][17];
    if (hdr->full_frame) {
        const int tiles_y = (st.height + 15) / 16;
        const int bx = (int)(blockIdx.x / tiles_y), by = (int)(blockIdx.x % tiles_y);
        if (bx * 16 >= st.width) return;  // (the grid also covers the list-order mapping)
        const int ty = threadIdx.x & 15, tx = threadIdx.x >> 4;  // reading: y fastest, as the list runs
        const int x = bx * 16 + tx, y = by * 16 + ty;
        if (x < st.width && y < st.height) {
            const int64_t p = (int64_t)x * st.height + y;
            float4 mean;
            const uint32_t px = resolve_pixel(st, g.smax, ray_rgba, p, x, y, mean);
            if (rgba_f32) reinterpret_cast<float4*>(rgba_f32)[p] = mean;
            s_tile[tx][ty] = px;
        }
        __syncthreads();
        if (image_u8) {
            const int wx = threadIdx.x & 15, wy = threadIdx.x >> 4;  // writing: x fastest, as the image runs
            const int ox = bx * 16 + wx, oy = by * 16 + wy;
            if (ox < st.width && oy < st.height) reinterpret_cast<uint32_t*>(image_u8)[(int64_t)oy * st.width + ox] = s_tile[wx][wy];
        }
        return;
    }
    int64_t p = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (p >= g.n_px) return;
    int x = g.pixels[2 * p], y = g.pixels[2 * p + 1];
    float4 mean;
    const uint32_t px = resolve_pixel(st, g.smax, ray_rgba, p, x, y, mean);
    if (rgba_f32) reinterpret_cast<float4*>(rgba_f32)[p] = mean;
    if (image_u8) reinterpret_cast<uint32_t*>(image_u8)[(int64_t)y * st.width + x] = px;
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
                                                             int cs, const uint32_t* chunk_list, uint32_t* table, uint8_t* voxels) {
    const int chunk = chunk_list ? (int)chunk_list[blockIdx.x] : (int)blockIdx.x;
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
// tile consumer: Window.draw_tile's alpha-over blit onto the persistent canvas (init.py:185-190)
// ---------------------------------------------------------------------------------------------
// pygame blits an RGBA tile (SRCALPHA) onto the SRCALPHA canvas with its own blender, surface.h ALPHA_BLEND
// (pygame 2.x; pygame is not installable here, so this restates the published macro and is PARITY UNPINNED):
//   dst alpha == 0:  dst = src
//   else:            dC = ((dC << 8) + (sC - dC) * sA + sC) >> 8   per colour channel,  dA = sA + dA - (sA * dA) / 255
// A tile's alpha is round(min(1, energy + shutter) * 255) (init.py:141): the blend is the reference's motion blur.
__device__ __forceinline__ uchar4 alpha_blend(uchar4 s, uchar4 d) {
    if (d.w == 0) return s;
    const int sa = s.w;
    uchar4 o;
    o.x = (unsigned char)((((int)d.x << 8) + ((int)s.x - (int)d.x) * sa + (int)s.x) >> 8);
    o.y = (unsigned char)((((int)d.y << 8) + ((int)s.y - (int)d.y) * sa + (int)s.y) >> 8);
    o.z = (unsigned char)((((int)d.z << 8) + ((int)s.z - (int)d.z) * sa + (int)s.z) >> 8);
    o.w = (unsigned char)(sa + (int)d.w - (sa * (int)d.w) / 255);
    return o;
}
// pixels == nullptr: every pixel of the window; else only the listed ones (a tile's other pixels are transparent,
// and a transparent source pixel leaves the canvas as it is)
__global__ void __launch_bounds__(VRT_BLOCK) canvas_blit_kernel(uchar4* canvas, const uchar4* tile, const int32_t* pixels,
                                                                int64_t n, int width) {
    const int64_t i = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t at = pixels ? (int64_t)pixels[2 * i + 1] * width + pixels[2 * i] : i;
    canvas[at] = alpha_blend(tile[at], canvas[at]);
}

// ---------------------------------------------------------------------------------------------
// host side of the C ABI
// ---------------------------------------------------------------------------------------------
static constexpr int D_SLOW = 113;   // draws in the retrace table (all outputs that need no state twist)
static constexpr int64_t SLOW_CAP_MIN = 1 << 18;  // rays per launch that may be re-traced with a 113-draw row:
static constexpr int64_t SLOW_CAP_MAX = 1 << 22;  // 1/64 of the launch, within these bounds
static constexpr int64_t FULL_CAP = 1 << 12;      // of those, rays that may be re-traced again with D_FULL_DEV draws

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// ray slots per march launch: every launch ends with a drain phase in which the last, longest rays finish in
// mostly empty waves, so fewer and larger launches are better (VRT_BATCH_LOG2 overrides, 12..30)
static int64_t batch_rays() {
    static int64_t b = 0;
    if (!b) {
        int l = env_int("VRT_BATCH_LOG2", 28);
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
// What a frame clears before its first launch, in one launch: the statistics, the first batch's counters, the per-frame
// pow memo (if any) and -- when the caller asks (vrt_traversed.reset) -- the traversed keys.
__global__ void __launch_bounds__(VRT_BLOCK) frame_begin_kernel(uint32_t* stats, int n_stats, uint32_t* count, int n_count, uint32_t* pow_memo,
                                                                int n_pow, unsigned long long* keys, int64_t n_keys) {
    const int64_t t = (int64_t)blockIdx.x * VRT_BLOCK + threadIdx.x, step = (int64_t)gridDim.x * VRT_BLOCK;
    for (int64_t i = t; i < n_keys; i += step) keys[i] = ~0ull;
    if (blockIdx.x != 0) return;
    for (int i = threadIdx.x; i < n_stats; i += VRT_BLOCK) stats[i] = 0u;
    for (int i = threadIdx.x; i < n_count; i += VRT_BLOCK) count[i] = 0u;
    for (int i = threadIdx.x; i < n_pow; i += VRT_BLOCK) pow_memo[i] = 0u;
}
static inline void frame_begin(uint64_t* d_stats, void* count, int n_count_words, void* pow_memo, int n_pow_words, const vrt_traversed* trav,
                               hipStream_t stream) {
    int64_t n_keys = 0;
    if (trav && trav->d_keys && trav->reset) n_keys = (int64_t)trav->dims[0] * trav->dims[1] * trav->dims[2];  // (fill_params checked them)
    const int64_t kb = (n_keys + VRT_BLOCK * 8 - 1) / (VRT_BLOCK * 8);
    hipLaunchKernelGGL(frame_begin_kernel, dim3((unsigned)(kb < 1 ? 1 : (kb > 1024 ? 1024 : kb))), dim3(VRT_BLOCK), 0, stream, (uint32_t*)d_stats,
                       (int)(2 * VRT_NSTATS), (uint32_t*)count, n_count_words, (uint32_t*)pow_memo, n_pow_words,
                       n_keys ? (unsigned long long*)trav->d_keys : nullptr, n_keys);
}
static inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

// march grid: persistent workgroups; each wave owns a contiguous range of the launch's rays
static int march_grid(int64_t n) {
    static int cap = -1;
    if (cap < 0) {
        cap = env_int("VRT_MARCH_GRID", 256 * VRT_WAVES_PER_SIMD);
        if (cap < 1) cap = 1;
    }
    int64_t g = (n + VRT_BLOCK * 4 - 1) / (VRT_BLOCK * 4);  // at least ~4 rays per lane
    if (g < 1) g = 1;
    return (int)(g < cap ? g : cap);
}

// The pow memo (see PowCache) caches a pure function, x -> vrt_pow(x, y) for one exponent y, so it may outlive the
// frame: one 4-KB table per (device, exponent), created by vrt_pow_memo_create, zeroed once and only ever inserted
// into afterwards.  Without it every workgroup of every frame starts cold and recomputes the same few dozen powers
// (0.3-0.4 ms per frame on MI355X, whatever the frame size); frames whose (device, exponent) has no table memoise
// into their workspace.  VRT_POW_MEMO=frame forces that.
struct PowMemo {
    int dev;
    double y;
    unsigned long long* buf;
};
static constexpr int MAX_MEMOS = 64;
static std::mutex g_memo_mu;
static PowMemo g_memos[MAX_MEMOS];
static int g_n_memos = 0;
static bool memo_per_frame() {
    static int per_frame = -1;
    if (per_frame < 0) {
        const char* e = getenv("VRT_POW_MEMO");
        per_frame = (e && e[0] == 'f') ? 1 : 0;
    }
    return per_frame != 0;
}
// lookup only: never allocates (vrt_render_tile / vrt_trace_rays stay capturable)
static unsigned long long* device_pow_memo(double y) {
    if (memo_per_frame()) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_memo_mu);
    for (int i = 0; i < g_n_memos; i++)
        if (g_memos[i].dev == dev && g_memos[i].y == y) return g_memos[i].buf;
    return nullptr;
}

// rays per hand-out for a launch of n rays: 512, but 256 / 128 for small launches (about one 512-ray chunk per wave
// would leave nothing to balance: config 2's 2 M rays march in 0.62 instead of 0.73 ms; a 1/8 share of config 3 in
// 1.17 instead of 1.18 ms); 64 is too few (1.89 ms: the lanes of a wave lose their coherence); VRT_CHUNK overrides
static int march_chunk(int64_t n) {
    static int c = -2;
    if (c == -2) {
        const char* e = getenv("VRT_CHUNK");
        c = e ? atoi(e) : -1;
        if (e && c < 0) c = 0;
    }
    if (c >= 0) return c;
    return n <= ((int64_t)1 << 22) ? 128 : (n <= ((int64_t)1 << 24) ? 256 : VRT_CHUNK);
}

// the voxel data is far larger than L2 + Infinity Cache (config 5): most lookups miss, and the scheduling thresholds that
// suit it differ (march_policy)
static bool march_big_scene(const vrt_scene* sc) {
    const int64_t bytes = (int64_t)sc->n_slots * sc->chunk_size * sc->chunk_size * sc->chunk_size;
    return bytes > ((int64_t)512 << 20);
}
// reference iterations fetched together per march pass: VRT_SPEC_DEEP (8) for scenes of resolutions 1 and 2 -- measured
// on MI355X after the per-position cost fell to ~18 VALU instructions: config 3 6.87 ms with 8 against 7.21 (4), 7.04 (6),
// 7.49 (12); config 5 305 ms against 392 (4), 326 (6), 316 (12) -- and for big scenes of any resolution; VRT_SPEC (4) for
// the generic-resolution kernel otherwise (8 positions spill registers there).  VRT_SPEC_DEEP=0/1 forces 4 / 8.
static bool march_deep(const vrt_scene* sc, int resmode) {
    static int c = -2;
    if (c == -2) c = env_int("VRT_SPEC_DEEP", -1);
    if (c >= 0) return c != 0;
    return resmode != 2 || march_big_scene(sc);
}

// lanes that wait for the HIT / ENDED body before a wave leaves the march loop for it, and the march iterations per
// pass at most while anything waits (VRT_T_HIT, VRT_T_END, VRT_MAX_ITERS override; scheduling only, never a result).
// Measured optima (tools/sweep.py): big scene 24 / 24 / 5; otherwise 24 / 32 / 3 for launches of more than 2^22 rays
// (config 3) and 32 / 40 / 4 for small ones (config 2, a 1/8 share of config 3), whose tail matters more.
static void march_policy(bool big_scene, int64_t n_rays, int32_t& t_hit, int32_t& t_end, int32_t& max_iters) {
    static int h = -1, e = -1, m = -1;
    if (h < 0) {
        h = env_int("VRT_T_HIT", 0);
        e = env_int("VRT_T_END", 0);
        m = env_int("VRT_MAX_ITERS", 0);
    }
    const bool small = n_rays <= ((int64_t)1 << 22);
    t_hit = h > 0 ? h : (big_scene || !small ? 24 : 32);
    t_end = e > 0 ? e : (big_scene ? 24 : (small ? 40 : 32));
    max_iters = m > 0 ? m : (big_scene ? 5 : (small ? 4 : 3));
    if (t_hit > 64) t_hit = 64;
    if (t_end > 64) t_end = 64;
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

int vrt_pow_memo_create(double falloff) {
    if (memo_per_frame()) return VRT_OK;
    const double y = 1 + falloff;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_memo_mu);
    for (int i = 0; i < g_n_memos; i++)
        if (g_memos[i].dev == dev && g_memos[i].y == y) return VRT_OK;
    if (g_n_memos >= MAX_MEMOS) return VRT_ERR_WORKSPACE;
    unsigned long long* buf = nullptr;
    HIP_TRY(hipMalloc((void**)&buf, 2 * VRT_PW_SLOTS * 8));
    hipError_t e = hipMemset(buf, 0, 2 * VRT_PW_SLOTS * 8);  // synchronous: visible to every stream
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(buf);
        g_last_hip_error = (int)e;
        return VRT_ERR_HIP;
    }
    g_memos[g_n_memos++] = PowMemo{dev, y, buf};
    return VRT_OK;
}

int vrt_release_caches(void) {
    std::lock_guard<std::mutex> lock(g_memo_mu);
    int rc = VRT_OK;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (int i = 0; i < g_n_memos; i++) {
        // kernels on any stream of the owning device may still read the table
        if (hipSetDevice(g_memos[i].dev) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipFree(g_memos[i].buf) != hipSuccess)
            rc = VRT_ERR_HIP;
    }
    if (have_cur) (void)hipSetDevice(cur);
    g_n_memos = 0;
    return rc;
}

static int64_t world_table_words(const int32_t* dims, int32_t cs) {
    int64_t words = 0, cells = 1;
    for (int a = 0; a < 3; a++) {
        if (dims[a] <= 0) return -1;
        words += (int64_t)dims[a] * cs + 2 * VRT_WT_GUARD;
        cells *= dims[a];
    }
    if (cells >= (1 << 24) || cells * cs * cs * cs > ((int64_t)1 << 30) || words * 4 > VRT_WT_LDS_MAX) return 0;
    return words;
}

int vrt_world_tables_bytes(const int32_t* dims, int32_t cs, int64_t* bytes) {
    if (!dims || !bytes || cs < 8 || (cs & (cs - 1)) || cs > 256) return VRT_ERR_ARG;
    const int64_t w = world_table_words(dims, cs);
    if (w < 0) return VRT_ERR_ARG;
    *bytes = w * 4;
    return VRT_OK;
}

int vrt_world_tables_build(const int32_t* dims, int32_t cs, uint32_t* d_tables, int64_t bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int64_t need = 0;
    int rc = vrt_world_tables_bytes(dims, cs, &need);
    if (rc != VRT_OK) return rc;
    if (need == 0 || !d_tables) return VRT_ERR_ARG;
    if (bytes < need) return VRT_ERR_WORKSPACE;
    int shift = 0;
    while ((1 << shift) < cs) shift++;
    hipLaunchKernelGGL(world_tables_kernel, dim3((unsigned)grid_for(need / 4)), dim3(VRT_BLOCK), 0, stream, (int)dims[0], (int)dims[1],
                       (int)dims[2], (int)cs, shift, d_tables);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int vrt_occupancy_build(const uint8_t* d_voxels, int64_t n_bytes, uint64_t* d_occupancy, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_bytes < 0 || (n_bytes & 63) || (n_bytes > 0 && (!d_voxels || !d_occupancy))) return VRT_ERR_ARG;
    if (n_bytes == 0) return VRT_OK;
    const int64_t words = n_bytes / 64;
    if (words * 4 > (int64_t)0x7fffffff * VRT_BLOCK) return VRT_ERR_ARG;
    hipLaunchKernelGGL(occupancy_kernel, dim3((unsigned)grid_for(words * 4)), dim3(VRT_BLOCK), 0, stream, d_voxels, words,
                       d_occupancy);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
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
    h.full_frame = n_px == (int64_t)st->width * st->height ? 1 : 0;
    h.pad = 0;
    HIP_TRY(hipMemcpyAsync(hdr, &h, sizeof h, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemsetAsync(bitmap, 0, (size_t)words * 4, stream));
    TileGeom g;
    g.pixels = d_pixels_xy;
    g.n_px = n_px;
    g.smax = smax;
    if (n_px > 0) hipLaunchKernelGGL(plan_mark_kernel, dim3(grid_for(n_px)), dim3(VRT_BLOCK), 0, stream, *st, g, bitmap, hdr);
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
    int64_t rays;   // ray slots of the tile
    int64_t batch;  // rays per march launch
    int64_t slow_cap;
    int64_t full_cap;  // rays per launch that may be re-traced a second time, with a D_FULL_DEV-draw row
    int64_t off_table, off_slow, off_full, off_tab, off_rgba, off_list, off_list_full, off_count, off_pow, total;
};
// external: VRT_WS_DRAW_TABLE / VRT_WS_RAY_TABLE bits -- tables the caller passes to vrt_render_tile need no room here
static WsLayout ws_layout(const vrt_settings* st, int64_t n_px, int64_t n_distinct, int fast_draws, int external) {
    WsLayout w;
    w.rays = n_px * vrt_max_samples(st);
    w.batch = w.rays < batch_rays() ? w.rays : batch_rays();
    if (w.batch < 1) w.batch = 1;
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o += align256(bytes); return r; };
    w.off_table = take((external & VRT_WS_DRAW_TABLE) ? 0 : (n_distinct > 0 ? n_distinct : 1) * (int64_t)fast_draws * 8);
    w.slow_cap = w.batch / 64;
    if (w.slow_cap < SLOW_CAP_MIN) w.slow_cap = w.batch < SLOW_CAP_MIN ? w.batch : SLOW_CAP_MIN;
    if (w.slow_cap > SLOW_CAP_MAX) w.slow_cap = SLOW_CAP_MAX;
    w.off_slow = take(w.slow_cap * VRT_SLOW_STRIDE * 8);
    w.full_cap = w.slow_cap < FULL_CAP ? w.slow_cap : FULL_CAP;
    w.off_full = take(w.full_cap * D_FULL_DEV * 8);
    const int64_t tab_records = ray_table_per_pixel(*st) ? n_px : w.rays;
    w.off_tab = take((external & VRT_WS_RAY_TABLE) ? 0 : (tab_records > 0 ? tab_records : 1) * 8 * VRT_RAY_WORDS);
    w.off_rgba = take(w.rays * 4);
    w.off_list = take(w.slow_cap * 4);
    w.off_list_full = take(w.full_cap * 4);
    w.off_count = take(256 + 8 * 128);  // retrace counts, launch-wide ray counters | the tiled hand-out's eight heads
    w.off_pow = take(2 * VRT_PW_SLOTS * 8);
    w.total = o;
    return w;
}

static int fast_draws_ok(int32_t d) { return d == 32 || d == 64; }

int vrt_workspace_bytes(const vrt_settings* st, int64_t n_px, int64_t n_distinct, int32_t fast_draws, int32_t external,
                        int64_t* bytes) {
    if (check_settings(st) != VRT_OK || n_px < 0 || n_distinct < 0 || !bytes || !fast_draws_ok(fast_draws)) return VRT_ERR_ARG;
    if (external & ~(VRT_WS_DRAW_TABLE | VRT_WS_RAY_TABLE)) return VRT_ERR_ARG;
    *bytes = ws_layout(st, n_px, n_distinct, fast_draws, external).total;
    return VRT_OK;
}

static inline bool within(double v, double lim) { return __builtin_fabs(v) <= lim; }  // false for NaN

static int fill_params(MarchParams& P, const vrt_scene* sc, const vrt_settings* st, const vrt_camera* cam,
                       const vrt_traversed* trav, uint64_t* d_stats) {
    if (!sc || !cam || check_settings(st) != VRT_OK) return VRT_ERR_ARG;
    if (sc->chunk_size != st->chunk_size || sc->n_materials < 0 || sc->n_materials > 255) return VRT_ERR_ARG;
    if (!sc->d_chunk_table || (sc->n_slots > 0 && !sc->d_voxels) || (sc->n_materials > 0 && !sc->d_materials))
        return VRT_ERR_ARG;
    if (sc->n_slots < 0 || sc->n_slots >= (1 << 24)) return VRT_ERR_ARG;
    if (!d_stats) return VRT_ERR_ARG;
    // The march keeps 4 * floor(pos) in 32-bit integers: the camera and everything a ray can reach must stay inside
    // +-2^28.  |vel|_inf <= 8 |rot|^2 + 1 for the reference's (not norm-preserving) quaternion product applied to a
    // unit lens quaternion (lib.py:353-358, 372-376); a ray travels at most dist_max - dist_min plus one void-skip step.
    {
        double q2 = 0;
        for (int a = 0; a < 4; a++) {
            if (!within(cam->rot[a], 1e3)) return VRT_ERR_ARG;
            q2 += cam->rot[a] * cam->rot[a];
        }
        if (!within(cam->lens, 1e6) || !within(st->dist_min, 0x1p28) || !within(st->dist_max, 0x1p28)) return VRT_ERR_ARG;
        const double vbound = 8 * q2 + 1;
        const double reach = (__builtin_fabs(st->dist_max) + __builtin_fabs(st->dist_min) + 2.0 * st->chunk_size + 2.0) * vbound;
        for (int a = 0; a < 3; a++)
            if (!(__builtin_fabs(cam->pos[a]) + reach < 0x1p28)) return VRT_ERR_ARG;
    }
    P.st = *st;
    P.cam = *cam;
    int shift = 0;
    while ((1 << shift) < st->chunk_size) shift++;
    P.cs = st->chunk_size;
    P.cs_shift = shift;
    // the march addresses the voxel bytes with 32-bit offsets (raw buffer loads): at most 4 GiB of voxels per scene
    if (((int64_t)sc->n_slots << (3 * shift)) > (int64_t)0xffffffffll) return VRT_ERR_ARG;
    P.vox_bytes = (uint32_t)((int64_t)sc->n_slots << (3 * shift));
    int64_t cells = 1;
    for (int a = 0; a < 3; a++) {
        if (sc->dims[a] <= 0 || (sc->origin[a] % st->chunk_size) != 0) return VRT_ERR_ARG;
        if (sc->origin[a] < -(1ll << 28) || sc->origin[a] + (int64_t)sc->dims[a] * st->chunk_size > (1ll << 28)) return VRT_ERR_ARG;
        P.origin_c[a] = (int32_t)(sc->origin[a] / st->chunk_size);
        P.t_origin_c[a] = 0;
        P.dims[a] = sc->dims[a];
        cells *= sc->dims[a];
    }
    P.n_materials = sc->n_materials;
    P.chunk_table = sc->d_chunk_table;
    P.voxels = sc->d_voxels;
    P.occ = sc->d_occupancy;
    P.materials = sc->d_materials;
    P.ct_cells = cells <= VRT_CT_LDS_MAX ? (int32_t)cells : 0;
    if ((sc->flags & VRT_SCENE_TABLE_IS_IDENTITY) && (int64_t)sc->n_slots != cells) return VRT_ERR_ARG;
    P.ct_identity = (sc->flags & VRT_SCENE_TABLE_IS_IDENTITY) ? 1 : 0;
    if ((sc->flags & VRT_SCENE_LAYOUT_DENSE) && (int64_t)sc->n_slots != cells) return VRT_ERR_ARG;
    // world-axis offset tables (march_step_w): blocks in table order, at most 2^30 bytes of them (the guard entries' 2^30
    // must lie beyond the buffer, and three of them must not wrap), tables of at most VRT_WT_LDS_MAX bytes.  VRT_WADDR=1
    // turns the look-ahead across chunk borders on: measured slower than stopping at them, kept as a
    // parity-tested variant
    P.wt_on = 0;
    P.wt_lds_off = 0;
    {
        int64_t entries = 0;
        for (int a = 0; a < 3; a++) {
            P.wt_cells[a] = (int32_t)((int64_t)sc->dims[a] * st->chunk_size);
            P.wt_min[a] = (int32_t)sc->origin[a];
            entries += (int64_t)sc->dims[a] * st->chunk_size + 2 * VRT_WT_GUARD;
        }
        // (read at every launch, like VRT_POOL: the parity tests run every case with and without it in one process)
        const int waddr = env_int("VRT_WADDR", VRT_WADDR_DEFAULT);
        P.wt_words = (int32_t)entries;
        P.wt_table = sc->d_world_tables;
        if (waddr && sc->d_world_tables && (sc->flags & (VRT_SCENE_LAYOUT_DENSE | VRT_SCENE_TABLE_IS_IDENTITY)) && sc->n_slots > 0 &&
            (int64_t)P.vox_bytes <= ((int64_t)1 << 30) && entries * 4 <= VRT_WT_LDS_MAX)
            P.wt_on = 1;
    }
    P.t_keys = nullptr;
    P.trav_words = 0;
    P.bm_window = -1;
    for (int a = 0; a < 3; a++) P.t_dims[a] = 0;
    if (trav && trav->d_keys) {
        int64_t tcells = 1;
        for (int a = 0; a < 3; a++) {
            if (trav->dims[a] <= 0 || (trav->origin[a] % st->chunk_size) != 0) return VRT_ERR_ARG;
            if (trav->origin[a] < -(1ll << 28) || trav->origin[a] + (int64_t)trav->dims[a] * st->chunk_size > (1ll << 28)) return VRT_ERR_ARG;
            P.t_origin_c[a] = (int32_t)(trav->origin[a] / st->chunk_size);
            P.t_dims[a] = trav->dims[a];
            tcells *= trav->dims[a];
        }
        if (tcells >= (1ll << 31)) return VRT_ERR_ARG;
        P.t_keys = trav->d_keys;
        const int trav_lds = env_int("VRT_TRAV_LDS", 1);  // (read at every launch: the parity tests render with and without)
        // the bitmap must leave room for VRT_WAVES_PER_SIMD workgroups per CU (160 KiB of LDS, 2 KiB of margin per
        // workgroup); the kernel's static LDS is about 10 KiB (march_pool_kernel, which also keeps its ray pools there,
        // checks its occupancy at the launch and gives the bitmap up if it must)
        const int64_t words = (tcells + 31) / 32;
        int64_t wt_bytes = 0;
        if (P.wt_on)
            for (int a = 0; a < 3; a++) wt_bytes += ((int64_t)P.wt_cells[a] + 2 * VRT_WT_GUARD) * 4;
        const int64_t room = 160 * 1024 / VRT_WAVES_PER_SIMD - 2 * 1024 -
                             (int64_t)(P.wt_on ? sizeof(MarchSharedT<true>) : sizeof(MarchShared)) -
                             (int64_t)sc->n_materials * 64 - (int64_t)P.ct_cells * 4 - wt_bytes - 64;
        P.bm_window = -1;
        const int win_env = env_int("VRT_TRAV_WINDOW", 1);  // (0: no bitmap for large boxes; 2: the window for every box of 32^3 cells and more -- tests)
        if (trav_lds && win_env != 2 && tcells <= VRT_TRAV_LDS_MAX && words * 4 <= room) {
            P.trav_words = (int32_t)words;
        } else if (trav_lds && win_env != 0 && 32 * 32 * 32 / 8 <= room && trav->dims[0] >= 32 && trav->dims[1] >= 32 &&
                   trav->dims[2] >= 32 && trav->dims[0] < 1024 + 32 && trav->dims[1] < 1024 + 32 && trav->dims[2] < 1024 + 32) {
            // The box is sized for the rays' reach, the visits fall where the rays are: around the camera.  A bitmap over the
            // 32^3 cells centred on the camera's (4 KiB) settles nearly all of them (config 5: recording `traversed` cost 14 %
            // of the march when every visit read its cell's key).
            int w = 0;
            for (int a = 0; a < 3; a++) {
                const double p = cam->pos[a];
                int64_t c = (int64_t)__builtin_floor(p / st->chunk_size) - P.t_origin_c[a] - 16;
                c = c < 0 ? 0 : (c > trav->dims[a] - 32 ? trav->dims[a] - 32 : c);
                w |= (int)c << (10 * a);
            }
            P.bm_window = w;
            P.trav_words = 32 * 32 * 32 / 32;
        }
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
    P.plan_hdr = nullptr;
    P.tile_heads = nullptr;
    P.tile_px_log2 = P.tile_log2 = P.tile_ntx = P.tile_nty = 0;
    P.retrace_cap = 0;
    P.list_cap = 0;
    P.chunk = march_chunk(0);  // the launch sites set it for their ray count
    P.first_draw = 0;
    P.brick_lds_off = 0;
    P.pool_lds_off = 0;
    P.pool_swap_min = 1;
    P.pool_refill_min = 1;
    P.pool_keep = 64;
    P.prefix_draws = 0;
    P.list_seed = 0;
    P.per_pixel = 0;
    P.lens = 0.0;
    march_policy(march_big_scene(sc), 0, P.t_hit, P.t_end, P.max_iters);  // the launch sites set it for their ray count
    return VRT_OK;
}

}  // extern "C"

// lookup variant (see march_kernel): 0 unless VRT_LOOKUP says otherwise
static int lookup_mode() {
    static int m = -1;
    if (m < 0) {
        m = env_int("VRT_LOOKUP", 0);
        if (m < 0 || m > 2) m = 0;
    }
    return m;
}
// the ray pool (march_pool_kernel): VRT_POOL=0|1; its thresholds VRT_POOL_T_HIT / VRT_POOL_T_END (rays of the wave's pool
// that wait for the HIT / ENDED body before a pass runs it), VRT_POOL_SWAP_MIN, VRT_POOL_REFILL_MIN (MarchParams)
#ifndef VRT_POOL_DEFAULT
#define VRT_POOL_DEFAULT 1
#endif
// (VRT_POOL and VRT_POOL_MIN_RAYS are read at every launch, not once per process: the parity tests run every case with
// both kernels in one process)
static bool march_pool() { return env_int("VRT_POOL", VRT_POOL_DEFAULT) != 0; }
// Which launches without a settled bitmap run the instances that compare a traversed key after the voxel reads went out
// (DEFER): VRT_DEFER_VISIT=1 (default) those over scenes far larger than the caches, 0 none, 2 all of them (the parity
// tests).  Read at every launch; scheduling only, never a result.
static bool march_defer(const MarchParams& P) {
    const int m = env_int("VRT_DEFER_VISIT", 1);
    return m >= 2 || (m == 1 && (int64_t)P.vox_bytes > ((int64_t)512 << 20));
}
// Measured optima on MI355X (tools/sweep_pool.py, same-run comparisons in profiles/r03_pool_sweep.md): config 3
// 40 / 60 / 8 / 8 / 40 / 3 (march 5.45 ms against 6.12 ms for march_kernel), config 5 48 / 32 / 4 / 8 / 40 / 5 (263.7 against
// 285.5 ms); VRT_POOL_T_HIT, _T_END, _SWAP_MIN, _REFILL_MIN, _KEEP, _ITERS override (scheduling only, never a result)
static void pool_policy(MarchParams& P, bool big_scene) {
    static int h = -2, e, sw, rf, kp, it;
    if (h == -2) {
        h = env_int("VRT_POOL_T_HIT", -1);
        e = env_int("VRT_POOL_T_END", -1);
        sw = env_int("VRT_POOL_SWAP_MIN", -1);
        rf = env_int("VRT_POOL_REFILL_MIN", -1);
        kp = env_int("VRT_POOL_KEEP", -1);
        it = env_int("VRT_POOL_ITERS", -1);
    }
    const int cap = 64 + VRT_POOL_SLOTS;
    auto pick = [](int env, int dflt, int lo, int hi) { const int v = env >= 0 ? env : dflt; return v < lo ? lo : (v > hi ? hi : v); };
    P.t_hit = pick(h, big_scene ? 48 : 40, 1, cap);
    P.t_end = pick(e, big_scene ? 32 : 60, 1, cap);
    P.pool_swap_min = pick(sw, big_scene ? 4 : 8, 1, 64);
    P.pool_refill_min = pick(rf, 8, 1, 64);
    P.pool_keep = pick(kp, 40, 1, 64);
    P.max_iters = pick(it, big_scene ? 5 : 3, 1, 64);
}
// dynamic LDS of a march launch: materials | chunk table | settled bitmap [| brick slots of lookup variant 2 | ray pools]
static inline size_t march_lds(MarchParams& P, bool bricks, bool pool) {
    P.snap_flags = (P.trav_words != 0 ? CF_HAS_BM | (P.bm_window >= 0 ? CF_BM_WINDOW : 0) : 0) | (P.ct_cells != 0 ? CF_CT_LDS : 0) | (P.ct_identity != 0 ? CF_CT_IDENTITY : 0) |
                   (P.t_keys != nullptr ? CF_HAS_KEYS : 0);
    size_t n = (size_t)P.n_materials * 64 + (size_t)P.ct_cells * 4 + (size_t)P.trav_words * 4;
    n = (n + 15) & ~(size_t)15;
    P.wt_lds_off = (int32_t)n;
    if (P.wt_on)
        for (int a = 0; a < 3; a++) n += ((size_t)P.wt_cells[a] + 2 * VRT_WT_GUARD) * 4;
    P.brick_lds_off = (int32_t)n;
    if (bricks) n += (size_t)VRT_BLOCK * 9 * 8;
    P.pool_lds_off = (int32_t)n;
    if (pool) n += (size_t)(VRT_BLOCK / VRT_WAVE) * VRT_POOL_WAVE_BYTES;
    return n + 16;
}
// workgroups of march_pool_kernel a CU holds with `dyn` bytes of dynamic LDS (the runtime's own occupancy calculation)
static int pool_blocks_per_cu(size_t dyn, bool w) {
    static std::mutex mu;
    static size_t seen_dyn[16];
    static int seen_n[16], n_seen = 0;
    const size_t key = dyn * 2 + (w ? 1 : 0);
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < n_seen; i++)
        if (seen_dyn[i] == key) return seen_n[i];
    int nb = 0;
    hipError_t e = w ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, march_pool_kernel<8, 1, 0, true>, VRT_BLOCK, dyn)
                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, march_pool_kernel<VRT_SPEC_DEEP, 1>, VRT_BLOCK, dyn);
    if (e != hipSuccess) nb = 0;
    if (env_int("VRT_POOL_VERBOSE", 0))
        fprintf(stderr, "vrt: march_pool_kernel%s: %zu + %zu bytes of LDS per workgroup -> %d workgroups per CU\n", w ? " (look-ahead)" : "",
                w ? sizeof(MarchSharedT<true>) : sizeof(MarchShared), dyn, nb);
    if (n_seen < 16) {
        seen_dyn[n_seen] = key;
        seen_n[n_seen++] = nb;
    }
    return nb;
}
// Does the frame's march of this launch use the ray pool?  Only if VRT_WAVES_PER_SIMD workgroups still fit a CU with the
// pools in LDS -- without the settled bitmap if need be (P.trav_words is cleared then).
static bool pool_plan(MarchParams& P) {
    if (!march_pool() || lookup_mode() != 0) return false;
    // a wave's pool holds up to 64 + VRT_POOL_SLOTS rays when the launch runs out of new ones, and drains alone: small
    // launches lose more in that tail than the fuller lanes win (config 2, 2 M rays: 0.70 against 0.59 ms; a 1/8 share of
    // config 3, 7.8 M rays: 0.94 against 0.99 ms; a 1/4 share: 1.52 against 1.71 ms) (VRT_POOL_MIN_RAYS)
    if (P.n < (int64_t)env_int("VRT_POOL_MIN_RAYS", 5 << 20)) return false;
    const int32_t words = P.trav_words, wt = P.wt_on;
    for (int keep_wt = wt; keep_wt >= 0; keep_wt--) {  // (the world-axis tables are given up before the pool is)
        P.wt_on = keep_wt;
        P.trav_words = words;
        if (pool_blocks_per_cu(march_lds(P, false, true), keep_wt != 0) >= VRT_WAVES_PER_SIMD) return true;
        P.trav_words = 0;
        if (pool_blocks_per_cu(march_lds(P, false, true), keep_wt != 0) >= VRT_WAVES_PER_SIMD) return true;
    }
    P.trav_words = words;
    P.wt_on = wt;
    return false;
}

// kernel variant: resolution mode from vrt_scene.max_resolution, speculation depth from the scene size
template <bool RECORD, bool LIST>
static int launch_march(MarchParams P, int grid, int resmode, bool deep, bool pool, hipStream_t stream) {
    if (RECORD || LIST) {  // debug records / re-traces: one generic variant
        P.wt_on = 0;
        // A re-trace launch hands its rays out in the order the frame's march appended them to the list, not in increasing
        // ray order: the smallest index a wave holds says nothing about the rays it will still take, so no cell may be
        // called settled (see trav_cell) -- every visit compares its key with the cell's.
        if (LIST) P.trav_words = 0;
        const size_t lds = march_lds(P, false, false);
        if (LIST && P.list_seed == 1)
            hipLaunchKernelGGL((march_kernel<VRT_SPEC, 2, RECORD, LIST, 0, 4, false, false, LIST ? 1 : 0>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);
        else if (LIST && P.list_seed == 2)
            hipLaunchKernelGGL((march_kernel<VRT_SPEC, 2, RECORD, LIST, 0, 4, false, false, LIST ? 2 : 0>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);
        else
            hipLaunchKernelGGL((march_kernel<VRT_SPEC, 2, RECORD, LIST>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);
        return VRT_OK;
    }
    if (pool) {
        pool_policy(P, (int64_t)P.vox_bytes > ((int64_t)512 << 20));
        const size_t lds = march_lds(P, false, true);
#define VRT_LAUNCH_POOL_W(SPEC_, RES_, W_)                                                                                 \
    do {                                                                                                                   \
        if (P.per_pixel)                                                                                                   \
            hipLaunchKernelGGL((march_pool_kernel<SPEC_, RES_, 1, W_>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);       \
        else                                                                                                               \
            hipLaunchKernelGGL((march_pool_kernel<SPEC_, RES_, 0, W_>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);       \
    } while (0)
#define VRT_LAUNCH_POOL(SPEC_, RES_) VRT_LAUNCH_POOL_W(SPEC_, RES_, false)
        // no settled bitmap, but keys to record (config 5: the box is too large for one): the DEFER instances (8 positions,
        // resolutions <= 2)
        // -- for scenes far larger than the caches only, where a step's voxel reads are misses worth overlapping with: config 5
        // 233.8 against 244.3 ms; config 3, whose pools leave the bitmap no room either, 5.39 against 5.33 ms with it
        // (a bitmap over the cells around the camera only -- boxes too large for their own -- goes with them)
        const bool defer = !P.wt_on && deep && VRT_SPEC_DEEP == 8 && resmode != 2 && P.t_keys && (P.trav_words == 0 || P.bm_window >= 0) && march_defer(P);
        // (the tiled hand-out: instances of the DEFER kernels with a ray table)
        if (!defer || P.per_pixel == 2) P.tile_heads = nullptr;
        if (P.per_pixel == 2) {  // no ray table (take_ray, PERPIX 3): vrt_render_tile asks for this with 8 positions only
            if (P.wt_on || !deep || VRT_SPEC_DEEP != 8 || resmode == 2) return VRT_ERR_ARG;
            if (defer && resmode == 0) hipLaunchKernelGGL((march_pool_kernel<8, 0, 3, false, true>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);
            else if (defer) hipLaunchKernelGGL((march_pool_kernel<8, 1, 3, false, true>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);
            else if (resmode == 0) hipLaunchKernelGGL((march_pool_kernel<8, 0, 3>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);
            else hipLaunchKernelGGL((march_pool_kernel<8, 1, 3>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);
        } else if (P.wt_on) {  // (only with 8 positions and resolutions <= 2: march_wt_ok)
            if (resmode == 0) VRT_LAUNCH_POOL_W(8, 0, true);
            else VRT_LAUNCH_POOL_W(8, 1, true);
        } else if (defer) {
#define VRT_LAUNCH_POOL_D(RES_)                                                                                                    \
    do {                                                                                                                         \
        if (P.tile_heads && P.per_pixel)                                                                                         \
            hipLaunchKernelGGL((march_pool_kernel<8, RES_, 1, false, true, true>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);   \
        else if (P.tile_heads)                                                                                                   \
            hipLaunchKernelGGL((march_pool_kernel<8, RES_, 0, false, true, true>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);   \
        else if (P.per_pixel)                                                                                                    \
            hipLaunchKernelGGL((march_pool_kernel<8, RES_, 1, false, true>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);         \
        else                                                                                                                     \
            hipLaunchKernelGGL((march_pool_kernel<8, RES_, 0, false, true>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);         \
    } while (0)
            if (resmode == 0) VRT_LAUNCH_POOL_D(0);
            else VRT_LAUNCH_POOL_D(1);
#undef VRT_LAUNCH_POOL_D
        } else if (deep) {
            if (resmode == 0) VRT_LAUNCH_POOL(VRT_SPEC_DEEP, 0);
            else if (resmode == 1) VRT_LAUNCH_POOL(VRT_SPEC_DEEP, 1);
            else VRT_LAUNCH_POOL(VRT_SPEC_DEEP, 2);
        } else {
            if (resmode == 0) VRT_LAUNCH_POOL(VRT_SPEC, 0);
            else if (resmode == 1) VRT_LAUNCH_POOL(VRT_SPEC, 1);
            else VRT_LAUNCH_POOL(VRT_SPEC, 2);
        }
#undef VRT_LAUNCH_POOL
#undef VRT_LAUNCH_POOL_W
        return VRT_OK;
    }
    const int lk = lookup_mode();
    if (lk != 0 && (!P.occ || resmode == 2)) return VRT_ERR_ARG;  // the measurement variants exist for resolutions <= 2
    // One ray per lane: where the key comparison can go behind the voxel reads (the DEFER instances) a bitmap over the cells
    // around the camera only is not worth its upkeep -- config 5 261.8 against 254.4 ms (the ray pool: 232.1 against 233.7)
    if (P.bm_window >= 0 && !P.wt_on && lk == 0 && deep && VRT_SPEC_DEEP == 8 && resmode != 2 && P.t_keys && march_defer(P))
        P.trav_words = 0;  // (march_kernel's DEFER instances have no code for one)
    const size_t lds = march_lds(P, lk == 2, false);
    if (P.wt_on) {  // (only with 8 positions, resolutions <= 2 and the byte lookup: march_wt_ok)
#define VRT_LAUNCH_W(RES_)                                                                                                          \
    do {                                                                                                                            \
        if (P.per_pixel)                                                                                                            \
            hipLaunchKernelGGL((march_kernel<8, RES_, false, false, 0, 2, true>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);       \
        else                                                                                                                        \
            hipLaunchKernelGGL((march_kernel<8, RES_, false, false, 0, 0, true>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);       \
    } while (0)
        if (resmode == 0) VRT_LAUNCH_W(0);
        else VRT_LAUNCH_W(1);
#undef VRT_LAUNCH_W
        return VRT_OK;
    }
    if (P.per_pixel == 2) {  // no ray table (take_ray, PERPIX 3): vrt_render_tile asks for this with 8 positions only
        if (lk != 0 || !deep || VRT_SPEC_DEEP != 8 || resmode == 2) return VRT_ERR_ARG;
        const bool defer = P.t_keys && (P.trav_words == 0 || P.bm_window >= 0) && march_defer(P);
        if (defer && resmode == 0) hipLaunchKernelGGL((march_kernel<8, 0, false, false, 0, 3, false, true>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);
        else if (defer) hipLaunchKernelGGL((march_kernel<8, 1, false, false, 0, 3, false, true>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);
        else if (resmode == 0) hipLaunchKernelGGL((march_kernel<8, 0, false, false, 0, 3>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);
        else hipLaunchKernelGGL((march_kernel<8, 1, false, false, 0, 3>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);
        return VRT_OK;
    }
    if (lk == 0 && deep && VRT_SPEC_DEEP == 8 && resmode != 2 && P.t_keys && (P.trav_words == 0 || P.bm_window >= 0) && march_defer(P)) {  // (see the pool's)
#define VRT_LAUNCH_D(RES_)                                                                                                                \
    do {                                                                                                                                  \
        if (P.per_pixel)                                                                                                                  \
            hipLaunchKernelGGL((march_kernel<8, RES_, false, false, 0, 2, false, true>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);      \
        else                                                                                                                              \
            hipLaunchKernelGGL((march_kernel<8, RES_, false, false, 0, 0, false, true>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P);      \
    } while (0)
        if (resmode == 0) VRT_LAUNCH_D(0);
        else VRT_LAUNCH_D(1);
#undef VRT_LAUNCH_D
        return VRT_OK;
    }
#define VRT_LAUNCH(SPEC_, RES_, LK_)                                                                                        \
    do {                                                                                                                    \
        if (P.per_pixel)                                                                                                    \
            hipLaunchKernelGGL((march_kernel<SPEC_, RES_, false, false, LK_, 2>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P); \
        else                                                                                                                \
            hipLaunchKernelGGL((march_kernel<SPEC_, RES_, false, false, LK_, 0>), dim3(grid), dim3(VRT_BLOCK), lds, stream, P); \
    } while (0)
#define VRT_LAUNCH_LK(SPEC_, RES_)                \
    do {                                          \
        if (lk == 0) VRT_LAUNCH(SPEC_, RES_, 0);  \
        else if (lk == 1) VRT_LAUNCH(SPEC_, RES_, 1); \
        else VRT_LAUNCH(SPEC_, RES_, 2);          \
    } while (0)
    if (deep) {
        if (resmode == 0) VRT_LAUNCH_LK(VRT_SPEC_DEEP, 0);
        else if (resmode == 1) VRT_LAUNCH_LK(VRT_SPEC_DEEP, 1);
        else VRT_LAUNCH(VRT_SPEC_DEEP, 2, 0);
    } else {
        if (resmode == 0) VRT_LAUNCH_LK(VRT_SPEC, 0);
        else if (resmode == 1) VRT_LAUNCH_LK(VRT_SPEC, 1);
        else VRT_LAUNCH(VRT_SPEC, 2, 0);
    }
#undef VRT_LAUNCH_LK
#undef VRT_LAUNCH
    return VRT_OK;
}
// may this frame's march look ahead across chunk borders (march_step_w)?  fill_params has checked the scene's layout and
// size; the kernel exists for 8 positions, resolutions <= 2 and the byte lookup
static inline bool march_wt_ok(const MarchParams& P, int resmode, bool deep) {
    return P.wt_on && resmode != 2 && deep && VRT_SPEC_DEEP == 8 && lookup_mode() == 0;
}
static inline int res_mode(const vrt_scene* sc) {
    static int force = -2;
    if (force == -2) force = env_int("VRT_RESMODE", -1);
    if (force >= 0 && force <= 2) return force == 2 ? 2 : (sc->max_resolution >= 1 && sc->max_resolution <= force + 1 ? force : 2);
    return sc->max_resolution == 1 ? 0 : (sc->max_resolution == 2 ? 1 : 2);
}

extern "C" {

int vrt_draw_table_bytes(int64_t n_distinct, int32_t fast_draws, int64_t* bytes) {
    if (n_distinct < 0 || !bytes || !fast_draws_ok(fast_draws)) return VRT_ERR_ARG;
    *bytes = align256((n_distinct > 0 ? n_distinct : 1) * (int64_t)fast_draws * 8);
    return VRT_OK;
}

// seeds the draw rows of a tile: one per distinct seed of the plan (static), one per ray slot (non-static)
static int seed_draw_table(const vrt_settings* st, const TileGeom& g, const void* d_plan, int64_t n_distinct, int32_t fast_draws,
                           double* d_table, hipStream_t stream) {
    const int64_t rays = g.n_px * g.smax;
    if (st->seed_nonce) {
        if (n_distinct != rays) return VRT_ERR_ARG;
        if (rays > 0)
            hipLaunchKernelGGL(rng_slots_kernel, dim3(grid_for(rays)), dim3(VRT_BLOCK), 0, stream, *st, g, (int)fast_draws, d_table);
    } else {
        if (n_distinct > rays) return VRT_ERR_ARG;
        const uint32_t* seed_list = (const uint32_t*)((const char*)d_plan + 64);
        if (n_distinct > 0)
            hipLaunchKernelGGL(rng_plan_kernel, dim3(grid_for(n_distinct)), dim3(VRT_BLOCK), 0, stream, seed_list, n_distinct,
                               (uint64_t)0, (int)fast_draws, d_table);
    }
    return VRT_OK;
}

int vrt_draw_table_build(const vrt_settings* st, const int32_t* d_pixels_xy, int64_t n_px, const void* d_plan,
                         int64_t n_distinct, int32_t fast_draws, double* d_table, int64_t table_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (check_settings(st) != VRT_OK || !plan_supported(st) || n_px < 0 || !d_plan || !d_table) return VRT_ERR_ARG;
    if (n_px > 0 && !d_pixels_xy) return VRT_ERR_ARG;
    int64_t need = 0;
    if (n_distinct < 0 || vrt_draw_table_bytes(n_distinct, fast_draws, &need) != VRT_OK) return VRT_ERR_ARG;
    if (table_bytes < need) return VRT_ERR_WORKSPACE;
    TileGeom g;
    g.pixels = d_pixels_xy;
    g.n_px = n_px;
    g.smax = vrt_max_samples(st);
    ProfScope ps(stream, VRT_PROF_RNG);
    int rc = seed_draw_table(st, g, d_plan, n_distinct, fast_draws, d_table, stream);
    if (rc != VRT_OK) return rc;
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int vrt_ray_table_bytes(const vrt_settings* st, int64_t n_px, int64_t* bytes) {
    if (check_settings(st) != VRT_OK || n_px < 0 || !bytes) return VRT_ERR_ARG;
    const int64_t records = ray_table_per_pixel(*st) ? n_px : n_px * vrt_max_samples(st);
    *bytes = align256((records > 0 ? records : 1) * 8 * VRT_RAY_WORDS);
    return VRT_OK;
}

int vrt_ray_table_build(const vrt_settings* st, double lens, const int32_t* d_pixels_xy, int64_t n_px, const void* d_plan,
                        const double* d_draw_table, int32_t fast_draws, double* d_ray_table, int64_t table_bytes,
                        void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int64_t need = 0;
    if (vrt_ray_table_bytes(st, n_px, &need) != VRT_OK || !plan_supported(st)) return VRT_ERR_ARG;
    if (!d_plan || !d_draw_table || !d_ray_table || (n_px > 0 && !d_pixels_xy) || !fast_draws_ok(fast_draws)) return VRT_ERR_ARG;
    if (!within(lens, 1e6)) return VRT_ERR_ARG;
    if (table_bytes < need) return VRT_ERR_WORKSPACE;
    const int smax = vrt_max_samples(st);
    const int64_t rays = n_px * smax;
    if (rays == 0) return VRT_OK;
    if (rays >= 4294967295ll) return VRT_ERR_ARG;
    TileGeom g;
    g.pixels = d_pixels_xy;
    g.n_px = n_px;
    g.smax = smax;
    const uint32_t* ray_seedidx = (const uint32_t*)((const char*)d_plan + 64 + align256(rays * 4));
    ProfScope ps(stream, VRT_PROF_RAYGEN);
    const int per_pixel = ray_table_per_pixel(*st) ? 1 : 0;
    hipLaunchKernelGGL(raygen_tile_kernel, dim3(grid_for(per_pixel ? n_px : rays)), dim3(VRT_BLOCK), 0, stream, *st, lens, g,
                       ray_seedidx, d_draw_table, (int)fast_draws, ray_tab_at(d_ray_table, rays), per_pixel);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int vrt_render_tile(const vrt_scene* scene, const vrt_settings* st, const vrt_camera* cam, const int32_t* d_pixels_xy,
                    int64_t n_px, const void* d_plan, int64_t n_distinct, int32_t fast_draws,
                    const double* d_draw_table, const double* d_ray_table, void* d_workspace, int64_t workspace_bytes,
                    float* d_rgba_f32, uint8_t* d_image_u8, uint32_t* d_ray_rgba, vrt_ray* d_rays, uint64_t* d_stats,
                    const vrt_traversed* trav, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MarchParams P;
    int rc = fill_params(P, scene, st, cam, trav, d_stats);
    if (rc != VRT_OK) return rc;
    if (n_px < 0 || (n_px > 0 && !d_pixels_xy) || !d_workspace || !d_plan) return VRT_ERR_ARG;
    if (!plan_supported(st)) return VRT_ERR_ARG;
    if (d_ray_table && !d_draw_table) return VRT_ERR_ARG;
    const int smax = vrt_max_samples(st);
    const int64_t rays = n_px * smax;
    if (rays >= 4294967295ll) return VRT_ERR_ARG;
    // the plan header (n_distinct, settings hash) is read back and validated by the caller once, after
    // vrt_plan_build; no host synchronisation happens here
    const char* pl = (const char*)d_plan;
    const uint32_t* ray_seedidx = (const uint32_t*)(pl + 64 + align256(rays * 4));
    if (n_distinct < 0 || n_distinct > rays || !fast_draws_ok(fast_draws)) return VRT_ERR_ARG;
    if (st->seed_nonce && n_distinct != rays) return VRT_ERR_ARG;  // non-static: one row per ray slot
    WsLayout w = ws_layout(st, n_px, n_distinct, fast_draws,
                           (d_draw_table ? VRT_WS_DRAW_TABLE : 0) | (d_ray_table ? VRT_WS_RAY_TABLE : 0));
    if (workspace_bytes < w.total) return VRT_ERR_WORKSPACE;
    char* ws = (char*)d_workspace;
    const double* table = d_draw_table ? d_draw_table : (const double*)(ws + w.off_table);
    double* t_slow = (double*)(ws + w.off_slow);
    uint32_t* rgba = d_ray_rgba ? d_ray_rgba : (uint32_t*)(ws + w.off_rgba);
    uint32_t* list = (uint32_t*)(ws + w.off_list);
    uint32_t* count = (uint32_t*)(ws + w.off_count);
    unsigned long long* pow_global = device_pow_memo(1 + st->falloff);
    const bool frame_memo = pow_global == nullptr;
    if (frame_memo) pow_global = (unsigned long long*)(ws + w.off_pow);
    frame_begin(d_stats, count, n_px > 0 ? 64 + 8 * 32 : 0, pow_global, frame_memo && n_px > 0 ? 4 * VRT_PW_SLOTS : 0, trav, stream);
    if (n_px == 0) return VRT_OK;
    TileGeom g;
    g.pixels = d_pixels_xy;
    g.n_px = n_px;
    g.smax = smax;
    if (!d_draw_table) {  // no table from vrt_draw_table_build: seed this frame's draws into the workspace
        ProfScope ps(stream, VRT_PROF_RNG);
        rc = seed_draw_table(st, g, d_plan, n_distinct, fast_draws, (double*)(ws + w.off_table), stream);
        if (rc != VRT_OK) return rc;
    }
    RayTab tab = ray_tab_at(d_ray_table ? const_cast<double*>(d_ray_table) : (double*)(ws + w.off_tab), rays);
    const int resmode = res_mode(scene);
    const bool deep = march_deep(scene, resmode);
    const bool big_scene = march_big_scene(scene);
    int per_pixel = ray_table_per_pixel(*st) ? 1 : 0;
    // A ray table that would be written now and read once is not written at all: the lanes of the march derive their
    // records from the draw rows (take_ray, PERPIX 3).  One record per pixel is cheap enough to keep its kernel; the
    // variants without a PERPIX 3 instance keep it too.  (VRT_FUSE_RAYGEN=0: always the kernel -- for measurements)
    if (!d_ray_table && !per_pixel && !d_rays && !P.wt_on && deep && VRT_SPEC_DEEP == 8 && resmode != 2 && lookup_mode() == 0 &&
        env_int("VRT_FUSE_RAYGEN", 1) != 0)
        per_pixel = 2;
    if (!d_ray_table && per_pixel != 2) {  // no table from vrt_ray_table_build: lens quaternions + lives of this frame
        ProfScope ps(stream, VRT_PROF_RAYGEN);
        hipLaunchKernelGGL(raygen_tile_kernel, dim3(grid_for(per_pixel ? n_px : rays)), dim3(VRT_BLOCK), 0, stream, *st, cam->lens,
                           g, ray_seedidx, table, (int)fast_draws, tab, per_pixel);
    }
    P.g = g;
    P.ray_seedidx = st->seed_nonce ? nullptr : ray_seedidx;
    P.ray_rgba = rgba;
    P.rays = d_rays;
    P.pow_global = pow_global;
    P.first_draw = 1 + (st->dof != 0.0 ? 2 : 0);
    P.per_pixel = per_pixel;
    P.lens = cam->lens;
    P.tab = tab;
    for (int64_t ray0 = 0; ray0 < rays; ray0 += w.batch) {
        const int64_t n = (rays - ray0) < w.batch ? (rays - ray0) : w.batch;
        if (ray0 != 0) clear_words(count, 256 + 8 * 128, stream);  // retrace counts + the launch-wide ray counters (first batch: frame_begin_kernel)
        P.ray0 = ray0;
        P.n = n;
        P.chunk = march_chunk(n);
        march_policy(big_scene, n, P.t_hit, P.t_end, P.max_iters);
        P.list = nullptr;
        P.list_count = nullptr;
        P.draws = table;
        P.n_draws = fast_draws;
        P.draw_stride = fast_draws;
        P.retrace_list = list;
        P.retrace_count = count;
        P.retrace_cap = (uint32_t)w.slow_cap;
        P.queue_head = (unsigned long long*)(count + 2);
        // (the pool variant may give up the settled bitmap for its LDS: decided on a copy, the re-traces keep theirs)
        MarchParams F = P;
        F.wt_on = !d_rays && march_wt_ok(P, resmode, deep);
        const bool pool = !d_rays && pool_plan(F);
        // Scenes beyond the caches, the whole window in one launch, no settled bitmap (its bound needs rays handed out in
        // increasing order): square pixel tiles in Morton order, an eighth of the window per XCD (tile_ticket).  Whether
        // the list IS the whole window in x-major order the plan's header says, on the device.  A measured variant (config 5:
        // 14 % fewer L2 misses, 0.5 % more time): VRT_TILED=1 asks for it, 2 also over scenes that fit the caches (tests)
        const int tile_env = env_int("VRT_TILED", 0);
        if (pool && (big_scene || tile_env == 2) && F.trav_words == 0 && ray0 == 0 && n == rays && n_px == (int64_t)st->width * st->height &&
            F.chunk > 0 && F.chunk % smax == 0 && st->height % (F.chunk / smax) == 0 && tile_env != 0) {
            const int px = F.chunk / smax;
            const int ntx = (st->width + px - 1) / px, nty = st->height / px;
            int lg = 1, lpx = 0;  // (at least 2 x 2 tiles: eight heads share 4^lg * px tickets)
            while ((1 << lg) < ntx || (1 << lg) < nty) lg++;
            while ((1 << lpx) < px) lpx++;
            if ((1 << lpx) == px && lpx + 2 * lg <= 31 && lpx + 2 * lg >= 3) {  // (a power-of-two tile edge; 32-bit tickets; eight heads)
                F.plan_hdr = (const PlanHeader*)d_plan;
                F.tile_heads = count + 64;
                F.tile_px_log2 = lpx;
                F.tile_log2 = lg;
                F.tile_ntx = ntx;
                F.tile_nty = nty;
            }
        }
        {
            ProfScope ps(stream, VRT_PROF_MARCH);
            rc = d_rays ? launch_march<true, false>(F, march_grid(n), resmode, deep, false, stream)
                        : launch_march<false, false>(F, march_grid(n), resmode, deep, pool, stream);
            if (rc != VRT_OK) return rc;
        }
        // rays that ran out of draws: per-ray 113-draw rows, device-side count (no host sync); the lane that takes a listed ray
        // seeds its row (take_ray, SEED: no launch of its own for that)
        ProfScope ps(stream, VRT_PROF_RETRACE);
        const int rgrid = 256;
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
        P.prefix_draws = pool ? fast_draws : 0;  // (see hit_body)
        P.list_seed = 1;
        if (d_rays) launch_march<true, true>(P, rgrid, resmode, deep, false, stream);
        else launch_march<false, true>(P, rgrid, resmode, deep, false, stream);
        P.prefix_draws = 0;
        // third tier: full-state MT19937, D_FULL_DEV draws per ray; usually empty (the kernel returns at once)
        P.list_seed = 2;
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
        if (d_rays) launch_march<true, true>(P, 64, resmode, deep, false, stream);
        else launch_march<false, true>(P, 64, resmode, deep, false, stream);
        P.list_seed = 0;
    }
    if (d_rgba_f32 || d_image_u8) {
        ProfScope ps(stream, VRT_PROF_RESOLVE);
        // (grid: enough for either mapping -- which one applies is in the plan header, on the device)
        const int64_t tiles = (int64_t)((st->width + 15) / 16) * ((st->height + 15) / 16);
        const int64_t rgrid = grid_for(n_px) > tiles ? grid_for(n_px) : tiles;
        hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)rgrid), dim3(VRT_BLOCK), 0, stream, *st, g, (const PlanHeader*)d_plan, rgba,
                           d_rgba_f32, d_image_u8);
    }
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int vrt_trace_workspace_bytes(int64_t n_rays, int64_t* bytes) {
    if (n_rays < 0 || !bytes) return VRT_ERR_ARG;
    *bytes = align256((n_rays > 0 ? n_rays : 1) * 8 * VRT_RAY_WORDS) + 256 + align256(2 * VRT_PW_SLOTS * 8);
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
    if (n_rays < 0 || n_rays >= 4294967295ll || n_draws < 0 || !d_rays || !d_workspace) return VRT_ERR_ARG;
    if (n_rays > 0 && (!d_dir_x || !d_dir_y || !d_detail || (n_draws > 0 && !d_draws))) return VRT_ERR_ARG;
    if (st->dof != 0.0 && n_draws < 2) return VRT_ERR_ARG;  // the lens jitter alone takes two draws (init.py:41-42)
    int64_t need = 0;
    vrt_trace_workspace_bytes(n_rays, &need);
    if (workspace_bytes < need) return VRT_ERR_WORKSPACE;
    char* tail = (char*)d_workspace + align256(n_rays * 8 * VRT_RAY_WORDS);
    unsigned long long* qh = (unsigned long long*)tail;
    P.pow_global = device_pow_memo(1 + st->falloff);
    const bool frame_memo = P.pow_global == nullptr;
    if (frame_memo) P.pow_global = (unsigned long long*)(tail + 256);
    frame_begin(d_stats, qh, n_rays > 0 ? 64 : 0, P.pow_global, frame_memo && n_rays > 0 ? 4 * VRT_PW_SLOTS : 0, trav, stream);
    if (n_rays == 0) return VRT_OK;
    RayTab tab = ray_tab_at((double*)d_workspace, n_rays);
    hipLaunchKernelGGL(raygen_explicit_kernel, dim3(grid_for(n_rays)), dim3(VRT_BLOCK), 0, stream, *st, cam->lens, d_dir_x, d_dir_y,
                       d_detail, d_draws, (int)n_draws, n_rays, tab);
    P.queue_head = qh;
    P.retrace_cap = 0;
    P.expl_detail = d_detail;
    P.tab = tab;
    P.ray0 = 0;
    P.n = n_rays;
    P.chunk = march_chunk(n_rays);
    P.draws = d_draws;
    P.n_draws = n_draws;
    P.draw_stride = n_draws;
    P.first_draw = (st->dof != 0.0) ? 2 : 0;
    P.rays = d_rays;
    launch_march<true, false>(P, march_grid(n_rays), 2, false, false, stream);
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

#ifdef VRT_DIAG
// diagnostic build only: read and clear the counters of the marches since the last call
int vrt_diag_read(unsigned long long* out, int n) {
    unsigned long long h[DG_N], z[DG_N] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return VRT_ERR_HIP;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_diag), sizeof h) != hipSuccess) return VRT_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, sizeof z) != hipSuccess) return VRT_ERR_HIP;
    for (int j = 0; j < n && j < DG_N; j++) out[j] = h[j];
    unsigned long long t[5], zt[5] = {0};
    if (hipMemcpyFromSymbol(t, HIP_SYMBOL(g_diag_t), sizeof t) != hipSuccess) return VRT_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag_t), zt, sizeof zt) != hipSuccess) return VRT_ERR_HIP;
    for (int j = 0; j < 5 && DG_N + j < n; j++) out[DG_N + j] = t[j];
    return DG_N + 5;
}
#endif

int vrt_profile_begin_kinds(uint32_t kinds) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (auto& e : g_prof) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    g_prof.clear();
    g_prof_kinds = kinds;
    g_prof_on = true;
    return VRT_OK;
}
int vrt_profile_begin(void) { return vrt_profile_begin_kinds(~0u); }

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
                 const int64_t* origin, const int32_t* dims, int32_t cs, const uint32_t* d_chunk_list, int64_t n_list,
                 uint32_t* d_world_table, uint8_t* d_voxels, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_objects < 0 || !origin || !dims || !d_world_table || !d_voxels) return VRT_ERR_ARG;
    if (n_objects > 0 && (!d_objects || !d_models || !d_remap)) return VRT_ERR_ARG;
    if (cs < 8 || cs > 256 || (cs & (cs - 1))) return VRT_ERR_ARG;
    const int64_t n = (int64_t)dims[0] * dims[1] * dims[2];
    if (dims[0] < 1 || dims[1] < 1 || dims[2] < 1 || n > (1 << 24) - 2) return VRT_ERR_ARG;
    if (n_list < 0 || n_list > n || (n_list > 0 && !d_chunk_list)) return VRT_ERR_ARG;
    for (int a = 0; a < 3; a++) {
        if (origin[a] % cs) return VRT_ERR_ARG;
        if (origin[a] < -(1ll << 30) || origin[a] + (int64_t)dims[a] * cs > (1ll << 30)) return VRT_ERR_ARG;
    }
    const int64_t blocks = d_chunk_list ? n_list : n;
    if (blocks == 0) return VRT_OK;
    hipLaunchKernelGGL(voxelize_kernel, dim3((unsigned)blocks), dim3(VRT_BLOCK), 0, stream, d_objects, (int)n_objects, d_models,
                       d_remap, (int)origin[0], (int)origin[1], (int)origin[2], (int)dims[1], (int)dims[2], (int)cs,
                       d_chunk_list, d_world_table, d_voxels);
    HIP_TRY(hipGetLastError());
    return VRT_OK;
}

int vrt_canvas_blit(uint8_t* d_canvas_rgba8, const uint8_t* d_tile_rgba8, int32_t width, int32_t height,
                    const int32_t* d_pixels_xy, int64_t n_px, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_canvas_rgba8 || !d_tile_rgba8 || width <= 0 || height <= 0 || n_px < 0) return VRT_ERR_ARG;
    const int64_t n = d_pixels_xy ? n_px : (int64_t)width * height;
    if (n == 0) return VRT_OK;
    hipLaunchKernelGGL(canvas_blit_kernel, dim3((unsigned)grid_for(n)), dim3(VRT_BLOCK), 0, stream, (uchar4*)d_canvas_rgba8,
                       (const uchar4*)d_tile_rgba8, d_pixels_xy, n, (int)width);
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
