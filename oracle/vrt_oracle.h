/* vrt_oracle.h -- CPU restatement of the reference's per-pixel trace path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package (python_raytracer_amd/)
 * imports, links or executes this library; it is used by tests/, by
 * __graft_entry__.smoke() as the checker, and by bench.py's `cpu_baseline` leg.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks this restatement (libm mode
 * ORC_LIBM_GLIBC) bit for bit against golden vectors produced by importing and running the
 * real reference in the build container (tests/golden/make_golden.py).  The reference ships
 * no tests or golden vectors of its own.
 *
 * Follows (reference file:line, relative to the reference checkout):
 *   Camera.tile             init.py:126-150
 *   Camera.trace            init.py:37-121
 *   Camera.chunk_get        init.py:28-33
 *   Frame.get_voxel         data.py:136-145 (closed form over a dense grid, see orc_lookup)
 *   lib.material            lib.py:448-460
 *   lib.material_background lib.py:463-476
 *   vec3 / quaternion / rgb lib.py:174-190, 228-244, 264-265, 276-280, 310-338, 353-358, 372-376, 393-395
 *   average / merge / rand / mix   lib.py:404-409, 419-428, 431-439
 *   random.seed / random.random    CPython 3.10 Modules/_randommodule.c (MT19937, init_by_array, genrand_res53)
 */
#ifndef VRT_ORACLE_H
#define VRT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_LIBM_GLIBC = 0, ORC_LIBM_PORTABLE = 1 };

enum { ORC_C_LOOKUP = 0, ORC_C_NBR, ORC_C_RESNAP, ORC_C_CHUNK_GET, ORC_C_HIT, ORC_C_DRAW, ORC_C_ADV, ORC_C_BROKE,
       ORC_NCOUNTERS };

typedef struct {
    int32_t width, height;       /* settings.width / height (data.py:19-20) */
    int32_t samples;             /* settings.samples (data.py:29) */
    int32_t chunk_size;          /* settings.chunk_size (data.py:39) */
    int32_t chunk_radius;        /* round(chunk_size / 2) (data.py:68) */
    int32_t has_background;      /* data.background is material_background (init.py:119) */
    uint64_t seed_nonce;         /* 0 for settings.static; otherwise every ray slot is seeded (y*width + x)*max_samples + s + nonce */
    double proportions;          /* data.py:66 */
    double shutter, falloff, dof, dist_min, dist_max, max_light, max_bounces;
    double lod_bounces, lod_samples, lod_random, lod_edge;
} orc_settings;

typedef struct {
    int64_t origin[3];           /* world position of grid[0][0][0]; multiple of chunk_size */
    int64_t dims[3];             /* extent in chunks */
    const uint8_t* present;      /* [dims0][dims1][dims2] 1 = camera has a Frame for this chunk */
    const uint8_t* res;          /* [dims0][dims1][dims2] Frame.resolution */
    const uint8_t* grid;         /* [X][Y][Z] material id (0 empty), voxels at world coords (q * resolution) */
    int32_t n_materials;
    const double* materials;     /* [n][7]: albedo r,g,b, roughness, absorption, ior, energy; id = row + 1 */
} orc_scene;

typedef struct {
    double pos[3];               /* Camera.pos */
    double rot[4];               /* Camera.rot x,y,z,w */
    double lens;                 /* Camera.lens = fov * pi / 8 (init.py:17) */
} orc_camera;

typedef struct {
    int32_t x, y, s;
    int32_t color[3];
    int32_t alpha;
    int32_t ntrav;
    int32_t counters[ORC_NCOUNTERS];
    double detail, energy, step, life, bounces;
    double pos[3], vel[3];
} orc_ray;

/* Render the listed pixels.  Ray records (optional) are written in the reference's call order
 * (pixel order as listed, then sample index).  traversed (optional) receives the order-preserving
 * union of every ray's traversed chunk positions, as the reference's tile() returns it.
 * Returns 0, or a negative error code (-1 bad argument, -2 capacity exceeded). */
int orc_render(const orc_scene* scene, const orc_settings* st, const orc_camera* cam,
               const int32_t* pixels_xy, int64_t n_px, int libm_mode, int n_threads,
               double* pix_mean, uint8_t* pix_rgba8,
               orc_ray* rays, int64_t rays_cap, int64_t* n_rays,
               int64_t* counters,
               int64_t* traversed, int64_t trav_cap, int64_t* n_trav);

/* Window.chunk_update's selection loop (init.py:447-452): for every world chunk that has voxel data, keep it for
 * the camera iff (not culling or its position is in `traversed`), at LOD
 * min(trunc(dist(chunk centre, camera) / (dist_max / (1 + chunk_lod))), chunk_lod); Frame.resolution = lod + 1.
 * world_present / out_present / out_res: [dims0][dims1][dims2]; traversed: [n][3] chunk positions.
 * (math.dist is restated as sqrt(dx^2 + dy^2 + dz^2); the LOD index only depends on it through trunc().) */
void orc_select_chunks(const int64_t* origin, const int64_t* dims, int32_t chunk_size, int32_t chunk_radius,
                       const uint8_t* world_present, const double* cam_pos, double dist_max, int32_t chunk_lod,
                       int32_t culling, const int64_t* traversed, int64_t n_trav, uint8_t* out_present, uint8_t* out_res);

/* MT19937 exactly as CPython's random.seed(int) / random.random(): writes n draws. */
void orc_rng_draws(uint64_t seed_lo, uint64_t seed_hi, int n, double* out);

/* libm probes for tests/test_math.py */
double orc_sin(int libm_mode, double x);
double orc_cos(int libm_mode, double x);
double orc_pow(int libm_mode, double x, double y);

/* number of samples of pixel (x, y) (init.py:131-134) */
/* Window.draw_tile alpha-over blit (init.py:185-190); PARITY UNPINNED (pygame unavailable) */
void orc_canvas_blit(uint8_t* canvas, const uint8_t* tile, int64_t n_pixels);

int32_t orc_pixel_samples(const orc_settings* st, int32_t x, int32_t y);

#ifdef __cplusplus
}
#endif
#endif
