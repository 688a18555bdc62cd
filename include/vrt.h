/* vrt.h -- C ABI of the MI355X voxel ray-march library (python_raytracer_amd/_vrt.so).
 *
 * The reference (MirceaKitsune/python_raytracer) is pure Python and has no FFI of its own; its
 * boundary for this path is the object API `Camera.tile(thread, t)` / `Camera.trace(...)`
 * (reference init.py:126-150, 37-121) reading `Camera.pos/rot/lens/chunks` (init.py:14-33),
 * `data.settings` (data.py:18-77) and `data.Material` attributes (data.py:85-93).  This header is
 * what a binding for that path calls instead of the Python loops; python_raytracer_amd/camera.py
 * is that binding (ctypes).  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: plain C, no exceptions; every function returns 0 or a negative vrt_status.
 * All `d_` pointers are DEVICE pointers owned by the caller (e.g. PyTorch-ROCm tensors) and are
 * only borrowed for the duration of the call; `stream` is a hipStream_t passed as void*.
 * Device memory is allocated only by vrt_pow_memo_create() and freed only by vrt_release_caches(); no other
 * function allocates, frees or synchronises (vrt_profile_end() waits for its events), so a frame may be captured
 * into a hipGraph.
 */
#ifndef VRT_H
#define VRT_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRT_ABI_VERSION 8

typedef enum {
    VRT_OK = 0,
    VRT_ERR_ARG = -1,        /* null pointer / bad size / unsupported setting */
    VRT_ERR_HIP = -2,        /* a HIP runtime call failed (vrt_last_hip_error) */
    VRT_ERR_WORKSPACE = -3,  /* workspace / plan buffer too small */
    VRT_ERR_NO_DEVICE = -4,
    VRT_ERR_PLAN = -5,       /* the tile plan was built for other pixels / settings */
} vrt_status;

/* Render settings: the subset of data.settings the path reads (reference data.py:26-50, 64-68). */
typedef struct vrt_settings {
    int32_t width, height;     /* settings.width / height */
    int32_t samples;           /* settings.samples */
    int32_t chunk_size;        /* settings.chunk_size: power of two >= 8 */
    int32_t chunk_radius;      /* round(chunk_size / 2) */
    int32_t has_background;    /* 1: data.background is lib.material_background, 0: None (init.py:119) */
    uint64_t seed_nonce;       /* 0 = settings.static: seed = (1+x)(1+y)(1+s) (init.py:137).  Non-zero = a non-static
                                  frame (init.py:147 re-seeds from OS entropy, no parity requirement): every ray slot
                                  gets its own stream, seed = (y * width + x) * max_samples + s + seed_nonce */
    double proportions;        /* settings.proportions (data.py:66) */
    double shutter, falloff, dof, dist_min, dist_max, max_light, max_bounces;
    double lod_bounces, lod_samples, lod_random, lod_edge;
} vrt_settings;

/* Camera state (reference init.py:15-17). */
typedef struct vrt_camera {
    double pos[3];
    double rot[4];             /* quaternion x, y, z, w */
    double lens;               /* fov * pi / 8 */
} vrt_camera;

/* Packed scene = what Camera.chunks holds (reference init.py:18), flattened.
 *   chunk_table[(cx*dims[1] + cy)*dims[2] + cz], chunk (cx,cy,cz) at world origin + c*chunk_size:
 *       0                      -> no Frame for this chunk (void)
 *       slot+1 | (res << 24)   -> voxel block `slot`, Frame.resolution `res` (1..255)
 *   voxels: n_slots blocks of chunk_size^3 bytes, material id per voxel (0 = empty), stored at WORLD
 *       coordinates q*res (reference data.py:136-138, 163-175).  Inside a block voxels are bricked:
 *       8^3 bricks ordered [bx][by][bz], each brick 512 B = 8 micro-bricks [mx][my][mz] of 4^3 voxels
 *       (64 B, [x][y][z]).  vrt_voxel_offset() gives the byte offset.
 *   materials: n_materials records of 8 doubles {r, g, b, roughness, absorption, ior, energy, 0};
 *       material id = record index + 1 (reference data.py:85-93 attributes read by lib.py:448-460). */
typedef struct vrt_scene {
    int64_t origin[3];         /* world coords of chunk (0,0,0); multiples of chunk_size */
    int32_t dims[3];           /* chunks per axis */
    int32_t chunk_size;
    int32_t n_slots;
    int32_t n_materials;       /* <= 255 */
    const uint32_t* d_chunk_table;
    const uint8_t* d_voxels;
    const double* d_materials;
    const uint64_t* d_occupancy; /* n_slots * chunk_size^3 / 64 words, bit b of word w = (d_voxels[64 w + b] != 0): one
                                    word per 4^3 micro-brick, eight consecutive words (one 64-byte line) per 8^3 brick.
                                    Derived from d_voxels by vrt_occupancy_build; only the measurement variants of the
                                    march (VRT_LOOKUP=1|2) read it, the shipped kernels read the bytes */
    int32_t max_resolution;      /* largest Frame.resolution in d_chunk_table if known (1, 2, ...), 0 = unknown (the
                                    generic kernel).  Selects the kernel variant: an OVERSTATED value only costs speed, an
                                    UNDERSTATED one gives wrong results (the resolution-1 and resolution <= 2 variants
                                    leave out the snapping larger resolutions need) */
    int32_t flags;               /* VRT_SCENE_* bits, 0 if nothing is known (ABI 5 called this field `pad`) */
    const uint32_t* d_world_tables; /* NULL, or the world-axis offset tables of a VRT_SCENE_LAYOUT_DENSE scene
                                    (vrt_world_tables_build: a function of dims and chunk_size only) */
} vrt_scene;
/* vrt_scene.flags: d_chunk_table[i] == (i + 1) | (1 << 24) for every cell -- a dense world, every chunk present at
 * resolution 1, its voxel blocks in table order (what vrt_synth_volume writes; n_slots must equal the number of cells).
 * A chunk table beyond the size the march keeps in LDS (4 096 cells) is otherwise read from memory at every chunk
 * border a ray crosses; with this bit the march computes the entry instead.  Only set it for a table that was checked. */
#define VRT_SCENE_TABLE_IS_IDENTITY 1
/* vrt_scene.flags: the voxel blocks lie in TABLE ORDER -- n_slots equals the number of cells and the block of cell i is
 * block i, whether or not the table lists the cell (d_chunk_table[i] is 0 or (i + 1) | res << 24; the bytes of an unlisted
 * cell's block are never interpreted).  The voxel address of a world cell is then a sum of three per-axis terms, so the
 * march can look ahead ACROSS chunk borders without reading the table (world-axis offset tables in LDS; DESIGN.md
 * section 4) and replays the reference's re-snaps (init.py:67-73) for the borders a ray really crossed afterwards.
 * vrt_voxelize / vrt_synth_volume / vrt_select_chunks produce and keep this layout; PackedScene builds it for small
 * scene boxes.  Implied by VRT_SCENE_TABLE_IS_IDENTITY.  Only set it for a table that was checked.  The look-ahead also
 * needs d_world_tables, at most 2^30 bytes of voxels, and tables that fit the march's LDS (vrt_world_tables_bytes says). */
#define VRT_SCENE_LAYOUT_DENSE 2

/* Box of chunk cells in which visited chunks are recorded (the `traversed` list of init.py:72-73, 143).
 * d_keys[(cx*dims[1]+cy)*dims[2]+cz] receives min over rays of (ray_index << 12 | resnap_index), or
 * UINT64_MAX if never visited; the caller fills it with 0xFF bytes before the call, or sets `reset` and vrt_render_tile
 * (and vrt_trace_rays) does (in the launch that clears the frame's counters: one kernel less per frame).  Sorting the visited
 * cells by key reproduces the reference's order-preserving union.  Visits outside the box are only counted
 * (d_stats[VRT_S_TRAV_OUTSIDE]): size the box for the rays' reach, (dist_max + 1 + chunk_size / 2) * max |vel|_inf
 * around the camera -- the primary velocity of a ROTATED camera can exceed 1 per component, because the
 * reference's quaternion product (lib.py:353-358) is not norm-preserving; see Camera._velocity_bound. */
typedef struct vrt_traversed {
    int64_t origin[3];         /* world coords of cell (0,0,0); multiples of chunk_size */
    int32_t dims[3];
    int32_t reset;             /* vrt_render_tile: nonzero = set every key to UINT64_MAX before the frame's rays record theirs
                                  (0: the keys are the caller's -- e.g. several tiles recorded into one box) */
    uint64_t* d_keys;          /* may be NULL: do not record */
} vrt_traversed;

/* Per-ray end state (the `ray` store returned by Camera.trace, init.py:50-59, plus event counters).
 * Only written when requested; 152 bytes. */
enum { VRT_C_LOOKUP = 0, VRT_C_NBR, VRT_C_RESNAP, VRT_C_CHUNK_GET, VRT_C_HIT, VRT_C_DRAW, VRT_C_ADV, VRT_C_BROKE,
       VRT_NCOUNTERS };
typedef struct vrt_ray {
    int32_t x, y, s;
    int32_t color[3];
    int32_t alpha;
    int32_t ntrav;
    int32_t counters[VRT_NCOUNTERS];
    double detail, energy, step, life, bounces;
    double pos[3], vel[3];
} vrt_ray;

/* Frame statistics written by vrt_render_tile into d_stats (16 x uint64, zeroed by the callee):
 *   [0..7] event counters summed over rays (VRT_C_*), [8] primary rays traced, [9] rays that needed more
 *   random draws than the fast table held and were re-traced, [10] rays whose draws exceeded every table
 *   (result invalid -> the Python wrapper raises), [11] chunk visits outside the traversed box, [12] workgroups of the
 *   frame's march that ran march_pool_kernel (rays regrouped between lanes through LDS; 0: march_kernel, one ray per lane
 *   -- which one runs is the library's choice by launch size and LDS room, and never changes a result).  Bits 32 and up of
 *   the same word count those of them that took their rays as square pixel tiles in Morton order, one eighth of the window
 *   per XCD, instead of in list order (VRT_TILED=1, a measured variant: whole-window launches over scenes beyond the caches). */
enum { VRT_S_RAYS = 8, VRT_S_RNG_RETRACED = 9, VRT_S_RNG_EXHAUSTED = 10, VRT_S_TRAV_OUTSIDE = 11, VRT_S_POOL_GROUPS = 12,
       VRT_S_STALLED = 13,  /* waves of march_pool_kernel that found nothing to run for 4096 passes in a row and gave up
                               (internal error: the frame is invalid, the Python wrapper raises) */
       VRT_S_LOOKAHEAD_GROUPS = 14,  /* workgroups of the frame's march whose look-ahead crossed chunk borders (march_step_w:
                                        VRT_SCENE_LAYOUT_DENSE scenes) */
       VRT_S_RAYGEN_GROUPS = 15,  /* workgroups of the frame's march that worked out their rays' lens quaternions and lives
                                     themselves (vrt_render_tile without d_ray_table: no ray table is written then) */
       VRT_NSTATS = 16 };

int vrt_abi_version(void);
const char* vrt_status_string(int status);
int vrt_last_hip_error(void);           /* hipError_t of the last failing HIP call on this thread */
int vrt_device_count(int* count);

/* The shader's pow (lib.py:450, 465) has one exponent per frame, 1 + falloff, and a handful of bases: it is memoised.
 * vrt_pow_memo_create allocates (hipMalloc + a synchronous clear, so not inside a stream capture) a 4-KiB table for
 * (current device, falloff) that the library keeps and every later frame with that falloff reuses; without it a frame
 * memoises into its workspace and starts cold (about 0.3 ms per frame).  At most 64 tables per process (then
 * VRT_ERR_WORKSPACE).  vrt_release_caches synchronises each device that owns one and frees them all. */
int vrt_pow_memo_create(double falloff);
int vrt_release_caches(void);

/* d_occupancy of a vrt_scene from its voxel bytes: n_bytes = n_slots * chunk_size^3 (a multiple of 64). */
int vrt_occupancy_build(const uint8_t* d_voxels, int64_t n_bytes, uint64_t* d_occupancy, void* stream);

/* World-axis offset tables of a scene whose voxel blocks lie in table order (VRT_SCENE_LAYOUT_DENSE): three arrays
 * X | Y | Z of (dims[a] * chunk_size + 64) words -- 32 guard cells either side of the world -- such that the voxel of
 * world cell (x, y, z) is byte X[x] + Y[y] + Z[z] of d_voxels.  This is what replaces Frame.get_voxel's "which chunk,
 * then which cell" (reference data.py:136-145, init.py:67-77) for the positions the march reads ahead: no chunk table
 * lookup between a position and its voxel.  vrt_world_tables_bytes: *bytes = size of the three tables, or 0 when the
 * march could not use them (more than 2^30 bytes of voxels, or tables beyond its LDS budget). */
int vrt_world_tables_bytes(const int32_t* dims, int32_t chunk_size, int64_t* bytes);
int vrt_world_tables_build(const int32_t* dims, int32_t chunk_size, uint32_t* d_tables, int64_t bytes, void* stream);

/* Byte offset of voxel (lx,ly,lz) inside a chunk block (host helper; same function the kernels use). */
int64_t vrt_voxel_offset(int32_t chunk_size, int32_t lx, int32_t ly, int32_t lz);

/* Maximum samples per pixel for these settings (init.py:133-134). */
int32_t vrt_max_samples(const vrt_settings* st);

/* ---- tile plan ---------------------------------------------------------------------------------------
 * A static index over a pixel list, built once per (pixel list, width, height, samples, lod_edge) like the
 * reference builds settings.pixels once at start-up (data.py:70-77): the per-sample seeds
 * (1+x)(1+y)(1+s) (init.py:137) collide heavily (a 4K x 8 spp frame has 62.7 M rays but 7.8 M distinct seeds),
 * and equal seeds give equal random streams, so every frame seeds MT19937 once per DISTINCT seed.
 * The plan holds the sorted distinct seeds and, per ray slot, the index of its seed.
 *   layout of d_plan: 64-byte header {u64 magic, n_px, n_slots, n_distinct, settings_hash, n_words, full_frame, 0}
 *                     (full_frame = 1: the pixel list is the whole window in x-major order, pixel p = (p / height,
 *                     p % height); vrt_render_tile then resolves in 16 x 16 tiles, writing whole image-row segments),
 *                     u32 seed_list[n_slots], u32 ray_seedidx[n_slots]   (0xFFFFFFFF = unused sample slot)
 * Requires (width * height * max_samples) < 2^32.  vrt_plan_build is asynchronous; read the first 64 bytes of
 * d_plan back (after the stream has finished) to learn n_distinct (header word 3). */
int vrt_plan_bytes(const vrt_settings* st, int64_t n_px, int64_t* plan_bytes, int64_t* scratch_bytes);
int vrt_plan_build(const vrt_settings* st, const int32_t* d_pixels_xy, int64_t n_px, void* d_plan, int64_t plan_bytes,
                   void* d_scratch, int64_t scratch_bytes, void* stream);

/* Workspace bytes vrt_render_tile needs: the frame's draw table (n_distinct rows of fast_draws doubles) and ray table
 * (64 bytes per ray slot) unless the caller passes its own (external: VRT_WS_* bits), 4 bytes per ray slot for the
 * per-sample results and the retrace tables (1/64 of the ray slots, between 2^18 and 2^22 rays, times 912 bytes, plus
 * 32 MiB for the third tier).  BASELINE config 3 (62.7 M rays): 7.5 GB, or 1.25 GB with both tables external.
 * fast_draws: random draws kept per distinct seed in the frame's table, 32 or 64; rays that consume more are re-traced
 * with a private 113-draw row (and the few that outrun that, with a 1024-draw row from a full-state MT19937), so the
 * choice changes speed only, never results (32 suits max_bounces <= ~4; scenes where many rays take > 9 rough hits
 * want 64). */
enum { VRT_WS_DRAW_TABLE = 1, VRT_WS_RAY_TABLE = 2 };
int vrt_workspace_bytes(const vrt_settings* st, int64_t n_px, int64_t n_distinct, int32_t fast_draws, int32_t external,
                        int64_t* bytes);

/* ---- draw table -------------------------------------------------------------------------------------
 * table[i * fast_draws + k] = k-th random.random() after random.seed(seed_list[i] + st->seed_nonce)
 * (init.py:137, 139; lib.py:434), for every distinct seed of the plan.  With static seeds (init.py:136-137) the
 * table depends only on the plan, fast_draws and seed_nonce, not on the frame: build it once with
 * vrt_draw_table_build and pass it to every vrt_render_tile call, or pass NULL there and the frame seeds its own
 * table into the workspace (what a non-static run, whose nonce changes every frame, needs anyway). */
int vrt_draw_table_bytes(int64_t n_distinct, int32_t fast_draws, int64_t* bytes);
int vrt_draw_table_build(const vrt_settings* st, const int32_t* d_pixels_xy, int64_t n_px, const void* d_plan,
                         int64_t n_distinct, int32_t fast_draws, double* d_table, int64_t table_bytes, void* stream);

/* ---- ray table --------------------------------------------------------------------------------------
 * Per ray slot (pixel, sample): the lens quaternion o = vec3(0, -lens_x, +lens_y).quaternion() of trace()
 * (init.py:41-43; lib.py:322-338) and the ray's life (dist_max - dist_min) * ray_detail (init.py:56, 139).  Both are
 * functions of the pixel, its draws, the settings and cam->lens only -- not of the camera's position or rotation --
 * so with static seeds they are frame-invariant like the draw table: build once, pass to every vrt_render_tile
 * (which then only multiplies by the camera rotation, lib.py:353-358, 372-376, when a lane picks the ray up), or
 * pass NULL: a table that would be written for one frame and read once is not written at all -- each lane of the march
 * works out the record of the ray it picks up from the draw row (d_stats[VRT_S_RAYGEN_GROUPS] counts the workgroups
 * that did; one record per pixel -- below -- is still built into the workspace, and so is the per-ray table for scenes
 * with resolutions > 2, whose march has no such variant).
 *   layout: one 64-byte record per ray slot (n_px * max_samples of them): doubles ox, oy, oz, ow, life (life < 0:
 *   unused sample slot) and the three draws of the ray's first rough hit (lib.py:457), copied from the draw table so
 *   that they arrive with the ray.
 *   With st->dof == 0, lod_random == 0 and lod_samples == 0 neither the lens quaternion nor the life depends on the
 *   sample (no lens jitter; detail / (1 + s * 0) * (1 - 0 * draw) is the pixel's detail exactly): the table then holds one
 *   record per PIXEL (n_px of them: ox, oy, oz, ow, life, the pixel's sample count as a double, 0, 0) and the march
 *   reads a ray's first-hit draws from the draw table -- 16 x less memory at BASELINE config 5 (1.1 GB instead of
 *   17.2 GB).  vrt_ray_table_bytes / _build and vrt_render_tile choose the layout from the settings alone. */
int vrt_ray_table_bytes(const vrt_settings* st, int64_t n_px, int64_t* bytes);
int vrt_ray_table_build(const vrt_settings* st, double lens, const int32_t* d_pixels_xy, int64_t n_px,
                        const void* d_plan, const double* d_draw_table, int32_t fast_draws, double* d_ray_table,
                        int64_t table_bytes, void* stream);

/* Camera.tile (init.py:126-150) for the pixel list d_pixels_xy ([n_px][2] int32, order = settings.pixels[t]).
 * d_plan / n_distinct: the plan built for this pixel list and the distinct-seed count read from its header.
 * d_draw_table: table built by vrt_draw_table_build for this plan, fast_draws and st->seed_nonce, or NULL.
 * d_ray_table: table built by vrt_ray_table_build from that draw table for these settings and cam->lens, or NULL
 *   (needs d_draw_table when given).
 * With st->seed_nonce != 0 every ray slot has its own seed: n_distinct must be n_px * max_samples.
 * Outputs (each may be NULL):
 *   d_rgba_f32   [n_px][4] float   per-pixel mean of the samples' [r,g,b,alpha] (lib.average, before set_at)
 *   d_image_u8   [height][width][4] RGBA8 full-window image; only the listed pixels are written (others keep
 *                what the caller put there -- zero it to get tile()'s transparent background)
 *   d_ray_rgba   [n_px * max_samples] uint32 r|g<<8|b<<16|alpha<<24 per sample (0 for unused sample slots)
 *   d_rays       [n_px * max_samples] vrt_ray debug records (slower path; unused slots have s = -1)
 *   d_stats      [VRT_NSTATS] uint64
 *   trav         traversed box (or NULL) */
int vrt_render_tile(const vrt_scene* scene, const vrt_settings* st, const vrt_camera* cam,
                    const int32_t* d_pixels_xy, int64_t n_px, const void* d_plan, int64_t n_distinct,
                    int32_t fast_draws, const double* d_draw_table, const double* d_ray_table, void* d_workspace,
                    int64_t workspace_bytes, float* d_rgba_f32, uint8_t* d_image_u8, uint32_t* d_ray_rgba, vrt_ray* d_rays,
                    uint64_t* d_stats, const vrt_traversed* trav, void* stream);

/* Camera.trace (init.py:37-121) for explicit rays: direction (dir_x, dir_y), detail and the random draws the
 * ray may consume (d_draws[i * n_draws + k] = k-th random.random() of ray i).  d_rays[i].counters[VRT_C_DRAW]
 * tells how many were consumed.  Rays that would need more draws are counted in d_stats[VRT_S_RNG_EXHAUSTED].
 * d_workspace: vrt_trace_workspace_bytes(n_rays) bytes. */
int vrt_trace_workspace_bytes(int64_t n_rays, int64_t* bytes);
int vrt_trace_rays(const vrt_scene* scene, const vrt_settings* st, const vrt_camera* cam,
                   const double* d_dir_x, const double* d_dir_y, const double* d_detail,
                   const double* d_draws, int32_t n_draws, int64_t n_rays,
                   void* d_workspace, int64_t workspace_bytes,
                   vrt_ray* d_rays, uint64_t* d_stats, const vrt_traversed* trav, void* stream);

/* MT19937 exactly as CPython random.seed(seed); [random.random() for _ in range(n_draws)]
 * (init.py:137, 139; lib.py:434): d_out[i * n_draws + k] = k-th draw of seed d_seeds[i]. 2 <= n_draws <= 4096
 * (up to 113 draws come from the register-only seeding kernel, more from a slower full-state generator). */
int vrt_rng_draws(const uint64_t* d_seeds, int64_t n_seeds, int32_t n_draws, double* d_out, void* stream);

/* One visible object of the world (data.Object, data.py:430-494, 589-600): the world box it occupies and the model
 * its sprite holds.  64 bytes. */
typedef struct vrt_object {
    int32_t mins[3];   /* world box [mins, maxs): ceil(pos) - size/2 .. floor(pos) + size/2 (data.py:591-599) */
    int32_t maxs[3];
    int32_t size[3];   /* sprite size = shape of the model array */
    int32_t turns[3];  /* quarter turns about x, y, z: round(rot / 90) % 4 (data.py:338-371) */
    int64_t model;     /* offset in d_models of this sprite frame's u8 ids [size.x][size.y][size.z], 0 = empty */
    int32_t remap;     /* offset in d_remap of its table: model id -> world material id (entry 0 unused) */
    int32_t pad;
} vrt_object;

/* The object loop of Window.chunk_update (init.py:398-444) on the device: every world voxel of the chunk box
 * [origin, origin + dims * chunk_size) takes the voxel of the LAST object (in array order, the dict union of
 * init.py:437-439) that has one at that position, read through the object's quarter-turn rotation
 * (Sprite.get_voxel, data.py:417-419).
 *   d_voxels      out: dims.x * dims.y * dims.z blocks of chunk_size^3 bytes in vrt_voxel_offset order, block index =
 *                 (cx * dims.y + cy) * dims.z + cz
 *   d_world_table out: [dims] block index + 1 | 1 << 24 where the chunk holds a voxel, else 0 -- the world table
 *                 vrt_select_chunks reads; with it and d_voxels a vrt_scene is complete.
 *   d_chunk_list  NULL: every chunk of the box is rebuilt.  Else n_list chunk indices ((cx * dims.y + cy) * dims.z +
 *                 cz): only these chunks are rebuilt (from ALL objects, in array order) and the rest of d_voxels /
 *                 d_world_table stays as it is -- the reference's per-object invalidation (init.py:398-429: an object
 *                 that moved, turned, appeared or vanished marks the chunks its old and new boxes touch, and only
 *                 those are recombined), for a large world in which one object moves. */
int vrt_voxelize(const vrt_object* d_objects, int32_t n_objects, const uint8_t* d_models, const uint8_t* d_remap,
                 const int64_t* origin, const int32_t* dims, int32_t chunk_size, const uint32_t* d_chunk_list,
                 int64_t n_list, uint32_t* d_world_table, uint8_t* d_voxels, void* stream);

/* Window.chunk_update's selection loop (init.py:447-452) on the device: which world chunks the camera renders this
 * frame and at which LOD.  The voxel blocks stay resident at full resolution; a chunk's LOD is only the resolution
 * byte of its table entry (a Frame of resolution r holds the voxels at coordinates divisible by r, which is what the
 * kernel's lookup snaps to), so re-selecting costs one tiny kernel and no voxel traffic.
 *   d_world_table  [dims] entries slot+1 (resolution bits ignored), 0 = no voxel data in this chunk
 *   d_camera_table [dims] out: slot+1 | (lod+1)<<24 where the chunk is kept, else 0
 *   kept iff (!culling || chunk position was traversed in `prev`) (init.py:447); prev = the vrt_traversed box a
 *   previous vrt_render_tile filled (NULL or NULL keys: nothing traversed)
 *   lod = min(trunc(dist(chunk centre, cam_pos) / (dist_max / (1 + chunk_lod))), chunk_lod) (init.py:448-449),
 *   chunk centre = position + round(chunk_size / 2). */
int vrt_select_chunks(const uint32_t* d_world_table, const int64_t* origin, const int32_t* dims, int32_t chunk_size,
                      const double* cam_pos, double dist_max, int32_t chunk_lod, int32_t culling,
                      const vrt_traversed* prev, uint32_t* d_camera_table, void* stream);

/* Window.draw_tile (init.py:185-190): alpha-over blit of a tile's RGBA8 window image (vrt_render_tile's d_image_u8)
 * onto the persistent RGBA8 canvas, [height][width][4] both.  d_pixels_xy: the tile's own pixels ([n_px][2]), or NULL
 * for every pixel of the window -- the same result, since a tile's other pixels are transparent.  The blend restates
 * pygame 2's ALPHA_BLEND (dst alpha 0: copy; else dC = ((dC << 8) + (sC - dC) * sA + sC) >> 8, dA = sA + dA -
 * sA * dA / 255); pygame cannot be run in the build environment, so this entry point is PARITY UNPINNED. */
int vrt_canvas_blit(uint8_t* d_canvas_rgba8, const uint8_t* d_tile_rgba8, int32_t width, int32_t height,
                    const int32_t* d_pixels_xy, int64_t n_px, void* stream);

/* Optional per-kernel timing for bench.py.  Between vrt_profile_begin() and vrt_profile_end() every kernel
 * launched by vrt_render_tile is bracketed by HIP events on its launch stream.  vrt_profile_end() waits for
 * those events (the only call in this header that blocks on the device) and returns, per kind, the summed
 * milliseconds and the launch count.  An event pair costs a frame ~12 us (measured: 37 us for the three pairs of a
 * BASELINE config 2 frame of 0.56 ms); vrt_profile_begin_kinds(1 << VRT_PROF_MARCH) times the march alone. */
enum { VRT_PROF_RNG = 0, VRT_PROF_MARCH = 1, VRT_PROF_RETRACE = 2, VRT_PROF_RESOLVE = 3, VRT_PROF_RAYGEN = 4,
       VRT_NPROF = 8 };
int vrt_profile_begin(void);
int vrt_profile_begin_kinds(uint32_t kinds);   /* bit k: time kind k */
int vrt_profile_end(double* ms, int64_t* launches);

/* Fill a packed voxel buffer with the synthetic dense volume of BASELINE config 5: edge n voxels (multiple of
 * chunk_size), centred on the world origin, all chunks present at resolution 1, slot = linear chunk index.
 * d_chunk_table: [(n/cs)^3] uint32, d_voxels: [n^3] bytes. */
int vrt_synth_volume(int32_t n, int32_t chunk_size, uint32_t* d_chunk_table, uint8_t* d_voxels, void* stream);

#ifdef __cplusplus
}
#endif
#endif
