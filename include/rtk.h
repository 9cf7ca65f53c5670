/*
 * rtk.h -- C ABI of the MI355X (gfx950) ray-trace core.
 *
 * The library replaces ONE path of mattkleiny/raytracer-rust: the per-pixel
 * loop `Camera::render(&self, scene: &Scene) -> Canvas` (src/scene/cameras.rs:72-85)
 * and everything it calls (`Scene::trace`, src/scene.rs:136-296).  The reference has no
 * FFI of its own; a Rust host would bind these entry points from an `extern "C"` block
 * and call `rtk_render` where it calls `camera.render(&scene)` today.
 *
 * Conventions
 *  - Every matrix is 16 doubles, row-major, element (row, column) at [column + 4*row]
 *    (src/maths/matrices.rs:69-71).
 *  - Every scene quantity crosses the boundary as f64, exactly as the reference holds
 *    it.  Narrowing to FP32 happens inside the library, and only where it cannot change
 *    a decision (see rtk_render).
 *  - The caller owns every buffer.  The library never frees caller memory and never
 *    throws or aborts across the boundary; every entry point returns RTK_OK (0) or a
 *    negative rtk_status and rtk_last_error() describes the failure.
 *  - A context is used by one thread at a time (the reference renderer is
 *    single-threaded, src/scene/cameras.rs:75-82); distinct contexts are independent.
 *  - There is no CPU fallback: without a usable HIP device rtk_create fails.
 */
#ifndef RTK_H
#define RTK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rtk_status {
  RTK_OK = 0,
  RTK_ERR_INVALID = -1,         /* NULL pointer, zero size, bad enum value            */
  RTK_ERR_SINGULAR_CAMERA = -2, /* camera transform has determinant 0; the reference
                                   panics here (src/scene/cameras.rs:59-62)            */
  RTK_ERR_UNSUPPORTED = -3,     /* input outside the supported envelope (see below)    */
  RTK_ERR_HIP = -4,             /* a HIP runtime call failed                           */
  RTK_ERR_NO_DEVICE = -5,       /* no gfx950 device / library built without a kernel   */
  RTK_ERR_NOMEM = -6,
  RTK_ERR_RCCL = -7             /* librccl could not be loaded, or an RCCL call failed */
} rtk_status;

/* Shape kinds: `PackedKind` (src/loader.rs:52-57). */
enum { RTK_SPHERE = 0, RTK_PLANE = 1 };

/* Texture kinds: `PackedTexture` (src/loader.rs:137-145) / `Texture` (src/scene/materials.rs:7-10). */
enum { RTK_TEX_SOLID = 0, RTK_TEX_CHECKER = 1, RTK_TEX_GRADIENT = 2, RTK_TEX_RING = 3, RTK_TEX_STRIPED = 4 };

/* Recursion limit of the reference: `Scene::MAX_DEPTH` (src/scene.rs:109). */
#define RTK_REFERENCE_MAX_DEPTH 5
/* Largest max_depth the device path accepts (bounds the per-pixel ray stack). */
#define RTK_MAX_DEPTH_LIMIT 8
/* Most lights a scene may hold (`Scene::add_light`, src/scene.rs:131-133, has no limit of its own). */
#define RTK_MAX_LIGHTS 32767
/* Frame tile used for the multi-GPU partition: RTK_TILE_W x RTK_TILE_H pixels. */
#define RTK_TILE_W 32
#define RTK_TILE_H 8

/* `Material` (src/scene/materials.rs:23-32).  For RTK_TEX_SOLID only color_a is used.
 * pattern_transformed = n > 0 wraps the pattern in n nested `TransformPattern`s
 * (src/graphics/patterns.rs:7-35; the type is generic over its pattern, so a TransformPattern may
 * wrap another): the object-space point is multiplied by the inverse of the OUTERMOST wrapper's
 * transform first (pattern_transform, row-major), then by the inverses of the n - 1 inner ones in
 * turn (inner_pattern_transforms: n - 1 matrices of 16 doubles, outer to inner; NULL for n <= 1),
 * and the pattern is sampled at the result.  A wrapper whose matrix is singular leaves the point as
 * it is, each on its own, as the reference's `if let Ok(inverse)` does.  The transforms need not be
 * affine.  Scene files cannot express any of this.  At most RTK_MAX_PATTERN_LEVELS wrappers. */
#define RTK_MAX_PATTERN_LEVELS 8
typedef struct rtk_material {
  uint32_t texture_kind;
  uint32_t pattern_transformed;
  double color_a[3];
  double color_b[3];
  double ambient, diffuse, specular, shininess;
  double transparency, reflectivity, refractivity;
  double pattern_transform[16];
  const double* inner_pattern_transforms;
} rtk_material;

/* `SceneNode<S>` (src/scene.rs:36-41).  The inverse transform is derived inside the
 * library with the reference's cofactor method (src/maths/matrices.rs:183-198); a
 * singular transform silently gets an identity inverse (src/scene.rs:55-62). */
typedef struct rtk_object {
  uint32_t kind; /* RTK_SPHERE | RTK_PLANE */
  uint32_t reserved_;
  double transform[16];
  double plane_normal[4]; /* `Plane::normal` (src/scene/shapes/planes.rs:10-12); ignored for spheres */
  rtk_material material;
} rtk_object;

/* `PointLight` (src/scene/lighting.rs:8-11).  position[3] is the reference's `w`: lights
 * loaded from YAML are built with `vec3` and carry w = 0 (src/loader.rs:35-40), lights
 * built with `point()` carry w = 1.  The value changes shadow rays and Phong shading,
 * so it is part of the interface. */
typedef struct rtk_light {
  double position[4];
  double intensity[3];
} rtk_light;

/* `Scene` (src/scene.rs:102-106). */
typedef struct rtk_scene {
  const rtk_object* objects;
  uint32_t n_objects;
  const rtk_light* lights;
  uint32_t n_lights;
  double ambient_color[3]; /* `Scene::ambient_color`; the reference only ever holds BLACK outside tests */
  uint32_t max_depth;      /* RTK_REFERENCE_MAX_DEPTH reproduces the reference */
  uint32_t reserved_;
} rtk_scene;

/* `Camera` (src/scene/cameras.rs:7-15). */
typedef struct rtk_camera {
  uint32_t width, height;
  double half_width, half_height;
  double field_of_view;
  double pixel_size;
  double transform[16];
} rtk_camera;

/* Ray counts use the reference's unit: one ray = one call of `Scene::intersect`
 * (src/scene.rs:146 for primary and secondary rays, src/scene.rs:217 for shadow rays). */
typedef struct rtk_stats {
  uint64_t rays_primary;
  uint64_t rays_shadow;
  uint64_t rays_secondary;
  uint64_t exact_tests; /* (ray, object) pairs that needed the f64 re-evaluation */
  double kernel_ms;     /* HIP-event time of the ray and shading kernels of the last call */
  double rays_ms;       /* ... of the ray kernels alone (camera + shadow + secondary rays); equals kernel_ms
                           when rays and shading run in one kernel (scenes with reflective or
                           transparent materials, rtk_trace_rays) */
  double gather_ms;     /* rtk_group_get_stats only: HIP-event time of the last frame's exchange on the root
                           device (first receive posted .. last band landed); 0 elsewhere */
} rtk_stats;

/* Which tiles of the frame a call renders.  Tiles are RTK_TILE_W x RTK_TILE_H pixels,
 * numbered row-major over ceil(width/RTK_TILE_W) x ceil(height/RTK_TILE_H).  A call
 * renders the tiles t with t % world == rank.  world == 1 renders the whole frame
 * straight into row-major images.  With world > 1 the output is compact:
 * local tile k (global tile k*world + rank) occupies pixels
 * [k*RTK_TILE_W*RTK_TILE_H, (k+1)*RTK_TILE_W*RTK_TILE_H) in row-major order inside the
 * tile; rtk_untile() rebuilds the frame from the gathered buffers. */
typedef struct rtk_tiling {
  uint32_t rank;
  uint32_t world;
} rtk_tiling;

/* The other partition: whole tile rows.  The tile rows [first_tile_row, first_tile_row +
 * n_tile_rows) of the frame are cut into bands of band_tile_rows tile rows (8 * band_tile_rows
 * pixel rows; the last band may be shorter), and band b goes to rank b % world.  A rank's
 * pixels are then whole rows of the frame: contiguous runs of the final row-major image, so a
 * receiver can land them in place and nothing has to be de-interleaved (pixels are
 * independent, src/scene/cameras.rs:75-82, so any partition renders the same frame).
 *  compact == 0: the outputs are whole W x H frames; the call writes only its own rows, at
 *                their final place.
 *  compact == 1: the outputs hold only this rank's tile rows, one after the other in
 *                ascending order, row-major at the frame's width: rtk_band_pixel_rows()
 *                rows (8 per tile row; rows of the last tile row that lie below the frame
 *                are not written). */
typedef struct rtk_banding {
  uint32_t rank, world;
  uint32_t band_tile_rows; /* >= 1 */
  uint32_t first_tile_row;
  uint32_t n_tile_rows;    /* 0: down to the bottom of the frame */
  uint32_t compact;
} rtk_banding;

typedef struct rtk_context rtk_context;           /* opaque: one HIP device, stream, scratch */
typedef struct rtk_scene_handle rtk_scene_handle; /* opaque: a scene resident in HBM        */

/* `Camera::new` (src/scene/cameras.rs:19-49): derives half_width, half_height and
 * pixel_size from (width, height, field_of_view) and sets the default view
 * look_at((0,1.5,-5), (0,1,0), (0,1,0)). */
int rtk_camera_init(rtk_camera* camera, uint32_t width, uint32_t height, double field_of_view);

/* Number of tiles a rank owns, and the number of pixels of its compact buffer. */
uint64_t rtk_tiles_for_rank(uint32_t width, uint32_t height, uint32_t rank, uint32_t world);
uint64_t rtk_compact_pixels(uint32_t width, uint32_t height, uint32_t rank, uint32_t world);

/* Tile rows a rank owns under a band partition (0 for a bad partition), and the pixel rows of
 * its compact buffer (8 per tile row). */
uint64_t rtk_band_tile_rows(uint32_t height, const rtk_banding* banding);
uint64_t rtk_band_pixel_rows(uint32_t height, const rtk_banding* banding);

/* Binds a context to HIP device `device_id`.  Fails with RTK_ERR_NO_DEVICE when the
 * device is absent or is not gfx950. */
int rtk_create(int device_id, rtk_context** out);
void rtk_destroy(rtk_context* ctx);

/* Flattens a scene for the device and copies it to HBM: the counterpart of building a
 * `Scene` with `add_object` / `add_light` (src/scene.rs:120-133) and of
 * `SceneNode::with_transform` pre-computing the inverse (src/scene.rs:55-62).
 * Supported envelope (anything else returns RTK_ERR_UNSUPPORTED):
 *  - object and camera transforms are affine (last row 0 0 0 1), as every constructor
 *    in src/maths/transforms.rs produces;
 *  - plane_normal[3] == 0 (the loader always passes vec3(0,1,0), src/loader.rs:90);
 *  - max_depth <= RTK_MAX_DEPTH_LIMIT, n_lights <= RTK_MAX_LIGHTS. */
int rtk_scene_upload(rtk_context* ctx, const rtk_scene* scene, rtk_scene_handle** out);
void rtk_scene_release(rtk_context* ctx, rtk_scene_handle* scene);

/* `Camera::render` (src/scene/cameras.rs:72-85) on the device, outputs in DEVICE memory.
 *  d_rgb_f32 : W*H*3 floats, pixel (x, y) at 3*(x + y*W)  -- `Canvas::pixels` layout
 *              (src/graphics.rs:40-46) narrowed to FP32; may be NULL.
 *  d_rgba8   : W*H*4 bytes, `(c * 255.0) as u8` per channel, alpha 255 -- the cast of
 *              `Canvas::to_image` (src/graphics.rs:59-74): truncation toward zero,
 *              saturation at 0 and 255, NaN -> 0; may be NULL.
 *  tiling    : NULL means {0, 1}.  With world > 1 both outputs use the compact layout
 *              described at rtk_tiling and hold rtk_compact_pixels() pixels.
 *  hip_stream: a hipStream_t, or NULL for the context's own stream (which is not ordered
 *              with HIP's default stream).  The call is asynchronous on that stream.  Render calls of one context share per-camera
 *              records and counters; the library orders them: a call on another stream than
 *              the previous call's first waits (on the device) for that call to finish.
 * Precision: object sweeps run a conservative FP32 filter; every candidate the filter
 * keeps, and all shading, is evaluated in f64 with the reference's formulas, so the
 * colour differs from an f64 render only by FMA contraction and summation order before
 * the final narrowing to FP32. */
int rtk_render_device(rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera,
                      const rtk_tiling* tiling, float* d_rgb_f32, uint8_t* d_rgba8,
                      void* hip_stream);

/* rtk_render_device for one rank of a band partition (see rtk_banding).  One sample per pixel. */
int rtk_render_device_bands(rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera,
                            const rtk_banding* banding, float* d_rgb_f32, uint8_t* d_rgba8, void* hip_stream);

/* rtk_render_device that allocates nothing.  It has no counterpart in the reference; it is for
 * callers that cannot call hipMalloc: code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  Everything the kernels need beyond the scene, the camera
 * and the outputs -- the per-camera filter records, the ray counters, and for scenes with a
 * light whose w is neither 0 nor 1 the per-pixel hit records -- lives in a fixed array inside
 * the library's device image (RTK_STATIC_SCRATCH_BYTES per GPU, held until the process ends)
 * instead of memory of the scene handle and the context.  Same arguments, outputs and ray
 * counts as rtk_render_device.  Limits (RTK_ERR_UNSUPPORTED beyond them): at most
 * RTK_STATIC_MAX_SPHERES spheres in the scene; at most RTK_STATIC_MAX_RECORD_PIXELS pixels per
 * call for scenes that need hit records (split larger frames with `tiling`).  Every call on a
 * GPU, from whatever context, shares that GPU's array: calls that overlap on one GPU race.
 * Calls of ONE context are ordered by the library like all its render calls.
 * While hip_stream is recording a HIP graph (hipStreamBeginCapture) the call only adds its kernels
 * and its counter reset to the graph: it records no timing events and does not become the call
 * rtk_get_stats and rtk_get_timing_history describe, and it is not ordered against the context's
 * frames outside the graph (which share the records it rebuilds): that ordering is the caller's. */
#define RTK_STATIC_MAX_SPHERES 32768u
#define RTK_STATIC_MAX_RECORD_PIXELS 131072u
#define RTK_STATIC_SCRATCH_BYTES (3u << 20)
int rtk_render_device_static(rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera,
                             const rtk_tiling* tiling, float* d_rgb_f32, uint8_t* d_rgba8, void* hip_stream);

/* Same render with HOST output buffers (whole frame, row-major); uploads the scene,
 * renders, copies back and synchronises.  stats may be NULL.  The frame is rendered in
 * horizontal blocks and each block travels to the host while the next one renders. */
int rtk_render(rtk_context* ctx, const rtk_scene* scene, const rtk_camera* camera,
               float* rgb_f32_out, uint8_t* rgba8_out, rtk_stats* stats);

/* ---- More than one sample per pixel.  No counterpart in the reference, which traces one ray
 * through each pixel centre (src/scene/cameras.rs:53-54); samples_per_axis == 1 is that.
 * With s = samples_per_axis the s*s samples of pixel (x, y) are the pixel centres
 * (s*x + i, s*y + j), 0 <= i, j < s, of the camera rtk_camera_supersample() derives: the same
 * view at s*width x s*height with pixel_size / s, so every sample is a ray the reference's
 * `Camera::ray_for_pixel` (src/scene/cameras.rs:52-69) would produce for that camera.  The
 * pixel is the mean of its samples' FP32 colours, summed in f64 in row-major order (j outer,
 * i inner) and divided by s*s; the mean is narrowed to FP32 and cast to 8 bits exactly as a
 * single colour is. */
#define RTK_MAX_SAMPLES_PER_AXIS 16
int rtk_camera_supersample(const rtk_camera* camera, uint32_t samples_per_axis, rtk_camera* out);

/* rtk_render_device with s*s samples per pixel.  The sample frame lives in a buffer of the
 * context that grows on demand, so sampled calls of one context must not overlap on
 * different streams.  Ray counts and kernel times of the call cover all samples and the
 * averaging kernel.  tiling: NULL or world == 1 writes row-major frames.  With world > 1
 * (samples_per_axis 1, 2, 4 or 8 only) the rank renders the 32x8 tiles t % world == rank of
 * the SAMPLE frame and averages each into a (32/s) x (8/s) tile of pixels; the outputs hold
 * rtk_compact_pixels_sampled() pixels, tile after tile, row-major inside a tile;
 * rtk_untile_sampled() rebuilds the frame from the gathered buffers. */
int rtk_render_device_sampled(rtk_context* ctx, rtk_scene_handle* scene, const rtk_camera* camera,
                              uint32_t samples_per_axis, const rtk_tiling* tiling, float* d_rgb_f32,
                              uint8_t* d_rgba8, void* hip_stream);
uint64_t rtk_compact_pixels_sampled(uint32_t width, uint32_t height, uint32_t samples_per_axis, uint32_t rank,
                                    uint32_t world);
/* rtk_untile_strided for the buffers of sampled, tiled renders (width, height: the frame's). */
int rtk_untile_sampled(rtk_context* ctx, uint32_t width, uint32_t height, uint32_t samples_per_axis, uint32_t world,
                       uint64_t rgb_rank_stride_bytes, uint64_t rgba8_rank_stride_bytes,
                       const float* d_rgb_f32_compact, float* d_rgb_f32_frame,
                       const uint8_t* d_rgba8_compact, uint8_t* d_rgba8_frame, void* hip_stream);

/* rtk_render with s*s samples per pixel (host outputs). */
int rtk_render_sampled(rtk_context* ctx, const rtk_scene* scene, const rtk_camera* camera, uint32_t samples_per_axis,
                       float* rgb_f32_out, uint8_t* rgba8_out, rtk_stats* stats);

/* The averaging step alone: d_samples_rgb_f32 holds (s*width) x (s*height) x 3 floats (what
 * rtk_render_device writes for the supersampled camera), outputs as rtk_render_device's.
 * Device pointers; either output may be NULL. */
int rtk_resolve(rtk_context* ctx, uint32_t width, uint32_t height, uint32_t samples_per_axis,
                const float* d_samples_rgb_f32, float* d_rgb_f32, uint8_t* d_rgba8, void* hip_stream);

/* `Scene::trace(&self, ray: Ray) -> Color` (src/scene.rs:136-138) for n caller-supplied
 * rays: origins and directions are n x 4 doubles (x, y, z, w), colors_out n x 3 doubles.
 * All three are HOST pointers; the call synchronises.  Origins must have w = 1 and
 * directions w = 0 (what `Camera::ray_for_pixel` produces). */
int rtk_trace_rays(rtk_context* ctx, rtk_scene_handle* scene, const double* origins,
                   const double* directions, uint64_t n, double* colors_out);

/* Rebuilds row-major frames from `world` compact buffers laid end to end in rank order
 * (what a gather to rank 0 produces).  Each rank's slot is `slot_pixels` pixels long
 * (>= rtk_compact_pixels of every rank).  Device pointers; either pair may be NULL. */
int rtk_untile(rtk_context* ctx, uint32_t width, uint32_t height, uint32_t world,
               uint64_t slot_pixels, const float* d_rgb_f32_compact, float* d_rgb_f32_frame,
               const uint8_t* d_rgba8_compact, uint8_t* d_rgba8_frame, void* hip_stream);

/* The same with an explicit distance in bytes between consecutive ranks' buffers, for
 * callers that gather FP32 and RGBA8 tiles of a rank in ONE message (per rank: its FP32
 * tiles, then its RGBA8 tiles; both strides are then the message size).  Strides must be
 * multiples of 4. */
int rtk_untile_strided(rtk_context* ctx, uint32_t width, uint32_t height, uint32_t world,
                       uint64_t rgb_rank_stride_bytes, uint64_t rgba8_rank_stride_bytes,
                       const float* d_rgb_f32_compact, float* d_rgb_f32_frame,
                       const uint8_t* d_rgba8_compact, uint8_t* d_rgba8_frame, void* hip_stream);

/* How scenes with reflective or transparent materials are traced.  No counterpart in the reference
 * (`Scene::trace` recurses, src/scene.rs:136-296); the frame is the same bit for bit either way.
 *   RTK_TRACE_DEPTH_FIRST: one lane walks a pixel's whole ray tree; needs no memory beyond the frame.
 *   RTK_TRACE_BY_LEVEL:    the nodes of all pixels are traced one tree level per launch from a queue in
 *                          device memory (104 bytes per node), the pixels are summed from the node
 *                          records afterwards in depth-first order.  No pixel holds a wavefront up, so
 *                          small and medium launches finish much sooner; large ones pay for the queue.
 *   RTK_TRACE_AUTO:        (default) by level where that is expected to be faster: scenes of more
 *                          than 8 objects, launches up to 64 million pixels (16 million below 256 spheres).
 * max_queue_nodes: capacity of the queue (0 keeps the current one; the default is 64 Mi nodes = 6.5 GB,
 * or the environment variable RTK_WAVEFRONT_NODES read by rtk_create, where 0 turns tracing by level
 * off).  The queue is allocated on first use at 6 nodes per pixel of the launch, up to the capacity;
 * launches that need more run in passes, and a pass whose nodes still do not fit is traced again
 * depth first.  Waits for the context's frames in flight when the capacity changes. */
#define RTK_TRACE_AUTO 0u
#define RTK_TRACE_DEPTH_FIRST 1u
#define RTK_TRACE_BY_LEVEL 2u
int rtk_set_trace_order(rtk_context* ctx, uint32_t order, uint64_t max_queue_nodes);

/* Waits for the stream of the last render and returns its ray counts and kernel time. */
int rtk_get_stats(rtk_context* ctx, rtk_stats* stats);

/* Kernel times (as rtk_stats.kernel_ms / rays_ms) of the most recent rtk_render_device /
 * rtk_trace_rays calls on this context, oldest first: lets a caller queue many frames
 * without waiting on each and read their times afterwards.  Fills min(capacity, calls so
 * far, RTK_TIMING_HISTORY) entries, waits for the newest of them, and returns how many it
 * filled (or an rtk_status < 0).  No counterpart in the reference. */
#define RTK_TIMING_HISTORY 64
int rtk_get_timing_history(rtk_context* ctx, uint32_t capacity, double* kernel_ms, double* rays_ms);

/* ---- One frame over several GPUs of one node, from ONE process.  No counterpart in the
 * reference (its loop is single-threaded, src/scene/cameras.rs:75-82); pixels are independent,
 * so the frame shards with no data-path collective.  Member r of the group renders the bands
 * (rtk_banding, world = n_members) dealt to r; member 0 (the root) writes its rows straight
 * into the caller's frame, every other member renders into a band buffer of its own and sends
 * each band with ncclSend; the root posts the matching ncclRecv with the band's final rows in
 * the caller's frame as the target, so nothing is staged or de-interleaved on the root.  One
 * RCCL communicator per distinct device (ncclCommInitAll), all sends and receives of a frame
 * in one ncclGroupStart / ncclGroupEnd.  Two frames may be in flight: the members' band
 * buffers alternate, so the exchange of frame k overlaps the render of frame k + 1.
 *
 * device_ids may name a device more than once: such members share that device's communicator
 * and exchange through RCCL's send-to-self, which lets a one-GPU machine run every step of the
 * multi-GPU frame.  librccl.so.1 is loaded when the first group is created; a group of one
 * member needs no exchange and does not load it.  If RCCL cannot be loaded or its communicators
 * do not come up, the same bands travel as peer-to-peer copies (hipMemcpyPeerAsync, one per
 * band, on the sending member's stream) into the same final rows; rtk_group_exchange() and
 * rtk_group_info.exchange say which.  The environment variable RTK_GROUP_EXCHANGE, read by
 * rtk_group_create, overrides the choice: "rccl" (RTK_ERR_RCCL rather than a fall-back) or "peer"
 * (copies from the start).  A frame for which only d_rgba8 is given moves 4 bytes per pixel and
 * allocates no FP32 band buffers.  Every rtk_group_* call leaves the calling thread's current HIP
 * device as it found it. */
typedef struct rtk_group rtk_group;
int rtk_group_create(const int* device_ids, int n_members, rtk_group** out);
void rtk_group_destroy(rtk_group* group);
int rtk_group_size(const rtk_group* group);
/* How the members' rows reach the root: "rccl", "peer copies", or "none" (a group of one). */
const char* rtk_group_exchange(const rtk_group* group);
/* Flattens the scene once and copies it to every member's device (replacing the group's
 * previous scene, if any). */
int rtk_group_upload(rtk_group* group, const rtk_scene* scene);
/* Renders one frame into DEVICE buffers on the root's device (device_ids[0]); either may be
 * NULL.  hip_stream: a stream of the root's device, or NULL for the root context's own.
 * Asynchronous: when the call returns the frame is queued, and hip_stream has been made to
 * wait for all of it (the root's own rows and every received band).  A caller that keeps two
 * frames in flight alternates two output buffers. */
int rtk_group_render_device(rtk_group* group, const rtk_camera* camera, float* d_rgb_f32, uint8_t* d_rgba8,
                            void* hip_stream);
/* Waits for the last frame and returns its ray counts summed over the members, the slowest
 * member's kernel times, and gather_ms. */
int rtk_group_get_stats(rtk_group* group, rtk_stats* stats);
/* Tile rows per band the group uses for a frame of `height` pixel rows. */
uint32_t rtk_group_band_tile_rows(const rtk_group* group, uint32_t height);
/* How many bands of a frame each member gets (about; a band is a whole number of tile rows):
 * more bands even out rows that cost differently, fewer mean fewer, larger messages to the root
 * (every band of every other member is one message per output).  0 restores the library's choice:
 * 4 per member, 3 from 4 members on, 2 from 8 members on.  Takes effect from the next frame. */
int rtk_group_set_bands_per_member(rtk_group* group, uint32_t bands_per_member);
/* The root's rows do not travel, so on frames whose exchange takes longer than a member's share the
 * root should render more than the others: with root_share = a the bands are dealt in rounds of
 * n_members - 1 + a, and the root holds the first a of every round (one launch, its rows in place).
 * 0 or 1: equal shares.  Takes effect from the next frame. */
int rtk_group_set_root_share(rtk_group* group, uint32_t root_share);
/* The other partition: ONE run of tile rows per member -- member r renders the tile rows
 * [first_tile_row[r], first_tile_row[r + 1]) of frames of `height` pixel rows (n_members + 1
 * boundaries from 0 to ceil(height / RTK_TILE_H), not decreasing; an empty run is allowed), so every
 * other member's rows reach the root as one message per output.  Frames of another height, and all
 * frames after a call with first_tile_row == NULL, are dealt in bands again. */
int rtk_group_set_rows(rtk_group* group, uint32_t height, const uint32_t* first_tile_row);
/* Lays such boundaries by measurement, for this scene, camera and set of outputs: every member
 * renders a run of rows alone and is timed, and the boundaries are moved until the largest of a
 * member's kernel time and -- for every member but the root -- the time its bytes take on one link is
 * as small as the cost per row measured so far allows (`rounds` measurements; 0: 4).  link_gbps:
 * the rate one member's rows reach the root at, in GB/s (0: the environment variable
 * RTK_GROUP_LINK_GBPS, else 100).  Synchronises; leaves the boundaries set (rtk_group_set_rows). */
int rtk_group_balance(rtk_group* group, const rtk_camera* camera, int with_rgb_f32, int with_rgba8, uint32_t rounds,
                      double link_gbps);
/* The last frame in numbers a scaling run can be read from.  Waits for the frame. */
#define RTK_GROUP_INFO_MEMBERS 16
typedef struct rtk_group_info {
  uint32_t n_members;
  uint32_t exchange;        /* 0 none (one member), 1 RCCL, 2 peer-to-peer copies */
  uint32_t root_share;      /* as set */
  uint32_t rows_mode;       /* 1: boundaries from rtk_group_set_rows / rtk_group_balance are set */
  uint32_t messages;        /* sends (= receives) or copies of the last frame's exchange */
  uint32_t reserved_;
  uint64_t bytes_into_root; /* bytes the exchange moved into the caller's frame */
  double gather_ms;         /* as rtk_stats.gather_ms */
  /* per member, the first RTK_GROUP_INFO_MEMBERS of them: tile rows, kernel time and rays of its share */
  uint32_t share_tile_rows[RTK_GROUP_INFO_MEMBERS];
  double share_ms[RTK_GROUP_INFO_MEMBERS];
  uint64_t share_rays[RTK_GROUP_INFO_MEMBERS];
} rtk_group_info;
int rtk_group_get_info(rtk_group* group, rtk_group_info* info);

/* Flattens a scene as rtk_scene_upload would -- validation, inverses, sweep order, every record
 * array -- WITHOUT touching a device, and returns a 64-bit digest of the result (and, if asked, the
 * bytes the scene would take in HBM).  No counterpart in the reference; it exists so that the host
 * half of the library can be checked without a GPU: two builds, or two threads, that flatten one
 * scene must agree on the digest. */
int rtk_flatten_digest(const rtk_scene* scene, uint64_t* digest, uint64_t* device_bytes);

/* Thread-local description of the last failure on this thread ("" if none). */
const char* rtk_last_error(void);

/* Library version string and the code-object architecture it was built for. */
const char* rtk_version(void);
/* The number of this header's struct layouts and entry points: it changes whenever a struct of
 * this header changes size or layout or an entry point changes meaning.  A host built against
 * this header compares it with RTK_ABI_VERSION before passing structs.  (2: rtk_stats has
 * gather_ms and RTK_ERR_RCCL exists; 3: rtk_group_info, shares and rows of a group,
 * rtk_material.inner_pattern_transforms.) */
#define RTK_ABI_VERSION 3
uint32_t rtk_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RTK_H */
