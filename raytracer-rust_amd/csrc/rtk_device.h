// rtk_device.h -- data layout shared by the host side (rtk_host.cpp, rtk_api.hip) and the kernels
// (rtk_kernels.hip, rtk_opaque.hip).  Everything here lives in HBM and is read-only during a render.
// Host code that is not compiled by hipcc (rtk_host.cpp) sees the same structs: the vector types come
// from HIP's header of plain structs, which needs neither the runtime nor a device compiler.
#ifndef RTK_DEVICE_H
#define RTK_DEVICE_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else
#include <hip/hip_vector_types.h>
#endif
#include <stdint.h>

namespace rtk {

constexpr int kBlock = 256;      // threads per workgroup = one 32x8 pixel tile (4 waves of 8x8)
constexpr int kGroup = 8;        // records per wavefront early-out branch (one ray per lane)
constexpr int kBundle = 64;      // records per wavefront-level test (one sphere per lane); arrays are padded to this
constexpr int kSuper = 8;        // clusters of kBundle spheres per ball of the upper level
constexpr int kMaxDepth = 8;     // == RTK_MAX_DEPTH_LIMIT
constexpr uint32_t kMaxLights = 32767;  // == RTK_MAX_LIGHTS

constexpr float kEps32 = 5.9604645e-8f;  // 2^-24, unit round-off of FP32

// One scene object in f64: what the exact intersection test and the shading need.
// `inv` holds rows 0..2 of the inverse transform (row 3 is 0 0 0 1 for the affine
// transforms the library accepts).  Objects keep the caller's order: the index in this
// array is the object's identity for tie-breaks (src/scene.rs:325-340, 164-166).
struct DevObject {
  double inv[12];
  double color_a[3];
  double color_b[3];
  double ambient, diffuse, specular, shininess;
  double transparency, reflectivity, refractivity;
  double plane_normal[3];
  uint32_t kind;
  uint32_t texture_kind;  // bits 0-7: the kind; bit 31: the pattern has a transform of its own, see DevScene::pattern_inv; bits 8-11 / 12-30: length / first matrix of its chain in DevScene::pattern_chain when it is wrapped more than once
};

// What the exact test of one sphere needs, in sweep (cluster) order like the filter records, so
// that a candidate's record is found from its position j without going through sph_object.
// For a sphere whose inverse transform is a diagonal 3x3 plus a translation (every sphere placed by
// translate * scale, whatever the three scale factors) the six numbers below ARE the inverse: the
// other six entries of its rows are zeros, and leaving their products out of the reference's row
// sums (src/maths/vectors.rs:205-223) changes no sum except, at most, the sign of a zero, which no
// decision of the intersection depends on.  Other spheres keep the full rows in DevObject::inv.
struct SphRecord {
  double scale[3];    // inv[0], inv[5], inv[10]
  double shift[3];    // inv[3], inv[7], inv[11]
  uint32_t object;    // index into DevScene::objects
  uint32_t diagonal;  // 1: scale/shift are the whole inverse; 0: use objects[object].inv
  uint64_t reserved_;
};
static_assert(sizeof(SphRecord) == 64, "one sphere record is one 64-byte scalar load");

struct DevLight {
  double pos[4];  // position incl. the reference's w (src/scene/lighting.rs:8-11)
  double intensity[3];
  double reserved_;
};

// Sphere record of the FP32 filter used by shadow and secondary rays:
//   x, y, z : bounding-sphere centre minus DevScene::center, rounded to FP32
//   w       : inflated squared radius (covers centre rounding), rounded up
// Record of the primary-ray filter (rebuilt per camera by prep_primary_kernel):
//   x, y, z : centre minus camera origin;  w : |L|^2 (1 - slack) - R'^2, rounded down
// Both arrays (and sph_object) are padded to a multiple of kBundle by repeating the last
// real record; testing a sphere twice changes no decision.  The one-ray-per-lane sweeps stop
// at n_spheres_grouped.
// The record arrays of a scene are read-only for every kernel that sees a DevScene.  In device
// code their pointers carry the constant address space: a load through one whose address is the
// same in every lane (an index taken from a ballot, a loop counter) then becomes a scalar load --
// the record arrives in scalar registers, through the scalar cache, and costs no vector
// registers; without the qualifier the compiler cannot prove the memory unwritten and issues a
// 64-lane vector load of one address.  Loads with per-lane addresses are vector loads either way.
// Same pointer size and value on both sides, so host and device agree on the layout.
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
#define RTK_RO __attribute__((address_space(4)))
#else
#define RTK_RO
#endif

// Shadow rays towards a light with w = 0 all run along one direction g, so "which spheres can
// block this point" is a question about the point's projection along g: a grid over the plane
// perpendicular to g, each cell listing the spheres whose shadow can reach some point above it.
// (The shadow of a sphere widens away from the light -- the reference's shadow ray has a w
// component, src/scene.rs:208-225 -- so the lists are built for points with (p - centre).g >= h_min,
// where every shading point on a sphere lies; points below take the cluster sweeps.)
// The lists are built from geometry -- discs a little larger (1e-4 relative, 1e-5 of the scene absolute)
// than the spheres' shadows.  The reference's f64 quadratic, evaluated for a point at distance D, carries
// a rounding error of some 1e-15 D^2 in its discriminant and so reports roots for lines that miss a sphere
// by up to that much (squared): geometry stops being a superset of "the reference finds a root" once
// that exceeds the discs' extra size, at D of about a hundred scene sizes.  Points further away than
// that take the cluster sweeps too, whose FP32 slack grows with the point's distance.
struct LightGrid {
  float e1[3], e2[3];        // orthonormal basis of the plane perpendicular to g
  float h_min;
  float u0, v0, inv_cell;    // cell (i, j) covers u0 + [i, i+1) / inv_cell, v0 + [j, j+1) / inv_cell
  uint32_t nx, ny;
  uint32_t cell_base;        // first cell of this light in DevScene::grid_cells
  float d_max2;              // squared distance from DevScene::center up to which the lists are supersets (below)
  uint32_t reserved_[2];
};
static_assert(sizeof(LightGrid) == 64, "one light grid header is 64 bytes");

// One cell of the grid that rays with a direction of their own walk: where its list lies, how long it
// is, and the first two entries of the list themselves -- most lists are no longer, and a lane that
// enters the cell then has all it needs after one round trip to memory.
struct TraceCell {
  uint32_t first, count;  // entries first .. first + count - 1 of DevScene::tg_filter / tg_slot
  uint32_t slot[2];       // sweep positions of the first two (unused ones: 0)
  float4 filter[2];       // ... and their filter records
};
static_assert(sizeof(TraceCell) == 48, "a cell record is three 16-byte loads");

struct DevScene {
  const RTK_RO DevObject* objects;
  const RTK_RO DevLight* lights;
  const RTK_RO float4* sph_filter;
  const RTK_RO double4* sph_exact;     // centre (world, f64) and bounding radius per filter record
  const RTK_RO uint32_t* sph_object;   // filter record -> object index
  const RTK_RO SphRecord* sph_record;  // filter record -> what its exact test needs (padded like the filter records)
  const RTK_RO uint32_t* plane_object; // plane list -> object index
  const RTK_RO double2* plane_light;   // [plane][light]: row 1 of the plane's inverse applied to (L.xyz, L.w - 1), and 1e-12 x the sum of that row's term magnitudes (plane_may_block)
  const RTK_RO double* pattern_inv;    // per object, rows 0..2 of the inverse pattern transform (12 f64); null when no pattern has one
  const RTK_RO double* pattern_chain;  // 16 f64 per matrix: the inverses, outer to inner, of patterns inside several TransformPatterns
  const RTK_RO float4* light_filter;   // per light: w = 0: unit line direction g = L/|L| and rho = -1/|L|; w = 1: position minus DevScene::center
  const RTK_RO uint32_t* light_kind;   // per light: 0 (w == 0), 1 (w == 1), 2 (anything else)
  const RTK_RO float4* k_table;        // [batch][sphere][k_width]: c_j.g_k - R_j'^2 rho_k, only when lights_uniform
  const RTK_RO float4* sph_tab;        // {cx, cy, cz, kappa |c|^2 - R'^2}: sphere record of the tabulated shadow filter
  const RTK_RO float4* sph_ball;       // {cx, cy, cz, R'}: record of the wavefront-level test; padding has R' = -1e30
  const RTK_RO float4* cluster_ball;   // one ball around each aligned run of kBundle spheres; padded to kBundle with R = -1e30
  const RTK_RO float4* super_ball;     // one ball around each aligned run of kSuper clusters
  const RTK_RO unsigned long long* cluster_plain; // per cluster, bit i: member i is a sphere of one radius (its inverse is sigma * identity plus a translation)
  const RTK_RO LightGrid* light_grid;  // per light, when shadow_grid
  const RTK_RO uint2* grid_cells;      // {first entry, entries} per cell, the lights' grids one after the other
  const RTK_RO uint32_t* grid_entries; // sweep positions of spheres
  const RTK_RO float4* grid_filter;    // ... and, entry by entry, their sph_filter records
  // The cell grid walked by rays with a direction of their own, when trace_grid: cubic cells of side
  // 1 / tg_inv_h from the corner tg_lo (world coordinates), tg_n cells per axis, x fastest; a cell lists
  // the spheres whose bounding ball (plus a margin) reaches into it: entries first .. first + count - 1.
  const RTK_RO TraceCell* tg_cells;    // per cell: its list's place and length, and the list's first two entries
  const RTK_RO float4* tg_filter;      // per entry: the sphere's sph_filter record
  const RTK_RO uint32_t* tg_slot;      // per entry: the sphere's sweep position
  uint32_t n_objects, n_lights, n_spheres, n_spheres_padded, n_planes;
  uint32_t n_clusters_padded;
  uint32_t n_spheres_grouped;  // n_spheres rounded up to kGroup: what the one-ray-per-lane sweeps cover
  uint32_t max_depth;
  uint32_t track_n1;     // some transparent object has refractivity != the others: keep n1
  uint32_t any_secondary; // some material reflects or transmits: rays recurse
  uint32_t lights_uniform; // every light has w == 0: shadow-ray line directions do not depend on the shading point
  uint32_t lights_bundled; // every light has w == 0 or w == 1: the shadow rays of a patch form lines / cones
  uint32_t k_width;        // lights per k_table row (4 or 8)
  uint32_t shadow_grid;    // the light grids exist (lights_uniform scenes with secondary rays and a cluster hierarchy)
  uint32_t trace_grid;     // the cell grid exists
  uint32_t lights_point;   // every light has w == 1 (built from a point): shadow rays are plain segments
  uint32_t tg_n[3];
  uint32_t reserved_;
  double tg_lo[3], tg_inv_h;
  double center[3];      // recentring offset of the FP32 filter
  double ambient[3];
  float r2_max;          // max inflated squared radius
  float r_max;           // max inflated radius
  float c_max;           // max |centre - center|
};

struct DevCamera {
  uint32_t width, height;
  double half_width, half_height, pixel_size;
  double inv[12];     // rows 0..2 of the inverse camera transform (hoisted out of the pixel loop)
  double origin[3];   // inverse * point(0,0,0)
};

// Ray counters.  Every wavefront adds its counts with atomics; one address takes about one
// atomic per 12 ns, so the counters are sharded (workgroup b adds to shard b % kCounterShards,
// the host sums the shards) -- a short kernel of 10^5 wavefronts would otherwise spend its
// whole run time queueing on one word.
constexpr int kCounterShards = 256;
// Counter buffers hold one element more: the work counter of launches with persistent wavefronts.
constexpr int kCounterSlots = kCounterShards + 1;
constexpr uint32_t kPersistentGroups = 2048;  // workgroups of such a launch at most (256 CUs x 4 resident x 2)
struct DevCounters {
  unsigned long long rays_primary, rays_shadow, rays_secondary, exact_tests;
};
#if defined(__HIPCC__)
__device__ __forceinline__ DevCounters* counter_shard(DevCounters* base) { return base + (blockIdx.x % kCounterShards); }
#endif

// Level-by-level tracing of scenes with secondary rays (rtk_wavefront.hip): a queued ray, what
// its node contributes, and the buffers of a context.  Node i of a pass is a camera ray for
// i < n_roots (no queue entry: the ray is generated from the pixel), else queue entry i - n_roots;
// outs[i] is its result.
// Queue entries lie in blocks of 64: within a block every field is a run of 64 values, so the 64 lanes of a
// wavefront, which take consecutive entries, read and write whole cache lines per field.  Fields (f64): origin,
// direction, the factor with which the ray enters its pixel.
constexpr uint32_t kRayOrigin = 0, kRayDirection = 3, kRayWeight = 6, kRayFields = 7;
constexpr size_t kRayBlockBytes = 64 * 8 * kRayFields;
static_assert(kRayBlockBytes % 128 == 0, "blocks start on cache lines");
struct WfNodeOut {                              // what the final walk over a pixel's tree reads of a node: half a line
  double t[3];                                  // the node's own term; zero when it has none (a sum that started at +0 is never
                                                // -0, so adding +0 changes nothing)
  uint32_t child_refraction, child_reflection;  // node indices; 0xffffffff: none; 0xfffffffe: cut off by the depth limit, weight in cuts[]
};
struct WfNodeCut {                              // beside it, read only where a child index says so
  double w_refraction, w_reflection;            // weights of the children cut off by the depth limit
};
static_assert(sizeof(WfNodeOut) == 32 && sizeof(WfNodeCut) == 16, "result records: half a 64-byte line, a quarter");
constexpr size_t kWavefrontNodeBytes = kRayBlockBytes / 64 + sizeof(WfNodeOut) + sizeof(WfNodeCut);  // device memory per queue node
constexpr uint32_t kWavefrontCtrlWords = 16;        // nodes so far, overflow flag, one level end per depth
constexpr uint32_t kWavefrontNodesPerPixel = 6;     // queue room a pass reserves per pixel (its camera ray included), to begin with
struct WavefrontBuffers {
  unsigned char* rays = nullptr;  // cap_nodes entries in blocks of 64 (kRayBlockBytes each)
  WfNodeOut* outs = nullptr;    // cap_nodes entries
  WfNodeCut* cuts = nullptr;    // cap_nodes entries, behind outs in the same allocation
  unsigned int* ctrl = nullptr;  // kWavefrontCtrlWords words
  DevCounters* counters = nullptr;  // kCounterShards shards: the launch's own ray counts
  unsigned int* host_overflow = nullptr;  // page-locked host word, set when a launch ran out of queue room
  uint32_t cap_nodes = 0;
  uint32_t nodes_per_pixel = kWavefrontNodesPerPixel;  // queue room a pass reserves per pixel
};

// Intermediate records of the three-kernel path, one entry per pixel of the launch
// (local tile k, slot s -> index k*256 + s).
struct OpaqueBuffers {
  uint32_t* hit_obj;  // object index of the closest hit, 0xffffffff = miss
  double* hit_t;      // its ray parameter
  uint16_t* shadow;   // 1: the hit is in shadow (the shadow-ray kernel counts its rays itself)
};

struct RenderTarget {
  float* rgb;        // may be null
  uint8_t* rgba8;    // may be null
  uint32_t tiles_x, tiles_y;
  uint32_t n_local_tiles;
  uint32_t layout;  // 0: row-major frame; 1: the rank's tiles one after the other (tile-major); 2: the rank's tile rows one after the other
  // Local tile k of the launch is entry g = k * step + offset of a row-major walk over tile rows
  // of tiles_x tiles; row q = g / tiles_x of that walk is row q % band_rows of the launch's band
  // q / band_rows, and that band starts at tile row row0 + (q / band_rows) * band_period + band_offset:
  // the launch owns band_rows tile rows out of every band_period, from band_offset on.
  //   tiles dealt round-robin to `world` ranks: step = world, offset = rank, band_rows = 1, band_period = 1, band_offset = 0
  //   bands of m tile rows dealt round-robin:   step = 1, offset = 0, band_rows = m, band_period = world * m, band_offset = rank * m
  //   a rank that holds a of every v bands:     band_rows = a * m, band_period = v * m, band_offset = (its first band) * m
  //   one run of rows [r0, r0 + n):             band_rows = n, band_period = n, band_offset = 0, row0 = r0
  uint32_t step, offset;
  uint32_t band_rows, band_period, band_offset, row0;
};

struct TileRef {
  uint32_t tx, ty;  // tile coordinates in the frame
  uint32_t lrow;    // row of the launch's own walk (the local tile row in band mode)
};
#if defined(__HIPCC__)
__device__ __forceinline__ TileRef tile_ref(const RenderTarget& T, uint32_t k) {
  TileRef r;
  const uint32_t g = k * T.step + T.offset;
  r.tx = g % T.tiles_x;
  r.lrow = g / T.tiles_x;
  r.ty = T.row0 + (r.lrow / T.band_rows) * T.band_period + T.band_offset + r.lrow % T.band_rows;
  return r;
}
#endif

}  // namespace rtk
#endif
