/*
 * rtx.h -- C ABI of librtx_hip.so, the MI355X (gfx950) implementation of the
 * ray tracer's per-pixel hot path: closest-hit BVH traversal, ray/box and
 * ray/triangle tests, hit attribute interpolation, Phong shading, point /
 * segment / parallelogram light shadow rays, mirror / glossy / transparency
 * bounces, texture and environment-map fetches, and the five frame drivers.
 *
 * The reference (stefan-creasta/RayTracer) has no FFI of its own: its seam for
 * this path is the C++ class BvhInterface (src/bvh_interface.h:12-41) and the
 * free functions of src/render.h:18-29.  Every entry point below names the
 * reference interface it replaces.  All structs are plain data, all pointers
 * are caller-owned, nothing is retained past return unless stated, and no
 * exception crosses this boundary: functions return 0 on success and a
 * negative rtx_status on failure, with text available from rtx_last_error().
 *
 * All arithmetic is IEEE fp32 evaluated in the reference's operation order
 * without fused multiply-add, so hit distances and triangle ids are
 * bit-identical to the CPU path.
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rtx_status {
    RTX_OK = 0,
    RTX_ERR_INVALID = -1,   /* bad argument / inconsistent description          */
    RTX_ERR_DEVICE = -2,    /* a HIP runtime call failed                        */
    RTX_ERR_LIMIT = -3,     /* a documented size limit was exceeded             */
    RTX_ERR_NOT_INIT = -4   /* rtx_init has not succeeded on this thread's GPU  */
} rtx_status;

/* ---- feature switches: Features + ExtraFeatures, src/common.h:61-87 ------- */
typedef struct rtx_features {
    int32_t enable_shading;
    int32_t enable_recursive;
    int32_t enable_hard_shadow;
    int32_t enable_soft_shadow;
    int32_t enable_normal_interp;
    int32_t enable_texture_mapping;
    int32_t enable_accel_structure;
    int32_t enable_environment_mapping;
    int32_t enable_bvh_sah_binning;          /* host builder only */
    int32_t enable_bilinear_texture_filtering;
    int32_t enable_mipmap_texture_filtering;
    int32_t enable_glossy_reflection;
    int32_t reserved[4];
} rtx_features;

/* ---- ray / hit: Ray (framework/include/framework/ray.h:9-13), HitInfo
 *      (src/common.h:14-21) -------------------------------------------------- */
typedef struct rtx_ray {
    float origin[3];
    float direction[3];
    float t;                 /* in: search limit (FLT_MAX for "none"); unchanged on return */
} rtx_ray;

#define RTX_TRI_NONE (-1)
/* sphere i is reported as triangle id -2 - i */
typedef struct rtx_hit {
    float   t;               /* closest hit distance, or the input t on a miss */
    int32_t tri;             /* index in the reference's meshTrianglePairs order
                                (mesh-major, src/bounding_volume_hierarchy.cpp:166-175) */
    float   normal[3];       /* HitInfo::normal as the reference leaves it (un-normalised) */
    float   bary[3];         /* HitInfo::barycentricCoord (area ratios)        */
    float   uv[2];           /* HitInfo::texCoord                              */
    int32_t material;        /* index into rtx_scene_desc::materials, -1 on a miss */
    int32_t hit;             /* 1 = hit, 0 = miss                              */
} rtx_hit;

/* ---- flattened acceleration structure -------------------------------------
 * Built on the host by the reference's builder
 * (src/bounding_volume_hierarchy.cpp:101-180) and flattened to 64-byte records.
 * A child reference r >= 0 is an inner node index; r < 0 is a leaf whose
 * triangles are records [~r, ~r + count) of the leaf-ordered triangle array
 * (the last one also carries RTX_TRI_LAST_IN_LEAF). */
typedef struct rtx_node {
    float   lo0[3], hi0[3];  /* box of child 0 (reference "left")              */
    float   lo1[3], hi1[3];  /* box of child 1 (reference "right")             */
    int32_t ref0, ref1;
    int32_t cnt0, cnt1;      /* triangle count when the child is a leaf        */
} rtx_node;                  /* 64 B */

typedef struct rtx_tri {
    float   v0[3], v1[3], v2[3];
    float   n[3];            /* normalize(cross(v1-v0, v2-v0)), src/intersect.cpp:50-58 */
    float   d;               /* dot(v0, n)                                     */
    int32_t id;              /* meshTrianglePairs index                        */
    int32_t material;
    int32_t flags;           /* bit 0: last triangle of its leaf               */
} rtx_tri;                   /* 64 B */
#define RTX_TRI_LAST_IN_LEAF 1

typedef struct rtx_tri_attr {
    float n0[3], n1[3], n2[3];   /* vertex normals                             */
    float uv0[2], uv1[2], uv2[2];
    float pad;
} rtx_tri_attr;              /* 64 B, same index as rtx_tri */

/* Material, framework/include/framework/mesh.h:22-34 */
typedef struct rtx_material {
    float   kd[3];
    float   ks[3];
    float   shininess;
    float   transparency;
    int32_t texture;         /* index into textures, -1 = none                 */
    int32_t pad[3];
} rtx_material;              /* 48 B */

/* One texture = a chain of mip levels of fp32 RGB texels (level 0 = the image,
 * framework/src/image.cpp:29-32; further levels = 2x2 box filter,
 * src/main.cpp:745-771).  Offsets count texels in rtx_scene_desc::texels. */
#define RTX_MAX_MIP_LEVELS 16
typedef struct rtx_texture {
    int32_t  levels;
    int32_t  width[RTX_MAX_MIP_LEVELS];
    int32_t  height[RTX_MAX_MIP_LEVELS];
    uint32_t offset[RTX_MAX_MIP_LEVELS];
    uint32_t count[RTX_MAX_MIP_LEVELS];   /* texels stored for the level */
} rtx_texture;

/* Lights, src/common.h:44-59.  POINT uses p[0]=position, c[0]=color.
 * SEGMENT: p[0],p[1]=endpoints, c[0],c[1].  PARALLELOGRAM: p[0]=v0,
 * p[1]=edge01, p[2]=edge02, c[0..3]. */
typedef enum rtx_light_type { RTX_LIGHT_POINT = 0, RTX_LIGHT_SEGMENT = 1, RTX_LIGHT_PARALLELOGRAM = 2 } rtx_light_type;
typedef struct rtx_light {
    int32_t type;
    float   p[3][3];
    float   c[4][3];
    int32_t pad[2];
} rtx_light;                 /* 96 B */

/* Sphere, src/common.h:38-42 */
typedef struct rtx_sphere {
    float   center[3];
    float   radius;
    int32_t material;
    int32_t pad[3];
} rtx_sphere;

/* EnvironmentMap, src/environment_mapping.h:9-33 */
typedef enum rtx_env_type { RTX_ENV_CYLINDRICAL = 0, RTX_ENV_SPHERICAL = 1 } rtx_env_type;
typedef struct rtx_envmap {
    int32_t texture;         /* index into textures, -1 = constant background  */
    int32_t mapping;         /* rtx_env_type                                   */
    float   vfov_radians;    /* verticalFOVFactor                              */
    float   background[3];
} rtx_envmap;

typedef struct rtx_scene_desc {
    const rtx_node*     nodes;       uint32_t n_nodes;
    int32_t             root_ref;    /* node index, leaf reference, or INT32_MIN for an empty tree */
    int32_t             root_cnt;
    float               root_lo[3], root_hi[3];
    const rtx_tri*      tris;        /* leaf order                              */
    const rtx_tri_attr* tri_attrs;   uint32_t n_tris;
    const rtx_material* materials;   uint32_t n_materials;
    const rtx_texture*  textures;    uint32_t n_textures;
    const float*        texels;      uint64_t n_texels;   /* RGB triplets       */
    const rtx_light*    lights;      uint32_t n_lights;
    const rtx_sphere*   spheres;     uint32_t n_spheres;
    rtx_envmap          envmap;
    int32_t             num_levels;  /* BvhInterface::numLevels                 */
    int32_t             num_leaves;  /* BvhInterface::numLeaves                 */
    /* EnvironmentMap::radianceBins (src/environment_mapping.cpp:42-105): rectangles of the
     * environment image as (lower.x, lower.y, upper.x, upper.y), built on the host; the
     * device draws its environment-light samples from them (src/light.cpp:47-66).
     * May be empty: a constant-colour environment has none. */
    const float*        env_bins;    uint32_t n_env_bins;
} rtx_scene_desc;

/* ---- camera: Trackball state (framework/src/trackball.cpp:64-110) --------- */
typedef struct rtx_camera {
    float look_at[3];
    float euler[3];          /* radians                                        */
    float distance;
    float fovy;              /* radians                                        */
    float half_width;        /* aspect * tan(fovy/2)                           */
    float half_height;       /* tan(fovy/2)                                    */
    float position[3];       /* look_at + quat(euler) * (0,0,-distance)        */
    float quat[4];           /* w,x,y,z of glm::quat(euler), host libm         */
} rtx_camera;

/* ---- frame drivers: src/render.h:18-26 ------------------------------------ */
typedef enum rtx_driver {
    RTX_DRIVER_PLAIN = 0,         /* renderRayTracing               src/render.cpp:69-90   */
    RTX_DRIVER_MRAYS = 1,         /* renderRayTracingMRaysPerPixel  src/render.cpp:152-183 */
    RTX_DRIVER_DOF = 2,           /* renderRayTracingDepthOfField   src/render.cpp:92-126  */
    RTX_DRIVER_TRANSPARENCY = 3,  /* renderRayTracingTransparency   src/render.cpp:129-150 (any number of
                                     translucent layers per camera ray; see rtx_stats.layer_overflows) */
    RTX_DRIVER_MOTION_BLUR = 4,   /* renderRayTracingMotionBlur     src/render.cpp:190-253 (second and later calls) */
    /* The four sampling drivers above in ONE frame.  The reference's router makes them mutually
     * exclusive (src/main.cpp:50-63), so the combination is this library's own: per pixel
     * time_steps x lens_samples x samples camera samples, sample s = (t * lens_samples + l) * samples + j,
     * each built from the drivers' own pieces in the order time, pixel, lens:
     *   time_steps > 1    the pose of src/render.cpp:224-235 at w = (jitter + t) / time_steps (one draw)
     *   samples > 1       the pixel position of src/render.cpp:170-174 (two draws)
     *   lens_samples > 1  the origin and direction of src/dof.cpp:9-39 about this sample's own camera ray (three draws)
     *   transparency != 0 calculateColorTransparency on the sample's ray (src/transparency.cpp:6-20), else getFinalColor
     * A dimension of size 1 is off and draws nothing; the draws come from the reference's
     * sequential engine, sample after sample.  The sum runs in sample order and is scaled like
     * src/render.cpp:178 (times 1 / count), like :249 (divided by count) when only time_steps is on,
     * and not at all for a single sample.  So each driver is the composition with the other
     * dimensions off, bit for bit: samples alone = MRAYS, lens_samples alone = DOF, time_steps alone =
     * MOTION_BLUR, transparency alone = TRANSPARENCY, nothing = PLAIN. */
    RTX_DRIVER_COMPOSED = 5
} rtx_driver;

typedef struct rtx_render_params {
    int32_t  width, height;
    int32_t  driver;          /* rtx_driver                                     */
    int32_t  ray_depth;       /* getFinalColor's rayDepth; -1 = the driver's own (1,0,0,1,1; COMPOSED: 0 with
                                 samples or lens_samples on, as the reference's sampling drivers pass, else 1) */
    int32_t  samples;         /* spp / lens samples / blur steps; <=0 = the driver's default (64,100,7; COMPOSED: 1) */
    float    aperture;        /* DoF, default 0.1                               */
    float    focal_length;    /* DoF, default 2.0                               */
    int32_t  shadow_samples;  /* sampleSize,   src/light.cpp:14   (50)          */
    int32_t  glossy_rays;     /* numberOfRays, src/shading.cpp:9  (5)           */
    uint32_t seed;            /* key of the counter RNG used for light / glossy draws */
    /* interleaved-tile partition: this call renders tiles t with
     * t % tile_world == tile_rank; tile_world <= 1 renders the whole frame.   */
    int32_t  tile_size;       /* square tile edge in pixels, 0 = default (32)   */
    int32_t  tile_rank, tile_world;
    rtx_camera prev_camera;   /* motion blur: camera of the previous call       */
    /* RTX_DRIVER_COMPOSED only (ignored by the other drivers); <= 0 counts as 1 */
    int32_t  lens_samples;    /* lens positions per pixel                       */
    int32_t  time_steps;      /* poses of the exposure per pixel                */
    int32_t  transparency;    /* != 0: the transparency chain on every camera sample's ray */
    int32_t  reserved[5];
} rtx_render_params;

typedef struct rtx_stats {
    uint64_t rays_primary;    /* BvhInterface::intersect-equivalent queries, by kind */
    uint64_t rays_shadow;
    uint64_t rays_bounce;
    uint64_t box_tests;       /* child boxes tested by the device traversal     */
    uint64_t tri_tests;       /* triangle records tested by the device traversal*/
    uint64_t shaded_hits;
    double   kernel_ms;       /* device time of the render / trace kernel (HIP events) */
    double   total_ms;        /* wall time of the call                          */
    uint64_t rays_traced;     /* of the rays counted above, those that walked the scene on the
                                 device: camera rays of 8x8 blocks that cannot see the scene's
                                 bounds are counted (the reference makes and tests them) but never
                                 made, and a light sample behind the surface needs no ray */
    uint64_t layer_overflows; /* transparency chains (RTX_DRIVER_TRANSPARENCY, RTX_DRIVER_COMPOSED) deeper than 16
                                 translucent surfaces: the first 16 layers are folded innermost first exactly as
                                 src/transparency.cpp:6-20 does, the deeper ones are summed front to back -- the
                                 same colour up to fp32 rounding of the order of summation; a chain has no
                                 bound, as in the reference */
} rtx_stats;

typedef struct rtx_scene rtx_scene;   /* device-resident scene, reference-counted */

/* Select the GPU this thread uses and create its stream.  Replaces nothing in
 * the reference (it has no device); must precede every other call. */
int rtx_init(int device);

/* Upload a flattened scene once.  Replaces the tail of
 * BvhInterface::BvhInterface (src/bvh_interface.cpp:6-9): the host has built
 * the tree, this copies it to HBM.  Returns NULL on failure.
 * Beside the records as given the scene keeps, for each point light among its first four, a copy
 * of the traversal records relative to the light's position (the shadow rays of such a light all
 * start there; about twice the size of the triangle records per light), made by a kernel here. */
rtx_scene* rtx_scene_create(const rtx_scene_desc* desc);
/* The same scene with its tree built on the GPU instead.  Replaces the whole of
 * BoundingVolumeHierarchy's constructor and bvhSplitHelper
 * (src/bounding_volume_hierarchy.cpp:101-180) for callers that cannot wait for the host
 * builder (10 M triangles): desc->tris / desc->tri_attrs hold the triangles in id order
 * (record i has id i, planes filled in, flags ignored), desc->nodes and root_* are ignored.
 * The tree is a linear BVH over Morton codes with leaves of up to four triangles (more where
 * triangles crowd), NOT the reference's tree: closest hits, and with them every frame, are
 * the same except where two triangles tie at exactly the same distance; numLevels /
 * numLeaves are this tree's (see rtx_scene_tree_info).  Returns NULL on failure. */
rtx_scene* rtx_scene_build_device(const rtx_scene_desc* desc);
/* BvhInterface::numLevels / numLeaves (src/bvh_interface.cpp:12-23) of the tree the scene
 * holds: the description's numbers after rtx_scene_create, the built tree's after
 * rtx_scene_build_device. */
typedef struct rtx_tree_info {
    int32_t levels, leaves;
    float   upload_ms;  /* rtx_scene_build_device: copying the triangle records to the GPU (0 otherwise) */
    float   build_ms;   /* rtx_scene_build_device: sorting, hierarchy, boxes, records (0 otherwise)       */
} rtx_tree_info;
int rtx_scene_tree_info(const rtx_scene* scene, rtx_tree_info* info);
/* BvhInterface is trivially copyable in the reference (src/main.cpp:218
 * re-assigns it), so handles are reference-counted. */
void rtx_scene_retain(rtx_scene* scene);
void rtx_scene_destroy(rtx_scene* scene);

/* Batch form of BvhInterface::intersect (src/bvh_interface.cpp:48-51 ->
 * src/bounding_volume_hierarchy.cpp:320-458): n closest-hit queries.
 * rays/hits are host buffers. */
int rtx_trace(const rtx_scene* scene, const rtx_ray* rays, size_t n, rtx_hit* hits,
              const rtx_features* features, rtx_stats* stats);
/* Same, with rays/hits already resident in HBM (device pointers). */
int rtx_trace_device(const rtx_scene* scene, const rtx_ray* d_rays, size_t n, rtx_hit* d_hits,
                     const rtx_features* features, rtx_stats* stats);

/* One frame: the five renderRayTracing* drivers (src/render.cpp:69-253) with
 * getFinalColor (src/render.cpp:16-66) underneath.  out_rgb is a host buffer of
 * width*height*3 floats in Screen layout, row 0 = top (src/screen.cpp:37-43).
 * With tile_world > 1 only this rank's tiles are written; the rest is left
 * untouched. */
int rtx_render(const rtx_scene* scene, const rtx_camera* camera, const rtx_features* features,
               const rtx_render_params* params, float* out_rgb, rtx_stats* stats);
/* Same with the framebuffer in HBM (device pointer); no host copy is made. */
int rtx_render_device(const rtx_scene* scene, const rtx_camera* camera, const rtx_features* features,
                      const rtx_render_params* params, float* d_out_rgb, rtx_stats* stats);
/* rtx_render_device for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Everything the kernels need beyond the scene, the parameters and the
 * framebuffer (work-queue counters, statistics, the sampler's skip-ahead tables, the per-block
 * costs and order of the frame scheduler) lives in one fixed array of about 3 MB per GPU inside
 * the library, so the call makes no hipMalloc / hipFree and no host allocation either.
 * With stats == NULL the call only queues kernels and memsets on the current stream (the
 * sampler tables of a new frame shape are made by a kernel, not copied from the host), waits
 * for nothing, and may be captured into a HIP graph.  With a stats pointer it additionally
 * waits for the frame and copies the counters to the host, which a capture cannot hold:
 * while the stream is being captured such a call is refused with RTX_ERR_INVALID.
 * Limits: width and height <= 4096 and at most 262 144 blocks of 8x8 pixels (tiles padded to
 * tile_size); beyond them RTX_ERR_LIMIT.  Every call on a GPU shares that GPU's array, which
 * keeps the previous call's block measurements: calls are serialised inside the library and
 * ordered on the current stream (rtx_set_stream); frames issued on different streams at
 * once would race.  Same frames, bit for bit, as rtx_render_device -- but not always at its
 * speed: the memory of the three-pass form of frames with area lights grows with the frame and
 * the camera-relative records of the pinhole drivers are allocated per scene, so this entry
 * point renders such frames with the one-kernel form and the plain records (measured at
 * 1920x1080: one parallelogram light with 64 samples 3.8 against 2.6 ms, with glossy
 * reflections to depth 4 82 against 17 ms, the headline frame 0.67 against 0.59 ms). */
int rtx_render_device_static(const rtx_scene* scene, const rtx_camera* camera, const rtx_features* features,
                             const rtx_render_params* params, float* d_out_rgb, rtx_stats* stats);
/* Same, but this rank's tiles are written compactly: tile k of this rank
 * (k-th tile with index % tile_world == tile_rank, row-major tile order) occupies
 * floats [k*tile_size*tile_size*3, ...) of d_tiles, rows top-down inside the
 * tile.  This is the buffer the multi-GPU gather moves. */
int rtx_render_tiles_device(const rtx_scene* scene, const rtx_camera* camera, const rtx_features* features,
                            const rtx_render_params* params, float* d_tiles, rtx_stats* stats);
/* Scatter gathered tile buffers (tile_world consecutive buffers of
 * rtx_tiles_per_rank*tile_size^2*3 floats each, rank-major) into a
 * Screen-layout frame.  Both pointers are device pointers. */
int rtx_detile_device(const float* d_all_tiles, const rtx_render_params* params, float* d_out_rgb);
/* Batch form of getFinalColor (src/render.h:29, src/render.cpp:16-66): the colour
 * carried by each of n caller-supplied rays.  Of params only ray_depth (<0 = the
 * reference's default 0), shadow_samples, glossy_rays and seed are read; ray i
 * draws its light / glossy samples from the stream keyed (seed, i, 0).  rays and
 * out_rgb (n x 3 floats) are host buffers; the _device form takes device pointers. */
int rtx_shade(const rtx_scene* scene, const rtx_ray* rays, size_t n, const rtx_features* features,
              const rtx_render_params* params, float* out_rgb, rtx_stats* stats);
int rtx_shade_device(const rtx_scene* scene, const rtx_ray* d_rays, size_t n, const rtx_features* features,
                     const rtx_render_params* params, float* d_out_rgb, rtx_stats* stats);

/* Number of tiles each rank's compact buffer holds (the maximum over ranks). */
int rtx_tiles_per_rank(const rtx_render_params* params);

/* Queue the work of this thread's frame slot on the caller's HIP stream (a hipStream_t, e.g.
 * the framework stream the device buffers are produced and consumed on) instead
 * of the library's own; NULL switches back.  With a stats pointer of NULL the
 * *_device entry points only enqueue and return, so frames can be pipelined
 * with the caller's other work on that stream.  The setting belongs to the frame slot
 * (rtx_select_slot; slot 0 unless another was selected), not to the thread: it applies to every
 * thread that works in the same slot of the same GPU.  Work still
 * queued on the previous stream is ordered before whatever is queued on the new one (an event
 * on the old stream that the new one waits for), because all calls in a slot share its work
 * counters and scheduling buffers. */
int rtx_set_stream(void* hip_stream);

/* Frames in flight.  A GPU has RTX_FRAME_SLOTS frame slots; each holds its own stream, work
 * counters, statistics, scheduling buffers and deferred-frame pools, so calls made in different
 * slots share nothing but the (read-only) scene and may run on the GPU at the same time: the
 * next frame's waves fill the compute units that the last, longest work units of the previous
 * frame leave idle.  The counterpart in the reference is its command line's one thread per
 * camera (src/main.cpp:560-585).  rtx_select_slot makes `slot` the one this THREAD's following
 * calls work in (created on first use; rtx_init selects slot 0).  Calls within one slot stay
 * serialised and ordered on its stream; ordering between slots is the caller's (events on the
 * slots' streams, or rtx_synchronize in each).  rtx_render_device_static works in slot 0 only. */
#define RTX_FRAME_SLOTS 4
int rtx_select_slot(int slot);
/* Whether kernels queued on two of the caller's HIP streams run at the same time: 1 = yes, 0 =
 * they take turns, < 0 = error.  The HIP runtime spreads a process's streams over a few hardware
 * queues, and streams that share a queue serialise, so frames in different slots only share the
 * GPU when the slots' streams pass this test (the library's own streams are chosen with it; a
 * caller that hands its streams to rtx_set_stream picks them with it).  Measured with two
 * 300-microsecond single-wave kernels; waits for both streams.  No counterpart in the reference. */
int rtx_streams_overlap(void* hip_stream_a, void* hip_stream_b);

/* Page-locked host memory for buffers that travel to or from the GPU (a framebuffer handed to
 * rtx_render, ray and hit arrays handed to rtx_trace): copies run at the bus's full rate
 * instead of through the runtime's staging buffer.  No counterpart in the reference, whose
 * Screen is plain host memory (src/screen.h:16-33); any host memory is accepted everywhere, this
 * is only faster.  Needs no rtx_init; returns NULL when no GPU or no such memory is available. */
void* rtx_host_alloc(size_t bytes);
void  rtx_host_free(void* buffer);

/* Self-check of the device's trigonometry.  Two places of the path turn an angle into a discrete
 * outcome: the motion-blur driver builds one camera per sample from interpolated Euler angles
 * (src/render.cpp:203-247, glm::quat(euler)), and the environment map turns a direction into
 * texture coordinates (src/environment_mapping.cpp:130-207).  The reference takes the sines,
 * cosines, tangents, arc cosines and arc tangents there from the host C library; the device
 * computes them by restatements of that library's routines, and this call returns what they
 * give: for i < n, out[i] = sin(x[i]), out[n + i] = cos(x[i]), out[2n + i] = tan(x[i]),
 * out[3n + i] = acos(x[i]), out[4n + i] = atan2(y[i], x[i]) (device pointers; out holds 5 n
 * floats), so that a test can hold them to the host's, bit for bit.  Runs in this thread's frame
 * slot, on its stream; waits for the result.  No counterpart in the reference. */
int rtx_selfcheck_trig(const float* x_device, const float* y_device, size_t n, float* out_device);

/* Block until all work queued in this thread's frame slot has finished. */
int rtx_synchronize(void);

/* Text of the most recent failure on the calling thread ("" if none). */
const char* rtx_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
