// The reference's hot-path API, backed by the MI355X library.  Same class and
// function names, argument order and meaning as the reference so that its
// callers (src/main.cpp:50-63,159,218,226,313,509,546,573) compile against this
// header unchanged apart from the namespace:
//
//   BvhInterface                       src/bvh_interface.h:12-41
//   Trackball (camera subset)          framework/include/framework/trackball.h:14-59
//   Screen    (framebuffer subset)     src/screen.h:10-33
//   renderRayTracing*, getFinalColor   src/render.h:18-29
//
// All work runs on the GPU through the C ABI of include/rtx.h.  There is no CPU
// fallback: if no device can be initialised, BvhInterface's constructor throws.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "bvh_build.h"
#include "rt_types.h"
#include "rtx.h"

namespace rt {

// Thrown where the reference would have thrown std::exception or called exit
// (missing files, bad scene): here also for device errors.
struct RtError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Bind the calling thread to a GPU (rtx_init).  BvhInterface's constructor calls
// selectDevice(0) itself if the thread has not chosen one.
void selectDevice(int device);
// The frame slot (0 .. RTX_FRAME_SLOTS - 1) the calling thread's frames are queued in
// (rtx_select_slot).  Frames of different slots run on the GPU at the same time, which is how
// threads that each render a camera of their own (src/main.cpp:560-585) share one GPU.
void selectFrameSlot(int slot);

// Process-wide sampling knobs, globals in the reference: sampleSize
// (src/light.cpp:14) and numberOfRays (src/shading.cpp:9).
extern int sampleSize;
extern int numberOfRays;
// Key of the counter RNG that replaces the reference's std::random_device draws.
extern uint32_t samplingSeed;

// Where BvhInterface's constructor builds the tree: false (default) on the host, with the
// reference's builder, so that numLevels / numLeaves and the tree are the reference's; true
// on the GPU (rtx_scene_build_device: a linear BVH, ~100x faster for millions of triangles,
// same closest hits, its own numLevels / numLeaves).  RTX_DEVICE_BUILD=1 in the environment
// turns it on as well.
extern bool buildTreeOnDevice;

class BvhInterface {
public:
    // Builds the tree (on the host with the reference's builder unless buildTreeOnDevice),
    // flattens it and uploads it once.  pScene must outlive the object (as in the
    // reference, which keeps the raw pointer).
    BvhInterface(Scene* pScene, const Features& features);
    BvhInterface(const BvhInterface& other);
    BvhInterface& operator=(const BvhInterface& other);
    ~BvhInterface();

    [[nodiscard]] int numLevels() const;
    [[nodiscard]] int numLeaves() const;
    // Visual-debug hooks of the reference (OpenGL); no-ops here.
    void debugDrawLevel(int level);
    void debugDrawLeaf(int leafIdx);

    // Closest hit of one ray; ray.t is the search limit on entry and the hit
    // distance on return.  One device launch per call: use intersectBatch for
    // more than a handful of rays.
    bool intersect(Ray& ray, HitInfo& hitInfo, const Features& features) const;
    // n rays at once; hits[i] as rtx_hit (include/rtx.h).  Not in the reference.
    void intersectBatch(const rtx_ray* rays, size_t n, rtx_hit* hits, const Features& features, rtx_stats* stats = nullptr) const;

    [[nodiscard]] const rtx_scene* deviceScene() const { return m_scene; }
    [[nodiscard]] const FlatScene& flat() const { return *m_flat; }
    [[nodiscard]] Scene* scene() const { return m_pScene; }

private:
    Scene* m_pScene;
    rtx_scene* m_scene;
    std::shared_ptr<FlatScene> m_flat;
};

// Camera state and ray generation of the reference's Trackball
// (framework/src/trackball.cpp:64-110).  The window is reduced to its aspect
// ratio (Window::getAspectRatio, framework/src/window.cpp:379-384).
class Trackball {
public:
    Trackball(float aspectRatio, float fovy, float distanceFromLookAt = 4.0f, float rotationX = 0.0f, float rotationY = 0.0f);
    Trackball(float aspectRatio, float fovy, const vec3& lookAt, float distanceFromLookAt = 4.0f, float rotationX = 0.0f, float rotationY = 0.0f);
    [[nodiscard]] vec3 position() const;
    [[nodiscard]] vec3 lookAt() const { return m_lookAt; }
    [[nodiscard]] vec3 rotationEulerAngles() const { return m_rotationEulerAngles; }
    [[nodiscard]] float distanceFromLookAt() const { return m_distanceFromLookAt; }
    void setCamera(const vec3 lookAt, const vec3 rotations, const float dist);
    [[nodiscard]] Ray generateRay(const vec2& pixel) const;
    // the plain-data camera the device consumes
    [[nodiscard]] rtx_camera deviceCamera() const;

private:
    float m_fovy, m_halfScreenSpaceHeight, m_halfScreenSpaceWidth;
    vec3 m_lookAt { 0.0f, 0.0f, 0.0f };
    float m_distanceFromLookAt;
    vec3 m_rotationEulerAngles { 0.f, 0.f, 0.f };
};

// Storage of a Screen's pixels: a std::vector whose memory is page-locked (rtx_host_alloc)
// where a GPU is present, so that a frame comes back from the device at the bus's full
// rate, and plain memory otherwise.
void* pixelStoreAlloc(size_t bytes);
void pixelStoreFree(void* p) noexcept;
template <class T>
struct PixelAllocator {
    using value_type = T;
    PixelAllocator() = default;
    template <class U>
    PixelAllocator(const PixelAllocator<U>&) noexcept {}
    T* allocate(size_t n) { return static_cast<T*>(pixelStoreAlloc(n * sizeof(T))); }
    void deallocate(T* p, size_t) noexcept { pixelStoreFree(p); }
    template <class U>
    bool operator==(const PixelAllocator<U>&) const noexcept { return true; }
    template <class U>
    bool operator!=(const PixelAllocator<U>&) const noexcept { return false; }
};
using PixelStore = std::vector<vec3, PixelAllocator<vec3>>;

// fp32 RGB framebuffer in the reference's layout: row 0 is the top row
// (src/screen.cpp:37-43).
class Screen {
public:
    explicit Screen(const ivec2& resolution, bool presentable = false);
    void clear(const vec3& color);
    void setPixel(int x, int y, const vec3& color);
    [[nodiscard]] ivec2 resolution() const { return m_resolution; }
    [[nodiscard]] int indexAt(int x, int y) const { return (m_resolution.y - 1 - y) * m_resolution.x + x; }
    [[nodiscard]] const PixelStore& pixels() const { return m_textureData; }
    [[nodiscard]] PixelStore& pixels() { return m_textureData; }
    // 4-channel BMP, clamp then truncate to 8 bits (src/screen.cpp:45-56)
    void writeBitmapToFile(const std::string& filePath) const;

private:
    ivec2 m_resolution;
    PixelStore m_textureData;
};

// src/render.h:18-29.  `scene` must be the scene the BvhInterface was built from.
void renderRayTracing(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features);
void renderRayTracingTransparency(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features);
void renderRayTracingDepthOfField(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features, float aperture = 0.1f, float focalLength = 2.0f, int samples = 100);
void renderRayTracingMRaysPerPixel(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features, int samples = 64);
// Stateful like the reference: the first call renders a still frame and
// remembers the camera; later calls blur between the remembered and the current one.
void renderRayTracingMotionBlur(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features, int steps = 7);
// Forget the remembered motion-blur camera (the reference never does).
void resetMotionBlur();
// The four sampling drivers above in ONE frame: per pixel timeSteps x lensSamples x jitterSamples camera
// samples (a count of 1 switches its dimension off), the transparency chain on every sample's ray if asked
// for.  The reference's router (src/main.cpp:50-63) picks one driver even when several of
// enableDepthOfField / enableMotionBlur / enableTransparency / enableMultipleRaysPerPixel are set; this is
// the frame in which all that are set take effect.  Each driver above is this one with the other
// dimensions off, bit for bit (see RTX_DRIVER_COMPOSED in rtx.h).  Shares renderRayTracingMotionBlur's
// remembered camera: with timeSteps > 1 the first call renders the exposure as a single pose.
struct ComposedSampling {
    int jitterSamples = 1, lensSamples = 1, timeSteps = 1;
    bool transparency = false;
    float aperture = 0.1f, focalLength = 2.0f; // src/render.h:22
    int rayDepth = -1;                         // -1: 0 with jitter or lens samples on (as the reference's sampling drivers pass), else 1
};
void renderRayTracingComposed(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features, const ComposedSampling& how);
// renderRayTracingRouter of src/main.cpp:50-63 with one change: two or more of the four switches set
// render the composed frame (the drivers' default counts: 64 jitter samples, 100 lens samples, 7 poses)
// instead of the first driver in the reference's priority order.
void renderRayTracingRouter(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features);

// src/render.h:29: the colour carried by one ray.  One device launch per call;
// getFinalColorBatch shades n rays at once (ray i samples lights / glossy
// directions from the stream keyed (samplingSeed, i, 0)).
vec3 getFinalColor(const Scene& scene, const BvhInterface& bvh, Ray ray, const Features& features, int rayDepth = 0);
void getFinalColorBatch(const BvhInterface& bvh, const rtx_ray* rays, size_t n, const Features& features, int rayDepth, vec3* out, rtx_stats* stats = nullptr);

// General form behind the five drivers: any driver with an explicit ray depth
// (the reference's drivers hard-code 0 or 1), optional interleaved-tile
// partition, statistics.
void renderFrame(const BvhInterface& bvh, const Trackball& camera, Screen& screen, const Features& features, rtx_render_params params, rtx_stats* stats = nullptr);
// The same into a caller's buffer of width x height RGB triplets in Screen layout.
void renderInto(const BvhInterface& bvh, const Trackball& camera, int width, int height, float* rgb, const Features& features, rtx_render_params params, rtx_stats* stats = nullptr);

rtx_features toDeviceFeatures(const Features& features);

} // namespace rt
