#include "raytracer.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <unordered_set>

namespace rt {

int sampleSize = 50;
int numberOfRays = 5;
uint32_t samplingSeed = 1;

namespace {

thread_local bool t_deviceChosen = false;

[[noreturn]] void fail(const char* what)
{
    throw RtError(std::string(what) + ": " + rtx_last_error());
}

void ensureDevice()
{
    if (!t_deviceChosen)
        selectDevice(0);
}

} // namespace

void selectDevice(int device)
{
    if (rtx_init(device) != RTX_OK)
        fail("rtx_init");
    t_deviceChosen = true;
}

void selectFrameSlot(int slot)
{
    ensureDevice();
    if (rtx_select_slot(slot) != RTX_OK)
        fail("rtx_select_slot");
}

rtx_features toDeviceFeatures(const Features& f)
{
    rtx_features o;
    std::memset(&o, 0, sizeof(o));
    o.enable_shading = f.enableShading;
    o.enable_recursive = f.enableRecursive;
    o.enable_hard_shadow = f.enableHardShadow;
    o.enable_soft_shadow = f.enableSoftShadow;
    o.enable_normal_interp = f.enableNormalInterp;
    o.enable_texture_mapping = f.enableTextureMapping;
    o.enable_accel_structure = f.enableAccelStructure;
    o.enable_environment_mapping = f.extra.enableEnvironmentMapping;
    o.enable_bvh_sah_binning = f.extra.enableBvhSahBinning;
    o.enable_bilinear_texture_filtering = f.extra.enableBilinearTextureFiltering;
    o.enable_mipmap_texture_filtering = f.extra.enableMipmapTextureFiltering;
    o.enable_glossy_reflection = f.extra.enableGlossyReflection;
    return o;
}

bool buildTreeOnDevice = false;

// ------------------------------------------------------------ BvhInterface --
BvhInterface::BvhInterface(Scene* pScene, const Features& features)
    : m_pScene(pScene)
    , m_scene(nullptr)
{
    if (!pScene)
        throw RtError("BvhInterface: null scene");
    ensureDevice();
    const char* env = std::getenv("RTX_DEVICE_BUILD");
    const bool onDevice = buildTreeOnDevice || (env && env[0] == '1');
    m_flat = std::make_shared<FlatScene>(flattenScene(*pScene, features, !onDevice));
    const rtx_scene_desc d = m_flat->desc();
    m_scene = onDevice ? rtx_scene_build_device(&d) : rtx_scene_create(&d);
    if (!m_scene)
        fail(onDevice ? "rtx_scene_build_device" : "rtx_scene_create");
    if (onDevice) {
        // numLevels / numLeaves are the built tree's; the triangle records are not needed again
        rtx_tree_info info {};
        rtx_scene_tree_info(m_scene, &info);
        m_flat->numLevels = info.levels;
        m_flat->numLeaves = info.leaves;
        m_flat->tris.clear();
        m_flat->tris.shrink_to_fit();
        m_flat->attrs.clear();
        m_flat->attrs.shrink_to_fit();
    }
}

BvhInterface::BvhInterface(const BvhInterface& o)
    : m_pScene(o.m_pScene)
    , m_scene(o.m_scene)
    , m_flat(o.m_flat)
{
    rtx_scene_retain(m_scene);
}

BvhInterface& BvhInterface::operator=(const BvhInterface& o)
{
    if (this != &o) {
        rtx_scene_retain(o.m_scene);
        rtx_scene_destroy(m_scene);
        m_pScene = o.m_pScene;
        m_scene = o.m_scene;
        m_flat = o.m_flat;
    }
    return *this;
}

BvhInterface::~BvhInterface() { rtx_scene_destroy(m_scene); }

int BvhInterface::numLevels() const { return m_flat->numLevels; }
int BvhInterface::numLeaves() const { return m_flat->numLeaves; }
void BvhInterface::debugDrawLevel(int) {}
void BvhInterface::debugDrawLeaf(int) {}

void BvhInterface::intersectBatch(const rtx_ray* rays, size_t n, rtx_hit* hits, const Features& features, rtx_stats* stats) const
{
    ensureDevice();
    const rtx_features f = toDeviceFeatures(features);
    if (rtx_trace(m_scene, rays, n, hits, &f, stats) != RTX_OK)
        fail("rtx_trace");
}

bool BvhInterface::intersect(Ray& ray, HitInfo& hitInfo, const Features& features) const
{
    rtx_ray r;
    std::memcpy(r.origin, &ray.origin, 12);
    std::memcpy(r.direction, &ray.direction, 12);
    r.t = ray.t;
    rtx_hit h;
    intersectBatch(&r, 1, &h, features);
    if (!h.hit)
        return false;
    ray.t = h.t;
    std::memcpy(&hitInfo.normal, h.normal, 12);
    if (h.tri >= 0) {
        // triangleIntersectUpdate, src/bounding_volume_hierarchy.cpp:277-295
        Mesh& mesh = m_pScene->meshes[size_t(m_flat->idMesh[size_t(h.tri)])];
        std::memcpy(&hitInfo.barycentricCoord, h.bary, 12);
        std::memcpy(&hitInfo.texCoord, h.uv, 8);
        hitInfo.material = mesh.material;
        hitInfo.mesh = &mesh;
        hitInfo.triangle = mesh.triangles[size_t(m_flat->idLocal[size_t(h.tri)])];
    } else {
        // sphere: the reference sets normal and material only (src/intersect.cpp:120-123)
        hitInfo.material = m_pScene->spheres[size_t(-2 - h.tri)].material;
    }
    return true;
}

// --------------------------------------------------------------- Trackball --
Trackball::Trackball(float aspect, float fovy, float dist, float rx, float ry)
    : Trackball(aspect, fovy, vec3(0.0f), dist, rx, ry)
{
}
Trackball::Trackball(float aspect, float fovy, const vec3& lookAt, float dist, float rx, float ry)
    : m_fovy(fovy)
    , m_halfScreenSpaceHeight(std::tan(fovy / 2.0f))
    , m_halfScreenSpaceWidth(aspect * m_halfScreenSpaceHeight)
    , m_lookAt(lookAt)
    , m_distanceFromLookAt(dist)
    , m_rotationEulerAngles(rx, ry, 0)
{
}
void Trackball::setCamera(const vec3 lookAt, const vec3 rotations, const float dist)
{
    m_lookAt = lookAt;
    m_rotationEulerAngles = rotations;
    m_distanceFromLookAt = dist;
}

namespace {
struct Quat {
    float w, x, y, z;
};
// glm::quat(eulerAngles), glm/detail/type_quat.inl:208-217
Quat quatFromEuler(vec3 e)
{
    const vec3 h = e * 0.5f;
    const vec3 c(std::cos(h.x), std::cos(h.y), std::cos(h.z));
    const vec3 s(std::sin(h.x), std::sin(h.y), std::sin(h.z));
    return Quat { c.x * c.y * c.z + s.x * s.y * s.z, s.x * c.y * c.z - c.x * s.y * s.z, c.x * s.y * c.z + s.x * c.y * s.z, c.x * c.y * s.z - s.x * s.y * c.z };
}
// quat * vec3, glm/detail/type_quat.inl:347-354
vec3 rotate(Quat q, vec3 v)
{
    const vec3 qv(q.x, q.y, q.z);
    const vec3 uv = cross(qv, v);
    const vec3 uuv = cross(qv, uv);
    return v + ((uv * q.w) + uuv) * 2.0f;
}
} // namespace

vec3 Trackball::position() const
{
    return m_lookAt + rotate(quatFromEuler(m_rotationEulerAngles), vec3(0, 0, -m_distanceFromLookAt));
}
Ray Trackball::generateRay(const vec2& pixel) const
{
    const vec3 dirCam = normalize(vec3(-pixel.x * m_halfScreenSpaceWidth, pixel.y * m_halfScreenSpaceHeight, 1.0f));
    Ray ray;
    ray.origin = position();
    ray.direction = rotate(quatFromEuler(m_rotationEulerAngles), dirCam);
    ray.t = std::numeric_limits<float>::max();
    return ray;
}
rtx_camera Trackball::deviceCamera() const
{
    rtx_camera c;
    std::memset(&c, 0, sizeof(c));
    std::memcpy(c.look_at, &m_lookAt, 12);
    std::memcpy(c.euler, &m_rotationEulerAngles, 12);
    c.distance = m_distanceFromLookAt;
    c.fovy = m_fovy;
    c.half_width = m_halfScreenSpaceWidth;
    c.half_height = m_halfScreenSpaceHeight;
    const vec3 p = position();
    std::memcpy(c.position, &p, 12);
    const Quat q = quatFromEuler(m_rotationEulerAngles);
    c.quat[0] = q.w, c.quat[1] = q.x, c.quat[2] = q.y, c.quat[3] = q.z;
    return c;
}

// ------------------------------------------------------------------ Screen --
Screen::Screen(const ivec2& resolution, bool)
    : m_resolution(resolution)
    , m_textureData(size_t(resolution.x) * size_t(resolution.y), vec3(0.0f))
{
}
// Page-locked blocks handed out by pixelStoreAlloc (the others came from operator new)
namespace {
std::mutex g_pinnedMutex;
std::unordered_set<void*> g_pinned;
}
void* pixelStoreAlloc(size_t bytes)
{
    if (void* p = rtx_host_alloc(bytes)) {
        std::lock_guard<std::mutex> lock(g_pinnedMutex);
        g_pinned.insert(p);
        return p;
    }
    return ::operator new(bytes);
}
void pixelStoreFree(void* p) noexcept
{
    if (!p)
        return;
    bool pinned = false;
    {
        std::lock_guard<std::mutex> lock(g_pinnedMutex);
        pinned = g_pinned.erase(p) != 0;
    }
    if (pinned)
        rtx_host_free(p);
    else
        ::operator delete(p);
}
void Screen::clear(const vec3& color) { std::fill(m_textureData.begin(), m_textureData.end(), color); }
void Screen::setPixel(int x, int y, const vec3& color) { m_textureData[size_t(indexAt(x, y))] = color; }
void Screen::writeBitmapToFile(const std::string& filePath) const
{
    // 32-bit BGRA bottom-up BMP of clamp(color,0,1)*255 truncated (src/screen.cpp:45-56)
    const int W = m_resolution.x, H = m_resolution.y;
    const uint32_t dataSize = uint32_t(W) * uint32_t(H) * 4u, fileSize = 54u + dataSize;
    std::vector<uint8_t> buf(fileSize, 0);
    auto put32 = [&](size_t off, uint32_t v) { std::memcpy(&buf[off], &v, 4); };
    auto put16 = [&](size_t off, uint16_t v) { std::memcpy(&buf[off], &v, 2); };
    buf[0] = 'B', buf[1] = 'M';
    put32(2, fileSize);
    put32(10, 54);
    put32(14, 40);
    put32(18, uint32_t(W));
    put32(22, uint32_t(H));
    put16(26, 1);
    put16(28, 32);
    put32(34, dataSize);
    auto q = [](float v) { return uint8_t(std::min(1.0f, std::max(0.0f, v)) * 255.0f); };
    for (int row = 0; row < H; row++) // BMP rows run bottom-up; Screen row 0 is the top
        for (int x = 0; x < W; x++) {
            const vec3& c = m_textureData[size_t(H - 1 - row) * size_t(W) + size_t(x)];
            uint8_t* px = &buf[54 + (size_t(row) * size_t(W) + size_t(x)) * 4];
            px[0] = q(c.z), px[1] = q(c.y), px[2] = q(c.x), px[3] = 255;
        }
    FILE* fp = std::fopen(filePath.c_str(), "wb");
    if (!fp)
        throw RtError("cannot open " + filePath);
    const size_t wrote = std::fwrite(buf.data(), 1, buf.size(), fp);
    std::fclose(fp);
    if (wrote != buf.size())
        throw RtError("short write to " + filePath);
}

// ----------------------------------------------------------------- drivers --
void renderInto(const BvhInterface& bvh, const Trackball& camera, int width, int height, float* rgb, const Features& features, rtx_render_params params, rtx_stats* stats)
{
    ensureDevice();
    params.width = width;
    params.height = height;
    if (params.shadow_samples <= 0)
        params.shadow_samples = sampleSize;
    if (params.glossy_rays <= 0)
        params.glossy_rays = numberOfRays;
    const rtx_features f = toDeviceFeatures(features);
    const rtx_camera cam = camera.deviceCamera();
    if (rtx_render(bvh.deviceScene(), &cam, &f, &params, rgb, stats) != RTX_OK)
        fail("rtx_render");
}

void renderFrame(const BvhInterface& bvh, const Trackball& camera, Screen& screen, const Features& features, rtx_render_params params, rtx_stats* stats)
{
    ensureDevice();
    const ivec2 res = screen.resolution();
    static_assert(sizeof(vec3) == 12, "vec3 must be three packed floats");
    renderInto(bvh, camera, res.x, res.y, &screen.pixels()[0].x, features, params, stats);
}

namespace {
rtx_render_params baseParams(int driver)
{
    rtx_render_params p;
    std::memset(&p, 0, sizeof(p));
    p.driver = driver;
    p.ray_depth = -1;
    p.seed = samplingSeed;
    return p;
}
// src/render.cpp:185-188
struct BlurState {
    bool rendered = false;
    rtx_camera start;
};
BlurState g_blur;
} // namespace

void getFinalColorBatch(const BvhInterface& bvh, const rtx_ray* rays, size_t n, const Features& features, int rayDepth, vec3* out, rtx_stats* stats)
{
    ensureDevice();
    rtx_render_params p = baseParams(RTX_DRIVER_PLAIN);
    p.ray_depth = rayDepth;
    p.shadow_samples = sampleSize;
    p.glossy_rays = numberOfRays;
    const rtx_features f = toDeviceFeatures(features);
    if (rtx_shade(bvh.deviceScene(), rays, n, &f, &p, n ? &out[0].x : nullptr, stats) != RTX_OK)
        fail("rtx_shade");
}

vec3 getFinalColor(const Scene&, const BvhInterface& bvh, Ray ray, const Features& features, int rayDepth)
{
    rtx_ray r;
    std::memcpy(r.origin, &ray.origin, 12);
    std::memcpy(r.direction, &ray.direction, 12);
    r.t = ray.t;
    vec3 out;
    getFinalColorBatch(bvh, &r, 1, features, rayDepth, &out);
    return out;
}

void renderRayTracing(const Scene&, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features)
{
    renderFrame(bvh, camera, screen, features, baseParams(RTX_DRIVER_PLAIN));
}
void renderRayTracingTransparency(const Scene&, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features)
{
    renderFrame(bvh, camera, screen, features, baseParams(RTX_DRIVER_TRANSPARENCY));
}
void renderRayTracingDepthOfField(const Scene&, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features, float aperture, float focalLength, int samples)
{
    rtx_render_params p = baseParams(RTX_DRIVER_DOF);
    p.aperture = aperture;
    p.focal_length = focalLength;
    p.samples = samples;
    renderFrame(bvh, camera, screen, features, p);
}
void renderRayTracingMRaysPerPixel(const Scene&, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features, int samples)
{
    rtx_render_params p = baseParams(RTX_DRIVER_MRAYS);
    p.samples = samples;
    renderFrame(bvh, camera, screen, features, p);
}
void renderRayTracingMotionBlur(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features, int steps)
{
    if (!g_blur.rendered) {
        renderRayTracing(scene, camera, bvh, screen, features);
        g_blur.start = camera.deviceCamera();
        g_blur.rendered = true;
        return;
    }
    rtx_render_params p = baseParams(RTX_DRIVER_MOTION_BLUR);
    p.samples = steps;
    p.prev_camera = g_blur.start;
    g_blur.start = camera.deviceCamera();
    renderFrame(bvh, camera, screen, features, p);
}
void resetMotionBlur() { g_blur.rendered = false; }

void renderRayTracingComposed(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features, const ComposedSampling& how)
{
    // (the exposure's poses extrapolate from the pose change since the previous call, exactly as in
    // renderRayTracingMotionBlur, whose state this shares: its first call renders a still frame)
    rtx_render_params p = baseParams(RTX_DRIVER_COMPOSED);
    p.samples = how.jitterSamples;
    p.lens_samples = how.lensSamples;
    p.time_steps = how.timeSteps;
    p.transparency = how.transparency ? 1 : 0;
    p.aperture = how.aperture;
    p.focal_length = how.focalLength;
    p.ray_depth = how.rayDepth;
    if (how.timeSteps > 1) {
        if (!g_blur.rendered) {
            p.time_steps = 1;
            g_blur.rendered = true;
        } else {
            p.prev_camera = g_blur.start;
        }
        g_blur.start = camera.deviceCamera();
    }
    (void)scene;
    renderFrame(bvh, camera, screen, features, p);
}
void renderRayTracingRouter(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features)
{
    const int on = int(features.extra.enableDepthOfField) + int(features.extra.enableMotionBlur) + int(features.extra.enableTransparency) + int(features.extra.enableMultipleRaysPerPixel);
    if (on >= 2) {
        ComposedSampling how;
        how.jitterSamples = features.extra.enableMultipleRaysPerPixel ? 64 : 1; // the drivers' default counts (src/render.h:22-26)
        how.lensSamples = features.extra.enableDepthOfField ? 100 : 1;
        how.timeSteps = features.extra.enableMotionBlur ? 7 : 1;
        how.transparency = features.extra.enableTransparency;
        renderRayTracingComposed(scene, camera, bvh, screen, features, how);
    } else if (features.extra.enableDepthOfField) {
        renderRayTracingDepthOfField(scene, camera, bvh, screen, features);
    } else if (features.extra.enableMotionBlur) {
        renderRayTracingMotionBlur(scene, camera, bvh, screen, features);
    } else if (features.extra.enableTransparency) {
        renderRayTracingTransparency(scene, camera, bvh, screen, features);
    } else if (features.extra.enableMultipleRaysPerPixel) {
        renderRayTracingMRaysPerPixel(scene, camera, bvh, screen, features);
    } else {
        renderRayTracing(scene, camera, bvh, screen, features);
    }
}

} // namespace rt
