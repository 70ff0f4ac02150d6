// Drop-in body of the reference's frame drivers and getFinalColor (src/render.h:18-29;
// replaces src/render.cpp): each driver is one rtx_render into the Screen's own pixel
// vector, getFinalColor one rtx_shade.
#include "rtx_binding.h"

#include "light.h"   // sampleSize  (src/light.cpp:14), the reference's own global
#include "render.h"
#include "screen.h"
#include "shading.h" // numberOfRays (src/shading.cpp:9)

#include <glm/gtc/quaternion.hpp>

#include <cstring>
#include <stdexcept>

struct TbHalfHeight { typedef float Trackball::*type; friend type memberOf(TbHalfHeight); };
struct TbHalfWidth { typedef float Trackball::*type; friend type memberOf(TbHalfWidth); };
struct TbFovy { typedef float Trackball::*type; friend type memberOf(TbFovy); };
template struct PrivateMember<TbHalfHeight, &Trackball::m_halfScreenSpaceHeight>;
template struct PrivateMember<TbHalfWidth, &Trackball::m_halfScreenSpaceWidth>;
template struct PrivateMember<TbFovy, &Trackball::m_fovy>;

rtx_camera toRtx(const Trackball& camera)
{
    rtx_camera c;
    std::memset(&c, 0, sizeof(c));
    const glm::vec3 lookAt = camera.lookAt(), euler = camera.rotationEulerAngles(), position = camera.position();
    const glm::quat q(euler); // the rotation Trackball::generateRay applies (host libm, as the reference)
    std::memcpy(c.look_at, &lookAt, 12);
    std::memcpy(c.euler, &euler, 12);
    c.distance = camera.distanceFromLookAt();
    c.fovy = camera.*memberOf(TbFovy());
    c.half_width = camera.*memberOf(TbHalfWidth());
    c.half_height = camera.*memberOf(TbHalfHeight());
    std::memcpy(c.position, &position, 12);
    c.quat[0] = q.w, c.quat[1] = q.x, c.quat[2] = q.y, c.quat[3] = q.z;
    return c;
}

namespace {

void run(const BvhInterface& bvh, const Trackball& camera, Screen& screen, const Features& features, rtx_render_params p)
{
    const glm::ivec2 res = screen.resolution();
    p.width = res.x;
    p.height = res.y;
    p.ray_depth = -1; // the driver's own depth (1, 0, 0, 1, 1)
    p.shadow_samples = sampleSize;
    p.glossy_rays = numberOfRays;
    p.seed = 1;
    const rtx_camera c = toRtx(camera);
    const rtx_features f = toRtx(features);
    // Screen::pixels() is row 0 = top, three packed floats per pixel: exactly what rtx_render writes
    bindThisThread();
    if (rtx_render(deviceSceneOf(bvh), &c, &f, &p, &screen.pixels()[0].x, nullptr) != RTX_OK)
        throw std::runtime_error(rtx_last_error());
}

// state of renderRayTracingMotionBlur between calls (src/render.cpp:185-188)
rtx_camera previousCamera;
bool rendered = false;

} // namespace

void renderRayTracing(const Scene&, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features)
{
    rtx_render_params p {};
    p.driver = RTX_DRIVER_PLAIN;
    run(bvh, camera, screen, features, p);
}

void renderRayTracingTransparency(const Scene&, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features)
{
    rtx_render_params p {};
    p.driver = RTX_DRIVER_TRANSPARENCY;
    run(bvh, camera, screen, features, p);
}

void renderRayTracingDepthOfField(const Scene&, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features, float aperture, float focalLength, int samples)
{
    rtx_render_params p {};
    p.driver = RTX_DRIVER_DOF;
    p.aperture = aperture;
    p.focal_length = focalLength;
    p.samples = samples;
    run(bvh, camera, screen, features, p);
}

void renderRayTracingMRaysPerPixel(const Scene&, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features, int samples)
{
    rtx_render_params p {};
    p.driver = RTX_DRIVER_MRAYS;
    p.samples = samples;
    run(bvh, camera, screen, features, p);
}

void renderRayTracingMotionBlur(const Scene& scene, const Trackball& camera, const BvhInterface& bvh, Screen& screen, const Features& features, int steps)
{
    if (!rendered) { // the first call renders the still frame and remembers the pose (src/render.cpp:194-201)
        renderRayTracing(scene, camera, bvh, screen, features);
        previousCamera = toRtx(camera);
        rendered = true;
        return;
    }
    rtx_render_params p {};
    p.driver = RTX_DRIVER_MOTION_BLUR;
    p.samples = steps;
    p.prev_camera = previousCamera;
    previousCamera = toRtx(camera);
    run(bvh, camera, screen, features, p);
}

glm::vec3 getFinalColor(const Scene&, const BvhInterface& bvh, Ray ray, const Features& features, int rayDepth)
{
    rtx_ray r;
    std::memcpy(r.origin, &ray.origin, 12);
    std::memcpy(r.direction, &ray.direction, 12);
    r.t = ray.t;
    rtx_render_params p {};
    p.ray_depth = rayDepth;
    p.shadow_samples = sampleSize;
    p.glossy_rays = numberOfRays;
    p.seed = 1;
    const rtx_features f = toRtx(features);
    glm::vec3 out;
    bindThisThread();
    if (rtx_shade(deviceSceneOf(bvh), &r, 1, &f, &p, &out.x, nullptr) != RTX_OK) // (n rays per call for batches)
        throw std::runtime_error(rtx_last_error());
    return out;
}
