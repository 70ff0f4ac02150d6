// TEST INFRASTRUCTURE (oracle/_ref build only).
//
// The reference's hot-path translation units are compiled unmodified from
// /root/reference.  Three of its other translation units cannot be compiled in
// this image because they include <GL/glu.h>, which the image does not have
// (src/draw.cpp, src/screen.cpp, framework/src/trackball.cpp), and no stand-in
// for that header is written.  The hot path links against a handful of symbols
// from them; this file defines those symbols itself:
//
//   * the visual-debug hooks of src/draw.h:18-28 as no-ops (they draw with
//     OpenGL and return nothing, so they cannot influence any computed value);
//   * Screen's framebuffer methods (src/screen.h:10-33) without the GL texture;
//   * the Trackball methods the frame drivers call (trackball.h:14-59), written
//     against the reference's own vendored glm so the quaternion arithmetic is
//     glm's.  These are this harness's statement of
//     framework/src/trackball.cpp:64-110, NOT reference code: primary-ray
//     generation is therefore pinned by restatement only.
#include "draw.h"
#include "screen.h"
#include <framework/trackball.h>
#include <glm/gtc/quaternion.hpp>
#include <cmath>
#include <limits>

bool enableDebugDraw = false;

void drawExampleOfCustomVisualDebug() {}
void drawRay(const Ray&, const glm::vec3&) {}
void drawAABB(const AxisAlignedBox&, DrawMode, const glm::vec3&, float) {}
void drawTriangle(const Vertex&, const Vertex&, const Vertex&) {}
void drawMesh(const Mesh&) {}
void drawSphere(const Sphere&) {}
void drawSphere(const glm::vec3&, float, const glm::vec3&) {}
void drawScene(const Scene&) {}

// ---- Screen: CPU framebuffer only -----------------------------------------
Screen::Screen(const glm::ivec2& resolution, bool presentable)
    : m_presentable(false)
    , m_resolution(resolution)
    , m_textureData(size_t(resolution.x) * size_t(resolution.y), glm::vec3(0.0f))
    , m_texture(0)
{
    (void)presentable;
}
void Screen::clear(const glm::vec3& color)
{
    for (auto& p : m_textureData)
        p = color;
}
void Screen::setPixel(int x, int y, const glm::vec3& color)
{
    m_textureData[size_t(indexAt(x, y))] = color;
}
glm::ivec2 Screen::resolution() const { return m_resolution; }
int Screen::indexAt(int x, int y) const { return (m_resolution.y - 1 - y) * m_resolution.x + x; }
const std::vector<glm::vec3>& Screen::pixels() const { return m_textureData; }
std::vector<glm::vec3>& Screen::pixels() { return m_textureData; }

// ---- Trackball -------------------------------------------------------------
// The harness passes a pointer to a float holding the aspect ratio where the
// reference passes a Window* (Window needs GLFW and a display).
Trackball::Trackball(Window* pWindow, float fovy, const glm::vec3& lookAt, float distFromLookAt, float rotationX, float rotationY)
    : m_pWindow(pWindow)
    , m_fovy(fovy)
    , m_halfScreenSpaceHeight(std::tan(fovy / 2.0f))
    , m_halfScreenSpaceWidth(*reinterpret_cast<const float*>(pWindow) * m_halfScreenSpaceHeight)
    , m_lookAt(lookAt)
    , m_distanceFromLookAt(distFromLookAt)
    , m_rotationEulerAngles(rotationX, rotationY, 0)
{
}
void Trackball::setCamera(const glm::vec3 lookAt, const glm::vec3 rotations, const float dist)
{
    m_lookAt = lookAt;
    m_rotationEulerAngles = rotations;
    m_distanceFromLookAt = dist;
}
glm::vec3 Trackball::lookAt() const { return m_lookAt; }
glm::vec3 Trackball::rotationEulerAngles() const { return m_rotationEulerAngles; }
float Trackball::distanceFromLookAt() const { return m_distanceFromLookAt; }
glm::vec3 Trackball::position() const
{
    const glm::quat q(m_rotationEulerAngles);
    return m_lookAt + q * glm::vec3(0, 0, -m_distanceFromLookAt);
}
Ray Trackball::generateRay(const glm::vec2& pixel) const
{
    const glm::quat q(m_rotationEulerAngles);
    Ray r;
    r.origin = position();
    r.direction = q * glm::normalize(glm::vec3(-pixel.x * m_halfScreenSpaceWidth, pixel.y * m_halfScreenSpaceHeight, 1.0f));
    r.t = std::numeric_limits<float>::max();
    return r;
}
