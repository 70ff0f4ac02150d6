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
//   * Screen's constructor (src/screen.cpp:16-30) without the GL texture;
//   * Trackball's constructor (framework/src/trackball.cpp:23-50) without the Window.
// The methods of the two classes that the frame drivers call -- Trackball::position /
// generateRay, Screen::setPixel and their neighbours -- are NOT restated here: their lines
// are extracted from the reference at build time (camera_screen_tu.cpp).
#include "draw.h"
#include "screen.h"
#include <framework/trackball.h>
#include <glm/gtc/quaternion.hpp>
#include <cmath>

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
// (clear, setPixel, resolution, pixels, indexAt: the reference's own lines, camera_screen_tu.cpp)

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
// (setCamera, position, lookAt, rotationEulerAngles, distanceFromLookAt, generateRay: the reference's
// own lines, camera_screen_tu.cpp)
