// TEST INFRASTRUCTURE (oracle/_ref build only).
// Primary-ray generation and the framebuffer layout, from the reference's own text.
// framework/src/trackball.cpp and src/screen.cpp cannot be compiled here as translation
// units (both include <GL/glu.h> through framework/opengl_includes.h, which the image lacks),
// but the methods the frame drivers call are free of OpenGL at the source level.  The
// Makefile extracts exactly those line ranges from /root/reference at build time into
// oracle/_ref/ (never committed, not sent to the GPU box):
//   trackball_extract.inc  framework/src/trackball.cpp:64-79   setCamera, position, lookAt
//                          framework/src/trackball.cpp:91-110  rotationEulerAngles,
//                                                              distanceFromLookAt, generateRay
//   screen_extract.inc     src/screen.cpp:32-43                clear, setPixel
//                          src/screen.cpp:105-123              resolution, pixels, indexAt
// and this file gives them a translation unit with the headers those lines need.  What stays
// the harness's own (hooks.cpp) are the two constructors, which take a Window / make a GL
// texture in the reference.
#include "screen.h"
#include <framework/trackball.h>
#include <glm/common.hpp>
#include <glm/gtc/quaternion.hpp>
#include <glm/vec3.hpp>
#include <glm/vec4.hpp>
#include <algorithm>
#include <limits>
#include "trackball_extract.inc"
#include "screen_extract.inc"
