// TEST INFRASTRUCTURE (oracle/_ref build only).
// getMipMap / transformToMipMap / conversionToMipMap are declared in
// src/shading.h:20 but defined in src/main.cpp:721-786, a translation unit that
// cannot be compiled here (GLFW window, ImGui, OpenGL).  The Makefile extracts
// exactly that line range from /root/reference into oracle/_ref/mipmap_extract.inc
// at build time (never committed) and this file gives it a translation unit.
#include "shading.h"
#include <framework/image.h>
#include <vector>
#include "mipmap_extract.inc"
