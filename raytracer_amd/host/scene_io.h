// Scene ingest on the host: Wavefront OBJ / MTL meshes and PNG / JPEG / BMP textures into the
// rt::Mesh / rt::Image types that BvhInterface flattens and uploads.
//
//   loadMesh   framework/src/mesh.cpp:52-176 (loadMesh + centerAndScaleToUnitMesh),
//              with the OBJ / MTL reading that the reference delegates to
//              tinyobjloader 2.0 (framework/third_party/tinyobjloader) written out
//              for the subset the reference's scenes use
//   loadImage  framework/src/image.cpp:13-36 (8-bit RGB, value / 255)
//
// The meshes come out exactly as the reference's loader leaves them: one mesh per
// run of equal material ids inside a shape, vertices de-duplicated in first-seen
// order, missing normals replaced by the triangle's geometric normal, optional
// centring and scaling into the unit ball.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_types.h"

namespace rt {

// Throws RtIoError (missing file, malformed input, unsupported feature).
struct RtIoError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Supported: v / vn / vt / f (triangles and quads, positive or relative indices,
// i, i/j, i//k, i/j/k), g, o, usemtl, mtllib (newmtl, Kd, Ks, Ns, d, Tr, map_Kd);
// everything else is skipped.  Faces with more than four corners are rejected.
std::vector<Mesh> loadMesh(const std::string& file, bool centerAndNormalize = false);

// Non-interlaced PNG: grey / palette at 1, 2, 4, 8 bits, grey+alpha / RGB / RGBA at 8,
// anything at 16 bits (high byte kept), as stb_image's 8-bit path delivers them.
// JPEG: Huffman-coded baseline and progressive frames of 1, 3 or 4 components with
// stb_image's inverse DCT, chroma upsampling and colour conversion (jpeg.cpp).
// BMP: uncompressed, 8 (palette), 24 or 32 bits per pixel.
Image loadImage(const std::string& file);

} // namespace rt
