// Host-side types of the drop-in layer.  They carry the reference's names and
// field meanings (src/common.h, src/scene.h, framework/include/framework/{mesh,
// ray,image}.h) so code written against the reference reads the same here; the
// vector types are plain structs with glm's memory layout (three / two packed
// floats), so a glm::vec3 can be reinterpreted as a vec3 and back.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <variant>
#include <vector>

namespace rt {

struct vec2 {
    float x = 0, y = 0;
};
struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
struct uvec3 {
    uint32_t x = 0, y = 0, z = 0;
    uint32_t operator[](int i) const { return (&x)[i]; }
};
struct ivec2 {
    int x = 0, y = 0;
};

// fp32 arithmetic in glm 0.9.9.9's scalar operation order; the plane of every
// triangle is computed with these so the device sees the reference's bits
inline vec3 operator+(vec3 a, vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline vec3 operator-(vec3 a, vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline vec3 operator*(vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3 operator*(float s, vec3 a) { return { s * a.x, s * a.y, s * a.z }; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 x, vec3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
inline float length(vec3 v) { return std::sqrt(dot(v, v)); }
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline float radians(float degrees) { return degrees * static_cast<float>(0.01745329251994329576923690768489); }

// framework/include/framework/ray.h:9-13
struct Ray {
    vec3 origin { 0.0f, 0.0f, 0.0f };
    vec3 direction { 0.0f, 0.0f, -1.0f };
    float t { std::numeric_limits<float>::max() };
};

// framework/include/framework/image.h:11-18 (pixels are fp32 RGB in [0,1] from 8-bit data)
struct Image {
    int width = 0, height = 0;
    std::vector<vec3> pixels;
};

// framework/include/framework/mesh.h:14-43
struct Vertex {
    vec3 position, normal;
    vec2 texCoord;
};
struct Material {
    vec3 kd;
    vec3 ks { 0.0f, 0.0f, 0.0f };
    float shininess { 1.0f };
    float transparency { 1.0f };
    std::shared_ptr<Image> kdTexture;
};
struct Mesh {
    std::vector<Vertex> vertices;
    std::vector<uvec3> triangles;
    Material material;
};

// src/common.h:14-59
struct HitInfo {
    vec3 normal;
    vec3 barycentricCoord;
    vec2 texCoord;
    Material material;
    uvec3 triangle;
    Mesh* mesh = nullptr;
};
struct Sphere {
    vec3 center { 0.0f, 0.0f, 0.0f };
    float radius = 1.0f;
    Material material;
};
struct PointLight {
    vec3 position, color;
};
struct SegmentLight {
    vec3 endpoint0, endpoint1, color0, color1;
};
struct ParallelogramLight {
    vec3 v0, edge01, edge02, color0, color1, color2, color3;
};

// src/common.h:61-87
struct ExtraFeatures {
    bool enableEnvironmentMapping = false;
    bool enableBvhSahBinning = false;
    bool enableMotionBlur = false;
    bool enableBloomEffect = false;
    bool enableBilinearTextureFiltering = false;
    bool enableMipmapTextureFiltering = false;
    bool enableMultipleRaysPerPixel = false;
    bool enableGlossyReflection = false;
    bool enableTransparency = false;
    bool enableDepthOfField = false;
    bool enableBvhDebugDrawVisited = false;
    bool enableBvhDebugDrawUnvisited = false;
    bool suppressBvhVisitDebug = false;
};
struct Features {
    bool enableShading = false;
    bool enableRecursive = false;
    bool enableHardShadow = false;
    bool enableSoftShadow = false;
    bool enableNormalInterp = false;
    bool enableTextureMapping = false;
    bool enableAccelStructure = false;
    ExtraFeatures extra = {};
};

// src/environment_mapping.h:9-33 (lookup side only)
enum EnvironmentMappingType { CYLINDRICAL, SPHERICAL };
struct EnvironmentMap {
    std::shared_ptr<Image> texture;     // null: constant background
    vec3 backgroundColor { 0.f, 0.f, 0.f };
    float verticalFieldOfViewDegrees = 180.f;
    EnvironmentMappingType mappingType = SPHERICAL;
};

// src/scene.h:31-37
struct Scene {
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<std::variant<PointLight, SegmentLight, ParallelogramLight>> lights;
    std::vector<EnvironmentMap*> environmentMap;
};

} // namespace rt
