// TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
//
// CPU restatement of the reference ray tracer's per-pixel hot path, written
// from the reference's algorithm (each function cites the file:line it
// follows), in plain scalar C++ with no third-party code.  It exists to check
// the HIP path: tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// are its only users.  Every fp32 operation is written in the order the
// reference (and glm 0.9.9.9's scalar code paths) evaluate it and the file is
// compiled with -ffp-contract=off, so distances and triangle ids come out
// bit-identical to the reference's.
//
// Parity pin: the reference holds no tests or golden vectors for this path, so
// this restatement is pinned against the reference's own translation units
// compiled in the build container (oracle/_ref, see oracle/Makefile): per
// function known-answer sweeps, ray batches and whole frames, committed as
// fixtures under tests/golden/ with the script that produced them.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <queue>
#include <vector>
#include <omp.h>

#include "rtx.h"

namespace {

// ------------------------------------------------------------------ vec3 ----
// glm scalar semantics: dot = (x+y)+z of products, cross / normalize / length
// as in glm/detail/func_geometric.inl:48-90, inversesqrt = 1/sqrt
// (glm/detail/func_exponential.inl:134-139).
struct V3 {
    float x, y, z;
};
inline V3 mk(float x, float y, float z) { return V3 { x, y, z }; }
inline V3 mk(const float* p) { return V3 { p[0], p[1], p[2] }; }
inline V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator-(V3 a) { return { -a.x, -a.y, -a.z }; }
inline V3 operator*(V3 a, V3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline V3 operator*(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline V3 operator*(float s, V3 a) { return { s * a.x, s * a.y, s * a.z }; }
inline V3 operator/(V3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline float dot(V3 a, V3 b)
{
    const V3 t = a * b;
    return t.x + t.y + t.z;
}
inline V3 cross(V3 x, V3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
inline float length(V3 v) { return std::sqrt(dot(v, v)); }
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline float distance(V3 p0, V3 p1) { return length(p1 - p0); }
inline void put(float* d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
inline bool eq(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

constexpr float kFltMax = std::numeric_limits<float>::max();
constexpr float kInf = std::numeric_limits<float>::infinity();

struct Ray {
    V3 o { 0, 0, 0 };
    V3 d { 0, 0, -1 };
    float t = kFltMax;
};

struct Material {
    V3 kd { 0, 0, 0 };
    V3 ks { 0, 0, 0 };
    float shininess = 1.0f;
    float transparency = 1.0f;
    int texture = -1;
};

struct Texture {
    // level 0 = the image; further levels by 2x2 box filter (src/main.cpp:745-771)
    std::vector<int> w, h;
    std::vector<std::vector<V3>> pix;
};

struct Vertex {
    V3 p, n;
    float u, v;
};
struct Mesh {
    std::vector<Vertex> verts;
    std::vector<uint32_t> tris; // 3 per triangle
    int material;
};
struct Sphere {
    V3 c;
    float r;
    int material;
};

// HitInfo, src/common.h:14-21
struct Hit {
    V3 normal { 0, 0, 0 };
    V3 bary { 0, 0, 0 };
    float u = 0, v = 0;
    int material = -1;
    int mesh = -1;   // -1: none / sphere
    int triLocal = -1;
    int triGlobal = RTX_TRI_NONE;
};

struct Pair { // MeshTrianglePair, src/bounding_volume_hierarchy.h:11-15
    int mesh;
    int tri;
    V3 centroid;
};
struct Node { // src/bounding_volume_hierarchy.h:17-27
    std::vector<size_t> children;
    V3 lo, hi;
    bool leaf;
    int depth, numLevels, numLeaves;
};

struct Counters {
    uint64_t raysPrimary = 0, raysShadow = 0, raysBounce = 0;
    uint64_t boxTests = 0, triTests = 0, shaded = 0;
    void add(const Counters& o)
    {
        raysPrimary += o.raysPrimary;
        raysShadow += o.raysShadow;
        raysBounce += o.raysBounce;
        boxTests += o.boxTests;
        triTests += o.triTests;
        shaded += o.shaded;
    }
};

struct Scene {
    std::vector<Mesh> meshes;
    std::vector<Material> materials;
    std::vector<Texture> textures;
    std::vector<Sphere> spheres;
    std::vector<rtx_light> lights;
    rtx_envmap env { -1, RTX_ENV_SPHERICAL, 1.f, { 0, 0, 0 } };
    bool hasEnv = false;
    struct Rect {
        float lx, ly, ux, uy; // AxisAlignedRectangle lower / upper (src/common.h:33-36)
    };
    std::vector<Rect> envBins; // EnvironmentMap::radianceBins
    // BVH
    std::vector<Pair> pairs;
    std::vector<Node> nodes;
    long root = -1;
    int numLevels = 0, numLeaves = 0;
};

struct Feat {
    bool shading, recursive, hardShadow, softShadow, normalInterp, textureMapping, accel;
    bool envMap, sah, bilinear, mipmap, glossy;
};
Feat toFeat(const rtx_features* f)
{
    return Feat { !!f->enable_shading, !!f->enable_recursive, !!f->enable_hard_shadow, !!f->enable_soft_shadow,
        !!f->enable_normal_interp, !!f->enable_texture_mapping, !!f->enable_accel_structure,
        !!f->enable_environment_mapping, !!f->enable_bvh_sah_binning, !!f->enable_bilinear_texture_filtering,
        !!f->enable_mipmap_texture_filtering, !!f->enable_glossy_reflection };
}

// --------------------------------------------------------- keyed uniforms ----
// One uniform per std::random_device construction of the reference
// (src/light.cpp:16-23, src/shading.cpp:56-65), as a pure function of
// (seed, pixel, sample, draw index).  Value = float(bits) / 2^32 clamped below 1, which
// is what std::uniform_real_distribution<float>(0,1) makes of a 32-bit engine.
inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
inline uint32_t sampleKey(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    const uint32_t h = mix32(seed ^ (pixel * 0x9e3779b1u));
    return mix32(h ^ (sample * 0xc2b2ae35u) ^ 0x27d4eb2fu);
}
inline float keyedUniform(uint32_t key, uint32_t counter)
{
    const uint32_t bits = mix32(key ^ (counter * 0x85ebca77u) ^ 0x68bc21ebu);
    float u = float(bits) / 4294967296.0f;
    if (u >= 1.0f)
        u = 0.99999994f;
    return u;
}
// draws of one camera sample, numbered in the reference's depth-first order
struct PixelRng {
    uint32_t seed = 0, key = 0, counter = 0;
    void start(uint32_t pixel, uint32_t sample)
    {
        key = sampleKey(seed, pixel, sample);
        counter = 0;
    }
    float next() { return keyedUniform(key, counter++); }
};

// ---- what libstdc++ 11 (the library the reference is built against here) makes
// of the engine the keyed stream stands in for.  sample2D (src/sampling.cpp:6-60)
// seeds ONE engine from a random_device draw and then calls std::shuffle and
// uniform_real_distribution on it, so the engine's later outputs matter: output
// m of the engine made from draw index `index` is bits(index) for m = 0 and
// mix32(bits(index) + m + 1) after that (oracle/ref_harness/rng_shim.h).
struct DrawEngine {
    uint32_t base, m = 0;
    DrawEngine(uint32_t key, uint32_t index) : base(mix32(key ^ (index * 0x85ebca77u) ^ 0x68bc21ebu)) {}
    uint32_t operator()()
    {
        const uint32_t v = m == 0 ? base : mix32(base + m + 1u);
        m++;
        return v;
    }
};
// std::uniform_int_distribution<unsigned long>{0, range - 1} on a 32-bit engine:
// Lemire's nearly divisionless method (bits/uniform_int_dist.h, _S_nd)
inline uint32_t uniformBelow(DrawEngine& g, uint32_t range)
{
    uint64_t product = uint64_t(g()) * uint64_t(range);
    uint32_t low = uint32_t(product);
    if (low < range) {
        const uint32_t threshold = uint32_t(-range) % range;
        while (low < threshold) {
            product = uint64_t(g()) * uint64_t(range);
            low = uint32_t(product);
        }
    }
    return uint32_t(product >> 32);
}
// std::shuffle (bits/stl_algo.h): two swap positions per engine call
template <class T>
void stdShuffle(T* a, uint32_t n, DrawEngine& g)
{
    if (n == 0)
        return;
    uint32_t i = 1;
    if (n % 2 == 0) {
        std::swap(a[i], a[uniformBelow(g, 2)]);
        i++;
    }
    while (i != n) {
        const uint32_t swapRange = i + 1;
        const uint32_t x = uniformBelow(g, swapRange * (swapRange + 1));
        std::swap(a[i], a[x / (swapRange + 1)]);
        i++;
        std::swap(a[i], a[x % (swapRange + 1)]);
        i++;
    }
}
// uniform_real_distribution<float>(0, 1) on the same engine: one output / 2^32
inline float uniformReal(DrawEngine& g)
{
    float u = float(g()) / 4294967296.0f;
    if (u >= 1.0f)
        u = 0.99999994f;
    return u;
}

// std::default_random_engine = minstd_rand0 (x <- 16807 x mod 2^31-1, seed 1),
// consumed as one sequential stream by the reference's single-thread drivers
// (src/render.cpp:101-102,160-161,219-220).  Skip-ahead makes every pixel's
// position in that stream directly addressable.
constexpr uint64_t kMinstdM = 2147483647ull;
inline uint64_t minstdPow(uint64_t k)
{
    uint64_t r = 1, b = 16807;
    while (k) {
        if (k & 1)
            r = (r * b) % kMinstdM;
        b = (b * b) % kMinstdM;
        k >>= 1;
    }
    return r;
}
struct Minstd {
    uint64_t x = 1;
    // state after `draws` outputs have been consumed from a default-seeded engine
    static Minstd at(uint64_t draws)
    {
        Minstd m;
        m.x = minstdPow(draws);
        return m;
    }
    // std::uniform_real_distribution<float>(a, b)(engine) of libstdc++:
    // generate_canonical<float,24> takes one output, (x - 1) / 2^31 (the range
    // 2147483646 rounds to 2^31 in fp32), clamped below 1.
    float uniform(float a, float b)
    {
        x = (x * 16807ull) % kMinstdM;
        float c = float(x - 1) / 2147483648.0f;
        if (c >= 1.0f)
            c = 0.99999994f;
        return c * (b - a) + a;
    }
};

// ------------------------------------------------------------ primitives ----
// src/intersect.cpp:129-158
bool slab(V3 lo, V3 hi, Ray& ray)
{
    float first = 0.0f, second = kInf;
    const float o[3] = { ray.o.x, ray.o.y, ray.o.z };
    const float d[3] = { ray.d.x, ray.d.y, ray.d.z };
    const float l[3] = { lo.x, lo.y, lo.z };
    const float h[3] = { hi.x, hi.y, hi.z };
    for (int i = 0; i < 3; i++) {
        const float inv = 1.0f / d[i];
        float t0 = (l[i] - o[i]) * inv;
        float t1 = (h[i] - o[i]) * inv;
        if (t0 > t1)
            std::swap(t0, t1);
        if (t0 > second || t1 < first)
            return false;
        if (t0 > first)
            first = t0;
        if (t1 < second)
            second = t1;
    }
    if (first <= 0.0f)
        first = second;
    if (ray.t < first)
        return false;
    ray.t = first;
    return true;
}

// src/bounding_volume_hierarchy.cpp:297-307
bool inBox(V3 lo, V3 hi, V3 x)
{
    return lo.x <= x.x && x.x <= hi.x && lo.y <= x.y && x.y <= hi.y && lo.z <= x.z && x.z <= hi.z;
}

// src/intersect.cpp:16-19,32-89
bool triangle(V3 v0, V3 v1, V3 v2, Ray& ray, Hit& hit)
{
    const float prevT = ray.t;
    if (length(cross(v2 - v0, v1 - v0)) == 0.f)
        return false;
    const V3 n = normalize(cross(v1 - v0, v2 - v0));
    const float D = dot(v0, n);
    // intersectRayWithPlane
    bool planeHit = false;
    if (!(dot(n, ray.d) == 0.f)) {
        const float t = (D - dot(n, ray.o)) / dot(n, ray.d);
        if (!(t <= 0.f) && t < ray.t) {
            ray.t = t;
            planeHit = true;
        }
    }
    if (planeHit) {
        const V3 p = ray.o + ray.t * ray.d;
        const float alpha = dot(n, cross(v2 - v1, p - v1));
        const float beta = dot(n, cross(p - v0, v2 - v0));
        const float gamma = dot(n, cross(v1 - v0, p - v0));
        if (!(alpha < 0 || beta < 0 || gamma < 0)) {
            const float inv = 1 / dot(n, cross(v1 - v0, v2 - v0));
            hit.bary = mk(alpha * inv, beta * inv, gamma * inv);
            hit.normal = n;
            return true;
        }
    }
    ray.t = prevT;
    return false;
}

// src/intersect.cpp:93-124.  The reference calls the C library's double sqrt
// here (unqualified sqrt on a float), so the two roots are evaluated in double
// and rounded to float once.
bool sphereTest(const Sphere& s, Ray& ray, Hit& hit)
{
    const float a = dot(ray.d, ray.d);
    const float b = 2 * dot(ray.d, ray.o - s.c);
    const float c = dot(s.c, s.c) + dot(ray.o, ray.o) - 2 * dot(ray.o, s.c) - s.r * s.r;
    const float delta = b * b - 4 * a * c;
    if (delta < 0)
        return false;
    const float t1 = float((double(-b) + std::sqrt(double(delta))) / double(2 * a));
    const float t2 = float((double(-b) - std::sqrt(double(delta))) / double(2 * a));
    if (t2 < 0) {
        if (ray.t > t2)
            ray.t = std::min(t1, ray.t);
        else
            return false;
    } else {
        if (t1 > 0.f && ray.t > t1)
            ray.t = std::min(t2, ray.t);
        else
            return false;
    }
    const V3 p = ray.o + ray.t * ray.d;
    hit.normal = normalize(p - s.c);
    hit.material = s.material;
    hit.mesh = -1;
    return true;
}

// src/interpolate.cpp:4-12
V3 barycentric(V3 v0, V3 v1, V3 v2, V3 p)
{
    const float tArea = length(cross(v1 - v2, v2 - v0));
    const float a = length(cross(v1 - p, v2 - p)) / tArea;
    const float b = length(cross(v0 - p, v2 - p)) / tArea;
    const float g = length(cross(v1 - p, v0 - p)) / tArea;
    return mk(a, b, g);
}

// ------------------------------------------------------------------- BVH ----
// src/bounding_volume_hierarchy.cpp:25-62
void triBox(const Scene& s, const Pair& p, V3& lo, V3& hi)
{
    const Mesh& m = s.meshes[size_t(p.mesh)];
    const V3 a = m.verts[m.tris[3 * size_t(p.tri)]].p, b = m.verts[m.tris[3 * size_t(p.tri) + 1]].p, c = m.verts[m.tris[3 * size_t(p.tri) + 2]].p;
    lo = mk(std::min({ a.x, b.x, c.x }), std::min({ a.y, b.y, c.y }), std::min({ a.z, b.z, c.z }));
    hi = mk(std::max({ a.x, b.x, c.x }), std::max({ a.y, b.y, c.y }), std::max({ a.z, b.z, c.z }));
}
inline V3 vmin(V3 a, V3 b) { return mk(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return mk(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)); }
void mergeBox(V3 alo, V3 ahi, V3 blo, V3 bhi, V3& lo, V3& hi)
{
    lo = vmin(vmin(alo, blo), vmin(ahi, bhi));
    hi = vmax(vmax(alo, blo), vmax(ahi, bhi));
}
inline float comp(V3 v, int axis) { return axis == 0 ? v.x : axis == 1 ? v.y : v.z; }

// src/bounding_volume_hierarchy.cpp:65-138.  Same std:: algorithms, same
// (descending-centroid) comparator, so the permutation matches the reference
// built against the same standard library.
size_t buildNode(Scene& s, size_t* idx, size_t n, int direction, int depth, int maxDepth, bool useSah)
{
    const std::vector<Pair>& pairs = s.pairs;
    if ((maxDepth > -1 && maxDepth <= depth) || n < 2) {
        V3 lo, hi;
        triBox(s, pairs[idx[0]], lo, hi);
        for (size_t i = 0; i < n; i++) {
            V3 l2, h2;
            triBox(s, pairs[idx[i]], l2, h2);
            mergeBox(lo, hi, l2, h2, lo, hi);
        }
        s.nodes.push_back(Node { std::vector<size_t>(idx, idx + n), lo, hi, true, depth, 1, 1 });
        return s.nodes.size() - 1;
    }
    auto cmp = [&pairs, direction](const size_t a, const size_t b) {
        return ((direction == 0) && (pairs[a].centroid.x > pairs[b].centroid.x)) || ((direction == 1) && (pairs[a].centroid.y > pairs[b].centroid.y))
            || ((direction == 2) && (pairs[a].centroid.z > pairs[b].centroid.z));
    };
    size_t split;
    if (useSah) {
        std::sort(idx, idx + n, cmp);
        const float minimum = comp(pairs[idx[0]].centroid, direction), maximum = comp(pairs[idx[n - 1]].centroid, direction);
        float minCost = kInf;
        size_t minIndex = 0;
        for (size_t i = 1; i < n; i++) {
            const float cc = comp(pairs[idx[i]].centroid, direction), pc = comp(pairs[idx[i - 1]].centroid, direction);
            // size_t -> float conversions of i and n-i, products, sum, negate
            const float cost = -((pc - minimum) * i + (maximum - cc) * (n - i));
            if (minCost > cost) {
                minCost = cost;
                minIndex = i;
            }
        }
        split = minIndex;
    } else {
        std::nth_element(idx, idx + n / 2, idx + n, cmp);
        split = n / 2;
    }
    const size_t left = buildNode(s, idx, split, (direction + 1) % 3, depth + 1, maxDepth, useSah);
    const size_t right = buildNode(s, idx + split, n - split, (direction + 1) % 3, depth + 1, maxDepth, useSah);
    V3 lo, hi;
    mergeBox(s.nodes[left].lo, s.nodes[left].hi, s.nodes[right].lo, s.nodes[right].hi, lo, hi);
    s.nodes.push_back(Node { { left, right }, lo, hi, false, depth, std::max(s.nodes[left].numLevels, s.nodes[right].numLevels) + 1,
        s.nodes[left].numLeaves + s.nodes[right].numLeaves });
    return s.nodes.size() - 1;
}

// src/bounding_volume_hierarchy.cpp:141-180
void buildBvh(Scene& s, bool useSah)
{
    s.pairs.clear();
    s.nodes.clear();
    size_t n = 0;
    for (const Mesh& m : s.meshes)
        n += m.tris.size() / 3;
    if (n == 0) {
        s.root = -1;
        s.numLevels = s.numLeaves = 0;
        return;
    }
    std::vector<size_t> idx;
    idx.reserve(n);
    for (size_t mi = 0; mi < s.meshes.size(); mi++) {
        const Mesh& m = s.meshes[mi];
        for (size_t j = 0; j < m.tris.size() / 3; j++) {
            const V3 p0 = m.verts[m.tris[3 * j]].p, p1 = m.verts[m.tris[3 * j + 1]].p, p2 = m.verts[m.tris[3 * j + 2]].p;
            idx.push_back(s.pairs.size());
            s.pairs.push_back(Pair { int(mi), int(j), (1.f / 3.f) * (p0 + p1 + p2) });
        }
    }
    // glm::ceil(0.8 * glm::log2((float) n)): log2 in fp32, product and ceil in double
    const int maxDepth = int(std::ceil(0.8 * double(std::log2(float(n)))));
    s.root = long(buildNode(s, idx.data(), n, 0, 0, maxDepth, useSah));
    s.numLevels = s.nodes[size_t(s.root)].numLevels;
    s.numLeaves = s.nodes[size_t(s.root)].numLeaves;
}

// src/bounding_volume_hierarchy.cpp:277-295
void hitUpdate(const Scene& s, int meshIdx, int triLocal, int triGlobal, Hit& hit, const Ray& ray, const Feat& f)
{
    const Mesh& m = s.meshes[size_t(meshIdx)];
    const Vertex& v0 = m.verts[m.tris[3 * size_t(triLocal)]];
    const Vertex& v1 = m.verts[m.tris[3 * size_t(triLocal) + 1]];
    const Vertex& v2 = m.verts[m.tris[3 * size_t(triLocal) + 2]];
    const V3 p = ray.o + ray.t * ray.d;
    hit.material = m.material;
    hit.bary = barycentric(v0.p, v1.p, v2.p, p);
    if (f.normalInterp)
        hit.normal = hit.bary.x * v0.n + hit.bary.y * v1.n + hit.bary.z * v2.n; // src/interpolate.cpp:14-18
    else
        hit.normal = v0.n;
    // src/interpolate.cpp:20-23
    hit.u = hit.bary.x * v0.u + hit.bary.y * v1.u + hit.bary.z * v2.u;
    hit.v = hit.bary.x * v0.v + hit.bary.y * v1.v + hit.bary.z * v2.v;
    hit.mesh = meshIdx;
    hit.triLocal = triLocal;
    hit.triGlobal = triGlobal;
}

struct QEntry {
    size_t node;
    float t;
};
struct QCmp { // src/bounding_volume_hierarchy.cpp:314-319
    bool operator()(const QEntry& a, const QEntry& b) const { return a.t > b.t; }
};

// src/bounding_volume_hierarchy.cpp:320-458
bool intersect(const Scene& s, Ray& ray, Hit& hit, const Feat& f, Counters& c)
{
    bool hitAny = false;
    for (size_t i = 0; i < s.spheres.size(); i++) {
        if (sphereTest(s.spheres[i], ray, hit)) {
            hitAny = true;
            hit.triGlobal = -2 - int(i);
        }
    }
    if (!f.accel) {
        int g = 0;
        for (size_t mi = 0; mi < s.meshes.size(); mi++) {
            const Mesh& m = s.meshes[mi];
            for (size_t j = 0; j < m.tris.size() / 3; j++, g++) {
                const V3 v0 = m.verts[m.tris[3 * j]].p, v1 = m.verts[m.tris[3 * j + 1]].p, v2 = m.verts[m.tris[3 * j + 2]].p;
                const float oldT = ray.t;
                c.triTests++;
                if (triangle(v0, v1, v2, ray, hit) && double(ray.t) > 1e-6 && ray.t < oldT) {
                    hitUpdate(s, int(mi), int(j), g, hit, ray, f);
                    hitAny = true;
                } else {
                    ray.t = oldT;
                }
            }
        }
        return hitAny;
    }
    if (s.root < 0)
        return hitAny;
    std::priority_queue<QEntry, std::vector<QEntry>, QCmp> pq;
    {
        const Node& rootNode = s.nodes[size_t(s.root)];
        const float prevT = ray.t;
        c.boxTests++;
        const bool rootHit = slab(rootNode.lo, rootNode.hi, ray);
        const float key = ray.t;
        ray.t = prevT;
        if (!rootHit)
            return hitAny;
        pq.push(QEntry { size_t(s.root), key });
    }
    bool hitTri = false;
    long minTri = -1;
    while (!pq.empty()) {
        const QEntry front = pq.top();
        pq.pop();
        if (ray.t < front.t)
            continue;
        const Node& nd = s.nodes[front.node];
        if (nd.leaf) {
            for (size_t child : nd.children) {
                const Pair& pr = s.pairs[child];
                const Mesh& m = s.meshes[size_t(pr.mesh)];
                const V3 v0 = m.verts[m.tris[3 * size_t(pr.tri)]].p, v1 = m.verts[m.tris[3 * size_t(pr.tri) + 1]].p, v2 = m.verts[m.tris[3 * size_t(pr.tri) + 2]].p;
                c.triTests++;
                if (triangle(v0, v1, v2, ray, hit)) {
                    hit.material = m.material;
                    hitTri = true;
                    minTri = long(child);
                }
            }
        } else {
            const float prevT = ray.t;
            for (int k = 0; k < 2; k++) {
                const size_t ci = nd.children[size_t(k)];
                const Node& ch = s.nodes[ci];
                ray.t = prevT;
                c.boxTests++;
                if (inBox(ch.lo, ch.hi, ray.o)) {
                    pq.push(QEntry { ci, 0.0f });
                } else if (slab(ch.lo, ch.hi, ray)) {
                    pq.push(QEntry { ci, ray.t });
                }
            }
            ray.t = prevT;
        }
    }
    if (hitTri) {
        const Pair& pr = s.pairs[size_t(minTri)];
        hitUpdate(s, pr.mesh, pr.tri, int(minTri), hit, ray, f);
    }
    return hitAny || hitTri;
}

// -------------------------------------------------------------- textures ----
// src/main.cpp:745-771.  The reference reads past the end of a level whose
// width or height is odd (undefined); those reads are treated as 0 here.
void buildMipChain(Texture& t)
{
    size_t last = 0;
    while (t.pix[last].size() > 1) {
        const size_t W = size_t(t.w[last]), H = size_t(t.h[last]);
        const std::vector<V3>& src = t.pix[last];
        auto at = [&](size_t i) { return i < src.size() ? src[i] : mk(0, 0, 0); };
        std::vector<V3> v;
        for (size_t i = 0; i < H; i += 2)
            for (size_t j = 0; j < W; j += 2) {
                V3 avg = at(i * W + j) + at(i * W + j + 1) + at((i + 1) * W + j) + at((i + 1) * W + j + 1);
                avg = avg * (1.0f / 4.0f);
                v.push_back(avg);
            }
        t.h.push_back(t.h[last] / 2);
        t.w.push_back(t.w[last] / 2);
        t.pix.push_back(std::move(v));
        last++;
        if (t.pix.size() >= RTX_MAX_MIP_LEVELS)
            break;
    }
}

// src/texture.cpp:6-16
V3 texelNearest(const Texture& t, float u, float v)
{
    const int W = t.w[0], H = t.h[0];
    int i = int(float(W) * u);
    int j = int(float(H) * (1 - v));
    i = std::max(0, std::min(W - 1, i));
    j = std::max(0, std::min(H - 1, j));
    return t.pix[0][size_t(j * W + i)];
}
inline float fmodGlm(float a, float b) { return a - b * std::floor(a / b); } // glm::mod
// src/texture.cpp:18-33 and src/shading.cpp:174-190 (same arithmetic per level)
V3 texelBilinear(const Texture& t, int level, float tu, float tv)
{
    const int W = t.w[size_t(level)], H = t.h[size_t(level)];
    const std::vector<V3>& px = t.pix[size_t(level)];
    const float tx = float(W - 1) * tu, ty = float(H - 1) * (1 - tv);
    float lx = std::floor(float(W - 1) * tu), ly = std::floor(float(H - 1) * (1 - tv));
    float ux = lx + 1, uy = ly + 1;
    const float u = tx - lx, v = ty - ly;
    lx = fmodGlm(lx, float(W));
    ly = fmodGlm(ly, float(H));
    ux = fmodGlm(ux, float(W));
    uy = fmodGlm(uy, float(H));
    auto at = [&](float idx) {
        const size_t i = size_t(idx);
        return i < px.size() ? px[i] : mk(0, 0, 0);
    };
    const V3 lowerLeft = at(ly * float(W) + lx);
    const V3 upperRight = at(uy * float(W) + ux);
    const V3 lowerRight = at(uy * float(W) + lx);
    const V3 upperLeft = at(ly * float(W) + ux);
    return lowerLeft * (1.0f - u) * (1.0f - v) + upperRight * u * v + lowerRight * (1.0f - u) * v + upperLeft * u * (1.0f - v);
}
// src/shading.cpp:107-114
V3 planeCoord(const Ray& ray, V3 point)
{
    const V3 w = normalize(ray.d);
    const V3 t = normalize(w - mk(0.1f, 0.0f, 0.0f));
    const V3 x = normalize(cross(t, w));
    const V3 y = normalize(cross(w, x));
    const V3 rToP = point - ray.o;
    return ray.o + dot(rToP, x) * x + dot(rToP, y) * y;
}
// src/shading.cpp:117-172
V3 texelTrilinear(const Scene& s, const Texture& t, const Ray& ray, const Hit& hit)
{
    if (hit.mesh < 0)
        return texelBilinear(t, 0, hit.u, hit.v);
    const Mesh& m = s.meshes[size_t(hit.mesh)];
    const V3 v0 = m.verts[m.tris[3 * size_t(hit.triLocal)]].p, v1 = m.verts[m.tris[3 * size_t(hit.triLocal) + 1]].p, v2 = m.verts[m.tris[3 * size_t(hit.triLocal) + 2]].p;
    const V3 p0 = planeCoord(ray, v0), p1 = planeCoord(ray, v1), p2 = planeCoord(ray, v2);
    const float areaV = length(cross(v0 - v1, v0 - v2));
    const float areaP = length(cross(p0 - p1, p0 - p2));
    const size_t levels = t.pix.size();
    float k = float(levels);
    if (double(areaP) >= 1e-6)
        k = (areaV / areaP - 1.0f) / 2.8f;
    const float k0 = std::floor(k);
    const float k1 = k0 + 1;
    const float a = k1 - k;
    if (k0 < 0)
        return texelBilinear(t, 0, hit.u, hit.v);
    if (k1 >= float(levels))
        return texelBilinear(t, int(levels) - 1, hit.u, hit.v);
    const V3 c0 = texelBilinear(t, int(k0), hit.u, hit.v);
    const V3 c1 = texelBilinear(t, int(k1), hit.u, hit.v);
    return a * c0 + (1 - a) * c1;
}

// src/environment_mapping.cpp:167-207 (the uv arithmetic is in double there
// because PI is a double literal)
constexpr double kPi = 3.14159265358979323846;
V3 envColor(const Scene& s, const Ray& ray, const Feat& f)
{
    const V3 bg = mk(s.env.background);
    if (s.env.texture < 0)
        return bg;
    const Texture& t = s.textures[size_t(s.env.texture)];
    const float x = float(double(std::atan2(ray.d.z, ray.d.x)) / (2 * kPi) + 0.5);
    const float cosUp = dot(normalize(ray.d), mk(0.f, 1.f, 0.f));
    float y;
    if (s.env.mapping == RTX_ENV_SPHERICAL)
        y = float(-(kPi / double(s.env.vfov_radians)) * (double(std::acos(cosUp)) / kPi - 0.5) + 0.5);
    else
        y = float(0.5 / double(std::tan(0.5f * s.env.vfov_radians)) / double(std::tan(std::acos(cosUp))) + 0.5);
    if (std::fabs(double(y) - 0.5) > 0.5)
        return bg;
    if (f.bilinear)
        return texelBilinear(t, 0, x, y);
    return texelNearest(t, x, y);
}

// --------------------------------------------------------------- shading ----
// src/shading.cpp:90-96
Ray reflectionRay(const Ray& ray, V3 normal)
{
    const V3 point = ray.o + ray.d * ray.t;
    const V3 r = normalize(ray.d) - 2 * dot(normalize(normal), normalize(ray.d)) * normalize(normal);
    Ray out;
    out.o = point + float(1e-5) * r;
    out.d = r;
    out.t = kFltMax;
    return out;
}

// src/shading.cpp:13-54
V3 computeShading(const Scene& s, V3 lightPos, V3 lightColor, const Feat& f, const Ray& ray, const Hit& hit)
{
    const Material& mat = s.materials[size_t(hit.material)];
    const V3 position = ray.o + ray.t * ray.d;
    const V3 lightDir = normalize(lightPos - position);
    const float ndl = dot(normalize(hit.normal), lightDir);
    if (ndl < 0.0f)
        return mk(0, 0, 0);
    Ray lr;
    lr.o = lightPos;
    lr.d = lightDir;
    const Ray refl = reflectionRay(lr, hit.normal);
    const float d = std::fabs(dot(normalize(refl.d), normalize(ray.d)));
    const V3 specular = lightColor * mat.ks * std::pow(d, mat.shininess);
    if (mat.texture >= 0 && f.textureMapping) {
        const Texture& t = s.textures[size_t(mat.texture)];
        V3 texel = texelNearest(t, hit.u, hit.v);
        if (f.bilinear) {
            if (f.mipmap)
                texel = texelTrilinear(s, t, ray, hit);
            else
                texel = texelBilinear(t, 0, hit.u, hit.v);
        }
        return lightColor * texel * ndl + specular;
    }
    return lightColor * mat.kd * ndl + specular;
}

// ---------------------------------------- environment light sampling ----
// getRadiance, src/environment_mapping.cpp:26-29 (double arithmetic, float result)
inline float radianceOf(V3 c) { return float(0.25 * (double(c.x) + 2. * double(c.y) + double(c.z))); }

// EnvironmentMap::buildRadianceHierarchy, src/environment_mapping.cpp:55-105
void buildRadianceBins(Scene& s, const Texture& t, Scene::Rect aar, float totalRadiance)
{
    const int W = t.w[0], H = t.h[0];
    const float sizeX = aar.ux - aar.lx, sizeY = aar.uy - aar.ly;
    const float minRadiance = float((W * H) / 1000);
    if (totalRadiance < minRadiance || double(sizeX + sizeY) < 0.1) {
        s.envBins.push_back(aar);
        return;
    }
    const float imageW = float(W), m1x = float(W - 1), m1y = float(H - 1);
    const bool splitVertically = sizeY / sizeX > 1.f;
    const float splitX = splitVertically ? aar.ux : float(0.5 * double(aar.lx + aar.ux));
    const float splitY = splitVertically ? float(0.5 * double(aar.ly + aar.uy)) : aar.uy;
    const float lowerScaledX = aar.lx * m1x, lowerScaledY = aar.ly * m1y;
    const float splitScaledX = splitX * m1x, splitScaledY = splitY * m1y;
    float lowerHalf = 0.f;
    for (int i = int(lowerScaledX); float(i) < splitScaledX; i++)
        for (int j = int(lowerScaledY); float(j) < splitScaledY; j++)
            lowerHalf += radianceOf(t.pix[0][size_t(float(i) + float(j) * imageW)]);
    const float upperHalf = totalRadiance - lowerHalf;
    const Scene::Rect lower { aar.lx, aar.ly, splitX, splitY };
    const Scene::Rect upper { splitVertically ? aar.lx : splitX, splitVertically ? splitY : aar.ly, aar.ux, aar.uy };
    if (double(lowerHalf) > 0.5 * double(totalRadiance))
        buildRadianceBins(s, t, lower, lowerHalf);
    else
        s.envBins.push_back(lower);
    if (double(upperHalf) > 0.5 * double(totalRadiance))
        buildRadianceBins(s, t, upper, upperHalf);
    else
        s.envBins.push_back(upper);
}
// EnvironmentMap's constructor, src/environment_mapping.cpp:42-53
void buildRadianceBins(Scene& s)
{
    s.envBins.clear();
    if (s.env.texture < 0)
        return;
    const Texture& t = s.textures[size_t(s.env.texture)];
    float total = 0.f;
    for (const V3& px : t.pix[0])
        total += radianceOf(px);
    buildRadianceBins(s, t, Scene::Rect { 0.f, 0.f, 1.f, 1.f }, total);
}

// sample2D, src/sampling.cpp:6-60: multi-jittered samples of a rectangle; the engine
// is seeded from draw `index` of the keyed stream.  out: h*k (x, y) pairs.
void sample2D(const Scene::Rect& rect, int h, int k, uint32_t key, uint32_t index, std::vector<float>& out)
{
    DrawEngine urng(key, index);
    std::vector<int> shufflingIndices(static_cast<size_t>(h));
    for (int i = 0; i < h; i++)
        shufflingIndices[size_t(i)] = i;
    struct Cell {
        float x, y;
    };
    std::vector<std::vector<Cell>> rows(static_cast<size_t>(k), std::vector<Cell>(static_cast<size_t>(h)));
    for (int j = 0; j < k; j++)
        for (int i = 0; i < h; i++)
            rows[size_t(j)][size_t(i)] = Cell { float(j), float(i) };
    for (int j = 0; j < k; j++)
        stdShuffle(rows[size_t(j)].data(), uint32_t(h), urng);
    for (int i = 0; i < h; i++) {
        stdShuffle(shufflingIndices.data(), uint32_t(h), urng);
        int j = 0;
        const Cell temp = rows[size_t(j)][size_t(i)];
        for (int x = 0; x < k; x++) {
            rows[size_t(j)][size_t(i)] = rows[size_t(shufflingIndices[size_t(x)])][size_t(i)];
            j = shufflingIndices[size_t(x)];
        }
        rows[size_t(j)][size_t(i)] = temp;
    }
    const float scaleX = float(1. / (k + 1)), scaleY = float(1. / (h + 1));
    const float sizeX = rect.ux - rect.lx, sizeY = rect.uy - rect.ly;
    out.clear();
    for (int j = 0; j < k; j++)
        for (int i = 0; i < h; i++) {
            const Cell c = rows[size_t(j)][size_t(i)];
            const float jx = uniformReal(urng);
            const float jy = uniformReal(urng);
            const float subX = c.x + jx, subY = c.y + jy;
            out.push_back(rect.lx + sizeX * (float(i / (h + 1.)) + scaleX * subX));
            out.push_back(rect.ly + sizeY * (float(j / (k + 1.)) + scaleY * subY));
        }
}

// EnvironmentMap::getRayForCoordinate, src/environment_mapping.cpp:130-165
V3 envDirection(const Scene& s, float cx, float cy)
{
    const float azimuth = float(2.0 * kPi * (double(cx) - 0.5));
    float pitch;
    if (s.env.mapping == RTX_ENV_SPHERICAL)
        pitch = float((0.5 - double(cy)) * double(s.env.vfov_radians));
    else
        pitch = float(std::atan(double(std::tan(0.5f * s.env.vfov_radians)) * (0.5 - double(cy)) / 0.5));
    const float px = std::cos(pitch), py = std::sin(pitch);
    return mk(px * std::cos(azimuth), py, px * std::sin(azimuth));
}

struct Ctx {
    const Scene& s;
    Feat f;
    int shadowSamples, glossyRays;
    PixelRng rng;
    Counters c;
};

enum RayKind { kPrimary, kShadow, kBounce };
bool trace(Ctx& cx, Ray& ray, Hit& hit, RayKind kind)
{
    if (kind == kPrimary)
        cx.c.raysPrimary++;
    else if (kind == kShadow)
        cx.c.raysShadow++;
    else
        cx.c.raysBounce++;
    return intersect(cx.s, ray, hit, cx.f, cx.c);
}

// src/light.cpp:70-102
float visibility(Ctx& cx, V3 samplePos, const Ray& ray, const Hit& hit)
{
    const V3 hp = ray.o + ray.t * ray.d;
    const float eps = 1e-6;
    V3 normal = hit.normal;
    if (dot(normalize(hit.normal), normalize(ray.o - hp)) < -eps)
        normal = -hit.normal;
    Ray sray;
    sray.o = samplePos;
    sray.d = hp - samplePos;
    sray.t = kFltMax;
    Hit h2;
    trace(cx, sray, h2, kShadow);
    const V3 second = sray.o + sray.t * sray.d;
    if (dot(normalize(samplePos - hp), normalize(normal)) < -eps)
        return 0.0f;
    if (double(distance(hp, second)) > 1e-3)
        return 0.0f;
    return 1.0f;
}

// EnvironmentMap::getSamplingRay, src/environment_mapping.cpp:107-128: rays towards
// samples of randomly chosen radiance bins, kept when they leave the surface
void envSamplingRays(Ctx& cx, V3 position, V3 normal, int n, std::vector<Ray>& out)
{
    const Scene& s = cx.s;
    out.clear();
    if (s.env.texture < 0) {
        Ray r;
        r.o = position;
        r.d = normal;
        r.t = kFltMax;
        out.push_back(r);
        return;
    }
    const int nPerTry = int(std::max(std::sqrt(n / 10.), 1.));
    std::vector<float> samples;
    for (int i = 0; int(out.size()) < n && i < 100; i++) {
        const int randomBin = int(std::floor(cx.rng.next() * float(s.envBins.size())));
        const Scene::Rect& rect = s.envBins[size_t(randomBin)];
        sample2D(rect, nPerTry, nPerTry, cx.rng.key, cx.rng.counter++, samples);
        for (size_t k = 0; k + 1 < samples.size(); k += 2) {
            Ray r;
            r.d = envDirection(s, samples[k], samples[k + 1]);
            const float eps = 0.0001f / dot(normalize(r.d), normal);
            r.o = position + eps * r.d;
            r.t = kFltMax;
            if (dot(r.d, normal) > 0.f)
                out.push_back(r);
        }
    }
}

// src/light.cpp:47-66
V3 sampleEnvironment(Ctx& cx, const Ray& ray, const Hit& hit)
{
    V3 avg = mk(0, 0, 0);
    const V3 origin = ray.o + ray.t * ray.d;
    Hit mine = hit;
    mine.normal = dot(hit.normal, ray.d) < 0 ? hit.normal : -hit.normal;
    std::vector<Ray> srays;
    envSamplingRays(cx, origin, mine.normal, cx.shadowSamples, srays);
    for (Ray& sray : srays) {
        Hit dummy;
        if (!cx.f.hardShadow || !trace(cx, sray, dummy, kShadow)) {
            const V3 pos = sray.o + 100000.f * sray.d;
            const V3 col = envColor(cx.s, sray, cx.f);
            avg = avg + computeShading(cx.s, pos, col, cx.f, ray, mine);
        }
    }
    return avg * float(1.0 / double(float(cx.shadowSamples)));
}

// src/light.cpp:139-189 with the samplers of src/light.cpp:26-45
V3 lightContribution(Ctx& cx, const Ray& ray, const Hit& hit)
{
    const Scene& s = cx.s;
    if (!cx.f.shading)
        return s.materials[size_t(hit.material)].kd;
    cx.c.shaded++;
    V3 med = mk(0, 0, 0);
    for (const rtx_light& l : s.lights) {
        if (l.type == RTX_LIGHT_POINT) {
            const V3 pos = mk(l.p[0]), col = mk(l.c[0]);
            if (cx.f.hardShadow) {
                // computeShading is evaluated before the visibility test
                const V3 sh = computeShading(s, pos, col, cx.f, ray, hit);
                med = med + sh * visibility(cx, pos, ray, hit);
            } else {
                med = med + computeShading(s, pos, col, cx.f, ray, hit);
            }
        } else if (l.type == RTX_LIGHT_SEGMENT) {
            if (cx.f.softShadow) {
                V3 avg = mk(0, 0, 0);
                for (int i = 0; i < cx.shadowSamples; i++) {
                    const V3 e0 = mk(l.p[0]), e1 = mk(l.p[1]);
                    const float alpha = cx.rng.next();
                    const V3 pos = alpha * (e1 - e0) + e0;
                    const V3 col = (1 - alpha) * mk(l.c[0]) + alpha * mk(l.c[1]);
                    const V3 sh = computeShading(s, pos, col, cx.f, ray, hit);
                    avg = avg + sh * visibility(cx, pos, ray, hit);
                }
                med = med + avg * float(1.0 / double(float(cx.shadowSamples)));
            }
        } else {
            if (cx.f.softShadow) {
                V3 avg = mk(0, 0, 0);
                for (int i = 0; i < cx.shadowSamples; i++) {
                    const float a1 = cx.rng.next();
                    const V3 c1 = (1 - a1) * mk(l.c[0]) + a1 * mk(l.c[1]);
                    const V3 c2 = (1 - a1) * mk(l.c[2]) + a1 * mk(l.c[3]);
                    const float a2 = cx.rng.next();
                    const V3 col = (1 - a2) * c1 + a2 * c2;
                    const V3 pos = mk(l.p[0]) + mk(l.p[1]) * a1 + mk(l.p[2]) * a2;
                    const V3 sh = computeShading(s, pos, col, cx.f, ray, hit);
                    avg = avg + sh * visibility(cx, pos, ray, hit);
                }
                med = med + avg * float(1.0 / double(float(cx.shadowSamples)));
            }
        }
    }
    // (a scene without an environment map behaves like the reference's default
    // EnvironmentMap(glm::vec3 {0}): one ray along the normal, black)
    if (cx.f.softShadow && cx.f.envMap)
        med = med + sampleEnvironment(cx, ray, hit);
    return med;
}

// src/shading.cpp:67-105: the fan is generated in full (two draws per ray,
// ua then va) before any of its rays is traced
void glossyFan(Ctx& cx, const Ray& reflection, float blur, std::vector<Ray>& out)
{
    for (int i = 1; i <= cx.glossyRays; i++) {
        const V3 w = normalize(reflection.d);
        const V3 t = normalize(w - mk(0.1f, 0.0f, 0.0f));
        const V3 u = normalize(cross(t, w));
        const V3 v = normalize(cross(w, u));
        const float ua = -blur / 2.0f + blur * cx.rng.next();
        const float va = -blur / 2.0f + blur * cx.rng.next();
        Ray rr = reflection;
        rr.d = rr.d + (ua * u + va * v);
        out.push_back(rr);
    }
}

// src/render.cpp:16-66
V3 finalColor(Ctx& cx, Ray ray, int depth, RayKind kind)
{
    Hit hit;
    if (trace(cx, ray, hit, kind)) {
        const V3 Lo = lightContribution(cx, ray, hit);
        if (cx.f.recursive) {
            const Material& mat = cx.s.materials[size_t(hit.material)];
            const Ray reflection = reflectionRay(ray, hit.normal);
            if (depth > 0 && !eq(mat.ks, mk(0, 0, 0))) {
                std::vector<Ray> rays;
                if (!cx.f.glossy)
                    rays.push_back(reflection);
                else
                    glossyFan(cx, reflection, 1.0f / mat.shininess, rays);
                V3 color = mk(0, 0, 0);
                int count = 0;
                for (const Ray& r : rays) {
                    const float cosinus = dot(normalize(r.d), normalize(hit.normal));
                    const float sinus = std::sqrt(1.0f - cosinus * cosinus);
                    if (cosinus > 0.0f && sinus > 0.0f) {
                        color = color + finalColor(cx, r, depth - 1, kBounce);
                        count++;
                    }
                }
                color = color * (1.0f / float(count));
                return Lo + color * mat.ks;
            }
            return Lo;
        }
        return Lo;
    }
    if (cx.f.envMap && cx.s.hasEnv)
        return envColor(cx.s, ray, cx.f);
    return mk(0, 0, 0);
}

// src/transparency.cpp:6-20.  The probe query and the query inside
// getFinalColor are the same ray; both are traced (and counted once, as the
// metric defines it: the duplicate is not a new ray).
V3 transparencyColor(Ctx& cx, Ray ray, int depth)
{
    Hit hit;
    {
        Counters scratch; // duplicate trace of the same ray is not counted
        Ray probe = ray;
        const bool any = intersect(cx.s, probe, hit, cx.f, scratch);
        (void)any;
        ray.t = probe.t;
    }
    const float transparency = hit.material >= 0 ? cx.s.materials[size_t(hit.material)].transparency : 1.0f;
    if (ray.t == kFltMax || transparency == 1.0f) {
        ray.t = kFltMax;
        return finalColor(cx, ray, depth, kPrimary);
    }
    const float opp = 1.0f - transparency;
    Ray next;
    const float advance = float(double(ray.t) + 1e-6); // glm::vec3(ray.t + 1e-6): double sum rounded to float
    next.o = ray.o + mk(advance, advance, advance) * ray.d;
    next.d = ray.d;
    next.t = kFltMax;
    ray.t = kFltMax;
    const V3 front = finalColor(cx, ray, depth, kPrimary);
    const V3 behind = transparencyColor(cx, next, depth);
    return transparency * front + opp * behind;
}

// ---------------------------------------------------------------- camera ----
// framework/src/trackball.cpp:71-74,101-110; glm::quat(euler) and quat * vec3
// as in glm/detail/type_quat.inl:208-217,347-354
struct Quat {
    float w, x, y, z;
};
Quat quatFromEuler(V3 e)
{
    const V3 h = e * 0.5f;
    const V3 c = mk(std::cos(h.x), std::cos(h.y), std::cos(h.z));
    const V3 s = mk(std::sin(h.x), std::sin(h.y), std::sin(h.z));
    Quat q;
    q.w = c.x * c.y * c.z + s.x * s.y * s.z;
    q.x = s.x * c.y * c.z - c.x * s.y * s.z;
    q.y = c.x * s.y * c.z + s.x * c.y * s.z;
    q.z = c.x * c.y * s.z - s.x * s.y * c.z;
    return q;
}
V3 rotate(Quat q, V3 v)
{
    const V3 qv = mk(q.x, q.y, q.z);
    const V3 uv = cross(qv, v);
    const V3 uuv = cross(qv, uv);
    return v + ((uv * q.w) + uuv) * 2.0f;
}
struct Camera {
    V3 lookAt, euler;
    float dist, halfW, halfH;
    Ray generate(float px, float py) const
    {
        const Quat q = quatFromEuler(euler);
        const V3 dirCam = normalize(mk(-px * halfW, py * halfH, 1.0f));
        Ray r;
        r.o = lookAt + rotate(q, mk(0, 0, -dist));
        r.d = rotate(q, dirCam);
        r.t = kFltMax;
        return r;
    }
};
Camera toCamera(const rtx_camera* c, int W, int H)
{
    Camera cam;
    cam.lookAt = mk(c->look_at);
    cam.euler = mk(c->euler);
    cam.dist = c->distance;
    cam.halfH = std::tan(c->fovy / 2.0f);
    cam.halfW = (float(W) / float(H)) * cam.halfH; // Window::getAspectRatio, framework/src/window.cpp:379-384
    return cam;
}

// --------------------------------------------------------------- drivers ----
int defaultDepth(int driver) { return (driver == RTX_DRIVER_MRAYS || driver == RTX_DRIVER_DOF) ? 0 : 1; }
// the sample of the drivers' motion-blur pose (src/render.cpp:224-235): the current camera moved on by w pose changes
Camera poseAt(const Camera& cam, const Camera& prev, float w)
{
    Camera c2 = cam;
    c2.lookAt = cam.lookAt + w * (cam.lookAt - prev.lookAt);
    c2.euler = cam.euler + w * (cam.euler - prev.euler);
    c2.dist = cam.dist + w * (cam.dist - prev.dist);
    return c2;
}

// src/render.cpp:69-253; one call renders one pixel
V3 renderPixel(Ctx& cx, const Camera& cam, const Camera& prev, const rtx_render_params& rp, int x, int y)
{
    const int W = rp.width, H = rp.height;
    const int driver = rp.driver;
    const int depth = rp.ray_depth >= 0 ? rp.ray_depth : defaultDepth(driver);
    const int n = rp.samples > 0 ? rp.samples : (driver == RTX_DRIVER_MRAYS ? 64 : driver == RTX_DRIVER_DOF ? 100 : 7);
    const uint64_t pixelIndex = uint64_t(y) * uint64_t(W) + uint64_t(x);
    cx.rng.start(uint32_t(pixelIndex), 0);
    const float ndcX = float(x) / float(W) * 2.0f - 1.0f, ndcY = float(y) / float(H) * 2.0f - 1.0f;
    switch (driver) {
    case RTX_DRIVER_PLAIN:
        return finalColor(cx, cam.generate(ndcX, ndcY), depth, kPrimary);
    case RTX_DRIVER_TRANSPARENCY:
        return transparencyColor(cx, cam.generate(ndcX, ndcY), depth);
    case RTX_DRIVER_MRAYS: {
        Minstd rng = Minstd::at(pixelIndex * 2ull * uint64_t(n));
        V3 col = mk(0, 0, 0);
        for (int i = 0; i < n; i++) {
            cx.rng.start(uint32_t(pixelIndex), uint32_t(i));
            const float jx = rng.uniform(0.f, 1.f);
            const float jy = rng.uniform(0.f, 1.f);
            const float qx = float(float(x) + jx) / float(W) * 2.0f - 1.0f;
            const float qy = float(float(y) + jy) / float(H) * 2.0f - 1.0f;
            col = col + finalColor(cx, cam.generate(qx, qy), depth, kPrimary);
        }
        const float inv = 1.0f / (float(n) * 1.0f);
        return col * mk(inv, inv, inv);
    }
    case RTX_DRIVER_DOF: {
        const float aperture = rp.aperture > 0.f ? rp.aperture : 0.1f;
        const float focal = rp.focal_length > 0.f ? rp.focal_length : 2.0f;
        Minstd rng = Minstd::at(pixelIndex * 3ull * uint64_t(n));
        const Ray cr = cam.generate(ndcX, ndcY);
        const V3 focalPoint = cr.o + focal * cr.d; // src/dof.cpp:9-11
        // all origins are drawn first (src/dof.cpp:17-28), then traced in order
        std::vector<V3> origins;
        for (int i = 0; i < n; i++) {
            const float sx = rng.uniform(-aperture / 2, aperture / 2);
            const float sy = rng.uniform(-aperture / 2, aperture / 2);
            const float sz = rng.uniform(-aperture / 2, aperture / 2);
            origins.push_back(cr.o + mk(sx, sy, sz));
        }
        V3 col = mk(0, 0, 0);
        uint32_t si = 0;
        for (const V3& o : origins) {
            cx.rng.start(uint32_t(pixelIndex), si++);
            Ray r;
            r.o = o;
            r.d = normalize(focalPoint - o);
            r.t = kFltMax;
            col = col + finalColor(cx, r, depth, kPrimary);
        }
        const float inv = 1.0f / float(origins.size());
        return col * mk(inv, inv, inv);
    }
    case RTX_DRIVER_MOTION_BLUR: {
        Minstd rng = Minstd::at(pixelIndex * uint64_t(n));
        const V3 scaleLookAt = cam.lookAt - prev.lookAt;
        const V3 scaleAngles = cam.euler - prev.euler;
        const float scaleDistance = cam.dist - prev.dist;
        V3 col = mk(0, 0, 0);
        for (int i = 0; i < n; i++) {
            cx.rng.start(uint32_t(pixelIndex), uint32_t(i));
            const float jitter = rng.uniform(0.f, 1.f);
            const float lerp = jitter + float(i);
            const float w = lerp / float(n);
            // the reference overwrites its stored start pose with the current
            // camera before this loop (src/render.cpp:211-213), so the samples
            // extrapolate forward from the current pose by the last frame's delta
            Camera c2 = cam;
            c2.lookAt = cam.lookAt + w * scaleLookAt;
            c2.euler = cam.euler + w * scaleAngles;
            c2.dist = cam.dist + w * scaleDistance;
            col = col + finalColor(cx, c2.generate(ndcX, ndcY), depth, kPrimary);
        }
        return col / float(n);
    }
    case RTX_DRIVER_COMPOSED: {
        // The four sampling drivers in one frame (the reference's router, src/main.cpp:50-63, runs one of
        // them; the combination is this build's own, stated in include/rtx.h): time step x lens sample x
        // pixel jitter, the pieces and their draws in that order per sample, a dimension of size 1 off.
        const int J = std::max(rp.samples, 1), L = std::max(rp.lens_samples, 1), T = std::max(rp.time_steps, 1), S = T * L * J;
        const uint64_t per = uint64_t(T > 1 ? 1 : 0) + uint64_t(J > 1 ? 2 : 0) + uint64_t(L > 1 ? 3 : 0);
        const int d = rp.ray_depth >= 0 ? rp.ray_depth : ((J > 1 || L > 1) ? 0 : 1);
        const float aperture = rp.aperture > 0.f ? rp.aperture : 0.1f;
        const float focal = rp.focal_length > 0.f ? rp.focal_length : 2.0f;
        Minstd rng = Minstd::at(pixelIndex * per * uint64_t(S));
        V3 col = mk(0, 0, 0);
        for (int s = 0; s < S; s++) {
            const int t = s / (L * J);
            cx.rng.start(uint32_t(pixelIndex), uint32_t(s));
            Camera c2 = cam;
            if (T > 1) { // src/render.cpp:224-235
                const float jitter = rng.uniform(0.f, 1.f);
                const float lerp = jitter + float(t);
                c2 = poseAt(cam, prev, lerp / float(T));
            }
            float qx = ndcX, qy = ndcY;
            if (J > 1) { // src/render.cpp:170-174
                const float jx = rng.uniform(0.f, 1.f);
                const float jy = rng.uniform(0.f, 1.f);
                qx = float(float(x) + jx) / float(W) * 2.0f - 1.0f;
                qy = float(float(y) + jy) / float(H) * 2.0f - 1.0f;
            }
            Ray r = c2.generate(qx, qy);
            if (L > 1) { // src/dof.cpp:9-39
                const V3 focalPoint = r.o + focal * r.d;
                const float sx = rng.uniform(-aperture / 2, aperture / 2);
                const float sy = rng.uniform(-aperture / 2, aperture / 2);
                const float sz = rng.uniform(-aperture / 2, aperture / 2);
                r.o = r.o + mk(sx, sy, sz);
                r.d = normalize(focalPoint - r.o);
            }
            const V3 value = rp.transparency ? transparencyColor(cx, r, d) : finalColor(cx, r, d, kPrimary);
            if (S == 1)
                return value;
            col = col + value;
        }
        if (J == 1 && L == 1)
            return col / float(S); // src/render.cpp:249
        const float inv = 1.0f / float(S); // src/render.cpp:178, :122
        return col * mk(inv, inv, inv);
    }
    default:
        return mk(0, 0, 0);
    }
}

Material toMaterial(const float* m, int tex)
{
    Material o;
    o.kd = mk(m);
    o.ks = mk(m + 3);
    o.shininess = m[6];
    o.transparency = m[7];
    o.texture = tex;
    return o;
}
int addTexture(Scene& s, const float* rgb, int w, int h)
{
    Texture t;
    t.w.push_back(w);
    t.h.push_back(h);
    t.pix.emplace_back(size_t(w) * size_t(h));
    std::memcpy(t.pix[0].data(), rgb, size_t(w) * size_t(h) * 3 * sizeof(float));
    buildMipChain(t);
    s.textures.push_back(std::move(t));
    return int(s.textures.size()) - 1;
}

} // namespace

extern "C" {

void* orc_scene_create() { return new Scene; }
void orc_scene_free(void* p) { delete static_cast<Scene*>(p); }

void orc_scene_add_mesh(void* p, const float* verts, int nv, const uint32_t* tris, int nt, const float* mat, const float* tex, int tw, int th)
{
    Scene& s = *static_cast<Scene*>(p);
    Mesh m;
    m.verts.resize(size_t(nv));
    for (int i = 0; i < nv; i++) {
        const float* v = verts + size_t(i) * 8;
        m.verts[size_t(i)] = Vertex { mk(v), mk(v + 3), v[6], v[7] };
    }
    m.tris.assign(tris, tris + size_t(nt) * 3);
    const int texIdx = tex ? addTexture(s, tex, tw, th) : -1;
    s.materials.push_back(toMaterial(mat, texIdx));
    m.material = int(s.materials.size()) - 1;
    s.meshes.push_back(std::move(m));
}
void orc_scene_add_light(void* p, const rtx_light* l) { static_cast<Scene*>(p)->lights.push_back(*l); }
void orc_scene_clear_lights(void* p) { static_cast<Scene*>(p)->lights.clear(); }
void orc_scene_add_sphere(void* p, const float* center, float radius, const float* mat)
{
    Scene& s = *static_cast<Scene*>(p);
    s.materials.push_back(toMaterial(mat, -1));
    s.spheres.push_back(Sphere { mk(center), radius, int(s.materials.size()) - 1 });
}
void orc_scene_set_envmap(void* p, const float* rgb, int w, int h, int mapping, float vfovDegrees, const float* bg)
{
    Scene& s = *static_cast<Scene*>(p);
    s.env.texture = rgb ? addTexture(s, rgb, w, h) : -1;
    s.env.mapping = mapping;
    s.env.vfov_radians = vfovDegrees * float(0.01745329251994329576923690768489); // glm::radians
    std::memcpy(s.env.background, bg, 3 * sizeof(float));
    buildRadianceBins(s);
    s.hasEnv = true;
}

void orc_build(void* p, const rtx_features* f) { buildBvh(*static_cast<Scene*>(p), !!f->enable_bvh_sah_binning); }
int orc_levels(void* p) { return static_cast<Scene*>(p)->numLevels; }
int orc_leaves(void* p) { return static_cast<Scene*>(p)->numLeaves; }
int64_t orc_num_nodes(void* p) { return int64_t(static_cast<Scene*>(p)->nodes.size()); }
int64_t orc_num_children(void* p)
{
    int64_t n = 0;
    for (const Node& nd : static_cast<Scene*>(p)->nodes)
        n += int64_t(nd.children.size());
    return n;
}
int64_t orc_dump(void* p, float* aabb, int32_t* isLeaf, int64_t* childOffset, int64_t* children)
{
    const Scene& s = *static_cast<Scene*>(p);
    int64_t off = 0;
    for (size_t i = 0; i < s.nodes.size(); i++) {
        put(aabb + 6 * i, s.nodes[i].lo);
        put(aabb + 6 * i + 3, s.nodes[i].hi);
        isLeaf[i] = s.nodes[i].leaf;
        childOffset[i] = off;
        for (size_t c : s.nodes[i].children)
            children[off++] = int64_t(c);
    }
    childOffset[s.nodes.size()] = off;
    return s.root;
}

// counters: 6 x uint64 (primary, shadow, bounce, boxTests, triTests, shaded); may be NULL
void orc_intersect(void* p, const rtx_ray* rays, int64_t n, const rtx_features* f, rtx_hit* out, uint64_t* counters, int threads)
{
    const Scene& s = *static_cast<Scene*>(p);
    const Feat feat = toFeat(f);
    Counters total;
    if (threads <= 0)
        threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
    {
        Counters c;
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < n; i++) {
            Ray r;
            r.o = mk(rays[i].origin);
            r.d = mk(rays[i].direction);
            r.t = rays[i].t;
            Hit h;
            c.raysPrimary++;
            const bool hit = intersect(s, r, h, feat, c);
            rtx_hit& o = out[i];
            std::memset(&o, 0, sizeof(o));
            o.t = r.t;
            o.hit = hit;
            o.tri = RTX_TRI_NONE;
            o.material = -1;
            if (hit) {
                put(o.normal, h.normal);
                o.tri = h.triGlobal;
                o.material = h.material;
                if (h.mesh >= 0) {
                    put(o.bary, h.bary);
                    o.uv[0] = h.u;
                    o.uv[1] = h.v;
                }
            }
        }
#pragma omp critical
        total.add(c);
    }
    if (counters) {
        counters[0] = total.raysPrimary;
        counters[1] = total.raysShadow;
        counters[2] = total.raysBounce;
        counters[3] = total.boxTests;
        counters[4] = total.triTests;
        counters[5] = total.shaded;
    }
}

int orc_render(void* p, const rtx_camera* cam, const rtx_features* f, const rtx_render_params* rp, float* outRgb, uint64_t* counters, int threads)
{
    const Scene& s = *static_cast<Scene*>(p);
    const Feat feat = toFeat(f);
    const int W = rp->width, H = rp->height;
    const Camera camera = toCamera(cam, W, H);
    const Camera prev = toCamera(&rp->prev_camera, W, H);
    Counters total;
    if (threads <= 0)
        threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
    {
        Ctx cx { s, feat, rp->shadow_samples > 0 ? rp->shadow_samples : 50, rp->glossy_rays > 0 ? rp->glossy_rays : 5, PixelRng {}, Counters {} };
        cx.rng.seed = rp->seed;
#pragma omp for schedule(dynamic, 1)
        for (int y = 0; y < H; y++) {
            for (int x = 0; x < W; x++) {
                const V3 col = renderPixel(cx, camera, prev, *rp, x, y);
                put(outRgb + (size_t(H - 1 - y) * size_t(W) + size_t(x)) * 3, col); // src/screen.cpp:37-43
            }
        }
#pragma omp critical
        total.add(cx.c);
    }
    if (counters) {
        counters[0] = total.raysPrimary;
        counters[1] = total.raysShadow;
        counters[2] = total.raysBounce;
        counters[3] = total.boxTests;
        counters[4] = total.triTests;
        counters[5] = total.shaded;
    }
    return 0;
}

// getFinalColor for a batch of rays; ray i draws from the stream keyed (seed, i, 0)
void orc_shade(void* p, const rtx_ray* rays, int64_t n, const rtx_features* f, const rtx_render_params* rp, float* out, uint64_t* counters, int threads)
{
    const Scene& s = *static_cast<Scene*>(p);
    const Feat feat = toFeat(f);
    const int depth = rp->ray_depth >= 0 ? rp->ray_depth : 0;
    Counters total;
    if (threads <= 0)
        threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
    {
        Ctx cx { s, feat, rp->shadow_samples > 0 ? rp->shadow_samples : 50, rp->glossy_rays > 0 ? rp->glossy_rays : 5, PixelRng {}, Counters {} };
        cx.rng.seed = rp->seed;
#pragma omp for schedule(dynamic, 64)
        for (int64_t i = 0; i < n; i++) {
            Ray r;
            r.o = mk(rays[i].origin);
            r.d = mk(rays[i].direction);
            r.t = rays[i].t;
            cx.rng.start(uint32_t(i), 0);
            put(out + 3 * i, finalColor(cx, r, depth, kPrimary));
        }
#pragma omp critical
        total.add(cx.c);
    }
    if (counters) {
        counters[0] = total.raysPrimary;
        counters[1] = total.raysShadow;
        counters[2] = total.raysBounce;
        counters[3] = total.boxTests;
        counters[4] = total.triTests;
        counters[5] = total.shaded;
    }
}

// ------------------------------------------------- per-function known answers
int orc_kat_slab(const float* lo, const float* hi, rtx_ray* ray)
{
    Ray r;
    r.o = mk(ray->origin);
    r.d = mk(ray->direction);
    r.t = ray->t;
    const bool ok = slab(mk(lo), mk(hi), r);
    ray->t = r.t;
    return ok;
}
int orc_kat_triangle(const float* a, const float* b, const float* c, rtx_ray* ray, float* bary, float* n)
{
    Ray r;
    r.o = mk(ray->origin);
    r.d = mk(ray->direction);
    r.t = ray->t;
    Hit h;
    const bool ok = triangle(mk(a), mk(b), mk(c), r, h);
    ray->t = r.t;
    put(bary, h.bary);
    put(n, h.normal);
    return ok;
}
int orc_kat_sphere(const float* center, float radius, rtx_ray* ray, float* n)
{
    Ray r;
    r.o = mk(ray->origin);
    r.d = mk(ray->direction);
    r.t = ray->t;
    Hit h;
    const bool ok = sphereTest(Sphere { mk(center), radius, 0 }, r, h);
    ray->t = r.t;
    put(n, h.normal);
    return ok;
}
void orc_kat_barycentric(const float* a, const float* b, const float* c, const float* pnt, float* out) { put(out, barycentric(mk(a), mk(b), mk(c), mk(pnt))); }
void orc_kat_shading(const float* lightPos, const float* lightCol, const rtx_ray* ray, const float* normal, const float* mat, float* out)
{
    Scene s;
    s.materials.push_back(toMaterial(mat, -1));
    Feat f {};
    f.shading = true;
    Hit h;
    h.normal = mk(normal);
    h.material = 0;
    Ray r;
    r.o = mk(ray->origin);
    r.d = mk(ray->direction);
    r.t = ray->t;
    put(out, computeShading(s, mk(lightPos), mk(lightCol), f, r, h));
}
void orc_kat_reflect(const rtx_ray* ray, const float* normal, rtx_ray* out)
{
    Ray r;
    r.o = mk(ray->origin);
    r.d = mk(ray->direction);
    r.t = ray->t;
    const Ray q = reflectionRay(r, mk(normal));
    put(out->origin, q.o);
    put(out->direction, q.d);
    out->t = q.t;
}
void orc_kat_texture(const float* rgb, int w, int h, int mode, const float* uv, int n, float* out)
{
    Scene s;
    const Texture& t = s.textures[size_t(addTexture(s, rgb, w, h))];
    for (int i = 0; i < n; i++)
        put(out + 3 * i, mode == 0 ? texelNearest(t, uv[2 * i], uv[2 * i + 1]) : texelBilinear(t, 0, uv[2 * i], uv[2 * i + 1]));
}
int orc_kat_mipmap(const float* rgb, int w, int h, int* sizes, float* pixels)
{
    Scene s;
    const Texture& t = s.textures[size_t(addTexture(s, rgb, w, h))];
    size_t off = 0;
    for (size_t l = 0; l < t.pix.size(); l++) {
        if (sizes) {
            sizes[2 * l] = t.w[l];
            sizes[2 * l + 1] = t.h[l];
        }
        if (pixels)
            std::memcpy(pixels + off, t.pix[l].data(), t.pix[l].size() * 3 * sizeof(float));
        off += t.pix[l].size() * 3;
    }
    return int(t.pix.size());
}
void orc_kat_envmap(const float* rgb, int w, int h, int mapping, float vfovDegrees, const float* bg, int bilinear, const float* dirs, int n, float* out)
{
    Scene s;
    orc_scene_set_envmap(&s, rgb, w, h, mapping, vfovDegrees, bg);
    Feat f {};
    f.bilinear = bilinear;
    for (int i = 0; i < n; i++) {
        Ray r;
        r.o = mk(0, 0, 0);
        r.d = mk(dirs + 3 * i);
        put(out + 3 * i, envColor(s, r, f));
    }
}
// radiance bins of an environment image: up to cap rectangles (lx, ly, ux, uy); returns their number
int orc_kat_env_bins(const float* rgb, int w, int h, float* out, int cap)
{
    Scene s;
    const float bg[3] = { 0, 0, 0 };
    orc_scene_set_envmap(&s, rgb, w, h, RTX_ENV_SPHERICAL, 180.f, bg);
    for (size_t i = 0; i < s.envBins.size() && int(i) < cap; i++) {
        out[4 * i + 0] = s.envBins[i].lx, out[4 * i + 1] = s.envBins[i].ly;
        out[4 * i + 2] = s.envBins[i].ux, out[4 * i + 3] = s.envBins[i].uy;
    }
    return int(s.envBins.size());
}
// sample2D of a rectangle with the engine seeded from draw `index` of (seed, pixel, sample 0)
void orc_kat_sample2d(const float* rect, int h, int k, uint32_t seed, uint32_t pixel, uint32_t index, float* out)
{
    std::vector<float> v;
    sample2D(Scene::Rect { rect[0], rect[1], rect[2], rect[3] }, h, k, sampleKey(seed, pixel, 0), index, v);
    std::copy(v.begin(), v.end(), out);
}
// EnvironmentMap::getSamplingRay for (seed, pixel), counter from 0; returns the number of rays (<= cap written)
int orc_kat_env_rays(const float* rgb, int w, int h, int mapping, float vfovDegrees, const float* position, const float* normal, int n,
    uint32_t seed, uint32_t pixel, rtx_ray* out, int cap)
{
    Scene s;
    const float bg[3] = { 0, 0, 0 };
    orc_scene_set_envmap(&s, rgb, w, h, mapping, vfovDegrees, bg);
    Ctx cx { s, Feat {}, n, 1, PixelRng {}, Counters {} };
    cx.rng.seed = seed;
    cx.rng.start(pixel, 0);
    std::vector<Ray> rays;
    envSamplingRays(cx, mk(position), mk(normal), n, rays);
    for (size_t i = 0; i < rays.size() && int(i) < cap; i++) {
        put(out[i].origin, rays[i].o);
        put(out[i].direction, rays[i].d);
        out[i].t = rays[i].t;
    }
    return int(rays.size());
}
void orc_kat_light_samples(const rtx_light* l, uint32_t seed, uint32_t pixel, int n, float* pos, float* col)
{
    PixelRng rng;
    rng.seed = seed;
    rng.start(pixel, 0);
    for (int i = 0; i < n; i++) {
        V3 pp, cc;
        if (l->type == RTX_LIGHT_SEGMENT) {
            const V3 e0 = mk(l->p[0]), e1 = mk(l->p[1]);
            const float alpha = rng.next();
            pp = alpha * (e1 - e0) + e0;
            cc = (1 - alpha) * mk(l->c[0]) + alpha * mk(l->c[1]);
        } else {
            const float a1 = rng.next();
            const V3 c1 = (1 - a1) * mk(l->c[0]) + a1 * mk(l->c[1]);
            const V3 c2 = (1 - a1) * mk(l->c[2]) + a1 * mk(l->c[3]);
            const float a2 = rng.next();
            cc = (1 - a2) * c1 + a2 * c2;
            pp = mk(l->p[0]) + mk(l->p[1]) * a1 + mk(l->p[2]) * a2;
        }
        put(pos + 3 * i, pp);
        put(col + 3 * i, cc);
    }
}
void orc_kat_glossy(const rtx_ray* reflection, float degreeBlur, int nRays, uint32_t seed, uint32_t pixel, rtx_ray* out)
{
    Scene s;
    Ctx cx { s, Feat {}, 50, nRays, PixelRng {}, Counters {} };
    cx.rng.seed = seed;
    cx.rng.start(pixel, 0);
    Ray r;
    r.o = mk(reflection->origin);
    r.d = mk(reflection->direction);
    r.t = reflection->t;
    std::vector<Ray> rays;
    glossyFan(cx, r, degreeBlur, rays);
    for (size_t i = 0; i < rays.size(); i++) {
        put(out[i].origin, rays[i].o);
        put(out[i].direction, rays[i].d);
        out[i].t = rays[i].t;
    }
}
void orc_kat_camera(const rtx_camera* cam, float fovy, float aspect, const float* ndc, int n, rtx_ray* out)
{
    Camera c;
    c.lookAt = mk(cam->look_at);
    c.euler = mk(cam->euler);
    c.dist = cam->distance;
    c.halfH = std::tan(fovy / 2.0f);
    c.halfW = aspect * c.halfH;
    for (int i = 0; i < n; i++) {
        const Ray r = c.generate(ndc[2 * i], ndc[2 * i + 1]);
        put(out[i].origin, r.o);
        put(out[i].direction, r.d);
        out[i].t = r.t;
    }
}
// draws of the driver-level engine: out[i] = i-th uniform in [a,b) after `skip` draws
void orc_kat_minstd(uint64_t skip, int n, float a, float b, float* out)
{
    Minstd m = Minstd::at(skip);
    for (int i = 0; i < n; i++)
        out[i] = m.uniform(a, b);
}

} // extern "C"
