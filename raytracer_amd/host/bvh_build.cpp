#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>

namespace rt {
namespace {

struct Box {
    vec3 lo, hi;
};
struct Prim {
    int mesh, tri;
    vec3 centroid;
    Box box;
};

inline float mn3(float a, float b, float c) { return std::min({ a, b, c }); }
inline float mx3(float a, float b, float c) { return std::max({ a, b, c }); }

// mergeAABs, src/bounding_volume_hierarchy.cpp:56-62
Box merge(const Box& a, const Box& b)
{
    auto mn = [](vec3 p, vec3 q) { return vec3(std::min(p.x, q.x), std::min(p.y, q.y), std::min(p.z, q.z)); };
    auto mx = [](vec3 p, vec3 q) { return vec3(std::max(p.x, q.x), std::max(p.y, q.y), std::max(p.z, q.z)); };
    return Box { mn(mn(a.lo, b.lo), mn(a.hi, b.hi)), mx(mx(a.lo, b.lo), mx(a.hi, b.hi)) };
}

// The build runs in two steps.  ORDER: the reference's recursion
// (bvhSplitHelper, src/bounding_volume_hierarchy.cpp:101-138) reduced to what
// decides the tree -- sorting / selecting each node's index range and choosing
// its split.  A node's two halves are disjoint ranges, so big subtrees are ordered
// by separate threads; each range is still sorted by the same std::sort call on
// the same sequence as in a serial build, so the result does not depend on the
// thread count.  EMIT: the final index order IS the leaf order, so the triangle
// records are written independently of each other (threads again), and the nodes
// are numbered in one cheap depth-first pass over the recorded splits.
struct Builder {
    const Scene& scene;
    const std::vector<Prim>& prims;
    FlatScene& out;
    int maxDepth;
    bool sah;

    struct Sub {
        int32_t ref, cnt;
        Box box;
        int levels, leaves;
    };
    struct Split {
        uint64_t key; // range start * 64 + depth: unique per inner node
        size_t at;
    };
    static constexpr size_t kParallelAbove = 1u << 17; // triangles in a subtree worth a thread of its own
    static constexpr int kParallelDepth = 5;           // at most 2^5 ordering threads

    std::atomic<bool> degenerate { false };

    bool isLeaf(size_t n, int depth) const { return (maxDepth > -1 && maxDepth <= depth) || n < 2; }

    void order(size_t* idx, size_t begin, size_t n, int direction, int depth, std::vector<Split>& splits)
    {
        if (isLeaf(n, depth))
            return;
        size_t* r = idx + begin;
        auto cmp = [this, direction](const size_t a, const size_t b) {
            return prims[a].centroid[direction] > prims[b].centroid[direction];
        };
        size_t split;
        if (sah) {
            // getSAHBestSplit, src/bounding_volume_hierarchy.cpp:76-98
            std::sort(r, r + n, cmp);
            const float minimum = prims[r[0]].centroid[direction], maximum = prims[r[n - 1]].centroid[direction];
            float minCost = std::numeric_limits<float>::infinity();
            size_t minIndex = 0;
            for (size_t i = 1; i < n; i++) {
                const float cost = -((prims[r[i - 1]].centroid[direction] - minimum) * i + (maximum - prims[r[i]].centroid[direction]) * (n - i));
                if (minCost > cost) {
                    minCost = cost;
                    minIndex = i;
                }
            }
            split = minIndex;
        } else {
            // getMedian, src/bounding_volume_hierarchy.cpp:65-73
            std::nth_element(r, r + n / 2, r + n, cmp);
            split = n / 2;
        }
        if (split == 0 || split >= n) {
            // (only NaN sweep costs can leave the split at the range's start; the recursion below
            // would take an empty half for a leaf.  Reported by build(), on the calling thread.)
            degenerate.store(true);
            return;
        }
        splits.push_back(Split { uint64_t(begin) * 64u + uint64_t(depth), split });
        const int next = (direction + 1) % 3;
        if (n >= kParallelAbove && depth < kParallelDepth) {
            std::vector<Split> other;
            std::thread t([&] { order(idx, begin + split, n - split, next, depth + 1, other); });
            order(idx, begin, split, next, depth + 1, splits);
            t.join();
            splits.insert(splits.end(), other.begin(), other.end());
        } else {
            order(idx, begin, split, next, depth + 1, splits);
            order(idx, begin + split, n - split, next, depth + 1, splits);
        }
    }

    void writeTriangle(size_t slot, size_t id, bool last)
    {
        const Prim& p = prims[id];
        const Mesh& m = scene.meshes[size_t(p.mesh)];
        const uvec3 t = m.triangles[size_t(p.tri)];
        const Vertex &a = m.vertices[t.x], &b = m.vertices[t.y], &c = m.vertices[t.z];
        rtx_tri r;
        std::memcpy(r.v0, &a.position, 12);
        std::memcpy(r.v1, &b.position, 12);
        std::memcpy(r.v2, &c.position, 12);
        // plane as src/intersect.cpp:50-58 computes it per test; the reference
        // rejects a triangle whose cross product has zero length
        // (src/intersect.cpp:16-19), encoded here as a NaN plane
        const float nan = std::numeric_limits<float>::quiet_NaN();
        vec3 n(nan, nan, nan);
        float d = nan;
        if (!(length(cross(c.position - a.position, b.position - a.position)) == 0.f)) {
            n = normalize(cross(b.position - a.position, c.position - a.position));
            d = dot(a.position, n);
        }
        std::memcpy(r.n, &n, 12);
        r.d = d;
        r.id = int32_t(id);
        r.material = p.mesh; // one material per mesh
        r.flags = last ? RTX_TRI_LAST_IN_LEAF : 0;
        out.tris[slot] = r;
        rtx_tri_attr at;
        std::memcpy(at.n0, &a.normal, 12);
        std::memcpy(at.n1, &b.normal, 12);
        std::memcpy(at.n2, &c.normal, 12);
        std::memcpy(at.uv0, &a.texCoord, 8);
        std::memcpy(at.uv1, &b.texCoord, 8);
        std::memcpy(at.uv2, &c.texCoord, 8);
        at.pad = 0.f;
        out.attrs[slot] = at;
    }

    // nodes in depth-first pre-order; marks the last triangle of every leaf
    Sub number(const size_t* idx, size_t begin, size_t n, int depth, const std::unordered_map<uint64_t, size_t>& splitOf, std::vector<uint8_t>& lastInLeaf)
    {
        if (isLeaf(n, depth)) {
            Box box = prims[idx[begin]].box;
            for (size_t i = 0; i < n; i++)
                box = merge(box, prims[idx[begin + i]].box);
            lastInLeaf[begin + n - 1] = 1;
            return Sub { ~int32_t(begin), int32_t(n), box, 1, 1 };
        }
        const size_t split = splitOf.at(uint64_t(begin) * 64u + uint64_t(depth));
        const int32_t self = int32_t(out.nodes.size());
        out.nodes.emplace_back();
        const Sub l = number(idx, begin, split, depth + 1, splitOf, lastInLeaf);
        const Sub r = number(idx, begin + split, n - split, depth + 1, splitOf, lastInLeaf);
        rtx_node& nd = out.nodes[size_t(self)];
        std::memcpy(nd.lo0, &l.box.lo, 12);
        std::memcpy(nd.hi0, &l.box.hi, 12);
        std::memcpy(nd.lo1, &r.box.lo, 12);
        std::memcpy(nd.hi1, &r.box.hi, 12);
        nd.ref0 = l.ref;
        nd.ref1 = r.ref;
        nd.cnt0 = l.ref < 0 ? l.cnt : 0;
        nd.cnt1 = r.ref < 0 ? r.cnt : 0;
        return Sub { self, 0, merge(l.box, r.box), std::max(l.levels, r.levels) + 1, l.leaves + r.leaves };
    }

    Sub build(size_t* idx, size_t n)
    {
        std::vector<Split> splits;
        order(idx, 0, n, 0, 0, splits);
        if (degenerate.load())
            throw std::invalid_argument("BVH build: a node could not be split (non-finite triangle centroids)");
        std::unordered_map<uint64_t, size_t> splitOf;
        splitOf.reserve(splits.size() * 2);
        for (const Split& s : splits)
            splitOf.emplace(s.key, s.at);
        std::vector<uint8_t> lastInLeaf(n, 0);
        const Sub root = number(idx, 0, n, 0, splitOf, lastInLeaf);
        out.tris.resize(n);
        out.attrs.resize(n);
        const size_t nThreads = n >= kParallelAbove ? std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 16) : 1;
        std::vector<std::thread> pool;
        for (size_t w = 1; w < nThreads; w++)
            pool.emplace_back([&, w] {
                for (size_t k = n * w / nThreads; k < n * (w + 1) / nThreads; k++)
                    writeTriangle(k, idx[k], lastInLeaf[k] != 0);
            });
        for (size_t k = 0; k < n / nThreads; k++)
            writeTriangle(k, idx[k], lastInLeaf[k] != 0);
        for (std::thread& t : pool)
            t.join();
        return root;
    }
};

// transformToMipMap, src/main.cpp:745-771.  The reference reads one row / column
// past a level whose size is odd (undefined there); those reads count as 0.
int addTexture(FlatScene& out, const Image& img)
{
    rtx_texture t;
    std::memset(&t, 0, sizeof(t));
    int w = img.width, h = img.height;
    std::vector<vec3> cur = img.pixels;
    cur.resize(size_t(w) * size_t(h));
    int level = 0;
    for (;;) {
        t.width[level] = w;
        t.height[level] = h;
        t.offset[level] = uint32_t(out.texels.size() / 3);
        t.count[level] = uint32_t(cur.size());
        for (const vec3& p : cur) {
            out.texels.push_back(p.x);
            out.texels.push_back(p.y);
            out.texels.push_back(p.z);
        }
        level++;
        if (cur.size() <= 1 || level >= RTX_MAX_MIP_LEVELS)
            break;
        std::vector<vec3> next;
        auto at = [&](size_t i) { return i < cur.size() ? cur[i] : vec3(0, 0, 0); };
        const size_t W = size_t(w), H = size_t(h);
        for (size_t i = 0; i < H; i += 2)
            for (size_t j = 0; j < W; j += 2) {
                vec3 avg = at(i * W + j) + at(i * W + j + 1) + at((i + 1) * W + j) + at((i + 1) * W + j + 1);
                next.push_back(avg * (1.0f / 4.0f));
            }
        w /= 2;
        h /= 2;
        cur.swap(next);
    }
    t.levels = level;
    out.textures.push_back(t);
    return int(out.textures.size()) - 1;
}

rtx_material toMaterial(FlatScene& out, const Material& m, std::map<const Image*, int>& texIndex)
{
    rtx_material r;
    std::memset(&r, 0, sizeof(r));
    std::memcpy(r.kd, &m.kd, 12);
    std::memcpy(r.ks, &m.ks, 12);
    r.shininess = m.shininess;
    r.transparency = m.transparency;
    r.texture = -1;
    if (m.kdTexture) {
        auto it = texIndex.find(m.kdTexture.get());
        if (it == texIndex.end())
            it = texIndex.emplace(m.kdTexture.get(), addTexture(out, *m.kdTexture)).first;
        r.texture = it->second;
    }
    return r;
}

// Radiance bins of the environment image: EnvironmentMap's constructor and
// buildRadianceHierarchy (src/environment_mapping.cpp:26-29,42-105).  The image is
// halved along its longer side again and again, each time following the half (or
// halves) holding more than half of the radiance, until a piece is dim or small;
// every piece left over is a bin.  Arithmetic as in the reference: radiance in
// double rounded to float, sums in float, pixel ranges by float-to-int truncation.
float radianceOf(const vec3& c) { return float(0.25 * (double(c.x) + 2. * double(c.y) + double(c.z))); }

void radianceBins(const Image& img, float lx, float ly, float ux, float uy, float totalRadiance, std::vector<float>& out)
{
    auto keep = [&](float a, float b, float c, float d) {
        out.push_back(a), out.push_back(b), out.push_back(c), out.push_back(d);
    };
    const float sizeX = ux - lx, sizeY = uy - ly;
    const float minRadiance = float((img.width * img.height) / 1000);
    if (totalRadiance < minRadiance || double(sizeX + sizeY) < 0.1) {
        keep(lx, ly, ux, uy);
        return;
    }
    const float imageW = float(img.width), m1x = float(img.width - 1), m1y = float(img.height - 1);
    const bool splitVertically = sizeY / sizeX > 1.f;
    const float splitX = splitVertically ? ux : float(0.5 * double(lx + ux));
    const float splitY = splitVertically ? float(0.5 * double(ly + uy)) : uy;
    const float fromX = lx * m1x, fromY = ly * m1y, toX = splitX * m1x, toY = splitY * m1y;
    float lowerHalf = 0.f;
    for (int i = int(fromX); float(i) < toX; i++)
        for (int j = int(fromY); float(j) < toY; j++)
            lowerHalf += radianceOf(img.pixels[size_t(float(i) + float(j) * imageW)]);
    const float upperHalf = totalRadiance - lowerHalf;
    if (double(lowerHalf) > 0.5 * double(totalRadiance))
        radianceBins(img, lx, ly, splitX, splitY, lowerHalf, out);
    else
        keep(lx, ly, splitX, splitY);
    const float l2x = splitVertically ? lx : splitX, l2y = splitVertically ? splitY : ly;
    if (double(upperHalf) > 0.5 * double(totalRadiance))
        radianceBins(img, l2x, l2y, ux, uy, upperHalf, out);
    else
        keep(l2x, l2y, ux, uy);
}

} // namespace

FlatScene flattenScene(const Scene& scene, const Features& features, bool buildTree)
{
    FlatScene out;
    std::map<const Image*, int> texIndex;
    // materials: one per mesh (index = mesh index), then one per sphere
    for (const Mesh& m : scene.meshes)
        out.materials.push_back(toMaterial(out, m.material, texIndex));
    for (const Sphere& s : scene.spheres) {
        rtx_sphere r;
        std::memset(&r, 0, sizeof(r));
        std::memcpy(r.center, &s.center, 12);
        r.radius = s.radius;
        r.material = int32_t(out.materials.size());
        out.materials.push_back(toMaterial(out, s.material, texIndex));
        out.spheres.push_back(r);
    }
    for (const auto& l : scene.lights) {
        rtx_light r;
        std::memset(&r, 0, sizeof(r));
        if (std::holds_alternative<PointLight>(l)) {
            const auto& q = std::get<PointLight>(l);
            r.type = RTX_LIGHT_POINT;
            std::memcpy(r.p[0], &q.position, 12);
            std::memcpy(r.c[0], &q.color, 12);
        } else if (std::holds_alternative<SegmentLight>(l)) {
            const auto& q = std::get<SegmentLight>(l);
            r.type = RTX_LIGHT_SEGMENT;
            std::memcpy(r.p[0], &q.endpoint0, 12);
            std::memcpy(r.p[1], &q.endpoint1, 12);
            std::memcpy(r.c[0], &q.color0, 12);
            std::memcpy(r.c[1], &q.color1, 12);
        } else {
            const auto& q = std::get<ParallelogramLight>(l);
            r.type = RTX_LIGHT_PARALLELOGRAM;
            std::memcpy(r.p[0], &q.v0, 12);
            std::memcpy(r.p[1], &q.edge01, 12);
            std::memcpy(r.p[2], &q.edge02, 12);
            std::memcpy(r.c[0], &q.color0, 12);
            std::memcpy(r.c[1], &q.color1, 12);
            std::memcpy(r.c[2], &q.color2, 12);
            std::memcpy(r.c[3], &q.color3, 12);
        }
        out.lights.push_back(r);
    }
    if (!scene.environmentMap.empty() && scene.environmentMap[0]) {
        const EnvironmentMap& e = *scene.environmentMap[0];
        out.envmap.texture = e.texture ? addTexture(out, *e.texture) : -1;
        // only level 0 of the environment image is ever sampled
        out.envmap.mapping = e.mappingType == SPHERICAL ? RTX_ENV_SPHERICAL : RTX_ENV_CYLINDRICAL;
        out.envmap.vfov_radians = radians(e.verticalFieldOfViewDegrees);
        std::memcpy(out.envmap.background, &e.backgroundColor, 12);
        if (e.texture && e.texture->pixels.size() >= size_t(e.texture->width) * size_t(e.texture->height) && !e.texture->pixels.empty()) {
            float total = 0.f;
            for (const vec3& px : e.texture->pixels)
                total += radianceOf(px);
            radianceBins(*e.texture, 0.f, 0.f, 1.f, 1.f, total, out.envBins);
        }
    }

    // meshTrianglePairs in mesh-major order, src/bounding_volume_hierarchy.cpp:166-175
    std::vector<Prim> prims;
    for (size_t mi = 0; mi < scene.meshes.size(); mi++) {
        const Mesh& m = scene.meshes[mi];
        for (size_t j = 0; j < m.triangles.size(); j++) {
            const uvec3 t = m.triangles[j];
            if (t.x >= m.vertices.size() || t.y >= m.vertices.size() || t.z >= m.vertices.size())
                throw std::out_of_range("triangle references a vertex outside its mesh");
            const vec3 p0 = m.vertices[t.x].position, p1 = m.vertices[t.y].position, p2 = m.vertices[t.z].position;
            for (const vec3& q : { p0, p1, p2 })
                if (!std::isfinite(q.x) || !std::isfinite(q.y) || !std::isfinite(q.z))
                    throw std::invalid_argument("mesh " + std::to_string(mi) + ": triangle " + std::to_string(j) + " has a vertex position that is not finite");
            Prim p;
            p.mesh = int(mi);
            p.tri = int(j);
            p.centroid = (1.f / 3.f) * (p0 + p1 + p2);
            p.box = Box { vec3(mn3(p0.x, p1.x, p2.x), mn3(p0.y, p1.y, p2.y), mn3(p0.z, p1.z, p2.z)),
                vec3(mx3(p0.x, p1.x, p2.x), mx3(p0.y, p1.y, p2.y), mx3(p0.z, p1.z, p2.z)) };
            prims.push_back(p);
            out.idMesh.push_back(int32_t(mi));
            out.idLocal.push_back(int32_t(j));
        }
    }
    const size_t n = prims.size();
    if (n == 0)
        return out; // empty tree (e.g. the Spheres scene): rootRef stays INT32_MIN
    if (!buildTree) {
        // the triangle records as they are, in id order, for the device's builder
        Builder b { scene, prims, out, -1, false };
        out.tris.resize(n);
        out.attrs.resize(n);
        const size_t nThreads = n >= Builder::kParallelAbove ? std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 16) : 1;
        std::vector<std::thread> pool;
        for (size_t w = 1; w < nThreads; w++)
            pool.emplace_back([&, w] {
                for (size_t k = n * w / nThreads; k < n * (w + 1) / nThreads; k++)
                    b.writeTriangle(k, k, false);
            });
        for (size_t k = 0; k < n / nThreads; k++)
            b.writeTriangle(k, k, false);
        for (std::thread& t : pool)
            t.join();
        return out;
    }
    std::vector<size_t> idx(n);
    for (size_t i = 0; i < n; i++)
        idx[i] = i;
    out.nodes.reserve(n / 2 + 1);
    // glm::ceil(0.8 * glm::log2((float) n)), src/bounding_volume_hierarchy.cpp:177
    const int maxDepth = int(std::ceil(0.8 * double(std::log2(float(n)))));
    Builder b { scene, prims, out, maxDepth, features.extra.enableBvhSahBinning };
    const Builder::Sub root = b.build(idx.data(), n);
    out.rootRef = root.ref;
    out.rootCnt = root.ref < 0 ? root.cnt : 0;
    std::memcpy(out.rootLo, &root.box.lo, 12);
    std::memcpy(out.rootHi, &root.box.hi, 12);
    out.numLevels = root.levels;
    out.numLeaves = root.leaves;
    return out;
}

rtx_scene_desc FlatScene::desc() const
{
    rtx_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    d.nodes = nodes.data();
    d.n_nodes = uint32_t(nodes.size());
    d.root_ref = rootRef;
    d.root_cnt = rootCnt;
    std::memcpy(d.root_lo, rootLo, 12);
    std::memcpy(d.root_hi, rootHi, 12);
    d.tris = tris.data();
    d.tri_attrs = attrs.data();
    d.n_tris = uint32_t(tris.size());
    d.materials = materials.data();
    d.n_materials = uint32_t(materials.size());
    d.textures = textures.data();
    d.n_textures = uint32_t(textures.size());
    d.texels = texels.data();
    d.n_texels = texels.size() / 3;
    d.lights = lights.data();
    d.n_lights = uint32_t(lights.size());
    d.spheres = spheres.data();
    d.n_spheres = uint32_t(spheres.size());
    d.envmap = envmap;
    d.num_levels = numLevels;
    d.num_leaves = numLeaves;
    d.env_bins = envBins.data();
    d.n_env_bins = uint32_t(envBins.size() / 4);
    return d;
}

} // namespace rt
