// Drop-in body of the reference's BvhInterface (src/bvh_interface.h:12-41; replaces
// src/bvh_interface.cpp and src/bounding_volume_hierarchy.cpp): the tree is built on the
// host by the reference's own split rules (rt::flattenScene), uploaded once, and
// BvhInterface::intersect becomes a one-ray rtx_trace.
#include "rtx_binding.h"

#include <atomic>
#include <cstring>
#include <stdexcept>

// ---- private members read (never written) by the binding (tags at namespace scope, so that
// the friend each instantiation defines is the function its tag declares)
struct EnvTexture { typedef const std::optional<Image> EnvironmentMap::*type; friend type memberOf(EnvTexture); };
struct EnvBackground { typedef const glm::vec3 EnvironmentMap::*type; friend type memberOf(EnvBackground); };
struct EnvFov { typedef const float EnvironmentMap::*type; friend type memberOf(EnvFov); };
struct EnvMapping { typedef const EnvironmentMappingType EnvironmentMap::*type; friend type memberOf(EnvMapping); };
struct BvhImpl { typedef BoundingVolumeHierarchy* BvhInterface::*type; friend type memberOf(BvhImpl); };

template struct PrivateMember<EnvTexture, &EnvironmentMap::texture>;
template struct PrivateMember<EnvBackground, &EnvironmentMap::backgroundColor>;
template struct PrivateMember<EnvFov, &EnvironmentMap::verticalFOVFactor>;
template struct PrivateMember<EnvMapping, &EnvironmentMap::mappingType>;
template struct PrivateMember<BvhImpl, &BvhInterface::m_impl>;

namespace {

rt::vec3 v3(const glm::vec3& v) { return rt::vec3(v.x, v.y, v.z); }

std::shared_ptr<rt::Image> toRt(const Image& im)
{
    auto out = std::make_shared<rt::Image>();
    out->width = im.width;
    out->height = im.height;
    out->pixels.resize(im.pixels.size());
    static_assert(sizeof(glm::vec3) == sizeof(rt::vec3), "three packed floats");
    if (!im.pixels.empty())
        std::memcpy(out->pixels.data(), im.pixels.data(), im.pixels.size() * sizeof(glm::vec3));
    return out;
}

rt::Material toRt(const Material& m)
{
    rt::Material o;
    o.kd = v3(m.kd);
    o.ks = v3(m.ks);
    o.shininess = m.shininess;
    o.transparency = m.transparency;
    if (m.kdTexture)
        o.kdTexture = toRt(*m.kdTexture);
    return o;
}

// The reference's Scene (src/scene.h:31-37) in the host layer's types: same fields, same
// order, so this is a field-by-field copy (vertices and triangles are bit copies).
rt::Scene toRt(const Scene& s, std::vector<rt::EnvironmentMap>& envStore, float& envFovRadians)
{
    rt::Scene o;
    static_assert(sizeof(Vertex) == sizeof(rt::Vertex) && sizeof(glm::uvec3) == sizeof(rt::uvec3), "same layouts");
    for (const Mesh& m : s.meshes) {
        rt::Mesh r;
        r.vertices.resize(m.vertices.size());
        r.triangles.resize(m.triangles.size());
        if (!m.vertices.empty())
            std::memcpy(static_cast<void*>(r.vertices.data()), m.vertices.data(), m.vertices.size() * sizeof(Vertex));
        if (!m.triangles.empty())
            std::memcpy(static_cast<void*>(r.triangles.data()), m.triangles.data(), m.triangles.size() * sizeof(glm::uvec3));
        r.material = toRt(m.material);
        o.meshes.push_back(std::move(r));
    }
    for (const Sphere& sp : s.spheres) {
        rt::Sphere r;
        r.center = v3(sp.center);
        r.radius = sp.radius;
        r.material = toRt(sp.material);
        o.spheres.push_back(r);
    }
    for (const auto& l : s.lights) {
        if (std::holds_alternative<PointLight>(l)) {
            const auto& q = std::get<PointLight>(l);
            o.lights.push_back(rt::PointLight { v3(q.position), v3(q.color) });
        } else if (std::holds_alternative<SegmentLight>(l)) {
            const auto& q = std::get<SegmentLight>(l);
            o.lights.push_back(rt::SegmentLight { v3(q.endpoint0), v3(q.endpoint1), v3(q.color0), v3(q.color1) });
        } else {
            const auto& q = std::get<ParallelogramLight>(l);
            o.lights.push_back(rt::ParallelogramLight { v3(q.v0), v3(q.edge01), v3(q.edge02), v3(q.color0), v3(q.color1), v3(q.color2), v3(q.color3) });
        }
    }
    envFovRadians = -1.f;
    if (!s.environmentMap.empty() && s.environmentMap[0]) { // only the first map is ever looked up (src/render.cpp:62)
        const EnvironmentMap& e = *s.environmentMap[0];
        rt::EnvironmentMap r;
        const std::optional<Image>& tex = e.*memberOf(EnvTexture());
        if (tex)
            r.texture = toRt(*tex);
        r.backgroundColor = v3(e.*memberOf(EnvBackground()));
        r.mappingType = e.*memberOf(EnvMapping()) == SPHERICAL ? rt::SPHERICAL : rt::CYLINDRICAL;
        envFovRadians = e.*memberOf(EnvFov()); // already radians (src/environment_mapping.cpp:44)
        envStore.push_back(r);
        o.environmentMap.push_back(&envStore.back());
    }
    return o;
}

} // namespace

rtx_features toRtx(const Features& f)
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

rtx_scene* deviceSceneOf(const BvhInterface& bvh) { return (bvh.*memberOf(BvhImpl()))->dev; }

void bindThisThread()
{
    thread_local bool bound = false;
    if (bound)
        return;
    static std::atomic<int> nextSlot { 0 };
    if (rtx_init(0) != RTX_OK || rtx_select_slot(nextSlot++ % RTX_FRAME_SLOTS) != RTX_OK)
        throw std::runtime_error(rtx_last_error());
    bound = true;
}

BvhInterface::BvhInterface(Scene* pScene, const Features& features)
{
    bindThisThread();
    rt::Features rf;
    rf.enableShading = features.enableShading, rf.enableRecursive = features.enableRecursive, rf.enableHardShadow = features.enableHardShadow;
    rf.enableSoftShadow = features.enableSoftShadow, rf.enableNormalInterp = features.enableNormalInterp;
    rf.enableTextureMapping = features.enableTextureMapping, rf.enableAccelStructure = features.enableAccelStructure;
    static_assert(sizeof(ExtraFeatures) == sizeof(rt::ExtraFeatures), "the same thirteen switches");
    std::memcpy(&rf.extra, &features.extra, sizeof(rf.extra));
    auto* h = new BoundingVolumeHierarchy;
    std::vector<rt::EnvironmentMap> envStore;
    envStore.reserve(1);
    float envFov = -1.f;
    h->flat = rt::flattenScene(toRt(*pScene, envStore, envFov), rf);
    if (envFov >= 0.f)
        h->flat.envmap.vfov_radians = envFov;
    h->scene = pScene;
    const rtx_scene_desc d = h->flat.desc();
    h->dev = rtx_scene_create(&d); // one upload; nothing of the tree stays on the host path
    if (!h->dev) {
        delete h;
        throw std::runtime_error(rtx_last_error());
    }
    m_impl = h; // (never freed and copied by value, as in the reference: src/main.cpp:218)
}

int BvhInterface::numLevels() const { return m_impl->flat.numLevels; }
int BvhInterface::numLeaves() const { return m_impl->flat.numLeaves; }
void BvhInterface::debugDrawLevel(int) {} // OpenGL visual debug: not on this path
void BvhInterface::debugDrawLeaf(int) {}

bool BvhInterface::intersect(Ray& ray, HitInfo& hitInfo, const Features& features) const
{
    rtx_ray r;
    std::memcpy(r.origin, &ray.origin, 12);
    std::memcpy(r.direction, &ray.direction, 12);
    r.t = ray.t;
    rtx_hit h;
    const rtx_features f = toRtx(features);
    bindThisThread();
    if (rtx_trace(m_impl->dev, &r, 1, &h, &f, nullptr) != RTX_OK)
        throw std::runtime_error(rtx_last_error());
    if (!h.hit)
        return false;
    ray.t = h.t;
    std::memcpy(&hitInfo.normal, h.normal, 12);
    if (h.tri >= 0) { // triangleIntersectUpdate's outputs, src/bounding_volume_hierarchy.cpp:277-295
        Mesh& mesh = m_impl->scene->meshes[size_t(m_impl->flat.idMesh[size_t(h.tri)])];
        std::memcpy(&hitInfo.barycentricCoord, h.bary, 12);
        std::memcpy(&hitInfo.texCoord, h.uv, 8);
        hitInfo.material = mesh.material;
        hitInfo.mesh = &mesh;
        hitInfo.triangle = mesh.triangles[size_t(m_impl->flat.idLocal[size_t(h.tri)])];
    } else {
        hitInfo.material = m_impl->scene->spheres[size_t(-2 - h.tri)].material; // src/intersect.cpp:120-123
    }
    return true;
}
