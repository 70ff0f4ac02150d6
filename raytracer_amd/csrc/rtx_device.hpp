// Device-side building blocks of the gfx950 ray tracer: fp32 vector algebra in
// the reference's operation order, ray/box and ray/triangle tests, closest-hit
// traversal with a per-lane LDS stack, hit attributes, textures, environment
// map, Phong shading and the samplers.  Compiled with -ffp-contract=off: no
// fused multiply-add anywhere, so every value that feeds a comparison is
// bit-identical to the CPU path's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtx.h"

namespace rtxd {

#define RTX_DEV __device__ __forceinline__

constexpr float kFltMax = 3.402823466e+38f;
constexpr int kWave = 64;
constexpr int kBlock = 256;

// ------------------------------------------------------------------ vec3 ----
// glm 0.9.9.9 scalar semantics (glm/detail/func_geometric.inl:48-90):
// dot = (x*x' + y*y') + z*z', normalize(v) = v * (1 / sqrt(dot(v, v))).
struct V3 {
    float x, y, z;
};
RTX_DEV V3 mk(float x, float y, float z) { return V3 { x, y, z }; }
RTX_DEV V3 mk(const float* p) { return V3 { p[0], p[1], p[2] }; }
RTX_DEV V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
RTX_DEV V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
RTX_DEV V3 operator-(V3 a) { return { -a.x, -a.y, -a.z }; }
RTX_DEV V3 operator*(V3 a, V3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
RTX_DEV V3 operator*(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
RTX_DEV V3 operator*(float s, V3 a) { return { s * a.x, s * a.y, s * a.z }; }
RTX_DEV V3 operator/(V3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
RTX_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RTX_DEV V3 cross(V3 x, V3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
RTX_DEV float length(V3 v) { return __builtin_sqrtf(dot(v, v)); }
RTX_DEV V3 normalize(V3 v) { return v * (1.0f / __builtin_sqrtf(dot(v, v))); }
RTX_DEV float distance(V3 p0, V3 p1) { return length(p1 - p0); }

// Two floats that one packed instruction (v_pk_add_f32 / v_pk_mul_f32) works on.
// Each half is an ordinary IEEE single operation, so packing changes the
// instruction count and nothing else.
typedef float F2 __attribute__((ext_vector_type(2)));
RTX_DEV F2 both(float v) { return F2 { v, v }; }

struct Ray {
    V3 o, d;
    float t;
};

// The records of the wave-shared walk re-expressed for rays that all start at ONE point (the
// camera rays of a pinhole camera; the shadow rays of a point light, which start at the light):
// `nodes` = the eight octant copies of DScene::waveNodes with every box plane replaced by
// (plane - origin) in that copy's mirror image, `tris` = DScene::waveTris with each plane's D
// replaced by D - dot(n, origin).  These are the very subtractions and the very dot product
// every ray of such a bundle would make at every visit (same operands, same fp32 operations,
// same order), made once per origin instead: the walk's arithmetic and results are bit for bit
// the same.  (Whole copies rather than side arrays: the walk then needs no more scalar
// registers than without a frame.)
// A frame's triangle records are kFrameTriQuads float4 long: three quads of a PRETEST in front
// of the six of DScene::waveTris.  With a_i = v_i - origin, the three vectors c_i = +-(a_i x
// a_(i+1)) (sign: such that a ray from the origin towards the triangle's inside has
// dot(d, c_i) > 0 for all three) tell by three dot products on which side of each edge a ray
// from the origin passes: a ray with dot(d, c_i) < -M * max|d_k| for some i passes the
// triangle by further than anything the reference's fp32 test could still take for a hit
// (M: per triangle, see originFrameKernel), and the wave skips the plane division and the
// edge tests of a triangle that all its rays pass that clearly -- most of a leaf's triangles
// for the narrow bundle of a wave.  Quads: (c_0.x, c_1.x, c_0.y, c_1.y), (c_0.z, c_1.z, c_2.x, c_2.y),
// (c_2.z, M, last-in-leaf flag: sign bit, -).
struct OriginFrame {
    const float4* nodes;
    const float4* tris;
};
constexpr int kLightFrames = 4; // point lights of a scene that get a frame at upload (the first ones)

// --------------------------------------------------------- device scene -----
struct DScene {
    // Node and triangle records live in ONE array, nodes first: record r of a child
    // reference is r itself for an inner node and ~r for a triangle (the device's
    // references are made at upload: leaf k of the flattened tree is ~(nNodes + k)), so
    // the traversal finds a lane's next record, whichever kind, with two instructions.
    const float4* nodes;     // 4 x float4 per rtx_node; the array's start
    const float4* tris;      // 4 x float4 per rtx_tri (leaf order) = nodes + 4 * nNodes
    int32_t nNodes;
    // The tree of the wave-shared walk (traverseBvhWave), read through the scalar cache: the
    // binary tree with every inner node merged with its inner children, i.e. up to four
    // boxes per node, in eight copies, one per direction octant.  Octant c's copy holds the
    // boxes in that octant's mirror image (see traverseBvhWave) with each node's children
    // sorted nearest first for its rays: 8 x float4 per node = lo.x of the four children,
    // hi.x, lo.y, hi.y, lo.z, hi.z, the four child references (>= 0: node, < 0: ~first
    // triangle of a leaf; an unused slot has an empty box), one spare.  Copy c starts at
    // node c * nWaveNodes.  Triangles (leaf order): 6 x float4 = (n, D), (v0, id word),
    // (v1, -), v2 - v0, v2 - v1, v1 - v0.
    const float4* waveNodes;
    const float4* waveTris;
    OriginFrame lightFrame[kLightFrames]; // light i's frame if it is a point light (nodes == nullptr: none)
    int32_t nWaveNodes;
    int32_t waveRoot;        // 0 (the root node), ~first triangle when the root is a leaf
    int32_t boxesNested;     // every child box lies within its parent's box, the root's children within the root box
    const float4* attrs;     // 4 x float4 per rtx_tri_attr
    const int32_t* idOrder;  // leaf-order record index of triangle id i (brute-force path)
    const rtx_material* mats;
    const rtx_texture* textures;
    const float* texels;
    const rtx_light* lights;
    const rtx_sphere* spheres;
    rtx_envmap env;
    const float4* envBins;   // radiance bins of the environment image: (lower.x, lower.y, upper.x, upper.y)
    int32_t nEnvBins;
    int32_t nTris, nLights, nSpheres, nMaterials;
    int32_t rootRef;
    float rootLo[3], rootHi[3];
    int32_t frontierOk;      // the four-wide records carry their leaf children's triangle counts and every index fits a frontier item (traverseFrontier)
};

constexpr int32_t kEmptyRoot = INT32_MIN;
#ifdef RTX_PROBE
// diagnostic build: the wave-shared walk's triangle tests -- all, those some lane passes the plane stage of, those
// some lane hits -- and its node visits, summed over the process
__device__ unsigned long long g_waveWalkStats[8];
#endif

// ------------------------------------------------------------ primitives ----
// Ray/box slab test, src/intersect.cpp:129-158.  Comparisons are written as
// compare+select (never fmin/fmax) so an axis-parallel ray's NaN (0 * inf)
// leaves `first`/`second` untouched exactly as in the reference.  Returns
// whether the box is entered before `tmax`; `key` is the entry distance, or the
// exit distance when the entry is not in front of the origin.
RTX_DEV bool slab(V3 lo, V3 hi, V3 o, V3 inv, float tmax, float& key)
{
    float first = 0.0f, second = __builtin_inff();
    bool fail = false;
    {
        float t0 = (lo.x - o.x) * inv.x, t1 = (hi.x - o.x) * inv.x;
        const bool sw = t0 > t1;
        const float a = sw ? t1 : t0, b = sw ? t0 : t1;
        fail |= (a > second) || (b < first);
        first = (a > first) ? a : first;
        second = (b < second) ? b : second;
    }
    {
        float t0 = (lo.y - o.y) * inv.y, t1 = (hi.y - o.y) * inv.y;
        const bool sw = t0 > t1;
        const float a = sw ? t1 : t0, b = sw ? t0 : t1;
        fail |= (a > second) || (b < first);
        first = (a > first) ? a : first;
        second = (b < second) ? b : second;
    }
    {
        float t0 = (lo.z - o.z) * inv.z, t1 = (hi.z - o.z) * inv.z;
        const bool sw = t0 > t1;
        const float a = sw ? t1 : t0, b = sw ? t0 : t1;
        fail |= (a > second) || (b < first);
        first = (a > first) ? a : first;
        second = (b < second) ? b : second;
    }
    if (first <= 0.0f)
        first = second;
    key = first;
    return !fail && !(tmax < first);
}

// src/bounding_volume_hierarchy.cpp:297-307
RTX_DEV bool inBox(V3 lo, V3 hi, V3 x)
{
    return lo.x <= x.x && x.x <= hi.x && lo.y <= x.y && x.y <= hi.y && lo.z <= x.z && x.z <= hi.z;
}

// Child visit rule of the reference's traversal
// (src/bounding_volume_hierarchy.cpp:412-439): a child containing the origin is
// always visited with key 0, otherwise the slab test decides.
RTX_DEV bool childTest(V3 lo, V3 hi, V3 o, V3 inv, float tmax, float& key)
{
    float k;
    const bool s = slab(lo, hi, o, inv, tmax, k);
    const bool in = inBox(lo, hi, o);
    key = in ? 0.0f : k;
    return in || s;
}

// The same visit rule for a ray whose three reciprocal direction components are
// finite: no product can be NaN then, so ordering each axis pair with min/max,
// taking first = max(0, entries) and second = min(exits), and rejecting on
// first > second gives bit for bit the reference's running comparisons; and
// lo <= o <= hi is read off the signs of the differences already computed.
RTX_DEV bool childTestFinite(V3 lo, V3 hi, V3 o, V3 inv, float tmax, float& key)
{
    const float dlx = lo.x - o.x, dly = lo.y - o.y, dlz = lo.z - o.z;
    const float dhx = hi.x - o.x, dhy = hi.y - o.y, dhz = hi.z - o.z;
    const float t0x = dlx * inv.x, t1x = dhx * inv.x;
    const float t0y = dly * inv.y, t1y = dhy * inv.y;
    const float t0z = dlz * inv.z, t1z = dhz * inv.z;
    float first = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
    const float second = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    const bool miss = first > second;
    if (first <= 0.0f)
        first = second;
    const bool in = fmaxf(fmaxf(dlx, dly), dlz) <= 0.0f && fminf(fminf(dhx, dhy), dhz) >= 0.0f;
    key = in ? 0.0f : first;
    return in || (!miss && !(tmax < first));
}

// childTestFinite for both children of a node at once.  The device keeps a node as
// X = (lo0.x, lo1.x, hi0.x, hi1.x), Y, Z likewise: the two children sit in the two
// halves of packed operands, so the 12 subtractions and 12 multiplications of the
// two slab tests are 6 + 6 packed instructions.
RTX_DEV void nodeTestFinite(float4 X, float4 Y, float4 Z, V3 o, V3 inv, float tmax, bool& h0, bool& h1, float& k0, float& k1)
{
    const F2 dlx = F2 { X.x, X.y } - both(o.x), dhx = F2 { X.z, X.w } - both(o.x);
    const F2 dly = F2 { Y.x, Y.y } - both(o.y), dhy = F2 { Y.z, Y.w } - both(o.y);
    const F2 dlz = F2 { Z.x, Z.y } - both(o.z), dhz = F2 { Z.z, Z.w } - both(o.z);
    const F2 t0x = dlx * both(inv.x), t1x = dhx * both(inv.x);
    const F2 t0y = dly * both(inv.y), t1y = dhy * both(inv.y);
    const F2 t0z = dlz * both(inv.z), t1z = dhz * both(inv.z);
    {
        float first = fmaxf(fmaxf(fminf(t0x.x, t1x.x), fminf(t0y.x, t1y.x)), fmaxf(fminf(t0z.x, t1z.x), 0.0f));
        const float second = fminf(fminf(fmaxf(t0x.x, t1x.x), fmaxf(t0y.x, t1y.x)), fmaxf(t0z.x, t1z.x));
        const bool miss = first > second;
        if (first <= 0.0f)
            first = second;
        const bool in = fmaxf(fmaxf(dlx.x, dly.x), dlz.x) <= 0.0f && fminf(fminf(dhx.x, dhy.x), dhz.x) >= 0.0f;
        k0 = in ? 0.0f : first;
        h0 = in || (!miss && !(tmax < first));
    }
    {
        float first = fmaxf(fmaxf(fminf(t0x.y, t1x.y), fminf(t0y.y, t1y.y)), fmaxf(fminf(t0z.y, t1z.y), 0.0f));
        const float second = fminf(fminf(fmaxf(t0x.y, t1x.y), fmaxf(t0y.y, t1y.y)), fmaxf(t0z.y, t1z.y));
        const bool miss = first > second;
        if (first <= 0.0f)
            first = second;
        const bool in = fmaxf(fmaxf(dlx.y, dly.y), dlz.y) <= 0.0f && fminf(fminf(dhx.y, dhy.y), dhz.y) >= 0.0f;
        k1 = in ? 0.0f : first;
        h1 = in || (!miss && !(tmax < first));
    }
}

// Ray/triangle, src/intersect.cpp:32-89, with the plane (n, D) precomputed on
// the host by the same arithmetic (src/intersect.cpp:50-58).  A degenerate
// triangle carries NaN in n and D, which fails `t < tmax` below just as the
// reference's explicit degeneracy test rejects it.
RTX_DEV bool triangleTest(V3 v0, V3 v1, V3 v2, V3 n, float D, V3 o, V3 d, float tmax, float& tOut)
{
    const float dn = dot(n, d);
    if (dn == 0.f)
        return false;
    const float t = (D - dot(n, o)) / dn;
    if (t <= 0.f || !(t < tmax))
        return false;
    const V3 p = o + t * d;
    const float alpha = dot(n, cross(v2 - v1, p - v1));
    const float beta = dot(n, cross(p - v0, v2 - v0));
    const float gamma = dot(n, cross(v1 - v0, p - v0));
    if (alpha < 0 || beta < 0 || gamma < 0)
        return false;
    tOut = t;
    return true;
}

// The same test on the device's triangle record X = (v0.x, v1.x, v2.x, n.x), Y, Z,
// D, with the operations paired into packed instructions: (dot(n,d), dot(n,o))
// together, the edge vectors (v2-v0, v2-v1) and (p-v0, p-v1) together, and the two
// cross products and dot products of beta and alpha together.  Every product, sum
// and difference is the one triangleTest makes, in the same order; the alpha half
// is carried negated (a - b and b - a are exact negations of each other, and so
// are sums of negated terms), so its sign test is flipped.
RTX_DEV bool triangleTestPacked(float4 X, float4 Y, float4 Z, float D, V3 o, V3 d, float tmax, float& tOut)
{
    const F2 nd = (both(X.w) * F2 { d.x, o.x } + both(Y.w) * F2 { d.y, o.y }) + both(Z.w) * F2 { d.z, o.z }; // (n.d, n.o)
    const float dn = nd.x;
    if (dn == 0.f)
        return false;
    const float t = (D - nd.y) / dn;
    if (t <= 0.f || !(t < tmax))
        return false;
    const float px = o.x + t * d.x, py = o.y + t * d.y, pz = o.z + t * d.z;
    // e = (p - v0, p - v1); ab = (v2 - v0, v2 - v1); c = v1 - v0
    const F2 ex = both(px) - F2 { X.x, X.y }, ey = both(py) - F2 { Y.x, Y.y }, ez = both(pz) - F2 { Z.x, Z.y };
    const F2 abx = both(X.z) - F2 { X.x, X.y }, aby = both(Y.z) - F2 { Y.x, Y.y }, abz = both(Z.z) - F2 { Z.x, Z.y };
    const float cx = X.y - X.x, cy = Y.y - Y.x, cz = Z.y - Z.x;
    // first half: cross(p - v0, v2 - v0) (beta); second half: -cross(v2 - v1, p - v1) (-alpha)
    const F2 wx = ey * abz - aby * ez;
    const F2 wy = ez * abx - abz * ex;
    const F2 wz = ex * aby - abx * ey;
    const F2 ba = (both(X.w) * wx + both(Y.w) * wy) + both(Z.w) * wz; // (beta, -alpha)
    // gamma = dot(n, cross(v1 - v0, p - v0))
    const float gx = cy * ez.x - ey.x * cz, gy = cz * ex.x - ez.x * cx, gz = cx * ey.x - ex.x * cy;
    const float gamma = (X.w * gx + Y.w * gy) + Z.w * gz;
    if (ba.y > 0 || ba.x < 0 || gamma < 0)
        return false;
    tOut = t;
    return true;
}

// src/intersect.cpp:93-124.  The roots are evaluated in double and rounded to
// float once, as the reference's unqualified sqrt() call does.
RTX_DEV bool sphereTest(const rtx_sphere& s, V3 o, V3 d, float& t, V3& normal)
{
    const V3 c = mk(s.center);
    const float a = dot(d, d);
    const float b = 2 * dot(d, o - c);
    const float cc = dot(c, c) + dot(o, o) - 2 * dot(o, c) - s.radius * s.radius;
    const float delta = b * b - 4 * a * cc;
    if (delta < 0)
        return false;
    const double sq = __builtin_sqrt(double(delta));
    const float t1 = float((double(-b) + sq) / double(2 * a));
    const float t2 = float((double(-b) - sq) / double(2 * a));
    if (t2 < 0) {
        if (t > t2)
            t = (t < t1) ? t : t1; // std::min(t1, ray.t)
        else
            return false;
    } else {
        if (t1 > 0.f && t > t1)
            t = (t < t2) ? t : t2;
        else
            return false;
    }
    const V3 p = o + t * d;
    normal = normalize(p - c);
    return true;
}

struct TriRec {
    V3 v0, v1, v2, n;
    float d;
    int32_t id, material, flags;
};
// The device's triangle record (64 B, leaf order): X = (v0.x, v1.x, v2.x, n.x),
// Y and Z likewise, W = (D, id | last-in-leaf << 31, material, flags) -- rtx_tri transposed at upload
// so that the pairs the packed test works on are neighbours.
RTX_DEV TriRec loadTri(const DScene& sc, int k)
{
    const float4* p = sc.tris + size_t(k) * 4;
    const float4 X = p[0], Y = p[1], Z = p[2], W = p[3];
    TriRec r;
    r.v0 = mk(X.x, Y.x, Z.x);
    r.v1 = mk(X.y, Y.y, Z.y);
    r.v2 = mk(X.z, Y.z, Z.z);
    r.n = mk(X.w, Y.w, Z.w);
    r.d = W.x;
    r.id = __float_as_int(W.y) & 0x7fffffff; // (the top bit marks the last triangle of a leaf)
    r.material = __float_as_int(W.z);
    r.flags = __float_as_int(W.w);
    return r;
}

// --------------------------------------------------------------- traversal --
// Per-lane traversal stack: the first kLdsStack entries of a lane live in LDS, entry e of lane l
// at stack[e][l], so a wave's 64 lanes always touch 64 different banks whatever their stack
// depths; deeper entries (trees of more than kLdsStack levels, and there only the walks that
// really leave that many subtrees pending) go to the lane's own scratch memory.  The LDS a
// workgroup takes, and with it the number of waves a CU holds, thus does not grow with the
// tree's depth: with all entries in LDS the 32-entry kernels held half the waves of the
// 16-entry ones, and the headline frame took 1.32 ms on them instead of 0.91 ms.
constexpr int kLdsStack = 16;
template <int STACK>
struct LaneStack {
    static constexpr int kLds = STACK < kLdsStack ? STACK : kLdsStack;
    static constexpr int kOver = STACK > kLdsStack ? STACK - kLdsStack : 1;
    int32_t* refs;  // [kLds][kBlock]
    float* keys;    // [kLds][kBlock], directly behind refs: the wave-shared walk uses the two as ONE array of 2 * kLds keys per lane
    int tid;
    int32_t* waveRefs; // [2 * STACK] of this wave: the wave-shared walk's subtree references (the same for every lane)
    int32_t overRefs[kOver];
    float overKeys[kOver];
    float overWaveKeys[2 * kOver];
    // the wave-shared walk's entries: a key per lane, one reference per wave
    RTX_DEV void pushWave(int sp, int32_t ref, float key)
    {
        if (STACK <= kLdsStack || sp < 2 * kLds)
            reinterpret_cast<float*>(refs)[sp * kBlock + tid] = key;
        else
            overWaveKeys[sp - 2 * kLds] = key;
        waveRefs[sp] = ref;
    }
    RTX_DEV float waveKey(int sp) const
    {
        if (STACK <= kLdsStack || sp < 2 * kLds)
            return reinterpret_cast<const float*>(refs)[sp * kBlock + tid];
        return overWaveKeys[sp - 2 * kLds];
    }
    RTX_DEV int32_t waveRef(int sp) const { return __builtin_amdgcn_readfirstlane(waveRefs[sp]); }
    RTX_DEV void push(int sp, int32_t ref, float key)
    {
        if (STACK <= kLdsStack || sp < kLds) {
            refs[sp * kBlock + tid] = ref;
            keys[sp * kBlock + tid] = key;
        } else {
            overRefs[sp - kLds] = ref;
            overKeys[sp - kLds] = key;
        }
    }
    RTX_DEV void pop(int sp, int32_t& ref, float& key)
    {
        if (STACK <= kLdsStack || sp < kLds) {
            ref = refs[sp * kBlock + tid];
            key = keys[sp * kBlock + tid];
        } else {
            ref = overRefs[sp - kLds];
            key = overKeys[sp - kLds];
        }
    }
};

struct TraceCounters {
    uint32_t boxTests, triTests;
#ifdef RTX_PROBE
    uint32_t frontierRounds = 0, frontierEntries = 0; // traverseFrontier: rounds of this wave, live entries this lane handled
    uint32_t outer = 0, inner = 0; // this lane's trips through the traversal loop and the leaf loop
    uint32_t waveOuter = 0, waveInner = 0; // trips in which this lane was the lowest active one: summed over the wave, the wave's trips
#endif
};

// Closest hit of one ray (BoundingVolumeHierarchy::intersect,
// src/bounding_volume_hierarchy.cpp:320-458) without the spheres.  The
// reference visits nodes best-first from a priority queue; this walks the same
// tree depth-first, nearer child first, with the same visit predicates: a
// child is entered only if `t >= key` held when it was tested and still holds
// when it is popped.  The closest hit is the same except when two candidates
// are within rounding of each other (exact-t ties keep the first one tested,
// and the test order differs).
// On return t is the closest distance (unchanged on a miss) and bestK the
// leaf-order record index of the hit triangle (-1 on a miss).
// With EARLY the walk stops as soon as any hit closer than tStop is known (used
// by shadow queries, for which such a hit already decides the answer), and the
// triangle search limit may have been shortened by the caller: rootLimit is the
// unshortened limit, which the root box test must see.
template <int STACK, bool COUNT, bool EARLY = false>
RTX_DEV void traverseBvh(const DScene& sc, LaneStack<STACK>& st, V3 o, V3 d, float& t, int& bestK, TraceCounters& tc, float tStop = 0.f, float rootLimit = 0.f)
{
    bestK = -1;
    if (sc.rootRef == kEmptyRoot)
        return;
    const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    // an axis-parallel ray (1/0 = inf) can produce 0 * inf = NaN in the slab test
    // and must take the comparison-by-comparison form
    const bool finiteInv = fabsf(inv.x) < __builtin_inff() && fabsf(inv.y) < __builtin_inff() && fabsf(inv.z) < __builtin_inff();
    {
        // The root box is tested with the plain slab routine, which reports the EXIT
        // distance for an origin inside the box and rejects when the search limit is
        // below it (src/intersect.cpp:149-153, src/bounding_volume_hierarchy.cpp:369-377).
        // A shadow query whose limit was shortened must therefore test the root with
        // the limit the reference had at that point (rootLimit), not the shortened one.
        float key;
        if (COUNT)
            tc.boxTests++;
        if (!slab(mk(sc.rootLo), mk(sc.rootHi), o, inv, EARLY ? rootLimit : t, key))
            return;
    }
    int sp = 0;
    int32_t cur = sc.rootRef;
    bool have = true;
    // One record per trip whatever its kind: node and triangle records are both four
    // float4, so every lane fetches its next record with the same four loads, the lanes
    // at inner nodes then test their two boxes and the lanes inside leaves their one
    // triangle.  No lane waits for the longest leaf among the wave's lanes.
    while (have) {
#ifdef RTX_PROBE
        tc.outer++;
        tc.waveOuter += uint32_t(__builtin_ctzll(__ballot(1))) == (threadIdx.x & 63u);
#endif
        const bool isNode = cur >= 0;
        const uint32_t record = uint32_t(cur ^ (cur >> 31)); // cur or ~cur
        const float4* p = sc.nodes + size_t(record) * 4;
        const float4 X = p[0], Y = p[1], Z = p[2], W = p[3];
        bool popNext = true;
        if (isNode) {
            float k0, k1;
            if (COUNT)
                tc.boxTests += 2;
            bool h0, h1;
            if (finiteInv) {
                nodeTestFinite(X, Y, Z, o, inv, t, h0, h1, k0, k1);
            } else {
                h0 = childTest(mk(X.x, Y.x, Z.x), mk(X.z, Y.z, Z.z), o, inv, t, k0);
                h1 = childTest(mk(X.y, Y.y, Z.y), mk(X.w, Y.w, Z.w), o, inv, t, k1);
            }
            const int32_t r0 = __float_as_int(W.x), r1 = __float_as_int(W.y);
            if (h0 && h1) {
                const bool firstIs1 = k1 < k0;
                st.push(sp++, firstIs1 ? r0 : r1, firstIs1 ? k0 : k1);
                cur = firstIs1 ? r1 : r0;
                popNext = false;
            } else if (h0 || h1) {
                cur = h0 ? r0 : r1;
                popNext = false;
            }
        } else {
            if (COUNT)
                tc.triTests++;
#ifdef RTX_PROBE
            tc.inner++;
            tc.waveInner += uint32_t(__builtin_ctzll(__ballot(1))) == (threadIdx.x & 63u);
#endif
            float th;
            if (triangleTestPacked(X, Y, Z, W.x, o, d, t, th)) {
                t = th;
                bestK = int(record) - sc.nNodes;
                if (EARLY && t < tStop)
                    return;
            }
            if (__float_as_int(W.y) >= 0) { // not the leaf's last triangle (top bit of the id word, set at upload)
                cur--;
                popNext = false;
            }
        }
        if (popNext) {
            have = false;
            while (sp > 0) {
                float key;
                st.pop(--sp, cur, key);
                if (!(t < key)) {
                    have = true;
                    break;
                }
            }
        }
    }
}

// ------------------------------------------------- wave-shared traversal ----
// The 64 rays of a wave walked through the tree TOGETHER: one node or one triangle
// at a time for the whole wave, its record fetched once through the scalar cache
// into scalar registers, one stack of (subtree, per-lane key) entries.  Every lane
// still decides for itself, with the reference's predicates evaluated on its own ray
// (childTestFinite's arithmetic, triangleTest's arithmetic), which subtrees it takes
// part in: a lane is active below a node only if it passed the visit test of every
// box on the way, and sits a subtree out otherwise.  So each lane tests exactly the
// boxes and triangles it could test walking alone; only the ORDER is the wave's
// (children by majority vote of the lanes that enter both), which matters only for
// exact-t ties between triangles, as for the depth-first order itself.
// What it buys on coherent rays (camera samples of neighbouring pixels, shadow rays
// of one light towards neighbouring points): no trip serves some lanes' boxes and
// other lanes' triangles at once, addressing / stack / loop control are scalar, a
// triangle whose plane no lane meets in range is dropped after a third of its test,
// and no vector registers or vector memory instructions are spent on the records.
//
// The rays of one call share the signs of their direction components (`oct`, bit a
// set: component a negative).  The walk then runs in the mirror image in which all
// three are positive: a node's record holds, per axis, the planes (lo, hi) and the
// mirrored planes (-hi, -lo), the origin is mirrored likewise, and |1/d| is used.
// Negation is exact, so (lo' - o') * |1/d| are the very numbers (lo - o) * (1/d) and
// (hi - o) * (1/d), already in the order min/max would put them: the per-axis
// ordering of the slab test costs nothing.
// (the compiler's own ballot on a condition: no detour through a 0 / 1 integer per lane)
RTX_DEV uint64_t laneMask(bool b) { return __builtin_amdgcn_ballot_w64(b); }
RTX_DEV bool anyLane(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0; }
// m's bit for this lane ? a : b, for a wave mask that scalar instructions made (the result of a
// vote combined with others: the compiler has no per-lane form of it, and rebuilding one costs
// more than this select).  gfx950 wants two wait states between a vector instruction that
// writes a scalar register and a vector instruction that reads it; the compiler keeps them for
// its own code but does not look inside an asm statement, so the statement brings its own (the
// masks here are written by scalar instructions as the code stands, which need none: the two
// idle cycles buy independence from how the compiler allocates them).
RTX_DEV float selectByMask(uint64_t m, float a, float b)
{
    float r;
    asm("s_nop 1\n\tv_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
RTX_DEV int selectByMask(uint64_t m, int a, int b) { return __float_as_int(selectByMask(m, __int_as_float(a), __int_as_float(b))); }
typedef float NF4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const NF4 UniformF4; // constant address space: uniform addresses load through the scalar cache
RTX_DEV const UniformF4* uniformRecords(const float4* p) { return (const UniformF4*)(uintptr_t)p; }
constexpr int kWaveNodeQuads = 8, kWaveTriQuads = 6, kFrameTriQuads = 9;
constexpr float kPretestMin = 1e-12f, kPretestMax = 1e12f; // the pretest is off for rays and triangles whose magnitudes fall outside (products could leave fp32's range)

// How a call of the wave-shared walk tests boxes:
//   BOX_MIRRORED  childTestFinite's arithmetic in the mirrored frame (finite 1/d, one octant)
//   BOX_OUTSIDE   the same for rays that start outside the root box, hence outside every box of
//                 a tree whose boxes are nested (checked at upload): "origin inside the box" is
//                 false throughout and is not evaluated
//   BOX_LITERAL   the comparison-by-comparison slab test on the unmirrored planes (octant 0's
//                 copy), for rays with an infinite reciprocal direction component (0 * inf =
//                 NaN must fall through the comparisons exactly as in the reference)
enum BoxMode { BOX_MIRRORED = 0, BOX_OUTSIDE = 1, BOX_LITERAL = 2 };

// The visit rule of one child: the key the reference would enter the child with (0 when the
// origin lies inside, else the slab test's entry distance) and the wave mask of the lanes that
// enter it -- those for which the slab test hits and !(limit < key), which is the reference's
// `hit && !(ray.t < first)`, or whose origin lies inside.  (The mask is put together from the
// separate comparisons' votes by scalar instructions; folding a miss into the key as +inf
// first cost a select per child.)  `partM`: the lanes taking part (their limit is finite).
// Mirrored frame: lo <= hi per axis and im > 0, so t0 = (lo - o) * im <= t1 = (hi - o) * im
// already (childTestFinite's numbers without the per-axis ordering).
// With ANYSIGN the lanes' directions need not share the frame's octant: a lane whose direction
// differs from it in some axis has a negative im there (the frame's mirroring applied to its own
// 1/d, an exact negation), and each axis' two products are put in order with min / max, which is
// childTestFinite's arithmetic on the very same numbers.
// Two children (the halves of F2: slots 0 and 1, or 2 and 3, of a four-wide node) at a time: the
// subtractions and products, which are the bulk of the test, as packed instructions.
// `inside` (REL only): bits 0 and 1 say that the frame's origin lies inside child a / b (made
// with the frame: every ray of the call starts there).
template <int MODE, bool ANYSIGN, bool REL>
RTX_DEV void childVisit2(F2 lox, F2 loy, F2 loz, F2 hix, F2 hiy, F2 hiz, V3 om, V3 im, float limit, uint64_t partM, uint32_t inside, float& keyA, uint64_t& enterA, float& keyB, uint64_t& enterB)
{
    if (MODE == BOX_LITERAL) {
        const bool ha = childTest(mk(lox.x, loy.x, loz.x), mk(hix.x, hiy.x, hiz.x), om, im, limit, keyA);
        const bool hb = childTest(mk(lox.y, loy.y, loz.y), mk(hix.y, hiy.y, hiz.y), om, im, limit, keyB);
        enterA = laneMask(ha) & partM; // (the literal test lets an origin inside the box in whatever the limit)
        enterB = laneMask(hb) & partM;
        return;
    }
    // (with a frame the planes come with the origin already subtracted)
    const F2 dlx = REL ? lox : lox - both(om.x), dly = REL ? loy : loy - both(om.y), dlz = REL ? loz : loz - both(om.z);
    const F2 dhx = REL ? hix : hix - both(om.x), dhy = REL ? hiy : hiy - both(om.y), dhz = REL ? hiz : hiz - both(om.z);
    F2 t0x = dlx * both(im.x), t0y = dly * both(im.y), t0z = dlz * both(im.z);
    F2 t1x = dhx * both(im.x), t1y = dhy * both(im.y), t1z = dhz * both(im.z);
    auto finish = [&](float ax, float ay, float az, float bx, float by, float bz, float lx, float ly, float lz, float hx, float hy, float hz, bool in, float& key, uint64_t& enterM) {
        if (ANYSIGN) {
            const float cx = fminf(ax, bx), cy = fminf(ay, by), cz = fminf(az, bz);
            bx = fmaxf(ax, bx), by = fmaxf(ay, by), bz = fmaxf(az, bz);
            ax = cx, ay = cy, az = cz;
        }
        const float first = fmaxf(fmaxf(ax, ay), fmaxf(az, 0.0f));
        const float second = fminf(fminf(bx, by), bz);
        const uint64_t missM = laneMask(first > second);
        key = first <= 0.0f ? second : first;
        enterM = laneMask(!(limit < key)) & ~missM;
        if (MODE != BOX_OUTSIDE) {
            if (REL) {
                if (in) { // (wave-uniform)
                    key = 0.0f;
                    enterM = partM;
                }
            } else {
                const uint64_t inM = laneMask(fmaxf(fmaxf(lx, ly), lz) <= 0.0f) & laneMask(fminf(fminf(hx, hy), hz) >= 0.0f);
                if (inM & partM) { // (rare below the first levels: most rays start outside most boxes)
                    key = selectByMask(inM, 0.0f, key);
                    enterM |= inM & partM;
                }
            }
        }
    };
    finish(t0x.x, t0y.x, t0z.x, t1x.x, t1y.x, t1z.x, dlx.x, dly.x, dlz.x, dhx.x, dhy.x, dhz.x, (inside & 1u) != 0, keyA, enterA);
    finish(t0x.y, t0y.y, t0z.y, t1x.y, t1y.y, t1z.y, dlx.y, dly.y, dlz.y, dhx.y, dhy.y, dhz.y, (inside & 2u) != 0, keyB, enterB);
}

// `alive`: the lanes of this call (passed the root box; for the mirrored modes: same octant,
// finite 1/d).  t and bestK as in traverseBvh; with EARLY a lane stops taking part once it
// knows a hit below tStop.
// REL: all rays of the call start at the origin `fr` was made for; its records are read instead
// (not for the literal test, nor for bundles that straddle octants).
template <int STACK, bool COUNT, bool EARLY, int MODE, bool ANYSIGN = false, bool REL = false>
RTX_DEV void traverseBvhWave(const DScene& sc, LaneStack<STACK>& st, V3 o, V3 d, V3 inv, bool alive, uint32_t oct, float& t, int& bestK, TraceCounters& tc, float tStop, OriginFrame fr = OriginFrame { nullptr, nullptr })
{
    static_assert(!REL || (MODE != BOX_LITERAL && !ANYSIGN), "origin frames serve the mirrored box tests of one octant");
    const bool nx = MODE != BOX_LITERAL && (oct & 1u) != 0, ny = MODE != BOX_LITERAL && (oct & 2u) != 0, nz = MODE != BOX_LITERAL && (oct & 4u) != 0; // wave-uniform
    // (with a frame the planes come with the origin already subtracted: x - 0 is x)
    const V3 om = REL ? mk(0.0f, 0.0f, 0.0f) : mk(nx ? -o.x : o.x, ny ? -o.y : o.y, nz ? -o.z : o.z);
    const V3 im = MODE == BOX_LITERAL ? inv : ANYSIGN ? mk(nx ? -inv.x : inv.x, ny ? -inv.y : inv.y, nz ? -inv.z : inv.z) : mk(fabsf(inv.x), fabsf(inv.y), fabsf(inv.z));
    const UniformF4* const nodes = uniformRecords(REL ? fr.nodes : sc.waveNodes) + size_t(MODE == BOX_LITERAL ? 0u : oct) * size_t(uint32_t(sc.nWaveNodes)) * kWaveNodeQuads;
    const UniformF4* const tris = uniformRecords(REL ? fr.tris : sc.waveTris);
    const float inf = __builtin_inff();
    // the pretest's threshold is M * -max|d_k| (no pretest for a direction of extreme magnitude)
    const float dmax = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
    const float negDmax = REL ? (dmax > kPretestMin && dmax < kPretestMax ? -dmax : -inf) : 0.0f;
    int sp = 0;
    int32_t cur = __builtin_amdgcn_readfirstlane(sc.waveRoot);
    // Which lanes take part is kept as wave masks (aliveM: still looking for a hit; actM: inside
    // the current subtree), combined from the votes with scalar instructions: a vote on a
    // condition put together per lane costs two more vector instructions each time.
    uint64_t aliveM = laneMask(alive), actM = aliveM;
    if (!actM)
        return;
    for (;;) {
        if (cur >= 0) {
#ifdef RTX_PROBE
            tc.waveOuter += (threadIdx.x & 63u) == 0u;
            tc.outer += (actM >> (threadIdx.x & 63u)) & 1u;
            if ((threadIdx.x & 63u) == 0u)
                atomicAdd(&g_waveWalkStats[3], 1ull);
#endif
            const UniformF4* p = nodes + size_t(uint32_t(cur)) * kWaveNodeQuads;
            const NF4 XL = p[0], XH = p[1], YL = p[2], YH = p[3], ZL = p[4], ZH = p[5], R = p[6];
            // (the children's references come with the boxes, not in a second round trip to memory
            // once a child turns out to be entered: the empty statement keeps the load up here)
            asm volatile("" ::"s"(R.x), "s"(R.y), "s"(R.z), "s"(R.w));
            const float limit = selectByMask(actM, t, -inf); // (a lane that sits this subtree out enters nothing)
            float k0, k1, k2, k3;
            uint64_t b0, b1, b2, b3; // the lanes that enter each child
            const uint32_t inside = REL && MODE != BOX_OUTSIDE ? uint32_t(__float_as_int(p[7].w)) : 0u; // (which children hold the frame's origin)
            childVisit2<MODE, ANYSIGN, REL>(F2 { XL.x, XL.y }, F2 { YL.x, YL.y }, F2 { ZL.x, ZL.y }, F2 { XH.x, XH.y }, F2 { YH.x, YH.y }, F2 { ZH.x, ZH.y }, om, im, limit, actM, inside, k0, b0, k1, b1);
            childVisit2<MODE, ANYSIGN, REL>(F2 { XL.z, XL.w }, F2 { YL.z, YL.w }, F2 { ZL.z, ZL.w }, F2 { XH.z, XH.w }, F2 { YH.z, YH.w }, F2 { ZH.z, ZH.w }, om, im, limit, actM, inside >> 2, k2, b2, k3, b3);
            if (MODE == BOX_LITERAL || ANYSIGN) {
                // (a test that orders each axis' two planes itself reads the inverted planes of
                // an unused slot as all of space: go by the child count)
                const int nChildren = __float_as_int(p[7].x);
                b2 = nChildren > 2 ? b2 : 0;
                b3 = nChildren > 3 ? b3 : 0;
            }
            if (COUNT)
                tc.boxTests += uint32_t((actM >> (threadIdx.x & 63u)) & 1u) * 4u;
            // The children are stored nearest first for this octant: the walk goes on in the
            // first one any lane enters and keeps the others, farthest first, for later.
            const int first = b0 ? 0 : b1 ? 1 : b2 ? 2 : b3 ? 3 : 4;
            if (first < 4) {
                const int32_t r0 = __float_as_int(R.x), r1 = __float_as_int(R.y), r2 = __float_as_int(R.z), r3 = __float_as_int(R.w);
                if (b3 && first < 3)
                    st.pushWave(sp++, r3, selectByMask(b3, k3, inf)); // (an infinite key: not this lane's subtree)
                if (b2 && first < 2)
                    st.pushWave(sp++, r2, selectByMask(b2, k2, inf));
                if (b1 && first < 1)
                    st.pushWave(sp++, r1, selectByMask(b1, k1, inf));
                cur = first == 0 ? r0 : first == 1 ? r1 : first == 2 ? r2 : r3;
                actM = first == 0 ? b0 : first == 1 ? b1 : first == 2 ? b2 : b3; // (the lanes that enter it)
                continue;
            }
        } else {
            // a leaf: its triangles one after the other, for the lanes that entered it
            // (candM: the lanes the triangle is tested for; last: the leaf ends with it)
            auto test = [&](const NF4& P, const NF4& A, const NF4& B, const NF4& E0, const NF4& E1, const NF4& C, uint32_t k, uint64_t candM, bool last) -> bool {
#ifdef RTX_PROBE
                tc.waveInner += (threadIdx.x & 63u) == 0u;
                tc.inner += (actM >> (threadIdx.x & 63u)) & 1u;
                if ((threadIdx.x & 63u) == 0u)
                    atomicAdd(&g_waveWalkStats[0], 1ull);
#endif
                // the plane part of triangleTest.  (Its `dn == 0` rejection needs no test of its own:
                // the quotient is then an infinity or a NaN, which fails one of the two comparisons.)
                const float dn = P.x * d.x + P.y * d.y + P.z * d.z;
                float num = P.w; // (with a frame: D - dot(n, origin) already)
                if (!REL) {
                    const float no = P.x * o.x + P.y * o.y + P.z * o.z;
                    num = P.w - no;
                }
                const float th = num / dn;
                const bool behind = th <= 0.f, nearer = th < t;
                const uint64_t okM = candM & ~laneMask(behind) & laneMask(nearer);
#ifdef RTX_PROBE
                if (okM && (threadIdx.x & 63u) == 0u)
                    atomicAdd(&g_waveWalkStats[1], 1ull);
#endif
                if (okM) {
                    const V3 n = mk(P.x, P.y, P.z);
                    const V3 pt = o + th * d;
                    const V3 e0 = pt - mk(A.x, A.y, A.z), e1 = pt - mk(B.x, B.y, B.z);
                    const float alpha = dot(n, cross(mk(E1.x, E1.y, E1.z), e1));
                    const float beta = dot(n, cross(e0, mk(E0.x, E0.y, E0.z)));
                    const float gamma = dot(n, cross(mk(C.x, C.y, C.z), e0));
                    const bool outside = alpha < 0 || beta < 0 || gamma < 0;
                    const uint64_t hitM = okM & ~laneMask(outside);
#ifdef RTX_PROBE
                    if (hitM && (threadIdx.x & 63u) == 0u)
                        atomicAdd(&g_waveWalkStats[2], 1ull);
#endif
                    if (hitM) {
                        t = selectByMask(hitM, th, t);
                        bestK = selectByMask(hitM, int(k), bestK);
                        if (EARLY) {
                            const uint64_t stopM = hitM & laneMask(th < tStop);
                            aliveM &= ~stopM;
                            actM &= ~stopM;
                        }
                    }
                }
                return last || !actM;
            };
            uint32_t k = uint32_t(~cur);
            if (REL) {
                // With a frame: the pretests of two triangles per round trip to memory, and a whole
                // record only for a triangle some lane's ray does not clearly pass.
                auto pretested = [&](const UniformF4* q, const NF4& G0, const NF4& G1, const NF4& G2, uint32_t kk) -> bool {
                    if (COUNT)
                        tc.triTests += uint32_t((actM >> (threadIdx.x & 63u)) & 1u);
                    // (not the reference's arithmetic: fused and packed operations are welcome)
                    const F2 e01 = __builtin_elementwise_fma(F2 { G0.x, G0.y }, both(d.x), __builtin_elementwise_fma(F2 { G0.z, G0.w }, both(d.y), F2 { G1.x, G1.y } * both(d.z)));
                    const float e2 = __builtin_fmaf(G1.z, d.x, __builtin_fmaf(G1.w, d.y, G2.x * d.z));
                    const bool clear = fminf(fminf(e01.x, e01.y), e2) < G2.y * negDmax; // passes the triangle by clearly
                    const uint64_t candM = actM & ~laneMask(clear);
                    const bool last = __float_as_int(G2.z) < 0;
#ifdef RTX_PROBE
                    if ((threadIdx.x & 63u) == 0u)
                        atomicAdd(&g_waveWalkStats[4], 1ull);
                    if (actM & ~candM) { // the whole test for the lanes the pretest drops: none may hit
                        const NF4 P = q[3], A = q[4], B = q[5], E0 = q[6], E1 = q[7], C = q[8];
                        const float th = P.w / (P.x * d.x + P.y * d.y + P.z * d.z);
                        const V3 n = mk(P.x, P.y, P.z);
                        const V3 pt = o + th * d;
                        const V3 f0 = pt - mk(A.x, A.y, A.z), f1 = pt - mk(B.x, B.y, B.z);
                        const float alpha = dot(n, cross(mk(E1.x, E1.y, E1.z), f1));
                        const float beta = dot(n, cross(f0, mk(E0.x, E0.y, E0.z)));
                        const float gamma = dot(n, cross(mk(C.x, C.y, C.z), f0));
                        const bool hits = !(th <= 0.f) && th < t && !(alpha < 0 || beta < 0 || gamma < 0);
                        const uint64_t wrongM = actM & ~candM & laneMask(hits);
                        if (wrongM && (threadIdx.x & 63u) == 0u)
                            atomicAdd(&g_waveWalkStats[5], (unsigned long long)__builtin_popcountll(wrongM));
                    }
#endif
                    if (!candM)
                        return last;
                    // (the whole record in one round trip: left alone, the loads sink behind the plane stage)
                    const NF4 P = q[3], A = q[4], B = q[5], E0 = q[6], E1 = q[7], C = q[8];
                    asm volatile("" ::"s"(P.x), "s"(A.x), "s"(B.x), "s"(E0.x), "s"(E1.x), "s"(C.x));
                    return test(P, A, B, E0, E1, C, kk, candM, last);
                };
                for (;;) {
                    const UniformF4* q = tris + size_t(k) * kFrameTriQuads;
                    const NF4 G00 = q[0], G01 = q[1], G02 = q[2];
                    const NF4 G10 = q[kFrameTriQuads + 0], G11 = q[kFrameTriQuads + 1], G12 = q[kFrameTriQuads + 2];
                    asm volatile("" ::"s"(G01.x), "s"(G02.x), "s"(G10.x), "s"(G11.x), "s"(G12.x));
                    if (pretested(q, G00, G01, G02, k) || pretested(q + kFrameTriQuads, G10, G11, G12, k + 1))
                        break;
                    k += 2;
                }
            } else {
                // Two whole records per round trip to memory (a leaf's triangles lie one behind the
                // other): the walk of a wave that has the SIMD to itself -- the long units a frame ends
                // on -- is a chain of such round trips, and fetching the edge vectors only after the
                // plane test, triangle by triangle, made it four times as long.  The record behind a
                // leaf's last one is another leaf's, or the array's padding, and is not looked at.
                for (;;) {
                    const UniformF4* q = tris + size_t(k) * kWaveTriQuads;
                    const NF4 P0 = q[0], A0 = q[1], B0 = q[2], E00 = q[3], E10 = q[4], C0 = q[5];
                    const NF4 P1 = q[6], A1 = q[7], B1 = q[8], E01 = q[9], E11 = q[10], C1 = q[11];
                    asm volatile("" ::"s"(B0.x), "s"(E00.x), "s"(E10.x), "s"(C0.x), "s"(P1.x), "s"(A1.x), "s"(B1.x), "s"(E01.x), "s"(E11.x), "s"(C1.x));
                    if (COUNT)
                        tc.triTests += uint32_t((actM >> (threadIdx.x & 63u)) & 1u);
                    if (test(P0, A0, B0, E00, E10, C0, k, actM, __float_as_int(A0.w) < 0))
                        break;
                    if (COUNT)
                        tc.triTests += uint32_t((actM >> (threadIdx.x & 63u)) & 1u);
                    if (test(P1, A1, B1, E01, E11, C1, k + 1, actM, __float_as_int(A1.w) < 0))
                        break;
                    k += 2;
                }
            }
        }
        bool have = false;
        while (sp > 0) {
            const float key = st.waveKey(--sp);
            const int32_t ref = st.waveRef(sp); // (read with the key, not after the vote)
            actM = aliveM & ~laneMask(t < key);
            if (actM) {
                cur = ref;
                have = true;
                break;
            }
        }
        if (!have)
            break;
    }
}

// Closest hits of the wave's rays (`lane`: this lane has one) by the wave-shared walk: one
// pass per direction octant present among them, and one for the rays whose reciprocal
// direction is not finite (axis-parallel rays).  t, bestK, tStop, rootLimit as in
// traverseBvh.  Must be called by all lanes of the wave together.
template <int STACK, bool COUNT, bool EARLY, bool SPREAD = false, bool REL = false>
RTX_DEV void traverseWaveTogether(const DScene& sc, LaneStack<STACK>& st, bool lane, V3 o, V3 d, float& t, int& bestK, TraceCounters& tc, float tStop = 0.f, float rootLimit = 0.f,
    OriginFrame fr = OriginFrame { nullptr, nullptr })
{
    bestK = -1;
    if (sc.rootRef == kEmptyRoot)
        return;
    const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const bool finiteInv = fabsf(inv.x) < __builtin_inff() && fabsf(inv.y) < __builtin_inff() && fabsf(inv.z) < __builtin_inff();
    bool in = lane;
    if (in) {
        // the root box, as in traverseBvh
        float key;
        if (COUNT)
            tc.boxTests++;
        in = slab(mk(sc.rootLo), mk(sc.rootHi), o, inv, EARLY ? rootLimit : t, key);
    }
    // a ray that starts outside the root box starts outside every box of a nested tree
    const bool outside = sc.boxesNested && !inBox(mk(sc.rootLo), mk(sc.rootHi), o);
    const uint32_t myOct = (inv.x < 0.f ? 1u : 0u) | (inv.y < 0.f ? 2u : 0u) | (inv.z < 0.f ? 4u : 0u);
    uint64_t rest = laneMask(in && finiteInv);
    if (SPREAD && rest) {
        // Rays that converge on a point or leave one in a narrow bundle (the samples of an area
        // light) can straddle a sign change of a direction component while still running side
        // by side: they are walked in ONE pass, in the frame most of them share per axis.
        const uint32_t total = uint32_t(__popcll(rest));
        const uint32_t oct = (2u * uint32_t(__popcll(laneMask(in && finiteInv && inv.x < 0.f))) > total ? 1u : 0u)
            | (2u * uint32_t(__popcll(laneMask(in && finiteInv && inv.y < 0.f))) > total ? 2u : 0u)
            | (2u * uint32_t(__popcll(laneMask(in && finiteInv && inv.z < 0.f))) > total ? 4u : 0u);
        if (laneMask(in && finiteInv && myOct != oct)) {
            if (!anyLane(in && finiteInv && !outside))
                traverseBvhWave<STACK, COUNT, EARLY, BOX_OUTSIDE, true>(sc, st, o, d, inv, in && finiteInv, oct, t, bestK, tc, tStop);
            else
                traverseBvhWave<STACK, COUNT, EARLY, BOX_MIRRORED, true>(sc, st, o, d, inv, in && finiteInv, oct, t, bestK, tc, tStop);
            rest = 0;
        }
    }
    while (rest) {
        const uint32_t oct = uint32_t(__builtin_amdgcn_readlane(int(myOct), __builtin_ctzll(rest)));
        const bool mine = in && finiteInv && myOct == oct;
        // (REL: the caller knows that all its rays start at the frame's origin)
        if (!anyLane(mine && !outside))
            traverseBvhWave<STACK, COUNT, EARLY, BOX_OUTSIDE, false, REL>(sc, st, o, d, inv, mine, oct, t, bestK, tc, tStop, fr);
        else
            traverseBvhWave<STACK, COUNT, EARLY, BOX_MIRRORED, false, REL>(sc, st, o, d, inv, mine, oct, t, bestK, tc, tStop, fr);
        rest &= ~laneMask(mine);
    }
    if (anyLane(in && !finiteInv))
        traverseBvhWave<STACK, COUNT, EARLY, BOX_LITERAL>(sc, st, o, d, inv, in && !finiteInv, 0u, t, bestK, tc, tStop);
}

// ------------------------------------------------------ frontier traversal --
// Closest hits of a FEW rays (the five rays of a glossy fan) with the WHOLE wave at work on them.
// Walked one node at a time, such a bundle is a chain of dependent fetches with most of the wave's
// lanes idle, and the walk of a deep glossy sample is hundreds of such bundles one after the other
// (their order is forced: the draws are numbered along the walk) -- the frame waits for that chain.
// Here every ray gets a group of G = 64 / rays lanes and a stack of its own in LDS, kept nearest
// on top as in any depth-first walk; but each round the group pops the TOP G entries at once, a
// lane each: a node entry tests the ray against the node's four child boxes (the reference's visit
// rule on that ray, in the ray's own octant copy, whose children are stored nearest first) and
// pushes the children entered, a leaf as a run of triangles; a triangle entry tests two triangles,
// lowers the ray's best hit with an LDS atomic and hands the rest of its run on in up to three
// parts.  Entries whose key lies behind the best hit known when they are popped are dropped, as in
// the other walks.  The fetches of a round are in flight together, so a ray's walk takes about as
// many round trips to memory as its path to the hit is deep: what a depth-first walk spends
// backing out of subtrees happens beside it.  Every ray still tests only boxes and triangles it
// could test walking alone; what differs is the ORDER, hence, as for the other walks, only which
// of two triangles met at exactly the same distance is reported (the one earlier in leaf order
// here).  The stacks live in the wave's columns of the traversal stacks' LDS, which no other walk
// uses meanwhile.  Returns false, with nothing decided, when a stack would outgrow its share: the
// caller then walks the rays the old way.
constexpr uint32_t kFrontierWords = 2u * uint32_t(kLdsStack) * 64u;   // LDS words of one wave's stacks
constexpr uint32_t kItemTri = 0x80000000u, kItemIndexMask = (1u << 25) - 1u; // triangle entry: kind | (run - 1) << 25 | first triangle
constexpr uint32_t kFrontierMaxIndex = kItemIndexMask, kFrontierMaxLeaf = 64u;
// The visit rule of one child for one ray in the ray's mirrored frame (childVisit's arithmetic, per lane)
RTX_DEV bool childTestMirrored(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 om, V3 im, float limit, float& key)
{
    const float dlx = lox - om.x, dly = loy - om.y, dlz = loz - om.z;
    const float dhx = hix - om.x, dhy = hiy - om.y, dhz = hiz - om.z;
    const float t0x = dlx * im.x, t0y = dly * im.y, t0z = dlz * im.z;
    const float t1x = dhx * im.x, t1y = dhy * im.y, t1z = dhz * im.z;
    const float first = fmaxf(fmaxf(t0x, t0y), fmaxf(t0z, 0.0f));
    const float second = fminf(fminf(t1x, t1y), t1z);
    const bool miss = first > second;
    key = first <= 0.0f ? second : first;
    const bool in = fmaxf(fmaxf(dlx, dly), dlz) <= 0.0f && fminf(fminf(dhx, dhy), dhz) >= 0.0f;
    key = in ? 0.0f : key;
    return in || (!miss && !(limit < key));
}
// Rays in lanes 0 .. nRays - 1 (`lane`: the ray is to be traced; nRays wave-uniform, 1 .. 64).
// `best`: 64 words of LDS of this wave.  Must be called by all lanes of the wave together; t,
// bestK as in traverseBvh.
template <int STACK, bool COUNT>
RTX_DEV bool traverseFrontier(const DScene& sc, LaneStack<STACK>& st, unsigned long long* best, uint32_t nRays, bool lane, V3 o, V3 d, float& t, int& bestK, TraceCounters& tc)
{
    static_assert(LaneStack<STACK>::kLds == kLdsStack, "the stacks take the traversal stacks' 2 x kLdsStack rows");
    bestK = -1;
    const uint32_t me = threadIdx.x & 63u;
    int32_t* const q = st.refs + (st.tid & ~63);
    auto word = [](uint32_t w) { return (w >> 6) * uint32_t(kBlock) + (w & 63u); }; // the wave's w-th word
    const uint32_t G = 64u / nRays, cap = (kFrontierWords / nRays) >> 1;              // lanes and stack entries per ray
    const uint32_t g = min(me / G, nRays - 1u), j = me - g * G;                        // this lane's ray and place in its group
    const bool grouped = me < nRays * G;
    // the ray's own part: root box, octant, mirrored frame
    const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const bool finiteInv = fabsf(inv.x) < __builtin_inff() && fabsf(inv.y) < __builtin_inff() && fabsf(inv.z) < __builtin_inff();
    bool in = lane && me < nRays;
    if (in) {
        float key; // the root box, as in traverseBvh
        if (COUNT)
            tc.boxTests++;
        in = slab(mk(sc.rootLo), mk(sc.rootHi), o, inv, t, key);
    }
    // (an axis-parallel ray takes the comparison-by-comparison test on the boxes as they are: octant 0's copy)
    const uint32_t myOct = finiteInv ? ((inv.x < 0.f ? 1u : 0u) | (inv.y < 0.f ? 2u : 0u) | (inv.z < 0.f ? 4u : 0u)) : 0u;
    const V3 myOm = finiteInv ? mk((myOct & 1u) ? -o.x : o.x, (myOct & 2u) ? -o.y : o.y, (myOct & 4u) ? -o.z : o.z) : o;
    const V3 myIm = finiteInv ? mk(fabsf(inv.x), fabsf(inv.y), fabsf(inv.z)) : inv;
    best[me] = (static_cast<unsigned long long>(__float_as_uint(t)) << 32) | 0xffffffffull;
    // the group's copy of its ray
    const int src = int(g);
    const V3 ro = mk(__shfl(o.x, src, 64), __shfl(o.y, src, 64), __shfl(o.z, src, 64));
    const V3 rd = mk(__shfl(d.x, src, 64), __shfl(d.y, src, 64), __shfl(d.z, src, 64));
    const V3 om = mk(__shfl(myOm.x, src, 64), __shfl(myOm.y, src, 64), __shfl(myOm.z, src, 64));
    const V3 im = mk(__shfl(myIm.x, src, 64), __shfl(myIm.y, src, 64), __shfl(myIm.z, src, 64));
    const bool rFinite = __shfl(int(finiteInv), src, 64) != 0;
    const float4* const nodes = sc.waveNodes + size_t(uint32_t(__shfl(int(myOct), src, 64))) * size_t(uint32_t(sc.nWaveNodes)) * kWaveNodeQuads;
    uint32_t sp = __shfl(int(in), src, 64) != 0 ? 1u : 0u; // entries on this group's stack (the same in all its lanes)
    const uint32_t stackAt = g * 2u * cap;                   // its first word
    if (grouped && j == 0u && sp) {
        q[word(stackAt)] = 0;                                // key 0
        q[word(stackAt + 1u)] = sc.waveRoot;
    }
    bool ok = true;
    while (anyLane(grouped && sp != 0u)) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t popped = min(G, sp);
        const bool have = grouped && j < popped;
        const uint32_t at = stackAt + 2u * (sp - popped + j);  // (lane 0 takes the farthest of the popped entries, the last lane the top)
        const float entryKey = have ? __int_as_float(q[word(at)]) : 0.f;
        const uint32_t entry = have ? uint32_t(q[word(at + 1u)]) : 0u;
        const float limit = __uint_as_float(uint32_t(best[g] >> 32));
        const bool live = have && !(limit < entryKey);
        const bool isTri = (entry & kItemTri) != 0u;
#ifdef RTX_PROBE
        tc.frontierRounds++;
        tc.frontierEntries += live ? 1u : 0u;
#endif
        // up to four entries to push, farthest first
        float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f;
        uint32_t e0 = 0u, e1 = 0u, e2 = 0u, e3 = 0u, n0 = 0u, n1 = 0u, n2 = 0u, n3 = 0u;
        if (live && !isTri) {
            const float4* p = nodes + size_t(entry) * kWaveNodeQuads;
            const float4 XL = p[0], XH = p[1], YL = p[2], YH = p[3], ZL = p[4], ZH = p[5], R = p[6], M = p[7];
            const int nChildren = __float_as_int(M.x);
            const uint32_t c01 = __float_as_uint(M.y), c23 = __float_as_uint(M.z);
            if (COUNT)
                tc.boxTests += uint32_t(nChildren);
            auto child = [&](int c, float lx, float ly, float lz, float hx, float hy, float hz, float r, uint32_t tris, float& key, uint32_t& e, uint32_t& n) {
                const bool enter = rFinite ? childTestMirrored(lx, ly, lz, hx, hy, hz, om, im, limit, key)
                                           : (c < nChildren && childTest(mk(lx, ly, lz), mk(hx, hy, hz), om, im, limit, key));
                const int32_t ref = __float_as_int(r);
                e = ref >= 0 ? uint32_t(ref) : (kItemTri | ((tris - 1u) << 25) | uint32_t(~ref));
                n = enter ? 1u : 0u;
            };
            // (the children are stored nearest first: the nearest goes on last)
            child(3, XL.w, YL.w, ZL.w, XH.w, YH.w, ZH.w, R.w, c23 >> 16, k0, e0, n0);
            child(2, XL.z, YL.z, ZL.z, XH.z, YH.z, ZH.z, R.z, c23 & 0xffffu, k1, e1, n1);
            child(1, XL.y, YL.y, ZL.y, XH.y, YH.y, ZH.y, R.y, c01 >> 16, k2, e2, n2);
            child(0, XL.x, YL.x, ZL.x, XH.x, YH.x, ZH.x, R.x, c01 & 0xffffu, k3, e3, n3);
        } else if (live) {
            const uint32_t first = entry & kItemIndexMask, run = ((entry >> 25) & 63u) + 1u;
            const float4* p = sc.tris + size_t(first) * 4;
            const float4 X0 = p[0], Y0 = p[1], Z0 = p[2], W0 = p[3];
            const bool two = run >= 2u;
            const float4* p2 = p + (two ? 4 : 0);
            const float4 X1 = p2[0], Y1 = p2[1], Z1 = p2[2], W1 = p2[3];
            if (COUNT)
                tc.triTests += two ? 2u : 1u;
            float th;
            float lim = limit;
            if (triangleTestPacked(X0, Y0, Z0, W0.x, ro, rd, lim, th)) {
                atomicMin(best + g, (static_cast<unsigned long long>(__float_as_uint(th)) << 32) | first);
                lim = th;
            }
            if (two && triangleTestPacked(X1, Y1, Z1, W1.x, ro, rd, lim, th))
                atomicMin(best + g, (static_cast<unsigned long long>(__float_as_uint(th)) << 32) | (first + 1u));
            // the rest of the run, in up to three parts for as many lanes of the next round
            if (run > 2u) {
                const uint32_t rest = run - 2u, a = (rest + 2u) / 3u, b = (rest - a + 1u) / 2u, c = rest - a - b;
                uint32_t from = first + 2u;
                e0 = kItemTri | ((a - 1u) << 25) | from, n0 = 1u, k0 = entryKey;
                from += a;
                if (b) {
                    e1 = kItemTri | ((b - 1u) << 25) | from, n1 = 1u, k1 = entryKey;
                    from += b;
                }
                if (c)
                    e2 = kItemTri | ((c - 1u) << 25) | from, n2 = 1u, k2 = entryKey;
            }
        }
        // where this lane's entries go: behind those of the group's lanes before it (votes and population
        // counts: no trip through the lanes' exchange network)
        const uint64_t v0 = laneMask(n0 != 0u), v1 = laneMask(n1 != 0u), v2 = laneMask(n2 != 0u), v3 = laneMask(n3 != 0u);
        const uint32_t mine = (n0 + n1) + (n2 + n3);
        const uint64_t groupMask = (G >= 64u ? ~0ull : ((1ull << G) - 1ull)) << (g * G);
        const uint64_t beforeMe = groupMask & ((1ull << me) - 1ull);
        const uint32_t beforeInGroup = uint32_t(__popcll(v0 & beforeMe) + __popcll(v1 & beforeMe)) + uint32_t(__popcll(v2 & beforeMe) + __popcll(v3 & beforeMe));
        const uint32_t groupTotal = uint32_t(__popcll(v0 & groupMask) + __popcll(v1 & groupMask)) + uint32_t(__popcll(v2 & groupMask) + __popcll(v3 & groupMask));
        const uint32_t newSp = sp - popped + groupTotal;
        if (anyLane(grouped && newSp > cap)) {
            ok = false;
            break;
        }
        uint32_t to = stackAt + 2u * (sp - popped + beforeInGroup);
        (void)mine;
        if (n0) {
            q[word(to)] = __float_as_int(k0);
            q[word(to + 1u)] = int32_t(e0);
            to += 2u;
        }
        if (n1) {
            q[word(to)] = __float_as_int(k1);
            q[word(to + 1u)] = int32_t(e1);
            to += 2u;
        }
        if (n2) {
            q[word(to)] = __float_as_int(k2);
            q[word(to + 1u)] = int32_t(e2);
            to += 2u;
        }
        if (n3) {
            q[word(to)] = __float_as_int(k3);
            q[word(to + 1u)] = int32_t(e3);
        }
        sp = grouped ? newSp : 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const unsigned long long found = best[me];
    if (ok && in && uint32_t(found) != 0xffffffffu) {
        t = __uint_as_float(uint32_t(found >> 32));
        bestK = int(uint32_t(found));
    }
    return ok;
}

// Brute-force path (features.enableAccelStructure == false),
// src/bounding_volume_hierarchy.cpp:330-360: every triangle in scene order,
// with the extra `t > 1e-6` acceptance rule.
template <bool COUNT>
RTX_DEV void traverseAll(const DScene& sc, V3 o, V3 d, float& t, int& bestK, TraceCounters& tc)
{
    bestK = -1;
    for (int i = 0; i < sc.nTris; i++) {
        const int k = sc.idOrder[i];
        const TriRec tr = loadTri(sc, k);
        if (COUNT)
            tc.triTests++;
        float th;
        if (triangleTest(tr.v0, tr.v1, tr.v2, tr.n, tr.d, o, d, t, th) && double(th) > 1e-6) {
            t = th;
            bestK = k;
        }
    }
}

// HitInfo as the hot path uses it
struct Hit {
    V3 normal;
    float u, v;
    int32_t k;         // leaf-order record of the triangle, -1 = sphere
    int32_t material;
    int32_t id;        // meshTrianglePairs index, or -2 - sphere
    V3 bary;
};

// Closest hit of one BvhInterface::intersect: spheres first
// (src/bounding_volume_hierarchy.cpp:327-328), then the triangles.  t is the
// search limit on entry and the hit distance on return; bestK is the leaf-order
// record of the winning triangle (-1 if none), sphere the index of the sphere
// that holds the hit when no triangle beat it (-1 if none).
// With EARLY, `clip` shortens the triangle search (never the sphere tests, whose
// acceptance rule compares the limit with the FAR root, src/intersect.cpp:107-119)
// and tStop ends it at the first hit below it; both are for shadow queries.
template <int STACK, bool COUNT, bool EARLY = false>
RTX_DEV bool closestHit(const DScene& sc, LaneStack<STACK>& st, const rtx_features& f, V3 o, V3 d, float& t, int& bestK, int& sphere, V3& sphereN, TraceCounters& tc, float tStop = 0.f, float clip = 0.f)
{
    sphere = -1;
    sphereN = mk(0, 0, 0);
    for (int i = 0; i < sc.nSpheres; i++) {
        V3 n;
        if (sphereTest(sc.spheres[i], o, d, t, n)) {
            sphereN = n;
            sphere = i;
        }
    }
    const float afterSpheres = t;
    if (EARLY)
        t = (clip < t) ? clip : t;
    if (f.enable_accel_structure)
        traverseBvh<STACK, COUNT, EARLY>(sc, st, o, d, t, bestK, tc, tStop, afterSpheres);
    else
        traverseAll<COUNT>(sc, o, d, t, bestK, tc);
    return bestK >= 0 || sphere >= 0;
}

// testVisibilityLightSample, src/light.cpp:70-102: 1 if the light sample at `pos`
// sees the shading point `hp`, else 0.  The reference shoots a ray from the
// sample towards the point (direction hp - pos, so the point sits at t ~ 1),
// takes its closest hit over all distances and calls the sample visible iff
// that hit lies within 1e-3 of the point; a ray that hits nothing counts as
// blocked; a sample behind the (viewer-side) surface is blocked whatever the ray
// finds.  Three shortcuts below give the same answer with less traversal:
//  - the facing test needs no ray at all;
//  - when the sample is at least 0.2 away, nothing at t >= 1.01 can be within
//    1e-3 of the point (0.01 * 0.2 = 2e-3), and "no hit below 1.01" is blocked
//    either way, so the search is limited to t < 1.01;
//  - any hit at t < 1 - 4e-3/|hp - pos| is farther than 1e-3 from the point and so
//    is every hit before it, so the walk may stop there.
// The margins (2x and 4x) dwarf fp32 rounding for coordinates up to 64; beyond
// that the query runs unshortened.
// (the facing test: false for a light sample behind the viewer-side surface)
RTX_DEV bool lightFacesSurface(V3 pos, V3 hp, V3 visN)
{
    return !(dot(normalize(pos - hp), normalize(visN)) < -1e-6f);
}
// (the ray part; the caller has made the facing test)
template <int STACK, bool COUNT>
RTX_DEV float lightVisibility(const DScene& sc, LaneStack<STACK>& st, const rtx_features& f, V3 pos, V3 hp, V3 visN, TraceCounters& tc)
{
    (void)visN;
    const V3 sd = hp - pos;
    const float len = length(sd);
    const float scale = fmaxf(fmaxf(fmaxf(fabsf(hp.x), fabsf(hp.y)), fabsf(hp.z)), fmaxf(fmaxf(fabsf(pos.x), fabsf(pos.y)), fabsf(pos.z)));
    float ts = kFltMax, clip = kFltMax, tStop = -__builtin_inff();
    if (len >= 0.2f && scale <= 64.0f) {
        clip = 1.01f;
        tStop = 1.0f - 4e-3f / len;
    }
    int bk, sph;
    V3 sn;
    closestHit<STACK, COUNT, true>(sc, st, f, pos, sd, ts, bk, sph, sn, tc, tStop, clip);
    const V3 second = pos + ts * sd;
    return (double(distance(hp, second)) > 1e-3) ? 0.0f : 1.0f;
}

// HitInfo of the winner (src/bounding_volume_hierarchy.cpp:277-295,
// src/interpolate.cpp:4-23; spheres: src/intersect.cpp:120-123)
RTX_DEV void hitAttributes(const DScene& sc, const rtx_features& f, V3 o, V3 d, float t, int bestK, int sphere, V3 sphereN, Hit& hit)
{
    hit.k = -1;
    hit.material = -1;
    hit.id = RTX_TRI_NONE;
    hit.normal = mk(0, 0, 0);
    hit.bary = mk(0, 0, 0);
    hit.u = hit.v = 0.f;
    if (bestK >= 0) {
        const TriRec tr = loadTri(sc, bestK);
        const float4* ap = sc.attrs + size_t(bestK) * 4;
        const float4 a = ap[0], b = ap[1], c = ap[2], e = ap[3];
        const V3 n0 = mk(a.x, a.y, a.z), n1 = mk(a.w, b.x, b.y), n2 = mk(b.z, b.w, c.x);
        const V3 p = o + t * d;
        // src/interpolate.cpp:4-12: area ratios from lengths
        const float tArea = length(cross(tr.v1 - tr.v2, tr.v2 - tr.v0));
        const float ba = length(cross(tr.v1 - p, tr.v2 - p)) / tArea;
        const float bb = length(cross(tr.v0 - p, tr.v2 - p)) / tArea;
        const float bg = length(cross(tr.v1 - p, tr.v0 - p)) / tArea;
        hit.bary = mk(ba, bb, bg);
        hit.normal = f.enable_normal_interp ? (ba * n0 + bb * n1 + bg * n2) : n0;
        hit.u = ba * c.y + bb * c.w + bg * e.y;
        hit.v = ba * c.z + bb * e.x + bg * e.z;
        hit.k = bestK;
        hit.material = tr.material;
        hit.id = tr.id;
    } else if (sphere >= 0) {
        hit.normal = sphereN;
        hit.material = sc.spheres[sphere].material;
        hit.id = -2 - sphere;
    }
}

template <int STACK, bool COUNT>
RTX_DEV bool intersectScene(const DScene& sc, LaneStack<STACK>& st, const rtx_features& f, V3 o, V3 d, float& t, Hit& hit, TraceCounters& tc)
{
    int bestK, sphere;
    V3 sphereN;
    const bool any = closestHit<STACK, COUNT>(sc, st, f, o, d, t, bestK, sphere, sphereN, tc);
    hitAttributes(sc, f, o, d, t, bestK, sphere, sphereN, hit);
    return any;
}

// closestHit / lightVisibility / intersectScene for a whole wave at once (`lane`: this lane
// has a query), with the triangles found by the wave-shared walk.  Same results per lane.
template <int STACK, bool COUNT, bool EARLY = false, bool SPREAD = false, bool REL = false>
RTX_DEV bool closestHitWave(const DScene& sc, LaneStack<STACK>& st, const rtx_features& f, bool lane, V3 o, V3 d, float& t, int& bestK, int& sphere, V3& sphereN, TraceCounters& tc, float tStop = 0.f, float clip = 0.f,
    OriginFrame fr = OriginFrame { nullptr, nullptr })
{
    sphere = -1;
    sphereN = mk(0, 0, 0);
    bestK = -1;
    if (lane)
        for (int i = 0; i < sc.nSpheres; i++) {
            V3 n;
            if (sphereTest(sc.spheres[i], o, d, t, n)) {
                sphereN = n;
                sphere = i;
            }
        }
    const float afterSpheres = t;
    if (EARLY)
        t = (clip < t) ? clip : t;
    if (f.enable_accel_structure)
        traverseWaveTogether<STACK, COUNT, EARLY, SPREAD, REL && !SPREAD>(sc, st, lane, o, d, t, bestK, tc, tStop, afterSpheres, fr);
    else if (lane)
        traverseAll<COUNT>(sc, o, d, t, bestK, tc);
    return lane && (bestK >= 0 || sphere >= 0);
}
template <int STACK, bool COUNT, bool SPREAD = false, bool REL = false>
RTX_DEV float lightVisibilityWave(const DScene& sc, LaneStack<STACK>& st, const rtx_features& f, bool lane, V3 pos, V3 hp, TraceCounters& tc, OriginFrame fr = OriginFrame { nullptr, nullptr })
{
    const V3 sd = hp - pos;
    const float len = length(sd);
    const float scale = fmaxf(fmaxf(fmaxf(fabsf(hp.x), fabsf(hp.y)), fabsf(hp.z)), fmaxf(fmaxf(fabsf(pos.x), fabsf(pos.y)), fabsf(pos.z)));
    float ts = kFltMax, clip = kFltMax, tStop = -__builtin_inff();
    if (len >= 0.2f && scale <= 64.0f) {
        clip = 1.01f;
        tStop = 1.0f - 4e-3f / len;
    }
    int bk, sph;
    V3 sn;
    closestHitWave<STACK, COUNT, true, SPREAD, REL>(sc, st, f, lane, pos, sd, ts, bk, sph, sn, tc, tStop, clip, fr);
    const V3 second = pos + ts * sd;
    return (double(distance(hp, second)) > 1e-3) ? 0.0f : 1.0f;
}
template <int STACK, bool COUNT, bool REL = false>
RTX_DEV bool intersectSceneWave(const DScene& sc, LaneStack<STACK>& st, const rtx_features& f, bool lane, V3 o, V3 d, float& t, Hit& hit, TraceCounters& tc, OriginFrame fr = OriginFrame { nullptr, nullptr })
{
    int bestK, sphere;
    V3 sphereN;
    const bool any = closestHitWave<STACK, COUNT, false, false, REL>(sc, st, f, lane, o, d, t, bestK, sphere, sphereN, tc, 0.f, 0.f, fr);
    hitAttributes(sc, f, o, d, t, bestK, sphere, sphereN, hit);
    return any;
}

// -------------------------------------------------------------- textures ----
RTX_DEV V3 texelAt(const DScene& sc, const rtx_texture& tx, int level, uint32_t idx)
{
    // the reference reads past the end of a level for odd sizes (undefined
    // there); such reads return 0 here
    if (idx >= tx.count[level])
        return mk(0, 0, 0);
    const float* p = sc.texels + (size_t(tx.offset[level]) + idx) * 3;
    return mk(p[0], p[1], p[2]);
}
// src/texture.cpp:6-16
RTX_DEV V3 texelNearest(const DScene& sc, const rtx_texture& tx, float u, float v)
{
    const int W = tx.width[0], H = tx.height[0];
    int i = int(float(W) * u);
    int j = int(float(H) * (1 - v));
    i = max(0, min(W - 1, i));
    j = max(0, min(H - 1, j));
    return texelAt(sc, tx, 0, uint32_t(j * W + i));
}
RTX_DEV float modGlm(float a, float b) { return a - b * __builtin_floorf(a / b); }
// src/texture.cpp:18-33, src/shading.cpp:174-190
RTX_DEV V3 texelBilinear(const DScene& sc, const rtx_texture& tx, int level, float tu, float tv)
{
    const int W = tx.width[level], H = tx.height[level];
    const float txp = float(W - 1) * tu, typ = float(H - 1) * (1 - tv);
    float lx = __builtin_floorf(float(W - 1) * tu), ly = __builtin_floorf(float(H - 1) * (1 - tv));
    float ux = lx + 1, uy = ly + 1;
    const float u = txp - lx, v = typ - ly;
    lx = modGlm(lx, float(W));
    ly = modGlm(ly, float(H));
    ux = modGlm(ux, float(W));
    uy = modGlm(uy, float(H));
    // float -> size_t conversions of the reference; negative / NaN indices are
    // undefined there and clamp to "out of range" here
    const V3 lowerLeft = texelAt(sc, tx, level, uint32_t(ly * float(W) + lx));
    const V3 upperRight = texelAt(sc, tx, level, uint32_t(uy * float(W) + ux));
    const V3 lowerRight = texelAt(sc, tx, level, uint32_t(uy * float(W) + lx));
    const V3 upperLeft = texelAt(sc, tx, level, uint32_t(ly * float(W) + ux));
    return lowerLeft * (1.0f - u) * (1.0f - v) + upperRight * u * v + lowerRight * (1.0f - u) * v + upperLeft * u * (1.0f - v);
}
// src/shading.cpp:107-114
RTX_DEV V3 planeCoord(V3 ro, V3 rd, V3 point)
{
    const V3 w = normalize(rd);
    const V3 t = normalize(w - mk(0.1f, 0.0f, 0.0f));
    const V3 x = normalize(cross(t, w));
    const V3 y = normalize(cross(w, x));
    const V3 rToP = point - ro;
    return ro + dot(rToP, x) * x + dot(rToP, y) * y;
}
// src/shading.cpp:117-172: level of detail from the ratio of the triangle's
// area to its area projected on the plane perpendicular to the ray
RTX_DEV V3 texelTrilinear(const DScene& sc, const rtx_texture& tx, V3 ro, V3 rd, const Hit& hit)
{
    if (hit.k < 0)
        return texelBilinear(sc, tx, 0, hit.u, hit.v);
    const TriRec tr = loadTri(sc, hit.k);
    const V3 p0 = planeCoord(ro, rd, tr.v0), p1 = planeCoord(ro, rd, tr.v1), p2 = planeCoord(ro, rd, tr.v2);
    const float areaV = length(cross(tr.v0 - tr.v1, tr.v0 - tr.v2));
    const float areaP = length(cross(p0 - p1, p0 - p2));
    const int levels = tx.levels;
    float k = float(levels);
    if (double(areaP) >= 1e-6)
        k = (areaV / areaP - 1.0f) / 2.8f;
    const float k0 = __builtin_floorf(k);
    const float k1 = k0 + 1;
    const float a = k1 - k;
    if (k0 < 0)
        return texelBilinear(sc, tx, 0, hit.u, hit.v);
    if (k1 >= float(levels))
        return texelBilinear(sc, tx, levels - 1, hit.u, hit.v);
    const V3 c0 = texelBilinear(sc, tx, int(k0), hit.u, hit.v);
    const V3 c1 = texelBilinear(sc, tx, int(k1), hit.u, hit.v);
    return a * c0 + (1 - a) * c1;
}

// --------------------------------------- the host library's trigonometry ----
// sinf / cosf with the host C library's results.  glm::quat(euler) calls std::sin / std::cos on
// floats, which on the hosts this runs beside is glibc's sinf / cosf (2.35,
// sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, sincosf_data.c: the algorithm of Arm's
// optimized routines): the argument is reduced and a degree-7 / degree-8 polynomial evaluated in
// double, the result rounded to float once.  Its error is below 0.6 ulp but it is not the
// correctly rounded value, and the device library's sinf / cosf round differently in about one
// argument in ten: a motion-blur sample's camera then differs in its last bit and its ray
// crosses a silhouette on the other side.  Restated here with the same coefficients and, as in
// the variant glibc runs on a CPU with fused multiply-add, one rounding per multiply-add:
// 749 million arguments below 120 in magnitude (every third float) give the host's bits without
// exception.  Arguments of 120 and beyond (nineteen turns of the camera) take the device
// library's routine.
struct HostTrig {
    static constexpr double hpiInv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    static constexpr double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    static constexpr double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    static __device__ __forceinline__ uint32_t top12(float x) { return (__float_as_uint(x) >> 20) & 0x7ffu; }
    // sine (n even) or cosine (n odd) of the reduced argument; `flip`: the cosine's sign in the
    // second half of the turn (glibc keeps a second table with the cosine's coefficients negated)
    static __device__ __forceinline__ float poly(double x, double x2, int n, bool flip)
    {
        if ((n & 1) == 0) {
            const double x3 = x * x2;
            const double t = __builtin_fma(x2, s3, s2);
            const double x7 = x3 * x2;
            const double s = __builtin_fma(x3, s1, x);
            return float(__builtin_fma(x7, t, s));
        }
        const double sg = flip ? -1.0 : 1.0; // (an exact negation of every coefficient, hence of every intermediate)
        const double x4 = x2 * x2;
        const double h = __builtin_fma(x2, sg * c4, sg * c3);
        const double l = __builtin_fma(x2, sg * c1, sg * c0);
        const double x6 = x4 * x2;
        const double c = __builtin_fma(x4, sg * c2, l);
        return float(__builtin_fma(x6, h, c));
    }
    // quarter turns in y, and what is left of it (|y| < 120: the product fits the conversion)
    static __device__ __forceinline__ double reduce(double x, int& n)
    {
        const double r = x * hpiInv;
        n = (int32_t(r) + 0x800000) >> 24;
        return __builtin_fma(-double(n), hpi, x);
    }
    static __device__ __forceinline__ float sinCos(float y, int cosine)
    {
        double x = double(y);
        if (top12(y) < top12(0x1.921FB6p-1f)) { // |y| below pi / 4
            if (top12(y) < top12(0x1p-12f))
                return cosine ? 1.0f : y;
            return poly(x, x * x, cosine, false);
        }
        int n;
        x = reduce(x, n);
        const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        return poly(x * sign, x * x, n ^ cosine, (n & 2) != 0);
    }
};
RTX_DEV float hostSinf(float y)
{
    return HostTrig::top12(y) < HostTrig::top12(120.0f) ? HostTrig::sinCos(y, 0) : sinf(y);
}
RTX_DEV float hostCosf(float y)
{
    return HostTrig::top12(y) < HostTrig::top12(120.0f) ? HostTrig::sinCos(y, 1) : cosf(y);
}
// atan2f, acosf and tanf of the same library, which the environment map's lookup turns a direction
// into texture coordinates with (src/environment_mapping.cpp:167-207): a last-bit difference there
// picks the neighbouring texel once in ten thousand lookups of a 4096-texel-wide map.  These are
// glibc 2.35's e_atan2f.c / s_atanf.c / e_acosf.c / k_tanf.c (the float forms of Sun's fdlibm:
// argument reduction to a short interval, an odd polynomial or a rational function, all in float)
// and, for tanf's argument, the quarter-turn reduction in double that s_sinf.c uses, split into a
// float head and tail (e_rem_pio2f.c).  Every operation is a single IEEE float operation in the
// order written there; 1.4 billion arguments of atanf, 1.07 billion of acosf, 400 million pairs of
// atan2f and all 2.2 billion floats below 120 in magnitude of tanf give the host's bits on the CPU.
RTX_DEV float hostAtanf(float x)
{
    const int32_t hx = __float_as_int(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {
        if (ix > 0x7f800000)
            return x + x;
        return hx > 0 ? 1.5707962513e+00f + 7.5497894159e-08f : -1.5707962513e+00f - 7.5497894159e-08f;
    }
    if (ix < 0x3ee00000) {
        if (ix < 0x31000000)
            return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) {
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else if (ix < 0x401c0000) {
            id = 2;
            x = (x - 1.5f) / (1.0f + 1.5f * x);
        } else {
            id = 3;
            x = -1.0f / x;
        }
    }
    float z = x * x;
    const float w = z * z;
    const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f + w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    if (id < 0)
        return x - x * (s1 + s2);
    const float hi = id == 0 ? 4.6364760399e-01f : id == 1 ? 7.8539812565e-01f : id == 2 ? 9.8279368877e-01f : 1.5707962513e+00f;
    const float lo = id == 0 ? 5.0121582440e-09f : id == 1 ? 3.7748947079e-08f : id == 2 ? 3.4473217170e-08f : 7.5497894159e-08f;
    z = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -z : z;
}
RTX_DEV float hostAtan2f(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = __float_as_int(x), ix = hx & 0x7fffffff, hy = __float_as_int(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000)
        return x + y;
    if (hx == 0x3f800000)
        return hostAtanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2); // 2 * sign(x) + sign(y)
    if (iy == 0)
        return m < 2 ? y : m == 2 ? pi + tiny : -pi - tiny;
    if (ix == 0)
        return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000)
            return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000)
        return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60)
        z = 0.0f;
    else
        z = hostAtanf(fabsf(y / x));
    return m == 0 ? z : m == 1 ? __uint_as_float(__float_as_uint(z) ^ 0x80000000u) : m == 2 ? pi - (z - pi_lo) : (z - pi_lo) - pi;
}
RTX_DEV float hostAcosf(float x)
{
    const float pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f;
    const float qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const int32_t hx = __float_as_int(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000)
        return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000)
        return (x - x) / (x - x); // (outside [-1, 1], or a NaN: NaN)
    const bool small = ix < 0x3f000000;
    if (small && ix <= 0x23000000)
        return pio2_hi + pio2_lo;
    const float z = small ? x * x : hx < 0 ? (1.0f + x) * 0.5f : (1.0f - x) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    if (small)
        return pio2_hi - (x - (pio2_lo - x * r));
    const float s = __builtin_sqrtf(z);
    if (hx < 0) {
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float w = r * s + c;
    return 2.0f * (df + w);
}
// (k_tanf.c: the tangent, or with iy = -1 the negative cotangent, of x + y on [-pi / 4, pi / 4])
RTX_DEV float hostKernelTanf(float x, float y, int iy)
{
    const float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    const int32_t hx = __float_as_int(x), ix = hx & 0x7fffffff;
    if (ix < 0x39000000 && int(x) == 0) {
        if ((ix | (iy + 1)) == 0)
            return 1.0f / fabsf(x);
        return iy == 1 ? x : -1.0f / x;
    }
    const bool big = ix >= 0x3f2ca140;
    if (big) {
        if (hx < 0) {
            x = -x;
            y = -y;
        }
        const float z = pio4 - x, w = pio4lo - y;
        x = z + w;
        y = 0.0f;
        if (fabsf(x) < 0x1p-13f)
            return float((1 - ((hx >> 30) & 2)) * iy) * (1.0f - float(2 * iy) * x);
    }
    float z = x * x;
    float w = z * z;
    float r = 1.3333334029e-01f + w * (2.1869488060e-02f + w * (3.5920790397e-03f + w * (5.8804126456e-04f + w * (7.8179444245e-05f + w * -1.8558637748e-05f))));
    float v = z * (5.3968254477e-02f + w * (8.8632395491e-03f + w * (1.4562094584e-03f + w * (2.4646313977e-04f + w * (7.1407252108e-05f + w * 2.5907305826e-05f)))));
    float s = z * x;
    r = y + z * (s * (r + v) + y);
    r += 3.3333334327e-01f * s;
    w = x + r;
    if (big) {
        v = float(iy);
        return float(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1)
        return w;
    z = __uint_as_float(__float_as_uint(w) & 0xfffff000u);
    v = r - (z - x);
    const float a = -1.0f / w;
    const float t = __uint_as_float(__float_as_uint(a) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}
RTX_DEV float hostTanf(float x)
{
    const int32_t ix = __float_as_int(x) & 0x7fffffff;
    if (ix <= 0x3f490fda)
        return hostKernelTanf(x, 0.0f, 1);
    if (HostTrig::top12(x) >= HostTrig::top12(120.0f))
        return tanf(x);
    double dx = double(x);
    const double r = dx * HostTrig::hpiInv;
    const int n = (int32_t(r) + 0x800000) >> 24;
    dx = dx - double(n) * HostTrig::hpi; // (this file of the library is built without fused multiply-add)
    const float head = float(dx), tail = float(dx - double(head));
    return hostKernelTanf(head, tail, 1 - ((n & 1) << 1));
}

// src/environment_mapping.cpp:167-207; the uv arithmetic is double there
RTX_DEV V3 envColor(const DScene& sc, const rtx_features& f, V3 d)
{
    const V3 bg = mk(sc.env.background);
    if (sc.env.texture < 0)
        return bg;
    const rtx_texture& tx = sc.textures[sc.env.texture];
    const double kPi = 3.14159265358979323846;
    const float x = float(double(hostAtan2f(d.z, d.x)) / (2 * kPi) + 0.5);
    const float cosUp = dot(normalize(d), mk(0.f, 1.f, 0.f));
    float y;
    if (sc.env.mapping == RTX_ENV_SPHERICAL)
        y = float(-(kPi / double(sc.env.vfov_radians)) * (double(hostAcosf(cosUp)) / kPi - 0.5) + 0.5);
    else
        y = float(0.5 / double(hostTanf(0.5f * sc.env.vfov_radians)) / double(hostTanf(hostAcosf(cosUp))) + 0.5);
    if (fabs(double(y) - 0.5) > 0.5)
        return bg;
    if (f.enable_bilinear_texture_filtering)
        return texelBilinear(sc, tx, 0, x, y);
    return texelNearest(sc, tx, x, y);
}

// --------------------------------------------------------------- shading ----
// src/shading.cpp:90-96 (direction only; the origin is point + 1e-5 * r)
RTX_DEV V3 reflectDir(V3 d, V3 normal)
{
    return normalize(d) - 2 * dot(normalize(normal), normalize(d)) * normalize(normal);
}

// src/shading.cpp:13-54.  `albedo` is kd or the texture lookup, already chosen.
RTX_DEV V3 phong(V3 lightPos, V3 lightColor, V3 position, V3 rayDir, V3 normal, V3 albedo, V3 ks, float shininess)
{
    const V3 lightDir = normalize(lightPos - position);
    const float ndl = dot(normalize(normal), lightDir);
    if (ndl < 0.0f)
        return mk(0, 0, 0);
    const V3 r = reflectDir(lightDir, normal);
    const float d = fabsf(dot(normalize(r), normalize(rayDir)));
    const V3 specular = lightColor * ks * powf(d, shininess);
    return lightColor * albedo * ndl + specular;
}

// diffuse colour of a hit: kd, or the texture with the reference's filter choice
// (src/shading.cpp:26-32)
RTX_DEV V3 albedoOf(const DScene& sc, const rtx_features& f, const rtx_material& m, V3 ro, V3 rd, const Hit& hit)
{
    if (m.texture >= 0 && f.enable_texture_mapping) {
        const rtx_texture& tx = sc.textures[m.texture];
        if (f.enable_bilinear_texture_filtering) {
            if (f.enable_mipmap_texture_filtering)
                return texelTrilinear(sc, tx, ro, rd, hit);
            return texelBilinear(sc, tx, 0, hit.u, hit.v);
        }
        return texelNearest(sc, tx, hit.u, hit.v);
    }
    return mk(m.kd);
}

// ------------------------------------------------------------------- RNG ----
// Keyed uniforms for light and glossy sampling: draw number `counter` of sample
// `sample` of pixel `pixel` is a pure function of its key, so any execution
// order gives the same image.  The value is float(bits) / 2^32 clamped below 1,
// which is what std::uniform_real_distribution<float>(0,1) makes of one output
// of a 32-bit engine.
RTX_DEV uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
RTX_DEV uint32_t sampleKey(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    const uint32_t h = mix32(seed ^ (pixel * 0x9e3779b1u));
    return mix32(h ^ (sample * 0xc2b2ae35u) ^ 0x27d4eb2fu);
}
RTX_DEV float keyedUniform(uint32_t key, uint32_t counter)
{
    const uint32_t bits = mix32(key ^ (counter * 0x85ebca77u) ^ 0x68bc21ebu);
    float u = float(bits) * 2.3283064365386963e-10f; // exact: / 2^32
    if (u >= 1.0f)
        u = 0.99999994f;
    return u;
}

// ------------------------------------------ environment light sampling ----
// The reference's sample2D (src/sampling.cpp:6-60) seeds ONE engine from a
// random_device draw and then runs std::shuffle and uniform_real_distribution on
// it.  With the keyed stream standing in for random_device, output m of the
// engine made from draw `index` is bits(index) for m = 0 and
// mix32(bits(index) + m + 1) after that; the shuffle and the integer
// distribution below are libstdc++'s (bits/stl_algo.h, bits/uniform_int_dist.h:
// two swap positions per engine call, Lemire's nearly divisionless reduction).
constexpr int kEnvMaxPerTry = 8; // side of the largest multi-jitter grid handled (light samples < 810)

struct DrawEngine {
    uint32_t base, m;
};
RTX_DEV DrawEngine engineAt(uint32_t key, uint32_t index)
{
    return DrawEngine { mix32(key ^ (index * 0x85ebca77u) ^ 0x68bc21ebu), 0u };
}
RTX_DEV uint32_t engineNext(DrawEngine& g)
{
    const uint32_t v = g.m == 0 ? g.base : mix32(g.base + g.m + 1u);
    g.m++;
    return v;
}
RTX_DEV uint32_t uniformBelow(DrawEngine& g, uint32_t range)
{
    uint64_t product = uint64_t(engineNext(g)) * uint64_t(range);
    uint32_t low = uint32_t(product);
    if (low < range) {
        const uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            product = uint64_t(engineNext(g)) * uint64_t(range);
            low = uint32_t(product);
        }
    }
    return uint32_t(product >> 32);
}
// std::shuffle of a[first .. first + n)
RTX_DEV void shuffleBytes(uint8_t* a, int first, uint32_t n, DrawEngine& g)
{
    if (n == 0)
        return;
    uint32_t i = 1;
    if (n % 2 == 0) {
        const uint32_t p = uniformBelow(g, 2);
        const uint8_t t = a[first + i];
        a[first + i] = a[first + p];
        a[first + p] = t;
        i++;
    }
    while (i != n) {
        const uint32_t swapRange = i + 1;
        const uint32_t x = uniformBelow(g, swapRange * (swapRange + 1));
        const uint32_t p0 = x / (swapRange + 1), p1 = x % (swapRange + 1);
        uint8_t t = a[first + i];
        a[first + i] = a[first + p0];
        a[first + p0] = t;
        i++;
        t = a[first + i];
        a[first + i] = a[first + p1];
        a[first + p1] = t;
        i++;
    }
}
// Sample number c (row-major: c = j * h + i) of sample2D(rect, h, h) whose engine is
// seeded from draw `index`: the shuffles are replayed, then the engine is moved to
// this sample's two jitter draws.
__device__ __attribute__((noinline)) void sample2DAt(float4 rect, int h, uint32_t key, uint32_t index, int c, float& sx, float& sy)
{
    DrawEngine g = engineAt(key, index);
    uint8_t cell[kEnvMaxPerTry * kEnvMaxPerTry]; // rows[j][i] = (j, i), packed j * 16 + i
    uint8_t order[kEnvMaxPerTry];
    for (int j = 0; j < h; j++)
        for (int i = 0; i < h; i++)
            cell[j * h + i] = uint8_t(j * 16 + i);
    for (int i = 0; i < h; i++)
        order[i] = uint8_t(i);
    for (int j = 0; j < h; j++)
        shuffleBytes(cell, j * h, uint32_t(h), g);
    for (int i = 0; i < h; i++) {
        shuffleBytes(order, 0, uint32_t(h), g);
        int j = 0;
        const uint8_t temp = cell[i];
        for (int x = 0; x < h; x++) {
            cell[j * h + i] = cell[int(order[x]) * h + i];
            j = order[x];
        }
        cell[j * h + i] = temp;
    }
    g.m += 2u * uint32_t(c);
    const int j = c / h, i = c - j * h;
    const float centerX = float(cell[c] >> 4), centerY = float(cell[c] & 15);
    float jx = float(engineNext(g)) * 2.3283064365386963e-10f;
    if (jx >= 1.0f)
        jx = 0.99999994f;
    float jy = float(engineNext(g)) * 2.3283064365386963e-10f;
    if (jy >= 1.0f)
        jy = 0.99999994f;
    const float scale = float(1. / double(h + 1));
    const float subX = centerX + jx, subY = centerY + jy;
    sx = rect.x + (rect.z - rect.x) * (float(double(i) / (double(h) + 1.)) + scale * subX);
    sy = rect.y + (rect.w - rect.y) * (float(double(j) / (double(h) + 1.)) + scale * subY);
}
// EnvironmentMap::getRayForCoordinate, src/environment_mapping.cpp:130-165
RTX_DEV V3 envDirection(const DScene& sc, float cx, float cy)
{
    const double kPi = 3.14159265358979323846;
    const float azimuth = float(2.0 * kPi * (double(cx) - 0.5));
    float pitch;
    if (sc.env.mapping == RTX_ENV_SPHERICAL)
        pitch = float((0.5 - double(cy)) * double(sc.env.vfov_radians));
    else
        pitch = float(atan(double(hostTanf(0.5f * sc.env.vfov_radians)) * (0.5 - double(cy)) / 0.5));
    const float px = hostCosf(pitch), py = hostSinf(pitch);
    return mk(px * hostCosf(azimuth), py, px * hostSinf(azimuth));
}

// std::default_random_engine (minstd_rand0: x <- 16807 x mod 2^31-1, seed 1) as
// the reference's single-thread drivers consume it (src/render.cpp:101-102,
// 160-161,219-220): one sequential stream over all pixels.  16807^k mod
// (2^31-1) gives the state after k draws, so each sample finds its own place in
// that stream.
constexpr uint64_t kMinstdM = 2147483647ull;
RTX_DEV uint64_t mulMod(uint64_t a, uint64_t b)
{
    uint64_t z = a * b; // < 2^62
    z = (z & kMinstdM) + (z >> 31);
    z = (z & kMinstdM) + (z >> 31);
    return z >= kMinstdM ? z - kMinstdM : z;
}
RTX_DEV uint64_t minstdAt(uint64_t draws)
{
    uint64_t r = 1, b = 16807;
    while (draws) {
        if (draws & 1)
            r = mulMod(r, b);
        b = mulMod(b, b);
        draws >>= 1;
    }
    return r;
}
// uniform_real_distribution<float>(a, b) of libstdc++ over one minstd output
RTX_DEV float minstdUniform(uint64_t& x, float a, float b)
{
    x = mulMod(x, 16807ull);
    float c = float(uint32_t(x - 1)) / 2147483648.0f;
    if (c >= 1.0f)
        c = 0.99999994f;
    return c * (b - a) + a;
}

// ---------------------------------------------------------------- camera ----
// quat * vec3, glm/detail/type_quat.inl:347-354
RTX_DEV V3 rotate(float qw, V3 qv, V3 v)
{
    const V3 uv = cross(qv, v);
    const V3 uuv = cross(qv, uv);
    return v + ((uv * qw) + uuv) * 2.0f;
}
// glm::quat(euler), glm/detail/type_quat.inl:208-217 (on the device: used only by
// the motion-blur driver, whose per-sample cameras cannot be prepared on the host)
RTX_DEV void quatFromEuler(V3 e, float& qw, V3& qv)
{
    const V3 h = e * 0.5f;
    const V3 c = mk(hostCosf(h.x), hostCosf(h.y), hostCosf(h.z));
    const V3 s = mk(hostSinf(h.x), hostSinf(h.y), hostSinf(h.z));
    qw = c.x * c.y * c.z + s.x * s.y * s.z;
    qv = mk(s.x * c.y * c.z - c.x * s.y * s.z, c.x * s.y * c.z + s.x * c.y * s.z, c.x * c.y * s.z - s.x * s.y * c.z);
}
// Trackball::generateRay, framework/src/trackball.cpp:101-110
RTX_DEV void cameraRay(const rtx_camera& cam, float px, float py, V3& o, V3& d)
{
    const V3 dirCam = normalize(mk(-px * cam.half_width, py * cam.half_height, 1.0f));
    o = mk(cam.position);
    d = rotate(cam.quat[0], mk(cam.quat[1], cam.quat[2], cam.quat[3]), dirCam);
}

} // namespace rtxd
