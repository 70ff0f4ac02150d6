// Acceleration structure built on the GPU (rtx_scene_build_device): a linear BVH over the
// Morton codes of the triangles' box centres (rocPRIM radix sort; hierarchy as in Karras,
// "Maximizing parallelism in the construction of BVHs, octrees and k-d trees", 2012), whose
// splits follow the cells of the Morton grid.  Subtrees of at most kLeafMax triangles are
// leaves; so are, from a depth on that keeps the tree within the traversal's stack, subtrees of
// up to kDeepLeafMax (the few deep chains a linear BVH grows where triangles crowd).  Boxes
// are made bottom-up.  Everything is written straight into the records the traversal kernels
// read (binary nodes + triangles in one array, the four-wide mirrored copies of the
// wave-shared walk, attributes in leaf order).
// The reference builds its tree on the host (src/bounding_volume_hierarchy.cpp:101-180); a
// closest hit does not depend on the tree's shape (only exact-t ties between two triangles
// can come out differently, as for any visiting order), so this tree gives the same frames.
// (Tried and dropped: the balanced tree that halves the Morton sequence level by level.  It is
// shallow by construction, but its boxes overlap: the headline frame took 1.62 ms on it against
// 0.91 ms on the host's tree.)
// Included by rtx_hip.hip (one translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "rtx.h"
#include "rtx_device.hpp"

namespace rtxbuild {
using namespace rtxd;

constexpr int kLeafMax = 4;      // a subtree with this many triangles or fewer is one leaf
constexpr int kDeepLeafMax = 64; // ... and from BuildBuffers::deepFrom levels down, one with this many

struct BuildBuffers {
    // inputs (device copies of the caller's arrays, in id order)
    const rtx_tri* tris;
    const rtx_tri_attr* attrs;
    uint32_t n;
    uint32_t deepFrom; // inner nodes at this depth or deeper with at most kDeepLeafMax triangles are leaves
    // per triangle
    float* primLo; // 3 floats each
    float* primHi;
    uint64_t *keys, *keysSorted;
    uint32_t *order, *orderSorted; // orderSorted[p] = id of the p-th triangle in Morton order
    // per inner node of the full hierarchy (n - 1 of them; node 0 is the root)
    int32_t *left, *right;       // >= 0 inner node, < 0: ~position of a single triangle
    uint32_t *first, *last;      // range of positions under the node
    int32_t* parent;             // of inner nodes
    int32_t* leafParent;         // of single triangles (by position)
    uint32_t* arrived;           // refit: children that have reported
    float *nodeLo, *nodeHi;
    uint32_t *alive, *wide;      // flags: an inner node of the collapsed tree / one that heads a four-wide node
    uint32_t *aliveIndex, *wideIndex; // exclusive running sums of the flags
    // scalars: [0..5] bounds of the boxes' centres as ordered integers, [6] error flag, [7] deepest inner node, [8] leaves
    uint32_t* scalars;
};

// floats as integers that order the same way
RTX_DEV uint32_t orderedBits(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float fromOrderedBits(uint32_t u)
{
    const uint32_t v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &v, 4);
    return f;
}

__global__ void __launch_bounds__(256) primBoundsKernel(BuildBuffers B)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    float lo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, hi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
    bool bad = false;
    if (i < B.n) {
        const rtx_tri& t = B.tris[i];
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(fminf(t.v0[a], t.v1[a]), t.v2[a]);
            hi[a] = fmaxf(fmaxf(t.v0[a], t.v1[a]), t.v2[a]);
            bad = bad || !(fabsf(t.v0[a]) < __builtin_inff()) || !(fabsf(t.v1[a]) < __builtin_inff()) || !(fabsf(t.v2[a]) < __builtin_inff());
            B.primLo[size_t(i) * 3 + a] = lo[a];
            B.primHi[size_t(i) * 3 + a] = hi[a];
        }
    }
    if (bad)
        atomicOr(&B.scalars[6], 1u);
    // bounds of the boxes' centres, for the Morton grid
    for (int a = 0; a < 3; a++) {
        float c = i < B.n ? 0.5f * (lo[a] + hi[a]) : __builtin_inff(), C = i < B.n ? 0.5f * (lo[a] + hi[a]) : -__builtin_inff();
        for (int off = 32; off > 0; off >>= 1) {
            c = fminf(c, __shfl_xor(c, off, 64));
            C = fmaxf(C, __shfl_xor(C, off, 64));
        }
        if ((threadIdx.x & 63u) == 0 && c <= C) {
            atomicMin(&B.scalars[a], orderedBits(c));
            atomicMax(&B.scalars[3 + a], orderedBits(C));
        }
    }
}

RTX_DEV uint64_t spread21(uint32_t v) // bit i of v to bit 3 i
{
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void __launch_bounds__(256) mortonKernel(BuildBuffers B)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= B.n)
        return;
    uint64_t key = 0;
    for (int a = 0; a < 3; a++) {
        const float lo = fromOrderedBits(B.scalars[a]), hi = fromOrderedBits(B.scalars[3 + a]);
        const float c = 0.5f * (B.primLo[size_t(i) * 3 + a] + B.primHi[size_t(i) * 3 + a]);
        const float extent = hi - lo;
        float u = extent > 0.f ? (c - lo) / extent : 0.f;
        u = fminf(fmaxf(u, 0.f), 1.f);
        const uint32_t cell = min(uint32_t(u * 2097152.0f), 2097151u);
        key |= spread21(cell) << a;
    }
    B.keys[i] = key;
    B.order[i] = i;
}

// length of the common prefix of the keys at positions i and j (ties broken by the position), -1 outside the array
RTX_DEV int commonPrefix(const uint64_t* keys, int n, int i, int j)
{
    if (j < 0 || j >= n)
        return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b)
        return 64 + __clz(int(uint32_t(i) ^ uint32_t(j)));
    return __clzll((long long)(a ^ b));
}

__global__ void __launch_bounds__(256) hierarchyKernel(BuildBuffers B)
{
    const int n = int(B.n), i = int(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1)
        return;
    const uint64_t* keys = B.keysSorted;
    // direction of the node's range and its other end (Karras 2012, section 4)
    const int d = commonPrefix(keys, n, i, i + 1) - commonPrefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dMin = commonPrefix(keys, n, i, i - d);
    int lMax = 2;
    while (commonPrefix(keys, n, i, i + lMax * d) > dMin)
        lMax *= 2;
    int l = 0;
    for (int t = lMax / 2; t >= 1; t /= 2)
        if (commonPrefix(keys, n, i, i + (l + t) * d) > dMin)
            l += t;
    const int j = i + l * d;
    // the split: the highest position that shares more than the range's prefix with i
    const int dNode = commonPrefix(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (commonPrefix(keys, n, i, i + (s + t) * d) > dNode)
            s += t;
        if (t == 1)
            break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int32_t leftRef = lo == gamma ? ~gamma : gamma, rightRef = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    B.left[i] = leftRef;
    B.right[i] = rightRef;
    B.first[i] = uint32_t(lo);
    B.last[i] = uint32_t(hi);
    if (leftRef >= 0)
        B.parent[leftRef] = i;
    else
        B.leafParent[~leftRef] = i;
    if (rightRef >= 0)
        B.parent[rightRef] = i;
    else
        B.leafParent[~rightRef] = i;
    if (i == 0)
        B.parent[0] = -1;
}

RTX_DEV float loadShared(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RTX_DEV void storeShared(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Boxes bottom-up: one thread per triangle climbs towards the root; at every node the first of
// the two children's threads to arrive stops, the second merges the children's boxes and goes on.
__global__ void __launch_bounds__(256) refitKernel(BuildBuffers B)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= B.n)
        return;
    int32_t node = B.leafParent[p];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&B.arrived[node], 1u) == 0)
            return;
        __threadfence();
        float lo[3], hi[3];
        for (int side = 0; side < 2; side++) {
            const int32_t ref = side ? B.right[node] : B.left[node];
            for (int a = 0; a < 3; a++) {
                float cl, ch;
                if (ref < 0) {
                    const uint32_t id = B.orderSorted[~ref];
                    cl = B.primLo[size_t(id) * 3 + a];
                    ch = B.primHi[size_t(id) * 3 + a];
                } else {
                    cl = loadShared(&B.nodeLo[size_t(ref) * 3 + a]);
                    ch = loadShared(&B.nodeHi[size_t(ref) * 3 + a]);
                }
                lo[a] = side ? fminf(lo[a], cl) : cl;
                hi[a] = side ? fmaxf(hi[a], ch) : ch;
            }
        }
        for (int a = 0; a < 3; a++) {
            storeShared(&B.nodeLo[size_t(node) * 3 + a], lo[a]);
            storeShared(&B.nodeHi[size_t(node) * 3 + a], hi[a]);
        }
        node = B.parent[node];
    }
}

// whether the subtree of inner node i, `depth` inner nodes below the root, is one leaf
RTX_DEV bool collapsed(const BuildBuffers& B, uint32_t count, uint32_t depth)
{
    return count <= uint32_t(kLeafMax) || (depth >= B.deepFrom && count <= uint32_t(kDeepLeafMax));
}
RTX_DEV uint32_t depthOf(const BuildBuffers& B, int32_t node)
{
    uint32_t depth = 0;
    for (int32_t up = B.parent[node]; up >= 0; up = B.parent[up])
        depth++;
    return depth;
}

// Which inner nodes stay, how deep they sit, which of them head a four-wide node of the
// wave-shared walk (even depth), and the tree's leaf count.  (A node that stays has a parent
// that stays: counts shrink and depths grow on the way down.)
__global__ void __launch_bounds__(256) markKernel(BuildBuffers B)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= B.n - 1)
        return;
    const uint32_t depth = depthOf(B, int32_t(i));
    const bool alive = !collapsed(B, B.last[i] - B.first[i] + 1, depth);
    if (alive) {
        atomicMax(&B.scalars[7], depth);
        uint32_t leaves = 0;
        for (int side = 0; side < 2; side++) {
            const int32_t ref = side ? B.right[i] : B.left[i];
            leaves += (ref < 0 || collapsed(B, B.last[ref] - B.first[ref] + 1, depth + 1)) ? 1u : 0u;
        }
        if (leaves)
            atomicAdd(&B.scalars[8], leaves);
    }
    B.alive[i] = alive ? 1u : 0u;
    B.wide[i] = (alive && (depth & 1u) == 0) ? 1u : 0u;
}

struct Outputs {
    float4* records;   // nNodes binary node records, then n triangle records
    float4* waveNodes; // 8 copies of nWide four-wide nodes, 8 float4 each
    float4* waveTris;  // 6 float4 per triangle
    rtx_tri_attr* attrs;
    int32_t* idOrder;
    uint32_t nNodes, nWide;
};

// a child of inner node `parent` (which sits `parentDepth` inner nodes below the root) as the collapsed tree sees it
struct Child {
    bool leaf;
    uint32_t first, count; // leaf: its triangles' positions
    int32_t node;          // inner: the node
    float lo[3], hi[3];
};
RTX_DEV Child childOf(const BuildBuffers& B, int32_t ref, uint32_t parentDepth)
{
    Child c;
    if (ref < 0) {
        const uint32_t p = uint32_t(~ref), id = B.orderSorted[p];
        c.leaf = true;
        c.first = p;
        c.count = 1;
        c.node = -1;
        for (int a = 0; a < 3; a++) {
            c.lo[a] = B.primLo[size_t(id) * 3 + a];
            c.hi[a] = B.primHi[size_t(id) * 3 + a];
        }
        return c;
    }
    c.first = B.first[ref];
    c.count = B.last[ref] - B.first[ref] + 1;
    c.leaf = collapsed(B, c.count, parentDepth + 1);
    c.node = ref;
    for (int a = 0; a < 3; a++) {
        c.lo[a] = B.nodeLo[size_t(ref) * 3 + a];
        c.hi[a] = B.nodeHi[size_t(ref) * 3 + a];
    }
    return c;
}

// triangle records, the wave-shared walk's triangle records, attributes and the id table, in leaf (Morton) order
__global__ void __launch_bounds__(256) triangleRecordsKernel(BuildBuffers B, Outputs O)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= B.n)
        return;
    const uint32_t id = B.orderSorted[p];
    const rtx_tri t = B.tris[id];
    float4* r = O.records + (size_t(O.nNodes) + p) * 4;
    r[0] = make_float4(t.v0[0], t.v1[0], t.v2[0], t.n[0]);
    r[1] = make_float4(t.v0[1], t.v1[1], t.v2[1], t.n[1]);
    r[2] = make_float4(t.v0[2], t.v1[2], t.v2[2], t.n[2]);
    r[3] = make_float4(t.d, __int_as_float(int32_t(id)), __int_as_float(t.material), __int_as_float(0));
    float4* w = O.waveTris + size_t(p) * 6;
    w[0] = make_float4(t.n[0], t.n[1], t.n[2], t.d);
    w[1] = make_float4(t.v0[0], t.v0[1], t.v0[2], __int_as_float(int32_t(id)));
    w[2] = make_float4(t.v1[0], t.v1[1], t.v1[2], 0.f);
    // single IEEE subtractions, exactly the ones the triangle test makes per ray
    w[3] = make_float4(t.v2[0] - t.v0[0], t.v2[1] - t.v0[1], t.v2[2] - t.v0[2], 0.f);
    w[4] = make_float4(t.v2[0] - t.v1[0], t.v2[1] - t.v1[1], t.v2[2] - t.v1[2], 0.f);
    w[5] = make_float4(t.v1[0] - t.v0[0], t.v1[1] - t.v0[1], t.v1[2] - t.v0[2], 0.f);
    O.attrs[p] = B.attrs[id];
    O.idOrder[id] = int32_t(p);
}

RTX_DEV void markLastOfLeaf(const Outputs& O, uint32_t position)
{
    // top bit of the id word: the last triangle of its leaf
    uint32_t* a = reinterpret_cast<uint32_t*>(O.records + (size_t(O.nNodes) + position) * 4 + 3) + 1;
    *a |= 0x80000000u;
    uint32_t* b = reinterpret_cast<uint32_t*>(O.waveTris + size_t(position) * 6 + 1) + 3;
    *b |= 0x80000000u;
}

// binary node records (both children's boxes side by side per axis, see traverseBvh)
__global__ void __launch_bounds__(256) nodeRecordsKernel(BuildBuffers B, Outputs O)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= B.n - 1 || !B.alive[i])
        return;
    const uint32_t depth = depthOf(B, int32_t(i));
    const Child c0 = childOf(B, B.left[i], depth), c1 = childOf(B, B.right[i], depth);
    int32_t ref[2], cnt[2];
    const Child* cs[2] = { &c0, &c1 };
    for (int k = 0; k < 2; k++) {
        if (cs[k]->leaf) {
            ref[k] = ~int32_t(O.nNodes + cs[k]->first); // triangle records follow the node records
            cnt[k] = int32_t(cs[k]->count);
            markLastOfLeaf(O, cs[k]->first + cs[k]->count - 1);
        } else {
            ref[k] = int32_t(B.aliveIndex[cs[k]->node]);
            cnt[k] = 0;
        }
    }
    float4* r = O.records + size_t(B.aliveIndex[i]) * 4;
    r[0] = make_float4(c0.lo[0], c1.lo[0], c0.hi[0], c1.hi[0]);
    r[1] = make_float4(c0.lo[1], c1.lo[1], c0.hi[1], c1.hi[1]);
    r[2] = make_float4(c0.lo[2], c1.lo[2], c0.hi[2], c1.hi[2]);
    r[3] = make_float4(__int_as_float(ref[0]), __int_as_float(ref[1]), __int_as_float(cnt[0]), __int_as_float(cnt[1]));
}

// four-wide nodes: an inner node at even depth merged with its inner children, one copy per
// direction octant in that octant's mirror image, children nearest first (see DScene::waveNodes)
__global__ void __launch_bounds__(256) wideRecordsKernel(BuildBuffers B, Outputs O)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= B.n - 1 || !B.wide[i])
        return;
    const uint32_t depth = depthOf(B, int32_t(i));
    Child kids[4];
    int n = 0;
    for (int side = 0; side < 2; side++) {
        const Child c = childOf(B, side ? B.right[i] : B.left[i], depth);
        if (c.leaf) {
            kids[n++] = c;
        } else {
            kids[n++] = childOf(B, B.left[c.node], depth + 1);
            kids[n++] = childOf(B, B.right[c.node], depth + 1);
        }
    }
    int32_t refs[4];
    for (int k = 0; k < n; k++)
        refs[k] = kids[k].leaf ? ~int32_t(kids[k].first) : int32_t(B.wideIndex[kids[k].node]);
    for (uint32_t oct = 0; oct < 8; oct++) {
        float lo[4][3], hi[4][3], along[4];
        int order[4] = { 0, 1, 2, 3 };
        for (int k = 0; k < n; k++) {
            along[k] = 0.f;
            for (int a = 0; a < 3; a++) {
                const bool neg = (oct >> a) & 1u;
                lo[k][a] = neg ? -kids[k].hi[a] : kids[k].lo[a];
                hi[k][a] = neg ? -kids[k].lo[a] : kids[k].hi[a];
                along[k] += lo[k][a] + hi[k][a];
            }
        }
        for (int a = 1; a < n; a++) // (stable insertion sort by the mirrored centre along the diagonal)
            for (int b = a; b > 0 && along[order[b]] < along[order[b - 1]]; b--) {
                const int t = order[b];
                order[b] = order[b - 1];
                order[b - 1] = t;
            }
        float o[32];
        for (int slot = 0; slot < 4; slot++) {
            const bool used = slot < n;
            const int k = used ? order[slot] : 0;
            for (int a = 0; a < 3; a++) {
                o[8 * a + slot] = used ? lo[k][a] : __builtin_inff();
                o[8 * a + 4 + slot] = used ? hi[k][a] : -__builtin_inff();
            }
            o[24 + slot] = __int_as_float(used ? refs[k] : INT32_MIN);
        }
        o[28] = __int_as_float(n);
        // the triangle counts of leaf children, slot by slot, 16 bits each (traverseFrontier)
        uint32_t counts[2] = { 0u, 0u };
        for (int slot = 0; slot < n; slot++) {
            const Child& kid = kids[order[slot]];
            if (kid.leaf) {
                counts[slot >> 1] |= (min(kid.count, 0xffffu) & 0xffffu) << (16 * (slot & 1));
                if (kid.count > kFrontierMaxLeaf && oct == 0)
                    atomicOr(&B.scalars[9], 1u); // (a leaf no frontier item can count: that walk is not used on this scene)
            }
        }
        o[29] = __uint_as_float(counts[0]);
        o[30] = __uint_as_float(counts[1]);
        o[31] = 0.f;
        float4* dst = O.waveNodes + (size_t(oct) * O.nWide + B.wideIndex[i]) * 8;
        for (int q = 0; q < 8; q++)
            dst[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    }
}

// A shallow cut through the tree, level by level while it stays within `limit` boxes (for the
// per-frame visibility rectangles); one thread.  out: count, then 6 floats per box.
__global__ void cutKernel(BuildBuffers B, int limit, float* out)
{
    constexpr int kMax = 64;
    Child cur[kMax], next[kMax];
    uint32_t depth[kMax], nextDepth[kMax];
    int n = 1;
    cur[0].leaf = false;
    cur[0].node = 0;
    cur[0].first = 0;
    cur[0].count = B.n;
    for (int a = 0; a < 3; a++) {
        cur[0].lo[a] = B.nodeLo[a];
        cur[0].hi[a] = B.nodeHi[a];
    }
    depth[0] = 0;
    limit = min(limit, kMax);
    for (;;) {
        int inner = 0;
        for (int k = 0; k < n; k++)
            inner += cur[k].leaf ? 0 : 1;
        if (inner == 0 || n + inner > limit)
            break;
        int m = 0;
        for (int k = 0; k < n; k++) {
            if (cur[k].leaf) {
                nextDepth[m] = depth[k];
                next[m++] = cur[k];
                continue;
            }
            nextDepth[m] = depth[k] + 1;
            next[m++] = childOf(B, B.left[cur[k].node], depth[k]);
            nextDepth[m] = depth[k] + 1;
            next[m++] = childOf(B, B.right[cur[k].node], depth[k]);
        }
        n = m;
        for (int k = 0; k < n; k++) {
            cur[k] = next[k];
            depth[k] = nextDepth[k];
        }
    }
    out[0] = float(n);
    for (int k = 0; k < n; k++)
        for (int a = 0; a < 3; a++) {
            out[1 + k * 6 + a] = cur[k].lo[a];
            out[1 + k * 6 + 3 + a] = cur[k].hi[a];
        }
}

} // namespace rtxbuild
