// librtx_hip.so: gfx950 kernels and the C ABI of include/rtx.h.
//
// Two persistent kernels do all the work:
//   traceKernel   batch form of BvhInterface::intersect: 64 rays per wave fetch
//   renderKernel  the frame drivers: every lane owns one camera sample and runs
//                 it through a small state machine (radiance ray -> shadow rays
//                 of the light loop -> reflection children -> ...), so all lanes
//                 of a wave share one traversal loop whatever their state
// Waves fetch work units from a global counter until it runs dry.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

#include "rtx.h"
#include "rtx_device.hpp"
#include "rtx_build.hpp"

using namespace rtxd;

namespace {

// Register budget of the render kernels as waves per SIMD: the lean kernel
// (point lights + hard shadows) fits 128 VGPRs, i.e. 4 waves; the full one is left
// to the compiler.
#ifndef RTX_LEAN_WAVES
#define RTX_LEAN_WAVES 4
#endif
#ifndef RTX_LAYERS_WAVES
#define RTX_LAYERS_WAVES 4
#endif
#ifndef RTX_REL_WAVES
#define RTX_REL_WAVES 4
// (Mode 8 on trees of at most 16 levels, whose 24-row stacks leave the LDS for six workgroups per CU.  At 6 waves per
// SIMD (80 vector registers) the headline frame takes 0.57 ms against 0.585, 64 samples 1.71 against 1.88, the
// screen-filling view 2.48 against 2.67 -- but the registers spilled per work unit become 0.7 GB of HBM writes and
// 0.37 GB of reads per frame (75 and 96 MB at 4 waves: rocprofv3 WRITE_SIZE / FETCH_SIZE), and frames of one sample per
// pixel get 10 % slower.  Not taken; -DRTX_REL_WAVES=6 builds it.)
#endif
#ifndef RTX_FULL_WAVES
#define RTX_FULL_WAVES 2
#endif
#ifndef RTX_WALK_WAVES
#define RTX_WALK_WAVES 2
#endif
#ifndef RTX_LIGHT_WAVES
#define RTX_LIGHT_WAVES 4
#endif
constexpr int kMaxDepth = 8;    // frames of the reflection recursion kept per lane
constexpr int kMaxLayers = 16;  // transparent layers along one camera ray that are kept one by one (deeper ones are summed front to back)

// ------------------------------------------------------------ statistics ----
enum { S_PRIMARY = 0, S_SHADOW, S_BOUNCE, S_BOX, S_TRI, S_SHADED, S_TRACED, S_OVERFLOW, S_COUNT };

// Ray counters are kept per wave (a ballot's population count is the same in
// every lane, so they live in scalar registers); the optional per-lane box /
// triangle test counters are summed across the wave at the end.
struct LaneCounters {
    uint32_t primary = 0, shadow = 0, bounce = 0, shaded = 0;
    uint32_t traced = 0;   // counted rays that did walk the scene (not: rays of blocks that cannot see it, light samples behind the surface)
    uint32_t overflow = 0; // camera samples whose transparency chain went deeper than kMaxLayers
    TraceCounters tc { 0, 0 };
    RTX_DEV void countPrimary(bool lane) { primary += __popcll(laneMask(lane)); }
    RTX_DEV void countShadow(bool lane) { shadow += __popcll(laneMask(lane)); }
    RTX_DEV void countBounce(bool lane) { bounce += __popcll(laneMask(lane)); }
    RTX_DEV void countShaded(bool lane) { shaded += __popcll(laneMask(lane)); }
    RTX_DEV void countTraced(bool lane) { traced += __popcll(laneMask(lane)); }
    RTX_DEV void countOverflow(bool lane) { overflow += __popcll(laneMask(lane)); }
};

#ifdef RTX_PROBE
// diagnostic build: per unit of the last frame (ticks, longest lane's loop trips, longest lane's leaf trips, sum of the lanes' loop trips)
constexpr uint32_t kProbeUnits = 1u << 20;
__device__ uint4 g_probe[kProbeUnits];
// the walk of a deferred frame, per unit: ticks of the 100 MHz clock spent making rays, walking the tree, on hit
// attributes, on the shading points' numbers (an atomic's round trip), on shading points / exchange / tape, on the
// state machine's advance; then the iterations that served fans and the others
struct WalkProbe {
    uint32_t make, walk, attributes, numbers, records, advance, fanTrips, otherTrips;
    uint32_t rounds, entries, pad0, pad1; // traverseFrontier: rounds, live entries handled (summed over the wave)
};
__device__ WalkProbe g_walkProbe[kProbeUnits];
struct WalkLaps {
    WalkProbe p {};
    uint64_t last = wall_clock64();
    RTX_DEV void lap(uint32_t& into)
    {
        const uint64_t now = wall_clock64();
        into += uint32_t(now - last);
        last = now;
    }
};
#define RTX_LAP(field) laps.lap(laps.p.field)
#else
#define RTX_LAP(field)
#endif

RTX_DEV uint64_t waveSum(uint32_t v)
{
    uint64_t s = v;
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off, 64);
    return s;
}

// The statistics words are kept in kStatSlots copies, each on its own 256-byte
// line, and added up on the host: thousands of waves finishing together would
// otherwise queue their atomics on one memory channel.
constexpr int kStatSlots = 64, kStatStride = 32; // stride in 64-bit words

RTX_DEV uint32_t waveIndex()
{
    return __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
}

template <bool COUNT>
RTX_DEV void flushCounters(const LaneCounters& c, unsigned long long* stats)
{
    uint64_t bx = 0, tr = 0;
    if (COUNT) {
        bx = waveSum(c.tc.boxTests);
        tr = waveSum(c.tc.triTests);
    }
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* slot = stats + size_t(waveIndex() % kStatSlots) * kStatStride;
        atomicAdd(&slot[S_PRIMARY], (unsigned long long)c.primary);
        atomicAdd(&slot[S_SHADOW], (unsigned long long)c.shadow);
        atomicAdd(&slot[S_BOUNCE], (unsigned long long)c.bounce);
        atomicAdd(&slot[S_SHADED], (unsigned long long)c.shaded);
        atomicAdd(&slot[S_TRACED], (unsigned long long)c.traced);
        if (c.overflow)
            atomicAdd(&slot[S_OVERFLOW], (unsigned long long)c.overflow);
        if (COUNT) {
            atomicAdd(&slot[S_BOX], (unsigned long long)bx);
            atomicAdd(&slot[S_TRI], (unsigned long long)tr);
        }
    }
}

// ------------------------------------------------------------ work queue ----
// The launch's 64-lane units are handed out in chunks of `chunk` consecutive
// units.  One atomic counter serves only ~90 claims per microsecond chip-wide (the
// atomics of all XCDs meet at one memory channel), far too few for half a million
// units per frame, so the chunks are dealt round-robin onto kQueues ticket
// counters, each on its own 256-byte line: counter c hands out chunks c,
// c + kQueues, c + 2 kQueues, ...  A wave draws from its home counter and claims
// the NEXT chunk before it starts on the current one, so the atomic's round trip
// is hidden behind the work.  When a counter is spent the wave reads all counters
// at once (lane i reads counter i) and moves to the next one that still has chunks.
// A frame whose units were ordered on the device, most expensive first, is handed
// out in three regions with chunk sizes of their own, made on the device together
// with the order (plan = {end of region 0, 1, 2 in units, chunk of region 0, 1, 2}):
// one unit per claim where a unit runs for a hundred microseconds, a few where the
// claim's round trip is what a unit costs.  Units past the last region are not
// handed out at all: they are whole 8x8 blocks that no ray can hit, which the
// kernel's waves share out by wave number since each costs the same.
constexpr int kQueues = 64, kQueueStride = 64; // stride in 32-bit words
constexpr int kRegions = 3;

struct WorkQueue {
    uint32_t* tickets;
    uint32_t nUnits, chunk, nChunks;
    const uint32_t* plan; // device, NULL: one region of `chunk` units per claim
    uint32_t aheadFrom;   // first region whose chunks are claimed ahead (0: all of them)
};
struct WorkCursor {
    uint32_t queue;  // wave-uniform: the counter this wave draws from
    uint32_t ticket; // lane 0: result of the claim in flight
    bool claimed;    // wave-uniform: a claim is in flight
    // wave-uniform: where the regions end, in units and in chunks (cumulative), and their chunk sizes
    uint32_t endUnit[kRegions], endChunk[kRegions], chunk[kRegions];
};
RTX_DEV void claimTicket(const WorkQueue& q, WorkCursor& w)
{
    uint32_t k = 0;
    if ((threadIdx.x & 63) == 0)
        k = atomicAdd(q.tickets + w.queue * kQueueStride, 1u);
    w.ticket = k;
    w.claimed = true;
}
RTX_DEV void openCursor(const WorkQueue& q, WorkCursor& w)
{
    w.queue = waveIndex() % kQueues;
    uint32_t unitsBefore = 0, chunksBefore = 0;
    for (int r = 0; r < kRegions; r++) {
        uint32_t end = q.nUnits, chunk = q.chunk;
        if (q.plan) {
            end = min(q.nUnits, max(unitsBefore, __builtin_amdgcn_readfirstlane(q.plan[r])));
            chunk = max(1u, __builtin_amdgcn_readfirstlane(q.plan[kRegions + r]));
        }
        chunksBefore += (end - unitsBefore + chunk - 1) / chunk;
        unitsBefore = end;
        w.endUnit[r] = end;
        w.endChunk[r] = chunksBefore;
        w.chunk[r] = chunk;
    }
    claimTicket(q, w);
}
// Units [first, last) of the next chunk; false once every chunk has been handed out.
RTX_DEV bool nextChunk(const WorkQueue& q, WorkCursor& w, uint32_t& first, uint32_t& last)
{
    for (;;) {
        if (!w.claimed)
            claimTicket(q, w);
        const uint32_t k = __builtin_amdgcn_readfirstlane(w.ticket);
        w.claimed = false;
        const uint64_t id = uint64_t(k) * kQueues + w.queue;
        if (id < w.endChunk[kRegions - 1]) {
            const uint32_t c = uint32_t(id);
            // claim the next chunk now, so that the atomic's round trip hides behind this
            // chunk's work (see makeQueue for when the longest blocks are left out)
            if (!q.plan || q.aheadFrom == 0 || c >= w.endChunk[q.aheadFrom - 1])
                claimTicket(q, w);
            if (c < w.endChunk[0]) {
                first = c * w.chunk[0];
                last = min(first + w.chunk[0], w.endUnit[0]);
            } else if (c < w.endChunk[1]) {
                first = w.endUnit[0] + (c - w.endChunk[0]) * w.chunk[1];
                last = min(first + w.chunk[1], w.endUnit[1]);
            } else {
                first = w.endUnit[1] + (c - w.endChunk[1]) * w.chunk[2];
                last = min(first + w.chunk[2], w.endUnit[2]);
            }
            return true;
        }
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t seen = __hip_atomic_load(q.tickets + lane * kQueueStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t open = __ballot(uint64_t(seen) * kQueues + lane < w.endChunk[kRegions - 1]);
        if (!open)
            return false;
        const uint32_t c = w.queue;
        const uint64_t ahead = ((open >> 1) >> c) | (open << (63 - c)); // bit j: counter (c + 1 + j) % 64
        w.queue = (c + 1 + uint32_t(__builtin_ctzll(ahead))) & 63u;
        claimTicket(q, w);
    }
}

// ---------------------------------------------------------- trace kernel ----
struct TraceArgs {
    DScene sc;
    rtx_features f;
    const rtx_ray* rays;
    rtx_hit* hits;
    uint64_t n;
    WorkQueue work;
    unsigned long long* stats;
};

template <int STACK, bool COUNT>
__global__ void __launch_bounds__(kBlock) traceKernel(TraceArgs A)
{
    __shared__ int32_t sStack[2 * LaneStack<STACK>::kLds * kBlock];
    __shared__ int32_t sWaveRefs[(kBlock / 64) * 2 * STACK];
    LaneStack<STACK> st { sStack, reinterpret_cast<float*>(sStack + LaneStack<STACK>::kLds * kBlock), int(threadIdx.x), sWaveRefs + (threadIdx.x >> 6) * 2 * STACK };
    LaneCounters lc;
    const int lane = threadIdx.x & 63;
    WorkCursor cursor;
    openCursor(A.work, cursor);
    uint32_t first, last;
    while (nextChunk(A.work, cursor, first, last)) {
        for (uint32_t unit = first; unit < last; unit++) {
        const uint64_t i = uint64_t(unit) * 64 + lane;
        lc.countPrimary(i < A.n);
        lc.countTraced(i < A.n);
        if (i < A.n) {
            const rtx_ray r = A.rays[i];
            float t = r.t;
            Hit h;
            const bool any = intersectScene<STACK, COUNT>(A.sc, st, A.f, mk(r.origin), mk(r.direction), t, h, lc.tc);
            rtx_hit o;
            o.t = t;
            o.tri = h.id;
            o.normal[0] = h.normal.x, o.normal[1] = h.normal.y, o.normal[2] = h.normal.z;
            o.bary[0] = h.bary.x, o.bary[1] = h.bary.y, o.bary[2] = h.bary.z;
            o.uv[0] = h.u, o.uv[1] = h.v;
            o.material = any ? h.material : -1;
            o.hit = any ? 1 : 0;
            A.hits[i] = o;
        }
        }
    }
    flushCounters<COUNT>(lc, A.stats);
}

// --------------------------------------------------------- render kernel ----
// n / d for n < 2^31 as a multiply-high and a shift (mul == 0: d is 1); exact, and
// scalar when n is wave-uniform.  The host picks mul = floor(2^(31+l) / d) + 1 and
// shift = l - 1 with l = ceil(log2 d).
struct FastDiv {
    uint32_t mul, shift;
};
RTX_DEV uint32_t fastDiv(uint32_t n, FastDiv d) { return d.mul ? (__umulhi(n, d.mul) >> d.shift) : n; }

// ---- deferred light sampling (frames with area lights) ------------------------------
// Such a frame is rendered in three passes instead of one state machine per camera sample:
//   walk     (renderKernel MODE 4) follows only the radiance rays of every camera sample (camera
//            ray, reflection / glossy children, depth first, exactly the order of
//            getFinalColor, src/render.cpp:16-66).  Each shaded hit becomes a ShadePoint record
//            with the number its light draws start at, and the sample's call tree is written
//            as a tape: MISS(value) | HIT(point) | OPEN(point, ks) ... CLOSE.
//   lights   (lightKernel, once per light that counts) the light loop of
//            computeLightContribution (src/light.cpp:139-189) for ALL shading points of the
//            frame: 64 / group points per wave pass, one lane per light sample, the samples'
//            contributions added in sample order along the lanes, one result per (point, light).
//   resolve  (resolveKernel) replays each sample's tape: Lo = sum of its lights' results in
//            light order, value = Lo + (sum of the children's values / count) * ks, the
//            samples of a pixel added in sample order, the driver's normalisation.
// The light draws are numbered along the depth-first walk (that numbering is what makes a frame
// reproducible), so a sample's tree has to be walked in order; but only the closest-hit queries
// decide its shape, and the light samples, the bulk of the rays, depend on nothing but their
// shading point.  Handing them to the whole chip at once takes them out of the walk's
// dependent chain: a pixel with a deep glossy tree (up to 781 shading points at depth 4 with
// fans of 5) no longer holds one wave for 781 light loops one after the other.
enum TapeOp : uint32_t { TAPE_MISS = 1, TAPE_HIT = 2, TAPE_OPEN = 3, TAPE_CLOSE = 4, TAPE_LINK = 5 };
constexpr uint32_t kTapeChunk = 8;             // entries per chunk; the last slot links to the next chunk when the tape goes on
constexpr uint32_t kHeadUntouched = 0xffffffffu; // heads: no sample of this rank here; 0: a black sample; else 1 + first chunk
constexpr uint32_t kNoPoint = 0xffffffffu;
constexpr int kPools = 64, kPoolStride = 64;   // sub-pools (a wave uses pool waveIndex % 64) and their counters' distance in words
struct ShadePoint { // what the light loop needs of one getFinalColor call
    float hp[3], rd[3], n[3], albedo[3], ks[3], shin;
    uint32_t key, counter; // the sample's key and the number of its first light draw here
    uint32_t flipped;      // the viewer-side normal of testVisibilityLightSample is -n
    uint32_t pad;
};
struct TapeEntry {
    uint32_t op, point;
    float a[3];  // MISS: the value; HIT / OPEN: Lo before the lights (the albedo when shading is off)
    float ks[3]; // OPEN: the factor of the children's mean
};
static_assert(sizeof(ShadePoint) == 80 && sizeof(TapeEntry) == 32, "record sizes");
struct Deferred {
    uint32_t* heads;       // per (output pixel, sample)
    TapeEntry* tape;       // pool c: chunks [c * tapePerPool, (c + 1) * tapePerPool)
    ShadePoint* points;    // pool c: points [c * pointsPerPool, ...)
    float* contrib;        // 3 floats per (point, counting light)
    uint32_t* counters;    // tape chunks of pool c at [c * kPoolStride], points at [(kPools + c) * kPoolStride], overflow flag at [2 * kPools * kPoolStride]
    uint32_t tapePerPool, pointsPerPool;
    uint32_t nCounting;    // lights that count under the frame's features (point lights; segment / parallelogram with soft shadows)
    uint32_t drawsPerPoint; // keyed draws one shading point's light loop consumes
    uint32_t fanLevels;    // > 0: glossy fans are traced side by side, their hits kept in LDS for this many levels
    uint32_t leafFans;     // the rays of a fan that ends the recursion are finished by the lanes that traced them
    uint32_t frontier;     // the rays of the fans a wave serves at a time are walked by the whole wave, a group of lanes per ray (traverseFrontier)
    uint32_t fansPerRound; // fans a wave serves per iteration at most (few, so that each ray's group has several lanes)
    uint2* primary;        // per (output pixel, sample): the camera ray's closest hit (t bits, triangle record or -1) from the camera pass; nullptr: the walk traces it
};
__host__ __device__ inline uint32_t* deferOverflow(const Deferred& D) { return D.counters + 2 * kPools * kPoolStride; }

constexpr int kMaxCutBoxes = 64;
struct RenderArgs {
    DScene sc;
    rtx_features f;
    rtx_camera cam, prev;
    OriginFrame camFrame; // the scene's records relative to cam.position (nodes == nullptr: none); only for drivers whose camera rays all start there
    int32_t W, H, driver, depth, samples, shadowSamples, glossyRays;
    uint32_t seed;
    float aperture, focal;
    // interleaved-tile partition
    int32_t tileSize, tileRank, tileWorld, tilesX, tilesY, tilesMine;
    int32_t compact; // 1: write this rank's tiles back to back, 0: Screen layout
    float* out;
    // work decomposition: a unit = pixelsPerUnit consecutive pixels of the tile
    // sequence x lanesPerPixel samples, repeated `rounds` times when a pixel has
    // more than 64 samples
    int32_t pixelsPerUnit, lanesPerPixel, rounds;
    WorkQueue work;
    unsigned long long* stats;
    // skip-ahead tables of the driver-level engine: its state before sample s of
    // pixel (x, y) is jumpX[x] * jumpY[y] * jumpS[s] mod 2^31-1
    const uint32_t *jumpX, *jumpY, *jumpS;
    int32_t jumpSamples; // entries of jumpS
    // When a unit's pixels never straddle an 8x8 block (pixelsPerUnit divides 64) the
    // block is found from the unit number alone, with wave-uniform arithmetic:
    // unit >> unitsPerBlockLog2 = block number in the tile sequence (-1: not so).
    int32_t unitsPerBlockLog2;
    uint32_t blocksPerTile, blocksPerRow;
    FastDiv divBlocksPerTile, divTilesX, divBlocksPerRow;
    // Pixels outside [cullX0, cullX1] x [cullY0, cullY1] (y bottom-up) cannot see the
    // scene's bounds from this camera: a block entirely outside is black without a
    // single ray being made (cullOn; see cullRectangle on the host side).
    int32_t cullOn, cullX0, cullX1, cullY0, cullY1;
    // The same statement one step finer: the boxes of a shallow cut through the tree
    // (every triangle lies in one of them), each projected like the root box.  A block
    // outside all of these rectangles cannot see the scene either.  Used when the blocks
    // are classified for the frame's order; 0 rectangles: the root rectangle alone counts.
    int32_t nCutRects;
    int32_t cutRects[kMaxCutBoxes][4]; // x0, x1, y0, y1 (y bottom-up), inclusive
    // 8x8 blocks are visited in the order blockOrder gives (position in the unit
    // sequence -> index into this rank's blocks, tile after tile), longest-running
    // first as measured on the previous frame; blockCost collects this frame's
    // measurement.  NULL: list order.
    const uint32_t* blockOrder;
    uint32_t* blockCost;
    uint32_t blocksMine; // tilesMine * blocksPerTile
    // Block i reports to blockCost[costIndex(A, i)]: blocks that are neighbours in the
    // list, and so in time, are costStride words apart, since atomics on one 256-byte
    // line all queue at one memory channel.  A unit shorter than costFloor ticks does
    // not report at all: the order kernel has already entered its block as cheap.
    uint32_t costStride, costFloor;
    Deferred defer;        // MODE 4 only
    // not NULL: the launch does nothing unless this word is non-zero (the one-kernel path standing by
    // for a deferred frame whose pools ran over)
    const uint32_t* runIf;
    // (last, so that the arguments above lie where the kernels' grouped argument loads have always found them)
    // RTX_DRIVER_COMPOSED: samples = timeSteps * lensSamples * jitterSamples, sample s = (t * lensSamples + l) *
    // jitterSamples + j; drawsPerSample of the sequential engine per sample (time 1, pixel 2, lens 3, each only
    // when its dimension is on)
    int32_t lensSamples, jitterSamples, timeSteps, drawsPerSample;
    int32_t transparency; // the transparency chain runs on every camera sample's ray (src/transparency.cpp:6-20)
    int32_t normalise;    // how a pixel's samples become its colour: NORM_*
};
RTX_DEV uint32_t costIndex(const RenderArgs& A, uint32_t block) { return (block & 63u) * A.costStride + (block >> 6); }

// q-th pixel of this rank's tile sequence: tiles in rank order, 8x8 blocks
// row-major inside a tile, Morton order inside a block.  Returns false for
// padding outside the image.  yTop counts rows from the top (Screen rows).
RTX_DEV bool pixelOfSequence(const RenderArgs& A, uint32_t q, int& x, int& yTop, uint32_t& outIndex)
{
    // q < 2^32: a padded frame has at most (32768 + 1023)^2 pixels
    const uint32_t ts = A.tileSize, tilePixels = ts * ts;
    const uint32_t k = q / tilePixels, r = q - k * tilePixels;
    if (k >= uint32_t(A.tilesMine))
        return false;
    const uint32_t tile = A.tileRank + k * A.tileWorld;
    const uint32_t ty = tile / uint32_t(A.tilesX), tx = tile - ty * uint32_t(A.tilesX);
    const uint32_t blocksPerRow = ts / 8;
    const uint32_t blk = r / 64, m = r % 64;
    const uint32_t by = blk / blocksPerRow, bx = blk - by * blocksPerRow;
    // de-interleave the 6 Morton bits
    const uint32_t mx = (m & 1) | ((m >> 1) & 2) | ((m >> 2) & 4);
    const uint32_t my = ((m >> 1) & 1) | ((m >> 2) & 2) | ((m >> 3) & 4);
    const uint32_t lx = bx * 8 + mx, ly = by * 8 + my;
    x = int(tx * ts + lx);
    yTop = int(ty * ts + ly);
    if (x >= A.W || yTop >= A.H)
        return false;
    outIndex = A.compact ? (k * tilePixels + ly * ts + lx) : (uint32_t(yTop) * uint32_t(A.W) + uint32_t(x));
    return true;
}

// The same within each row of 16 lanes (DPP row_shr:1): lanes 0, 16, 32, 48 receive 0.
RTX_DEV float fromPreviousLaneOfRow(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}
// Where lane `slot`'s pixel of `unit` is, and whether the whole unit can be skipped.
struct PixelPlace {
    int x, yTop;
    uint32_t outIndex;
    bool ok;     // the pixel exists (inside the image, inside this rank's tiles)
    bool culled; // wave-uniform: no pixel of the unit can see the scene's bounds
    int32_t costSlot; // wave-uniform: entry of blockCost this unit reports to, -1: none
};
// The 8x8 block with number `block` in the unit sequence (all wave-uniform)
struct BlockPlace {
    uint32_t own;     // index into this rank's blocks
    uint32_t k;       // index into this rank's tile list
    uint32_t x0, y0;  // block corner in the image, rows counted from the top
    uint32_t bx, by;  // block position inside its tile
    bool listed;      // the tile exists
    bool culled;      // no pixel of the block can see the scene's bounds
};
RTX_DEV BlockPlace placeBlock(const RenderArgs& A, uint32_t block)
{
    BlockPlace b;
    const uint32_t ts = A.tileSize;
    b.listed = block < A.blocksMine;
    b.own = (A.blockOrder && b.listed) ? A.blockOrder[block] : block;
    b.k = fastDiv(b.own, A.divBlocksPerTile);
    const uint32_t blk = b.own - b.k * A.blocksPerTile;
    const uint32_t tile = A.tileRank + b.k * A.tileWorld;
    const uint32_t ty = fastDiv(tile, A.divTilesX), tx = tile - ty * uint32_t(A.tilesX);
    b.by = fastDiv(blk, A.divBlocksPerRow);
    b.bx = blk - b.by * A.blocksPerRow;
    b.x0 = tx * ts + b.bx * 8;
    b.y0 = ty * ts + b.by * 8;
    b.culled = false;
    if (A.cullOn) {
        const int yHigh = A.H - 1 - int(b.y0), yLow = yHigh - 7; // the block's rows in bottom-up numbering
        b.culled = int(b.x0) + 7 < A.cullX0 || int(b.x0) > A.cullX1 || yHigh < A.cullY0 || yLow > A.cullY1;
    }
    return b;
}
RTX_DEV uint32_t blockPixelIndex(const RenderArgs& A, const BlockPlace& b, uint32_t mx, uint32_t my)
{
    const uint32_t ts = A.tileSize;
    return A.compact ? (b.k * ts * ts + (b.by * 8 + my) * ts + b.bx * 8 + mx) : ((b.y0 + my) * uint32_t(A.W) + b.x0 + mx);
}

RTX_DEV PixelPlace placePixel(const RenderArgs& A, uint32_t unit, int slot, bool laneUsed)
{
    PixelPlace pl;
    pl.x = pl.yTop = 0;
    pl.outIndex = 0;
    pl.culled = false;
    pl.costSlot = -1;
    if (A.unitsPerBlockLog2 < 0) {
        pl.ok = laneUsed && pixelOfSequence(A, unit * uint32_t(A.pixelsPerUnit) + uint32_t(slot), pl.x, pl.yTop, pl.outIndex);
        return pl;
    }
    const uint32_t part = unit & ((1u << A.unitsPerBlockLog2) - 1u);
    const BlockPlace b = placeBlock(A, unit >> A.unitsPerBlockLog2);
    if (A.blockCost && b.listed)
        pl.costSlot = int32_t(b.own);
    pl.culled = b.culled;
    // Morton position inside the block: de-interleave its 6 bits
    const uint32_t m = part * uint32_t(A.pixelsPerUnit) + uint32_t(slot);
    const uint32_t mx = (m & 1) | ((m >> 1) & 2) | ((m >> 2) & 4);
    const uint32_t my = ((m >> 1) & 1) | ((m >> 2) & 2) | ((m >> 3) & 4);
    pl.x = int(b.x0 + mx);
    pl.yTop = int(b.y0 + my);
    pl.ok = laneUsed && b.listed && pl.x < A.W && pl.yTop < A.H;
    pl.outIndex = blockPixelIndex(A, b, mx, my);
    return pl;
}

// One pixel (or one point's light result): three floats with ONE 12-byte store, so that
// a wave writing neighbouring pixels covers whole lines with a single instruction instead of
// three strided ones.
struct __attribute__((packed, aligned(4))) Rgb {
    float r, g, b;
};
RTX_DEV void storeRgb(float* dst, V3 c) { *reinterpret_cast<Rgb*>(dst) = Rgb { c.x, c.y, c.z }; }

RTX_DEV float fromLane(float v, int lane) // lane: wave-uniform
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// The value the previous lane of the wave holds (lane 0 receives 0): DPP
// wave_shr:1.  Must run with every lane of the wave enabled, since a disabled
// source lane reads as 0.
RTX_DEV float fromPreviousLane(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}

// state of the driver-level engine before the first draw of `sample` of pixel (x, y)
RTX_DEV uint64_t driverRngState(const RenderArgs& A, int x, int y, uint32_t pixelIndex, int sample, int drawsPerSample)
{
    if (sample < A.jumpSamples)
        return mulMod(mulMod(A.jumpX[x], A.jumpY[y]), A.jumpS[sample]);
    return minstdAt((uint64_t(pixelIndex) * uint64_t(A.samples) + uint64_t(sample)) * uint64_t(drawsPerSample));
}

struct Frame { // one suspended getFinalColor call (src/render.cpp:16-66)
    V3 Lo, ks, acc;
    V3 reflO, reflD, nrm;
    float blur;
    int32_t count, child, nChildren;
    uint32_t rngBase;
    int32_t traced; // the fan's rays have been traced ahead (fanT / fanK / fanSphere / fanSphereN hold their hits)
};

// A pixel from its samples: the one sample as it is (src/render.cpp:80-83), the sum in sample order times
// 1 / count (src/render.cpp:178, :122), the sum divided by the count (src/render.cpp:249)
enum Normalise { NORM_SINGLE = 0, NORM_RECIPROCAL = 1, NORM_DIVIDE = 2 };
RTX_DEV V3 normalisePixel(V3 accum, int how, int samples)
{
    if (how == NORM_RECIPROCAL) {
        const float inv = 1.0f / float(samples);
        return accum * mk(inv, inv, inv);
    }
    if (how == NORM_DIVIDE)
        return accum / float(samples);
    return accum;
}
enum PathState { PS_DONE = 0, PS_RADIANCE = 1, PS_SHADOW = 2, PS_AREA = 3, PS_ENV = 4, PS_FAN = 5 };
constexpr int kFanMax = 8; // glossy fans of up to this many rays are traced by the wave's lanes side by side
enum Phase { PH_NONE = -1, PH_NEXT_LIGHT, PH_ENV, PH_AFTER_LIGHTS, PH_NEXT_CHILD, PH_RETURN };

RTX_DEV int lane0() { return int(threadIdx.x & 63u); }
RTX_DEV float laneValue(float v, int srcLane) { return __shfl(v, srcLane, 64); }
RTX_DEV V3 laneValue(V3 v, int srcLane) { return mk(__shfl(v.x, srcLane, 64), __shfl(v.y, srcLane, 64), __shfl(v.z, srcLane, 64)); }

// One camera sample from its first ray to its colour.  Every lane of the wave
// calls this together; lanes without a sample pass active = false.
//
// Each lane runs the body of getFinalColor (src/render.cpp:16-66) as a small state
// machine around ONE traversal site: per iteration of the loop below a lane traces
// the ray it has in flight (camera / reflection ray, or the shadow ray of a point
// light), then advances to its next ray.  Segment and parallelogram lights are
// sampled by the whole wave instead: a lane that reaches such a light parks in
// PS_AREA, and the next iteration is a cooperative one in which the 64 lanes each
// take one of the light's samples for that lane's shading point (several points
// at a time when a light has 32 samples or fewer), trace its shadow ray, and the
// contributions are added up in sample order along the lanes and handed back.
// A path's long sequential run of shadow samples thus becomes one traversal step
// of a full wave, however few lanes of the wave have a shading point.
template <int STACK, bool COUNT, bool FULL, bool ENVL = false, bool FAN = false>
RTX_DEV V3 tracePath(const RenderArgs& A, LaneStack<STACK>& st, V3 camO, V3 camD, bool active, uint32_t rngKey, LaneCounters& lc, float firstLimit = kFltMax)
{
    const DScene& sc = A.sc;
    const rtx_features& f = A.f;
    const bool transparencyDriver = FULL && A.transparency != 0;

    int state = active ? PS_RADIANCE : PS_DONE;
    V3 ro = camO, rd = camD; // ray in flight
    V3 result = mk(0, 0, 0);
    uint32_t counter = 0;
    int level = 0, depthLeft = A.depth;

    // shading point of the current getFinalColor call
    V3 Ro = ro, Rd = rd, hp = mk(0, 0, 0), visN = mk(0, 0, 0), albedo = mk(0, 0, 0), ks = mk(0, 0, 0), hitN = mk(0, 0, 0);
    float shin = 1.f, Rt = 0.f;
    V3 Lo = mk(0, 0, 0), sh = mk(0, 0, 0);
    int li = 0;

    Frame frames[FULL ? kMaxDepth : 1];
    // Glossy fans traced ahead.  The rays of a fan are all drawn before the first of them
    // is followed (src/shading.cpp:98-105), so their closest hits do not depend on anything
    // the recursion does in between: a lane that has drawn a fan parks in PS_FAN, the next
    // iteration is a cooperative one in which the wave's lanes each trace one ray of that
    // lane's fan, and the hits are handed back.  The lane then follows its children one
    // after the other as before (their draws are numbered along that walk), but each
    // child's own traversal is already done.  What a deep glossy path costs per shading
    // point falls from glossyRays traversals one after the other to one.
    // (compiled in only for FAN: the storage costs the other paths registers)
    float fanT[FAN ? kMaxDepth : 1][FAN ? kFanMax : 1];
    int32_t fanK[FAN ? kMaxDepth : 1][FAN ? kFanMax : 1], fanSphere[FAN ? kMaxDepth : 1][FAN ? kFanMax : 1];
    V3 fanSphereN[FAN ? kMaxDepth : 1][FAN ? kFanMax : 1];
    bool preTraced = false; // the radiance ray in flight was traced ahead: pre* hold its hit
    float preT = kFltMax;
    int32_t preK = -1, preSphere = -1;
    V3 preSphereN = mk(0, 0, 0);
    const int fanGroup = min(max(A.glossyRays, 1), kFanMax), fanPerRound = 64 / fanGroup;
    const int fanSlot = lane0() / fanGroup, fanChild = lane0() - fanSlot * fanGroup;
    // transparency chain (src/transparency.cpp:6-20): alpha and colour of each
    // layer in front of the last one.  The reference folds them innermost first,
    // value = alpha_i * colour_i + (1 - alpha_i) * value; the first kMaxLayers layers are kept and
    // folded exactly so.  A chain has no bound there, so layers deeper than that are summed front to
    // back as they come (tailSum += tailThrough * alpha * colour, tailThrough *= 1 - alpha): the same
    // value up to the rounding of a different order of summation, under a factor
    // prod (1 - alpha_i) of the kMaxLayers layers in front.
    float layerAlpha[FULL ? kMaxLayers : 1];
    V3 layerColor[FULL ? kMaxLayers : 1];
    int nLayers = 0;
    bool lastLayer = true;
    float pendingAlpha = 1.0f, tailThrough = 1.0f;
    V3 nextLayerO = mk(0, 0, 0), tailSum = mk(0, 0, 0);

    // shadow query in flight: search limit, early-stop distance and whether the
    // light sample is behind the surface (then no traversal is needed); see
    // lightVisibility in rtx_device.hpp for why these leave the answer unchanged
    float shadowLimit = kFltMax, shadowStop = -__builtin_inff();
    bool shadowBehind = false;

    // cooperative sampling geometry (wave-uniform): `group` lanes work on one
    // shading point, perRound points are served per iteration; a light with more
    // than 64 samples takes several iterations per point (areaRound, areaCarry)
    const int lane = threadIdx.x & 63;
    const int nS = A.shadowSamples;
    const int group = nS < 64 ? nS : 64;
    const int perRound = 64 / group;
    const int mySlot = lane / group, mySample = lane - mySlot * group;
    int areaRound = 0;
    V3 areaCarry = mk(0, 0, 0);

    // environment light (sampleEnvironment, src/light.cpp:47-66): which try and which
    // sample of the try comes next, how many rays were kept, the sum so far
    // (compiled in only for ENVL: its code costs the other paths registers)
    const bool envLight = FULL && ENVL && f.enable_soft_shadow && f.enable_environment_mapping;
    const int envPerTry = int(fmax(sqrt(double(nS) / 10.), 1.)); // src/environment_mapping.cpp:110
    int envTry = 0, envCand = 0, envCount = 0;
    V3 envAvg = mk(0, 0, 0);

    bool shadedPending = false; // counted at the loop head, where the wave is converged
    bool overflowPending = false;
    bool firstQuery = true;
    while (__any(state != PS_DONE)) {
        const uint64_t pending = FULL ? __ballot(state == PS_AREA) : 0ull;
        const bool coop = pending != 0;
        const uint64_t pendingFan = (FAN && !coop) ? __ballot(state == PS_FAN) : 0ull;
        const bool coopFan = pendingFan != 0;
        lc.countShaded(shadedPending);
        shadedPending = false;
        if (FULL) {
            lc.countOverflow(overflowPending);
            overflowPending = false;
        }

        // ---- the ray this lane traces in this iteration
        V3 qo = ro, qd = rd;
        float t = kFltMax, qStop = -__builtin_inff(), qLimit = kFltMax;
        bool trace = false;
        // cooperative iteration: the shading point this lane samples for
        uint64_t served = 0;
        bool worker = false, workerBehind = false;
        V3 workerHp = mk(0, 0, 0), contribution = mk(0, 0, 0);
        if (FULL && coop) {
            int mine = -1;
            uint64_t m = pending;
            for (int slot = 0; slot < perRound && m; slot++) {
                const int p = __builtin_ctzll(m);
                m &= m - 1;
                served |= 1ull << p;
                if (mySlot == slot)
                    mine = p;
            }
            const int sampleIndex = areaRound * 64 + mySample;
            worker = mine >= 0 && sampleIndex < nS;
            const int src = mine >= 0 ? mine : 0;
            workerHp = laneValue(hp, src);
            const V3 pRd = laneValue(Rd, src), pN = laneValue(hitN, src), pAlbedo = laneValue(albedo, src), pKs = laneValue(ks, src), pVisN = laneValue(visN, src);
            const float pShin = laneValue(shin, src);
            const uint32_t pKey = uint32_t(__shfl(int(rngKey), src, 64)), pCounter = uint32_t(__shfl(int(counter), src, 64));
            const int srcLight = __shfl(li, src, 64); // every lane takes part: the source lane need not be a worker
            const int pLight = worker ? srcLight : 0;
            lc.countShadow(worker);
            if (worker) {
                const rtx_light& L = sc.lights[pLight];
                V3 pos, col;
                if (L.type == RTX_LIGHT_SEGMENT) {
                    // src/light.cpp:26-32
                    const V3 e0 = mk(L.p[0]), e1 = mk(L.p[1]);
                    const float a = keyedUniform(pKey, pCounter + uint32_t(sampleIndex));
                    pos = a * (e1 - e0) + e0;
                    col = (1 - a) * mk(L.c[0]) + a * mk(L.c[1]);
                } else {
                    // src/light.cpp:36-45
                    const float a1 = keyedUniform(pKey, pCounter + 2u * uint32_t(sampleIndex));
                    const V3 c1 = (1 - a1) * mk(L.c[0]) + a1 * mk(L.c[1]);
                    const V3 c2 = (1 - a1) * mk(L.c[2]) + a1 * mk(L.c[3]);
                    const float a2 = keyedUniform(pKey, pCounter + 2u * uint32_t(sampleIndex) + 1u);
                    col = (1 - a2) * c1 + a2 * c2;
                    pos = mk(L.p[0]) + mk(L.p[1]) * a1 + mk(L.p[2]) * a2;
                }
                contribution = phong(pos, col, workerHp, pRd, pN, pAlbedo, pKs, pShin);
                workerBehind = dot(normalize(pos - workerHp), normalize(pVisN)) < -1e-6f;
                qo = pos;
                qd = workerHp - pos;
                const float len = length(qd);
                const float scale = fmaxf(fmaxf(fmaxf(fabsf(workerHp.x), fabsf(workerHp.y)), fabsf(workerHp.z)), fmaxf(fmaxf(fabsf(pos.x), fabsf(pos.y)), fabsf(pos.z)));
                const bool shorten = len >= 0.2f && scale <= 64.0f;
                qLimit = shorten ? 1.01f : kFltMax;
                qStop = shorten ? 1.0f - 4e-3f / len : -__builtin_inff();
                trace = !workerBehind;
            }
        } else if (FAN && coopFan) {
            // lane = one ray of a parked lane's fan (returnGlossyRay, src/shading.cpp:67-88)
            int mine = -1;
            uint64_t m = pendingFan;
            for (int slot = 0; slot < fanPerRound && m; slot++) {
                const int p = __builtin_ctzll(m);
                m &= m - 1;
                served |= 1ull << p;
                if (fanSlot == slot)
                    mine = p;
            }
            const int src = mine >= 0 ? mine : 0;
            const Frame& own = frames[level > 0 ? level - 1 : 0]; // (a parked lane's fan is its innermost frame)
            const V3 fD = laneValue(own.reflD, src), fO = laneValue(own.reflO, src), fN = laneValue(own.nrm, src);
            const float fBlur = laneValue(own.blur, src);
            const uint32_t fBase = uint32_t(__shfl(int(own.rngBase), src, 64)), fKey = uint32_t(__shfl(int(rngKey), src, 64));
            const int fCount = __shfl(own.nChildren, src, 64);
            worker = mine >= 0 && fanChild < fCount;
            if (worker) {
                const V3 w = normalize(fD);
                const V3 tt = normalize(w - mk(0.1f, 0.0f, 0.0f));
                const V3 u = normalize(cross(tt, w));
                const V3 v = normalize(cross(w, u));
                const float ua = -fBlur / 2.0f + fBlur * keyedUniform(fKey, fBase + 2u * uint32_t(fanChild));
                const float va = -fBlur / 2.0f + fBlur * keyedUniform(fKey, fBase + 2u * uint32_t(fanChild) + 1u);
                const V3 dir = fD + (ua * u + va * v);
                const float cosinus = dot(normalize(dir), fN);
                const float sinus = __builtin_sqrtf(1.0f - cosinus * cosinus);
                qo = fO;
                qd = dir;
                trace = cosinus > 0.0f && sinus > 0.0f; // the others are never followed (src/render.cpp:31-38)
            }
        } else {
            if (firstQuery)
                t = firstLimit;
            lc.countShadow(state == PS_SHADOW || state == PS_ENV);
            lc.countPrimary(state == PS_RADIANCE && level == 0);
            lc.countBounce(state == PS_RADIANCE && level != 0);
            trace = (state == PS_RADIANCE && !(FAN && preTraced)) || state == PS_ENV || (state == PS_SHADOW && !shadowBehind);
            if (state == PS_SHADOW) {
                qStop = shadowStop;
                qLimit = shadowLimit;
            }
            if (state == PS_ENV)
                qStop = kFltMax; // only "hit anything?" matters: stop at the first hit (t drops below its initial kFltMax)
            firstQuery = false;
        }

        // ---- the one traversal site
        Hit hit;
        bool any = false;
        int bestK = -1, sphere = -1;
        V3 sphereN = mk(0, 0, 0);
        {
            lc.countTraced(trace);
            if (trace)
                any = closestHit<STACK, COUNT, true>(sc, st, f, qo, qd, t, bestK, sphere, sphereN, lc.tc, qStop, qLimit);
            if (FAN && !coop && !coopFan && state == PS_RADIANCE && preTraced) {
                t = preT;
                bestK = preK;
                sphere = preSphere;
                sphereN = preSphereN;
                any = bestK >= 0 || sphere >= 0;
                preTraced = false;
            }
            if (!coop && !coopFan && state == PS_RADIANCE)
                hitAttributes(sc, f, ro, rd, t, bestK, sphere, sphereN, hit);
        }

        int phase = PH_NONE;
        V3 value = mk(0, 0, 0);
        if (FULL && coop) {
            // testVisibilityLightSample (src/light.cpp:70-102) for this lane's sample, then
            // `avg += contribution` in sample order (src/light.cpp:160-163, 173-176)
            const V3 second = qo + t * qd;
            const float vis = (workerBehind || double(distance(workerHp, second)) > 1e-3) ? 0.0f : 1.0f;
            const V3 c = contribution * vis;
            const int cnt = min(group, nS - areaRound * 64);
            const bool head = mySample == 0;
            float rx = head ? areaCarry.x + c.x : c.x, ry = head ? areaCarry.y + c.y : c.y, rz = head ? areaCarry.z + c.z : c.z;
            for (int j = 1; j < cnt; j++) {
                const float nx = fromPreviousLane(rx) + c.x, ny = fromPreviousLane(ry) + c.y, nz = fromPreviousLane(rz) + c.z;
                rx = head ? rx : nx;
                ry = head ? ry : ny;
                rz = head ? rz : nz;
            }
            if ((areaRound + 1) * 64 < nS) {
                // more samples of the same point to come (one point per iteration here)
                areaCarry = mk(fromLane(rx, cnt - 1), fromLane(ry, cnt - 1), fromLane(rz, cnt - 1));
                areaRound++;
            } else {
                // the k-th served lane finds its sum in the last lane of group k
                const int k = __popcll(served & ((1ull << lane) - 1ull));
                const int from = min(k * group + cnt - 1, 63);
                const V3 avg = mk(__shfl(rx, from, 64), __shfl(ry, from, 64), __shfl(rz, from, 64));
                if ((served >> lane) & 1ull) {
                    const rtx_light& L = sc.lights[li];
                    Lo = Lo + avg * float(1.0 / double(float(nS)));
                    counter += (L.type == RTX_LIGHT_SEGMENT ? 1u : 2u) * uint32_t(nS);
                    li++;
                    state = PS_RADIANCE; // left again below
                    phase = PH_NEXT_LIGHT;
                }
                areaRound = 0;
                areaCarry = mk(0, 0, 0);
            }
        } else if (FAN && coopFan) {
            // the k-th served lane finds ray j of its fan in lane k * fanGroup + j
            const int k = __popcll(served & ((1ull << lane) - 1ull));
            const bool mineServed = (served >> lane) & 1ull;
            const int lv = level > 0 ? level - 1 : 0;
            for (int j = 0; j < fanGroup; j++) {
                const int from = min(k * fanGroup + j, 63);
                const float jt = __shfl(t, from, 64);
                const int jk = __shfl(bestK, from, 64), js = __shfl(sphere, from, 64);
                const V3 jn = laneValue(sphereN, from);
                if (mineServed) {
                    fanT[lv][j] = jt;
                    fanK[lv][j] = jk;
                    fanSphere[lv][j] = js;
                    fanSphereN[lv][j] = jn;
                }
            }
            if (mineServed) {
                frames[lv].traced = 1;
                state = PS_RADIANCE; // left again below
                phase = PH_NEXT_CHILD;
            }
        } else if (state == PS_SHADOW) {
            // testVisibilityLightSample, src/light.cpp:70-102: the shadow ray runs
            // from the light sample to the shading point (un-normalised direction)
            // and must end within 1e-3 of it; a ray that hits nothing counts as blocked
            const V3 second = ro + t * rd;
            float vis = 1.0f;
            if (shadowBehind)
                vis = 0.0f;
            else if (double(distance(hp, second)) > 1e-3)
                vis = 0.0f;
            Lo = Lo + sh * vis;
            li++;
            phase = PH_NEXT_LIGHT;
        } else if (FULL && ENVL && state == PS_ENV) {
            // src/light.cpp:56-59: an unblocked ray brings the map's colour from far away
            if (!any) {
                const V3 envN = dot(hitN, Rd) < 0 ? hitN : -hitN;
                envAvg = envAvg + phong(ro + 100000.f * rd, envColor(sc, f, rd), hp, Rd, envN, albedo, ks, shin);
            }
            phase = PH_ENV;
        } else if (state == PS_RADIANCE) {
            // result of a radiance ray = body of getFinalColor
            if (transparencyDriver && level == 0) {
                const float alpha = (any && hit.material >= 0) ? sc.mats[hit.material].transparency : 1.0f;
                lastLayer = !any || alpha == 1.0f;
                overflowPending = !lastLayer && nLayers == kMaxLayers; // (counted once per camera sample: reported in rtx_stats)
                if (!lastLayer) {
                    pendingAlpha = alpha;
                    const float adv = float(double(t) + 1e-6);
                    nextLayerO = ro + mk(adv, adv, adv) * rd;
                }
            }
            if (!any) {
                value = (f.enable_environment_mapping && FULL) ? envColor(sc, f, rd) : mk(0, 0, 0);
                phase = PH_RETURN;
            } else {
                Ro = ro, Rd = rd, Rt = t;
                hp = Ro + Rt * Rd;
                hitN = hit.normal;
                const rtx_material& m = sc.mats[hit.material];
                ks = mk(m.ks);
                shin = m.shininess;
                if (!f.enable_shading) {
                    Lo = mk(m.kd);
                    phase = PH_AFTER_LIGHTS;
                } else {
                    shadedPending = true;
                    albedo = FULL ? albedoOf(sc, f, m, Ro, Rd, hit) : mk(m.kd);
                    visN = (dot(normalize(hitN), normalize(Ro - hp)) < -1e-6f) ? -hitN : hitN;
                    Lo = mk(0, 0, 0);
                    li = 0;
                    phase = PH_NEXT_LIGHT;
                }
            }
        }

        // advance until the next ray is in flight, the lane waits for the wave to
        // sample an area light, or the sample is finished
        bool launched = phase == PH_NONE;
        while (!launched) {
            if (phase == PH_NEXT_LIGHT) {
                // computeLightContribution, src/light.cpp:139-189
                phase = envLight ? PH_ENV : PH_AFTER_LIGHTS;
                while (li < sc.nLights) {
                    const rtx_light& L = sc.lights[li];
                    if (L.type == RTX_LIGHT_POINT) {
                        const V3 pos = mk(L.p[0]), col = mk(L.c[0]);
                        if (!f.enable_hard_shadow) {
                            Lo = Lo + phong(pos, col, hp, Rd, hitN, albedo, ks, shin);
                            li++;
                            continue;
                        }
                        sh = phong(pos, col, hp, Rd, hitN, albedo, ks, shin);
                        ro = pos;
                        rd = hp - pos;
                        shadowBehind = dot(normalize(pos - hp), normalize(visN)) < -1e-6f;
                        const float len = length(rd);
                        const float scale = fmaxf(fmaxf(fmaxf(fabsf(hp.x), fabsf(hp.y)), fabsf(hp.z)), fmaxf(fmaxf(fabsf(pos.x), fabsf(pos.y)), fabsf(pos.z)));
                        const bool shorten = len >= 0.2f && scale <= 64.0f;
                        shadowLimit = shorten ? 1.01f : kFltMax;
                        shadowStop = shorten ? 1.0f - 4e-3f / len : -__builtin_inff();
                        state = PS_SHADOW;
                        launched = true;
                        break;
                    }
                    if (FULL && f.enable_soft_shadow && A.shadowSamples > 0) {
                        state = PS_AREA; // sampled by the whole wave in the next iteration
                        launched = true;
                        break;
                    }
                    li++; // segment / parallelogram lights only count with soft shadows (src/light.cpp:154,167)
                }
                if (launched)
                    break;
                // all lights done: the environment light, if any, starts from scratch
                envTry = envCand = envCount = 0;
                envAvg = mk(0, 0, 0);
            }
            if (FULL && ENVL && phase == PH_ENV) {
                // sampleEnvironment + EnvironmentMap::getSamplingRay (src/light.cpp:47-66,
                // src/environment_mapping.cpp:107-128): up to 100 tries, each a random radiance
                // bin sampled on a small multi-jittered grid; rays leaving the surface are kept
                // until there are at least shadowSamples of them.  Two keyed draws per try:
                // the bin, then the seed of the grid's engine.
                const V3 envN = dot(hitN, Rd) < 0 ? hitN : -hitN;
                for (;;) {
                    V3 dir = envN, org = hp;
                    bool have = false, finished = false;
                    if (sc.env.texture < 0) {
                        // constant-colour environment: one ray along the normal
                        finished = envTry != 0;
                        have = !finished;
                        envTry = 1;
                    } else if (envCand == 0 && !(envCount < nS && envTry < 100)) {
                        finished = true;
                        counter += 2u * uint32_t(envTry);
                    } else {
                        const float binDraw = keyedUniform(rngKey, counter + 2u * uint32_t(envTry));
                        const int bin = min(int(floorf(binDraw * float(sc.nEnvBins))), sc.nEnvBins - 1);
                        float sx, sy;
                        sample2DAt(sc.envBins[bin], envPerTry, rngKey, counter + 2u * uint32_t(envTry) + 1u, envCand, sx, sy);
                        dir = envDirection(sc, sx, sy);
                        const float eps = 0.0001f / dot(normalize(dir), envN);
                        org = hp + eps * dir;
                        have = dot(dir, envN) > 0.f;
                        if (++envCand == envPerTry * envPerTry) {
                            envCand = 0;
                            envTry++;
                        }
                        if (have)
                            envCount++;
                    }
                    if (finished) {
                        Lo = Lo + envAvg * float(1.0 / double(float(nS)));
                        phase = PH_AFTER_LIGHTS;
                        break;
                    }
                    if (!have)
                        continue;
                    if (f.enable_hard_shadow) {
                        ro = org;
                        rd = dir;
                        state = PS_ENV;
                        launched = true;
                        break;
                    }
                    envAvg = envAvg + phong(org + 100000.f * dir, envColor(sc, f, dir), hp, Rd, envN, albedo, ks, shin);
                }
                if (launched)
                    break;
            }
            if (phase == PH_AFTER_LIGHTS) {
                // src/render.cpp:22-52
                const bool bounce = FULL && f.enable_recursive && depthLeft > 0 && !(ks.x == 0.f && ks.y == 0.f && ks.z == 0.f) && level < kMaxDepth;
                if (!bounce) {
                    value = Lo;
                    phase = PH_RETURN;
                } else {
                    Frame& F = frames[level];
                    F.Lo = Lo;
                    F.ks = ks;
                    F.acc = mk(0, 0, 0);
                    F.count = 0;
                    F.child = 0;
                    const V3 r = reflectDir(Rd, hitN);
                    F.reflO = (Ro + Rd * Rt) + float(1e-5) * r;
                    F.reflD = r;
                    F.nrm = normalize(hitN);
                    if (f.enable_glossy_reflection) {
                        // the whole fan is drawn before any of it is traced
                        // (src/shading.cpp:98-105): reserve its draws now
                        F.nChildren = A.glossyRays;
                        F.blur = 1.0f / shin;
                        F.rngBase = counter;
                        counter += 2u * uint32_t(A.glossyRays);
                    } else {
                        F.nChildren = 1;
                        F.blur = 0.f;
                        F.rngBase = 0;
                    }
                    F.traced = 0;
                    level++;
                    depthLeft--;
                    phase = PH_NEXT_CHILD;
                    if (FAN && f.enable_glossy_reflection && F.nChildren > 1 && F.nChildren <= kFanMax) {
                        state = PS_FAN; // traced by the whole wave in the next iteration
                        launched = true;
                        break;
                    }
                }
            }
            if (phase == PH_NEXT_CHILD) {
                Frame& F = frames[level - 1];
                phase = PH_RETURN;
                bool exhausted = true;
                while (F.child < F.nChildren) {
                    const int k = F.child++;
                    V3 dir = F.reflD;
                    if (f.enable_glossy_reflection) {
                        // returnGlossyRay, src/shading.cpp:67-88
                        const V3 w = normalize(F.reflD);
                        const V3 tt = normalize(w - mk(0.1f, 0.0f, 0.0f));
                        const V3 u = normalize(cross(tt, w));
                        const V3 v = normalize(cross(w, u));
                        const float ua = -F.blur / 2.0f + F.blur * keyedUniform(rngKey, F.rngBase + 2u * uint32_t(k));
                        const float va = -F.blur / 2.0f + F.blur * keyedUniform(rngKey, F.rngBase + 2u * uint32_t(k) + 1u);
                        dir = F.reflD + (ua * u + va * v);
                    }
                    const float cosinus = dot(normalize(dir), F.nrm);
                    const float sinus = __builtin_sqrtf(1.0f - cosinus * cosinus);
                    if (cosinus > 0.0f && sinus > 0.0f) {
                        ro = F.reflO;
                        rd = dir;
                        if (FAN && F.traced) {
                            preTraced = true;
                            preT = fanT[level - 1][k];
                            preK = fanK[level - 1][k];
                            preSphere = fanSphere[level - 1][k];
                            preSphereN = fanSphereN[level - 1][k];
                        }
                        state = PS_RADIANCE;
                        launched = true;
                        exhausted = false;
                        break;
                    }
                }
                if (launched)
                    break;
                if (exhausted) {
                    // color *= 1/count (0 * inf = NaN when no child was traced), Lo + color * ks
                    const V3 color = F.acc * (1.0f / float(F.count));
                    value = F.Lo + color * F.ks;
                    level--;
                    depthLeft++;
                }
            }
            if (phase == PH_RETURN) {
                if (level > 0) {
                    Frame& F = frames[level - 1];
                    F.acc = F.acc + value;
                    F.count++;
                    phase = PH_NEXT_CHILD;
                    continue;
                }
                if (transparencyDriver && !lastLayer) {
                    if (nLayers < kMaxLayers) {
                        layerAlpha[nLayers] = pendingAlpha;
                        layerColor[nLayers] = value;
                    } else {
                        tailSum = tailSum + (tailThrough * pendingAlpha) * value;
                        tailThrough = tailThrough * (1.0f - pendingAlpha);
                    }
                    nLayers++;
                    ro = nextLayerO;
                    rd = camD;
                    state = PS_RADIANCE;
                    launched = true;
                    break;
                }
                if (transparencyDriver) {
                    if (nLayers > kMaxLayers)
                        value = tailSum + tailThrough * value;
                    // transparency * front + (1 - transparency) * behind, innermost first
                    for (int i = min(nLayers, kMaxLayers) - 1; i >= 0; i--)
                        value = layerAlpha[i] * layerColor[i] + (1.0f - layerAlpha[i]) * value;
                }
                result = value;
                state = PS_DONE;
                launched = true;
            }
        }
    }
    lc.countShaded(shadedPending);
    if (FULL)
        lc.countOverflow(overflowPending);
    return result;
}

// Visibility of point light li for the wave's shading points, the shadow rays walking the tree together: by the
// light's origin frame if the scene has one for it (made at upload for the first kLightFrames point lights; the
// choice is the same for the whole wave), which answers exactly as the scene's own records do.
template <int STACK, bool COUNT>
RTX_DEV float pointLightVisibilityWave(const DScene& sc, LaneStack<STACK>& st, const rtx_features& f, bool facing, V3 pos, V3 hp, int li, TraceCounters& tc)
{
    if (li < kLightFrames) {
        const OriginFrame fr = li == 0 ? sc.lightFrame[0] : li == 1 ? sc.lightFrame[1] : li == 2 ? sc.lightFrame[2] : sc.lightFrame[3];
        if (fr.nodes)
            return lightVisibilityWave<STACK, COUNT, false, true>(sc, st, f, facing, pos, hp, tc, fr);
    }
    return lightVisibilityWave<STACK, COUNT>(sc, st, f, facing, pos, hp, tc);
}

// The same sample when nothing but point lights, Phong shading and hard shadows
// are in play (no reflection, area light, texture, environment or transparency):
// straight-line code with one traversal for the camera ray and one per point
// light, which keeps far fewer values alive across the traversal loops than the
// general state machine above.
// TOGETHER: the wave's rays walk the tree together (coherent bundles: the samples of neighbouring
// pixels from one camera position); otherwise lane by lane (the depth-of-field driver, whose
// samples start all over the lens and cross at the focal point: walked together such a wave
// visits the union of its lanes' nodes, 32 ms instead of 10 ms on the 10 M-triangle scene).
// REL: the camera rays all start at the camera position and every light that counts is a point
// light, and the frames of those origins are at hand (A.camFrame, sc.lightFrame).
template <int STACK, bool COUNT, bool TOGETHER = true, bool REL = false>
RTX_DEV V3 tracePathLean(const RenderArgs& A, LaneStack<STACK>& st, V3 o, V3 d, bool active, LaneCounters& lc)
{
    const DScene& sc = A.sc;
    const rtx_features& f = A.f;
    V3 result = mk(0, 0, 0); // a miss is black without environment mapping (src/render.cpp:61-64)
    float t = kFltMax;
    Hit hit;
    bool any = false;
    lc.countPrimary(active);
    lc.countTraced(active);
    if (TOGETHER)
        any = intersectSceneWave<STACK, COUNT, REL>(sc, st, f, active, o, d, t, hit, lc.tc, A.camFrame);
    else if (active)
        any = intersectScene<STACK, COUNT>(sc, st, f, o, d, t, hit, lc.tc);
    const bool shade = any && f.enable_shading;
    lc.countShaded(shade);
    if (!anyLane(any))
        return result; // no camera ray of the wave hit anything: every sample is black, no light needs a look
    V3 hp = mk(0, 0, 0), hitN = mk(0, 0, 0), albedo = mk(0, 0, 0), ks = mk(0, 0, 0), visN = mk(0, 0, 0);
    float shin = 1.f;
    if (any) {
        const rtx_material& m = sc.mats[hit.material];
        result = mk(m.kd); // shading disabled: the albedo (src/light.cpp:184-188)
        hp = o + t * d;
        hitN = hit.normal;
        albedo = mk(m.kd);
        ks = mk(m.ks);
        shin = m.shininess;
        // testVisibilityLightSample's viewer-side normal (src/light.cpp:72-77)
        visN = (dot(normalize(hitN), normalize(o - hp)) < -1e-6f) ? -hitN : hitN;
    }
    if (shade)
        result = mk(0, 0, 0);
    auto light = [&](int li) {
        const rtx_light& L = sc.lights[li];
        if (L.type != RTX_LIGHT_POINT)
            return; // segment / parallelogram lights only count with soft shadows (src/light.cpp:154,167)
        const V3 pos = mk(L.p[0]);
        V3 sh = mk(0, 0, 0);
        if (shade)
            sh = phong(pos, mk(L.c[0]), hp, d, hitN, albedo, ks, shin);
        if (f.enable_hard_shadow) {
            // src/light.cpp:70-102
            lc.countShadow(shade);
            const bool facing = shade && lightFacesSurface(pos, hp, visN);
            lc.countTraced(facing);
            float visible = 0.0f;
            if (TOGETHER && REL) // (the shadow rays start at the light: its frame, which the host has checked)
                visible = lightVisibilityWave<STACK, COUNT, false, true>(sc, st, f, facing, pos, hp, lc.tc,
                    li == 0 ? sc.lightFrame[0] : li == 1 ? sc.lightFrame[1] : li == 2 ? sc.lightFrame[2] : sc.lightFrame[3]);
            else if (TOGETHER)
                visible = pointLightVisibilityWave<STACK, COUNT>(sc, st, f, facing, pos, hp, li, lc.tc);
            // (lane-by-lane frames keep their shadow rays lane by lane: walked together through the light's frame they
            // win where triangles are large on the screen, 2.31 -> 2.12 ms for 16 lens samples on the headline scene, and
            // lose where they are not, 5.2 -> 8.1 ms for 4 lens samples on the 10 M-triangle scene)
            else if (facing)
                visible = lightVisibility<STACK, COUNT>(sc, st, f, pos, hp, visN, lc.tc);
            if (shade)
                sh = sh * (facing ? visible : 0.0f);
        }
        if (shade)
            result = result + sh;
    };
    for (int li = 0; li < sc.nLights; li++)
        light(li);
    return result;
}

// computeLightContribution for point lights only (src/light.cpp:139-189 with :70-102 for the hard shadows), as the
// lean paths use it: the shading point (hp, seen along rd), its material, the viewer-side normal.  Every lane of
// the wave calls this together; `shade`: the lane has a point to shade.  TOGETHER: the shadow rays of a light
// walk the tree together (they all start at the light), else lane by lane.
template <int STACK, bool COUNT, bool TOGETHER>
RTX_DEV V3 pointLightsLean(const DScene& sc, LaneStack<STACK>& st, const rtx_features& f, bool shade, V3 hp, V3 rd, V3 hitN, V3 albedo, V3 ks, float shin, V3 visN, LaneCounters& lc)
{
    V3 Lo = mk(0, 0, 0);
    for (int li = 0; li < sc.nLights; li++) {
        const rtx_light& L = sc.lights[li];
        if (L.type != RTX_LIGHT_POINT)
            continue; // segment / parallelogram lights only count with soft shadows (src/light.cpp:154,167)
        const V3 pos = mk(L.p[0]);
        V3 sh = mk(0, 0, 0);
        if (shade)
            sh = phong(pos, mk(L.c[0]), hp, rd, hitN, albedo, ks, shin);
        if (f.enable_hard_shadow) {
            lc.countShadow(shade);
            const bool facing = shade && lightFacesSurface(pos, hp, visN);
            lc.countTraced(facing);
            float visible = 0.0f;
            if (TOGETHER)
                visible = pointLightVisibilityWave<STACK, COUNT>(sc, st, f, facing, pos, hp, li, lc.tc);
            else if (facing)
                visible = lightVisibility<STACK, COUNT>(sc, st, f, pos, hp, visN, lc.tc);
            if (shade)
                sh = sh * (facing ? visible : 0.0f);
        }
        if (shade)
            Lo = Lo + sh;
    }
    return Lo;
}

// Whether the wave's rays (`lane`: the lane has one, starting at o) may be walked by the camera position's
// origin frame: there is one and every ray starts exactly there.
RTX_DEV bool fromCamera(const RenderArgs& A, bool lane, V3 o)
{
    return A.camFrame.nodes != nullptr && !anyLane(lane && !(o.x == A.cam.position[0] && o.y == A.cam.position[1] && o.z == A.cam.position[2]));
}

// getFinalColor (src/render.cpp:16-66) with ONE level of mirror reflection under point lights: what the
// reference's own drivers ask for when `recursive` is on (they pass depth 1 or 0, src/render.cpp:83,147,247)
// and no glossy fan, area light, texture or environment map is in play.  Straight-line code like
// tracePathLean -- the camera rays and their shadow rays walk the tree together, then the lanes whose
// surface reflects (ks != 0, direction passes the test of src/render.cpp:31-38) trace their one child and its
// shadow rays lane by lane (reflected rays of neighbouring pixels leave a curved surface every which way) --
// so that a frame does not pay for the general state machine's 256 registers because a switch is on.
// The arithmetic is tracePath's, term by term: child value added onto a zero, times 1 / count, times ks.
template <int STACK, bool COUNT>
RTX_DEV V3 tracePathLeanMirror(const RenderArgs& A, LaneStack<STACK>& st, V3 o, V3 d, bool active, LaneCounters& lc)
{
    const DScene& sc = A.sc;
    const rtx_features& f = A.f;
    float t = kFltMax;
    Hit hit;
    lc.countPrimary(active);
    lc.countTraced(active);
    // (A.camFrame: the host makes one when every camera ray of the frame starts at the camera position)
    const bool any = fromCamera(A, active, o) ? intersectSceneWave<STACK, COUNT, true>(sc, st, f, active, o, d, t, hit, lc.tc, A.camFrame) : intersectSceneWave<STACK, COUNT>(sc, st, f, active, o, d, t, hit, lc.tc);
    const bool shade = any && f.enable_shading;
    lc.countShaded(shade);
    if (!anyLane(any))
        return mk(0, 0, 0); // a miss is black without environment mapping (src/render.cpp:61-64)
    V3 hp = mk(0, 0, 0), hitN = mk(0, 0, 0), albedo = mk(0, 0, 0), ks = mk(0, 0, 0), visN = mk(0, 0, 0), Lo = mk(0, 0, 0);
    float shin = 1.f;
    if (any) {
        const rtx_material& m = sc.mats[hit.material];
        hp = o + t * d;
        hitN = hit.normal;
        albedo = mk(m.kd);
        ks = mk(m.ks);
        shin = m.shininess;
        visN = (dot(normalize(hitN), normalize(o - hp)) < -1e-6f) ? -hitN : hitN;
        Lo = mk(m.kd); // shading disabled: the albedo (src/light.cpp:184-188)
    }
    const V3 lit = pointLightsLean<STACK, COUNT, true>(sc, st, f, shade, hp, d, hitN, albedo, ks, shin, visN, lc);
    if (shade)
        Lo = lit;
    // src/render.cpp:22-52
    const bool bounce = any && f.enable_recursive && A.depth > 0 && !(ks.x == 0.f && ks.y == 0.f && ks.z == 0.f);
    if (!anyLane(bounce))
        return Lo;
    const V3 r = reflectDir(d, hitN);
    const V3 childO = (o + d * t) + float(1e-5) * r;
    const float cosinus = dot(normalize(r), normalize(hitN));
    const float sinus = __builtin_sqrtf(1.0f - cosinus * cosinus);
    const bool followed = bounce && cosinus > 0.0f && sinus > 0.0f;
    V3 child = mk(0, 0, 0);
    lc.countBounce(followed);
    lc.countTraced(followed);
    {
        float ct = kFltMax;
        Hit ch;
        bool cAny = false;
        if (followed)
            cAny = intersectScene<STACK, COUNT>(sc, st, f, childO, r, ct, ch, lc.tc);
        const bool cShade = cAny && f.enable_shading;
        lc.countShaded(cShade);
        V3 chp = mk(0, 0, 0), cN = mk(0, 0, 0), cAlbedo = mk(0, 0, 0), cKs = mk(0, 0, 0), cVisN = mk(0, 0, 0);
        float cShin = 1.f;
        if (cAny) {
            const rtx_material& m = sc.mats[ch.material];
            chp = childO + ct * r;
            cN = ch.normal;
            cAlbedo = mk(m.kd);
            cKs = mk(m.ks);
            cShin = m.shininess;
            cVisN = (dot(normalize(cN), normalize(childO - chp)) < -1e-6f) ? -cN : cN;
            child = mk(m.kd);
        }
        if (anyLane(cShade)) {
            const V3 cLit = pointLightsLean<STACK, COUNT, false>(sc, st, f, cShade, chp, r, cN, cAlbedo, cKs, cShin, cVisN, lc);
            if (cShade)
                child = cLit;
        }
    }
    if (!bounce)
        return Lo;
    // color += child; color *= 1 / count (0 * inf = NaN when the child was not followed); Lo + color * ks
    const V3 acc = followed ? mk(0, 0, 0) + child : mk(0, 0, 0);
    const V3 color = acc * (1.0f / float(followed ? 1 : 0));
    return Lo + color * ks;
}

// calculateColorTransparency (src/transparency.cpp:6-20) around the lean path: the transparency chain of a
// camera sample under point lights, no reflection.  Layer after layer -- trace, shade, and where the surface is
// translucent keep (alpha, colour) and go on from just behind it (src/transparency.cpp:15) -- with the lanes
// whose chain has ended sitting the later layers out; then the fold of tracePath, innermost first, the first
// kMaxLayers layers kept one by one and deeper ones summed front to back.  Straight-line code at the lean
// path's registers; the layers live in the lane's scratch memory, touched once per layer.  It pays where the
// lanes of a wave end their chains together: the samples of one pixel (a jittered, lens or time-sampled frame);
// 64 different pixels at one sample each lose lanes layer by layer and stay with the general state machine,
// whose lanes in different layers share one traversal.
template <int STACK, bool COUNT, bool TOGETHER>
RTX_DEV V3 tracePathLeanLayers(const RenderArgs& A, LaneStack<STACK>& st, V3 o, V3 d, bool active, LaneCounters& lc)
{
    const DScene& sc = A.sc;
    const rtx_features& f = A.f;
    float layerAlpha[kMaxLayers];
    V3 layerColor[kMaxLayers];
    int nLayers = 0;
    float tailThrough = 1.0f;
    V3 tailSum = mk(0, 0, 0), last = mk(0, 0, 0), from = o;
    bool going = active, firstLayer = true;
    while (anyLane(going)) {
        float t = kFltMax;
        Hit hit;
        bool any = false;
        lc.countPrimary(going);
        lc.countTraced(going);
        if (TOGETHER && firstLayer && fromCamera(A, going, from)) // (the chain's first ray starts at the camera position: its frame, if the host made one)
            any = intersectSceneWave<STACK, COUNT, true>(sc, st, f, going, from, d, t, hit, lc.tc, A.camFrame);
        else if (TOGETHER)
            any = intersectSceneWave<STACK, COUNT>(sc, st, f, going, from, d, t, hit, lc.tc);
        else if (going)
            any = intersectScene<STACK, COUNT>(sc, st, f, from, d, t, hit, lc.tc);
        firstLayer = false;
        const bool shade = any && f.enable_shading;
        lc.countShaded(shade);
        V3 hp = mk(0, 0, 0), hitN = mk(0, 0, 0), albedo = mk(0, 0, 0), ks = mk(0, 0, 0), visN = mk(0, 0, 0), value = mk(0, 0, 0);
        float shin = 1.f, alpha = 1.0f;
        if (any) {
            const rtx_material& m = sc.mats[hit.material];
            hp = from + t * d;
            hitN = hit.normal;
            albedo = mk(m.kd);
            ks = mk(m.ks);
            shin = m.shininess;
            visN = (dot(normalize(hitN), normalize(from - hp)) < -1e-6f) ? -hitN : hitN;
            value = mk(m.kd); // shading disabled: the albedo (src/light.cpp:184-188)
            alpha = m.transparency;
        }
        if (anyLane(shade)) {
            const V3 lit = pointLightsLean<STACK, COUNT, TOGETHER>(sc, st, f, shade, hp, d, hitN, albedo, ks, shin, visN, lc);
            if (shade)
                value = lit;
        }
        const bool more = going && any && alpha != 1.0f;
        lc.countOverflow(more && nLayers == kMaxLayers); // (counted once per camera sample: reported in rtx_stats)
        if (more) {
            if (nLayers < kMaxLayers) {
                layerAlpha[nLayers] = alpha;
                layerColor[nLayers] = value;
            } else {
                tailSum = tailSum + (tailThrough * alpha) * value;
                tailThrough = tailThrough * (1.0f - alpha);
            }
            nLayers++;
            const float adv = float(double(t) + 1e-6);
            from = from + mk(adv, adv, adv) * d;
        } else if (going) {
            last = value;
        }
        going = more;
    }
    V3 value = last;
    if (nLayers > kMaxLayers)
        value = tailSum + tailThrough * value;
    // transparency * front + (1 - transparency) * behind, innermost first
    for (int i = min(nLayers, kMaxLayers) - 1; i >= 0; i--)
        value = layerAlpha[i] * layerColor[i] + (1.0f - layerAlpha[i]) * value;
    return value;
}

// ---- pass 1 of a deferred frame: the walk ---------------------------------------------
// A lane's tape writer: entries go into chunks of kTapeChunk taken from the wave's pool.
struct TapeWriter {
    uint32_t at = 0, left = 0; // next entry (index into the tape), free entries of the current chunk
    uint32_t head = 0;         // what goes into heads[]: 0 while nothing was written
    bool failed = false;       // the pool ran over: the frame falls back to the one-kernel path
};
RTX_DEV void tapeEmit(const Deferred& D, TapeWriter& w, uint32_t op, uint32_t point, V3 a, V3 ks)
{
    if (w.failed)
        return;
    if (w.left <= 1) {
        const uint32_t pool = waveIndex() % kPools;
        const uint32_t k = atomicAdd(D.counters + pool * kPoolStride, 1u);
        if (k >= D.tapePerPool) {
            w.failed = true;
            atomicOr(deferOverflow(D), 1u);
            return;
        }
        const uint32_t chunk = pool * D.tapePerPool + k;
        if (w.left == 1) {
            TapeEntry link {};
            link.op = TAPE_LINK;
            link.point = chunk;
            D.tape[w.at] = link;
        } else {
            w.head = chunk + 1;
        }
        w.at = chunk * kTapeChunk;
        w.left = kTapeChunk;
    }
    TapeEntry e;
    e.op = op;
    e.point = point;
    e.a[0] = a.x, e.a[1] = a.y, e.a[2] = a.z;
    e.ks[0] = ks.x, e.ks[1] = ks.y, e.ks[2] = ks.z;
    D.tape[w.at++] = e;
    w.left--;
}

struct WalkFrame { // one getFinalColor call whose children are being followed
    V3 reflO, reflD, nrm;
    float blur;
    uint32_t rngBase;
    int32_t child, nChildren, traced;
};
constexpr int32_t kNotFollowed = -2; // FanStore::k of a fan's ray whose direction fails the test of src/render.cpp:31-38
// Hits of the glossy fans traced ahead, in LDS: entry (level, ray) of the workgroup's path `slot` at
// [(level * rays + ray) * slots + slot].  Only the lanes that carry a camera sample hold fans (one lane of a wave
// where a unit is one pixel), so the store has a slot per such lane, not per lane: a workgroup's share of LDS
// then leaves the number of waves a CU holds to the kernel's registers.
struct FanStore {
    float* t;
    int32_t* k;
    int slot, slots, rays; // rays: the fan size
    unsigned long long* best; // this wave's 64 words for traverseFrontier
    RTX_DEV int at(int level, int ray) const { return (level * rays + ray) * slots + slot; }
};

// One camera sample's radiance rays, depth first; every lane of the wave calls this together.
// The control flow is tracePath's with the light loop taken out: a shaded hit reserves the draws
// its light loop will make (D.drawsPerPoint) and moves on.  Glossy fans of 2..kFanMax rays are
// traced by the wave's lanes side by side as in tracePath (D.fanLevels > 0).
template <int STACK, bool COUNT>
RTX_DEV void walkPath(const RenderArgs& A, LaneStack<STACK>& st, FanStore fan, V3 camO, V3 camD, bool active, uint32_t rngKey, uint32_t headIndex, LaneCounters& lc, uint32_t probeUnit = 0)
{
#ifdef RTX_PROBE
    WalkLaps laps;
#else
    (void)probeUnit;
#endif
    const DScene& sc = A.sc;
    const rtx_features& f = A.f;
    const Deferred& D = A.defer;
    const int lane = lane0();
    const bool fans = D.fanLevels > 0;

    int state = active ? PS_RADIANCE : PS_DONE;
    V3 ro = camO, rd = camD;
    uint32_t counter = 0;
    int level = 0, depthLeft = A.depth;
    WalkFrame frames[kMaxDepth];
    bool preTraced = false;
    float preT = kFltMax;
    int32_t preK = -1;
    if (D.primary && active) { // the camera pass has traced this sample's camera ray already
        const uint2 seen = D.primary[headIndex];
        preTraced = true;
        preT = __uint_as_float(seen.x);
        preK = int32_t(seen.y);
    }
    const int fanGroup = min(max(A.glossyRays, 1), kFanMax), fanPerRound = min(64 / fanGroup, int(D.fansPerRound));
    const int fanSlot = lane / fanGroup, fanChild = lane - fanSlot * fanGroup;
    TapeWriter tape;
    bool shadedPending = false;
    uint32_t missedAhead = 0; // children settled in the advance loop (see there), still to be counted as bounce rays
    // A frame's walk ends on its deepest samples, each a chain of fans traced one after the other by a wave that
    // shares its SIMD with a wave of short units: once a sample has shown itself deep (a fan below the second
    // level) its wave asks for the SIMD's issue slots first (measured: a 156-fan sample runs 35 us per fan beside
    // another wave, 20 us alone).
    bool raised = false;

    while (__any(state != PS_DONE)) {
        RTX_LAP(advance);
        if (!raised && anyLane(state == PS_FAN && level >= 3)) {
            __builtin_amdgcn_s_setprio(3);
            raised = true;
        }
        while (anyLane(missedAhead != 0u)) { // (the ray counters are the wave's: counted where the wave is converged)
            lc.countBounce(missedAhead != 0u);
            missedAhead -= missedAhead != 0u ? 1u : 0u;
        }
        const uint64_t pendingFan = fans ? __ballot(state == PS_FAN) : 0ull;
        const bool coopFan = pendingFan != 0;
        lc.countShaded(shadedPending);
        shadedPending = false;

        V3 qo = ro, qd = rd;
        float t = kFltMax;
        bool trace = false;
        uint64_t served = 0;
        // fan iteration: the key and the counter of the lane this lane traces for, and whether
        // that fan's rays are the last level of the recursion
        uint32_t workerKey = 0, workerCounter = 0;
        bool workerLeaf = false;
        if (coopFan) {
            // lane = one ray of a parked lane's fan (returnGlossyRay, src/shading.cpp:67-88)
            int mine = -1;
            uint64_t m = pendingFan;
            for (int slot = 0; slot < fanPerRound && m; slot++) {
                const int p = __builtin_ctzll(m);
                m &= m - 1;
                served |= 1ull << p;
                if (fanSlot == slot)
                    mine = p;
            }
            const int src = mine >= 0 ? mine : 0;
            const WalkFrame& own = frames[level > 0 ? level - 1 : 0]; // (a parked lane's fan is its innermost frame)
            const V3 fD = laneValue(own.reflD, src), fO = laneValue(own.reflO, src), fN = laneValue(own.nrm, src);
            const float fBlur = laneValue(own.blur, src);
            const uint32_t fBase = uint32_t(__shfl(int(own.rngBase), src, 64)), fKey = uint32_t(__shfl(int(rngKey), src, 64));
            const int fCount = __shfl(own.nChildren, src, 64);
            workerKey = fKey;
            workerCounter = uint32_t(__shfl(int(counter), src, 64));
            // (every lane takes part in the exchange: a lane that sat it out could not be read from)
            const int fDepthLeft = __shfl(depthLeft, src, 64);
            workerLeaf = mine >= 0 && fDepthLeft == 0 && f.enable_shading && D.leafFans; // the fan's rays end the recursion (src/render.cpp:22)
            if (mine >= 0 && fanChild < fCount) {
                const V3 w = normalize(fD);
                const V3 tt = normalize(w - mk(0.1f, 0.0f, 0.0f));
                const V3 u = normalize(cross(tt, w));
                const V3 v = normalize(cross(w, u));
                const float ua = -fBlur / 2.0f + fBlur * keyedUniform(fKey, fBase + 2u * uint32_t(fanChild));
                const float va = -fBlur / 2.0f + fBlur * keyedUniform(fKey, fBase + 2u * uint32_t(fanChild) + 1u);
                const V3 dir = fD + (ua * u + va * v);
                const float cosinus = dot(normalize(dir), fN);
                const float sinus = __builtin_sqrtf(1.0f - cosinus * cosinus);
                qo = fO;
                qd = dir;
                trace = cosinus > 0.0f && sinus > 0.0f; // the others are never followed (src/render.cpp:31-38)
            }
        } else {
            lc.countPrimary(state == PS_RADIANCE && level == 0);
            lc.countBounce(state == PS_RADIANCE && level != 0);
            trace = state == PS_RADIANCE && !preTraced;
        }

        Hit hit;
        bool any = false;
        int bestK = -1, sphere = -1;
        V3 sphereN = mk(0, 0, 0);
        lc.countTraced(trace);
        RTX_LAP(make);
#ifdef RTX_PROBE
        (coopFan ? laps.p.fanTrips : laps.p.otherTrips)++;
#endif
        bool walked = false;
        if (coopFan && D.frontier) { // (no spheres in a frame whose fans are traced ahead)
            walked = traverseFrontier<STACK, COUNT>(sc, st, fan.best, uint32_t(__popcll(served)) * uint32_t(fanGroup), trace, qo, qd, t, bestK, lc.tc);
            any = trace && bestK >= 0;
        }
        // (otherwise, and for the rays that are not a fan's, lane by lane; the wave-shared walk is kept out of this
        // kernel: a deep sample's wave runs alone on its SIMD and pays for every line of code it has to fetch)
        if (!walked && trace)
            any = closestHit<STACK, COUNT>(sc, st, f, qo, qd, t, bestK, sphere, sphereN, lc.tc);
        if (!coopFan && state == PS_RADIANCE && preTraced) {
            t = preT;
            bestK = preK;
            any = bestK >= 0;
            preTraced = false;
        }
        RTX_LAP(walk);
        const bool landed = !coopFan && state == PS_RADIANCE;
        // The rays of a fan that ends the recursion are whole getFinalColor calls by themselves
        // (hit or miss, the lights, no children): the lanes that traced them make their shading
        // points on the spot, all at once, and the parked lane only writes the tape.  The light
        // draws of ray j start after those of the fan's earlier rays that hit something.
        const bool leafHit = coopFan && workerLeaf && trace && any;
        if (landed)
            hitAttributes(sc, f, ro, rd, t, bestK, sphere, sphereN, hit);
        else if (leafHit)
            hitAttributes(sc, f, qo, qd, t, bestK, sphere, sphereN, hit);
        lc.countBounce(coopFan && workerLeaf && trace);
        RTX_LAP(attributes);

        // numbers for this iteration's shading points: one atomic per wave, the lanes in order
        const bool wantsPoint = (landed || leafHit) && any && f.enable_shading && !tape.failed;
        uint32_t point = kNoPoint;
        {
            const uint64_t want = __ballot(wantsPoint);
            if (want) {
                const uint32_t pool = waveIndex() % kPools, n = uint32_t(__popcll(want));
                uint32_t base = 0;
                if (lane == int(__builtin_ctzll(want)))
                    base = atomicAdd(D.counters + (kPools + pool) * kPoolStride, n);
                base = uint32_t(__builtin_amdgcn_readlane(int(base), int(__builtin_ctzll(want))));
                const uint32_t mine = base + uint32_t(__popcll(want & ((1ull << lane) - 1ull)));
                if (wantsPoint) {
                    if (mine < D.pointsPerPool) {
                        point = pool * D.pointsPerPool + mine;
                    } else {
                        tape.failed = true;
                        atomicOr(deferOverflow(D), 1u);
                    }
                }
            }
        }

        RTX_LAP(numbers);
        int phase = PH_NONE;
        if (coopFan) {
            if (anyLane(workerLeaf)) {
                const uint64_t hits = __ballot(leafHit && f.enable_shading);
                if (leafHit && f.enable_shading) {
                    shadedPending = true; // (counted for the wave, whichever lane reports it)
                    if (point != kNoPoint) {
                        const uint64_t group = ((fanGroup < 64 ? (1ull << fanGroup) : 0ull) - 1ull) << (fanSlot * fanGroup);
                        const uint32_t earlier = uint32_t(__popcll(hits & group & ((1ull << lane) - 1ull)));
                        const V3 hp = qo + t * qd;
                        const rtx_material& m = sc.mats[hit.material];
                        const V3 albedo = albedoOf(sc, f, m, qo, qd, hit);
                        ShadePoint P;
                        P.hp[0] = hp.x, P.hp[1] = hp.y, P.hp[2] = hp.z;
                        P.rd[0] = qd.x, P.rd[1] = qd.y, P.rd[2] = qd.z;
                        P.n[0] = hit.normal.x, P.n[1] = hit.normal.y, P.n[2] = hit.normal.z;
                        P.albedo[0] = albedo.x, P.albedo[1] = albedo.y, P.albedo[2] = albedo.z;
                        P.ks[0] = m.ks[0], P.ks[1] = m.ks[1], P.ks[2] = m.ks[2];
                        P.shin = m.shininess;
                        P.key = workerKey;
                        P.counter = workerCounter + earlier * D.drawsPerPoint;
                        P.flipped = (dot(normalize(hit.normal), normalize(qo - hp)) < -1e-6f) ? 1u : 0u;
                        P.pad = 0;
                        D.points[point] = P;
                    }
                }
            }
            // the k-th served lane finds ray j of its fan in lane k * fanGroup + j
            const int k = __popcll(served & ((1ull << lane) - 1ull));
            const bool mineServed = (served >> lane) & 1ull;
            const int lv = level > 0 ? level - 1 : 0;
            const bool myLeaf = mineServed && depthLeft == 0 && f.enable_shading && D.leafFans;
            // what a ray of a last-level fan leaves on the tape: 0 not followed, 1 a miss, 2 a hit
            const int outcome = !trace ? 0 : !any ? 1 : 2;
            for (int j = 0; j < fanGroup; j++) {
                const int from = min(k * fanGroup + j, 63);
                const float jt = __shfl(t, from, 64);
                const int jk = __shfl(trace ? bestK : kNotFollowed, from, 64);
                const int jOutcome = __shfl(outcome, from, 64);
                const uint32_t jPoint = uint32_t(__shfl(int(point), from, 64));
                const int jFailed = __shfl(int(tape.failed), from, 64);
                if (myLeaf && j < frames[lv].nChildren) {
                    if (jFailed)
                        tape.failed = true;
                    if (jOutcome == 1) {
                        tapeEmit(D, tape, TAPE_MISS, kNoPoint, mk(0, 0, 0), mk(0, 0, 0)); // (no environment map in a deferred frame)
                    } else if (jOutcome == 2) {
                        tapeEmit(D, tape, TAPE_HIT, jPoint, mk(0, 0, 0), mk(0, 0, 0));
                        counter += D.drawsPerPoint;
                    }
                } else if (mineServed) {
                    fan.t[fan.at(lv, j)] = jt;
                    fan.k[fan.at(lv, j)] = jk;
                }
            }
            if (myLeaf) {
                // every child has returned
                tapeEmit(D, tape, TAPE_CLOSE, kNoPoint, mk(0, 0, 0), mk(0, 0, 0));
                level--;
                depthLeft++;
                state = PS_RADIANCE; // left again below
                phase = PH_RETURN;
            } else if (mineServed) {
                frames[lv].traced = 1;
                state = PS_RADIANCE; // left again below
                phase = PH_NEXT_CHILD;
            }
        } else if (state == PS_RADIANCE) {
            if (!any) {
                // src/render.cpp:61-64
                const V3 value = f.enable_environment_mapping ? envColor(sc, f, rd) : mk(0, 0, 0);
                if (!(level == 0 && tape.head == 0 && value.x == 0.f && value.y == 0.f && value.z == 0.f)) // (a black camera sample needs no tape)
                    tapeEmit(D, tape, TAPE_MISS, kNoPoint, value, mk(0, 0, 0));
                phase = PH_RETURN;
            } else {
                const V3 hp = ro + t * rd, hitN = hit.normal;
                const rtx_material& m = sc.mats[hit.material];
                const V3 ks = mk(m.ks);
                const float shin = m.shininess;
                V3 before = mk(m.kd); // shading off: the albedo is the light contribution (src/light.cpp:184-188)
                if (f.enable_shading) {
                    shadedPending = true;
                    before = mk(0, 0, 0);
                    if (point != kNoPoint) {
                        const V3 albedo = albedoOf(sc, f, m, ro, rd, hit);
                        ShadePoint P;
                        P.hp[0] = hp.x, P.hp[1] = hp.y, P.hp[2] = hp.z;
                        P.rd[0] = rd.x, P.rd[1] = rd.y, P.rd[2] = rd.z;
                        P.n[0] = hitN.x, P.n[1] = hitN.y, P.n[2] = hitN.z;
                        P.albedo[0] = albedo.x, P.albedo[1] = albedo.y, P.albedo[2] = albedo.z;
                        P.ks[0] = ks.x, P.ks[1] = ks.y, P.ks[2] = ks.z;
                        P.shin = shin;
                        P.key = rngKey;
                        P.counter = counter;
                        // testVisibilityLightSample's viewer-side normal (src/light.cpp:72-77)
                        P.flipped = (dot(normalize(hitN), normalize(ro - hp)) < -1e-6f) ? 1u : 0u;
                        P.pad = 0;
                        D.points[point] = P;
                    }
                    counter += D.drawsPerPoint;
                }
                // src/render.cpp:22-52
                const bool bounce = f.enable_recursive && depthLeft > 0 && !(ks.x == 0.f && ks.y == 0.f && ks.z == 0.f) && level < kMaxDepth;
                if (!bounce) {
                    tapeEmit(D, tape, TAPE_HIT, point, before, mk(0, 0, 0));
                    phase = PH_RETURN;
                } else {
                    tapeEmit(D, tape, TAPE_OPEN, point, before, ks);
                    WalkFrame& F = frames[level];
                    F.child = 0;
                    const V3 r = reflectDir(rd, hitN);
                    F.reflO = (ro + rd * t) + float(1e-5) * r;
                    F.reflD = r;
                    F.nrm = normalize(hitN);
                    if (f.enable_glossy_reflection) {
                        // the whole fan is drawn before any of it is traced (src/shading.cpp:98-105)
                        F.nChildren = A.glossyRays;
                        F.blur = 1.0f / shin;
                        F.rngBase = counter;
                        counter += 2u * uint32_t(A.glossyRays);
                    } else {
                        F.nChildren = 1;
                        F.blur = 0.f;
                        F.rngBase = 0;
                    }
                    F.traced = 0;
                    level++;
                    depthLeft--;
                    phase = PH_NEXT_CHILD;
                    if (fans && f.enable_glossy_reflection && F.nChildren > 1 && F.nChildren <= kFanMax && level <= int(D.fanLevels)) {
                        state = PS_FAN; // traced by the whole wave in the next iteration
                        phase = PH_NONE;
                    }
                }
            }
        }
        if (tape.failed && state != PS_DONE) { // the frame is rendered again by the one-kernel path: stop here
            state = PS_DONE;
            phase = PH_NONE;
        }
        RTX_LAP(records);

        bool launched = phase == PH_NONE;
        while (!launched) {
            if (phase == PH_NEXT_CHILD) {
                WalkFrame& F = frames[level - 1];
                phase = PH_RETURN;
                while (F.child < F.nChildren) {
                    const int k = F.child++;
                    if (F.traced) {
                        // What the lanes that traced the fan found for this child (kNotFollowed: its direction
                        // failed the test of src/render.cpp:31-38).  A child that is not followed, or that hit
                        // nothing, is settled here without its direction being made again: a miss is a whole
                        // getFinalColor call already -- black (src/render.cpp:61-64; a deferred frame has no
                        // environment map), no draws, nothing below it -- and goes on the tape at once instead of
                        // taking a turn of the loop above, which most of a frame's fans, whose rays leave the
                        // mesh, would pay five times.
                        preK = fan.k[fan.at(level - 1, k)];
                        if (preK == kNotFollowed)
                            continue;
                        if (preK < 0) {
                            tapeEmit(D, tape, TAPE_MISS, kNoPoint, mk(0, 0, 0), mk(0, 0, 0));
                            missedAhead++;
                            continue;
                        }
                        preT = fan.t[fan.at(level - 1, k)];
                        preTraced = true;
                    }
                    V3 dir = F.reflD;
                    if (f.enable_glossy_reflection) {
                        // returnGlossyRay, src/shading.cpp:67-88
                        const V3 w = normalize(F.reflD);
                        const V3 tt = normalize(w - mk(0.1f, 0.0f, 0.0f));
                        const V3 u = normalize(cross(tt, w));
                        const V3 v = normalize(cross(w, u));
                        const float ua = -F.blur / 2.0f + F.blur * keyedUniform(rngKey, F.rngBase + 2u * uint32_t(k));
                        const float va = -F.blur / 2.0f + F.blur * keyedUniform(rngKey, F.rngBase + 2u * uint32_t(k) + 1u);
                        dir = F.reflD + (ua * u + va * v);
                    }
                    const float cosinus = dot(normalize(dir), F.nrm);
                    const float sinus = __builtin_sqrtf(1.0f - cosinus * cosinus);
                    if (preTraced || (cosinus > 0.0f && sinus > 0.0f)) { // (a child traced ahead has passed the test)
                        ro = F.reflO;
                        rd = dir;
                        state = PS_RADIANCE;
                        launched = true;
                        break;
                    }
                }
                if (launched)
                    break;
                // every child has returned: color * (1 / count), Lo + color * ks (resolveKernel)
                tapeEmit(D, tape, TAPE_CLOSE, kNoPoint, mk(0, 0, 0), mk(0, 0, 0));
                level--;
                depthLeft++;
            }
            if (phase == PH_RETURN) {
                if (level > 0) {
                    phase = PH_NEXT_CHILD;
                    continue;
                }
                state = PS_DONE;
                launched = true;
            }
        }
    }
    lc.countShaded(shadedPending);
    while (anyLane(missedAhead != 0u)) {
        lc.countBounce(missedAhead != 0u);
        missedAhead -= missedAhead != 0u ? 1u : 0u;
    }
    if (active && !tape.failed)
        D.heads[headIndex] = tape.head;
    if (raised)
        __builtin_amdgcn_s_setprio(0);
#ifdef RTX_PROBE
    RTX_LAP(advance);
    laps.p.rounds = lc.tc.frontierRounds;
    laps.p.entries = uint32_t(waveSum(lc.tc.frontierEntries));
    lc.tc.frontierRounds = lc.tc.frontierEntries = 0;
    if (lane == 0 && probeUnit < kProbeUnits)
        g_walkProbe[probeUnit] = laps.p;
#endif
}

// MODE 0: point lights + hard shadows only (tracePathLean); 1: the general path;
// 2: the general path with environment light sampling; 3: the general path with
// glossy fans traced by the wave's lanes side by side; 4: the walk of a deferred
// frame (walkPath: radiance rays only, no pixel is written); 5: MODE 0 with every ray walking the
// tree on its own
template <int STACK, bool COUNT, int MODE>
__global__ void __launch_bounds__(kBlock, (MODE == 11 || MODE == 12) ? RTX_LAYERS_WAVES : (MODE == 8 && STACK <= kLdsStack) ? RTX_REL_WAVES : (MODE == 0 || MODE >= 5) ? RTX_LEAN_WAVES : MODE == 4 ? RTX_WALK_WAVES : RTX_FULL_WAVES) renderKernel(RenderArgs A) // (mode 10: tracePathLeanMirror; 11 / 12: tracePathLeanLayers, rays together / alone)
{
    constexpr bool FULL = MODE >= 1 && MODE <= 4; // (mode 9, the camera pass, needs none of either kind's machinery beyond ray generation)
    constexpr bool BLUR = FULL || MODE == 6 || MODE == 7 || MODE == 9 || MODE >= 10; // (the motion-blur driver's per-sample cameras: kept out of the other lean kernels' registers)
    // (Mode 8 walks the tree only with the wave-shared walk, whose stack holds at most three entries per
    // two levels of the binary tree -- a four-wide node spans two levels unless its children are leaves --
    // i.e. 24 keys per lane for the 16-level kernels: LDS for five and six workgroups per CU.)
    constexpr int kStackRows = (MODE == 8 && STACK <= kLdsStack) ? (3 * STACK + 1) / 2 : 2 * LaneStack<STACK>::kLds;
    __shared__ int32_t sStack[kStackRows * kBlock];
    __shared__ int32_t sWaveRefs[(kBlock / 64) * 2 * STACK];
    extern __shared__ int32_t sFan[]; // MODE 4: fanLevels * fan size * kBlock distances, then as many triangle records
    __shared__ unsigned long long sBest[MODE == 4 ? kBlock : 1]; // MODE 4: traverseFrontier's best hit per ray
    if (A.runIf && __hip_atomic_load(A.runIf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        return;
    LaneStack<STACK> st { sStack, reinterpret_cast<float*>(sStack + LaneStack<STACK>::kLds * kBlock), int(threadIdx.x), sWaveRefs + (threadIdx.x >> 6) * 2 * STACK };
    LaneCounters lc;
    const int lane = threadIdx.x & 63;
    const int slot = lane / A.lanesPerPixel, sub = lane % A.lanesPerPixel;
    const bool laneUsed = slot < A.pixelsPerUnit;
    // (The composed driver's parameters are read only by the kernels that carry per-sample poses -- the host
    // sends every composed frame to those -- and the other lean kernels go by the driver alone: their
    // traversal loops run at the scalar registers' ceiling, and each kernel argument more that is kept
    // across them costs reloads per trip, 5 % of the headline frame when this was one code path.)
    const int normalise = (BLUR && A.driver == RTX_DRIVER_COMPOSED) ? A.normalise
        : (A.driver == RTX_DRIVER_MRAYS || A.driver == RTX_DRIVER_DOF) ? int(NORM_RECIPROCAL) : A.driver == RTX_DRIVER_MOTION_BLUR ? int(NORM_DIVIDE) : int(NORM_SINGLE);
    const bool averaging = normalise != NORM_SINGLE;
    // the lane that ends up holding its pixel's sum and stores it
    const bool writer = laneUsed && sub == A.lanesPerPixel - 1;

    // An 8x8 block that cannot see the scene: every camera ray misses the scene's
    // bounds, so the rays are counted, the pixels are black (src/render.cpp:61-64) and
    // nothing is traced.  One wave-wide store, lane = pixel.
    auto blackBlock = [&](const BlockPlace& b) {
        if (MODE == 9)
            return; // (the camera pass leaves counting and pixels to the walk)
        const uint32_t mx = uint32_t(lane) & 7u, my = uint32_t(lane) >> 3;
        const bool ok = b.listed && int(b.x0 + mx) < A.W && int(b.y0 + my) < A.H;
        lc.primary += uint32_t(__popcll(__ballot(ok))) * uint32_t(A.samples);
        if (ok) {
            float* dst = A.out + size_t(blockPixelIndex(A, b, mx, my)) * 3;
            storeRgb(dst, mk(0, 0, 0));
        }
    };
    WorkCursor cursor;
    openCursor(A.work, cursor);
    uint32_t firstUnit, lastUnit;
    while (nextChunk(A.work, cursor, firstUnit, lastUnit)) {
        // longest unit seen per block, reported once per chunk and block
        int32_t costSlot = -1;
        uint32_t costTicks = 0;
        auto reportCost = [&] {
            if (costSlot >= 0 && lane == 0 && costTicks >= A.costFloor)
                atomicMax(A.blockCost + costIndex(A, uint32_t(costSlot)), costTicks);
        };
        // (chunks shorter than a block: expensive or cheap units, not empty blocks)
        const bool mayHoldBlocks = A.cullOn && A.unitsPerBlockLog2 >= 0 && lastUnit - firstUnit >= (1u << A.unitsPerBlockLog2);
        for (uint32_t unit = firstUnit; unit < lastUnit; unit++) {
        // a whole 8x8 block inside this chunk that cannot see the scene: one wave-wide
        // store of black pixels (lane = pixel) instead of its units one by one
        if (mayHoldBlocks && (unit & ((1u << A.unitsPerBlockLog2) - 1u)) == 0 && unit + (1u << A.unitsPerBlockLog2) <= lastUnit) {
            const BlockPlace b = placeBlock(A, unit >> A.unitsPerBlockLog2);
            if (b.culled) {
                blackBlock(b);
                if (A.blockCost && b.listed && int32_t(b.own) != costSlot) {
                    reportCost();
                    costSlot = int32_t(b.own);
                    costTicks = 1;
                }
                unit += (1u << A.unitsPerBlockLog2) - 1u;
                continue;
            }
        }
        V3 accum = mk(0, 0, 0);
        const PixelPlace place = placePixel(A, unit, slot, laneUsed);
        if (place.costSlot != costSlot) {
            reportCost();
            costSlot = place.costSlot;
            costTicks = 1; // measured, however cheap
        }
        if (place.culled) {
            // every camera ray of the unit misses the scene's bounds: the rays are
            // counted, the pixels are black (src/render.cpp:61-64), nothing is traced
            if (MODE == 9)
                continue;
            for (int round = 0; round < A.rounds; round++)
                lc.countPrimary(place.ok && round * A.lanesPerPixel + sub < A.samples);
            if (writer && place.ok) {
                float* dst = A.out + size_t(place.outIndex) * 3;
                storeRgb(dst, mk(0, 0, 0));
            }
            continue;
        }
        const uint64_t unitBegan = place.costSlot >= 0 ? wall_clock64() : 0;
#ifdef RTX_PROBE
        const uint64_t probeBegan = wall_clock64();
        const uint32_t probeOuter = lc.tc.outer, probeInner = lc.tc.inner, probeWaveOuter = lc.tc.waveOuter, probeWaveInner = lc.tc.waveInner;
#endif
        const uint32_t storeIndex = place.ok ? place.outIndex : 0xffffffffu; // where this lane's pixel goes in the output; ~0: nowhere
        V3 carry = mk(0, 0, 0); // sum of the rounds so far (a pixel with more than 64 samples owns the wave)
        for (int round = 0; round < A.rounds; round++) {
            const int sample = round * A.lanesPerPixel + sub;
            V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
            bool active = false;
            uint32_t key = 0;
            {
                // the pixel's coordinates are only needed to make the camera ray; only the
                // output index is kept across the traversal loops
                const int x = place.x, yTop = place.yTop;
                const bool pixelOk = place.ok;
                const int y = A.H - 1 - yTop; // the reference's y runs bottom-up (src/screen.cpp:37-43)
                const uint32_t pixelIndex = uint32_t(y) * uint32_t(A.W) + uint32_t(x);
                // pixel corner, not centre (src/render.cpp:76-79)
                const float ndcX = float(x) / float(A.W) * 2.0f - 1.0f, ndcY = float(y) / float(A.H) * 2.0f - 1.0f;
                active = pixelOk && sample < A.samples;
                if (FULL)
                    key = sampleKey(A.seed, pixelIndex, uint32_t(sample));
                if (active) {
                    switch (A.driver) {
                    case RTX_DRIVER_MRAYS: {
                        // src/render.cpp:170-176: x jitter then y jitter, draws 2*(pixel*spp+i)+{1,2}
                        uint64_t rng = driverRngState(A, x, y, pixelIndex, sample, 2);
                        const float jx = minstdUniform(rng, 0.f, 1.f);
                        const float jy = minstdUniform(rng, 0.f, 1.f);
                        cameraRay(A.cam, float(float(x) + jx) / float(A.W) * 2.0f - 1.0f, float(float(y) + jy) / float(A.H) * 2.0f - 1.0f, o, d);
                    } break;
                    case RTX_DRIVER_DOF: {
                        // src/render.cpp:113-117, src/dof.cpp:9-39
                        V3 co, cd;
                        cameraRay(A.cam, ndcX, ndcY, co, cd);
                        const V3 focalPoint = co + A.focal * cd;
                        uint64_t rng = driverRngState(A, x, y, pixelIndex, sample, 3);
                        const float sx = minstdUniform(rng, -A.aperture / 2, A.aperture / 2);
                        const float sy = minstdUniform(rng, -A.aperture / 2, A.aperture / 2);
                        const float sz = minstdUniform(rng, -A.aperture / 2, A.aperture / 2);
                        o = co + mk(sx, sy, sz);
                        d = normalize(focalPoint - o);
                    } break;
                    case RTX_DRIVER_MOTION_BLUR:
                        if (BLUR) {
                            // src/render.cpp:203-247: the samples extrapolate from the current
                            // pose by the pose change since the previous call
                            uint64_t rng = driverRngState(A, x, y, pixelIndex, sample, 1);
                            const float jitter = minstdUniform(rng, 0.f, 1.f);
                            const float w = (jitter + float(sample)) / float(A.samples);
                            const V3 lookAt = mk(A.cam.look_at) + w * (mk(A.cam.look_at) - mk(A.prev.look_at));
                            const V3 euler = mk(A.cam.euler) + w * (mk(A.cam.euler) - mk(A.prev.euler));
                            const float dist = A.cam.distance + w * (A.cam.distance - A.prev.distance);
                            float qw;
                            V3 qv;
                            quatFromEuler(euler, qw, qv);
                            const V3 dirCam = normalize(mk(-ndcX * A.cam.half_width, ndcY * A.cam.half_height, 1.0f));
                            o = lookAt + rotate(qw, qv, mk(0, 0, -dist));
                            d = rotate(qw, qv, dirCam);
                        }
                        break;
                    case RTX_DRIVER_COMPOSED: if (BLUR) {
                        // time step x lens sample x pixel jitter (see rtx_driver): the drivers' own pieces,
                        // their draws in the order time, pixel, lens
                        uint64_t rng = driverRngState(A, x, y, pixelIndex, sample, A.drawsPerSample);
                        float w = 0.f;
                        if (A.timeSteps > 1) {
                            const int step = sample / (A.lensSamples * A.jitterSamples);
                            const float jitter = minstdUniform(rng, 0.f, 1.f);
                            w = (jitter + float(step)) / float(A.timeSteps);
                        }
                        float qx = ndcX, qy = ndcY;
                        if (A.jitterSamples > 1) {
                            const float jx = minstdUniform(rng, 0.f, 1.f);
                            const float jy = minstdUniform(rng, 0.f, 1.f);
                            qx = float(float(x) + jx) / float(A.W) * 2.0f - 1.0f;
                            qy = float(float(y) + jy) / float(A.H) * 2.0f - 1.0f;
                        }
                        if (A.timeSteps > 1) {
                            const V3 lookAt = mk(A.cam.look_at) + w * (mk(A.cam.look_at) - mk(A.prev.look_at));
                            const V3 euler = mk(A.cam.euler) + w * (mk(A.cam.euler) - mk(A.prev.euler));
                            const float dist = A.cam.distance + w * (A.cam.distance - A.prev.distance);
                            float qw;
                            V3 qv;
                            quatFromEuler(euler, qw, qv);
                            const V3 dirCam = normalize(mk(-qx * A.cam.half_width, qy * A.cam.half_height, 1.0f));
                            o = lookAt + rotate(qw, qv, mk(0, 0, -dist));
                            d = rotate(qw, qv, dirCam);
                        } else {
                            cameraRay(A.cam, qx, qy, o, d);
                        }
                        if (A.lensSamples > 1) {
                            const V3 focalPoint = o + A.focal * d;
                            const float sx = minstdUniform(rng, -A.aperture / 2, A.aperture / 2);
                            const float sy = minstdUniform(rng, -A.aperture / 2, A.aperture / 2);
                            const float sz = minstdUniform(rng, -A.aperture / 2, A.aperture / 2);
                            o = o + mk(sx, sy, sz);
                            d = normalize(focalPoint - o);
                        }
                    } break;
                    default:
                        cameraRay(A.cam, ndcX, ndcY, o, d);
                        break;
                    }
                }
            }
            if (MODE == 9) {
                // The camera pass of a deferred frame whose walk holds few pixels per wave: the
                // camera rays of a full wave of pixels, by the very query the walk would make
                // for each (closestHit, ray by ray), their hits left for the walk to pick up.
                float t = kFltMax;
                int bestK = -1, sphere = -1;
                V3 sphereN = mk(0, 0, 0);
                lc.countTraced(active);
                if (active) {
                    closestHit<STACK, COUNT>(A.sc, st, A.f, o, d, t, bestK, sphere, sphereN, lc.tc);
                    A.defer.primary[storeIndex * uint32_t(A.samples) + uint32_t(sample)] = make_uint2(__float_as_uint(t), uint32_t(bestK));
                }
                continue;
            }
            if (MODE == 4) {
                const int fanRays = min(max(A.glossyRays, 1), kFanMax);
                const int pathLanes = A.pixelsPerUnit * A.lanesPerPixel, fanSlots = (kBlock / 64) * pathLanes; // (lanes of a wave that carry a sample)
                const FanStore fan { reinterpret_cast<float*>(sFan), sFan + A.defer.fanLevels * fanRays * fanSlots, int(threadIdx.x >> 6) * pathLanes + min(lane, pathLanes - 1), fanSlots, fanRays,
                    sBest + (threadIdx.x & ~63u) };
                walkPath<STACK, COUNT>(A, st, fan, o, d, active, key, storeIndex * uint32_t(A.samples) + uint32_t(sample), lc, unit);
                continue; // (the pixel is written by resolveKernel)
            }
            const V3 c = (MODE == 4 || MODE == 9) ? mk(0, 0, 0) : MODE == 10 ? tracePathLeanMirror<STACK, COUNT>(A, st, o, d, active, lc) : MODE >= 11 ? tracePathLeanLayers<STACK, COUNT, MODE == 11>(A, st, o, d, active, lc) : FULL ? tracePath<STACK, COUNT, true, MODE == 2, MODE == 3>(A, st, o, d, active, key, lc) : tracePathLean<STACK, COUNT, MODE == 0 || MODE == 6 || MODE == 8, MODE == 8>(A, st, o, d, active, lc);
            if (!averaging) {
                accum = c;
            } else if (!anyLane(!(c.x == 0.f && c.y == 0.f && c.z == 0.f))) {
                // every sample of the wave is black: adding zeros in order leaves each pixel's
                // sum where it was (+0 at the start, src/render.cpp:168), so the chain is skipped
                accum = carry;
            } else {
                // The reference adds the samples of a pixel one after another, in sample
                // order, onto a zero (src/render.cpp:168-178).  The samples sit in
                // consecutive lanes, so that chain is a running sum along the lanes:
                // after step j the lane of sample j holds ((0 + c0) + c1) + ... + cj.
                const int n = min(A.lanesPerPixel, A.samples - round * A.lanesPerPixel); // samples of this round
                const bool head = sub == 0;
                // (selects written per component: a select between two V3 objects is
                // lowered to an indexed copy through scratch memory)
                float rx = head ? carry.x + c.x : c.x, ry = head ? carry.y + c.y : c.y, rz = head ? carry.z + c.z : c.z;
                if (A.lanesPerPixel == 16) {
                    // a pixel's samples fill one DPP row: shifting within the row feeds the
                    // head lane the 0 the chain starts from, so no select is needed
                    rx = ry = rz = 0.0f;
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        rx = fromPreviousLaneOfRow(rx) + c.x;
                        ry = fromPreviousLaneOfRow(ry) + c.y;
                        rz = fromPreviousLaneOfRow(rz) + c.z;
                    }
                } else
                for (int j = 1; j < n; j++) {
                    const float nx = fromPreviousLane(rx) + c.x, ny = fromPreviousLane(ry) + c.y, nz = fromPreviousLane(rz) + c.z;
                    rx = head ? rx : nx;
                    ry = head ? ry : ny;
                    rz = head ? rz : nz;
                }
                const V3 run = mk(rx, ry, rz);
                if (A.rounds > 1) {
                    // one pixel per wave: its sum so far sits in lane n - 1
                    carry = mk(fromLane(run.x, n - 1), fromLane(run.y, n - 1), fromLane(run.z, n - 1));
                    accum = carry;
                } else {
                    accum = run;
                }
            }
        }
        if (MODE != 4 && MODE != 9 && writer && storeIndex != 0xffffffffu) {
            {
                const uint32_t outIndex = storeIndex;
                const V3 px = normalisePixel(accum, normalise, A.samples);
                float* dst = A.out + size_t(outIndex) * 3;
                storeRgb(dst, px);
            }
        }
#ifdef RTX_PROBE
        {
            uint32_t mo = lc.tc.outer - probeOuter, mi = lc.tc.inner - probeInner;
            const uint32_t so = uint32_t(waveSum(mo));
            const uint32_t wo = uint32_t(waveSum(lc.tc.waveOuter - probeWaveOuter)), wi = uint32_t(waveSum(lc.tc.waveInner - probeWaveInner));
            for (int off = 32; off > 0; off >>= 1) {
                mo = max(mo, uint32_t(__shfl_xor(int(mo), off, 64)));
                mi = max(mi, uint32_t(__shfl_xor(int(mi), off, 64)));
            }
            if (lane == 0 && unit < kProbeUnits)
                g_probe[unit] = make_uint4(uint32_t(wall_clock64() - probeBegan), mo | (mi << 16), wo | (wi << 16), uint32_t(probeBegan));
        }
#endif
        // the longest unit of each block decides how early the block starts next frame
        if (place.costSlot >= 0)
            costTicks = max(costTicks, uint32_t(min(wall_clock64() - unitBegan, uint64_t(0xfffffffeu))));
        }
        reportCost();
    }
    if (A.work.plan) {
        // the ordered sequence ends with the blocks that cannot see the scene
        const uint32_t nWaves = gridDim.x * (kBlock / 64);
        for (uint32_t block = (cursor.endUnit[kRegions - 1] >> A.unitsPerBlockLog2) + waveIndex(); block < A.blocksMine; block += nWaves)
            blackBlock(placeBlock(A, block));
    }
    flushCounters<COUNT>(lc, A.stats);
}

// Block order for this frame: the rank's blocks that can see the scene by falling
// cost class (the longest unit each held on the previous frame, in half octaves;
// never measured counts as the most expensive), then the blocks that cannot.
// Inside a class the blocks keep their list order, so neighbours on the screen with
// like costs stay neighbours in the sequence and the waves working side by side
// keep touching the same part of the scene.  The work queue's plan comes with it:
// region 0 = the blocks up to class kLastSingleClass, one unit per claim; region 1 =
// up to kLastHeavyClass, one to three units per claim depending on how many units
// each wave will get; region 2 = the rest of the blocks that can see the scene, four.
// A stable counting sort on the classes in three small kernels, one wave per 64
// consecutive blocks (a "step"): classifyBlocksKernel finds each block's class, counts
// the classes of its step into table[class][step] and resets the block's cost for the
// new measurement: 1 (cheap until a unit reports otherwise) for a block that can see
// the scene, 0 (never measured) for the others; scanBlocksKernel (one workgroup) turns
// the table into places, class after class, and writes the plan; placeBlocksKernel
// writes the order.
constexpr int kMaxOrderedBlocks = 1 << 20, kScanThreads = 1024, kCostClasses = 32;
// costs are in ticks of the 100 MHz wall clock; class 15 starts at 232 us, class 23 at 14.5 us
constexpr uint32_t kLastSingleClass = 15, kLastHeavyClass = 23, kEmptyClass = kCostClasses - 1;
constexpr uint32_t kCostFloorTicks = 256; // 2.5 us: well inside the cheapest classes
RTX_DEV uint32_t costClass(uint32_t c) // 0 = most expensive ... 30 = cheapest
{
    if (c == 0)
        return 0;
    const uint32_t msb = 31u - uint32_t(__clz(int(c)));
    const uint32_t level = 2 * msb + (msb ? (c >> (msb - 1)) & 1u : 0u); // half octaves; 18 = 5.12 us
    if (level >= 18)
        return level >= 44 ? 0 : 44 - level; // 1 .. 26
    return level >= 16 ? 27 : level >= 14 ? 28 : level >= 10 ? 29 : 30;
}
// The classes among a wave's 64 blocks, one after another (neighbouring blocks share
// few): visit(class, lanes that hold it) is called with wave-uniform arguments.
template <typename F>
RTX_DEV void forEachClass(uint32_t cls, bool valid, F visit)
{
    uint64_t rest = __ballot(valid);
    while (rest) {
        const uint32_t b = uint32_t(__builtin_amdgcn_readlane(int(cls), __builtin_ctzll(rest)));
        const uint64_t same = __ballot(valid && cls == b);
        visit(b, same);
        rest &= ~same;
    }
}
__global__ void __launch_bounds__(256) classifyBlocksKernel(RenderArgs A, uint8_t* classes, uint32_t* table, uint32_t nSteps, uint32_t* clear, uint32_t clearWords)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u, step = i >> 6;
    // (the frame's ticket counters and statistics are zeroed on the way)
    for (uint32_t w = i; w < clearWords; w += gridDim.x * 256u)
        clear[w] = 0;
    const bool valid = i < A.blocksMine;
    // (the rectangles are staged in LDS: read one by one from the kernel's arguments
    // they cost this kernel three times its other work)
    __shared__ int4 rects[kMaxCutBoxes];
    if (threadIdx.x < uint32_t(A.nCutRects))
        rects[threadIdx.x] = make_int4(A.cutRects[threadIdx.x][0], A.cutRects[threadIdx.x][1], A.cutRects[threadIdx.x][2], A.cutRects[threadIdx.x][3]);
    __syncthreads();
    uint32_t cls = 0;
    if (valid) {
        const BlockPlace b = placeBlock(A, i); // (A.blockOrder is NULL here: list order)
        bool empty = b.culled;
        if (!empty && A.nCutRects > 0) {
            const int bx0 = int(b.x0), yHigh = A.H - 1 - int(b.y0), yLow = yHigh - 7;
            bool seen = false;
#pragma unroll 8
            for (int r = 0; r < A.nCutRects; r++) {
                const int4 q = rects[r];
                seen = seen || !(bx0 + 7 < q.x || bx0 > q.y || yHigh < q.z || yLow > q.w);
            }
            empty = !seen;
        }
        uint32_t& mine = A.blockCost[costIndex(A, i)];
        cls = empty ? kEmptyClass : costClass(mine);
        classes[i] = uint8_t(cls);
        mine = empty ? 0u : 1u;
    }
    uint32_t counts = 0; // lane c: blocks of class c in this step
    forEachClass(cls, valid, [&](uint32_t b, uint64_t same) { counts += lane == b ? uint32_t(__popcll(same)) : 0u; });
    if (lane < kCostClasses && step < nSteps)
        table[lane * nSteps + step] = counts;
}
__global__ void __launch_bounds__(kScanThreads) scanBlocksKernel(uint32_t* table, uint32_t nSteps, int unitsPerBlockLog2, uint32_t* plan, uint32_t nWaves, uint32_t lastSingleClass, uint32_t lastHeavyClass)
{
    // Exclusive prefix sum over the table read as one array.  Each wave owns a run of
    // whole 64-entry rows (lane = entry, so the accesses coalesce): it adds its run up,
    // the waves' totals are exchanged, then it goes over the run again with a running
    // sum along the lanes and a carry from row to row.
    __shared__ uint32_t waveTotal[kScanThreads / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, total = kCostClasses * nSteps;
    const uint32_t rows = (total + 63u) / 64u, rowsPerWave = (rows + kScanThreads / 64 - 1) / (kScanThreads / 64);
    const uint32_t lo = min(total, wave * rowsPerWave * 64u), hi = min(total, lo + rowsPerWave * 64u);
    auto alongLanes = [&](uint32_t v) { // inclusive running sum
        for (uint32_t o = 1; o < 64u; o <<= 1) {
            const uint32_t up = uint32_t(__shfl_up(int(v), o, 64));
            v += lane >= o ? up : 0u;
        }
        return v;
    };
    uint32_t mine = 0;
    for (uint32_t base = lo; base < hi; base += 64u)
        mine += base + lane < hi ? table[base + lane] : 0u;
    mine = alongLanes(mine);
    if (lane == 63u)
        waveTotal[wave] = mine;
    __syncthreads();
    uint32_t carry = 0;
    for (uint32_t w = 0; w < wave; w++)
        carry += waveTotal[w];
    for (uint32_t base = lo; base < hi; base += 64u) {
        const uint32_t v = base + lane < hi ? table[base + lane] : 0u;
        const uint32_t upTo = alongLanes(v);
        if (base + lane < hi)
            table[base + lane] = carry + upTo - v;
        carry += uint32_t(__builtin_amdgcn_readlane(int(upTo), 63));
    }
    __syncthreads(); // (one workgroup: its global writes are visible to it after the barrier)
    if (threadIdx.x == 0) {
        const uint32_t end0 = table[(lastSingleClass + 1) * nSteps] << unitsPerBlockLog2, end1 = table[(lastHeavyClass + 1) * nSteps] << unitsPerBlockLog2;
        plan[0] = end0;
        plan[1] = end1;
        plan[2] = table[kEmptyClass * nSteps] << unitsPerBlockLog2;
        plan[3] = 1;
        plan[4] = min(3u, max(1u, (end1 - end0) / (nWaves * 32u)));
        plan[5] = 4;
    }
}
__global__ void __launch_bounds__(256) placeBlocksKernel(const uint8_t* classes, uint32_t n, const uint32_t* table, uint32_t nSteps, uint32_t* order)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u, step = i >> 6;
    const bool valid = i < n;
    const uint32_t cls = valid ? classes[i] : 0u;
    const uint32_t places = (lane < kCostClasses && step < nSteps) ? table[lane * nSteps + step] : 0u; // lane c: where the step's class c goes
    forEachClass(cls, valid, [&](uint32_t b, uint64_t same) {
        if (valid && cls == b) {
            const uint32_t before = __builtin_amdgcn_mbcnt_hi(uint32_t(same >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(same), 0u));
            order[uint32_t(__builtin_amdgcn_readlane(int(places), int(b))) + before] = i;
        }
    });
}

// ---- passes 2 and 3 of a deferred frame -------------------------------------------------
struct LightArgs {
    DScene sc;
    rtx_features f;
    Deferred defer;
    int32_t light;          // index into the scene's lights
    int32_t slot;           // its place among the lights that count (index into a point's results)
    int32_t nS;             // samples (1 for a point light)
    uint32_t counterOffset; // keyed draws the point's earlier lights consume
    const uint32_t* unitStart; // [kPools + 1]: first unit of each pool's points for this light (made by planLightsKernel)
    WorkQueue work;
    unsigned long long* stats;
};
// After the walk: how many units (64 / group points each) every pool's points make for a light
// whose samples fill `group` lanes, as a running sum (unitStart), and the work queue's plan.
// One wave.  A frame whose pools ran over gets no units at all.
__global__ void __launch_bounds__(64) planLightsKernel(Deferred D, uint32_t group, uint32_t* unitStart, uint32_t* plan)
{
    const uint32_t lane = threadIdx.x;
    const bool over = *deferOverflow(D) != 0;
    const uint32_t count = over ? 0u : min(D.counters[(kPools + lane) * kPoolStride], D.pointsPerPool);
    const uint32_t perUnit = 64u / group;
    uint32_t v = (count + perUnit - 1u) / perUnit;
    for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t up = uint32_t(__shfl_up(int(v), o, 64));
        v += lane >= o ? up : 0u;
    }
    unitStart[lane + 1] = v;
    if (lane == 0)
        unitStart[0] = 0;
    if (lane == 63) {
        plan[0] = plan[1] = plan[2] = v;
        plan[3] = plan[4] = plan[5] = 1;
    }
}

// The light loop's body for one light and every shading point of the frame
// (computeLightContribution, src/light.cpp:139-189; sampleSegmentLight / sampleParallelogramLight,
// :26-45; testVisibilityLightSample, :70-102).  A unit = 64 / group shading points of one pool,
// group = min(samples, 64) lanes each: lane = one light sample, its shadow ray traced, the
// contributions added in sample order along the lanes (a light with more than 64 samples takes
// several rounds per point, the sum carried over), the mean written to the point's result.
// A point light is a light of one sample: 64 points per unit.
template <int STACK, bool COUNT>
__global__ void __launch_bounds__(kBlock, RTX_LIGHT_WAVES) lightKernel(LightArgs L)
{
    __shared__ int32_t sStack[2 * LaneStack<STACK>::kLds * kBlock];
    LaneStack<STACK> st { sStack, reinterpret_cast<float*>(sStack + LaneStack<STACK>::kLds * kBlock), int(threadIdx.x), nullptr };
    LaneCounters lc;
    const DScene& sc = L.sc;
    const Deferred& D = L.defer;
    const int lane = threadIdx.x & 63;
    const rtx_light& light = sc.lights[L.light];
    const int type = __builtin_amdgcn_readfirstlane(light.type);
    const bool isPoint = type == RTX_LIGHT_POINT;
    const int nS = isPoint ? 1 : L.nS;
    const int group = nS < 64 ? nS : 64, perUnit = 64 / group, rounds = (nS + 63) / 64;
    const int mySlot = lane / group, mySample = lane - mySlot * group;
    const bool shadowed = !isPoint || L.f.enable_hard_shadow; // (area lights always test visibility, src/light.cpp:160,173)
    const uint32_t startOfMine = L.unitStart[lane]; // lane c: first unit of pool c
    WorkCursor cursor;
    openCursor(L.work, cursor);
    uint32_t first, last;
    while (nextChunk(L.work, cursor, first, last)) {
        for (uint32_t unit = first; unit < last; unit++) {
            const uint32_t pool = uint32_t(__popcll(__ballot(startOfMine <= unit))) - 1u;
            const uint32_t inPool = (unit - uint32_t(__builtin_amdgcn_readlane(int(startOfMine), int(pool)))) * uint32_t(perUnit) + uint32_t(mySlot);
            const uint32_t count = min(__builtin_amdgcn_readfirstlane(D.counters[(kPools + pool) * kPoolStride]), D.pointsPerPool);
            const bool valid = mySlot < perUnit && inPool < count;
            const uint32_t point = pool * D.pointsPerPool + (valid ? inPool : 0u);
            V3 hp = mk(0, 0, 0), Rd = mk(0, 0, 1), N = mk(0, 0, 1), albedo = mk(0, 0, 0), ks = mk(0, 0, 0);
            float shin = 1.f;
            uint32_t key = 0, counter = 0;
            bool flipped = false;
            if (valid) {
                const ShadePoint& P = D.points[point];
                hp = mk(P.hp), Rd = mk(P.rd), N = mk(P.n), albedo = mk(P.albedo), ks = mk(P.ks);
                shin = P.shin;
                key = P.key;
                counter = P.counter + L.counterOffset;
                flipped = P.flipped != 0;
            }
            V3 carry = mk(0, 0, 0), result = mk(0, 0, 0);
            for (int round = 0; round < rounds; round++) {
                const int sampleIndex = round * 64 + mySample;
                const bool worker = valid && sampleIndex < nS;
                V3 pos = mk(light.p[0]), col = mk(light.c[0]);
                if (type == RTX_LIGHT_SEGMENT) {
                    // src/light.cpp:26-32
                    const V3 e0 = mk(light.p[0]), e1 = mk(light.p[1]);
                    const float a = keyedUniform(key, counter + uint32_t(sampleIndex));
                    pos = a * (e1 - e0) + e0;
                    col = (1 - a) * mk(light.c[0]) + a * mk(light.c[1]);
                } else if (type == RTX_LIGHT_PARALLELOGRAM) {
                    // src/light.cpp:36-45
                    const float a1 = keyedUniform(key, counter + 2u * uint32_t(sampleIndex));
                    const V3 c1 = (1 - a1) * mk(light.c[0]) + a1 * mk(light.c[1]);
                    const V3 c2 = (1 - a1) * mk(light.c[2]) + a1 * mk(light.c[3]);
                    const float a2 = keyedUniform(key, counter + 2u * uint32_t(sampleIndex) + 1u);
                    col = (1 - a2) * c1 + a2 * c2;
                    pos = mk(light.p[0]) + mk(light.p[1]) * a1 + mk(light.p[2]) * a2;
                }
                V3 c = mk(0, 0, 0);
                if (worker)
                    c = phong(pos, col, hp, Rd, N, albedo, ks, shin);
                if (shadowed) {
                    lc.countShadow(worker);
                    const V3 visN = flipped ? -N : N;
                    const bool facing = worker && lightFacesSurface(pos, hp, visN);
                    lc.countTraced(facing);
                    // (lane by lane: the samples of an area light start all over the light, a bundle
                    // too wide to be walked through the tree together; measured 1.7x slower that way)
                    const float visible = facing ? lightVisibility<STACK, COUNT>(sc, st, L.f, pos, hp, visN, lc.tc) : 0.0f;
                    c = c * (facing ? visible : 0.0f);
                }
                if (isPoint) {
                    result = c;
                } else {
                    // `avg += contribution` in sample order (src/light.cpp:160-163, 173-176)
                    const int cnt = min(group, nS - round * 64);
                    const bool head = mySample == 0;
                    float rx = head ? carry.x + c.x : c.x, ry = head ? carry.y + c.y : c.y, rz = head ? carry.z + c.z : c.z;
                    for (int j = 1; j < cnt; j++) {
                        const float nx = fromPreviousLane(rx) + c.x, ny = fromPreviousLane(ry) + c.y, nz = fromPreviousLane(rz) + c.z;
                        rx = head ? rx : nx;
                        ry = head ? ry : ny;
                        rz = head ? rz : nz;
                    }
                    if (round + 1 < rounds)
                        carry = mk(fromLane(rx, cnt - 1), fromLane(ry, cnt - 1), fromLane(rz, cnt - 1)); // (one point per unit here)
                    else
                        result = mk(rx, ry, rz) * float(1.0 / double(float(nS))); // the sum sits in the group's last lane
                }
            }
            const int lastSample = isPoint ? 0 : min(group, nS - (rounds - 1) * 64) - 1;
            if (valid && mySample == lastSample) {
                float* dst = D.contrib + (size_t(point) * D.nCounting + uint32_t(L.slot)) * 3;
                storeRgb(dst, result);
            }
        }
    }
    flushCounters<COUNT>(lc, L.stats);
}

struct ResolveArgs {
    Deferred defer;
    float* out;
    uint32_t nPixels;   // output pixels (the frame, or this rank's tile buffer)
    int32_t samples, normalise;
};
// getFinalColor's arithmetic replayed from a sample's tape (src/render.cpp:16-66)
RTX_DEV V3 resolveTape(const Deferred& D, uint32_t head)
{
    if (head == 0)
        return mk(0, 0, 0);
    // the calls whose children are being added up: the innermost in registers, the others on a stack
    V3 stackLo[kMaxDepth], stackKs[kMaxDepth], stackAcc[kMaxDepth];
    int32_t stackCount[kMaxDepth];
    V3 Lo = mk(0, 0, 0), ks = mk(0, 0, 0), acc = mk(0, 0, 0);
    int32_t count = 0;
    int level = 0;
    uint32_t at = (head - 1) * kTapeChunk;
    const uint32_t nPoints = D.pointsPerPool * uint32_t(kPools);
    // The replay of a deep sample (hundreds of entries) is a chain of dependent loads when taken entry by
    // entry, each entry's lights after it.  A chunk's entries lie side by side: all eight are fetched at
    // once, then the lights' results of all of them, and only then are they replayed in order.  (Entries
    // behind the sample's end are whatever an earlier frame left there: they are loaded, never used, and
    // their points are looked at only when they lie inside the pools.)
    for (;;) {
        TapeEntry e[kTapeChunk];
        const TapeEntry* chunk = D.tape + at;
#pragma unroll
        for (uint32_t i = 0; i < kTapeChunk; i++)
            e[i] = chunk[i];
        V3 value[kTapeChunk];
#pragma unroll
        for (uint32_t i = 0; i < kTapeChunk; i++) {
            value[i] = mk(e[i].a);
            // Lo: the lights in scene order (src/light.cpp:146-181)
            if ((e[i].op == TAPE_HIT || e[i].op == TAPE_OPEN) && e[i].point < nPoints) {
                const float* c = D.contrib + size_t(e[i].point) * D.nCounting * 3;
                for (uint32_t j = 0; j < D.nCounting; j++)
                    value[i] = value[i] + mk(c + 3 * j);
            }
        }
        bool linked = false;
#pragma unroll
        for (uint32_t i = 0; i < kTapeChunk && !linked; i++) {
            const uint32_t op = e[i].op;
            if (op == TAPE_LINK) {
                at = e[i].point * kTapeChunk;
                linked = true;
                break;
            }
            V3 v = value[i];
            if (op == TAPE_OPEN) {
                if (level > 0) {
                    stackLo[level - 1] = Lo, stackKs[level - 1] = ks, stackAcc[level - 1] = acc;
                    stackCount[level - 1] = count;
                }
                Lo = v;
                ks = mk(e[i].ks);
                acc = mk(0, 0, 0);
                count = 0;
                level++;
                continue;
            }
            if (op == TAPE_CLOSE) {
                // color *= 1/count (0 * inf = NaN when no child was followed), Lo + color * ks
                const V3 color = acc * (1.0f / float(count));
                v = Lo + color * ks;
                level--;
                if (level > 0) {
                    Lo = stackLo[level - 1], ks = stackKs[level - 1], acc = stackAcc[level - 1];
                    count = stackCount[level - 1];
                }
            }
            if (level == 0)
                return v;
            acc = acc + v;
            count++;
        }
        if (!linked)
            return mk(0, 0, 0); // (a chunk without a link or an end: not written by the walk)
    }
}
__global__ void __launch_bounds__(256) resolveKernel(ResolveArgs R)
{
    const Deferred& D = R.defer;
    if (*deferOverflow(D) != 0)
        return;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= R.nPixels)
        return;
    const uint32_t* heads = D.heads + size_t(i) * uint32_t(R.samples);
    if (heads[0] == kHeadUntouched)
        return; // another rank's pixel, or a block that cannot see the scene (written black by the walk)
    // the samples of a pixel are added one after another onto a zero (src/render.cpp:168-178)
    const bool averaging = R.normalise != NORM_SINGLE;
    V3 accum = mk(0, 0, 0);
    for (int s = 0; averaging && s < R.samples; s++)
        accum = accum + resolveTape(D, heads[s]);
    V3 px = normalisePixel(accum, R.normalise, R.samples);
    if (!averaging)
        px = resolveTape(D, heads[0]); // (not a sum: the one sample as it is, -0 included)
    float* dst = R.out + size_t(i) * 3;
    storeRgb(dst, px);
}
// A deferred frame whose pools ran over is rendered again by the one-kernel path, which counts
// its rays itself: the walk's counts are dropped.  One wave.
__global__ void __launch_bounds__(64) dropCountsIfOverflowKernel(Deferred D, unsigned long long* stats)
{
    if (*deferOverflow(D) == 0)
        return;
    for (uint32_t w = threadIdx.x; w < uint32_t(kStatSlots * kStatStride); w += 64u)
        stats[w] = 0;
}

// Batch form of getFinalColor (src/render.cpp:16-66): lane = caller-supplied ray.
struct ShadeArgs {
    RenderArgs R;          // scene, features, depth, sampling knobs, counters (frame fields unused)
    const rtx_ray* rays;
    float* out;            // n x RGB
    uint64_t n;
};

template <int STACK, bool COUNT, bool ENVL>
__global__ void __launch_bounds__(kBlock, RTX_FULL_WAVES) shadeKernel(ShadeArgs S)
{
    __shared__ int32_t sStack[2 * LaneStack<STACK>::kLds * kBlock];
    __shared__ int32_t sWaveRefs[(kBlock / 64) * 2 * STACK];
    LaneStack<STACK> st { sStack, reinterpret_cast<float*>(sStack + LaneStack<STACK>::kLds * kBlock), int(threadIdx.x), sWaveRefs + (threadIdx.x >> 6) * 2 * STACK };
    LaneCounters lc;
    const RenderArgs& A = S.R;
    const int lane = threadIdx.x & 63;
    WorkCursor cursor;
    openCursor(A.work, cursor);
    uint32_t first, last;
    while (nextChunk(A.work, cursor, first, last)) {
        for (uint32_t unit = first; unit < last; unit++) {
            const uint64_t i = uint64_t(unit) * 64 + lane;
            const bool active = i < S.n;
            V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
            float limit = kFltMax;
            if (active) {
                const rtx_ray r = S.rays[i];
                o = mk(r.origin);
                d = mk(r.direction);
                limit = r.t;
            }
            const V3 c = tracePath<STACK, COUNT, true, ENVL>(A, st, o, d, active, sampleKey(A.seed, uint32_t(i), 0u), lc, limit);
            if (active) {
                S.out[i * 3 + 0] = c.x;
                S.out[i * 3 + 1] = c.y;
                S.out[i * 3 + 2] = c.z;
            }
        }
    }
    flushCounters<COUNT>(lc, A.stats);
}

// scatter gathered compact tile buffers into a Screen-layout frame
// Makes an OriginFrame: thread i < 8 * nWide rewrites four-wide node i of the eight octant
// copies (every plane minus the origin's coordinate in that copy's mirror image; references
// and child count copied), thread j < nTris + 1 writes triangle record j (the padding record
// included): the pretest's three quads (c_0 and c_1 side by side for packed arithmetic), then the wave record with D - dot(n, origin) in place
// of D -- the operations traverseBvhWave makes per ray without a frame.
//
// The pretest (see OriginFrame), in double from the fp32 vertices: c_i = a_i x a_(i+1) with
// a_i = v_i - origin, all three negated if the ray towards the centroid has them negative.
// A ray from the origin that meets the triangle's plane at p has dot(d, c_i) = |c_i| |d| rho /
// |p - origin|, rho = p's distance from the plane through the origin and edge i, so dropping
// rays with dot(d, c_i) < -|c_i| |d| F / h (h: the origin's distance from the triangle's plane,
// <= |p - origin|) drops only rays whose p lies further than F outside edge i.  F bounds how far
// outside the triangle the reference's fp32 test can still see a hit: the rounding of its plane
// distance and of p (a few ulp of the coordinates' magnitude, |origin| + |a|; stretched along a
// grazing ray by |p - origin| / h, which the division by h above already carries) and the tilt of
// its fp32 unit normal against the exact one (a point at distance r <= `size` from v0 shifts by
// r * tilt, stretched the same way), each taken 16 to 40 times over.  Max norms stand in for
// the Euclidean ones (|c_i| |d| <= 3 cmax dmax).  M = inf (no pretest) for a triangle whose
// numbers leave that reasoning: degenerate, origin in its plane, extreme magnitudes.
__global__ void __launch_bounds__(256) originFrameKernel(const float4* waveNodes, uint32_t nWide, const float4* waveTris, const float4* tris, uint32_t nTris, float ox, float oy, float oz, float4* relNodes, float4* relTris)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 8u * nWide) {
        const uint32_t oct = i / nWide;
        const float mx = (oct & 1u) ? -ox : ox, my = (oct & 2u) ? -oy : oy, mz = (oct & 4u) ? -oz : oz;
        const float4* src = waveNodes + size_t(i) * 8;
        float4* dst = relNodes + size_t(i) * 8;
        float4 r[6];
        for (int q = 0; q < 6; q++) {
            const float m = q < 2 ? mx : q < 4 ? my : mz;
            const float4 v = src[q];
            dst[q] = r[q] = make_float4(v.x - m, v.y - m, v.z - m, v.w - m);
        }
        dst[6] = src[6];
        // which children hold the origin: the walk's own test (every plane difference <= 0 below, >= 0 above), made here once
        const float4 lo = make_float4(fmaxf(fmaxf(r[0].x, r[2].x), r[4].x), fmaxf(fmaxf(r[0].y, r[2].y), r[4].y), fmaxf(fmaxf(r[0].z, r[2].z), r[4].z), fmaxf(fmaxf(r[0].w, r[2].w), r[4].w));
        const float4 hi = make_float4(fminf(fminf(r[1].x, r[3].x), r[5].x), fminf(fminf(r[1].y, r[3].y), r[5].y), fminf(fminf(r[1].z, r[3].z), r[5].z), fminf(fminf(r[1].w, r[3].w), r[5].w));
        const int inside = (lo.x <= 0.0f && hi.x >= 0.0f ? 1 : 0) | (lo.y <= 0.0f && hi.y >= 0.0f ? 2 : 0) | (lo.z <= 0.0f && hi.z >= 0.0f ? 4 : 0) | (lo.w <= 0.0f && hi.w >= 0.0f ? 8 : 0);
        float4 last = src[7];
        last.w = __int_as_float(inside); // (the record's spare word)
        dst[7] = last;
    }
    if (i <= nTris) {
        const float4* src = waveTris + size_t(i) * kWaveTriQuads;
        float4* dst = relTris + size_t(i) * kFrameTriQuads;
        const float4 P = src[0];
        const float no = P.x * ox + P.y * oy + P.z * oz;
        dst[3] = make_float4(P.x, P.y, P.z, P.w - no);
        for (int q = 1; q < 6; q++)
            dst[3 + q] = src[q];
        const float inf = __builtin_inff();
        const float lastWord = __int_as_float(__float_as_int(src[1].w) < 0 ? int(0x80000000u) : 0);
        double c[3][3] = {};
        float M = inf;
        if (i < nTris) {
            // the triangle's record of the lane-by-lane walk holds the vertices: X = (v0.x, v1.x, v2.x, n.x), Y, Z
            const float4 X = tris[size_t(i) * 4 + 0], Y = tris[size_t(i) * 4 + 1], Z = tris[size_t(i) * 4 + 2];
            const double vx[3] = { X.x, X.y, X.z }, vy[3] = { Y.x, Y.y, Y.z }, vz[3] = { Z.x, Z.y, Z.z };
            double a[3][3];
            double amax = 0;
            for (int v = 0; v < 3; v++) {
                a[v][0] = vx[v] - double(ox), a[v][1] = vy[v] - double(oy), a[v][2] = vz[v] - double(oz);
                amax = fmax(amax, fmax(fabs(a[v][0]), fmax(fabs(a[v][1]), fabs(a[v][2]))));
            }
            double cmax = 0;
            for (int v = 0; v < 3; v++) {
                const double* p = a[v];
                const double* q = a[(v + 1) % 3];
                c[v][0] = p[1] * q[2] - p[2] * q[1], c[v][1] = p[2] * q[0] - p[0] * q[2], c[v][2] = p[0] * q[1] - p[1] * q[0];
                cmax = fmax(cmax, fmax(fabs(c[v][0]), fmax(fabs(c[v][1]), fabs(c[v][2]))));
            }
            const double g[3] = { (a[0][0] + a[1][0] + a[2][0]) / 3, (a[0][1] + a[1][1] + a[2][1]) / 3, (a[0][2] + a[1][2] + a[2][2]) / 3 };
            double e[3];
            for (int v = 0; v < 3; v++)
                e[v] = g[0] * c[v][0] + g[1] * c[v][1] + g[2] * c[v][2];
            const bool pos = e[0] > 0 && e[1] > 0 && e[2] > 0, neg = e[0] < 0 && e[1] < 0 && e[2] < 0;
            // the exact unit normal, the origin's distance from the plane, the longest edge
            const double ux = vx[1] - vx[0], uy = vy[1] - vy[0], uz = vz[1] - vz[0];
            const double wx = vx[2] - vx[0], wy = vy[2] - vy[0], wz = vz[2] - vz[0];
            const double zx = vx[2] - vx[1], zy = vy[2] - vy[1], zz = vz[2] - vz[1];
            double nx = uy * wz - uz * wy, ny = uz * wx - ux * wz, nz = ux * wy - uy * wx;
            const double nl = sqrt(nx * nx + ny * ny + nz * nz);
            const double size = sqrt(fmax(ux * ux + uy * uy + uz * uz, fmax(wx * wx + wy * wy + wz * wz, zx * zx + zy * zy + zz * zz)));
            if ((pos || neg) && nl > 0) {
                nx /= nl, ny /= nl, nz /= nl;
                const double h = fabs(nx * a[0][0] + ny * a[0][1] + nz * a[0][2]);
                const double tilt = fmax(fabs(nx - double(X.w)), fmax(fabs(ny - double(Y.w)), fabs(nz - double(Z.w)))) + 1e-6;
                const double omax = fmax(fabs(double(ox)), fmax(fabs(double(oy)), fabs(double(oz))));
                // (a plane constant that is not exactly dot(v0, n) moves the reference's plane along n)
                const double dErr = fabs(double(P.w) - (double(X.w) * vx[0] + double(Y.w) * vy[0] + double(Z.w) * vz[0]));
                const double F = 4e-6 * (omax + 3 * amax) + 4 * size * tilt + 4 * dErr;
                const double m = 3 * cmax * fmax(1e-5, F / h); // (h == 0: inf)
                const bool sane = cmax > double(kPretestMin) && cmax < double(kPretestMax) && tilt < 0.01 && m < 1e30 && m == m;
                if (sane) {
                    M = float(m) * 1.0001f;
                    if (neg)
                        for (int v = 0; v < 3; v++)
                            c[v][0] = -c[v][0], c[v][1] = -c[v][1], c[v][2] = -c[v][2];
                }
            }
        }
        dst[0] = make_float4(float(c[0][0]), float(c[1][0]), float(c[0][1]), float(c[1][1]));
        dst[1] = make_float4(float(c[0][2]), float(c[1][2]), float(c[2][0]), float(c[2][1]));
        dst[2] = make_float4(float(c[2][2]), M, lastWord, 0.0f);
    }
}

// The skip-ahead tables of the driver-level engine (see RenderArgs::jumpX): entry i of each is a
// power of that table's step, g^i, gRow^i, gS^i mod 2^31 - 1, by square-and-multiply per entry.
RTX_DEV uint32_t powMod(uint64_t base, uint32_t e)
{
    uint64_t r = 1;
    while (e) {
        if (e & 1u)
            r = mulMod(r, base);
        base = mulMod(base, base);
        e >>= 1;
    }
    return uint32_t(r);
}
__global__ void __launch_bounds__(256) jumpTablesKernel(uint32_t* jump, uint32_t maxDim, uint32_t W, uint32_t H, uint32_t nS, uint64_t g, uint64_t gRow, uint64_t gS)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < W)
        jump[i] = powMod(g, i);
    if (i < H)
        jump[maxDim + i] = powMod(gRow, i);
    if (i < nS)
        jump[2 * maxDim + i] = powMod(gS, i);
}

__global__ void detileKernel(const float* tiles, float* out, int W, int H, int ts, int tilesX, int tilesY, int world, int tilesPerRank)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t n = uint64_t(W) * H;
    if (i >= n)
        return;
    const int x = int(i % W), yTop = int(i / W);
    const int tx = x / ts, ty = yTop / ts;
    const int tile = ty * tilesX + tx;
    const int rank = tile % world, k = tile / world;
    const uint64_t src = (uint64_t(rank) * tilesPerRank + k) * ts * ts + uint64_t(yTop - ty * ts) * ts + (x - tx * ts);
    out[i * 3 + 0] = tiles[src * 3 + 0];
    out[i * 3 + 1] = tiles[src * 3 + 1];
    out[i * 3 + 2] = tiles[src * 3 + 2];
}

// ================================================================= host ====
thread_local char g_err[512] = "";
void setErr(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#define HIP_TRY(expr)                                                                 \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            setErr("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RTX_ERR_DEVICE;                                                    \
        }                                                                             \
    } while (0)

// The device memory the kernels work in beyond a call's inputs and outputs: the work
// queue's counters, the statistics, the skip-ahead tables of the driver-level RNG, and the
// per-block costs / order / classes of the frame scheduler.  Shared by every call that
// uses it (calls are serialised per GPU by DeviceCtx::launch and ordered on one stream).
struct Arena {
    uint32_t* tickets = nullptr;              // kQueues work counters, kQueueStride words apart
    unsigned long long* stats = nullptr;      // kStatSlots copies of the S_COUNT words, kStatStride apart (follows the tickets)
    uint32_t* queueBounds = nullptr;          // the work queue's plan, written by scanBlocksKernel
    // skip-ahead tables of the driver-level RNG (jumpX | jumpY | jumpS), rebuilt when the frame shape changes
    uint32_t* jump = nullptr;
    int jumpMaxDim = 0, jumpMaxSamples = 0;   // capacity: entries of jumpX and jumpY each, entries of jumpS
    int jumpW = -1, jumpH = -1, jumpS = -1, jumpD = -1;
    // longest unit per 8x8 block on the previous frame and the visiting order derived from
    // it; valid for the frame shape / partition recorded here
    uint32_t *blockCost = nullptr, *blockOrder = nullptr; // maxBlocks (+ 64) words each
    uint8_t* blockClass = nullptr;                         // maxBlocks bytes
    uint32_t* classTable = nullptr;                        // kCostClasses x (maxBlocks / 64) words
    uint32_t maxBlocks = 0;
    int costW = -1, costH = -1, costTile = -1, costRank = -1, costWorld = -1, costDriver = -1;
};
constexpr size_t kTicketBytes = size_t(kQueues) * kQueueStride * sizeof(uint32_t), kStatBytes = size_t(kStatSlots) * kStatStride * sizeof(unsigned long long);
constexpr int kJumpMaxDim = 32768, kJumpMaxSamples = 4096;

// The arena of rtx_render_device_static: one file-scope array per GPU instead of
// allocations, for callers that must not allocate.  Frames up to kFixedMaxDim pixels a
// side and kFixedMaxBlocks 8x8 blocks (tile padding included).
constexpr int kFixedMaxDim = 4096, kFixedMaxSamples = 1024;
constexpr uint32_t kFixedMaxBlocks = 4096u * 4096u / 64u;
constexpr size_t kFixedJumpWords = 2 * size_t(kFixedMaxDim) + kFixedMaxSamples;
constexpr size_t alignUp256(size_t n) { return (n + 255) & ~size_t(255); }
constexpr size_t kFixedTickets = 0;
constexpr size_t kFixedBounds = kFixedTickets + alignUp256(kTicketBytes + kStatBytes);
constexpr size_t kFixedJump = kFixedBounds + 256;
constexpr size_t kFixedCost = kFixedJump + alignUp256((2 * size_t(kFixedMaxDim) + kFixedMaxSamples) * sizeof(uint32_t));
constexpr size_t kFixedOrder = kFixedCost + alignUp256((size_t(kFixedMaxBlocks) + 64) * sizeof(uint32_t));
constexpr size_t kFixedClass = kFixedOrder + alignUp256(size_t(kFixedMaxBlocks) * sizeof(uint32_t));
constexpr size_t kFixedTable = kFixedClass + alignUp256(size_t(kFixedMaxBlocks) + 4);
constexpr size_t kFixedBytes = kFixedTable + alignUp256(size_t(kCostClasses) * (kFixedMaxBlocks / 64) * sizeof(uint32_t));
// In host code this name is only a handle for hipGetSymbolAddress: the array's device
// address is looked up once per GPU in rtx_init and kept in DeviceCtx::fixedBase.
static __device__ __align__(256) unsigned char scratch[kFixedBytes];

struct DeviceCtx {
    int device = -1;
    int slot = 0;                      // which of the GPU's frame slots this is (rtx_select_slot)
    hipStream_t stream = nullptr;      // stream work is queued on (own or caller's)
    hipStream_t ownStream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evSwitch = nullptr;            // orders the old stream's work before the new stream's (rtx_set_stream)
    int numCUs = 0;
    // the OriginFrame of the camera position last rendered from (see ensureCameraFrame)
    struct CameraFrame {
        float4 *nodes = nullptr, *tris = nullptr;
        size_t nodeBytes = 0, triBytes = 0;
        bool valid = false, haveLast = false;
        uint64_t sceneId = 0, lastSceneId = 0;
        float pos[3] = {}, lastPos[3] = {};
    } cameraFrame;
    std::mutex launch;                        // one frame / batch at a time per GPU
    Arena heap;                               // allocated in rtx_init
    Arena fixed;                              // views into this GPU's copy of `scratch`
    unsigned char* fixedBase = nullptr;       // device address of `scratch` on this GPU
    // device framebuffer behind the host-buffer entry point (rtx_render), kept from call to
    // call and grown when a larger frame comes; hostFrame serialises its users
    std::mutex hostFrame;
    float* frame = nullptr;
    size_t frameBytes = 0;
    // memory of the deferred frames (see Deferred), grown on demand and kept
    struct DeferPools {
        uint32_t* heads = nullptr;
        size_t headWords = 0;
        TapeEntry* tape = nullptr;
        uint32_t tapePerPool = 0;
        ShadePoint* points = nullptr;
        float* contrib = nullptr;
        uint32_t pointsPerPool = 0, contribLights = 0;
        uint32_t* counters = nullptr;   // the pools' counters and the overflow flag, then the light launches' ticket counters, unit tables and plans
        uint32_t* overflowHost = nullptr; // page-locked copy of the last deferred frame's overflow flag
        uint32_t wantPoints = 0;        // points per pool asked for by the last frame that ran over
    } defer;
};
constexpr int kMaxDeferLights = 8;
constexpr size_t kDeferCounterWords = 2 * size_t(kPools) * kPoolStride + 64;
constexpr size_t kDeferTicketWords = size_t(kQueues) * kQueueStride;
constexpr size_t kDeferUnitWords = 128, kDeferPlanWords = 64;
constexpr size_t kDeferControlWords = kDeferCounterWords + kMaxDeferLights * (kDeferTicketWords + kDeferUnitWords + kDeferPlanWords);

uint64_t hostMulMod(uint64_t a, uint64_t b) { return (a * b) % 2147483647ull; }
uint64_t hostPowMod(uint64_t base, uint64_t e)
{
    uint64_t r = 1;
    base %= 2147483647ull;
    while (e) {
        if (e & 1)
            r = hostMulMod(r, base);
        base = hostMulMod(base, base);
        e >>= 1;
    }
    return r;
}
thread_local DeviceCtx* g_ctx = nullptr;
std::mutex g_ctxMutex;
constexpr int kSlots = RTX_FRAME_SLOTS; // frame slots per GPU (rtx_select_slot)
DeviceCtx g_ctxs[64][kSlots];

} // namespace

struct rtx_scene {
    int device = -1;
    int refs = 1;
    DScene d {};
    void* buffers[32] = {};
    int nBuffers = 0;
    uint64_t id = 0;           // never reused (a slot's camera frame remembers which scene it was made for)
    std::vector<std::array<float, 3>> lightPos; // p[0] of every light
    int numLevels = 0;
    bool hasEnv = false;
    bool hasTextures = false;  // some material may carry a texture
    bool hasAreaLight = false; // a segment or parallelogram light
    std::vector<int32_t> lightTypes;
    int numLeaves = 0;         // leaves of the tree (the description's count, or the device builder's)
    float uploadMs = 0.f, buildMs = 0.f; // rtx_scene_build_device: the triangles' way to the GPU, and the build there
    // boxes of a shallow cut through the tree (at most kMaxCutBoxes; every triangle lies
    // in one of them), for the per-frame visibility rectangles
    std::vector<std::array<float, 6>> cutBoxes;
};

namespace {

template <class T>
int upload(rtx_scene* s, const T* src, size_t count, const T** dst)
{
    *dst = nullptr;
    // an empty array still gets a valid pointer so kernels never see NULL
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    s->buffers[s->nBuffers++] = p;
    HIP_TRY(hipMemset(p, 0, bytes));
    if (count)
        HIP_TRY(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(p);
    return RTX_OK;
}

std::atomic<uint64_t> g_sceneIds { 0 };
bool originFramesOff() { return getenv("RTX_NO_ORIGIN_FRAMES") != nullptr; } // diagnostic

size_t frameNodeBytes(const rtx_scene* s) { return std::max<size_t>(size_t(uint32_t(s->d.nWaveNodes)) * 8, 1) * 8 * sizeof(float4); }
size_t frameTriBytes(const rtx_scene* s) { return (size_t(uint32_t(s->d.nTris)) + 1) * kFrameTriQuads * sizeof(float4); }
void launchOriginFrame(const rtx_scene* s, const float* pos, float4* nodes, float4* tris, hipStream_t stream)
{
    const uint32_t nWide = uint32_t(s->d.nWaveNodes), nTris = uint32_t(s->d.nTris);
    const uint64_t threads = std::max<uint64_t>(8ull * nWide, uint64_t(nTris) + 1);
    hipLaunchKernelGGL(originFrameKernel, dim3(uint32_t((threads + 255) / 256)), dim3(256), 0, stream, s->d.waveNodes, nWide, s->d.waveTris, s->d.tris, nTris, pos[0], pos[1], pos[2], nodes, tris);
}

// The frames of the scene's first point lights (their shadow rays all start at the light),
// made once at upload.  The scene's wave records must be in place.
int makeLightFrames(rtx_scene* s, hipStream_t stream)
{
    s->id = ++g_sceneIds;
    for (int i = 0; i < kLightFrames; i++)
        s->d.lightFrame[i] = OriginFrame { nullptr, nullptr };
    if (s->d.nTris <= 0 || !s->d.waveTris || originFramesOff())
        return RTX_OK;
    for (size_t li = 0; li < s->lightTypes.size() && li < size_t(kLightFrames); li++) {
        if (s->lightTypes[li] != RTX_LIGHT_POINT)
            continue;
        void *nodes = nullptr, *num = nullptr; // (the node copies, the triangle copy)
        HIP_TRY(hipMalloc(&nodes, frameNodeBytes(s)));
        s->buffers[s->nBuffers++] = nodes;
        HIP_TRY(hipMalloc(&num, frameTriBytes(s)));
        s->buffers[s->nBuffers++] = num;
        launchOriginFrame(s, s->lightPos[li].data(), static_cast<float4*>(nodes), static_cast<float4*>(num), stream);
        HIP_TRY(hipGetLastError());
        s->d.lightFrame[li] = OriginFrame { static_cast<const float4*>(nodes), static_cast<const float4*>(num) };
    }
    HIP_TRY(hipStreamSynchronize(stream));
    return RTX_OK;
}

// The frame of the camera position for drivers whose camera rays all start there, kept per
// frame slot and remade (one small kernel on the slot's stream) when the scene or the position
// changes.  A scene whose copy is large (> 256 MB) gets one only from the second frame in a
// row from the same position: for a moving camera the copy would cost what it saves.
int ensureCameraFrame(DeviceCtx* c, const rtx_scene* s, const float* pos, OriginFrame& out)
{
    out = OriginFrame { nullptr, nullptr };
    DeviceCtx::CameraFrame& F = c->cameraFrame;
    if (F.valid && F.sceneId == s->id && std::memcmp(F.pos, pos, sizeof(F.pos)) == 0) {
        out = OriginFrame { F.nodes, F.tris };
        return RTX_OK;
    }
    const size_t nodeBytes = frameNodeBytes(s), triBytes = frameTriBytes(s);
    const bool repeated = F.haveLast && F.lastSceneId == s->id && std::memcmp(F.lastPos, pos, sizeof(F.lastPos)) == 0;
    F.haveLast = true;
    F.lastSceneId = s->id;
    std::memcpy(F.lastPos, pos, sizeof(F.lastPos));
    if (nodeBytes + triBytes > (size_t(256) << 20) && !repeated)
        return RTX_OK;
    F.valid = false;
    if (nodeBytes > F.nodeBytes || triBytes > F.triBytes) {
        HIP_TRY(hipStreamSynchronize(c->stream)); // nothing queued may still read what is freed below
        if (F.nodes)
            (void)hipFree(F.nodes);
        if (F.tris)
            (void)hipFree(F.tris);
        F.nodes = F.tris = nullptr, F.nodeBytes = F.triBytes = 0;
        HIP_TRY(hipMalloc((void**)&F.nodes, nodeBytes));
        F.nodeBytes = nodeBytes;
        HIP_TRY(hipMalloc((void**)&F.tris, triBytes));
        F.triBytes = triBytes;
    }
    launchOriginFrame(s, pos, F.nodes, F.tris, c->stream);
    HIP_TRY(hipGetLastError());
    F.valid = true;
    F.sceneId = s->id;
    std::memcpy(F.pos, pos, sizeof(F.pos));
    out = OriginFrame { F.nodes, F.tris };
    return RTX_OK;
}

// Calls f with the kernels' stack size as a compile-time constant.  A development build
// (-DRTX_DEV_BUILD: tools/dev_build.sh) instantiates only the 16-entry kernels without test counters,
// an eighth of the library, for experiments on trees of at most 16 levels; pickStack refuses deeper ones there.
template <class F>
void withStack(int stack, F f)
{
#ifdef RTX_DEV_BUILD
    (void)stack;
    f(std::integral_constant<int, 16> {});
#else
    switch (stack) {
    case 16: f(std::integral_constant<int, 16> {}); break;
    case 24: f(std::integral_constant<int, 24> {}); break;
    case 32: f(std::integral_constant<int, 32> {}); break;
    default: f(std::integral_constant<int, 64> {}); break;
    }
#endif
}
#ifdef RTX_DEV_BUILD
#define RTX_IF_COUNT(count) if constexpr (false)
#else
#define RTX_IF_COUNT(count) if (count)
#endif

int pickStack(int numLevels)
{
    // nearer-child-first depth-first traversal holds at most one entry per level
    if (const char* e = getenv("RTX_MIN_STACK")) // diagnostic: run a shallow tree on a deeper stack's kernel
        numLevels = std::max(numLevels, atoi(e));
    if (numLevels <= 16)
        return 16;
#ifdef RTX_DEV_BUILD
    return -1;
#endif
    if (numLevels <= 24)
        return 24;
    if (numLevels <= 32)
        return 32;
    if (numLevels <= 64)
        return 64;
    return -1;
}

// Persistent grid: as many 4-wave blocks as the kernel's registers and LDS let
// the chip hold at once.  No block waits on another, so a smaller share of the
// chip (a busy GPU) only makes the launch slower, never stuck.
int gridFor(const DeviceCtx* c, const void* kernel, size_t dynamicLds = 0)
{
    int perCU = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, kBlock, dynamicLds) != hipSuccess || perCU < 1)
        perCU = 2;
    if (const char* e = getenv("RTX_GRID_PER_CU"))
        perCU = std::max(1, atoi(e));
    return c->numCUs * perCU;
}
// Units per claim: small, so that the last chunks of a frame are short, but with
// enough units that a wave's claim for the next chunk returns before it is needed.
WorkQueue makeQueue(const DeviceCtx* c, const Arena* a, uint32_t nUnits)
{
    WorkQueue q;
    q.tickets = a->tickets;
    q.nUnits = nUnits;
    const uint32_t waves = uint32_t(c->numCUs) * 16u;
    q.chunk = std::min(8u, std::max(1u, nUnits / (waves * 32u)));
    if (const char* e = getenv("RTX_CHUNK"))
        q.chunk = uint32_t(std::max(1, atoi(e)));
    q.nChunks = uint32_t((uint64_t(nUnits) + q.chunk - 1) / q.chunk);
    q.plan = nullptr;
    // A frame spread thinly over the chip (one rank's share of it: a handful of chunks
    // per wave) ends on its longest units, and a wave that holds its next chunk while
    // it works through one of them keeps that chunk from the waves that have fallen
    // idle; there the chunks of the longest blocks are claimed when needed.  A frame
    // that keeps every wave busy for long is better off with every claim made ahead.
    q.aheadFrom = nUnits < waves * 24u ? 1 : 0;
    if (const char* e = getenv("RTX_CLAIM_AHEAD_FROM")) // diagnostic
        q.aheadFrom = uint32_t(std::min(kRegions - 1, std::max(0, atoi(e))));
    return q;
}

template <int STACK>
int launchTrace(DeviceCtx* c, const TraceArgs& A, bool count)
{
    RTX_IF_COUNT(count) {
        const int grid = std::max(1, std::min<int>(gridFor(c, (const void*)traceKernel<STACK, true>), int((A.work.nChunks + 3) / 4)));
        hipLaunchKernelGGL((traceKernel<STACK, true>), dim3(grid), dim3(kBlock), 0, c->stream, A);
    } else {
        const int grid = std::max(1, std::min<int>(gridFor(c, (const void*)traceKernel<STACK, false>), int((A.work.nChunks + 3) / 4)));
        hipLaunchKernelGGL((traceKernel<STACK, false>), dim3(grid), dim3(kBlock), 0, c->stream, A);
    }
    return RTX_OK;
}

template <int STACK>
int launchRender(DeviceCtx* c, const RenderArgs& A, bool count, int mode, size_t dynamicLds = 0)
{
    if (const char* e = getenv("RTX_EXTRA_LDS")) // diagnostic: fewer workgroups per CU (how the frame's time goes with the waves per SIMD)
        dynamicLds += size_t(std::max(0, atoi(e)));
#define RTX_LAUNCH(CNT, MODEV)                                                                                    \
    {                                                                                                             \
        const void* k = (const void*)renderKernel<STACK, CNT, MODEV>;                                             \
        if (dynamicLds)                                                                                           \
            (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, int(dynamicLds));            \
        const int grid = std::max(1, std::min<int>(gridFor(c, k, dynamicLds), int((A.work.nChunks + 3) / 4)));    \
        hipLaunchKernelGGL((renderKernel<STACK, CNT, MODEV>), dim3(grid), dim3(kBlock), dynamicLds, c->stream, A); \
    }
#define RTX_LAUNCH_MODES(CNT)                                                                                     \
    switch (mode) {                                                                                               \
    case 12: RTX_LAUNCH(CNT, 12) break;                                                                           \
    case 11: RTX_LAUNCH(CNT, 11) break;                                                                           \
    case 10: RTX_LAUNCH(CNT, 10) break;                                                                           \
    case 9: RTX_LAUNCH(CNT, 9) break;                                                                             \
    case 8: RTX_LAUNCH(CNT, 8) break;                                                                             \
    case 7: RTX_LAUNCH(CNT, 7) break;                                                                             \
    case 6: RTX_LAUNCH(CNT, 6) break;                                                                             \
    case 5: RTX_LAUNCH(CNT, 5) break;                                                                             \
    case 4: RTX_LAUNCH(CNT, 4) break;                                                                             \
    case 3: RTX_LAUNCH(CNT, 3) break;                                                                             \
    case 2: RTX_LAUNCH(CNT, 2) break;                                                                             \
    case 1: RTX_LAUNCH(CNT, 1) break;                                                                             \
    default: RTX_LAUNCH(CNT, 0) break;                                                                            \
    }
    RTX_IF_COUNT(count)
        RTX_LAUNCH_MODES(true)
    else
        RTX_LAUNCH_MODES(false)
#undef RTX_LAUNCH_MODES
#undef RTX_LAUNCH
    return RTX_OK;
}

int checkCtx(const rtx_scene* scene)
{
    if (!g_ctx) {
        setErr("rtx_init has not been called on this thread");
        return RTX_ERR_NOT_INIT;
    }
    if (scene && scene->device != g_ctx->device) {
        setErr("scene lives on device %d, this thread uses device %d", scene->device, g_ctx->device);
        return RTX_ERR_INVALID;
    }
    // the host framework may have made another GPU current on this thread since rtx_init
    HIP_TRY(hipSetDevice(g_ctx->device));
    return RTX_OK;
}

double nowMs()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// environment switch for the per-ray work counters (box / triangle tests): they
// cost registers, so they are off unless asked for
bool wantCounters()
{
    const char* e = getenv("RTX_COUNT_TESTS");
    return e && e[0] == '1';
}

// With stats == NULL the call only queues work and returns; with stats it waits
// for the kernel and reads the counters back.
int fetchStats(DeviceCtx* c, const Arena* a, rtx_stats* stats, double t0)
{
    if (!stats)
        return RTX_OK;
    float ms = 0.f;
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    {
        unsigned long long slots[size_t(kStatSlots) * kStatStride]; // (on the stack: this path must not allocate)
        HIP_TRY(hipMemcpy(slots, a->stats, kStatBytes, hipMemcpyDeviceToHost));
        unsigned long long h[S_COUNT] = {};
        for (int i = 0; i < kStatSlots; i++)
            for (int j = 0; j < S_COUNT; j++)
                h[j] += slots[size_t(i) * kStatStride + j];
        stats->rays_primary = h[S_PRIMARY];
        stats->rays_shadow = h[S_SHADOW];
        stats->rays_bounce = h[S_BOUNCE];
        stats->box_tests = h[S_BOX];
        stats->tri_tests = h[S_TRI];
        stats->shaded_hits = h[S_SHADED];
        stats->rays_traced = h[S_TRACED];
        stats->layer_overflows = h[S_OVERFLOW];
        stats->kernel_ms = ms;
        stats->total_ms = nowMs() - t0;
    }
    return RTX_OK;
}

int traceImpl(const rtx_scene* scene, const rtx_ray* dRays, size_t n, rtx_hit* dHits, const rtx_features* f, rtx_stats* stats, double t0)
{
    DeviceCtx* c = g_ctx;
    std::lock_guard<std::mutex> launchLock(c->launch);
    if (n > (size_t(1) << 37)) {
        setErr("rtx_trace: %zu rays exceed the 2^37 limit", n);
        return RTX_ERR_LIMIT;
    }
    const int stack = pickStack(scene->numLevels);
    if (stack < 0) {
        setErr("BVH has %d levels; the device traversal stack holds 64", scene->numLevels);
        return RTX_ERR_LIMIT;
    }
    TraceArgs A;
    A.sc = scene->d;
    A.f = *f;
    A.rays = dRays;
    A.hits = dHits;
    A.n = n;
    const Arena* a = &c->heap;
    A.work.nUnits = uint32_t((n + 63) / 64);
    A.work = makeQueue(c, a, A.work.nUnits);
    A.stats = a->stats;
    HIP_TRY(hipMemsetAsync(a->tickets, 0, kTicketBytes + kStatBytes, c->stream)); // the statistics follow the tickets
    if (stats)
        HIP_TRY(hipEventRecord(c->ev0, c->stream));
    if (n) {
        const bool count = wantCounters();
        withStack(stack, [&](auto S) { launchTrace<decltype(S)::value>(c, A, count); });
        HIP_TRY(hipGetLastError());
    }
    if (stats)
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
    return fetchStats(c, a, stats, t0);
}

struct TileGeom {
    int ts, tilesX, tilesY, world, rank, tilesMine, tilesPerRank;
};
int tileGeom(const rtx_render_params* p, TileGeom& g)
{
    g.ts = p->tile_size > 0 ? p->tile_size : 32;
    if (g.ts % 8 != 0 || g.ts > 1024) {
        setErr("tile_size must be a multiple of 8 up to 1024 (got %d)", g.ts);
        return RTX_ERR_INVALID;
    }
    if (p->width <= 0 || p->height <= 0 || p->width > 32768 || p->height > 32768) {
        setErr("frame size %dx%d out of range", p->width, p->height);
        return RTX_ERR_INVALID;
    }
    g.world = p->tile_world > 1 ? p->tile_world : 1;
    g.rank = p->tile_world > 1 ? p->tile_rank : 0;
    if (g.rank < 0 || g.rank >= g.world) {
        setErr("tile_rank %d outside tile_world %d", g.rank, g.world);
        return RTX_ERR_INVALID;
    }
    g.tilesX = (p->width + g.ts - 1) / g.ts;
    g.tilesY = (p->height + g.ts - 1) / g.ts;
    const int total = g.tilesX * g.tilesY;
    g.tilesPerRank = (total + g.world - 1) / g.world;
    g.tilesMine = total > g.rank ? (total - g.rank + g.world - 1) / g.world : 0;
    return RTX_OK;
}

// Environment-light sampling (soft shadows + environment mapping) needs the radiance
// bins of a textured environment and a multi-jitter grid the kernel can hold.
int checkEnvLight(const rtx_scene* scene, const rtx_features* f, int shadowSamples)
{
    if (!(f->enable_soft_shadow && f->enable_environment_mapping))
        return RTX_OK;
    if (scene->d.env.texture >= 0 && scene->d.nEnvBins <= 0) {
        setErr("environment light sampling needs rtx_scene_desc.env_bins for a textured environment map");
        return RTX_ERR_INVALID;
    }
    if (int(std::sqrt(double(shadowSamples) / 10.)) > kEnvMaxPerTry) {
        setErr("shadow_samples %d: environment light sampling handles fewer than %d", shadowSamples, 10 * (kEnvMaxPerTry + 1) * (kEnvMaxPerTry + 1));
        return RTX_ERR_LIMIT;
    }
    return RTX_OK;
}

FastDiv makeFastDiv(uint32_t d)
{
    if (d <= 1)
        return FastDiv { 0u, 0u };
    uint32_t l = 0;
    while ((1ull << l) < d)
        l++;
    return FastDiv { uint32_t((1ull << (31 + l)) / d + 1ull), l - 1 };
}

// Pixels that can possibly see the scene from this camera: the bounding rectangle,
// in pixel coordinates (y bottom-up as in src/render.cpp:76-79), of the projection
// of the root box grown by a thousandth of its size and distance (of a sphere
// around it when a corner lies behind the camera).  A camera ray through any other
// pixel, whatever its jitter inside the pixel, passes the true box at a distance
// far beyond what rounding in the reference's slab test could bridge, so it misses
// the root box there too and the pixel is black.  Returns false when no such
// statement can be made (camera inside the bounds).
// `lens` > 0 (depth of field, src/dof.cpp:9-39): a sample's ray starts up to `lens` from the
// camera position and runs through the point at distance `focal` on the pixel's central ray,
// so at distance s along the central ray it is at most lens * |1 - s / focal| away from it:
// the box is grown by the largest such distance over its depth range, and a central ray that
// misses the grown box has no sample ray that touches the box itself.
// `sway`, `turn` > 0 (motion blur, src/render.cpp:203-247, with `cam` the pose halfway through
// the exposure): a sample's camera stands at most `sway` from that pose's position and its
// rotation differs by at most `turn` radians, so a point at distance s along a sample's ray is
// at most sway + s * turn from the halfway pose's ray through the same pixel.
bool cullRectangle(const float* boxLo, const float* boxHi, const rtx_camera* cam, int W, int H, int32_t& x0, int32_t& x1, int32_t& y0, int32_t& y1, double lens = 0.0,
    double focal = 1.0, double sway = 0.0, double turn = 0.0)
{
    // world -> camera space: the conjugate of the camera's rotation applied to p - position
    const double qw = cam->quat[0], qx = -cam->quat[1], qy = -cam->quat[2], qz = -cam->quat[3];
    auto toCamera = [&](const double p[3], double out[3]) {
        const double c[3] = { p[0] - double(cam->position[0]), p[1] - double(cam->position[1]), p[2] - double(cam->position[2]) };
        const double ux = qy * c[2] - qz * c[1], uy = qz * c[0] - qx * c[2], uz = qx * c[1] - qy * c[0];
        const double vx = qy * uz - qz * uy, vy = qz * ux - qx * uz, vz = qx * uy - qy * ux;
        out[0] = c[0] + 2.0 * (qw * ux + vx), out[1] = c[1] + 2.0 * (qw * uy + vy), out[2] = c[2] + 2.0 * (qw * uz + vz);
    };
    double centre[3], half[3], half2 = 0, extent = 0;
    for (int i = 0; i < 3; i++) {
        centre[i] = 0.5 * (double(boxLo[i]) + double(boxHi[i]));
        half[i] = 0.5 * (double(boxHi[i]) - double(boxLo[i]));
        half2 += half[i] * half[i];
        extent = std::max(extent, 2.0 * half[i]);
    }
    double cc[3];
    toCamera(centre, cc);
    const double dist = std::sqrt(cc[0] * cc[0] + cc[1] * cc[1] + cc[2] * cc[2]);
    if (!std::isfinite(dist) || !std::isfinite(half2))
        return false;
    // slopes x/z, y/z of the rays that can touch the bounds
    double sLo[2] = { -INFINITY, -INFINITY }, sHi[2] = { INFINITY, INFINITY };
    double grow = 1e-3 * (extent + dist) + 1e-6;
    if (lens > 0.0 || sway > 0.0 || turn > 0.0) {
        const double far = dist + std::sqrt(half2);
        if (lens > 0.0)
            grow += 1.001 * lens * std::max(1.0, far / focal - 1.0);
        grow += 1.001 * (sway + (far + sway) * turn);
        half2 = 0;
        for (int i = 0; i < 3; i++) // (the sphere fallback below uses the grown half extents)
            half2 += (half[i] + grow) * (half[i] + grow);
    }
    bool inFront = true;
    double lo2[2] = { INFINITY, INFINITY }, hi2[2] = { -INFINITY, -INFINITY };
    for (int corner = 0; corner < 8 && inFront; corner++) {
        double p[3], pc[3];
        for (int i = 0; i < 3; i++)
            p[i] = centre[i] + (((corner >> i) & 1) ? 1.0 : -1.0) * (half[i] + grow);
        toCamera(p, pc);
        if (!(pc[2] > 1e-3 * (extent + dist))) {
            inFront = false;
            break;
        }
        for (int a = 0; a < 2; a++) {
            lo2[a] = std::min(lo2[a], pc[a] / pc[2]);
            hi2[a] = std::max(hi2[a], pc[a] / pc[2]);
        }
    }
    if (inFront) {
        for (int a = 0; a < 2; a++)
            sLo[a] = lo2[a], sHi[a] = hi2[a];
    } else {
        const double r = std::sqrt(half2) * 1.001 + 1e-3 * dist + 1e-6;
        if (!(dist > r))
            return false;
        const double kHalfPi = 1.57079632679489661923;
        for (int a = 0; a < 2; a++) {
            const double d2 = std::sqrt(cc[a] * cc[a] + cc[2] * cc[2]);
            if (!(d2 > r))
                continue; // the sphere's shadow on this plane contains the camera: no bound on this axis
            const double ang = std::atan2(cc[a], cc[2]), spread = std::asin(r / d2);
            if (ang - spread >= kHalfPi || ang + spread <= -kHalfPi) { // entirely behind the camera
                x0 = y0 = 1;
                x1 = y1 = 0;
                return true;
            }
            sLo[a] = (ang - spread <= -kHalfPi + 1e-9) ? -INFINITY : std::tan(ang - spread);
            sHi[a] = (ang + spread >= kHalfPi - 1e-9) ? INFINITY : std::tan(ang + spread);
        }
    }
    auto toPixel = [](double v) { return int32_t(std::min(std::max(v, -1e9), 1e9)); };
    // image plane x = -ndcX * half_width, y = ndcY * half_height; pixel = (ndc + 1) / 2 * size
    const double fx0 = (-sHi[0] / double(cam->half_width) + 1.0) * 0.5 * W, fx1 = (-sLo[0] / double(cam->half_width) + 1.0) * 0.5 * W;
    const double fy0 = (sLo[1] / double(cam->half_height) + 1.0) * 0.5 * H, fy1 = (sHi[1] / double(cam->half_height) + 1.0) * 0.5 * H;
    x0 = toPixel(std::floor(fx0) - 2.0);
    x1 = toPixel(std::ceil(fx1) + 2.0);
    y0 = toPixel(std::floor(fy0) - 2.0);
    y1 = toPixel(std::ceil(fy1) + 2.0);
    return true;
}

// Memory of a deferred frame: a head per camera sample, and pools for shading points, their
// lights' results and the tapes, sized from the frame (half as many shading points as camera
// samples, two tape chunks per point) and doubled after a frame that ran over.
int ensureDeferPools(DeviceCtx* c, size_t headWords, uint32_t nCounting)
{
    DeviceCtx::DeferPools& P = c->defer;
    if (!P.counters) {
        HIP_TRY(hipMalloc((void**)&P.counters, kDeferControlWords * sizeof(uint32_t)));
        if (hipHostMalloc((void**)&P.overflowHost, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            P.overflowHost = nullptr; // (the pools then never grow: such frames keep falling back)
        } else {
            *P.overflowHost = 0;
        }
    }
    if (P.overflowHost && *P.overflowHost) { // the previous deferred frame ran over
        P.wantPoints = std::min<uint64_t>(uint64_t(P.pointsPerPool) * 2, (1ull << 28) / kPools);
        *P.overflowHost = 0;
    }
    uint64_t perPool = std::min<uint64_t>(std::max<uint64_t>(headWords / 2, 1ull << 20), 1ull << 25) / kPools;
    perPool = std::max<uint64_t>(perPool, P.wantPoints);
    if (const char* e = getenv("RTX_DEFER_POINTS")) // diagnostic: shading points per pool
        perPool = uint64_t(std::max(1, atoi(e)));
    const bool growHeads = headWords > P.headWords, growPools = perPool > P.pointsPerPool || (getenv("RTX_DEFER_POINTS") && perPool != P.pointsPerPool), growContrib = growPools || nCounting > P.contribLights;
    if (growHeads || growPools || growContrib)
        HIP_TRY(hipStreamSynchronize(c->stream)); // nothing queued may still use what is freed below
    if (growHeads) {
        if (P.heads)
            (void)hipFree(P.heads);
        P.heads = nullptr;
        P.headWords = 0;
        // (the camera pass's hits, two words per sample, follow the heads)
        HIP_TRY(hipMalloc((void**)&P.heads, (((headWords + 1) & ~size_t(1)) + 2 * headWords) * sizeof(uint32_t)));
        P.headWords = headWords;
    }
    if (growPools) {
        if (P.points)
            (void)hipFree(P.points);
        if (P.tape)
            (void)hipFree(P.tape);
        P.points = nullptr;
        P.tape = nullptr;
        P.pointsPerPool = P.tapePerPool = 0;
        HIP_TRY(hipMalloc((void**)&P.points, size_t(perPool) * kPools * sizeof(ShadePoint)));
        if (hipMalloc((void**)&P.tape, size_t(perPool) * 2 * kPools * kTapeChunk * sizeof(TapeEntry)) != hipSuccess) {
            (void)hipFree(P.points);
            P.points = nullptr;
            setErr("out of device memory for the deferred frame's pools");
            return RTX_ERR_DEVICE;
        }
        P.pointsPerPool = uint32_t(perPool);
        P.tapePerPool = uint32_t(perPool * 2);
    }
    if (growContrib) {
        if (P.contrib)
            (void)hipFree(P.contrib);
        P.contrib = nullptr;
        P.contribLights = 0;
        HIP_TRY(hipMalloc((void**)&P.contrib, size_t(P.pointsPerPool) * kPools * nCounting * 3 * sizeof(float)));
        P.contribLights = nCounting;
    }
    return RTX_OK;
}

template <int STACK>
void launchLights(DeviceCtx* c, const LightArgs& L, bool count)
{
    RTX_IF_COUNT(count) {
        const int grid = gridFor(c, (const void*)lightKernel<STACK, true>);
        hipLaunchKernelGGL((lightKernel<STACK, true>), dim3(grid), dim3(kBlock), 0, c->stream, L);
    } else {
        const int grid = gridFor(c, (const void*)lightKernel<STACK, false>);
        hipLaunchKernelGGL((lightKernel<STACK, false>), dim3(grid), dim3(kBlock), 0, c->stream, L);
    }
}

int renderImpl(const rtx_scene* scene, const rtx_camera* cam, const rtx_features* f, const rtx_render_params* p, float* dOut, bool compact, rtx_stats* stats, double t0, bool fixedArena = false)
{
    DeviceCtx* c = g_ctx;
    Arena* a = fixedArena ? &c->fixed : &c->heap;
    std::lock_guard<std::mutex> launchLock(c->launch);
    TileGeom g;
    if (int rc = tileGeom(p, g))
        return rc;
    const int driver = p->driver;
    if (driver < RTX_DRIVER_PLAIN || driver > RTX_DRIVER_COMPOSED) {
        setErr("unknown driver %d", driver);
        return RTX_ERR_INVALID;
    }
    if (fixedArena) {
        const uint64_t blocks = uint64_t(g.tilesMine) * uint64_t(g.ts / 8) * uint64_t(g.ts / 8);
        if (p->width > kFixedMaxDim || p->height > kFixedMaxDim || blocks > kFixedMaxBlocks) {
            setErr("rtx_render_device_static: frame %dx%d (%llu blocks of 8x8 pixels with tile padding) exceeds the fixed workspace (%dx%d pixels, %u blocks)", p->width,
                p->height, (unsigned long long)blocks, kFixedMaxDim, kFixedMaxDim, kFixedMaxBlocks);
            return RTX_ERR_LIMIT;
        }
    }
    const int stack = pickStack(scene->numLevels);
    if (stack < 0) {
        setErr("BVH has %d levels; the device traversal stack holds 64", scene->numLevels);
        return RTX_ERR_LIMIT;
    }
    RenderArgs A;
    A.sc = scene->d;
    A.f = *f;
    A.cam = *cam;
    A.prev = p->prev_camera;
    A.W = p->width;
    A.H = p->height;
    A.driver = driver;
    // the composed driver's dimensions (1 = off); the single drivers are the composition with one dimension on
    const bool composed = driver == RTX_DRIVER_COMPOSED;
    const int64_t nJitter = composed ? std::max(p->samples, 1) : 1, nLens = composed ? std::max(p->lens_samples, 1) : 1, nTime = composed ? std::max(p->time_steps, 1) : 1;
    if (nJitter * nLens * nTime > (int64_t(1) << 20)) {
        setErr("composed driver: %lld x %lld x %lld samples per pixel exceed the limit of 2^20", (long long)nTime, (long long)nLens, (long long)nJitter);
        return RTX_ERR_LIMIT;
    }
    A.jitterSamples = int32_t(nJitter), A.lensSamples = int32_t(nLens), A.timeSteps = int32_t(nTime);
    A.drawsPerSample = (nTime > 1 ? 1 : 0) + (nJitter > 1 ? 2 : 0) + (nLens > 1 ? 3 : 0);
    A.transparency = (driver == RTX_DRIVER_TRANSPARENCY || (composed && p->transparency != 0)) ? 1 : 0;
    const bool lensOn = driver == RTX_DRIVER_DOF || nLens > 1, posesOn = driver == RTX_DRIVER_MOTION_BLUR || nTime > 1;
    const bool sampled = driver == RTX_DRIVER_MRAYS || driver == RTX_DRIVER_DOF || driver == RTX_DRIVER_MOTION_BLUR || (composed && A.drawsPerSample > 0);
    A.normalise = !sampled ? NORM_SINGLE : (driver == RTX_DRIVER_MOTION_BLUR || (composed && nJitter == 1 && nLens == 1)) ? NORM_DIVIDE : NORM_RECIPROCAL;
    A.depth = p->ray_depth >= 0 ? p->ray_depth : ((driver == RTX_DRIVER_MRAYS || driver == RTX_DRIVER_DOF || (composed && (nJitter > 1 || nLens > 1))) ? 0 : 1);
    if (A.depth > kMaxDepth) {
        setErr("ray_depth %d exceeds the limit of %d", A.depth, kMaxDepth);
        return RTX_ERR_LIMIT;
    }
    A.samples = !sampled ? 1 : composed ? int32_t(nJitter * nLens * nTime) : (p->samples > 0 ? p->samples : (driver == RTX_DRIVER_MRAYS ? 64 : driver == RTX_DRIVER_DOF ? 100 : 7));
    A.shadowSamples = p->shadow_samples > 0 ? p->shadow_samples : 50;
    if (int rc = checkEnvLight(scene, f, A.shadowSamples))
        return rc;
    A.glossyRays = p->glossy_rays > 0 ? p->glossy_rays : 5;
    A.seed = p->seed;
    A.aperture = p->aperture > 0.f ? p->aperture : 0.1f;
    A.focal = p->focal_length > 0.f ? p->focal_length : 2.0f;
    A.tileSize = g.ts;
    A.tileRank = g.rank;
    A.tileWorld = g.world;
    A.tilesX = g.tilesX;
    A.tilesY = g.tilesY;
    A.tilesMine = g.tilesMine;
    A.compact = compact ? 1 : 0;
    A.out = dOut;
    A.lanesPerPixel = std::min(A.samples, 64);
    A.pixelsPerUnit = 64 / A.lanesPerPixel;
    // A unit must not straddle 8x8 blocks for the blocks to be ordered and skipped: a
    // sample count that does not divide 64 (7 blur steps: 9 pixels per wave) gives up
    // the odd pixel (8 pixels, 56 of 64 lanes).
    while (64 % A.pixelsPerUnit != 0)
        A.pixelsPerUnit--;
    const int mostPixelsPerUnit = A.pixelsPerUnit;
    // With area lights every shading point costs the wave a traversal step per 64
    // light samples however few camera samples it carries, so a unit holds few
    // pixels: short units balance better and their camera rays are the small part.
    // Frames with area lights are rendered in three passes (walk, lights, resolve: see Deferred)
    // where the walk's control flow covers the features in play.
    uint32_t nCounting = 0, drawsPerPoint = 0;
    for (const int32_t type : scene->lightTypes)
        if (type == RTX_LIGHT_POINT) {
            nCounting++;
        } else if (f->enable_soft_shadow) { // segment / parallelogram lights only count with soft shadows (src/light.cpp:154,167)
            nCounting++;
            drawsPerPoint += (type == RTX_LIGHT_SEGMENT ? 1u : 2u) * uint32_t(A.shadowSamples);
        }
    const uint64_t outPixels = compact ? uint64_t(g.tilesMine) * g.ts * g.ts : uint64_t(A.W) * uint64_t(A.H);
    const bool deferred = !fixedArena && f->enable_shading && f->enable_soft_shadow && scene->hasAreaLight && !f->enable_environment_mapping
        && !A.transparency && nCounting >= 1 && nCounting <= uint32_t(kMaxDeferLights) && outPixels * uint64_t(A.samples) < 0xffffffffull
        && g.tilesMine > 0 && !getenv("RTX_NO_DEFER");
    if (deferred && f->enable_recursive && f->enable_glossy_reflection && A.depth > 0 && A.glossyRays > 1) {
        // A deferred frame's walk is bound by its deepest camera sample, whose tree of up to
        // glossyRays^depth shading points is walked in order (the draws are numbered along that
        // walk): the deeper the tree, the fewer pixels share a wave, so that a wave serves few
        // fans per pass and the frame does not wait for waves that hold several deep pixels.
        double tree = 1;
        for (int i = 0; i < A.depth && tree < 1e6; i++)
            tree *= A.glossyRays;
        A.pixelsPerUnit = std::min(A.pixelsPerUnit, tree <= 25 ? 8 : tree <= 125 ? 2 : 1);
    }
    if (f->enable_soft_shadow && scene->hasAreaLight && A.shadowSamples >= 16 && !deferred) {
        A.pixelsPerUnit = std::min(A.pixelsPerUnit, 8);
        // Under glossy recursion a camera sample walks a tree of up to glossyRays^depth
        // shading points one after the other (its draws are numbered in that order), and the
        // lanes of a wave take turns at the cooperative light sampling: the deeper the tree,
        // the fewer pixels share a wave, so that a frame is not left waiting for a few waves
        // that hold several deep pixels each.
        if (f->enable_recursive && f->enable_glossy_reflection && A.depth > 0 && A.glossyRays > 1) {
            double tree = 1;
            for (int i = 0; i < A.depth && tree < 1e6; i++)
                tree *= A.glossyRays;
            A.pixelsPerUnit = std::min(A.pixelsPerUnit, tree <= 25 ? 4 : tree <= 125 ? 2 : 1);
        }
    }
    if (const char* e = getenv("RTX_PIXELS_PER_UNIT")) { // diagnostic: 1, 2, 4, ... 64
        const int v = atoi(e);
        if (v >= 1 && v <= mostPixelsPerUnit && 64 % v == 0)
            A.pixelsPerUnit = v;
    }
    A.rounds = (A.samples + 63) / 64;
    const uint64_t seqPixels = uint64_t(g.tilesMine) * g.ts * g.ts;
    const uint64_t units = (seqPixels + A.pixelsPerUnit - 1) / A.pixelsPerUnit;
    if (units > 0xffffffffull) {
        setErr("frame too large for one launch");
        return RTX_ERR_LIMIT;
    }
    A.work.nUnits = uint32_t(units);
    A.work = makeQueue(c, a, A.work.nUnits);
    A.stats = a->stats;

    // the full kernel carries the reflection / transparency / area-light /
    // texture / environment machinery; the lean one is enough otherwise
    // (a switch that cannot take effect does not count: recursion with a ray depth of 0, which is what the
    // reference's jittered and depth-of-field drivers pass; texture mapping in a scene without textures)
    const bool recursiveOn = f->enable_recursive && A.depth > 0, texturesOn = f->enable_texture_mapping && scene->hasTextures;
    const bool full = recursiveOn || f->enable_soft_shadow || texturesOn || f->enable_environment_mapping || A.transparency;
    // (glossy fans are traced side by side where a wave holds few pixels, i.e. under area
    // lights: with a full wave of paths the lanes' own traversals already run side by side)
    const bool fans = f->enable_recursive && f->enable_glossy_reflection && A.depth > 0 && A.glossyRays > 1 && A.glossyRays <= kFanMax
        && A.pixelsPerUnit * A.lanesPerPixel <= 8 && !getenv("RTX_NO_FANS");
    // (the lens samples of the depth-of-field driver make no coherent bundle; 64 neighbouring pixels
    // with one sample each share enough of the tree only while triangles are large on the screen --
    // measured at 3840x2160: together wins up to ~0.3 triangles per pixel, 0.60 against 0.87 ms at
    // 0.04, and loses beyond, 1.41 against 1.26 ms at 1.2; RTX_LEAN_WALK=0/1 overrides)
    const bool sparse = double(scene->d.nTris) <= 0.3 * double(A.W) * double(A.H);
    // (the motion-blur driver's samples of a pixel come from as many camera poses: a bundle only
    // while the poses stay close.  How far they stray from the middle pose, in pixels at the
    // distance of what the camera looks at -- measured, 7 samples: together wins below some 60-80
    // pixels on 10 M triangles at 3840x2160 (5.3 against 7.8 ms at 8 pixels, 11.7 against 12.0 at
    // 81) and loses beyond (35 against 18 ms at 116), and up to some 200 pixels where triangles
    // are large on the screen (headline scene: 1.29 against 1.45 ms at 200))
    double swayPixels = 0.0;
    if (posesOn) {
        double dl = 0.0, turn = 0.0;
        for (int i = 0; i < 3; i++) {
            const double dLook = double(cam->look_at[i]) - double(p->prev_camera.look_at[i]);
            dl += dLook * dLook;
            turn += 0.5 * std::fabs(double(cam->euler[i]) - double(p->prev_camera.euler[i]));
        }
        const double dDist = double(cam->distance) - double(p->prev_camera.distance), distMid = std::fabs(double(cam->distance) + 0.5 * dDist);
        const double sway = 0.5 * std::sqrt(dl) + 0.5 * std::fabs(dDist) + (distMid + 0.5 * std::fabs(dDist)) * turn;
        swayPixels = sway / (std::max(distMid, 1e-9) * std::max(double(cam->half_height), 1e-9)) * 0.5 * double(A.H);
        if (!std::isfinite(swayPixels))
            swayPixels = INFINITY;
    }
    // (The lens samples of a pixel start all over the lens and cross at the focal point: how far they lie
    // apart at the distance of what the camera looks at, in pixels.  Where triangles are large on the screen
    // such a bundle still shares its way through the tree -- headline scene, 16 lens samples, aperture 0.1 focused at 2 from 3 away (25 pixels): 2.05
    // against 2.37 ms lane by lane; on the 10 M-triangle scene at 3840x2160 it is 32 against 10 ms.)
    double lensPixels = 0.0;
    if (lensOn) {
        const double dist = std::fabs(double(cam->distance)), focal = std::fabs(double(A.focal));
        const double spread = 0.5 * std::sqrt(3.0) * std::fabs(double(A.aperture)) * (std::fabs(dist - focal) / std::max(focal, 1e-9) + 0.25);
        lensPixels = spread / (std::max(dist, 1e-9) * std::max(double(cam->half_height), 1e-9)) * 0.5 * double(A.H);
        if (!std::isfinite(lensPixels))
            lensPixels = INFINITY;
    }
    const bool bundles = lensOn ? (sparse && swayPixels + lensPixels <= 64.0) : (!posesOn || swayPixels <= (sparse ? 256.0 : 64.0));
    const bool together = getenv("RTX_LEAN_WALK") ? atoi(getenv("RTX_LEAN_WALK")) != 0 : bundles && (A.lanesPerPixel >= 2 || sparse);
    // (the motion-blur driver's lean kernels carry the per-sample camera's trigonometry: modes 6 / 7)
    // (every composed frame goes to the kernels that carry per-sample poses, the only lean ones that read its parameters)
    // (one level of mirror reflection under point lights, coherent camera rays: the lean mirror path, mode 10)
    const bool leanMirror = recursiveOn && !f->enable_glossy_reflection && A.depth == 1 && !f->enable_soft_shadow && !texturesOn && !f->enable_environment_mapping && !A.transparency
        && together && !getenv("RTX_NO_LEAN_MIRROR");
    // (the transparency chain under point lights where a wave holds a few pixels' samples: the lean layers path, modes 11 / 12)
    const bool leanLayers = A.transparency && !recursiveOn && !f->enable_soft_shadow && !texturesOn && !f->enable_environment_mapping
        && A.lanesPerPixel >= (getenv("RTX_LEAN_LAYERS_FROM") ? atoi(getenv("RTX_LEAN_LAYERS_FROM")) : 4);
    const int mode = !full ? ((posesOn || composed) ? (together ? 6 : 7) : (together ? 0 : 5)) : leanMirror ? 10 : leanLayers ? (together ? 11 : 12) : (f->enable_soft_shadow && f->enable_environment_mapping) ? 2 : fans ? 3 : 1;
    // Mode 8 = mode 0 with every ray of the frame read from an origin's frame: the camera rays
    // of these drivers all start at the camera position, the shadow rays at their point light.
    // (The fixed-workspace entry point must not allocate and goes without: same arithmetic,
    // same pixels.)
    A.camFrame = OriginFrame { nullptr, nullptr };
    int launchMode = mode;
    if (mode == 0 && !fixedArena && !originFramesOff() && (driver == RTX_DRIVER_PLAIN || driver == RTX_DRIVER_MRAYS) && f->enable_accel_structure && scene->d.nTris > 0) {
        bool lightsHaveFrames = scene->lightTypes.size() <= size_t(kLightFrames);
        for (size_t li = 0; li < scene->lightTypes.size() && lightsHaveFrames; li++) // (other kinds do not count in the lean kernels)
            lightsHaveFrames = scene->lightTypes[li] != RTX_LIGHT_POINT || scene->d.lightFrame[li].nodes != nullptr;
        if (lightsHaveFrames) {
            if (int rc = ensureCameraFrame(c, scene, cam->position, A.camFrame))
                return rc;
            if (A.camFrame.nodes)
                launchMode = 8;
        }
    }
    // The lean mirror and layers paths read their camera rays' walk from the camera position's frame when
    // there is one (every camera ray of the frame must start there: no lens, no per-sample poses); their
    // point lights' shadow rays go by the lights' frames wherever the scene has them.
    if ((mode == 10 || mode == 11) && !fixedArena && !originFramesOff() && !lensOn && !posesOn && f->enable_accel_structure && scene->d.nTris > 0)
        if (int rc = ensureCameraFrame(c, scene, cam->position, A.camFrame))
            return rc;
    A.runIf = nullptr;
    A.defer = Deferred {};
    size_t walkLds = 0;
    if (deferred) {
        if (int rc = ensureDeferPools(c, size_t(outPixels) * size_t(A.samples), nCounting))
            return rc;
        const DeviceCtx::DeferPools& P = c->defer;
        A.defer.heads = P.heads;
        A.defer.tape = P.tape;
        A.defer.points = P.points;
        A.defer.contrib = P.contrib;
        A.defer.counters = P.counters;
        A.defer.tapePerPool = P.tapePerPool;
        A.defer.pointsPerPool = P.pointsPerPool;
        A.defer.nCounting = nCounting;
        A.defer.drawsPerPoint = drawsPerPoint;
        // glossy fans traced side by side: their hits wait in LDS, for as many levels as 64 KB hold
        // (deeper fans, and scenes with spheres, whose hits are not triangle records: ray by ray)
        A.defer.fanLevels = 0;
        A.defer.leafFans = getenv("RTX_NO_LEAF_FANS") ? 0u : 1u; // (diagnostic switch)
        A.defer.frontier = (f->enable_accel_structure && scene->d.frontierOk && scene->d.waveRoot >= 0 && !getenv("RTX_NO_FRONTIER")) ? 1u : 0u;
        // (few fans at a time, so that each ray's group has three or more lanes: measured with 8 pixels per
        // wave at depth 2, 4 fans per iteration 4.98 ms against 5.9 ms with all 12)
        A.defer.fansPerRound = A.defer.frontier ? 4 : 64;
        if (const char* e = getenv("RTX_FANS_PER_ROUND")) // diagnostic
            A.defer.fansPerRound = uint32_t(std::max(1, atoi(e)));
        if (f->enable_recursive && f->enable_glossy_reflection && A.depth > 0 && A.glossyRays > 1 && A.glossyRays <= kFanMax && scene->d.nSpheres == 0 && !getenv("RTX_NO_FANS")) {
            const size_t perLevel = size_t(A.glossyRays) * size_t(kBlock / 64) * size_t(A.pixelsPerUnit * A.lanesPerPixel) * 8;
            A.defer.fanLevels = uint32_t(std::min<size_t>(size_t(A.depth), (64 * 1024) / perLevel));
            walkLds = size_t(A.defer.fanLevels) * perLevel;
        }
    }

    // wave-uniform block arithmetic and the culling rectangle
    A.unitsPerBlockLog2 = -1;
    if (64 % A.pixelsPerUnit == 0) {
        A.unitsPerBlockLog2 = 0;
        while ((A.pixelsPerUnit << A.unitsPerBlockLog2) < 64)
            A.unitsPerBlockLog2++;
    }
    A.blocksPerRow = uint32_t(g.ts / 8);
    A.blocksPerTile = A.blocksPerRow * A.blocksPerRow;
    A.divBlocksPerTile = makeFastDiv(A.blocksPerTile);
    A.divTilesX = makeFastDiv(uint32_t(g.tilesX));
    A.divBlocksPerRow = makeFastDiv(A.blocksPerRow);
    // A miss is black and nothing but the BVH's triangles can be hit: then blocks whose
    // camera rays (of every sample: lens positions, poses of the exposure) cannot reach
    // the root box are skipped.
    A.cullOn = 0;
    A.cullX0 = A.cullY0 = 0;
    A.cullX1 = A.cullY1 = 0;
    A.nCutRects = 0;
    // (depth of field: sample origins lie within half the diagonal of the aperture cube of the camera position)
    const double lens = lensOn ? 0.5 * std::sqrt(3.0) * double(A.aperture) * 1.0001 : 0.0, lensFocal = double(A.focal);
    // (motion blur: the samples' poses run from the current one to one full pose change beyond
    // it; the rectangle is taken from the pose halfway, grown by how far the others stray from it)
    rtx_camera cullCam = *cam;
    double sway = 0.0, turn = 0.0;
    if (posesOn) {
        double dl = 0, e[3], q[4];
        for (int i = 0; i < 3; i++) {
            const double dLook = double(cam->look_at[i]) - double(p->prev_camera.look_at[i]), dEuler = double(cam->euler[i]) - double(p->prev_camera.euler[i]);
            cullCam.look_at[i] = float(double(cam->look_at[i]) + 0.5 * dLook);
            e[i] = double(cam->euler[i]) + 0.5 * dEuler;
            cullCam.euler[i] = float(e[i]);
            dl += dLook * dLook;
            turn += 0.5 * std::fabs(dEuler);
        }
        const double dDist = double(cam->distance) - double(p->prev_camera.distance), distMid = double(cam->distance) + 0.5 * dDist;
        cullCam.distance = float(distMid);
        // glm::quat(euler), glm/detail/type_quat.inl:208-217
        const double cx = std::cos(0.5 * e[0]), cy = std::cos(0.5 * e[1]), cz = std::cos(0.5 * e[2]), sx = std::sin(0.5 * e[0]), sy = std::sin(0.5 * e[1]), sz = std::sin(0.5 * e[2]);
        q[0] = cx * cy * cz + sx * sy * sz;
        q[1] = sx * cy * cz - cx * sy * sz;
        q[2] = cx * sy * cz + sx * cy * sz;
        q[3] = cx * cy * sz - sx * sy * cz;
        // position = lookAt + q * (0, 0, -distance): v + 2 (w (qv x v) + qv x (qv x v))
        const double v[3] = { 0, 0, -distMid };
        const double uv[3] = { q[2] * v[2] - q[3] * v[1], q[3] * v[0] - q[1] * v[2], q[1] * v[1] - q[2] * v[0] };
        const double uuv[3] = { q[2] * uv[2] - q[3] * uv[1], q[3] * uv[0] - q[1] * uv[2], q[1] * uv[1] - q[2] * uv[0] };
        for (int i = 0; i < 3; i++) {
            cullCam.position[i] = float(double(cullCam.look_at[i]) + v[i] + 2.0 * (q[0] * uv[i] + uuv[i]));
            cullCam.quat[i + 1] = float(q[i + 1]);
        }
        cullCam.quat[0] = float(q[0]);
        turn = turn * 1.001 + 1e-5; // (float poses on the device, double here)
        sway = 0.5 * std::sqrt(dl) + 0.5 * std::fabs(dDist) + (std::fabs(distMid) + 0.5 * std::fabs(dDist)) * turn;
        sway = sway * 1.001 + 1e-5 * (1.0 + std::fabs(distMid));
        if (!std::isfinite(sway) || !std::isfinite(turn))
            sway = turn = INFINITY; // (no rectangle can be bounded then)
    }
    const rtx_camera* const cullFrom = &cullCam;
    // (every driver: a camera ray that misses everything is black in all of them, src/render.cpp:61-64,
    // src/transparency.cpp:9-11)
    if (!f->enable_environment_mapping && scene->d.nSpheres == 0
        && A.unitsPerBlockLog2 >= 0 && !getenv("RTX_NO_CULL"))
    {
        if (scene->d.rootRef == kEmptyRoot) { // nothing to hit at all
            A.cullOn = 1;
            A.cullX0 = A.cullY0 = 1;
        } else {
            A.cullOn = cullRectangle(scene->d.rootLo, scene->d.rootHi, cullFrom, A.W, A.H, A.cullX0, A.cullX1, A.cullY0, A.cullY1, lens, lensFocal, sway, turn) ? 1 : 0;
        }
        // the finer rectangles; a box that cannot be bounded on the screen (the camera is
        // inside or beside it) sees every pixel, and then only the root rectangle counts
        if (A.cullOn && !getenv("RTX_NO_FINE_CULL")) {
            bool bounded = true;
            for (const auto& box : scene->cutBoxes) {
                int32_t r[4];
                if (!cullRectangle(box.data(), box.data() + 3, cullFrom, A.W, A.H, r[0], r[1], r[2], r[3], lens, lensFocal, sway, turn)) {
                    bounded = false;
                    break;
                }
                if (r[1] < r[0] || r[3] < r[2])
                    continue; // entirely behind the camera
                std::memcpy(A.cutRects[A.nCutRects++], r, sizeof(r));
            }
            if (!bounded || scene->cutBoxes.size() < 2)
                A.nCutRects = 0;
        }
    }

    // skip-ahead tables for the drivers that draw from the sequential engine
    A.jumpX = a->jump;
    A.jumpY = a->jump + a->jumpMaxDim;
    A.jumpS = a->jump + 2 * a->jumpMaxDim;
    A.jumpSamples = 0;
    if (sampled) {
        const int d = composed ? A.drawsPerSample : driver == RTX_DRIVER_MRAYS ? 2 : driver == RTX_DRIVER_DOF ? 3 : 1;
        const int ns = std::min(A.samples, a->jumpMaxSamples);
        A.jumpSamples = ns;
        if (a->jumpW != A.W || a->jumpH != A.H || a->jumpS != A.samples || a->jumpD != d) {
            // (filled by a kernel on the stream: no host table, no copy from pageable memory and no wait, so
            // the first frame of a shape is queued like every other one)
            const uint64_t perPixel = uint64_t(d) * uint64_t(A.samples);
            const uint64_t g = hostPowMod(16807, perPixel), gRow = hostPowMod(g, uint64_t(A.W)), gS = hostPowMod(16807, uint64_t(d));
            const uint32_t entries = uint32_t(std::max(std::max(A.W, A.H), ns));
            hipLaunchKernelGGL(jumpTablesKernel, dim3((entries + 255u) / 256u), dim3(256), 0, c->stream, a->jump, uint32_t(a->jumpMaxDim), uint32_t(A.W), uint32_t(A.H), uint32_t(ns), g, gRow, gS);
            HIP_TRY(hipGetLastError());
            a->jumpW = A.W, a->jumpH = A.H, a->jumpS = A.samples, a->jumpD = d;
        }
    }
    // Longest blocks first: some units (grazing shadow rays, silhouettes) run fifty times
    // the average, and a frame is not done before the last of them, so the 8x8 blocks
    // are started in the order of the longest unit each held on the previous frame of
    // this shape.  The measurement rides along in the kernel; the order is made on the
    // device, on the stream, with no host round trip.  The image does not depend on it.
    A.blockOrder = nullptr;
    A.blockCost = nullptr;
    bool ordered = false;
    A.blocksMine = uint32_t(g.tilesMine) * A.blocksPerTile;
    A.costStride = (A.blocksMine + 63u) / 64u;
    A.costFloor = 0;
    if (A.unitsPerBlockLog2 >= 0 && A.blocksMine > 1 && A.blocksMine <= a->maxBlocks && !getenv("RTX_NO_BLOCK_ORDER")) {
        A.blockCost = a->blockCost;
        const bool measured = !(a->costW != A.W || a->costH != A.H || a->costTile != g.ts || a->costRank != g.rank || a->costWorld != g.world || a->costDriver != driver);
        if (!measured) {
            // Nothing measured yet for frames of this shape.  Where blocks can be told
            // unable to see the scene the frame is ordered all the same (every other block
            // counts as never measured, i.e. expensive, and keeps its place in the list),
            // so that those blocks stay out of the queue; otherwise it runs in list order.
            HIP_TRY(hipMemsetAsync(a->blockCost, 0, size_t(A.costStride) * 64 * sizeof(uint32_t), c->stream));
            a->costW = A.W, a->costH = A.H, a->costTile = g.ts, a->costRank = g.rank, a->costWorld = g.world, a->costDriver = driver;
        }
        ordered = measured || A.cullOn;
        if (ordered) {
            auto envClass = [](const char* name, uint32_t fallback) { const char* e = getenv(name); return e ? uint32_t(atoi(e)) : fallback; };
            const uint32_t nSteps = (A.blocksMine + 63u) / 64u, groups = (A.blocksMine + 255u) / 256u;
            hipLaunchKernelGGL(classifyBlocksKernel, dim3(groups), dim3(256), 0, c->stream, A, a->blockClass, a->classTable, nSteps, a->tickets, uint32_t((kTicketBytes + kStatBytes) / sizeof(uint32_t)));
            hipLaunchKernelGGL(scanBlocksKernel, dim3(1), dim3(kScanThreads), 0, c->stream, a->classTable, nSteps, A.unitsPerBlockLog2, a->queueBounds, uint32_t(c->numCUs) * 16u,
                envClass("RTX_SINGLE_CLASS", kLastSingleClass), envClass("RTX_HEAVY_CLASS", kLastHeavyClass));
            hipLaunchKernelGGL(placeBlocksKernel, dim3(groups), dim3(256), 0, c->stream, a->blockClass, A.blocksMine, a->classTable, nSteps, a->blockOrder);
            A.blockOrder = a->blockOrder;
            A.costFloor = kCostFloorTicks;
            A.work.plan = a->queueBounds;
            A.work.nChunks = A.work.nUnits; // (for the grid size: at most)
        }
    }
    if (!ordered)
        HIP_TRY(hipMemsetAsync(a->tickets, 0, kTicketBytes + kStatBytes, c->stream)); // the statistics follow the tickets
    const bool count = wantCounters();
    auto launch = [&](const RenderArgs& R, int m, size_t lds) {
        withStack(stack, [&](auto S) { launchRender<decltype(S)::value>(c, R, count, m, lds); });
    };
    if (deferred && A.work.nUnits) {
        const Deferred& D = A.defer;
        HIP_TRY(hipMemsetAsync(D.heads, 0xff, size_t(outPixels) * size_t(A.samples) * sizeof(uint32_t), c->stream));
        HIP_TRY(hipMemsetAsync(D.counters, 0, kDeferControlWords * sizeof(uint32_t), c->stream));
        if (stats)
            HIP_TRY(hipEventRecord(c->ev0, c->stream)); // kernel_ms covers the passes
        // A walk that holds one or two pixels per wave (deep glossy trees) would trace each camera
        // ray with a lane or two: a camera pass traces them a full wave of pixels at a time first
        // -- the same query per ray, so the same hits -- and the walk picks them up.  (Measured on
        // the glossy depth 4 frame: 14.1 -> 12.9 ms per frame with two frames in flight, 17.7 ->
        // 17.9 ms alone; with eight pixels per wave, depth 2, it cost 7 % and is not taken.)
        const int cameraPassUpTo = getenv("RTX_CAMERA_PASS_MAX") ? atoi(getenv("RTX_CAMERA_PASS_MAX")) : 2; // diagnostic
        if (scene->d.nSpheres == 0 && A.pixelsPerUnit * A.lanesPerPixel <= cameraPassUpTo && A.pixelsPerUnit < mostPixelsPerUnit && !getenv("RTX_NO_CAMERA_PASS")) {
            A.defer.primary = reinterpret_cast<uint2*>(c->defer.heads + ((c->defer.headWords + 1) & ~size_t(1)));
            RenderArgs Cm = A;
            Cm.pixelsPerUnit = mostPixelsPerUnit;
            Cm.unitsPerBlockLog2 = 0;
            while ((Cm.pixelsPerUnit << Cm.unitsPerBlockLog2) < 64)
                Cm.unitsPerBlockLog2++;
            Cm.work = makeQueue(c, a, uint32_t((seqPixels + Cm.pixelsPerUnit - 1) / Cm.pixelsPerUnit));
            Cm.blockOrder = nullptr; // (list order, nothing measured: the pass is short)
            Cm.blockCost = nullptr;
            Cm.costFloor = 0;
            launch(Cm, 9, 0);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemsetAsync(a->tickets, 0, kTicketBytes, c->stream)); // the work queue starts over for the walk
        }
        // The walk's long units run for milliseconds (a deep glossy sample's fans one after the other): a wave
        // that claimed its next chunk ahead would sit on it that long while other waves idle (measured: units
        // of 156 fans early in the sequence began 4 ms into an 8 ms walk).  Only the cheap region's chunks,
        // whose round trip is what a unit costs, are claimed ahead.
        A.work.aheadFrom = getenv("RTX_CLAIM_AHEAD_FROM") ? A.work.aheadFrom : uint32_t(kRegions - 1);
        launch(A, 4, walkLds);
        HIP_TRY(hipGetLastError());
        uint32_t slot = 0, offset = 0;
        for (size_t li = 0; li < scene->lightTypes.size(); li++) {
            const int32_t type = scene->lightTypes[li];
            if (type != RTX_LIGHT_POINT && !f->enable_soft_shadow)
                continue;
            uint32_t* const control = D.counters + kDeferCounterWords + size_t(slot) * (kDeferTicketWords + kDeferUnitWords + kDeferPlanWords);
            LightArgs L;
            L.sc = scene->d;
            L.f = *f;
            L.defer = D;
            L.light = int32_t(li);
            L.slot = int32_t(slot);
            L.nS = type == RTX_LIGHT_POINT ? 1 : A.shadowSamples;
            L.counterOffset = offset;
            L.unitStart = control + kDeferTicketWords;
            L.work.tickets = control;
            L.work.nUnits = 0xffffffffu; // (the plan, made on the device after the walk, says how many)
            L.work.chunk = 1;
            L.work.nChunks = 0xffffffffu;
            L.work.plan = control + kDeferTicketWords + kDeferUnitWords;
            L.work.aheadFrom = 0;
            L.stats = a->stats;
            hipLaunchKernelGGL(planLightsKernel, dim3(1), dim3(64), 0, c->stream, D, uint32_t(std::min(L.nS, 64)), control + kDeferTicketWords, control + kDeferTicketWords + kDeferUnitWords);
            withStack(stack, [&](auto S) { launchLights<decltype(S)::value>(c, L, count); });
            HIP_TRY(hipGetLastError());
            if (type != RTX_LIGHT_POINT)
                offset += (type == RTX_LIGHT_SEGMENT ? 1u : 2u) * uint32_t(A.shadowSamples);
            slot++;
        }
        ResolveArgs R;
        R.defer = D;
        R.out = dOut;
        R.nPixels = uint32_t(outPixels);
        R.samples = A.samples;
        R.normalise = A.normalise;
        hipLaunchKernelGGL(resolveKernel, dim3((R.nPixels + 255u) / 256u), dim3(256), 0, c->stream, R);
        // A frame whose pools ran over (the flag says so on the device) is rendered again by the
        // one-kernel path, in the walk's units; that launch returns at once otherwise.
        hipLaunchKernelGGL(dropCountsIfOverflowKernel, dim3(1), dim3(64), 0, c->stream, D, a->stats);
        HIP_TRY(hipMemsetAsync(a->tickets, 0, kTicketBytes, c->stream));
        RenderArgs B = A;
        B.defer = Deferred {};
        B.runIf = deferOverflow(D);
        launch(B, (fans && A.pixelsPerUnit * A.lanesPerPixel <= 8) ? 3 : 1, 0);
        HIP_TRY(hipGetLastError());
        if (c->defer.overflowHost)
            HIP_TRY(hipMemcpyAsync(c->defer.overflowHost, deferOverflow(D), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        if (stats)
            HIP_TRY(hipEventRecord(c->ev1, c->stream));
        return fetchStats(c, a, stats, t0);
    }
    if (stats) // (an event record costs the stream a few microseconds around the kernel: only when the time is asked for)
        HIP_TRY(hipEventRecord(c->ev0, c->stream)); // kernel_ms is the render kernel's time
    if (A.work.nUnits) {
        launch(A, launchMode, 0);
        HIP_TRY(hipGetLastError());
    }
    if (stats)
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
    return fetchStats(c, a, stats, t0);
}

template <int STACK>
void launchShade(DeviceCtx* c, const ShadeArgs& S, bool count, bool envLight)
{
#define RTX_LAUNCH(CNT, ENVV)                                                                                  \
    {                                                                                                          \
        const void* k = (const void*)shadeKernel<STACK, CNT, ENVV>;                                            \
        const int grid = std::max(1, std::min<int>(gridFor(c, k), int((S.R.work.nChunks + 3) / 4)));           \
        hipLaunchKernelGGL((shadeKernel<STACK, CNT, ENVV>), dim3(grid), dim3(kBlock), 0, c->stream, S);        \
    }
    RTX_IF_COUNT(count && envLight)
        RTX_LAUNCH(true, true)
    else RTX_IF_COUNT(count)
        RTX_LAUNCH(true, false)
    else if (envLight)
        RTX_LAUNCH(false, true)
    else
        RTX_LAUNCH(false, false)
#undef RTX_LAUNCH
}

int shadeImpl(const rtx_scene* scene, const rtx_ray* dRays, size_t n, const rtx_features* f, const rtx_render_params* p, float* dOut, rtx_stats* stats, double t0)
{
    DeviceCtx* c = g_ctx;
    std::lock_guard<std::mutex> launchLock(c->launch);
    if (n > (size_t(1) << 32) - 64) {
        setErr("rtx_shade: %zu rays exceed the 2^32 limit (the ray index keys the sampler)", n);
        return RTX_ERR_LIMIT;
    }
    const int stack = pickStack(scene->numLevels);
    if (stack < 0) {
        setErr("BVH has %d levels; the device traversal stack holds 64", scene->numLevels);
        return RTX_ERR_LIMIT;
    }
    ShadeArgs S;
    memset(&S, 0, sizeof(S));
    RenderArgs& A = S.R;
    A.sc = scene->d;
    A.f = *f;
    A.driver = RTX_DRIVER_PLAIN;
    A.depth = p->ray_depth >= 0 ? p->ray_depth : 0; // getFinalColor's default rayDepth (src/render.h:29)
    if (A.depth > kMaxDepth) {
        setErr("ray_depth %d exceeds the limit of %d", A.depth, kMaxDepth);
        return RTX_ERR_LIMIT;
    }
    A.samples = 1;
    A.shadowSamples = p->shadow_samples > 0 ? p->shadow_samples : 50;
    if (int rc = checkEnvLight(scene, f, A.shadowSamples))
        return rc;
    A.glossyRays = p->glossy_rays > 0 ? p->glossy_rays : 5;
    A.seed = p->seed;
    const Arena* a = &c->heap;
    A.work.nUnits = uint32_t((n + 63) / 64);
    A.work = makeQueue(c, a, A.work.nUnits);
    A.stats = a->stats;
    S.rays = dRays;
    S.out = dOut;
    S.n = n;
    HIP_TRY(hipMemsetAsync(a->tickets, 0, kTicketBytes + kStatBytes, c->stream)); // the statistics follow the tickets
    if (stats)
        HIP_TRY(hipEventRecord(c->ev0, c->stream));
    if (n) {
        const bool count = wantCounters();
        withStack(stack, [&](auto K) { launchShade<decltype(K)::value>(c, S, count, f->enable_soft_shadow && f->enable_environment_mapping); });
        HIP_TRY(hipGetLastError());
    }
    if (stats)
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
    return fetchStats(c, a, stats, t0);
}

// What both ways of making a scene (rtx_scene_create, rtx_scene_build_device) check of a
// description besides its tree: everything the kernels index must be in range.
bool shadingInputsOk(const rtx_scene_desc* d)
{
    for (uint32_t i = 0; i < d->n_tris; i++) {
        if (d->tris[i].material < 0 || uint32_t(d->tris[i].material) >= d->n_materials) {
            setErr("rtx_scene_create: triangle %u has material %d of %u", i, d->tris[i].material, d->n_materials);
            return false;
        }
        if (d->tris[i].id < 0 || uint32_t(d->tris[i].id) >= d->n_tris) {
            setErr("rtx_scene_create: triangle %u has id %d of %u", i, d->tris[i].id, d->n_tris);
            return false;
        }
    }
    for (uint32_t i = 0; i < d->n_spheres; i++)
        if (d->spheres[i].material < 0 || uint32_t(d->spheres[i].material) >= d->n_materials) {
            setErr("rtx_scene_create: sphere %u has material %d of %u", i, d->spheres[i].material, d->n_materials);
            return false;
        }
    for (uint32_t i = 0; i < d->n_materials; i++)
        if (d->materials[i].texture >= int32_t(d->n_textures)) {
            setErr("rtx_scene_create: material %u has texture %d of %u", i, d->materials[i].texture, d->n_textures);
            return false;
        }
    if (d->envmap.texture >= int32_t(d->n_textures)) {
        setErr("rtx_scene_create: environment map texture %d of %u", d->envmap.texture, d->n_textures);
        return false;
    }
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rtx_texture& t = d->textures[i];
        if (t.levels < 1 || t.levels > RTX_MAX_MIP_LEVELS) {
            setErr("rtx_scene_create: texture %u has %d levels", i, t.levels);
            return false;
        }
        for (int l = 0; l < t.levels; l++)
            if (uint64_t(t.offset[l]) + t.count[l] > d->n_texels || t.width[l] < 0 || t.height[l] < 0) {
                setErr("rtx_scene_create: texture %u level %d outside the texel pool", i, l);
                return false;
            }
    }
    if (d->n_env_bins && !d->env_bins) {
        setErr("rtx_scene_create: %u environment bins but NULL array", d->n_env_bins);
        return false;
    }
    for (uint32_t i = 0; i < d->n_lights; i++)
        if (d->lights[i].type < RTX_LIGHT_POINT || d->lights[i].type > RTX_LIGHT_PARALLELOGRAM) {
            setErr("rtx_scene_create: light %u has unknown type %d", i, d->lights[i].type);
            return false;
        }

    return true;
}

// Materials, textures, lights, spheres and the environment map of a scene: uploaded as they are.
int uploadShading(rtx_scene* s, const rtx_scene_desc* d)
{
    int rc = upload(s, d->materials, d->n_materials, &s->d.mats);
    rc = rc ? rc : upload(s, d->textures, d->n_textures, &s->d.textures);
    rc = rc ? rc : upload(s, d->texels, size_t(d->n_texels) * 3, &s->d.texels);
    rc = rc ? rc : upload(s, d->lights, d->n_lights, &s->d.lights);
    rc = rc ? rc : upload(s, d->spheres, d->n_spheres, &s->d.spheres);
    const float* bins = nullptr;
    rc = rc ? rc : upload(s, d->env_bins, size_t(d->n_env_bins) * 4, &bins);
    s->d.envBins = reinterpret_cast<const float4*>(bins);
    s->d.env = d->envmap;
    s->d.nLights = int32_t(d->n_lights);
    s->d.nEnvBins = int32_t(d->n_env_bins);
    for (uint32_t i = 0; i < d->n_materials; i++)
        s->hasTextures = s->hasTextures || d->materials[i].texture >= 0;
    for (uint32_t i = 0; i < d->n_lights; i++) {
        s->hasAreaLight = s->hasAreaLight || d->lights[i].type != RTX_LIGHT_POINT;
        s->lightTypes.push_back(d->lights[i].type);
        s->lightPos.push_back({ d->lights[i].p[0][0], d->lights[i].p[0][1], d->lights[i].p[0][2] });
    }
    s->d.nSpheres = int32_t(d->n_spheres);
    s->d.nMaterials = int32_t(d->n_materials);
    return rc;
}

// Keeps one wave busy for `ticks` of the 100 MHz wall clock (streamsOverlap).
// the device's restatements of the host library's routines, for rtx_selfcheck_trig
__global__ void __launch_bounds__(256) trigKernel(const float* x, const float* y, uint32_t n, float* out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) {
        out[i] = rtxd::hostSinf(x[i]);
        out[size_t(n) + i] = rtxd::hostCosf(x[i]);
        out[2 * size_t(n) + i] = rtxd::hostTanf(x[i]);
        out[3 * size_t(n) + i] = rtxd::hostAcosf(x[i]);
        out[4 * size_t(n) + i] = rtxd::hostAtan2f(y[i], x[i]);
    }
}

__global__ void __launch_bounds__(64) spinKernel(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(8);
}

// Whether kernels queued on streams `a` and `b` run at the same time.  The runtime spreads a
// process's streams over a few hardware queues, and two streams that land on the same queue
// take turns however independent their work is.  Measured: one wave spins for 300 us on each
// stream; together they take 300 us on different queues and 600 us on one.  1 = they overlap,
// 0 = they take turns, < 0 = error.  The device is current.
int streamsOverlap(hipStream_t a, hipStream_t b)
{
    if (a == b)
        return 0;
    constexpr unsigned long long kSpinTicks = 30000; // 300 us
    hipEvent_t e0 = nullptr, e1 = nullptr, eb = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess)
        e = hipEventCreate(&e1);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&eb, hipEventDisableTiming);
    float ms = 0.f;
    for (int pass = 0; pass < 2 && e == hipSuccess; pass++) { // (the first pass loads the kernel and wakes the queues)
        const unsigned long long ticks = pass ? kSpinTicks : 1ull;
        e = hipEventRecord(e0, a);
        if (e != hipSuccess)
            break;
        hipLaunchKernelGGL(spinKernel, dim3(1), dim3(64), 0, a, ticks);
        hipLaunchKernelGGL(spinKernel, dim3(1), dim3(64), 0, b, ticks);
        e = hipGetLastError();
        if (e == hipSuccess)
            e = hipEventRecord(eb, b);
        if (e == hipSuccess)
            e = hipStreamWaitEvent(a, eb, 0);
        if (e == hipSuccess)
            e = hipEventRecord(e1, a);
        if (e == hipSuccess)
            e = hipEventSynchronize(e1);
        if (e == hipSuccess)
            e = hipEventElapsedTime(&ms, e0, e1);
    }
    for (hipEvent_t ev : { e0, e1, eb })
        if (ev)
            (void)hipEventDestroy(ev);
    if (e != hipSuccess) {
        setErr("stream overlap probe failed: %s", hipGetErrorString(e));
        return RTX_ERR_DEVICE;
    }
    return ms < 1.6f * float(kSpinTicks) * 1e-5f ? 1 : 0;
}

// Makes a context (one frame slot of one GPU) ready: its stream, events and working memory.
// Everything is made in locals and committed at the end, so that a failure halfway leaves
// nothing behind and a later call starts from scratch.  The caller holds g_ctxMutex and has
// made `device` current.
int initCtx(DeviceCtx* c, int device, int slot)
{
    if (c->device >= 0)
        return RTX_OK;
    hipStream_t stream = nullptr;
    hipEvent_t events[3] = {};
    void* mem[6] = {};
    const size_t bytes[6] = { kTicketBytes + kStatBytes, // one allocation: cleared together
        (2 * size_t(kJumpMaxDim) + kJumpMaxSamples) * sizeof(uint32_t), (size_t(kMaxOrderedBlocks) + 64) * sizeof(uint32_t),
        size_t(kMaxOrderedBlocks) * sizeof(uint32_t), size_t(kMaxOrderedBlocks) + 4, size_t(kCostClasses) * (kMaxOrderedBlocks / 64) * sizeof(uint32_t) + 256 };
    hipDeviceProp_t prop;
    unsigned char* fixedBase = nullptr;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    // a further slot is there to run beside the others: its stream must not share a hardware
    // queue with theirs (a handful of tries: the runtime deals its queues out in turn)
    hipStream_t rejected[8] = {};
    for (int tries = 0; e == hipSuccess && slot > 0 && tries < 8; tries++) {
        bool beside = true;
        for (int k = 0; k < kSlots && beside; k++) {
            const DeviceCtx* other = &g_ctxs[device][k];
            if (other != c && other->device >= 0 && streamsOverlap(other->ownStream, stream) == 0)
                beside = false;
        }
        if (beside)
            break;
        rejected[tries] = stream;
        stream = nullptr;
        e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    }
    for (hipStream_t r : rejected)
        if (r)
            (void)hipStreamDestroy(r);
    for (int i = 0; i < 3 && e == hipSuccess; i++)
        e = hipEventCreate(&events[i]);
    for (int i = 0; i < 6 && e == hipSuccess; i++)
        e = hipMalloc(&mem[i], bytes[i]);
    // this GPU's copy of the file-scope array (each device holds its own, at its own address);
    // it belongs to slot 0
    if (e == hipSuccess && slot == 0)
        e = hipGetSymbolAddress(reinterpret_cast<void**>(&fixedBase), HIP_SYMBOL(scratch));
    if (e != hipSuccess) {
        setErr("rtx_init(%d) failed: %s", device, hipGetErrorString(e));
        for (void* m : mem)
            if (m)
                (void)hipFree(m);
        for (hipEvent_t ev : events)
            if (ev)
                (void)hipEventDestroy(ev);
        if (stream)
            (void)hipStreamDestroy(stream);
        return RTX_ERR_DEVICE;
    }
    c->numCUs = prop.multiProcessorCount;
    c->ownStream = c->stream = stream;
    c->ev0 = events[0], c->ev1 = events[1], c->evSwitch = events[2];
    Arena& h = c->heap;
    h.tickets = static_cast<uint32_t*>(mem[0]);
    h.stats = reinterpret_cast<unsigned long long*>(h.tickets + kTicketBytes / sizeof(uint32_t));
    h.jump = static_cast<uint32_t*>(mem[1]);
    h.jumpMaxDim = kJumpMaxDim, h.jumpMaxSamples = kJumpMaxSamples;
    h.blockCost = static_cast<uint32_t*>(mem[2]);
    h.blockOrder = static_cast<uint32_t*>(mem[3]);
    h.blockClass = static_cast<uint8_t*>(mem[4]);
    h.classTable = static_cast<uint32_t*>(mem[5]);
    h.queueBounds = h.classTable + size_t(kCostClasses) * (kMaxOrderedBlocks / 64);
    h.maxBlocks = uint32_t(kMaxOrderedBlocks);
    c->fixedBase = fixedBase;
    if (fixedBase) {
        Arena& x = c->fixed;
        x.tickets = reinterpret_cast<uint32_t*>(fixedBase + kFixedTickets);
        x.stats = reinterpret_cast<unsigned long long*>(x.tickets + kTicketBytes / sizeof(uint32_t));
        x.queueBounds = reinterpret_cast<uint32_t*>(fixedBase + kFixedBounds);
        x.jump = reinterpret_cast<uint32_t*>(fixedBase + kFixedJump);
        x.jumpMaxDim = kFixedMaxDim, x.jumpMaxSamples = kFixedMaxSamples;
        x.blockCost = reinterpret_cast<uint32_t*>(fixedBase + kFixedCost);
        x.blockOrder = reinterpret_cast<uint32_t*>(fixedBase + kFixedOrder);
        x.blockClass = fixedBase + kFixedClass;
        x.classTable = reinterpret_cast<uint32_t*>(fixedBase + kFixedTable);
        x.maxBlocks = kFixedMaxBlocks;
    }
    c->slot = slot;
    c->device = device;
    return RTX_OK;
}

} // namespace

extern "C" {

const char* rtx_last_error(void) { return g_err; }

int rtx_init(int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        setErr("no HIP device available");
        return RTX_ERR_DEVICE;
    }
    if (device < 0 || device >= count || device >= 64) {
        setErr("device %d out of range (0..%d)", device, count - 1);
        return RTX_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(g_ctxMutex);
    DeviceCtx* c = &g_ctxs[device][0];
    if (int rc = initCtx(c, device, 0))
        return rc;
    g_ctx = c;
    g_err[0] = 0;
    return RTX_OK;
}

int rtx_select_slot(int slot)
{
    if (int rc = checkCtx(nullptr))
        return rc;
    if (slot < 0 || slot >= kSlots) {
        setErr("rtx_select_slot: slot %d out of range (0..%d)", slot, kSlots - 1);
        return RTX_ERR_INVALID;
    }
    const int device = g_ctx->device;
    std::lock_guard<std::mutex> lock(g_ctxMutex);
    DeviceCtx* c = &g_ctxs[device][slot];
    if (int rc = initCtx(c, device, slot))
        return rc;
    g_ctx = c;
    return RTX_OK;
}

rtx_scene* rtx_scene_create(const rtx_scene_desc* d)
{
    if (checkCtx(nullptr))
        return nullptr;
    if (!d) {
        setErr("rtx_scene_create: NULL description");
        return nullptr;
    }
    // shape checks: everything the kernels index must be in range
    if (d->n_tris && (!d->tris || !d->tri_attrs)) {
        setErr("rtx_scene_create: %u triangles but NULL arrays", d->n_tris);
        return nullptr;
    }
    if (d->n_tris > (1u << 30) || d->n_nodes > (1u << 30)) {
        setErr("rtx_scene_create: scene too large");
        return nullptr;
    }
    {
        const struct { const void* p; uint64_t n; const char* what; } arrays[] = { { d->nodes, d->n_nodes, "nodes" }, { d->materials, d->n_materials, "materials" },
            { d->textures, d->n_textures, "textures" }, { d->texels, d->n_texels, "texels" }, { d->lights, d->n_lights, "lights" }, { d->spheres, d->n_spheres, "spheres" } };
        for (const auto& a : arrays)
            if (a.n && !a.p) {
                setErr("rtx_scene_create: %llu %s but a NULL array", (unsigned long long)a.n, a.what);
                return nullptr;
            }
    }
    const bool emptyTree = d->root_ref == INT32_MIN;
    int walkedLevels = 0; // inner nodes on the longest path from the root, plus the leaf
    bool boxesNested = true; // (NaN planes make it false)
    if (!emptyTree) {
        auto refOk = [&](int32_t r) { return r >= 0 ? uint32_t(r) < d->n_nodes : uint32_t(~r) < d->n_tris; };
        if (!refOk(d->root_ref)) {
            setErr("rtx_scene_create: root reference %d out of range", d->root_ref);
            return nullptr;
        }
        for (uint32_t i = 0; i < d->n_nodes; i++)
            if (!refOk(d->nodes[i].ref0) || !refOk(d->nodes[i].ref1)) {
                setErr("rtx_scene_create: node %u has a child reference out of range", i);
                return nullptr;
            }
        // every leaf must be terminated inside the array
        if (d->n_tris && !(d->tris[d->n_tris - 1].flags & RTX_TRI_LAST_IN_LEAF)) {
            setErr("rtx_scene_create: last triangle record lacks the end-of-leaf flag");
            return nullptr;
        }
        // The nodes reachable from the root must form a tree: the traversal's stack is sized by
        // its depth, and a node reached twice (a cycle, or a shared subtree that could hide one)
        // would let a walk run on forever or past its stack.
        std::vector<uint8_t> seen(d->n_nodes, 0);
        std::vector<uint32_t> parentOf(d->n_nodes, 0);
        std::vector<std::pair<int32_t, int>> todo; // (inner node, its depth)
        if (d->root_ref >= 0)
            todo.push_back({ d->root_ref, 1 });
        walkedLevels = 1;
        while (!todo.empty()) {
            const auto [node, depth] = todo.back();
            todo.pop_back();
            if (seen[size_t(node)]) {
                setErr("rtx_scene_create: node %d is reachable twice (the nodes must form a tree)", node);
                return nullptr;
            }
            seen[size_t(node)] = 1;
            walkedLevels = std::max(walkedLevels, depth + 1);
            {
                // do this node's child boxes lie within the box its parent holds for it?
                const float* lo = node == d->root_ref ? d->root_lo : nullptr;
                const float* hi = node == d->root_ref ? d->root_hi : nullptr;
                if (!lo) {
                    const rtx_node& up = d->nodes[parentOf[size_t(node)]];
                    const bool second = up.ref1 == node;
                    lo = second ? up.lo1 : up.lo0;
                    hi = second ? up.hi1 : up.hi0;
                }
                const rtx_node& nd = d->nodes[node];
                for (int a = 0; a < 3; a++)
                    boxesNested = boxesNested && nd.lo0[a] >= lo[a] && nd.lo1[a] >= lo[a] && nd.hi0[a] <= hi[a] && nd.hi1[a] <= hi[a];
            }
            for (const int32_t child : { d->nodes[node].ref0, d->nodes[node].ref1 })
                if (child >= 0) {
                    parentOf[size_t(child)] = uint32_t(node);
                    todo.push_back({ child, depth + 1 });
                }
        }
    }
    if (!shadingInputsOk(d))
        return nullptr;

    rtx_scene* s = new (std::nothrow) rtx_scene;
    if (!s) {
        setErr("out of host memory");
        return nullptr;
    }
    s->device = g_ctx->device;
    s->numLevels = std::max(d->num_levels, walkedLevels); // (never less than the tree really has: it sizes the traversal stack)
    s->numLeaves = d->num_leaves;
    // A shallow cut through the tree, level by level while it stays within
    // kMaxCutBoxes: its boxes together hold every triangle.
    if (d->root_ref >= 0) {
        struct Entry { int32_t ref; std::array<float, 6> box; };
        std::vector<Entry> cut;
        Entry root;
        root.ref = d->root_ref;
        for (int i = 0; i < 3; i++)
            root.box[size_t(i)] = d->root_lo[i], root.box[size_t(3 + i)] = d->root_hi[i];
        cut.push_back(root);
        for (;;) {
            size_t inner = 0;
            for (const Entry& e : cut)
                inner += e.ref >= 0;
            if (inner == 0 || cut.size() + inner > size_t(kMaxCutBoxes))
                break;
            std::vector<Entry> next;
            for (const Entry& e : cut) {
                if (e.ref < 0) {
                    next.push_back(e);
                    continue;
                }
                const rtx_node& n = d->nodes[e.ref];
                Entry a, b;
                a.ref = n.ref0, b.ref = n.ref1;
                for (int i = 0; i < 3; i++) {
                    a.box[size_t(i)] = n.lo0[i], a.box[size_t(3 + i)] = n.hi0[i];
                    b.box[size_t(i)] = n.lo1[i], b.box[size_t(3 + i)] = n.hi1[i];
                }
                next.push_back(a);
                next.push_back(b);
            }
            cut.swap(next);
        }
        for (const Entry& e : cut)
            s->cutBoxes.push_back(e.box);
    }
    std::vector<int32_t> idOrder(d->n_tris);
    for (uint32_t k = 0; k < d->n_tris; k++)
        idOrder[size_t(d->tris[k].id)] = int32_t(k);
    const rtx_node* nodes;
    const rtx_tri* tris;
    const rtx_tri_attr* attrs;
    // device-side record layouts (see loadTri / traverseBvh): both children of a node,
    // and the three vertices and the normal of a triangle, side by side per axis
    std::vector<rtx_node> packedNodes(d->n_nodes);
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        const rtx_node& a = d->nodes[i];
        float* o = reinterpret_cast<float*>(&packedNodes[i]);
        for (int ax = 0; ax < 3; ax++) {
            o[4 * ax + 0] = a.lo0[ax], o[4 * ax + 1] = a.lo1[ax];
            o[4 * ax + 2] = a.hi0[ax], o[4 * ax + 3] = a.hi1[ax];
        }
        // the device's references: triangle records follow the node records in one array
        int32_t refs[4] = { a.ref0, a.ref1, a.cnt0, a.cnt1 };
        for (int c = 0; c < 2; c++)
            if (refs[c] < 0)
                refs[c] = ~int32_t(d->n_nodes + uint32_t(~refs[c]));
        std::memcpy(o + 12, refs, 16);
    }
    std::vector<rtx_tri> packedTris(d->n_tris);
    for (uint32_t i = 0; i < d->n_tris; i++) {
        const rtx_tri& a = d->tris[i];
        float* o = reinterpret_cast<float*>(&packedTris[i]);
        for (int ax = 0; ax < 3; ax++) {
            o[4 * ax + 0] = a.v0[ax], o[4 * ax + 1] = a.v1[ax];
            o[4 * ax + 2] = a.v2[ax], o[4 * ax + 3] = a.n[ax];
        }
        std::memcpy(o + 12, &a.d, 16);
        // the traversal reads "last triangle of its leaf" from the top bit of the id word,
        // which sits beside D, the one other word of this quarter the triangle test needs
        if (a.flags & RTX_TRI_LAST_IN_LEAF) {
            const uint32_t marked = uint32_t(a.id) | 0x80000000u;
            std::memcpy(o + 13, &marked, 4);
        }
    }
    static_assert(sizeof(rtx_node) == 64 && sizeof(rtx_tri) == 64, "one record size");
    std::vector<rtx_node> records(size_t(d->n_nodes) + size_t(d->n_tris));
    if (d->n_nodes)
        std::memcpy(records.data(), packedNodes.data(), size_t(d->n_nodes) * sizeof(rtx_node));
    if (d->n_tris)
        std::memcpy(records.data() + d->n_nodes, packedTris.data(), size_t(d->n_tris) * sizeof(rtx_tri));
    // The tree of the wave-shared walk (see DScene::waveNodes): every inner node merged with its
    // inner children, in eight mirrored copies with the children sorted nearest first.
    struct Wide {
        int32_t ref[4];   // >= 0: binary node to be merged next, < 0: leaf reference of the flat tree; INT32_MIN: unused
        int32_t cnt[4];   // triangles of a leaf child
        float lo[4][3], hi[4][3];
        int n;
    };
    bool frontierOk = d->n_tris <= kFrontierMaxIndex && d->root_ref >= 0;
    std::vector<Wide> wide;
    int waveStackNeed = 0;
    if (d->root_ref >= 0) {
        // (iterative, parents before children: a node's index is its position in `wide`)
        std::vector<int32_t> binaryOf { d->root_ref }; // binary node each wide node is made from
        for (size_t at = 0; at < binaryOf.size(); at++) {
            const rtx_node& top = d->nodes[binaryOf[at]];
            Wide w;
            w.n = 0;
            auto add = [&](int32_t ref, int32_t cnt, const float* lo, const float* hi) {
                w.ref[w.n] = ref;
                w.cnt[w.n] = ref < 0 ? cnt : 0;
                if (ref < 0 && frontierOk) {
                    // (the other walks find a leaf's end by its last triangle's flag; the frontier walk goes by
                    // the description's count, which must say the same)
                    const uint64_t k = uint32_t(~ref);
                    frontierOk = cnt >= 1 && uint32_t(cnt) <= kFrontierMaxLeaf && k + uint64_t(cnt) <= d->n_tris;
                    for (int64_t j = 0; frontierOk && j < cnt; j++)
                        frontierOk = ((d->tris[k + uint64_t(j)].flags & RTX_TRI_LAST_IN_LEAF) != 0) == (j == cnt - 1);
                }
                std::memcpy(w.lo[w.n], lo, 12);
                std::memcpy(w.hi[w.n], hi, 12);
                w.n++;
            };
            for (int side = 0; side < 2; side++) {
                const int32_t ref = side ? top.ref1 : top.ref0;
                if (ref >= 0) {
                    const rtx_node& mid = d->nodes[ref];
                    add(mid.ref0, mid.cnt0, mid.lo0, mid.hi0);
                    add(mid.ref1, mid.cnt1, mid.lo1, mid.hi1);
                } else {
                    add(ref, side ? top.cnt1 : top.cnt0, side ? top.lo1 : top.lo0, side ? top.hi1 : top.hi0);
                }
            }
            for (int c = 0; c < w.n; c++)
                if (w.ref[c] >= 0) { // an inner node: becomes a wide node of its own
                    binaryOf.push_back(w.ref[c]);
                    w.ref[c] = int32_t(binaryOf.size() - 1);
                } // (a leaf keeps its reference: ~first triangle in leaf order)
            wide.push_back(w);
        }
        // entries the walk's stack can hold at once: all siblings of every node on a path
        std::vector<int> need(wide.size(), 0);
        for (size_t i = wide.size(); i-- > 0;) {
            int below = 0;
            for (int c = 0; c < wide[i].n; c++)
                if (wide[i].ref[c] >= 0)
                    below = std::max(below, need[size_t(wide[i].ref[c])]);
            need[i] = wide[i].n - 1 + below;
        }
        waveStackNeed = need[0];
    }
    const size_t nWide = wide.size();
    std::vector<float> waveNodes(nWide * 8 * 32, 0.f), waveTris((size_t(d->n_tris) + 1) * 24, 0.f); // (one record of padding: the walk fetches triangles in pairs)
    for (uint32_t oct = 0; oct < 8; oct++)
        for (size_t i = 0; i < nWide; i++) {
            const Wide& w = wide[i];
            // this octant's mirror image of the boxes, and the children nearest first: by the
            // mirrored box centre along the diagonal every mirrored direction has a positive
            // share of (any order gives the same hits; a good one finds them sooner)
            float lo[4][3], hi[4][3], along[4];
            int order[4] = { 0, 1, 2, 3 };
            for (int c = 0; c < w.n; c++) {
                along[c] = 0.f;
                for (int a = 0; a < 3; a++) {
                    const bool neg = (oct >> a) & 1u;
                    lo[c][a] = neg ? -w.hi[c][a] : w.lo[c][a];
                    hi[c][a] = neg ? -w.lo[c][a] : w.hi[c][a];
                    along[c] += lo[c][a] + hi[c][a];
                }
            }
            std::stable_sort(order, order + w.n, [&](int x, int y) { return along[x] < along[y]; });
            float* o = &waveNodes[(size_t(oct) * nWide + i) * 32];
            for (int slot = 0; slot < 4; slot++) {
                const bool used = slot < w.n;
                const int c = used ? order[slot] : 0;
                for (int a = 0; a < 3; a++) {
                    o[8 * a + slot] = used ? lo[c][a] : INFINITY;       // (an unused slot: the empty box, entered by no ray)
                    o[8 * a + 4 + slot] = used ? hi[c][a] : -INFINITY;
                }
                const int32_t ref = used ? w.ref[c] : INT32_MIN;
                std::memcpy(o + 24 + slot, &ref, 4);
            }
            std::memcpy(o + 28, &w.n, 4); // number of children
            // the triangle counts of leaf children, slot by slot, 16 bits each (traverseFrontier)
            uint32_t counts[2] = { 0u, 0u };
            for (int slot = 0; slot < w.n; slot++)
                counts[slot >> 1] |= (uint32_t(w.cnt[order[slot]]) & 0xffffu) << (16 * (slot & 1));
            std::memcpy(o + 29, counts, 8);
        }
    for (uint32_t i = 0; i < d->n_tris; i++) {
        const rtx_tri& a = d->tris[i];
        float* o = &waveTris[size_t(i) * 24];
        o[0] = a.n[0], o[1] = a.n[1], o[2] = a.n[2], o[3] = a.d;
        for (int ax = 0; ax < 3; ax++) {
            o[4 + ax] = a.v0[ax];
            o[8 + ax] = a.v1[ax];
            // single IEEE subtractions, exactly the ones the triangle test makes per ray
            o[12 + ax] = a.v2[ax] - a.v0[ax];
            o[16 + ax] = a.v2[ax] - a.v1[ax];
            o[20 + ax] = a.v1[ax] - a.v0[ax];
        }
        std::memcpy(o + 7, reinterpret_cast<const float*>(&packedTris[i]) + 13, 4); // id word, top bit = last of its leaf
    }
    if (waveStackNeed > 2 * pickStack(std::max(d->num_levels, walkedLevels)) && pickStack(std::max(d->num_levels, walkedLevels)) > 0) {
        setErr("rtx_scene_create: the merged tree needs %d stack entries, more than its %d levels allow for", waveStackNeed, std::max(d->num_levels, walkedLevels));
        delete s;
        return nullptr;
    }
    const float *dWaveNodes = nullptr, *dWaveTris = nullptr;
    int rc = upload(s, records.data(), records.size(), &nodes);
    rc = rc ? rc : upload(s, waveNodes.data(), waveNodes.size(), &dWaveNodes);
    rc = rc ? rc : upload(s, waveTris.data(), waveTris.size(), &dWaveTris);
    tris = reinterpret_cast<const rtx_tri*>(nodes ? nodes + d->n_nodes : nullptr);
    rc = rc ? rc : upload(s, d->tri_attrs, d->n_tris, &attrs);
    rc = rc ? rc : upload(s, idOrder.data(), idOrder.size(), &s->d.idOrder);
    rc = rc ? rc : uploadShading(s, d);
    if (rc) {
        rtx_scene_destroy(s);
        return nullptr;
    }
    s->d.nodes = reinterpret_cast<const float4*>(nodes);
    s->d.tris = reinterpret_cast<const float4*>(tris);
    s->d.attrs = reinterpret_cast<const float4*>(attrs);
    s->d.boxesNested = boxesNested ? 1 : 0;
    s->d.frontierOk = frontierOk ? 1 : 0;
    s->d.nWaveNodes = int32_t(nWide);
    s->d.waveRoot = d->root_ref >= 0 ? 0 : (emptyTree ? kEmptyRoot : ~int32_t(uint32_t(~d->root_ref)));
    s->d.waveNodes = reinterpret_cast<const float4*>(dWaveNodes);
    s->d.waveTris = reinterpret_cast<const float4*>(dWaveTris);
    s->d.nNodes = int32_t(d->n_nodes);
    s->d.nTris = int32_t(d->n_tris);
    s->d.rootRef = (d->root_ref < 0 && d->root_ref != kEmptyRoot) ? ~int32_t(d->n_nodes + uint32_t(~d->root_ref)) : d->root_ref;
    memcpy(s->d.rootLo, d->root_lo, sizeof(s->d.rootLo));
    memcpy(s->d.rootHi, d->root_hi, sizeof(s->d.rootHi));
    if (makeLightFrames(s, g_ctx->stream)) {
        rtx_scene_destroy(s);
        return nullptr;
    }
    return s;
}

// The scene with its acceleration structure built on the GPU (see rtx_build.hpp).  d->tris and
// d->tri_attrs hold the triangles in id order (record i has id i); d->nodes / root_* are ignored.
rtx_scene* rtx_scene_build_device(const rtx_scene_desc* d)
{
    using namespace rtxbuild;
    if (checkCtx(nullptr))
        return nullptr;
    if (!d || (d->n_tris && (!d->tris || !d->tri_attrs))) {
        setErr("rtx_scene_build_device: NULL description or triangle arrays");
        return nullptr;
    }
    if (d->n_tris > (1u << 30)) {
        setErr("rtx_scene_build_device: scene too large");
        return nullptr;
    }
    {
        const struct { const void* p; uint64_t n; const char* what; } arrays[] = { { d->materials, d->n_materials, "materials" }, { d->textures, d->n_textures, "textures" },
            { d->texels, d->n_texels, "texels" }, { d->lights, d->n_lights, "lights" }, { d->spheres, d->n_spheres, "spheres" } };
        for (const auto& a : arrays)
            if (a.n && !a.p) {
                setErr("rtx_scene_build_device: %llu %s but a NULL array", (unsigned long long)a.n, a.what);
                return nullptr;
            }
    }
    if (!shadingInputsOk(d))
        return nullptr;
    for (uint32_t i = 0; i < d->n_tris; i++)
        if (uint32_t(d->tris[i].id) != i) {
            setErr("rtx_scene_build_device: triangle record %u carries id %d (the records must come in id order)", i, d->tris[i].id);
            return nullptr;
        }
    const uint32_t n = d->n_tris;
    if (n <= uint32_t(kLeafMax)) {
        // too few triangles for a tree: one leaf (or none), through the ordinary upload
        rtx_scene_desc one = *d;
        std::vector<rtx_tri> tris(d->tris, d->tris + n);
        float lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
        for (uint32_t i = 0; i < n; i++) {
            tris[i].flags = i + 1 == n ? RTX_TRI_LAST_IN_LEAF : 0;
            for (int a = 0; a < 3; a++) {
                const float mn = std::min({ tris[i].v0[a], tris[i].v1[a], tris[i].v2[a] }), mx = std::max({ tris[i].v0[a], tris[i].v1[a], tris[i].v2[a] });
                lo[a] = i ? std::min(lo[a], mn) : mn;
                hi[a] = i ? std::max(hi[a], mx) : mx;
            }
        }
        one.tris = tris.data();
        one.nodes = nullptr;
        one.n_nodes = 0;
        one.root_ref = n ? ~0 : INT32_MIN;
        one.root_cnt = int32_t(n);
        std::memcpy(one.root_lo, lo, 12);
        std::memcpy(one.root_hi, hi, 12);
        one.num_levels = n ? 1 : 0;
        one.num_leaves = n ? 1 : 0;
        return rtx_scene_create(&one);
    }
    DeviceCtx* c = g_ctx;
    std::lock_guard<std::mutex> launchLock(c->launch);
    hipStream_t stream = c->stream;
    std::vector<void*> temps;
    auto release = [&] {
        (void)hipStreamSynchronize(stream);
        for (void* t : temps)
            (void)hipFree(t);
        temps.clear();
    };
    bool ok = true;
    auto grab = [&](size_t bytes) -> void* {
        void* ptr = nullptr;
        if (ok && hipMalloc(&ptr, std::max<size_t>(bytes, 16)) != hipSuccess) {
            (void)hipGetLastError();
            ok = false;
            ptr = nullptr;
        }
        if (ptr)
            temps.push_back(ptr);
        return ptr;
    };
    BuildBuffers B {};
    B.n = n;
    // A linear BVH grows deep chains where triangles crowd (equal Morton codes: up to 32 more
    // levels, one per bit of the position that breaks the tie).  From this depth on, subtrees of up
    // to kDeepLeafMax triangles are leaves, which keeps the tree within the traversal's 64-entry
    // stack.  (Taking such leaves from a shallower depth, to fit a smaller stack, was measured and
    // dropped: on the headline scene leaves of up to 64 triangles from level 14 on cost 1.65 ms a
    // frame against 1.15 ms on the full tree.)
    B.deepFrom = 56;
    if (const char* e = getenv("RTX_BUILD_DEEP_FROM")) // diagnostic
        B.deepFrom = uint32_t(std::max(1, atoi(e)));
    rtx_tri* dTris = static_cast<rtx_tri*>(grab(size_t(n) * sizeof(rtx_tri)));
    rtx_tri_attr* dAttrs = static_cast<rtx_tri_attr*>(grab(size_t(n) * sizeof(rtx_tri_attr)));
    B.tris = dTris;
    B.attrs = dAttrs;
    B.primLo = static_cast<float*>(grab(size_t(n) * 12));
    B.primHi = static_cast<float*>(grab(size_t(n) * 12));
    B.keys = static_cast<uint64_t*>(grab(size_t(n) * 8));
    B.keysSorted = static_cast<uint64_t*>(grab(size_t(n) * 8));
    B.order = static_cast<uint32_t*>(grab(size_t(n) * 4));
    B.orderSorted = static_cast<uint32_t*>(grab(size_t(n) * 4));
    B.left = static_cast<int32_t*>(grab(size_t(n) * 4));
    B.right = static_cast<int32_t*>(grab(size_t(n) * 4));
    B.first = static_cast<uint32_t*>(grab(size_t(n) * 4));
    B.last = static_cast<uint32_t*>(grab(size_t(n) * 4));
    B.parent = static_cast<int32_t*>(grab(size_t(n) * 4));
    B.leafParent = static_cast<int32_t*>(grab(size_t(n) * 4));
    B.arrived = static_cast<uint32_t*>(grab(size_t(n) * 4));
    B.nodeLo = static_cast<float*>(grab(size_t(n) * 12));
    B.nodeHi = static_cast<float*>(grab(size_t(n) * 12));
    B.alive = static_cast<uint32_t*>(grab(size_t(n) * 4));
    B.wide = static_cast<uint32_t*>(grab(size_t(n) * 4));
    B.aliveIndex = static_cast<uint32_t*>(grab(size_t(n) * 4));
    B.wideIndex = static_cast<uint32_t*>(grab(size_t(n) * 4));
    B.scalars = static_cast<uint32_t*>(grab(64));
    float* dCut = static_cast<float*>(grab((1 + size_t(kMaxCutBoxes) * 6) * sizeof(float)));
    size_t sortBytes = 0, scanBytes = 0;
    if (ok) {
        ok = rocprim::radix_sort_pairs(nullptr, sortBytes, B.keys, B.keysSorted, B.order, B.orderSorted, size_t(n), 0u, 63u, stream) == hipSuccess
            && rocprim::exclusive_scan(nullptr, scanBytes, B.alive, B.aliveIndex, 0u, size_t(n - 1), rocprim::plus<uint32_t>(), stream) == hipSuccess;
    }
    void* dSortTemp = grab(std::max(sortBytes, scanBytes));
    if (!ok) {
        setErr("rtx_scene_build_device: out of device memory for %u triangles", n);
        release();
        return nullptr;
    }
    const uint32_t scalarInit[16] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const dim3 perTri((n + 255u) / 256u), perNode((n - 1 + 255u) / 256u), block(256);
    const double tStart = nowMs();
    hipError_t e = hipMemcpyAsync(dTris, d->tris, size_t(n) * sizeof(rtx_tri), hipMemcpyHostToDevice, stream);
    e = e ? e : hipMemcpyAsync(dAttrs, d->tri_attrs, size_t(n) * sizeof(rtx_tri_attr), hipMemcpyHostToDevice, stream);
    e = e ? e : hipMemcpyAsync(B.scalars, scalarInit, sizeof(scalarInit), hipMemcpyHostToDevice, stream);
    e = e ? e : hipMemsetAsync(B.arrived, 0, size_t(n) * 4, stream);
    e = e ? e : hipStreamSynchronize(stream);
    const double tUploaded = nowMs();
    if (!e) {
        hipLaunchKernelGGL(primBoundsKernel, perTri, block, 0, stream, B);
        hipLaunchKernelGGL(mortonKernel, perTri, block, 0, stream, B);
        e = rocprim::radix_sort_pairs(dSortTemp, sortBytes, B.keys, B.keysSorted, B.order, B.orderSorted, size_t(n), 0u, 63u, stream);
    }
    if (!e) {
        hipLaunchKernelGGL(hierarchyKernel, perNode, block, 0, stream, B);
        hipLaunchKernelGGL(refitKernel, perTri, block, 0, stream, B);
        hipLaunchKernelGGL(markKernel, perNode, block, 0, stream, B);
        e = rocprim::exclusive_scan(dSortTemp, scanBytes, B.alive, B.aliveIndex, 0u, size_t(n - 1), rocprim::plus<uint32_t>(), stream);
        e = e ? e : rocprim::exclusive_scan(dSortTemp, scanBytes, B.wide, B.wideIndex, 0u, size_t(n - 1), rocprim::plus<uint32_t>(), stream);
    }
    uint32_t scalars[16] = {}, tail[4] = {};
    e = e ? e : hipMemcpyAsync(scalars, B.scalars, sizeof(scalars), hipMemcpyDeviceToHost, stream);
    e = e ? e : hipMemcpyAsync(&tail[0], B.aliveIndex + (n - 2), 4, hipMemcpyDeviceToHost, stream);
    e = e ? e : hipMemcpyAsync(&tail[1], B.alive + (n - 2), 4, hipMemcpyDeviceToHost, stream);
    e = e ? e : hipMemcpyAsync(&tail[2], B.wideIndex + (n - 2), 4, hipMemcpyDeviceToHost, stream);
    e = e ? e : hipMemcpyAsync(&tail[3], B.wide + (n - 2), 4, hipMemcpyDeviceToHost, stream);
    e = e ? e : hipStreamSynchronize(stream);
    e = e ? e : hipGetLastError();
    if (e) {
        setErr("rtx_scene_build_device: %s", hipGetErrorString(e));
        release();
        return nullptr;
    }
    if (scalars[6]) {
        setErr("rtx_scene_build_device: a triangle has a vertex position that is not finite");
        release();
        return nullptr;
    }
    const uint32_t nNodes = tail[0] + tail[1], nWide = tail[2] + tail[3];
    const int levels = int(scalars[7]) + 2; // inner nodes on the longest path from the root, plus the leaf
    if (pickStack(levels) < 0) {
        setErr("rtx_scene_build_device: the tree has %d levels; the device traversal stack holds 64 (many triangles share one place: use the host builder)", levels);
        release();
        return nullptr;
    }
    rtx_scene* s = new (std::nothrow) rtx_scene;
    if (!s) {
        setErr("out of host memory");
        release();
        return nullptr;
    }
    s->device = c->device;
    s->numLevels = levels;
    s->numLeaves = int(scalars[8]);
    Outputs O {};
    O.nNodes = nNodes;
    O.nWide = nWide;
    auto keep = [&](size_t bytes) -> void* {
        void* ptr = nullptr;
        if (ok && hipMalloc(&ptr, std::max<size_t>(bytes, 16)) != hipSuccess) {
            (void)hipGetLastError();
            ok = false;
            ptr = nullptr;
        }
        if (ptr)
            s->buffers[s->nBuffers++] = ptr;
        return ptr;
    };
    O.records = static_cast<float4*>(keep((size_t(nNodes) + n) * 64));
    O.waveNodes = static_cast<float4*>(keep(size_t(nWide) * 8 * 128));
    O.waveTris = static_cast<float4*>(keep((size_t(n) + 1) * 96)); // (one record of padding: the walk fetches triangles in pairs)
    O.attrs = static_cast<rtx_tri_attr*>(keep(size_t(n) * sizeof(rtx_tri_attr)));
    O.idOrder = static_cast<int32_t*>(keep(size_t(n) * 4));
    float rootBox[6] = {}, cut[1 + kMaxCutBoxes * 6] = {};
    uint32_t bigLeaf = 1; // (set by the records' kernel when a leaf holds more triangles than a frontier item counts)
    if (ok) {
        // (the padding record is read -- never used -- by the pair fetch and copied into origin frames: zero, as rtx_scene_create leaves it)
        e = hipMemsetAsync(reinterpret_cast<unsigned char*>(O.waveTris) + size_t(n) * 96, 0, 96, stream);
        ok = e == hipSuccess;
    }
    if (ok) {
        hipLaunchKernelGGL(triangleRecordsKernel, perTri, block, 0, stream, B, O);
        hipLaunchKernelGGL(nodeRecordsKernel, perNode, block, 0, stream, B, O);
        hipLaunchKernelGGL(wideRecordsKernel, perNode, block, 0, stream, B, O);
        hipLaunchKernelGGL(cutKernel, dim3(1), dim3(1), 0, stream, B, kMaxCutBoxes, dCut);
        e = hipMemcpyAsync(rootBox, B.nodeLo, 12, hipMemcpyDeviceToHost, stream);
        e = e ? e : hipMemcpyAsync(rootBox + 3, B.nodeHi, 12, hipMemcpyDeviceToHost, stream);
        e = e ? e : hipMemcpyAsync(cut, dCut, sizeof(cut), hipMemcpyDeviceToHost, stream);
        e = e ? e : hipMemcpyAsync(&bigLeaf, B.scalars + 9, sizeof(bigLeaf), hipMemcpyDeviceToHost, stream);
        e = e ? e : hipStreamSynchronize(stream);
        e = e ? e : hipGetLastError();
        ok = e == hipSuccess;
    }
    release();
    s->uploadMs = float(tUploaded - tStart);
    s->buildMs = float(nowMs() - tUploaded); // (ends with the records in place and the temporary buffers freed)
    if (ok && uploadShading(s, d) != RTX_OK)
        ok = false;
    if (!ok) {
        if (e)
            setErr("rtx_scene_build_device: %s", hipGetErrorString(e));
        else if (!g_err[0])
            setErr("rtx_scene_build_device: out of device memory");
        rtx_scene_destroy(s);
        return nullptr;
    }
    for (int k = 0; k < int(cut[0]) && k < kMaxCutBoxes; k++) {
        std::array<float, 6> box;
        std::memcpy(box.data(), cut + 1 + k * 6, sizeof(float) * 6);
        s->cutBoxes.push_back(box);
    }
    s->d.nodes = O.records;
    s->d.tris = O.records + size_t(nNodes) * 4;
    s->d.attrs = reinterpret_cast<const float4*>(O.attrs);
    s->d.idOrder = O.idOrder;
    s->d.boxesNested = 1; // every box is the union of what lies below it
    s->d.frontierOk = (bigLeaf == 0 && n <= kFrontierMaxIndex) ? 1 : 0;
    s->d.nWaveNodes = int32_t(nWide);
    s->d.waveRoot = 0;
    s->d.waveNodes = O.waveNodes;
    s->d.waveTris = O.waveTris;
    s->d.nNodes = int32_t(nNodes);
    s->d.nTris = int32_t(n);
    s->d.rootRef = 0;
    std::memcpy(s->d.rootLo, rootBox, 12);
    std::memcpy(s->d.rootHi, rootBox + 3, 12);
    if (makeLightFrames(s, g_ctx->stream)) {
        rtx_scene_destroy(s);
        return nullptr;
    }
    return s;
}

// numLevels / numLeaves of the scene's tree as uploaded or built (BvhInterface::numLevels, numLeaves)
int rtx_scene_tree_info(const rtx_scene* scene, rtx_tree_info* info)
{
    if (!scene || !info) {
        setErr("rtx_scene_tree_info: NULL argument");
        return RTX_ERR_INVALID;
    }
    info->levels = scene->numLevels;
    info->leaves = scene->numLeaves;
    info->upload_ms = scene->uploadMs;
    info->build_ms = scene->buildMs;
    return RTX_OK;
}

void rtx_scene_retain(rtx_scene* s)
{
    if (s) {
        std::lock_guard<std::mutex> lock(g_ctxMutex);
        s->refs++;
    }
}

void rtx_scene_destroy(rtx_scene* s)
{
    if (!s)
        return;
    {
        std::lock_guard<std::mutex> lock(g_ctxMutex);
        if (--s->refs > 0)
            return;
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(s->device);
    for (int i = 0; i < s->nBuffers; i++)
        (void)hipFree(s->buffers[i]);
    if (prev >= 0)
        (void)hipSetDevice(prev);
    delete s;
}

int rtx_trace_device(const rtx_scene* scene, const rtx_ray* dRays, size_t n, rtx_hit* dHits, const rtx_features* f, rtx_stats* stats)
{
    const double t0 = nowMs();
    if (int rc = checkCtx(scene))
        return rc;
    if (!scene || !f || (n && (!dRays || !dHits))) {
        setErr("rtx_trace_device: NULL argument");
        return RTX_ERR_INVALID;
    }
    return traceImpl(scene, dRays, n, dHits, f, stats, t0);
}

int rtx_trace(const rtx_scene* scene, const rtx_ray* rays, size_t n, rtx_hit* hits, const rtx_features* f, rtx_stats* stats)
{
    const double t0 = nowMs();
    if (int rc = checkCtx(scene))
        return rc;
    if (!scene || !f || (n && (!rays || !hits))) {
        setErr("rtx_trace: NULL argument");
        return RTX_ERR_INVALID;
    }
    if (n == 0)
        return traceImpl(scene, nullptr, 0, nullptr, f, stats, t0);
    rtx_ray* dRays = nullptr;
    rtx_hit* dHits = nullptr;
    HIP_TRY(hipMalloc((void**)&dRays, n * sizeof(rtx_ray)));
    if (hipMalloc((void**)&dHits, n * sizeof(rtx_hit)) != hipSuccess) {
        (void)hipFree(dRays);
        setErr("rtx_trace: out of device memory for %zu hits", n);
        return RTX_ERR_DEVICE;
    }
    int rc = RTX_OK;
    if (hipMemcpyAsync(dRays, rays, n * sizeof(rtx_ray), hipMemcpyHostToDevice, g_ctx->stream) != hipSuccess) {
        setErr("rtx_trace: ray upload failed");
        rc = RTX_ERR_DEVICE;
    }
    rc = rc ? rc : traceImpl(scene, dRays, n, dHits, f, stats, t0);
    if (!rc && (hipStreamSynchronize(g_ctx->stream) != hipSuccess || hipMemcpy(hits, dHits, n * sizeof(rtx_hit), hipMemcpyDeviceToHost) != hipSuccess)) {
        setErr("rtx_trace: hit download failed");
        rc = RTX_ERR_DEVICE;
    }
    (void)hipFree(dRays);
    (void)hipFree(dHits);
    if (stats)
        stats->total_ms = nowMs() - t0;
    return rc;
}

int rtx_render_device(const rtx_scene* scene, const rtx_camera* cam, const rtx_features* f, const rtx_render_params* p, float* dOut, rtx_stats* stats)
{
    const double t0 = nowMs();
    if (int rc = checkCtx(scene))
        return rc;
    if (!scene || !cam || !f || !p || !dOut) {
        setErr("rtx_render_device: NULL argument");
        return RTX_ERR_INVALID;
    }
    return renderImpl(scene, cam, f, p, dOut, false, stats, t0);
}

int rtx_render_device_static(const rtx_scene* scene, const rtx_camera* cam, const rtx_features* f, const rtx_render_params* p, float* dOut, rtx_stats* stats)
{
    const double t0 = nowMs();
    if (int rc = checkCtx(scene))
        return rc;
    if (!scene || !cam || !f || !p || !dOut) {
        setErr("rtx_render_device_static: NULL argument");
        return RTX_ERR_INVALID;
    }
    if (g_ctx->slot != 0) {
        setErr("rtx_render_device_static: the fixed workspace belongs to frame slot 0 (this thread has selected slot %d)", g_ctx->slot);
        return RTX_ERR_INVALID;
    }
    if (stats) {
        // reading the counters back waits for the frame and copies to the host: not while the stream is being captured
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(g_ctx->stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) {
            setErr("rtx_render_device_static: stats must be NULL while the stream is being captured (reading them back waits for the frame)");
            return RTX_ERR_INVALID;
        }
        (void)hipGetLastError();
    }
    return renderImpl(scene, cam, f, p, dOut, false, stats, t0, true);
}

int rtx_render_tiles_device(const rtx_scene* scene, const rtx_camera* cam, const rtx_features* f, const rtx_render_params* p, float* dTiles, rtx_stats* stats)
{
    const double t0 = nowMs();
    if (int rc = checkCtx(scene))
        return rc;
    if (!scene || !cam || !f || !p || !dTiles) {
        setErr("rtx_render_tiles_device: NULL argument");
        return RTX_ERR_INVALID;
    }
    return renderImpl(scene, cam, f, p, dTiles, true, stats, t0);
}

int rtx_render(const rtx_scene* scene, const rtx_camera* cam, const rtx_features* f, const rtx_render_params* p, float* out, rtx_stats* stats)
{
    const double t0 = nowMs();
    if (int rc = checkCtx(scene))
        return rc;
    if (!scene || !cam || !f || !p || !out) {
        setErr("rtx_render: NULL argument");
        return RTX_ERR_INVALID;
    }
    TileGeom g;
    if (int rc = tileGeom(p, g))
        return rc;
    const size_t bytes = size_t(p->width) * size_t(p->height) * 3 * sizeof(float);
    DeviceCtx* c = g_ctx;
    std::lock_guard<std::mutex> frameLock(c->hostFrame);
    if (c->frameBytes < bytes) {
        if (c->frame)
            (void)hipFree(c->frame);
        c->frame = nullptr;
        c->frameBytes = 0;
        HIP_TRY(hipMalloc((void**)&c->frame, bytes));
        c->frameBytes = bytes;
    }
    float* const dOut = c->frame;
    int rc = RTX_OK;
    // pixels of other ranks' tiles keep the caller's values
    if (g.world > 1 && (hipMemcpyAsync(dOut, out, bytes, hipMemcpyHostToDevice, g_ctx->stream) != hipSuccess || hipStreamSynchronize(g_ctx->stream) != hipSuccess)) {
        setErr("rtx_render: framebuffer upload failed");
        rc = RTX_ERR_DEVICE;
    }
    rc = rc ? rc : renderImpl(scene, cam, f, p, dOut, false, stats, t0);
    if (!rc && (hipStreamSynchronize(g_ctx->stream) != hipSuccess || hipMemcpy(out, dOut, bytes, hipMemcpyDeviceToHost) != hipSuccess)) {
        setErr("rtx_render: framebuffer download failed");
        rc = RTX_ERR_DEVICE;
    }
    if (stats)
        stats->total_ms = nowMs() - t0;
    return rc;
}

int rtx_shade_device(const rtx_scene* scene, const rtx_ray* dRays, size_t n, const rtx_features* f, const rtx_render_params* p, float* dOut, rtx_stats* stats)
{
    const double t0 = nowMs();
    if (int rc = checkCtx(scene))
        return rc;
    if (!scene || !f || !p || (n && (!dRays || !dOut))) {
        setErr("rtx_shade_device: NULL argument");
        return RTX_ERR_INVALID;
    }
    return shadeImpl(scene, dRays, n, f, p, dOut, stats, t0);
}

int rtx_shade(const rtx_scene* scene, const rtx_ray* rays, size_t n, const rtx_features* f, const rtx_render_params* p, float* out, rtx_stats* stats)
{
    const double t0 = nowMs();
    if (int rc = checkCtx(scene))
        return rc;
    if (!scene || !f || !p || (n && (!rays || !out))) {
        setErr("rtx_shade: NULL argument");
        return RTX_ERR_INVALID;
    }
    if (n == 0)
        return shadeImpl(scene, nullptr, 0, f, p, nullptr, stats, t0);
    rtx_ray* dRays = nullptr;
    float* dOut = nullptr;
    HIP_TRY(hipMalloc((void**)&dRays, n * sizeof(rtx_ray)));
    if (hipMalloc((void**)&dOut, n * 3 * sizeof(float)) != hipSuccess) {
        (void)hipFree(dRays);
        setErr("rtx_shade: out of device memory for %zu colours", n);
        return RTX_ERR_DEVICE;
    }
    int rc = RTX_OK;
    if (hipMemcpyAsync(dRays, rays, n * sizeof(rtx_ray), hipMemcpyHostToDevice, g_ctx->stream) != hipSuccess) {
        setErr("rtx_shade: ray upload failed");
        rc = RTX_ERR_DEVICE;
    }
    rc = rc ? rc : shadeImpl(scene, dRays, n, f, p, dOut, stats, t0);
    if (!rc && (hipStreamSynchronize(g_ctx->stream) != hipSuccess || hipMemcpy(out, dOut, n * 3 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)) {
        setErr("rtx_shade: colour download failed");
        rc = RTX_ERR_DEVICE;
    }
    (void)hipFree(dRays);
    (void)hipFree(dOut);
    if (stats)
        stats->total_ms = nowMs() - t0;
    return rc;
}

int rtx_tiles_per_rank(const rtx_render_params* p)
{
    TileGeom g;
    if (!p) {
        setErr("rtx_tiles_per_rank: NULL argument");
        return RTX_ERR_INVALID;
    }
    if (int rc = tileGeom(p, g))
        return rc;
    return g.tilesPerRank;
}

int rtx_detile_device(const float* dAllTiles, const rtx_render_params* p, float* dOut)
{
    if (int rc = checkCtx(nullptr))
        return rc;
    if (!dAllTiles || !p || !dOut) {
        setErr("rtx_detile_device: NULL argument");
        return RTX_ERR_INVALID;
    }
    TileGeom g;
    if (int rc = tileGeom(p, g))
        return rc;
    const uint64_t n = uint64_t(p->width) * p->height;
    const int block = 256;
    hipLaunchKernelGGL(detileKernel, dim3(uint32_t((n + block - 1) / block)), dim3(block), 0, g_ctx->stream, dAllTiles, dOut, p->width, p->height, g.ts,
        g.tilesX, g.tilesY, g.world, g.tilesPerRank);
    HIP_TRY(hipGetLastError());
    return RTX_OK;
}

int rtx_streams_overlap(void* hipStreamA, void* hipStreamB)
{
    if (int rc = checkCtx(nullptr))
        return rc;
    if (!hipStreamA || !hipStreamB) {
        setErr("rtx_streams_overlap: NULL stream");
        return RTX_ERR_INVALID;
    }
    return streamsOverlap(static_cast<hipStream_t>(hipStreamA), static_cast<hipStream_t>(hipStreamB));
}

int rtx_set_stream(void* hipStream)
{
    if (int rc = checkCtx(nullptr))
        return rc;
    DeviceCtx* c = g_ctx;
    std::lock_guard<std::mutex> launchLock(c->launch);
    const hipStream_t next = hipStream ? static_cast<hipStream_t>(hipStream) : c->ownStream;
    if (next != c->stream) {
        // The work counters, the block order and the other working buffers are shared by
        // every call in this frame slot: what is still queued on the old stream must have
        // finished with them before anything queued on the new stream touches them.
        HIP_TRY(hipEventRecord(c->evSwitch, c->stream));
        HIP_TRY(hipStreamWaitEvent(next, c->evSwitch, 0));
        c->stream = next;
    }
    return RTX_OK;
}

int rtx_selfcheck_trig(const float* x, const float* y, size_t n, float* out)
{
    if (int rc = checkCtx(nullptr))
        return rc;
    if (n > (size_t(1) << 28) || (n && (!x || !y || !out))) {
        setErr("rtx_selfcheck_trig: %zu arguments (at most 2^28), or a null array", n);
        return RTX_ERR_INVALID;
    }
    DeviceCtx* c = g_ctx;
    std::lock_guard<std::mutex> launchLock(c->launch);
    if (n) {
        hipLaunchKernelGGL(trigKernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, c->stream, x, y, uint32_t(n), out);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTX_OK;
}

#ifdef RTX_PROBE
int rtx_debug_walk_probe(void* out, size_t nUnits) // nUnits x 8 words (WalkProbe) of the last deferred frame's walk
{
    if (nUnits > kProbeUnits)
        return RTX_ERR_LIMIT;
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walkProbe), nUnits * sizeof(WalkProbe)) == hipSuccess ? RTX_OK : RTX_ERR_DEVICE;
}
int rtx_debug_wave_walk_stats(unsigned long long* out, int reset) // 8 counts of the wave-shared walk: whole triangle tests, plane stage passed, hit; node visits; pretests, lanes a pretest dropped that the whole test takes (must be 0)
{
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rtxd::g_waveWalkStats), 8 * sizeof(unsigned long long)) != hipSuccess)
        return RTX_ERR_DEVICE;
    if (reset) {
        const unsigned long long zero[8] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(rtxd::g_waveWalkStats), zero, sizeof(zero));
    }
    return RTX_OK;
}
int rtx_debug_probe(void* out, size_t nUnits)
{
    if (nUnits > kProbeUnits)
        return RTX_ERR_LIMIT;
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe), nUnits * sizeof(uint4)) == hipSuccess ? RTX_OK : RTX_ERR_DEVICE;
}
#endif

void* rtx_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError(); // (not an error of the path: the caller falls back to plain memory)
        setErr("rtx_host_alloc: no page-locked memory for %zu bytes", bytes);
        return nullptr;
    }
    return p;
}

void rtx_host_free(void* p)
{
    if (p && hipHostFree(p) != hipSuccess)
        (void)hipGetLastError();
}

int rtx_synchronize(void)
{
    if (int rc = checkCtx(nullptr))
        return rc;
    HIP_TRY(hipStreamSynchronize(g_ctx->stream));
    return RTX_OK;
}

} // extern "C"
