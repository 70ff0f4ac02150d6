// TEST INFRASTRUCTURE (oracle/_ref build only).  Force-included (-include) in
// front of the reference translation units that seed a fresh
// std::random_device + std::mt19937 for every random number
// (src/light.cpp:16-23, src/shading.cpp:56-65, src/environment_mapping.cpp:17-24,
// src/sampling.cpp:8-10).  It renames those two types to harness types so that
// the k-th random_device construction of a camera sample yields a uniform that
// is a pure function of (seed, pixel, sample, k): the same counter-keyed stream
// the HIP kernels and the CPU restatement consume in the reference's
// depth-first order.
#pragma once
#include <cstdint>
#include <random>

extern thread_local uint32_t rtx_shim_seed;
extern thread_local uint32_t rtx_shim_pixel;
extern thread_local uint32_t rtx_shim_sample;
extern thread_local uint32_t rtx_shim_counter;

static inline uint32_t rtx_shim_mix(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// uniform bits for draw number `counter` of sample `sample` of pixel `pixel`
static inline uint32_t rtx_shim_bits(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t counter)
{
    uint32_t h = rtx_shim_mix(seed ^ (pixel * 0x9e3779b1u));
    h = rtx_shim_mix(h ^ (sample * 0xc2b2ae35u) ^ 0x27d4eb2fu);
    return rtx_shim_mix(h ^ (counter * 0x85ebca77u) ^ 0x68bc21ebu);
}

namespace std {
struct rtx_shim_random_device {
    using result_type = uint32_t;
    // returns the draw index; the engine below turns it into bits
    result_type operator()() { return rtx_shim_counter++; }
};
struct rtx_shim_engine {
    using result_type = uint32_t;
    uint32_t index, sub = 0;
    explicit rtx_shim_engine(uint32_t drawIndex) : index(drawIndex) {}
    static constexpr result_type min() { return 0u; }
    static constexpr result_type max() { return 0xffffffffu; }
    // first output = the keyed uniform; later outputs of the same engine are
    // only ever discarded by the reference (src/shading.cpp:61-62)
    result_type operator()()
    {
        uint32_t v = rtx_shim_bits(rtx_shim_seed, rtx_shim_pixel, rtx_shim_sample, index);
        if (sub++) v = rtx_shim_mix(v + sub);
        return v;
    }
};
}
#define random_device rtx_shim_random_device
#define mt19937 rtx_shim_engine
