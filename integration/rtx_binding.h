// Shared by the two drop-in translation units (integration/bvh_interface.cpp,
// integration/render.cpp).  Compiled against the REFERENCE's own, unmodified headers
// (src/bvh_interface.h, src/render.h, src/scene.h, src/common.h, framework/trackball.h):
// the class declarations the callers see do not change.
#pragma once
#include "bvh_interface.h" // the reference's (src/bvh_interface.h:12-41)
#include "common.h"
#include "scene.h"
#include <framework/trackball.h>

#include <bvh_build.h> // raytracer_amd/host: rt::flattenScene, rt::FlatScene
#include <rtx.h>       // include/rtx.h: the C ABI of librtx_hip.so

// What BvhInterface::m_impl points at (opaque to every caller, src/bvh_interface.h:9):
// the flattened tree as uploaded, its device handle and the caller's Scene.
class BoundingVolumeHierarchy {
public:
    rt::FlatScene flat;
    rtx_scene* dev = nullptr;
    Scene* scene = nullptr;
};

// The reference's headers are left untouched, and a few values the binding must read
// have no accessor there (Trackball's field of view and half extents, EnvironmentMap's
// image, BvhInterface::m_impl).  An explicit template instantiation may name a private
// member ([temp.spec]); it defines the friend function that hands the member pointer out.
template <class Tag, typename Tag::type Member>
struct PrivateMember {
    friend typename Tag::type memberOf(Tag) { return Member; }
};

rtx_features toRtx(const Features& features);
rtx_camera toRtx(const Trackball& camera);
rtx_scene* deviceSceneOf(const BvhInterface& bvh);
// Binds the calling thread to the GPU on its first call through the binding.  The reference's
// command line renders every camera on a thread of its own (src/main.cpp:560-563) and its
// viewer intersects from OpenMP threads: the thread that constructs the BvhInterface works in
// frame slot 0, each further thread in the next slot in turn, so that the cameras' frames share
// the GPU instead of queueing behind one another.  Throws std::runtime_error without a GPU.
void bindThisThread();
