#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + secondary) on the dragon scene,
1920x1080, 16 spp jittered, Phong shading + hard shadows (BASELINE.json
configs[1]), on 1..8 MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step is one frame.  The scene, its BVH and the framebuffer live in HBM before
the timed region starts; nothing crosses PCIe inside it.  With N > 1 the frame
is cut into interleaved 32x32 tiles (tile t -> rank t % N), every rank renders
its tiles and the framebuffer is assembled on rank 0 by one RCCL gather per
frame; total work is fixed, so scaling is "strong".  Frames are double-buffered:
the gather of frame i runs on the collective's stream while frame i + 1 is
rendered, and every frame is assembled on rank 0 inside the timed region.

data/dragon.obj is not part of the reference checkout, so the geometry is the
seeded stand-in of raytracer_amd.scene.dragon_standin (81 920 triangles, unit
ball, same light and camera as SceneType::Dragon); pass --obj dragon.obj to use the
real mesh (loaded and normalised as the reference does), or --obj-npz for a scene
saved by raytracer_amd.scene.SceneData.save.

One JSON line is printed by rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def algorithmic_bytes(stats_rays, shaded, pixels, n_aabb, n_tri):
    """Implementation-neutral bytes one frame must move: per ray 28 B in, 24 B per
    box test and 36 B per triangle test the reference's visit rule performs, 16 B
    out; per shaded hit three 32-B vertices and a 32-B material; 12 B per pixel."""
    return stats_rays * (28.0 + 24.0 * n_aabb + 36.0 * n_tri + 16.0) + shaded * (3 * 32 + 32) + pixels * 12.0


def cpu_baseline(scene, features, cam, args, wl):
    """The CPU path timed on this box's host cores on a bounded sample of the same
    workload: the reference's own renderRayTracingMRaysPerPixel when oracle/_ref
    was built and shipped, else the CPU restatement.  Also returns the per-ray
    work counts of the reference's visit rule (from the instrumented restatement)."""
    from raytracer_amd import abi
    from raytracer_amd.scene import make_params
    from oracle.bindings import OracleEngine, RefEngine, have_ref
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    orc = OracleEngine(scene, features)
    # per-ray work of the reference's visit rule on a 1/16-area frame of the same view
    shrink = 4 if args.workload == "headline" else 16 # (the other workloads cost more per pixel)
    small = make_params(args.width // shrink, args.height // shrink, **wl["params"])
    mode = dict(mode=wl["ref_mode"])                  # 1: the harness's pixel loop around the reference's getFinalColor (depth, keyed draws)
    orc.render(cam, features, small, threads=threads)
    c = orc.counters.astype(np.float64)
    rays_small = c[0] + c[1] + c[2]
    n_aabb, n_tri = c[3] / rays_small, c[4] / rays_small
    eng = RefEngine(scene, features) if have_ref() else orc
    kind = eng.kind
    # probe, then size the sample for roughly 15 s of CPU work
    t0 = time.perf_counter()
    eng.render(cam, features, small, threads=threads, **mode)
    probe = time.perf_counter() - t0
    scale = min(4.0, max(1.0, (15.0 / max(probe, 1e-3)) ** 0.5))
    w, h = int(small.width * scale) // 8 * 8, int(small.height * scale) // 8 * 8
    sample = make_params(w, h, **wl["params"])
    t0 = time.perf_counter()
    eng.render(cam, features, sample, threads=threads, **mode)
    dt = time.perf_counter() - t0
    orc.render(cam, features, sample, threads=threads)
    rays = float(orc.counters[:3].sum())
    orc.close()
    if eng is not orc:
        eng.close()
    return dict(value=rays / dt / 1e6, unit="Mrays/s", cores=threads, kind=kind,
                sample=f"{w}x{h} frame of the same scene/camera and parameters, {int(rays)} rays in {dt:.2f} s "
                       f"({('reference renderRayTracingMRaysPerPixel, OpenMP' if wl['ref_mode'] == 0 else 'reference getFinalColor in a pixel loop, OpenMP') if kind == 'reference' else 'CPU restatement, OpenMP'})"), n_aabb, n_tri


def workload(name, args):
    """The scene, features, frame parameters and description of a workload:
    headline      BASELINE configs[1] (the metric's own configuration; the default)
    area_glossy   BASELINE configs[2]: the same mesh with a glossy material under one parallelogram
                  light, 64 shadow samples, glossy reflections (fans of 5) to depth 4, 1 spp
    winding       the headline configuration on the harder stand-in mesh (a coiled body with thin wings, legs
                  and spikes: rays cross it several times)
    big_scene     BASELINE configs[4]'s geometry: 122 instances of the mesh baked to 9 994 240
                  triangles, 3840x2160, 16 spp jittered"""
    from raytracer_amd import abi
    from raytracer_amd.scene import SceneData, dragon_scene, instanced_scene, winding_dragon_scene
    base = dict(shading=1, hard_shadow=1, normal_interp=1, accel_structure=1, bvh_sah_binning=1)
    def dragon():
        """SceneType::Dragon (src/scene.cpp:68-73): the mesh, one white point light at (-1, 1, -1).  A user's
        OBJ goes through the host loader with the reference's unit normalisation (framework/src/mesh.cpp:151-176)."""
        if args.obj:
            from raytracer_amd import engine
            s = engine.load_obj(args.obj, normalize=True)
            s.lights.append(abi.point_light((-1, 1, -1), (1, 1, 1)))
            return s, f"{os.path.basename(args.obj)} ({s.n_triangles} tris, user-supplied, scaled to the unit cube)"
        if args.obj_npz:
            s = SceneData.load(args.obj_npz)
            return s, f"{os.path.basename(args.obj_npz)} ({s.n_triangles} tris, user-supplied scene file)"
        s = dragon_scene()
        return s, f"dragon stand-in ({s.n_triangles} tris, seeded; dragon.obj is absent from the reference checkout)"
    if name == "headline":
        scene, what = dragon()
        return dict(scene=scene, features=abi.features(**base), params=dict(driver=abi.DRIVER_MRAYS, samples=args.spp), ref_mode=0,
                    metric="Mrays/s (primary+secondary), dragon 1080p 16spp",
                    text=f"{what} {args.width}x{args.height}, {args.spp} spp jittered, Phong + hard shadows, 1 point light")
    if name == "winding":
        scene = winding_dragon_scene()
        return dict(scene=scene, features=abi.features(**base), params=dict(driver=abi.DRIVER_MRAYS, samples=args.spp), ref_mode=0,
                    metric="Mrays/s (primary+secondary), dragon 1080p 16spp",
                    text=f"winding dragon stand-in ({scene.n_triangles} tris, seeded: coiled body, thin wings, legs, spikes) "
                         f"{args.width}x{args.height}, {args.spp} spp jittered, Phong + hard shadows, 1 point light")
    if name == "area_glossy":
        scene, what = dragon()
        scene.lights = [abi.parallelogram_light((-0.2, 0.9, 0.0), (0.4, 0, 0), (0, 0, 0.4), (1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1))]
        for m in scene.meshes:   # a glossy material: ks != 0, shininess 50
            m.material[3:6] = 0.5
            m.material[6] = 50.0
        f = abi.features(**dict(base, hard_shadow=0, soft_shadow=1, recursive=1, glossy_reflection=1))
        return dict(scene=scene, features=f, params=dict(driver=abi.DRIVER_PLAIN, shadow_samples=64, glossy_rays=5, ray_depth=4), ref_mode=1,
                    metric="Mrays/s (primary+secondary), dragon 1080p area light 64 + glossy depth 4",
                    text=f"{what}, ks 0.5, shininess 50, {args.width}x{args.height}, 1 spp, "
                         "one parallelogram light with 64 shadow samples, glossy reflections (5 rays) to depth 4")
    if name == "big_scene":
        scene = instanced_scene(122)
        return dict(scene=scene, features=abi.features(**base), params=dict(driver=abi.DRIVER_MRAYS, samples=args.spp), ref_mode=0,
                    metric="Mrays/s (primary+secondary), 10 M triangles 3840x2160 16spp",
                    text=f"122 instances of the dragon stand-in baked to {scene.n_triangles} triangles, {args.width}x{args.height}, "
                         f"{args.spp} spp jittered, Phong + hard shadows, 1 point light")
    sys.exit(f"unknown workload {name}")


def measured_counters(workload_name):
    """Per-launch hardware counters of the dominant kernel from the newest rocprofv3 summary committed under
    profiles/ (tools/profile_counters.sh + tools/round_profiles.py): SQ_INSTS_VALU, HBM bytes (2 x FETCH_SIZE +
    WRITE_SIZE, KB), the file they come from, and whether the library they were taken on is the one loaded now."""
    import glob
    import hashlib
    suffix = "" if workload_name == "headline" else "_" + workload_name
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]{suffix}_render_counters.json")))
    if not files:
        return {}
    with open(files[-1]) as fh:
        d = json.load(fh)
    c = d.get("counters_per_launch_mean", {})
    out = {"file": os.path.relpath(files[-1], ROOT), "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU")}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["hbm_bytes_per_launch"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    with open(os.path.join(ROOT, "raytracer_amd", "librtx_hip.so"), "rb") as fh:
        now = hashlib.sha256(fh.read()).hexdigest()
    # (a summary without a hash predates the check: it cannot be told apart from a stale one)
    out["stale"] = d.get("library_sha256") != now
    return out


def valu_peak(waves_per_simd=4):
    """Vector-ALU issue rate of this GPU, measured now: wave-instructions per second chip-wide
    for the traversal loop's instruction mix at the render kernel's occupancy, and per cycle per
    SIMD for that mix and for plain v_mul_f32 (the fastest kind).  None if the diagnostic
    library was not built."""
    so = os.path.join(ROOT, "raytracer_amd", "libvalu_peak.so")
    if not os.path.exists(so):
        return None
    lib = C.CDLL(so)
    lib.valu_peak_run.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    out = (C.c_double * 4)()
    if lib.valu_peak_run(lib.valu_peak_mix_kind(), waves_per_simd, 4000, out) != 0:
        return None
    res = dict(ipc=out[0], clock_mhz=out[2], chip_ginstr_s=out[3] / 1e9)
    if lib.valu_peak_run(0, waves_per_simd, 4000, out) != 0:
        return None
    res["ipc_mul"] = out[0]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--obj", default=None, help="Wavefront OBJ (with its MTL / textures beside it) to use instead of the stand-in, e.g. a user-supplied dragon.obj: "
                                                "loaded and scaled into the unit cube as the reference's SceneType::Dragon does (loadMesh(..., normalize = true))")
    ap.add_argument("--obj-npz", default=None, help="scene .npz (raytracer_amd.scene.SceneData.save) to use instead of the stand-in")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 path)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--check", action="store_true", help="rank 0 compares the assembled frame with a single-GPU render of it")
    ap.add_argument("--no-every-block", action="store_true", help="skip the second timing of the frame with no block declared unable to see the scene (roofline.kernel_ms_every_block_traced)")
    ap.add_argument("--force-pipeline", action="store_true", help="rehearsal: run the N > 1 code path (tiles, gather, assembly) in a one-rank group")
    ap.add_argument("--workload", default="headline", choices=["headline", "winding", "area_glossy", "big_scene"],
                    help="headline = BASELINE configs[1], the metric's configuration; area_glossy = configs[2]; big_scene = configs[4]'s geometry")
    ap.add_argument("--frames-in-flight", type=int, default=2,
                    help="frames kept in flight on each GPU (1..4): frame i runs in the library's frame slot i %% F on that slot's own stream, "
                         "so the next frame's waves fill the compute units the previous frame's last, longest work units leave idle; "
                         "1 = one frame after the other on one stream")
    ap.add_argument("--tree", default="auto", choices=["auto", "host", "gpu"],
                    help="where the BVH is built: host = the reference's tree (default for headline, area_glossy); gpu = rtx_scene_build_device (default for big_scene)")
    args = ap.parse_args()
    if args.workload == "big_scene" and (args.width, args.height) == (1920, 1080):
        args.width, args.height = 3840, 2160

    import torch
    import torch.distributed as dist
    from raytracer_amd import abi, engine
    from raytracer_amd.distributed import FramePipeline
    from raytracer_amd.scene import SceneData, dragon_scene, make_camera, make_params

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs one process per GPU: launch with python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    grouped = world > 1 or args.force_pipeline
    if grouped:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    hip = engine.hip_lib()
    wl = workload(args.workload, args)
    scene, features = wl["scene"], wl["features"]
    cam_in = make_camera()

    t_build = time.perf_counter()
    # the reference's tree, built on the host and uploaded once; the scene of millions of triangles gets
    # its tree on the GPU (a Morton-order tree, same hits), so that the ranks of a node do not each run a
    # threaded host build of seconds on the same cores
    device_build = args.tree == "gpu" or (args.tree == "auto" and args.workload == "big_scene")
    gpu = engine.GpuEngine(scene, features, device=local_rank, device_build=device_build)
    t_build = time.perf_counter() - t_build
    dscene = gpu.device_scene()
    cam = engine.device_camera(cam_in, args.width, args.height)
    rp = make_params(args.width, args.height, tile_size=args.tile, tile_rank=rank, tile_world=world, **wl["params"])
    # queue the library's work on torch streams so events and collectives order with it: one
    # stream per frame slot (rtx_select_slot), each slot with its own working memory in the library
    F = max(1, min(4, args.frames_in_flight))
    streams = engine.concurrent_streams(F, dev)   # (streams that share a hardware queue would take turns)
    F = len(streams)
    for k in reversed(range(F)):    # (slot 0 is left selected)
        assert hip.rtx_select_slot(k) == 0, hip.rtx_last_error()
        assert streams[k].cuda_stream != 0
        assert hip.rtx_set_stream(C.c_void_p(streams[k].cuda_stream)) == 0, hip.rtx_last_error()
    torch.cuda.set_stream(streams[0])
    stream = streams[0]

    import contextlib

    @contextlib.contextmanager
    def in_slot(k):
        """The library's frame slot k and its stream current for the calls inside."""
        assert hip.rtx_select_slot(k) == 0, hip.rtx_last_error()
        try:
            with torch.cuda.stream(streams[k]):
                yield
        finally:
            hip.rtx_select_slot(0)

    pixels = args.width * args.height
    frames = [torch.zeros((args.height, args.width, 3), dtype=torch.float32, device=dev) for _ in range(F)]
    frame = frames[0]
    def check(rc):
        if rc != 0:
            raise RuntimeError(hip.rtx_last_error().decode())

    pipe = None
    last = {"slot": 0, "n": 0}
    if grouped:
        per = hip.rtx_tiles_per_rank(C.byref(rp))
        def enter(k):
            last["slot"] = k
            return in_slot(k)
        pipe = FramePipeline(
            lambda: torch.zeros((per, args.tile, args.tile, 3), dtype=torch.float32, device=dev),
            lambda tiles: check(hip.rtx_render_tiles_device(dscene, C.byref(cam), C.byref(features), C.byref(rp), C.c_void_p(tiles.data_ptr()), None)),
            lambda stack: check(hip.rtx_detile_device(C.c_void_p(stack.data_ptr()), C.byref(rp), C.c_void_p(frames[last["slot"]].data_ptr()))),
            slots=F, enter=enter)

    def step(stats=None, serial=False):
        if pipe is not None:
            if stats is not None:   # the counted frame: this rank's tiles alone, waited for
                check(hip.rtx_render_tiles_device(dscene, C.byref(cam), C.byref(features), C.byref(rp), C.c_void_p(pipe.tiles[0].data_ptr()), C.byref(stats)))
            else:
                pipe.submit()
        elif stats is not None or serial or F == 1:
            check(hip.rtx_render_device(dscene, C.byref(cam), C.byref(features), C.byref(rp), C.c_void_p(frame.data_ptr()), C.byref(stats) if stats is not None else None))
        else:
            k = last["n"] % F
            last["n"] += 1
            with in_slot(k):
                check(hip.rtx_render_device(dscene, C.byref(cam), C.byref(features), C.byref(rp), C.c_void_p(frames[k].data_ptr()), None))

    def join_slots():
        """Stream 0 waits for everything queued in the other slots."""
        for k in range(1, F):
            e = torch.cuda.Event()
            e.record(streams[k])
            streams[0].wait_event(e)

    # one counted frame: rays per frame (deterministic) and this rank's kernel time
    st = abi.Stats()
    step(st)
    torch.cuda.synchronize()
    # (a slot's first frame measures its blocks for the order, and a scene of millions of triangles
    # gets its camera-relative records in a slot with the second frame from the same position)
    for _ in range(max(args.warmup, 3 * F)):
        step()
    if pipe is not None:
        pipe.drain()
    torch.cuda.synchronize()
    # the kernel's own duration, for the roofline: frames one after the other in slot 0, HIP
    # events on the launch stream (frames in flight together share the chip, so their
    # individual durations say nothing about the kernel)
    serial_ms = None
    if pipe is None:
        for _ in range(2):
            step(serial=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.steps):
            step(serial=True)
        e1.record(stream)
        torch.cuda.synchronize()
        serial_ms = e0.elapsed_time(e1) / args.steps
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")   # where the small reductions live
    counts = torch.tensor([st.rays_primary, st.rays_shadow, st.rays_bounce, st.shaded_hits, st.rays_traced], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(counts)
    rays_frame = float(counts[:3].sum())
    shaded_frame = float(counts[3])

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    if pipe is not None:
        pipe.drain()          # the last frame's gather and assembly belong to the timed region
    join_slots()
    ev1.record(stream)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    device_ms = ev0.elapsed_time(ev1) / args.steps     # HIP events around the timed region, per frame (frames in flight overlap)
    kernel_ms = serial_ms if serial_ms is not None else device_ms
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax[0])

    # for the record: the same frame with every block traced (no block is declared
    # unable to see the scene; each of its camera rays then fails the root box test)
    traced_all_ms = None
    if not grouped and not args.no_every_block:
        os.environ["RTX_NO_CULL"] = "1"
        best = 1e9
        for _ in range(4):
            s2 = abi.Stats()
            step(s2)
            best = min(best, s2.kernel_ms)
        del os.environ["RTX_NO_CULL"]
        assert (s2.rays_primary, s2.rays_shadow) == (st.rays_primary, st.rays_shadow)
        assert s2.rays_traced >= st.rays_traced
        traced_all_ms = best

    # for the record: the same frame on a tree built on the GPU (rtx_scene_build_device; the timed
    # frames above use the host-built tree, which is the reference's)
    device_tree = None
    if not grouped and not args.no_every_block and not device_build:
        t_dev = time.perf_counter()
        gpu2 = engine.GpuEngine(scene, features, device=local_rank, device_build=True)
        t_dev = time.perf_counter() - t_dev
        best = 1e9
        for _ in range(6):
            s3 = abi.Stats()
            check(hip.rtx_render_device(gpu2.device_scene(), C.byref(cam), C.byref(features), C.byref(rp), C.c_void_p(frame.data_ptr()), C.byref(s3)))
            best = min(best, s3.kernel_ms)
        assert (s3.rays_primary, s3.rays_shadow, s3.rays_bounce) == (st.rays_primary, st.rays_shadow, st.rays_bounce)
        info = gpu2.tree_info()
        device_tree = {"kernel_ms": round(best, 4), "levels": info["levels"], "leaves": info["leaves"], "upload_ms": round(info["upload_ms"], 2),
                       "build_ms": round(info["build_ms"], 2), "host_records_upload_build_s": round(t_dev, 3)}
        gpu2.close()

    # for the record, two harder views of the same scene (kernel time, one frame after the other):
    # a camera close enough that the mesh fills the screen, and a camera that moves between frames, so
    # that no frame finds a block order learnt from an identical predecessor nor camera-relative
    # records made for its position (every frame remakes them: that kernel is inside the time)
    extra = {}
    if not grouped and not args.no_every_block and args.workload in ("headline", "winding"):
        def timed_views(views, reps):
            cams = [engine.device_camera(make_camera(**v), args.width, args.height) for v in views]
            rays, traced = [], []
            for c2 in cams:       # counted once, untimed
                s4 = abi.Stats()
                check(hip.rtx_render_device(dscene, C.byref(c2), C.byref(features), C.byref(rp), C.c_void_p(frame.data_ptr()), C.byref(s4)))
                rays.append(s4.rays)
                traced.append(s4.rays_traced)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                for c2 in cams:
                    check(hip.rtx_render_device(dscene, C.byref(c2), C.byref(features), C.byref(rp), C.c_void_p(frame.data_ptr()), None))
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / (reps * len(cams))
            return ms, sum(rays) / len(rays), sum(traced) / len(traced)
        ms, r, tr = timed_views([dict(distance=1.2)], max(args.steps, 4))
        extra["screen_filling_view"] = {"camera": "distance 1.2 (default 3)", "ms_per_frame": round(ms, 4), "rays_per_frame": int(r), "rays_traced": int(tr),
                                        "mrays_s": round(r / ms / 1e3, 1), "mrays_s_on_rays_traced": round(tr / ms / 1e3, 1)}
        orbit = [dict(rotation_degrees=(20 + 9 * ((7 * i) % 5 - 2), 20 + 37 * i, 0), distance=3.0 - 0.11 * (i % 7)) for i in range(max(args.steps, 8))]
        ms, r, tr = timed_views(orbit, 1)
        extra["moving_camera"] = {"camera": f"{len(orbit)} different poses, one per frame (orbit in steps of 37 degrees, tilt and distance varied); includes remaking the camera-relative records",
                                  "ms_per_frame": round(ms, 4), "rays_per_frame": int(r), "rays_traced": int(tr),
                                  "mrays_s": round(r / ms / 1e3, 1), "mrays_s_on_rays_traced": round(tr / ms / 1e3, 1)}

    if rank == 0 and args.check and grouped:
        ref = torch.zeros_like(frame)
        one = make_params(args.width, args.height, tile_size=args.tile, **wl["params"])
        rc = hip.rtx_render_device(dscene, C.byref(cam), C.byref(features), C.byref(one), C.c_void_p(ref.data_ptr()), None)
        torch.cuda.synchronize()
        assert rc == 0
        for k in range(F):
            # (bit patterns: the reference's NaN pixels -- a glossy surface hit from behind -- must be NaN in both)
            if not torch.equal(ref.view(torch.int32), frames[k].view(torch.int32)):
                bad = (ref.view(torch.int32) != frames[k].view(torch.int32)).any(dim=2)
                ys, xs = torch.nonzero(bad, as_tuple=True)
                raise AssertionError(f"assembled multi-rank frame of slot {k} differs from the single-GPU frame in {int(bad.sum())} pixels, first at "
                                     f"{list(zip(xs[:4].tolist(), ys[:4].tolist()))}, largest difference {float((ref - frames[k]).abs().nan_to_num().max()):.3e}")
        print(f"check ok: the assembled frame of each of {F} slot(s) equals the single-GPU frame", file=sys.stderr)
    if rank == 0:
        value = rays_frame * args.steps / elapsed / 1e6
        out = {
            "metric": wl["metric"],
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["text"] + (f", Morton-order BVH built on the GPU ({gpu.levels()} levels, {gpu.leaves()} leaves)" if device_build
                                                    else f", sweep-split BVH ({gpu.levels()} levels, {gpu.leaves()} leaves)"),
                       "rays_per_frame": int(rays_frame), "primary": int(counts[0]), "shadow": int(counts[1]), "bounce": int(counts[2]),
                       # one "ray" = one BvhInterface::intersect call of the reference.  The device does not make
                       # the camera rays of 8x8 blocks that cannot see the scene's bounds (the reference tests
                       # and misses them): rays_traced are the rays that did walk the scene on the device
                       "rays_traced": int(counts[4]), "value_on_rays_traced": round(float(counts[4]) * args.steps / elapsed / 1e6, 2),
                       "partition": "whole frame" if not grouped else f"interleaved {args.tile}x{args.tile} tiles + RCCL gather to rank 0",
                       # frame i is queued in frame slot i % F (own stream and working memory): the frames of
                       # different slots share the chip, each frame is rendered (and assembled) in full
                       "frames_in_flight": F, "device_ms_per_frame": round(device_ms, 4),
                       "bvh_build_upload_s": round(t_build, 3)},
        }
        if device_tree is not None:
            out["config"]["tree_built_on_gpu"] = device_tree
        if not grouped:
            cb, n_aabb, n_tri = (None, None, None)
            if not args.no_cpu_baseline:
                cb, n_aabb, n_tri = cpu_baseline(scene, features, cam_in, args, wl)
            # The hardware counters of this command come from rocprofv3 passes (they cannot be read from inside
            # the run): the newest summary under profiles/ carries the hash of the library it was taken on, and
            # the line says so when that is not the library running now.
            # (they describe the workload's own command: another mesh or frame size has none)
            own_command = not args.obj and not args.obj_npz and args.spp == 16 and args.tile == 32 and \
                (args.width, args.height) == ((3840, 2160) if args.workload == "big_scene" else (1920, 1080))
            counters = measured_counters(args.workload) if own_command else {}
            traffic = counters.get("hbm_bytes_per_launch")
            valu_instr = counters.get("SQ_INSTS_VALU")
            kernel_waves = 4    # (the frame kernel's launch bound)
            vp = valu_peak(kernel_waves)
            if valu_instr is not None and vp is not None:
                # The kernel is bound by vector-ALU issue (pointer chasing through cache-resident records: no
                # dense contraction, 4 % of the HBM rate).  Peak: a SIMD issues one wave64 instruction every
                # two cycles at best (a wave64 instruction occupies the 32-lane ALU for two passes).
                simds = torch.cuda.get_device_properties(dev).multi_processor_count * 4
                peak = 0.5 * simds * vp["clock_mhz"] * 1e6 / 1e9
                achieved = valu_instr / (kernel_ms * 1e-3) / 1e9
                out["roofline"] = {"bound": "valu issue", "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "G wave-instr/s",
                                   "frac": round(achieved / peak, 4), "traffic": traffic,
                                   "kernel": "renderKernel" if args.workload != "area_glossy" else "renderKernel (walk) + lightKernel + resolveKernel",
                                   # one frame after the other on one stream (HIP events over args.steps frames), the figure
                                   # the kernel trace's average duration agrees with; the timed region above keeps
                                   # frames_in_flight frames on the chip at once
                                   "kernel_ms": round(kernel_ms, 4), "valu_instr_per_launch": int(valu_instr),
                                   "peak_is": f"0.5 wave64 instructions per cycle per SIMD x {simds} SIMDs x the clock measured now ({round(vp['clock_mhz'])} MHz)",
                                   # the same rate against what a SIMD issues of the kernel's OWN instruction mix (three fifths of
                                   # its instructions are 4.3-cycle kinds: packed products, compares, selects, min / max), measured
                                   # now on this GPU at the kernel's waves per SIMD (tools/valu_peak.hip)
                                   "frac_of_mix_ceiling": round(achieved / vp["chip_ginstr_s"], 4), "waves_per_simd": kernel_waves,
                                   "mix_ceiling_instr_per_cycle_per_simd": round(vp["ipc"], 4), "v_mul_f32_instr_per_cycle_per_simd": round(vp["ipc_mul"], 4),
                                   "counters_from": counters.get("file"), "counters_stale": counters.get("stale")}
                if traced_all_ms is not None:
                    out["roofline"]["kernel_ms_every_block_traced"] = round(traced_all_ms, 4)
                    out["roofline"]["mrays_s_every_block_traced"] = round(rays_frame / traced_all_ms / 1e3, 1)
            if n_aabb is not None:
                # The byte count of an implementation that fetched every box and triangle of the reference's visit
                # rule from memory, per ray; this kernel does not move those bytes (one scalar-cache fetch serves 64
                # lanes, the scene is cache-resident, blocks that cannot see the scene are never traced), so the
                # nominal fraction can exceed 1 and only frac_measured (HBM counters / time / peak) is a roofline.
                alg = algorithmic_bytes(rays_frame, shaded_frame, pixels, n_aabb, n_tri)
                nominal = alg / (kernel_ms * 1e-3) / 1e9
                out["roofline_hbm_nominal"] = {"bound": "hbm", "achieved": round(nominal, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                               "frac": round(nominal / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": int(alg),
                                               "box_tests_per_ray": round(n_aabb, 3), "tri_tests_per_ray": round(n_tri, 3), "traffic": traffic}
                if traffic is not None:
                    out["roofline_hbm_nominal"]["frac_measured"] = round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                if "roofline" not in out:   # (no counter summary for this workload under profiles/: the nominal figure alone)
                    out["roofline"] = dict(out["roofline_hbm_nominal"], kernel_ms=round(kernel_ms, 4))
            if extra:
                out["config"].update(extra)
            if cb is not None:
                out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    gpu.close()
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
