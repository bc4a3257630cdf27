"""Benchmark of the shading path: Mpixels/s for deferred PBR shading + bloom + ACES at 4K with 64
shadowed point lights (the headline configuration: 3840x2160, 64 point lights with 1024^2 cube shadow
maps + a 4096^2 directional PCF shadow map), plus the IBL precompute time.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one frame through composition -> bloom 1 -> (halo exchange) -> bloom 2 + post-process on
inputs already resident in HBM.  With N > 1 every rank owns a 3840x2160 strip of a 3840 x (2160 N)
frame (weak scaling); strips swap 4 bloom rows with their neighbours over RCCL and the RGBA8 strips
are gathered on rank 0 inside the timed region.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT, LIGHTS, CUBE_RES, DIR_RES = 3840, 2160, 64, 1024, 4096
SEED = 0x5EED0003
PEAK_FP32_TFLOPS = 157.3   # MI355X FP32 vector peak
PEAK_HBM_GBS = 8000.0      # MI355X HBM3E peak
OPS_PER_PIXEL = 504        # canonical FP32 ops per pixel outside the light loop
OPS_PER_PIXEL_LIGHT = 810  # canonical FP32 ops per (pixel, light) pair


def build_inputs(rank: int, device, rows: int, n_lights: int, cube_res: int, dir_res: int):
    """Replicated scene data from SEED, this rank's G-buffer strip from SEED + 1000 + rank; IBL images
    computed on the GPU by the library from a synthetic environment (reference sizes 512/32/128/512)."""
    import torch
    from vulkanrenderer_amd import _lib, passes as P, synth

    rng = np.random.default_rng(SEED)
    lights = synth.make_lights(rng, n_lights)
    dir_shadow = synth.make_dir_shadow(rng, dir_res)
    cube_shadow = synth.make_cube_shadows(rng, n_lights, cube_res)
    env = P.to_device_f16(synth.make_env_cube(rng, 512, sun_peak=20.0), device)
    irr = torch.empty((6, 32, 32, 4), dtype=torch.float16, device=device)
    pre = torch.empty((6, 128, 128, 4), dtype=torch.float16, device=device)
    lut = torch.empty((512, 512, 2), dtype=torch.float16, device=device)
    P.irradiance(env, irr, _lib.IRRADIANCE_SEQUENTIAL)
    P.prefilter(env, pre, 0.0)
    P.brdf_lut(lut)
    torch.cuda.synchronize()
    albedo, normal, arm, position = synth.make_gbuffer(np.random.default_rng(SEED + 1000 + rank), WIDTH, rows)
    f = synth.FrameInputs(WIDTH, rows, albedo, normal, arm, position, synth.default_scene_info(), lights,
                          dir_shadow, cube_shadow, P.f16_bits(irr), P.f16_bits(pre), P.f16_bits(lut))
    # the point-shadow cubes are resident in the library's tiled layout (DeviceFrame.upload re-lays them once)
    return f, P.DeviceFrame.upload(f, device)


def time_shadow_retile(f, device, reps: int = 5) -> dict:
    """Row-major -> 8x4-tile re-layout of all point-shadow cubes: what a caller pays per frame if its
    shadow pass hands over row-major depth maps (a native shadow pass would write tiles directly)."""
    import torch
    from vulkanrenderer_amd import _lib, passes as P
    lin = torch.from_numpy(f.cube_shadow).to(device)
    P.tile_depth_cubes(lin, _lib.CUBE_LAYOUT_TILED_8X4_X25)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        P.tile_depth_cubes(lin, _lib.CUBE_LAYOUT_TILED_8X4_X25)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return {"ms": round(ms, 4), "GB/s": round(2 * lin.numel() * 4 / (ms * 1e-3) / 1e9, 1),
            "bytes": 2 * lin.numel() * 4}


def time_ibl(device) -> dict:
    """IBL precompute at the BASELINE configuration: 4096x2048 HDR equirect -> 2048^2 cube ->
    irradiance 64^2, 5-mip GGX prefilter (128^2 .. 8^2, 1024 samples) and 512^2 BRDF LUT."""
    import torch
    from vulkanrenderer_amd import _lib, passes as P, synth

    rng = np.random.default_rng(SEED + 4)
    src = P.to_device_f16(synth.make_equirect(rng, 4096, 2048, hdr=True), device)
    env = torch.empty((6, 2048, 2048, 4), dtype=torch.float16, device=device)
    irr = torch.empty((6, 64, 64, 4), dtype=torch.float16, device=device)
    pres = [torch.empty((6, 128 >> m, 128 >> m, 4), dtype=torch.float16, device=device) for m in range(5)]
    lut = torch.empty((512, 512, 2), dtype=torch.float16, device=device)

    def run():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        P.equirect_to_cube(src, _lib.SRC_RGBA16F, env)
        tiled = P.tile_env_cube(env)  # the copy the two sampling passes read; counted with the equirect stage
        ev[1].record()
        P.irradiance(tiled, irr, _lib.IRRADIANCE_SEQUENTIAL, env_layout=_lib.ENV_LAYOUT_TILED_4X4)
        ev[2].record()
        for m in range(5):
            P.prefilter(tiled, pres[m], m / 4.0, env_layout=_lib.ENV_LAYOUT_TILED_4X4)
        ev[3].record()
        P.brdf_lut(lut)
        ev[4].record()
        torch.cuda.synchronize()
        return [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]

    run()
    t = run()
    return {"equirect_ms": round(t[0], 3), "irradiance_ms": round(t[1], 3), "prefilter_ms": round(t[2], 3),
            "brdf_lut_ms": round(t[3], 3), "total_ms": round(sum(t), 3),
            "config": "equirect 4096x2048 RGBA16F -> env 2048^2 (+ 4x4-tiled copy), irradiance 64^2 in the shader's summation order, "
                      "prefilter 128^2 x5 mips, LUT 512^2"}


def time_coverage(f, dev_frame, renderer, device) -> dict:
    """Light-volume coverage (VKR_COVERAGE_MESH) at the benchmark size: the mask kernel, and the shading kernel
    consuming the masks.  With the reference's light radii every volume contains the camera, so every mask
    bit is set and the image is the full-coverage one; a second scene shrinks the volumes to show what the
    per-light masks save when lights are local."""
    import dataclasses
    import torch
    from vulkanrenderer_amd import passes as P, synth

    verts, tris = synth.make_sphere_mesh()
    dverts, dtris = torch.from_numpy(verts).to(device), torch.from_numpy(tris.astype(np.int32)).to(device)

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {}
    want = renderer.color.clone()
    masked = dataclasses.replace(dev_frame, coverage=P.light_coverage(dev_frame, dverts, dtris))
    color, thr = torch.empty_like(renderer.color), torch.empty_like(renderer.threshold)
    out["reference_radii"] = {"coverage_ms": round(timed(lambda: P.light_coverage(dev_frame, dverts, dtris)), 4),
                              "shade_ms": round(timed(lambda: P.shade(masked, color, thr)), 4),
                              "covered_fraction": 1.0 if bool((masked.coverage == -1).all()) else None,
                              "equals_full_coverage_image": bool(torch.equal(color, want))}
    lights = f.lights.copy()
    for i in range(len(lights)):
        lights["color"][i] *= 0.02
        r = synth.light_volume_radius(lights["color"][i])
        m = np.diag([r, r, r, 1.0]); m[3, :3] = lights["position"][i]
        lights["model"][i] = m.astype(np.float32).ravel()
    small = dataclasses.replace(dev_frame, lights=torch.from_numpy(np.frombuffer(lights.tobytes(), dtype=np.uint8).copy()).to(device))
    small.coverage = P.light_coverage(small, dverts, dtris)
    bits = small.coverage.view(torch.uint8)
    frac = float(sum(int(((bits >> k) & 1).sum()) for k in range(8))) / (len(lights) * f.width * f.height)
    out["local_lights"] = {"coverage_ms": round(timed(lambda: P.light_coverage(small, dverts, dtris)), 4),
                           "shade_ms": round(timed(lambda: P.shade(small, color, thr)), 4),
                           "covered_fraction": round(frac, 4),
                           "note": "same lights with colours x0.02: volume radius 16*sqrt(max/5) shrinks to ~1-2.5 units"}
    out["mesh"] = f"UV sphere, {len(verts)} vertices, {len(tris)} triangles (the reference volume's topology)"
    return out


def time_shadow_pass(f, dev_frame, device) -> dict:
    """The passes that make the shadow maps and the G-buffer (VulkanRenderer.cs:325-428), on the synthetic scene's
    mesh: a closed slab under the height field, 129^2 grid.  Not part of a step (the benchmark frame keeps its
    synthetic maps resident); what a frame with moving lights would add."""
    import torch
    from vulkanrenderer_amd import _lib, passes as P, synth
    verts, tris = synth.make_scene_mesh(128)
    draws = [(torch.from_numpy(verts).to(device), torch.from_numpy(tris.astype(np.int32)).to(device), np.eye(4))]
    res, dres, nl = f.cube_shadow.shape[-1], f.dir_shadow.shape[-1], len(f.lights)

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    uv, nrm, tan = synth.mesh_attributes(verts, tris, uv_scale=0.25)
    alb, nmap, arm = synth.make_material(np.random.default_rng(SEED + 9), 1024)
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(device)
    gdraws = [dict(positions=draws[0][0], triangles=draws[0][1], texcoords=dev(uv, np.float32), normals=dev(nrm, np.float32),
                   tangents=dev(tan, np.float32), albedo=dev(alb, np.uint8), normal=dev(nmap, np.uint8), arm=dev(arm, np.uint8), model=np.eye(4))]
    scene = f.scene.copy()
    scene["camera_projection"] = synth.perspective(60.0, f.width / f.height, 0.1, 100.0).astype(np.float32).ravel()
    ga = P.gbuffer(gdraws, scene, f.width, f.height, device)[0]
    env = P.tile_env_cube(P.to_device_f16(synth.make_env_cube(np.random.default_rng(SEED + 10), 2048), device))
    return {"point_shadow_ms": round(timed(lambda: P.point_shadow_maps(draws, dev_frame.lights, nl, res)), 4),
            "dir_shadow_ms": round(timed(lambda: P.dir_shadow_map(draws, f.scene["light_space_matrix"], dres, device)), 4),
            "geometry_pass_ms": round(timed(lambda: P.gbuffer(gdraws, scene, f.width, f.height, device)), 4),
            "geometry_pass_covered_fraction": round(float((ga[..., 3] != 0).float().mean()), 4),
            "skybox_ms": round(timed(lambda: P.skybox(env, scene, f.width, f.height, env_layout=_lib.ENV_LAYOUT_TILED_4X4)), 4),
            "config": f"{nl} lights x 6 x {res}^2 D32F (written as 8x4 tiles) + {dres}^2 directional map; mesh {len(verts)} vertices, {len(tris)} triangles"}


def measured_traffic(lights: int, cube_res: int, dir_res: int) -> dict:
    """HBM-side bytes per launch from the committed rocprofv3 counter passes (profiles/r1_traffic.json);
    counters cannot be read from inside this process, so the figures apply only to the workload they
    were collected on and are dropped (null) for any other."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except OSError:
        return {}
    w = rec.get("workload", {})
    if (w.get("width"), w.get("rows"), w.get("lights"), w.get("cube_res"), w.get("dir_res")) != (WIDTH, HEIGHT, lights, cube_res, dir_res):
        return {}
    return {k: v["traffic_bytes"] for k, v in rec.get("kernels", {}).items()}


def cpu_baseline(f, gpu_color, gpu_thr, sample_rows: int) -> dict:
    """The oracle (CPU restatement, OpenMP) on the first `sample_rows` rows of the same frame; also
    checks the GPU's rows against it bit for bit."""
    from tests import oracle
    sub = f.rows(0, sample_rows)
    t0 = time.perf_counter()
    color, thr = oracle.shade(sub)
    b1 = oracle.bloom_blur(thr, True, oracle.WRAP_REPEAT)
    b2 = oracle.bloom_blur(b1, False, oracle.WRAP_CLAMP)
    oracle.postprocess(color, b2)
    dt = time.perf_counter() - t0
    exact = bool(np.array_equal(color, gpu_color[:sample_rows]) and np.array_equal(thr, gpu_thr[:sample_rows]))
    return {"value": round(sample_rows * f.width / dt / 1e6, 4), "unit": "Mpixels/s", "cores": oracle.omp_threads(),
            "kind": "port", "sample": f"rows 0..{sample_rows} of the {f.width}x{f.height} frame, {len(f.lights)} lights "
                                      f"(composition + bloom x2 + postprocess), {dt:.2f} s wall",
            "gpu_rows_bit_exact": exact}


def main() -> None:
    global WIDTH, HEIGHT
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=WIDTH, help="strip width per GPU (default: the headline 3840)")
    ap.add_argument("--height", type=int, default=HEIGHT, help="strip rows per GPU (default: the headline 2160)")
    ap.add_argument("--lights", type=int, default=LIGHTS)
    ap.add_argument("--cube-res", type=int, default=CUBE_RES)
    ap.add_argument("--dir-res", type=int, default=DIR_RES)
    ap.add_argument("--cpu-rows", type=int, default=None, help="rows of the frame the CPU baseline shades (default: all; 0 = skip)")
    ap.add_argument("--no-ibl", action="store_true", help="skip the IBL precompute timing")
    ap.add_argument("--no-coverage", action="store_true", help="skip the light-volume coverage and shadow-pass timings")
    args = ap.parse_args()
    WIDTH, HEIGHT = args.width, args.height
    args.cpu_rows = HEIGHT if args.cpu_rows is None else min(args.cpu_rows, HEIGHT)

    import torch
    import torch.distributed as dist
    from vulkanrenderer_amd import _lib
    from vulkanrenderer_amd.strips import StripPlan, StripRenderer

    _lib.load()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # One rank per GPU over RCCL.  VKR_DIST_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs
    # than ranks (ranks then share devices; RCCL refuses two ranks on one device).
    backend = os.environ.get("VKR_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    f, dev_frame = build_inputs(rank, device, HEIGHT, args.lights, args.cube_res, args.dir_res)
    plan = StripPlan.make(rank, world, HEIGHT * world)
    # two frames in flight, like the reference (VulkanRenderer.cs:60): the gather of frame k overlaps frame k+1
    renderer = StripRenderer(plan, WIDTH, device=device, frames_in_flight=2 if world > 1 else 1)

    def step():
        renderer.render(dev_frame)
        if world > 1:
            renderer.gather(0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations with HIP events on the launch stream (separate loop, same inputs)
    names = ["shade", "bloom_h", "bloomv_post"]
    acc = {n: 0.0 for n in names}
    be = renderer.backend
    lo, hi = (4, 4 + plan.rows - 1)
    for _ in range(args.steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        be.shade(dev_frame, renderer.color, renderer.threshold)
        ev[1].record()
        be.bloom_h(renderer.threshold, renderer.bloom1, plan.rows)
        ev[2].record()
        be.bloomv_post(renderer.bloom1, renderer.color, renderer.ldr, lo, hi)
        ev[3].record()
        torch.cuda.synchronize()
        for i, n in enumerate(names):
            acc[n] += ev[i].elapsed_time(ev[i + 1])
    kernel_ms = {n: acc[n] / args.steps for n in names}
    barrier()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    px = WIDTH * HEIGHT
    ms_per_step = elapsed / args.steps * 1e3
    # background texels (cleared G-buffer: ~10% of the synthetic frame) are discarded by the composition
    # shader and take nothing from the lights, so they are not counted as shading work
    shaded_px = int((f.normal.view(np.float16)[..., :3] != 0).any(axis=-1).sum())
    flops = (OPS_PER_PIXEL + OPS_PER_PIXEL_LIGHT * args.lights) * shaded_px
    shade_tflops = flops / (kernel_ms["shade"] * 1e-3) / 1e12
    traffic = measured_traffic(args.lights, args.cube_res, args.dir_res)
    out = {
        "metric": "Mpixels/s deferred PBR+bloom+ACES @4K 64 shadowed lights",
        "value": round(px * world / (elapsed / args.steps) / 1e6, 3),
        "unit": "Mpixels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{WIDTH}x{HEIGHT} strip per GPU ({WIDTH}x{HEIGHT * world} frame), {args.lights} point lights with "
                               f"{args.cube_res}^2 cube shadow maps + {args.dir_res}^2 directional PCF shadow; composition + bloom x2 + ACES/gamma",
                   "parallelism": f"strips{world}",
                   "storage": "RGBA16F G-buffer and targets, RGBA8 output, D32F shadow cubes in 8x4-texel tiles holding 25 x depth (the product light.frag forms per tap)"},
        "roofline": {"bound": "valu", "kernel": "shade_kernel", "achieved": round(shade_tflops, 3), "peak": PEAK_FP32_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(shade_tflops / PEAK_FP32_TFLOPS, 4), "traffic": traffic.get("shade_kernel"),
                     "algorithmic": f"({OPS_PER_PIXEL} + {OPS_PER_PIXEL_LIGHT} x {args.lights}) FP32 ops x {shaded_px} shaded pixels per launch "
                                    f"({px - shaded_px} background pixels of the {px} do no shading work)",
                     "avg_launch_ms": round(kernel_ms["shade"], 4)},
        "roofline_post": [
            {"bound": "hbm", "kernel": "bloom_h_kernel", "achieved": round(16 * px / (kernel_ms["bloom_h"] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
             "unit": "GB/s", "frac": round(16 * px / (kernel_ms["bloom_h"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "avg_launch_ms": round(kernel_ms["bloom_h"], 4),
             "algorithmic": "16 B/pixel", "traffic": traffic.get("bloom_h_kernel")},
            {"bound": "hbm", "kernel": "bloom_v_kernel<fused post>", "achieved": round(20 * px / (kernel_ms["bloomv_post"] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
             "unit": "GB/s", "frac": round(20 * px / (kernel_ms["bloomv_post"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "avg_launch_ms": round(kernel_ms["bloomv_post"], 4),
             "algorithmic": "20 B/pixel (8 bloom-1 + 8 colour in, 4 RGBA8 out)", "traffic": traffic.get("bloom_v_kernel")}],
    }
    if world == 1 and dev_frame.cube_layout != 0:
        rt = time_shadow_retile(f, device)
        rt["note"] = ("not part of a step: the shadow cubes are inputs resident in the tiled layout; "
                      "ms_per_step + this = a frame whose shadow pass hands over row-major maps")
        rt["value_with_retile_every_frame"] = round(px / ((ms_per_step + rt["ms"]) * 1e-3) / 1e6, 3)
        out["shadow_retile"] = rt
    if world == 1:
        import ctypes
        fp32, copy = ctypes.c_float(), ctypes.c_float()
        _lib.check(_lib.load().vkr_measure_peaks(ctypes.byref(fp32), ctypes.byref(copy)))
        out["measured_peaks"] = {"fp32_fma_loop_tflops": round(fp32.value, 1), "copy_1GiB_GBps": round(copy.value, 1),
                                 "shade_frac_of_measured": round(shade_tflops / fp32.value, 4),
                                 "bloom_h_frac_of_measured": round(16 * px / (kernel_ms["bloom_h"] * 1e-3) / 1e9 / copy.value, 4),
                                 "bloom_v_frac_of_measured": round(20 * px / (kernel_ms["bloomv_post"] * 1e-3) / 1e9 / copy.value, 4)}
    if world == 1 and not args.no_coverage:
        out["light_volume_coverage"] = time_coverage(f, dev_frame, renderer, device)
        out["shadow_pass"] = time_shadow_pass(f, dev_frame, device)
        sp, cv = out["shadow_pass"], out["light_volume_coverage"]["reference_radii"]
        whole = sp["point_shadow_ms"] + sp["dir_shadow_ms"] + sp["geometry_pass_ms"] + sp["skybox_ms"] + cv["coverage_ms"] + ms_per_step
        out["frame_from_meshes"] = {"ms": round(whole, 3), "Mpixels/s": round(px / (whole * 1e-3) / 1e6, 1),
                                    "note": "shadow passes + geometry pass + skybox + light-volume coverage + the timed step: a frame in which "
                                            "every light moves and nothing but meshes, materials, lights and the camera is given"}
    if not args.no_ibl:
        out["ibl_precompute"] = time_ibl(device)
    if world == 1 and args.cpu_rows > 0:
        from vulkanrenderer_amd import passes as P
        out["cpu_baseline"] = cpu_baseline(f, P.f16_bits(renderer.color), P.f16_bits(renderer.threshold), args.cpu_rows)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
