"""Benchmark of the shading path: Mpixels/s for deferred PBR shading + bloom + ACES at 4K with 64
shadowed point lights (the headline configuration: 3840x2160, 64 point lights with 1024^2 cube shadow
maps + a 4096^2 directional PCF shadow map), plus the IBL precompute time.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python bench.py --gpus 8                      # starts its own 8 ranks (one process per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one frame through composition -> (halo exchange) -> bloom 1 + bloom 2 + post-process on
inputs already resident in HBM.  With N > 1 the SAME 3840x2160 frame is cut into N horizontal strips
(strong scaling): every rank shades its strip, swaps 4 thresholded rows with its neighbours over
RCCL, and the RGBA8 strips are gathered on rank 0 inside the timed region; rank 0 then renders the
whole frame alone and checks the gathered image against it byte for byte.  Two further N > 1 shapes are
reported as extra keys: a full 3840x2160 strip per rank (weak scaling) and a 16384x2048 strip with 256
lights per rank (the 16384^2 / 8-GPU configuration).  Rank 0 prints one JSON line.

Started without WORLD_SIZE in the environment and with --gpus N > 1, the script starts N child
processes of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set) before anything
touches a GPU, and exits with their status.  VKR_DIST_BACKEND=gloo is a rehearsal switch for machines
with fewer GPUs than ranks (ranks then share devices; RCCL refuses two ranks on one device);
--transport-only rehearses the launcher, the halo exchange and the gather on a machine without any GPU.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT, LIGHTS, CUBE_RES, DIR_RES = 3840, 2160, 64, 1024, 4096
C5_WIDTH, C5_ROWS, C5_LIGHTS = 16384, 2048, 256  # per-GPU share of the 16384^2 / 256-light / 8-GPU configuration
SEED = 0x5EED0003
PEAK_FP32_TFLOPS = 157.3   # MI355X FP32 vector peak
PEAK_HBM_GBS = 8000.0      # MI355X HBM3E peak
L2_RANDOM_LINE_GBS = 34450.0  # random 128-byte lines out of an L2-resident table, all CUs (profiles/r2_gather_bench.txt)
OPS_PER_PIXEL = 504        # canonical FP32 ops per pixel outside the light loop
OPS_PER_PIXEL_LIGHT = 810  # canonical FP32 ops per (pixel, light) pair
# canonical FP32 ops per IBL sample / texel (the same counting convention as the two figures above)
OPS_IRRADIANCE_SAMPLE, OPS_PREFILTER_SAMPLE, OPS_BRDF_SAMPLE, OPS_EQUIRECT_TEXEL = 72, 110, 60, 80
IRRADIANCE_SAMPLES, GGX_SAMPLES = 252 * 63, 1024


# ---------------------------------------------------------------------------------------------
# launcher: one child process per rank, started before any GPU call
# ---------------------------------------------------------------------------------------------

def launch_ranks(n: int, timeout_s: float) -> int:
    """Starts `n` copies of this command as ranks 0..n-1 and waits for them.  Rank 0 inherits stdout
    (the JSON line); all inherit stderr.  If a rank fails or the time limit passes, the others are
    terminated (by their process ids) and the status is non-zero."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + timeout_s
    status = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is not None:
                live.remove(p)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 1
        if status != 0 or time.monotonic() > deadline:
            if not status:
                status = 124
                print(f"bench.py: ranks still running after {timeout_s:.0f} s; stopping them", file=sys.stderr)
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    return status


# ---------------------------------------------------------------------------------------------
# transport rehearsal without a GPU
# ---------------------------------------------------------------------------------------------

class PatternBackend:
    """Stand-in for the HIP passes on CPU tensors, for --transport-only: every pass writes bytes that
    depend on the frame row and, for the last pass, on the 4 rows above and below, so that the gathered
    image is right only if the strip split, both halo directions, the clamp ranges and the gather are.
    No kernel runs and nothing is timed as throughput."""

    def empty(self, shape, dtype):
        import torch
        return torch.zeros(shape, dtype=dtype)

    def shade(self, frame, color, thr, y0=0, y1=None):
        import torch
        y1 = frame.rows if y1 is None else y1
        g = torch.arange(frame.row0 + y0, frame.row0 + y1, dtype=torch.float32)
        thr[y0:y1] = ((g * 7) % 61)[:, None, None].to(torch.float16)
        color[y0:y1] = ((g * 3) % 17)[:, None, None].to(torch.float16)

    def bloom_post(self, thr, hdr, out, lo, hi, row0=0):
        import torch
        from vulkanrenderer_amd.strips import HALO
        rows = hdr.shape[0]
        idx = (torch.arange(rows)[:, None] + HALO + row0 + torch.arange(-4, 5)[None, :]).clamp(lo, hi)
        s = thr.float()[idx].sum(1) + hdr.float()
        out[...] = (s % 256).to(torch.uint8)


def run_transport_only(args, rank: int, world: int) -> None:
    import types
    import torch
    import torch.distributed as dist
    from vulkanrenderer_amd.strips import StripPlan, StripRenderer

    if world > 1:
        dist.init_process_group("gloo")
    w, h = min(args.width, 64), min(args.height, 97)
    plan = StripPlan.make(rank, world, h)
    r = StripRenderer(plan, w, backend=PatternBackend(), frames_in_flight=2 if world > 1 else 1)
    frame = types.SimpleNamespace(rows=plan.rows, row0=plan.row0, width=w)
    parts = None
    for _ in range(args.warmup + args.steps):
        r.render(frame)
        parts = r.gather(0)
    if world > 1:
        dist.barrier()
    if rank == 0:
        whole = StripRenderer(StripPlan.make(0, 1, h), w, backend=PatternBackend())
        want = whole.render(types.SimpleNamespace(rows=h, row0=0, width=w))
        same = bool(torch.equal(torch.cat(parts, 0), want))
        print(json.dumps({"metric": "Mpixels/s deferred PBR+bloom+ACES @4K 64 shadowed lights", "value": None, "unit": "Mpixels/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "transport_only": True,
                          "data": "none: rehearsal of the launcher, halo exchange and gather on CPU tensors; no kernel ran",
                          "distributed": {"backend": "gloo", "world": world}, "frame": f"{w}x{h}",
                          "gathered_equals_single_rank": same}), flush=True)
        if not same:
            raise SystemExit("transport rehearsal: the gathered image differs from the single-rank image")
    if world > 1:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------
# inputs
# ---------------------------------------------------------------------------------------------

def build_scene(device, n_lights: int, cube_res: int, dir_res: int):
    """Replicated scene data from SEED: lights, shadow maps (cubes re-laid into the library's tiled layout on the
    device), and IBL images computed on the GPU by the library from a synthetic environment (reference
    sizes 512 / 32 / 128 / 512).  Returns the host-side record and a DeviceFrame without a G-buffer."""
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
    return dict(lights=lights, dir_shadow=dir_shadow, cube_shadow=cube_shadow, scene=synth.default_scene_info(),
                irradiance=P.f16_bits(irr), prefilter=P.f16_bits(pre), brdf_lut=P.f16_bits(lut))


def frame_of(scene: dict, gbuffer, width: int, rows: int):
    from vulkanrenderer_amd import synth
    albedo, normal, arm, position = gbuffer
    return synth.FrameInputs(width, rows, albedo, normal, arm, position, scene["scene"], scene["lights"], scene["dir_shadow"],
                             scene["cube_shadow"], scene["irradiance"], scene["prefilter"], scene["brdf_lut"])


def strip_device_frame(whole_dev, f_rows, row0: int, rows: int, device):
    """A DeviceFrame for rows [row0, row0 + rows) of a frame whose scene data is already on the device."""
    import dataclasses
    from vulkanrenderer_amd import passes as P
    return dataclasses.replace(whole_dev, rows=rows,
                               albedo=P.to_device_f16(f_rows.albedo, device), normal=P.to_device_f16(f_rows.normal, device),
                               arm=P.to_device_f16(f_rows.arm, device), position=P.to_device_f16(f_rows.position, device))


# ---------------------------------------------------------------------------------------------
# timed loops
# ---------------------------------------------------------------------------------------------

def timed_frames(renderer, dev_frame, world: int, steps: int, warmup: int, device) -> float:
    """`warmup` untimed frames, then exactly `steps` frames between barrier + synchronize; seconds, max over ranks."""
    import torch
    import torch.distributed as dist

    def step():
        renderer.render(dev_frame)
        if world > 1:
            renderer.gather(0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def phase_times(renderer, dev_frame, world: int, steps: int, device) -> dict:
    """Milliseconds per frame of the phases of StripRenderer.render / gather (HIP events on the streams the work
    runs on; mean over `steps` frames, max over ranks).  halo_wait is what of the exchange the interior work did
    not hide; gather is the RGBA8 transfer to rank 0 (on the side stream: it overlaps the next frame)."""
    import torch
    import torch.distributed as dist
    names = ["shade", "interior_bloom_post", "halo_wait", "edge_bloom_post", "gather"]
    acc = {n: 0.0 for n in names}
    for _ in range(steps):
        renderer.marks = []
        renderer.render(dev_frame)
        if world > 1:
            renderer.gather(0)
        torch.cuda.synchronize()
        ev = dict(renderer.marks)
        order = ["frame_start"] + names[:4]
        for a, b in zip(order, order[1:]):
            acc[b] += ev[a].elapsed_time(ev[b])
        if "gather" in ev:
            acc["gather"] += ev["gather_start"].elapsed_time(ev["gather"])
    renderer.marks = None
    t = torch.tensor([acc[n] / steps for n in names], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return {n: round(float(v), 4) for n, v in zip(names, t.tolist())}


# what the prediction below assumes of the fabric between GPUs (nothing here can measure it: the development box has
# one GPU): a point-to-point message of a few hundred KiB between neighbours, and the rate of one xGMI link
XGMI_SMALL_MESSAGE_MS = 0.03
XGMI_LINK_GBS = 50.0


def predict_strong_scaling(whole, scene_dev, width: int, height: int, worlds, steps: int, device) -> dict:
    """Frame times at `worlds` GPUs predicted from this one GPU: every strip of an N-way split is rendered here, alone,
    in a steady loop (composition + post chain, the clamped form: a strip without neighbours costs what one with halo
    rows costs); the slowest strip sets the frame, the halo exchange is priced at one small message that the interior
    rows do not hide, and the RGBA8 gather on rank 0 at N - 1 links' rate (it overlaps the next frame when two frames
    are in flight, so it is listed and not added).  Unmeasured on hardware: a figure to hold the first N-GPU run against."""
    import torch
    from vulkanrenderer_amd.strips import StripPlan, StripRenderer
    out = {}
    for n in worlds:
        per_strip = []
        for k in range(n):
            plan = StripPlan.make(k, n, height)
            frame = strip_device_frame(scene_dev, whole.rows(plan.row0, plan.row0 + plan.rows), 0, plan.rows, device)
            r = StripRenderer(StripPlan.make(0, 1, plan.rows), width, device=device, frames_in_flight=2)
            per_strip.append(timed_frames(r, frame, 1, steps, 2, device) / steps * 1e3)
            del r, frame
        torch.cuda.empty_cache()
        slowest = max(per_strip)
        halo = XGMI_SMALL_MESSAGE_MS if n > 1 else 0.0
        gather_bytes = width * height * 4 * (n - 1) // n
        gather = gather_bytes / (XGMI_LINK_GBS * 1e9 * max(min(n - 1, 7), 1)) * 1e3 + (XGMI_SMALL_MESSAGE_MS if n > 1 else 0.0)
        ms = slowest + halo
        out[str(n)] = {"strip_rows": height // n, "strip_ms_slowest": round(slowest, 4), "strip_ms_mean": round(sum(per_strip) / n, 4),
                       "halo_ms_assumed": halo, "gather_ms_at_link_rate_overlapped": round(gather, 4),
                       "predicted_ms_per_frame": round(ms, 4), "predicted_Mpixels/s": round(width * height / (ms * 1e-3) / 1e6, 1)}
    base = out.get("1", {}).get("predicted_ms_per_frame")
    if base:
        for n, rec in out.items():
            rec["predicted_speedup"] = round(base / rec["predicted_ms_per_frame"], 2)
    out["note"] = ("predicted from single-GPU strip timings, unmeasured on hardware; assumes "
                   f"{XGMI_SMALL_MESSAGE_MS} ms per neighbour message and {XGMI_LINK_GBS:g} GB/s per xGMI link")
    return out


def predict_ibl_sharded(worlds, device, reps: int = 3) -> dict:
    """The IBL precompute at the 2048^2 configuration split over N ranks, predicted from this one GPU: the work of every
    rank of an N-way split (environment and its tile copy on every rank, the rank's rows of the irradiance cube, of every
    prefilter mip and of the LUT) is run here alone; the slowest rank sets the time.  The all-gathers (1.2 MB in all) are
    priced at one small message each.  Unmeasured on hardware."""
    import torch
    from vulkanrenderer_amd import _lib, passes as P, synth
    from vulkanrenderer_amd.ibl import row_share
    src = P.to_device_f16(synth.make_equirect(np.random.default_rng(SEED + 77), 4096, 2048, hdr=True), device)
    env = torch.empty((6, 2048, 2048, 4), dtype=torch.float16, device=device)
    irr = torch.empty((6, 64, 64, 4), dtype=torch.float16, device=device)
    pres = [torch.empty((6, 128 >> m, 128 >> m, 4), dtype=torch.float16, device=device) for m in range(5)]
    lut = torch.empty((512, 512, 2), dtype=torch.float16, device=device)
    box = {}

    def rank_work(rank, world):
        P.equirect_to_cube(src, _lib.SRC_RGBA16F, env)
        foot = P.tile_env_footprints(env)
        r0, r1, _ = row_share(6 * 64, rank, world)
        box["ws"] = P.irradiance_by_region_rows(foot, 2048, irr, (r0, r1), workspace=box.get(("ws", rank, world)))
        box[("ws", rank, world)] = box["ws"]
        P.prefilter_mips(foot, pres, [row_share(6 * (128 >> m), rank, world)[:2] for m in range(5)], env_layout=_lib.ENV_LAYOUT_FOOTPRINT_TILES, env_n=2048)
        r0, r1, _ = row_share(512, rank, world)
        P.brdf_lut(lut, (r0, r1))

    out = {}
    for n in worlds:
        per_rank = []
        for k in range(n):
            rank_work(k, n)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                rank_work(k, n)
            e1.record()
            torch.cuda.synchronize()
            per_rank.append(e0.elapsed_time(e1) / reps)
        gathers = 7 * XGMI_SMALL_MESSAGE_MS if n > 1 else 0.0  # irradiance, five mips, LUT
        out[str(n)] = {"rank_ms_slowest": round(max(per_rank), 3), "rank_ms_mean": round(sum(per_rank) / n, 3), "all_gathers_ms_assumed": round(gathers, 3),
                       "predicted_ms": round(max(per_rank) + gathers, 3)}
    out["note"] = "predicted from one GPU running each rank's share alone; the environment (0.13 ms) and its tile copy (0.16 ms) are made on every rank; unmeasured on hardware"
    return out


def kernel_times(renderer, dev_frame, rows: int, steps: int) -> dict:
    """Average duration of the two kernels of a step (shade; both bloom passes + post-process as one kernel), each
    between two HIP events on the launch stream, and of the three-pass form of the post chain the one kernel
    replaces (bloom pass 1, bloom pass 2 fused with the post-process draw) on the same images."""
    import torch
    from vulkanrenderer_amd import _lib, passes as P
    names = ["shade", "bloom_post", "bloom_h", "bloomv_post"]
    acc = {n: 0.0 for n in names}
    be = renderer.backend
    lo, hi = 4, 4 + rows - 1
    b1 = torch.empty_like(renderer.threshold)
    ldr2 = torch.empty_like(renderer.ldr)
    for _ in range(steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        be.shade(dev_frame, renderer.color, renderer.threshold)
        ev[1].record()
        be.bloom_post(renderer.threshold_padded, renderer.color, renderer.ldr, lo, hi)
        ev[2].record()
        P.bloom_blur(renderer.threshold, b1, True, _lib.WRAP_REPEAT)
        ev[3].record()
        P.bloomv_postprocess(b1, renderer.color, ldr2)
        ev[4].record()
        torch.cuda.synchronize()
        for i, n in enumerate(names):
            acc[n] += ev[i].elapsed_time(ev[i + 1])
    out = {n: acc[n] / steps for n in names}
    out["three_pass_form_equal"] = bool(torch.equal(renderer.ldr, ldr2))
    return out


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


def time_shadow_ranges(dev_frame, renderer, compare: bool = True, reps: int = 5) -> dict:
    """The depth ranges of the point-shadow cubes: how long they take to build (a caller whose shadow pass runs every
    frame pays this every frame), and the shading kernel without them on the same frame (every tap read: the
    shader's own access pattern), with the two images compared."""
    import torch
    from vulkanrenderer_amd import passes as P
    import dataclasses
    def avg_ms(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    build_ms = avg_ms(lambda: P.cube_shadow_ranges(dev_frame.cube_shadow, dev_frame.cube_layout, dev_frame.cube_ranges))
    cube_bytes = dev_frame.cube_shadow.numel() * 4
    out = {"build_ms": round(build_ms, 4), "build_reads_GB/s": round(cube_bytes / (build_ms * 1e-3) / 1e9, 1),
           "bytes": dev_frame.cube_ranges.numel(), "cube_bytes": cube_bytes,
           "tile_classes_bytes": None if dev_frame.shade_workspace is None else dev_frame.shade_workspace.numel()}
    if compare:
        color, thr = torch.empty_like(renderer.color), torch.empty_like(renderer.threshold)
        plain = dataclasses.replace(dev_frame, cube_ranges=None)
        P.shade(dev_frame, renderer.color, renderer.threshold)
        without_ms = min(avg_ms(lambda: P.shade(plain, color, thr)) for _ in range(2))
        same = bool(torch.equal(color.view(torch.int16), renderer.color.view(torch.int16)) and torch.equal(thr.view(torch.int16), renderer.threshold.view(torch.int16)))
        out.update({"shade_ms_without_ranges": round(without_ms, 4), "same_image_without_ranges": same})
    return out


def time_ibl(device, reps: int = 3) -> tuple[dict, list]:
    """IBL precompute at the BASELINE configuration: 4096x2048 HDR equirect -> 2048^2 cube ->
    irradiance 64^2, 5-mip GGX prefilter (128^2 .. 8^2, 1024 samples) and 512^2 BRDF LUT.  Returns the
    per-stage milliseconds and one roofline record per kernel (average of `reps` launches, HIP events)."""
    import torch
    from vulkanrenderer_amd import _lib, passes as P, synth

    rng = np.random.default_rng(SEED + 4)
    src = P.to_device_f16(synth.make_equirect(rng, 4096, 2048, hdr=True), device)
    env = torch.empty((6, 2048, 2048, 4), dtype=torch.float16, device=device)
    irr = torch.empty((6, 64, 64, 4), dtype=torch.float16, device=device)
    pres = [torch.empty((6, 128 >> m, 128 >> m, 4), dtype=torch.float16, device=device) for m in range(5)]
    lut = torch.empty((512, 512, 2), dtype=torch.float16, device=device)
    irr_seq, irr_reg = torch.empty_like(irr), torch.empty_like(irr)
    box = {}

    def run():
        """The precompute as a renderer runs it: the irradiance pass in the library's default order (by region of the
        environment with row sums, from footprint tiles)."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ev[0].record()
        P.equirect_to_cube(src, _lib.SRC_RGBA16F, env)
        ev[1].record()
        box["foot"] = P.tile_env_footprints(env)  # the copy the irradiance and prefilter passes read
        ev[2].record()
        box["ws"] = P.irradiance_by_region_rows(box["foot"], 2048, irr, workspace=box.get("ws"))
        ev[3].record()
        P.prefilter_mips(box["foot"], pres, env_layout=_lib.ENV_LAYOUT_FOOTPRINT_TILES, env_n=2048)  # mip 0, then mips 1-4 in one launch
        ev[4].record()
        P.brdf_lut(lut)
        ev[5].record()
        torch.cuda.synchronize()
        return [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]

    run()
    t = np.mean([run() for _ in range(reps)], axis=0)
    eq, foot, irr_ms, pre_total, lut_ms = t
    uncovered = P.irradiance_uncovered(box["ws"])

    def timed(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # the prefilter mips one launch each (what the per-mip roofline records are made from), from both tiled layouts
    box["tiled"] = P.tile_env_cube(env)
    tile = timed(lambda: P.tile_env_cube(env))
    pre_ms = np.array([timed(lambda: P.prefilter(box["foot"], pres[m], m / 4.0, env_layout=_lib.ENV_LAYOUT_FOOTPRINT_TILES, env_n=2048)) for m in range(5)])
    pre_ms_4x4 = np.array([timed(lambda: P.prefilter(box["tiled"], pres[m], m / 4.0, env_layout=_lib.ENV_LAYOUT_TILED_4X4)) for m in range(5)])
    # beside it: the shader's own summation order (one running sum per texel), and the first by-region order
    seq_ms = timed(lambda: P.irradiance(box["tiled"], irr_seq, _lib.IRRADIANCE_SEQUENTIAL, env_layout=_lib.ENV_LAYOUT_TILED_4X4))
    ws2 = P.irradiance_by_region(box["tiled"], irr_reg, env_layout=_lib.ENV_LAYOUT_TILED_4X4)
    region_ms = timed(lambda: P.irradiance_by_region(box["tiled"], irr_reg, env_layout=_lib.ENV_LAYOUT_TILED_4X4, workspace=ws2))
    a = irr_seq.float().cpu().numpy()
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(a), 2.0 ** -14))) - 10)

    def against_shader_order(img):
        b = img.float().cpu().numpy()
        return {"texels_that_differ": int((P.f16_bits(irr_seq) != P.f16_bits(img)).any(-1).sum()), "of_texels": int(a.shape[0] * a.shape[1] * a.shape[2]),
                "max_difference_binary16_ulps": round(float((np.abs(a - b) / ulp).max()), 2)}

    orders = {"by_region_rows (default, timed in total_ms)": {"irradiance_ms": round(float(irr_ms), 3), "footprint_tiles_ms": round(float(foot), 3),
                                                              "workspace_bytes": int(box["ws"].numel()), "footprint_tile_bytes": int(box["foot"].numel()),
                                                              "texels_with_samples_unaccounted_for": uncovered, "against_the_shader_order": against_shader_order(irr)},
              "sequential (the shader's running sum)": {"irradiance_ms": round(float(seq_ms), 3), "total_ms": round(float(t.sum() - irr_ms + seq_ms + tile), 3)},
              "by_region": {"irradiance_ms": round(float(region_ms), 3), "total_ms": round(float(t.sum() - irr_ms + region_ms + tile), 3),
                            "texels_with_samples_unaccounted_for": P.irradiance_uncovered(ws2), "against_the_shader_order": against_shader_order(irr_reg)},
              "note": "every order is bit-exact against its own restatement in oracle/ (orc_irradiance, order 0 / 2 / 3); the by-region orders add the same "
                      "15 876 samples per texel in another order and stay within one binary16 ulp of the shader's"}
    stages = {"equirect_ms": round(float(eq + foot), 3), "irradiance_ms": round(float(irr_ms), 3), "prefilter_ms": round(float(pre_total), 3),
              "brdf_lut_ms": round(float(lut_ms), 3), "total_ms": round(float(t.sum()), 3),
              "total_ms_with_the_shader_summation_order": round(float(t.sum() - irr_ms + seq_ms + tile), 3),
              "config": "equirect 4096x2048 RGBA16F -> env 2048^2 (+ its copy in footprint tiles, which the irradiance and prefilter passes read), "
                        "irradiance 64^2 in the library's default order (VKR_IRRADIANCE_BY_REGION_ROWS), prefilter 128^2 x5 mips in two launches "
                        "(mip 0; mips 1-4 together), LUT 512^2",
              "prefilter_launch_per_mip_ms": {"footprint_tiles": [round(float(v), 4) for v in pre_ms], "sum": round(float(pre_ms.sum()), 4),
                                              "tiles_4x4": [round(float(v), 4) for v in pre_ms_4x4], "sum_4x4": round(float(pre_ms_4x4.sum()), 4)},
              "irradiance_orders": orders}

    def valu(kernel, ops, ms, algorithmic, **more):
        tf = ops / (ms * 1e-3) / 1e12
        return dict(kernel=kernel, bound="valu", algorithmic=algorithmic, achieved=round(tf, 3), peak=PEAK_FP32_TFLOPS, unit="TFLOP/s",
                    frac=round(tf / PEAK_FP32_TFLOPS, 4), avg_launch_ms=round(float(ms), 4), **more)

    def hbm(kernel, nbytes, ms, algorithmic):
        gbs = nbytes / (ms * 1e-3) / 1e9
        return dict(kernel=kernel, bound="hbm", algorithmic=algorithmic, achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                    frac=round(gbs / PEAK_HBM_GBS, 4), avg_launch_ms=round(float(ms), 4))

    irr_texels, lut_texels = 6 * 64 * 64, 512 * 512
    env_bytes, src_bytes = env.numel() * 2, src.numel() * 2
    # every irradiance sample gathers the 2x2 footprint of a bilinear lookup (4 texels x 8 B) from the 201 MB cube
    gather_bytes = irr_texels * IRRADIANCE_SAMPLES * 32
    samples = IRRADIANCE_SAMPLES * irr_texels
    roof = [
        hbm("equirect_kernel", env_bytes + src_bytes, eq, f"{env_bytes} B written (8 B/texel) + {src_bytes} B source read once"),
        hbm("tile_rgba16f_kernel", 2 * env_bytes, tile, f"{env_bytes} B read + {env_bytes} B written"),
        hbm("footprint_tiles_kernel", env_bytes + int(box["foot"].numel()), foot, f"{env_bytes} B read + {int(box['foot'].numel())} B written (16/9 of the cube)"),
        valu("irradiance_rows_kernel (+ irradiance_regions_sum_kernel)", OPS_IRRADIANCE_SAMPLE * samples, irr_ms,
             f"{OPS_IRRADIANCE_SAMPLE} FP32 ops x {IRRADIANCE_SAMPLES} samples x {irr_texels} texels (the default order; the kernel also decides about 1.05 "
             "cells per sample, which the figure does not count)",
             # what bounds it: every sample pulls the one 128-byte line that holds its 2x2 footprint from L2 (hit rate 97 %); the ceiling is the
             # rate at which random lines come out of an L2-resident table
             l2_lines={"bound": "L2 -> L1 cache lines", "bytes": samples * 128, "achieved": round(samples * 128 / (irr_ms * 1e-3) / 1e9, 1),
                       "peak": L2_RANDOM_LINE_GBS, "unit": "GB/s", "frac": round(samples * 128 / (irr_ms * 1e-3) / 1e9 / L2_RANDOM_LINE_GBS, 4),
                       "source": "one 128-byte line per sample (footprint tiles); peak = random 128-byte lines from a 2 MiB table, "
                                 "profiles/r2_gather_bench.txt (scratch/ubench/gather_bench), = 16 channels x 128 B x 8 XCDs x 2.1 GHz"}),
        valu("irradiance_texel_kernel (the shader's order)", OPS_IRRADIANCE_SAMPLE * samples, seq_ms,
             f"{OPS_IRRADIANCE_SAMPLE} FP32 ops x {IRRADIANCE_SAMPLES} samples x {irr_texels} texels",
             gather={"bytes": gather_bytes, "GB/s": round(gather_bytes / (seq_ms * 1e-3) / 1e9, 1),
                     "note": "32 B of texels per bilinear sample; bound by L2-miss line traffic (70.6 GB per launch, profiles/r2_counters.json)"}),
        valu("irradiance_region_kernel (+ irradiance_regions_sum_kernel)", OPS_IRRADIANCE_SAMPLE * samples, region_ms,
             f"{OPS_IRRADIANCE_SAMPLE} FP32 ops x {IRRADIANCE_SAMPLES} samples x {irr_texels} texels (the first by-region order)"),
    ]
    for m in range(5):
        n = 128 >> m
        roof.append(valu(f"prefilter_texel_kernel mip {m} (roughness {m / 4:g})", OPS_PREFILTER_SAMPLE * GGX_SAMPLES * 6 * n * n, pre_ms[m],
                         f"{OPS_PREFILTER_SAMPLE} FP32 ops x {GGX_SAMPLES} samples x {6 * n * n} texels"))
    roof.append(valu("prefilter_texel_kernel, 5 mips", OPS_PREFILTER_SAMPLE * GGX_SAMPLES * sum(6 * (128 >> m) ** 2 for m in range(5)), pre_total,
                     f"{OPS_PREFILTER_SAMPLE} FP32 ops x {GGX_SAMPLES} samples x 130944 texels, two launches (mip 0; mips 1-4 together) as the precompute runs them"))
    roof.append(valu("brdf_lut_kernel", OPS_BRDF_SAMPLE * GGX_SAMPLES * lut_texels, lut_ms,
                     f"{OPS_BRDF_SAMPLE} FP32 ops x {GGX_SAMPLES} samples x {lut_texels} texels"))
    return stages, roof


def time_ibl_sharded(rank: int, world: int, device, reps: int = 3) -> dict:
    """The same precompute with every rank computing an equal share of the rows of the irradiance cube, of each
    prefilter mip and of the LUT, and the shares all-gathered (every rank needs whole images for shading); the
    environment cube is resampled on every rank.  Milliseconds per run, slowest rank; all ranks call this."""
    import torch
    import torch.distributed as dist
    from vulkanrenderer_amd import _lib, passes as P, synth
    from vulkanrenderer_amd.ibl import IblConfig, IblPrecompute
    src = P.to_device_f16(synth.make_equirect(np.random.default_rng(SEED + 77), 4096, 2048, hdr=True), device)

    def timed(order):
        cfg = IblConfig(env_res=2048, irradiance_res=64, prefilter_res=128, prefilter_mips=5, lut_res=512, irradiance_order=order)
        pre = IblPrecompute(cfg, rank, world, device=device)
        pre.run(src, _lib.SRC_RGBA16F)
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = pre.run(src, _lib.SRC_RGBA16F)
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        del res
        return round(float(t.item()), 3)

    return {"ms": timed(_lib.IRRADIANCE_FASTEST), "ms_with_the_shader_summation_order": timed(_lib.IRRADIANCE_SEQUENTIAL), "world": world,
            "config": "equirect 4096x2048 -> env 2048^2 on every rank; rows of irradiance 64^2, prefilter 128^2 x5 mips and LUT 512^2 split "
                      "over the ranks and all-gathered; ms: the irradiance pass in the library's default order (by region with row sums)"}


def time_coverage(f, dev_frame, renderer, device) -> dict:
    """Light-volume coverage (VKR_COVERAGE_MESH) at the benchmark size: the mask kernel, and the shading kernel
    consuming the masks.  With the reference's light radii every volume contains the camera, so every mask
    bit is set and the image is the full-coverage one; a second scene shrinks the volumes to show what the
    per-light masks save when lights are local."""
    import dataclasses
    import torch
    from vulkanrenderer_amd import passes as P, synth

    verts, tris = synth.reference_light_volume()
    dverts, dtris = torch.from_numpy(verts).to(device), torch.from_numpy(tris.astype(np.int32)).to(device)

    def timed(fn, reps=10):
        for _ in range(3):  # clocks and caches settle over the first launches of a different kernel
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {}
    want = renderer.color.clone()
    masks, flags = P.light_coverage_with_flags(dev_frame, dverts, dtris)
    masked = dataclasses.replace(dev_frame, coverage=masks, coverage_flags=flags)
    color, thr = torch.empty_like(renderer.color), torch.empty_like(renderer.threshold)
    out["reference_radii"] = {"coverage_ms": round(timed(lambda: P.light_coverage_with_flags(dev_frame, dverts, dtris)), 4),
                              "shade_ms": round(timed(lambda: P.shade(masked, color, thr)), 4),
                              "lights_flagged_as_covering_every_pixel": int(flags[:-1].sum()), "lights": int(flags.numel() - 1),
                              "covered_fraction": 1.0 if bool((masked.coverage == -1).all()) else None,
                              "equals_full_coverage_image": bool(torch.equal(color, want)),
                              "note": "the camera is inside every volume: the classification kernel says so per light, the masks are "
                                      "written as all ones without rasterising, and the shading kernel takes its loop without masks"}
    unflagged = dataclasses.replace(dev_frame, coverage=P.light_coverage(dev_frame, dverts, dtris))
    out["reference_radii_rasterised"] = {"coverage_ms": round(timed(lambda: P.light_coverage(dev_frame, dverts, dtris)), 4),
                                         "shade_ms": round(timed(lambda: P.shade(unflagged, color, thr)), 4),
                                         "masks_equal": bool(torch.equal(unflagged.coverage, masks)),
                                         "note": "the same lights without the per-light words: every volume rasterised, every mask read"}
    lights = f.lights.copy()
    for i in range(len(lights)):
        lights["color"][i] *= 0.02
        r = synth.light_volume_radius(lights["color"][i])
        m = np.diag([r, r, r, 1.0]); m[3, :3] = lights["position"][i]
        lights["model"][i] = m.astype(np.float32).ravel()
    small = dataclasses.replace(dev_frame, lights=torch.from_numpy(np.frombuffer(lights.tobytes(), dtype=np.uint8).copy()).to(device))
    small.coverage, small.coverage_flags = P.light_coverage_with_flags(small, dverts, dtris)
    bits = small.coverage.view(torch.uint8)
    frac = float(sum(int(((bits >> k) & 1).sum()) for k in range(8))) / (len(lights) * f.width * f.height)
    out["local_lights"] = {"coverage_ms": round(timed(lambda: P.light_coverage_with_flags(small, dverts, dtris)), 4),
                           "shade_ms": round(timed(lambda: P.shade(small, color, thr)), 4),
                           "covered_fraction": round(frac, 4),
                           "note": "same lights with colours x0.02: volume radius 16*sqrt(max/5) shrinks to ~1-2.5 units"}
    out["mesh"] = f"the reference's light volume (assets/models/sphere/sphere.glb as arrays, tests/golden/light_volume.npz): {len(verts)} vertices, {len(tris)} triangles"
    # the rasterising form priced by the bytes it cannot avoid: one 64-bit mask per light and 8x8-pixel tile
    mask_bytes = len(f.lights) * ((f.width + 7) // 8) * ((f.height + 7) // 8) * 8
    for key in ("reference_radii_rasterised", "local_lights"):
        ms = out[key]["coverage_ms"]
        gbs = mask_bytes / (ms * 1e-3) / 1e9
        out[key]["roofline"] = dict(kernel="coverage_classify_kernel + coverage_kernel", bound="hbm", algorithmic=f"{mask_bytes} B of masks written", achieved=round(gbs, 1),
                                    peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 5), avg_launch_ms=ms,
                                    note="the masks are 4 MB; the time is the binary64 set-up of 960 triangles per (light, 128^2-pixel block), far from any memory bound")
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
    # A frame from draw calls end to end (no sum of separately timed passes): every pass a rank runs between "meshes,
    # materials, lights and camera are given" and "8-bit rows are ready" -- point shadows, directional shadow, geometry
    # pass, skybox, light-volume coverage, composition, post chain -- enqueued frame after frame on one stream; for the
    # whole frame, and for the rows one rank of eight owns (every rank renders all shadow maps: that part does not shrink).
    import dataclasses
    from vulkanrenderer_amd.strips import HALO, HipBackend, StripPlan
    sv, st = synth.reference_light_volume()
    volume = (torch.from_numpy(sv).to(device), torch.from_numpy(st.astype(np.int32)).to(device))
    be = HipBackend(device)

    def from_draws(plan, reps=8):
        # Two frames in flight, as the reference keeps them (MaxFramesInFlight, VulkanRenderer.cs:60): each has its own
        # targets, tile-class workspace and stream, so the clear, the shadow passes and the depth ranges of frame k + 1
        # (bound by memory and by binary64 arithmetic) run beside the composition pass of frame k.
        r = plan.rows
        lo, hi = (HALO if plan.is_top else 0), (HALO + r - 1 if plan.is_bottom else r + 2 * HALO - 1)
        slots = []
        for _ in range(2):
            ws = torch.empty(P.shade_workspace_bytes(f.width, r, nl), dtype=torch.uint8, device=device)
            slots.append(dict(stream=torch.cuda.Stream(device=device),
                              base=dataclasses.replace(dev_frame, rows=r, scene=scene, coverage=None, coverage_flags=None, background=None, shade_workspace=ws),
                              color=torch.empty((r, f.width, 4), dtype=torch.float16, device=device),
                              padded=torch.empty((r + 2 * HALO, f.width, 4), dtype=torch.float16, device=device),
                              ldr=torch.empty((r, f.width, 4), dtype=torch.uint8, device=device)))

        def frame(k):
            s = slots[k % 2]
            s["stream"].wait_stream(torch.cuda.current_stream(device))  # the frame's inputs
            with torch.cuda.stream(s["stream"]):
                fr = be.frame_from_draws(s["base"], gdraws, res, dres, plan.total_rows, plan.row0, (env, _lib.ENV_LAYOUT_TILED_4X4))
                fr = be.with_coverage(fr, volume, plan.total_rows, plan.row0)
                be.shade(fr, s["color"], s["padded"][HALO:HALO + r])
                be.bloom_post(s["padded"], s["color"], s["ldr"], lo, hi)  # a strip's halo rows would arrive here; their content does not change the time

        def join():
            for s in slots:
                torch.cuda.current_stream(device).wait_stream(s["stream"])
        frame(0); frame(1); join(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(reps):
            frame(k)
        join()
        e1.record(); torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / reps, 4)
    padded_note = ("two frames in flight, each on its own stream with its own targets; halo rows are not exchanged (one GPU): "
                   "the post kernel reads whatever the halo rows hold")
    end_to_end = {"whole_frame_ms": from_draws(StripPlan.make(0, 1, f.height)),
                  "rows_of_rank_3_of_8_ms": from_draws(StripPlan.make(3, 8, f.height)), "note": padded_note}
    point_ms = timed(lambda: P.point_shadow_maps(draws, dev_frame.lights, nl, res))
    dir_ms = timed(lambda: P.dir_shadow_map(draws, f.scene["light_space_matrix"], dres, device))
    gpass_ms = timed(lambda: P.gbuffer(gdraws, scene, f.width, f.height, device))
    sky_ms = timed(lambda: P.skybox(env, scene, f.width, f.height, env_layout=_lib.ENV_LAYOUT_TILED_4X4))
    cubes, cube_layout = P.point_shadow_maps(draws, dev_frame.lights, nl, res)
    cube_cov = float((cubes < (25.0 if cube_layout == _lib.CUBE_LAYOUT_TILED_8X4_X25 else 1.0)).float().mean())
    dmap = P.dir_shadow_map(draws, f.scene["light_space_matrix"], dres, device)
    gcov = float((ga[..., 3] != 0).float().mean())

    def hbm(kernel, nbytes, ms, algorithmic):
        gbs = nbytes / (ms * 1e-3) / 1e9
        return dict(kernel=kernel, bound="hbm", algorithmic=algorithmic, achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                    frac=round(gbs / PEAK_HBM_GBS, 4), avg_launch_ms=round(float(ms), 4))
    # Rasterising passes, priced by the bytes a pass cannot avoid: every target texel is cleared and every covered one
    # written at least once more.  Their arithmetic (binary64 edge functions per sample of every triangle's bounding box,
    # L2 atomics for the depth test) is what the time goes into; the fraction says how far that is from the memory floor.
    cube_bytes = cubes.numel() * 4
    px = f.width * f.height
    roof = [hbm("fill_depth_kernel + point_shadow_kernel", int(cube_bytes * (1 + cube_cov)), point_ms,
                f"{cube_bytes} B cleared + {cube_cov:.3f} of them written again (covered texels), {len(tris)} triangles x {nl} lights x 6 faces set up"),
            hbm("fill_depth_kernel + dir_shadow_kernel", int(dmap.numel() * 4 * (1 + float((dmap < 1).float().mean()))), dir_ms,
                f"{dmap.numel() * 4} B cleared + the covered texels written again"),
            hbm("gpass_visibility_kernel + gpass_resolve_kernel", int(px * (8 + 8 * gcov + 8 + 32)), gpass_ms,
                "per pixel: 8 B visibility key reset, written where covered and read back, 32 B of G-buffer written; material texels from cache"),
            hbm("skybox_kernel", px * 8, sky_ms, "8 B per pixel written; one bilinear cube lookup per pixel from cache")]
    return {"from_draw_calls_end_to_end": end_to_end,
            "point_shadow_ms": round(point_ms, 4), "point_shadow_covered_fraction": round(cube_cov, 4),
            "dir_shadow_ms": round(dir_ms, 4),
            "geometry_pass_ms": round(gpass_ms, 4),
            "geometry_pass_covered_fraction": round(gcov, 4),
            "skybox_ms": round(sky_ms, 4),
            "roofline": roof,
            "config": f"{nl} lights x 6 x {res}^2 D32F (written as 8x4 tiles) + {dres}^2 directional map; mesh {len(verts)} vertices, {len(tris)} triangles"}


COUNTER_FILE = os.path.join("profiles", "r3_counters.json")


def measured_counters(width: int, rows: int, lights: int, cube_res: int, dir_res: int) -> dict:
    """Per-launch hardware-counter figures from the committed rocprofv3 --pmc passes of this command
    (profiles/r2_counters.json): HBM-side bytes, vector instructions, gather instructions.  Counters cannot
    be read from inside this process, so the figures apply only to the workload they were collected on and
    are dropped for any other."""
    try:
        with open(os.path.join(ROOT, COUNTER_FILE)) as fh:
            rec = json.load(fh)
    except OSError:
        return {}
    w = rec.get("workload", {})
    if (w.get("width"), w.get("rows"), w.get("lights"), w.get("cube_res"), w.get("dir_res")) != (width, rows, lights, cube_res, dir_res):
        return {}
    return rec.get("kernels", {})


PAIR_COUNT_FILE = "profiles/r3_shade_pair_counts.json"
PAIR_COUNT_NAMES = ["no_live_lane", "one_face_settled_per_pixel_by_a_range", "one_face_taps_read", "edge_or_corner_taps_read",
                    "all_live_lanes_fully_shadowed_no_brdf", "brdf_evaluated", "tile_class_skip", "tile_class_lit"]


def shade_pair_counts(dev_frame, renderer) -> dict:
    """Counts kept by a library built with -DVKR_EXPERIMENT_COUNT over one launch of the composition pass: how many
    (wave, light) pairs the tile classes skipped or took as lit, how many the per-pixel range test settled, how many read
    their 20 taps, how many evaluated the BRDF."""
    import ctypes
    import torch
    from vulkanrenderer_amd import _lib, passes as P
    lib = _lib.load()
    fn = getattr(lib, "vkr_experiment_shade_counts", None)
    if fn is None:
        raise SystemExit("--pair-counts needs a library built with -DVKR_EXPERIMENT_COUNT (scratch/build_variant.sh, VKR_HIP_LIB)")
    buf = (ctypes.c_ulonglong * 8)()
    torch.cuda.synchronize()
    fn(buf, 1)
    P.shade(dev_frame, renderer.color, renderer.threshold)
    torch.cuda.synchronize()
    fn(buf, 1)
    c = dict(zip(PAIR_COUNT_NAMES, (int(v) for v in buf)))
    c["pairs"] = c["tile_class_skip"] + c["no_live_lane"] + c["tile_class_lit"] + c["one_face_settled_per_pixel_by_a_range"] + c["one_face_taps_read"] + c["edge_or_corner_taps_read"]
    # a pair without a live lane is counted before the tile class is looked at; a lit tile with live lanes goes on to the BRDF
    c["workload"] = {"width": dev_frame.width, "rows": dev_frame.rows, "lights": dev_frame.n_lights, "cube_res": dev_frame.cube_shadow.shape[-1]}
    return c


def stored_pair_counts(width: int, rows: int, lights: int, cube_res: int) -> dict:
    try:
        with open(os.path.join(ROOT, PAIR_COUNT_FILE)) as fh:
            rec = json.load(fh)
    except OSError:
        return {}
    if rec.get("workload") != {"width": width, "rows": rows, "lights": lights, "cube_res": cube_res}:
        return {}
    rec["source"] = f"{PAIR_COUNT_FILE}: one launch of this workload by a library built with -DVKR_EXPERIMENT_COUNT (bench.py --pair-counts), not counted in this run"
    return rec


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


def extra_shapes(args, rank: int, world: int, device, scene: dict, scene_dev, weak: bool = True) -> dict:
    """The two other N > 1 shapes: a full-size strip per rank (weak scaling) and the per-GPU share of the
    16384^2 / 256-light frame.  Inputs are synthetic repeats: timing only, the parity of these shapes is the
    test suite's business."""
    import dataclasses
    import torch
    from vulkanrenderer_amd import passes as P, synth
    from vulkanrenderer_amd.strips import StripPlan, StripRenderer
    out = {}
    steps, warmup = max(args.steps // 2, 3), 2

    # (a) weak: every rank owns a WIDTH x HEIGHT strip of a WIDTH x (HEIGHT * world) frame
    if weak:
      g = synth.make_gbuffer(np.random.default_rng(SEED + 1000 + rank), WIDTH, HEIGHT)
      f = frame_of(scene, g, WIDTH, HEIGHT)
      dev_frame = strip_device_frame(scene_dev, f, 0, HEIGHT, device)
      r = StripRenderer(StripPlan.make(rank, world, HEIGHT * world), WIDTH, device=device, frames_in_flight=2)
      t = timed_frames(r, dev_frame, world, steps, warmup, device)
      out["weak_full_strip_per_rank"] = {"frame": f"{WIDTH}x{HEIGHT * world}", "strip": f"{WIDTH}x{HEIGHT}", "lights": args.lights, "scaling": "weak",
                                         "ms_per_step": round(t / steps * 1e3, 4), "Mpixels/s": round(WIDTH * HEIGHT * world / (t / steps) / 1e6, 1),
                                         "phases_ms": phase_times(r, dev_frame, world, steps, device)}
      del r, dev_frame, f, g
      torch.cuda.empty_cache()

    # (b) the 16384^2 / 256-light configuration: 16384 x 2048 rows per rank, 256 lights with 1024^2 cubes
    reps = C5_LIGHTS // args.lights
    if reps >= 1 and C5_LIGHTS % args.lights == 0:
        g = synth.make_gbuffer(np.random.default_rng(SEED + 2000 + rank), 4096, C5_ROWS)
        wide = [np.ascontiguousarray(np.tile(a, (1, C5_WIDTH // 4096, 1))) for a in g]
        lights = np.concatenate([scene["lights"]] * reps)
        dev_lights = torch.from_numpy(np.frombuffer(lights.tobytes(), dtype=np.uint8).copy()).to(device)
        cubes = scene_dev.cube_shadow.repeat(reps, 1, 1, 1)
        base = dataclasses.replace(scene_dev, width=C5_WIDTH, lights=dev_lights, n_lights=C5_LIGHTS, cube_shadow=cubes,
                                   cube_ranges=None if scene_dev.cube_ranges is None else scene_dev.cube_ranges.repeat(reps, 1))
        f = synth.FrameInputs(C5_WIDTH, C5_ROWS, *wide, scene["scene"], lights, scene["dir_shadow"], scene["cube_shadow"],
                              scene["irradiance"], scene["prefilter"], scene["brdf_lut"])
        dev_frame = strip_device_frame(base, f, 0, C5_ROWS, device)
        r = StripRenderer(StripPlan.make(rank, world, C5_ROWS * world), C5_WIDTH, device=device, frames_in_flight=2)
        t = timed_frames(r, dev_frame, world, 3, 1, device)
        out["c5_strip_per_rank"] = {"frame": f"{C5_WIDTH}x{C5_ROWS * world}", "strip": f"{C5_WIDTH}x{C5_ROWS}", "lights": C5_LIGHTS, "scaling": "weak",
                                    "ms_per_step": round(t / 3 * 1e3, 3), "Mpixels/s": round(C5_WIDTH * C5_ROWS * world / (t / 3) / 1e6, 1),
                                    "phases_ms": phase_times(r, dev_frame, world, 3, device),
                                    "inputs": f"{C5_LIGHTS} lights = the {args.lights} synthetic lights and their {args.cube_res}^2 depth cubes repeated x{reps} "
                                              f"({cubes.numel() * 4 / 2**30:.1f} GiB resident); G-buffer = a 4096-wide synthetic strip repeated x4"}
        del r, dev_frame, f, wide, g, cubes, base
        torch.cuda.empty_cache()
    return out


def main() -> None:
    global WIDTH, HEIGHT
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=WIDTH, help="frame width (default: the headline 3840)")
    ap.add_argument("--height", type=int, default=HEIGHT, help="frame rows, split over the GPUs (default: the headline 2160)")
    ap.add_argument("--lights", type=int, default=LIGHTS)
    ap.add_argument("--cube-res", type=int, default=CUBE_RES)
    ap.add_argument("--dir-res", type=int, default=DIR_RES)
    ap.add_argument("--cpu-rows", type=int, default=None, help="rows of the frame the CPU baseline shades (default: all; 0 = skip)")
    ap.add_argument("--no-ibl", action="store_true", help="skip the IBL precompute timing")
    ap.add_argument("--no-coverage", action="store_true", help="skip the light-volume coverage and shadow-pass timings")
    ap.add_argument("--pair-counts", action="store_true", help="with a library built with -DVKR_EXPERIMENT_COUNT (VKR_HIP_LIB): what happens to the "
                    "(wave, light) pairs of one launch of the shading kernel, added to the JSON line as shade_pair_counts")
    ap.add_argument("--no-ranges-compare", action="store_true", help="do not launch the shading kernel without depth ranges beside the timed one "
                    "(counter passes: every shade_kernel launch of the run is then the timed form)")
    ap.add_argument("--no-extra", action="store_true", help="N > 1: skip the extra shapes (a full strip per rank and the 16384-wide / 256-light strip)")
    ap.add_argument("--extra", action="store_true", help="N = 1: also time the 16384x2048 / 256-light strip (off by default, so that every launch of the "
                    "shading kernel in a default run has the shape the roofline is quoted on and a kernel trace of the run averages one shape)")
    ap.add_argument("--predict", action="store_true", help="N = 1: add the predicted strong-scaling table for 1, 2, 4, 8 GPUs (every strip of those splits rendered "
                    "alone on this GPU; off by default for the reason --extra is)")
    ap.add_argument("--no-predict", action="store_true", help="N > 1: skip the prediction for this N that is printed beside the measured phases")
    ap.add_argument("--transport-only", action="store_true", help="rehearse launcher, halo exchange and gather on CPU tensors (no GPU, no kernels)")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="seconds the self-started ranks may take")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")

    # Self-launch: with --gpus N > 1 and no rank environment, start the N ranks as fresh child processes.
    # Nothing has touched a GPU at this point (torch is not even imported), and the parent never does.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, args.launch_timeout))

    WIDTH, HEIGHT = args.width, args.height
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.transport_only:
        return run_transport_only(args, rank, world)

    import torch
    import torch.distributed as dist
    from vulkanrenderer_amd import _lib, passes as P, synth
    from vulkanrenderer_amd.strips import StripPlan, StripRenderer

    _lib.load()
    args.cpu_rows = HEIGHT if args.cpu_rows is None else min(args.cpu_rows, HEIGHT)
    # One rank per GPU over RCCL.  VKR_DIST_BACKEND=gloo: rehearsal on a box with fewer GPUs than ranks.
    backend = os.environ.get("VKR_DIST_BACKEND", "nccl")
    n_dev = max(torch.cuda.device_count(), 1)
    if backend == "nccl" and world > n_dev:
        raise SystemExit(f"{world} ranks over RCCL need {world} GPUs; {n_dev} visible (VKR_DIST_BACKEND=gloo rehearses on fewer)")
    dev_index = local_rank if backend == "nccl" else local_rank % n_dev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    def note(msg: str) -> None:  # progress on stderr: a multi-rank run is silent for a minute or more otherwise
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    note(f"{world} rank(s), backend {backend if world > 1 else 'none'}: building the scene")
    # the frame: every rank builds the same scene and the same whole-frame G-buffer and keeps its own rows
    scene = build_scene(device, args.lights, args.cube_res, args.dir_res)
    whole = frame_of(scene, synth.make_gbuffer(np.random.default_rng(SEED + 1000), WIDTH, HEIGHT), WIDTH, HEIGHT)
    plan = StripPlan.make(rank, world, HEIGHT)
    f = whole.rows(plan.row0, plan.row0 + plan.rows)
    dev_frame = P.DeviceFrame.upload(f, device)  # scene data + this rank's rows; cubes re-laid into tiles once
    # two frames in flight, like the reference (VulkanRenderer.cs:60), each on a stream of its own: the last waves and
    # the gather of frame k overlap the first waves of frame k+1
    renderer = StripRenderer(plan, WIDTH, device=device, frames_in_flight=2)

    note("timed frames")
    elapsed = timed_frames(renderer, dev_frame, world, args.steps, args.warmup, device)
    phases = phase_times(renderer, dev_frame, world, args.steps, device)
    kernel_ms = kernel_times(renderer, dev_frame, plan.rows, args.steps)

    # N > 1: the gathered image must be the single-GPU image, byte for byte
    gathered_ok = None
    if world > 1:
        renderer.render(dev_frame)
        parts = renderer.gather(0)
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            got = torch.cat(parts, 0)
            alone = StripRenderer(StripPlan.make(0, 1, HEIGHT), WIDTH, device=device)
            want = alone.render(strip_device_frame(dev_frame, whole, 0, HEIGHT, device))
            gathered_ok = bool(torch.equal(got, want))
            del alone, want, got
        # leave the strip targets holding the strip's own frame for the checks below
        renderer.render(dev_frame)
        torch.cuda.synchronize()

    extra = {}
    note("extra shapes" if (world > 1 and not args.no_extra) or (world == 1 and args.extra) else "no extra shapes")
    if (world > 1 and not args.no_extra) or (world == 1 and args.extra):
        # N > 1: a full strip per rank (weak scaling) and the 16384-wide / 256-light strip per rank; N = 1 (--extra): the latter alone
        extra = extra_shapes(args, rank, world, device, scene, dev_frame, weak=world > 1)
        if world > 1:
            dist.barrier()

    # N > 1: the IBL precompute with its rows split over the ranks and all-gathered (vulkanrenderer_amd/ibl.py)
    ibl_sharded = None
    if world > 1 and not args.no_ibl:
        note("IBL precompute split over the ranks")
        ibl_sharded = time_ibl_sharded(rank, world, device)
    note("per-kernel figures on rank 0")

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    px = WIDTH * HEIGHT
    strip_px = WIDTH * plan.rows
    ms_per_step = elapsed / args.steps * 1e3
    # background texels (cleared G-buffer: ~10% of the synthetic frame) are discarded by the composition
    # shader and take nothing from the lights, so they are not counted as shading work
    shaded_px = int((f.normal.view(np.float16)[..., :3] != 0).any(axis=-1).sum())
    flops = (OPS_PER_PIXEL + OPS_PER_PIXEL_LIGHT * args.lights) * shaded_px
    shade_tflops = flops / (kernel_ms["shade"] * 1e-3) / 1e12
    ctr = measured_counters(WIDTH, plan.rows, args.lights, args.cube_res, args.dir_res)
    uses_classes = dev_frame.cube_ranges is not None and args.lights >= _lib.SHADOW_RANGES_MIN_LIGHTS  # passes.shade's default
    shade_ctr = ctr.get("shade_kernel", {})
    roofline = {"bound": "valu", "kernel": "shade_kernel" + (" (+ light_classes_kernel)" if uses_classes else ""), "achieved": round(shade_tflops, 3), "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s", "frac": round(shade_tflops / PEAK_FP32_TFLOPS, 4), "traffic": shade_ctr.get("traffic_bytes"),
                "traffic_source": f"{COUNTER_FILE}: separate rocprofv3 --pmc passes of this command, (2 x FETCH_SIZE + WRITE_SIZE) x 1024, not read in this run" if shade_ctr else None,
                "algorithmic": f"({OPS_PER_PIXEL} + {OPS_PER_PIXEL_LIGHT} x {args.lights}) canonical FP32 ops x {shaded_px} shaded pixels per launch "
                               f"({strip_px - shaded_px} background pixels of the {strip_px} do no shading work); a canonical-work rate: "
                               "the kernel reaches the same bits in fewer issued instructions (see valu_lane_instructions), skips lights "
                               "that add exactly zero to a whole wave and takes the outcome of a light's 20 shadow taps from one depth range "
                               "where the range settles it (see shadow_ranges), so the fraction depends on the scene and can pass 1",
                "avg_launch_ms": round(kernel_ms["shade"], 4)}
    if shade_ctr:
        roofline["valu_wave_instructions"] = shade_ctr.get("SQ_INSTS_VALU")
        if shade_ctr.get("SQ_INSTS_VALU"):
            roofline["valu_lane_instructions"] = shade_ctr["SQ_INSTS_VALU"] * 64
            roofline["canonical_ops_per_lane_instruction"] = round(flops / (shade_ctr["SQ_INSTS_VALU"] * 64), 3)
        # how busy the two units are that bound the kernel (same counter file): the 256 texture addressers, against the
        # cycles the launch was active on an XCD; the vector ALUs, whose busy time is tallied in units of four cycles
        if shade_ctr.get("TA_TA_BUSY_sum") and shade_ctr.get("GRBM_GUI_ACTIVE"):
            cycles = shade_ctr["GRBM_GUI_ACTIVE"] / 8
            roofline["texture_addresser_busy_frac"] = round(shade_ctr["TA_TA_BUSY_sum"] / 256 / cycles, 3)
            if shade_ctr.get("SQ_INSTS_VMEM_RD"):
                roofline["gather_instructions"] = int(shade_ctr["SQ_INSTS_VMEM_RD"])
                roofline["addresser_cycles_per_gather"] = round(shade_ctr["TA_TA_BUSY_sum"] / shade_ctr["SQ_INSTS_VMEM_RD"], 2)
            if shade_ctr.get("SQ_ACTIVE_INST_VALU"):
                roofline["valu_busy_frac"] = round(min(shade_ctr["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cycles), 1.0), 3)
    pair_counts = shade_pair_counts(dev_frame, renderer) if args.pair_counts else stored_pair_counts(WIDTH, plan.rows, args.lights, args.cube_res)
    if pair_counts:
        roofline["wave_light_pairs"] = pair_counts
        # the canonical count again, for the work the kernel still performs: 681 of the 810 operations per pixel and light are
        # the 20 shadow taps, which only the pairs that read them perform
        tap_frac = (pair_counts["one_face_taps_read"] + pair_counts["edge_or_corner_taps_read"]) / max(pair_counts["pairs"], 1)
        performed = (OPS_PER_PIXEL + (OPS_PER_PIXEL_LIGHT - 681 + 681 * tap_frac) * args.lights) * shaded_px
        roofline["performed_work"] = {"pairs_reading_taps_frac": round(tap_frac, 4), "achieved": round(performed / (kernel_ms["shade"] * 1e-3) / 1e12, 3),
                                      "frac": round(performed / (kernel_ms["shade"] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4),
                                      "note": "canonical operations of the taps counted only for the (wave, light) pairs that read them; what is left is mostly the "
                                              "Cook-Torrance evaluation, whose canonical operations are close to one issued instruction each (a tap's 34 are about 8)"}
    def hbm_rec(kernel, bytes_per_px, ms, algorithmic, ctr_name):
        gbs = bytes_per_px * strip_px / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                "avg_launch_ms": round(ms, 4), "algorithmic": algorithmic, "traffic": ctr.get(ctr_name, {}).get("traffic_bytes")}
    post = [hbm_rec("bloom_post_fused_kernel", 20, kernel_ms["bloom_post"],
                    "20 B/pixel (8 thresholded + 8 colour in, 4 RGBA8 out); both bloom passes and the post-process draw, neither bloom image stored; "
                    "the kernel is bound by vector-instruction issue, not by memory", "bloom_post_fused_kernel")]
    post[0]["replaces"] = {"kernels": [hbm_rec("bloom_h_kernel", 16, kernel_ms["bloom_h"], "16 B/pixel", "bloom_h_kernel"),
                                       hbm_rec("bloom_v_kernel<fused post>", 20, kernel_ms["bloomv_post"], "20 B/pixel (8 bloom-1 + 8 colour in, 4 RGBA8 out)", "bloom_v_kernel")],
                           "ms": round(kernel_ms["bloom_h"] + kernel_ms["bloomv_post"], 4), "same_bytes_out": kernel_ms["three_pass_form_equal"]}
    out = {
        "metric": "Mpixels/s deferred PBR+bloom+ACES @4K 64 shadowed lights",
        "value": round(px / (elapsed / args.steps) / 1e6, 3),
        "unit": "Mpixels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,  # the same frame at every N (one strip at N = 1)
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{WIDTH}x{HEIGHT} frame" + (f" cut into {world} strips of {plan.rows} rows, one per GPU" if world > 1 else "") +
                               f", {args.lights} point lights with {args.cube_res}^2 cube shadow maps + {args.dir_res}^2 directional PCF shadow; "
                               "composition + bloom x2 + ACES/gamma" + ("; halo exchange and RGBA8 gather on rank 0 inside the timed region" if world > 1 else ""),
                   "parallelism": f"strips{world}",
                   "storage": "RGBA16F G-buffer and targets, RGBA8 output, D32F shadow cubes in 8x4-texel tiles holding 25 x depth (the product light.frag forms per tap)"
                              + (", with their depth ranges beside them" if uses_classes else "")},
        "distributed": {"backend": backend if world > 1 else None, "world": world,
                        "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if world > 1 and backend == "nccl" else None,
                        "devices": torch.cuda.device_count()},
        "phases_ms": phases,
        "roofline": roofline,
        "roofline_post": post,
    }
    if world > 1:
        out["gathered_equals_single_gpu"] = gathered_ok
    if (world > 1 and not args.no_predict) or (world == 1 and args.predict):
        # N > 1: this N, to hold beside the measured phases above; N = 1 (--predict): the curve for 1, 2, 4, 8 GPUs
        note("strips of the N-way splits rendered alone (prediction)")
        out["predicted_strong_scaling"] = predict_strong_scaling(whole, dev_frame, WIDTH, HEIGHT, (1, 2, 4, 8) if world == 1 else (1, world),
                                                                 max(args.steps // 2, 3), device)
        if not args.no_ibl:
            out["predicted_ibl_precompute_sharded"] = predict_ibl_sharded((1, 2, 4, 8) if world == 1 else (1, world), device)
    out.update(extra)
    if uses_classes:
        sr = time_shadow_ranges(dev_frame, renderer, compare=not args.no_ranges_compare)
        sr["shade_ms"] = round(kernel_ms["shade"], 4)  # light_classes_kernel + shade_kernel, as timed for the roofline record
        sr["note"] = ("build_ms is not part of a step: the ranges are built when the cubes change, as the cubes themselves are inputs of a step; "
                      "ms_per_step + build_ms = a frame whose shadow pass ran and whose ranges were rebuilt.  The per-tile light classes "
                      "are made from the G-buffer inside every step (light_classes_kernel, timed with the shading kernel)")
        sr["value_with_ranges_built_every_frame"] = round(px / ((ms_per_step + sr["build_ms"]) * 1e-3) / 1e6, 3)
        if "shade_ms_without_ranges" in sr:
            sr["value_without_ranges"] = round(px / ((ms_per_step + sr["shade_ms_without_ranges"] - sr["shade_ms"]) * 1e-3) / 1e6, 3)
            plain_tflops = flops / (sr["shade_ms_without_ranges"] * 1e-3) / 1e12
            roofline["without_ranges"] = {"avg_launch_ms": sr["shade_ms_without_ranges"], "achieved": round(plain_tflops, 3),
                                          "frac": round(plain_tflops / PEAK_FP32_TFLOPS, 4),
                                          "note": "the same kernel reading all 20 taps of every light of every pixel, as light.frag does: the canonical count describes this form"}
        out["shadow_ranges"] = sr
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
                                 "bloom_post_frac_of_measured": round(post[0]["achieved"] / copy.value, 4),
                                 "bloom_h_frac_of_measured": round(post[0]["replaces"]["kernels"][0]["achieved"] / copy.value, 4),
                                 "bloom_v_frac_of_measured": round(post[0]["replaces"]["kernels"][1]["achieved"] / copy.value, 4)}
        roofline["frac_of_measured"] = out["measured_peaks"]["shade_frac_of_measured"]
    if world == 1 and not args.no_coverage:
        out["light_volume_coverage"] = time_coverage(f, dev_frame, renderer, device)
        out["shadow_pass"] = time_shadow_pass(f, dev_frame, device)
        sp, cv = out["shadow_pass"], out["light_volume_coverage"]["reference_radii"]
        whole_ms = sp["point_shadow_ms"] + sp["dir_shadow_ms"] + sp["geometry_pass_ms"] + sp["skybox_ms"] + cv["coverage_ms"] + ms_per_step
        e2e = sp["from_draw_calls_end_to_end"]
        out["frame_from_meshes"] = {"ms": e2e["whole_frame_ms"], "Mpixels/s": round(px / (e2e["whole_frame_ms"] * 1e-3) / 1e6, 1),
                                    "sum_of_passes_ms": round(whole_ms, 3), "rows_of_one_rank_of_8_ms": e2e["rows_of_rank_3_of_8_ms"],
                                    "note": "shadow passes + geometry pass + skybox + light-volume coverage + composition + post chain, enqueued frame after "
                                            "frame and timed end to end: a frame in which every light moves and nothing but meshes, materials, lights and "
                                            "the camera is given; rows_of_one_rank_of_8: the same for the 270 rows one rank of eight owns (all shadow maps "
                                            "are rendered on every rank).  The G-buffer is the mesh's (geometry_pass_covered_fraction of the pixels carry "
                                            "geometry, the rest shows the skybox), so the composition pass has less to shade here than in the timed step, "
                                            "whose synthetic G-buffer is 90 % covered; sum_of_passes_ms adds the passes timed one by one to the timed step"}
    if ibl_sharded is not None:
        out["ibl_precompute_sharded"] = ibl_sharded
    if not args.no_ibl:
        out["ibl_precompute"], out["roofline_ibl"] = time_ibl(device)
    if world == 1 and args.cpu_rows > 0:
        out["cpu_baseline"] = cpu_baseline(f, P.f16_bits(renderer.color), P.f16_bits(renderer.threshold), args.cpu_rows)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
