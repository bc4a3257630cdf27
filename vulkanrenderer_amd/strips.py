"""Strip-parallel frame chain: one process per GPU, the frame cut into horizontal strips.

The reference renders on one device; nothing in it partitions a frame.  The shading path shards
naturally by rows: the composition pass and the post-process draw are per-pixel, bloom pass 1
(horizontal, REPEAT) is per-row, and bloom pass 2 (vertical, CLAMP_TO_EDGE) reads 4 rows above and
below each output row.  So each rank shades its strip, swaps 4 boundary rows of the thresholded
image (the source of bloom pass 1) with its neighbours (point-to-point over RCCL / xGMI, or gloo on
CPU), and finishes with the one kernel that runs both bloom passes and the tonemap: a rank blurs the
neighbours' four rows horizontally itself, which costs less than storing and re-reading a pass-1
image.  There is no reduction anywhere, so the assembled image is bit-identical to a single-GPU run.

Scene constants, lights, shadow maps and IBL images are replicated; only the G-buffer is split.
"""
from __future__ import annotations

import dataclasses

import torch
import torch.distributed as dist

from . import _lib

HALO = 4  # bloom_blur.frag:20-23 reaches 4 texels to each side


@dataclasses.dataclass(frozen=True)
class StripPlan:
    """Rows [row0, row0 + rows) of a `total_rows`-row frame belong to `rank`."""
    rank: int
    world: int
    total_rows: int
    row0: int
    rows: int

    @staticmethod
    def make(rank: int, world: int, total_rows: int) -> "StripPlan":
        base, extra = divmod(total_rows, world)  # the first `extra` strips are one row taller
        row0 = rank * base + min(rank, extra)
        rows = base + (1 if rank < extra else 0)
        if rows < HALO:
            raise ValueError(f"strip of rank {rank} has {rows} rows; every strip needs at least {HALO}")
        return StripPlan(rank, world, total_rows, row0, rows)

    @property
    def is_top(self) -> bool:
        return self.rank == 0

    @property
    def is_bottom(self) -> bool:
        return self.rank == self.world - 1


class HipBackend:
    """The product backend: libvkr_hip.so passes on torch CUDA tensors."""

    def __init__(self, device):
        from . import passes
        _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("the strip renderer needs a GPU: there is no CPU path")
        self.P = passes
        self.device = device
        self._side = None  # a second stream for the passes that run beside the point-shadow pass

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def shade(self, frame, color, thr, y0=0, y1=None):
        """Rows [y0, y1) of the strip (all of it by default); y0 is a multiple of 8 when masks are attached."""
        if y0 == 0 and (y1 is None or y1 == frame.rows):
            self.P.shade(frame, color, thr)
            return
        cut = lambda t: None if t is None else t[y0:y1]
        tiles_x = (frame.width + 7) // 8
        part = dataclasses.replace(frame, rows=y1 - y0, albedo=cut(frame.albedo), normal=cut(frame.normal), arm=cut(frame.arm),
                                   position=cut(frame.position), background=cut(frame.background),
                                   coverage=None if frame.coverage is None else frame.coverage[:, y0 // 8:(y1 + 7) // 8].contiguous())
        assert frame.coverage is None or (y0 % 8 == 0 and part.coverage.shape[-1] == tiles_x)
        self.P.shade(part, color[y0:y1], thr[y0:y1])

    def frame_from_draws(self, frame, draws, shadow_res, dir_res, frame_rows, row0, env=None):
        """This strip's G-buffer and the (replicated) shadow maps rendered from draw calls: `frame` supplies the
        scene constants, lights and IBL images; `draws` are the dicts passes.gbuffer takes.  Each rank renders
        all shadow maps itself: re-rendering 64 cubes (1.8 ms) is cheaper than moving 1.5 GB between GPUs.
        `env` = (environment cube tensor, VKR_ENV_LAYOUT_*): the skybox is drawn behind the strip's geometry."""
        import dataclasses
        P = self.P
        # The skybox, the geometry pass and the directional depth map go to a second stream: the point-shadow pass is long
        # and bound by binary64 arithmetic at half-empty waves, and these passes fit beside it.
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side
        side.wait_stream(main)
        sd = [(d["positions"], d["triangles"], d["model"]) for d in draws]
        with torch.cuda.stream(side):
            background = None
            if env is not None:
                background = P.skybox(env[0], frame.scene, frame.width, frame.rows, frame_rows, row0, env_layout=env[1])
            ga, gn, gm, gp = P.gbuffer(draws, frame.scene, frame.width, frame.rows, self.device, frame_rows, row0)
            dmap = P.dir_shadow_map(sd, frame.scene["light_space_matrix"], dir_res, self.device)
        cubes, layout = P.point_shadow_maps(sd, frame.lights, frame.n_lights, shadow_res)
        ranges = P.cube_shadow_ranges(cubes, layout) if frame.n_lights >= _lib.SHADOW_RANGES_MIN_LIGHTS else None
        main.wait_stream(side)
        for t in (background, ga, gn, gm, gp, dmap):  # made on the side stream, used and released on this one
            if t is not None:
                t.record_stream(main)
        out = dataclasses.replace(frame, albedo=ga, normal=gn, arm=gm, position=gp, cube_shadow=cubes, cube_layout=layout, dir_shadow=dmap,
                                  cube_ranges=ranges)
        return out if background is None else dataclasses.replace(out, background=background)

    def with_coverage(self, frame, volume_mesh, frame_rows, row0):
        """The strip's frame with light-volume masks attached; `volume_mesh` = (verts, triangles) device tensors."""
        import dataclasses
        verts, tris = volume_mesh
        masks, flags = self.P.light_coverage_with_flags(frame, verts, tris, frame_rows, row0)
        return dataclasses.replace(frame, coverage=masks, coverage_flags=flags)

    def bloom_post(self, thr, hdr, out, lo, hi, row0=0):
        # `thr` is the halo-padded thresholded image; output row k of `hdr`/`out` is strip row row0 + k, centred
        # on row HALO + row0 + k of `thr`; vertical taps clamp to rows [lo, hi] of `thr`
        self.P.bloom_postprocess(thr, hdr, out, 0, src_row_offset=HALO + row0, clamp_v=(lo, hi))


class StripRenderer:
    """Runs the per-frame chain on this rank's strip.  `backend` defaults to the HIP kernels."""

    def __init__(self, plan: StripPlan, width: int, device=None, backend=None, group=None, frames_in_flight: int = 1):
        """`frames_in_flight` > 1 (the reference keeps 2, VulkanRenderer.cs:60): that many sets of targets, each with a
        HIP stream of its own.  Consecutive frames then overlap on the GPU: while the last waves of frame k walk
        their lights and its RGBA8 strip travels to rank 0, the first waves of frame k+1 are already running.
        With one frame in flight everything runs on the caller's current stream."""
        self.plan, self.width, self.group = plan, width, group
        self.backend = backend if backend is not None else HipBackend(device)
        r = plan.rows
        self._slots = []
        for _ in range(max(frames_in_flight, 1)):
            # the thresholded image carries HALO rows of each neighbour above and below the strip's own rows
            padded = self.backend.empty((r + 2 * HALO, width, 4), torch.float16)
            self._slots.append(dict(color=self.backend.empty((r, width, 4), torch.float16), threshold_padded=padded,
                                    threshold=padded[HALO:HALO + r], ldr=self.backend.empty((r, width, 4), torch.uint8),
                                    stream=None, gathered=None))
        if self._slots[0]["ldr"].is_cuda and len(self._slots) > 1:
            for slot in self._slots:
                slot["stream"] = torch.cuda.Stream(device=slot["ldr"].device)
        self._slot = 0
        self._use(self._slots[0])
        self.marks = None  # a list: render() and gather() append (phase name, event recorded when the phase's work ends)

    def _use(self, slot) -> None:
        # the targets of the frame being rendered (or rendered last)
        self.color, self.threshold, self.threshold_padded, self.ldr = slot["color"], slot["threshold"], slot["threshold_padded"], slot["ldr"]

    def _on_slot_stream(self, slot):
        """Context in which the work of `slot`'s frame is enqueued: its own stream, ordered after what the caller's
        current stream holds so far (the frame's inputs), or the caller's stream when there is one frame in flight."""
        import contextlib
        if slot["stream"] is None:
            return contextlib.nullcontext()
        slot["stream"].wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(slot["stream"])

    def start_halo_exchange(self) -> list:
        """The first HALO rows of the strip's thresholded image go to the rank above, the last HALO to the rank
        below.  Returns the pending requests (empty on a single rank)."""
        p, r, t = self.plan, self.plan.rows, self.threshold_padded
        if p.world == 1:
            return []
        ops = []
        if not p.is_top:
            ops.append(dist.P2POp(dist.isend, t[HALO:2 * HALO], p.rank - 1, self.group))
            ops.append(dist.P2POp(dist.irecv, t[0:HALO], p.rank - 1, self.group))
        if not p.is_bottom:
            ops.append(dist.P2POp(dist.isend, t[r:r + HALO], p.rank + 1, self.group))
            ops.append(dist.P2POp(dist.irecv, t[r + HALO:r + 2 * HALO], p.rank + 1, self.group))
        return dist.batch_isend_irecv(ops)

    def exchange_halo(self) -> None:
        for req in self.start_halo_exchange():
            req.wait()

    def render_draws(self, frame, draws, shadow_res: int, dir_res: int, volume_mesh=None, env=None) -> torch.Tensor:
        """A frame from draw calls: this strip's part of the geometry pass (placed in the frame by plan.row0),
        the shadow passes, the skybox when an environment cube is given, then render().  `frame` carries scene
        constants, lights and IBL images."""
        p = self.plan
        if env is None:
            return self.render(self.backend.frame_from_draws(frame, draws, shadow_res, dir_res, p.total_rows, p.row0), volume_mesh)
        return self.render(self.backend.frame_from_draws(frame, draws, shadow_res, dir_res, p.total_rows, p.row0, env), volume_mesh)

    def _mark(self, name: str) -> None:
        if self.marks is not None and self.ldr.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            self.marks.append((name, ev))

    def render(self, frame, volume_mesh=None) -> torch.Tensor:
        """composition of the strip -> halo exchange of its boundary thresholded rows, overlapped with the bloom +
        post-process kernel over the rows that reach no halo row -> the rows next to a neighbour; returns the
        RGBA8 strip.  With several frames in flight the strip is written on the frame's own stream: synchronize()
        makes the caller's stream wait for it before the caller reads the tensor.  With `volume_mesh` each light's draw is limited to the pixels its volume covers; the strip's
        place in the frame (plan.row0 of plan.total_rows) enters the projection, nothing is exchanged for it.

        The strip is shaded in ONE launch: every wave walks all lights in order, so a launch fills and drains the
        machine for about one wave lifetime (4 us per light) whatever its size; launches that follow each other
        hide most of that, but each extra one still costs 0.05-0.1 ms at 64 lights -- more than the exchange
        (tens of microseconds) that shading the boundary rows first would hide."""
        slot = self._slots[self._slot]
        self._use(slot)
        if slot["stream"] is not None:
            # the frame's tensors were allocated on the caller's stream: tell the allocator that this stream reads them,
            # so that dropping the frame does not hand their memory to a new allocation while the kernels still run
            for v in vars(frame).values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(slot["stream"])
        with self._on_slot_stream(slot):
            return self._render(frame, volume_mesh)

    def _render(self, frame, volume_mesh):
        p, r = self.plan, self.plan.rows
        be = self.backend
        if volume_mesh is not None:
            frame = be.with_coverage(frame, volume_mesh, p.total_rows, p.row0)
        # frame edges clamp (CLAMP_TO_EDGE); interior strip edges read the neighbours' rows
        lo = HALO if p.is_top else 0
        hi = HALO + r - 1 if p.is_bottom else r + 2 * HALO - 1
        # rows [top, bottom) reach no halo row: they run while the halo rows are in flight
        top = 0 if p.is_top else HALO
        bottom = r if p.is_bottom else r - HALO
        if getattr(frame, "cube_ranges", None) is not None and frame.n_lights > 0:
            # the composition pass writes its per-tile light classes into a workspace: one per set of targets, so that
            # frames in flight on different streams do not share it
            slot = self._slots[self._slot]
            need = be.P.shade_workspace_bytes(frame.width, frame.rows, frame.n_lights)
            if slot.get("shade_workspace") is None or slot["shade_workspace"].numel() < need:
                slot["shade_workspace"] = be.empty((need,), torch.uint8)
            frame = dataclasses.replace(frame, shade_workspace=slot["shade_workspace"])
        self._mark("frame_start")
        be.shade(frame, self.color, self.threshold)
        self._mark("shade")
        pending = self.start_halo_exchange()
        if bottom > top:
            be.bloom_post(self.threshold_padded, self.color[top:bottom], self.ldr[top:bottom], lo, hi, row0=top)
        self._mark("interior_bloom_post")
        for req in pending:
            req.wait()
        self._mark("halo_wait")  # what of the exchange the interior work did not hide
        if top > 0:
            be.bloom_post(self.threshold_padded, self.color[:top], self.ldr[:top], lo, hi, row0=0)
        if bottom < r:
            be.bloom_post(self.threshold_padded, self.color[max(bottom, top):], self.ldr[max(bottom, top):], lo, hi, row0=max(bottom, top))
        self._mark("edge_bloom_post")
        return self.ldr

    def gather(self, dst: int = 0):
        """Collects the RGBA8 strips of the frame just rendered on rank `dst` (list of tensors there, None
        elsewhere).  With several frames in flight the transfer is enqueued on the frame's own stream and the next
        render() goes on at once on another; the returned tensors are complete once that stream (or the device) is
        synchronised, and stay untouched until as many further frames as there are frames in flight have been
        gathered (every set of targets has receive buffers of its own)."""
        slot = self._slots[self._slot]
        self._slot = (self._slot + 1) % len(self._slots)
        if self.plan.world == 1:
            return [slot["ldr"]]
        with self._on_slot_stream(slot):
            self._mark("gather_start")
            parts = self._gather_now(slot, dst)
            self._mark("gather")
        return parts

    def synchronize(self) -> None:
        """Makes the caller's current stream wait for every frame in flight."""
        for slot in self._slots:
            if slot["stream"] is not None:
                torch.cuda.current_stream().wait_stream(slot["stream"])

    def _gather_now(self, slot, dst: int):
        p, ldr = self.plan, slot["ldr"]
        if p.rank == dst:
            if slot["gathered"] is None:
                slot["gathered"] = [self.backend.empty((StripPlan.make(k, p.world, p.total_rows).rows, self.width, 4), torch.uint8)
                                    for k in range(p.world)]
            parts = list(slot["gathered"])
            ops = [dist.P2POp(dist.irecv, parts[k], k, self.group) for k in range(p.world) if k != dst]
            parts[dst] = ldr
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            return parts
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, ldr, dst, self.group)]):
            req.wait()
        return None
