"""IBL precompute split over GPUs (one process per GPU).

The reference runs the four precompute steps once, on one device, in its constructor
(VulkanRenderer.cs:271-275: CreateSkyboxCubemap, CreateIrradianceCubemap, CreatePrefilteredCubemap,
CreateBRDFLUTTexture).  Every output texel is independent, so the work splits by rows: the
environment cube is resampled redundantly on every rank (cheaper than moving 201 MB), then each
rank computes an equal slice of the rows of the irradiance cube, of every prefilter mip and of the
BRDF LUT, and the slices are all-gathered (every rank needs the full images for shading).  Each
mip is split on its own, so the per-rank cost is equal although mip 0 holds 75 % of the texels.
No arithmetic crosses a rank boundary: the gathered images are bit-identical to a one-GPU run.
"""
from __future__ import annotations

import dataclasses

import torch
import torch.distributed as dist

from . import _lib


@dataclasses.dataclass(frozen=True)
class IblConfig:
    env_res: int = 512          # VulkanRenderer.cs:653 (skybox cube)
    irradiance_res: int = 32    # VulkanRenderer.cs:734
    prefilter_res: int = 128    # VulkanRenderer.cs:815
    prefilter_mips: int = 5     # roughness = mip / (mips - 1), VulkanRenderer.cs:874
    lut_res: int = 512          # VulkanRenderer.cs:966
    # IRRADIANCE_FASTEST: by region of the environment with row sums, 1 binary16 ulp from the shader's own running sum
    # (IRRADIANCE_SEQUENTIAL, five times slower at 2048^2 faces) in under 1 % of the texels
    irradiance_order: int = _lib.IRRADIANCE_FASTEST


def row_share(total_rows: int, rank: int, world: int) -> tuple[int, int, int]:
    """(row_begin, row_end, rows_per_rank): equal chunks of ceil(total/world) rows, the last ones short or empty."""
    per = -(-total_rows // world)
    return min(rank * per, total_rows), min((rank + 1) * per, total_rows), per


class HipIblBackend:
    """The product backend: libvkr_hip.so kernels on torch CUDA tensors."""

    def __init__(self, device):
        from . import passes
        _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("the IBL precompute needs a GPU: there is no CPU path")
        self.P = passes
        self.device = device

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def equirect(self, src, src_format, env):
        self.P.equirect_to_cube(src, src_format, env)

    def sampling_copy(self, env):
        """The copy of the environment the irradiance and prefilter passes read, as (tensor, layout, face size): footprint
        tiles (every 2x2 footprint of a LINEAR lookup in one cache line)."""
        n = env.shape[1]
        if n <= 8192:
            return self.P.tile_env_footprints(env), _lib.ENV_LAYOUT_FOOTPRINT_TILES, n
        return env, _lib.ENV_LAYOUT_LINEAR, n

    def irradiance(self, sampled, out, rows, order=_lib.IRRADIANCE_FASTEST, linear=None):
        """`sampled` = sampling_copy(); `linear` the row-major cube, which the orders other than the default one read.
        The by-region orders raise if a texel did not find all of its samples."""
        tiles, layout, env_n = sampled
        if order in (_lib.IRRADIANCE_FASTEST, _lib.IRRADIANCE_BY_REGION_ROWS) and layout == _lib.ENV_LAYOUT_FOOTPRINT_TILES:
            ws = self.P.irradiance_by_region_rows(tiles, env_n, out, rows)
            self.P._require_all_samples(ws, "VKR_IRRADIANCE_BY_REGION_ROWS")
        else:
            self.P.irradiance(linear, out, order, rows=rows, env_layout=_lib.ENV_LAYOUT_LINEAR)

    def prefilter_mips(self, sampled, cubes, rows):
        tiles, layout, env_n = sampled
        self.P.prefilter_mips(tiles, cubes, rows, env_layout=layout, env_n=env_n)

    def brdf_lut(self, out, rows):
        self.P.brdf_lut(out, rows=rows)


class IblPrecompute:
    """Computes this rank's rows of every IBL image and all-gathers them."""

    def __init__(self, cfg: IblConfig, rank: int = 0, world: int = 1, device=None, backend=None, group=None):
        self.cfg, self.rank, self.world, self.group = cfg, rank, world, group
        self.backend = backend if backend is not None else HipIblBackend(device)

    def _padded_cube(self, n: int):
        """A cube image whose 6n rows are padded to a multiple of the world size: (flat rows view, cube view)."""
        _, _, per = row_share(6 * n, self.rank, self.world)
        flat = self.backend.empty((per * self.world, n, 4), torch.float16)
        return flat, flat[:6 * n].view(6, n, n, 4)

    def _all_gather_rows(self, flat: torch.Tensor, per: int) -> None:
        if self.world == 1:
            return
        mine = flat[self.rank * per:(self.rank + 1) * per]
        if flat.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal on a box with fewer GPUs than ranks: gloo's own staging of device tensors stalls now and then
            # when ranks share a GPU, so the rows go through host memory here
            host = flat.cpu()
            dist.all_gather_into_tensor(host, host[self.rank * per:(self.rank + 1) * per].clone(), group=self.group)
            flat.copy_(host)
            return
        dist.all_gather_into_tensor(flat, mine, group=self.group)

    def run(self, src: torch.Tensor, src_format: int) -> dict:
        cfg, be = self.cfg, self.backend
        env = be.empty((6, cfg.env_res, cfg.env_res, 4), torch.float16)
        be.equirect(src, src_format, env)  # replicated
        sampled = be.sampling_copy(env)

        n = cfg.irradiance_res
        irr_flat, irr = self._padded_cube(n)
        r0, r1, per = row_share(6 * n, self.rank, self.world)
        be.irradiance(sampled, irr, (r0, r1), order=cfg.irradiance_order, linear=env)
        self._all_gather_rows(irr_flat, per)

        # every mip is split on its own; the rank's shares of all mips go to the device together
        flats, mips, shares = [], [], []
        for m in range(cfg.prefilter_mips):
            n = cfg.prefilter_res >> m
            flat, cube = self._padded_cube(n)
            flats.append(flat); mips.append(cube); shares.append(row_share(6 * n, self.rank, self.world))
        be.prefilter_mips(sampled, mips, [(r0, r1) for r0, r1, _ in shares])
        for flat, (_, _, per) in zip(flats, shares):
            self._all_gather_rows(flat, per)

        n = cfg.lut_res
        r0, r1, per = row_share(n, self.rank, self.world)
        lut_flat = be.empty((per * self.world, n, 2), torch.float16)
        lut = lut_flat[:n]
        be.brdf_lut(lut, (r0, r1))
        self._all_gather_rows(lut_flat, per)
        return {"env": env, "irradiance": irr, "prefilter": mips, "brdf_lut": lut}
