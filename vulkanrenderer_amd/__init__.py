"""vulkanrenderer_amd: the screen-space shading path of VulkanRenderer (deferred composition with
shadowed lights, IBL precompute, bloom + ACES post chain) as hand-written gfx950 kernels behind a
C ABI (include/vkr_hip.h).

    _lib      ctypes binding of libvkr_hip.so (no CPU fallback: missing library = ImportError)
    renderer  HipRenderer, the host-side mirror of the reference's VulkanRenderer interface
    passes    the stateless passes on torch-owned device memory (used by the strip-parallel path)
    strips    one-process-per-GPU strip partition with RCCL halo exchange
    synth     seeded synthetic G-buffer / lights / shadow maps / environment images
"""
from . import _lib  # noqa: F401
from ._lib import VkrError  # noqa: F401
