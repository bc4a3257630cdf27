/*
 * vkr_hip.h -- C ABI of libvkr_hip.so: VulkanRenderer's screen-space shading path on MI355X.
 *
 * The reference (a C# / Vulkan renderer) has no native seam of its own for this path: its C#
 * pass objects call straight into libvulkan through Silk.NET.  The seam bound here is the one
 * its pass interface implies -- one entry point per render pass the reference records, taking
 * the same parameter blocks (SceneInfo / LightInfo, byte for byte) and the same images
 * (tight row-major RGBA16F / D32F).  Each declaration cites the reference code it replaces.
 *
 * Two layers:
 *   layer 1  vkr_shade / vkr_bloom_blur / ...   stateless passes over caller-owned DEVICE memory,
 *            launched on a caller-supplied HIP stream (NULL = the default stream); asynchronous.
 *   layer 2  vkr_create / vkr_end_frame / ...   a renderer object that owns its device images and
 *            takes HOST pointers; this is what a P/Invoke binding calls.
 *
 * Conventions: every function returns 0 (VKR_OK) or a negative VKR_ERR_* code and records a
 * message retrievable with vkr_last_error() (thread-local).  No function is thread-safe with
 * respect to the same vkr_ctx; different contexts may be used from different threads.
 * Images: RGBA16F = 4 x uint16 (binary16) per texel, RG16F = 2 x uint16, D32F = float,
 * RGBA8 = 4 x uint8; rows are tightly packed, row 0 is the top row of the frame.
 * Cubes: 6 faces in the order +X,-X,+Y,-Y,+Z,-Z, each face row-major.
 */
#ifndef VKR_HIP_H
#define VKR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define VKR_EXPORT __attribute__((visibility("default")))
#else
#define VKR_EXPORT
#endif

#define VKR_OK 0
#define VKR_ERR_INVALID (-1)     /* bad argument (null pointer, size <= 0, unknown enum) */
#define VKR_ERR_HIP (-2)         /* a HIP runtime call failed; see vkr_last_error() */
#define VKR_ERR_UNSUPPORTED (-3) /* valid request this build does not implement */
#define VKR_ERR_ORDER (-4)       /* BeginFrame / EndFrame called out of order */
#define VKR_ERR_NO_DEVICE (-5)   /* no gfx950 device visible */

/* VulkanRenderer.cs:11-20 -- SceneInfo, explicit layout, 240 bytes.  Matrices are the 16 floats
 * of Silk.NET's row-major Matrix4X4<float> in memory order. */
typedef struct VkrSceneInfo {
    float camera_view[16];        /* offset   0 */
    float camera_projection[16];  /* offset  64 */
    float camera_position[3];     /* offset 128 */
    float _pad0;
    float dir_light_direction[3]; /* offset 144 */
    float _pad1;
    float dir_light_color[3];     /* offset 160 */
    float _pad2;
    float light_space_matrix[16]; /* offset 176 */
} VkrSceneInfo;

/* VulkanRenderer.cs:22-28 -- LightInfo, explicit layout, 92 bytes. */
typedef struct VkrLightInfo {
    float model[16];   /* offset  0: Scale(volume radius) * Translate(position), Light.cs:14 */
    float position[3]; /* offset 64 */
    float _pad0;
    float color[3];    /* offset 80 */
} VkrLightInfo;

/* Which pixels a point light's draw touches (the reference rasterises a sphere mesh per light,
 * GraphicsPipeline.cs:100-117).  Both modes are implemented. */
#define VKR_COVERAGE_FULLSCREEN 0 /* every light volume covers every pixel (exact while the camera is inside every volume) */
#define VKR_COVERAGE_MESH 1       /* per-pixel coverage of the light-volume mesh, from vkr_light_coverage() */

/* Memory layout of the D32F point-shadow cubes (the reference keeps them in OPTIMAL-tiled Vulkan
 * images, VulkanRenderer.cs:1348-1370, whose layout is the driver's; here it is explicit). */
#define VKR_CUBE_LAYOUT_LINEAR 0    /* [light][face][row][column], tight */
#define VKR_CUBE_LAYOUT_TILED_8X4 1 /* each face as 8x4-texel tiles of 128 bytes (rows of 8 texels inside a tile), the
                                       tiles stored column by column: texel (i, j) of a face of size n is float number
                                       (i / 8) * 8n + 8j + i % 8 of the face; face size % 8 == 0; made from the linear
                                       layout by vkr_tile_depth_cubes() */
#define VKR_CUBE_LAYOUT_TILED_8X4_X25 2 /* the same tiles holding RN(25 * depth), the product light.frag:133 forms for
                                           every tap (25 = the shadow passes' far plane, VulkanRenderer.cs:331): the
                                           shading kernel then compares without multiplying; made by
                                           vkr_relayout_depth_cubes() or written directly by vkr_point_shadow_maps() */

/* Memory layout of the RGBA16F environment cube read by the irradiance and prefilter passes. */
#define VKR_ENV_LAYOUT_LINEAR 0    /* [face][row][column], tight */
#define VKR_ENV_LAYOUT_TILED_4X4 1 /* each face as row-major 4x4-texel tiles of 128 bytes; face size % 4 == 0;
                                      made from the linear layout by vkr_tile_rgba16f_faces() */
#define VKR_ENV_LAYOUT_FOOTPRINT_TILES 2
/* The third layout, read by vkr_ibl_irradiance_by_region_rows only and made by vkr_tile_rgba16f_footprints(): "footprint
 * tiles" -- 4x4-texel tiles of 128 bytes at a stride of THREE texels in both directions over each face extended by a
 * one-texel apron (the neighbouring faces' texels, as seamless cube filtering folds them), (n / 3 + 1)^2 tiles per face,
 * row-major by tile: tile (k, m) holds the texels (3k-1 .. 3k+2) x (3m-1 .. 3m+2).  The 2x2 footprint of any LINEAR
 * lookup then lies inside one tile (one cache line), at the price of 16/9 of the memory. */

#define VKR_WRAP_REPEAT 0 /* textureSampler, VulkanHelper.cs:123-154 */
#define VKR_WRAP_CLAMP 1  /* bloomSampler,   VulkanHelper.cs:156-187 */

#define VKR_SRC_RGBA8_SRGB 0 /* Texture.cs:21 (the reference's 8-bit skybox image) */
#define VKR_SRC_RGBA16F 1

#define VKR_POST_SRGB_TARGET 1 /* store through an _SRGB target format (SCDevice.cs:655-666) */

#define VKR_IRRADIANCE_SEQUENTIAL 0 /* one running sum, the shader's own order */
#define VKR_IRRADIANCE_PHI_SLICES 1 /* every phi slice summed from zero, slice sums added in phi order */
#define VKR_IRRADIANCE_BY_REGION 2  /* samples grouped by the 512x512-texel region of the environment their lookup
                                       starts in (regions numbered (row * per_side + column) * 6 + face); within a
                                       region the k-th sample in the shader's order goes to accumulator k mod 64, the
                                       64 accumulators are combined pairwise (partner = index xor 32, 16, .. 1) and the
                                       region sums added in region order.  Nearly three times as fast at 2048^2 faces (the
                                       lookups hit the cache); differs from the shader's order in rounding only */
#define VKR_IRRADIANCE_BY_REGION_ROWS 3 /* samples grouped by region again, the regions cut for the footprint-tile layout
                                       (the base texel of a lookup's 2x2 footprint, shifted by one to lie in [0, n]^2, in
                                       cells of 3x3 positions, C = ceil(cells / ceil(cells / 128)) cells per region side
                                       with cells = n / 3 + 1, regions numbered (row * per_side + column) * 6 + face); within
                                       a region every phi row of the shader's loop is summed on its own (theta ascending),
                                       the row sums are added in phi order and the region sums in region order.  The
                                       fastest order; differs from the shader's order in rounding only */
#define VKR_IRRADIANCE_FASTEST (-1) /* layer 2 only: the library's choice, currently VKR_IRRADIANCE_BY_REGION_ROWS */

/* Version of this interface: it changes whenever a struct below gains, loses or re-orders a field, an argument
 * list changes, or a layout constant changes its meaning -- anything that makes a host built against an older
 * header pass memory the library would misread.  A host compares vkr_abi_version() (what the loaded library was
 * built with) against the VKR_ABI_VERSION it was compiled with and refuses to go on when they differ.
 * History: 1 the first interface; 2 VkrShadeArgs.coverage_flags, VkrCoverageArgs.light_flags,
 * VkrConfig.frame_rows / row0, VKR_CUBE_LAYOUT_TILED_8X4 stores its tiles column by column;
 * 3 VKR_IRRADIANCE_BY_REGION_ROWS and the layer-2 default order, the workspace word of the by-region call. */
#define VKR_ABI_VERSION 4
VKR_EXPORT const char *vkr_last_error(void);
VKR_EXPORT int32_t vkr_abi_version(void);
/* Number of gfx950 devices visible to the process (0 if none); never fails. */
VKR_EXPORT int32_t vkr_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Layer 1: passes over device memory
 * ------------------------------------------------------------------------------------------ */

/* Replaces the composition render pass: the fullscreen composition.frag draw
 * (VulkanRenderer.cs:445-459) followed by one additive light.frag draw per light
 * (VulkanRenderer.cs:479-498), over a strip of `rows` rows.  Strips are independent. */
typedef struct VkrShadeArgs {
    int32_t width, rows;
    const void *albedo, *normal, *arm, *position; /* RGBA16F, rows*width texels each */
    const void *background;                       /* RGBA16F or NULL (cleared to 0,0,0,1) */
    VkrSceneInfo scene;
    const VkrLightInfo *lights;                   /* device memory, n_lights entries */
    int32_t n_lights;
    int32_t coverage_mode;                        /* VKR_COVERAGE_* */
    const uint64_t *coverage_masks;               /* VKR_COVERAGE_MESH: masks written by vkr_light_coverage() for
                                                     this strip (n_lights * ceil(rows/8) * ceil(width/8)) */
    const float *dir_shadow;  int32_t dir_res;    /* D32F dir_res^2 */
    const float *cube_shadow; int32_t cube_res;   /* D32F n_lights * 6 * cube_res^2 */
    int32_t cube_layout;                          /* VKR_CUBE_LAYOUT_* of cube_shadow */
    const void *irradiance;   int32_t irr_res;    /* RGBA16F cube */
    const void *prefilter;    int32_t pref_res;   /* RGBA16F cube (mip 0) */
    const void *brdf_lut;     int32_t lut_res;    /* RG16F */
    void *out_color, *out_threshold;              /* RGBA16F, rows*width texels each */
    const uint32_t *coverage_flags;               /* VKR_COVERAGE_MESH: VkrCoverageArgs.light_flags of the call that wrote
                                                     coverage_masks, or NULL; when it says that every light covers every
                                                     pixel the masks are not read */
    const void *cube_ranges;                      /* depth ranges of cube_shadow written by vkr_cube_shadow_ranges(), or NULL */
    void *workspace;                              /* device memory of vkr_shade_workspace_bytes() bytes, or NULL.  With both
                                                     cube_ranges and workspace the pass first sorts the lights of every 8x8
                                                     tile by what the ranges prove for all of the tile's pixels (all 20 taps
                                                     in shadow: the light is skipped; none: taken as lit; else: taps read),
                                                     written to workspace.  Same output either way, bit for bit. */
    int64_t workspace_bytes;
} VkrShadeArgs;
VKR_EXPORT int64_t vkr_shade_workspace_bytes(int32_t width, int32_t rows, int32_t n_lights);
VKR_EXPORT int32_t vkr_shade(const VkrShadeArgs *args, void *stream);

/* Depth ranges of the cube shadow maps, an acceleration structure for vkr_shade with no counterpart in the reference
 * (light.frag:131-137 reads its 20 taps per light unconditionally): the smallest and largest value the taps compare
 * against over half-overlapping blocks of each face, at four block sizes.  Where one range shows that all 20 taps of
 * every pixel of an 8x8 tile are in shadow, or none, vkr_shade takes that outcome without reading the taps; the image is
 * the same bit for bit.  `ranges` is device memory of vkr_cube_shadow_ranges_bytes(n_lights, res) bytes; rebuild it
 * whenever `cubes` changes.  `layout` is the VKR_CUBE_LAYOUT_* of `cubes`. */
VKR_EXPORT int64_t vkr_cube_shadow_ranges_bytes(int32_t n_lights, int32_t res);
VKR_EXPORT int32_t vkr_cube_shadow_ranges(const float *cubes, int32_t n_lights, int32_t res, int32_t layout, void *ranges, void *stream);

/* Which pixels the light.frag draw of each light touches: the light-volume mesh (the reference's
 * unit sphere, assets/models/sphere/sphere.glb, or any closed mesh) is placed by each light's model
 * matrix (Light.cs:10-26), transformed as light.vert:26 does, and its back faces are rasterised
 * (FRONT faces culled, CCW front, no depth test: GraphicsPipeline.cs:100-117; depth clip 0 <= z <= w).
 * Output: one 64-bit mask per (light, 8x8-pixel tile of the strip), bit (y&7)*8 + (x&7), tiles
 * row-major, lights outermost -- the form VkrShadeArgs.coverage_masks takes. */
typedef struct VkrCoverageArgs {
    int32_t width, rows;          /* the image: a strip of the frame */
    int32_t frame_rows, row0;     /* rows of the whole frame; the strip's first row in it */
    VkrSceneInfo scene;           /* camera_view and camera_projection are read */
    const VkrLightInfo *lights;   /* device memory; model matrices are read */
    int32_t n_lights;
    const float *verts;           /* device memory, 3 floats per vertex, at most 1024 vertices */
    int32_t n_verts;
    const uint32_t *indices;      /* device memory, 3 per triangle, counter-clockwise seen from outside */
    int32_t n_tris;
    uint64_t *masks;              /* device memory, n_lights * ceil(rows/8) * ceil(width/8) */
    uint32_t *light_flags;        /* device memory, n_lights + 1 words, or NULL.  Word i becomes 1 when light i's volume
                                     provably covers every pixel (the camera sees every triangle of the closed mesh from
                                     behind and every triangle a pixel ray can hit lies between the near and the far
                                     plane): such a light's masks are written as all ones without rasterising anything.
                                     Word n_lights becomes the number of lights whose word is 0.  Pass it for a CLOSED mesh
                                     only (every edge shared by two triangles of opposite direction): the conclusion rests
                                     on back faces alone, and an open volume seen from inside would be taken to cover the
                                     frame.  Layer 2 checks the mesh in vkr_set_light_volume_mesh and passes NULL otherwise. */
} VkrCoverageArgs;
VKR_EXPORT int32_t vkr_light_coverage(const VkrCoverageArgs *args, void *stream);

/* Shadow-map generation: the passes that produce the depth images vkr_shade reads.
 * One draw of the reference's solidModelDrawCalls list (VulkanRenderer.cs:357-366): a mesh and its model matrix. */
typedef struct VkrMeshDraw {
    const float *positions;    /* device memory, 3 floats per vertex */
    int32_t n_verts;
    const uint32_t *indices;   /* device memory, 3 per triangle, counter-clockwise seen from outside */
    int32_t n_tris;
    float model[16];           /* DrawCall.ModelMatrix, row-major bytes as the reference pushes them */
} VkrMeshDraw;

/* Replaces the point-shadow render passes (VulkanRenderer.cs:325-376; point_shadow.vert/.geom/.frag):
 * for every light a depth cube holding length(fragment - light) / far_plane of the nearest surface the
 * FRONT-culled, LESS-tested draws leave, cleared to 1.0.  Shadow matrices as VulkanRenderer.cs:327-339
 * (90 degree perspective, near plane 1).  `draws` is a host array; its meshes and `lights` are device memory. */
typedef struct VkrPointShadowArgs {
    const VkrMeshDraw *draws; int32_t n_draws;
    const VkrLightInfo *lights; int32_t n_lights;   /* positions are read */
    int32_t res;                /* face size (PointShadowMapResolution, 256 in the reference) */
    float far_plane;            /* 25 in the reference (VulkanRenderer.cs:331) */
    int32_t layout;             /* VKR_CUBE_LAYOUT_* of `out` */
    float *out;                 /* device memory, n_lights * 6 * res^2 */
} VkrPointShadowArgs;
VKR_EXPORT int32_t vkr_point_shadow_maps(const VkrPointShadowArgs *args, void *stream);

/* Replaces the directional depth pass (VulkanRenderer.cs:406-428; depth.vert): clip = lightSpaceMatrix * model * pos,
 * the same raster state, the fixed-function depth z/w, row-major res^2 D32F cleared to 1.0. */
typedef struct VkrDirShadowArgs {
    const VkrMeshDraw *draws; int32_t n_draws;
    float light_space[16];      /* SceneInfo.lightSpaceMatrix */
    int32_t res;
    float *out;
} VkrDirShadowArgs;
VKR_EXPORT int32_t vkr_dir_shadow_map(const VkrDirShadowArgs *args, void *stream);

/* Geometry pass: replaces the gpass.vert / gpass.frag draws of VulkanRenderer.cs:378-404 (BACK faces culled,
 * depth test LESS), producing the four RGBA16F G-buffer images vkr_shade reads.  Material images are
 * single-level RGBA8 (albedo and ao/roughness/metalness _SRGB, normal map _UNORM: Texture.cs:21,
 * Material.cs:20-22), sampled LINEAR / REPEAT. */
typedef struct VkrTexture {
    const void *rgba8;         /* device memory, height * width * 4 bytes */
    int32_t width, height;
} VkrTexture;
typedef struct VkrGpassDraw {
    const float *positions, *texcoords, *normals, *tangents;  /* device memory: 3, 2, 3, 3 floats per vertex (Mesh.cs vertex layout) */
    int32_t n_verts;
    const uint32_t *indices;   /* device memory, 3 per triangle, counter-clockwise seen from outside */
    int32_t n_tris;
    float model[16];           /* DrawCall.ModelMatrix */
    VkrTexture albedo, normal, arm;  /* Model.Material */
} VkrGpassDraw;
typedef struct VkrGpassArgs {
    const VkrGpassDraw *draws; int32_t n_draws;  /* host array, in submission order; copied to the device by an
                                                    asynchronous copy on `stream`: pageable memory may be reused when the
                                                    call returns, pinned memory must stay valid until the stream gets there */
    VkrSceneInfo scene;                          /* camera_view and camera_projection are read */
    int32_t width, rows;                         /* the image: a strip of the frame */
    int32_t frame_rows, row0;
    void *albedo, *normal, *arm, *position;      /* device memory, RGBA16F rows*width texels each */
    void *workspace; int64_t workspace_bytes;    /* device memory, at least vkr_gbuffer_workspace_bytes() */
} VkrGpassArgs;
VKR_EXPORT int64_t vkr_gbuffer_workspace_bytes(int32_t width, int32_t rows, int32_t n_draws);
/* The size of a workspace that also holds one 256-byte record per triangle of the call's draws (the vertex shader's
 * outputs at its corners and its edge functions).  Given at least this much, vkr_gbuffer shades every triangle's
 * corners once and the resolve step reads the winner's record, where it otherwise re-runs gpass.vert three times per
 * pixel: the same bits, the resolve step 2x faster.  No counterpart in the reference (its vertex shader runs per vertex
 * in the pipeline).  -1 for arguments out of range. */
VKR_EXPORT int64_t vkr_gbuffer_workspace_bytes_with_triangles(int32_t width, int32_t rows, int32_t n_draws, int64_t n_triangles);
VKR_EXPORT int32_t vkr_gbuffer(const VkrGpassArgs *args, void *stream);

/* vkr_gbuffer for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc).  It has no counterpart in the reference.  It is not meant for recording into a HIP graph: the draw
 * records are copied from the host array with an asynchronous copy and the launches are sized from host-side
 * triangle counts, so a replayed graph would draw the records of the call that was captured.  The visibility keys and
 * the draw records live in a fixed array inside the library's code object instead of a workspace
 * (args->workspace and args->workspace_bytes are ignored), so the call allocates and frees nothing; each GPU
 * the library is loaded on holds VKR_GBUFFER_STATIC_MAX_PIXELS * 8 bytes + 16 KiB for it until the process
 * ends.  Results equal vkr_gbuffer's bit for bit.  Limits: width * rows <= VKR_GBUFFER_STATIC_MAX_PIXELS and
 * n_draws <= VKR_GBUFFER_STATIC_MAX_DRAWS, otherwise VKR_ERR_INVALID.  All calls on one GPU share the array:
 * calls that overlap in time on one GPU (two streams, two threads) race; calls on one stream do not. */
#define VKR_GBUFFER_STATIC_MAX_PIXELS 2097152 /* 1920 x 1080 fits */
#define VKR_GBUFFER_STATIC_MAX_DRAWS 64
VKR_EXPORT int32_t vkr_gbuffer_static(const VkrGpassArgs *args, void *stream);

/* Replaces the skybox draw that opens the composition pass (VulkanRenderer.cs:436-443; cubemap.vert, skybox.frag):
 * colour = texture(skybox cube, view ray of the pixel), alpha 1.  The result is the `background` of VkrShadeArgs,
 * kept where the composition draw discards.  Assumes the reference's symmetric perspective (Camera.cs:30-35). */
VKR_EXPORT int32_t vkr_skybox(const void *env, int32_t env_n, int32_t env_layout, const VkrSceneInfo *scene,
                              int32_t width, int32_t rows, int32_t frame_rows, int32_t row0, void *out_rgba16f, void *stream);

/* Re-lays `n_faces` D32F faces of `res`^2 texels from VKR_CUBE_LAYOUT_LINEAR to
 * VKR_CUBE_LAYOUT_TILED_8X4 (res % 8 == 0; src and dst must not overlap).  No counterpart in the
 * reference, whose depth images are written tiled by the rasteriser. */
VKR_EXPORT int32_t vkr_tile_depth_cubes(const float *linear, float *tiled, int32_t n_faces, int32_t res, void *stream);
/* The same for either tiled layout (VKR_CUBE_LAYOUT_TILED_8X4 or VKR_CUBE_LAYOUT_TILED_8X4_X25). */
VKR_EXPORT int32_t vkr_relayout_depth_cubes(const float *linear, float *out, int32_t n_faces, int32_t res, int32_t layout, void *stream);

/* Replaces one bloom_blur.frag draw (VulkanRenderer.cs:504-540).  Writes `rows` output rows;
 * output row r is centred on source row r + src_row_offset.  Taps along the blur axis wrap
 * (REPEAT) or clamp to [clamp_lo, clamp_hi] (source row / column indices) -- a strip with halo
 * rows passes the valid range of its source buffer. */
typedef struct VkrBloomArgs {
    const void *src; void *dst;  /* RGBA16F */
    int32_t width, rows;         /* output size */
    int32_t src_rows;            /* rows in src (== rows unless src carries halo rows) */
    int32_t src_row_offset;
    int32_t horizontal;          /* non-zero: blur along x (bloom pass 1) */
    int32_t wrap;                /* VKR_WRAP_* along the blur axis */
    int32_t clamp_lo, clamp_hi;  /* used when wrap == VKR_WRAP_CLAMP */
} VkrBloomArgs;
VKR_EXPORT int32_t vkr_bloom_blur(const VkrBloomArgs *args, void *stream);

/* Replaces the postprocess.frag draw (VulkanRenderer.cs:544-557): hdr + bloom -> ACES ->
 * pow(1/2.2) -> RGBA8. */
VKR_EXPORT int32_t vkr_postprocess(const void *hdr, const void *bloom, void *out_rgba8, int32_t width, int32_t rows,
                        int32_t flags, void *stream);

/* Bloom pass 2 and the post-process draw in one kernel (same results as vkr_bloom_blur with
 * horizontal = 0 followed by vkr_postprocess; the intermediate bloom-2 image is never stored).
 * `bloom1` is the pass-1 output, addressed like VkrBloomArgs.src. */
typedef struct VkrBloomPostArgs {
    const void *bloom1; const void *hdr; void *out_rgba8;
    int32_t width, rows;
    int32_t src_rows, src_row_offset;
    int32_t wrap, clamp_lo, clamp_hi;
    int32_t flags;
} VkrBloomPostArgs;
VKR_EXPORT int32_t vkr_bloomv_postprocess(const VkrBloomPostArgs *args, void *stream);

/* Both bloom_blur.frag draws and the postprocess.frag draw (VulkanRenderer.cs:504-557) in one kernel: same
 * results as vkr_bloom_blur (horizontal, then vertical) followed by vkr_postprocess, bit for bit, but neither
 * bloom image is stored (20 bytes per pixel move instead of 36).  `threshold` is the source of bloom pass 1
 * (the composition pass's second attachment), addressed like VkrBloomArgs.src: output row r is centred on
 * source row r + src_row_offset, so a strip passes its threshold rows plus four halo rows on each side.
 * Pass 1 wraps along x by wrap_h (the reference: VKR_WRAP_REPEAT), pass 2 along y by wrap_v over source rows
 * (the reference: VKR_WRAP_CLAMP with the valid row range of `threshold`). */
typedef struct VkrBloomChainArgs {
    const void *threshold; const void *hdr; void *out_rgba8;
    int32_t width, rows;
    int32_t src_rows, src_row_offset;
    int32_t wrap_h, clamp_lo_h, clamp_hi_h;
    int32_t wrap_v, clamp_lo_v, clamp_hi_v;
    int32_t flags;                              /* VKR_POST_* */
} VkrBloomChainArgs;
VKR_EXPORT int32_t vkr_bloom_postprocess(const VkrBloomChainArgs *args, void *stream);

/* IBL precompute.  Cube images are addressed by "face rows": row r of a 6*n-row image is row
 * r % n of face r / n; [row_begin, row_end) selects the part this call computes, so the faces
 * can be split over GPUs. */

/* Replaces CreateSkyboxCubemap's six draws (VulkanRenderer.cs:653-732). */
VKR_EXPORT int32_t vkr_ibl_equirect_to_cube(const void *src, int32_t src_w, int32_t src_h, int32_t src_format,
                                 void *cube, int32_t n, int32_t row_begin, int32_t row_end, void *stream);
/* Re-lays `n_faces` RGBA16F faces of `n`^2 texels from VKR_ENV_LAYOUT_LINEAR to VKR_ENV_LAYOUT_TILED_4X4
 * (n % 4 == 0; src and dst must not overlap).  No counterpart in the reference, whose images are
 * OPTIMAL-tiled by the driver. */
VKR_EXPORT int32_t vkr_tile_rgba16f_faces(const void *linear, void *tiled, int32_t n_faces, int32_t n, void *stream);
/* Replaces CreateIrradianceCubemap's six draws (VulkanRenderer.cs:734-813). */
VKR_EXPORT int32_t vkr_ibl_irradiance(const void *env, int32_t env_n, int32_t env_layout, void *out, int32_t n, int32_t order,
                           int32_t row_begin, int32_t row_end, void *stream);
/* The same six draws summed in the VKR_IRRADIANCE_BY_REGION order (environment faces up to 8192^2, irradiance faces up to
 * 4096^2).  `workspace` is device memory of at least
 * vkr_ibl_irradiance_by_region_workspace_bytes(env_n, n, row_begin, row_end) bytes (-1: bad sizes), 4-byte aligned,
 * used until the work on `stream` has finished; the call itself allocates nothing.  The 32-bit word 256 bytes
 * before the end of that size then holds the number of texels whose 15 876 samples were not all found: 0, unless
 * the kernel's culling has a flaw; vkr_create_irradiance_cubemap checks it. */
VKR_EXPORT int64_t vkr_ibl_irradiance_by_region_workspace_bytes(int32_t env_n, int32_t n, int32_t row_begin, int32_t row_end);
VKR_EXPORT int32_t vkr_ibl_irradiance_by_region(const void *env, int32_t env_n, int32_t env_layout, void *out, int32_t n,
                           int32_t row_begin, int32_t row_end, void *workspace, int64_t workspace_bytes, void *stream);
/* The same six draws summed in the VKR_IRRADIANCE_BY_REGION_ROWS order, from the environment in footprint tiles
 * (vkr_tile_rgba16f_footprints of the whole cube into vkr_env_footprint_tiles_bytes(env_n) bytes, 8-byte aligned; -1: bad
 * size).  Sizes, workspace and its check word as for vkr_ibl_irradiance_by_region; a texel whose samples were not all
 * found is also written as NaN (by both calls).  Neither call has a counterpart in the reference beyond the pass itself. */
VKR_EXPORT int64_t vkr_env_footprint_tiles_bytes(int32_t n);
VKR_EXPORT int32_t vkr_tile_rgba16f_footprints(const void *linear_cube, void *tiles, int32_t n, void *stream);
/* All roughness mips of the prefiltered cube in as few launches as serve them best (mips too small to fill the chip
 * alone share one): VulkanRenderer.cs:855-884, roughness of mip m = m / (mips - 1) (:874).  out[m]: the RGBA16F cube of
 * mip m, face size base_res >> m; rows [row_begin[m], row_end[m]) of its 6 * (base_res >> m) rows are computed.  The
 * images equal those of one vkr_ibl_prefilter call per mip.  env_layout: any VKR_ENV_LAYOUT_* (footprint tiles: `env`
 * is the buffer vkr_tile_rgba16f_footprints filled, env_res the face size of the cube it was made from). */
#define VKR_PREFILTER_MAX_MIPS 8
typedef struct VkrPrefilterMipsArgs {
    const void *env; int32_t env_res, env_layout;
    int32_t base_res, mips;
    void *out[VKR_PREFILTER_MAX_MIPS];
    int32_t row_begin[VKR_PREFILTER_MAX_MIPS], row_end[VKR_PREFILTER_MAX_MIPS];
} VkrPrefilterMipsArgs;
VKR_EXPORT int32_t vkr_ibl_prefilter_mips(const VkrPrefilterMipsArgs *args, void *stream);

VKR_EXPORT int64_t vkr_ibl_irradiance_by_region_rows_workspace_bytes(int32_t env_n, int32_t n, int32_t row_begin, int32_t row_end);
VKR_EXPORT int32_t vkr_ibl_irradiance_by_region_rows(const void *env_tiles, int32_t env_n, void *out, int32_t n,
                           int32_t row_begin, int32_t row_end, void *workspace, int64_t workspace_bytes, void *stream);
/* Replaces the six draws of one mip of CreatePrefilteredCubemap (VulkanRenderer.cs:855-884). */
VKR_EXPORT int32_t vkr_ibl_prefilter(const void *env, int32_t env_n, int32_t env_layout, void *out, int32_t n, float roughness,
                          int32_t row_begin, int32_t row_end, void *stream);
/* Replaces CreateBRDFLUTTexture's draw (VulkanRenderer.cs:966-989); rows of the n^2 RG16F image. */
VKR_EXPORT int32_t vkr_ibl_brdf_lut(void *out, int32_t n, int32_t row_begin, int32_t row_end, void *stream);

/* ------------------------------------------------------------------------------------------
 * Layer 2: renderer object (host pointers in, device images owned by the context)
 * ------------------------------------------------------------------------------------------ */

typedef struct vkr_ctx vkr_ctx;

/* The reference's compile-time constants (VulkanRenderer.cs:60-76,1274,1300,1323,1348) as
 * run-time parameters.  Zero in any size field selects the reference's value. */
/* Depth ranges of the point-shadow cubes in the composition pass (vkr_cube_shadow_ranges, VkrShadeArgs.workspace): by
 * default a frame with at least VKR_SHADOW_RANGES_MIN_LIGHTS point lights builds and uses them (with few lights the
 * pass that sorts the lights per tile costs more than the taps it saves); the flags force them on or off. */
#define VKR_SHADOW_RANGES_ALWAYS 0x100
#define VKR_SHADOW_RANGES_NEVER 0x200
#define VKR_SHADOW_RANGES_MIN_LIGHTS 16
typedef struct VkrConfig {
    int32_t device;            /* HIP device ordinal */
    int32_t width, height;     /* frame size            (reference: 1920 x 1080, Program.cs:48) */
    int32_t max_lights;        /* MaxLights             (32) */
    int32_t dir_shadow_res;    /* ShadowMapResolution   (1024) */
    int32_t point_shadow_res;  /* PointShadowMapResolution (256) */
    int32_t env_res;           /* environment cube face (512) */
    int32_t irradiance_res;    /* (32) */
    int32_t prefilter_res;     /* (128) */
    int32_t prefilter_mips;    /* (5) */
    int32_t brdf_lut_res;      /* (512) */
    int32_t frames_in_flight;  /* MaxFramesInFlight     (2) */
    int32_t flags;             /* VKR_POST_* | VKR_SHADOW_RANGES_* */
    int32_t frame_rows, row0;  /* the context renders rows [row0, row0 + height) of a frame of frame_rows rows
                                  (see "Strips" below); 0, 0: the whole frame */
} VkrConfig;

#define VKR_TARGET_COLOR 0      /* composition colour, RGBA16F WxH */
#define VKR_TARGET_THRESHOLD 1  /* thresholded colour, RGBA16F WxH */
#define VKR_TARGET_BLOOM1 2     /* RGBA16F WxH */
#define VKR_TARGET_BLOOM2 3     /* RGBA16F WxH */
#define VKR_TARGET_LDR 4        /* RGBA8 WxH */
#define VKR_TARGET_ENV 5        /* RGBA16F cube */
#define VKR_TARGET_IRRADIANCE 6 /* RGBA16F cube */
#define VKR_TARGET_PREFILTER 7  /* RGBA16F cube, mip selected by `mip` */
#define VKR_TARGET_BRDF_LUT 8   /* RG16F */

VKR_EXPORT int32_t vkr_create(const VkrConfig *config, vkr_ctx **out);        /* VulkanRenderer..ctor, VulkanRenderer.cs:177 */
VKR_EXPORT int32_t vkr_destroy(vkr_ctx *ctx);                                 /* Dispose */
VKR_EXPORT int32_t vkr_resize(vkr_ctx *ctx, int32_t width, int32_t height);   /* RecreateSwapchain, VulkanRenderer.cs:1087-1127 */

VKR_EXPORT int32_t vkr_update_scene_info(vkr_ctx *ctx, const VkrSceneInfo *info);               /* UpdateSceneInfo, :635-646 */
VKR_EXPORT int32_t vkr_set_lights(vkr_ctx *ctx, const VkrLightInfo *lights, int32_t n, int32_t coverage_mode); /* Lights, :65 */
/* The mesh every light volume is an instance of (the reference loads assets/models/sphere/sphere.glb, a unit
 * sphere of 559 vertices / 960 triangles, VulkanRenderer.cs:78,269); needed for VKR_COVERAGE_MESH. */
VKR_EXPORT int32_t vkr_set_light_volume_mesh(vkr_ctx *ctx, const float *verts, int32_t n_verts, const uint32_t *indices, int32_t n_tris);
VKR_EXPORT int32_t vkr_begin_frame(vkr_ctx *ctx);                                                /* BeginFrame, :286-318 */
/* EndFrame, :320-623: composition pass, bloom 1, bloom 2, post-process, then advances the frame index.  The two
 * bloom images are not stored (VKR_TARGET_BLOOM1 / _BLOOM2 exist only after vkr_pass_bloom). */
VKR_EXPORT int32_t vkr_end_frame(vkr_ctx *ctx);
VKR_EXPORT int32_t vkr_current_frame(vkr_ctx *ctx);

/* Meshes, materials and the frame's draw calls (Mesh.cs, Material.cs, Model.cs).  vkr_draw_model is DrawModel
 * (VulkanRenderer.cs:630-633): it queues a mesh with a material and a model matrix between BeginFrame and
 * EndFrame.  With draw calls in a frame, EndFrame renders the point-light depth cubes and the directional depth
 * map from them (VulkanRenderer.cs:325-376, 406-428); if every draw has a material (material >= 0) and its mesh
 * has attributes, it also renders the G-buffer (VulkanRenderer.cs:378-404); if only some have, EndFrame fails with
 * VKR_ERR_INVALID.  What is not rendered is taken from the vkr_upload_* calls. */
VKR_EXPORT int32_t vkr_create_mesh(vkr_ctx *ctx, const float *positions, int32_t n_verts, const uint32_t *indices, int32_t n_tris, int32_t *mesh_out);
VKR_EXPORT int32_t vkr_set_mesh_attributes(vkr_ctx *ctx, int32_t mesh, const float *texcoords, const float *normals, const float *tangents);
/* Three single-level RGBA8 images: albedo (_SRGB), normal map (_UNORM), ao/roughness/metalness (_SRGB). */
VKR_EXPORT int32_t vkr_create_material(vkr_ctx *ctx, const void *albedo, int32_t albedo_w, int32_t albedo_h,
                                       const void *normal, int32_t normal_w, int32_t normal_h,
                                       const void *arm, int32_t arm_w, int32_t arm_h, int32_t *material_out);
VKR_EXPORT int32_t vkr_draw_model(vkr_ctx *ctx, int32_t mesh, int32_t material, const float *model16);

/* Images of passes the frame does not render itself (see vkr_draw_model), for the current frame slot (there are
 * frames_in_flight slots; vkr_current_frame() names the one the next EndFrame renders).  An upload stays in its slot
 * until it is overwritten.  vkr_create and vkr_resize set every slot to the reference's clear values (albedo
 * (0,0,0,0), normal / arm / position (0,0,0,1), depth 1.0: VulkanRenderer.cs:1155-1161), so a slot nothing was
 * uploaded to or rendered into shades as an empty scene. */
VKR_EXPORT int32_t vkr_upload_gbuffer(vkr_ctx *ctx, const void *albedo, const void *normal, const void *arm, const void *position);
/* An image in place of the skybox draw (VulkanRenderer.cs:436-443) for the current frame.  Without one, EndFrame
 * draws the skybox from the environment cube when there is one (vkr_create_skybox_cubemap / vkr_upload_env_cube)
 * and the frame's projection matrix has x and y scales; otherwise texels the composition draw discards stay (0,0,0,1). */
VKR_EXPORT int32_t vkr_upload_background(vkr_ctx *ctx, const void *rgba16f_or_null);
VKR_EXPORT int32_t vkr_upload_dir_shadow(vkr_ctx *ctx, const float *depth);
VKR_EXPORT int32_t vkr_upload_point_shadow(vkr_ctx *ctx, int32_t light, const float *depth_cube);

/* IBL precompute (synchronous, like the reference's QueueWaitIdle after each). */
VKR_EXPORT int32_t vkr_create_skybox_cubemap(vkr_ctx *ctx, const void *equirect, int32_t w, int32_t h, int32_t src_format); /* :653 */
VKR_EXPORT int32_t vkr_upload_env_cube(vkr_ctx *ctx, const void *rgba16f_cube);  /* an environment cube given directly */
VKR_EXPORT int32_t vkr_create_irradiance_cubemap(vkr_ctx *ctx, int32_t order);   /* :734; order: VKR_IRRADIANCE_* (the by-region
                                                                                  * orders allocate their workspace for the call) */
VKR_EXPORT int32_t vkr_create_prefiltered_cubemap(vkr_ctx *ctx);                 /* :815 */
VKR_EXPORT int32_t vkr_create_brdf_lut(vkr_ctx *ctx);                            /* :966 */

/* Individual passes of EndFrame, run at once on the context's stream.  vkr_pass_composition includes the shadow,
 * geometry, coverage and skybox work of the frame's draw calls; the bloom passes store VKR_TARGET_BLOOM1 / _BLOOM2. */
VKR_EXPORT int32_t vkr_pass_composition(vkr_ctx *ctx);
VKR_EXPORT int32_t vkr_pass_bloom(vkr_ctx *ctx, int32_t horizontal);
VKR_EXPORT int32_t vkr_pass_postprocess(vkr_ctx *ctx);

/* RenderStage (RenderStage.cs:99-190) for a caller that keeps the reference's recording shape: per stage
 * BeginCommands -> BeginRenderPass -> the stage's draws -> EndRenderPass -> EndCommands, as VulkanRenderer.EndFrame
 * does for each of its stages (VulkanRenderer.cs:346-374, 380-403, 408-427, 432-500, 504-521, 523-540, 544-557), then
 * one submission (VulkanRenderer.cs:559-595).  `stage` names the reference's stage object; `frame` is its
 * commandBufferIndex and must be vkr_current_frame().  Recording runs nothing: vkr_submit_frame runs the recorded
 * stages in the reference's submission order (the numbering below), then advances the frame index like EndFrame.
 * vkr_stage_draw records all draws the reference records into that stage (per light for the point-shadow stage and
 * the light volumes; per draw call for the geometry and shadow stages).  A render pass without draws leaves its
 * attachments at the reference's clear values -- also when vkr_stage_draw was called in a frame with no vkr_draw_model
 * calls, for the two shadow stages and the geometry stage: their draw loops are then empty, as in the reference.  A
 * stage that recorded nothing leaves its images as they are (uploaded, or from an earlier frame): a host that uploads
 * the G-buffer or the shadow maps instead of drawing them does not record those stages.  Every bloom image is stored, as in the reference (vkr_end_frame stores none).
 * Calls out of order fail with VKR_ERR_ORDER, as the reference throws on a failed vkBegin/EndCommandBuffer
 * (RenderStage.cs:106-121).  vkr_begin_frame resets every stage (VulkanRenderer.cs:307-314). */
#define VKR_STAGE_POINT_SHADOW 0  /* pointShadowRenderStages[0 .. Lights.Count) */
#define VKR_STAGE_GEOMETRY 1      /* geometryRenderStage */
#define VKR_STAGE_DIR_SHADOW 2    /* depthMapRenderStage */
#define VKR_STAGE_COMPOSITION 3   /* compositionRenderStage: skybox, composition and light-volume draws */
#define VKR_STAGE_BLOOM1 4        /* bloomRenderStage1 */
#define VKR_STAGE_BLOOM2 5        /* bloomRenderStage2 */
#define VKR_STAGE_POSTPROCESS 6   /* postProcessRenderStage */
#define VKR_STAGE_COUNT 7
VKR_EXPORT int32_t vkr_stage_reset_commands(vkr_ctx *ctx, int32_t stage, int32_t frame);     /* ResetCommandBuffer, :187 */
VKR_EXPORT int32_t vkr_stage_begin_commands(vkr_ctx *ctx, int32_t stage, int32_t frame);     /* BeginCommands, :99 */
VKR_EXPORT int32_t vkr_stage_begin_render_pass(vkr_ctx *ctx, int32_t stage, int32_t frame);  /* BeginRenderPass, :132 */
VKR_EXPORT int32_t vkr_stage_draw(vkr_ctx *ctx, int32_t stage, int32_t frame);               /* the vkCmdDraw* calls EndFrame records */
VKR_EXPORT int32_t vkr_stage_end_render_pass(vkr_ctx *ctx, int32_t stage, int32_t frame);    /* EndRenderPass, :177 */
VKR_EXPORT int32_t vkr_stage_end_commands(vkr_ctx *ctx, int32_t stage, int32_t frame);       /* EndCommands, :116 */
VKR_EXPORT int32_t vkr_submit_frame(vkr_ctx *ctx);                                           /* QueueSubmit + present + advance, VulkanRenderer.cs:559-622 */

/* Strips: one frame on several GPUs.  The reference renders on one device; its passes shard by rows, so a host gives
 * each GPU a context for rows [row0, row0 + height) of the frame (VkrConfig.frame_rows / row0; vkr_strip_plan cuts a
 * frame into n near-equal strips), feeds every context the same scene, lights, draw calls and IBL source and its own
 * rows of any uploaded image, and ends a frame in three steps instead of vkr_end_frame:
 *   vkr_strip_shade(ctx)      every pass up to the composition pass, for the strip (asynchronous);
 *   the halo rows             bloom pass 2 reads VKR_STRIP_HALO_ROWS rows of the thresholded image beyond each strip
 *                             edge that is not a frame edge.  Contexts of one process: vkr_strip_exchange(upper, lower)
 *                             per neighbouring pair (device-to-device copies, over xGMI between GPUs, ordered on the
 *                             contexts' streams: the host waits for nothing).  Contexts of different processes:
 *                             vkr_strip_halo gives the device pointers to send from and receive into (RCCL / MPI /
 *                             IPC of the host's choice, after vkr_sync and before vkr_strip_finish);
 *   vkr_strip_finish(ctx)     bloom 1, bloom 2 and post-process for the strip; advances the frame index.
 * vkr_read returns the strip's rows; the strips stacked in row order are the frame, bit for bit the image one
 * context renders alone (no pass reduces across rows).  Shadow maps and IBL images are rendered per context. */
#define VKR_STRIP_HALO_ROWS 4 /* bloom_blur.frag:20-23 reaches 4 texels to each side */
#define VKR_STRIP_ABOVE 0
#define VKR_STRIP_BELOW 1
VKR_EXPORT int32_t vkr_strip_plan(int32_t frame_rows, int32_t n_strips, int32_t k, int32_t *row0, int32_t *rows);
VKR_EXPORT int32_t vkr_strip_shade(vkr_ctx *ctx);
VKR_EXPORT int32_t vkr_strip_halo(vkr_ctx *ctx, int32_t side, const void **send, void **recv, int64_t *bytes);
VKR_EXPORT int32_t vkr_strip_exchange(vkr_ctx *upper, vkr_ctx *lower);
VKR_EXPORT int32_t vkr_strip_finish(vkr_ctx *ctx);
VKR_EXPORT int32_t vkr_resize_strip(vkr_ctx *ctx, int32_t width, int32_t height, int32_t frame_rows, int32_t row0);

/* Copies a target of frame `frame` (or an IBL image) to host memory; waits for pending work. */
VKR_EXPORT int32_t vkr_read(vkr_ctx *ctx, int32_t frame, int32_t target, int32_t mip, void *host, int64_t host_bytes);
VKR_EXPORT int32_t vkr_sync(vkr_ctx *ctx); /* WaitIdle */

/* Milliseconds the passes of the most recent EndFrame took on the device (HIP events).  EndFrame runs both bloom
 * passes and the post-process draw as one kernel: its time is bloom_v_post_ms, and bloom_h_ms is 0. */
typedef struct VkrTiming {
    float shade_ms, bloom_h_ms, bloom_v_post_ms, frame_ms;
} VkrTiming;
VKR_EXPORT int32_t vkr_get_timing(vkr_ctx *ctx, VkrTiming *out);

/* ------------------------------------------------------------------------------------------
 * Diagnostics
 * ------------------------------------------------------------------------------------------ */

/* Sweeps all 2^32 binary32 patterns on the current device and counts where the kernels' fast
 * correctly rounded 1/x and sqrt(x) differ from IEEE division / square root.  counts7[0..4] are
 * mismatch counts (rcp one Newton step, rcp two steps, sqrt, rcp of the IEEE sqrt,
 * the sqrt + reciprocal pair as shipped); counts7[5..6] the number of patterns tested for rcp and
 * for sqrt.  No counterpart in the reference: it backs the bit-exactness claim of the kernels. */
VKR_EXPORT int32_t vkr_selftest_math(uint64_t *counts7);

/* Measures, on the current device, the FP32 vector rate of an fma loop (TFLOP/s) and the rate of a
 * 1 GiB device-to-device copy (GB/s, bytes read + bytes written): the peaks a kernel on this
 * machine can actually reach, to set beside the nominal ones.  Allocates 2 GiB for the call. */
VKR_EXPORT int32_t vkr_measure_peaks(float *fp32_tflops, float *copy_gbs);

#ifdef __cplusplus
}
#endif
#endif /* VKR_HIP_H */
