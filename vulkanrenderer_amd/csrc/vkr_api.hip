// libvkr_hip.so: error reporting and the renderer object (layer 2 of include/vkr_hip.h).
// The renderer object owns the device images the reference keeps per frame in flight
// (G-buffer, shadow maps, composition / bloom / final targets: VulkanRenderer.cs:1129-1267)
// and the IBL images, and runs the passes of EndFrame (VulkanRenderer.cs:430-557) on one stream.
#include <stdarg.h>
#include <string.h>
#include <new>
#include <map>
#include <utility>
#include <vector>
#include "vkr_host.h"

namespace vkr {

static thread_local char g_error[512] = "";

int32_t fail(int32_t code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

struct Frame {
    void *gbuf[4] = {nullptr, nullptr, nullptr, nullptr};  // albedo, normal, arm, position
    void *background = nullptr;
    bool has_background = false;
    float *dir_shadow = nullptr;
    float *cube_shadow = nullptr;
    void *cube_ranges = nullptr;  // depth ranges of cube_shadow (vkr_cube_shadow_ranges), rebuilt before a composition pass
    int ranges_lights = 0;        // how many lights' ranges are current; 0 after any change of the cubes
    void *shade_workspace = nullptr;  // vkr_shade_workspace_bytes(width, height, max_lights): the per-tile light classes
    VkrLightInfo *lights = nullptr;
    int n_lights = 0;
    int coverage = VKR_COVERAGE_FULLSCREEN;
    VkrSceneInfo scene;
    void *color = nullptr, *threshold = nullptr, *bloom1 = nullptr, *bloom2 = nullptr, *ldr = nullptr;
    void *threshold_padded = nullptr;  // the allocation `threshold` lies in: VKR_STRIP_HALO_ROWS rows of the neighbouring strips above and below it
    uint64_t *coverage_buf = nullptr;  // light-volume masks, max_lights * tiles
    bool bloom1_valid = false, bloom2_valid = false;  // stored only by vkr_pass_bloom
};

}  // namespace vkr

struct vkr_ctx {
    VkrConfig cfg;
    hipStream_t stream = nullptr;
    std::vector<vkr::Frame> frames;
    int frame = 0;
    bool in_frame = false;
    void *env = nullptr, *irradiance = nullptr, *brdf_lut = nullptr;
    std::vector<void *> prefilter;  // one image per mip
    bool have_env = false, have_irr = false, have_pref = false, have_lut = false;
    void *env_tiled = nullptr;      // the environment cube as 4x4-texel tiles (what the skybox draw samples)
    void *env_foot = nullptr;       // the environment cube in footprint tiles (what the IBL passes sample), made on first use
    bool env_foot_valid = false;    // env_foot holds the current environment
    float *cube_staging = nullptr;  // one linear depth cube on its way into the tiled storage
    float *mesh_verts = nullptr;    // light-volume mesh (vkr_set_light_volume_mesh)
    uint32_t *mesh_indices = nullptr;
    int mesh_n_verts = 0, mesh_n_tris = 0;
    bool mesh_closed = false;       // every directed edge of the volume mesh has exactly one opposite twin (by position)
    struct Mesh { float *positions; uint32_t *indices; int n_verts, n_tris; float *texcoords, *normals, *tangents; };
    struct Material { VkrTexture albedo, normal, arm; };
    std::vector<Mesh> meshes;            // vkr_create_mesh
    std::vector<Material> materials;     // vkr_create_material
    std::vector<VkrMeshDraw> draw_calls; // vkr_draw_model since the last BeginFrame (solidModelDrawCalls)
    std::vector<int> draw_mesh, draw_material;  // per draw call; material -1: the draw only casts shadows
    void *gp_workspace = nullptr;        // visibility keys + draw records of the geometry pass
    int64_t gp_workspace_bytes = 0;
    bool cubes_tiled = false;       // point shadow cubes are stored VKR_CUBE_LAYOUT_TILED_8X4_X25
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool timing_valid = false;
    // strips (vkr_strip_*): "this frame's composition pass is enqueued", "the neighbour's rows have arrived"
    hipEvent_t ev_shaded = nullptr, ev_pulled = nullptr;
    bool strip_shaded = false;
    // RenderStage mirror (vkr_stage_*): what each stage's command buffer of the current frame holds
    enum StageState { STAGE_INITIAL, STAGE_RECORDING, STAGE_IN_PASS, STAGE_PASS_ENDED, STAGE_EXECUTABLE };
    struct Stage { StageState state = STAGE_INITIAL; bool cleared = false, drawn = false; };
    Stage stages[VKR_STAGE_COUNT];
};

namespace vkr {

static size_t rgba16f_bytes(int w, int h) { return (size_t)w * h * 8; }
static size_t halo_bytes(const vkr_ctx *c) { return rgba16f_bytes(c->cfg.width, VKR_STRIP_HALO_ROWS); }
static size_t cube_bytes(int n) { return (size_t)6 * n * n * 8; }

static void free_frame_sized(vkr_ctx *c)
{
    for (Frame &f : c->frames) {
        for (void *&g : f.gbuf) { if (g) (void)hipFree(g); g = nullptr; }
        void **imgs[] = {&f.background, &f.color, &f.threshold_padded, &f.bloom1, &f.bloom2, &f.ldr, (void **)&f.coverage_buf, &f.shade_workspace};
        for (void **p : imgs) { if (*p) (void)hipFree(*p); *p = nullptr; }
        f.threshold = nullptr;
        f.has_background = false;
        f.bloom1_valid = f.bloom2_valid = false;
    }
}

static int32_t alloc_frame_sized(vkr_ctx *c)
{
    const size_t img = rgba16f_bytes(c->cfg.width, c->cfg.height);
    for (Frame &f : c->frames) {
        for (void *&g : f.gbuf) VKR_HIP_CHECK(hipMalloc(&g, img));
        VKR_HIP_CHECK(hipMalloc(&f.background, img));
        VKR_HIP_CHECK(hipMalloc(&f.color, img));
        VKR_HIP_CHECK(hipMalloc(&f.threshold_padded, img + 2 * halo_bytes(c)));
        f.threshold = (char *)f.threshold_padded + halo_bytes(c);
        VKR_HIP_CHECK(hipMalloc(&f.bloom1, img));
        VKR_HIP_CHECK(hipMalloc(&f.bloom2, img));
        VKR_HIP_CHECK(hipMalloc(&f.ldr, (size_t)c->cfg.width * c->cfg.height * 4));
        const size_t tiles = (size_t)((c->cfg.width + 7) / 8) * ((c->cfg.height + 7) / 8);
        // the masks, then the per-light words of VkrCoverageArgs.light_flags (max_lights + 1, rounded up to whole masks)
        VKR_HIP_CHECK(hipMalloc((void **)&f.coverage_buf, ((size_t)c->cfg.max_lights * tiles + (size_t)c->cfg.max_lights / 2 + 1) * sizeof(uint64_t)));
        VKR_HIP_CHECK(hipMalloc(&f.shade_workspace, (size_t)std::max<int64_t>(vkr_shade_workspace_bytes(c->cfg.width, c->cfg.height, c->cfg.max_lights), 16)));
    }
    return VKR_OK;
}

// What the reference's render passes clear their attachments to (VulkanRenderer.cs:1155-1161: albedo (0,0,0,0),
// normal / arm / position (0,0,0,1); depth 1.0), so that a frame slot nothing was uploaded to or rendered into
// shades as an empty scene instead of whatever the allocation held.
static constexpr unsigned long long RGBA16F_0001 = 0x3C00000000000000ull;  // binary16 (0, 0, 0, 1)
static constexpr unsigned long long D32F_ONE_PAIR = 0x3F8000003F800000ull;  // two texels of depth 1.0
static constexpr unsigned long long D32F_25_PAIR = 0x41C8000041C80000ull;   // two texels of 25 x 1.0 (the _X25 layout)

static int32_t clear_frame_sized(vkr_ctx *c)
{
    const size_t px = (size_t)c->cfg.width * c->cfg.height;
    for (Frame &f : c->frames) {
        int32_t rc = fill_u64(f.gbuf[0], px, 0ull, c->stream);
        for (int k = 1; k < 4 && rc == VKR_OK; k++) rc = fill_u64(f.gbuf[k], px, RGBA16F_0001, c->stream);
        void *imgs[] = {f.background, f.color, f.bloom1, f.bloom2};
        for (void *img : imgs) if (rc == VKR_OK) rc = fill_u64(img, px, RGBA16F_0001, c->stream);
        if (rc == VKR_OK) rc = fill_u64(f.threshold_padded, px + 2 * (size_t)c->cfg.width * VKR_STRIP_HALO_ROWS, RGBA16F_0001, c->stream);
        if (rc == VKR_OK) rc = fill_u64(f.ldr, px / 2, 0xFF000000FF000000ull, c->stream);
        if (rc != VKR_OK) return rc;
        if (px & 1) VKR_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)((char *)f.ldr + (px - 1) * 4), (int)0xFF000000u, 1, c->stream));  // (0, 0, 0, 255)
    }
    VKR_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VKR_OK;
}

static int32_t clear_shadow_maps(vkr_ctx *c)
{
    const size_t dir = (size_t)c->cfg.dir_shadow_res * c->cfg.dir_shadow_res;
    const size_t cube = (size_t)c->cfg.max_lights * 6 * c->cfg.point_shadow_res * c->cfg.point_shadow_res;
    for (Frame &f : c->frames) {
        int32_t rc = fill_u64(f.dir_shadow, dir / 2, D32F_ONE_PAIR, c->stream);
        if (rc == VKR_OK) rc = fill_u64(f.cube_shadow, cube / 2, c->cubes_tiled ? D32F_25_PAIR : D32F_ONE_PAIR, c->stream);
        f.ranges_lights = 0;
        if (rc != VKR_OK) return rc;
        const float one = 1.0f;  // a map with an odd number of texels: the last one
        if (dir & 1) VKR_HIP_CHECK(hipMemcpyAsync(f.dir_shadow + dir - 1, &one, 4, hipMemcpyHostToDevice, c->stream));
        if (cube & 1) VKR_HIP_CHECK(hipMemcpyAsync(f.cube_shadow + cube - 1, &one, 4, hipMemcpyHostToDevice, c->stream));
        VKR_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return VKR_OK;
}

static void destroy(vkr_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_frame_sized(c);
    for (Frame &f : c->frames) {
        if (f.dir_shadow) (void)hipFree(f.dir_shadow);
        if (f.cube_shadow) (void)hipFree(f.cube_shadow);
        if (f.cube_ranges) (void)hipFree(f.cube_ranges);
        if (f.lights) (void)hipFree(f.lights);
    }
    void *ibl[] = {c->env, c->env_tiled, c->env_foot, c->irradiance, c->brdf_lut, c->cube_staging, c->mesh_verts, c->mesh_indices};
    for (auto &m : c->meshes) {
        (void)hipFree(m.positions); (void)hipFree(m.indices);
        if (m.texcoords) (void)hipFree(m.texcoords);
        if (m.normals) (void)hipFree(m.normals);
        if (m.tangents) (void)hipFree(m.tangents);
    }
    for (auto &m : c->materials) { (void)hipFree((void *)m.albedo.rgba8); (void)hipFree((void *)m.normal.rgba8); (void)hipFree((void *)m.arm.rgba8); }
    if (c->gp_workspace) (void)hipFree(c->gp_workspace);
    for (void *p : ibl) if (p) (void)hipFree(p);
    for (void *p : c->prefilter) if (p) (void)hipFree(p);
    for (hipEvent_t e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->ev_shaded) (void)hipEventDestroy(c->ev_shaded);
    if (c->ev_pulled) (void)hipEventDestroy(c->ev_pulled);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// the per-light words of VkrCoverageArgs.light_flags live behind the masks
static uint32_t *coverage_flags(vkr_ctx *c, Frame &f)
{
    const size_t tiles = (size_t)((c->cfg.width + 7) / 8) * ((c->cfg.height + 7) / 8);
    return (uint32_t *)(f.coverage_buf + (size_t)c->cfg.max_lights * tiles);
}

static int32_t run_coverage(vkr_ctx *c, Frame &f);
static int32_t run_shadow_passes(vkr_ctx *c, Frame &f);
static int32_t run_geometry_pass(vkr_ctx *c, Frame &f);
static int32_t run_skybox(vkr_ctx *c, Frame &f);

#define VKR_CTX(c)                                                   \
    VKR_REQUIRE((c) != nullptr, "%s: context is null", __func__);    \
    VKR_HIP_CHECK(hipSetDevice((c)->cfg.device))

// The skybox is drawn when there is an environment cube, no uploaded background for the frame, and a camera
// whose projection has x and y scales (a zeroed SceneInfo has none: nothing to look through).
static bool draws_skybox(const vkr_ctx *c, const Frame &f)
{
    return !f.has_background && c->have_env && f.scene.camera_projection[0] != 0.0f && f.scene.camera_projection[5] != 0.0f;
}

static int32_t fill_shade_args(vkr_ctx *c, Frame &f, VkrShadeArgs &a)
{
    if (!(c->have_irr && c->have_pref && c->have_lut))
        return fail(VKR_ERR_ORDER, "composition pass needs the irradiance cube, the prefiltered cube and the BRDF LUT; create them first");
    memset(&a, 0, sizeof(a));
    a.width = c->cfg.width; a.rows = c->cfg.height;
    a.albedo = f.gbuf[0]; a.normal = f.gbuf[1]; a.arm = f.gbuf[2]; a.position = f.gbuf[3];
    a.background = f.has_background || draws_skybox(c, f) ? f.background : nullptr;  // uploaded, or drawn by run_skybox
    a.scene = f.scene;
    a.lights = f.lights; a.n_lights = f.n_lights; a.coverage_mode = f.coverage;
    a.dir_shadow = f.dir_shadow; a.dir_res = c->cfg.dir_shadow_res;
    a.cube_shadow = f.cube_shadow; a.cube_res = c->cfg.point_shadow_res;
    a.cube_layout = c->cubes_tiled ? VKR_CUBE_LAYOUT_TILED_8X4_X25 : VKR_CUBE_LAYOUT_LINEAR;
    a.cube_ranges = f.n_lights > 0 ? f.cube_ranges : nullptr;  // brought up to date by shade() below
    a.workspace = f.shade_workspace;
    a.workspace_bytes = vkr_shade_workspace_bytes(c->cfg.width, c->cfg.height, c->cfg.max_lights);
    a.irradiance = c->irradiance; a.irr_res = c->cfg.irradiance_res;
    a.prefilter = c->prefilter[0]; a.pref_res = c->cfg.prefilter_res;
    a.brdf_lut = c->brdf_lut; a.lut_res = c->cfg.brdf_lut_res;
    a.out_color = f.color; a.out_threshold = f.threshold;
    a.coverage_masks = f.coverage == VKR_COVERAGE_MESH ? f.coverage_buf : nullptr;
    a.coverage_flags = f.coverage == VKR_COVERAGE_MESH && c->mesh_closed ? coverage_flags(c, f) : nullptr;
    return VKR_OK;
}

// The composition pass proper: the depth ranges of the point-shadow cubes are rebuilt first when the cubes changed
// since they were last built (a shadow pass ran, a cube was uploaded or cleared) or more lights are in use.
static int32_t shade(vkr_ctx *c, Frame &f, VkrShadeArgs &a)
{
    const bool ranges = !(c->cfg.flags & VKR_SHADOW_RANGES_NEVER) && f.n_lights > 0 &&
                        ((c->cfg.flags & VKR_SHADOW_RANGES_ALWAYS) || f.n_lights >= VKR_SHADOW_RANGES_MIN_LIGHTS);
    if (!ranges) {
        a.cube_ranges = nullptr; a.workspace = nullptr; a.workspace_bytes = 0;
    } else if (f.ranges_lights < f.n_lights) {
        const int32_t rc = vkr_cube_shadow_ranges(f.cube_shadow, f.n_lights, a.cube_res, a.cube_layout, f.cube_ranges, c->stream);
        if (rc != VKR_OK) return rc;
        f.ranges_lights = f.n_lights;
    }
    return vkr_shade(&a, c->stream);
}

static constexpr int GP_MAX_DRAWS = 1024;  // draw records the geometry-pass workspace has room for

// The geometry render pass of EndFrame (VulkanRenderer.cs:378-404).  It runs when every draw call of the
// frame has a material and vertex attributes; otherwise the uploaded G-buffer is used.
static int32_t run_geometry_pass(vkr_ctx *c, Frame &f)
{
    if (c->draw_calls.empty()) return VKR_OK;
    // Either every draw of the frame can be shaded (material and vertex attributes) and the pass runs, or none can
    // and the draws only cast shadows over an uploaded G-buffer.  A mixture would silently drop geometry.
    size_t shadeable = 0;
    for (size_t k = 0; k < c->draw_calls.size(); k++)
        if (c->draw_material[k] >= 0 && c->meshes[c->draw_mesh[k]].texcoords) shadeable++;
    if (shadeable == 0) return VKR_OK;
    if (shadeable != c->draw_calls.size())
        return fail(VKR_ERR_INVALID, "%zu of the frame's %zu draw calls have a material and vertex attributes and the rest do not: "
                    "the geometry pass needs them for every draw (or for none, with the G-buffer uploaded)", shadeable, c->draw_calls.size());
    std::vector<VkrGpassDraw> draws(c->draw_calls.size());
    for (size_t k = 0; k < draws.size(); k++) {
        const vkr_ctx::Mesh &m = c->meshes[c->draw_mesh[k]];
        const vkr_ctx::Material &mat = c->materials[c->draw_material[k]];
        VkrGpassDraw &d = draws[k];
        d.positions = m.positions; d.texcoords = m.texcoords; d.normals = m.normals; d.tangents = m.tangents;
        d.n_verts = m.n_verts; d.indices = m.indices; d.n_tris = m.n_tris;
        memcpy(d.model, c->draw_calls[k].model, sizeof(d.model));
        d.albedo = mat.albedo; d.normal = mat.normal; d.arm = mat.arm;
    }
    if (draws.size() > (size_t)GP_MAX_DRAWS) return fail(VKR_ERR_UNSUPPORTED, "more than %d draw calls in a frame", GP_MAX_DRAWS);
    // Room for a record per triangle as well, so that the corners are shaded once per triangle.  The workspace grows when
    // a frame brings more triangles than any before it (the frames in flight are waited for first: they use it).
    int64_t triangles = 0;
    for (const VkrGpassDraw &d : draws) triangles += d.n_tris;
    const int64_t wanted = vkr_gbuffer_workspace_bytes_with_triangles(c->cfg.width, c->cfg.height, GP_MAX_DRAWS, triangles);
    if (wanted < 0) return fail(VKR_ERR_UNSUPPORTED, "too many triangles in a frame");
    if (!c->gp_workspace || c->gp_workspace_bytes < wanted) {
        if (c->gp_workspace) {
            VKR_HIP_CHECK(hipStreamSynchronize(c->stream));
            (void)hipFree(c->gp_workspace);
            c->gp_workspace = nullptr;
        }
        c->gp_workspace_bytes = wanted + wanted / 4;  // headroom: a scene that grows a little does not reallocate every frame
        VKR_HIP_CHECK(hipMalloc(&c->gp_workspace, (size_t)c->gp_workspace_bytes));
    }
    VkrGpassArgs a;
    memset(&a, 0, sizeof(a));
    a.draws = draws.data(); a.n_draws = (int32_t)draws.size();
    a.scene = f.scene;
    a.width = c->cfg.width; a.rows = c->cfg.height; a.frame_rows = c->cfg.frame_rows; a.row0 = c->cfg.row0;
    a.albedo = f.gbuf[0]; a.normal = f.gbuf[1]; a.arm = f.gbuf[2]; a.position = f.gbuf[3];
    a.workspace = c->gp_workspace; a.workspace_bytes = c->gp_workspace_bytes;
    return vkr_gbuffer(&a, c->stream);
}

// The shadow render passes of EndFrame (VulkanRenderer.cs:325-376, 406-428) over the frame's draw calls.
// Without draw calls the uploaded maps stay as they are.
static int32_t run_point_shadows(vkr_ctx *c, Frame &f)
{
    if (c->draw_calls.empty()) return VKR_OK;
    VkrPointShadowArgs ps;
    memset(&ps, 0, sizeof(ps));
    ps.draws = c->draw_calls.data(); ps.n_draws = (int32_t)c->draw_calls.size();
    ps.lights = f.lights; ps.n_lights = f.n_lights;
    ps.res = c->cfg.point_shadow_res; ps.far_plane = 25.0f;  // VulkanRenderer.cs:331
    ps.layout = c->cubes_tiled ? VKR_CUBE_LAYOUT_TILED_8X4_X25 : VKR_CUBE_LAYOUT_LINEAR;
    ps.out = f.cube_shadow;
    f.ranges_lights = 0;
    return vkr_point_shadow_maps(&ps, c->stream);
}

static int32_t run_dir_shadow(vkr_ctx *c, Frame &f)
{
    if (c->draw_calls.empty()) return VKR_OK;
    VkrDirShadowArgs ds;
    memset(&ds, 0, sizeof(ds));
    ds.draws = c->draw_calls.data(); ds.n_draws = (int32_t)c->draw_calls.size();
    memcpy(ds.light_space, f.scene.light_space_matrix, sizeof(ds.light_space));
    ds.res = c->cfg.dir_shadow_res; ds.out = f.dir_shadow;
    return vkr_dir_shadow_map(&ds, c->stream);
}

static int32_t run_shadow_passes(vkr_ctx *c, Frame &f)
{
    const int32_t rc = run_point_shadows(c, f);
    return rc != VKR_OK ? rc : run_dir_shadow(c, f);
}

// The light-volume draws' coverage (what rasterising sphere.glb per light decides in the reference).
static int32_t run_coverage(vkr_ctx *c, Frame &f)
{
    if (f.coverage != VKR_COVERAGE_MESH || f.n_lights == 0) return VKR_OK;
    if (!c->mesh_verts) return fail(VKR_ERR_ORDER, "VKR_COVERAGE_MESH needs a light-volume mesh: call vkr_set_light_volume_mesh first");
    VkrCoverageArgs a;
    memset(&a, 0, sizeof(a));
    a.width = c->cfg.width; a.rows = c->cfg.height; a.frame_rows = c->cfg.frame_rows; a.row0 = c->cfg.row0;
    a.scene = f.scene;
    a.lights = f.lights; a.n_lights = f.n_lights;
    a.verts = c->mesh_verts; a.n_verts = c->mesh_n_verts;
    a.indices = c->mesh_indices; a.n_tris = c->mesh_n_tris;
    a.masks = f.coverage_buf;
    // "covers every pixel" is concluded from back faces alone, which holds for a closed mesh only
    a.light_flags = c->mesh_closed ? coverage_flags(c, f) : nullptr;
    return vkr_light_coverage(&a, c->stream);
}

// The skybox draw that opens the composition pass (VulkanRenderer.cs:436-443), from the environment cube.
// An image given with vkr_upload_background for this frame takes its place.
static int32_t run_skybox(vkr_ctx *c, Frame &f)
{
    if (!draws_skybox(c, f)) return VKR_OK;
    const void *env = c->env_tiled ? c->env_tiled : c->env;
    const int32_t layout = c->env_tiled ? VKR_ENV_LAYOUT_TILED_4X4 : VKR_ENV_LAYOUT_LINEAR;
    return vkr_skybox(env, c->cfg.env_res, layout, &f.scene, c->cfg.width, c->cfg.height, c->cfg.frame_rows, c->cfg.row0, f.background, c->stream);
}

static VkrBloomArgs bloom_args(vkr_ctx *c, Frame &f, int horizontal)
{
    VkrBloomArgs b;
    memset(&b, 0, sizeof(b));
    // pass 1 reads the thresholded target through textureSampler (REPEAT), pass 2 reads pass 1's
    // output through bloomSampler (CLAMP_TO_EDGE): VulkanRenderer.cs:231-235
    b.src = horizontal ? f.threshold : f.bloom1;
    b.dst = horizontal ? f.bloom1 : f.bloom2;
    b.width = c->cfg.width; b.rows = c->cfg.height; b.src_rows = c->cfg.height; b.src_row_offset = 0;
    b.horizontal = horizontal;
    b.wrap = horizontal ? VKR_WRAP_REPEAT : VKR_WRAP_CLAMP;
    b.clamp_lo = 0; b.clamp_hi = c->cfg.height - 1;
    return b;
}

// Whether a triangle mesh is closed and consistently wound: every directed edge between two different positions occurs
// once, and so does its opposite.  Vertices are matched by position (bit for bit), not by index: meshes that carry
// texture coordinates repeat the positions along a seam (the reference's sphere.glb has 559 vertices at 482 positions).
static bool mesh_is_closed(const float *verts, const uint32_t *indices, int n_tris)
{
    struct Key { uint32_t x, y, z; bool operator<(const Key &o) const { return x != o.x ? x < o.x : (y != o.y ? y < o.y : z < o.z); } };
    auto key = [&](uint32_t v) {
        Key k;
        float p[3] = {verts[3 * v] + 0.0f, verts[3 * v + 1] + 0.0f, verts[3 * v + 2] + 0.0f};  // -0 and +0 are one position
        memcpy(&k, p, 12);
        return k;
    };
    std::map<Key, uint32_t> ids;
    std::map<std::pair<uint32_t, uint32_t>, int> edges;
    for (int t = 0; t < n_tris; t++) {
        uint32_t id[3];
        for (int k = 0; k < 3; k++) id[k] = ids.emplace(key(indices[3 * t + k]), (uint32_t)ids.size()).first->second;
        for (int k = 0; k < 3; k++) {
            const uint32_t a = id[k], b = id[(k + 1) % 3];
            if (a != b) edges[{a, b}]++;
        }
    }
    for (const auto &e : edges) {
        if (e.second != 1) return false;
        auto twin = edges.find({e.first.second, e.first.first});
        if (twin == edges.end() || twin->second != 1) return false;
    }
    return !edges.empty();
}

}  // namespace vkr

using namespace vkr;

extern "C" const char *vkr_last_error(void) { return g_error; }
extern "C" int32_t vkr_abi_version(void) { return VKR_ABI_VERSION; }

extern "C" int32_t vkr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int usable = 0;
    for (int d = 0; d < n; d++) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) usable++;
    }
    return usable;
}

extern "C" int32_t vkr_create(const VkrConfig *config, vkr_ctx **out)
{
    VKR_REQUIRE(config && out, "vkr_create: null argument");
    *out = nullptr;
    VkrConfig cfg = *config;
    // zero selects the reference's constants (VulkanRenderer.cs:60-76,1274,1300,1323,1348; Program.cs:48)
    if (cfg.width == 0) cfg.width = 1920;
    if (cfg.height == 0) cfg.height = 1080;
    if (cfg.max_lights == 0) cfg.max_lights = 32;
    if (cfg.dir_shadow_res == 0) cfg.dir_shadow_res = 1024;
    if (cfg.point_shadow_res == 0) cfg.point_shadow_res = 256;
    if (cfg.env_res == 0) cfg.env_res = 512;
    if (cfg.irradiance_res == 0) cfg.irradiance_res = 32;
    if (cfg.prefilter_res == 0) cfg.prefilter_res = 128;
    if (cfg.prefilter_mips == 0) cfg.prefilter_mips = 5;
    if (cfg.brdf_lut_res == 0) cfg.brdf_lut_res = 512;
    if (cfg.frames_in_flight == 0) cfg.frames_in_flight = 2;
    VKR_REQUIRE(cfg.width > 0 && cfg.height > 0, "vkr_create: bad frame size %dx%d", cfg.width, cfg.height);
    if (cfg.frame_rows == 0) { cfg.frame_rows = cfg.height; cfg.row0 = 0; }  // the context renders the whole frame
    VKR_REQUIRE(cfg.row0 >= 0 && cfg.row0 + cfg.height <= cfg.frame_rows, "vkr_create: rows [%d,%d) lie outside a frame of %d rows", cfg.row0, cfg.row0 + cfg.height, cfg.frame_rows);
    VKR_REQUIRE(cfg.frame_rows == cfg.height || cfg.height >= VKR_STRIP_HALO_ROWS, "vkr_create: a strip needs at least %d rows (got %d)", VKR_STRIP_HALO_ROWS, cfg.height);
    VKR_REQUIRE(cfg.max_lights > 0 && cfg.frames_in_flight > 0 && cfg.frames_in_flight <= 8, "vkr_create: bad max_lights / frames_in_flight");
    VKR_REQUIRE(cfg.dir_shadow_res > 0 && cfg.dir_shadow_res <= 32768 && cfg.point_shadow_res > 0 && cfg.point_shadow_res <= 8192, "vkr_create: bad shadow map size");
    VKR_REQUIRE(cfg.env_res > 0 && cfg.irradiance_res > 0 && cfg.prefilter_res > 0 && cfg.brdf_lut_res > 0, "vkr_create: bad IBL size");
    VKR_REQUIRE(cfg.env_res <= 8192 && cfg.irradiance_res <= 4096, "vkr_create: environment faces of at most 8192^2 and irradiance faces of at most 4096^2 are supported (%d, %d)", cfg.env_res, cfg.irradiance_res);
    VKR_REQUIRE(cfg.prefilter_mips <= VKR_PREFILTER_MAX_MIPS, "vkr_create: %d prefilter mips; at most %d are supported", cfg.prefilter_mips, VKR_PREFILTER_MAX_MIPS);
    VKR_REQUIRE(cfg.prefilter_mips > 0 && (cfg.prefilter_res >> (cfg.prefilter_mips - 1)) > 0, "vkr_create: %d mips do not fit a %d^2 prefilter cube", cfg.prefilter_mips, cfg.prefilter_res);

    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { (void)hipGetLastError(); return fail(VKR_ERR_NO_DEVICE, "vkr_create: no HIP device visible"); }
    VKR_REQUIRE(cfg.device >= 0 && cfg.device < n, "vkr_create: device %d outside [0,%d)", cfg.device, n);
    VKR_HIP_CHECK(hipSetDevice(cfg.device));

    vkr_ctx *c = new (std::nothrow) vkr_ctx;
    if (!c) return fail(VKR_ERR_INVALID, "vkr_create: out of host memory");
    c->cfg = cfg;
    c->frames.resize(cfg.frames_in_flight);
    c->prefilter.assign(cfg.prefilter_mips, nullptr);
    int32_t rc = VKR_OK;
    auto body = [&]() -> int32_t {
        VKR_HIP_CHECK(hipStreamCreate(&c->stream));
        for (hipEvent_t &e : c->ev) VKR_HIP_CHECK(hipEventCreate(&e));
        VKR_HIP_CHECK(hipEventCreateWithFlags(&c->ev_shaded, hipEventDisableTiming));
        VKR_HIP_CHECK(hipEventCreateWithFlags(&c->ev_pulled, hipEventDisableTiming));
        int32_t r = alloc_frame_sized(c);
        if (r != VKR_OK) return r;
        for (Frame &f : c->frames) {
            memset(&f.scene, 0, sizeof(f.scene));
            VKR_HIP_CHECK(hipMalloc((void **)&f.dir_shadow, (size_t)cfg.dir_shadow_res * cfg.dir_shadow_res * 4));
            VKR_HIP_CHECK(hipMalloc((void **)&f.cube_shadow, (size_t)cfg.max_lights * 6 * cfg.point_shadow_res * cfg.point_shadow_res * 4));
            VKR_HIP_CHECK(hipMalloc(&f.cube_ranges, (size_t)vkr_cube_shadow_ranges_bytes(cfg.max_lights, cfg.point_shadow_res)));
            VKR_HIP_CHECK(hipMalloc((void **)&f.lights, (size_t)cfg.max_lights * sizeof(VkrLightInfo)));
        }
        c->cubes_tiled = cfg.point_shadow_res % 8 == 0;
        if (c->cubes_tiled) VKR_HIP_CHECK(hipMalloc((void **)&c->cube_staging, (size_t)6 * cfg.point_shadow_res * cfg.point_shadow_res * 4));
        if ((r = clear_frame_sized(c)) != VKR_OK) return r;
        if ((r = clear_shadow_maps(c)) != VKR_OK) return r;
        VKR_HIP_CHECK(hipMalloc(&c->env, cube_bytes(cfg.env_res)));
        if (cfg.env_res % 4 == 0) VKR_HIP_CHECK(hipMalloc(&c->env_tiled, cube_bytes(cfg.env_res)));
        VKR_HIP_CHECK(hipMalloc(&c->irradiance, cube_bytes(cfg.irradiance_res)));
        VKR_HIP_CHECK(hipMalloc(&c->brdf_lut, (size_t)cfg.brdf_lut_res * cfg.brdf_lut_res * 4));
        for (int m = 0; m < cfg.prefilter_mips; m++) VKR_HIP_CHECK(hipMalloc(&c->prefilter[m], cube_bytes(cfg.prefilter_res >> m)));
        return VKR_OK;
    };
    rc = body();
    if (rc != VKR_OK) { destroy(c); return rc; }
    *out = c;
    return VKR_OK;
}

extern "C" int32_t vkr_destroy(vkr_ctx *ctx)
{
    destroy(ctx);
    return VKR_OK;
}

extern "C" int32_t vkr_resize_strip(vkr_ctx *c, int32_t width, int32_t height, int32_t frame_rows, int32_t row0)
{
    VKR_CTX(c);
    VKR_REQUIRE(width > 0 && height > 0, "vkr_resize: bad size %dx%d", width, height);
    VKR_REQUIRE(row0 >= 0 && row0 + height <= frame_rows, "vkr_resize: rows [%d,%d) lie outside a frame of %d rows", row0, row0 + height, frame_rows);
    VKR_REQUIRE(frame_rows == height || height >= VKR_STRIP_HALO_ROWS, "vkr_resize: a strip needs at least %d rows (got %d)", VKR_STRIP_HALO_ROWS, height);
    VKR_REQUIRE(!c->in_frame, "vkr_resize: called between BeginFrame and EndFrame");
    VKR_HIP_CHECK(hipStreamSynchronize(c->stream));
    free_frame_sized(c);
    c->cfg.width = width; c->cfg.height = height; c->cfg.frame_rows = frame_rows; c->cfg.row0 = row0;
    c->timing_valid = false;
    if (c->gp_workspace) { (void)hipFree(c->gp_workspace); c->gp_workspace = nullptr; }
    int32_t rc = alloc_frame_sized(c);
    return rc == VKR_OK ? clear_frame_sized(c) : rc;
}

extern "C" int32_t vkr_resize(vkr_ctx *c, int32_t width, int32_t height)
{
    VKR_REQUIRE(c != nullptr, "vkr_resize: context is null");
    VKR_REQUIRE(c->cfg.frame_rows == c->cfg.height, "vkr_resize: the context renders a strip of a frame; use vkr_resize_strip");
    return vkr_resize_strip(c, width, height, height, 0);
}

extern "C" int32_t vkr_update_scene_info(vkr_ctx *c, const VkrSceneInfo *info)
{
    VKR_CTX(c);
    VKR_REQUIRE(info != nullptr, "vkr_update_scene_info: info is null");
    c->frames[c->frame].scene = *info;
    return VKR_OK;
}

extern "C" int32_t vkr_set_lights(vkr_ctx *c, const VkrLightInfo *lights, int32_t n, int32_t coverage_mode)
{
    VKR_CTX(c);
    VKR_REQUIRE(n >= 0 && n <= c->cfg.max_lights, "vkr_set_lights: %d lights outside [0,%d]", n, c->cfg.max_lights);
    VKR_REQUIRE(n == 0 || lights != nullptr, "vkr_set_lights: lights is null");
    VKR_REQUIRE(coverage_mode == VKR_COVERAGE_FULLSCREEN || coverage_mode == VKR_COVERAGE_MESH, "vkr_set_lights: unknown coverage mode %d", coverage_mode);
    if (coverage_mode == VKR_COVERAGE_MESH && !c->mesh_verts) return fail(VKR_ERR_ORDER, "vkr_set_lights: VKR_COVERAGE_MESH needs a light-volume mesh (vkr_set_light_volume_mesh)");
    Frame &f = c->frames[c->frame];
    if (n) {
        VKR_HIP_CHECK(hipMemcpyAsync(f.lights, lights, (size_t)n * sizeof(VkrLightInfo), hipMemcpyHostToDevice, c->stream));
        VKR_HIP_CHECK(hipStreamSynchronize(c->stream));  // `lights` is borrowed for this call only
    }
    f.n_lights = n; f.coverage = coverage_mode;
    return VKR_OK;
}

extern "C" int32_t vkr_begin_frame(vkr_ctx *c)
{
    VKR_CTX(c);
    if (c->in_frame) return fail(VKR_ERR_ORDER, "Tried to begin frame before previous frame was finished!");  // VulkanRenderer.cs:288
    c->in_frame = true;
    c->strip_shaded = false;
    c->draw_calls.clear();  // VulkanRenderer.cs:316
    c->draw_mesh.clear(); c->draw_material.clear();
    for (vkr_ctx::Stage &st : c->stages) st = vkr_ctx::Stage();  // ResetCommandBuffer of every stage, VulkanRenderer.cs:307-314
    return VKR_OK;
}

extern "C" int32_t vkr_current_frame(vkr_ctx *c)
{
    VKR_REQUIRE(c != nullptr, "vkr_current_frame: context is null");
    return c->frame;
}

static int32_t upload(vkr_ctx *c, void *dst, const void *src, size_t bytes)
{
    VKR_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    VKR_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VKR_OK;
}

extern "C" int32_t vkr_set_light_volume_mesh(vkr_ctx *c, const float *verts, int32_t n_verts, const uint32_t *indices, int32_t n_tris)
{
    VKR_CTX(c);
    VKR_REQUIRE(verts && indices && n_verts > 0 && n_tris > 0, "vkr_set_light_volume_mesh: mesh missing");
    VKR_REQUIRE(n_verts <= 1024, "vkr_set_light_volume_mesh: %d vertices; at most 1024 are supported", n_verts);
    for (int64_t i = 0; i < (int64_t)n_tris * 3; i++)
        VKR_REQUIRE(indices[i] < (uint32_t)n_verts, "vkr_set_light_volume_mesh: index %u at %lld is outside the %d vertices", indices[i], (long long)i, n_verts);
    const bool closed = mesh_is_closed(verts, indices, n_tris);
    // the new mesh replaces the old one only once it is complete on the device
    float *new_verts = nullptr;
    uint32_t *new_indices = nullptr;
    VKR_HIP_CHECK(hipMalloc((void **)&new_verts, (size_t)n_verts * 12));
    hipError_t e = hipMalloc((void **)&new_indices, (size_t)n_tris * 12);
    int32_t rc = e == hipSuccess ? upload(c, new_verts, verts, (size_t)n_verts * 12) : fail(VKR_ERR_HIP, "vkr_set_light_volume_mesh: %s", hipGetErrorString(e));
    if (rc == VKR_OK) rc = upload(c, new_indices, indices, (size_t)n_tris * 12);
    if (rc != VKR_OK) { (void)hipFree(new_verts); if (new_indices) (void)hipFree(new_indices); return rc; }
    VKR_HIP_CHECK(hipStreamSynchronize(c->stream));  // frames that still read the old mesh
    if (c->mesh_verts) (void)hipFree(c->mesh_verts);
    if (c->mesh_indices) (void)hipFree(c->mesh_indices);
    c->mesh_verts = new_verts; c->mesh_indices = new_indices;
    c->mesh_n_verts = n_verts; c->mesh_n_tris = n_tris;
    c->mesh_closed = closed;
    return VKR_OK;
}

extern "C" int32_t vkr_create_mesh(vkr_ctx *c, const float *positions, int32_t n_verts, const uint32_t *indices, int32_t n_tris, int32_t *mesh_out)
{
    VKR_CTX(c);
    VKR_REQUIRE(positions && indices && mesh_out && n_verts > 0 && n_tris > 0, "vkr_create_mesh: mesh missing");
    for (int64_t i = 0; i < (int64_t)n_tris * 3; i++)
        VKR_REQUIRE(indices[i] < (uint32_t)n_verts, "vkr_create_mesh: index %u at %lld is outside the %d vertices", indices[i], (long long)i, n_verts);
    vkr_ctx::Mesh m = {nullptr, nullptr, n_verts, n_tris, nullptr, nullptr, nullptr};
    VKR_HIP_CHECK(hipMalloc((void **)&m.positions, (size_t)n_verts * 12));
    hipError_t e = hipMalloc((void **)&m.indices, (size_t)n_tris * 12);
    int32_t rc = e == hipSuccess ? upload(c, m.positions, positions, (size_t)n_verts * 12) : fail(VKR_ERR_HIP, "vkr_create_mesh: %s", hipGetErrorString(e));
    if (rc == VKR_OK) rc = upload(c, m.indices, indices, (size_t)n_tris * 12);
    if (rc != VKR_OK) { (void)hipFree(m.positions); if (m.indices) (void)hipFree(m.indices); return rc; }
    c->meshes.push_back(m);
    *mesh_out = (int32_t)c->meshes.size() - 1;
    return VKR_OK;
}

extern "C" int32_t vkr_set_mesh_attributes(vkr_ctx *c, int32_t mesh, const float *texcoords, const float *normals, const float *tangents)
{
    VKR_CTX(c);
    VKR_REQUIRE(mesh >= 0 && mesh < (int32_t)c->meshes.size(), "vkr_set_mesh_attributes: unknown mesh %d", mesh);
    VKR_REQUIRE(texcoords && normals && tangents, "vkr_set_mesh_attributes: an attribute array is null");
    vkr_ctx::Mesh &m = c->meshes[mesh];
    const size_t n = (size_t)m.n_verts;
    if (!m.texcoords) {  // all three arrays or none: `texcoords` is what marks a mesh as shadeable
        float *t = nullptr, *nn = nullptr, *tg = nullptr;
        if (hipMalloc((void **)&t, n * 8) != hipSuccess || hipMalloc((void **)&nn, n * 12) != hipSuccess || hipMalloc((void **)&tg, n * 12) != hipSuccess) {
            (void)hipGetLastError();
            if (t) (void)hipFree(t);
            if (nn) (void)hipFree(nn);
            return fail(VKR_ERR_HIP, "vkr_set_mesh_attributes: out of device memory");
        }
        m.texcoords = t; m.normals = nn; m.tangents = tg;
    }
    int32_t rc = upload(c, m.texcoords, texcoords, n * 8);
    if (rc == VKR_OK) rc = upload(c, m.normals, normals, n * 12);
    if (rc == VKR_OK) rc = upload(c, m.tangents, tangents, n * 12);
    return rc;
}

extern "C" int32_t vkr_create_material(vkr_ctx *c, const void *albedo, int32_t aw, int32_t ah, const void *normal, int32_t nw, int32_t nh,
                                       const void *arm, int32_t mw, int32_t mh, int32_t *material_out)
{
    VKR_CTX(c);
    VKR_REQUIRE(albedo && normal && arm && material_out, "vkr_create_material: an image is null");
    VKR_REQUIRE(aw > 0 && ah > 0 && nw > 0 && nh > 0 && mw > 0 && mh > 0, "vkr_create_material: bad image size");
    vkr_ctx::Material m;
    memset(&m, 0, sizeof(m));
    const void *src[3] = {albedo, normal, arm};
    VkrTexture *dst[3] = {&m.albedo, &m.normal, &m.arm};
    const int w[3] = {aw, nw, mw}, h[3] = {ah, nh, mh};
    int32_t rc = VKR_OK;
    for (int k = 0; k < 3 && rc == VKR_OK; k++) {
        void *dev = nullptr;
        hipError_t e = hipMalloc(&dev, (size_t)w[k] * h[k] * 4);
        if (e != hipSuccess) { rc = fail(VKR_ERR_HIP, "vkr_create_material: %s", hipGetErrorString(e)); break; }
        dst[k]->rgba8 = dev; dst[k]->width = w[k]; dst[k]->height = h[k];
        rc = upload(c, dev, src[k], (size_t)w[k] * h[k] * 4);
    }
    if (rc != VKR_OK) {
        for (VkrTexture *t : dst) if (t->rgba8) (void)hipFree((void *)t->rgba8);
        return rc;
    }
    c->materials.push_back(m);
    *material_out = (int32_t)c->materials.size() - 1;
    return VKR_OK;
}

extern "C" int32_t vkr_draw_model(vkr_ctx *c, int32_t mesh, int32_t material, const float *model16)
{
    VKR_CTX(c);
    VKR_REQUIRE(mesh >= 0 && mesh < (int32_t)c->meshes.size(), "vkr_draw_model: unknown mesh %d", mesh);
    VKR_REQUIRE(material >= -1 && material < (int32_t)c->materials.size(), "vkr_draw_model: unknown material %d", material);
    VKR_REQUIRE(model16 != nullptr, "vkr_draw_model: model matrix is null");
    if (!c->in_frame) return fail(VKR_ERR_ORDER, "vkr_draw_model: call between BeginFrame and EndFrame");
    const vkr_ctx::Mesh &m = c->meshes[mesh];
    VkrMeshDraw d;
    d.positions = m.positions; d.n_verts = m.n_verts; d.indices = m.indices; d.n_tris = m.n_tris;
    memcpy(d.model, model16, sizeof(d.model));
    c->draw_calls.push_back(d);
    c->draw_mesh.push_back(mesh); c->draw_material.push_back(material);
    return VKR_OK;
}

extern "C" int32_t vkr_upload_gbuffer(vkr_ctx *c, const void *albedo, const void *normal, const void *arm, const void *position)
{
    VKR_CTX(c);
    VKR_REQUIRE(albedo && normal && arm && position, "vkr_upload_gbuffer: an image is null");
    Frame &f = c->frames[c->frame];
    const void *src[4] = {albedo, normal, arm, position};
    for (int i = 0; i < 4; i++) {
        int32_t rc = upload(c, f.gbuf[i], src[i], rgba16f_bytes(c->cfg.width, c->cfg.height));
        if (rc != VKR_OK) return rc;
    }
    return VKR_OK;
}

extern "C" int32_t vkr_upload_background(vkr_ctx *c, const void *img)
{
    VKR_CTX(c);
    Frame &f = c->frames[c->frame];
    f.has_background = img != nullptr;
    if (!img) return VKR_OK;
    return upload(c, f.background, img, rgba16f_bytes(c->cfg.width, c->cfg.height));
}

extern "C" int32_t vkr_upload_dir_shadow(vkr_ctx *c, const float *depth)
{
    VKR_CTX(c);
    VKR_REQUIRE(depth != nullptr, "vkr_upload_dir_shadow: depth is null");
    return upload(c, c->frames[c->frame].dir_shadow, depth, (size_t)c->cfg.dir_shadow_res * c->cfg.dir_shadow_res * 4);
}

extern "C" int32_t vkr_upload_point_shadow(vkr_ctx *c, int32_t light, const float *depth_cube)
{
    VKR_CTX(c);
    VKR_REQUIRE(depth_cube != nullptr, "vkr_upload_point_shadow: depth cube is null");
    VKR_REQUIRE(light >= 0 && light < c->cfg.max_lights, "vkr_upload_point_shadow: light %d outside [0,%d)", light, c->cfg.max_lights);
    size_t per = (size_t)6 * c->cfg.point_shadow_res * c->cfg.point_shadow_res;
    float *dst = c->frames[c->frame].cube_shadow + per * light;
    c->frames[c->frame].ranges_lights = 0;
    if (!c->cubes_tiled) return upload(c, dst, depth_cube, per * 4);
    // host rows -> staging -> 8x4-texel tiles (what the shade kernel's gathers are laid out for)
    VKR_HIP_CHECK(hipMemcpyAsync(c->cube_staging, depth_cube, per * 4, hipMemcpyHostToDevice, c->stream));
    int32_t rc = vkr_relayout_depth_cubes(c->cube_staging, dst, 6, c->cfg.point_shadow_res, VKR_CUBE_LAYOUT_TILED_8X4_X25, c->stream);
    if (rc != VKR_OK) return rc;
    VKR_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VKR_OK;
}

extern "C" int32_t vkr_create_skybox_cubemap(vkr_ctx *c, const void *equirect, int32_t w, int32_t h, int32_t src_format)
{
    VKR_CTX(c);
    VKR_REQUIRE(equirect != nullptr && w > 0 && h > 0, "vkr_create_skybox_cubemap: bad source image");
    VKR_REQUIRE(src_format == VKR_SRC_RGBA8_SRGB || src_format == VKR_SRC_RGBA16F, "vkr_create_skybox_cubemap: unknown source format %d", src_format);
    size_t bytes = (size_t)w * h * (src_format == VKR_SRC_RGBA8_SRGB ? 4 : 8);
    void *src = nullptr;
    VKR_HIP_CHECK(hipMalloc(&src, bytes));
    int32_t rc = upload(c, src, equirect, bytes);
    if (rc == VKR_OK) rc = vkr_ibl_equirect_to_cube(src, w, h, src_format, c->env, c->cfg.env_res, 0, 6 * c->cfg.env_res, c->stream);
    if (rc == VKR_OK && c->env_tiled) rc = vkr_tile_rgba16f_faces(c->env, c->env_tiled, 6, c->cfg.env_res, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);  // the reference waits with QueueWaitIdle (VulkanRenderer.cs:714)
    (void)hipFree(src);
    if (rc != VKR_OK) return rc;
    if (e != hipSuccess) return fail(VKR_ERR_HIP, "vkr_create_skybox_cubemap: %s", hipGetErrorString(e));
    c->have_env = true;
    c->env_foot_valid = false;
    return VKR_OK;
}

extern "C" int32_t vkr_upload_env_cube(vkr_ctx *c, const void *cube)
{
    VKR_CTX(c);
    VKR_REQUIRE(cube != nullptr, "vkr_upload_env_cube: cube is null");
    int32_t rc = upload(c, c->env, cube, cube_bytes(c->cfg.env_res));
    if (rc == VKR_OK && c->env_tiled) {
        rc = vkr_tile_rgba16f_faces(c->env, c->env_tiled, 6, c->cfg.env_res, c->stream);
        if (rc == VKR_OK) VKR_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    if (rc == VKR_OK) c->have_env = true;
    c->env_foot_valid = false;
    return rc;
}

// The environment in footprint tiles (every 2x2 footprint of a LINEAR lookup in one cache line): the copy the
// irradiance pass in its default order and the prefilter passes read.  Made when first asked for after the
// environment changed; enqueued on the context's stream.
static int32_t ensure_env_footprints(vkr_ctx *c)
{
    if (c->env_foot_valid) return VKR_OK;
    const int64_t bytes = vkr_env_footprint_tiles_bytes(c->cfg.env_res);
    if (bytes < 0) return VKR_ERR_INVALID;
    if (!c->env_foot) VKR_HIP_CHECK(hipMalloc(&c->env_foot, (size_t)bytes));
    const int32_t rc = vkr_tile_rgba16f_footprints(c->env, c->env_foot, c->cfg.env_res, c->stream);
    if (rc == VKR_OK) c->env_foot_valid = true;
    return rc;
}

extern "C" int32_t vkr_create_irradiance_cubemap(vkr_ctx *c, int32_t order)
{
    VKR_CTX(c);
    if (!c->have_env) return fail(VKR_ERR_ORDER, "vkr_create_irradiance_cubemap: no environment cube yet");
    const void *env = c->env_tiled ? c->env_tiled : c->env;
    const int32_t layout = c->env_tiled ? VKR_ENV_LAYOUT_TILED_4X4 : VKR_ENV_LAYOUT_LINEAR;
    const int32_t n = c->cfg.irradiance_res;
    if (order == VKR_IRRADIANCE_FASTEST) order = VKR_IRRADIANCE_BY_REGION_ROWS;
    if (order == VKR_IRRADIANCE_BY_REGION || order == VKR_IRRADIANCE_BY_REGION_ROWS) {
        // a one-off step: the workspace lives for this call only
        const bool rows = order == VKR_IRRADIANCE_BY_REGION_ROWS;
        const int64_t bytes = rows ? vkr_ibl_irradiance_by_region_rows_workspace_bytes(c->cfg.env_res, n, 0, 6 * n)
                                   : vkr_ibl_irradiance_by_region_workspace_bytes(c->cfg.env_res, n, 0, 6 * n);
        if (bytes < 0) return VKR_ERR_INVALID;
        int32_t rc = rows ? ensure_env_footprints(c) : VKR_OK;
        if (rc != VKR_OK) return rc;
        char *ws = nullptr;
        VKR_HIP_CHECK(hipMalloc((void **)&ws, (size_t)bytes));
        if (rows)
            rc = vkr_ibl_irradiance_by_region_rows(c->env_foot, c->cfg.env_res, c->irradiance, n, 0, 6 * n, ws, bytes, c->stream);
        else
            rc = vkr_ibl_irradiance_by_region(env, c->cfg.env_res, layout, c->irradiance, n, 0, 6 * n, ws, bytes, c->stream);
        uint32_t uncovered = 0;
        hipError_t e = hipStreamSynchronize(c->stream);
        if (rc == VKR_OK && e == hipSuccess) e = hipMemcpy(&uncovered, ws + bytes - 256, 4, hipMemcpyDeviceToHost);
        (void)hipFree(ws);
        if (rc != VKR_OK) return rc;
        if (e != hipSuccess) return fail(VKR_ERR_HIP, "vkr_create_irradiance_cubemap: %s", hipGetErrorString(e));
        if (uncovered != 0) return fail(VKR_ERR_HIP, "vkr_create_irradiance_cubemap: %u texels did not find all of their samples", uncovered);
        c->have_irr = true;
        return VKR_OK;
    }
    int32_t rc = vkr_ibl_irradiance(env, c->cfg.env_res, layout, c->irradiance, n, order, 0, 6 * n, c->stream);
    if (rc != VKR_OK) return rc;
    VKR_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->have_irr = true;
    return VKR_OK;
}

extern "C" int32_t vkr_create_prefiltered_cubemap(vkr_ctx *c)
{
    VKR_CTX(c);
    if (!c->have_env) return fail(VKR_ERR_ORDER, "vkr_create_prefiltered_cubemap: no environment cube yet");
    int32_t rc = ensure_env_footprints(c);
    if (rc != VKR_OK) return rc;
    VkrPrefilterMipsArgs a;
    memset(&a, 0, sizeof(a));
    a.env = c->env_foot; a.env_res = c->cfg.env_res; a.env_layout = VKR_ENV_LAYOUT_FOOTPRINT_TILES;
    a.base_res = c->cfg.prefilter_res; a.mips = c->cfg.prefilter_mips;  // roughness of mip m = m / (mips - 1), VulkanRenderer.cs:874
    for (int m = 0; m < a.mips; m++) { a.out[m] = c->prefilter[m]; a.row_begin[m] = 0; a.row_end[m] = 6 * (c->cfg.prefilter_res >> m); }
    if ((rc = vkr_ibl_prefilter_mips(&a, c->stream)) != VKR_OK) return rc;
    VKR_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->have_pref = true;
    return VKR_OK;
}

extern "C" int32_t vkr_create_brdf_lut(vkr_ctx *c)
{
    VKR_CTX(c);
    int32_t rc = vkr_ibl_brdf_lut(c->brdf_lut, c->cfg.brdf_lut_res, 0, c->cfg.brdf_lut_res, c->stream);
    if (rc != VKR_OK) return rc;
    VKR_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->have_lut = true;
    return VKR_OK;
}

extern "C" int32_t vkr_pass_composition(vkr_ctx *c)
{
    VKR_CTX(c);
    VkrShadeArgs a;
    int32_t rc = fill_shade_args(c, c->frames[c->frame], a);
    if (rc != VKR_OK) return rc;
    if ((rc = run_shadow_passes(c, c->frames[c->frame])) != VKR_OK) return rc;
    if ((rc = run_geometry_pass(c, c->frames[c->frame])) != VKR_OK) return rc;
    if ((rc = run_coverage(c, c->frames[c->frame])) != VKR_OK) return rc;
    if ((rc = run_skybox(c, c->frames[c->frame])) != VKR_OK) return rc;
    return shade(c, c->frames[c->frame], a);
}

extern "C" int32_t vkr_pass_bloom(vkr_ctx *c, int32_t horizontal)
{
    VKR_CTX(c);
    Frame &f = c->frames[c->frame];
    VkrBloomArgs b = bloom_args(c, f, horizontal != 0);
    int32_t rc = vkr_bloom_blur(&b, c->stream);
    if (rc == VKR_OK) (horizontal ? f.bloom1_valid : f.bloom2_valid) = true;
    return rc;
}

extern "C" int32_t vkr_pass_postprocess(vkr_ctx *c)
{
    VKR_CTX(c);
    Frame &f = c->frames[c->frame];
    if (!f.bloom2_valid) return fail(VKR_ERR_ORDER, "vkr_pass_postprocess: bloom pass 2 has not run for this frame");
    return vkr_postprocess(f.color, f.bloom2, f.ldr, c->cfg.width, c->cfg.height, c->cfg.flags & VKR_POST_SRGB_TARGET, c->stream);
}

// ---- RenderStage mirror --------------------------------------------------------------------------------------
// The reference records every pass of a frame into a RenderStage (RenderStage.cs:99-190: BeginCommands,
// BeginRenderPass, the draws, EndRenderPass, EndCommands) and submits the command buffers at the end of EndFrame
// (VulkanRenderer.cs:559-595).  Here a stage remembers what was recorded; vkr_submit_frame runs the recorded
// stages on the context's stream in the reference's submission order.

static const char *const STAGE_NAMES[VKR_STAGE_COUNT] = {"point shadow", "geometry", "directional shadow", "composition", "bloom 1", "bloom 2", "post-process"};

static int32_t stage_of(vkr_ctx *c, int32_t stage, int32_t frame, const char *fn, vkr_ctx::Stage **out)
{
    if (stage < 0 || stage >= VKR_STAGE_COUNT) return fail(VKR_ERR_INVALID, "%s: unknown stage %d", fn, stage);
    if (frame != c->frame) return fail(VKR_ERR_INVALID, "%s: command buffer %d is not the current frame's (%d)", fn, frame, c->frame);
    *out = &c->stages[stage];
    return VKR_OK;
}

extern "C" int32_t vkr_stage_reset_commands(vkr_ctx *c, int32_t stage, int32_t frame)
{
    VKR_CTX(c);
    vkr_ctx::Stage *st;
    int32_t rc = stage_of(c, stage, frame, __func__, &st);
    if (rc != VKR_OK) return rc;
    *st = vkr_ctx::Stage();
    return VKR_OK;
}

extern "C" int32_t vkr_stage_begin_commands(vkr_ctx *c, int32_t stage, int32_t frame)
{
    VKR_CTX(c);
    vkr_ctx::Stage *st;
    int32_t rc = stage_of(c, stage, frame, __func__, &st);
    if (rc != VKR_OK) return rc;
    if (st->state != vkr_ctx::STAGE_INITIAL && st->state != vkr_ctx::STAGE_EXECUTABLE)
        return fail(VKR_ERR_ORDER, "Failed to begin command buffer! (the %s stage is already recording)", STAGE_NAMES[stage]);  // RenderStage.cs:108
    *st = vkr_ctx::Stage();  // beginning an executable command buffer resets it
    st->state = vkr_ctx::STAGE_RECORDING;
    return VKR_OK;
}

extern "C" int32_t vkr_stage_begin_render_pass(vkr_ctx *c, int32_t stage, int32_t frame)
{
    VKR_CTX(c);
    vkr_ctx::Stage *st;
    int32_t rc = stage_of(c, stage, frame, __func__, &st);
    if (rc != VKR_OK) return rc;
    if (st->state != vkr_ctx::STAGE_RECORDING)
        return fail(VKR_ERR_ORDER, "vkr_stage_begin_render_pass: the %s stage is not recording (BeginCommands first, one render pass per command buffer)", STAGE_NAMES[stage]);
    st->state = vkr_ctx::STAGE_IN_PASS;
    st->cleared = true;  // every attachment of the reference's passes is loaded with its clear value
    return VKR_OK;
}

extern "C" int32_t vkr_stage_draw(vkr_ctx *c, int32_t stage, int32_t frame)
{
    VKR_CTX(c);
    vkr_ctx::Stage *st;
    int32_t rc = stage_of(c, stage, frame, __func__, &st);
    if (rc != VKR_OK) return rc;
    if (st->state != vkr_ctx::STAGE_IN_PASS)
        return fail(VKR_ERR_ORDER, "vkr_stage_draw: the %s stage is not inside a render pass", STAGE_NAMES[stage]);
    st->drawn = true;
    return VKR_OK;
}

extern "C" int32_t vkr_stage_end_render_pass(vkr_ctx *c, int32_t stage, int32_t frame)
{
    VKR_CTX(c);
    vkr_ctx::Stage *st;
    int32_t rc = stage_of(c, stage, frame, __func__, &st);
    if (rc != VKR_OK) return rc;
    if (st->state != vkr_ctx::STAGE_IN_PASS)
        return fail(VKR_ERR_ORDER, "vkr_stage_end_render_pass: the %s stage is not inside a render pass", STAGE_NAMES[stage]);
    st->state = vkr_ctx::STAGE_PASS_ENDED;
    return VKR_OK;
}

extern "C" int32_t vkr_stage_end_commands(vkr_ctx *c, int32_t stage, int32_t frame)
{
    VKR_CTX(c);
    vkr_ctx::Stage *st;
    int32_t rc = stage_of(c, stage, frame, __func__, &st);
    if (rc != VKR_OK) return rc;
    if (st->state != vkr_ctx::STAGE_RECORDING && st->state != vkr_ctx::STAGE_PASS_ENDED)
        return fail(VKR_ERR_ORDER, "Failed to end command buffer! (the %s stage is %s)", STAGE_NAMES[stage],  // RenderStage.cs:120
                    st->state == vkr_ctx::STAGE_IN_PASS ? "inside a render pass" : "not recording");
    st->state = vkr_ctx::STAGE_EXECUTABLE;
    return VKR_OK;
}

// A render pass that was begun and ended without draws leaves its attachments at their clear values.
static int32_t clear_stage_attachments(vkr_ctx *c, Frame &f, int32_t stage)
{
    const size_t px = (size_t)c->cfg.width * c->cfg.height;
    int32_t rc = VKR_OK;
    switch (stage) {
    case VKR_STAGE_POINT_SHADOW: {
        const size_t cube = (size_t)c->cfg.max_lights * 6 * c->cfg.point_shadow_res * c->cfg.point_shadow_res;  // even: 6 faces
        f.ranges_lights = 0;
        return fill_u64(f.cube_shadow, cube / 2, c->cubes_tiled ? D32F_25_PAIR : D32F_ONE_PAIR, c->stream);
    }
    case VKR_STAGE_DIR_SHADOW: {
        const size_t dir = (size_t)c->cfg.dir_shadow_res * c->cfg.dir_shadow_res;
        rc = fill_u64(f.dir_shadow, dir / 2, D32F_ONE_PAIR, c->stream);
        const float one = 1.0f;
        if (rc == VKR_OK && (dir & 1)) VKR_HIP_CHECK(hipMemcpyAsync(f.dir_shadow + dir - 1, &one, 4, hipMemcpyHostToDevice, c->stream));
        return rc;
    }
    case VKR_STAGE_GEOMETRY:
        rc = fill_u64(f.gbuf[0], px, 0ull, c->stream);
        for (int k = 1; k < 4 && rc == VKR_OK; k++) rc = fill_u64(f.gbuf[k], px, RGBA16F_0001, c->stream);
        return rc;
    case VKR_STAGE_COMPOSITION:
        rc = fill_u64(f.color, px, RGBA16F_0001, c->stream);
        return rc != VKR_OK ? rc : fill_u64(f.threshold, px, RGBA16F_0001, c->stream);
    case VKR_STAGE_BLOOM1: f.bloom1_valid = true; return fill_u64(f.bloom1, px, RGBA16F_0001, c->stream);
    case VKR_STAGE_BLOOM2: f.bloom2_valid = true; return fill_u64(f.bloom2, px, RGBA16F_0001, c->stream);
    default:
        rc = fill_u64(f.ldr, px / 2, 0xFF000000FF000000ull, c->stream);
        if (rc == VKR_OK && (px & 1)) VKR_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)((char *)f.ldr + (px - 1) * 4), (int)0xFF000000u, 1, c->stream));
        return rc;
    }
}

static int32_t run_stage(vkr_ctx *c, Frame &f, int32_t stage)
{
    int32_t rc;
    switch (stage) {
    // A frame without draw calls records zero vkCmdDraw calls into these three passes (their loops over
    // solidModelDrawCalls are empty, VulkanRenderer.cs:357-366, 391-400, 417-424): begun and ended, they clear.
    case VKR_STAGE_POINT_SHADOW: return c->draw_calls.empty() ? clear_stage_attachments(c, f, stage) : run_point_shadows(c, f);
    case VKR_STAGE_GEOMETRY: return c->draw_calls.empty() ? clear_stage_attachments(c, f, stage) : run_geometry_pass(c, f);
    case VKR_STAGE_DIR_SHADOW: return c->draw_calls.empty() ? clear_stage_attachments(c, f, stage) : run_dir_shadow(c, f);
    case VKR_STAGE_COMPOSITION: {  // skybox draw, composition draw, one light-volume draw per light
        VkrShadeArgs a;
        if ((rc = fill_shade_args(c, f, a)) != VKR_OK) return rc;
        if ((rc = run_coverage(c, f)) != VKR_OK) return rc;
        if ((rc = run_skybox(c, f)) != VKR_OK) return rc;
        return shade(c, f, a);
    }
    case VKR_STAGE_BLOOM1: return vkr_pass_bloom(c, 1);
    case VKR_STAGE_BLOOM2: return vkr_pass_bloom(c, 0);
    default: return vkr_pass_postprocess(c);
    }
}

extern "C" int32_t vkr_submit_frame(vkr_ctx *c)
{
    VKR_CTX(c);
    if (!c->in_frame) return fail(VKR_ERR_ORDER, "Tried to end frame before beginning a new one!");  // VulkanRenderer.cs:322
    if (c->cfg.frame_rows != c->cfg.height) return fail(VKR_ERR_UNSUPPORTED, "vkr_submit_frame: stage recording is for contexts that render a whole frame; a strip ends its frame with vkr_strip_shade / vkr_strip_finish");
    Frame &f = c->frames[c->frame];
    for (int s = 0; s < VKR_STAGE_COUNT; s++)
        if (c->stages[s].state != vkr_ctx::STAGE_INITIAL && c->stages[s].state != vkr_ctx::STAGE_EXECUTABLE)
            return fail(VKR_ERR_ORDER, "Unable to submit graphics queue: the %s stage is still recording", STAGE_NAMES[s]);  // VulkanRenderer.cs:593
    // the stages are numbered in the reference's submission order (VulkanRenderer.cs:563-584)
    for (int s = 0; s < VKR_STAGE_COUNT; s++) {
        const vkr_ctx::Stage &st = c->stages[s];
        if (st.state != vkr_ctx::STAGE_EXECUTABLE || !st.cleared) continue;  // nothing recorded: the stage's images stay as they are
        const int32_t rc = st.drawn ? run_stage(c, f, s) : clear_stage_attachments(c, f, s);
        if (rc != VKR_OK) return rc;
    }
    c->timing_valid = false;
    c->in_frame = false;
    c->frame = (c->frame + 1) % c->cfg.frames_in_flight;  // VulkanRenderer.cs:622
    return VKR_OK;
}

// ---- strips: a context that owns rows [row0, row0 + height) of a frame ---------------------------------------

extern "C" int32_t vkr_strip_plan(int32_t frame_rows, int32_t n_strips, int32_t k, int32_t *row0, int32_t *rows)
{
    VKR_REQUIRE(row0 && rows, "vkr_strip_plan: null output");
    VKR_REQUIRE(n_strips > 0 && k >= 0 && k < n_strips, "vkr_strip_plan: strip %d of %d", k, n_strips);
    const int32_t base = frame_rows / n_strips, extra = frame_rows % n_strips;  // the first `extra` strips are one row taller
    VKR_REQUIRE(n_strips == 1 ? frame_rows > 0 : base >= VKR_STRIP_HALO_ROWS, "vkr_strip_plan: %d rows over %d strips leaves strips of fewer than %d rows", frame_rows, n_strips, VKR_STRIP_HALO_ROWS);
    *row0 = k * base + (k < extra ? k : extra);
    *rows = base + (k < extra ? 1 : 0);
    return VKR_OK;
}

static bool has_neighbour(const vkr_ctx *c, int side) { return side == 0 ? c->cfg.row0 > 0 : c->cfg.row0 + c->cfg.height < c->cfg.frame_rows; }

extern "C" int32_t vkr_strip_shade(vkr_ctx *c)
{
    VKR_CTX(c);
    if (!c->in_frame) return fail(VKR_ERR_ORDER, "Tried to end frame before beginning a new one!");  // VulkanRenderer.cs:322
    if (c->strip_shaded) return fail(VKR_ERR_ORDER, "vkr_strip_shade: this frame's composition pass is already enqueued; vkr_strip_finish comes next");
    Frame &f = c->frames[c->frame];
    VkrShadeArgs a;
    int32_t rc = fill_shade_args(c, f, a);
    if (rc != VKR_OK) return rc;
    VKR_HIP_CHECK(hipEventRecord(c->ev[0], c->stream));
    if ((rc = run_shadow_passes(c, f)) != VKR_OK) return rc;   // EndFrame's order: point shadows, geometry, directional shadow
    if ((rc = run_geometry_pass(c, f)) != VKR_OK) return rc;
    if ((rc = run_coverage(c, f)) != VKR_OK) return rc;  // counted with the composition pass in the timing, like the skybox
    if ((rc = run_skybox(c, f)) != VKR_OK) return rc;
    if ((rc = shade(c, f, a)) != VKR_OK) return rc;
    VKR_HIP_CHECK(hipEventRecord(c->ev[1], c->stream));
    VKR_HIP_CHECK(hipEventRecord(c->ev_shaded, c->stream));
    c->strip_shaded = true;
    return VKR_OK;
}

extern "C" int32_t vkr_strip_halo(vkr_ctx *c, int32_t side, const void **send, void **recv, int64_t *bytes)
{
    VKR_CTX(c);
    VKR_REQUIRE(side == VKR_STRIP_ABOVE || side == VKR_STRIP_BELOW, "vkr_strip_halo: side %d", side);
    VKR_REQUIRE(has_neighbour(c, side), "vkr_strip_halo: the strip has no neighbour %s it (it touches the frame edge)", side == VKR_STRIP_ABOVE ? "above" : "below");
    Frame &f = c->frames[c->frame];
    const size_t hb = halo_bytes(c), rows_b = rgba16f_bytes(c->cfg.width, c->cfg.height);
    char *base = (char *)f.threshold_padded;
    if (send) *send = side == VKR_STRIP_ABOVE ? base + hb : base + rows_b;        // the strip's first / last rows
    if (recv) *recv = side == VKR_STRIP_ABOVE ? base : base + hb + rows_b;         // the rows beyond them
    if (bytes) *bytes = (int64_t)hb;
    return VKR_OK;
}

extern "C" int32_t vkr_strip_exchange(vkr_ctx *upper, vkr_ctx *lower)
{
    VKR_REQUIRE(upper && lower && upper != lower, "vkr_strip_exchange: two different contexts are needed");
    VKR_REQUIRE(upper->cfg.width == lower->cfg.width && upper->cfg.frame_rows == lower->cfg.frame_rows &&
                upper->cfg.row0 + upper->cfg.height == lower->cfg.row0,
                "vkr_strip_exchange: rows [%d,%d) and [%d,%d) of frames %d / %d rows tall, %d / %d wide are not neighbouring strips of one frame",
                upper->cfg.row0, upper->cfg.row0 + upper->cfg.height, lower->cfg.row0, lower->cfg.row0 + lower->cfg.height,
                upper->cfg.frame_rows, lower->cfg.frame_rows, upper->cfg.width, lower->cfg.width);
    if (!upper->strip_shaded || !lower->strip_shaded) return fail(VKR_ERR_ORDER, "vkr_strip_exchange: both strips need vkr_strip_shade for the frame first");
    const void *up_send, *lo_send;
    void *up_recv, *lo_recv;
    int64_t bytes = 0;
    int32_t rc = vkr_strip_halo(upper, VKR_STRIP_BELOW, &up_send, &up_recv, &bytes);
    if (rc == VKR_OK) rc = vkr_strip_halo(lower, VKR_STRIP_ABOVE, &lo_send, &lo_recv, &bytes);
    if (rc != VKR_OK) return rc;
    const int du = upper->cfg.device, dl = lower->cfg.device;
    if (du != dl) {  // direct copies over xGMI where the devices allow it; otherwise the runtime stages them
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dl, du) == hipSuccess && can) { VKR_HIP_CHECK(hipSetDevice(dl)); (void)hipDeviceEnablePeerAccess(du, 0); (void)hipGetLastError(); }
        if (hipDeviceCanAccessPeer(&can, du, dl) == hipSuccess && can) { VKR_HIP_CHECK(hipSetDevice(du)); (void)hipDeviceEnablePeerAccess(dl, 0); (void)hipGetLastError(); }
    }
    // each strip pulls its neighbour's rows on its own stream once the neighbour's composition pass is done ...
    VKR_HIP_CHECK(hipSetDevice(dl));
    VKR_HIP_CHECK(hipStreamWaitEvent(lower->stream, upper->ev_shaded, 0));
    VKR_HIP_CHECK(hipMemcpyPeerAsync(lo_recv, dl, up_send, du, (size_t)bytes, lower->stream));
    VKR_HIP_CHECK(hipEventRecord(lower->ev_pulled, lower->stream));
    VKR_HIP_CHECK(hipSetDevice(du));
    VKR_HIP_CHECK(hipStreamWaitEvent(upper->stream, lower->ev_shaded, 0));
    VKR_HIP_CHECK(hipMemcpyPeerAsync(up_recv, du, lo_send, dl, (size_t)bytes, upper->stream));
    VKR_HIP_CHECK(hipEventRecord(upper->ev_pulled, upper->stream));
    // ... and neither goes on (and later overwrites the rows it lent) before the other has them
    VKR_HIP_CHECK(hipStreamWaitEvent(upper->stream, lower->ev_pulled, 0));
    VKR_HIP_CHECK(hipSetDevice(dl));
    VKR_HIP_CHECK(hipStreamWaitEvent(lower->stream, upper->ev_pulled, 0));
    return VKR_OK;
}

extern "C" int32_t vkr_strip_finish(vkr_ctx *c)
{
    VKR_CTX(c);
    if (!c->in_frame || !c->strip_shaded) return fail(VKR_ERR_ORDER, "vkr_strip_finish: vkr_strip_shade has not run for this frame");
    Frame &f = c->frames[c->frame];
    VKR_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
    // both bloom passes and the post-process draw run as one kernel; neither bloom image is stored.  The source is
    // the thresholded image with the neighbours' rows around it: the frame's edges clamp, the strip's edges do not.
    const int H = VKR_STRIP_HALO_ROWS;
    VkrBloomChainArgs bc;
    memset(&bc, 0, sizeof(bc));
    bc.threshold = f.threshold_padded; bc.hdr = f.color; bc.out_rgba8 = f.ldr;
    bc.width = c->cfg.width; bc.rows = c->cfg.height; bc.src_rows = c->cfg.height + 2 * H; bc.src_row_offset = H;
    bc.wrap_h = VKR_WRAP_REPEAT; bc.clamp_lo_h = 0; bc.clamp_hi_h = c->cfg.width - 1;   // textureSampler along x
    bc.wrap_v = VKR_WRAP_CLAMP;                                                          // bloomSampler along y
    bc.clamp_lo_v = has_neighbour(c, VKR_STRIP_ABOVE) ? 0 : H;
    bc.clamp_hi_v = has_neighbour(c, VKR_STRIP_BELOW) ? c->cfg.height + 2 * H - 1 : H + c->cfg.height - 1;
    bc.flags = c->cfg.flags & VKR_POST_SRGB_TARGET;
    int32_t rc = vkr_bloom_postprocess(&bc, c->stream);
    if (rc != VKR_OK) return rc;
    VKR_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
    f.bloom1_valid = f.bloom2_valid = false;
    c->timing_valid = true;
    c->strip_shaded = false;
    c->in_frame = false;
    c->frame = (c->frame + 1) % c->cfg.frames_in_flight;  // VulkanRenderer.cs:622
    return VKR_OK;
}

extern "C" int32_t vkr_end_frame(vkr_ctx *c)
{
    VKR_CTX(c);
    if (!c->in_frame) return fail(VKR_ERR_ORDER, "Tried to end frame before beginning a new one!");  // VulkanRenderer.cs:322
    if (c->cfg.frame_rows != c->cfg.height)
        return fail(VKR_ERR_ORDER, "vkr_end_frame: the context renders a strip of a frame: vkr_strip_shade, the halo rows, then vkr_strip_finish");
    const int32_t rc = vkr_strip_shade(c);
    return rc != VKR_OK ? rc : vkr_strip_finish(c);
}

extern "C" int32_t vkr_read(vkr_ctx *c, int32_t frame, int32_t target, int32_t mip, void *host, int64_t host_bytes)
{
    VKR_CTX(c);
    VKR_REQUIRE(host != nullptr, "vkr_read: host buffer is null");
    const void *src = nullptr;
    size_t bytes = 0;
    const size_t img = rgba16f_bytes(c->cfg.width, c->cfg.height);
    if (target <= VKR_TARGET_LDR) {
        VKR_REQUIRE(frame >= 0 && frame < c->cfg.frames_in_flight, "vkr_read: frame %d outside [0,%d)", frame, c->cfg.frames_in_flight);
        Frame &f = c->frames[frame];
        switch (target) {
        case VKR_TARGET_COLOR: src = f.color; bytes = img; break;
        case VKR_TARGET_THRESHOLD: src = f.threshold; bytes = img; break;
        case VKR_TARGET_BLOOM1:
            if (!f.bloom1_valid) return fail(VKR_ERR_ORDER, "vkr_read: bloom-1 is only stored by vkr_pass_bloom(ctx, 1)");
            src = f.bloom1; bytes = img; break;
        case VKR_TARGET_BLOOM2:
            if (!f.bloom2_valid) return fail(VKR_ERR_ORDER, "vkr_read: bloom-2 is only stored by vkr_pass_bloom(ctx, 0)");
            src = f.bloom2; bytes = img; break;
        case VKR_TARGET_LDR: src = f.ldr; bytes = (size_t)c->cfg.width * c->cfg.height * 4; break;
        default: return fail(VKR_ERR_INVALID, "vkr_read: unknown target %d", target);
        }
    } else {
        switch (target) {
        case VKR_TARGET_ENV: src = c->env; bytes = cube_bytes(c->cfg.env_res); break;
        case VKR_TARGET_IRRADIANCE: src = c->irradiance; bytes = cube_bytes(c->cfg.irradiance_res); break;
        case VKR_TARGET_PREFILTER:
            VKR_REQUIRE(mip >= 0 && mip < c->cfg.prefilter_mips, "vkr_read: mip %d outside [0,%d)", mip, c->cfg.prefilter_mips);
            src = c->prefilter[mip]; bytes = cube_bytes(c->cfg.prefilter_res >> mip); break;
        case VKR_TARGET_BRDF_LUT: src = c->brdf_lut; bytes = (size_t)c->cfg.brdf_lut_res * c->cfg.brdf_lut_res * 4; break;
        default: return fail(VKR_ERR_INVALID, "vkr_read: unknown target %d", target);
        }
    }
    VKR_REQUIRE(host_bytes >= 0 && (size_t)host_bytes >= bytes, "vkr_read: host buffer holds %lld bytes, target needs %zu", (long long)host_bytes, bytes);
    VKR_HIP_CHECK(hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, c->stream));
    VKR_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VKR_OK;
}

extern "C" int32_t vkr_sync(vkr_ctx *c)
{
    VKR_CTX(c);
    VKR_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VKR_OK;
}

extern "C" int32_t vkr_get_timing(vkr_ctx *c, VkrTiming *out)
{
    VKR_CTX(c);
    VKR_REQUIRE(out != nullptr, "vkr_get_timing: out is null");
    if (!c->timing_valid) return fail(VKR_ERR_ORDER, "vkr_get_timing: no frame has ended yet");
    VKR_HIP_CHECK(hipEventSynchronize(c->ev[3]));
    VKR_HIP_CHECK(hipEventElapsedTime(&out->shade_ms, c->ev[0], c->ev[1]));
    VKR_HIP_CHECK(hipEventElapsedTime(&out->bloom_h_ms, c->ev[1], c->ev[2]));
    VKR_HIP_CHECK(hipEventElapsedTime(&out->bloom_v_post_ms, c->ev[2], c->ev[3]));
    VKR_HIP_CHECK(hipEventElapsedTime(&out->frame_ms, c->ev[0], c->ev[3]));
    return VKR_OK;
}
