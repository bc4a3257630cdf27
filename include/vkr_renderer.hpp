// vkr_renderer.hpp -- C++ host-side mirror of the reference's renderer interface for the shading
// path, over the C ABI of vkr_hip.h.  Header-only; link against libvkr_hip.so.
//
// The reference drives this path from C# (src/Renderer/VulkanRenderer.cs); where its toolchain is
// not available this class plays the same role with the same method names, call order and error
// behaviour (every non-success status becomes an exception, as the reference throws on any
// non-Success VkResult):
//
//   UpdateSceneInfo(info)        VulkanRenderer.cs:635
//   Lights                       VulkanRenderer.cs:65   (draw order = list order)
//   BeginFrame() / EndFrame()    VulkanRenderer.cs:286 / :320
//   CreateSkyboxCubemap() ...    VulkanRenderer.cs:653, :734, :815, :966
//   WaitIdle(), Dispose (RAII)   as in the reference's IDisposable wrappers
//   HipRenderStage               RenderStage.cs:99-190 (BeginCommands / BeginRenderPass / EndRenderPass / EndCommands),
//                                for a host that keeps EndFrame's per-stage recording; SubmitFrame() is its QueueSubmit
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include "vkr_hip.h"

namespace vkr {

struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string &what) : std::runtime_error(what), code(c) {}
};

inline void check(int32_t rc)
{
    if (rc < 0) throw Error(rc, std::string("libvkr_hip: ") + vkr_last_error());
}

// Light.cs:5-27
struct Light {
    float Position[3];
    float Color[3];

    float CalculateVolumeSize() const  // Light.cs:20-26
    {
        float m = Color[0] > Color[1] ? Color[0] : Color[1];
        m = m > Color[2] ? m : Color[2];
        return 16.0f * std::sqrt(m / 5.0f);
    }
    VkrLightInfo ToInfo() const  // Light.cs:10-18: Model = Scale(volume) * Translate(position), row-vector convention
    {
        VkrLightInfo info{};
        const float r = CalculateVolumeSize();
        info.model[0] = info.model[5] = info.model[10] = r;
        info.model[15] = 1.0f;
        for (int i = 0; i < 3; i++) {
            info.model[12 + i] = Position[i];
            info.position[i] = Position[i];
            info.color[i] = Color[i];
        }
        return info;
    }
};

class HipRenderer;

// RenderStage.cs: one object per pass of the frame.  The five calls have the reference's names and arguments
// (commandBufferIndex = the current frame); Draw() stands for the vkCmdBind* / vkCmdDraw* calls EndFrame records
// between BeginRenderPass and EndRenderPass of that stage.  Nothing runs until HipRenderer::SubmitFrame().
class HipRenderStage {
public:
    void ResetCommandBuffer(uint32_t commandBufferIndex) { check(vkr_stage_reset_commands(ctx_, stage_, (int32_t)commandBufferIndex)); }
    void BeginCommands(uint32_t commandBufferIndex) { check(vkr_stage_begin_commands(ctx_, stage_, (int32_t)commandBufferIndex)); }
    // framebufferIndex is the frame in flight again (the swapchain image for the post-process stage): the
    // renderer object owns one set of attachments per frame in flight, so it selects nothing here
    void BeginRenderPass(uint32_t commandBufferIndex, uint32_t framebufferIndex)
    {
        (void)framebufferIndex;
        check(vkr_stage_begin_render_pass(ctx_, stage_, (int32_t)commandBufferIndex));
    }
    void Draw(uint32_t commandBufferIndex) { check(vkr_stage_draw(ctx_, stage_, (int32_t)commandBufferIndex)); }
    void EndRenderPass(uint32_t commandBufferIndex) { check(vkr_stage_end_render_pass(ctx_, stage_, (int32_t)commandBufferIndex)); }
    void EndCommands(uint32_t commandBufferIndex) { check(vkr_stage_end_commands(ctx_, stage_, (int32_t)commandBufferIndex)); }

private:
    friend class HipRenderer;
    HipRenderStage(vkr_ctx *ctx, int32_t stage) : ctx_(ctx), stage_(stage) {}
    vkr_ctx *ctx_;
    int32_t stage_;
};

class HipRenderer {
public:
    std::vector<Light> Lights;  // VulkanRenderer.cs:65

    explicit HipRenderer(const VkrConfig &config)
    {
        // a library built from another revision of vkr_hip.h would misread the structs this host passes
        if (vkr_abi_version() != VKR_ABI_VERSION)
            throw Error(VKR_ERR_UNSUPPORTED, "libvkr_hip: the loaded library has interface version " + std::to_string(vkr_abi_version()) +
                                             ", this host was compiled against " + std::to_string(VKR_ABI_VERSION));
        check(vkr_create(&config, &ctx_));
    }
    ~HipRenderer() { Dispose(); }
    HipRenderer(const HipRenderer &) = delete;
    HipRenderer &operator=(const HipRenderer &) = delete;

    void Dispose()
    {
        if (ctx_) { vkr_destroy(ctx_); ctx_ = nullptr; }
    }

    void UpdateSceneInfo(const VkrSceneInfo &info) { check(vkr_update_scene_info(ctx_, &info)); }
    void BeginFrame() { check(vkr_begin_frame(ctx_)); }
    void EndFrame()
    {
        PushLights();
        check(vkr_end_frame(ctx_));
    }
    // The stage objects of the reference (VulkanRenderer.cs:80-88): Stage(VKR_STAGE_GEOMETRY) is geometryRenderStage, ...
    HipRenderStage Stage(int32_t stage) { return HipRenderStage(ctx_, stage); }
    // What EndFrame does before and after its stages: the light records go up (they are push constants of the light
    // draws in the reference, VulkanRenderer.cs:485-497), ...
    void PushLights()
    {
        std::vector<VkrLightInfo> infos;
        infos.reserve(Lights.size());
        for (const Light &l : Lights) infos.push_back(l.ToInfo());
        check(vkr_set_lights(ctx_, infos.data(), (int32_t)infos.size(), has_volume_mesh_ ? VKR_COVERAGE_MESH : VKR_COVERAGE_FULLSCREEN));
    }
    // ... and the recorded stages are submitted (VulkanRenderer.cs:559-622): they run in the reference's submission
    // order, then the frame index advances.
    void SubmitFrame() { check(vkr_submit_frame(ctx_)); }
    // One frame on several GPUs: every renderer owns a strip of the frame (VkrConfig.frame_rows / row0) and ends a
    // frame with StripShade(), ExchangeHalo() per neighbouring pair, StripFinish().
    void StripShade() { PushLights(); check(vkr_strip_shade(ctx_)); }
    void StripFinish() { check(vkr_strip_finish(ctx_)); }
    static void ExchangeHalo(HipRenderer &upper, HipRenderer &lower) { check(vkr_strip_exchange(upper.ctx_, lower.ctx_)); }
    // Mesh.cs / VulkanRenderer.DrawModel (VulkanRenderer.cs:630-633): with draw calls in a frame, EndFrame renders
    // the shadow maps from them.
    int32_t CreateMesh(const float *positions, int32_t n_verts, const uint32_t *indices, int32_t n_tris)
    {
        int32_t mesh = -1;
        check(vkr_create_mesh(ctx_, positions, n_verts, indices, n_tris, &mesh));
        return mesh;
    }
    void SetMeshAttributes(int32_t mesh, const float *texcoords, const float *normals, const float *tangents)
    {
        check(vkr_set_mesh_attributes(ctx_, mesh, texcoords, normals, tangents));
    }
    int32_t CreateMaterial(const void *albedo, int32_t aw, int32_t ah, const void *normal, int32_t nw, int32_t nh, const void *arm, int32_t mw, int32_t mh)
    {
        int32_t material = -1;
        check(vkr_create_material(ctx_, albedo, aw, ah, normal, nw, nh, arm, mw, mh, &material));
        return material;
    }
    // material -1: the draw casts shadows only and the G-buffer comes from UploadGBuffer
    void DrawModel(int32_t mesh, const float (&model_matrix)[16], int32_t material = -1) { check(vkr_draw_model(ctx_, mesh, material, model_matrix)); }
    // The reference's sphereMesh (VulkanRenderer.cs:78,269): once set, each light's draw touches only the
    // pixels its volume covers, as the reference's rasterised volumes do.
    void SetLightVolumeMesh(const float *verts, int32_t n_verts, const uint32_t *indices, int32_t n_tris)
    {
        check(vkr_set_light_volume_mesh(ctx_, verts, n_verts, indices, n_tris));
        has_volume_mesh_ = true;
    }
    void WaitIdle() { check(vkr_sync(ctx_)); }
    void Resize(int32_t width, int32_t height) { check(vkr_resize(ctx_, width, height)); }  // RecreateSwapchain
    int32_t CurrentFrame() const { return vkr_current_frame(ctx_); }

    // for a caller that renders the geometry pass, the shadow maps or the skybox itself: the images of those passes
    void UploadGBuffer(const void *albedo, const void *normal, const void *arm, const void *position)
    {
        check(vkr_upload_gbuffer(ctx_, albedo, normal, arm, position));
    }
    void UploadBackground(const void *rgba16f_or_null) { check(vkr_upload_background(ctx_, rgba16f_or_null)); }
    void UploadDirectionalShadowMap(const float *depth) { check(vkr_upload_dir_shadow(ctx_, depth)); }
    void UploadPointShadowMap(int32_t light, const float *depth_cube) { check(vkr_upload_point_shadow(ctx_, light, depth_cube)); }

    // IBL precompute (VulkanRenderer.cs:271-275)
    void CreateSkyboxCubemap(const void *equirect, int32_t w, int32_t h, int32_t format) { check(vkr_create_skybox_cubemap(ctx_, equirect, w, h, format)); }
    void UploadEnvironmentCubemap(const void *cube) { check(vkr_upload_env_cube(ctx_, cube)); }
    // the default order is the library's fastest; it differs from the shader's own running sum (VKR_IRRADIANCE_SEQUENTIAL) in rounding only
    void CreateIrradianceCubemap(int32_t order = VKR_IRRADIANCE_FASTEST) { check(vkr_create_irradiance_cubemap(ctx_, order)); }
    void CreatePrefilteredCubemap() { check(vkr_create_prefiltered_cubemap(ctx_)); }
    void CreateBRDFLUTTexture() { check(vkr_create_brdf_lut(ctx_)); }

    void Read(int32_t frame, int32_t target, int32_t mip, void *host, int64_t bytes) { check(vkr_read(ctx_, frame, target, mip, host, bytes)); }
    VkrTiming Timing()
    {
        VkrTiming t{};
        check(vkr_get_timing(ctx_, &t));
        return t;
    }

private:
    vkr_ctx *ctx_ = nullptr;
    bool has_volume_mesh_ = false;
};

}  // namespace vkr
