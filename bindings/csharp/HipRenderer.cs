// P/Invoke binding of libvkr_hip.so for the reference's C# code base (namespace Renderer).
// It keeps the names the reference's callers use -- UpdateSceneInfo, Lights, BeginFrame, EndFrame,
// CreateSkyboxCubemap, ... -- so Program.cs drives it exactly as it drives VulkanRenderer, for the
// composition / bloom / post-process part of a frame and the IBL precompute.
// SceneInfo and LightInfo are the reference's own explicit-layout structs (VulkanRenderer.cs:11-28)
// and cross the boundary without marshalling.
//
// Not compiled in this repository (no .NET SDK in the build image); it is declarative on purpose.
using System.Runtime.InteropServices;
using Silk.NET.Maths;

namespace Renderer;

[StructLayout(LayoutKind.Sequential)]
public struct VkrConfig
{
    public int Device, Width, Height, MaxLights, DirShadowRes, PointShadowRes, EnvRes,
               IrradianceRes, PrefilterRes, PrefilterMips, BrdfLutRes, FramesInFlight, Flags,
               FrameRows, Row0;   // rows [Row0, Row0 + Height) of a frame of FrameRows rows; 0, 0 = the whole frame
}

[StructLayout(LayoutKind.Sequential)]
public struct VkrTiming { public float ShadeMs, BloomHMs, BloomVPostMs, FrameMs; }

internal static unsafe class Native
{
    const string Lib = "vkr_hip";
    [DllImport(Lib)] public static extern sbyte* vkr_last_error();
    [DllImport(Lib)] public static extern int vkr_abi_version();
    [DllImport(Lib)] public static extern int vkr_device_count();   // gfx950 devices visible to the process: how many strips a frame can be cut into
    [DllImport(Lib)] public static extern int vkr_create(in VkrConfig config, out nint ctx);
    [DllImport(Lib)] public static extern int vkr_destroy(nint ctx);
    [DllImport(Lib)] public static extern int vkr_resize(nint ctx, int width, int height);
    [DllImport(Lib)] public static extern int vkr_update_scene_info(nint ctx, in SceneInfo info);
    [DllImport(Lib)] public static extern int vkr_set_lights(nint ctx, LightInfo* lights, int n, int coverageMode);
    [DllImport(Lib)] public static extern int vkr_set_light_volume_mesh(nint ctx, float* verts, int nVerts, uint* indices, int nTris);
    [DllImport(Lib)] public static extern int vkr_create_mesh(nint ctx, float* positions, int nVerts, uint* indices, int nTris, out int mesh);
    [DllImport(Lib)] public static extern int vkr_set_mesh_attributes(nint ctx, int mesh, float* texcoords, float* normals, float* tangents);
    [DllImport(Lib)] public static extern int vkr_create_material(nint ctx, void* albedo, int aw, int ah, void* normal, int nw, int nh, void* arm, int mw, int mh, out int material);
    [DllImport(Lib)] public static extern int vkr_draw_model(nint ctx, int mesh, int material, in Matrix4X4<float> model);
    [DllImport(Lib)] public static extern int vkr_begin_frame(nint ctx);
    [DllImport(Lib)] public static extern int vkr_end_frame(nint ctx);
    [DllImport(Lib)] public static extern int vkr_upload_gbuffer(nint ctx, void* albedo, void* normal, void* arm, void* position);
    [DllImport(Lib)] public static extern int vkr_upload_background(nint ctx, void* rgba16f);
    [DllImport(Lib)] public static extern int vkr_upload_dir_shadow(nint ctx, float* depth);
    [DllImport(Lib)] public static extern int vkr_upload_point_shadow(nint ctx, int light, float* depthCube);
    [DllImport(Lib)] public static extern int vkr_create_skybox_cubemap(nint ctx, void* equirect, int w, int h, int format);
    [DllImport(Lib)] public static extern int vkr_create_irradiance_cubemap(nint ctx, int order);
    [DllImport(Lib)] public static extern int vkr_create_prefiltered_cubemap(nint ctx);
    [DllImport(Lib)] public static extern int vkr_create_brdf_lut(nint ctx);
    [DllImport(Lib)] public static extern int vkr_read(nint ctx, int frame, int target, int mip, void* host, long hostBytes);
    [DllImport(Lib)] public static extern int vkr_sync(nint ctx);
    [DllImport(Lib)] public static extern int vkr_get_timing(nint ctx, out VkrTiming timing);
    [DllImport(Lib)] public static extern int vkr_current_frame(nint ctx);
    [DllImport(Lib)] public static extern int vkr_upload_env_cube(nint ctx, void* rgba16fCube);
    // single passes, run at once
    [DllImport(Lib)] public static extern int vkr_pass_composition(nint ctx);
    [DllImport(Lib)] public static extern int vkr_pass_bloom(nint ctx, int horizontal);
    [DllImport(Lib)] public static extern int vkr_pass_postprocess(nint ctx);
    // RenderStage.cs:99-190
    [DllImport(Lib)] public static extern int vkr_stage_reset_commands(nint ctx, int stage, int frame);
    [DllImport(Lib)] public static extern int vkr_stage_begin_commands(nint ctx, int stage, int frame);
    [DllImport(Lib)] public static extern int vkr_stage_begin_render_pass(nint ctx, int stage, int frame);
    [DllImport(Lib)] public static extern int vkr_stage_draw(nint ctx, int stage, int frame);
    [DllImport(Lib)] public static extern int vkr_stage_end_render_pass(nint ctx, int stage, int frame);
    [DllImport(Lib)] public static extern int vkr_stage_end_commands(nint ctx, int stage, int frame);
    [DllImport(Lib)] public static extern int vkr_submit_frame(nint ctx);
    // strips: one frame on several GPUs
    [DllImport(Lib)] public static extern int vkr_strip_plan(int frameRows, int nStrips, int k, out int row0, out int rows);
    [DllImport(Lib)] public static extern int vkr_strip_shade(nint ctx);
    [DllImport(Lib)] public static extern int vkr_strip_halo(nint ctx, int side, out nint send, out nint recv, out long bytes);
    [DllImport(Lib)] public static extern int vkr_strip_exchange(nint upper, nint lower);
    [DllImport(Lib)] public static extern int vkr_strip_finish(nint ctx);
    [DllImport(Lib)] public static extern int vkr_resize_strip(nint ctx, int width, int height, int frameRows, int row0);
}

// VKR_STAGE_*: the reference's stage objects (VulkanRenderer.cs:80-88) in its submission order (:563-584)
public enum HipStage { PointShadow = 0, Geometry = 1, DirectionalShadow = 2, Composition = 3, Bloom1 = 4, Bloom2 = 5, PostProcess = 6 }

// RenderStage.cs with the same five calls and arguments, so that EndFrame can stay written as it is
// (VulkanRenderer.cs:346-557): BeginCommands(currentFrame); BeginRenderPass(currentFrame, currentFrame); the stage's
// vkCmdBind* / vkCmdDraw* calls become one Draw(currentFrame); EndRenderPass(currentFrame); EndCommands(currentFrame).
// Nothing runs until HipRenderer.SubmitFrame(), which takes the place of vk.QueueSubmit + QueuePresent (:559-622).
public sealed class HipRenderStage
{
    readonly nint ctx;
    readonly int stage;
    internal HipRenderStage(nint ctx, HipStage stage) { this.ctx = ctx; this.stage = (int) stage; }

    public void ResetCommandBuffer(uint commandBufferIndex) => HipRenderer.Check(Native.vkr_stage_reset_commands(ctx, stage, (int) commandBufferIndex));
    public void BeginCommands(uint commandBufferIndex) => HipRenderer.Check(Native.vkr_stage_begin_commands(ctx, stage, (int) commandBufferIndex));
    // framebufferIndex selects nothing: the context owns one set of attachments per frame in flight
    public void BeginRenderPass(uint commandBufferIndex, uint framebufferIndex) => HipRenderer.Check(Native.vkr_stage_begin_render_pass(ctx, stage, (int) commandBufferIndex));
    public void Draw(uint commandBufferIndex) => HipRenderer.Check(Native.vkr_stage_draw(ctx, stage, (int) commandBufferIndex));
    public void EndRenderPass(uint commandBufferIndex) => HipRenderer.Check(Native.vkr_stage_end_render_pass(ctx, stage, (int) commandBufferIndex));
    public void EndCommands(uint commandBufferIndex) => HipRenderer.Check(Native.vkr_stage_end_commands(ctx, stage, (int) commandBufferIndex));
}

public sealed unsafe class HipRenderer : IDisposable
{
    nint ctx;
    bool disposedValue;
    bool hasVolumeMesh;
    public List<Light> Lights = new();   // VulkanRenderer.cs:65

    // VKR_ABI_VERSION of the vkr_hip.h these declarations were written against: a library of another revision
    // would misread the structs passed below
    public const int AbiVersion = 4;
    // VkrConfig.Flags: 1 = store through an _SRGB target; depth ranges of the point-shadow cubes in the composition pass
    // (default: frames with at least 16 point lights use them; the image is the same either way)
    public const int PostSrgbTarget = 1, ShadowRangesAlways = 0x100, ShadowRangesNever = 0x200;

    public HipRenderer(in VkrConfig config)
    {
        if (Native.vkr_abi_version() != AbiVersion)
            throw new Exception($"libvkr_hip has interface version {Native.vkr_abi_version()}, this binding needs {AbiVersion}");
        Check(Native.vkr_create(config, out ctx));
    }

    internal static void Check(int rc)
    {
        // the reference throws Exception on any non-Success VkResult (e.g. VulkanRenderer.cs:589-594)
        if (rc < 0) throw new Exception(Marshal.PtrToStringUTF8((nint)Native.vkr_last_error()));
    }

    public void UpdateSceneInfo(SceneInfo sceneInfo) => Check(Native.vkr_update_scene_info(ctx, sceneInfo));
    public void BeginFrame() => Check(Native.vkr_begin_frame(ctx));

    public void PushLights()
    {
        var infos = stackalloc LightInfo[Lights.Count];
        for (int i = 0; i < Lights.Count; i++) infos[i] = Lights[i].ToInfo();   // Light.cs:10
        Check(Native.vkr_set_lights(ctx, infos, Lights.Count, hasVolumeMesh ? 1 : 0));   // 1 = VKR_COVERAGE_MESH
    }

    // the whole frame in one call: both bloom passes and the post-process draw run as one kernel
    public void EndFrame()
    {
        PushLights();
        Check(Native.vkr_end_frame(ctx));
    }

    // EndFrame written stage by stage, as in the reference (VulkanRenderer.cs:320-623); the stage fields take the
    // place of geometryRenderStage, compositionRenderStage, ... (:80-88)
    public uint CurrentFrame => (uint) Native.vkr_current_frame(ctx);
    public HipRenderStage Stage(HipStage stage) => new(ctx, stage);
    public void SubmitFrame() => Check(Native.vkr_submit_frame(ctx));
    public void EndFrameByStages(bool hasDrawCalls)
    {
        uint currentFrame = CurrentFrame;
        PushLights();
        foreach (HipStage id in Enum.GetValues<HipStage>())
        {
            // without draw calls the shadow maps and the G-buffer are the uploaded ones: those stages record nothing
            if (!hasDrawCalls && id <= HipStage.DirectionalShadow) continue;
            var stage = Stage(id);
            stage.BeginCommands(currentFrame);
            stage.BeginRenderPass(currentFrame, currentFrame);
            stage.Draw(currentFrame);
            stage.EndRenderPass(currentFrame);
            stage.EndCommands(currentFrame);
        }
        SubmitFrame();
    }

    // One frame on several GPUs from this one process: a HipRenderer per GPU, each created with its strip
    // (VkrConfig.Device / Height / FrameRows / Row0 from StripPlan); per frame every renderer gets the same
    // BeginFrame / UpdateSceneInfo / Lights / DrawModel calls, then
    //     foreach r: r.StripShade();   foreach neighbouring pair: HipRenderer.ExchangeHalo(upper, lower);   foreach r: r.StripFinish();
    // and Read(frame, VKR_TARGET_LDR) of renderer k fills rows [Row0, Row0 + Height) of the image.
    public static int DeviceCount => Native.vkr_device_count();
    public static (int row0, int rows) StripPlan(int frameRows, int nStrips, int k)
    {
        Check(Native.vkr_strip_plan(frameRows, nStrips, k, out int row0, out int rows));
        return (row0, rows);
    }
    public void StripShade() { PushLights(); Check(Native.vkr_strip_shade(ctx)); }
    public void StripFinish() => Check(Native.vkr_strip_finish(ctx));
    public static void ExchangeHalo(HipRenderer upper, HipRenderer lower) => Check(Native.vkr_strip_exchange(upper.ctx, lower.ctx));

    // single passes, run at once on the context's stream (for tools that want one pass of a frame)
    public void CompositionPass() { PushLights(); Check(Native.vkr_pass_composition(ctx)); }
    public void BloomPass(bool horizontal) => Check(Native.vkr_pass_bloom(ctx, horizontal ? 1 : 0));
    public void PostProcessPass() => Check(Native.vkr_pass_postprocess(ctx));
    public void UploadEnvironmentCubemap(void* rgba16fCube) => Check(Native.vkr_upload_env_cube(ctx, rgba16fCube));

    // Mesh.cs keeps positions and indices on the device; DrawModel is VulkanRenderer.cs:630-633
    public int CreateMesh(float* positions, int nVerts, uint* indices, int nTris)
    {
        Check(Native.vkr_create_mesh(ctx, positions, nVerts, indices, nTris, out int mesh));
        return mesh;
    }
    public void SetMeshAttributes(int mesh, float* texcoords, float* normals, float* tangents) => Check(Native.vkr_set_mesh_attributes(ctx, mesh, texcoords, normals, tangents));
    public int CreateMaterial(void* albedo, int aw, int ah, void* normal, int nw, int nh, void* arm, int mw, int mh)
    {
        Check(Native.vkr_create_material(ctx, albedo, aw, ah, normal, nw, nh, arm, mw, mh, out int material));
        return material;
    }
    public void DrawModel(int mesh, int material, Matrix4X4<float> modelMatrix) => Check(Native.vkr_draw_model(ctx, mesh, material, modelMatrix));

    // sphereMesh of the reference (VulkanRenderer.cs:78,269): positions and indices of the light volume
    public void SetLightVolumeMesh(float* verts, int nVerts, uint* indices, int nTris)
    {
        Check(Native.vkr_set_light_volume_mesh(ctx, verts, nVerts, indices, nTris));
        hasVolumeMesh = true;
    }

    public void WaitIdle() => Check(Native.vkr_sync(ctx));
    public void Resize(int width, int height) => Check(Native.vkr_resize(ctx, width, height));

    public void UploadGBuffer(void* albedo, void* normal, void* arm, void* position) => Check(Native.vkr_upload_gbuffer(ctx, albedo, normal, arm, position));
    public void UploadDirectionalShadowMap(float* depth) => Check(Native.vkr_upload_dir_shadow(ctx, depth));
    public void UploadPointShadowMap(int light, float* depthCube) => Check(Native.vkr_upload_point_shadow(ctx, light, depthCube));

    public void CreateSkyboxCubemap(void* equirect, int w, int h, int format) => Check(Native.vkr_create_skybox_cubemap(ctx, equirect, w, h, format));
    /// <summary>Summation order of the irradiance pass: the shader's own running sum, or (about twice as fast, other
    /// rounding in the last binary16 digit of under 1 % of the texels) the samples grouped by region of the environment.</summary>
    public const int IrradianceSequential = 0, IrradianceByRegion = 2, IrradianceByRegionRows = 3, IrradianceFastest = -1;
    // the default order is the library's fastest; it differs from the shader's own running sum (IrradianceSequential) in rounding only
    public void CreateIrradianceCubemap(int order = IrradianceFastest) => Check(Native.vkr_create_irradiance_cubemap(ctx, order));
    public void CreatePrefilteredCubemap() => Check(Native.vkr_create_prefiltered_cubemap(ctx));
    public void CreateBRDFLUTTexture() => Check(Native.vkr_create_brdf_lut(ctx));

    public void Read(int frame, int target, int mip, void* host, long bytes) => Check(Native.vkr_read(ctx, frame, target, mip, host, bytes));

    public void Dispose()
    {
        if (!disposedValue) { Native.vkr_destroy(ctx); ctx = 0; disposedValue = true; }
        GC.SuppressFinalize(this);
    }
    ~HipRenderer() { if (!disposedValue) Native.vkr_destroy(ctx); }
}
