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
               IrradianceRes, PrefilterRes, PrefilterMips, BrdfLutRes, FramesInFlight, Flags;
}

[StructLayout(LayoutKind.Sequential)]
public struct VkrTiming { public float ShadeMs, BloomHMs, BloomVPostMs, FrameMs; }

internal static unsafe class Native
{
    const string Lib = "vkr_hip";
    [DllImport(Lib)] public static extern sbyte* vkr_last_error();
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
}

public sealed unsafe class HipRenderer : IDisposable
{
    nint ctx;
    bool disposedValue;
    bool hasVolumeMesh;
    public List<Light> Lights = new();   // VulkanRenderer.cs:65

    public HipRenderer(in VkrConfig config) => Check(Native.vkr_create(config, out ctx));

    static void Check(int rc)
    {
        // the reference throws Exception on any non-Success VkResult (e.g. VulkanRenderer.cs:589-594)
        if (rc < 0) throw new Exception(Marshal.PtrToStringUTF8((nint)Native.vkr_last_error()));
    }

    public void UpdateSceneInfo(SceneInfo sceneInfo) => Check(Native.vkr_update_scene_info(ctx, sceneInfo));
    public void BeginFrame() => Check(Native.vkr_begin_frame(ctx));

    public void EndFrame()
    {
        var infos = stackalloc LightInfo[Lights.Count];
        for (int i = 0; i < Lights.Count; i++) infos[i] = Lights[i].ToInfo();   // Light.cs:10
        Check(Native.vkr_set_lights(ctx, infos, Lights.Count, hasVolumeMesh ? 1 : 0));   // 1 = VKR_COVERAGE_MESH
        Check(Native.vkr_end_frame(ctx));
    }

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
    public void CreateIrradianceCubemap() => Check(Native.vkr_create_irradiance_cubemap(ctx, 0));
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
