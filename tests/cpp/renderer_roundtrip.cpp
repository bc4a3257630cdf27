// Drives vkr::HipRenderer (include/vkr_renderer.hpp) the way the reference's Program.OnRender drives
// VulkanRenderer (Program.cs:109-160): IBL precompute once, then BeginFrame / UpdateSceneInfo /
// Lights / EndFrame, twice (both frames in flight).  Inputs come from a file written by the test;
// the targets of the last frame go to the output file for comparison with the CPU oracle.
//
// input file (little endian): int32 w, h, n_lights, dir_res, cube_res, env_res, irr_res, pref_res, lut_res;
//   scene 240 B; per light 6 floats (position, colour); albedo, normal, arm, position (w*h*8 B each);
//   dir shadow (dir_res^2 floats); cube shadows (n_lights*6*cube_res^2 floats); env cube (6*env_res^2*8 B)
//   optional trailer: int32 n_draws; per draw: int32 n_verts, n_tris; positions (3 f32 per vertex), texcoords (2),
//   normals (3), tangents (3); indices (3 u32 per triangle); model (16 f32); three RGBA8 images, each int32 w, h + bytes.
//   With draws the frame is rendered from them (CreateMesh / CreateMaterial / DrawModel): nothing is uploaded.
// A third argument "stages" ends each frame the way VulkanRenderer.EndFrame is written (VulkanRenderer.cs:346-557):
// every pass recorded into its RenderStage by BeginCommands / BeginRenderPass / draws / EndRenderPass / EndCommands,
// then one submission; the shadow and geometry stages are recorded only when the frame has draw calls.
// output file: colour, threshold (w*h*8 B each), ldr (w*h*4 B), irradiance cube, prefilter mip 0, LUT
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "vkr_renderer.hpp"

static void read_exact(FILE *f, void *dst, size_t n)
{
    if (fread(dst, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}

int main(int argc, char **argv)
{
    if (argc != 3 && argc != 4) { fprintf(stderr, "usage: %s <in> <out> [stages]\n", argv[0]); return 2; }
    const bool by_stages = argc == 4 && std::string(argv[3]) == "stages";
    FILE *in = fopen(argv[1], "rb");
    if (!in) { perror(argv[1]); return 2; }
    int32_t h9[9];
    read_exact(in, h9, sizeof h9);
    const int w = h9[0], h = h9[1], nl = h9[2], dir_res = h9[3], cube_res = h9[4], env_res = h9[5], irr_res = h9[6], pref_res = h9[7], lut_res = h9[8];
    VkrSceneInfo scene;
    read_exact(in, &scene, sizeof scene);
    std::vector<vkr::Light> lights(nl);
    for (auto &l : lights) { read_exact(in, l.Position, 12); read_exact(in, l.Color, 12); }
    const size_t px = (size_t)w * h;
    std::vector<uint16_t> albedo(px * 4), normal(px * 4), arm(px * 4), position(px * 4);
    read_exact(in, albedo.data(), px * 8); read_exact(in, normal.data(), px * 8);
    read_exact(in, arm.data(), px * 8); read_exact(in, position.data(), px * 8);
    std::vector<float> dir((size_t)dir_res * dir_res), cubes((size_t)nl * 6 * cube_res * cube_res);
    read_exact(in, dir.data(), dir.size() * 4);
    read_exact(in, cubes.data(), cubes.size() * 4);
    std::vector<uint16_t> env((size_t)6 * env_res * env_res * 4);
    read_exact(in, env.data(), env.size() * 2);
    struct Draw {
        std::vector<float> pos, uv, nrm, tan; std::vector<uint32_t> idx; float model[16];
        std::vector<uint8_t> tex[3]; int32_t tw[3], th[3]; int32_t n_verts, n_tris;
    };
    std::vector<Draw> draws;
    int32_t n_draws = 0;
    if (fread(&n_draws, 4, 1, in) == 1 && n_draws > 0) {
        draws.resize(n_draws);
        for (Draw &d : draws) {
            read_exact(in, &d.n_verts, 4); read_exact(in, &d.n_tris, 4);
            d.pos.resize((size_t)d.n_verts * 3); d.uv.resize((size_t)d.n_verts * 2); d.nrm.resize((size_t)d.n_verts * 3); d.tan.resize((size_t)d.n_verts * 3);
            d.idx.resize((size_t)d.n_tris * 3);
            read_exact(in, d.pos.data(), d.pos.size() * 4); read_exact(in, d.uv.data(), d.uv.size() * 4);
            read_exact(in, d.nrm.data(), d.nrm.size() * 4); read_exact(in, d.tan.data(), d.tan.size() * 4);
            read_exact(in, d.idx.data(), d.idx.size() * 4); read_exact(in, d.model, 64);
            for (int k = 0; k < 3; k++) {
                read_exact(in, &d.tw[k], 4); read_exact(in, &d.th[k], 4);
                d.tex[k].resize((size_t)d.tw[k] * d.th[k] * 4);
                read_exact(in, d.tex[k].data(), d.tex[k].size());
            }
        }
    }
    fclose(in);

    try {
        VkrConfig cfg{};
        cfg.width = w; cfg.height = h; cfg.max_lights = nl > 0 ? nl : 1;
        cfg.dir_shadow_res = dir_res; cfg.point_shadow_res = cube_res;
        cfg.env_res = env_res; cfg.irradiance_res = irr_res; cfg.prefilter_res = pref_res; cfg.prefilter_mips = 1; cfg.brdf_lut_res = lut_res;
        cfg.frames_in_flight = 2;
        vkr::HipRenderer r(cfg);
        r.UploadEnvironmentCubemap(env.data());
        r.CreateIrradianceCubemap();
        r.CreatePrefilteredCubemap();
        r.CreateBRDFLUTTexture();
        r.Lights = lights;
        std::vector<int32_t> meshes, materials;
        for (const Draw &d : draws) {  // Mesh.cs / Material.cs
            meshes.push_back(r.CreateMesh(d.pos.data(), d.n_verts, d.idx.data(), d.n_tris));
            r.SetMeshAttributes(meshes.back(), d.uv.data(), d.nrm.data(), d.tan.data());
            materials.push_back(r.CreateMaterial(d.tex[0].data(), d.tw[0], d.th[0], d.tex[1].data(), d.tw[1], d.th[1], d.tex[2].data(), d.tw[2], d.th[2]));
        }
        int32_t frame = 0;
        for (int k = 0; k < 2; k++) {
            frame = r.CurrentFrame();
            r.BeginFrame();
            r.UpdateSceneInfo(scene);
            if (draws.empty()) {
                r.UploadGBuffer(albedo.data(), normal.data(), arm.data(), position.data());
                r.UploadDirectionalShadowMap(dir.data());
                for (int i = 0; i < nl; i++) r.UploadPointShadowMap(i, cubes.data() + (size_t)i * 6 * cube_res * cube_res);
            } else {
                for (size_t i = 0; i < draws.size(); i++) {  // Program.OnRender: renderer.DrawModel(model, transform)
                    float m[16];
                    for (int j = 0; j < 16; j++) m[j] = draws[i].model[j];
                    r.DrawModel(meshes[i], m, materials[i]);
                }
            }
            if (!by_stages) { r.EndFrame(); continue; }
            const uint32_t currentFrame = (uint32_t)frame;
            r.PushLights();
            for (int32_t id = 0; id < VKR_STAGE_COUNT; id++) {
                const bool from_draws = id == VKR_STAGE_POINT_SHADOW || id == VKR_STAGE_GEOMETRY || id == VKR_STAGE_DIR_SHADOW;
                if (from_draws && draws.empty()) continue;  // uploaded images take these stages' place
                vkr::HipRenderStage stage = r.Stage(id);
                stage.BeginCommands(currentFrame);
                stage.BeginRenderPass(currentFrame, currentFrame);
                stage.Draw(currentFrame);
                stage.EndRenderPass(currentFrame);
                stage.EndCommands(currentFrame);
            }
            // recording out of order fails the way a failed vkBeginCommandBuffer / vkEndCommandBuffer throws
            bool threw = false;
            try { r.Stage(VKR_STAGE_BLOOM1).BeginRenderPass(currentFrame, currentFrame); } catch (const vkr::Error &) { threw = true; }
            if (!threw) { fprintf(stderr, "BeginRenderPass on an ended command buffer did not throw\n"); return 3; }
            r.SubmitFrame();
        }
        r.WaitIdle();
        bool threw = false;  // EndFrame without BeginFrame throws, as VulkanRenderer.cs:322
        try { r.EndFrame(); } catch (const vkr::Error &) { threw = true; }
        if (!threw) { fprintf(stderr, "EndFrame without BeginFrame did not throw\n"); return 3; }

        std::vector<uint16_t> color(px * 4), thr(px * 4), irr((size_t)6 * irr_res * irr_res * 4), pre((size_t)6 * pref_res * pref_res * 4), lut((size_t)lut_res * lut_res * 2);
        std::vector<uint8_t> ldr(px * 4);
        r.Read(frame, VKR_TARGET_COLOR, 0, color.data(), (int64_t)color.size() * 2);
        r.Read(frame, VKR_TARGET_THRESHOLD, 0, thr.data(), (int64_t)thr.size() * 2);
        r.Read(frame, VKR_TARGET_LDR, 0, ldr.data(), (int64_t)ldr.size());
        r.Read(0, VKR_TARGET_IRRADIANCE, 0, irr.data(), (int64_t)irr.size() * 2);
        r.Read(0, VKR_TARGET_PREFILTER, 0, pre.data(), (int64_t)pre.size() * 2);
        r.Read(0, VKR_TARGET_BRDF_LUT, 0, lut.data(), (int64_t)lut.size() * 2);
        FILE *out = fopen(argv[2], "wb");
        if (!out) { perror(argv[2]); return 2; }
        fwrite(color.data(), 2, color.size(), out); fwrite(thr.data(), 2, thr.size(), out); fwrite(ldr.data(), 1, ldr.size(), out);
        fwrite(irr.data(), 2, irr.size(), out); fwrite(pre.data(), 2, pre.size(), out); fwrite(lut.data(), 2, lut.size(), out);
        fclose(out);
    } catch (const vkr::Error &e) {
        fprintf(stderr, "vkr error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
