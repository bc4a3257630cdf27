// Microbenchmark: cost of per-lane dword gathers on gfx950 by instruction form and locality.
// Each wave issues ITER x 8 independent gathers from a table; addresses are precomputed per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITER = 256;

// mode 0: saddr + 32-bit voffset (global_load_dword v, v, s[..])
// mode 1: 64-bit per-lane address (global_load_dword v, v[..], off)
// mode 2: raw buffer load, offen
// mode 3: dwordx2 via saddr+voffset, mode 4: dwordx4
template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const float *table, const unsigned *offsets, float *out, unsigned mask)
{
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    unsigned off[8];
#pragma unroll
    for (int k = 0; k < 8; k++) off[k] = offsets[(size_t)k * gridDim.x * 256 + tid];
    float acc = 0.0f;
    const char *base = (const char *)table;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)table, 0, 0x7fffffff, 0x00020000);
    for (int it = 0; it < ITER; it++) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            unsigned o = (off[k] + it * 4u * 33u) & mask;  // moves every iteration, stays in the table
            if (MODE == 0) v[k] = *(const float *)(base + o);
            else if (MODE == 1) {
                const char *p = base + o;
                asm volatile("" : "+v"(p));
                v[k] = *(const float *)p;
            } else if (MODE == 2) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o, 0, 0));
            else if (MODE == 3) { float2 t = *(const float2 *)(base + (o & ~7u)); v[k] = t.x + t.y; }
            else { float4 t = *(const float4 *)(base + (o & ~15u)); v[k] = t.x + t.y + t.z + t.w; }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc += v[k];
    }
    out[tid] = acc;
}

int main()
{
    const int blocks = 256 * 8, threads = blocks * 256;  // 8 blocks x 4 waves = 32 waves per CU
    const size_t table_bytes = 256u << 20;  // the largest pattern: the size class of a 2048^2 RGBA16F cube (201 MB), most of the 256 MiB Infinity Cache
    float *table; unsigned *offs; float *out;
    CHECK(hipMalloc(&table, table_bytes));
    CHECK(hipMemset(table, 0, table_bytes));
    CHECK(hipMalloc(&offs, (size_t)8 * threads * 4));
    CHECK(hipMalloc(&out, (size_t)threads * 4));
    std::vector<unsigned> h((size_t)8 * threads);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct Pattern { const char *name; int kind; unsigned mask; };
    // kind 0: lane l reads dword l of a per-wave random row (fully coalesced, 2 lines)
    // kind 1: 8x8 lanes -> 8 rows of a 4 KiB-pitch image, 8 adjacent dwords per row (8 lines, like a shadow-map footprint)
    // kind 2: every lane its own random line within `mask`
    // kind 3: 4 rows x 16 dwords
    Pattern pats[] = {{"coalesced row", 0, (16u << 10) - 1}, {"8 rows x 8 dwords, 16 KiB table", 1, (16u << 10) - 1},
                      {"4 rows x 16 dwords, 16 KiB table", 3, (16u << 10) - 1},
                      {"8 rows x 8 dwords, 4 MiB table", 1, (4u << 20) - 1}, {"random lines, 16 KiB table", 2, (16u << 10) - 1},
                      {"random lines, 2 MiB table", 2, (2u << 20) - 1}, {"random lines, 64 MiB table", 2, (64u << 20) - 1},
                      {"random lines, 256 MiB table", 2, (256u << 20) - 1}};
    for (auto &pt : pats) {
        srand(1);
        for (int k = 0; k < 8; k++)
            for (int t = 0; t < threads; t++) {
                int lane = t & 63, wave = t >> 6;
                unsigned wbase = ((unsigned)rand() * 2654435761u + wave * 40503u) & pt.mask & ~4095u;
                unsigned o;
                if (pt.kind == 0) o = wbase + lane * 4;
                else if (pt.kind == 1) o = wbase + (lane >> 3) * 4096 / 8 * 1 + (lane & 7) * 4 + (lane >> 3) * 0;  // rows 512 B apart
                else if (pt.kind == 3) o = wbase + (lane >> 4) * 1024 + (lane & 15) * 4;
                else o = ((unsigned)rand() * 2654435761u) & ~3u;
                h[(size_t)k * threads + t] = o & pt.mask;
            }
        CHECK(hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        printf("%-36s", pt.name);
        for (int mode = 0; mode < 5; mode++) {
            auto launch = [&]() {
                switch (mode) {
                case 0: hipLaunchKernelGGL(gather_kernel<0>, dim3(blocks), dim3(256), 0, 0, table, offs, out, pt.mask); break;
                case 1: hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, table, offs, out, pt.mask); break;
                case 2: hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(256), 0, 0, table, offs, out, pt.mask); break;
                case 3: hipLaunchKernelGGL(gather_kernel<3>, dim3(blocks), dim3(256), 0, 0, table, offs, out, pt.mask); break;
                default: hipLaunchKernelGGL(gather_kernel<4>, dim3(blocks), dim3(256), 0, 0, table, offs, out, pt.mask); break;
                }
            };
            launch();
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            double instr_per_cu = (double)blocks * 4 * ITER * 8 / 256.0;
            printf("  m%d %6.1f cyc", mode, ms * 1e-3 * 2.4e9 / instr_per_cu);
            // every lane of a random-line gather brings in its own 128-byte line
            if (pt.kind == 2 && mode == 0) printf(" (%.2f TB/s of lines)", (double)blocks * 4 * ITER * 8 * 64 * 128 / (ms * 1e-3) / 1e12);
        }
        printf("\n");
    }
    return 0;
}
