// Microbenchmark 2: does the cost of a per-lane dword gather depend on the number of active lanes
// or of distinct addresses?  Plus LDS gather and ds_bpermute rates.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITER = 256;

// ACTIVE lanes of each wave issue the gathers (others are masked off by a branch)
template <int ACTIVE>
__global__ __launch_bounds__(256) void masked_gather(const float *table, const unsigned *offsets, float *out, unsigned mask)
{
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    unsigned off[8];
#pragma unroll
    for (int k = 0; k < 8; k++) off[k] = offsets[(size_t)k * gridDim.x * 256 + tid];
    float acc = 0.0f;
    const char *base = (const char *)table;
    if ((threadIdx.x & 63) < ACTIVE) {
        for (int it = 0; it < ITER; it++) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = *(const float *)(base + ((off[k] + it * 132u) & mask));
#pragma unroll
            for (int k = 0; k < 8; k++) acc += v[k];
        }
    }
    out[tid] = acc;
}

// mode 0: ds_read_b32 random gather; mode 1: ds_bpermute_b32
template <int MODE>
__global__ __launch_bounds__(256) void lds_kernel(const unsigned *offsets, float *out)
{
    __shared__ float lds[4096];
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    unsigned off[8];
#pragma unroll
    for (int k = 0; k < 8; k++) off[k] = offsets[(size_t)k * gridDim.x * 256 + tid];
    float acc = 0.0f;
    for (int it = 0; it < ITER; it++) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            unsigned o = (off[k] + it * 132u) & 16383u;
            if (MODE == 0) v[k] = *(const float *)((const char *)lds + (o & ~3u));
            else v[k] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(o & 252u), __builtin_bit_cast(int, acc + (float)k)));
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc += v[k];
    }
    out[tid] = acc;
}

int main()
{
    const int blocks = 256 * 8, threads = blocks * 256;
    const size_t table_bytes = 4u << 20;
    float *table; unsigned *offs; float *out;
    CHECK(hipMalloc(&table, table_bytes)); CHECK(hipMemset(table, 0, table_bytes));
    CHECK(hipMalloc(&offs, (size_t)8 * threads * 4)); CHECK(hipMalloc(&out, (size_t)threads * 4));
    std::vector<unsigned> h((size_t)8 * threads);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const unsigned mask = (16u << 10) - 1;
    auto timeit = [&](auto launch) {
        launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e-3 * 2.4e9 / ((double)blocks * 4 * ITER * 8 / 256.0);
    };
    // distinct = number of distinct addresses per wave-instruction (lanes share addresses in groups)
    for (int distinct : {64, 16, 4, 1}) {
        srand(2);
        for (int k = 0; k < 8; k++)
            for (int t = 0; t < threads; t++) {
                int lane = t & 63;
                unsigned g = lane / (64 / distinct);
                h[(size_t)k * threads + t] = (((unsigned)(t >> 6) * 977u + k * 131u) * 4096u + (g >> 3) * 512 + (g & 7) * 4) & mask;
            }
        CHECK(hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        printf("8 rows x 8 dwords pattern, %2d distinct addresses per instr:", distinct);
        printf("  64 lanes %5.1f", timeit([&] { hipLaunchKernelGGL(masked_gather<64>, dim3(blocks), dim3(256), 0, 0, table, offs, out, mask); }));
        printf("  32 lanes %5.1f", timeit([&] { hipLaunchKernelGGL(masked_gather<32>, dim3(blocks), dim3(256), 0, 0, table, offs, out, mask); }));
        printf("  16 lanes %5.1f", timeit([&] { hipLaunchKernelGGL(masked_gather<16>, dim3(blocks), dim3(256), 0, 0, table, offs, out, mask); }));
        printf("   8 lanes %5.1f cyc/instr/CU\n", timeit([&] { hipLaunchKernelGGL(masked_gather<8>, dim3(blocks), dim3(256), 0, 0, table, offs, out, mask); }));
    }
    // the same 8x8 footprint with other row pitches (bytes); table mask 1 MiB so that 8 rows always fit
    for (unsigned pitch : {512u, 1024u, 2048u, 4096u, 4096u + 128u, 4096u + 256u, 8192u, 16384u}) {
        for (int k = 0; k < 8; k++)
            for (int t = 0; t < threads; t++) {
                int lane = t & 63;
                h[(size_t)k * threads + t] = (((unsigned)(t >> 6) * 977u + k * 131u) * 128u) % (64u << 10) + (lane >> 3) * pitch + (lane & 7) * 4;
            }
        CHECK(hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        printf("8 rows x 8 dwords, row pitch %5u B: %5.1f cyc/instr/CU\n", pitch,
               timeit([&] { hipLaunchKernelGGL(masked_gather<64>, dim3(blocks), dim3(256), 0, 0, table, offs, out, (1u << 20) - 1); }));
    }
    for (unsigned pitch : {4096u, 4096u + 128u}) {
        for (int k = 0; k < 8; k++)
            for (int t = 0; t < threads; t++) {
                int lane = t & 63;
                h[(size_t)k * threads + t] = (((unsigned)(t >> 6) * 977u + k * 131u) * 128u) % (64u << 10) + (lane >> 4) * pitch + (lane & 15) * 4;
            }
        CHECK(hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        printf("4 rows x 16 dwords, row pitch %5u B: %5.1f cyc/instr/CU\n", pitch,
               timeit([&] { hipLaunchKernelGGL(masked_gather<64>, dim3(blocks), dim3(256), 0, 0, table, offs, out, (1u << 20) - 1); }));
    }
    srand(3);
    for (auto &v : h) v = (unsigned)rand() * 2654435761u;
    CHECK(hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("LDS random ds_read_b32 gather: %5.1f cyc/instr/CU\n", timeit([&] { hipLaunchKernelGGL(lds_kernel<0>, dim3(blocks), dim3(256), 0, 0, offs, out); }));
    printf("ds_bpermute_b32:               %5.1f cyc/instr/CU\n", timeit([&] { hipLaunchKernelGGL(lds_kernel<1>, dim3(blocks), dim3(256), 0, 0, offs, out); }));
    return 0;
}
