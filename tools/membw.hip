// HBM streaming calibration on MI355X: what in-place read-modify-write patterns reach,
// next to a plain copy.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/membw tools/membw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void copy_k(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void rmw_k(float4* a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i]; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; a[i] = v;
    }
}
__global__ void read_k(const float4* __restrict__ a, float* out, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ void write_k(float4* a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// One block moves one contiguous chunk of U x 256 float4: U loads per lane are in flight before the first store,
// the grid covers the buffer exactly (no grid-stride loop).  NTL / NTS: non-temporal loads / stores.
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NTL, bool NTS, bool INPLACE>
__global__ void __launch_bounds__(256) chunk_k(const float4* __restrict__ a4, float4* __restrict__ b4) {
    const f4* a = reinterpret_cast<const f4*>(a4);
    f4* b = reinterpret_cast<f4*>(b4);
    const size_t base = (size_t)blockIdx.x * (U * 256) + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (NTL) v[u] = __builtin_nontemporal_load(&a[base + (size_t)u * 256]);
        else v[u] = a[base + (size_t)u * 256];
    }
    f4* out = INPLACE ? const_cast<f4*>(a) : b;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (INPLACE) { v[u].x += 1.f; v[u].y *= 1.0001f; }
        if (NTS) __builtin_nontemporal_store(v[u], &out[base + (size_t)u * 256]);
        else out[base + (size_t)u * 256] = v[u];
    }
}

// each wave owns a private contiguous region of `region` float4 and sweeps it `passes` times,
// ROWS rows of 64 float4 (1 KiB) per step: load ROWS, modify, store ROWS (the CN sweep pattern)
template <int ROWS, bool PREFETCH>
__global__ void __launch_bounds__(256) wave_rmw_k(float4* a, size_t region, int passes) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    float4* base = a + wave * region + lane;
    const size_t steps = region / (64 * ROWS);
    for (int p = 0; p < passes; ++p) {
        if (!PREFETCH) {
            for (size_t s = 0; s < steps; ++s) {
                float4 v[ROWS];
#pragma unroll
                for (int r = 0; r < ROWS; ++r) v[r] = base[(s * ROWS + r) * 64];
#pragma unroll
                for (int r = 0; r < ROWS; ++r) { v[r].x += 1.f; v[r].y *= 1.0001f; base[(s * ROWS + r) * 64] = v[r]; }
            }
        } else {
            float4 nx[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) nx[r] = base[r * 64];
            for (size_t s = 0; s < steps; ++s) {
                float4 v[ROWS];
#pragma unroll
                for (int r = 0; r < ROWS; ++r) v[r] = nx[r];
                if (s + 1 < steps) {
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) nx[r] = base[((s + 1) * ROWS + r) * 64];
                }
#pragma unroll
                for (int r = 0; r < ROWS; ++r) { v[r].x += 1.f; v[r].y *= 1.0001f; base[(s * ROWS + r) * 64] = v[r]; }
            }
        }
    }
}

template <class F> float time_ms(F f, int reps = 3) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 32.0) * (1ull << 30);
    const size_t n4 = bytes / 16;
    float4 *a, *b; float* o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    for (int blocks : {2048, 4096, 8192, 16384}) {
        float t;
        t = time_ms([&] { hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, 0, a, b, n4); });
        printf("copy   grid %5d: %7.1f GB/s (read+write)\n", blocks, 2.0 * bytes / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL(rmw_k, dim3(blocks), dim3(256), 0, 0, a, n4); });
        printf("rmw    grid %5d: %7.1f GB/s (read+write)\n", blocks, 2.0 * bytes / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL(read_k, dim3(blocks), dim3(256), 0, 0, a, o, n4); });
        printf("read   grid %5d: %7.1f GB/s\n", blocks, 1.0 * bytes / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL(write_k, dim3(blocks), dim3(256), 0, 0, a, n4); });
        printf("write  grid %5d: %7.1f GB/s\n", blocks, 1.0 * bytes / t / 1e6);
    }
    {
        float t;
#define CHUNK(U, NTL, NTS, INP, label)                                                                              \
        t = time_ms([&] { hipLaunchKernelGGL((chunk_k<U, NTL, NTS, INP>), dim3((unsigned)(n4 / (U * 256))), dim3(256), 0, 0, a, b); }); \
        printf("%-6s chunk of %2d x 256 float4 per block, nt loads %d, nt stores %d: %7.1f GB/s (read+write)\n", label, U, NTL, NTS, 2.0 * bytes / t / 1e6);
        CHUNK(4, false, false, false, "copy") CHUNK(8, false, false, false, "copy") CHUNK(16, false, false, false, "copy")
        CHUNK(8, false, true, false, "copy") CHUNK(8, true, true, false, "copy") CHUNK(16, true, true, false, "copy")
        CHUNK(4, true, true, false, "copy")
        CHUNK(8, false, false, true, "rmw") CHUNK(8, false, true, true, "rmw") CHUNK(8, true, true, true, "rmw") CHUNK(16, true, true, true, "rmw")
#undef CHUNK
    }
    // per-wave private regions: waves = blocks*4; region = bytes / waves
    for (int blocks : {512, 1024, 2048}) {
        const size_t waves = (size_t)blocks * 4;
        const size_t region = (n4 / waves) / (64 * 12) * (64 * 12);
        const int passes = 4;
        const double moved = 2.0 * passes * (double)region * 16 * waves;
        float t;
        t = time_ms([&] { hipLaunchKernelGGL((wave_rmw_k<6, false>), dim3(blocks), dim3(256), 0, 0, a, region, passes); });
        printf("wave-private rmw rows=6  no-prefetch blocks %4d (region %.1f MB): %7.1f GB/s\n", blocks, region * 16 / 1e6, moved / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL((wave_rmw_k<6, true>), dim3(blocks), dim3(256), 0, 0, a, region, passes); });
        printf("wave-private rmw rows=6  prefetch    blocks %4d: %7.1f GB/s\n", blocks, moved / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL((wave_rmw_k<12, true>), dim3(blocks), dim3(256), 0, 0, a, region, passes); });
        printf("wave-private rmw rows=12 prefetch    blocks %4d: %7.1f GB/s\n", blocks, moved / t / 1e6);
        t = time_ms([&] { hipLaunchKernelGGL((wave_rmw_k<3, true>), dim3(blocks), dim3(256), 0, 0, a, region, passes); });
        printf("wave-private rmw rows=3  prefetch    blocks %4d: %7.1f GB/s\n", blocks, moved / t / 1e6);
    }
    return 0;
}
