// What HBM delivers to the variable-node sweep's access pattern on MI355X: in-place read-modify-write of
// rows of 64 x GW floats (GW = 1, 2, 4: granules of 256 B, 512 B, 1 KiB) that lie scattered over a tile of
// R rows, every row touched exactly once per sweep, D rows gathered per node before any is written back.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/gather_bw tools/gather_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float fv1 __attribute__((ext_vector_type(1)));
typedef float fv2 __attribute__((ext_vector_type(2)));
typedef float fv4 __attribute__((ext_vector_type(4)));
template <int GW> struct Vec;
template <> struct Vec<1> { typedef float type; };
template <> struct Vec<2> { typedef fv2 type; };
template <> struct Vec<4> { typedef fv4 type; };

// tile t holds R rows of 64 * GW floats; wave (t, c) handles nodes c * npw .. + npw - 1; node v gathers rows
// perm(v * D + k), k < D, with perm(x) = x * P mod R (P coprime to R): scattered, and a bijection, so a
// sweep touches every row once.  SEQ: perm = identity (the check-node sweep's contiguous rows).
template <int GW, int D, bool NT, bool SEQ>
__global__ void __launch_bounds__(256) gather_rmw_k(float* a, long long R, long long P, int npw, int waves_per_tile, long long tiles) {
    typedef typename Vec<GW>::type V;
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long tile = wave / waves_per_tile, c = wave % waves_per_tile;
    if (tile >= tiles) return;  // the last block may hold waves past the last tile
    V* base = reinterpret_cast<V*>(a + tile * R * 64 * GW) + lane;
    for (int v = 0; v < npw; ++v) {
        const long long node = c * npw + v;
        if ((node + 1) * D > R) break;
        V x[D];
        long long row[D];
#pragma unroll
        for (int k = 0; k < D; ++k) row[k] = SEQ ? node * D + k : ((node * D + k) * P) % R;
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = NT ? __builtin_nontemporal_load(&base[row[k] * 64]) : base[row[k] * 64];
        V s = x[0];
#pragma unroll
        for (int k = 1; k < D; ++k) s += x[k];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const V y = s - x[k];
            if (NT) __builtin_nontemporal_store(y, &base[row[k] * 64]); else base[row[k] * 64] = y;
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

template <int GW, int D, bool NT, bool SEQ>
void run(float* a, size_t bytes, long long R, int npw) {
    const long long P = 48271;  // coprime to R = 121344 = 2^9 * 3 * 79
    const long long tile_bytes = R * 64 * GW * 4;
    const long long tiles = (long long)(bytes / tile_bytes);
    const long long nodes = R / D;
    const int wpt = (int)((nodes + npw - 1) / npw);
    const long long waves = tiles * wpt;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    const double moved = 2.0 * (double)tiles * (double)(nodes * D) * 64 * GW * 4;
    const float t = time_ms([&] { hipLaunchKernelGGL((gather_rmw_k<GW, D, NT, SEQ>), dim3(blocks), dim3(256), 0, 0, a, R, P, npw, wpt, tiles); });
    printf("%-9s granule %4d B, %2d rows per node, %3d nodes per wave, nt %d: %7.1f GB/s (read+write), %lld tiles of %.1f MB, %lld waves\n",
           SEQ ? "contiguous" : "scattered", 256 * GW, D, npw, (int)NT, moved / t / 1e6, tiles, tile_bytes / 1e6, waves);
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 24.0) * (1ull << 30);
    float* a;
    CK(hipMalloc(&a, bytes));
    CK(hipMemset(a, 0, bytes));
    const long long R = 121344;  // edges of the NR-BG1-sized code
    run<1, 3, false, true>(a, bytes, R, 32);  run<1, 8, false, true>(a, bytes, R, 32);  run<1, 19, false, true>(a, bytes, R, 8);
    run<1, 3, false, false>(a, bytes, R, 32); run<1, 8, false, false>(a, bytes, R, 32); run<1, 30, false, false>(a, bytes, R, 32);
    run<1, 3, true, false>(a, bytes, R, 32);  run<1, 8, true, false>(a, bytes, R, 32);  run<1, 30, true, false>(a, bytes, R, 32);
    run<2, 3, false, false>(a, bytes, R, 32); run<2, 8, false, false>(a, bytes, R, 32); run<2, 30, false, false>(a, bytes, R, 32);
    run<2, 3, true, false>(a, bytes, R, 32);  run<2, 8, true, false>(a, bytes, R, 32);  run<2, 30, true, false>(a, bytes, R, 32);
    run<4, 3, false, false>(a, bytes, R, 32); run<4, 8, false, false>(a, bytes, R, 32);
    run<4, 3, true, false>(a, bytes, R, 32);  run<4, 8, true, false>(a, bytes, R, 32);  run<4, 8, true, true>(a, bytes, R, 32);
    run<1, 8, false, false>(a, bytes, R, 8);  run<1, 8, false, false>(a, bytes, R, 128);
    return 0;
}
