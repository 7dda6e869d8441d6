// Micro-benchmark: what would a Newton-3 j-force write-back cost on gfx950?
// Each wave walks `n_entries` pseudo-random j-clusters near its own tile and adds an 8-atom force
// record with f32 global atomics, optionally interleaved with dependent FMA work that stands in for
// the pair evaluations of the entry.  Variants: 0 = no atomics (VALU only), 1 = 3 instr x 8 lanes,
// 2 = 1 instr x 24 lanes, 3 = plain (non-atomic) store of the record to a per-entry buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int VARIANT>
__global__ __launch_bounds__(256) void k(float* __restrict__ force, float4* __restrict__ scratch, int n_clusters,
                                         int n_entries, int work, float seed) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int my_cluster = (wave >> 2) * 8;
    unsigned h = wave * 2654435761u;
    float ax = seed + lane, ay = seed * 2 + lane, az = seed * 3 + lane;
    for (int e = 0; e < n_entries; ++e) {
        h = h * 1664525u + 1013904223u;
        int jc = my_cluster + (int)((h >> 8) % 4096u) - 2048;
        jc = jc < 0 ? jc + n_clusters : (jc >= n_clusters ? jc - n_clusters : jc);
        float fx = ax, fy = ay, fz = az;
        for (int w = 0; w < work; ++w) {       // dependent chains, 3 independent streams
            fx = fmaf(fx, 1.0001f, 0.5f); fy = fmaf(fy, 0.9999f, 0.25f); fz = fmaf(fz, 1.0002f, 0.125f);
        }
        // reduce over ii = lane & 7 (3 DPP-able steps per component)
        for (int m = 1; m < 8; m <<= 1) {
            fx += __shfl_xor(fx, m); fy += __shfl_xor(fy, m); fz += __shfl_xor(fz, m);
        }
        int jj = lane >> 3, ii = lane & 7;
        if (VARIANT == 1) {
            if (ii == 0) {
                float* p = force + (size_t)(jc * 8 + jj) * 4;
                atomicAdd(p, fx); atomicAdd(p + 1, fy); atomicAdd(p + 2, fz);
            }
        } else if (VARIANT == 2) {
            if (ii < 3) {
                float v = ii == 0 ? fx : (ii == 1 ? fy : fz);
                atomicAdd(force + (size_t)(jc * 8 + jj) * 4 + ii, v);
            }
        } else if (VARIANT == 3) {
            if (ii == 0) scratch[((size_t)wave * n_entries + e) * 8 + jj] = make_float4(fx, fy, fz, 0.f);
        }
        ax += fx * 1e-9f; ay += fy * 1e-9f; az += fz * 1e-9f;
    }
    if (ax + ay + az == 12345.678f) force[0] = ax;
}

template <int V> float run(float* f, float4* s, int nc, int tiles, int ne, int work) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<V><<<tiles, 256>>>(f, s, nc, ne, work, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) k<V><<<tiles, 256>>>(f, s, nc, ne, work, 1.0f);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 5;
}

int main() {
    int tiles = 16400, nc = tiles * 8;
    float* f; CK(hipMalloc(&f, (size_t)nc * 8 * 16)); CK(hipMemset(f, 0, (size_t)nc * 8 * 16));
    int ne = 52;                                  // 16400 tiles x 4 waves x 52 = 3.4 M entries (half list at 1 M atoms)
    float4* s; CK(hipMalloc(&s, (size_t)tiles * 4 * ne * 8 * 16));
    for (int work : {0, 50, 100}) {
        printf("work %3d FMAx3/entry: none %.3f ms | 3x8-lane atomics %.3f | 1x24-lane atomics %.3f | store-to-scratch %.3f\n",
               work, run<0>(f, s, nc, tiles, ne, work), run<1>(f, s, nc, tiles, ne, work),
               run<2>(f, s, nc, tiles, ne, work), run<3>(f, s, nc, tiles, ne, work));
    }
    return 0;
}
