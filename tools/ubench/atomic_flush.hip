// Micro-benchmark 2: flush patterns for the j-forces of a chunk of 8 entries (8 x 8 atoms x xyz = 192
// floats, each entry one 128-B line of the float4 force array).  Which instruction shape does the
// memory system like?   A: 8 x 24 lanes (one entry per instruction, xyz contiguous)
//                       B: 3 x 64 lanes strided (x of all 64 atoms, then y, then z: every line hit 3x in a row)
//                       C: 3 x 64 lanes packed (float k*64+lane of the [64][3] array: 2 of 8 lines shared)
//                       D: 4 x 48 lanes (two whole entries per instruction: no line shared)
//                       E: 2 x 96?? n/a      N: none
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int V>
__global__ __launch_bounds__(256) void k(float* __restrict__ force, int n_clusters, int n_chunks, int work) {
    __shared__ float4 s_g[4][64];
    __shared__ uint32_t s_jc[4][8];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + wave;
    const int my_cluster = (gw >> 2) * 8;
    unsigned h = gw * 2654435761u;
    float acc = lane * 0.001f;
    const int ii = lane & 7, jj = lane >> 3;
    for (int c = 0; c < n_chunks; ++c) {
        if (lane < 8) {
            unsigned hh = (h + lane * 40503u) * 1664525u + 1013904223u;
            int jc = my_cluster + (int)((hh >> 8) & 4095u) - 2048;
            jc = jc < 0 ? jc + n_clusters : (jc >= n_clusters ? jc - n_clusters : jc);
            s_jc[wave][lane] = (uint32_t)jc;
        }
        h = h * 1664525u + 1013904223u;
        float f = acc;
        for (int w = 0; w < work; ++w) f = fmaf(f, 1.0001f, 0.5f);
        s_g[wave][lane] = make_float4(f, f * 0.5f, f * 0.25f, 0.f);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
        const float* sgf = reinterpret_cast<const float*>(s_g[wave]);
        if (V == 0) {          // A: per entry, 24 lanes
            for (int e = 0; e < 8; ++e) {
                const uint32_t jc = s_jc[wave][e];
                if (ii < 3) atomicAdd(force + (size_t)jc * 32 + jj * 4 + ii, sgf[(e * 8 + jj) * 4 + ii]);
            }
        } else if (V == 1) {   // B: strided
            const uint32_t slot = s_jc[wave][lane >> 3] * 8 + (lane & 7);
            const float4 g = s_g[wave][lane];
            atomicAdd(force + (size_t)slot * 4, g.x); atomicAdd(force + (size_t)slot * 4 + 1, g.y);
            atomicAdd(force + (size_t)slot * 4 + 2, g.z);
        } else if (V == 2) {   // C: packed
            for (int kk = 0; kk < 3; ++kk) {
                const int fidx = kk * 64 + lane, atom = fidx / 3, comp = fidx - atom * 3;
                const uint32_t slot = s_jc[wave][atom >> 3] * 8 + (atom & 7);
                atomicAdd(force + (size_t)slot * 4 + comp, sgf[atom * 4 + comp]);
            }
        } else if (V == 3) {   // D: two entries per instruction, 48 lanes
            if (ii < 6) {
                const int half = ii >= 3, comp = ii - 3 * half;
                for (int kk = 0; kk < 4; ++kk) {
                    const int e = 2 * kk + half;
                    const uint32_t jc = s_jc[wave][e];
                    atomicAdd(force + (size_t)jc * 32 + jj * 4 + comp, sgf[(e * 8 + jj) * 4 + comp]);
                }
            }
        }
        acc += f * 1e-9f;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
    }
    if (acc == 12345.678f) force[0] = acc;
}

template <int V> float run(float* f, int nc, int tiles, int nch, int work) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<V><<<tiles, 256>>>(f, nc, nch, work);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) k<V><<<tiles, 256>>>(f, nc, nch, work);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 5;
}

int main() {
    int tiles = 16400, nc = tiles * 8;
    float* f; CK(hipMalloc(&f, (size_t)nc * 8 * 16)); CK(hipMemset(f, 0, (size_t)nc * 8 * 16));
    int nch = 7;   // 16400 tiles x 4 waves x 7 chunks x 8 = 3.7 M entries
    for (int work : {0, 400, 1600}) {
        printf("work %4d: none %.3f | A 8x24 %.3f | B 3x64 strided %.3f | C 3x64 packed %.3f | D 4x48 %.3f ms\n", work,
               run<9>(f, nc, tiles, nch, work), run<0>(f, nc, tiles, nch, work), run<1>(f, nc, tiles, nch, work),
               run<2>(f, nc, tiles, nch, work), run<3>(f, nc, tiles, nch, work));
    }
    return 0;
}
