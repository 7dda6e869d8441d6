// Micro-benchmark: how fast does a CU retire ds_add_f32?  The PME charge spread accumulates 64 mesh points per atom in an LDS
// canvas; two different kernels (one workgroup per tile, one per mesh brick) both ran at ~0.4 lane-adds per clock and CU.
// Variants (one 19^3-float canvas per 256-thread workgroup, R rounds of 16 updates per lane):
//   0  ds_add_f32, canvas addressing of the spread (16 atoms x 4 y-lanes per wave, 4 x 4 (x, z) points per lane)
//   1  ds_add_f32, every lane its own bank (address = lane + 64 k): the conflict-free ceiling
//   2  plain read + add + write (NOT atomic: wrong sums, right traffic), canvas addressing
//   3  ds_add_rtn_f32 (returning), canvas addressing
//   4  ds_add_u32 (integer), canvas addressing
//   5  ds_add_f32, all 64 lanes of a wave into one atom's 4 x 4 x 4 points (lane = point), 16 rounds per atom
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_rate.hip -o lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int CB = 19, VOL = CB * CB * CB;

template <int V>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, int rounds) {
    __shared__ float s_q[VOL];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < VOL; i += 256) s_q[i] = 0.f;
    __syncthreads();
    unsigned h = (blockIdx.x * 256u + (unsigned)(tid >> 2)) * 2654435761u + 12345u;
    const int yb = tid & 3;
    float ret = 0.f;
    for (int r = 0; r < rounds; ++r) {
        h = h * 1664525u + 1013904223u;
        const int lx = (h >> 8) & 15, ly = (h >> 14) & 15, lz = (h >> 20) & 15;       // one "atom" per 4 lanes
        const float v = 1.0f + (float)(h & 255u) * 0.001f;
        if (V == 0 || V == 2 || V == 3 || V == 4) {
            float* row = s_q + (lx * CB + ly + yb) * CB + lz;
#pragma unroll
            for (int qa = 0; qa < 4; ++qa) {
#pragma unroll
                for (int qc = 0; qc < 4; ++qc) {
                    float* p = row + qa * CB * CB + qc;
                    if (V == 0) atomicAdd(p, v);
                    else if (V == 2) { *(volatile float*)p = *(volatile float*)p + v; }
                    else if (V == 3) ret += atomicAdd(p, v);
                    else atomicAdd(reinterpret_cast<unsigned*>(p), (unsigned)(h & 3u));
                }
            }
        } else if (V == 1) {
#pragma unroll
            for (int q = 0; q < 16; ++q) atomicAdd(s_q + lane + 64 * ((q + (int)(h >> 28)) & 63), v);
        } else {      // V == 5: the wave works on ONE atom at a time: lane = (a, b, c) point of its 4^3 support
            const unsigned hw = __builtin_amdgcn_readfirstlane(h);
            const int wx = (hw >> 8) & 15, wy = (hw >> 14) & 15, wz = (hw >> 20) & 15;
            const int a = lane >> 4, b = (lane >> 2) & 3, c = lane & 3;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int ox = (wx + q) & 15, oy = (wy + 3 * q) & 15;
                atomicAdd(s_q + ((ox + a) * CB + oy + b) * CB + wz + c, v);
            }
        }
    }
    __syncthreads();
    float s = ret;
    for (int i = tid; i < VOL; i += 256) s += s_q[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V>
static void run(const char* name, float* d_out, int blocks, int rounds) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d_out, rounds);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d_out, rounds);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double adds = (double)blocks * 256 * rounds * 16;
    printf("%-52s %8.1f us  %7.1f G lane-adds/s  = %5.2f per clock and CU (2.4 GHz, 256 CUs)\n", name, ms * 1e3, adds / ms * 1e-6,
           adds / (ms * 1e-3) / 2.4e9 / 256);
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 2560, rounds = argc > 2 ? atoi(argv[2]) : 64;
    float* d_out; CK(hipMalloc(&d_out, sizeof(float) * blocks * 256));
    printf("%d workgroups x 256 threads, %d rounds x 16 updates per lane, canvas %d^3 floats\n", blocks, rounds, CB);
    run<0>("0 ds_add_f32, spread addressing (16 atoms/wave)", d_out, blocks, rounds);
    run<1>("1 ds_add_f32, conflict-free", d_out, blocks, rounds);
    run<2>("2 plain read-add-write, spread addressing", d_out, blocks, rounds);
    run<3>("3 ds_add_rtn_f32, spread addressing", d_out, blocks, rounds);
    run<4>("4 ds_add_u32, spread addressing", d_out, blocks, rounds);
    run<5>("5 ds_add_f32, one atom per wave, lane = point", d_out, blocks, rounds);
    return 0;
}
