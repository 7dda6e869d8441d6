// mdx_hostutil.cpp - host-only helpers of libmdx.so that need x86 intrinsics (a HIP translation unit cannot include <immintrin.h>).
// The fingerprint of a system's static arrays, by which the stateless scorer (mdx_single_point, mdx_api.hip) recognises the next
// pose of the molecules it scored last  [ref: compute_energy_snapshot is called pose after pose, src/docking/mod.rs:235].
#include <cstdint>
#include <cstddef>
#include <cstring>
#if defined(__x86_64__) || defined(__i386__)
#define MDX_HOST_X86 1
#include <immintrin.h>
#endif

static uint64_t fp_mix_scalar(uint64_t h, const void* p, size_t bytes) {
    // four independent multiply-rotate lanes over 32-byte blocks (the multiplies pipeline: ~8 B per cycle), folded at the end
    const unsigned char* b = (const unsigned char*)p;
    const uint64_t K = 0x9E3779B97F4A7C15ull;
    uint64_t x0 = h ^ 0x243F6A8885A308D3ull, x1 = h ^ 0x13198A2E03707344ull, x2 = h ^ 0xA4093822299F31D0ull, x3 = h ^ 0x082EFA98EC4E6C89ull;
    auto rotl = [](uint64_t v, int r) { return (v << r) | (v >> (64 - r)); };
    size_t k = 0;
    for (; k + 32 <= bytes; k += 32) {
        uint64_t w[4]; std::memcpy(w, b + k, 32);
        x0 = rotl(x0 ^ w[0], 29) * K; x1 = rotl(x1 ^ w[1], 31) * K; x2 = rotl(x2 ^ w[2], 33) * K; x3 = rotl(x3 ^ w[3], 37) * K;
    }
    for (; k + 8 <= bytes; k += 8) { uint64_t w; std::memcpy(&w, b + k, 8); x0 = rotl(x0 ^ w, 29) * K; }
    uint64_t w = 0; std::memcpy(&w, b + k, bytes - k);
    x1 = rotl(x1 ^ w ^ (uint64_t)bytes, 31) * K;
    h = (x0 ^ rotl(x1, 17) ^ rotl(x2, 31) ^ rotl(x3, 47)) * 0xBF58476D1CE4E5B9ull;
    return h ^ (h >> 31);
}
// The same job on AVX2 hosts, 128 bytes per iteration: four 256-bit accumulators, each adding the 32 x 32 -> 64-bit product of
// the two halves of every keyed 64-bit word plus the word itself with its halves swapped (the accumulation step of XXH3).  ~30 B
// per cycle against ~8: the 6 MB of static arrays of a 51 k-atom complex take ~70 us instead of ~230, which hides behind the
// 150 us the device needs for the pose (mdx_single_point).  Not a cryptographic hash: it tells a docking loop's next pose from
// another molecule set.  The key advances with the stripe (as XXH3 walks its secret): two arrays that hold the same 128-byte blocks
// in a different order - charges of atoms 0-31 swapped with those of atoms 32-63 - mix differently.
#ifdef MDX_HOST_X86
__attribute__((target("avx2"))) static uint64_t fp_mix_avx2(uint64_t h, const void* p, size_t bytes) {
    const unsigned char* b = (const unsigned char*)p;
    const __m256i key[4] = {_mm256_set_epi64x(0x243F6A8885A308D3ll, 0x13198A2E03707344ll, (long long)0xA4093822299F31D0ull, 0x082EFA98EC4E6C89ll),
                            _mm256_set_epi64x(0x452821E638D01377ll, (long long)0xBE5466CF34E90C6Cull, (long long)0xC0AC29B7C97C50DDull, 0x3F84D5B5B5470917ll),
                            _mm256_set_epi64x((long long)0x9216D5D98979FB1Bull, (long long)0xD1310BA698DFB5ACull, 0x2FFD72DBD01ADFB7ll, (long long)0xB8E1AFED6A267E96ull),
                            _mm256_set_epi64x((long long)0xBA7C9045F12C7F99ull, 0x24A19947B3916CF7ll, 0x0801F2E2858EFC16ll, 0x636920D871574E69ll)};
    __m256i acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = _mm256_xor_si256(key[i], _mm256_set1_epi64x((long long)h));
    size_t k = 0;
    __m256i walk = _mm256_setzero_si256();
    const __m256i stride = _mm256_set_epi64x((long long)0x9E3779B97F4A7C15ull, (long long)0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ll, (long long)0x85EBCA77C2B2AE63ull);   // odd words
    for (; k + 128 <= bytes; k += 128) {
        walk = _mm256_add_epi64(walk, stride);
        for (int i = 0; i < 4; ++i) {
            const __m256i d = _mm256_loadu_si256((const __m256i*)(b + k + 32 * i));
            const __m256i dk = _mm256_xor_si256(d, _mm256_add_epi64(key[i], walk));
            acc[i] = _mm256_add_epi64(acc[i], _mm256_mul_epu32(dk, _mm256_shuffle_epi32(dk, 0x31)));
            acc[i] = _mm256_add_epi64(acc[i], _mm256_shuffle_epi32(d, 0x4E));
        }
    }
    uint64_t lanes[16];
    for (int i = 0; i < 4; ++i) _mm256_storeu_si256((__m256i*)(lanes + 4 * i), acc[i]);
    uint64_t x = h ^ (uint64_t)bytes;
    for (int i = 0; i < 16; ++i) { x = (x ^ lanes[i]) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; }
    return fp_mix_scalar(x, b + k, bytes - k);      // (the tail, and the final avalanche)
}
#endif
uint64_t mdx_fp_mix(uint64_t h, const void* p, size_t bytes) {
    if (!p) return (h ^ 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
#ifdef MDX_HOST_X86
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && bytes >= 512) return fp_mix_avx2(h, p, bytes);
#endif
    return fp_mix_scalar(h, p, bytes);      // (any other host: the portable flavour)
}
