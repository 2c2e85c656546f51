// Philox4x32-10 counter-based generator, usable from host C++ and HIP device
// code.  A sample is a pure function of (seed, stream offset, element index):
// element i takes lane (i & 3) of the block with counter (i >> 2), so results
// do not depend on the launch geometry, the number of GPUs, or on how a flat
// arena is split into calls - and a hipGraph replay reproduces them.
//
// uniform01     : 24-bit mantissa uniform in [0, 1)
// normal pair   : Box-Muller on (u1 in (0,1], u2 in [0,1))
// laplace       : inverse CDF, sign(v) * log1p(-2|v|) with v = u - 1/2 (the same
//                 sign convention as the host generator in
//                 nanofed_amd/privacy/noise/generators.py)
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define NF_HD __host__ __device__ __forceinline__
#else
#define NF_HD inline
#endif

namespace nf {

struct Philox4 {
    uint32_t v[4];
};

NF_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return static_cast<uint32_t>((static_cast<uint64_t>(a) * b) >> 32);
#endif
}

NF_HD Philox4 philox4x32_10(uint64_t seed, uint64_t offset, uint64_t block) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t k0 = static_cast<uint32_t>(seed);
    uint32_t k1 = static_cast<uint32_t>(seed >> 32);
    uint32_t c0 = static_cast<uint32_t>(block);
    uint32_t c1 = static_cast<uint32_t>(block >> 32);
    uint32_t c2 = static_cast<uint32_t>(offset);
    uint32_t c3 = static_cast<uint32_t>(offset >> 32);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    return Philox4{{c0, c1, c2, c3}};
}

NF_HD float u01(uint32_t x) { return static_cast<float>(x >> 8) * 5.9604644775390625e-8f; }
NF_HD float u01_open_low(uint32_t x) {
    return (static_cast<float>(x >> 8) + 1.0f) * 5.9604644775390625e-8f;
}

// Four samples for block `b`: mode 0 uniform [0,1), 1 standard normal,
// 2 unit Laplace, 3 uniform [-1,1).
NF_HD void philox_samples(uint64_t seed, uint64_t offset, uint64_t b, int mode,
                          float out[4]) {
    const Philox4 r = philox4x32_10(seed, offset, b);
    if (mode == 1) {
        const float TWO_PI = 6.283185307179586f;
        for (int h = 0; h < 2; ++h) {
            const float u1 = u01_open_low(r.v[2 * h]);
            const float u2 = u01(r.v[2 * h + 1]);
#if defined(__HIP_DEVICE_COMPILE__)
            // hardware log2 and sin/cos (their argument is in revolutions): a few
            // instructions instead of the range-reducing library routines; agrees
            // with the host path to about 1e-6 absolute
            (void)TWO_PI;
            const float rad = sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
            out[2 * h] = rad * __builtin_amdgcn_cosf(u2);
            out[2 * h + 1] = rad * __builtin_amdgcn_sinf(u2);
#else
            const float rad = sqrtf(-2.0f * logf(u1));
            out[2 * h] = rad * cosf(TWO_PI * u2);
            out[2 * h + 1] = rad * sinf(TWO_PI * u2);
#endif
        }
    } else if (mode == 2) {
        for (int j = 0; j < 4; ++j) {
            const float v = u01(r.v[j]) - 0.5f;
            const float s = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
            out[j] = s * log1pf(-2.0f * fabsf(v));
        }
    } else if (mode == 3) {
        for (int j = 0; j < 4; ++j) out[j] = 2.0f * u01(r.v[j]) - 1.0f;
    } else {
        for (int j = 0; j < 4; ++j) out[j] = u01(r.v[j]);
    }
}

}  // namespace nf
