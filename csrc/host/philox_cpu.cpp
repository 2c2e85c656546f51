// CPU implementation of the Philox sampling entry points, sharing
// csrc/common/philox.h with the HIP kernels so that a CPU run draws the same
// stream as a GPU run (bit-identical for the uniform modes; equal up to libm
// rounding for the normal and Laplace modes).
//
// out[i] = (accumulate ? out[i] : 0) + scale * sample(seed, offset, first + i)

#include "../common/philox.h"

extern "C" {

void nf_philox_cpu(float* out, long n, unsigned long long seed,
                   unsigned long long offset, long first, int mode, float scale,
                   int accumulate) {
    long i = 0;
    while (i < n) {
        const unsigned long long e = static_cast<unsigned long long>(first + i);
        float s[4];
        nf::philox_samples(seed, offset, e >> 2, mode, s);
        for (int lane = static_cast<int>(e & 3); lane < 4 && i < n; ++lane, ++i) {
            const float v = scale * s[lane];
            out[i] = accumulate ? out[i] + v : v;
        }
    }
}

}  // extern "C"
