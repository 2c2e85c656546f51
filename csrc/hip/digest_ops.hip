// SHA-256 of fixed-size chunks of arena rows: the leaves of the tree digest that a client
// signs and the aggregating side verifies (ops/digest_ops.py).  The reference signs the
// serialised update on the host (nanofed/server/validation.py:138-212); here only
// 32 bytes per chunk (1 KiB, or 64 KiB for rows beyond 16 MiB) leave the device and the
// host hashes those.
//
// One thread hashes one chunk (a whole number of 64-byte blocks) with the standard
// padding for its length; thread i of a row reads bytes [i * chunk, (i + 1) * chunk).

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__constant__ uint32_t kK[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u,
    0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu,
    0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu,
    0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u,
    0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu,
    0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu,
    0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u,
    0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u,
    0xc67178f2u};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

// one compression: w[0..15] are the block's words (big-endian values)
__device__ void compress(uint32_t (&h)[8], uint32_t (&w)[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    // 4 x 16 rounds; the inner loop is unrolled so that the 16-word schedule stays in
    // registers (indices are compile-time constants)
#pragma unroll 1
    for (int t0 = 0; t0 < 64; t0 += 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (t0 > 0) {
                const uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
                const uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3);
                const uint32_t s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
                w[i] = w[i] + s0 + w[(i + 9) & 15] + s1;
            }
            const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
            const uint32_t ch = (e & f) ^ (~e & g);
            const uint32_t t1 = hh + S1 + ch + kK[t0 + i] + w[i];
            const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
            const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
            const uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

__global__ void __launch_bounds__(64)
sha256_chunks_kernel(const uint8_t* __restrict__ data, long row_bytes, long row_stride,
                     int chunk_bytes, int chunks, uint8_t* __restrict__ out) {
    const int chunk = blockIdx.x * 64 + threadIdx.x;
    if (chunk >= chunks) return;
    const int row = blockIdx.y;
    const long begin = static_cast<long>(chunk) * chunk_bytes;
    const long len = min(static_cast<long>(chunk_bytes), row_bytes - begin);  // multiple of 64
    const uint32_t* src =
        reinterpret_cast<const uint32_t*>(data + static_cast<long>(row) * row_stride + begin);
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                     0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint32_t w[16];
    for (long blk = 0; blk < len / 64; ++blk) {
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = __builtin_bswap32(src[blk * 16 + i]);
        compress(h, w);
    }
    // padding block: 0x80, zeros, the length in bits (big-endian 64 bit)
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = 0u;
    w[0] = 0x80000000u;
    const unsigned long long bits = static_cast<unsigned long long>(len) * 8ull;
    w[14] = static_cast<uint32_t>(bits >> 32);
    w[15] = static_cast<uint32_t>(bits);
    compress(h, w);
    uint32_t* dst = reinterpret_cast<uint32_t*>(
        out + (static_cast<long>(row) * chunks + chunk) * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = __builtin_bswap32(h[i]);
}

}  // namespace

extern "C" {

// out[row][chunk][32] = SHA-256 of bytes [chunk * chunk_bytes, ...) of row `row` (the last
// chunk may be shorter).  row_bytes and chunk_bytes are multiples of 64, data is 4-byte
// aligned, row_stride (bytes between rows) a multiple of 4.
int nf_sha256_chunks(const void* data, long row_bytes, long row_stride, int rows, int chunk_bytes,
                     void* out, hipStream_t s) {
    if (rows < 1 || rows > 65535 || row_bytes < 64 || row_bytes % 64 != 0 || chunk_bytes < 64 ||
        chunk_bytes % 64 != 0 || row_stride % 4 != 0 || row_stride < row_bytes)
        return -1;
    if ((reinterpret_cast<uintptr_t>(data) & 3) != 0 || (reinterpret_cast<uintptr_t>(out) & 3) != 0)
        return -2;
    const long chunks = (row_bytes + chunk_bytes - 1) / chunk_bytes;
    if (chunks > (1L << 30)) return -3;
    hipLaunchKernelGGL(sha256_chunks_kernel, dim3(static_cast<unsigned>((chunks + 63) / 64), rows),
                       dim3(64), 0, s, static_cast<const uint8_t*>(data), row_bytes, row_stride,
                       chunk_bytes, static_cast<int>(chunks), static_cast<uint8_t*>(out));
    return static_cast<int>(hipGetLastError());
}

}  // extern "C"
