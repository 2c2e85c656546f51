// Index arithmetic shared by the kernels and a host build.
//
// An LDS swizzle or a tile order that is wrong does not fail loudly on the GPU: it
// reads another lane's data or, worse, writes outside its tile.  Every such map
// therefore lives here as a plain function that compiles for the device (included by
// the .hip sources) and for the host (csrc/host/index_math.cpp exports them, and
// tests/unit/test_kernel_index_math.py checks bounds, alignment and bijectivity over
// the whole domain the kernels use).
#pragma once

#if defined(__HIPCC__)
#define NF_IX __host__ __device__ __forceinline__
#else
#define NF_IX inline
#endif

namespace nfix {

// ---- bf16 GEMM (gemm_ops.hip) ----------------------------------------------------
// byte offset of piece `chunk` (8 k = 16 bytes) of row `row` in a k-contiguous operand
// tile with 128-byte rows (64 k per stage)
NF_IX int gemm_kc_off(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}
// byte offset of byte position `xb` of k row `k` in a k-major operand tile whose rows
// are `rowb` bytes (a multiple of 256): 64-byte groups are XOR-ed with k & 3
NF_IX int gemm_km_off(int rowb, int k, int xb) {
    return k * rowb + ((((xb >> 6) ^ (k & 3)) << 6) | (xb & 63));
}
// workgroup id -> linear tile index: ids are dealt round-robin to 8 XCDs; the ids of
// one XCD (congruent mod 8) get one contiguous range of tiles
NF_IX int gemm_tile_index(int id, int total_tiles) {
    const int q = total_tiles >> 3, r = total_tiles & 7;
    const int xcd = id & 7, slot = id >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + slot;
}

// ---- MNIST CNN (mnist_cnn.hip) -----------------------------------------------------
// conv2 forward band image: pixel p (= row * 26 + col) owns 64 bytes (32 channels), its
// 16-byte chunk `chunk` sits at chunk position chunk ^ ((col >> 1) & 3)
NF_IX int mnist_fwd_swz(int pixel, int col, int chunk) {
    return pixel * 64 + ((chunk ^ ((col >> 1) & 3)) << 4);
}
// conv2 data-gradient patch: 13 x 28 pixels of 128 bytes (64 channels)
constexpr int kMnistDgPatchRows = 13, kMnistDgPatchCols = 28;
NF_IX int mnist_dg_swz(int pr, int pc, int part) {
    return (pr * kMnistDgPatchCols + pc) * 128 + ((part ^ (((pc >> 1) + 5 * pr) & 7)) << 4);
}

// ---- MFMA 32x32x16 accumulator layout (nf_device.h) ----------------------------------
// row of register `reg` (0..15) of lane `lane` in the 32 x 32 result; its column is lane & 31
NF_IX int mfma32_acc_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

// ---- 3x3 convolution tiles (conv_ops.hip) ---------------------------------------------
// A tile is NI bands of RI image rows; its LDS patch holds, per band, RI + 2 rows of
// W + 2 pixels (the halo).  H, W, RI are powers of two.
// patch pixel pp of the tile whose first image row (index over b * H + y) is row0:
// element offset of its first channel in the NHWC activation tensor, -1 in the halo
// outside the image
NF_IX int conv_patch_source(int H, int W, int logH, int RI, int Cin, int row0, int pp) {
    const int PW = W + 2, PR = RI + 2;
    const int band = pp / (PR * PW);
    const int rem = pp - band * PR * PW;
    const int pr = rem / PW, pc = rem - pr * PW;
    const int grow = row0 + band * RI;
    const int img = grow >> logH, y0 = grow & (H - 1);
    const int yy = y0 + pr - 1, xx = pc - 1;
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) return -1;
    return ((img * H + yy) * W + xx) * Cin;
}
// patch index (without tap shift) of output pixel m of a tile
NF_IX int conv_patch_index(int W, int logW, int RI, int logRI, int m) {
    const int band = m >> (logRI + logW);
    const int r = (m >> logW) & (RI - 1);
    const int xx = m & (W - 1);
    return (band * (RI + 2) + r) * (W + 2) + xx;
}

}  // namespace nfix
