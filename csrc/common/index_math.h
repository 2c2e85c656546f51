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

// linear tile index -> tile row * 65536 + tile column.  group_m <= 1: row after row.  Else
// bands of group_m tile rows, each walked column after column, so that consecutive tiles (the
// ones in flight on an XCD) share group_m row panels and in_flight / group_m column panels;
// the last band may have fewer rows.
NF_IX int gemm_tile_rc(int t, int tiles_m, int tiles_n, int group_m) {
    if (group_m <= 1) {
        const int tm = t / tiles_n;
        return tm * 65536 + (t - tm * tiles_n);
    }
    const int band_sz = group_m * tiles_n;
    const int band = t / band_sz, r = t - band * band_sz;
    const int left = tiles_m - band * group_m;
    const int rows = left < group_m ? left : group_m;
    const int tn = r / rows;
    return (band * group_m + (r - tn * rows)) * 65536 + tn;
}

// ---- MNIST CNN (mnist_cnn.hip) -----------------------------------------------------
// Both convolution images are padded, not swizzled: a fragment address is then one
// per-lane base plus a compile-time offset (tap row, tap column, k step), so the inner
// loops compute no addresses at all.
//
// conv2 forward band image: 10 rows of 26 pixels, 64 bytes (32 channels) per pixel, rows
// 1680 bytes apart (26 * 64 + 16).  In 16-byte units a pixel is 4 and a row 105 (odd):
// with the window-major row order of the MFMA blocks (lane n = position n & 3 of pooling
// window n >> 2 of a 4 x 8 pixel block) the 16 lanes of a ds_read_b128 group then read
// 16 different 16-byte slots of the 256-byte bank row.
constexpr int kMnistFwdPixPitch = 64, kMnistFwdRowPitch = 26 * 64 + 16;
NF_IX int mnist_fwd_off(int row, int col, int chunk) {
    return row * kMnistFwdRowPitch + col * kMnistFwdPixPitch + (chunk << 4);
}
// (dy, dx) of row m of a forward MFMA block inside its 4 x 8 pixel block
NF_IX int mnist_fwd_block_dy(int m) { return 2 * (m >> 4) + ((m >> 1) & 1); }
NF_IX int mnist_fwd_block_dx(int m) { return 2 * ((m >> 2) & 3) + (m & 1); }
// conv2 data-gradient patch: 13 rows of 28 pixels, 144 bytes per pixel (128 of channels
// + 16), rows 4256 bytes apart (28 * 144 + 224).  In 16-byte units a pixel is 9 and a row
// 266 = 10 (mod 16), so the pixels of a 32-pixel block - consecutive pixels of the
// 26-wide image, across row ends - advance by 9 (mod 16) each: any 16 lanes of a read
// group hit 16 different slots, whatever the tap shift.
constexpr int kMnistDgPatchRows = 13, kMnistDgPatchCols = 28;
constexpr int kMnistDgPixPitch = 144, kMnistDgRowPitch = 28 * 144 + 224;
NF_IX int mnist_dg_off(int pr, int pc, int part) {
    return pr * kMnistDgRowPitch + pc * kMnistDgPixPitch + (part << 4);
}
// the four groups of 16 lanes that a ds_read_b128 serves in one LDS cycle each
NF_IX int lds_b128_group(int lane) {
    const int l = lane & 31;
    const int g = (l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28)) ? 0 : 1;
    return g + 2 * (lane >> 5);
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

// Forward / data-gradient kernel: the patch image is padded per row and per band so that
// the 16 lanes of a ds_read_b128 group - which, for images narrower than 32 pixels, sit in
// several image rows (and bands) - fall on 16 different 16-byte slots of the 256-byte bank
// row.  Pixel (band, pr, pc) of the patch starts at band * bp + pr * rp + pc * pitch bytes;
// rp >= (W + 2) * pitch, bp >= (RI + 2) * rp, all multiples of 16.
// patch pixel pp (numbered band, row, column) -> byte offset
NF_IX int conv_patch_byte(int W, int RI, int pp, int pitch, int rp, int bp) {
    const int PW = W + 2, PR = RI + 2;
    const int band = pp / (PR * PW);
    const int rem = pp - band * PR * PW;
    const int pr = rem / PW, pc = rem - pr * PW;
    return band * bp + pr * rp + pc * pitch;
}
// output pixel m of a tile -> byte offset of its patch pixel without tap shift (tap
// (ky, kx) adds ky * rp + kx * pitch)
NF_IX int conv_patch_frag_byte(int W, int logW, int RI, int logRI, int m, int pitch, int rp,
                               int bp) {
    const int band = m >> (logRI + logW);
    const int r = (m >> logW) & (RI - 1);
    const int xx = m & (W - 1);
    return band * bp + r * rp + xx * pitch;
}
// LDS cycles of the fragment reads of a tile of tp pixels (32 pixels per read, 16-byte
// pieces): per lane group the largest number of different addresses on one slot, summed
// over the two groups of each 32-pixel block.  2 per block = conflict-free.
NF_IX int conv_patch_read_cycles(int W, int logW, int RI, int logRI, int tp, int pitch, int rp,
                                 int bp) {
    // lanes of the two ds_read_b128 groups of a 32-lane half
    const int groups[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                               {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int cycles = 0;
    for (int m0 = 0; m0 < tp; m0 += 32)
        for (int gi = 0; gi < 2; ++gi) {
            // the 16 lanes read 16 different pixels: a slot costs one cycle per lane on it
            int on_slot[16] = {0}, worst = 1;
            for (int i = 0; i < 16; ++i) {
                const int a = conv_patch_frag_byte(W, logW, RI, logRI, m0 + groups[gi][i], pitch,
                                                   rp, bp);
                const int n = ++on_slot[(a >> 4) & 15];
                if (n > worst) worst = n;
            }
            cycles += worst;
        }
    return cycles;
}
// Smallest padding (in 16-byte steps, fewer than 16 per row and per band) with the fewest
// read cycles.
NF_IX void conv_patch_pitches(int W, int logW, int RI, int logRI, int NI, int tp, int pitch,
                              int* rp, int* bp) {
    int best_cycles = 1 << 30, best_bytes = 1 << 30;
    for (int e = 0; e < 16; ++e)
        for (int be = 0; be < (NI > 1 ? 16 : 1); ++be) {
            const int r = (W + 2) * pitch + 16 * e, b = (RI + 2) * r + 16 * be;
            const int c = conv_patch_read_cycles(W, logW, RI, logRI, tp, pitch, r, b);
            if (c < best_cycles || (c == best_cycles && NI * b < best_bytes)) {
                best_cycles = c;
                best_bytes = NI * b;
                *rp = r;
                *bp = b;
            }
        }
}

}  // namespace nfix
