// 3x3 convolution, stride 1, zero padding 1, for channels-last bf16 activations
// on gfx950: forward (which also serves the data gradient, on flipped weights)
// and the weight gradient, as implicit GEMMs on the 32x32x16 bf16 MFMA.
//
// Layouts: activations [B][H][W][C] (channels-last), weights [Cout][3][3][Cin]
// (the channels-last storage of a [Cout, Cin, 3, 3] tensor).  H and W are powers
// of two, channel counts multiples of 64.
//
// Forward.  A workgroup of four waves computes a tile of TP consecutive pixels
// (raster order over image, row, column) by 64 output channels.  The pixels of
// a tile form NI bands of RI image rows; the bands are staged in LDS with a halo
// of one pixel, so the nine taps are nine shifted reads of one patch.  The
// reduction runs over chunks of CH (32 or 64) input channels: per chunk the patch
// and the [64 oc][9 taps][CH ic] weight slice are staged (pitch 2 CH + 16 bytes:
// 16-byte row-major fragment reads), which makes 9 CH / 16 MFMA k-steps; the next
// chunk is fetched into registers meanwhile.  In images narrower than 32 pixels the 32
// pixels of a fragment read lie in several patch rows, so the rows (and bands) of the
// patch are padded by a few 16-byte steps, chosen on the host so that the lanes of every
// ds_read_b128 group hit different bank groups (nfix::conv_patch_pitches).  Waves are arranged PXW x KSW: PXW
// pixel groups of 64 pixels, KSW waves sharing the k-steps of one pixel group; the
// partial accumulators of a pixel group are summed through LDS at the end.  Deep
// layers (few pixels, many channels) use 1 x 4 with chunks of 64 so that a layer
// still makes >= 128 workgroups; wide layers 2 x 2 with chunks of 32; launches large
// enough for 512 workgroups of 256 pixels use 4 x 1: every wave runs all k-steps of its
// 64 pixels, a chunk's weight slice (46 KB through registers and ds_write_b128, the
// largest LDS cost of the kernel) is staged once for twice the pixels, there are twice
// the MFMAs between two barriers and nothing is summed through LDS at the end
// (8 x 64 images: 84 -> 65 us forward, 78 -> 62 us data gradient at 32x32x64;
// 58 -> 48 us at 8x8x256, four images per tile).
//
// Weight gradient.  dW[oc][tap][ic] = sum over pixels dY[px][oc] * X[px + tap][ic]:
// both operands have the reduction index (pixels) as the slow dimension in
// memory, so both fragments come from transposing LDS reads.  A workgroup owns a
// 64 oc x 64 ic tile for a range of pixels; each wave accumulates a 32 x 32
// sub-tile for all nine taps (144 accumulator registers; one dY fragment feeds
// nine MFMAs).  Pixel ranges write partial slabs (bf16: the kernel is bound by
// this traffic, and the sum itself is bf16) that a second kernel sums in fp32 in a
// fixed order (no atomics: results do not depend on scheduling).
//
// Grouped launches (the *_g entry points): the batch holds G groups of B images, one
// after the other - the clients that share a GPU, each with weights of its own.  Group
// g's weights start `w_stride` elements after those of group g - 1 (rows of a parameter
// arena); a pixel tile never spans two groups, so a workgroup picks its weights from its
// tile index.  The weight gradient sums each group's pixel ranges separately.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "nf_device.h"

using namespace nfd;

namespace {

constexpr int kBRows = 64 * 9;            // weight rows of an output-channel tile
constexpr int kMaxAPieces = 7;            // 16-byte patch pieces per thread and chunk (at most)
constexpr int kRedBytes = 4 * 4 * 16 * 64 * 4;  // accumulators of four waves
constexpr int kColsumBytes = 4 * 2 * 64 * 4;    // column-sum staging behind them
constexpr int kFwdLdsCap = 124 * 1024;    // dynamic LDS the forward kernels may ask for

// forward staging for chunks of CH input channels
template <int CH>
struct FwdStage {
    static constexpr int kPitch = CH * 2 + 16;        // bytes per staged pixel / weight row
    static constexpr int kPieces = CH / 8;            // 16-byte pieces per row
    static constexpr int kBBytes = kBRows * kPitch;
    static constexpr int kBPerThread = kBRows * kPieces / 256;
    static constexpr int kMaxPatch = kMaxAPieces * 256 / kPieces;  // 448 (CH 32), 224 (CH 64)
    static constexpr int kSteps = 9 * CH / 16;        // MFMA k-steps per chunk
    // data-gradient variant: weights staged [tap][CH oc][64 ic], ic contiguous
    static constexpr int kTPitch = 64 * 2 + 32;       // transposing reads
    static constexpr int kTBBytes = 9 * CH * kTPitch;
    static_assert(9 * CH * 8 / 256 == kBPerThread, "same piece count in both layouts");
};

struct ConvGeo {
    int H, W, logH, logW;  // image size
    int RI, logRI;         // image rows per band
    int NI;                // bands per tile
    int np;                // patch pixels of a tile: NI * (RI + 2) * (W + 2)
    int Cin, Cout;
    int tpg;               // pixel tiles per group (of the launch's tile size)
    long wgs;              // weight elements between consecutive groups
    int rp, bp;            // forward kernel: bytes between patch rows / bands in LDS (padded)
    int xorder;            // forward kernel: 1-D launch in weight-sharing order (see launch_fwd)
};

// Patch pixel pp of the tile whose first image row (index over b * H + y) is row0:
// element offset of its first channel in the activation tensor, or -1 in the halo
// outside the image.
__device__ __forceinline__ int patch_source(const ConvGeo& g, int row0, int pp) {
    return nfix::conv_patch_source(g.H, g.W, g.logH, g.RI, g.Cin, row0, pp);
}

// Patch index (without tap shift) of pixel m of a tile.
__device__ __forceinline__ int patch_index(const ConvGeo& g, int m) {
    return nfix::conv_patch_index(g.W, g.logW, g.RI, g.logRI, m);
}
// The same as a byte offset into the forward kernel's padded patch image.
__device__ __forceinline__ int patch_frag_byte(const ConvGeo& g, int m, int pitch) {
    return nfix::conv_patch_frag_byte(g.W, g.logW, g.RI, g.logRI, m, pitch, g.rp, g.bp);
}

// What the data-gradient launch needs to finish the elementwise part of the BatchNorm
// backward pass that consumes its result (kBnBwd): the launch then stores
//   gz = (dX + add) where mask > 0, else 0
// (mask: the ReLU output that followed the BatchNorm, add: the gradient from the second
// branch that read that output, may be null) and its column sums are those of gz and
// gz * xpre (xpre: the BatchNorm's input), which is all the BatchNorm backward needs from
// a pass over the pixels.  All three tensors have the shape of the launch's output.
// Where the BatchNorm had no residual input its ReLU output is a function of xpre alone, and
// the mask is recomputed from xpre with the arithmetic of the kernel that wrote the output
// (bn_apply_kernel: fma(x, sc, sh) with sc = gamma / std, sh = fma(-mean, sc, beta)) instead of
// being read: mask == nullptr, stats [groups][2][C] (mean, 1/std), gamma / beta of group 0 with
// pstride floats between groups.
struct BnBwdIn {
    const bf16* mask;
    const bf16* xpre;
    const bf16* add;
    const float* stats;
    const float* gamma;
    const float* beta;
    long pstride;
};

// kDgrad: the launch computes a data gradient.  x is then dY, y is dX, g.Cin / g.Cout are
// the layer's output / input channel counts, and w is the layer's weight as it is,
// [g.Cin][9][g.Cout]: tap t of the launch uses the layer's tap 8 - t, and the weight
// slice is staged with the reduction index (the layer's oc) as the slow dimension and
// read through transposing LDS loads - no flipped copy of the weights is made.
// NAP: patch pieces per thread this instantiation stages (ceil(patch pixels * pieces per
// pixel / 256) of the shapes it is launched for): an unused one still costs its
// predicated load and store in every chunk (5 us of 65 at 32x32x64).
template <int PXW, int KSW, int CH, bool kDgrad, int NAP, bool kBnBwd = false>
__global__ void __launch_bounds__(256)
conv3x3_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w, bf16* __restrict__ y,
                   float* __restrict__ colsum, ConvGeo g, BnBwdIn fz) {
    static_assert(PXW * KSW == 4, "four waves");
    static_assert(kDgrad || !kBnBwd, "the BatchNorm backward epilogue belongs to the data gradient");
    static_assert(NAP >= 1 && NAP <= kMaxAPieces, "named prefetch registers");
    using St = FwdStage<CH>;
    constexpr int TP = 64 * PXW;
    constexpr int kPitch = St::kPitch, kPieces = St::kPieces;
    constexpr int kLog = CH == 32 ? 2 : 3;  // log2(kPieces)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kBRegion = kDgrad ? St::kTBBytes : St::kBBytes;
    char* s_b = smem;                 // [64 oc * 9 taps][CH ic]  (kDgrad: [9][CH][64])
    char* s_a = smem + kBRegion;      // [patch pixel][CH ic]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pw = wave % PXW, kw = KSW == 1 ? 0 : wave / PXW;
    int tile = blockIdx.x, oc0 = blockIdx.y * 64;
    if (g.xorder) {
        // the workgroups of one XCD (ids congruent mod 8) take a contiguous run of
        // (group, output-channel tile, pixel tile) triples, pixel tiles fastest: the pixel
        // tiles that read the same weight slice follow each other on one L2
        const int t = nfix::gemm_tile_index(blockIdx.x, gridDim.x);
        const int blk = t / g.tpg, oct = g.Cout >> 6;
        const int grp = blk / oct;
        tile = grp * g.tpg + (t - blk * g.tpg);
        oc0 = (blk - grp * oct) * 64;
    }
    const int row0 = (tile * TP) >> g.logW;
    w += (tile / g.tpg) * g.wgs;  // this group's weights

    // staging plan of the activation patch: pieces tid + 256 i, 16 bytes each
    int a_src[kMaxAPieces], a_dst[kMaxAPieces];
#pragma unroll
    for (int i = 0; i < kMaxAPieces; ++i) {
        const int q = tid + i * 256, pp = q >> kLog, p = q & (kPieces - 1);
        if (i < NAP && pp < g.np) {
            const int src = patch_source(g, row0, pp);
            a_src[i] = src < 0 ? -1 : src + p * 8;
            a_dst[i] = nfix::conv_patch_byte(g.W, g.RI, pp, kPitch, g.rp, g.bp) + p * 16;
        } else {
            a_src[i] = -1;
            a_dst[i] = -1;
        }
    }
    // fragment rows of this lane
    int a_row[2], b_row[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
        a_row[mb] = patch_frag_byte(g, pw * 64 + mb * 32 + (lane & 31), kPitch) + (lane >> 5) * 16;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
        b_row[nb] = (nb * 32 + (lane & 31)) * 9 * kPitch + (lane >> 5) * 16;

    f32x16 acc[2][2] = {{{0}, {0}}, {{0}, {0}}};
    const int s_lo = (St::kSteps * kw) / KSW, s_hi = (St::kSteps * (kw + 1)) / KSW;
    const int nchunks = g.Cin / CH;
    const bf16* wsrc = w + static_cast<long>(oc0) * 9 * g.Cin;

    // chunk c + 1 travels memory -> registers while chunk c is being multiplied
    // (named registers: an indexed array that lives across loop iterations ends up
    // in scratch memory)
#define NF_B_LIST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) \
    X(14) X(15) X(16) X(17)
#define NF_A_LIST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6)
#define NF_DECL_B(i) uint4 pb##i = make_uint4(0u, 0u, 0u, 0u);
#define NF_DECL_A(i) uint4 pa##i = make_uint4(0u, 0u, 0u, 0u);
    NF_B_LIST(NF_DECL_B)
    NF_A_LIST(NF_DECL_A)
    static_assert(St::kBPerThread <= 18 && kMaxAPieces == 7, "named prefetch registers");
    int chunk_off = 0;  // c * CH of the chunk being fetched
#define NF_LOAD_B(i)                                                                      \
    if constexpr (i < St::kBPerThread) {                                                  \
        if constexpr (kDgrad) {                                                           \
            const int q = tid + i * 256, row = q >> 3, p = q & 7;                         \
            const int tap = row / CH, kk = row - tap * CH;                                \
            pb##i = *reinterpret_cast<const uint4*>(                                      \
                w + (static_cast<long>(chunk_off + kk) * 9 + (8 - tap)) * g.Cout + oc0 +  \
                p * 8);                                                                   \
        } else {                                                                          \
            const int q = tid + i * 256, row = q >> kLog, p = q & (kPieces - 1);          \
            pb##i = *reinterpret_cast<const uint4*>(wsrc + static_cast<long>(row) * g.Cin + \
                                                    chunk_off + p * 8);                   \
        }                                                                                 \
    }
#define NF_LOAD_A(i)                                                                      \
    if constexpr (i < NAP) {                                                              \
        pa##i = make_uint4(0u, 0u, 0u, 0u);                                               \
        if (a_src[i] >= 0) pa##i = *reinterpret_cast<const uint4*>(x + a_src[i] + chunk_off); \
    }
#define NF_STORE_B(i)                                                                     \
    if constexpr (i < St::kBPerThread) {                                                  \
        if constexpr (kDgrad) {                                                           \
            const int q = tid + i * 256, row = q >> 3, p = q & 7;                         \
            *reinterpret_cast<uint4*>(s_b + row * St::kTPitch + p * 16) = pb##i;          \
        } else {                                                                          \
            const int q = tid + i * 256, row = q >> kLog, p = q & (kPieces - 1);          \
            *reinterpret_cast<uint4*>(s_b + row * kPitch + p * 16) = pb##i;               \
        }                                                                                 \
    }
#define NF_STORE_A(i)   \
    if constexpr (i < NAP) \
        if (a_dst[i] >= 0) *reinterpret_cast<uint4*>(s_a + a_dst[i]) = pa##i;
    NF_B_LIST(NF_LOAD_B)
    NF_A_LIST(NF_LOAD_A)
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        NF_B_LIST(NF_STORE_B)
        NF_A_LIST(NF_STORE_A)
        if (c + 1 < nchunks) {
            chunk_off = (c + 1) * CH;
            NF_B_LIST(NF_LOAD_B)
            NF_A_LIST(NF_LOAD_A)
        }
        __syncthreads();
#undef NF_B_LIST
#undef NF_A_LIST
#undef NF_DECL_B
#undef NF_DECL_A
#undef NF_LOAD_B
#undef NF_LOAD_A
#undef NF_STORE_B
#undef NF_STORE_A
#pragma unroll 3
        for (int s = s_lo; s < s_hi; ++s) {
            const int tap = s / (CH / 16), sub = s - tap * (CH / 16);
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int aoff = ky * g.rp + kx * kPitch + sub * 32;
            const int boff = tap * kPitch + sub * 32;
            const bf16x8 A0 = *reinterpret_cast<const bf16x8*>(s_a + a_row[0] + aoff);
            const bf16x8 A1 = *reinterpret_cast<const bf16x8*>(s_a + a_row[1] + aoff);
            bf16x8 B0, B1;
            if constexpr (kDgrad) {
                const char* bt = s_b + tap * (CH * St::kTPitch);
                B0 = load_tr_frag(bt, St::kTPitch, sub * 16, 0, lane);
                B1 = load_tr_frag(bt, St::kTPitch, sub * 16, 32, lane);
            } else {
                B0 = *reinterpret_cast<const bf16x8*>(s_b + b_row[0] + boff);
                B1 = *reinterpret_cast<const bf16x8*>(s_b + b_row[1] + boff);
            }
            acc[0][0] = mfma32(A0, B0, acc[0][0]);
            acc[0][1] = mfma32(A0, B1, acc[0][1]);
            acc[1][0] = mfma32(A1, B0, acc[1][0]);
            acc[1][1] = mfma32(A1, B1, acc[1][1]);
        }
    }

    // Epilogue.  The k-split partial accumulators of a pixel group are summed through LDS;
    // the bf16 results are laid out [pixel][64 oc] in LDS (the main loop's regions are
    // free by now) and leave as 16-byte pieces of whole 128-byte row segments: thread t owns
    // channel piece t & 7 of pixels (t >> 3) + 32 i, so its column sums stay in registers
    // over its pieces and everything that is read beside them (kBnBwd) is a 16-byte load.
    __syncthreads();
    constexpr int kBlocksPerWave = 4 / KSW;
    constexpr int kSP = 64 * 2 + 16;  // bytes per staged pixel
    static_assert(TP * kSP <= kRedBytes, "the staged tile ends before the column-sum staging");
    [[maybe_unused]] float* red = reinterpret_cast<float*>(smem);  // [wave][block mb*2+nb][reg][lane]
    if constexpr (KSW > 1) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    red[((wave * 4 + mb * 2 + nb) * 16 + r) * 64 + lane] = acc[mb][nb][r];
        __syncthreads();
    }
    bf16 vb[kBlocksPerWave][16];
#pragma unroll
    for (int j = 0; j < kBlocksPerWave; ++j) {
        [[maybe_unused]] const int blk = kw * kBlocksPerWave + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = 0.f;
            if constexpr (KSW > 1) {
#pragma unroll
                for (int k2 = 0; k2 < KSW; ++k2)
                    v += red[(((k2 * PXW + pw) * 4 + blk) * 16 + r) * 64 + lane];
            } else {
                v = acc[j >> 1][j & 1][r];  // the wave has run every k-step of its 64 pixels itself
            }
            vb[j][r] = f2bf(v);
        }
    }
    if constexpr (KSW > 1) __syncthreads();  // the partial sums have been read: reuse the region
    char* stage = smem;
#pragma unroll
    for (int j = 0; j < kBlocksPerWave; ++j) {
        const int blk = kw * kBlocksPerWave + j, mb = blk >> 1, nb = blk & 1;
        char* sp = stage + (pw * 64 + mb * 32) * kSP + (nb * 32 + (lane & 31)) * 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) *reinterpret_cast<bf16*>(sp + acc_row(r, lane) * kSP) = vb[j][r];
    }
    __syncthreads();

    constexpr int kPPT = TP / 32;  // pieces per thread
    const int chunk = tid & 7, prow = tid >> 3;
    const long gbase = (static_cast<long>(tile) * TP + prow) * g.Cout + oc0 + chunk * 8;
    const long gstep = 32L * g.Cout;
    [[maybe_unused]] bf16x8 mk[kPPT], xp[kPPT], ad[kPPT];
    [[maybe_unused]] float sc[8], sh[8];
    [[maybe_unused]] const bool remask = kBnBwd && fz.mask == nullptr;
    if constexpr (kBnBwd) {
        // everything the pieces need, in flight before the first is used
#pragma unroll
        for (int i = 0; i < kPPT; ++i) {
            xp[i] = *reinterpret_cast<const bf16x8*>(fz.xpre + gbase + i * gstep);
            if (!remask) mk[i] = *reinterpret_cast<const bf16x8*>(fz.mask + gbase + i * gstep);
        }
        if (remask) {
            const int grp = tile / g.tpg, c0 = oc0 + chunk * 8;
            const float* st = fz.stats + static_cast<long>(grp) * 2 * g.Cout;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                sc[k] = st[g.Cout + c0 + k] * fz.gamma[grp * fz.pstride + c0 + k];
                sh[k] = fmaf(-st[c0 + k], sc[k], fz.beta[grp * fz.pstride + c0 + k]);
            }
        }
        if (fz.add != nullptr) {
#pragma unroll
            for (int i = 0; i < kPPT; ++i)
                ad[i] = *reinterpret_cast<const bf16x8*>(fz.add + gbase + i * gstep);
        } else {
#pragma unroll
            for (int i = 0; i < kPPT; ++i)
#pragma unroll
                for (int k = 0; k < 8; ++k) ad[i][k] = f2bf(0.f);
        }
    }
    // column sums of the values as stored: y and y^2 (kBnBwd: gz and gz * xpre)
    float cs1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float cs2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kPPT; ++i) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + (prow + 32 * i) * kSP + chunk * 16);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float f = bf2f(v[k]);
            if constexpr (kBnBwd) {
                // (the rounding of the separate kernels: the data gradient to bf16, the sum of
                // the two branches to bf16)
                const bool on = remask ? fmaf(bf2f(xp[i][k]), sc[k], sh[k]) > 0.f
                                       : bf2f(mk[i][k]) > 0.f;
                f = on ? f + bf2f(ad[i][k]) : 0.f;
                o[k] = f2bf(f);
                f = bf2f(o[k]);
                cs2[k] = fmaf(f, bf2f(xp[i][k]), cs2[k]);
            } else {
                o[k] = v[k];
                cs2[k] = fmaf(f, f, cs2[k]);
            }
            cs1[k] += f;
        }
        *reinterpret_cast<bf16x8*>(y + gbase + i * gstep) = o;
    }
    // Per-channel sums over the tile's pixels, for a BatchNorm that follows (kBnBwd: whose
    // backward pass follows): colsum[tile][0][oc], colsum[tile][1][oc].
    if (colsum != nullptr) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                cs1[k] += __shfl_xor(cs1[k], off, 64);
                cs2[k] += __shfl_xor(cs2[k], off, 64);
            }
        }
        float* cs = reinterpret_cast<float*>(smem + kRedBytes);  // [wave][2][64]
        if (lane < 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                cs[(wave * 2 + 0) * 64 + lane * 8 + k] = cs1[k];
                cs[(wave * 2 + 1) * 64 + lane * 8 + k] = cs2[k];
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int stat = tid >> 6, oc = tid & 63;
            const float total = (cs[(0 * 2 + stat) * 64 + oc] + cs[(1 * 2 + stat) * 64 + oc]) +
                                (cs[(2 * 2 + stat) * 64 + oc] + cs[(3 * 2 + stat) * 64 + oc]);
            colsum[(static_cast<long>(tile) * 2 + stat) * g.Cout + oc0 + oc] = total;
        }
    }
}

// ---------------------------------------------------------------------------
// weights [Cout][9][Cin] -> [Cin][9][Cout] with the taps reversed: the forward
// kernel on these computes the data gradient.
__global__ void __launch_bounds__(256)
flip_weights_kernel(const bf16* __restrict__ w, bf16* __restrict__ wt, int Cin, int Cout) {
    // one workgroup per (64 oc, 64 ic, tap): transpose through LDS
    __shared__ bf16 t[64][66];
    const int tap = blockIdx.x, oc0 = blockIdx.y * 64, ic0 = blockIdx.z * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + i * 256, o = e >> 6, c = e & 63;
        t[o][c] = w[(static_cast<long>(oc0 + o) * 9 + tap) * Cin + ic0 + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + i * 256, c = e >> 6, o = e & 63;
        wt[(static_cast<long>(ic0 + c) * 9 + (8 - tap)) * Cout + oc0 + o] = t[o][c];
    }
}

// ---------------------------------------------------------------------------
constexpr int kWgTP = 128;       // pixels per iteration
constexpr int kWgPitch = 160;    // 64 channels + 32 bytes (transposing reads)
constexpr int kWgDyBytes = kWgTP * kWgPitch;  // 20480
constexpr int kWgMaxPatch = 288;
constexpr int kWgLds = kWgDyBytes + kWgMaxPatch * kWgPitch;  // 66560

// NXP: 16-byte patch pieces per thread (ceil(patch pixels * 8 / 256) of the shapes the
// instantiation is launched for).
template <int NXP>
__global__ void __launch_bounds__(256)
conv3x3_wgrad_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                     bf16* __restrict__ part, ConvGeo g, int nit, long slab_stride) {
    static_assert(NXP >= 1 && NXP <= 9, "named prefetch registers");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_dy = smem;               // [128 px][64 oc], pitch 160
    char* s_x = smem + kWgDyBytes;   // [patch pixel][64 ic], pitch 160
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ob = wave & 1, ib = wave >> 1, h = lane >> 5;
    const int ict = g.Cin >> 6;
    const int oc0 = (blockIdx.y / ict) * 64, ic0 = (blockIdx.y % ict) * 64;
    const int PW = g.W + 2, PR = g.RI + 2;

    // Patch staging plan, the part that does not depend on the tile: piece tid + 256 i is
    // 8 channels of patch pixel (band, pr, pc).  A tile starts at an image row boundary
    // (row0 = tile * 128 / W counts rows over all images), and the patch pixel's source is
    // pixel ((row0 + band RI + pr - 1) W + pc - 1) of the tensor whenever it lies inside its
    // image - so its offset from the tile's first pixel is the same for every tile
    // (poff), and only whether the row is inside the image has to be asked per tile:
    //   plan = -1 past the patch, else (column inside the image) << 24 | band RI << 8 | pr
    int plan[NXP], poff[NXP];
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
        const int q = tid + i * 256, pp = q >> 3, p = q & 7;
        plan[i] = -1;
        poff[i] = 0;
        if (pp < g.np) {
            const int band = pp / (PR * PW);
            const int rem = pp - band * PR * PW;
            const int pr = rem / PW, pc = rem - pr * PW;
            const int xin = (pc >= 1 && pc <= g.W) ? 1 : 0;
            plan[i] = (xin << 24) | ((band * g.RI) << 8) | pr;
            poff[i] = ((band * g.RI + pr - 1) * g.W + pc - 1) * g.Cin + ic0 + p * 8;
        }
    }
    const int dyoff = (tid >> 3) * g.Cout + oc0 + (tid & 7) * 8;  // piece 0; piece i: + 32 i Cout
    // transposed-read lane geometry (see nfd::load_tr_frag)
    const int li = lane & 15, lq = li >> 2;
    const int lcol = (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;
    const char* xb = s_x + (ib * 32) * 2 + lcol;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x16{0};

    // tile t + 1 travels memory -> registers while tile t is being multiplied
    // (named registers: an indexed array across loop iterations lands in scratch).
    // The loads are unconditional - a piece outside its image reads the tile's first
    // pixel and is zeroed when it is stored (bit i of `inside`) - so that fetching a tile
    // is straight-line code the compiler can spread between the MFMAs.
#define NF_DY_LIST(X) X(0) X(1) X(2) X(3)
#define NF_X_LIST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#define NF_DECL_DY(i) uint4 pdy##i = make_uint4(0u, 0u, 0u, 0u);
#define NF_DECL_X(i) [[maybe_unused]] uint4 px##i = make_uint4(0u, 0u, 0u, 0u);
    NF_DY_LIST(NF_DECL_DY)
    NF_X_LIST(NF_DECL_X)
    const bf16* xt = x;    // first pixel of the tile being fetched
    const bf16* dyt = dy;
    int r0m = 0;           // its first row within the image
    unsigned inside = 0, inside_next = 0;
#define NF_SET_TILE(t)                                              \
    {                                                               \
        const long first = static_cast<long>(t) * kWgTP;           \
        xt = x + first * g.Cin;                                     \
        dyt = dy + first * g.Cout;                                  \
        r0m = static_cast<int>(first >> g.logW) & (g.H - 1);        \
        inside_next = 0;                                            \
    }
#define NF_LOAD_DY(i) \
    pdy##i = *reinterpret_cast<const uint4*>(dyt + dyoff + i * 32 * g.Cout);
#define NF_LOAD_X(i)                                                                      \
    if constexpr (i < NXP) {                                                              \
        const int yy = ((r0m + ((plan[i] >> 8) & 0xFFFF)) & (g.H - 1)) + (plan[i] & 255) - 1; \
        const bool ok = plan[i] >= (1 << 24) && static_cast<unsigned>(yy) < static_cast<unsigned>(g.H); \
        inside_next |= ok ? (1u << i) : 0u;                                               \
        px##i = *reinterpret_cast<const uint4*>(xt + (ok ? poff[i] : 0));                 \
    }
#define NF_STORE_DY(i)                                                                    \
    {                                                                                     \
        const int q = tid + i * 256, pix = q >> 3, p = q & 7;                             \
        *reinterpret_cast<uint4*>(s_dy + pix * kWgPitch + p * 16) = pdy##i;               \
    }
#define NF_STORE_X(i)                                                                     \
    if constexpr (i < NXP) {                                                              \
        if (plan[i] >= 0) {                                                               \
            const int q = tid + i * 256, pp = q >> 3, p = q & 7;                          \
            *reinterpret_cast<uint4*>(s_x + pp * kWgPitch + p * 16) =                     \
                (inside >> i) & 1u ? px##i : make_uint4(0u, 0u, 0u, 0u);                  \
        }                                                                                 \
    }
    NF_SET_TILE(blockIdx.x * nit)
    NF_DY_LIST(NF_LOAD_DY)
    NF_X_LIST(NF_LOAD_X)
    for (int it = 0; it < nit; ++it) {
        __syncthreads();
        inside = inside_next;
        NF_DY_LIST(NF_STORE_DY)
        NF_X_LIST(NF_STORE_X)
        __syncthreads();
        if (it + 1 < nit) {
            NF_SET_TILE(blockIdx.x * nit + it + 1)
            NF_DY_LIST(NF_LOAD_DY)
            NF_X_LIST(NF_LOAD_X)
        }
#undef NF_DY_LIST
#undef NF_X_LIST
#undef NF_DECL_DY
#undef NF_DECL_X
#undef NF_SET_TILE
#undef NF_LOAD_DY
#undef NF_LOAD_X
#undef NF_STORE_DY
#undef NF_STORE_X
        // Eight k-steps of 16 pixels, software-pipelined: the 20 transposing reads of step
        // k + 1 (2 for the dY fragment, 18 for the nine shifted X fragments) are issued
        // between the nine MFMAs of step k, two per MFMA (the scheduling hints below pin
        // that order; left alone, the compiler puts each step's reads directly before its
        // MFMAs and every step starts with a full LDS round trip).
        bf16x8 fa[2], fb[2][9];
        auto fetch = [&](int ks, bf16x8& A, bf16x8 (&B)[9]) {
            A = load_tr_frag(s_dy, kWgPitch, ks * 16, ob * 32, lane);
            // the 8 pixels of this lane half: two groups of 4 consecutive pixels
            const int m0 = ks * 16 + h * 8;
            const char* r0 = xb + (patch_index(g, m0) + lq) * kWgPitch;
            const char* r1 = xb + (patch_index(g, m0 + 4) + lq) * kWgPitch;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int toff = ((t / 3) * PW + (t % 3)) * kWgPitch;
                const bf16x4 lo = lds_tr4(r0 + toff);
                const bf16x4 hi = lds_tr4(r1 + toff);
                B[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        fetch(0, fa[0], fb[0]);
#pragma unroll
        for (int ks = 0; ks < kWgTP / 16; ++ks) {
            if (ks + 1 < kWgTP / 16) fetch(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = mfma32(fa[ks & 1], fb[ks & 1][t], acc[t]);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 20, 0);  // step 0's reads
#pragma unroll
        for (int ks = 0; ks + 1 < kWgTP / 16; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // two LDS reads
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
    }
    bf16* slab = part + static_cast<long>(blockIdx.x) * slab_stride;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oc = oc0 + ob * 32 + acc_row(r, lane);
            slab[(static_cast<long>(oc) * 9 + t) * g.Cin + ic0 + ib * 32 + (lane & 31)] =
                f2bf(acc[t][r]);
        }
}

// dw[i] = sum over slabs in a fixed order.  A workgroup owns 64 groups of four
// consecutive elements; its four waves each sum every fourth slab (so that few,
// long columns still spread over the chip) and combine through LDS.
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const bf16* __restrict__ part, bf16* __restrict__ dw, long n4, int slabs,
                    long dw_stride) {
    __shared__ float4 red[3][64];
    part += static_cast<long>(blockIdx.y) * slabs * n4 * 4;  // blockIdx.y: group
    dw += blockIdx.y * dw_stride;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long i = static_cast<long>(blockIdx.x) * 64 + lane;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        const bf16x4* p = reinterpret_cast<const bf16x4*>(part) + i;
#pragma unroll 4
        for (int s = grp; s < slabs; s += 4) {
            const bf16x4 v = p[static_cast<long>(s) * n4];
            a.x += bf2f(v[0]); a.y += bf2f(v[1]); a.z += bf2f(v[2]); a.w += bf2f(v[3]);
        }
    }
    if (grp > 0) red[grp - 1][lane] = a;
    __syncthreads();
    if (grp == 0 && i < n4) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float4 v = red[k][lane];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        bf16x4 o;
        o[0] = f2bf(a.x); o[1] = f2bf(a.y); o[2] = f2bf(a.z); o[3] = f2bf(a.w);
        *reinterpret_cast<bf16x4*>(dw + i * 4) = o;
    }
}

// ---------------------------------------------------------------------------
inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Geometry of tiles of tp pixels; false when the shape is not supported.
bool make_geo(int B, int H, int W, int Cin, int Cout, int tp, int max_patch, ConvGeo* g) {
    if (!pow2(H) || !pow2(W) || W < 4 || W > 64 || H > 64 || B <= 0) return false;
    if (Cin % 64 != 0 || Cout % 64 != 0 || Cin <= 0 || Cout <= 0) return false;
    if (tp % W != 0) return false;
    const long M = static_cast<long>(B) * H * W;
    if (M % tp != 0 || M * (Cin > Cout ? Cin : Cout) >= (1L << 31)) return false;
    const int rows = tp / W;
    g->H = H; g->W = W; g->logH = ilog2(H); g->logW = ilog2(W);
    g->RI = rows < H ? rows : H;
    g->logRI = ilog2(g->RI);
    g->NI = rows / g->RI;
    g->np = g->NI * (g->RI + 2) * (W + 2);
    g->Cin = Cin; g->Cout = Cout;
    g->tpg = static_cast<int>(M / tp);
    g->wgs = 0;
    g->xorder = 0;
    g->rp = g->bp = 0;  // set by fwd_geo for the forward kernel
    return g->np <= max_patch;
}

// forward tile choice: 256-pixel tiles (4 x 1 waves: every wave runs all k-steps of its 64
// pixels, a chunk's weights are staged once for twice the pixels and nothing is summed
// through LDS at the end) while they make >= 512 workgroups and the patch of a tile fits;
// 128-pixel tiles (2 x 2 waves, chunks of 32 channels) while they make >= 256; else
// 64-pixel tiles (1 x 4 waves, chunks of 64).  NANOFED_CONV_TILE256=0 leaves the first out.
int fwd_tile_pixels(int B, int H, int W, int Cout) {
    static const bool big = [] {
        const char* e = getenv("NANOFED_CONV_TILE256");
        return e == nullptr || e[0] != '0';
    }();
    const long M = static_cast<long>(B) * H * W;
    if (big && M % 256 == 0 && (M / 256) * (Cout / 64) >= 512 && 256 % W == 0) {
        const int rows = 256 / W, ri = rows < H ? rows : H, ni = rows / ri;
        if (ni * (ri + 2) * (W + 2) <= FwdStage<32>::kMaxPatch) return 256;
    }
    return (M / 128) * (Cout / 64) >= 256 && M % 128 == 0 ? 128 : 64;
}

bool fwd_geo(int B, int H, int W, int Cin, int Cout, ConvGeo* g) {
    const int tp = fwd_tile_pixels(B, H, W, Cout);
    const int max_patch = tp == 256 ? FwdStage<32>::kMaxPatch
                          : 5 * 256 / (tp == 128 ? FwdStage<32>::kPieces : FwdStage<64>::kPieces);
    if (!make_geo(B, H, W, Cin, Cout, tp, max_patch, g)) return false;
    const int pitch = tp >= 128 ? FwdStage<32>::kPitch : FwdStage<64>::kPitch;
    nfix::conv_patch_pitches(g->W, g->logW, g->RI, g->logRI, g->NI, tp, pitch, &g->rp, &g->bp);
    // the larger of the two weight regions (forward, data gradient) beside the patch
    const int wbytes = tp >= 128 ? (FwdStage<32>::kTBBytes > FwdStage<32>::kBBytes ? FwdStage<32>::kTBBytes : FwdStage<32>::kBBytes)
                                 : (FwdStage<64>::kTBBytes > FwdStage<64>::kBBytes ? FwdStage<64>::kTBBytes : FwdStage<64>::kBBytes);
    // NANOFED_CONV_PATCH_PAD=0 keeps the plain rows (for measurements)
    static const bool pad = [] {
        const char* e = getenv("NANOFED_CONV_PATCH_PAD");
        return e == nullptr || e[0] != '0';
    }();
    if (!pad || wbytes + g->NI * g->bp > kFwdLdsCap) {  // no (room for the) padding: plain rows
        g->rp = (W + 2) * pitch;
        g->bp = (g->RI + 2) * g->rp;
    }
    return wbytes + g->NI * g->bp <= kFwdLdsCap;
}

int g_wgrad_min_wgs = 192;

inline int set_lds(const void* fn, int bytes) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess
               ? 0
               : -20;
}

// pixel ranges of the weight gradient: as few slabs as keep g_wgrad_min_wgs
// workgroups in flight, at least one iteration of 128 pixels each
int wgrad_iterations(long M, int Cin, int Cout, int G = 1) {
    const long tiles = static_cast<long>(Cin / 64) * (Cout / 64) * G;
    const long iters = M / kWgTP;  // per group
    long nit = 1;
    while (nit * 2 <= iters && (iters / (nit * 2)) * tiles >= g_wgrad_min_wgs && iters % (nit * 2) == 0) nit *= 2;
    return static_cast<int>(nit);
}

// One instantiation: raises its LDS limit once, launches it.
template <int PXW, int KSW, int CH, bool kDgrad, int NAP, bool kBnBwd>
int launch_one(dim3 grid, int lds, hipStream_t s, const bf16* x, const bf16* w, bf16* y,
               float* colsum, const ConvGeo& g, const BnBwdIn& fz) {
    static const int lds_ok = set_lds(
        reinterpret_cast<const void*>(conv3x3_fwd_kernel<PXW, KSW, CH, kDgrad, NAP, kBnBwd>), kFwdLdsCap);
    if (lds_ok != 0) return -20;
    hipLaunchKernelGGL((conv3x3_fwd_kernel<PXW, KSW, CH, kDgrad, NAP, kBnBwd>), grid, dim3(256), lds, s,
                       x, w, y, colsum, g, fz);
    return static_cast<int>(hipGetLastError());
}

// ... with or without the BatchNorm backward epilogue (data gradient only)
template <int PXW, int KSW, int CH, bool kDgrad, int NAP>
int launch_pick(dim3 grid, int lds, hipStream_t s, const bf16* x, const bf16* w, bf16* y,
                float* colsum, const ConvGeo& g, const BnBwdIn* fz) {
    if constexpr (kDgrad) {
        if (fz != nullptr)
            return launch_one<PXW, KSW, CH, true, NAP, true>(grid, lds, s, x, w, y, colsum, g, *fz);
    }
    return launch_one<PXW, KSW, CH, kDgrad, NAP, false>(grid, lds, s, x, w, y, colsum, g, BnBwdIn{});
}

// B: images of the whole launch (all groups), G groups of B / G images
template <bool kDgrad>
int launch_fwd(const void* x, const void* w, void* y, float* colsum, int B, int H, int W,
               int Cin, int Cout, hipStream_t s, int G = 1, long w_stride = 0,
               const BnBwdIn* fz = nullptr) {
    if (G < 1 || B % G != 0) return -1;
    const int tp = fwd_tile_pixels(B, H, W, Cout);
    ConvGeo g;
    if (!fwd_geo(B, H, W, Cin, Cout, &g)) return -1;
    if (!al16(x) || !al16(w) || !al16(y)) return -2;
    if (fz != nullptr && (!kDgrad || fz->xpre == nullptr || colsum == nullptr)) return -4;
    if (fz != nullptr && fz->mask == nullptr &&
        (fz->stats == nullptr || fz->gamma == nullptr || fz->beta == nullptr))
        return -4;
    const long M = static_cast<long>(B) * H * W;
    if ((M / G) % tp != 0 || (w_stride & 7) != 0) return -3;
    g.tpg = static_cast<int>(M / G / tp);
    g.wgs = w_stride;
    // Workgroup order.  Ids are dealt round-robin to the 8 XCDs (one L2 each).  With pixel
    // tiles as the fast grid dimension the tiles that share a weight slice - the largest
    // thing a workgroup reads: 1152 Cin bytes against 2.6 TP Cin for its patch - land on 8
    // different L2s and the slice is fetched 8 times (the 512-channel stage fetched 227 MB
    // from HBM per launch for 38 MB of weights and 8 MB of activations).  The other order
    // keeps a slice on one L2 and fetches each patch once per output-channel tile instead;
    // taken when that is less traffic by this estimate.  NANOFED_CONV_XORDER=0|1 pins it.
    {
        const char* e = getenv("NANOFED_CONV_XORDER");  // (read per launch: tests switch it)
        const int pinned = e == nullptr ? -1 : (e[0] != '0' ? 1 : 0);
        const double wbytes = 2.0 * G * Cout * 9.0 * Cin, abytes = 2.0 * M * Cin;
        const int oct = Cout / 64, share = g.tpg < 8 ? g.tpg : 8;
        const bool better = wbytes + abytes * oct < wbytes * share + abytes;
        g.xorder = pinned >= 0 ? pinned : (better ? 1 : 0);
    }
    const dim3 grid = g.xorder ? dim3(static_cast<unsigned>(M / tp) * (Cout / 64))
                               : dim3(static_cast<unsigned>(M / tp), Cout / 64);
    const bf16* xp = static_cast<const bf16*>(x);
    const bf16* wp = static_cast<const bf16*>(w);
    bf16* yp = static_cast<bf16*>(y);
    constexpr int kLdsCap = kFwdLdsCap;
    // patch pieces per thread of this shape (the instantiations cover 5, 6 and 7)
    const int pieces = (g.np * (tp >= 128 ? 4 : 8) + 255) / 256;
    if (tp >= 128) {
        using St = FwdStage<32>;
        int lds = (kDgrad ? St::kTBBytes : St::kBBytes) + g.NI * g.bp;
        if (lds < kRedBytes + kColsumBytes) lds = kRedBytes + kColsumBytes;
        if (lds > kLdsCap) return -20;
        if (tp == 256 && pieces <= 6)
            return launch_pick<4, 1, 32, kDgrad, 6>(grid, lds, s, xp, wp, yp, colsum, g, fz);
        if (tp == 256 && pieces <= 7)
            return launch_pick<4, 1, 32, kDgrad, 7>(grid, lds, s, xp, wp, yp, colsum, g, fz);
        if (tp == 128 && pieces <= 5)
            return launch_pick<2, 2, 32, kDgrad, 5>(grid, lds, s, xp, wp, yp, colsum, g, fz);
        return -21;
    }
    using St = FwdStage<64>;
    int lds = (kDgrad ? St::kTBBytes : St::kBBytes) + g.NI * g.bp;
    if (lds < kRedBytes + kColsumBytes) lds = kRedBytes + kColsumBytes;
    if (lds > kLdsCap || pieces > 5) return pieces > 5 ? -21 : -20;
    return launch_pick<1, 4, 64, kDgrad, 5>(grid, lds, s, xp, wp, yp, colsum, g, fz);
}

}  // namespace

extern "C" {

// 1 when nf_conv3x3_fwd / nf_conv3x3_wgrad handle this shape.
int nf_conv3x3_supported(int B, int H, int W, int Cin, int Cout) {
    ConvGeo g;
    if (!fwd_geo(B, H, W, Cin, Cout, &g)) return 0;
    if (!fwd_geo(B, H, W, Cout, Cin, &g)) return 0;
    return make_geo(B, H, W, Cin, Cout, kWgTP, kWgMaxPatch, &g) ? 1 : 0;
}

// Pixel tiles of the forward kernel for this shape: the number of partial rows
// nf_conv3x3_fwd writes into colsum.
int nf_conv3x3_fwd_tiles(int B, int H, int W, int Cout) {
    return static_cast<int>(static_cast<long>(B) * H * W / fwd_tile_pixels(B, H, W, Cout));
}

// y[B][H][W][Cout] = conv3x3(x[B][H][W][Cin], w[Cout][3][3][Cin]), bf16.
// colsum (may be null): [tiles][2][Cout] floats, per-tile sums of y and y^2 over the
// tile's pixels for every channel (what a following BatchNorm needs).
int nf_conv3x3_fwd(const void* x, const void* w, void* y, float* colsum, int B, int H, int W,
                   int Cin, int Cout, hipStream_t s) {
    return launch_fwd<false>(x, w, y, colsum, B, H, W, Cin, Cout, s);
}

// dx[B][H][W][Cin] = data gradient of that convolution: dy[B][H][W][Cout] against the
// same, unmodified w[Cout][3][3][Cin].
int nf_conv3x3_dgrad(const void* dy, const void* w, void* dx, int B, int H, int W, int Cin,
                     int Cout, hipStream_t s) {
    // as a forward launch: reduction over the layer's Cout, output channels = its Cin
    return launch_fwd<true>(dy, w, dx, nullptr, B, H, W, Cout, Cin, s);
}

// wt[Cin][3][3][Cout] = w[Cout][3][3][Cin] with reversed taps.
int nf_conv3x3_flip_weights(const void* w, void* wt, int Cin, int Cout, hipStream_t s) {
    if (Cin % 64 != 0 || Cout % 64 != 0 || Cin <= 0 || Cout <= 0) return -1;
    hipLaunchKernelGGL(flip_weights_kernel, dim3(9, Cout / 64, Cin / 64), dim3(256), 0, s,
                       static_cast<const bf16*>(w), static_cast<bf16*>(wt), Cin, Cout);
    return static_cast<int>(hipGetLastError());
}

// Tuning knob: the weight gradient splits the pixels into as few ranges as still
// make this many workgroups (default 192).
void nf_conv3x3_set_wgrad_min_workgroups(int n) { g_wgrad_min_wgs = n < 1 ? 1 : n; }

}  // extern "C"

namespace {

// weight gradient of G groups of B / G images each; dw of group g starts dw_stride
// elements after that of group g - 1
int wgrad_launch(const void* x, const void* dy, void* dw, void* workspace, int B, int H, int W,
                 int Cin, int Cout, int G, long dw_stride, hipStream_t s) {
    ConvGeo g;
    if (G < 1 || B % G != 0) return -1;
    if (!make_geo(B, H, W, Cin, Cout, kWgTP, kWgMaxPatch, &g)) return -1;
    if (!al16(x) || !al16(dy) || !al16(dw) || !al16(workspace) || (dw_stride & 7) != 0) return -2;
    const long Mg = static_cast<long>(B / G) * H * W;
    if (Mg % kWgTP != 0) return -3;
    const int nit = wgrad_iterations(Mg, Cin, Cout, G);
    const int slabs = static_cast<int>(Mg / kWgTP / nit);  // per group
    const dim3 grid(slabs * G, (Cout / 64) * (Cin / 64));
    static const int lds_ok =
        set_lds(reinterpret_cast<const void*>(conv3x3_wgrad_kernel<6>), kWgLds) |
        set_lds(reinterpret_cast<const void*>(conv3x3_wgrad_kernel<7>), kWgLds) |
        set_lds(reinterpret_cast<const void*>(conv3x3_wgrad_kernel<9>), kWgLds);
    if (lds_ok != 0) return -20;
    const bf16* xp = static_cast<const bf16*>(x);
    const bf16* dyp = static_cast<const bf16*>(dy);
    // one pixel range per group: its tile is the gradient itself, written in place
    const bool direct = slabs == 1;
    bf16* wp = static_cast<bf16*>(direct ? dw : workspace);
    const long sstride = direct ? dw_stride : static_cast<long>(Cout) * 9 * Cin;
    const int pieces = (g.np * 8 + 255) / 256;  // patch pieces per thread (at most 9)
    if (pieces <= 6)
        hipLaunchKernelGGL(conv3x3_wgrad_kernel<6>, grid, dim3(256), kWgLds, s, xp, dyp, wp, g, nit, sstride);
    else if (pieces <= 7)
        hipLaunchKernelGGL(conv3x3_wgrad_kernel<7>, grid, dim3(256), kWgLds, s, xp, dyp, wp, g, nit, sstride);
    else
        hipLaunchKernelGGL(conv3x3_wgrad_kernel<9>, grid, dim3(256), kWgLds, s, xp, dyp, wp, g, nit, sstride);
    if (direct) return static_cast<int>(hipGetLastError());
    const long n4 = static_cast<long>(Cout) * 9 * Cin / 4;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(static_cast<unsigned>((n4 + 63) / 64), G),
                       dim3(256), 0, s, static_cast<const bf16*>(workspace), static_cast<bf16*>(dw),
                       n4, slabs, dw_stride);
    return static_cast<int>(hipGetLastError());
}

}  // namespace

extern "C" {

// bf16 elements of workspace nf_conv3x3_wgrad needs
long nf_conv3x3_wgrad_workspace(int B, int H, int W, int Cin, int Cout) {
    const long M = static_cast<long>(B) * H * W;
    if (M % kWgTP != 0 || Cin % 64 != 0 || Cout % 64 != 0) return -1;
    const int nit = wgrad_iterations(M, Cin, Cout);
    return (M / kWgTP / nit) * static_cast<long>(Cout) * 9 * Cin;
}

// dw[Cout][3][3][Cin] (bf16) = weight gradient of the convolution above.
int nf_conv3x3_wgrad(const void* x, const void* dy, void* dw, void* workspace, int B, int H,
                     int W, int Cin, int Cout, hipStream_t s) {
    return wgrad_launch(x, dy, dw, workspace, B, H, W, Cin, Cout, 1, 0, s);
}

// -- grouped launches: G groups of B images each (G * B images in x / y / dy) ----------

int nf_conv3x3_supported_g(int G, int B, int H, int W, int Cin, int Cout) {
    if (G < 1 || B < 1) return 0;
    ConvGeo g;
    const long Mg = static_cast<long>(B) * H * W;
    if (!fwd_geo(G * B, H, W, Cin, Cout, &g) || Mg % fwd_tile_pixels(G * B, H, W, Cout) != 0) return 0;
    if (!fwd_geo(G * B, H, W, Cout, Cin, &g) || Mg % fwd_tile_pixels(G * B, H, W, Cin) != 0) return 0;
    return make_geo(G * B, H, W, Cin, Cout, kWgTP, kWgMaxPatch, &g) && Mg % kWgTP == 0 ? 1 : 0;
}

// pixel tiles per group of the grouped forward launch (rows of colsum per group)
int nf_conv3x3_fwd_tiles_g(int G, int B, int H, int W, int Cout) {
    return static_cast<int>(static_cast<long>(B) * H * W / fwd_tile_pixels(G * B, H, W, Cout));
}

int nf_conv3x3_fwd_g(const void* x, const void* w, void* y, float* colsum, int G, int B, int H,
                     int W, int Cin, int Cout, long w_stride, hipStream_t s) {
    return launch_fwd<false>(x, w, y, colsum, G * B, H, W, Cin, Cout, s, G, w_stride);
}

int nf_conv3x3_dgrad_g(const void* dy, const void* w, void* dx, int G, int B, int H, int W,
                       int Cin, int Cout, long w_stride, hipStream_t s) {
    return launch_fwd<true>(dy, w, dx, nullptr, G * B, H, W, Cout, Cin, s, G, w_stride);
}

// The grouped data gradient with the elementwise part of the consuming BatchNorm's backward
// pass in its epilogue: dx = (data gradient + add) where mask > 0, else 0; colsum
// [G * tiles][2][Cin] (tiles = nf_conv3x3_fwd_tiles_g(G, B, H, W, Cin)) receives per-tile
// sums of dx and dx * xpre per channel.  mask, xpre, add (may be null): [G * B][H][W][Cin].
// mask == nullptr: the BatchNorm had no residual input and the mask is recomputed from xpre
// (its input) with stats [G][2][Cin] (mean, 1/std) and gamma / beta (group 0's vectors,
// pstride floats between groups).
int nf_conv3x3_dgrad_bn_g(const void* dy, const void* w, void* dx, const void* mask,
                          const void* xpre, const void* add, float* colsum, const float* stats,
                          const float* gamma, const float* beta, long pstride, int G, int B,
                          int H, int W, int Cin, int Cout, long w_stride, hipStream_t s) {
    const BnBwdIn fz{static_cast<const bf16*>(mask), static_cast<const bf16*>(xpre),
                     static_cast<const bf16*>(add), stats, gamma, beta, pstride};
    return launch_fwd<true>(dy, w, dx, colsum, G * B, H, W, Cout, Cin, s, G, w_stride, &fz);
}

long nf_conv3x3_wgrad_workspace_g(int G, int B, int H, int W, int Cin, int Cout) {
    const long Mg = static_cast<long>(B) * H * W;
    if (G < 1 || Mg % kWgTP != 0 || Cin % 64 != 0 || Cout % 64 != 0) return -1;
    const int nit = wgrad_iterations(Mg, Cin, Cout, G);
    return G * (Mg / kWgTP / nit) * static_cast<long>(Cout) * 9 * Cin;
}

int nf_conv3x3_wgrad_g(const void* x, const void* dy, void* dw, void* workspace, int G, int B,
                       int H, int W, int Cin, int Cout, long dw_stride, hipStream_t s) {
    return wgrad_launch(x, dy, dw, workspace, G * B, H, W, Cin, Cout, G, dw_stride, s);
}

}  // extern "C"
