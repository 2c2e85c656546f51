// Patch gather (im2col) and its adjoint (col2im) for channels-last bf16 activations:
// the memory-side half of convolutions that run as plain GEMMs on gemm_ops.hip -
// strided 3x3, 1x1 shortcut and the 3-channel stem convolution of ResNet-18, and the
// weight gradient of stages whose images are too small for the implicit-GEMM kernel.
//
//   col[m][(ky * KW + kx) * C + c] = x[b][oy * S - P + ky][ox * S - P + kx][c]   (0 outside)
//   m = (b * Ho + oy) * Wo + ox,   rows of Kp >= KH * KW * C elements (Kp a multiple of 8,
//   the tail of a row is zero)
//
// With the weight stored [Cout][KH][KW][Cin] (channels-last storage of a
// [Cout, Cin, KH, KW] tensor) the forward pass is y[M][Cout] = col · Wᵀ, the weight
// gradient dW[Cout][Kp] = dyᵀ · col and the data gradient col2im(dy · W).
//
// col2im is written as a gather: one thread owns 8 channels of one input pixel and adds
// the (at most ceil(KH / S) * ceil(KW / S)) patch entries that were copies of it, in a
// fixed order, in fp32 - no atomics, results do not depend on scheduling.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nf_device.h"

using namespace nfd;

namespace {

struct PatchGeo {
    int B, H, W, C;
    int Ho, Wo;
    int KH, KW, S, P;
    int Kp;  // row length of col in elements
};

// one thread per 16-byte piece of col (C a multiple of 8)
__global__ void __launch_bounds__(256) im2col8_kernel(const bf16* __restrict__ x,
                                                      bf16* __restrict__ col, PatchGeo g) {
    const int ppr = g.Kp >> 3;                       // pieces per row
    const int cp = g.C >> 3;                         // pieces per pixel
    const long total = static_cast<long>(g.B) * g.Ho * g.Wo * ppr;
    for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total;
         i += static_cast<long>(gridDim.x) * 256) {
        const long m = i / ppr;
        const int piece = static_cast<int>(i - m * ppr);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        const int tap = piece / cp;
        if (tap < g.KH * g.KW) {
            const int c8 = piece - tap * cp;
            const int ky = tap / g.KW, kx = tap - ky * g.KW;
            const int ox = static_cast<int>(m % g.Wo);
            const long t = m / g.Wo;
            const int oy = static_cast<int>(t % g.Ho);
            const int b = static_cast<int>(t / g.Ho);
            const int iy = oy * g.S - g.P + ky, ix = ox * g.S - g.P + kx;
            if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                v = *reinterpret_cast<const uint4*>(
                    x + ((static_cast<long>(b) * g.H + iy) * g.W + ix) * g.C + c8 * 8);
        }
        *reinterpret_cast<uint4*>(col + m * g.Kp + piece * 8) = v;
    }
}

// any channel count (the 3-channel stem): one thread per element of col
__global__ void __launch_bounds__(256) im2col1_kernel(const bf16* __restrict__ x,
                                                      bf16* __restrict__ col, PatchGeo g) {
    const long total = static_cast<long>(g.B) * g.Ho * g.Wo * g.Kp;
    const int K = g.KH * g.KW * g.C;
    for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total;
         i += static_cast<long>(gridDim.x) * 256) {
        const long m = i / g.Kp;
        const int k = static_cast<int>(i - m * g.Kp);
        bf16 v = f2bf(0.f);
        if (k < K) {
            const int tap = k / g.C, c = k - tap * g.C;
            const int ky = tap / g.KW, kx = tap - ky * g.KW;
            const int ox = static_cast<int>(m % g.Wo);
            const long t = m / g.Wo;
            const int oy = static_cast<int>(t % g.Ho);
            const int b = static_cast<int>(t / g.Ho);
            const int iy = oy * g.S - g.P + ky, ix = ox * g.S - g.P + kx;
            if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                v = x[((static_cast<long>(b) * g.H + iy) * g.W + ix) * g.C + c];
        }
        col[i] = v;
    }
}

// dx[b][iy][ix][c8 .. c8 + 7] = sum of the col entries that copied it (C a multiple of 8)
__global__ void __launch_bounds__(256) col2im8_kernel(const bf16* __restrict__ col,
                                                      bf16* __restrict__ dx, PatchGeo g) {
    const int cp = g.C >> 3;
    const long total = static_cast<long>(g.B) * g.H * g.W * cp;
    for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total;
         i += static_cast<long>(gridDim.x) * 256) {
        const long pix = i / cp;
        const int c8 = static_cast<int>(i - pix * cp);
        const int ix = static_cast<int>(pix % g.W);
        const long t = pix / g.W;
        const int iy = static_cast<int>(t % g.H);
        const int b = static_cast<int>(t / g.H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < g.KH; ++ky) {
            const int ty = iy + g.P - ky;
            if (ty < 0 || ty % g.S != 0) continue;
            const int oy = ty / g.S;
            if (oy >= g.Ho) continue;
            for (int kx = 0; kx < g.KW; ++kx) {
                const int tx = ix + g.P - kx;
                if (tx < 0 || tx % g.S != 0) continue;
                const int ox = tx / g.S;
                if (ox >= g.Wo) continue;
                const long m = (static_cast<long>(b) * g.Ho + oy) * g.Wo + ox;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(
                    col + m * g.Kp + (ky * g.KW + kx) * g.C + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
            }
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e]);
        *reinterpret_cast<bf16x8*>(dx + pix * g.C + c8 * 8) = o;
    }
}

// col2im of G groups of g.B images with the elementwise part of the backward pass of the
// BatchNorm whose (ReLU) output the convolution read:
//   gz = bf16(bf16(gathered sum) + add) where mask > 0, else 0       -> dx
// (add: the gradient from the other branch that read that output, may be null; mask: that
// output) and per workgroup the sums over its pixels of gz and gz * xpre per channel (xpre: the
// BatchNorm's input) -> partial[group][block][2][C], which is all the BatchNorm backward needs
// from a pass over the pixels.  blockIdx.y = group; a thread keeps its 8 channels over its loop
// (the stride, 256 * gridDim.x pieces, is a multiple of the pieces per pixel), so its sums
// stay in registers and meet those of the other threads in LDS in a fixed order.
constexpr int kJoinPad = 17;

__global__ void __launch_bounds__(256)
col2im8_join_kernel(const bf16* __restrict__ col, bf16* __restrict__ dx,
                    const bf16* __restrict__ add, const bf16* __restrict__ mask,
                    const bf16* __restrict__ xpre, float* __restrict__ partial, PatchGeo g) {
    __shared__ float red[256 * kJoinPad];
    const int cp = g.C >> 3, tid = threadIdx.x;
    const long per_group = static_cast<long>(g.B) * g.H * g.W;
    const long pix0 = blockIdx.y * per_group;  // first pixel / image of the group
    const long img0 = static_cast<long>(blockIdx.y) * g.B;
    const long total = per_group * cp;
    float sb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float sg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long i = static_cast<long>(blockIdx.x) * 256 + tid; i < total;
         i += static_cast<long>(gridDim.x) * 256) {
        const long pix = i / cp;
        const int c8 = static_cast<int>(i - pix * cp);
        const int ix = static_cast<int>(pix % g.W);
        const long t = pix / g.W;
        const int iy = static_cast<int>(t % g.H);
        const long b = img0 + t / g.H;
        const long at = (pix0 + pix) * g.C + c8 * 8;
        // the three other operands travel while the patch entries are gathered
        const bf16x8 mk = *reinterpret_cast<const bf16x8*>(mask + at);
        const bf16x8 xp = *reinterpret_cast<const bf16x8*>(xpre + at);
        bf16x8 ad;
        if (add != nullptr) ad = *reinterpret_cast<const bf16x8*>(add + at);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < g.KH; ++ky) {
            const int ty = iy + g.P - ky;
            if (ty < 0 || ty % g.S != 0) continue;
            const int oy = ty / g.S;
            if (oy >= g.Ho) continue;
            for (int kx = 0; kx < g.KW; ++kx) {
                const int tx = ix + g.P - kx;
                if (tx < 0 || tx % g.S != 0) continue;
                const int ox = tx / g.S;
                if (ox >= g.Wo) continue;
                const long m = (b * g.Ho + oy) * g.Wo + ox;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(
                    col + m * g.Kp + (ky * g.KW + kx) * g.C + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
            }
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // (the roundings of the separate kernels: the gathered sum, then the joined sum)
            float f = bf2f(f2bf(acc[e]));
            if (add != nullptr) f += bf2f(ad[e]);
            o[e] = f2bf(bf2f(mk[e]) > 0.f ? f : 0.f);
            f = bf2f(o[e]);
            sb[e] += f;
            sg[e] = fmaf(f, bf2f(xp[e]), sg[e]);
        }
        *reinterpret_cast<bf16x8*>(dx + at) = o;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[tid * kJoinPad + e] = sb[e];
        red[tid * kJoinPad + 8 + e] = sg[e];
    }
    __syncthreads();
    // thread tid owned channel piece tid % cp: channel c was summed by threads c / 8 + j cp
    float* out = partial + (static_cast<long>(blockIdx.y) * gridDim.x + blockIdx.x) * 2 * g.C;
    for (int o = tid; o < 2 * g.C; o += 256) {
        const int stat = o / g.C, c = o - stat * g.C;
        const int c8 = c >> 3, e = c & 7;
        float total_c = 0.f;
        for (int j = c8; j < 256; j += cp) total_c += red[j * kJoinPad + stat * 8 + e];
        out[o] = total_c;
    }
}

inline int check_geo(const PatchGeo& g) {
    if (g.B < 1 || g.H < 1 || g.W < 1 || g.C < 1 || g.KH < 1 || g.KW < 1 || g.S < 1 || g.P < 0)
        return -1;
    if (g.Ho != (g.H + 2 * g.P - g.KH) / g.S + 1 || g.Wo != (g.W + 2 * g.P - g.KW) / g.S + 1)
        return -2;
    if (g.Ho < 1 || g.Wo < 1) return -2;
    if (g.Kp % 8 != 0 || g.Kp < g.KH * g.KW * g.C) return -3;
    return 0;
}

inline int grid_for(long items) {
    const long blocks = (items + 255) / 256;
    return static_cast<int>(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
}

}  // namespace

extern "C" {

// col must hold B * Ho * Wo rows of Kp elements; x and col 16-byte aligned when C % 8 == 0.
int nf_im2col(const void* x, void* col, int B, int H, int W, int C, int Ho, int Wo, int KH,
              int KW, int S, int P, int Kp, hipStream_t stream) {
    const PatchGeo g{B, H, W, C, Ho, Wo, KH, KW, S, P, Kp};
    const int rc = check_geo(g);
    if (rc) return rc;
    const long rows = static_cast<long>(B) * Ho * Wo;
    if (C % 8 == 0) {
        hipLaunchKernelGGL(im2col8_kernel, dim3(grid_for(rows * (Kp / 8))), dim3(256), 0, stream,
                           static_cast<const bf16*>(x), static_cast<bf16*>(col), g);
    } else {
        hipLaunchKernelGGL(im2col1_kernel, dim3(grid_for(rows * Kp)), dim3(256), 0, stream,
                           static_cast<const bf16*>(x), static_cast<bf16*>(col), g);
    }
    return static_cast<int>(hipGetLastError());
}

// dx [B][H][W][C] from col [B * Ho * Wo][Kp]; C must be a multiple of 8.
int nf_col2im(const void* col, void* dx, int B, int H, int W, int C, int Ho, int Wo, int KH,
              int KW, int S, int P, int Kp, hipStream_t stream) {
    const PatchGeo g{B, H, W, C, Ho, Wo, KH, KW, S, P, Kp};
    const int rc = check_geo(g);
    if (rc) return rc;
    if (C % 8 != 0) return -4;
    hipLaunchKernelGGL(col2im8_kernel, dim3(grid_for(static_cast<long>(B) * H * W * (C / 8))),
                       dim3(256), 0, stream, static_cast<const bf16*>(col),
                       static_cast<bf16*>(dx), g);
    return static_cast<int>(hipGetLastError());
}

// Workgroups per group of nf_col2im_join_g = rows of partial sums it writes per group.
int nf_col2im_join_blocks(int B, int H, int W, int C) {
    const long blocks = (static_cast<long>(B) * H * W * (C / 8) + 255) / 256;
    return static_cast<int>(blocks < 1 ? 1 : (blocks > 128 ? 128 : blocks));
}

// col2im for G groups of B images (col: [G * B * Ho * Wo][Kp], dx: [G * B][H][W][C]) joined
// with the second gradient branch, the ReLU mask and the per-channel partial sums of the
// BatchNorm backward that consumes dx (see col2im8_join_kernel).  C: 64, 128, 256 or 512
// (the pieces per pixel must divide 256).  partial: [G][nf_col2im_join_blocks][2][C] floats.
int nf_col2im_join_g(const void* col, void* dx, const void* add, const void* mask,
                     const void* xpre, float* partial, int G, int B, int H, int W, int C, int Ho,
                     int Wo, int KH, int KW, int S, int P, int Kp, hipStream_t stream) {
    const PatchGeo g{B, H, W, C, Ho, Wo, KH, KW, S, P, Kp};
    const int rc = check_geo(g);
    if (rc) return rc;
    if (G < 1 || G > 65535 || C % 8 != 0 || 256 % (C / 8) != 0) return -4;
    if (mask == nullptr || xpre == nullptr || partial == nullptr) return -5;
    hipLaunchKernelGGL(col2im8_join_kernel, dim3(nf_col2im_join_blocks(B, H, W, C), G), dim3(256),
                       0, stream, static_cast<const bf16*>(col), static_cast<bf16*>(dx),
                       static_cast<const bf16*>(add), static_cast<const bf16*>(mask),
                       static_cast<const bf16*>(xpre), partial, g);
    return static_cast<int>(hipGetLastError());
}

}  // extern "C"
