// Training-mode BatchNorm for channels-last bf16 activations, fused with the
// residual add and the ReLU that follow it in a ResNet block.
//
// The activation is a row-major matrix x[R][C] (R = N*H*W pixels, C channels,
// C/8 a power of two between 8 and 64 lanes wide: C = 64, 128, 256, 512), so
// the per-channel statistics are column sums:
//
//   nf_bn_fwd   1. partial column sums of x and x*x per workgroup (fp32)
//               2. fixed-order final sum -> mean, 1/std; running statistics
//               3. y = relu(x * scale + shift (+ residual)), 16-byte vectors
//   nf_bn_bwd   1. partial sums of dz and dz * xhat, where dz = dy masked by y > 0
//                  (nf_bn_bwd_pre_g: taken from the kernel that produced dz instead)
//               2. final sums -> d(gamma), d(beta) (fp32)
//               3. dx = (dz - mean(dz) - xhat * mean(dz * xhat)) * gamma / std,
//                  and d(residual) = dz
//
// A lane owns 8 adjacent channels (one 16-byte load); a wave covers
// 64 / (C/8) rows per iteration.  No atomics: results are reproducible.
//
// Grouped launches (nf_bn_fwd_g / nf_bn_bwd_g): x holds G independent batches of R
// rows each, one after the other (the clients of a GPU, every one with its own
// parameters and statistics); blockIdx.y is the group, parameters of group g live
// `pstride` floats after those of group g - 1 (rows of a parameter arena), and the
// parameter gradients can be written as bf16 into rows of a flat gradient buffer.
// The backward pass optionally takes its incoming gradient as the sum of two tensors
// (the two branches that consumed a residual block's input).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nf_device.h"

using namespace nfd;

namespace {

constexpr int kMaxBlocks = 128;

__device__ __forceinline__ void load8(const bf16* p, float (&f)[8]) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = bf2f(v[k]);
}

__device__ __forceinline__ void store8(bf16* p, const float (&f)[8]) {
    bf16x8 v;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = f2bf(f[k]);
    *reinterpret_cast<bf16x8*>(p) = v;
}

// Sums a[k], b[k] over the lanes of a wave that own the same channel chunk
// (lanes whose index differs only in the bits above log2(lpr)), then over the
// four waves through LDS; wave 0 writes partial[block][0][c], partial[block][1][c].
__device__ __forceinline__ void reduce_and_store(float (&a)[8], float (&b)[8], int lpr, int C,
                                                 float* partial, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = lpr; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a[k] += __shfl_xor(a[k], off, 64);
            b[k] += __shfl_xor(b[k], off, 64);
        }
    }
    // red: [4 waves][2][C]
    if (lane < lpr) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[(wave * 2 + 0) * C + lane * 8 + k] = a[k];
            red[(wave * 2 + 1) * C + lane * 8 + k] = b[k];
        }
    }
    __syncthreads();
    float* out = partial + static_cast<long>(blockIdx.x) * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256)
        out[i] = (red[i] + red[2 * C + i]) + (red[4 * C + i] + red[6 * C + i]);
}

__global__ void __launch_bounds__(256) bn_stats_partial_kernel(const bf16* x, float* partial,
                                                               long R, int C, int lpr,
                                                               long rows_per_block) {
    extern __shared__ float red[];
    x += static_cast<long>(blockIdx.y) * R * C;
    partial += static_cast<long>(blockIdx.y) * gridDim.x * 2 * C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = lane & (lpr - 1), sub = lane / lpr, rpw = 64 / lpr;
    const long r0 = blockIdx.x * rows_per_block;
    const long r1 = min(R, r0 + rows_per_block);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long r = r0 + wave * rpw + sub; r < r1; r += 4 * rpw) {
        float v[8];
        load8(x + r * C + chunk * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s[k] += v[k];
            q[k] = fmaf(v[k], v[k], q[k]);
        }
    }
    reduce_and_store(s, q, lpr, C, partial, red);
}

// Fixed-order sum of the per-workgroup partials of kFinalCh channels: thread (rl, cl) adds
// rows rl, rl + 16, ... of channel cl (eight loads of each statistic in flight: the kernel is
// a few round trips to memory and nothing else - a convolution leaves up to 256 partial rows
// per group), the 16 row lanes meet in LDS and are added pairwise in a fixed order.  Returns
// the channel of this thread for rl == 0 (with the two sums), -1 for every other thread.
constexpr int kFinalCh = 16;

__device__ __forceinline__ int final_sums(const float* partial, int nblocks, int C, float& s0,
                                          float& s1) {
    __shared__ float red[16][2][kFinalCh];
    const int cl = threadIdx.x & (kFinalCh - 1), rl = threadIdx.x / kFinalCh;
    const int c = blockIdx.x * kFinalCh + cl;
    float a = 0.f, b = 0.f;
    if (c < C) {
        const float* p = partial + c;
        const long row = 2L * C;
        int k = rl;
        for (; k + 7 * 16 < nblocks; k += 8 * 16) {
            float va[8], vb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                va[i] = p[(k + 16 * i) * row];
                vb[i] = p[(k + 16 * i) * row + C];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a += va[i];
                b += vb[i];
            }
        }
        for (; k < nblocks; k += 16) {
            a += p[k * row];
            b += p[k * row + C];
        }
    }
    red[rl][0][cl] = a;
    red[rl][1][cl] = b;
    __syncthreads();
    if (rl != 0 || c >= C) return -1;
    float t0[8], t1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        t0[i] = red[2 * i][0][cl] + red[2 * i + 1][0][cl];
        t1[i] = red[2 * i][1][cl] + red[2 * i + 1][1][cl];
    }
    s0 = ((t0[0] + t0[1]) + (t0[2] + t0[3])) + ((t0[4] + t0[5]) + (t0[6] + t0[7]));
    s1 = ((t1[0] + t1[1]) + (t1[2] + t1[3])) + ((t1[4] + t1[5]) + (t1[6] + t1[7]));
    return c;
}

// stats[0][c] = mean, stats[1][c] = 1/std (biased variance), running statistics
// updated with the unbiased variance, as torch.nn.BatchNorm2d does.
__global__ void __launch_bounds__(256) bn_stats_final_kernel(const float* partial, float* stats,
                                                             float* running_mean,
                                                             float* running_var, int nblocks,
                                                             int C, float inv_count,
                                                             float unbias, float momentum,
                                                             float eps, long pstride) {
    float s, q;
    partial += static_cast<long>(blockIdx.y) * nblocks * 2 * C;
    stats += static_cast<long>(blockIdx.y) * 2 * C;
    if (running_mean != nullptr) {
        running_mean += blockIdx.y * pstride;
        running_var += blockIdx.y * pstride;
    }
    const int c = final_sums(partial, nblocks, C, s, q);
    if (c < 0) return;
    const float mean = s * inv_count;
    const float var = fmaxf(q * inv_count - mean * mean, 0.f);
    stats[c] = mean;
    stats[C + c] = rsqrtf(var + eps);
    if (running_mean != nullptr) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * unbias;
    }
}

// A thread keeps the 8 channels of its first piece for the whole loop (the stride of the
// loop, 8 * 256 * gridDim.x elements, is a multiple of C), so the per-channel factors
// are loaded once and every piece costs one 16-byte load per operand, 8 FMAs and one store.
template <bool kRelu, bool kRes>
__global__ void __launch_bounds__(256) bn_apply_kernel(const bf16* x, const bf16* res,
                                                       const float* stats, const float* gamma,
                                                       const float* beta, bf16* y, long n8,
                                                       int C, long pstride) {
    {
        const long gofs = static_cast<long>(blockIdx.y) * n8 * 8;
        x += gofs;
        y += gofs;
        if (kRes) res += gofs;
        stats += static_cast<long>(blockIdx.y) * 2 * C;
        gamma += blockIdx.y * pstride;
        beta += blockIdx.y * pstride;
    }
    const long first = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
    const int c0 = static_cast<int>((first * 8) % C);
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = stats[C + c0 + k] * gamma[c0 + k];
        sh[k] = fmaf(-stats[c0 + k], sc[k], beta[c0 + k]);
    }
    for (long i = first; i < n8; i += static_cast<long>(gridDim.x) * 256) {
        float v[8], o[8];
        load8(x + i * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fmaf(v[k], sc[k], sh[k]);
        if (kRes) {
            float r[8];
            load8(res + i * 8, r);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] += r[k];
        }
        if (kRelu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = fmaxf(o[k], 0.f);
        }
        store8(y + i * 8, o);
    }
}

// gradient arriving at the layer: dy (+ dy2 when the layer's output fed two branches)
__device__ __forceinline__ void load_dy(const bf16* dy, const bf16* dy2, long at, float (&d)[8]) {
    load8(dy + at, d);
    if (dy2 != nullptr) {
        float e[8];
        load8(dy2 + at, e);
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] += e[k];
    }
}

template <bool kRelu>
__global__ void __launch_bounds__(256) bn_bwd_partial_kernel(const bf16* dy, const bf16* dy2,
                                                             const bf16* y, const bf16* x,
                                                             const float* stats, float* partial,
                                                             long R, int C, int lpr,
                                                             long rows_per_block) {
    extern __shared__ float red[];
    {
        const long gofs = static_cast<long>(blockIdx.y) * R * C;
        dy += gofs;
        if (dy2 != nullptr) dy2 += gofs;
        x += gofs;
        if (kRelu) y += gofs;
        stats += static_cast<long>(blockIdx.y) * 2 * C;
        partial += static_cast<long>(blockIdx.y) * gridDim.x * 2 * C;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = lane & (lpr - 1), sub = lane / lpr, rpw = 64 / lpr;
    const long r0 = blockIdx.x * rows_per_block;
    const long r1 = min(R, r0 + rows_per_block);
    float mu[8], rs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = stats[chunk * 8 + k];
        rs[k] = stats[C + chunk * 8 + k];
    }
    float sb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float sg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long r = r0 + wave * rpw + sub; r < r1; r += 4 * rpw) {
        const long at = r * C + chunk * 8;
        float d[8], xv[8];
        load_dy(dy, dy2, at, d);
        load8(x + at, xv);
        if (kRelu) {
            float yv[8];
            load8(y + at, yv);
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = yv[k] > 0.f ? d[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            sb[k] += d[k];
            sg[k] = fmaf(d[k], (xv[k] - mu[k]) * rs[k], sg[k]);
        }
    }
    reduce_and_store(sb, sg, lpr, C, partial, red);
}

// coef[0][c] = mean(dz), coef[1][c] = mean(dz * xhat); d(gamma), d(beta) are overwritten
// (fp32, or bf16 when out_bf16 is set; group g writes ostride elements after group g - 1).
// raw_stats (may be null): the second partial sum is that of dz * x, not of dz * xhat (the
// partials come from the kernel that produced dz, which does not know the statistics):
// sum dz xhat = (sum dz x - mean sum dz) / std, with this group's mean and 1/std from it.
__global__ void __launch_bounds__(256) bn_bwd_final_kernel(const float* partial, float* coef,
                                                           void* dgamma, void* dbeta,
                                                           int nblocks, int C, float inv_count,
                                                           int out_bf16, long ostride,
                                                           const float* raw_stats) {
    float sb, sg;
    partial += static_cast<long>(blockIdx.y) * nblocks * 2 * C;
    coef += static_cast<long>(blockIdx.y) * 2 * C;
    const int c = final_sums(partial, nblocks, C, sb, sg);
    if (c < 0) return;
    if (raw_stats != nullptr) {
        const float* st = raw_stats + static_cast<long>(blockIdx.y) * 2 * C;
        sg = (sg - st[c] * sb) * st[C + c];
    }
    const long o = blockIdx.y * ostride + c;
    if (out_bf16) {
        static_cast<bf16*>(dbeta)[o] = f2bf(sb);
        static_cast<bf16*>(dgamma)[o] = f2bf(sg);
    } else {
        static_cast<float*>(dbeta)[o] = sb;
        static_cast<float*>(dgamma)[o] = sg;
    }
    coef[c] = sb * inv_count;
    coef[C + c] = sg * inv_count;
}

// dx = (dz - mean(dz) - xhat mean(dz xhat)) gamma / std, written as dz fa - x fb + fc with
// per-channel factors the thread computes once (its channels do not change over the loop)
template <bool kRelu, bool kRes>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const bf16* dy, const bf16* dy2,
                                                           const bf16* y, const bf16* x,
                                                           const float* stats, const float* coef,
                                                           const float* gamma, bf16* dx,
                                                           bf16* dres, long n8, int C,
                                                           long pstride) {
    {
        const long gofs = static_cast<long>(blockIdx.y) * n8 * 8;
        dy += gofs;
        if (dy2 != nullptr) dy2 += gofs;
        x += gofs;
        dx += gofs;
        if (kRelu) y += gofs;
        if (kRes) dres += gofs;
        stats += static_cast<long>(blockIdx.y) * 2 * C;
        coef += static_cast<long>(blockIdx.y) * 2 * C;
        gamma += blockIdx.y * pstride;
    }
    const long first = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
    const int c0 = static_cast<int>((first * 8) % C);
    float fa[8], fb[8], fc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float mu = stats[c0 + k], rs = stats[C + c0 + k];
        const float k0 = coef[c0 + k], k1 = coef[C + c0 + k];
        fa[k] = rs * gamma[c0 + k];
        fb[k] = rs * k1 * fa[k];
        fc[k] = fmaf(mu, fb[k], -k0 * fa[k]);
    }
    for (long i = first; i < n8; i += static_cast<long>(gridDim.x) * 256) {
        float d[8], xv[8], o[8];
        load_dy(dy, dy2, i * 8, d);
        load8(x + i * 8, xv);
        if (kRelu) {
            float yv[8];
            load8(y + i * 8, yv);
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = yv[k] > 0.f ? d[k] : 0.f;
        }
        if (kRes) store8(dres + i * 8, d);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fmaf(d[k], fa[k], fmaf(-xv[k], fb[k], fc[k]));
        store8(dx + i * 8, o);
    }
}

inline int lanes_per_row(int C) {
    const int lpr = C / 8;
    if (C % 8 != 0 || lpr < 8 || lpr > 64 || (lpr & (lpr - 1)) != 0) return 0;
    return lpr;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// at least eight row iterations per workgroup, so the partial pass stays short
inline int blocks_for(long R, int lpr) {
    const long rows_per_iter = 4L * (64 / lpr) * 8;
    long b = (R + rows_per_iter - 1) / rows_per_iter;
    if (b > kMaxBlocks) b = kMaxBlocks;
    if (b < 1) b = 1;
    return static_cast<int>(b);
}

// workgroups of an apply pass over n8 pieces per group: about 8 pieces per thread once
// all groups together fill the chip (>= 1024 workgroups), never more than 2048 per group
inline int grid_cap(long n8, int G = 1) {
    long b = (n8 + 255) / 256;
    while (b > 1 && (b / 2) * G >= 1024 && b * 256 * 8 > n8 + 255) b = (b + 1) / 2;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return static_cast<int>(b);
}

int bn_forward(const void* x, const void* res, void* y, const float* gamma, const float* beta,
               float* running_mean, float* running_var, float* stats, float* workspace, int G,
               long R, int C, int relu, float momentum, float eps, int pre_blocks, long pstride,
               hipStream_t s) {
    const int lpr = lanes_per_row(C);
    if (lpr == 0 || R <= 0 || pre_blocks < 0 || G < 1 || G > 65535) return -1;
    if (!al16(x) || !al16(y) || (res != nullptr && !al16(res))) return -2;
    if (G > 1 && (R * C) % 8 != 0) return -3;
    int nb = pre_blocks;
    if (pre_blocks == 0) {
        nb = blocks_for(R, lpr);
        const long rpb = (R + nb - 1) / nb;
        const size_t lds = static_cast<size_t>(8) * C * sizeof(float);
        hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nb, G), dim3(256), lds, s,
                           static_cast<const bf16*>(x), workspace, R, C, lpr, rpb);
    }
    const float unbias = R > 1 ? static_cast<float>(R) / static_cast<float>(R - 1) : 1.f;
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + kFinalCh - 1) / kFinalCh, G), dim3(256), 0, s, workspace,
                       stats, running_mean, running_var, nb, C, 1.f / static_cast<float>(R),
                       unbias, momentum, eps, pstride);
    const long n8 = R * C / 8;
    const dim3 grid(grid_cap(n8, G), G), block(256);
    const bf16* xp = static_cast<const bf16*>(x);
    const bf16* rp = static_cast<const bf16*>(res);
    bf16* yp = static_cast<bf16*>(y);
    if (relu && res != nullptr)
        hipLaunchKernelGGL((bn_apply_kernel<true, true>), grid, block, 0, s, xp, rp, stats, gamma, beta, yp, n8, C, pstride);
    else if (relu)
        hipLaunchKernelGGL((bn_apply_kernel<true, false>), grid, block, 0, s, xp, rp, stats, gamma, beta, yp, n8, C, pstride);
    else if (res != nullptr)
        hipLaunchKernelGGL((bn_apply_kernel<false, true>), grid, block, 0, s, xp, rp, stats, gamma, beta, yp, n8, C, pstride);
    else
        hipLaunchKernelGGL((bn_apply_kernel<false, false>), grid, block, 0, s, xp, rp, stats, gamma, beta, yp, n8, C, pstride);
    return static_cast<int>(hipGetLastError());
}

// pre (may be null): [G][pre_blocks][2][C] partial sums of dz and dz * x, written by the
// kernel that produced dy = dz (already masked and joined: relu, dy2 and dres must be unset);
// the pass over the pixels that would compute them is skipped.
int bn_backward(const void* dy, const void* dy2, const void* y, const void* x, const float* stats,
                const float* gamma, void* dx, void* dres, void* dgamma, void* dbeta,
                float* workspace, int G, long R, int C, int relu, long pstride, int out_bf16,
                long ostride, hipStream_t s, const float* pre = nullptr, int pre_blocks = 0) {
    const int lpr = lanes_per_row(C);
    if (lpr == 0 || R <= 0 || G < 1 || G > 65535) return -1;
    if (!al16(dy) || !al16(x) || !al16(dx) || (relu && !al16(y)) ||
        (dres != nullptr && !al16(dres)) || (dy2 != nullptr && !al16(dy2)))
        return -2;
    if (G > 1 && (R * C) % 8 != 0) return -3;
    if (pre != nullptr && (pre_blocks < 1 || relu || dy2 != nullptr || dres != nullptr)) return -4;
    const int nb = pre != nullptr ? pre_blocks : blocks_for(R, lpr);
    const long rpb = (R + nb - 1) / nb;
    const size_t lds = static_cast<size_t>(8) * C * sizeof(float);
    float* coef = workspace + static_cast<long>(G) * kMaxBlocks * 2 * C;
    const bf16* dyp = static_cast<const bf16*>(dy);
    const bf16* dy2p = static_cast<const bf16*>(dy2);
    const bf16* yp = static_cast<const bf16*>(y);
    const bf16* xp = static_cast<const bf16*>(x);
    if (pre != nullptr)
        ;  // the producer of dy left the partial sums
    else if (relu)
        hipLaunchKernelGGL(bn_bwd_partial_kernel<true>, dim3(nb, G), dim3(256), lds, s, dyp, dy2p,
                           yp, xp, stats, workspace, R, C, lpr, rpb);
    else
        hipLaunchKernelGGL(bn_bwd_partial_kernel<false>, dim3(nb, G), dim3(256), lds, s, dyp, dy2p,
                           yp, xp, stats, workspace, R, C, lpr, rpb);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + kFinalCh - 1) / kFinalCh, G), dim3(256), 0, s,
                       pre != nullptr ? pre : workspace, coef, dgamma, dbeta, nb, C,
                       1.f / static_cast<float>(R), out_bf16, ostride,
                       pre != nullptr ? stats : static_cast<const float*>(nullptr));
    const long n8 = R * C / 8;
    const dim3 grid(grid_cap(n8, G), G), block(256);
    bf16* dxp = static_cast<bf16*>(dx);
    bf16* drp = static_cast<bf16*>(dres);
    if (relu && dres != nullptr)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true>), grid, block, 0, s, dyp, dy2p, yp, xp, stats, coef, gamma, dxp, drp, n8, C, pstride);
    else if (relu)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false>), grid, block, 0, s, dyp, dy2p, yp, xp, stats, coef, gamma, dxp, drp, n8, C, pstride);
    else if (dres != nullptr)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<false, true>), grid, block, 0, s, dyp, dy2p, yp, xp, stats, coef, gamma, dxp, drp, n8, C, pstride);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<false, false>), grid, block, 0, s, dyp, dy2p, yp, xp, stats, coef, gamma, dxp, drp, n8, C, pstride);
    return static_cast<int>(hipGetLastError());
}

}  // namespace

extern "C" {

// 1 when the channel count is supported (C/8 a power of two in [8, 64]).
int nf_bn_supported(int C) { return lanes_per_row(C) != 0; }

// floats of workspace per group (the backward pass needs 2 * C more per group)
int nf_bn_workspace_floats(int C) { return kMaxBlocks * 2 * C; }

// stats: 2*C floats (mean, 1/std) kept for the backward pass.  running_* may
// be null.  res may be null.  y may not alias x when relu is set (the backward
// pass reads both).  pre_blocks > 0: workspace already holds that many rows of
// per-channel partial sums [block][2][C] of x and x^2 (written by the kernel that
// produced x), so the pass over x that computes them is skipped.
int nf_bn_fwd(const void* x, const void* res, void* y, const float* gamma, const float* beta,
              float* running_mean, float* running_var, float* stats, float* workspace, long R,
              int C, int relu, float momentum, float eps, int pre_blocks, hipStream_t s) {
    return bn_forward(x, res, y, gamma, beta, running_mean, running_var, stats, workspace, 1, R,
                      C, relu, momentum, eps, pre_blocks, 0, s);
}

// Inference: y = relu((x - stats[0]) stats[1] gamma + beta (+ res)) with given per-channel
// mean (stats[0][c]) and 1/std (stats[1][c]) - the apply pass of nf_bn_fwd alone.
int nf_bn_apply(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                const float* stats, long R, int C, int relu, hipStream_t s) {
    if (lanes_per_row(C) == 0 || R <= 0) return -1;
    if (!al16(x) || !al16(y) || (res != nullptr && !al16(res))) return -2;
    const long n8 = R * C / 8;
    const dim3 grid(grid_cap(n8, 1), 1), block(256);
    const bf16* xp = static_cast<const bf16*>(x);
    const bf16* rp = static_cast<const bf16*>(res);
    bf16* yp = static_cast<bf16*>(y);
    if (relu && res != nullptr)
        hipLaunchKernelGGL((bn_apply_kernel<true, true>), grid, block, 0, s, xp, rp, stats, gamma, beta, yp, n8, C, 0L);
    else if (relu)
        hipLaunchKernelGGL((bn_apply_kernel<true, false>), grid, block, 0, s, xp, rp, stats, gamma, beta, yp, n8, C, 0L);
    else if (res != nullptr)
        hipLaunchKernelGGL((bn_apply_kernel<false, true>), grid, block, 0, s, xp, rp, stats, gamma, beta, yp, n8, C, 0L);
    else
        hipLaunchKernelGGL((bn_apply_kernel<false, false>), grid, block, 0, s, xp, rp, stats, gamma, beta, yp, n8, C, 0L);
    return static_cast<int>(hipGetLastError());
}

// dgamma / dbeta (fp32) are overwritten.  dres (may be null) receives the
// gradient of the residual input.  workspace: nf_bn_workspace_floats(C) + 2*C floats.
int nf_bn_bwd(const void* dy, const void* y, const void* x, const float* stats,
              const float* gamma, void* dx, void* dres, float* dgamma, float* dbeta,
              float* workspace, long R, int C, int relu, hipStream_t s) {
    return bn_backward(dy, nullptr, y, x, stats, gamma, dx, dres, dgamma, dbeta, workspace, 1, R,
                       C, relu, 0, 0, 0, s);
}

// G independent batches of R rows, stored one after the other in x / res / y.
// gamma, beta, running_mean, running_var: group 0's vectors; group g's start pstride
// floats later.  stats: [G][2][C].  workspace: G * nf_bn_workspace_floats(C) floats;
// with pre_blocks > 0 it holds [G][pre_blocks][2][C] partial sums.
int nf_bn_fwd_g(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                float* running_mean, float* running_var, float* stats, float* workspace, int G,
                long R, int C, int relu, float momentum, float eps, int pre_blocks, long pstride,
                hipStream_t s) {
    if (pre_blocks > kMaxBlocks * 64) return -1;
    return bn_forward(x, res, y, gamma, beta, running_mean, running_var, stats, workspace, G, R,
                      C, relu, momentum, eps, pre_blocks, pstride, s);
}

// Grouped backward.  dy2 (may be null): second addend of the incoming gradient.
// dgamma / dbeta: group 0's output vectors, fp32 or (out_bf16) bf16, group g's ostride
// elements later.  workspace: G * (nf_bn_workspace_floats(C) + 2 * C) floats.
int nf_bn_bwd_g(const void* dy, const void* dy2, const void* y, const void* x,
                const float* stats, const float* gamma, void* dx, void* dres, void* dgamma,
                void* dbeta, float* workspace, int G, long R, int C, int relu, long pstride,
                int out_bf16, long ostride, hipStream_t s) {
    return bn_backward(dy, dy2, y, x, stats, gamma, dx, dres, dgamma, dbeta, workspace, G, R, C,
                       relu, pstride, out_bf16, ostride, s);
}

// Grouped backward whose incoming gradient dz is final (masked by the ReLU, both branches
// added) and whose per-channel partial sums exist already: pre holds [G][pre_blocks][2][C]
// sums of dz and dz * x over disjoint pixel ranges (nf_conv3x3_dgrad_bn_g writes them).
// Runs the fixed-order final sum and the apply pass (reads dz and x, writes dx).
// workspace: G * (nf_bn_workspace_floats(C) + 2 * C) floats.
int nf_bn_bwd_pre_g(const void* dz, const void* x, const float* stats, const float* gamma,
                    void* dx, void* dgamma, void* dbeta, const float* pre, float* workspace,
                    int G, long R, int C, int pre_blocks, long pstride, int out_bf16,
                    long ostride, hipStream_t s) {
    if (pre == nullptr || pre_blocks < 1 || pre_blocks > kMaxBlocks * 64) return -1;
    return bn_backward(dz, nullptr, nullptr, x, stats, gamma, dx, nullptr, dgamma, dbeta,
                       workspace, G, R, C, 0, pstride, out_bf16, ostride, s, pre, pre_blocks);
}

}  // extern "C"
