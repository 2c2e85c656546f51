// Kernels around the convolutions of the ResNet-18 trainer that steps all clients of a
// GPU at once (engine/resnet_native.py): the batch is G groups (clients) of B images,
// every group with parameters of its own in a row of the parameter arena.
//
//   nf_stem_patches   uint8 NCHW images, gathered by row index and normalised, as the
//                     patch matrix of the 3x3 / stride 1 / padding 1 stem convolution
//                     (rows of 32 bf16: 27 patch entries, zero tail) - the image never
//                     exists as a float tensor
//   nf_pool_fc_ce     global average pooling, the classifier, softmax cross-entropy and
//                     their backward pass up to d(activation), one workgroup per image
//   nf_fc_grads       the classifier's weight / bias gradient and the loss / accuracy
//                     accumulators, one workgroup per group
//   nf_sgd_rows       SGD on rows of the fp32 arena from rows of a bf16 gradient buffer,
//                     refreshing the bf16 copy of the parameters
//
// The reference trains one client per process through torch.optim.SGD
// (nanofed/trainer/base.py:141-159); it has no ResNet, this serves the convolution-heavy
// configuration of the benchmark suite.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nf_device.h"

using namespace nfd;

namespace {

constexpr int kStemK = 32;     // patch row length (27 used)
constexpr int kMaxClasses = 16;
constexpr int kFcC = 512;      // channels of the pooled feature

struct Norm3 {
    float scale[3];  // 1 / (255 std)
    float shift[3];  // -mean / std
};

// one thread per output pixel: 27 byte loads, four 16-byte stores
__global__ void __launch_bounds__(256)
stem_patches_kernel(const uint8_t* __restrict__ images, const int64_t* __restrict__ rows,
                    bf16* __restrict__ col, long npix, int H, int W, Norm3 nm) {
    const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= npix) return;
    const int hw = H * W;
    const long b = i / hw;
    const int p = static_cast<int>(i - b * hw);
    const int oy = p / W, ox = p - oy * W;
    const uint8_t* img = images + rows[b] * 3L * hw;
    bf16 v[kStemK];
#pragma unroll
    for (int k = 27; k < kStemK; ++k) v[k] = f2bf(0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float f = 0.f;  // the padding of the NORMALISED image is zero
                if (in) f = fmaf(static_cast<float>(img[c * hw + iy * W + ix]), nm.scale[c], nm.shift[c]);
                v[(ky * 3 + kx) * 3 + c] = f2bf(f);
            }
        }
    bf16x8* dst = reinterpret_cast<bf16x8*>(col + i * kStemK);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = v[q * 8 + e];
        dst[q] = o;
    }
}

// x [N][P][512] bf16 (channels-last feature map of P pixels), w [G][classes][512] bf16 with
// group stride ws, bias fp32 with group stride bs.  Writes pooled [N][512] bf16,
// dlogits [N][16] fp32 (already divided by the group's batch size), loss / correct per
// image and dx [N][P][512] bf16.  Thread t owns channels 2t, 2t + 1.
__global__ void __launch_bounds__(256)
pool_fc_ce_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                  const float* __restrict__ bias, const int64_t* __restrict__ labels,
                  bf16* __restrict__ pooled, float* __restrict__ dlogits,
                  float* __restrict__ loss_rows, float* __restrict__ correct_rows,
                  bf16* __restrict__ dx, int P, int classes, int B, long ws, long bs, int train) {
    __shared__ float red[4][kMaxClasses];
    __shared__ float s_dl[kMaxClasses];
    const int img = blockIdx.x, g = img / B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16* xi = x + static_cast<long>(img) * P * kFcC + 2 * tid;
    float a0 = 0.f, a1 = 0.f;
    for (int p = 0; p < P; ++p) {
        const uint32_t u = *reinterpret_cast<const uint32_t*>(xi + static_cast<long>(p) * kFcC);
        a0 += __uint_as_float(u << 16);
        a1 += __uint_as_float(u & 0xFFFF0000u);
    }
    const float inv = 1.f / static_cast<float>(P);
    // the pooled feature is a bf16 tensor (as under autocast): the classifier sees its rounding
    const bf16 p0 = f2bf(a0 * inv), p1 = f2bf(a1 * inv);
    pooled[static_cast<long>(img) * kFcC + 2 * tid] = p0;
    pooled[static_cast<long>(img) * kFcC + 2 * tid + 1] = p1;
    const float f0 = bf2f(p0), f1 = bf2f(p1);
    const bf16* wg = w + g * ws + 2 * tid;
    float wl0[kMaxClasses], wl1[kMaxClasses];
#pragma unroll
    for (int j = 0; j < kMaxClasses; ++j) {
        wl0[j] = 0.f;
        wl1[j] = 0.f;
        if (j < classes) {
            wl0[j] = bf2f(wg[j * kFcC]);
            wl1[j] = bf2f(wg[j * kFcC + 1]);
        }
        const float part = wave_sum(fmaf(f0, wl0[j], f1 * wl1[j]));
        if (lane == 0) red[wave][j] = part;
    }
    __syncthreads();
    if (tid == 0) {
        const int label = static_cast<int>(labels[img]);
        float lg[kMaxClasses];
        float mx = -3.0e38f;
        int arg = 0;
        for (int j = 0; j < classes; ++j) {
            const float v = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]) + bias[g * bs + j];
            lg[j] = bf2f(f2bf(v));  // bf16 logits
            if (lg[j] > mx) { mx = lg[j]; arg = j; }
        }
        float se = 0.f;
        for (int j = 0; j < classes; ++j) se += __expf(lg[j] - mx);
        const float lse = mx + __logf(se);
        loss_rows[img] = lse - lg[label];
        correct_rows[img] = arg == label ? 1.f : 0.f;
        const float invb = 1.f / static_cast<float>(B);
        for (int j = 0; j < kMaxClasses; ++j) {
            float d = 0.f;
            if (j < classes) d = (__expf(lg[j] - lse) - (j == label ? 1.f : 0.f)) * invb;
            d = bf2f(f2bf(d));  // the gradient of bf16 logits is a bf16 tensor
            s_dl[j] = d;
            dlogits[static_cast<long>(img) * kMaxClasses + j] = d;
        }
    }
    if (!train) return;
    __syncthreads();
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxClasses; ++j) {
        d0 = fmaf(s_dl[j], wl0[j], d0);
        d1 = fmaf(s_dl[j], wl1[j], d1);
    }
    // d(pooled) is bf16; every pixel receives 1 / P of it
    const bf16 o0 = f2bf(bf2f(f2bf(d0)) * inv), o1 = f2bf(bf2f(f2bf(d1)) * inv);
    const uint32_t packed = (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, o1)) << 16) |
                            static_cast<uint32_t>(__builtin_bit_cast(uint16_t, o0));
    bf16* dxi = dx + static_cast<long>(img) * P * kFcC + 2 * tid;
    for (int p = 0; p < P; ++p)
        *reinterpret_cast<uint32_t*>(dxi + static_cast<long>(p) * kFcC) = packed;
}

// dW[g][j][c] = sum_b dlogits[b][j] pooled[b][c], db[g][j] = sum_b dlogits[b][j] (bf16, group
// stride gs); acc[g] += (mean loss, mean correct, 1).  Thread t owns channels t, t + 256.
__global__ void __launch_bounds__(256)
fc_grads_kernel(const bf16* __restrict__ pooled, const float* __restrict__ dlogits,
                const float* __restrict__ loss_rows, const float* __restrict__ correct_rows,
                bf16* __restrict__ dw, bf16* __restrict__ db, float* __restrict__ acc,
                int classes, int B, long gs) {
    extern __shared__ float s_dl[];  // [B][16]
    const int g = blockIdx.x, tid = threadIdx.x;
    const long img0 = static_cast<long>(g) * B;
    for (int i = tid; i < B * kMaxClasses; i += 256) s_dl[i] = dlogits[img0 * kMaxClasses + i];
    __syncthreads();
    float a0[kMaxClasses], a1[kMaxClasses];
#pragma unroll
    for (int j = 0; j < kMaxClasses; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
    for (int b = 0; b < B; ++b) {
        const float f0 = bf2f(pooled[(img0 + b) * kFcC + tid]);
        const float f1 = bf2f(pooled[(img0 + b) * kFcC + tid + 256]);
#pragma unroll
        for (int j = 0; j < kMaxClasses; ++j) {
            a0[j] = fmaf(s_dl[b * kMaxClasses + j], f0, a0[j]);
            a1[j] = fmaf(s_dl[b * kMaxClasses + j], f1, a1[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < kMaxClasses; ++j)
        if (j < classes) {
            dw[g * gs + j * kFcC + tid] = f2bf(a0[j]);
            dw[g * gs + j * kFcC + tid + 256] = f2bf(a1[j]);
        }
    if (tid < classes) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += s_dl[b * kMaxClasses + tid];
        db[g * gs + tid] = f2bf(s);
    }
    if (tid == 0 && acc != nullptr) {
        float l = 0.f, c = 0.f;
        for (int b = 0; b < B; ++b) {
            l += loss_rows[img0 + b];
            c += correct_rows[img0 + b];
        }
        acc[g * 3 + 0] += l / static_cast<float>(B);
        acc[g * 3 + 1] += c / static_cast<float>(B);
        acc[g * 3 + 2] += 1.f;
    }
}

// p[r][i] -= lr (g[r][i] + wd p[r][i]); shadow[r][i] = bf16(p[r][i]); 8 elements per thread
__global__ void __launch_bounds__(256)
sgd_rows_kernel(float* p, const bf16* g, bf16* shadow, long n8, long ps, long gs, long ss,
                float lr, float wd) {
    float* pr = p + blockIdx.y * ps;
    const bf16* gr = g + blockIdx.y * gs;
    bf16* sr = shadow + blockIdx.y * ss;
    for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n8;
         i += static_cast<long>(gridDim.x) * 256) {
        const bf16x8 gv = reinterpret_cast<const bf16x8*>(gr)[i];
        float4 a = reinterpret_cast<float4*>(pr)[2 * i];
        float4 b = reinterpret_cast<float4*>(pr)[2 * i + 1];
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        bf16x8 sv;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] -= lr * fmaf(wd, v[k], bf2f(gv[k]));
            sv[k] = f2bf(v[k]);
        }
        reinterpret_cast<float4*>(pr)[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(pr)[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
        reinterpret_cast<bf16x8*>(sr)[i] = sv;
    }
}

inline bool al(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

}  // namespace

extern "C" {

// col [N * H * W][32] bf16 from images [*][3][H][W] uint8 and rows [N] int64.
// scale[c] = 1 / (255 std[c]), shift[c] = -mean[c] / std[c].
int nf_stem_patches(const void* images, const void* rows, void* col, int N, int H, int W,
                    const float* scale, const float* shift, hipStream_t s) {
    if (N < 1 || H < 1 || W < 1 || !al(col, 16)) return -1;
    Norm3 nm;
    for (int c = 0; c < 3; ++c) {
        nm.scale[c] = scale[c];
        nm.shift[c] = shift[c];
    }
    const long npix = static_cast<long>(N) * H * W;
    hipLaunchKernelGGL(stem_patches_kernel, dim3(static_cast<unsigned>((npix + 255) / 256)),
                       dim3(256), 0, s, static_cast<const uint8_t*>(images),
                       static_cast<const int64_t*>(rows), static_cast<bf16*>(col), npix, H, W, nm);
    return static_cast<int>(hipGetLastError());
}

// G groups of B images with a [P][512] feature map each.  w / bias: group 0's classifier
// ([classes][512] bf16, [classes] fp32), group g's ws / bs elements later.  dx may be
// null with train == 0 (evaluation: pooled, loss and correct only).
int nf_pool_fc_ce(const void* x, const void* w, const float* bias, const void* labels,
                  void* pooled, float* dlogits, float* loss_rows, float* correct_rows, void* dx,
                  int G, int B, int P, int C, int classes, long ws, long bs, int train,
                  hipStream_t s) {
    if (G < 1 || B < 1 || P < 1 || C != kFcC || classes < 1 || classes > kMaxClasses) return -1;
    if (train && dx == nullptr) return -2;
    if (!al(x, 4) || !al(w, 4) || (ws & 1) || (dx != nullptr && !al(dx, 4))) return -3;
    hipLaunchKernelGGL(pool_fc_ce_kernel, dim3(G * B), dim3(256), 0, s,
                       static_cast<const bf16*>(x), static_cast<const bf16*>(w), bias,
                       static_cast<const int64_t*>(labels), static_cast<bf16*>(pooled), dlogits,
                       loss_rows, correct_rows, static_cast<bf16*>(dx), P, classes, B, ws, bs,
                       train);
    return static_cast<int>(hipGetLastError());
}

// dw / db: group 0's gradient slots inside a bf16 gradient buffer with rows of gs elements.
// acc (may be null): [G][3] running sums of (mean loss, mean accuracy, steps).
int nf_fc_grads(const void* pooled, const float* dlogits, const float* loss_rows,
                const float* correct_rows, void* dw, void* db, float* acc, int G, int B, int C,
                int classes, long gs, hipStream_t s) {
    if (G < 1 || B < 1 || B > 1024 || C != kFcC || classes < 1 || classes > kMaxClasses) return -1;
    hipLaunchKernelGGL(fc_grads_kernel, dim3(G), dim3(256), B * kMaxClasses * sizeof(float), s,
                       static_cast<const bf16*>(pooled), dlogits, loss_rows, correct_rows,
                       static_cast<bf16*>(dw), static_cast<bf16*>(db), acc, classes, B, gs);
    return static_cast<int>(hipGetLastError());
}

// rows x n elements (n a multiple of 8); row strides ps (fp32 parameters), gs (bf16
// gradients), ss (bf16 copy) in elements, multiples of 8.
int nf_sgd_rows(float* p, const void* g, void* shadow, int rows, long n, long ps, long gs,
                long ss, float lr, float wd, hipStream_t s) {
    if (rows < 1 || n < 8 || (n & 7) || (ps & 7) || (gs & 7) || (ss & 7)) return -1;
    if (!al(p, 16) || !al(g, 16) || !al(shadow, 16)) return -2;
    long blocks = ((n >> 3) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sgd_rows_kernel, dim3(static_cast<unsigned>(blocks), rows), dim3(256), 0, s,
                       p, static_cast<const bf16*>(g), static_cast<bf16*>(shadow), n >> 3, ps, gs,
                       ss, lr, wd);
    return static_cast<int>(hipGetLastError());
}

}  // extern "C"
