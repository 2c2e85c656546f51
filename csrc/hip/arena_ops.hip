// Flat-arena kernels for gfx950: everything the federation does to a model
// between local-training steps is an elementwise pass or a reduction over one
// contiguous fp32 buffer (the parameter arena), so each operation is a single
// launch regardless of how many tensors the model has.
//
//   nf_weighted_sum       out = sum_k w_k * src_k           (FedAvg / share sum)
//   nf_sqnorm_finite      sum of squares + count of non-finite values
//   nf_clip_noise         out = in * min(1, C/(||in||+eps)) + N(0, std^2)
//   nf_sgd_step           p -= lr * (g + wd * p); optional grad zeroing and
//                         bf16 shadow copy of p
//   nf_philox_fill        out (+)= scale * {uniform, normal, laplace, sym-uniform}
//   nf_scale_accumulate   out = a * out + b * in
//
// All are HBM-bandwidth bound: 16-byte vector accesses, grid capped at 2048
// blocks of 256 threads with a grid-stride loop, wave64 shuffle reductions.
// Scalars that depend on device results (weights, the squared norm) are read
// from device memory, so sequences of these kernels can be captured in a
// hipGraph without host round trips.  Nothing here uses hipMemsetAsync:
// reduction partials are fully overwritten by the first stage.

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include "../common/philox.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;
constexpr int kMaxSources = 64;

inline int grid_for(long n_vec) {
    long blocks = (n_vec + kThreads - 1) / kThreads;
    if (blocks < 1) blocks = 1;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    return static_cast<int>(blocks);
}

struct SourceTable {
    const float* src[kMaxSources];
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Block-wide sum of two values; result valid in thread 0.
__device__ __forceinline__ void block_sum2(float& a, float& b) {
    __shared__ float sa[kThreads / 64];
    __shared__ float sb[kThreads / 64];
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sa[wave] = a; sb[wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) { ta += sa[w]; tb += sb[w]; }
        a = ta; b = tb;
    }
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
weighted_sum_kernel(float* __restrict__ out, SourceTable table,
                    const float* __restrict__ weights, int k, long n,
                    int accumulate) {
    __shared__ float w[kMaxSources];
    if (threadIdx.x < k) w[threadIdx.x] = weights[threadIdx.x];
    __syncthreads();

    const long n4 = n >> 2;
    const long stride = static_cast<long>(gridDim.x) * blockDim.x;
    for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4;
         i += stride) {
        float4 acc = accumulate ? reinterpret_cast<const float4*>(out)[i]
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < k; ++s) {
            const float ws = w[s];
            if (ws == 0.f) continue;  // excluded source: never read (may hold NaNs)
            const float4 v = reinterpret_cast<const float4*>(table.src[s])[i];
            acc.x = fmaf(ws, v.x, acc.x);
            acc.y = fmaf(ws, v.y, acc.y);
            acc.z = fmaf(ws, v.z, acc.z);
            acc.w = fmaf(ws, v.w, acc.w);
        }
        reinterpret_cast<float4*>(out)[i] = acc;
    }
    // tail (n not a multiple of 4)
    const long tail0 = n4 << 2;
    const long t = tail0 + static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t < n) {
        float acc = accumulate ? out[t] : 0.f;
        for (int s = 0; s < k; ++s)
            if (w[s] != 0.f) acc = fmaf(w[s], table.src[s][t], acc);
        out[t] = acc;
    }
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
sqnorm_finite_stage1(const float* __restrict__ x, long n,
                     float* __restrict__ partial /* [2 * gridDim.x] */) {
    float sq = 0.f, bad = 0.f;
    const long n4 = n >> 2;
    const long stride = static_cast<long>(gridDim.x) * blockDim.x;
    for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4;
         i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = isfinite(e[j]);
            sq += ok ? e[j] * e[j] : 0.f;
            bad += ok ? 0.f : 1.f;
        }
    }
    const long t = (n4 << 2) + static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t < n) {
        const float e = x[t];
        const bool ok = isfinite(e);
        sq += ok ? e * e : 0.f;
        bad += ok ? 0.f : 1.f;
    }
    block_sum2(sq, bad);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sq;
        partial[2 * blockIdx.x + 1] = bad;
    }
}

__global__ void __launch_bounds__(kThreads)
sqnorm_finite_stage2(const float* __restrict__ partial, int blocks,
                     float* __restrict__ out /* [2] */) {
    float sq = 0.f, bad = 0.f;
    for (int i = threadIdx.x; i < blocks; i += blockDim.x) {
        sq += partial[2 * i];
        bad += partial[2 * i + 1];
    }
    block_sum2(sq, bad);
    if (threadIdx.x == 0) { out[0] = sq; out[1] = bad; }
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
clip_noise_kernel(const float* __restrict__ in, float* __restrict__ out, long n,
                  const float* __restrict__ sqnorm, float max_norm, float noise_std,
                  unsigned long long seed, unsigned long long offset) {
    const float norm = sqrtf(sqnorm[0]);
    const float coef = fminf(max_norm / (norm + 1e-6f), 1.0f);
    const long n4 = (n + 3) >> 2;
    const long stride = static_cast<long>(gridDim.x) * blockDim.x;
    for (long b = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; b < n4;
         b += stride) {
        float z[4];
        nf::philox_samples(seed, offset, static_cast<unsigned long long>(b), 1, z);
        const long base = b << 2;
        if (base + 3 < n) {
            const float4 v = reinterpret_cast<const float4*>(in)[b];
            float4 r;
            r.x = fmaf(v.x, coef, noise_std * z[0]);
            r.y = fmaf(v.y, coef, noise_std * z[1]);
            r.z = fmaf(v.z, coef, noise_std * z[2]);
            r.w = fmaf(v.w, coef, noise_std * z[3]);
            reinterpret_cast<float4*>(out)[b] = r;
        } else {
            for (int j = 0; j < 4 && base + j < n; ++j)
                out[base + j] = fmaf(in[base + j], coef, noise_std * z[j]);
        }
    }
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
sgd_step_kernel(float* __restrict__ p, float* __restrict__ g, long n, float lr,
                float weight_decay, int zero_grad,
                __hip_bfloat16* __restrict__ shadow) {
    const long n4 = n >> 2;
    const long stride = static_cast<long>(gridDim.x) * blockDim.x;
    for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4;
         i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        pv.x -= lr * fmaf(weight_decay, pv.x, gv.x);
        pv.y -= lr * fmaf(weight_decay, pv.y, gv.y);
        pv.z -= lr * fmaf(weight_decay, pv.z, gv.z);
        pv.w -= lr * fmaf(weight_decay, pv.w, gv.w);
        reinterpret_cast<float4*>(p)[i] = pv;
        if (zero_grad)
            reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (shadow) {
            __hip_bfloat16 h[4] = {__float2bfloat16(pv.x), __float2bfloat16(pv.y),
                                   __float2bfloat16(pv.z), __float2bfloat16(pv.w)};
            reinterpret_cast<uint2*>(shadow)[i] = *reinterpret_cast<uint2*>(h);
        }
    }
    const long t = (n4 << 2) + static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t < n) {
        float pv = p[t];
        pv -= lr * fmaf(weight_decay, pv, g[t]);
        p[t] = pv;
        if (zero_grad) g[t] = 0.f;
        if (shadow) shadow[t] = __float2bfloat16(pv);
    }
}

// ---------------------------------------------------------------------------
// SGD over a flat fp32 parameter buffer whose gradients are separate tensors
// (what autograd hands out when it is not made to accumulate into a flat buffer):
// entry e updates p[offset[e] .. offset[e] + numel[e]) from grad[e], fp32 or bf16.
// The table travels by value in the kernel arguments, so a captured graph keeps it.
constexpr int kSgdMaxTensors = 96;
struct SgdTable {
    const void* grad[kSgdMaxTensors];
    long offset[kSgdMaxTensors];
    int numel[kSgdMaxTensors];
    unsigned bf16_mask[kSgdMaxTensors / 32];
};

__global__ void __launch_bounds__(kThreads)
sgd_multi_kernel(float* __restrict__ params, __hip_bfloat16* __restrict__ shadow_base, SgdTable t,
                 float lr, float weight_decay) {
    const int e = blockIdx.y;
    const int n = t.numel[e];
    float* p = params + t.offset[e];
    __hip_bfloat16* shadow = shadow_base ? shadow_base + t.offset[e] : nullptr;
    const bool g_bf16 = (t.bf16_mask[e >> 5] >> (e & 31)) & 1u;
    const float* gf = static_cast<const float*>(t.grad[e]);
    const __hip_bfloat16* gh = static_cast<const __hip_bfloat16*>(t.grad[e]);
    const int n4 = n >> 2;
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        float4 gv;
        if (g_bf16) {
            const uint2 raw = reinterpret_cast<const uint2*>(gh)[i];
            gv.x = __uint_as_float(raw.x << 16);
            gv.y = __uint_as_float(raw.x & 0xFFFF0000u);
            gv.z = __uint_as_float(raw.y << 16);
            gv.w = __uint_as_float(raw.y & 0xFFFF0000u);
        } else {
            gv = reinterpret_cast<const float4*>(gf)[i];
        }
        pv.x -= lr * fmaf(weight_decay, pv.x, gv.x);
        pv.y -= lr * fmaf(weight_decay, pv.y, gv.y);
        pv.z -= lr * fmaf(weight_decay, pv.z, gv.z);
        pv.w -= lr * fmaf(weight_decay, pv.w, gv.w);
        reinterpret_cast<float4*>(p)[i] = pv;
        if (shadow) {
            __hip_bfloat16 h[4] = {__float2bfloat16(pv.x), __float2bfloat16(pv.y),
                                   __float2bfloat16(pv.z), __float2bfloat16(pv.w)};
            reinterpret_cast<uint2*>(shadow)[i] = *reinterpret_cast<uint2*>(h);
        }
    }
    const int tail = (n4 << 2) + blockIdx.x * blockDim.x + threadIdx.x;
    if (tail < n) {
        float pv = p[tail];
        const float g = g_bf16 ? __bfloat162float(gh[tail]) : gf[tail];
        pv -= lr * fmaf(weight_decay, pv, g);
        p[tail] = pv;
        if (shadow) shadow[tail] = __float2bfloat16(pv);
    }
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
philox_fill_kernel(float* __restrict__ out, long n, unsigned long long seed,
                   unsigned long long offset, long first, int mode, float scale,
                   int accumulate) {
    // element e = first + i; block b = e >> 2. Iterate over the blocks that
    // overlap [first, first + n).
    const long b0 = first >> 2;
    const long b1 = (first + n + 3) >> 2;
    const long stride = static_cast<long>(gridDim.x) * blockDim.x;
    for (long b = b0 + static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; b < b1;
         b += stride) {
        float s[4];
        nf::philox_samples(seed, offset, static_cast<unsigned long long>(b), mode, s);
        const long i0 = (b << 2) - first;
        if (i0 >= 0 && i0 + 3 < n && ((reinterpret_cast<uintptr_t>(out + i0) & 15) == 0)) {
            float4 v = accumulate ? *reinterpret_cast<float4*>(out + i0)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            v.x = fmaf(scale, s[0], v.x);
            v.y = fmaf(scale, s[1], v.y);
            v.z = fmaf(scale, s[2], v.z);
            v.w = fmaf(scale, s[3], v.w);
            *reinterpret_cast<float4*>(out + i0) = v;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long i = i0 + j;
                if (i >= 0 && i < n) {
                    const float base = accumulate ? out[i] : 0.f;
                    out[i] = fmaf(scale, s[j], base);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
scale_accumulate_kernel(float* __restrict__ out, const float* __restrict__ in, long n,
                        float a, float b) {
    const long n4 = n >> 2;
    const long stride = static_cast<long>(gridDim.x) * blockDim.x;
    for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4;
         i += stride) {
        float4 o = reinterpret_cast<float4*>(out)[i];
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        o.x = fmaf(b, v.x, a * o.x);
        o.y = fmaf(b, v.y, a * o.y);
        o.z = fmaf(b, v.z, a * o.z);
        o.w = fmaf(b, v.w, a * o.w);
        reinterpret_cast<float4*>(out)[i] = o;
    }
    const long t = (n4 << 2) + static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t < n) out[t] = fmaf(b, in[t], a * out[t]);
}

// ---------------------------------------------------------------------------
// Screening of all G client rows of a pool in two launches.
//   stage 1, grid (bx, G): block (x, g) reduces its slice of row g (optionally of
//            row g - ref) to [sum of squares of finite values, non-finite count]
//   stage 2, grid (G):     fixed-order sum of the bx partials of row g, then the
//            verdict: records[g][col_sq] = sum of squares,
//            records[g][col_valid] *= (no non-finite value) && (norm bound holds)
constexpr int kScreenBlocks = 64;

__global__ void __launch_bounds__(kThreads)
screen_rows_stage1(const float* __restrict__ rows, long row_stride, long n,
                   const float* __restrict__ ref, float* __restrict__ partial) {
    const int g = blockIdx.y;
    const float* x = rows + static_cast<long>(g) * row_stride;
    float sq = 0.f, bad = 0.f;
    const long n4 = n >> 2;
    const long stride = static_cast<long>(gridDim.x) * blockDim.x;
    for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        if (ref != nullptr) {
            const float4 r = reinterpret_cast<const float4*>(ref)[i];
            v.x -= r.x; v.y -= r.y; v.z -= r.z; v.w -= r.w;
        }
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = isfinite(e[j]);
            sq += ok ? e[j] * e[j] : 0.f;
            bad += ok ? 0.f : 1.f;
        }
    }
    const long t = (n4 << 2) + static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t < n) {
        const float e = ref != nullptr ? x[t] - ref[t] : x[t];
        const bool ok = isfinite(e);
        sq += ok ? e * e : 0.f;
        bad += ok ? 0.f : 1.f;
    }
    block_sum2(sq, bad);
    if (threadIdx.x == 0) {
        float* p = partial + (static_cast<long>(g) * gridDim.x + blockIdx.x) * 2;
        p[0] = sq;
        p[1] = bad;
    }
}

__global__ void __launch_bounds__(64)
screen_rows_stage2(const float* __restrict__ partial, int blocks, float* __restrict__ records,
                   int rec_stride, int col_sq, int col_valid, float max_sq) {
    const int g = blockIdx.x;
    if (threadIdx.x != 0) return;
    const float* p = partial + static_cast<long>(g) * blocks * 2;
    float sq = 0.f, bad = 0.f;
    for (int i = 0; i < blocks; ++i) { sq += p[2 * i]; bad += p[2 * i + 1]; }
    float* rec = records + static_cast<long>(g) * rec_stride;
    const bool valid = bad == 0.f && (max_sq <= 0.f || sq <= max_sq);
    rec[col_sq] = sq;
    rec[col_valid] = valid ? rec[col_valid] : 0.f;
}

// ---------------------------------------------------------------------------
// FedAvg weights from the gathered per-client records, one workgroup:
//   1. optional anomaly screen: client k is dropped when the norm of its update
//      is more than z_thr standard deviations (unbiased, of the OTHER valid
//      clients) away from their mean and at least min_stats others exist
//   2. weights[k] = valid_k n_k / sum(valid n)
//   3. flags = [round succeeded, valid clients, total samples, anomalous clients]
// K is small (clients of the whole job), sums are taken by thread 0 in index order.
__global__ void __launch_bounds__(64)
round_weights_kernel(float* __restrict__ records, int K, int M, int col_samples, int col_valid,
                     int col_sq, float required, float z_thr, int min_stats,
                     float* __restrict__ weights, float* __restrict__ flags) {
    if (threadIdx.x != 0) return;
    int anomalous = 0;
    if (z_thr > 0.f) {
        double S = 0.0, Q = 0.0;
        int m = 0;
        for (int k = 0; k < K; ++k) {
            if (records[k * M + col_valid] > 0.f) {
                const double x = sqrt(static_cast<double>(records[k * M + col_sq]));
                S += x; Q += x * x; ++m;
            }
        }
        const int others = m - 1;
        if (others >= min_stats && others >= 2) {
            for (int k = 0; k < K; ++k) {
                if (!(records[k * M + col_valid] > 0.f)) continue;
                const double x = sqrt(static_cast<double>(records[k * M + col_sq]));
                const double mean = (S - x) / others;
                double var = (Q - x * x - others * mean * mean) / (others - 1);
                if (var < 0.0) var = 0.0;
                const double z = fabs(x - mean) / (sqrt(var) + 1e-8);
                if (z > z_thr) { weights[k] = -1.f; ++anomalous; }
                else weights[k] = 0.f;
            }
            for (int k = 0; k < K; ++k)
                if (records[k * M + col_valid] > 0.f && weights[k] < 0.f)
                    records[k * M + col_valid] = 0.f;
        }
    }
    float total = 0.f, nvalid = 0.f;
    for (int k = 0; k < K; ++k) {
        const float v = records[k * M + col_valid] > 0.f ? 1.f : 0.f;
        total += v * records[k * M + col_samples];
        nvalid += v;
    }
    const float inv = 1.0f / fmaxf(total, 1e-30f);
    for (int k = 0; k < K; ++k) {
        const float v = records[k * M + col_valid] > 0.f ? 1.f : 0.f;
        weights[k] = v * records[k * M + col_samples] * inv;
    }
    flags[0] = (nvalid >= required && total > 0.f) ? 1.f : 0.f;
    flags[1] = nvalid;
    flags[2] = total;
    flags[3] = static_cast<float>(anomalous);
}

// dst = src when flag[0] > 0 (quorum select: the old model survives a failed round)
__global__ void __launch_bounds__(kThreads)
select_copy_kernel(float* __restrict__ dst, const float* __restrict__ src,
                   const float* __restrict__ flag, long n) {
    if (!(flag[0] > 0.f)) return;
    const long n4 = n >> 2;
    const long stride = static_cast<long>(gridDim.x) * blockDim.x;
    for (long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    const long t = (n4 << 2) + static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = src[t];
}

// ---------------------------------------------------------------------------
// Pairwise-mask secure aggregation on flat rows.
//   mask:   row = prescale * row + sum_p sign[p] * scale * U(seed[p])   (one client,
//           all its peers, one pass)
//   repair: total -= the masks that did not cancel because exactly one client of
//           a pair is present (present[] and the K x K seed table live in device
//           memory, so no host decision is needed between reduce and repair)
// The uniform draw of a pair is philox(seed, offset, block) in MODE_SYM_UNIFORM,
// the same stream philox_fill produces.
constexpr int kMaxPeers = 64;
struct PeerTable {
    unsigned long long seed[kMaxPeers];
    float sign[kMaxPeers];
};

__global__ void __launch_bounds__(kThreads)
pairwise_mask_kernel(float* __restrict__ row, long n, float prescale, PeerTable t, int peers,
                     float scale, unsigned long long offset) {
    const long n4 = (n + 3) >> 2;
    const long stride = static_cast<long>(gridDim.x) * blockDim.x;
    for (long b = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; b < n4; b += stride) {
        float m[4] = {0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < peers; ++p) {
            float s[4];
            nf::philox_samples(t.seed[p], offset, static_cast<unsigned long long>(b), 3, s);
            const float a = t.sign[p] * scale;
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = fmaf(a, s[j], m[j]);
        }
        const long base = b << 2;
        if (base + 3 < n) {
            float4 v = reinterpret_cast<float4*>(row)[b];
            v.x = fmaf(prescale, v.x, m[0]); v.y = fmaf(prescale, v.y, m[1]);
            v.z = fmaf(prescale, v.z, m[2]); v.w = fmaf(prescale, v.w, m[3]);
            reinterpret_cast<float4*>(row)[b] = v;
        } else {
            for (int j = 0; j < 4 && base + j < n; ++j)
                row[base + j] = fmaf(prescale, row[base + j], m[j]);
        }
    }
}

__global__ void __launch_bounds__(kThreads)
pairwise_repair_kernel(float* __restrict__ total, long n, const float* __restrict__ present,
                       const unsigned long long* __restrict__ seeds /* [K][K] */, int K,
                       float scale, unsigned long long offset) {
    const long n4 = (n + 3) >> 2;
    const long stride = static_cast<long>(gridDim.x) * blockDim.x;
    for (long b = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; b < n4; b += stride) {
        float m[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < K; ++i) {
            const bool pi = present[i] > 0.f;
            for (int j = i + 1; j < K; ++j) {
                const bool pj = present[j] > 0.f;
                if (pi == pj) continue;  // both masks arrived (they cancel) or neither did
                // the survivor applied +scale when it is the smaller id, -scale otherwise
                const float a = pi ? scale : -scale;
                float s[4];
                nf::philox_samples(seeds[i * K + j], offset, static_cast<unsigned long long>(b), 3, s);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaf(a, s[e], m[e]);
            }
        }
        const long base = b << 2;
        if (base + 3 < n) {
            float4 v = reinterpret_cast<float4*>(total)[b];
            v.x -= m[0]; v.y -= m[1]; v.z -= m[2]; v.w -= m[3];
            reinterpret_cast<float4*>(total)[b] = v;
        } else {
            for (int j = 0; j < 4 && base + j < n; ++j) total[base + j] -= m[j];
        }
    }
}

}  // namespace

#define NF_CHECK_LAUNCH() (static_cast<int>(hipGetLastError()))

extern "C" {

int nf_max_sources() { return kMaxSources; }
int nf_reduce_workspace_floats() { return 2 * kMaxBlocks + 2; }

// srcs: host array of k device pointers; weights: device pointer to k floats.
int nf_weighted_sum(float* out, const float* const* srcs, const float* weights,
                    int k, long n, int accumulate, hipStream_t stream) {
    if (k < 1 || k > kMaxSources || n < 0) return -1;
    if (n == 0) return 0;
    SourceTable table;
    for (int i = 0; i < kMaxSources; ++i) table.src[i] = i < k ? srcs[i] : nullptr;
    const int grid = grid_for((n + 3) >> 2);
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(grid), dim3(kThreads), 0, stream,
                       out, table, weights, k, n, accumulate);
    return NF_CHECK_LAUNCH();
}

// workspace: device buffer of nf_reduce_workspace_floats() floats.
// result: device buffer of 2 floats (sum of squares over finite values,
// number of non-finite values).
int nf_sqnorm_finite(const float* x, long n, float* workspace, float* result,
                     hipStream_t stream) {
    if (n < 0) return -1;
    const int grid = grid_for((n + 3) >> 2);
    hipLaunchKernelGGL(sqnorm_finite_stage1, dim3(grid), dim3(kThreads), 0, stream,
                       x, n, workspace);
    hipLaunchKernelGGL(sqnorm_finite_stage2, dim3(1), dim3(kThreads), 0, stream,
                       workspace, grid, result);
    return NF_CHECK_LAUNCH();
}

int nf_screen_workspace_floats(int rows) { return rows * kScreenBlocks * 2; }
int nf_max_peers() { return kMaxPeers; }

// rows: [G][row_stride] (n used elements each); ref: null or n floats subtracted
// from every row first; workspace: nf_screen_workspace_floats(G) floats;
// records: [G][rec_stride], columns col_sq / col_valid are written.
int nf_screen_rows(const float* rows, int G, long row_stride, long n, const float* ref,
                   float* workspace, float* records, int rec_stride, int col_sq, int col_valid,
                   float max_sq, hipStream_t stream) {
    if (G < 1 || n < 0 || row_stride < n || col_sq < 0 || col_valid < 0 ||
        col_sq >= rec_stride || col_valid >= rec_stride)
        return -1;
    if ((row_stride & 3) != 0) return -2;  // every row must keep float4 alignment
    int bx = grid_for((n + 3) >> 2);
    if (bx > kScreenBlocks) bx = kScreenBlocks;
    hipLaunchKernelGGL(screen_rows_stage1, dim3(bx, G), dim3(kThreads), 0, stream, rows,
                       row_stride, n, ref, workspace);
    hipLaunchKernelGGL(screen_rows_stage2, dim3(G), dim3(64), 0, stream, workspace, bx, records,
                       rec_stride, col_sq, col_valid, max_sq);
    return NF_CHECK_LAUNCH();
}

int nf_round_weights(float* records, int K, int M, int col_samples, int col_valid, int col_sq,
                     float required, float z_thr, int min_stats, float* weights, float* flags,
                     hipStream_t stream) {
    if (K < 1 || M < 1 || col_samples >= M || col_valid >= M || col_sq >= M) return -1;
    hipLaunchKernelGGL(round_weights_kernel, dim3(1), dim3(64), 0, stream, records, K, M,
                       col_samples, col_valid, col_sq, required, z_thr, min_stats, weights, flags);
    return NF_CHECK_LAUNCH();
}

int nf_select_copy(float* dst, const float* src, const float* flag, long n, hipStream_t stream) {
    if (n <= 0) return n == 0 ? 0 : -1;
    hipLaunchKernelGGL(select_copy_kernel, dim3(grid_for((n + 3) >> 2)), dim3(kThreads), 0, stream,
                       dst, src, flag, n);
    return NF_CHECK_LAUNCH();
}

// seeds / signs: host arrays of `peers` entries (<= nf_max_peers() per call).
int nf_pairwise_mask(float* row, long n, float prescale, const unsigned long long* seeds,
                     const float* signs, int peers, float scale, unsigned long long offset,
                     hipStream_t stream) {
    if (n <= 0) return n == 0 ? 0 : -1;
    if (peers < 0 || peers > kMaxPeers) return -1;
    PeerTable t = {};
    for (int p = 0; p < peers; ++p) { t.seed[p] = seeds[p]; t.sign[p] = signs[p]; }
    hipLaunchKernelGGL(pairwise_mask_kernel, dim3(grid_for((n + 3) >> 2)), dim3(kThreads), 0,
                       stream, row, n, prescale, t, peers, scale, offset);
    return NF_CHECK_LAUNCH();
}

// present: device, K floats; seeds: device, K*K uint64 (entry [i][j], i < j, is used).
int nf_pairwise_repair(float* total, long n, const float* present,
                       const unsigned long long* seeds, int K, float scale,
                       unsigned long long offset, hipStream_t stream) {
    if (n <= 0) return n == 0 ? 0 : -1;
    if (K < 2) return -1;
    hipLaunchKernelGGL(pairwise_repair_kernel, dim3(grid_for((n + 3) >> 2)), dim3(kThreads), 0,
                       stream, total, n, present, seeds, K, scale, offset);
    return NF_CHECK_LAUNCH();
}

int nf_clip_noise(const float* in, float* out, long n, const float* sqnorm,
                  float max_norm, float noise_std, unsigned long long seed,
                  unsigned long long offset, hipStream_t stream) {
    if (n <= 0) return n == 0 ? 0 : -1;
    const int grid = grid_for((n + 3) >> 2);
    hipLaunchKernelGGL(clip_noise_kernel, dim3(grid), dim3(kThreads), 0, stream, in,
                       out, n, sqnorm, max_norm, noise_std, seed, offset);
    return NF_CHECK_LAUNCH();
}

int nf_sgd_step(float* p, float* g, long n, float lr, float weight_decay,
                int zero_grad, void* bf16_shadow, hipStream_t stream) {
    if (n <= 0) return n == 0 ? 0 : -1;
    const int grid = grid_for((n + 3) >> 2);
    hipLaunchKernelGGL(sgd_step_kernel, dim3(grid), dim3(kThreads), 0, stream, p, g,
                       n, lr, weight_decay, zero_grad,
                       static_cast<__hip_bfloat16*>(bf16_shadow));
    return NF_CHECK_LAUNCH();
}

// Multi-tensor SGD (see sgd_multi_kernel).  grads[e] must be 16-byte aligned,
// offsets multiples of 4 elements; is_bf16[e] != 0 marks bf16 gradients.
// shadow (may be null) is a bf16 buffer parallel to params.
int nf_sgd_multi(float* params, void* shadow, const void* const* grads, const long* offsets,
                 const long* numels, const int* is_bf16, int count, float lr,
                 float weight_decay, hipStream_t stream) {
    if (count < 0) return -1;
    for (int first = 0; first < count; first += kSgdMaxTensors) {
        const int m = count - first < kSgdMaxTensors ? count - first : kSgdMaxTensors;
        SgdTable t = {};
        long largest = 0;
        for (int e = 0; e < m; ++e) {
            const long n = numels[first + e];
            if (n < 0 || n >= (1L << 31) || (offsets[first + e] & 3) != 0) return -1;
            if ((reinterpret_cast<uintptr_t>(grads[first + e]) & 15) != 0) return -2;
            t.grad[e] = grads[first + e];
            t.offset[e] = offsets[first + e];
            t.numel[e] = static_cast<int>(n);
            if (is_bf16[first + e]) t.bf16_mask[e >> 5] |= 1u << (e & 31);
            if (n > largest) largest = n;
        }
        long bx = (largest / 4 + kThreads - 1) / kThreads;
        if (bx > 64) bx = 64;
        if (bx < 1) bx = 1;
        hipLaunchKernelGGL(sgd_multi_kernel, dim3(static_cast<unsigned>(bx), m), dim3(kThreads), 0,
                           stream, params, static_cast<__hip_bfloat16*>(shadow), t, lr,
                           weight_decay);
    }
    return static_cast<int>(hipGetLastError());
}

int nf_philox_fill(float* out, long n, unsigned long long seed,
                   unsigned long long offset, long first, int mode, float scale,
                   int accumulate, hipStream_t stream) {
    if (n <= 0) return n == 0 ? 0 : -1;
    const int grid = grid_for((n + 3) >> 2);
    hipLaunchKernelGGL(philox_fill_kernel, dim3(grid), dim3(kThreads), 0, stream, out,
                       n, seed, offset, first, mode, scale, accumulate);
    return NF_CHECK_LAUNCH();
}

int nf_scale_accumulate(float* out, const float* in, long n, float a, float b,
                        hipStream_t stream) {
    if (n <= 0) return n == 0 ? 0 : -1;
    const int grid = grid_for((n + 3) >> 2);
    hipLaunchKernelGGL(scale_accumulate_kernel, dim3(grid), dim3(kThreads), 0, stream,
                       out, in, n, a, b);
    return NF_CHECK_LAUNCH();
}

}  // extern "C"
