// Transformer building blocks for gfx950, bf16 activations with fp32 statistics.
// These are the memory-bound pieces around the library GEMMs of a GPT-2 style
// block (the model of the large-arena federated configuration):
//
//   nf_embed_fwd        x[r] = wte[token[r]] + wpe[r % T]
//   nf_embed_bwd        d(wte) += scatter of d(x) by token, without atomics: rows are
//                       visited in token-sorted order, one workgroup per distinct
//                       token; d(wpe)[t] = sum over the batch
//   nf_ln_fwd           optional residual add (x_out = x_in + delta) fused with
//                       LayerNorm; one wave per row, the row lives in registers
//   nf_ln_bwd           d(x) (+ incoming residual gradient), one wave per row; then a
//                       column pass for d(gamma), d(beta) and the column sums of d(x)
//                       (= bias gradient of the projection feeding the residual
//                       stream), whose partials a third launch sums in a fixed order
//   nf_gelu_fwd/bwd     tanh-GELU on the pre-activation (the bias comes from the GEMM
//                       epilogue); nf_gelu_bwd_colsum also produces the bias gradient
//                       of the layer in front of it
//   nf_colsum           bias gradient: column sums of a [R, N] matrix, two fixed-order
//                       stages
//   nf_ce_fwd_bwd       softmax cross-entropy over a padded-stride logits matrix:
//                       per-row loss and, in place, d(logits) = (p - onehot) * scale
//   nf_sgd_step_bf16g   SGD on fp32 master weights from bf16 gradients, refreshing
//                       the bf16 copy the next forward pass reads
//
// Every reduction has a fixed association order, so a step is reproducible.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nf_device.h"

using namespace nfd;

namespace {

constexpr int kLnMaxChunks = 8;  // 64 lanes x 4 values x 8 chunks = 2048 columns

constexpr float kGeluK = 0.7978845608028654f;  // sqrt(2/pi)
constexpr float kGeluC = 0.044715f;

__device__ __forceinline__ float tanh_fast(float u) {
    // tanh(u) = 1 - 2 / (exp(2u) + 1); exact limits at +-inf
    const float e = __expf(2.f * u);
    return 1.f - 2.f / (e + 1.f);
}

__device__ __forceinline__ float gelu_grad(float v) {
    const float t = tanh_fast(kGeluK * (v + kGeluC * v * v * v));
    const float dt = (1.f - t * t) * kGeluK * (1.f + 3.f * kGeluC * v * v);
    return 0.5f * (1.f + t) + 0.5f * v * dt;
}

__device__ __forceinline__ void load4(const bf16* p, float (&f)[4]) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = bf2f(a[i]);
}

__device__ __forceinline__ void store4(bf16* p, const float (&f)[4]) {
    bf16x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = f2bf(f[i]);
    *reinterpret_cast<bf16x4*>(p) = a;
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) embed_fwd_kernel(const long* tokens, const bf16* wte,
                                                        const bf16* wpe, bf16* x, int R, int T,
                                                        int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= R) return;
    const long tok = tokens[row];
    const bf16* te = wte + tok * C;
    const bf16* pe = wpe + static_cast<long>(row % T) * C;
    bf16* out = x + static_cast<long>(row) * C;
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
        float a[4], b[4];
        load4(te + c0, a);
        load4(pe + c0, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] += b[i];
        store4(out + c0, a);
    }
}

// sorted[i] = token of the i-th row in token order, order[i] = that row.
// One workgroup per position; only the first position of every run works.
__global__ void __launch_bounds__(256) embed_bwd_wte_kernel(const long* sorted, const long* order,
                                                            const bf16* dx, bf16* dwte, int R,
                                                            int C) {
    const int i = blockIdx.x;
    const long tok = sorted[i];
    if (i > 0 && sorted[i - 1] == tok) return;
    for (int c0 = threadIdx.x * 4; c0 < C; c0 += 1024) {
        float acc[4];
        load4(dwte + tok * C + c0, acc);
        for (int j = i; j < R && sorted[j] == tok; ++j) {
            float v[4];
            load4(dx + order[j] * C + c0, v);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += v[q];
        }
        store4(dwte + tok * C + c0, acc);
    }
}

__global__ void __launch_bounds__(256) embed_bwd_wpe_kernel(const bf16* dx, bf16* dwpe, int B,
                                                            int T, int C) {
    const int t = blockIdx.x;
    for (int c0 = threadIdx.x * 4; c0 < C; c0 += 1024) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            float v[4];
            load4(dx + (static_cast<long>(b) * T + t) * C + c0, v);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += v[q];
        }
        store4(dwpe + static_cast<long>(t) * C + c0, acc);
    }
}

// ---------------------------------------------------------------------------
// NCH = ceil(C / 256): the number of 4-value chunks a lane holds of its row.
template <bool kHasDelta, int NCH>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const bf16* x_in, const bf16* delta,
                                                     bf16* x_out, const bf16* gamma,
                                                     const bf16* beta, bf16* y, float* mean,
                                                     float* rstd, int R, int C, float eps,
                                                     int Rg, long ps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= R) return;
    if (ps != 0) {  // grouped: rows [g Rg, (g + 1) Rg) use the vectors of group g
        const long shift = (row / Rg) * ps;
        gamma += shift;
        beta += shift;
    }
    const long base = static_cast<long>(row) * C;
    float v[NCH][4];
    float s = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c0 = ch * 256 + lane * 4;
        if (c0 < C) {
            load4(x_in + base + c0, v[ch]);
            if (kHasDelta) {
                float d[4];
                load4(delta + base + c0, d);
                // the statistics use the value that is stored, so the backward pass
                // (which reads x_out) sees exactly the same row
#pragma unroll
                for (int i = 0; i < 4; ++i) v[ch][i] = bf2f(f2bf(v[ch][i] + d[i]));
                store4(x_out + base + c0, v[ch]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s += v[ch][i];
        }
    }
    const float mu = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c0 = ch * 256 + lane * 4;
        if (c0 < C) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = v[ch][i] - mu;
                q += d * d;
            }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / C + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c0 = ch * 256 + lane * 4;
        if (c0 < C) {
            float g[4], b[4], o[4];
            load4(gamma + c0, g);
            load4(beta + c0, b);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (v[ch][i] - mu) * rs * g[i] + b[i];
            store4(y + base + c0, o);
        }
    }
}

// d(x) and the three column sums in one pass: workgroup (chunk, group) takes the rows of its
// chunk, wave w every NW-th row (one row at a time in registers, the next row's dy, x and
// residual gradient already requested), and keeps per lane the running sums of its columns:
// d(gamma) = sum dy*xhat, d(beta) = sum dy and the column sums of the finished d(x) as
// stored (the bias gradient of the projection in front of the residual) ->
// partial[chunk][0..2][c].  dy, x and d(x) are not read a second time by a column-sum pass.
// dx may alias dres: a row is read and written by one wave only, and read first.
template <bool kHasRes, int NCH, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 4 : 1) ln_bwd_fused_kernel(
    const bf16* __restrict__ dy, const bf16* __restrict__ x, const bf16* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, const bf16* dres, bf16* dx,
    float* __restrict__ partial, int R, int C, int rows_per_chunk, long ps) {
    __shared__ float red[NW][NCH * 256];
    {   // blockIdx.y: group of R rows
        const long rofs = static_cast<long>(blockIdx.y) * R;
        dy += rofs * C;
        x += rofs * C;
        dx += rofs * C;
        if (kHasRes) dres += rofs * C;
        mean += rofs;
        rstd += rofs;
        gamma += blockIdx.y * ps;
        partial += (static_cast<long>(blockIdx.y) * gridDim.x + blockIdx.x) * 3 * C;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = min(R, r0 + rows_per_chunk);
    float ag[NCH][4], ab[NCH][4], as[NCH][4];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int i = 0; i < 4; ++i) { ag[ch][i] = 0.f; ab[ch][i] = 0.f; as[ch][i] = 0.f; }
    bf16x4 nx[NCH], nd[NCH], nr[NCH];
    auto request = [&](int r) {
        const long base = static_cast<long>(r) * C;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c0 = ch * 256 + lane * 4;
            if (c0 < C) {
                nx[ch] = *reinterpret_cast<const bf16x4*>(x + base + c0);
                nd[ch] = *reinterpret_cast<const bf16x4*>(dy + base + c0);
                if (kHasRes) nr[ch] = *reinterpret_cast<const bf16x4*>(dres + base + c0);
            }
        }
    };
    if (r0 + wave < r1) request(r0 + wave);
    for (int r = r0 + wave; r < r1; r += NW) {
        const long base = static_cast<long>(r) * C;
        const float mu = mean[r], rs = rstd[r];
        float xh[NCH][4], dyw[NCH][4];
        bf16x4 rv[NCH];  // (kept packed: the kernel has to fit 128 registers for 16 waves per CU)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c0 = ch * 256 + lane * 4;
            if (c0 < C) {
                float gm[4];
                load4(gamma + c0, gm);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = bf2f(nd[ch][i]);
                    xh[ch][i] = (bf2f(nx[ch][i]) - mu) * rs;
                    dyw[ch][i] = d * gm[i];
                    s1 += dyw[ch][i];
                    s2 += dyw[ch][i] * xh[ch][i];
                    ag[ch][i] = fmaf(d, xh[ch][i], ag[ch][i]);
                    ab[ch][i] += d;
                }
                if (kHasRes) rv[ch] = nr[ch];
            }
        }
        if (r + NW < r1) request(r + NW);
        const float m1 = wave_sum(s1) / C, m2 = wave_sum(s2) / C;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c0 = ch * 256 + lane * 4;
            if (c0 < C) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[i] = (dyw[ch][i] - m1 - xh[ch][i] * m2) * rs;
                    if (kHasRes) o[i] += bf2f(rv[ch][i]);
                    as[ch][i] += bf2f(f2bf(o[i]));  // the sum of what is stored
                }
                store4(dx + base + c0, o);
            }
        }
    }
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        if (which > 0) __syncthreads();
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                red[wave][ch * 256 + lane * 4 + i] =
                    which == 0 ? ag[ch][i] : (which == 1 ? ab[ch][i] : as[ch][i]);
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 64 * NW) {
            float t = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
            if (NW == 8) t += (red[4][c] + red[5][c]) + (red[6][c] + red[7][c]);
            partial[which * C + c] = t;
        }
    }
}

// 64 columns per workgroup; wave w sums a quarter of the partial rows.
__global__ void __launch_bounds__(256) ln_bwd_reduce_kernel(const float* partial, bf16* dgamma,
                                                            bf16* dbeta, bf16* dxsum, int nblocks,
                                                            int C, long gs) {
    __shared__ float red[4][64];
    partial += static_cast<long>(blockIdx.y) * nblocks * 3 * C;  // blockIdx.y: group
    dgamma += blockIdx.y * gs;
    dbeta += blockIdx.y * gs;
    if (dxsum != nullptr) dxsum += blockIdx.y * gs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (i < 3 * C) {
        const int lo = (nblocks * wave) >> 2, hi = (nblocks * (wave + 1)) >> 2;
        int b = lo;
        for (; b + 1 < hi; b += 2) {
            s0 += partial[static_cast<long>(b) * 3 * C + i];
            s1 += partial[static_cast<long>(b + 1) * 3 * C + i];
        }
        if (b < hi) s0 += partial[static_cast<long>(b) * 3 * C + i];
    }
    red[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave != 0 || i >= 3 * C) return;
    const bf16 r = f2bf((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
    if (i < C) dgamma[i] = r;
    else if (i < 2 * C) dbeta[i - C] = r;
    else if (dxsum != nullptr) dxsum[i - 2 * C] = r;
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gelu_fwd_kernel(const bf16* x, bf16* y, long n8) {
    for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n8;
         i += static_cast<long>(gridDim.x) * 256) {
        const bf16x8 a = reinterpret_cast<const bf16x8*>(x)[i];
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = bf2f(a[k]);
            const float t = tanh_fast(kGeluK * (v + kGeluC * v * v * v));
            o[k] = f2bf(0.5f * v * (1.f + t));
        }
        reinterpret_cast<bf16x8*>(y)[i] = o;
    }
}

__global__ void __launch_bounds__(256) gelu_bwd_kernel(const bf16* x, const bf16* dy, bf16* dx,
                                                       long n8) {
    for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n8;
         i += static_cast<long>(gridDim.x) * 256) {
        const bf16x8 a = reinterpret_cast<const bf16x8*>(x)[i];
        const bf16x8 d = reinterpret_cast<const bf16x8*>(dy)[i];
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            o[k] = f2bf(bf2f(d[k]) * gelu_grad(bf2f(a[k])));
        }
        reinterpret_cast<bf16x8*>(dx)[i] = o;
    }
}

// ---------------------------------------------------------------------------
// Column sums.  Narrow variant (any even N): workgroup (cx, chunk) = columns
// [128*cx, 128*cx+128), rows of the chunk; a lane owns two adjacent columns,
// wave w takes every 4th row.
__global__ void __launch_bounds__(256) colsum_partial_kernel(const bf16* m, float* partial, int R,
                                                             int N, long ld, int rows_per_chunk) {
    __shared__ float red[4][128];
    m += static_cast<long>(blockIdx.z) * R * ld;  // blockIdx.z: group of R rows
    partial += static_cast<long>(blockIdx.z) * gridDim.y * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 128 + lane * 2;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(R, r0 + rows_per_chunk);
    float a0 = 0.f, a1 = 0.f;
    if (c < N) {
        for (int r = r0 + wave; r < r1; r += 4) {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(m + static_cast<long>(r) * ld + c);
            a0 += __uint_as_float(w << 16);
            a1 += __uint_as_float(w & 0xFFFF0000u);
        }
    }
    red[wave][lane * 2] = a0;
    red[wave][lane * 2 + 1] = a1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int col = blockIdx.x * 128 + threadIdx.x;
        if (col < N)
            partial[static_cast<long>(blockIdx.y) * N + col] =
                (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

// Wide variant (N and ld multiples of 8, 16-byte aligned rows): a lane owns 8
// adjacent columns, a wave 512.  kGelu: the matrix is d(act); every element is
// first turned into d(pre-activation) with the tanh-GELU derivative at x and
// written to dx, and the sums are taken of that.
template <bool kGelu>
__global__ void __launch_bounds__(256) colsum8_partial_kernel(const bf16* m, const bf16* x,
                                                              bf16* dx, float* partial, int R,
                                                              int N, long ld, int rows_per_chunk) {
    __shared__ float red[4][512];
    {   // blockIdx.z: group of R rows
        const long gofs = static_cast<long>(blockIdx.z) * R * ld;
        m += gofs;
        if (kGelu) {
            x += gofs;
            dx += gofs;
        }
        partial += static_cast<long>(blockIdx.z) * gridDim.y * N;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 512 + lane * 8;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(R, r0 + rows_per_chunk);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < N) {
#pragma unroll 4
        for (int r = r0 + wave; r < r1; r += 4) {
            const long at = static_cast<long>(r) * ld + c;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(m + at);
            if (kGelu) {
                const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + at);
                bf16x8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    o[k] = f2bf(bf2f(v[k]) * gelu_grad(bf2f(xv[k])));
                    acc[k] += bf2f(o[k]);
                }
                *reinterpret_cast<bf16x8*>(dx + at) = o;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += bf2f(v[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[wave][lane * 8 + k] = acc[k];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int col = blockIdx.x * 512 + i;
        if (col < N)
            partial[static_cast<long>(blockIdx.y) * N + col] =
                (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
}

// 64 columns per workgroup; wave w sums a quarter of the chunks.
__global__ void __launch_bounds__(256) colsum_final_kernel(const float* partial, bf16* out,
                                                           int chunks, int N, long gs) {
    __shared__ float red[4][64];
    partial += static_cast<long>(blockIdx.y) * chunks * N;  // blockIdx.y: group
    out += blockIdx.y * gs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (col < N) {
        const int lo = (chunks * wave) >> 2, hi = (chunks * (wave + 1)) >> 2;
        for (int k = lo; k < hi; ++k) s += partial[static_cast<long>(k) * N + col];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < N)
        out[col] = f2bf((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

// ---------------------------------------------------------------------------
// One workgroup per row.  Columns >= V (padding up to the row stride) are
// ignored on input and zeroed on output.
__global__ void __launch_bounds__(256) ce_fwd_bwd_kernel(bf16* logits, const long* targets,
                                                         float* loss_rows, float* correct_rows,
                                                         int V, long ld, float scale) {
    __shared__ float sm[4], ss[4];
    __shared__ int sarg[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16* row = logits + static_cast<long>(blockIdx.x) * ld;
    const long tgt = targets[blockIdx.x];
    const int nvec = static_cast<int>(ld >> 3);
    // pass 1: running maximum (with its column) and sum of exponentials
    float m = -INFINITY, s = 0.f;
    int arg = 0;
    for (int i = tid; i < nvec; i += 256) {
        const bf16x8 a = reinterpret_cast<const bf16x8*>(row)[i];
        float f[8];
        float cm = -INFINITY;
        int carg = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f[k] = (i * 8 + k < V) ? bf2f(a[k]) : -INFINITY;
            if (f[k] > cm) { cm = f[k]; carg = i * 8 + k; }
        }
        if (cm > m) {
            s *= __expf(m - cm);
            m = cm;
            arg = carg;
        }
        if (m > -INFINITY) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s += __expf(f[k] - m);
        }
    }
    // combine lanes, then waves (ties resolve to the lower column)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(m, off, 64), os = __shfl_xor(s, off, 64);
        const int oa = __shfl_xor(arg, off, 64);
        const float nm = fmaxf(m, om);
        const float mine = (m > -INFINITY) ? s * __expf(m - nm) : 0.f;
        const float theirs = (om > -INFINITY) ? os * __expf(om - nm) : 0.f;
        if (om > m || (om == m && oa < arg)) arg = oa;
        s = mine + theirs;
        m = nm;
    }
    if (lane == 0) { sm[wave] = m; ss[wave] = s; sarg[wave] = arg; }
    __syncthreads();
    float M = sm[0];
    int A = sarg[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (sm[w] > M || (sm[w] == M && sarg[w] < A)) { M = sm[w]; A = sarg[w]; }
    float S = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) S += (sm[w] > -INFINITY) ? ss[w] * __expf(sm[w] - M) : 0.f;
    const float lse = M + __logf(S);
    if (tid == 0) {
        loss_rows[blockIdx.x] = lse - bf2f(row[tgt]);
        correct_rows[blockIdx.x] = (A == tgt) ? 1.f : 0.f;
    }
    __syncthreads();  // row[tgt] is read before anybody overwrites it
    // pass 2: d(logits) in place
    for (int i = tid; i < nvec; i += 256) {
        const bf16x8 a = reinterpret_cast<const bf16x8*>(row)[i];
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int col = i * 8 + k;
            float p = (col < V) ? __expf(bf2f(a[k]) - lse) : 0.f;
            if (col == tgt) p -= 1.f;
            o[k] = f2bf(p * scale);
        }
        reinterpret_cast<bf16x8*>(row)[i] = o;
    }
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sgd_bf16g_kernel(float* p, const bf16* g, bf16* shadow,
                                                        long n8, float lr, float wd) {
    for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n8;
         i += static_cast<long>(gridDim.x) * 256) {
        const bf16x8 gv = reinterpret_cast<const bf16x8*>(g)[i];
        float4 a = reinterpret_cast<float4*>(p)[2 * i];
        float4 b = reinterpret_cast<float4*>(p)[2 * i + 1];
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        bf16x8 sv;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] -= lr * fmaf(wd, v[k], bf2f(gv[k]));
            sv[k] = f2bf(v[k]);
        }
        reinterpret_cast<float4*>(p)[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(p)[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
        reinterpret_cast<bf16x8*>(shadow)[i] = sv;
    }
}

inline int grid_cap(long items, int cap = 4096) {
    long b = (items + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return static_cast<int>(b);
}

inline bool aligned(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

}  // namespace

#define NF_NN_OK() (static_cast<int>(hipGetLastError()))

extern "C" {

int nf_ln_max_cols() { return kLnMaxChunks * 256; }

int nf_embed_fwd(const long* tokens, const void* wte, const void* wpe, void* x, int R, int T,
                 int C, hipStream_t s) {
    if (R <= 0 || T <= 0 || C <= 0 || C % 4 != 0) return -1;
    if (!aligned(wte, 8) || !aligned(wpe, 8) || !aligned(x, 8)) return -2;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, s, tokens,
                       static_cast<const bf16*>(wte), static_cast<const bf16*>(wpe),
                       static_cast<bf16*>(x), R, T, C);
    return NF_NN_OK();
}

// `sorted` / `order`: tokens in ascending order and the rows they came from.
// d(wte) is accumulated into (the caller has already put the LM-head gradient
// there when the weights are tied); d(wpe) is overwritten.
int nf_embed_bwd(const long* sorted, const long* order, const void* dx, void* dwte, void* dwpe,
                 int B, int T, int C, hipStream_t s) {
    if (B <= 0 || T <= 0 || C <= 0 || C % 4 != 0) return -1;
    if (!aligned(dx, 8) || !aligned(dwte, 8) || !aligned(dwpe, 8)) return -2;
    hipLaunchKernelGGL(embed_bwd_wte_kernel, dim3(B * T), dim3(256), 0, s, sorted, order,
                       static_cast<const bf16*>(dx), static_cast<bf16*>(dwte), B * T, C);
    hipLaunchKernelGGL(embed_bwd_wpe_kernel, dim3(T), dim3(256), 0, s,
                       static_cast<const bf16*>(dx), static_cast<bf16*>(dwpe), B, T, C);
    return NF_NN_OK();
}

// delta == nullptr: y = LN(x_in).  Otherwise x_out = x_in + delta, y = LN(x_out)
// (x_out may alias x_in).  G > 1: G groups of R rows stacked in the matrices, gamma / beta
// of group g start ps elements after those of group g - 1.
int nf_ln_fwd_g(const void* x_in, const void* delta, void* x_out, const void* gamma,
                const void* beta, void* y, float* mean, float* rstd, int G, int R, int C,
                float eps, long ps, hipStream_t s) {
    if (G < 1 || R <= 0 || C <= 0 || C % 4 != 0 || C > kLnMaxChunks * 256) return -1;
    if (!aligned(x_in, 8) || !aligned(y, 8) || !aligned(gamma, 8) || !aligned(beta, 8)) return -2;
    if (delta != nullptr && (x_out == nullptr || !aligned(delta, 8) || !aligned(x_out, 8))) return -2;
    if (G > 1 && (ps % 4 != 0 || static_cast<long>(G) * R >= (1L << 31))) return -3;
    const int RT = G * R;
    const long pstride = G > 1 ? ps : 0;
    const dim3 grid((RT + 3) / 4), block(256);
    const bf16* xi = static_cast<const bf16*>(x_in);
    const bf16* dl = static_cast<const bf16*>(delta);
    bf16* xo = static_cast<bf16*>(x_out);
    const bf16* ga = static_cast<const bf16*>(gamma);
    const bf16* be = static_cast<const bf16*>(beta);
    bf16* yo = static_cast<bf16*>(y);
#define NF_LN_FWD(N)                                                                          \
    case N:                                                                                   \
        if (delta != nullptr)                                                                 \
            hipLaunchKernelGGL((ln_fwd_kernel<true, N>), grid, block, 0, s, xi, dl, xo, ga, be, yo, \
                               mean, rstd, RT, C, eps, R, pstride);                           \
        else                                                                                  \
            hipLaunchKernelGGL((ln_fwd_kernel<false, N>), grid, block, 0, s, xi, nullptr, nullptr, \
                               ga, be, yo, mean, rstd, RT, C, eps, R, pstride);               \
        break;
    switch ((C + 255) / 256) {
        NF_LN_FWD(1) NF_LN_FWD(2) NF_LN_FWD(3) NF_LN_FWD(4)
        NF_LN_FWD(5) NF_LN_FWD(6) NF_LN_FWD(7) NF_LN_FWD(8)
    default: return -1;
    }
#undef NF_LN_FWD
    return NF_NN_OK();
}

int nf_ln_fwd(const void* x_in, const void* delta, void* x_out, const void* gamma,
              const void* beta, void* y, float* mean, float* rstd, int R, int C, float eps,
              hipStream_t s) {
    return nf_ln_fwd_g(x_in, delta, x_out, gamma, beta, y, mean, rstd, 1, R, C, eps, 0, s);
}

// row chunks of the column-sum pass
int nf_ln_bwd_blocks(int R) {
    (void)R;
    return 64;
}

// partial: G * nf_ln_bwd_blocks(R) * 3 * C floats of workspace.  dres may be null;
// dx may alias dres (not dy).  dxsum (may be null) receives the column sums of dx.
// G > 1: G groups of R rows; gamma of group g starts ps elements after that of group
// g - 1, dgamma / dbeta / dxsum of group g start gs elements after those of group g - 1.
int nf_ln_bwd_g(const void* dy, const void* x, const void* gamma, const float* mean,
                const float* rstd, const void* dres, void* dx, void* dgamma, void* dbeta,
                void* dxsum, float* partial, int G, int R, int C, long ps, long gs,
                hipStream_t s) {
    if (G < 1 || G > 65535 || R <= 0 || C <= 0 || C % 8 != 0 || C > kLnMaxChunks * 256) return -1;
    if (!aligned(dy, 16) || !aligned(x, 16) || !aligned(dx, 16) || !aligned(gamma, 8)) return -2;
    if (dres != nullptr && !aligned(dres, 8)) return -2;
    if (G > 1 && (ps % 4 != 0 || static_cast<long>(G) * R >= (1L << 31))) return -3;
    const bf16* dyp = static_cast<const bf16*>(dy);
    const bf16* xp = static_cast<const bf16*>(x);
    const bf16* ga = static_cast<const bf16*>(gamma);
    const bf16* dr = static_cast<const bf16*>(dres);
    bf16* dxp = static_cast<bf16*>(dx);
    const int RT = G * R;
    const long pstride = G > 1 ? ps : 0;
    const int nb = nf_ln_bwd_blocks(R);
    const int rows = (R + nb - 1) / nb;
    const dim3 grid(nb, G);
    (void)RT;
    // eight waves per workgroup, two workgroups per CU (twice the rows in flight) while the
    // row and its sums fit 128 registers
#define NF_LN_BWD(N, W)                                                                       \
    case N:                                                                                   \
        if (dres != nullptr)                                                                  \
            hipLaunchKernelGGL((ln_bwd_fused_kernel<true, N, W>), grid, dim3(64 * W), 0, s, dyp, \
                               xp, ga, mean, rstd, dr, dxp, partial, R, C, rows, pstride);    \
        else                                                                                  \
            hipLaunchKernelGGL((ln_bwd_fused_kernel<false, N, W>), grid, dim3(64 * W), 0, s, dyp, \
                               xp, ga, mean, rstd, nullptr, dxp, partial, R, C, rows, pstride); \
        break;
    switch ((C + 255) / 256) {
        NF_LN_BWD(1, 8) NF_LN_BWD(2, 8) NF_LN_BWD(3, 8) NF_LN_BWD(4, 4)
        NF_LN_BWD(5, 4) NF_LN_BWD(6, 4) NF_LN_BWD(7, 4) NF_LN_BWD(8, 4)
    default: return -1;
    }
#undef NF_LN_BWD
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((3 * C + 63) / 64, G), dim3(256), 0, s, partial,
                       static_cast<bf16*>(dgamma), static_cast<bf16*>(dbeta),
                       static_cast<bf16*>(dxsum), nb, C, G > 1 ? gs : 0);
    return NF_NN_OK();
}

int nf_ln_bwd(const void* dy, const void* x, const void* gamma, const float* mean,
              const float* rstd, const void* dres, void* dx, void* dgamma, void* dbeta,
              void* dxsum, float* partial, int R, int C, hipStream_t s) {
    return nf_ln_bwd_g(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, dxsum, partial, 1, R, C,
                       0, 0, s);
}

int nf_gelu_fwd(const void* x, void* y, long n, hipStream_t s) {
    if (n <= 0 || n % 8 != 0 || !aligned(x, 16) || !aligned(y, 16)) return -1;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_cap(n >> 3)), dim3(256), 0, s,
                       static_cast<const bf16*>(x), static_cast<bf16*>(y), n >> 3);
    return NF_NN_OK();
}

int nf_gelu_bwd(const void* x, const void* dy, void* dx, long n, hipStream_t s) {
    if (n <= 0 || n % 8 != 0 || !aligned(x, 16) || !aligned(dy, 16) || !aligned(dx, 16)) return -1;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_cap(n >> 3)), dim3(256), 0, s,
                       static_cast<const bf16*>(x), static_cast<const bf16*>(dy),
                       static_cast<bf16*>(dx), n >> 3);
    return NF_NN_OK();
}

int nf_colsum_chunks() { return 64; }

// out[g][c] = sum over the R rows of group g of m[r][c]; partial: G * nf_colsum_chunks() * N
// floats of workspace; out of group g starts gs elements after that of group g - 1.
int nf_colsum_g(const void* m, void* out, float* partial, int G, int R, int N, long ld, long gs,
                hipStream_t s) {
    if (G < 1 || G > 65535 || R <= 0 || N <= 0 || N % 2 != 0 || ld % 2 != 0 || ld < N ||
        !aligned(m, 4))
        return -1;
    const int chunks = nf_colsum_chunks();
    const int rows = (R + chunks - 1) / chunks;
    if (N % 8 == 0 && ld % 8 == 0 && aligned(m, 16))
        hipLaunchKernelGGL(colsum8_partial_kernel<false>, dim3((N + 511) / 512, chunks, G),
                           dim3(256), 0, s, static_cast<const bf16*>(m), nullptr, nullptr, partial,
                           R, N, ld, rows);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 127) / 128, chunks, G), dim3(256), 0, s,
                           static_cast<const bf16*>(m), partial, R, N, ld, rows);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 63) / 64, G), dim3(256), 0, s, partial,
                       static_cast<bf16*>(out), chunks, N, G > 1 ? gs : 0);
    return NF_NN_OK();
}

int nf_colsum(const void* m, void* out, float* partial, int R, int N, long ld, hipStream_t s) {
    return nf_colsum_g(m, out, partial, 1, R, N, ld, 0, s);
}

// dx = dy * gelu'(x) for contiguous [G * R][N] matrices (dx may alias dy) and
// dbias[g][c] = sum over the rows of group g of dx[r][c], in one pass.
int nf_gelu_bwd_colsum_g(const void* x, const void* dy, void* dx, void* dbias, float* partial,
                         int G, int R, int N, long gs, hipStream_t s) {
    if (G < 1 || G > 65535 || R <= 0 || N <= 0 || N % 8 != 0 || !aligned(x, 16) ||
        !aligned(dy, 16) || !aligned(dx, 16))
        return -1;
    const int chunks = nf_colsum_chunks();
    const int rows = (R + chunks - 1) / chunks;
    hipLaunchKernelGGL(colsum8_partial_kernel<true>, dim3((N + 511) / 512, chunks, G), dim3(256), 0,
                       s, static_cast<const bf16*>(dy), static_cast<const bf16*>(x),
                       static_cast<bf16*>(dx), partial, R, N, static_cast<long>(N), rows);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 63) / 64, G), dim3(256), 0, s, partial,
                       static_cast<bf16*>(dbias), chunks, N, G > 1 ? gs : 0);
    return NF_NN_OK();
}

int nf_gelu_bwd_colsum(const void* x, const void* dy, void* dx, void* dbias, float* partial, int R,
                       int N, hipStream_t s) {
    return nf_gelu_bwd_colsum_g(x, dy, dx, dbias, partial, 1, R, N, 0, s);
}

// logits: [R][ld] bf16 with V <= ld valid columns, ld a multiple of 8, base
// 16-byte aligned; targets must lie in [0, V) (checked by the caller).
int nf_ce_fwd_bwd(void* logits, const long* targets, float* loss_rows, float* correct_rows, int R,
                  int V, long ld, float scale, hipStream_t s) {
    if (R <= 0 || V <= 0 || ld < V || ld % 8 != 0 || !aligned(logits, 16)) return -1;
    hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(R), dim3(256), 0, s, static_cast<bf16*>(logits),
                       targets, loss_rows, correct_rows, V, ld, scale);
    return NF_NN_OK();
}

int nf_sgd_step_bf16g(float* p, const void* g, void* shadow, long n, float lr, float wd,
                      hipStream_t s) {
    if (n <= 0 || n % 8 != 0 || !aligned(p, 16) || !aligned(g, 16) || !aligned(shadow, 16))
        return -1;
    hipLaunchKernelGGL(sgd_bf16g_kernel, dim3(grid_cap(n >> 3)), dim3(256), 0, s, p,
                       static_cast<const bf16*>(g), static_cast<bf16*>(shadow), n >> 3, lr, wd);
    return NF_NN_OK();
}

}  // extern "C"
