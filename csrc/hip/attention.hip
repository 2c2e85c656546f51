// Causal multi-head attention for gfx950, head dimension 64, bf16 operands,
// fp32 accumulation, on the packed layout a GPT-2 block produces:
//   qkv [B*T][3C]  (q | k | v, head h at columns h*64 .. h*64+63 of each third)
//   out [B*T][C]
// Flash-style: the T x T score matrix is never written; the forward pass keeps
// a running maximum and sum per query and saves the log-sum-exp, the backward
// pass recomputes the probabilities from it.
//
// All GEMMs are MFMA 32x32x16.  The score tiles are computed TRANSPOSED where
// the softmax needs per-query reductions (keys along the accumulator registers,
// one query per lane), so those reductions are register-local, and the
// probability tile is fed straight back as an MFMA operand: an accumulator
// holds, for its lane, the keys (j&3) + 4*(lane>>5) + 8*(j>>2) of each group of
// 16, and the other operand is read from LDS (ds_read_b64_tr_b16) in exactly
// that order, so P never goes through memory.
//
//   attn_fwd        one workgroup = 128 queries (4 waves x 32) of one (batch, head);
//                   K/V tiles of 64 keys double-buffered in LDS
//   attn_bwd_delta  D[q] = sum_d dO[q][d] * O[q][d]
//   attn_bwd_dkdv   one workgroup = 128 keys; streams Q / dO tiles of 64 queries
//   attn_bwd_dq     one workgroup = 128 queries; streams K / V tiles
// The two backward kernels recompute S instead of sharing dS through atomics,
// so every output element has one writer and a fixed summation order.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nf_device.h"

using namespace nfd;

namespace {

constexpr int kHd = 64;
constexpr int kRowPitch = 144;  // LDS bytes per 64-wide bf16 row read row-wise (16-byte reads)
constexpr int kTrPitch = 160;   // ... per row read with the transposing loads
constexpr int kStage = 64;      // keys (or queries) per LDS stage
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kScale = 0.125f;  // 1 / sqrt(64)
constexpr float kNegBig = -1.0e30f;

// 8 values tile[k0 + kperm(h, j)][x0 + (lane & 31)], j = 0..7, where
// kperm(h, j) = (j & 3) + 4*h + 8*(j >> 2), h = lane >> 5: the order in which
// an MFMA accumulator hands its rows to the next MFMA.
__device__ __forceinline__ bf16x8 tr_frag_perm(const char* tile, int pitch, int k0, int x0,
                                               int lane) {
    const int i = lane & 15;
    const char* a = tile + (k0 + 4 * (lane >> 5) + (i >> 2)) * pitch +
                    (x0 + 16 * ((lane >> 4) & 1) + 4 * (i & 3)) * 2;
    const bf16x4 lo = lds_tr4(a);
    const bf16x4 hi = lds_tr4(a + 8 * pitch);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 row_frag(const char* tile, int pitch, int row, int col) {
    return *reinterpret_cast<const bf16x8*>(tile + row * pitch + col * 2);
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int first) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(v[first + j]);
    return o;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// A 64 x 64 bf16 tile: rows [r0, r0+64) of a matrix with row stride ld, as two
// uint4 per thread (256 threads).
struct TileRegs {
    uint4 v[2];
};

__device__ __forceinline__ TileRegs tile_load(const bf16* base, long ld, int r0, int tid) {
    TileRegs t;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int idx = tid + 256 * p;
        t.v[p] = *reinterpret_cast<const uint4*>(base + static_cast<long>(r0 + (idx >> 3)) * ld +
                                                 (idx & 7) * 8);
    }
    return t;
}

__device__ __forceinline__ void tile_store(char* lds, int pitch, const TileRegs& t, int tid) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int idx = tid + 256 * p;
        *reinterpret_cast<uint4*>(lds + (idx >> 3) * pitch + (idx & 7) * 16) = t.v[p];
    }
}

// ---------------------------------------------------------------------------
constexpr int kFwdLds = 2 * kStage * (kRowPitch + kTrPitch);

// One 32-key x 32-query tile of the forward pass for one wave: S^T = K Q^T,
// online softmax update of (m, l), O^T += V^T P^T.  kDiag: the tile straddles
// the diagonal, so keys after the query are masked.
template <bool kDiag>
__device__ __forceinline__ void fwd_subtile(const char* Ks, const char* Vs, int sub, int key0,
                                            int q0, int lane, float sc2, const bf16x8 (&qf)[4],
                                            float& m, float& l, f32x16& o0, f32x16& o1) {
    const int n = lane & 31, hh = lane >> 5;
    f32x16 s = {0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        s = mfma32(row_frag(Ks, kRowPitch, sub * 32 + n, ks * 16 + 8 * hh), qf[ks], s);
    // raw scores; the 1/sqrt(d) * log2(e) factor is folded into the exponent
    float mx = kNegBig;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (kDiag && key0 + acc_row(r, lane) > q0 + n) s[r] = kNegBig;
        mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // the running maximum rarely moves after the first tiles: rescale the
    // accumulators only when some query of the wave needs it
    if (__any(mx > m)) {
        const float mn = fmaxf(m, mx);
        const float alpha = fast_exp2((m - mn) * sc2);
        m = mn;
        l *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }
    const float msc = m * sc2;
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(fmaf(s[r], sc2, -msc));
        s[r] = p;
        rs += p;
    }
    l += rs;
    const bf16x8 p0 = pack8(s, 0), p1 = pack8(s, 8);
    o0 = mfma32(tr_frag_perm(Vs, kTrPitch, sub * 32, 0, lane), p0, o0);
    o1 = mfma32(tr_frag_perm(Vs, kTrPitch, sub * 32, 32, lane), p0, o1);
    o0 = mfma32(tr_frag_perm(Vs, kTrPitch, sub * 32 + 16, 0, lane), p1, o0);
    o1 = mfma32(tr_frag_perm(Vs, kTrPitch, sub * 32 + 16, 32, lane), p1, o1);
}

// KG groups of four waves share one 128-query tile: group g takes the key stages
// g, g + KG, ... (each group stages its own K/V tiles), and the partial
// (max, sum, O) triples are merged through LDS at the end.  More waves per SIMD
// hide the LDS / exp / MFMA latencies of a single wave's dependent chain.
template <int KG>
__global__ void __launch_bounds__(256 * KG) __attribute__((amdgpu_waves_per_eu(4, 4)))
attn_fwd_kernel(const bf16* qkv, bf16* out, float* lse,
                                                            int T, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
    char* smem = smem_all + grp * kFwdLds;
    auto k_buf = [&](int i) { return smem + (i & 1) * kStage * kRowPitch; };
    auto v_buf = [&](int i) { return smem + 2 * kStage * kRowPitch + (i & 1) * kStage * kTrPitch; };
    // grid = (batch*head, tile): workgroups go to the XCDs round-robin by linear
    // id, so every tile of one head lands on the same XCD and its K/V stay in
    // that XCD's L2; the long tiles of all heads are dispatched first
    const int qt = gridDim.y - 1 - blockIdx.y;
    const int bh = blockIdx.x;
    const int b = bh / H, h = bh % H;
    const int lane = tid & 63, wave = tid >> 6, n = lane & 31, hh = lane >> 5;
    const int C = H * kHd;
    const long ld = 3L * C;
    const bf16* qb = qkv + static_cast<long>(b) * T * ld + h * kHd;
    const bf16* kb = qb + C;
    const bf16* vb = qb + 2 * C;
    const int q0 = qt * 128 + wave * 32;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8*>(qb + static_cast<long>(q0 + n) * ld + ks * 16 + 8 * hh);
    f32x16 o0 = {0}, o1 = {0};
    float m = kNegBig, l = 0.f;
    const float sc2 = kScale * kLog2e;
    const int stages = 2 * (qt + 1);
    const int iters = (stages + KG - 1) / KG;  // the same for every group: barriers stay aligned

    TileRegs kr, vr;
    if (grp < stages) {
        kr = tile_load(kb, ld, grp * kStage, tid);
        vr = tile_load(vb, ld, grp * kStage, tid);
        tile_store(k_buf(0), kRowPitch, kr, tid);
        tile_store(v_buf(0), kTrPitch, vr, tid);
    }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        const int st = it * KG + grp;
        const bool next = st + KG < stages;
        const char* Ks = k_buf(it);
        const char* Vs = v_buf(it);
        if (next) {
            kr = tile_load(kb, ld, (st + KG) * kStage, tid);
            vr = tile_load(vb, ld, (st + KG) * kStage, tid);
        }
        if (st < stages) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int key0 = st * kStage + sub * 32;
                if (key0 > q0 + 31) continue;  // wave-uniform: nothing visible in this tile
                // only the tiles on the diagonal need the per-element causal test
                if (key0 + 31 > q0)
                    fwd_subtile<true>(Ks, Vs, sub, key0, q0, lane, sc2, qf, m, l, o0, o1);
                else
                    fwd_subtile<false>(Ks, Vs, sub, key0, q0, lane, sc2, qf, m, l, o0, o1);
            }
        }
        if (next) {
            tile_store(k_buf(it + 1), kRowPitch, kr, tid);
            tile_store(v_buf(it + 1), kTrPitch, vr, tid);
        }
        __syncthreads();
    }
    // merge the groups pairwise: the upper half of the live groups parks its
    // (m, l, O) in its own LDS region, the lower half folds it in
    for (int half = KG / 2; half >= 1; half >>= 1) {
        if (grp >= half && grp < 2 * half) {
            float* park = reinterpret_cast<float*>(smem);
            park[0 * 256 + tid] = m;
            park[1 * 256 + tid] = l;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                park[(2 + r) * 256 + tid] = o0[r];
                park[(18 + r) * 256 + tid] = o1[r];
            }
        }
        __syncthreads();
        if (grp < half) {
            const float* park = reinterpret_cast<const float*>(smem_all + (grp + half) * kFwdLds);
            const float m2 = park[0 * 256 + tid], l2 = park[1 * 256 + tid];
            const float mn = fmaxf(m, m2);
            const float a1 = fast_exp2((m - mn) * sc2), a2 = fast_exp2((m2 - mn) * sc2);
            m = mn;
            l = l * a1 + l2 * a2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o0[r] = o0[r] * a1 + park[(2 + r) * 256 + tid] * a2;
                o1[r] = o1[r] * a1 + park[(18 + r) * 256 + tid] * a2;
            }
        }
        __syncthreads();
    }
    if (grp != 0) return;
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    bf16* orow = out + (static_cast<long>(b) * T + q0 + n) * C + h * kHd;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bf16x4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = f2bf(o0[4 * g + e] * inv);
            c[e] = f2bf(o1[4 * g + e] * inv);
        }
        *reinterpret_cast<bf16x4*>(orow + 8 * g + 4 * hh) = a;
        *reinterpret_cast<bf16x4*>(orow + 32 + 8 * g + 4 * hh) = c;
    }
    if (hh == 0)
        lse[(static_cast<long>(bh)) * T + q0 + n] = m * kScale + __logf(l);
}

// ---------------------------------------------------------------------------
// Eight lanes per (row, head), 8 values each: delta = sum_d dO * O.
__global__ void __launch_bounds__(256) attn_bwd_delta_kernel(const bf16* out, const bf16* dout,
                                                             float* delta, int B, int T, int H) {
    const long task = static_cast<long>(blockIdx.x) * 32 + (threadIdx.x >> 3);
    const long total = static_cast<long>(B) * T * H;
    const int part = threadIdx.x & 7;
    float v = 0.f;
    long row = 0;
    int h = 0;
    if (task < total) {
        row = task / H;
        h = static_cast<int>(task % H);
        const long at = row * H * kHd + h * kHd + part * 8;
        const bf16x8 o = *reinterpret_cast<const bf16x8*>(out + at);
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(dout + at);
#pragma unroll
        for (int k = 0; k < 8; ++k) v += bf2f(o[k]) * bf2f(d[k]);
    }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    if (task < total && part == 0) {
        const long b = row / T, t = row % T;
        delta[(b * H + h) * T + t] = v;
    }
}

// P = exp2(S * sc2 - lse2) and dS = P * (dP - delta) * scale for one accumulator
// tile, in place (s <- P, dp <- dS).  kRows: the per-query values (lse2, delta)
// vary along the accumulator rows and are read from LDS (dK/dV kernel); otherwise
// the lane owns one query and passes them as the first elements.  kDiag: apply
// the causal mask (query index < key index -> 0).
template <bool kDiag, bool kRows>
__device__ __forceinline__ void bwd_probs(f32x16& s, f32x16& dp, const float* lse2,
                                          const float* del, float sc2, int row0, int col, int hh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float lv[4], dv[4];
        if (kRows) {
            const float4 l4 = *reinterpret_cast<const float4*>(lse2 + 8 * g + 4 * hh);
            const float4 d4 = *reinterpret_cast<const float4*>(del + 8 * g + 4 * hh);
            lv[0] = l4.x; lv[1] = l4.y; lv[2] = l4.z; lv[3] = l4.w;
            dv[0] = d4.x; dv[1] = d4.y; dv[2] = d4.z; dv[3] = d4.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { lv[e] = lse2[0]; dv[e] = del[0]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            float p = fast_exp2(fmaf(s[r], sc2, -lv[e]));
            // kRows: rows are queries, the lane's column a key; else rows are keys
            if (kDiag) {
                const int rr = row0 + 8 * g + 4 * hh + e;
                if (kRows ? (rr < col) : (rr > col)) p = 0.f;
            }
            s[r] = p;
            dp[r] = p * (dp[r] - dv[e]) * kScale;
        }
    }
}

// ---------------------------------------------------------------------------
// LDS per stage: Q and dO tiles, each in both pitches (row reads feed S and dP,
// transposing reads feed dK and dV), plus lse and delta of the 64 queries.
constexpr int kTileBoth = kStage * (kRowPitch + kTrPitch);
constexpr int kDkvStage = 2 * kTileBoth + 2 * kStage * 4;
constexpr int kDkvLds = 2 * kDkvStage;

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
attn_bwd_dkdv_kernel(const bf16* qkv, const bf16* dout,
                                                            const float* lse, const float* delta,
                                                            bf16* dqkv, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int kt = blockIdx.y;  // key tile 0 sees every query: longest first
    const int bh = blockIdx.x;  // all tiles of a head on one XCD (see attn_fwd_kernel)
    const int b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hh = lane >> 5;
    const int C = H * kHd;
    const long ld = 3L * C;
    const bf16* qb = qkv + static_cast<long>(b) * T * ld + h * kHd;
    const bf16* dob = dout + static_cast<long>(b) * T * C + h * kHd;
    const float* lseb = lse + static_cast<long>(bh) * T;
    const float* delb = delta + static_cast<long>(bh) * T;
    const int key0 = kt * 128 + wave * 32;
    // K and V rows of this wave's keys: B operands of S = Q K^T and dP = dO V^T
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *reinterpret_cast<const bf16x8*>(qb + C + static_cast<long>(key0 + n) * ld + ks * 16 + 8 * hh);
        vf[ks] = *reinterpret_cast<const bf16x8*>(qb + 2 * C + static_cast<long>(key0 + n) * ld + ks * 16 + 8 * hh);
    }
    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    const float sc2 = kScale * kLog2e;
    const int first = kt * 2, stages = T / kStage;

    auto stage_ptr = [&](int st) { return smem + (st & 1) * kDkvStage; };
    auto fill = [&](char* base, const TileRegs& q, const TileRegs& d, float ls, float de) {
        tile_store(base, kRowPitch, q, tid);
        tile_store(base + kStage * kRowPitch, kTrPitch, q, tid);
        tile_store(base + kTileBoth, kRowPitch, d, tid);
        tile_store(base + kTileBoth + kStage * kRowPitch, kTrPitch, d, tid);
        if (tid < kStage) {
            reinterpret_cast<float*>(base + 2 * kTileBoth)[tid] = ls * kLog2e;
            reinterpret_cast<float*>(base + 2 * kTileBoth + kStage * 4)[tid] = de;
        }
    };
    TileRegs qr = tile_load(qb, ld, first * kStage, tid);
    TileRegs dr = tile_load(dob, C, first * kStage, tid);
    float ls = 0.f, de = 0.f;
    if (tid < kStage) { ls = lseb[first * kStage + tid]; de = delb[first * kStage + tid]; }
    fill(stage_ptr(first), qr, dr, ls, de);
    __syncthreads();
    for (int st = first; st < stages; ++st) {
        const char* base = stage_ptr(st);
        const char* Qrow = base;
        const char* Qtr = base + kStage * kRowPitch;
        const char* Drow = base + kTileBoth;
        const char* Dtr = base + kTileBoth + kStage * kRowPitch;
        const float* lse2 = reinterpret_cast<const float*>(base + 2 * kTileBoth);
        const float* del = lse2 + kStage;
        if (st + 1 < stages) {
            qr = tile_load(qb, ld, (st + 1) * kStage, tid);
            dr = tile_load(dob, C, (st + 1) * kStage, tid);
            if (tid < kStage) { ls = lseb[(st + 1) * kStage + tid]; de = delb[(st + 1) * kStage + tid]; }
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int qs0 = st * kStage + sub * 32;
            if (qs0 + 31 < key0) continue;  // every query of the tile precedes every key
            f32x16 s = {0}, dp = {0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma32(row_frag(Qrow, kRowPitch, sub * 32 + n, ks * 16 + 8 * hh), kf[ks], s);
                dp = mfma32(row_frag(Drow, kRowPitch, sub * 32 + n, ks * 16 + 8 * hh), vf[ks], dp);
            }
            // s[r], dp[r]: query qs0 + acc_row(r), key key0 + n
            if (qs0 < key0 + 31)
                bwd_probs<true, true>(s, dp, lse2 + sub * 32, del + sub * 32, sc2, qs0, key0 + n, hh);
            else
                bwd_probs<false, true>(s, dp, lse2 + sub * 32, del + sub * 32, sc2, qs0, key0 + n, hh);
            const bf16x8 p0 = pack8(s, 0), p1 = pack8(s, 8);
            const bf16x8 s0 = pack8(dp, 0), s1 = pack8(dp, 8);
            dv0 = mfma32(p0, tr_frag_perm(Dtr, kTrPitch, sub * 32, 0, lane), dv0);
            dv1 = mfma32(p0, tr_frag_perm(Dtr, kTrPitch, sub * 32, 32, lane), dv1);
            dv0 = mfma32(p1, tr_frag_perm(Dtr, kTrPitch, sub * 32 + 16, 0, lane), dv0);
            dv1 = mfma32(p1, tr_frag_perm(Dtr, kTrPitch, sub * 32 + 16, 32, lane), dv1);
            dk0 = mfma32(s0, tr_frag_perm(Qtr, kTrPitch, sub * 32, 0, lane), dk0);
            dk1 = mfma32(s0, tr_frag_perm(Qtr, kTrPitch, sub * 32, 32, lane), dk1);
            dk0 = mfma32(s1, tr_frag_perm(Qtr, kTrPitch, sub * 32 + 16, 0, lane), dk0);
            dk1 = mfma32(s1, tr_frag_perm(Qtr, kTrPitch, sub * 32 + 16, 32, lane), dk1);
        }
        if (st + 1 < stages) fill(stage_ptr(st + 1), qr, dr, ls, de);
        __syncthreads();
    }
    // dk/dv[r]: key key0 + acc_row(r), column n (+32)
    bf16* dkb = dqkv + static_cast<long>(b) * T * ld + C + h * kHd;
    bf16* dvb = dkb + C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long off = static_cast<long>(key0 + acc_row(r, lane)) * ld + n;
        dkb[off] = f2bf(dk0[r]);
        dkb[off + 32] = f2bf(dk1[r]);
        dvb[off] = f2bf(dv0[r]);
        dvb[off + 32] = f2bf(dv1[r]);
    }
}

// ---------------------------------------------------------------------------
// LDS per stage: K in both pitches (row reads for S^T, transposing reads for
// dQ), V row-wise.
constexpr int kDqStage = kTileBoth + kStage * kRowPitch;
constexpr int kDqLds = 2 * kDqStage;

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
attn_bwd_dq_kernel(const bf16* qkv, const bf16* dout,
                                                          const float* lse, const float* delta,
                                                          bf16* dqkv, int T, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int qt = gridDim.y - 1 - blockIdx.y;
    const int bh = blockIdx.x;  // all tiles of a head on one XCD (see attn_fwd_kernel)
    const int b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hh = lane >> 5;
    const int C = H * kHd;
    const long ld = 3L * C;
    const bf16* qb = qkv + static_cast<long>(b) * T * ld + h * kHd;
    const bf16* kb = qb + C;
    const bf16* vb = qb + 2 * C;
    const int q0 = qt * 128 + wave * 32;
    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = *reinterpret_cast<const bf16x8*>(qb + static_cast<long>(q0 + n) * ld + ks * 16 + 8 * hh);
        dof[ks] = *reinterpret_cast<const bf16x8*>(
            dout + (static_cast<long>(b) * T + q0 + n) * C + h * kHd + ks * 16 + 8 * hh);
    }
    const float lse2 = lse[static_cast<long>(bh) * T + q0 + n] * kLog2e;
    const float del = delta[static_cast<long>(bh) * T + q0 + n];
    f32x16 dq0 = {0}, dq1 = {0};
    const float sc2 = kScale * kLog2e;
    const int stages = 2 * (qt + 1);

    auto fill = [&](char* base, const TileRegs& k, const TileRegs& v) {
        tile_store(base, kRowPitch, k, tid);
        tile_store(base + kStage * kRowPitch, kTrPitch, k, tid);
        tile_store(base + kTileBoth, kRowPitch, v, tid);
    };
    TileRegs kr = tile_load(kb, ld, 0, tid), vr = tile_load(vb, ld, 0, tid);
    fill(smem, kr, vr);
    __syncthreads();
    for (int st = 0; st < stages; ++st) {
        const char* base = smem + (st & 1) * kDqStage;
        const char* Krow = base;
        const char* Ktr = base + kStage * kRowPitch;
        const char* Vrow = base + kTileBoth;
        if (st + 1 < stages) {
            kr = tile_load(kb, ld, (st + 1) * kStage, tid);
            vr = tile_load(vb, ld, (st + 1) * kStage, tid);
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int key0 = st * kStage + sub * 32;
            if (key0 > q0 + 31) continue;
            f32x16 s = {0}, dp = {0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma32(row_frag(Krow, kRowPitch, sub * 32 + n, ks * 16 + 8 * hh), qf[ks], s);
                dp = mfma32(row_frag(Vrow, kRowPitch, sub * 32 + n, ks * 16 + 8 * hh), dof[ks], dp);
            }
            // s[r], dp[r]: key key0 + acc_row(r), query q0 + n
            if (key0 + 31 > q0)
                bwd_probs<true, false>(s, dp, &lse2, &del, sc2, key0, q0 + n, hh);
            else
                bwd_probs<false, false>(s, dp, &lse2, &del, sc2, key0, q0 + n, hh);
            const bf16x8 s0 = pack8(dp, 0), s1 = pack8(dp, 8);
            dq0 = mfma32(tr_frag_perm(Ktr, kTrPitch, sub * 32, 0, lane), s0, dq0);
            dq1 = mfma32(tr_frag_perm(Ktr, kTrPitch, sub * 32, 32, lane), s0, dq1);
            dq0 = mfma32(tr_frag_perm(Ktr, kTrPitch, sub * 32 + 16, 0, lane), s1, dq0);
            dq1 = mfma32(tr_frag_perm(Ktr, kTrPitch, sub * 32 + 16, 32, lane), s1, dq1);
        }
        if (st + 1 < stages) fill(smem + ((st + 1) & 1) * kDqStage, kr, vr);
        __syncthreads();
    }
    // dq[r]: column (d) acc_row(r) (+32), query q0 + n
    bf16* drow = dqkv + (static_cast<long>(b) * T + q0 + n) * ld + h * kHd;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bf16x4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = f2bf(dq0[4 * g + e]);
            c[e] = f2bf(dq1[4 * g + e]);
        }
        *reinterpret_cast<bf16x4*>(drow + 8 * g + 4 * hh) = a;
        *reinterpret_cast<bf16x4*>(drow + 32 + 8 * g + 4 * hh) = c;
    }
}

inline bool ok16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int set_lds(const void* fn, int bytes) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess
               ? 0
               : -20;
}

template <int KG>
int launch_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, hipStream_t s) {
    static const int attr = set_lds(reinterpret_cast<const void*>(attn_fwd_kernel<KG>), KG * kFwdLds);
    if (attr) return attr;
    hipLaunchKernelGGL(attn_fwd_kernel<KG>, dim3(B * H, T / 128), dim3(256 * KG), KG * kFwdLds, s,
                       static_cast<const bf16*>(qkv), static_cast<bf16*>(out), lse, T, H);
    return static_cast<int>(hipGetLastError());
}

}  // namespace

extern "C" {

// Sequence lengths the kernels accept: a multiple of the 128-row workgroup tile.
int nf_attn_seq_multiple() { return 128; }

// Wave groups per 128-query tile (1, 2 or 4) for the forward kernel; tuning hook.
static int g_fwd_kg = 2;
void nf_attn_set_split(int fwd_kg) {
    if (fwd_kg == 1 || fwd_kg == 2 || fwd_kg == 4) g_fwd_kg = fwd_kg;
}

int nf_attn_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, hipStream_t s) {
    if (B <= 0 || H <= 0 || T <= 0 || T % 128 != 0) return -1;
    if (!ok16(qkv) || !ok16(out)) return -2;
    if (g_fwd_kg == 4) return launch_fwd<4>(qkv, out, lse, B, T, H, s);
    if (g_fwd_kg == 2) return launch_fwd<2>(qkv, out, lse, B, T, H, s);
    return launch_fwd<1>(qkv, out, lse, B, T, H, s);
}

// delta: B*H*T floats of scratch.  dqkv is fully overwritten.
int nf_attn_bwd(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv,
                float* delta, int B, int T, int H, hipStream_t s) {
    if (B <= 0 || H <= 0 || T <= 0 || T % 128 != 0) return -1;
    if (!ok16(qkv) || !ok16(out) || !ok16(dout) || !ok16(dqkv)) return -2;
    static const int attr1 = set_lds(reinterpret_cast<const void*>(attn_bwd_dkdv_kernel), kDkvLds);
    static const int attr2 = set_lds(reinterpret_cast<const void*>(attn_bwd_dq_kernel), kDqLds);
    if (attr1 || attr2) return -20;
    const long tasks = static_cast<long>(B) * T * H;
    hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3(static_cast<unsigned>((tasks + 31) / 32)), dim3(256),
                       0, s, static_cast<const bf16*>(out), static_cast<const bf16*>(dout), delta,
                       B, T, H);
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, dim3(B * H, T / 128), dim3(256), kDkvLds, s,
                       static_cast<const bf16*>(qkv), static_cast<const bf16*>(dout), lse, delta,
                       static_cast<bf16*>(dqkv), T, H);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(B * H, T / 128), dim3(256), kDqLds, s,
                       static_cast<const bf16*>(qkv), static_cast<const bf16*>(dout), lse, delta,
                       static_cast<bf16*>(dqkv), T, H);
    return static_cast<int>(hipGetLastError());
}

}  // extern "C"
