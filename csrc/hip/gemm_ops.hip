// bf16 GEMM on MFMA 32x32x16 for gfx950, fp32 accumulation - the linear layers of
// the transformer trainer (the reference's linear layers are nn.Linear,
// nanofed/models/mnist.py:24,27; its step is nanofed/trainer/base.py:141-159).
//
//   C[M][N] = op(A) * op(B)   with the reduction index K
//
// A is given either "k-contiguous" ([M][K], a row of A is a run of k) or "k-major"
// ([K][M], consecutive rows are consecutive k); the same for B ([N][K] or [K][N]).
// The three layouts of a linear layer need no transposed copy of anything:
//
//   forward   y  = x  W^T      A = x  [M][K]  k-contiguous, B = W [N][K] k-contiguous
//   dgrad     dx = dy W        A = dy [M][K]  k-contiguous, B = W [K][N] k-major
//   wgrad     dW = dy^T x      A = dy [K][M]  k-major,      B = x [K][N] k-major
//
// Structure: one workgroup of 8 waves per CU and output tile.  K advances in steps
// of 64 through a ring of LDS stages that is filled by LDS-DMA (global_load_lds,
// 16 bytes per lane, no registers, no ds_write): the loads of a later K step are
// issued one piece at a time between the MFMAs of the current one, each wave waits
// only for its own loads of the step it is about to read (a counted s_waitcnt
// vmcnt) and one barrier per K step both publishes that step and frees the stage
// the next loads go into.  A DMA piece (1 KiB per wave) costs its wave about 60 issue
// cycles among bare MFMAs and two to three times that beside a burst of LDS reads
// (neither the DMA latency nor the barrier is what costs time: a build that never
// waited for either ran at the same speed, one without DMAs at 1.6-1.7 PFLOP/s), so
// where a piece is issued and the MFMA work per piece decide the speed: three tile
// geometries,
//   256 x 256 (wave 128 x 64, 4 MFMAs per piece, 2 stages)   large outputs
//   256 x 128 (wave  64 x 64, 2.7 MFMAs per piece, 3 stages)
//   128 x 128 (wave  64 x 32, 2 MFMAs per piece, 4 stages)   few output tiles
// chosen by the caller from the number of tiles each would launch.
//
// An LDS-DMA writes the 64 lanes' pieces to consecutive addresses, so a stage cannot
// be padded; bank conflicts are avoided by permuting the 16-byte pieces instead -
// each lane fetches the piece of the source that belongs at its (linear) LDS slot:
//   k-contiguous tile [128 rows][64 k] (128-byte rows): piece c of row r sits at
//     slot c ^ ((r >> 1) & 7) - the 16 lanes of a ds_read_b128 group then cover all
//     16 slots of the 256-byte bank row
//   k-major tile [64 k][128 x] (256-byte rows): the four 64-byte windows of row k
//     are permuted by k & 3, so the four k rows of a transposing ds_read_b64_tr_b16
//     hit different windows
// Edges: rows beyond M / N are clamped on load and masked on store (their products
// land in rows / columns that are never stored); a K tail shorter than 64 cannot
// be zero-filled by a DMA and is staged through registers, element-masked (the
// buffers may hold anything beyond K).  M, N, K are arbitrary; leading dimensions
// must be multiples of 8 elements.
//
// Epilogue (plain mode): + bias[n], optional second output gelu(.), results
// converted to bf16 and staged through LDS so that global stores are 16-byte
// pieces of full row segments.  Split-K mode (few output tiles, long K):
// blockIdx.y owns a K range and writes an fp32 slab; gemm_reduce_kernel sums the
// slabs in a fixed order (and applies bias / gelu) - no workgroup waits for another.
//
// Workgroup ids are dealt round-robin to the 8 XCDs (one L2 each): tile_of gives the
// ids of one XCD a contiguous range of tiles, so that neighbours in L2 share operand
// rows.

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <cstdlib>

#include "nf_device.h"

using namespace nfd;

namespace {

constexpr int BK = 64;
constexpr int kThreads = 512;             // 8 waves
constexpr int kOutRows = 128;             // rows of the output tile staged per epilogue pass

struct GemmArgs {
    const bf16* A;
    const bf16* B;
    bf16* C;          // plain mode output [M][ldc]
    bf16* C2;         // optional gelu(C) output, same shape
    const bf16* bias; // optional [N]
    float* slabs;     // split-K mode: [splits][M][N] fp32
    long lda, ldb, ldc;
    int M, N, K;
    int splits;       // 1 = plain mode
    int tiles_m, tiles_n;
    long sA, sB, sC;  // batched launches: elements between consecutive problems (blockIdx.z)
    long sBias;       // ... between their bias vectors (C2 moves with C)
    int group_m;      // tile rows that are walked together (column after column)
};

// tanh-GELU, the arithmetic of gelu_fwd_kernel in nn_ops.hip (same bits)
__device__ __forceinline__ float gelu_tanh(float v) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float e = __expf(2.f * (k0 * (v + k1 * v * v * v)));
    const float t = 1.f - 2.f / (e + 1.f);
    return 0.5f * v * (1.f + t);
}

// 16-byte piece (8 bf16) with the elements from `valid` on zeroed
__device__ __forceinline__ uint4 mask_tail(uint4 v, int valid) {
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int nv = valid - 2 * i;  // elements 2i, 2i + 1 live in word i
        const unsigned m = nv >= 2 ? 0xFFFFFFFFu : (nv == 1 ? 0x0000FFFFu : 0u);
        w[i] &= m;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// byte offset of piece `chunk` (8 k) of row `row` in a k-contiguous tile (128-byte rows)
__device__ __forceinline__ int kc_off(int row, int chunk) { return nfix::gemm_kc_off(row, chunk); }
// byte offset of x byte position `xb` of k row `k` in a k-major tile with rows of ROWB bytes
template <int ROWB>
__device__ __forceinline__ int km_off(int k, int xb) {
    return nfix::gemm_km_off(ROWB, k, xb);
}

// MFMA fragment of a k-major tile: for lane l the 8 values tile[k0 + 8 (l >> 5) + j][x0 + (l & 31)]
template <int ROWB>
__device__ __forceinline__ bf16x8 km_frag(const char* tile, int k0, int x0, int lane) {
    const int i = lane & 15;
    const int k = k0 + 8 * (lane >> 5) + (i >> 2);
    const int xb = (x0 + 16 * ((lane >> 4) & 1) + 4 * (i & 3)) * 2;
    const bf16x4 lo = lds_tr4(tile + km_off<ROWB>(k, xb));
    const bf16x4 hi = lds_tr4(tile + km_off<ROWB>(k + 4, xb));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// One LDS-DMA piece as a buffer load: the tile's first element of the K step is the
// (wave-uniform) base of the descriptor, the lane's place in the tile a 32-bit byte offset
// that does not change over the K loop, the LDS address a scalar - so a piece costs its
// wave s_mov m0 + buffer_load ... lds and no vector instruction (as a global load with a
// 64-bit address per lane it was five VALU instructions and a v_readfirstlane for M0).
__device__ __forceinline__ void dma16(const bf16* step_base, unsigned lane_bytes, char* lds_wave_base) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(step_base), 0, -1, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
        static_cast<int>(lane_bytes), 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Workgroup id -> (tile row, tile column): the id space is cut into 8 contiguous
// ranges of tile indices, one per XCD (ids congruent mod 8), tiles numbered
// column-fastest.
__device__ __forceinline__ void tile_of(int id, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
    const int rc = nfix::gemm_tile_rc(nfix::gemm_tile_index(id, tiles_m * tiles_n), tiles_m, tiles_n,
                                      group_m);
    tm = rc >> 16;
    tn = rc & 65535;
}

// Tile geometry.  BM x BN outputs per workgroup; the 8 waves form WM x (8 / WM)
// blocks of (BM / WM) x (BN / WN) outputs = RM x RN MFMA tiles per wave; STAGES LDS
// stages of one K step each.  An operand tile of X rows/columns is X / 8 DMA pieces
// of 1 KiB (64 lanes x 16 B), X / 64 per wave.
template <int BM_, int BN_, int WM_, int STAGES_>
struct Geo {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = 8 / WM_, STAGES = STAGES_;
    static constexpr int RM = BM / (32 * WM), RN = BN / (32 * WN);
    static constexpr int PA = BM / 64, PB = BN / 64;          // DMA pieces per wave and K step
    static constexpr int TILE_A = BM * 128, TILE_B = BN * 128;  // bytes (both layouts)
    static constexpr int STAGE = TILE_A + TILE_B;
    static constexpr int OUT_PITCH = BN * 2 + 16;
    static constexpr int OUT_BYTES = kOutRows * OUT_PITCH;
    static constexpr int LDS = STAGES * STAGE > OUT_BYTES ? STAGES * STAGE : OUT_BYTES;
};

template <class G, bool A_KM, bool B_KM>
__global__ void __launch_bounds__(kThreads) gemm_kernel(GemmArgs g) {
    constexpr int BM = G::BM, BN = G::BN, RM = G::RM, RN = G::RN, PA = G::PA, PB = G::PB;
    constexpr int STAGES = G::STAGES, D = STAGES - 1;  // K steps in flight
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: LDS bases stay scalar
    const int wm = wave / G::WN, wn = wave % G::WN;
    const int n = lane & 31, h = lane >> 5;
    int tm, tn;
    tile_of(blockIdx.x, g.tiles_m, g.tiles_n, g.group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    // batched launch: problem blockIdx.z of gridDim.z equally shaped ones
    g.A += blockIdx.z * g.sA;
    g.B += blockIdx.z * g.sB;
    g.C += blockIdx.z * g.sC;
    if (g.C2 != nullptr) g.C2 += blockIdx.z * g.sC;
    if (g.bias != nullptr) g.bias += blockIdx.z * g.sBias;

    // K range of this workgroup (split-K: contiguous ranges of whole K steps); only
    // the very last step of the matrix can be partial
    const int nfull = g.K / BK;
    const int ksteps = (g.K + BK - 1) / BK;
    const int per = (ksteps + g.splits - 1) / g.splits;
    const int ks0 = blockIdx.y * per;
    const int ks1 = min(ks0 + per, ksteps);
    const int kf1 = min(ks1, nfull);        // full steps [ks0, kf1) come by DMA
    const int nk = kf1 - ks0;
    const bool tail = ks1 > nfull && ks0 <= nfull;  // this workgroup owns the partial step

    // The 1 KiB pieces of each operand tile this wave fetches per K step, as the
    // source address of the lane for K step 0 and the distance between steps.
    //   k-contiguous (128-byte rows): piece q holds rows 8q .. 8q + 7; lane -> row 8q
    //     + (lane >> 3), LDS slot lane & 7, i.e. source chunk (lane & 7) ^ ((row >> 1) & 7)
    //   k-major (rows of 2 X bytes, X = 128 or 256): piece q holds 1024 / (2 X) k rows;
    //     lane -> k row and 16-byte slot of it, whose 64-byte window is permuted by k & 3
    // (kept as 32-bit element offsets from A / B: the matrices of a layer are far
    // smaller than 2^31 elements, checked by the launcher)
    int srca[PA];
    int srcb[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int q = PA * wave + j;
        if (!A_KM) {
            const int r = 8 * q + (lane >> 3);
            srca[j] = static_cast<int>(min(m0 + r, g.M - 1) * g.lda) + (((lane & 7) ^ ((r >> 1) & 7)) << 3);
        } else {
            constexpr int SPR = BM / 8;  // 16-byte slots per k row
            const int k = q * (64 / SPR) + lane / SPR, pc = lane % SPR;
            const int x = m0 + (((((pc >> 2) ^ (k & 3)) << 2) | (pc & 3)) << 3);
            srca[j] = static_cast<int>(k * g.lda) + (x < g.M ? x : 0);
        }
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int q = PB * wave + j;
        if (!B_KM) {
            const int r = 8 * q + (lane >> 3);
            srcb[j] = static_cast<int>(min(n0 + r, g.N - 1) * g.ldb) + (((lane & 7) ^ ((r >> 1) & 7)) << 3);
        } else {
            constexpr int SPR = BN / 8;
            const int k = q * (64 / SPR) + lane / SPR, pc = lane % SPR;
            const int x = n0 + (((((pc >> 2) ^ (k & 3)) << 2) | (pc & 3)) << 3);
            srcb[j] = static_cast<int>(k * g.ldb) + (x < g.N ? x : 0);
        }
    }
    const int stepa = A_KM ? static_cast<int>(BK * g.lda) : BK;
    const int stepb = B_KM ? static_cast<int>(BK * g.ldb) : BK;
    char* const dsta = smem + PA * wave * 1024;
    char* const dstb = smem + G::TILE_A + PB * wave * 1024;

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A quarter of a K step (one 16-deep MFMA group) from stage `cur`
    auto step_quarter = [&](int cur, int s) {
        const char* sa = smem + cur * G::STAGE;
        const char* sb = sa + G::TILE_A;
        bf16x8 af[RM], bfr[RN];
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            const int r0 = wm * (BM / G::WM) + i * 32;
            if (!A_KM) af[i] = *reinterpret_cast<const bf16x8*>(sa + kc_off(r0 + n, 2 * s + h));
            else af[i] = km_frag<BM * 2>(sa, s * 16, r0, lane);
        }
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int c0 = wn * (BN / G::WN) + j * 32;
            if (!B_KM) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + kc_off(c0 + n, 2 * s + h));
            else bfr[j] = km_frag<BN * 2>(sb, s * 16, c0, lane);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j)
                acc[i][j] = mfma32(af[i], bfr[j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    // Half a K step (two 16-deep groups): all fragments of the two groups are requested
    // first (their LDS latency overlaps), then multiplied.  Operands read through the
    // transposing LDS loads take this form: two instructions per fragment, more latency
    // to cover per group.
    auto step_half = [&](int cur, int half) {
        const char* sa = smem + cur * G::STAGE;
        const char* sb = sa + G::TILE_A;
        bf16x8 af[2][RM], bfr[2][RN];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = 2 * half + u;
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                const int r0 = wm * (BM / G::WM) + i * 32;
                if (!A_KM) af[u][i] = *reinterpret_cast<const bf16x8*>(sa + kc_off(r0 + n, 2 * s + h));
                else af[u][i] = km_frag<BM * 2>(sa, s * 16, r0, lane);
            }
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                const int c0 = wn * (BN / G::WN) + j * 32;
                if (!B_KM) bfr[u][j] = *reinterpret_cast<const bf16x8*>(sb + kc_off(c0 + n, 2 * s + h));
                else bfr[u][j] = km_frag<BN * 2>(sb, s * 16, c0, lane);
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    acc[i][j] = mfma32(af[u][i], bfr[u][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto step = [&](int cur) {
        step_half(cur, 0);
        step_half(cur, 1);
    };
    // one DMA piece of the K step (A pieces first, then B pieces)
    auto issue_piece = [&](int kt, int stage, int p) {
        if (p < PA)
            dma16(g.A + static_cast<long>(kt) * stepa, static_cast<unsigned>(srca[p]) * 2u,
                  dsta + stage * G::STAGE + p * 1024);
        else
            dma16(g.B + static_cast<long>(kt) * stepb, static_cast<unsigned>(srcb[p - PA]) * 2u,
                  dstb + stage * G::STAGE + (p - PA) * 1024);
    };
    auto issue_all = [&](int kt, int stage) {
#pragma unroll
        for (int p = 0; p < PA + PB; ++p) issue_piece(kt, stage, p);
    };
    // a quarter step whose share of the DMA pieces is issued between its MFMAs
    auto step_quarter_dma = [&](int cur, auto S, int kt, int stage) {
        constexpr int sq = decltype(S)::value;
        constexpr int P = PA + PB, lo = sq * P / 4, hi = (sq + 1) * P / 4, NM = RM * RN;
        const char* sa = smem + cur * G::STAGE;
        const char* sb = sa + G::TILE_A;
        bf16x8 af[RM], bfr[RN];
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            const int r0 = wm * (BM / G::WM) + i * 32;
            af[i] = *reinterpret_cast<const bf16x8*>(sa + kc_off(r0 + n, 2 * sq + h));
        }
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int c0 = wn * (BN / G::WN) + j * 32;
            bfr[j] = *reinterpret_cast<const bf16x8*>(sb + kc_off(c0 + n, 2 * sq + h));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                acc[i][j] = mfma32(af[i], bfr[j], acc[i][j]);
                const int idx = i * RN + j;
                if (idx == 0 && hi > lo) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_piece(kt, stage, lo);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (idx == NM / 2 && hi > lo + 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_piece(kt, stage, lo + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
    };

    // half a K step (two 16-deep groups) whose share of the DMA pieces is issued between
    // its MFMAs (operands read through the transposing LDS loads)
    auto step_half_dma = [&](int cur, auto H, int kt, int stage) {
        constexpr int hf = decltype(H)::value;
        constexpr int P = PA + PB, lo = hf * P / 2, hi = (hf + 1) * P / 2, NM = 2 * RM * RN;
        constexpr int NP = hi - lo;
        const char* sa = smem + cur * G::STAGE;
        const char* sb = sa + G::TILE_A;
        bf16x8 af[2][RM], bfr[2][RN];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = 2 * hf + u;
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                const int r0 = wm * (BM / G::WM) + i * 32;
                if (!A_KM) af[u][i] = *reinterpret_cast<const bf16x8*>(sa + kc_off(r0 + n, 2 * s + h));
                else af[u][i] = km_frag<BM * 2>(sa, s * 16, r0, lane);
            }
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                const int c0 = wn * (BN / G::WN) + j * 32;
                if (!B_KM) bfr[u][j] = *reinterpret_cast<const bf16x8*>(sb + kc_off(c0 + n, 2 * s + h));
                else bfr[u][j] = km_frag<BN * 2>(sb, s * 16, c0, lane);
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j) {
                    acc[i][j] = mfma32(af[u][i], bfr[u][j], acc[i][j]);
                    const int idx = (u * RM + i) * RN + j;
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        if (idx == q * NM / NP) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_piece(kt, stage, lo + q);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
        __builtin_amdgcn_s_setprio(0);
    };

    if (nk > 0) {
        // every K step is PA + PB DMAs per wave; steps past the range repeat its last
        // one into a stage nobody reads, so that the outstanding count is the same in
        // every iteration and `vmcnt((D - 1) * pieces)` means "my step t has landed"
#pragma unroll
        for (int t = 0; t < D; ++t) issue_all(min(ks0 + t, kf1 - 1), t);
        for (int t = 0; t < nk; ++t) {
            wait_vmcnt<(D - 1) * (PA + PB)>();
            __builtin_amdgcn_s_barrier();  // step t is in LDS for everybody; step t - 1 has been read
            // The DMAs of step t + D are issued between the MFMAs of step t, a piece at a
            // time: a piece costs its wave about 60 issue cycles among bare MFMAs and two to
            // three times that next to a burst of fragment reads, and the matrix pipe keeps
            // working on what was issued before.  (At 4096^3: 128 x 128 tile 163 -> 151 us,
            // 256 x 128 150 -> 136 us against blocks of DMAs issued by half of the waves
            // at a time; the loop without any DMA runs at 1.6-1.7 PFLOP/s.)
            const int kn = min(ks0 + t + D, kf1 - 1), sn = (t + D) % STAGES, cur = t % STAGES;
            if constexpr (!A_KM && !B_KM) {
                step_quarter_dma(cur, std::integral_constant<int, 0>{}, kn, sn);
                step_quarter_dma(cur, std::integral_constant<int, 1>{}, kn, sn);
                step_quarter_dma(cur, std::integral_constant<int, 2>{}, kn, sn);
                step_quarter_dma(cur, std::integral_constant<int, 3>{}, kn, sn);
            } else {
                step_half_dma(cur, std::integral_constant<int, 0>{}, kn, sn);
                step_half_dma(cur, std::integral_constant<int, 1>{}, kn, sn);
            }
        }
        wait_vmcnt<0>();
    }
    __syncthreads();  // no DMA in flight, no wave still reading: LDS can be reused

    if (tail) {
        // the partial last K step through registers: elements at k >= K are zeroed,
        // k rows >= K of a k-major tile too
        const int k0 = nfull * BK;
        char* sa = smem;
        char* sb = smem + G::TILE_A;
        for (int p = tid; p < BM * 8; p += kThreads) {  // 16-byte pieces of the A tile
            uint4 va;
            if (!A_KM) {
                const int r = p >> 3, c = p & 7;
                const int valid = g.K - (k0 + c * 8);
                va = mask_tail(*reinterpret_cast<const uint4*>(
                                   g.A + min(m0 + r, g.M - 1) * g.lda + (valid > 0 ? k0 + c * 8 : 0)),
                               valid);
                *reinterpret_cast<uint4*>(sa + kc_off(r, c)) = va;
            } else {
                constexpr int SPR = BM / 8;
                const int k = p / SPR, sl = p % SPR;
                const int x = m0 + sl * 8;
                const bool ok = k0 + k < g.K && x < g.M;
                va = *reinterpret_cast<const uint4*>(ok ? g.A + (k0 + k) * g.lda + x : g.A);
                va = mask_tail(va, ok ? 8 : 0);
                *reinterpret_cast<uint4*>(sa + km_off<BM * 2>(k, sl * 16)) = va;
            }
        }
        for (int p = tid; p < BN * 8; p += kThreads) {
            uint4 vb;
            if (!B_KM) {
                const int r = p >> 3, c = p & 7;
                const int valid = g.K - (k0 + c * 8);
                vb = mask_tail(*reinterpret_cast<const uint4*>(
                                   g.B + min(n0 + r, g.N - 1) * g.ldb + (valid > 0 ? k0 + c * 8 : 0)),
                               valid);
                *reinterpret_cast<uint4*>(sb + kc_off(r, c)) = vb;
            } else {
                constexpr int SPR = BN / 8;
                const int k = p / SPR, sl = p % SPR;
                const int x = n0 + sl * 8;
                const bool ok = k0 + k < g.K && x < g.N;
                vb = *reinterpret_cast<const uint4*>(ok ? g.B + (k0 + k) * g.ldb + x : g.B);
                vb = mask_tail(vb, ok ? 8 : 0);
                *reinterpret_cast<uint4*>(sb + km_off<BN * 2>(k, sl * 16)) = vb;
            }
        }
        __syncthreads();
        step(0);
        __syncthreads();
    }

    const int wrow = wm * (BM / G::WM), wcol = wn * (BN / G::WN);
    if (g.splits > 1) {
        // fp32 slab of this K range: row segments of 32 columns per half wave
        float* slab = g.slabs + (static_cast<long>(blockIdx.z) * g.splits + blockIdx.y) * g.M * g.N;
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                const int col = n0 + wcol + j * 32 + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wrow + i * 32 + acc_row(r, lane);
                    if (row < g.M && col < g.N)
                        slab[static_cast<long>(row) * g.N + col] = acc[i][j][r];
                }
            }
        return;
    }

    // plain mode: bias (+ gelu) in registers, 128 rows of the bf16 tile at a time
    // through LDS, 16-byte stores
    float bias[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int col = n0 + wcol + j * 32 + n;
        bias[j] = (g.bias != nullptr && col < g.N) ? bf2f(g.bias[col]) : 0.f;
    }
    // The tile is rounded to bf16 first (two values per register): half the registers of
    // the fp32 accumulators stay alive through the staging below - with the 128
    // accumulator registers of the 256 x 256 tile the staging code's temporaries used to
    // push 45 registers into scratch.
    unsigned pk[RM][RN][8];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bf16x2 two = {f2bf(acc[i][j][2 * r] + bias[j]), f2bf(acc[i][j][2 * r + 1] + bias[j])};
                unsigned word = __builtin_bit_cast(unsigned, two);
                // (an empty asm: the optimiser may not split the pair into two registers again)
                asm volatile("" : "+v"(word));
                pk[i][j][r] = word;
            }
    // (one copy of the staging code for both outputs: `which` is a run-time value)
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
        if (which == 1 && g.C2 == nullptr) break;
#pragma unroll
        for (int part = 0; part < BM / kOutRows; ++part) {
            if (which + part > 0) __syncthreads();
            if (wrow / kOutRows == part) {  // this wave's rows belong to the part being staged
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            unsigned word = pk[i][j][r >> 1];
                            // (the halves are taken apart here, inside the part's branch, and
                            // not hoisted above it into 128 registers)
                            asm volatile("" : "+v"(word));
                            bf16x2 two = __builtin_bit_cast(bf16x2, word);
                            if (which == 1) {
                                two[0] = f2bf(gelu_tanh(bf2f(two[0])));
                                two[1] = f2bf(gelu_tanh(bf2f(two[1])));
                            }
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const int row = wrow % kOutRows + i * 32 + acc_row(r + h, lane);
                                *reinterpret_cast<bf16*>(smem + row * G::OUT_PITCH +
                                                         (wcol + j * 32 + n) * 2) = two[h];
                            }
                        }
            }
            __syncthreads();
            bf16* out = which == 0 ? g.C : g.C2;
            constexpr int PPR = BN / 8;  // 16-byte pieces per row
            for (int p = tid; p < kOutRows * PPR; p += kThreads) {
                const int lr = p / PPR, pc = p % PPR;
                const int row = m0 + part * kOutRows + lr, c = n0 + pc * 8;
                // whole 16-byte pieces: a piece that starts inside N may run into the
                // padding of the row (ldc >= N rounded up to 8 is checked by the launcher)
                if (row < g.M && c < g.N)
                    *reinterpret_cast<uint4*>(out + static_cast<long>(row) * g.ldc + c) =
                        *reinterpret_cast<const uint4*>(smem + lr * G::OUT_PITCH + pc * 16);
            }
        }
    }
}

// out[m][n] (bf16, ldc) = bias[n] + sum over the slabs, in slab order; with out2 also
// out2 = gelu(out).  N a multiple of 4: float4 slab reads, 8-byte stores; other
// widths take the scalar loop.
__global__ void __launch_bounds__(256) gemm_reduce_kernel(const float* __restrict__ slabs,
                                                          bf16* __restrict__ out,
                                                          bf16* __restrict__ out2,
                                                          const bf16* __restrict__ bias, long ldc,
                                                          int M, int N, int splits, long sC,
                                                          long sBias) {
    const long total = static_cast<long>(M) * N;
    slabs += static_cast<long>(blockIdx.y) * splits * total;  // blockIdx.y: problem of a batch
    out += blockIdx.y * sC;
    if (out2 != nullptr) out2 += blockIdx.y * sC;
    if (bias != nullptr) bias += blockIdx.y * sBias;
    const long stride = static_cast<long>(gridDim.x) * blockDim.x;
    if ((N & 3) == 0) {
        const long total4 = total >> 2;
        for (long q = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; q < total4; q += stride) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < splits; ++k) {
                const float4 v = reinterpret_cast<const float4*>(slabs + k * total)[q];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const long e = q << 2;
            const long row = e / N, col = e - row * N;
            if (bias != nullptr) {
                s.x += bf2f(bias[col]); s.y += bf2f(bias[col + 1]);
                s.z += bf2f(bias[col + 2]); s.w += bf2f(bias[col + 3]);
            }
            bf16x4 o;
            o[0] = f2bf(s.x); o[1] = f2bf(s.y); o[2] = f2bf(s.z); o[3] = f2bf(s.w);
            *reinterpret_cast<bf16x4*>(out + row * ldc + col) = o;
            if (out2 != nullptr) {
                bf16x4 a;
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = f2bf(gelu_tanh(bf2f(o[i])));
                *reinterpret_cast<bf16x4*>(out2 + row * ldc + col) = a;
            }
        }
        return;
    }
    for (long e = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += slabs[k * total + e];
        const long row = e / N, col = e - row * N;
        if (bias != nullptr) s += bf2f(bias[col]);
        const bf16 o = f2bf(s);
        out[row * ldc + col] = o;
        if (out2 != nullptr) out2[row * ldc + col] = f2bf(gelu_tanh(bf2f(o)));
    }
}

using Geo256 = Geo<256, 256, 2, 2>;      // 128 x 64 per wave, 2 stages of 64 KiB
using Geo256x128 = Geo<256, 128, 4, 3>;  // 64 x 64 per wave, 3 stages of 48 KiB
using Geo128 = Geo<128, 128, 2, 4>;      // 64 x 32 per wave, 4 stages of 32 KiB
using Geo128x2 = Geo<128, 128, 2, 2>;    // the same with 2 stages: two workgroups share a CU
using Geo128x192 = Geo<128, 192, 4, 2>;  // 32 x 96 per wave, 2 stages of 40 KiB, two workgroups per CU
                                         // (k-contiguous operands only: the k-major swizzle needs
                                         // rows of a multiple of 256 bytes)

template <class G, bool A_KM, bool B_KM>
int launch(GemmArgs g, int batch, hipStream_t stream) {
    static const hipError_t lds_ok = hipFuncSetAttribute(
        reinterpret_cast<const void*>(gemm_kernel<G, A_KM, B_KM>),
        hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (lds_ok != hipSuccess) return -20;
    g.tiles_m = (g.M + G::BM - 1) / G::BM;
    g.tiles_n = (g.N + G::BN - 1) / G::BN;
    if (g.tiles_m >= 32768 || g.tiles_n >= 65536) return -8;  // (row, column) packed in one int
    hipLaunchKernelGGL((gemm_kernel<G, A_KM, B_KM>), dim3(g.tiles_m * g.tiles_n, g.splits, batch),
                       dim3(kThreads), G::LDS, stream, g);
    return static_cast<int>(hipGetLastError());
}

template <bool A_KM, bool B_KM>
int launch_geo(int geo, const GemmArgs& g, int batch, hipStream_t stream) {
    switch (geo) {
    case 0: return launch<Geo128, A_KM, B_KM>(g, batch, stream);
    case 1: return launch<Geo256x128, A_KM, B_KM>(g, batch, stream);
    case 2: return launch<Geo256, A_KM, B_KM>(g, batch, stream);
    case 3: return launch<Geo128x2, A_KM, B_KM>(g, batch, stream);
    case 4:
        if constexpr (!A_KM && !B_KM) return launch<Geo128x192, false, false>(g, batch, stream);
        else return -7;
    default: return -7;
    }
}

}  // namespace

extern "C" {

int nf_gemm_geometries() { return 5; }

// layout: 0 = NT (A [M][K], B [N][K]), 1 = NN (A [M][K], B [K][N]), 2 = TN (A [K][M], B [K][N]).
// splits > 1: `slabs` must hold batch * splits * M * N floats; the bf16 result is written to C
// by a second launch, which also applies bias / C2.  bias / C2 may be null.
// geo: workgroup tile, see nf_gemm_geometries.
// batch > 1: that many problems of one shape in one launch; problem b reads A + b sA,
// B + b sB, bias + b sBias and writes C + b sC, C2 + b sC (elements; multiples of 8).
int nf_gemm_bf16_batched(int layout, int geo, const void* A, const void* B, void* C, void* C2,
                         const void* bias, float* slabs, int M, int N, int K, long lda, long ldb,
                         long ldc, int splits, int batch, long sA, long sB, long sC, long sBias,
                         hipStream_t stream) {
    if (M < 1 || N < 1 || K < 1 || splits < 1 || splits > 64 || batch < 1 || batch > 65535) return -1;
    if ((lda & 7) || (ldb & 7) || (ldc & 7)) return -2;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) |
         reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(C2)) & 15)
        return -3;
    if (splits > 1 && slabs == nullptr) return -4;
    if (batch > 1 && ((sA | sB | sC | sBias) & 7) != 0) return -9;
    // leading dimensions must cover the rows the kernel reads (16-byte pieces)
    const long need_a = layout == 2 ? ((M + 7) & ~7) : ((K + 7) & ~7);
    const long need_b = layout == 0 ? ((K + 7) & ~7) : ((N + 7) & ~7);
    if (lda < need_a || ldb < need_b || ldc < ((N + 7) & ~7)) return -5;
    // operand elements are addressed with 32-bit offsets
    const long rows_a = layout == 2 ? K : M, rows_b = layout == 0 ? N : K;
    if (rows_a * lda >= (1L << 31) || rows_b * ldb >= (1L << 31)) return -8;
    GemmArgs g;
    g.A = static_cast<const bf16*>(A);
    g.B = static_cast<const bf16*>(B);
    g.C = static_cast<bf16*>(C);
    g.C2 = static_cast<bf16*>(C2);
    g.bias = static_cast<const bf16*>(bias);
    g.slabs = slabs;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K;
    g.splits = splits;
    g.tiles_m = g.tiles_n = 0;
    g.sA = batch > 1 ? sA : 0; g.sB = batch > 1 ? sB : 0; g.sC = batch > 1 ? sC : 0;
    g.sBias = batch > 1 ? sBias : 0;
    // Tile rows walked together: with a short reduction a 256 x 256 tile lives a few
    // microseconds and an XCD's 32 tiles in flight fetch their operand panels at about the same
    // time - four rows x eight columns share 12 panels where one row x 32 columns needs 33
    // (8 clients x 4096 x 3072 x 768: data gradient 182 -> 167 us, forward 184 -> 172 us).  Long
    // reductions and the k-major weight gradients measured no gain or a loss; the LM head's
    // forward (197 tile columns) gained 2 % with eight rows for 8 clients and lost 11 % for one.
    const char* gm = getenv("NANOFED_GEMM_GROUPM");
    if (gm != nullptr)
        g.group_m = atoi(gm);
    else if (geo == 2 && layout != 2 && K <= 1024 && M >= 2048 && N < 16384 &&
             static_cast<long>((M + 255) / 256) * ((N + 255) / 256) * batch >= 1024)
        g.group_m = 4;  // (several rounds of workgroups: with one round the order changes nothing)
    else
        g.group_m = 1;
    int rc;
    switch (layout) {
    case 0: rc = launch_geo<false, false>(geo, g, batch, stream); break;
    case 1: rc = launch_geo<false, true>(geo, g, batch, stream); break;
    case 2: rc = launch_geo<true, true>(geo, g, batch, stream); break;
    default: return -6;
    }
    if (rc) return rc;
    if (splits > 1) {
        const long total = static_cast<long>(M) * N;
        long blocks = (total / 4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3(static_cast<unsigned>(blocks), batch), dim3(256),
                           0, stream, slabs, static_cast<bf16*>(C), static_cast<bf16*>(C2),
                           static_cast<const bf16*>(bias), ldc, M, N, splits, g.sC, g.sBias);
        return static_cast<int>(hipGetLastError());
    }
    return 0;
}

int nf_gemm_bf16(int layout, int geo, const void* A, const void* B, void* C, void* C2,
                 const void* bias, float* slabs, int M, int N, int K, long lda, long ldb, long ldc,
                 int splits, hipStream_t stream) {
    return nf_gemm_bf16_batched(layout, geo, A, B, C, C2, bias, slabs, M, N, K, lda, ldb, ldc,
                                splits, 1, 0, 0, 0, 0, stream);
}

}  // extern "C"
