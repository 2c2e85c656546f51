// Hand-written gfx950 training step of the MNIST CNN
//   conv1 1->32 k3, ReLU, conv2 32->64 k3, ReLU, maxpool2, dropout(.25),
//   fc1 9216->128, ReLU, dropout(.5), fc2 128->10, log-softmax + NLL
// (reference model: nanofed/models/mnist.py:6-28; loss: nanofed/trainer/torch.py:14;
//  step: nanofed/trainer/base.py:141-159) for the G simulated clients that share
// one GPU, all of them in every launch (blockIdx.y = client).
//
// One optimisation step is six to eight launches, none of which synchronises with the
// host; dropout offsets come from a device-side step counter, batch rows from
// a device index table, so a whole epoch can be recorded in one hipGraph.
//
//   conv1_fwd      direct 3x3 conv on the uint8 dataset rows (gather + normalise
//                  fused), ReLU, bf16 NHWC out
//   conv2_fwd      implicit GEMM on MFMA 32x32x16 bf16, weights stationary in
//                  registers, activations staged through LDS; bias + ReLU +
//                  2x2 max-pool + dropout in the epilogue (pooling is register
//                  local thanks to a window-major row order); writes pooled
//                  activations and a 1-byte argmax/alive code
//   fc1_fwd        split-K MFMA GEMM, operands streamed straight to registers,
//                  fp32 partial slabs (deterministic reduction)
//   head           slab reduce + bias + ReLU + dropout, fc2, log-softmax, NLL,
//                  accuracy, d(logits), fc2 backward, d(h1)  - one launch
//   fc1_dgrad      dp2 = dh1 W1 (W1 tile transposed by LDS tr-reads); dropout and
//                  ReLU backward in the epilogue; writes the pooled gradient
//                  (the 4x larger d(conv2 out) is never materialised: its two
//                  consumers expand it from dp2 + the arg-max code while
//                  filling LDS); also reduces conv2's bias gradient
//   fc1_wgrad      dW1 = dh1^T p2, both operands transposed on the fly with
//                  ds_read_b64_tr_b16; optionally applies the SGD update of
//                  fc1.weight (98 % of the parameters) in its epilogue
//   conv2_dgrad    implicit GEMM (K = 9 taps x 64), zero-halo LDS patch with XOR
//                  swizzle; epilogue applies conv1's ReLU mask (bit words) and
//                  feeds the masked accumulator tile straight back into MFMA as
//                  the A operand of conv1's weight gradient (im2col of the
//                  input as B), so d(a1) never reaches memory
//   conv2_wgrad    pixels are the reduction dimension of both operands:
//                  tr-reads for both; register prefetch of the next band;
//                  fp32 partial tiles per workgroup
//   finalize       fixed-order sum of all partial slabs into the gradient arena,
//                  loss/accuracy accumulation; when fc1_wgrad already applied
//                  fc1.weight's update it also applies SGD to every other
//                  parameter and ends the step
//   sgd_pack       SGD on the flat arena + bf16 shadow + transposed conv2 shadow
//                  (unfused steps, refresh of the copies after a broadcast)
//
// Storage ("native") layout of the parameters: conv2.weight [oc][kh][kw][c],
// fc1.weight [j][(ph*12+pw)*64 + c]; everything else as in PyTorch.
// Batch size is 64 (two 32-row MFMA blocks).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#include "../common/philox.h"
#include "nf_device.h"

using namespace nfd;

extern "C" {

struct MnistCtx {
    int G, B, S, WPC;
    int train;            // 1: dropout active
    int client_base;      // global id of client 0 (dropout streams)
    int fuse_sgd;         // 1: fc1_wgrad applies fc1.weight's SGD update itself
    int dp_pass;          // 0: plain SGD; 1: norm pass of DP-SGD; 2: clipped (re-weighting) pass
    int split;            // conv2 fwd/dgrad workgroups per image: 1 (whole image) or 3 (one band)
    int conv1_split;      // conv1 workgroups per image: 1, 2 or 4
    const uint8_t* images;  // [N][784]
    const int32_t* labels;  // [N]
    const int32_t* idx;     // [G][idx_stride] dataset rows of every step
    long idx_stride;
    const int32_t* nvalid;  // [G][nvalid_stride] valid samples per step
    long nvalid_stride;
    float* arena;           // [G][arena_stride] fp32 master parameters
    long arena_stride;
    float* grad;            // [G][grad_stride]
    long grad_stride;
    bf16* shadow;           // [G][shadow_stride] bf16 copy of the arena
    long shadow_stride;
    bf16* w2t;              // [G][w2t_stride] conv2 weights as [tap][c][oc]
    long w2t_stride;
    long n_arena;
    int off_c1w, off_c1b, off_c2w, off_c2b, off_f1w, off_f1b, off_f2w, off_f2b;
    bf16* a1;               // [G][B][676][32]
    bf16* p2;               // [G][B][144][64]
    uint8_t* pmask;         // [G][B][144][64] bits 0-1 argmax, bit 2 alive
    float* fc1_slabs;       // [G][S][B][128]
    bf16* dh1;              // [G][B][128]
    bf16* dp2;              // [G][B][144][64] gradient of the pooled activations
    uint32_t* a1mask;       // [G][B][676] bit c = conv1 output (pixel, c) > 0
    float* conv1_part;      // [G][B*split][320]
    float* conv2_part;      // [G][WPC][18432]
    float* c2b_part;        // [G][144][64] conv2.bias partial sums: one row per pooled pixel, over the batch
    float* head_part;       // [G][B/8][1440]
    float* metrics;         // [G][4] loss sum, accuracy sum, batches, -
    float* eval_out;        // [G][B/8][2] loss sum, correct count per workgroup of the last step (eval mode)
    unsigned long long* step_counter;  // [G] optimisation steps taken by each client
    unsigned long long seed;
    float lr, wd, mean, inv_std, p_drop1, p_drop2;
    float dp_clip, dp_noise_std;  // per-sample L2 bound C, noise std on the SUM (sigma * C)
    float* dp_normsq;       // [G][B][4] squared per-sample gradient norms: fc2, conv1, conv2.weight; the
                            // fourth word of samples 0 / 1: low / high half of the step counter as
                            // dp_scale saw it (the clipped pass keys its noise by it)
    float* dp_rowscale;     // [G][B] clip factor of every sample
    float* dp_c2b;          // [G][B][64] conv2.bias gradient of every sample (DP-SGD)
    int ragged;             // 1: some client runs out of batches before the others
    int cur_step;           // step of the launch (filled in by nf_mnist_stage)
    // Batches larger than B: an optimisation step is `micro` consecutive table steps
    // (micro-batches of up to B samples) whose gradients are added up in `grad` before
    // one SGD update; ntotal holds, per table step, the valid samples of the whole batch
    // (the divisor of the mean loss).  micro == 1 and ntotal == nullptr: a table step is
    // an optimisation step.
    int micro;
    int accumulate;         // this launch adds to `grad` instead of overwriting it (filled in per micro-batch)
    const int32_t* ntotal;  // [G][nvalid_stride] or nullptr
    int bwd_triple;         // 1: conv2_wgrad shares the launch of conv2_dgrad + fc1_wgrad
    // non-null: the images are rows of normalised floats [N][784] (a batch handed over as a
    // tensor, e.g. by PrivateTrainer.train_batch) instead of the uint8 dataset
    const float* images_f32;
};

}  // extern "C"

namespace {

constexpr int kA1Pix = 676, kA1C = 32, kP2Pix = 144, kC2 = 64, kFc1K = 9216;
constexpr int kHeadPart = 1440, kConv2Part = 18432, kConv1Part = 320, kHeadRows = 8;
constexpr int kFc1Tiles = 72;

// With shards of unequal size a client can be out of batches while others still
// train: every kernel of the step then skips that client's workgroups (the test is
// on a kernel argument first, so equal shards pay nothing).
__device__ __forceinline__ bool client_idle(const MnistCtx& c, int g) {
    return c.ragged != 0 && c.nvalid[g * c.nvalid_stride + c.cur_step] == 0;
}

__device__ __forceinline__ unsigned long long client_seed(const MnistCtx& c, int g) {
    return c.seed + 0x9E3779B97F4A7C15ull * static_cast<unsigned long long>(c.client_base + g + 1);
}

// The normalised input image of dataset row `row` into xs[0..783] (threads 0..195, four
// pixels each): from the uint8 dataset (normalisation fused into the load) or, when the
// batch came as a float tensor, as it is.
__device__ __forceinline__ void load_image(const MnistCtx& c, int row, int tid, float* xs) {
    if (tid >= 196) return;
    if (c.images_f32 != nullptr) {
        const float4 v = reinterpret_cast<const float4*>(c.images_f32 + static_cast<long>(row) * 784)[tid];
        xs[4 * tid + 0] = v.x; xs[4 * tid + 1] = v.y; xs[4 * tid + 2] = v.z; xs[4 * tid + 3] = v.w;
        return;
    }
    const uchar4 v = reinterpret_cast<const uchar4*>(c.images + static_cast<long>(row) * 784)[tid];
    const float k = 1.0f / 255.0f;
    xs[4 * tid + 0] = (v.x * k - c.mean) * c.inv_std;
    xs[4 * tid + 1] = (v.y * k - c.mean) * c.inv_std;
    xs[4 * tid + 2] = (v.z * k - c.mean) * c.inv_std;
    xs[4 * tid + 3] = (v.w * k - c.mean) * c.inv_std;
}

// ---------------------------------------------------------------------------
// conv1 as a matrix product on the fp32 matrix cores: D[channel][pixel] = W[channel][k] *
// X[k][pixel] with k = 9 taps + 1 bias row (X = 1), five v_mfma_f32_32x32x2_f32 per block
// of 32 pixels - exact fp32 products, as the scalar loop this replaces (72 FMAs per pixel
// and channel group, most of the kernel's instructions).  With the channels as rows a lane
// ends up with four runs of 4 consecutive channels of ONE pixel; a half-swap between the
// lanes of a pixel (v_permlane32_swap) turns them into two 16-byte stores.
//
// ReLU mask word of a pixel: the bit of channel c is conv1_mask_bit(c) - the order in which
// the accumulator registers of the two lane halves hold the channels.
__device__ __forceinline__ int conv1_mask_bit(int c) {
    return (c & 3) + 4 * (c >> 3) + 16 * ((c >> 2) & 1);
}

// (at least two waves per SIMD: with a budget of 256 registers the compiler keeps the
// matrix results in VGPRs, where the epilogue reads them without v_accvgpr_read copies)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
conv1_fwd_kernel(MnistCtx c, int idx_off) {
    if (client_idle(c, blockIdx.y)) return;
    const int b = blockIdx.x / c.conv1_split, part = blockIdx.x % c.conv1_split;
    const int g = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
    __shared__ float xs[784 + 4];  // the image, then the 1.0 that multiplies the bias
    load_image(c, c.idx[g * c.idx_stride + idx_off + b], tid, xs);
    if (tid == 196) xs[784] = 1.0f;
    // A operand: row = channel n, k = 2 s + h.  k = 0..8 the taps, k = 9 the bias.
    const float* wa = c.arena + g * c.arena_stride;
    float wk[5];
    int xoff[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int k = 2 * s + h;
        wk[s] = k < 9 ? wa[c.off_c1w + n * 9 + k] : wa[c.off_c1b + n];
        xoff[s] = (k / 3) * 28 + k % 3;  // (k = 9 is replaced below)
    }
    __syncthreads();

    bf16* out = c.a1 + (static_cast<long>(g) * c.B + b) * (kA1Pix * kA1C);
    uint32_t* mask_out = c.a1mask + (static_cast<long>(g) * c.B + b) * kA1Pix;
    constexpr int kBlocks = (kA1Pix + 31) / 32;  // 22, the last one holds 4 pixels
    const int blk0 = part * kBlocks / c.conv1_split, blk1 = (part + 1) * kBlocks / c.conv1_split;
    for (int blk = blk0 + wave; blk < blk1; blk += 4) {
        const int p = blk * 32 + n;
        const int pc = min(p, kA1Pix - 1);
        const int y = pc / 26, x = pc - y * 26;
        const int pb = y * 28 + x;
        f32x16 acc = {0};
#pragma unroll
        for (int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[s], xs[pb + xoff[s]], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[4], xs[h ? 784 : pb + 58], acc, 0, 0, 0);
        // register r: channel (r & 3) + 8 * (r >> 2) + 4 * h of pixel p
        unsigned bits = 0;
        unsigned pk[8];
        float neg[16];  // 0 - acc: its sign bit says acc > 0, and max(-neg, 0) is the ReLU
#pragma unroll
        for (int r = 0; r < 16; ++r) neg[r] = 0.f - acc[r];
#pragma unroll
        for (int r = 15; r >= 0; --r)  // bit r <- (acc[r] > 0)
            bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(neg[r]), 31);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // (the maximum is taken of an arithmetic result, which needs no canonicalising
            // instruction in front of it; of the matrix result itself it would)
            const bf16x2 two = {f2bf(fmaxf(-neg[2 * i], 0.f)), f2bf(fmaxf(-neg[2 * i + 1], 0.f))};
            pk[i] = __builtin_bit_cast(unsigned, two);
        }
        // pk[2 q'], pk[2 q' + 1]: channels 8 q' + 4 h .. + 3.  Swap the upper half's q' = 0
        // (2) with the lower half's q' = 1 (3): h = 0 then holds channels 0..7 (16..23),
        // h = 1 channels 8..15 (24..31) of the pixel.
#pragma unroll
        for (int pair = 0; pair < 2; ++pair)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const auto sw = __builtin_amdgcn_permlane32_swap(pk[4 * pair + w], pk[4 * pair + 2 + w],
                                                                 false, false);
                pk[4 * pair + w] = sw[0];
                pk[4 * pair + 2 + w] = sw[1];
            }
        const unsigned other = __shfl_xor(bits, 32, 64);
        if (p < kA1Pix) {
            uint4* dst = reinterpret_cast<uint4*>(out + p * kA1C + 8 * h);
            dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            dst[2] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            if (h == 0) mask_out[p] = bits | (other << 16);
        }
    }
}

// ---------------------------------------------------------------------------
constexpr int kFwdLds = 64 * 592;  // filter staging (37888 B) > band image (10 * 1680)

// LDS image of a band: 10 rows of 26 pixels, 64 bytes per pixel, rows 1680 bytes apart
// (nfix::mnist_fwd_off).  No swizzle: with the window-major row order of the MFMA blocks
// the odd row pitch alone makes every ds_read_b128 lane group conflict-free, and a
// fragment address is a per-lane base plus a compile-time offset.
constexpr int kFwdRowPitch = nfix::kMnistFwdRowPitch, kFwdPixPitch = nfix::kMnistFwdPixPitch;
constexpr int kFwdRowPieces = kFwdRowPitch / 16 - 1;  // 104 pieces of 16 bytes per image row

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv2_fwd_kernel(MnistCtx c) {
    if (client_idle(c, blockIdx.y)) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x / c.split, g = blockIdx.y, tid = threadIdx.x;
    const int band0 = c.split == 1 ? 0 : blockIdx.x - b * c.split;
    const int band1 = c.split == 1 ? 3 : band0 + 1;
    const int lane = tid & 63, wave = tid >> 6;
    const int nb = wave & 1, mgrp = wave >> 1;
    const int n = lane & 31, h = lane >> 5;

    // stationary operand: this wave's 32 output channels, all of K = 9 x 32.
    // The whole filter bank (36 KB, rows of 576 B padded to 592) goes through
    // LDS once per workgroup: coalesced fetch, conflict-free fragment reads.
    bf16x8 Bf[18];
    {
        const uint4* wsrc =
            reinterpret_cast<const uint4*>(c.shadow + g * c.shadow_stride + c.off_c2w);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int p = tid + i * 256, row = p / 36, part = p - row * 36;
            *reinterpret_cast<uint4*>(smem + row * 592 + part * 16) = wsrc[p];
        }
        __syncthreads();
        const char* wrow = smem + (nb * 32 + n) * 592 + 16 * h;
#pragma unroll
        for (int s = 0; s < 18; ++s) Bf[s] = *reinterpret_cast<const bf16x8*>(wrow + s * 32);
        // the band loop starts with a barrier before this area is overwritten
    }
    const float bias = (c.arena + g * c.arena_stride)[c.off_c2b + nb * 32 + n];

    // row m of a 32-pixel block is position q of pooling window wi (m = 4*wi + q)
    const int dy = nfix::mnist_fwd_block_dy(n), dx = nfix::mnist_fwd_block_dx(n);
    // this wave's blocks of a band: rows 4 * mgrp .. + 3, columns 8 * i .. + 7, i = 0, 1, 2
    const char* const afrag = smem + nfix::mnist_fwd_off(4 * mgrp + dy, dx, h);

    const bf16* a1img = c.a1 + (static_cast<long>(g) * c.B + b) * (kA1Pix * kA1C);
    const float keep_scale = c.train ? 1.0f / (1.0f - c.p_drop1) : 1.0f;
    const unsigned long long ctr = c.train ? c.step_counter[g] : 0ull;
    const unsigned long long seed = client_seed(c, g);
    // dropout decisions are 8-bit draws: p is honoured to 1/256
    const unsigned drop_thr = c.train ? static_cast<unsigned>(c.p_drop1 * 256.0f + 0.5f) : 0u;
    // pooled pixel (2 * mgrp, h) of band 0, this lane's channel: every store of the
    // kernel is this plus a compile-time offset (band, block, window row / column)
    const long out0 = ((static_cast<long>(g) * c.B + b) * kP2Pix + 2 * mgrp * 12 + h) * kC2 +
                      nb * 32 + n;
    bf16* const p2out = c.p2 + out0;
    uint8_t* const pmout = c.pmask + out0;

    // fixed piece -> LDS address of this thread's share of a band (pieces
    // tid + 256 i; only threads 0..15 own a fifth piece)
    const int p4 = min(tid + 1024, 1039);
    const bool has4 = tid + 1024 < 1040;
#define NF_FWD_DST(p) (((p) / kFwdRowPieces) * kFwdRowPitch + ((p) % kFwdRowPieces) * 16)
    const int dst0 = NF_FWD_DST(tid), dst1 = NF_FWD_DST(tid + 256), dst2 = NF_FWD_DST(tid + 512),
              dst3 = NF_FWD_DST(tid + 768), dst4 = NF_FWD_DST(p4);
#undef NF_FWD_DST
    // band b+1 travels HBM -> registers while band b is being multiplied
    // (named registers: an indexed array here ends up in scratch memory)
    uint4 pre0, pre1, pre2, pre3, pre4;
#define NF_FWD_ISSUE(band_)                                                               \
    {                                                                                     \
        const uint4* src_ = reinterpret_cast<const uint4*>(a1img + (band_) * 8 * 26 * kA1C); \
        pre0 = src_[tid]; pre1 = src_[tid + 256]; pre2 = src_[tid + 512];                 \
        pre3 = src_[tid + 768]; pre4 = src_[p4];                                          \
    }
    NF_FWD_ISSUE(band0)
    for (int band = band0; band < band1; ++band) {
        __syncthreads();
        *reinterpret_cast<uint4*>(smem + dst0) = pre0;
        *reinterpret_cast<uint4*>(smem + dst1) = pre1;
        *reinterpret_cast<uint4*>(smem + dst2) = pre2;
        *reinterpret_cast<uint4*>(smem + dst3) = pre3;
        if (has4) *reinterpret_cast<uint4*>(smem + dst4) = pre4;
        if (band + 1 < band1) NF_FWD_ISSUE(band + 1)
        __syncthreads();
#undef NF_FWD_ISSUE

        unsigned rnd[3] = {~0u, ~0u, ~0u};
        if (c.train) {
            const unsigned long long blk =
                ((static_cast<unsigned long long>(b) * 3 + band) * 2 + mgrp) * 64 + nb * 32 + n;
            const nf::Philox4 r = nf::philox4x32_10(seed, 2 * ctr, blk);
            rnd[0] = r.v[0]; rnd[1] = r.v[1]; rnd[2] = r.v[2];
        }
        const int band_off = band * 4 * 12 * kC2;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            f32x16 acc = {0};
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                const int tap = s >> 1, kh = tap / 3, kw = tap - kh * 3;
                const bf16x8 A = *reinterpret_cast<const bf16x8*>(
                    afrag + kh * kFwdRowPitch + (8 * i + kw) * kFwdPixPitch + (s & 1) * 32);
                acc = mfma32(A, Bf[s], acc);
            }
            const unsigned rbits = rnd[i];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                // max and arg-max (first of equals) of the window; the bias is common to
                // the four positions, so it is added to the maximum only
                const float a0 = acc[4 * rg], a1 = acc[4 * rg + 1], a2 = acc[4 * rg + 2],
                            a3 = acc[4 * rg + 3];
                const float mx = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
                // (flat selects and a bitwise AND: nested ?: and && become branches here)
                const bool c0 = a0 == mx, c1 = a1 == mx, c2 = a2 == mx;
                int am = c2 ? 2 : 3;
                am = c1 ? 1 : am;
                am = c0 ? 0 : am;
                const float best = mx + bias;
                const bool alive = (best > 0.f) & (((rbits >> (8 * rg)) & 0xFFu) >= drop_thr);
                const float val = alive ? best * keep_scale : 0.f;
                // pooled pixel (band * 4 + 2 * mgrp + (rg >> 1), 4 * i + 2 * (rg & 1) + h)
                const int o = band_off + ((rg >> 1) * 12 + 4 * i + 2 * (rg & 1)) * kC2;
                p2out[o] = f2bf(val);
                pmout[o] = static_cast<uint8_t>(am + (alive ? 4 : 0));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// conv1 and conv2 forward in ONE launch.  Both are per image: the workgroup that owns an
// image (or one band of it) computes conv1's output for its rows on the fp32 matrix cores
// (conv1_fwd_kernel's scheme) straight into the LDS image that conv2's MFMA loop reads
// (conv2_fwd_kernel's scheme, all rows resident, so no staging and no barriers between
// bands), and writes it to memory only for the backward pass.  One launch, one dependent
// kernel boundary and one 5.5 MB round trip per client and step less than the two
// kernels; results are bit-identical to theirs.
constexpr int kFusedXs = 784 + 4;
constexpr int kFusedImgBytes = 26 * kFwdRowPitch;  // 43680
constexpr int kFusedLds = kFusedImgBytes + kFusedXs * 4;

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv12_fwd_kernel(MnistCtx c, int idx_off) {
    if (client_idle(c, blockIdx.y)) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem + kFusedImgBytes);
    const int b = blockIdx.x / c.split, g = blockIdx.y, tid = threadIdx.x;
    const int band0 = c.split == 1 ? 0 : blockIdx.x - b * c.split;
    const int band1 = c.split == 1 ? 3 : band0 + 1;
    const int lane = tid & 63, wave = tid >> 6;
    const int nb = wave & 1, mgrp = wave >> 1;
    const int n = lane & 31, h = lane >> 5;
    const long img = static_cast<long>(g) * c.B + b;

    // input image (its loads are the longest dependent chain of the kernel: first)
    load_image(c, c.idx[g * c.idx_stride + idx_off + b], tid, xs);
    if (tid == 196) xs[784] = 1.0f;
    // conv2's stationary operand through the (still unused) image area, as in conv2_fwd
    bf16x8 Bf[18];
    {
        const uint4* wsrc =
            reinterpret_cast<const uint4*>(c.shadow + g * c.shadow_stride + c.off_c2w);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int p = tid + i * 256, row = p / 36, part = p - row * 36;
            *reinterpret_cast<uint4*>(smem + row * 592 + part * 16) = wsrc[p];
        }
    }
    // conv1's A operand: row = channel n, k = 2 s + h (k = 9: the bias)
    const float* wa = c.arena + g * c.arena_stride;
    float wk[5];
    int xoff[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int k = 2 * s + h;
        wk[s] = k < 9 ? wa[c.off_c1w + n * 9 + k] : wa[c.off_c1b + n];
        xoff[s] = (k / 3) * 28 + k % 3;
    }
    const float bias = wa[c.off_c2b + nb * 32 + n];
    __syncthreads();
    {
        const char* wrow = smem + (nb * 32 + n) * 592 + 16 * h;
#pragma unroll
        for (int s = 0; s < 18; ++s) Bf[s] = *reinterpret_cast<const bf16x8*>(wrow + s * 32);
    }
    __syncthreads();  // the filter bank is in registers: the area becomes the image

    // ---- conv1 for image rows row0 .. row0 + nrows - 1 (the bands' rows plus two)
    const int row0 = band0 * 8, nrows = (band1 - band0) * 8 + 2;
    const int own_rows = band1 == 3 ? nrows : nrows - 2;  // rows another band's workgroup does not write
    const int npix = nrows * 26;
    bf16* a1out = c.a1 + img * (kA1Pix * kA1C);
    uint32_t* mask_out = c.a1mask + img * kA1Pix;
    for (int blk = wave; blk * 32 < npix; blk += 4) {
        const int p = blk * 32 + n;
        const int pc = min(p, npix - 1);
        const int yl = pc / 26, x = pc - yl * 26;
        const int pb = (row0 + yl) * 28 + x;
        f32x16 acc = {0};
#pragma unroll
        for (int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[s], xs[pb + xoff[s]], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[4], xs[h ? 784 : pb + 58], acc, 0, 0, 0);
        unsigned bits = 0;
        unsigned pk[8];
        float neg[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) neg[r] = 0.f - acc[r];
#pragma unroll
        for (int r = 15; r >= 0; --r)
            bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(neg[r]), 31);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x2 two = {f2bf(fmaxf(-neg[2 * i], 0.f)), f2bf(fmaxf(-neg[2 * i + 1], 0.f))};
            pk[i] = __builtin_bit_cast(unsigned, two);
        }
#pragma unroll
        for (int pair = 0; pair < 2; ++pair)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const auto sw = __builtin_amdgcn_permlane32_swap(pk[4 * pair + w], pk[4 * pair + 2 + w],
                                                                 false, false);
                pk[4 * pair + w] = sw[0];
                pk[4 * pair + 2 + w] = sw[1];
            }
        const unsigned other = __shfl_xor(bits, 32, 64);
        if (p < npix) {
            const uint4 lo = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            const uint4 hi = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            // channels 8 h .. + 7 and 16 + 8 h .. + 7: chunks h and 2 + h of the pixel
            char* ldst = smem + nfix::mnist_fwd_off(yl, x, h);
            *reinterpret_cast<uint4*>(ldst) = lo;
            *reinterpret_cast<uint4*>(ldst + 32) = hi;
            if (yl < own_rows) {
                const int gp = (row0 + yl) * 26 + x;
                uint4* dst = reinterpret_cast<uint4*>(a1out + gp * kA1C + 8 * h);
                dst[0] = lo;
                dst[2] = hi;
                if (h == 0) mask_out[gp] = bits | (other << 16);
            }
        }
    }
    __syncthreads();

    // ---- conv2 over the resident rows
    const int dy = nfix::mnist_fwd_block_dy(n), dx = nfix::mnist_fwd_block_dx(n);
    const char* const afrag0 = smem + nfix::mnist_fwd_off(4 * mgrp + dy, dx, h);
    const float keep_scale = c.train ? 1.0f / (1.0f - c.p_drop1) : 1.0f;
    const unsigned long long ctr = c.train ? c.step_counter[g] : 0ull;
    const unsigned long long seed = client_seed(c, g);
    const unsigned drop_thr = c.train ? static_cast<unsigned>(c.p_drop1 * 256.0f + 0.5f) : 0u;
    const long out0 = (img * kP2Pix + 2 * mgrp * 12 + h) * kC2 + nb * 32 + n;
    bf16* const p2out = c.p2 + out0;
    uint8_t* const pmout = c.pmask + out0;
    for (int band = band0; band < band1; ++band) {
        unsigned rnd[3] = {~0u, ~0u, ~0u};
        if (c.train) {
            const unsigned long long blk =
                ((static_cast<unsigned long long>(b) * 3 + band) * 2 + mgrp) * 64 + nb * 32 + n;
            const nf::Philox4 r = nf::philox4x32_10(seed, 2 * ctr, blk);
            rnd[0] = r.v[0]; rnd[1] = r.v[1]; rnd[2] = r.v[2];
        }
        const char* const afrag = afrag0 + (band - band0) * 8 * kFwdRowPitch;
        const int band_off = band * 4 * 12 * kC2;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            f32x16 acc = {0};
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                const int tap = s >> 1, kh = tap / 3, kw = tap - kh * 3;
                const bf16x8 A = *reinterpret_cast<const bf16x8*>(
                    afrag + kh * kFwdRowPitch + (8 * i + kw) * kFwdPixPitch + (s & 1) * 32);
                acc = mfma32(A, Bf[s], acc);
            }
            const unsigned rbits = rnd[i];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float a0 = acc[4 * rg], a1 = acc[4 * rg + 1], a2 = acc[4 * rg + 2],
                            a3 = acc[4 * rg + 3];
                const float mx = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
                const bool c0 = a0 == mx, c1 = a1 == mx, c2 = a2 == mx;
                int am = c2 ? 2 : 3;
                am = c1 ? 1 : am;
                am = c0 ? 0 : am;
                const float best = mx + bias;
                const bool alive = (best > 0.f) & (((rbits >> (8 * rg)) & 0xFFu) >= drop_thr);
                const float val = alive ? best * keep_scale : 0.f;
                const int o = band_off + ((rg >> 1) * 12 + 4 * i + 2 * (rg & 1)) * kC2;
                p2out[o] = f2bf(val);
                pmout[o] = static_cast<uint8_t>(am + (alive ? 4 : 0));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Split-K slice of h1 = p2 W1^T: K chunk of 288 (18 MFMA steps) per workgroup.
// Both operand tiles are fetched as whole 576-byte row segments (coalesced,
// full cache lines) into LDS and read back as MFMA fragments; rows are padded
// to 592 bytes so the 16 rows of a ds_read_b128 lane group fall on 16
// different 16-byte slots.  (Fragment-shaped global loads - 64 lanes touching
// 64 different cache lines per instruction - made this kernel address-bound.)
constexpr int kFc1KC = 288, kFc1S = kFc1K / kFc1KC;      // 32 slabs
constexpr int kFc1Pitch = kFc1KC * 2 + 16;                // 592
// A workgroup computes 64 batch rows x 64 of the 128 hidden units of a slab (blockIdx.x =
// 2 * slab + half): 76 KB of LDS instead of the 114 KB of a whole slab, so that a workgroup of
// another chain's kernel fits on the CU beside it, and twice the workgroups (64 per client)
// for a launch that does not fill the chip.  A wave owns one 32 x 32 tile; every output adds
// the same products in the same order as before.
constexpr int kFc1FwdLds = (64 + 64) * kFc1Pitch;         // 75776

__global__ void __launch_bounds__(256) fc1_fwd_kernel(MnistCtx c) {
    if (client_idle(c, blockIdx.y)) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                    // [64 j][288 k]
    char* s_a = smem + 64 * kFc1Pitch;   // [64 b][288 k]
    const int kc = blockIdx.x >> 1, nh = blockIdx.x & 1, g = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int cb = wave & 1, rb = wave >> 1;  // column / row block of this wave's tile
    const bf16* wsrc = c.shadow + g * c.shadow_stride + c.off_f1w +
                       static_cast<long>(nh) * 64 * kFc1K + kc * kFc1KC;
    const bf16* asrc = c.p2 + static_cast<long>(g) * c.B * kFc1K + kc * kFc1KC;
    constexpr int kPiecesPerRow = kFc1KC / 8;  // 36
#pragma unroll 3
    for (int p = tid; p < 64 * kPiecesPerRow; p += 256) {
        const int row = p / kPiecesPerRow, part = p - row * kPiecesPerRow;
        *reinterpret_cast<uint4*>(s_w + row * kFc1Pitch + part * 16) =
            *reinterpret_cast<const uint4*>(wsrc + static_cast<long>(row) * kFc1K + part * 8);
        *reinterpret_cast<uint4*>(s_a + row * kFc1Pitch + part * 16) =
            *reinterpret_cast<const uint4*>(asrc + static_cast<long>(row) * kFc1K + part * 8);
    }
    __syncthreads();
    const char* ap = s_a + (rb * 32 + n) * kFc1Pitch + 16 * h;
    const char* bp = s_w + (cb * 32 + n) * kFc1Pitch + 16 * h;
    f32x16 acc = {0};
#pragma unroll
    for (int ks = 0; ks < kFc1KC / 16; ++ks)
        acc = mfma32(*reinterpret_cast<const bf16x8*>(ap + ks * 32),
                     *reinterpret_cast<const bf16x8*>(bp + ks * 32), acc);
    float* out = c.fc1_slabs + ((static_cast<long>(g) * c.S + kc) * c.B) * 128 + nh * 64 + cb * 32 + n;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(rb * 32 + acc_row(r, lane)) * 128] = acc[r];
}

// ---------------------------------------------------------------------------
// Sum over the 32 lanes of a wave half, in every one of them: quad and row butterflies as
// DPP adds (one instruction each), then the two 16-lane rows of the half exchange their
// sums (v_permlane16_swap).
template <int kCtrl>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, 0xF, 0xF, true));
}
__device__ __forceinline__ float half_wave_sum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1, 0, 3, 2]
    v += dpp_f<0x4E>(v);   // quad_perm [2, 3, 0, 1]
    v += dpp_f<0x141>(v);  // row_half_mirror
    v += dpp_f<0x140>(v);  // row_mirror
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

// 8 batch rows per workgroup; thread = (row r, 4 consecutive hidden units); a row is one
// half of a wave.  Everything up to d(logits) happens in registers (the 32 lanes of a row
// reduce their partial logits with DPP adds and all hold the row's logits), so the kernel
// has one barrier: before the sums over the 8 rows (fc2's weight gradient, the bias
// gradients, the loss).  The slabs, fc2's weights and the label are fetched first, all at
// once - the kernel is a chain of latencies, not of arithmetic.
// (write_dh1 = false: the caller's fc1_wgrad scales the rows of the norm pass's d(h1) by the
// clip factors itself, see dp_clip_kernel)
__device__ __forceinline__ void head_body(const MnistCtx& c, int idx_off, int step, int hb, int g,
                                          bool write_dh1) {
    if (client_idle(c, g)) return;
    const int tid = threadIdx.x;
    __shared__ float h1s[kHeadRows][132];
    __shared__ float dvs[kHeadRows][132];  // d(h1 pre-activation)
    __shared__ float dl[kHeadRows][12];
    __shared__ float red[2][kHeadRows];
    const int r0 = hb * kHeadRows;
    const float* wa = c.arena + g * c.arena_stride;
    const int r = tid >> 5, j0 = (tid & 31) * 4;
    const int b = r0 + r;
    // the label is two dependent loads away: start them before anything else
    const int y = c.labels[c.idx[g * c.idx_stride + idx_off + b]];
    const float* base = c.fc1_slabs + (static_cast<long>(g) * c.S * c.B + b) * 128 + j0;
    const long sstride = static_cast<long>(c.B) * 128;
    // all slabs in flight at once (check_ctx pins S to kFc1S)
    float4 sl[kFc1S];
#pragma unroll
    for (int s = 0; s < kFc1S; ++s) sl[s] = *reinterpret_cast<const float4*>(base + s * sstride);
    // fc2: this thread's 4 columns of the 10 rows, and the biases
    float4 w2r[10];
#pragma unroll
    for (int o = 0; o < 10; ++o) w2r[o] = *reinterpret_cast<const float4*>(wa + c.off_f2w + o * 128 + j0);
    const float4 bv = *reinterpret_cast<const float4*>(wa + c.off_f1b + j0);
    float b2[10];
#pragma unroll
    for (int o = 0; o < 10; ++o) b2[o] = wa[c.off_f2b + o];
    const int nv = c.nvalid[g * c.nvalid_stride + step];
    const int nt = c.ntotal != nullptr ? c.ntotal[g * c.nvalid_stride + step] : nv;

    // A: reduce the split-K slabs (two chains, even and odd slabs), bias, ReLU, dropout
    float h[4], dm[4];
    {
        float4 s0 = bv, s1 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < kFc1S; s += 2) {
            s0.x += sl[s].x; s0.y += sl[s].y; s0.z += sl[s].z; s0.w += sl[s].w;
            s1.x += sl[s + 1].x; s1.y += sl[s + 1].y; s1.z += sl[s + 1].z; s1.w += sl[s + 1].w;
        }
        const float acc[4] = {s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w};
        float u[4] = {1.f, 1.f, 1.f, 1.f};
        float sc = 1.0f;
        if (c.train) {
            const unsigned long long ctr = c.step_counter[g];
            const unsigned long long blk = (static_cast<unsigned long long>(b) * 128 + j0) >> 2;
            nf::philox_samples(client_seed(c, g), 2 * ctr + 1, blk, 0, u);
            sc = 1.0f / (1.0f - c.p_drop2);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool alive = (acc[e] > 0.f) & (!c.train | (u[e] >= c.p_drop2));
            h[e] = alive ? acc[e] * sc : 0.f;
            dm[e] = alive ? sc : 0.f;
        }
        *reinterpret_cast<float4*>(&h1s[r][j0]) = make_float4(h[0], h[1], h[2], h[3]);
    }

    // B: logits of the row, in every lane of its wave half
    float lg[10];
#pragma unroll
    for (int o = 0; o < 10; ++o) {
        float p = h[0] * w2r[o].x;
        p = fmaf(h[1], w2r[o].y, p);
        p = fmaf(h[2], w2r[o].z, p);
        p = fmaf(h[3], w2r[o].w, p);
        lg[o] = half_wave_sum(p) + b2[o];
    }

    // C: log-softmax, NLL, accuracy, d(logits)
    const bool valid = b < nv;
    float m = lg[0];
    int pred = 0;
#pragma unroll
    for (int o = 1; o < 10; ++o)
        if (lg[o] > m) { m = lg[o]; pred = o; }
    float se = 0.f;
#pragma unroll
    for (int o = 0; o < 10; ++o) se += __expf(lg[o] - m);
    const float lse = m + __logf(se);
    const float inv = nt > 0 ? 1.0f / static_cast<float>(nt) : 0.f;
    float dlr[10];
    float ly = 0.f;
#pragma unroll
    for (int o = 0; o < 10; ++o) {
        const float p = __expf(lg[o] - lse);
        dlr[o] = valid ? (p - (o == y ? 1.f : 0.f)) * inv : 0.f;
        ly = o == y ? lg[o] : ly;
    }
    if (c.dp_pass != 0) {
        const long bi = static_cast<long>(g) * c.B + b;
        if (c.dp_pass == 1) {
            // per-sample gradient of fc2: dl (x) h1 and dl
            float hq = fmaf(h[0], h[0], fmaf(h[1], h[1], fmaf(h[2], h[2], h[3] * h[3])));
            const float hsq = 1.f + half_wave_sum(hq);
            float dsq = 0.f;
#pragma unroll
            for (int o = 0; o < 10; ++o) dsq = fmaf(dlr[o], dlr[o], dsq);
            if ((tid & 31) == 0) c.dp_normsq[bi * 4 + 0] = hsq * dsq;
        } else {
            const float scl = c.dp_rowscale[bi];
#pragma unroll
            for (int o = 0; o < 10; ++o) dlr[o] *= scl;
        }
    }
    if ((tid & 31) == 0) {
#pragma unroll
        for (int o = 0; o < 10; ++o) dl[r][o] = dlr[o];
        red[0][r] = valid ? (lse - ly) : 0.f;
        red[1][r] = (valid && pred == y) ? 1.f : 0.f;
    }
    const bool eval_only = !c.train && c.eval_out != nullptr;
    if (!eval_only) {
        // E: d(h1 pre-activation), from registers
        bf16x4 o4;
        float v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = 0.f;
#pragma unroll
            for (int o = 0; o < 10; ++o) {
                const float w = e == 0 ? w2r[o].x : (e == 1 ? w2r[o].y : (e == 2 ? w2r[o].z : w2r[o].w));
                v = fmaf(dlr[o], w, v);
            }
            v *= dm[e];
            v4[e] = v;
            o4[e] = f2bf(v);
        }
        if (write_dh1)
            *reinterpret_cast<bf16x4*>(c.dh1 + (static_cast<long>(g) * c.B + b) * 128 + j0) = o4;
        *reinterpret_cast<float4*>(&dvs[r][j0]) = make_float4(v4[0], v4[1], v4[2], v4[3]);
    }
    __syncthreads();

    float loss_sum = 0.f, correct = 0.f;
    if (tid == 0) {
        for (int i = 0; i < kHeadRows; ++i) { loss_sum += red[0][i]; correct += red[1][i]; }
    }
    if (eval_only) {
        // one slot per workgroup, overwritten every step: the host adds the slots of
        // the clients that had a batch, in a fixed order (no float atomics)
        if (tid == 0) {
            float* e = c.eval_out + (static_cast<long>(g) * (c.B / kHeadRows) + hb) * 2;
            e[0] = loss_sum;
            e[1] = correct;
        }
        return;
    }

    float* part = c.head_part + (static_cast<long>(g) * (c.B / kHeadRows) + hb) * kHeadPart;
    // D: fc2 weight / bias gradient (partial over this block's rows)
    for (int i = tid; i < 1280; i += 256) {
        const int o = i >> 7, j = i & 127;
        float acc = 0.f;
#pragma unroll
        for (int rr = 0; rr < kHeadRows; ++rr) acc = fmaf(dl[rr][o], h1s[rr][j], acc);
        part[i] = acc;
    }
    if (tid < 10) {
        float acc = 0.f;
        for (int rr = 0; rr < kHeadRows; ++rr) acc += dl[rr][tid];
        part[1280 + tid] = acc;
    }
    if (tid < 128) {
        float acc = 0.f;
        for (int rr = 0; rr < kHeadRows; ++rr) acc += dvs[rr][tid];
        part[1296 + tid] = acc;
    }
    if (tid == 0) { part[1424] = loss_sum; part[1425] = correct; }
}

__global__ void __launch_bounds__(256) head_kernel(MnistCtx c, int idx_off, int step) {
    head_body(c, idx_off, step, blockIdx.x, blockIdx.y, true);
}

// ---------------------------------------------------------------------------
constexpr int kTrPitch128 = 320;  // 256 B of data + 64 B: 4 consecutive k rows hit distinct banks

// The product is formed transposed, D[channel][batch row] = W1 tile^T * dh1^T (the same two
// fragments as for D[batch row][channel], operands exchanged), so that a lane ends up with
// runs of 4 consecutive channels of ONE pooled pixel of one sample; after a half-swap
// between the two lanes of a sample (v_permlane32_swap) it owns 8 consecutive channels:
// 8-byte loads of the arg-max / alive codes and 16-byte stores of the gradient instead of
// one byte and two bytes per lane and instruction.
__global__ void __launch_bounds__(256) fc1_dgrad_kernel(MnistCtx c) {
    if (client_idle(c, blockIdx.y)) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [128 j][128 k'] pitch 320
    const int tile = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
    const bf16* w = c.shadow + g * c.shadow_stride + c.off_f1w + tile * 128;
    for (int p = tid; p < 2048; p += 256) {
        const int row = p >> 4, part = p & 15;
        *reinterpret_cast<uint4*>(smem + row * kTrPitch128 + part * 16) =
            *reinterpret_cast<const uint4*>(w + static_cast<long>(row) * kFc1K + part * 8);
    }
    const bf16* dh = c.dh1 + static_cast<long>(g) * c.B * 128 + 8 * h;
    bf16x8 Af[2][8];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            Af[mb][ks] = *reinterpret_cast<const bf16x8*>(dh + (mb * 32 + n) * 128 + ks * 16);
    // the codes this lane will need in the epilogue: fetched now, used after the GEMM.
    // Sample mb * 32 + n, pooled pixel pp, channels cbase + 16 * pair + 8 * h .. + 7.
    const int pp = tile * 2 + (wave >> 1), cbase = (wave & 1) * 32;
    const long o0 = ((static_cast<long>(g) * c.B + n) * kP2Pix + pp) * kC2 + cbase + 8 * h;
    uint2 mk[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int pair = 0; pair < 2; ++pair)
            mk[mb][pair] = *reinterpret_cast<const uint2*>(
                c.pmask + o0 + static_cast<long>(mb) * 32 * kP2Pix * kC2 + 16 * pair);
    __syncthreads();
    f32x16 acc[2] = {{0}, {0}};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 Wfr = load_tr_frag(smem, kTrPitch128, ks * 16, wave * 32, lane);
        acc[0] = mfma32(Wfr, Af[0][ks], acc[0]);
        acc[1] = mfma32(Wfr, Af[1][ks], acc[1]);
    }
    // dropout + ReLU backward on the pooled gradient.  conv2's bias gradient is the sum of
    // these (bf16-rounded) values over samples and pixels: every wave leaves its
    // [64 samples][32 channels] tile in LDS as well and sums its columns with four MFMAs
    // against a matrix of ones.
    const float keep_scale = c.train ? 1.0f / (1.0f - c.p_drop1) : 1.0f;
    __syncthreads();  // the weight tile is no longer needed: its space takes the tiles
    constexpr int kSumPitch = 80;  // 64 B of channels + 16: conflict-free transposed reads
    char* const tile_w = smem + wave * (64 * kSumPitch);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        // register r: channel cbase + (r & 3) + 8 * (r >> 2) + 4 * h
        unsigned pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x2 two = {f2bf(acc[mb][2 * i] * keep_scale), f2bf(acc[mb][2 * i + 1] * keep_scale)};
            pk[i] = __builtin_bit_cast(unsigned, two);
        }
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
#pragma unroll
            for (int ww = 0; ww < 2; ++ww) {
                const auto sw = __builtin_amdgcn_permlane32_swap(pk[4 * pair + ww], pk[4 * pair + 2 + ww],
                                                                 false, false);
                pk[4 * pair + ww] = sw[0];
                pk[4 * pair + 2 + ww] = sw[1];
            }
            // pk[4 pair .. + 3]: channels cbase + 16 pair + 8 h + 0..7; byte e of the codes
            // belongs to channel e, its bit 2 says "alive"
            const unsigned mw[2] = {mk[mb][pair].x, mk[mb][pair].y};
            unsigned out[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lo = __builtin_amdgcn_sbfe(static_cast<int>(mw[j >> 1]), 16 * (j & 1) + 2, 1);
                const int hi = __builtin_amdgcn_sbfe(static_cast<int>(mw[j >> 1]), 16 * (j & 1) + 10, 1);
                out[j] = pk[4 * pair + j] & __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            }
            const uint4 o4 = make_uint4(out[0], out[1], out[2], out[3]);
            *reinterpret_cast<uint4*>(c.dp2 + o0 + static_cast<long>(mb) * 32 * kP2Pix * kC2 + 16 * pair) = o4;
            *reinterpret_cast<uint4*>(tile_w + (mb * 32 + n) * kSumPitch + (16 * pair + 8 * h) * 2) = o4;
        }
    }
    // column sums: D[any row][channel] = sum over the 64 samples (k) of tile[k][channel]
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = f2bf(1.0f);
    f32x16 colsum = {0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        colsum = mfma32(ones, load_tr_frag(tile_w, kSumPitch, ks * 16, 0, lane), colsum);
    // register 0 of the lower lane half is row 0: channel cbase + n of pooled pixel pp
    if (h == 0)
        c.c2b_part[(static_cast<long>(g) * 2 * kFc1Tiles + pp) * 64 + cbase + n] = colsum[0];
}

// ---------------------------------------------------------------------------
// scale_rows: d(h1) in memory is the norm pass's; row b is multiplied by the clip factor of
// sample b while it is staged (DP-SGD, clipped pass in one launch with the head).
__device__ __forceinline__ void fc1_wgrad_body(const MnistCtx& c, int step, int tile, int g,
                                               bool scale_rows = false) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_dh = smem;                    // [64 b][128 j]
    char* s_p2 = smem + 64 * kTrPitch128; // [64 b][128 k']
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, n = lane & 31;
    if (client_idle(c, g) || (c.fuse_sgd && c.nvalid[g * c.nvalid_stride + step] == 0)) return;
    const bf16* dh = c.dh1 + static_cast<long>(g) * c.B * 128;
    const bf16* p2 = c.p2 + static_cast<long>(g) * c.B * kFc1K + tile * 128;
    for (int p = tid; p < 1024; p += 256) {
        const int row = p >> 4, part = p & 15;
        bf16x8 dv = *reinterpret_cast<const bf16x8*>(dh + row * 128 + part * 8);
        if (scale_rows) {
            const float scl = c.dp_rowscale[static_cast<long>(g) * c.B + row];
#pragma unroll
            for (int k = 0; k < 8; ++k) dv[k] = f2bf(bf2f(dv[k]) * scl);
        }
        *reinterpret_cast<bf16x8*>(s_dh + row * kTrPitch128 + part * 16) = dv;
        *reinterpret_cast<uint4*>(s_p2 + row * kTrPitch128 + part * 16) =
            *reinterpret_cast<const uint4*>(p2 + static_cast<long>(row) * kFc1K + part * 8);
    }
    const long col = static_cast<long>(tile) * 128 + wave * 32 + n;
    float* pa = c.arena + g * c.arena_stride + c.off_f1w + col;
    // start the read of the master weights early when the update is fused
    __syncthreads();
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 Bfr = load_tr_frag(s_p2, kTrPitch128, ks * 16, wave * 32, lane);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const bf16x8 Afr = load_tr_frag(s_dh, kTrPitch128, ks * 16, mb * 32, lane);
            acc[mb] = mfma32(Afr, Bfr, acc[mb]);
        }
    }
    if (c.fuse_sgd) {
        bf16* sh = c.shadow + g * c.shadow_stride + c.off_f1w + col;
        const float lr = c.lr, wd = c.wd;
        // DP-SGD (clipped pass): Gaussian noise of std sigma*C/n on the mean gradient.
        // One Philox block serves the four consecutive rows of an accumulator group.
        const bool noisy = c.dp_pass == 2 && c.dp_noise_std > 0.f;
        float nstd = 0.f;
        unsigned long long nctr = 0ull;
        if (noisy) {
            nstd = c.dp_noise_std / static_cast<float>(c.nvalid[g * c.nvalid_stride + step]);
            nctr = 2ull * c.step_counter[g] + 0x8000000000000001ull;
        }
        if (!noisy) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long o = static_cast<long>(mb * 32 + acc_row(r, lane)) * kFc1K;
                    float pv = pa[o];
                    pv -= lr * fmaf(wd, pv, acc[mb][r]);
                    pa[o] = pv;
                    sh[o] = f2bf(pv);
                }
        } else {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    float z[4];
                    const unsigned long long rg = mb * 8 + 2 * rq + (lane >> 5);  // row / 4
                    nf::philox_samples(client_seed(c, g), nctr, rg * kFc1K + col, 1, z);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = rq * 4 + k;
                        const long o = static_cast<long>(mb * 32 + acc_row(r, lane)) * kFc1K;
                        float pv = pa[o];
                        pv -= lr * fmaf(wd, pv, fmaf(nstd, z[k], acc[mb][r]));
                        pa[o] = pv;
                        sh[o] = f2bf(pv);
                    }
                }
        }
    } else {
        float* gr = c.grad + g * c.grad_stride + c.off_f1w + col;
        if (c.accumulate) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    gr[static_cast<long>(mb * 32 + acc_row(r, lane)) * kFc1K] += acc[mb][r];
        } else {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    gr[static_cast<long>(mb * 32 + acc_row(r, lane)) * kFc1K] = acc[mb][r];
        }
    }
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
fc1_wgrad_kernel(MnistCtx c, int step) {
    fc1_wgrad_body(c, step, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------
// d(conv2 out) from the pooled gradient and the arg-max codes: the value of a
// channel belongs to the window position that won the max.
// The 2x2 window of one pooled pixel, 8 channels: position q (= 2 * dy + dx)
// receives the pooled gradient of the channels whose arg-max code is q, zero
// for the others.  One load of gradient and codes serves all four positions.
//
// A 32-bit word holds two channels; which halves survive at position q is a function of
// their two 2-bit codes, i.e. of 16 cases: a 256-byte table in LDS holds, per case, the
// four AND masks of q = 0..3, so a word costs one ds_read_b128 and four ANDs instead of
// a chain of shifts, compares and selects (the expansion used to be half of the VALU
// instructions of conv2_dgrad).  Any 16 lanes read whole, 16-byte-aligned entries:
// different entries lie in different banks, equal ones are broadcast.
constexpr int kPoolLutBytes = 256;

__device__ __forceinline__ void pool_lut_init(char* lut, int tid) {
    if (tid < 64) {
        const unsigned entry = tid >> 2, q = tid & 3;
        const unsigned lo = (entry & 3u) == q ? 0x0000FFFFu : 0u;
        const unsigned hi = (entry >> 2) == q ? 0xFFFF0000u : 0u;
        reinterpret_cast<unsigned*>(lut)[tid] = lo | hi;
    }
}

struct PoolWindow {
    unsigned w[4];
    uint4 mk[4];  // mk[k]: masks of word k for q = 0..3
    // v: pooled gradient of 8 channels; m: their codes (bits 0-1 arg-max position)
    __device__ __forceinline__ PoolWindow(uint4 v, uint2 m, const char* lut) {
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        // byte offset of the entry of a channel pair: 16 * (code_lo | code_hi << 2)
        unsigned z0 = (m.x << 4) & 0x30303030u, z1 = (m.y << 4) & 0x30303030u;
        z0 |= z0 >> 6;
        z1 |= z1 >> 6;
        const char* t = static_cast<const char*>(__builtin_assume_aligned(lut, 16));
        mk[0] = *reinterpret_cast<const uint4*>(t + (z0 & 0xF0u));
        mk[1] = *reinterpret_cast<const uint4*>(t + ((z0 >> 16) & 0xF0u));
        mk[2] = *reinterpret_cast<const uint4*>(t + (z1 & 0xF0u));
        mk[3] = *reinterpret_cast<const uint4*>(t + ((z1 >> 16) & 0xF0u));
    }
    __device__ __forceinline__ uint4 at(int q) const {
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned msk = q == 0 ? mk[k].x : (q == 1 ? mk[k].y : (q == 2 ? mk[k].z : mk[k].w));
            o[k] = w[k] & msk;
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
};

// LDS image of the data-gradient patch (nfix::mnist_dg_off): 13 rows of 28 pixels, 144
// bytes per pixel, rows 4256 bytes apart - padded, not swizzled, so that the 36 fragment
// reads of a block are one per-lane base plus compile-time offsets.
constexpr int kDgPatchRows = nfix::kMnistDgPatchRows;
constexpr int kDgRowPitch = nfix::kMnistDgRowPitch, kDgPixPitch = nfix::kMnistDgPixPitch;
constexpr int kDgPatchBytes = kDgPatchRows * kDgRowPitch;         // 55328
constexpr int kDgPixPad = 704;                                    // 22 blocks of 32
// patch | normalised input image + 784 ones | pixel -> image offset | pixel -> ReLU bits |
// conv1 gradient tiles of the four waves | pooling table
constexpr int kDgLds = kDgPatchBytes + 2 * 784 * 4 + kDgPixPad * 4 * 2 + 4 * 32 * 10 * 4 + kPoolLutBytes;

__device__ __forceinline__ void conv2_dgrad_body(const MnistCtx& c, int idx_off, int bx, int g) {
    if (client_idle(c, g)) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem + kDgPatchBytes);       // normalised input image, then ones
    int* soff = reinterpret_cast<int*>(xs + 2 * 784);                 // a1 pixel -> offset in xs
    unsigned* smask = reinterpret_cast<unsigned*>(soff + kDgPixPad);  // a1 pixel -> ReLU bits
    float* red = reinterpret_cast<float*>(smask + kDgPixPad);
    const char* lut = reinterpret_cast<const char*>(red + 4 * 32 * 10);
    const int b = bx / c.split, tid = threadIdx.x;
    const int chunk0 = c.split == 1 ? 0 : bx - b * c.split;
    const int chunk1 = c.split == 1 ? 3 : chunk0 + 1;
    const int lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
    const long img = static_cast<long>(g) * c.B + b;

    // stationary operand: W2 as [k = (tap, oc)][col = c].  The 36 KB are fetched
    // once per workgroup with coalesced loads into the (still unused) patch area
    // and handed to the four waves from LDS; 128-byte rows, chunk XOR (row >> 1) & 7.
    bf16x8 Bf[36];
    {
        const uint4* wsrc = reinterpret_cast<const uint4*>(c.w2t + g * c.w2t_stride);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int p = tid + i * 256, row = p >> 3, part = p & 7;
            *reinterpret_cast<uint4*>(smem + row * 128 + ((part ^ ((row >> 1) & 7)) << 4)) = wsrc[p];
        }
        pool_lut_init(const_cast<char*>(lut), tid);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 36; ++ks) {
            const int row = (ks >> 2) * 32 + n, part = (ks & 3) * 2 + h;
            Bf[ks] = *reinterpret_cast<const bf16x8*>(
                smem + row * 128 + ((part ^ ((row >> 1) & 7)) << 4));
        }
        // the chunk loop starts with a barrier before the patch is overwritten
    }

    {
        load_image(c, c.idx[g * c.idx_stride + idx_off + b], tid, xs);
        if (tid < 196)
            *reinterpret_cast<float4*>(xs + 784 + 4 * tid) = make_float4(1.f, 1.f, 1.f, 1.f);
        const uint32_t* am = c.a1mask + img * kA1Pix;
        for (int p = tid; p < kDgPixPad; p += 256) {
            const int y = p / 26, x = p - y * 26;
            soff[p] = p < kA1Pix ? y * 28 + x : 0;
            smask[p] = p < kA1Pix ? am[p] : 0u;
        }
    }
    // conv1 gradient tile Z[c][t]: t = 0..8 the taps, t = 9 the bias (its "input" is the
    // block of ones behind the image); columns t > 9 read the image and are never stored
    f32x16 Z = {0};
    const int tapoff = n < 9 ? (n / 3) * 28 + (n % 3) : (n == 9 ? 784 : 0);
    const int mbit = conv1_mask_bit(n);  // this lane's channel in the ReLU mask words
    const bf16* dp2img = c.dp2 + img * kP2Pix * kC2;
    const uint8_t* maskimg = c.pmask + img * kP2Pix * kC2;

    for (int chunk = chunk0; chunk < chunk1; ++chunk) {
        const int mb0 = chunk * 8;
        const int nmb = chunk < 2 ? 8 : 6;
        const int pix0 = mb0 * 32;
        const int pixl = min(pix0 + nmb * 32, kA1Pix) - 1;
        const int ymin = pix0 / 26, ymax = pixl / 26;
        const int oy_base = ymin - 2;
        const int nrows = ymax - ymin + 3;
        __syncthreads();
        // One item = one pooled pixel x 8 channels: a single load of the pooled
        // gradient and the arg-max codes feeds the four positions of its 2x2 window.
        // Pooled rows/columns one step outside the image produce the zero halo.
        // Loads are unconditional (clamped coordinates) so that the unrolled
        // iterations keep several in flight.
        const int py_base = ((oy_base + 2) >> 1) - 1;  // floor(oy_base / 2), oy_base >= -2
        const int nprow = ((oy_base + nrows + 1) >> 1) - py_base;
        const int nitems = nprow * 14 * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int it = tid + i * 256;
            const int itc = min(it, nitems - 1);
            const int part = itc & 7, pi = itc >> 3;
            const int prow = pi / 14, pcol = pi - prow * 14;
            const int py = py_base + prow, px = pcol - 1;
            const bool inside = py >= 0 && py < 12 && px >= 0 && px < 12;
            const int pooled = min(max(py, 0), 11) * 12 + min(max(px, 0), 11);
            uint4 v = *reinterpret_cast<const uint4*>(dp2img + pooled * kC2 + part * 8);
            const uint2 m = *reinterpret_cast<const uint2*>(maskimg + pooled * kC2 + part * 8);
            if (!inside) v = make_uint4(0u, 0u, 0u, 0u);
            const PoolWindow win(v, m, lut);
            // window rows 2 * py and 2 * py + 1 of the image, relative to the patch
            const int pr = 2 * py - oy_base;
            char* const dst = smem + nfix::mnist_dg_off(pr, 2 * px + 2, part);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int prq = pr + (q >> 1);
                if (it < nitems && prq >= 0 && prq < nrows)
                    *reinterpret_cast<uint4*>(dst + (q >> 1) * kDgRowPitch + (q & 1) * kDgPixPitch) =
                        win.at(q);
            }
        }
        __syncthreads();

#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
            const int mb = mb0 + wave * 2 + i;
            if (mb >= mb0 + nmb) break;  // wave-uniform
            const int p = min(mb * 32 + n, kA1Pix - 1);
            const int y = p / 26, x = p - y * 26;
            // tap (kh, kw) reads patch pixel (y - oy_base - kh, x + 2 - kw): the address of
            // tap (2, 2) plus compile-time offsets
            const char* const afrag = smem + nfix::mnist_dg_off(y - oy_base - 2, x, h);
            f32x16 acc = {0};
#pragma unroll
            for (int ks = 0; ks < 36; ++ks) {
                const int tap = ks >> 2, kh = tap / 3, kw = tap - kh * 3;
                const bf16x8 A = *reinterpret_cast<const bf16x8*>(
                    afrag + (2 - kh) * kDgRowPitch + (2 - kw) * kDgPixPitch + (ks & 3) * 32);
                acc = mfma32(A, Bf[ks], acc);
            }
            // ReLU mask of conv1's output; the masked tile is the A operand
            // (k = its own row index, in accumulator order) of
            //   Z[c][t] += sum_pixels d(a1)[pixel][c] * x[pixel shifted by tap t]
            const int pb = mb * 32 + 4 * h;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                unsigned Aw[4];
                bf16x8 Bx;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j0 = 2 * jj, j1 = j0 + 1;
                    const int pp0 = pb + 16 * s + 8 * (j0 >> 2) + (j0 & 3), pp1 = pp0 + 1;
                    // bit n of a pixel's word, spread over a bf16: 0 or 0xFFFF
                    const int m0 = __builtin_amdgcn_sbfe(static_cast<int>(smask[pp0]), mbit, 1);
                    const int m1 = __builtin_amdgcn_sbfe(static_cast<int>(smask[pp1]), mbit, 1);
                    const unsigned keep = __builtin_amdgcn_perm(m1, m0, 0x05040100u);
                    const bf16x2 pair = {f2bf(acc[8 * s + j0]), f2bf(acc[8 * s + j1])};
                    Aw[jj] = __builtin_bit_cast(unsigned, pair) & keep;
                    Bx[j0] = f2bf(xs[soff[pp0] + tapoff]);
                    Bx[j1] = f2bf(xs[soff[pp1] + tapoff]);
                }
                const uint4 Au = make_uint4(Aw[0], Aw[1], Aw[2], Aw[3]);
                Z = mfma32(__builtin_bit_cast(bf16x8, Au), Bx, Z);
            }
        }
    }
    // Z: lane (t = n, half h) holds channels acc_row(r): combine the four waves
    __syncthreads();
    if (n < 10) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + acc_row(r, lane)) * 10 + n] = Z[r];
    }
    __syncthreads();
    float sq = 0.f;
    for (int i = tid; i < 320; i += 256) {
        const int ch = i / 10, t = i - ch * 10;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += red[(w * 32 + ch) * 10 + t];
        float* part = c.conv1_part + (img * c.split + chunk0) * kConv1Part;
        if (t < 9) part[ch * 9 + t] = s;
        else part[288 + ch] = s;
        sq = fmaf(s, s, sq);
    }
    if (c.dp_pass == 1) {  // this workgroup's tile IS one sample's conv1 gradient
        sq = wave_sum(sq);
        __syncthreads();
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0) c.dp_normsq[img * 4 + 1] = red[0] + red[1] + red[2] + red[3];
    }
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv2_dgrad_kernel(MnistCtx c, int idx_off) {
    conv2_dgrad_body(c, idx_off, blockIdx.x, blockIdx.y);
}

// conv2_dgrad and fc1_wgrad in one launch (they are independent once fc1_dgrad
// is done).  With one or two clients on the GPU neither fills the chip on its
// own, so the workgroups of both run side by side; the first n_dg indices of
// blockIdx.x are conv2_dgrad's.
constexpr int kPairLds = kDgLds > 2 * 64 * kTrPitch128 ? kDgLds : 2 * 64 * kTrPitch128;

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
bwd_pair_kernel(MnistCtx c, int idx_off, int step, int n_dg) {
    if (static_cast<int>(blockIdx.x) < n_dg)
        conv2_dgrad_body(c, idx_off, blockIdx.x, blockIdx.y);
    else
        fc1_wgrad_body(c, step, blockIdx.x - n_dg, blockIdx.y);
}

// ---------------------------------------------------------------------------
constexpr int kWgDaPitch = 192;  // 128 B of channels + 64 B
constexpr int kWgDaBytes = 192 * kWgDaPitch;  // 36864
constexpr int kWgA1Bytes = 260 * 64;          // 16640
constexpr int kWgLds = kWgDaBytes + kWgA1Bytes + kPoolLutBytes;

__device__ __forceinline__ void conv2_wgrad_body(const MnistCtx& c, int wg, int g) {
    if (client_idle(c, g)) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_da = smem;               // [192 px][64 oc], pitch 192
    char* s_a1 = smem + kWgDaBytes;  // [260 px][32 c], pitch 64
    char* lut = s_a1 + kWgA1Bytes;   // pooling-window masks (PoolWindow)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
    const int ocb = wave & 1, tg = wave >> 1;
    const int nit = 3 * c.B / c.WPC;  // image-bands per workgroup
    const int it0 = wg * nit;
    f32x16 acc[3] = {{0}, {0}, {0}};
    // per-lane pieces of the tap-shifted transposed read (see load_tr_frag)
    const int li = lane & 15;
    const int lq = li >> 2;
    const int lcol = (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;

    // A band of 8 output rows is 4 x 12 pooled pixels x 8 channel groups = 384
    // items, one per thread: the thread loads its 16 B of pooled gradient and 8 B
    // of arg-max codes (contiguous over the workgroup) and writes the four
    // positions of the 2x2 window.  The loads for the next band are issued before
    // the MFMA loop of the current one and held in named registers.
    const int wpi = tid >> 3, wprow = wpi / 12, wpcol = wpi - wprow * 12;
    char* const da_dst = s_da + (2 * wprow * 24 + 2 * wpcol) * kWgDaPitch + (tid & 7) * 16;
    const int q2 = min(tid + 768, 1039);
    const bool has2 = tid + 768 < 1040;
    uint4 pv, pa0, pa1, pa2;
    uint2 pm;
#define NF_WG_ISSUE(it_)                                                                \
    {                                                                                   \
        const int gi_ = it0 + (it_);                                                    \
        const int im_ = gi_ / 3, band_ = gi_ - im_ * 3;                                 \
        const long img_ = static_cast<long>(g) * c.B + im_;                             \
        const long off_ = (img_ * kP2Pix + band_ * 48) * kC2 + tid * 8;                 \
        pv = *reinterpret_cast<const uint4*>(c.dp2 + off_);                             \
        pm = *reinterpret_cast<const uint2*>(c.pmask + off_);                           \
        const uint4* sa1_ = reinterpret_cast<const uint4*>(                             \
            c.a1 + (img_ * kA1Pix + band_ * 8 * 26) * kA1C);                            \
        pa0 = sa1_[tid]; pa1 = sa1_[tid + 384]; pa2 = sa1_[q2];                         \
    }
    NF_WG_ISSUE(0)
    pool_lut_init(lut, tid);
    for (int it = 0; it < nit; ++it) {
        __syncthreads();
        {
            const PoolWindow win(pv, pm, lut);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint4*>(da_dst + ((q >> 1) * 24 + (q & 1)) * kWgDaPitch) = win.at(q);
        }
        *reinterpret_cast<uint4*>(s_a1 + tid * 16) = pa0;
        *reinterpret_cast<uint4*>(s_a1 + (tid + 384) * 16) = pa1;
        if (has2) *reinterpret_cast<uint4*>(s_a1 + q2 * 16) = pa2;
        if (it + 1 < nit) NF_WG_ISSUE(it + 1)
        __syncthreads();
#undef NF_WG_ISSUE
#pragma unroll 2
        for (int ks = 0; ks < 12; ++ks) {
            const bf16x8 Afr = load_tr_frag(s_da, kWgDaPitch, ks * 16, ocb * 32, lane);
            // the 8 pixels of this lane half: groups gi0, gi0 + 1 of 4 pixels
            const int gi0 = 4 * ks + 2 * h;
            const int oy0 = gi0 / 6, ox0 = 4 * (gi0 - oy0 * 6);
            const int gi1 = gi0 + 1;
            const int oy1 = gi1 / 6, ox1 = 4 * (gi1 - oy1 * 6);
#pragma unroll
            for (int t = 0; t < 3; ++t) {  // tap (kh, kw) = (tg, t)
                const bf16x4 lo = lds_tr4(s_a1 + ((oy0 + tg) * 26 + ox0 + t + lq) * 64 + lcol);
                const bf16x4 hi = lds_tr4(s_a1 + ((oy1 + tg) * 26 + ox1 + t + lq) * 64 + lcol);
                const bf16x8 Bfr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                acc[t] = mfma32(Afr, Bfr, acc[t]);
            }
        }
    }
    if (c.dp_pass == 1) {  // ipw == 1: the accumulators are one sample's conv2.weight gradient
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sq = fmaf(acc[t][r], acc[t][r], sq);
        sq = wave_sum(sq);
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int w = 0; w < 6; ++w) tot += red[w];
            c.dp_normsq[(static_cast<long>(g) * c.B + wg) * 4 + 2] = tot;
        }
        // the tile itself is kept: the clipped gradient is the clip-factor weighted sum
        // of these per-sample tiles (finalize), not a second convolution pass
    }
    float* part = c.conv2_part + (static_cast<long>(g) * c.WPC + wg) * kConv2Part;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int tap = tg * 3 + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oc = ocb * 32 + acc_row(r, lane);
            part[(oc * 9 + tap) * 32 + n] = acc[t][r];
        }
    }
}

__global__ void __launch_bounds__(384) conv2_wgrad_kernel(MnistCtx c) {
    conv2_wgrad_body(c, blockIdx.x, blockIdx.y);
}

// The same for a workgroup of four waves (the triple launch below: with 256-thread
// workgroups two of them share a CU, with 384-thread ones only one fits).  Waves: output
// channel half x tap group, taps 0-4 and 5-8; every tap's accumulator adds the same products
// in the same order as in the six-wave body, so the results have the same bits.  Staging: the
// 384 pooled-gradient items of a band are items tid and tid + 256, the 1040 pieces of the
// conv1 activation band pieces tid + 256 i.
__device__ __forceinline__ void conv2_wgrad_body4(const MnistCtx& c, int wg, int g) {
    if (client_idle(c, g)) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_da = smem;               // [192 px][64 oc], pitch 192
    char* s_a1 = smem + kWgDaBytes;  // [260 px][32 c], pitch 64
    char* lut = s_a1 + kWgA1Bytes;   // pooling-window masks (PoolWindow)
    const int tid = threadIdx.x;
    const int lane = tid & 63, n = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ocb = wave & 1, tg = wave >> 1;
    const int tap0 = tg * 5, ntap = tg == 0 ? 5 : 4;
    const int nit = 3 * c.B / c.WPC;  // image-bands per workgroup
    const int it0 = wg * nit;
    f32x16 acc[5] = {{0}, {0}, {0}, {0}, {0}};
    const int li = lane & 15;
    const int lq = li >> 2;
    const int lcol = (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;
    // byte offset of tap (kh, kw) in the activation band: (kh * 26 + kw) pixels
    int toff[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int t9 = min(tap0 + t, 8);
        toff[t] = ((t9 / 3) * 26 + t9 % 3) * 64;
    }

    const bool item2 = tid < 128;  // items tid + 256 exist for the first 128 threads
    const int i2 = item2 ? tid + 256 : tid;
    auto da_of = [&](int item) {
        const int wpi = item >> 3, wprow = wpi / 12, wpcol = wpi - wprow * 12;
        return s_da + (2 * wprow * 24 + 2 * wpcol) * kWgDaPitch + (item & 7) * 16;
    };
    char* const da_dst0 = da_of(tid);
    char* const da_dst1 = da_of(i2);
    const bool has4 = tid + 1024 < 1040;
    const int q4 = min(tid + 1024, 1039);
    uint4 pv0, pv1, pa0, pa1, pa2, pa3, pa4;
    uint2 pm0, pm1;
#define NF_WG4_ISSUE(it_)                                                               \
    {                                                                                   \
        const int gi_ = it0 + (it_);                                                    \
        const int im_ = gi_ / 3, band_ = gi_ - im_ * 3;                                 \
        const long img_ = static_cast<long>(g) * c.B + im_;                             \
        const long off_ = (img_ * kP2Pix + band_ * 48) * kC2;                           \
        pv0 = *reinterpret_cast<const uint4*>(c.dp2 + off_ + tid * 8);                  \
        pm0 = *reinterpret_cast<const uint2*>(c.pmask + off_ + tid * 8);                \
        pv1 = *reinterpret_cast<const uint4*>(c.dp2 + off_ + i2 * 8);                   \
        pm1 = *reinterpret_cast<const uint2*>(c.pmask + off_ + i2 * 8);                 \
        const uint4* sa1_ = reinterpret_cast<const uint4*>(                             \
            c.a1 + (img_ * kA1Pix + band_ * 8 * 26) * kA1C);                            \
        pa0 = sa1_[tid]; pa1 = sa1_[tid + 256]; pa2 = sa1_[tid + 512];                  \
        pa3 = sa1_[tid + 768]; pa4 = sa1_[q4];                                          \
    }
    NF_WG4_ISSUE(0)
    pool_lut_init(lut, tid);
    for (int it = 0; it < nit; ++it) {
        __syncthreads();
        {
            const PoolWindow win(pv0, pm0, lut);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint4*>(da_dst0 + ((q >> 1) * 24 + (q & 1)) * kWgDaPitch) = win.at(q);
        }
        if (item2) {
            const PoolWindow win(pv1, pm1, lut);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint4*>(da_dst1 + ((q >> 1) * 24 + (q & 1)) * kWgDaPitch) = win.at(q);
        }
        *reinterpret_cast<uint4*>(s_a1 + tid * 16) = pa0;
        *reinterpret_cast<uint4*>(s_a1 + (tid + 256) * 16) = pa1;
        *reinterpret_cast<uint4*>(s_a1 + (tid + 512) * 16) = pa2;
        *reinterpret_cast<uint4*>(s_a1 + (tid + 768) * 16) = pa3;
        if (has4) *reinterpret_cast<uint4*>(s_a1 + q4 * 16) = pa4;
        if (it + 1 < nit) NF_WG4_ISSUE(it + 1)
        __syncthreads();
#undef NF_WG4_ISSUE
#pragma unroll 2
        for (int ks = 0; ks < 12; ++ks) {
            const bf16x8 Afr = load_tr_frag(s_da, kWgDaPitch, ks * 16, ocb * 32, lane);
            const int gi0 = 4 * ks + 2 * h;
            const int oy0 = gi0 / 6, ox0 = 4 * (gi0 - oy0 * 6);
            const int gi1 = gi0 + 1;
            const int oy1 = gi1 / 6, ox1 = 4 * (gi1 - oy1 * 6);
            const char* b0 = s_a1 + (oy0 * 26 + ox0 + lq) * 64 + lcol;
            const char* b1 = s_a1 + (oy1 * 26 + ox1 + lq) * 64 + lcol;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                if (t < ntap) {
                    const bf16x4 lo = lds_tr4(b0 + toff[t]);
                    const bf16x4 hi = lds_tr4(b1 + toff[t]);
                    const bf16x8 Bfr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[t] = mfma32(Afr, Bfr, acc[t]);
                }
            }
        }
    }
    if (c.dp_pass == 1) {  // one image per workgroup: the accumulators are one sample's gradient
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t)
            if (t < ntap) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sq = fmaf(acc[t][r], acc[t][r], sq);
            }
        sq = wave_sum(sq);
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        if (lane == 0) red[wave] = sq;
        __syncthreads();
        if (tid == 0)
            c.dp_normsq[(static_cast<long>(g) * c.B + wg) * 4 + 2] = (red[0] + red[1]) + (red[2] + red[3]);
        // (the tile itself is kept: finalize forms the clipped gradient from these tiles)
    }
    float* part = c.conv2_part + (static_cast<long>(g) * c.WPC + wg) * kConv2Part;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        if (t < ntap) {
            const int tap = tap0 + t;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oc = ocb * 32 + acc_row(r, lane);
                part[(oc * 9 + tap) * 32 + n] = acc[t][r];
            }
        }
    }
}

// conv2_dgrad, fc1_wgrad and conv2_wgrad in one launch: all three only need what
// fc1_dgrad left behind.  Four-wave workgroups throughout (conv2_wgrad in its four-wave
// form), so that two workgroups share a CU and the launch of one client - 192 + 72 + 192
// workgroups - is resident at once (with the six-wave body the launch had 384-thread
// workgroups, one per CU, and ran in two rounds).  The DP norm pass keeps the six-wave
// kernel.  Workgroup indices: n_dg of conv2_dgrad, then n_fw of fc1_wgrad, then conv2_wgrad's.
constexpr int kTripleLds = kPairLds > kWgLds ? kPairLds : kWgLds;

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
bwd_triple_kernel(MnistCtx c, int idx_off, int step, int n_dg, int n_fw) {
    const int bx = blockIdx.x;
    if (bx < n_dg)
        conv2_dgrad_body(c, idx_off, bx, blockIdx.y);
    else if (bx < n_dg + n_fw)
        fc1_wgrad_body(c, step, bx - n_dg, blockIdx.y);
    else
        conv2_wgrad_body4(c, bx - n_dg - n_fw, blockIdx.y);
}

// The norm pass of DP-SGD: conv2_dgrad and conv2_wgrad (a workgroup per image each) in one
// launch of four-wave workgroups.
constexpr int kDpPairLds = kDgLds > kWgLds ? kDgLds : kWgLds;

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
dp_pair_kernel(MnistCtx c, int idx_off, int n_dg) {
    const int bx = blockIdx.x;
    if (bx < n_dg)
        conv2_dgrad_body(c, idx_off, bx, blockIdx.y);
    else
        conv2_wgrad_body4(c, bx - n_dg, blockIdx.y);
}

// The clipped pass of DP-SGD: the head (fc2 gradients, metrics, fc1 bias gradient from the
// clipped rows) and fc1_wgrad in one launch.  Both only need the clip factors: fc1_wgrad reads
// the norm pass's d(h1) and scales its rows itself, so the head does not write d(h1) here.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
dp_clip_kernel(MnistCtx c, int idx_off, int step, int n_head) {
    const int bx = blockIdx.x;
    if (bx < n_head)
        head_body(c, idx_off, step, bx, blockIdx.y, false);
    else
        fc1_wgrad_body(c, step, bx - n_head, blockIdx.y, true);
}

// ---------------------------------------------------------------------------
// DP-SGD, between the norm pass and the clipped pass: one workgroup per sample
// adds the analytic parts (fc1: ||dh1||^2 (||p2||^2 + 1); conv2.bias: per-channel
// pixel sums of dp2) to the tile norms and writes the clip factor
//   min(1, C / ||g_b||),  g_b = n * (gradient of the mean loss w.r.t. sample b).
__global__ void __launch_bounds__(256) dp_scale_kernel(MnistCtx c, int step) {
    if (client_idle(c, blockIdx.y)) return;
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ float red[3][4];
    __shared__ float chs[4][64];
    const long bi = static_cast<long>(g) * c.B + b;
    float p2sq = 0.f;
    const bf16* p2 = c.p2 + bi * kFc1K;
    for (int i = tid; i < kFc1K / 8; i += 256) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p2 + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = bf2f(v[e]); p2sq = fmaf(f, f, p2sq); }
    }
    float dhsq = 0.f;
    if (tid < 128) { const float f = bf2f(c.dh1[bi * 128 + tid]); dhsq = f * f; }
    // conv2 bias: thread (strip = tid / 64, oc = tid % 64) sums 36 pooled pixels
    {
        const int oc = tid & 63, strip = tid >> 6;
        const bf16* d = c.dp2 + bi * kP2Pix * kC2 + oc;
        float sacc = 0.f;
        for (int px = strip * 36; px < strip * 36 + 36; ++px) sacc += bf2f(d[px * kC2]);
        chs[strip][oc] = sacc;
    }
    p2sq = wave_sum(p2sq);
    dhsq = wave_sum(dhsq);
    __syncthreads();
    float cbsq = 0.f;
    if (tid < 64) {
        const float t = chs[0][tid] + chs[1][tid] + chs[2][tid] + chs[3][tid];
        cbsq = t * t;
        c.dp_c2b[bi * 64 + tid] = t;
    }
    cbsq = wave_sum(cbsq);
    if (lane == 0) { red[0][wave] = p2sq; red[1][wave] = dhsq; red[2][wave] = cbsq; }
    __syncthreads();
    if (tid == 0) {
        const float P = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const float D = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const float Cb = red[2][0];
        const float* ns = c.dp_normsq + bi * 4;
        const float total = ns[0] + ns[1] + ns[2] + D * (P + 1.f) + Cb;
        const int nv = c.nvalid[g * c.nvalid_stride + step];
        const float norm = static_cast<float>(nv) * sqrtf(fmaxf(total, 0.f));
        c.dp_rowscale[bi] = fminf(1.f, c.dp_clip / (norm + 1e-6f));
        // The step's last launch (finalize) draws noise by the step counter in all of its
        // workgroups and advances it from one of them: they read this copy instead, taken
        // while nothing writes the counter.
        if (b < 2) {
            const unsigned long long ctr = c.step_counter[g];
            reinterpret_cast<unsigned*>(c.dp_normsq)[bi * 4 + 3] =
                static_cast<unsigned>(b == 0 ? ctr : ctr >> 32);
        }
    }
}

// ---------------------------------------------------------------------------
constexpr int kSmallTotal = kConv1Part + kConv2Part + 64 + kHeadPart;

// Sum `count` slabs of one element: wave w of the block adds slabs
// [w*count/4, (w+1)*count/4) in order, the four partial sums meet in LDS and are
// added in a fixed order, so the result does not depend on timing.
__device__ __forceinline__ float sum_slabs(const float* p, int count, long stride, int wave) {
    const int lo = (count * wave) >> 2, hi = (count * (wave + 1)) >> 2;
    float s0 = 0.f, s1 = 0.f;
    int i = lo;
    // 16 loads in flight per lane (the kernel is a few round trips to memory, nothing else);
    // the order of the additions is that of the two-at-a-time loop below
    for (; i + 15 < hi; i += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = p[(i + k) * stride];
#pragma unroll
        for (int k = 0; k < 16; k += 2) { s0 += v[k]; s1 += v[k + 1]; }
    }
#pragma unroll 4
    for (; i + 1 < hi; i += 2) {
        s0 += p[(i + 0) * stride];
        s1 += p[(i + 1) * stride];
    }
    if (i < hi) s0 += p[i * stride];
    return s0 + s1;
}

// The same with a weight per slab (DP-SGD: slab b is sample b's gradient tile, the
// weight its clip factor).
__device__ __forceinline__ float sum_slabs_weighted(const float* p, const float* w, int count,
                                                    long stride, int wave) {
    const int lo = (count * wave) >> 2, hi = (count * (wave + 1)) >> 2;
    float s0 = 0.f, s1 = 0.f;
    int i = lo;
#pragma unroll 4
    for (; i + 1 < hi; i += 2) {
        s0 = fmaf(w[i + 0], p[(i + 0) * stride], s0);
        s1 = fmaf(w[i + 1], p[(i + 1) * stride], s1);
    }
    if (i < hi) s0 = fmaf(w[i], p[i * stride], s0);
    return s0 + s1;
}

// 64 elements per workgroup, the slabs of each split over the four waves.
// kSgd: also apply the SGD update of the element (every parameter except
// fc1.weight is covered by exactly one thread), refresh its bf16 copies and
// advance the client's step counter - the whole tail of a step whose
// fc1.weight update already happened in fc1_wgrad.
template <bool kSgd>
__global__ void __launch_bounds__(256) finalize_kernel(MnistCtx c, int step) {
    if (client_idle(c, blockIdx.y)) return;
    __shared__ float part[4][64];
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    float* gr = c.grad + g * c.grad_stride;
    const int nv = c.nvalid[g * c.nvalid_stride + step];
    int dst = -1;      // arena offset of the parameter this thread owns
    int w2t_at = -1;   // conv2.weight only: position in the [tap][c][oc] copy
    int metric = -1;
    const float* src = nullptr;
    int count = 0;
    long stride = 0;
    // DP-SGD, clipped pass: the convolution tiles and the conv2.bias rows are per
    // sample and unclipped (written by the norm pass); weight them here
    const bool dp_weighted = c.dp_pass == 2;
    bool weighted = false;
    if (e < kConv1Part) {
        count = c.B * c.split;
        stride = kConv1Part;
        src = c.conv1_part + static_cast<long>(g) * count * kConv1Part + e;
        dst = e < 288 ? c.off_c1w + e : c.off_c1b + e - 288;
        weighted = dp_weighted;
    } else if (e < kConv1Part + kConv2Part) {
        const int e2 = e - kConv1Part;
        count = c.WPC;
        stride = kConv2Part;
        src = c.conv2_part + static_cast<long>(g) * c.WPC * kConv2Part + e2;
        dst = c.off_c2w + e2;
        weighted = dp_weighted;
        const int oc = e2 / 288;
        w2t_at = (e2 - oc * 288) * 64 + oc;
    } else if (e < kConv1Part + kConv2Part + 64) {
        const int e2 = e - kConv1Part - kConv2Part;
        count = 2 * kFc1Tiles;
        stride = 64;
        src = c.c2b_part + static_cast<long>(g) * count * 64 + e2;
        if (dp_weighted) {
            count = c.B;
            src = c.dp_c2b + static_cast<long>(g) * c.B * 64 + e2;
            weighted = true;
        }
        dst = c.off_c2b + e2;
    } else if (e < kSmallTotal) {
        const int e3 = e - kConv1Part - kConv2Part - 64;
        count = c.B / kHeadRows;
        stride = kHeadPart;
        src = c.head_part + static_cast<long>(g) * count * kHeadPart + e3;
        if (e3 < 1280) dst = c.off_f2w + e3;
        else if (e3 < 1290) dst = c.off_f2b + e3 - 1280;
        else if (e3 >= 1296 && e3 < 1424) dst = c.off_f1b + e3 - 1296;
        else if (e3 == 1424 || e3 == 1425) metric = e3 - 1424;
    }
    float* p = c.arena + g * c.arena_stride + (dst >= 0 ? dst : 0);
    const float pv = (kSgd && wave == 0) ? *p : 0.f;  // in flight during the sums
    float s = 0.f;
    if (src != nullptr)
        s = weighted ? sum_slabs_weighted(src, c.dp_rowscale + static_cast<long>(g) * c.B, count,
                                          stride, wave)
                     : sum_slabs(src, count, stride, wave);
    part[wave][lane] = s;
    __syncthreads();
    if (wave != 0) return;
    s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (metric >= 0 && nv > 0) {
        // mean over the whole batch; a batch is counted once, with its first micro-batch
        const int nt = c.ntotal != nullptr ? c.ntotal[g * c.nvalid_stride + step] : nv;
        float* m = c.metrics + g * 4;
        if (metric == 0) { m[0] += s / nt; if (!c.accumulate) m[2] += 1.0f; }
        else m[1] += s / nt;
    }
    if (dst >= 0) gr[dst] = (!kSgd && c.accumulate) ? gr[dst] + s : s;
    if (kSgd && nv > 0) {
        if (dst >= 0) {
            if (c.dp_pass == 2 && c.dp_noise_std > 0.f) {
                // the element's share of the noise block sgd_pack would draw for it
                // (the counter as dp_scale copied it: another workgroup of this launch
                // advances the counter itself)
                float z[4];
                const unsigned* aux =
                    reinterpret_cast<const unsigned*>(c.dp_normsq) + static_cast<long>(g) * c.B * 4;
                const unsigned long long ctr =
                    aux[3] | (static_cast<unsigned long long>(aux[7]) << 32);
                const unsigned long long nctr = 2ull * ctr + 0x8000000000000000ull;
                nf::philox_samples(client_seed(c, g), nctr,
                                   static_cast<unsigned long long>(dst >> 2), 1, z);
                s = fmaf(c.dp_noise_std / static_cast<float>(nv), z[dst & 3], s);
            }
            const float v = pv - c.lr * fmaf(c.wd, pv, s);
            *p = v;
            const bf16 h = f2bf(v);
            c.shadow[g * c.shadow_stride + dst] = h;
            if (w2t_at >= 0) c.w2t[g * c.w2t_stride + w2t_at] = h;
        }
        // every reader of this step's counter ran in an earlier launch (the DP noise above
        // is keyed by the copy dp_scale took)
        if (e == 0) c.step_counter[g] += 1ull;
    }
}

// ---------------------------------------------------------------------------
// force = 1: refresh the bf16 copies only (lr is ignored, nothing is skipped,
// the step counter stays).
__global__ void __launch_bounds__(256) sgd_pack_kernel(MnistCtx c, int step, int force) {
    const int g = blockIdx.y;
    float lr = c.lr, wd = c.wd;
    if (force) { lr = 0.f; wd = 0.f; }
    else if (c.nvalid[g * c.nvalid_stride + step] == 0) return;  // client has no batch left
    float* p = c.arena + g * c.arena_stride;
    const float* gr = c.grad + g * c.grad_stride;
    bf16* sh = c.shadow + g * c.shadow_stride;
    bf16* w2t = c.w2t + g * c.w2t_stride;
    const long n4 = c.n_arena >> 2;
    const bool skip_fc1 = !force && c.fuse_sgd;  // fc1.weight was updated by fc1_wgrad
    // DP-SGD: Gaussian noise of std sigma*C on the SUM of clipped gradients,
    // i.e. sigma*C/n on the mean that the gradient arena holds
    const bool noisy = !force && c.dp_pass == 2 && c.dp_noise_std > 0.f;
    float nstd = 0.f;
    unsigned long long nctr = 0ull;
    if (noisy) {
        nstd = c.dp_noise_std / static_cast<float>(c.nvalid[g * c.nvalid_stride + step]);
        nctr = 2ull * c.step_counter[g] + 0x8000000000000000ull;
    }
    const long f1_lo = c.off_f1w >> 2, f1_hi = (c.off_f1w + 128L * kFc1K) >> 2;
    for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n4;
         i += static_cast<long>(gridDim.x) * 256) {
        if (skip_fc1 && i >= f1_lo && i < f1_hi) {
            // jump over the segment in whole grid strides
            const long stride = static_cast<long>(gridDim.x) * 256;
            i += ((f1_hi - i + stride - 1) / stride - 1) * stride;
            continue;
        }
        float4 pv = reinterpret_cast<float4*>(p)[i];
        if (!force) {
            float4 gv = reinterpret_cast<const float4*>(gr)[i];
            if (noisy) {
                float z[4];
                nf::philox_samples(client_seed(c, g), nctr, static_cast<unsigned long long>(i), 1, z);
                gv.x = fmaf(nstd, z[0], gv.x); gv.y = fmaf(nstd, z[1], gv.y);
                gv.z = fmaf(nstd, z[2], gv.z); gv.w = fmaf(nstd, z[3], gv.w);
            }
            pv.x -= lr * fmaf(wd, pv.x, gv.x);
            pv.y -= lr * fmaf(wd, pv.y, gv.y);
            pv.z -= lr * fmaf(wd, pv.z, gv.z);
            pv.w -= lr * fmaf(wd, pv.w, gv.w);
            reinterpret_cast<float4*>(p)[i] = pv;
        }
        bf16x4 hv;
        hv[0] = f2bf(pv.x); hv[1] = f2bf(pv.y); hv[2] = f2bf(pv.z); hv[3] = f2bf(pv.w);
        reinterpret_cast<bf16x4*>(sh)[i] = hv;
        const long base = i << 2;
        if (base >= c.off_c2w && base < c.off_c2w + 18432) {
            const int e = static_cast<int>(base - c.off_c2w);  // 4 consecutive c of one (oc, tap)
            const int oc = e / 288, r = e - oc * 288;
#pragma unroll
            for (int q = 0; q < 4; ++q) w2t[(r + q) * 64 + oc] = hv[q];
        }
    }
}

// A client's counter advances only when it really took a step, so its dropout
// and noise streams do not depend on which other clients share the GPU.
__global__ void bump_step_kernel(MnistCtx c, int step) {
    const int g = threadIdx.x;
    if (g < c.G && c.nvalid[g * c.nvalid_stride + step] > 0) c.step_counter[g] += 1ull;
}

inline int check_ctx(const MnistCtx* c) {
    if (c->B != 64) return -2;
    if (c->G < 1 || c->G > 1024 || c->S != kFc1S) return -3;
    if (c->WPC < 1 || (3 * c->B) % c->WPC != 0) return -4;
    if (c->split != 1 && c->split != 3) return -7;
    if (c->conv1_split != 1 && c->conv1_split != 2 && c->conv1_split != 4) return -8;
    if (c->n_arena % 64 != 0) return -5;
    if (c->micro < 1 || (c->micro > 1 && (c->fuse_sgd || c->dp_pass != 0 || c->ntotal == nullptr)))
        return -13;
    if (c->dp_pass != 0 &&
        (c->WPC != c->B || c->split != 1 || c->dp_normsq == nullptr ||
         c->dp_rowscale == nullptr || c->dp_c2b == nullptr))
        return -6;
    return 0;
}

}  // namespace

#define NF_LAUNCH_OK() (static_cast<int>(hipGetLastError()))

extern "C" {

int nf_mnist_ctx_size() { return static_cast<int>(sizeof(MnistCtx)); }

// stage: 0 conv1_fwd, 1 conv2_fwd, 2 fc1_fwd, 3 head, 4 fc1_dgrad, 5 fc1_wgrad,
//        6 conv2_dgrad, 7 conv2_wgrad, 8 finalize, 9 sgd_pack, 10 pack only,
//        11 dp_scale, 12 finalize + SGD of everything but fc1.weight (fuse_sgd only),
//        13 conv2_dgrad + fc1_wgrad in one launch, 14 advance the step counters,
//        15 sgd_pack without advancing the counters,
//        16 conv2_dgrad + fc1_wgrad + conv2_wgrad in one launch, 17 conv1 + conv2 forward in one launch
int nf_mnist_stage(const MnistCtx* c, int stage, int idx_off, int step, hipStream_t s) {
    const int rc = check_ctx(c);
    if (rc) return rc;
    MnistCtx with_step = *c;
    with_step.cur_step = step;
    c = &with_step;
    const dim3 img_grid(c->B, c->G);
    const dim3 split_grid(c->B * c->split, c->G);
    switch (stage) {
    case 0: hipLaunchKernelGGL(conv1_fwd_kernel, dim3(c->B * c->conv1_split, c->G), dim3(256), 0, s, *c, idx_off); break;
    case 1: hipLaunchKernelGGL(conv2_fwd_kernel, split_grid, dim3(256), kFwdLds, s, *c); break;
    case 17: hipLaunchKernelGGL(conv12_fwd_kernel, split_grid, dim3(256), kFusedLds, s, *c, idx_off); break;
    case 2: {
        static const hipError_t lds_ok = hipFuncSetAttribute(
            reinterpret_cast<const void*>(fc1_fwd_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, kFc1FwdLds);
        if (lds_ok != hipSuccess) return -20;
        hipLaunchKernelGGL(fc1_fwd_kernel, dim3(2 * c->S, c->G), dim3(256), kFc1FwdLds, s, *c);
        break;
    }
    case 3: hipLaunchKernelGGL(head_kernel, dim3(c->B / kHeadRows, c->G), dim3(256), 0, s, *c, idx_off, step); break;
    case 4: hipLaunchKernelGGL(fc1_dgrad_kernel, dim3(kFc1Tiles, c->G), dim3(256), 128 * kTrPitch128, s, *c); break;
    case 5: hipLaunchKernelGGL(fc1_wgrad_kernel, dim3(kFc1Tiles, c->G), dim3(256), 2 * 64 * kTrPitch128, s, *c, step); break;
    case 6: {
        static const hipError_t lds_ok = hipFuncSetAttribute(
            reinterpret_cast<const void*>(conv2_dgrad_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, kDgLds);
        if (lds_ok != hipSuccess) return -20;
        hipLaunchKernelGGL(conv2_dgrad_kernel, split_grid, dim3(256), kDgLds, s, *c, idx_off);
        break;
    }
    case 7: hipLaunchKernelGGL(conv2_wgrad_kernel, dim3(c->WPC, c->G), dim3(384), kWgLds, s, *c); break;
    case 8: hipLaunchKernelGGL(finalize_kernel<false>, dim3((kSmallTotal + 63) / 64, c->G), dim3(256), 0, s, *c, step); break;
    case 13: {
        // conv2_dgrad + fc1_wgrad side by side (plain SGD only)
        if (c->dp_pass != 0) return -12;
        static const hipError_t lds_ok = hipFuncSetAttribute(
            reinterpret_cast<const void*>(bwd_pair_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, kPairLds);
        if (lds_ok != hipSuccess) return -20;
        const int n_dg = c->B * c->split;
        hipLaunchKernelGGL(bwd_pair_kernel, dim3(n_dg + kFc1Tiles, c->G), dim3(256), kPairLds, s,
                           *c, idx_off, step, n_dg);
        break;
    }
    case 16: {
        // conv2_dgrad + fc1_wgrad + conv2_wgrad side by side (plain SGD only)
        if (c->dp_pass != 0) return -12;
        static const hipError_t lds_ok = hipFuncSetAttribute(
            reinterpret_cast<const void*>(bwd_triple_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, kTripleLds);
        if (lds_ok != hipSuccess) return -20;
        const int n_dg = c->B * c->split;
        hipLaunchKernelGGL(bwd_triple_kernel, dim3(n_dg + kFc1Tiles + c->WPC, c->G), dim3(256),
                           kTripleLds, s, *c, idx_off, step, n_dg, kFc1Tiles);
        break;
    }
    case 18: {
        if (c->dp_pass != 1) return -12;
        static const hipError_t lds_ok = hipFuncSetAttribute(
            reinterpret_cast<const void*>(dp_pair_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, kDpPairLds);
        if (lds_ok != hipSuccess) return -20;
        const int n_dg = c->B * c->split;
        hipLaunchKernelGGL(dp_pair_kernel, dim3(n_dg + c->WPC, c->G), dim3(256), kDpPairLds, s, *c,
                           idx_off, n_dg);
        break;
    }
    case 19: {
        if (c->dp_pass != 2 || !c->fuse_sgd) return -12;
        const int n_head = c->B / kHeadRows;
        hipLaunchKernelGGL(dp_clip_kernel, dim3(n_head + kFc1Tiles, c->G), dim3(256),
                           2 * 64 * kTrPitch128, s, *c, idx_off, step, n_head);
        break;
    }
    case 12:
        if (!c->fuse_sgd || c->dp_pass == 1) return -12;
        hipLaunchKernelGGL(finalize_kernel<true>, dim3((kSmallTotal + 63) / 64, c->G), dim3(256), 0, s, *c, step);
        break;
    case 9:
        hipLaunchKernelGGL(sgd_pack_kernel, dim3(128, c->G), dim3(256), 0, s, *c, step, 0);
        hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(1024), 0, s, *c, step);
        break;
    case 14: hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(1024), 0, s, *c, step); break;
    case 15: hipLaunchKernelGGL(sgd_pack_kernel, dim3(128, c->G), dim3(256), 0, s, *c, step, 0); break;
    case 11: hipLaunchKernelGGL(dp_scale_kernel, img_grid, dim3(256), 0, s, *c, step); break;
    case 10: hipLaunchKernelGGL(sgd_pack_kernel, dim3(128, c->G), dim3(256), 0, s, *c, step, 1); break;
    default: return -10;
    }
    return NF_LAUNCH_OK();
}

// One full optimisation step for all G clients.  With DP-SGD enabled
// (ctx->dp_pass != 0) the backward runs twice: a norm pass that measures every
// sample's gradient norm (conv tiles are per image; fc layers analytically),
// then - after dp_scale turned the norms into clip factors - the head and the fc1
// weight gradient run again with d(logits) rows scaled by those factors, while the
// convolution gradients are the factor-weighted sums of the per-sample tiles the norm
// pass left behind (finalize).  Noise is added by sgd_pack.
int nf_mnist_train_step(const MnistCtx* c, int idx_off, int step, hipStream_t s) {
    if (c->dp_pass == 0) {
        if (c->fuse_sgd && c->G <= 2) {
            // few clients: the independent backward kernels share one launch - two of
            // them, or all three when the caller says the chip has room (one client on
            // the GPU: 14.5 -> 13.8 ms per round; with 8 clients in four chains the
            // triple costs 3 %)
            static const int pair_seq[] = {17, 2, 3, 4, 13, 7, 12, -1};
            static const int triple_seq[] = {17, 2, 3, 4, 16, 12, -1, -1};
            const int* few = c->bwd_triple ? triple_seq : pair_seq;
            for (int i = 0; i < 8 && few[i] >= 0; ++i) {
                const int stage = few[i];
                const int rc = nf_mnist_stage(c, stage, idx_off, step, s);
                if (rc) return rc;
            }
            return 0;
        }
        // with fc1.weight updated inside fc1_wgrad, one launch finishes the step
        static const int fused_seq[] = {17, 2, 3, 4, 5, 6, 7, 12};
        static const int plain_seq[] = {17, 2, 3, 4, 5, 6, 7, 8, 9};
        const int* seq = c->fuse_sgd ? fused_seq : plain_seq;
        const int count = c->fuse_sgd ? 8 : 9;
        for (int i = 0; i < count; ++i) {
            const int rc = nf_mnist_stage(c, seq[i], idx_off, step, s);
            if (rc) return rc;
        }
        return 0;
    }
    MnistCtx p = *c;
    static const int fwd[] = {17, 2};
    static const int norm_pass_split[] = {3, 4, 6, 7, 11};
    static const int norm_pass_pair[] = {3, 4, 18, 11, -1};
    static const bool split_norm = getenv("NANOFED_MNIST_DP_PAIR") != nullptr &&
                                   getenv("NANOFED_MNIST_DP_PAIR")[0] == '0';
    const int* norm_pass = split_norm ? norm_pass_split : norm_pass_pair;
    static const int clip_unfused[] = {3, 5, 8, 9};
    static const int clip_fused_split[] = {3, 5, 12};  // SGD + noise inside fc1_wgrad / finalize
    static const int clip_fused_pair[] = {19, 12, -1};
    static const bool split_clip = getenv("NANOFED_MNIST_DP_CLIP_PAIR") != nullptr &&
                                   getenv("NANOFED_MNIST_DP_CLIP_PAIR")[0] == '0';
    const int* clip_fused = split_clip ? clip_fused_split : clip_fused_pair;
    for (int st : fwd) { const int rc = nf_mnist_stage(&p, st, idx_off, step, s); if (rc) return rc; }
    p.dp_pass = 1;
    for (int i = 0; i < 5 && norm_pass[i] >= 0; ++i) {
        const int rc = nf_mnist_stage(&p, norm_pass[i], idx_off, step, s);
        if (rc) return rc;
    }
    p.dp_pass = 2;
    if (p.fuse_sgd) {
        for (int i = 0; i < 3 && clip_fused[i] >= 0; ++i) {
            const int rc = nf_mnist_stage(&p, clip_fused[i], idx_off, step, s);
            if (rc) return rc;
        }
    } else {
        for (int st : clip_unfused) { const int rc = nf_mnist_stage(&p, st, idx_off, step, s); if (rc) return rc; }
    }
    return 0;
}

// One optimisation step made of ctx->micro micro-batches (table steps step * micro ...):
// forward and backward per micro-batch with the gradients added up in ctx->grad, the
// step counters (dropout streams) advanced after each, then one SGD update.
static int train_step_micro(const MnistCtx* c, int step, hipStream_t s) {
    MnistCtx p = *c;
    for (int mb = 0; mb < c->micro; ++mb) {
        const int ts = step * c->micro + mb;
        p.accumulate = mb > 0 ? 1 : 0;
        static const int micro_seq[] = {17, 2, 3, 4, 5, 6, 7, 8};
        for (int stage : micro_seq) {
            const int rc = nf_mnist_stage(&p, stage, ts * c->B, ts, s);
            if (rc) return rc;
        }
        const int rc = nf_mnist_stage(&p, 14, 0, ts, s);
        if (rc) return rc;
    }
    p.accumulate = 0;
    // the batch is empty exactly when its first micro-batch is
    return nf_mnist_stage(&p, 15, 0, step * c->micro, s);
}

// `steps` consecutive optimisation steps (one epoch) in one call.
int nf_mnist_train_steps(const MnistCtx* c, int first_step, int steps, hipStream_t s) {
    if (c->micro > 1) {
        for (int i = 0; i < steps; ++i) {
            const int rc = train_step_micro(c, first_step + i, s);
            if (rc) return rc;
        }
        return 0;
    }
    for (int i = 0; i < steps; ++i) {
        const int rc = nf_mnist_train_step(c, (first_step + i) * c->B, first_step + i, s);
        if (rc) return rc;
    }
    return 0;
}

// Forward + loss/accuracy only (ctx->train must be 0 and eval_out set).
int nf_mnist_eval_step(const MnistCtx* c, int idx_off, int step, hipStream_t s) {
    if (c->train != 0 || c->eval_out == nullptr) return -11;
    static const int eval_seq[] = {17, 2, 3};
    for (int stage : eval_seq) {
        const int rc = nf_mnist_stage(c, stage, idx_off, step, s);
        if (rc) return rc;
    }
    return 0;
}

int nf_mnist_pack(const MnistCtx* c, hipStream_t s) { return nf_mnist_stage(c, 10, 0, 0, s); }

}  // extern "C"
