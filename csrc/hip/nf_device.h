// Device-side helpers shared by the gfx950 kernels: MFMA fragment types,
// wave64 reductions and the LDS transposed-read fragment loader.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common/index_math.h"

namespace nfd {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NF_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))

// D = A(32x16) * B(16x32) + C, bf16 in, fp32 accumulate.
//   A fragment: lane l holds A[row l&31][k = 8*(l>>5) + j], j = 0..7
//   B fragment: lane l holds B[k = 8*(l>>5) + j][col l&31]
//   C/D:        reg r of lane l is C[row (r&3) + 8*(r>>2) + 4*(l>>5)][col l&31]
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int acc_row(int reg, int lane) { return nfix::mfma32_acc_row(reg, lane); }

// One 4x16 transposed LDS read: within each group of 16 lanes, lane 4q+p
// supplies the address of row q, columns 4p..4p+3; lane i receives column i
// of the four rows.  EXEC must be full.
__device__ __forceinline__ bf16x4 lds_tr4(const char* addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(NF_LDS(bf16x4, addr));
}

// MFMA operand fragment (A or B of mfma32) from an LDS tile stored with the
// reduction index as the slow dimension: tile[k][x], x contiguous, `pitch`
// bytes between consecutive k.  Returns for lane l the 8 values
// tile[k0 + 8*(l>>5) + j][x0 + (l&31)], j = 0..7.
// pitch and the tile base must be multiples of 8 bytes.
__device__ __forceinline__ bf16x8 load_tr_frag(const char* tile, int pitch, int k0,
                                               int x0, int lane) {
    const int i = lane & 15;
    const char* a = tile + (k0 + 8 * (lane >> 5) + (i >> 2)) * pitch +
                    (x0 + 16 * ((lane >> 4) & 1) + 4 * (i & 3)) * 2;
    const bf16x4 lo = lds_tr4(a);
    const bf16x4 hi = lds_tr4(a + 4 * pitch);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float bf2f(bf16 v) { return static_cast<float>(v); }
__device__ __forceinline__ bf16 f2bf(float v) { return static_cast<bf16>(v); }

}  // namespace nfd
