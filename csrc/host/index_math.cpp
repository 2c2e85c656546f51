// Host exports of the kernels' index arithmetic (csrc/common/index_math.h) for the
// bounds / bijectivity tests.  Nothing here runs in production paths.
#include "../common/index_math.h"

extern "C" {

int nf_ix_gemm_kc_off(int row, int chunk) { return nfix::gemm_kc_off(row, chunk); }
int nf_ix_gemm_km_off(int rowb, int k, int xb) { return nfix::gemm_km_off(rowb, k, xb); }
int nf_ix_gemm_tile_index(int id, int total) { return nfix::gemm_tile_index(id, total); }
int nf_ix_gemm_tile_rc(int t, int tiles_m, int tiles_n, int group_m) {
    return nfix::gemm_tile_rc(t, tiles_m, tiles_n, group_m);
}
int nf_ix_mnist_fwd_off(int row, int col, int chunk) { return nfix::mnist_fwd_off(row, col, chunk); }
int nf_ix_mnist_fwd_block_dy(int m) { return nfix::mnist_fwd_block_dy(m); }
int nf_ix_mnist_fwd_block_dx(int m) { return nfix::mnist_fwd_block_dx(m); }
int nf_ix_mnist_dg_off(int pr, int pc, int part) { return nfix::mnist_dg_off(pr, pc, part); }
int nf_ix_lds_b128_group(int lane) { return nfix::lds_b128_group(lane); }
int nf_ix_mfma32_acc_row(int reg, int lane) { return nfix::mfma32_acc_row(reg, lane); }
int nf_ix_conv_patch_source(int H, int W, int logH, int RI, int Cin, int row0, int pp) {
    return nfix::conv_patch_source(H, W, logH, RI, Cin, row0, pp);
}
int nf_ix_conv_patch_index(int W, int logW, int RI, int logRI, int m) {
    return nfix::conv_patch_index(W, logW, RI, logRI, m);
}
int nf_ix_conv_patch_byte(int W, int RI, int pp, int pitch, int rp, int bp) {
    return nfix::conv_patch_byte(W, RI, pp, pitch, rp, bp);
}
int nf_ix_conv_patch_frag_byte(int W, int logW, int RI, int logRI, int m, int pitch, int rp,
                               int bp) {
    return nfix::conv_patch_frag_byte(W, logW, RI, logRI, m, pitch, rp, bp);
}
int nf_ix_conv_patch_read_cycles(int W, int logW, int RI, int logRI, int tp, int pitch, int rp,
                                 int bp) {
    return nfix::conv_patch_read_cycles(W, logW, RI, logRI, tp, pitch, rp, bp);
}
// rp and bp packed as rp | bp << 16 would overflow for large bands: two calls instead
int nf_ix_conv_patch_row_pitch(int W, int logW, int RI, int logRI, int NI, int tp, int pitch) {
    int rp = 0, bp = 0;
    nfix::conv_patch_pitches(W, logW, RI, logRI, NI, tp, pitch, &rp, &bp);
    return rp;
}
int nf_ix_conv_patch_band_pitch(int W, int logW, int RI, int logRI, int NI, int tp, int pitch) {
    int rp = 0, bp = 0;
    nfix::conv_patch_pitches(W, logW, RI, logRI, NI, tp, pitch, &rp, &bp);
    return bp;
}

}  // extern "C"
