// bf16 ACTIVATION STORAGE on the lean depth-wise strip kernel (dw_lean.h, H16): defined in dwconv.hip, called by bf16_dw.hip.
#pragma once
#include "tsii_common.h"

namespace tsii {

// stride-1 3x3 layers the H16 form takes: dilation 1, and dilations 2 / 4 by phases; a function of the OUTPUT grid only (the
// partial-row counts of the fused forms -- tsii_bf16_dw_stat_rows / tsii_bf16_dw_bwd_stat_rows -- see nothing else)
bool hdw_lean_ok(int hout, int wout, int c, int s, int d);
// partial rows of the K6b statistics / K6c reductions the H16 form writes for that output grid
int64_t hdw_lean_rows(int n, int hout, int wout, int c, int s, int d);
// in [n, hin, win, c] -> out [n, hout, wout, c] (bf16), w = fp32 [c][9]; flip: taps (2 - ky, 2 - kx) (the adjoint);
// in_sc / in_sh (+ in_neg, in_hi): BatchNorm + activation of the producer applied on load (K6b) or NULL; stats: K6b partials
// [rows][4][c] or NULL; bn_y (bf16, the output grid) ... bn_part [rows][2][c]: K6c reductions or NULL.
// -> 0 launched, 1 the geometry is outside the form's limits (the caller keeps its own kernels), < 0 error
int launch_hdw_lean(const void* in, const float* w, const float* bias, int n, int hin, int win, int c, int d, int pad_h, int pad_w,
                    int hout, int wout, int flip, const float* in_sc, const float* in_sh, float in_neg, float in_hi, float* stats,
                    const void* bn_y, const float* bn_mean, const float* bn_var, const float* bn_gamma, const float* bn_beta,
                    float bn_eps, float bn_neg, float bn_hi, float* bn_part, void* out, hipStream_t st);

}  // namespace tsii
