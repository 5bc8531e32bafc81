// Internal interface between gemm.hip (ud_gemm entry point, fp32-MFMA kernel) and gemm_x3.hip (split-bf16 kernel).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/unidefense_hip.h"

// plain GEMM modes with 16-byte-loadable operands and no ragged vector tails
bool ud_gemm_x3_eligible(const ud_gemm_desc& d, bool a_vec, bool b_vec);
int ud_gemm_x3_launch(const ud_gemm_desc& d, hipStream_t s, bool f16);
// descriptors with half_mask != 0 (half-stored operands / result), fp16 MFMA
int ud_gemm_x3_launch_half(const ud_gemm_desc& d, hipStream_t s);
// rows per tile of the configuration the split-bf16 kernel would run this descriptor with
int ud_gemm_x3_tile_rows(const ud_gemm_desc& d);
