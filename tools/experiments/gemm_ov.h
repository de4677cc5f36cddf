// gemm_ov.h — launchers of the overlapped-epilogue GEMM experiment (gemm_ov.hip).
#pragma once
#include "../../zig_gpt2_amd/csrc/zg_kernels.h"

namespace zg {
// bf16 result, 192-wide tiles, the epilogue of a tile under the next tile's main loop
bool gemm_ov_args_ok(int M, const GemmPlanes& pl, int ldc);
int launch_gemm_ov(const bf16_t* A, const bf16_t* B, const float* bias, void* C, int M, int N, const GemmPlanes& pl, int ldc, bool gelu,
                   hipStream_t s);
int gemm_ov_stamps(unsigned long long* out, size_t n_words);
}  // namespace zg
