// exp_api.hip — C entry points of libzgpt2_exp.so: kernels that were measured and did NOT make the product library
// (zig_gpt2_amd/csrc), kept runnable beside it.  Links libzgpt2_hip.so for the runtime (stream, error plumbing).
#include "gemm_ov.h"
#include "../../zig_gpt2_amd/csrc/zg_runtime.h"

using namespace zg;

extern "C" {

// C[M,N] = act(A[M,K] B[N,K]^T + bias) rounded to bf16 on gemm_ov_kernel (the epilogue of a tile under the next tile's main
// loop).  Same operands as zg_gemm_bf16_nt(..., out_bf16 = 1); bitwise equal to gemm_s4, slower at two tiles per workgroup.
int zg_exp_gemm_ov(const uint16_t* A, const uint16_t* B, const float* bias, uint16_t* C, size_t M, size_t N, size_t K, int gelu) {
    ZG_TRY(require_init());
    ZG_REQUIRE(K >= 128 && K % 64 == 0 && N % 8 == 0, ZG_ERR_UNSUPPORTED, "exp_gemm_ov: K = %zu, N = %zu", K, N);
    GemmPlanes pl{};
    pl.lda = (int)K;
    pl.ldb = (int)K;
    pl.kpp = (int)(K / 64);
    pl.npairs = 1;
    ZG_REQUIRE(gemm_ov_args_ok((int)M, pl, (int)N), ZG_ERR_UNSUPPORTED, "exp_gemm_ov: shape beyond the kernel's packed arguments");
    return launch_gemm_ov(A, B, bias, C, (int)M, (int)N, pl, (int)N, gelu != 0, ctx().stream);
}

int zg_exp_gemm_ov_stamps(unsigned long long* out, size_t n_words) { return gemm_ov_stamps(out, n_words); }

}  // extern "C"
