// gemv.hip — the decode-regime Linear: y[M,N] = x[M,K] * W[N,K]^T (+ bias) for M <= 8.
//
// Replaces the cblas_sgemm call of Linear.forward (reference src/ops.zig:21-46) when the batch
// is tiny (decode: M = number of lock-step sequences).  HBM-bandwidth bound: every weight byte is
// read exactly once, 16 B per lane, K-contiguous rows ([out,in] layout of ops.Linear.weight), so
// one row is read by a group of LPR lanes with fully coalesced 16-B loads and reduced with DPP
// row rotations (M == 1: no barrier, except where a workgroup shares one copy of a wide input).
//
// Fused around the dot products (the reference does these as separate host loops / ops):
//   prologue  PRO_LAYERNORM   LayerNorm.forward of the input row   (src/ops.zig:82-104)
//             PRO_ATTN_MERGE  combine split-KV attention partials  (src/ops.zig:284-305 tail)
//   epilogue  EPI_RESIDUAL    state.o + state.x residual adds      (src/main.zig:136-145)
//             EPI_GELU        ops.gelu                             (src/ops.zig:221-228)
//             EPI_QKV         split_qkv + KV-cache append          (src/ops.zig:146-157)
//             EPI_ARGMAX      greedy sampler partial argmax        (replaces src/main.zig:198-207)
//
// This file: which kernel a launch takes (launch_gemv) and how it is laid out (gemv_plan and the shape tests the model
// tier asks before it builds its step).  The kernels: gemv_valu.hip (any prologue / epilogue, M <= 8, vector ALUs),
// gemv_ksplit.hip (M == 1: K-split and linearised-LayerNorm forms), gemv_mfma16.hip / gemv_pl4.hip (2..8 sequences on the
// matrix cores: 16-wave and four-wave plane-fed forms, wave-per-tile lm_head); shared device helpers in gemv_internal.h.
#include "gemv_internal.h"

namespace zg {

namespace {

int launch_gemv_mfma(const GemvArgs& a, int grid, hipStream_t s) {
    const int pairs = pl4_pairs(a);
    if (pairs > 0) return gemv_launch_pl4(a, pairs, grid, s);                 // plane-fed four-wave kernel
    if (a.kslices != 4) {
        const int steps = lm_wpt_steps(a);
        if (steps > 0) return gemv_launch_lm_wpt(a, steps, grid, s);          // lm_head, one wave per tile
    }
    return gemv_launch_mfma16(a, grid, s);
}

}  // namespace

int gemv_lanes_per_row(int K) {
    const int nch = K / 8;
    if (nch <= 16 * 8) return 16;
    if (nch <= 32 * 8) return 32;
    return 64;
}

// Wide, thin, un-normalised Linears of the lock-step batch (mlp c_proj) are cut into four K slices over as many
// workgroups when the caller provided the combine workspace.
int gemv_kslices(const GemvArgs& a) {
    if (a.sk_ws == nullptr || a.sk_cnt == nullptr || a.M < 2 || a.M > kMfmaRows) return 1;
    if (a.prologue != PRO_NONE || a.epilogue == EPI_ARGMAX || a.epilogue == EPI_QKV) return 1;
    if (a.K < 2048 || a.K % 128 != 0 || a.K / 4 > 3072 || (a.N + 15) / 16 > a.sk_tiles) return 1;
    return 4;
}

// Rows per wave: enough waves to cover the chip (256 CUs x 4 SIMDs x 2) without dropping below
// one double pass (2 * 64/LPR rows) per wave.
// The matrix-core path serves the model tier's lock-step batch: bf16 weights, 2..8 rows, K a
// multiple of 32 whose three input planes fit in LDS, a fused LayerNorm no wider than 2048.
bool gemv_use_mfma(const GemvArgs& a, int weight_type) {
    if (weight_type != WT_BF16 || a.M < 2 || a.M > kMfmaRows) return false;
    if (gemv_kslices(a) > 1) return true;
    if (a.K % 32 != 0 || a.K / 32 < 4 || a.K / 32 > 96) return false;
    if (a.prologue == PRO_LAYERNORM && a.K > 2048) return false;
    if (gemv_mfma_lds(a.K, gemv_mfma_waves(a)) <= 160 * 1024) return true;
    // wide K: only as single-tile workgroups whose partial tiles alias the planes
    constexpr int wgs = 768;
    return a.epilogue != EPI_ARGMAX && (a.N + 15) / 16 <= wgs && gemv_mfma_lds(a.K, 16, true) <= 160 * 1024;
}

// Can a Linear with this M and K run at all?  (The batched kernels keep all M input rows in LDS.)
namespace {
inline size_t valu_lds(int mt, int K) { return ((size_t)mt * K + 4 * mt * 2 + 64) * sizeof(float); }
inline int valu_mt(int M) { return M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : 8)); }
// Rows of a plain (no prologue) Linear are independent: when M rows of K floats exceed the LDS the batch is
// run as row groups that fit (the weights are streamed once per group).
inline bool splittable(const GemvArgs& a) {
    return a.prologue == PRO_NONE && (a.epilogue == EPI_STORE || a.epilogue == EPI_RESIDUAL || a.epilogue == EPI_GELU);
}
inline int row_group(const GemvArgs& a) {
    int g = valu_mt(a.M);
    while (g > 1 && valu_lds(g, a.K) > 160 * 1024) g >>= 1;
    return g;
}
}  // namespace

bool gemv_supported(const GemvArgs& a, int weight_type) {
    if (a.M <= 1 || gemv_use_mfma(a, weight_type)) return true;
    if (valu_lds(valu_mt(a.M), a.K) <= 160 * 1024) return true;
    return splittable(a) && valu_lds(row_group(a), a.K) <= 160 * 1024;
}

int gemv_plan(GemvArgs& a, int weight_type) {
    if (gemv_use_mfma(a, weight_type)) {
        const int ntiles = (a.N + 15) / 16;
        a.kslices = gemv_kslices(a);
        if (a.kslices > 1) {  // one tile per workgroup and slice
            a.rows_per_wave = 1;
            return ntiles;
        }
        if (lm_wpt_steps(a) > 0) {  // one wave per tile: every wave of a workgroup gets the same number of tiles
            a.rows_per_wave = lm_wpt_tiles_per_wg();
            return (ntiles + a.rows_per_wave - 1) / a.rows_per_wave;
        }
        constexpr int wgs = 768;
        int tpw = (ntiles + wgs - 1) / wgs;  // at most ~4 workgroups per CU for the widest matrices
        if (tpw < 1) tpw = 1;
        a.rows_per_wave = tpw;            // tiles per workgroup on this path
        return (ntiles + tpw - 1) / tpw;
    }
    const int rpp2 = 2 * (64 / gemv_lanes_per_row(a.K));
    const int target_waves = 256 * 4 * 2;
    int rpw = (a.N + target_waves - 1) / target_waves;
    rpw = ((rpw + rpp2 - 1) / rpp2) * rpp2;
    if (rpw < rpp2) rpw = rpp2;
    a.rows_per_wave = rpw;
    const int waves = (a.N + rpw - 1) / rpw;
    // M == 1 without the argmax tail: one-wave workgroups while the matrix has at most ~8 waves per CU
    int wpw = 4;
    if (a.M == 1 && a.epilogue != EPI_ARGMAX) wpw = waves <= 2048 ? 1 : 4;
    if (a.M == 1 && a.epilogue != EPI_ARGMAX && a.prologue == PRO_NONE && a.K >= 2048 && a.K <= 8192)
        wpw = 2;  // measured in situ: 2 >= 4 at K = 3072 (124M) and K = 6400 (XL)
    if (a.M == 1 && a.epilogue != EPI_ARGMAX && a.prologue == PRO_ATTN_MERGE) wpw = 4;
    a.waves_per_wg = wpw;
    return (waves + wpw - 1) / wpw;
}

// Mirrors the dispatch of launch_gemv below for a planned launch (prefetch.hip follows the same tiles).
int gemv_rows_per_wg(const GemvArgs& a, int weight_type) {
    if (gemv_use_mfma(a, weight_type)) return a.kslices > 1 ? 0 : 16 * a.rows_per_wave;
    const int nchq = a.K / 32;
    if (gemv_use_ksplit(a)) return 2 * (64 / (nchq <= 32 ? 16 : (nchq <= 224 ? 32 : 64)));  // launch_ksplit
    if (gemv_use_lnk(a)) return 4 * (64 / (nchq <= 32 ? 16 : (nchq <= 96 ? 32 : 64)));      // launch_lnk
    if (a.M > 1) return 0;
    return a.waves_per_wg * a.rows_per_wave;
}

bool gemv_planes_ok(const GemvArgs& a, int weight_type) {
    if (a.epilogue == EPI_ARGMAX || !gemv_use_mfma(a, weight_type)) return false;
    return a.prologue == PRO_NONE || (a.prologue == PRO_LAYERNORM && a.ln_c2 != nullptr && a.ln_c3 != nullptr && a.K <= 2048);
}

bool gemv_pl4_ok(const GemvArgs& a, int weight_type) {
    if (!gemv_planes_ok(a, weight_type)) return false;
    GemvArgs b = a;
    (void)gemv_plan(b, weight_type);
    if (b.pl_in == nullptr) b.pl_in = reinterpret_cast<const bf16_t*>(a.W);  // any non-null: only the shape is judged
    return pl4_pairs(b) > 0;
}

bool gemv_planes_producer_ok(const GemvArgs& a, int weight_type) { return a.epilogue != EPI_ARGMAX && gemv_use_mfma(a, weight_type); }

int launch_gemv(const GemvArgs& a, int weight_type, int grid, hipStream_t s) {
    ZG_REQUIRE(a.pl_in == nullptr || gemv_planes_ok(a, weight_type), ZG_ERR_UNSUPPORTED, "gemv: input planes given to a launch outside the matrix-core path");
    ZG_REQUIRE(a.pl_out == nullptr || gemv_use_mfma(a, weight_type), ZG_ERR_UNSUPPORTED, "gemv: output planes asked of a launch outside the matrix-core path");
    if (gemv_use_mfma(a, weight_type)) return launch_gemv_mfma(a, grid, s);
    if (gemv_use_ksplit(a)) return gemv_launch_ksplit(a, weight_type, s);
    if (gemv_use_lnk(a)) return gemv_launch_lnk(a, weight_type, s);
    if (a.M > 1 && valu_lds(valu_mt(a.M), a.K) > 160 * 1024 && splittable(a)) {
        const int g = row_group(a);
        for (int m0 = 0; m0 < a.M; m0 += g) {
            GemvArgs b = a;
            b.M = a.M - m0 < g ? a.M - m0 : g;
            b.x = a.x + (size_t)m0 * a.x_stride;
            b.y = a.y + (size_t)m0 * a.y_stride;
            if (a.resid) b.resid = a.resid + (size_t)m0 * a.resid_stride;
            GemvArgs p = b;
            const int gb = gemv_plan(p, weight_type);  // M == 1 groups are planned differently
            ZG_TRY(gemv_launch_valu(p, weight_type, gb, s));
        }
        return ZG_OK;
    }
    return gemv_launch_valu(a, weight_type, grid, s);
}

}  // namespace zg
