"""ctypes binding of libzgpt2_hip.so (the C ABI of include/zgpt2.h).

There is no fallback: if the HIP library is missing, load() raises.  build() compiles it in-tree
with hipcc for gfx950 (zig_gpt2_amd/csrc/Makefile), so the .so travels with the repo snapshot.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("ZGPT2_LIB") or os.path.join(_HERE, "lib", "libzgpt2_hip.so")  # ZGPT2_LIB: diagnostic builds
HEADER = os.path.join(os.path.dirname(_HERE), "include", "zgpt2.h")

_lib = None

f32p = C.POINTER(C.c_float)
szp = C.POINTER(C.c_size_t)
vp = C.c_void_p
sz = C.c_size_t


class ZgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libzgpt2_hip error {code}: {msg}")
        self.code = code


class GptConfig(C.Structure):
    """zg_gpt_config == GPTConfig of src/main.zig:5-23."""

    _fields_ = [("vocab_size", sz), ("context_size", sz), ("n_layer", sz), ("n_heads", sz), ("n_embed", sz)]


class GptOptions(C.Structure):
    """zg_gpt_options of include/zgpt2.h."""

    _fields_ = [("share_weights_with", vp), ("own_stream", C.c_int), ("stream_priority", C.c_int)]


# name -> (restype, argtypes); every symbol include/zgpt2.h declares
SIGNATURES = {
    "zg_init": (C.c_int, [C.c_int]),
    "zg_init_ex": (C.c_int, [C.c_int, sz]),
    "zg_shutdown": (C.c_int, []),
    "zg_last_error": (C.c_char_p, []),
    "zg_set_stream": (C.c_int, [vp]),
    "zg_synchronize": (C.c_int, []),
    "zg_register_tensor": (C.c_int, [vp, sz]),
    "zg_unregister_tensor": (C.c_int, [vp]),
    "zg_unregister_all": (C.c_int, []),
    "zg_debug_gemm_launches": (C.c_ulonglong, []),
    "zg_debug_gemm_stamps": (C.c_int, [vp, sz]),
    "zg_debug_last_kernel": (C.c_int, [C.c_char_p, sz]),
    "zg_debug_attn_prefill": (C.c_int, [vp, vp, sz, sz, sz, sz, vp, vp, sz, vp, sz, C.c_int]),
    "zg_debug_prefill_route": (C.c_int, [C.c_int, C.c_int]),
    "zg_debug_prefill_linear": (C.c_int, [vp, vp, vp, vp, sz, sz, sz, C.c_int, C.c_int, C.c_int, vp, sz]),
    "zg_linear_forward": (C.c_int, [sz, sz, vp, vp, vp, sz, vp, sz]),
    "zg_embedding_forward": (C.c_int, [sz, vp, sz, vp, sz, vp, sz]),
    "zg_layernorm_forward": (C.c_int, [sz, vp, vp, C.c_float, vp, sz]),
    "zg_attn_forward": (C.c_int, [sz, sz, vp, vp, vp, vp, sz, vp, sz] + [vp, sz] * 8),
    "zg_split_qkv": (C.c_int, [sz, sz, vp, sz, sz, vp, sz]),
    "zg_transpose": (C.c_int, [sz, sz, sz, vp, sz, vp, sz]),
    "zg_scaled_dot_product_attention": (C.c_int, [vp, sz, vp, sz, vp, sz, sz, sz, sz, vp, sz, vp, sz]),
    "zg_gemm_bf16_nt": (C.c_int, [vp, vp, vp, vp, sz, sz, sz, C.c_int, C.c_int]),
    "zg_f32_to_bf16": (C.c_int, [vp, vp, sz]),
    "zg_gelu": (C.c_int, [vp, sz]),
    "zg_softmax": (C.c_int, [vp, sz]),
    "zg_gpt_create": (C.c_int, [C.POINTER(vp), C.POINTER(GptConfig), sz, C.c_uint]),
    "zg_gpt_destroy": (C.c_int, [vp]),
    "zg_gpt_create_ex": (C.c_int, [C.POINTER(vp), C.POINTER(GptConfig), sz, C.c_uint, vp]),
    "zg_gpt_stream": (C.c_int, [vp, C.POINTER(vp)]),
    "zg_gpt_generate_enqueue_many": (C.c_int, [vp, sz, vp, sz, vp, sz]),
    "zg_gpt_generate_fetch_many": (C.c_int, [vp, sz, sz, vp, sz]),
    "zg_gpt_load_block_tensor": (C.c_int, [vp, sz, C.c_int, vp, sz]),
    "zg_gpt_load_tensor": (C.c_int, [vp, C.c_int, vp, sz]),
    "zg_gpt_weight_arena": (C.c_int, [vp, C.POINTER(vp), szp]),
    "zg_dist_available": (C.c_int, []),
    "zg_dist_unique_id": (C.c_int, [vp, sz]),
    "zg_dist_init": (C.c_int, [vp, sz, C.c_int, C.c_int]),
    "zg_dist_world": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "zg_gpt_broadcast_weights": (C.c_int, [vp, C.c_int, f32p]),
    "zg_dist_allgather": (C.c_int, [vp, vp, sz]),
    "zg_dist_finalize": (C.c_int, []),
    "zg_gpt_step_bytes": (C.c_int, [vp, sz, szp, szp]),
    "zg_gpt_forward": (C.c_int, [vp, sz, vp, sz, C.c_int, vp, sz]),
    "zg_gpt_prefill": (C.c_int, [vp, vp, sz, sz, C.c_int, vp, sz]),
    "zg_gpt_argmax": (C.c_int, [vp, vp, sz]),
    "zg_gpt_sample": (C.c_int, [vp, sz, vp, sz, C.c_float, vp, C.c_uint64, vp, vp, sz]),
    "zg_gpt_hidden": (C.c_int, [vp, vp, sz]),
    "zg_gpt_generate_greedy": (C.c_int, [vp, vp, sz, vp, sz, vp, sz]),
    "zg_gpt_generate_enqueue": (C.c_int, [vp, vp, sz, vp, sz]),
    "zg_gpt_generate_fetch": (C.c_int, [vp, sz, vp, sz]),
    "zg_gpt_generate_sample_enqueue": (C.c_int, [vp, vp, sz, vp, sz, C.c_float, C.c_uint64]),
    "zg_gpt_generate_sample": (C.c_int, [vp, vp, sz, vp, sz, C.c_float, C.c_uint64, vp, sz]),
    "zg_gpt_time_kernel": (C.c_int, [vp, C.c_int, C.c_int, f32p, szp]),
    "zg_gpt_profile_step": (C.c_int, [vp, sz, C.c_int, f32p, sz]),
    "zg_debug_prefetch_stats": (C.c_int, [vp, vp, sz]),
    "zg_bpe_create": (C.c_int, [C.POINTER(vp), vp, vp, sz, vp, vp, sz]),
    "zg_bpe_destroy": (C.c_int, [vp]),
    "zg_bpe_encode": (C.c_int, [vp, C.c_char_p, sz, vp, sz, szp]),
    "zg_bpe_decode": (C.c_int, [vp, vp, sz, vp, sz, szp]),
}

# flags / slots of include/zgpt2.h
GPT_WEIGHTS_BF16, GPT_WEIGHTS_F32, GPT_NO_GRAPH, GPT_KV_F16, GPT_NO_PREFILL, GPT_PREFILL_2PLANE, GPT_NO_PREFETCH = 0, 1, 2, 4, 8, 16, 32
GPT_KV_B24 = 64
GPT_SAMPLED_GENERATE = 128
BLOCK_SLOTS = ["ln_1_g", "ln_1_b", "c_attn_w", "c_attn_b", "c_proj_w", "c_proj_b",
               "ln_2_g", "ln_2_b", "c_fc_w", "c_fc_b", "mlp_proj_w", "mlp_proj_b"]
TOP_SLOTS = ["wte", "wpe", "ln_f_g", "ln_f_b"]
TIME_LM_HEAD = 6


def build(force=False):
    """hipcc --offload-arch=gfx950 build of every HIP source into lib/libzgpt2_hip.so."""
    args = ["make", "-C", CSRC, "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"build did not produce {SO_PATH}")
    return SO_PATH


def load():
    """Load the HIP library; raises if it is not built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: the HIP extension is required (run `python -c 'import __graft_entry__ as g; g.build()'`)"
        )
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7,
    # requested by torch as "libamdhip64.so").  If this library pulled in /opt/rocm's copy first, a
    # later `import torch` would load a second runtime that finds no GPU.  Importing torch first makes
    # the dynamic linker resolve our DT_NEEDED libamdhip64.so.7 to the copy torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise ZgError(code, load().zg_last_error().decode(errors="replace"))


def ptr(a):
    """Raw address of a numpy array (host) or torch tensor (host or device); None -> NULL."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a.ctypes.data
