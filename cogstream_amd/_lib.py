"""ctypes binding of libcogs_hip.so (include/cogs.h). The product path has no CPU fallback:
importing this module without the built library raises, and every op checks its status code."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_HERE = Path(__file__).resolve().parent
import os as _os

LIB_PATH = Path(_os.environ.get("COGS_LIB_PATH", str(_HERE / "libcogs_hip.so")))  # another BUILD of the library (tools/build_alt.sh)

DT_BF16, DT_F32 = 0, 1
OK, E_INVALID, E_HIP, E_UNSUPPORTED, E_WORKSPACE = 0, -1, -2, -3, -4      # cogs_status (include/cogs.h)
ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_SWIGLU = 0, 1, 2, 3
ATTN_BLOCK_DIAG, ATTN_REF_EAGER_GLOBAL = 0, 1


class CogsError(RuntimeError):
    pass


if not LIB_PATH.exists():
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -m cogstream_amd.build` "
        "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path."
    )

lib = C.CDLL(str(LIB_PATH))

c_void_p, c_int, c_int64, c_float, c_size_t = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class GemmDesc(C.Structure):
    _fields_ = [
        ("dtype", c_int),
        ("A", c_void_p), ("lda", c_int64),
        ("W", c_void_p), ("ldw", c_int64),
        ("C", c_void_p), ("ldc", c_int64),
        ("bias", c_void_p),
        ("residual", c_void_p), ("ldr", c_int64),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("act", c_int),
        ("out_f32", c_int),
        ("rope_cos", c_void_p),
        ("rope_sin", c_void_p),
        ("rope_cols", c_int),
        ("head_dim", c_int),
        ("rope_lut", c_void_p),
        ("rope_rowpos", c_void_p),
        ("rope_maxpos", c_int),
        ("row_stats", c_void_p),
        ("ln_ab", c_void_p),
        ("col_c", c_void_p),
        ("hm_rows", c_int64),
        ("hm_cols", c_int),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("dtype", c_int),
        ("Q", c_void_p), ("K", c_void_p), ("V", c_void_p), ("O", c_void_p),
        ("ldq", c_int64), ("ldk", c_int64), ("ldv", c_int64), ("ldo", c_int64),
        ("cu_seqlens", c_void_p), ("nseg", c_int), ("max_seqlen", c_int),
        ("row_lo", c_void_p), ("row_hi", c_void_p), ("bias", c_float),
        ("q_len", c_int), ("kv_len", c_int),
        ("hq", c_int), ("hkv", c_int), ("head_dim", c_int),
        ("scale", c_float),
        ("causal", c_int), ("q_pos0", c_int),
        ("force_rowwise", c_int),
        ("nsplit", c_int), ("ws", c_void_p), ("ws_bytes", c_size_t),
        ("q_prescaled", c_int),
        ("head_stride", c_int64),
    ]


class VitLayer(C.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "ln1_g", "ln1_b", "qkv_w", "qkv_b", "o_w", "o_b", "ln2_g", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
        "qkv_c", "fc1_c")]


class VitWeights(C.Structure):
    _fields_ = [
        ("dtype", c_int),
        ("hidden", c_int), ("inter_pad", c_int), ("layers", c_int), ("heads", c_int),
        ("patch_dim", c_int), ("patch_pad", c_int),
        ("ln_eps", c_float),
        ("patch_w", c_void_p), ("patch_b", c_void_p), ("post_ln_g", c_void_p), ("post_ln_b", c_void_p),
        ("layer", C.POINTER(VitLayer)),
    ]


class ProjWeights(C.Structure):
    _fields_ = [("dtype", c_int), ("in_dim", c_int), ("out_dim", c_int),
                ("w1", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p)]


class LlmLayer(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("in_ln", "qkv_w", "qkv_b", "o_w", "post_ln", "gu_w", "down_w")]


class LlmWeights(C.Structure):
    _fields_ = [
        ("dtype", c_int),
        ("hidden", c_int), ("inter", c_int), ("layers", c_int), ("heads", c_int), ("kv_heads", c_int),
        ("head_dim", c_int), ("vocab", c_int),
        ("rms_eps", c_float), ("rope_theta", c_float),
        ("final_norm", c_void_p), ("lm_head", c_void_p),
        ("layer", C.POINTER(LlmLayer)),
    ]


class KV(C.Structure):
    _fields_ = [("k", c_void_p), ("v", c_void_p), ("max_len", c_int), ("len", c_int)]


# every symbol include/cogs.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "cogs_status_string": (C.c_char_p, [c_int]),
    "cogs_version": (C.c_char_p, []),
    "cogs_debug_set": (c_int, [C.c_char_p, c_int64]),
    "cogs_debug_get": (c_int, [C.c_char_p, C.POINTER(c_int64)]),
    "cogs_debug_list": (C.c_char_p, []),
    "cogs_create": (c_int, [c_int, C.POINTER(c_void_p)]),
    "cogs_destroy": (c_int, [c_void_p]),
    "cogs_vit_set_streams": (c_int, [c_void_p, c_int]),
    "cogs_profile_begin": (c_int, [c_void_p]),
    "cogs_profile_end": (c_int, [c_void_p, c_void_p, C.POINTER(c_float), C.POINTER(c_int)]),
    "cogs_gemm": (c_int, [c_void_p, C.POINTER(GemmDesc)]),
    "cogs_ln_finalize": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "cogs_attention": (c_int, [c_void_p, C.POINTER(AttnDesc)]),
    "cogs_layernorm": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float]),
    "cogs_rmsnorm": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float]),
    "cogs_ln_merge": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float]),
    "cogs_pixdiff_mask": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    "cogs_frame_mean_to_slot0": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int]),
    "cogs_gather_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "cogs_mean_rows": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "cogs_cosine": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "cogs_kmeans_workspace_bytes": (c_int, [c_int, c_int64, c_int, C.POINTER(c_size_t)]),
    "cogs_kmeans_sqdist": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_int, c_void_p,
                                   c_void_p, c_size_t]),
    "cogs_kmeans_assign": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    "cogs_kmeans_update": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "cogs_select_near_centroid": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "cogs_kmeans_pp_step": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_size_t]),
    "cogs_kmeans_pp": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_size_t]),
    "cogs_kmeans_margins": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "cogs_kmeans_lloyd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_int, c_float,
                                  c_void_p, c_int, c_void_p, c_void_p, c_void_p, C.POINTER(c_int), C.POINTER(c_int),
                                  C.POINTER(c_int), c_void_p, c_size_t]),
    "cogs_pack_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int]),
    "cogs_preprocess_workspace_bytes": (c_int, [c_int, c_int, c_int, C.POINTER(c_size_t)]),
    "cogs_preprocess_frames": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                       c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_size_t]),
    "cogs_argmax": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "cogs_logits_process": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_int, c_float,
                                    c_void_p]),
    "cogs_topk": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "cogs_sample_workspace_bytes": (c_size_t, []),
    "cogs_sample": (c_int, [c_void_p, c_void_p, c_int, c_float, c_int, C.c_double, c_void_p, C.c_uint64, C.c_uint64, c_void_p,
                            c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "cogs_vit_load": (c_int, [c_void_p, C.POINTER(VitWeights)]),
    "cogs_vit_workspace_bytes": (c_int, [c_void_p, c_int64, C.POINTER(c_size_t)]),
    "cogs_vit_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, C.POINTER(c_int64), C.POINTER(c_int64), c_int,
                                c_int, c_void_p, c_void_p, c_size_t]),
    "cogs_allgather_tokens": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "cogs_proj_load": (c_int, [c_void_p, C.POINTER(ProjWeights)]),
    "cogs_vit_encode_project": (c_int, [c_void_p, c_void_p, c_void_p, c_int, C.POINTER(c_int64), C.POINTER(c_int64), c_int,
                                        c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t]),
    "cogs_project": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t]),
    "cogs_llm_load": (c_int, [c_void_p, C.POINTER(LlmWeights)]),
    "cogs_llm_workspace_bytes": (c_int, [c_void_p, c_int, c_int, C.POINTER(c_size_t)]),
    "cogs_llm_forward_segments": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t]),
    "cogs_llm_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, C.POINTER(KV), c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_size_t]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not match include/cogs.h
    _fn.restype = _res
    _fn.argtypes = _args


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib.cogs_status_string(status).decode()
        raise CogsError(f"{what or 'cogs call'} failed: {msg} ({status})")


def debug_set(name: str, value: int) -> None:
    """one diagnostic switch of the library (csrc/debug.h; `debug_list()` prints the table)"""
    check(lib.cogs_debug_set(name.encode(), int(value)), f"cogs_debug_set({name})")


def debug_get(name: str) -> int:
    v = c_int64(0)
    check(lib.cogs_debug_get(name.encode(), C.byref(v)), f"cogs_debug_get({name})")
    return int(v.value)


def debug_list() -> str:
    return lib.cogs_debug_list().decode()


class debug_switch:
    """with debug_switch("gemm_pp64", 0): ...   -- sets a switch and restores the previous value on exit"""

    def __init__(self, name: str, value: int):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = debug_get(self.name)
        debug_set(self.name, self.value)
        return self

    def __exit__(self, *exc):
        debug_set(self.name, self.old)
        return False


def debug_from_spec(spec: str) -> dict:
    """"name=value[,name=value...]" -> applied switches (bench.py --debug, tools/*)"""
    done = {}
    for item in filter(None, (x.strip() for x in spec.split(","))):
        k, _, v = item.partition("=")
        debug_set(k.strip(), int(v))
        done[k.strip()] = int(v)
    return done


def debug_from_argv(argv: list) -> dict:
    """strip every `--debug name=value[,...]` pair out of argv (in place) and apply it: lets any tool run under an A/B
    switch without its own option parsing"""
    done = {}
    while "--debug" in argv:
        i = argv.index("--debug")
        done.update(debug_from_spec(argv[i + 1]))
        del argv[i:i + 2]
    return done


def dtype_code(torch_dtype) -> int:
    import torch

    if torch_dtype == torch.bfloat16:
        return DT_BF16
    if torch_dtype == torch.float32:
        return DT_F32
    raise CogsError(f"unsupported dtype {torch_dtype}: the HIP path computes in bfloat16 or float32")


def ptr(t) -> int:
    """device pointer of a torch tensor (None -> NULL)"""
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
