"""ctypes binding of libgava_hip.so (include/gava_hip.h).

The product path has NO fallback: if the library is missing or a call is rejected this module
raises.  torch is used only to own device memory and to name the current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GAVA_HIP_LIB") or os.path.join(_HERE, "libgava_hip.so")   # env: A/B experiment builds only

PREC_F16, PREC_BF16 = 0, 1
KERNEL_AUTO, KERNEL_256, KERNEL_PAIR, KERNEL_PP = 0, 3, 4, 5     # gava_gemm_args.kernel
EPI_H16, EPI_H16_QGELU, EPI_F32, EPI_F32_PATCH, EPI_H16_QGELU_BWD = 0, 1, 2, 3, 4
PREC_NAMES = {"fp16": PREC_F16, "f16": PREC_F16, "bf16": PREC_BF16}
PREC_TORCH = {PREC_F16: torch.float16, PREC_BF16: torch.bfloat16}

EXPORTS = ["gava_abi_version", "gava_gemm", "gava_layernorm", "gava_attention",
           "gava_vision_workspace_bytes", "gava_vision_forward", "gava_text_workspace_bytes",
           "gava_text_forward", "gava_similarity_head", "gava_convert_h16", "gava_debug_set_buffer",
           "gava_preprocess_clip", "gava_layernorm_backward", "gava_qgelu_backward", "gava_attention_backward",
           "gava_text_forward_train", "gava_vision_forward_train", "gava_attention_backward_workspace_bytes", "gava_vision_forward_keep", "gava_row_stats",
           "gava_probe_fc1_enable", "gava_probe_fc1_read", "gava_clip_geometry", "gava_patchify", "gava_attention_f32",
           "gava_gemm_aligned_walk", "gava_vision_pair_stream", "gava_struct_sizes"]

_vp, _fp, _ip = C.c_void_p, C.c_void_p, C.c_void_p  # all device pointers travel as void*


class GemmArgs(C.Structure):
    _fields_ = [("A", _vp), ("lda", C.c_int64), ("W", _vp), ("ldw", C.c_int64), ("bias", _fp),
                ("out", _vp), ("ldo", C.c_int64), ("resid", _fp), ("ldr", C.c_int64),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("epilogue", C.c_int), ("prec", C.c_int),
                ("scale_cols", C.c_int), ("scale", C.c_float),
                ("pos", _fp), ("time", _fp), ("n_patches", C.c_int), ("T", C.c_int),
                ("frames", _fp), ("frame_size", C.c_int), ("patch", C.c_int), ("split_out", C.c_int),
                ("aux", _vp), ("aux_prec", C.c_int), ("aux_out", _vp),
                ("x16_out", _vp), ("ld_x16", C.c_int64), ("rowsum_out", _fp),
                ("fold_stats", _fp), ("fold_s", _fp), ("fold_t", _fp), ("cu_reserve", C.c_int),
                ("rowsum_reduced", C.c_int), ("fold_partials", _fp),
                ("clips", _vp), ("clip_lut", _fp), ("kernel", C.c_int),
                ("w_lo", C.c_int), ("A8", _vp), ("lda8", C.c_int64), ("W8", _vp), ("ldw8", C.c_int64), ("w8_exp", C.c_int),
                ("out8", _vp), ("ldo8", C.c_int64), ("x8_out", _vp), ("ld_x8", C.c_int64),
                ("resid16", _vp), ("resid_lo", _vp), ("xlo_out", _vp)]


class LayerNormArgs(C.Structure):
    _fields_ = [("inp", _fp), ("in_stride", C.c_int64), ("in_row_index", _ip), ("gamma", _fp), ("beta", _fp),
                ("out16", _vp), ("out16_stride", C.c_int64), ("out32", _fp), ("out32_stride", C.c_int64),
                ("rows", C.c_int), ("D", C.c_int), ("prec", C.c_int), ("split_out", C.c_int),
                ("gamma2", _fp), ("beta2", _fp),
                ("out_hi", _vp), ("out_lo", _vp), ("out_hl_stride", C.c_int64)]


class AttentionArgs(C.Structure):
    _fields_ = [("q", _vp), ("k", _vp), ("v", _vp), ("ld_qkv", C.c_int64),
                ("side_k", _vp), ("side_v", _vp), ("ld_side", C.c_int64),
                ("out", _vp), ("ld_out", C.c_int64),
                ("batch", C.c_int), ("heads", C.c_int), ("n_q", C.c_int), ("n_kmain", C.c_int),
                ("n_g", C.c_int), ("T", C.c_int), ("has_summary", C.c_int),
                ("causal", C.c_int), ("prec", C.c_int), ("split_out", C.c_int),
                ("q_batch_rows", C.c_int), ("ld_q", C.c_int64)]


class ClipDesc(C.Structure):
    _fields_ = [("frames", _vp), ("n_frames", C.c_int), ("height", C.c_int), ("width", C.c_int),
                ("t_st", C.c_int), ("rate", C.c_int), ("h_st", C.c_int), ("w_st", C.c_int),
                ("scale_h", C.c_float), ("scale_w", C.c_float)]


class VisionLayer(C.Structure):
    _fields_ = [(n, _vp) for n in (
        "w_qkv", "b_qkv", "w_out", "b_out", "w_fc1", "b_fc1", "w_fc2", "b_fc2",
        "ln1_g", "ln1_b", "ln2_g", "ln2_b", "w_cls", "b_cls", "sln_g", "sln_b",
        "w_sqkv", "b_sqkv", "w_sout", "b_sout", "local_prompts", "global_prompts",
        "w_qkv_fold", "qkv_fold_s", "qkv_fold_t", "w_fc1_fold", "fc1_fold_s", "fc1_fold_t",
        "w_q_split", "w_out_split", "w_fc1_split", "w_fc2_split")]


class VisionModel(C.Structure):
    _fields_ = [("B", C.c_int), ("T_in", C.c_int), ("T_model", C.c_int),
                ("size", C.c_int), ("P", C.c_int), ("D", C.c_int), ("H", C.c_int), ("layers", C.c_int),
                ("F", C.c_int), ("E", C.c_int), ("G", C.c_int), ("prec", C.c_int),
                ("w_patch", _vp), ("b_patch", _fp), ("cls_token", _fp), ("pos_embed", _fp), ("time_embed", _fp),
                ("lnpre_g", _fp), ("lnpre_b", _fp), ("lnpost_g", _fp), ("lnpost_b", _fp),
                ("w_proj", _vp), ("layer", C.POINTER(VisionLayer)),
                ("clips", _vp), ("clip_lut", _fp), ("w_lo", C.c_int), ("layer8", _vp)]


class VisionLayer8(C.Structure):
    _fields_ = [(n, _vp) for n in ("w_qkv_fold8", "w_fc1_fold8", "w_fc28", "qkv_fold_s8", "fc1_fold_s8")] + \
               [(n, C.c_int) for n in ("qkv_fold_exp", "fc1_fold_exp", "fc2_exp")]


class TextLayer(C.Structure):
    _fields_ = [(n, _vp) for n in (
        "w_qkv", "b_qkv", "w_out", "b_out", "w_fc", "b_fc", "w_proj", "b_proj",
        "ln1_g", "ln1_b", "ln2_g", "ln2_b")]


class AttentionF32Args(C.Structure):
    _fields_ = [("q", _fp), ("k", _fp), ("v", _fp), ("ld", C.c_int64), ("out", _vp), ("ld_out", C.c_int64),
                ("batch", C.c_int), ("heads", C.c_int), ("L", C.c_int), ("causal", C.c_int), ("prec", C.c_int),
                ("split_out", C.c_int), ("scale", C.c_float)]


class TextModel(C.Structure):
    _fields_ = [("n_prompts", C.c_int), ("L", C.c_int), ("W", C.c_int), ("H", C.c_int), ("layers", C.c_int),
                ("E", C.c_int), ("n_ctx", C.c_int), ("prec", C.c_int), ("split", C.c_int), ("attn_f32", C.c_int),
                ("token_embedding", _fp), ("positional_embedding", _fp), ("lnf_g", _fp), ("lnf_b", _fp),
                ("w_tproj", _vp), ("layer", C.POINTER(TextLayer))]


class PreprocessArgs(C.Structure):
    _fields_ = [("frames", _vp), ("n_frames", C.c_int), ("height", C.c_int), ("width", C.c_int),
                ("mean", C.c_float * 3), ("std", C.c_float * 3),
                ("T", C.c_int), ("rate", C.c_int), ("size", C.c_int),
                ("out", _fp), ("out_stride_c", C.c_int64), ("out_stride_t", C.c_int64),
                ("first_temporal_view", C.c_int), ("first_spatial_view", C.c_int), ("lut", _fp)]


class PatchifyArgs(C.Structure):
    _fields_ = [("x", _fp), ("clips", _vp), ("clip_lut", _fp),
                ("B", C.c_int), ("T", C.c_int), ("size", C.c_int), ("patch", C.c_int), ("prec", C.c_int),
                ("out", _vp), ("ldo", C.c_int64)]


class VisionSaved(C.Structure):
    _fields_ = [("e0", _fp), ("x", _fp), ("x1", _fp), ("qkv", _vp), ("pre", _vp), ("sidekv", _vp),
                ("last_q", _vp), ("last_x1", _fp), ("last_pre", _vp)]


class LayerNormBwdArgs(C.Structure):
    _fields_ = [("x", _fp), ("x_stride", C.c_int64), ("x_row_index", _ip), ("gamma", _fp),
                ("dy", _fp), ("dy_stride", C.c_int64),
                ("dx", _fp), ("dx_stride", C.c_int64), ("dx_row_index", _ip),
                ("dgamma", _fp), ("dbeta", _fp), ("rows", C.c_int), ("D", C.c_int), ("accumulate", C.c_int),
                ("dx16", _vp), ("dx16_stride", C.c_int64), ("prec", C.c_int)]


class AttentionBwdArgs(C.Structure):
    _fields_ = [("q", _vp), ("k", _vp), ("v", _vp), ("ld_qkv", C.c_int64), ("dout", _vp), ("ld_dout", C.c_int64),
                ("dq", _vp), ("dk", _vp), ("dv", _vp), ("ld_dqkv", C.c_int64),
                ("batch", C.c_int), ("heads", C.c_int), ("n", C.c_int), ("causal", C.c_int), ("prec", C.c_int),
                ("q_scale", C.c_float),
                ("side_k", _vp), ("side_v", _vp), ("ld_side", C.c_int64),
                ("dside_k", _fp), ("dside_v", _fp), ("ld_dside", C.c_int64),
                ("n_g", C.c_int), ("T", C.c_int), ("has_summary", C.c_int), ("n_q", C.c_int), ("workspace", _vp),
                ("act_prec_set", C.c_int), ("act_prec", C.c_int),
                ("q_batch_rows", C.c_int), ("ld_q", C.c_int64), ("ld_dq", C.c_int64)]


_lib = None


class GavaError(RuntimeError):
    pass


_ERR = {-1: "GAVA_EINVAL (unsupported shape/alignment)", -2: "GAVA_EWORKSPACE (workspace too small)",
        -3: "GAVA_ELAUNCH (kernel launch failed)"}


def load():
    """Load libgava_hip.so; raises GavaError when it has not been built (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GavaError(f"{LIB_PATH} is missing: run `python -m gava_clip_amd.build` "
                        "(the HIP path has no fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.gava_abi_version.restype = C.c_int
    # the library reports the hash of the header it was compiled against; the ctypes mirrors in this file follow the header
    # in the tree (tests/test_host_cpu.py compares every struct size with the C compiler's): a stale .so would be called with
    # shifted structs, so it is refused here
    from .build import abi_hash, HEADER
    try:
        want = abi_hash()
    except OSError as e:
        raise GavaError(f"cannot read {os.path.normpath(HEADER)} ({e}): the ctypes mirrors in gava_clip_amd/hip.py are checked against "
                        "the ABI hash of that header, keep include/gava_hip.h next to the package") from None
    have = lib.gava_abi_version()
    if have != want:
        raise GavaError(f"{LIB_PATH} was built from another include/gava_hip.h (ABI {have:#x}, header in the tree {want:#x}): "
                        "rebuild with `python -m gava_clip_amd.build --force`")
    for name, args in (("gava_gemm", [C.POINTER(GemmArgs), _vp]),
                       ("gava_layernorm", [C.POINTER(LayerNormArgs), _vp]),
                       ("gava_attention", [C.POINTER(AttentionArgs), _vp])):
        f = getattr(lib, name)
        f.argtypes, f.restype = args, C.c_int
    lib.gava_attention_f32.argtypes, lib.gava_attention_f32.restype = [C.POINTER(AttentionF32Args), _vp], C.c_int
    lib.gava_vision_workspace_bytes.argtypes = [C.POINTER(VisionModel)]
    lib.gava_vision_workspace_bytes.restype = C.c_size_t
    lib.gava_vision_forward.argtypes = [C.POINTER(VisionModel), _fp, _fp, _fp, _fp, _vp, C.c_size_t, _vp]
    lib.gava_vision_forward.restype = C.c_int
    lib.gava_text_workspace_bytes.argtypes = [C.POINTER(TextModel)]
    lib.gava_text_workspace_bytes.restype = C.c_size_t
    lib.gava_text_forward.argtypes = [C.POINTER(TextModel), _ip, _fp, _ip, _fp, _vp, C.c_size_t, _vp]
    lib.gava_text_forward.restype = C.c_int
    lib.gava_similarity_head.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _vp]
    lib.gava_similarity_head.restype = C.c_int
    lib.gava_convert_h16.argtypes = [_fp, _vp, C.c_size_t, C.c_int, _vp]
    lib.gava_convert_h16.restype = C.c_int
    lib.gava_preprocess_clip.argtypes = [C.POINTER(PreprocessArgs), _vp]
    lib.gava_preprocess_clip.restype = C.c_int
    lib.gava_patchify.argtypes = [C.POINTER(PatchifyArgs), _vp]
    lib.gava_patchify.restype = C.c_int
    lib.gava_layernorm_backward.argtypes = [C.POINTER(LayerNormBwdArgs), _vp]
    lib.gava_layernorm_backward.restype = C.c_int
    lib.gava_qgelu_backward.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp]
    lib.gava_qgelu_backward.restype = C.c_int
    lib.gava_attention_backward.argtypes = [C.POINTER(AttentionBwdArgs), _vp]
    lib.gava_attention_backward.restype = C.c_int
    lib.gava_attention_backward_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.gava_attention_backward_workspace_bytes.restype = C.c_size_t
    lib.gava_clip_geometry.argtypes = [C.POINTER(ClipDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.gava_clip_geometry.restype = C.c_int
    lib.gava_gemm_aligned_walk.argtypes, lib.gava_gemm_aligned_walk.restype = [C.c_int, C.c_int, C.c_int], C.c_int
    lib.gava_probe_fc1_enable.argtypes = [C.c_int]
    lib.gava_probe_fc1_read.argtypes = [C.POINTER(C.c_float), C.c_int]
    lib.gava_row_stats.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, _vp]
    lib.gava_row_stats.restype = C.c_int
    lib.gava_vision_forward_keep.argtypes = [C.POINTER(VisionModel), _fp, _fp, _fp, C.POINTER(VisionSaved), _vp, C.c_size_t, _vp]
    lib.gava_vision_forward_keep.restype = C.c_int
    lib.gava_vision_forward_train.argtypes = [C.POINTER(VisionModel), _fp, _fp, _fp, _fp, _fp, _vp, C.c_size_t, _vp]
    lib.gava_vision_forward_train.restype = C.c_int
    lib.gava_text_forward_train.argtypes = [C.POINTER(TextModel), _ip, _fp, _ip, _fp, _fp, _vp, C.c_size_t, _vp]
    lib.gava_text_forward_train.restype = C.c_int
    lib.gava_vision_pair_stream.argtypes, lib.gava_vision_pair_stream.restype = [C.POINTER(VisionModel)], C.c_int
    # the header the library was compiled from against the mirrors above (gava_abi_version ties library and header)
    mirrors = [GemmArgs, LayerNormArgs, AttentionArgs, AttentionF32Args, ClipDesc, VisionLayer, VisionLayer8, VisionModel, TextLayer,
               TextModel, LayerNormBwdArgs, AttentionBwdArgs, VisionSaved, PreprocessArgs, PatchifyArgs]
    sizes = (C.c_size_t * len(mirrors))()
    lib.gava_struct_sizes.argtypes, lib.gava_struct_sizes.restype = [C.POINTER(C.c_size_t), C.c_int], C.c_int
    if lib.gava_struct_sizes(sizes, len(mirrors)) != len(mirrors):
        raise GavaError("libgava_hip.so and gava_clip_amd/hip.py disagree on the number of ABI structs")
    for cls, sz in zip(mirrors, sizes):
        if C.sizeof(cls) != sz:
            raise GavaError(f"ctypes mirror {cls.__name__} is {C.sizeof(cls)} bytes, the library's struct {sz}: gava_clip_amd/hip.py is out of "
                            f"step with include/gava_hip.h")
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        raise GavaError(f"{what} failed: {_ERR.get(code, code)}")


def stream_ptr(device=None):
    """torch's current HIP stream on `device` (default: the current device).  The drivers key their per-device launch
    context by the CURRENT device, so callers that may be handed a tensor of another device wrap the call in
    `torch.cuda.device(t.device)` (model.py does)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Device pointer of a CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda, "gava_clip_amd kernels take device tensors only"
    return C.c_void_p(t.data_ptr())


def h16_dtype(prec):
    return PREC_TORCH[prec]


# ---- thin per-op wrappers (used by the unit tests; the model uses the fused drivers) ----------

def gemm(A, W, bias, out, *, epilogue, prec, resid=None, scale_cols=0, scale=1.0,
         pos=None, time=None, n_patches=0, T=0, M=None, split_out=False, frames=None, frame_size=0, patch=0, aux=None,
         aux_prec=None, aux_out=None, x16_out=None, rowsum_out=None, fold_stats=None, fold_s=None, fold_t=None,
         cu_reserve=0, rowsum_reduced=False, fold_partials=None, clips=None, clip_lut=None, kernel=0, w_lo=0, K=None,
         A8=None, W8=None, w8_exp=0, out8=None, x8_out=None, resid16=None, resid_lo=None, xlo_out=None):
    a = GemmArgs()
    a.resid16, a.resid_lo, a.xlo_out = ptr(resid16), ptr(resid_lo), ptr(xlo_out)
    a.kernel = kernel
    a.w_lo, a.w8_exp = w_lo, w8_exp
    a.A8, a.lda8, a.W8, a.ldw8 = ptr(A8), (A8.stride(0) if A8 is not None else 0), ptr(W8), (W8.stride(0) if W8 is not None else 0)
    a.out8, a.ldo8 = ptr(out8), (out8.stride(0) if out8 is not None else 0)
    a.x8_out, a.ld_x8 = ptr(x8_out), (x8_out.stride(0) if x8_out is not None else 0)
    a.clips, a.clip_lut = ptr(clips), ptr(clip_lut)
    a.cu_reserve = cu_reserve
    a.rowsum_reduced, a.fold_partials = int(rowsum_reduced), ptr(fold_partials)
    a.x16_out, a.ld_x16, a.rowsum_out = ptr(x16_out), (x16_out.stride(0) if x16_out is not None else 0), ptr(rowsum_out)
    a.fold_stats, a.fold_s, a.fold_t = ptr(fold_stats), ptr(fold_s), ptr(fold_t)
    a.aux, a.aux_out = ptr(aux), ptr(aux_out)
    a.aux_prec = prec if aux_prec is None else aux_prec
    a.A, a.lda, a.W, a.ldw = ptr(A), (A.stride(0) if A is not None else W.stride(0)), ptr(W), W.stride(0)
    a.frames, a.frame_size, a.patch = ptr(frames), frame_size, patch
    a.bias, a.out, a.ldo = ptr(bias), ptr(out), (out.stride(0) if out is not None else 0)
    a.resid, a.ldr = ptr(resid), (resid.stride(0) if resid is not None else (resid16.stride(0) if resid16 is not None else 0))
    a.M, a.N, a.K = (A.shape[0] if M is None else M), W.shape[0], (W.shape[1] // (2 if w_lo == 1 else 1) if K is None else K)
    a.epilogue, a.prec, a.scale_cols, a.scale = epilogue, prec, scale_cols, scale
    a.pos, a.time, a.n_patches, a.T = ptr(pos), ptr(time), n_patches, T
    a.split_out = int(split_out)
    check(load().gava_gemm(C.byref(a), stream_ptr()), "gava_gemm")


def attention_f32(q, k, v, out, *, batch, heads, L, prec, causal=False, split_out=False, scale=0.125):
    a = AttentionF32Args()
    a.q, a.k, a.v, a.ld = ptr(q), ptr(k), ptr(v), q.stride(0)
    a.out, a.ld_out = ptr(out), out.stride(0)
    a.batch, a.heads, a.L, a.causal, a.prec, a.split_out, a.scale = batch, heads, L, int(causal), prec, int(split_out), scale
    check(load().gava_attention_f32(C.byref(a), stream_ptr()), "gava_attention_f32")


def layernorm(x, gamma, beta, *, out16=None, out32=None, prec, rows=None, in_stride=None, row_index=None,
              split_out=False, gamma2=None, beta2=None, out_hi=None, out_lo=None):
    a = LayerNormArgs()
    a.out_hi, a.out_lo, a.out_hl_stride = ptr(out_hi), ptr(out_lo), (out_hi.stride(0) if out_hi is not None else 0)
    a.gamma2, a.beta2 = ptr(gamma2), ptr(beta2)
    a.inp, a.in_stride, a.in_row_index = ptr(x), (x.stride(0) if in_stride is None else in_stride), ptr(row_index)
    a.gamma, a.beta = ptr(gamma), ptr(beta)
    a.out16, a.out16_stride = ptr(out16), (out16.stride(0) if out16 is not None else 0)
    a.out32, a.out32_stride = ptr(out32), (out32.stride(0) if out32 is not None else 0)
    a.rows, a.D, a.prec = (x.shape[0] if rows is None else rows), x.shape[-1], prec
    a.split_out = int(split_out)
    check(load().gava_layernorm(C.byref(a), stream_ptr()), "gava_layernorm")


def attention(q, k, v, out, *, batch, heads, n_q, n_kmain, prec, causal=False,
              side_k=None, side_v=None, n_g=0, T=0, has_summary=False, split_out=False, q_batch_rows=0):
    a = AttentionArgs()
    a.q, a.k, a.v, a.ld_qkv = ptr(q), ptr(k), ptr(v), q.stride(0)
    a.side_k, a.side_v = ptr(side_k), ptr(side_v)
    a.ld_side = side_k.stride(0) if side_k is not None else 0
    a.out, a.ld_out = ptr(out), out.stride(0)
    a.batch, a.heads, a.n_q, a.n_kmain = batch, heads, n_q, n_kmain
    a.n_g, a.T, a.has_summary, a.causal, a.prec = n_g, T, int(has_summary), int(causal), prec
    a.split_out = int(split_out)
    a.q_batch_rows, a.ld_q = q_batch_rows, (q.stride(0) if q_batch_rows else 0)
    check(load().gava_attention(C.byref(a), stream_ptr()), "gava_attention")


def split_pack_weight(w, prec):
    """[N][K] fp32 -> [N][3K] h16 = [W_hi | W_hi | W_lo], the weight side of the split-precision GEMM."""
    w = w.detach().float().contiguous()
    hi = convert_h16(w, prec)
    lo = convert_h16(w - hi.float(), prec)
    return torch.cat([hi, hi, lo], dim=1).contiguous()


def pack_w8(w, prec):
    """The 8-bit lo operand of a weight (gava_gemm_args.w_lo = 2): (hi16, W8, exp, row sums) with hi16 = h16(w), W8 = e4m3 bytes
    of 2^exp (w - hi16) in rows 4K bytes apart (K used), exp chosen so that the largest |w - hi16| lands in [128, 256), and
    the fp32 row sums of hi16 + 2^-exp W8."""
    import math
    w = w.detach().float().contiguous()
    hi = convert_h16(w, prec)
    lo = (w - hi.float()).cpu()
    mx = float(lo.abs().max())
    e = 7 - int(math.floor(math.log2(mx))) if mx > 0 else 0
    e = max(-100, min(100, e))
    q = (lo * (2.0 ** e)).to(torch.float8_e4m3fn)          # OCP e4m3 (gfx950), round to nearest even, on the host
    N, K = w.shape
    w8 = torch.zeros(N, 4 * K, dtype=torch.uint8)
    w8[:, :K] = q.view(torch.uint8)
    s8 = (hi.float().cpu().double() + q.to(torch.float32).double() * (2.0 ** -e)).sum(1).float()
    return hi, w8.to(w.device), e, s8.to(w.device).contiguous()


def convert_h16(x, prec):
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=h16_dtype(prec), device=x.device)
    check(load().gava_convert_h16(ptr(x), ptr(out), x.numel(), prec, stream_ptr()), "gava_convert_h16")
    return out


def clip_lut(mean, std, device):
    """fp32 [3][256] with (v/255 - mean[c]) / std[c] for every byte value, evaluated with the reference's own torch expression
    (video_dataset/dataset.py:119,121) on the host: the GPU kernels look the normalised value up."""
    v = torch.arange(256, dtype=torch.float32).view(1, 256) / 255.
    lut = (v - torch.tensor(mean, dtype=torch.float32).view(3, 1)) / torch.tensor(std, dtype=torch.float32).view(3, 1)
    return lut.contiguous().to(device)


def preprocess_clip(frames_u8, out, *, T, rate, size, mean, std, first_temporal_view=False, first_spatial_view=False, lut=None):
    """frames_u8: uint8 [n][H][W][3] (device); out: fp32 view [3][T][size][size] whose last two dims are contiguous."""
    assert frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4 and frames_u8.shape[-1] == 3 and frames_u8.is_contiguous()
    assert out.dtype == torch.float32 and tuple(out.shape) == (3, T, size, size)
    assert out.stride(3) == 1 and out.stride(2) == size, "output planes must be contiguous"
    a = PreprocessArgs()
    a.frames, a.n_frames, a.height, a.width = ptr(frames_u8), frames_u8.shape[0], frames_u8.shape[1], frames_u8.shape[2]
    a.mean, a.std = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    a.T, a.rate, a.size = T, rate, size
    a.out, a.out_stride_c, a.out_stride_t = ptr(out), out.stride(0), out.stride(1)
    a.first_temporal_view, a.first_spatial_view = int(first_temporal_view), int(first_spatial_view)
    a.lut = ptr(lut)
    check(load().gava_preprocess_clip(C.byref(a), stream_ptr()), "gava_preprocess_clip")


def patchify(out, *, B, T, size, patch, prec, x=None, clips=None, clip_lut=None):
    """16-bit patch matrix [B*T*(size/patch)^2, ldo] of the fp32 clips `x` (B,3,T,size,size) or of the uint8 videos behind
    `clips` (clip_descriptors): the A operand of the patch-embedding GEMM (gemm(A, ..., epilogue=EPI_F32_PATCH))."""
    a = PatchifyArgs()
    a.x, a.clips, a.clip_lut = ptr(x), ptr(clips), ptr(clip_lut)
    a.B, a.T, a.size, a.patch, a.prec = B, T, size, patch, prec
    a.out, a.ldo = ptr(out), out.stride(0)
    check(load().gava_patchify(C.byref(a), stream_ptr()), "gava_patchify")


# ---- backward ops ------------------------------------------------------------

def layernorm_backward(x, gamma, dy, dx, *, accumulate=False, x_row_index=None, dx_row_index=None, rows=None,
                       dgamma=None, dbeta=None, dx16=None, prec=PREC_BF16):
    a = LayerNormBwdArgs()
    a.dx16, a.dx16_stride, a.prec = ptr(dx16), (dx16.stride(0) if dx16 is not None else 0), prec
    a.x, a.x_stride, a.x_row_index = ptr(x), x.stride(0), ptr(x_row_index)
    a.gamma, a.dy, a.dy_stride = ptr(gamma), ptr(dy), dy.stride(0)
    a.dx, a.dx_stride, a.dx_row_index = ptr(dx), dx.stride(0), ptr(dx_row_index)
    a.dgamma, a.dbeta = ptr(dgamma), ptr(dbeta)
    a.rows, a.D, a.accumulate = (dy.shape[0] if rows is None else rows), x.shape[-1], int(accumulate)
    check(load().gava_layernorm_backward(C.byref(a), stream_ptr()), "gava_layernorm_backward")


def qgelu_backward(pre, dh, dpre, prec):
    assert pre.is_contiguous() and dh.is_contiguous() and dpre.is_contiguous() and pre.numel() == dh.numel() == dpre.numel()
    check(load().gava_qgelu_backward(ptr(pre), ptr(dh), ptr(dpre), pre.numel(), prec, stream_ptr()), "gava_qgelu_backward")


def attention_backward(q, k, v, dout, dq, dk, dv, *, batch, heads, n, prec, causal=False, q_scale=1.0,
                       side_k=None, side_v=None, dside_k=None, dside_v=None, n_g=0, T=0, has_summary=False, n_q=0,
                       act_prec=None, q_batch_rows=0):
    a = AttentionBwdArgs()
    a.q_batch_rows, a.ld_q, a.ld_dq = q_batch_rows, (q.stride(0) if q_batch_rows else 0), (dq.stride(0) if q_batch_rows else 0)
    a.act_prec_set, a.act_prec = int(act_prec is not None), (act_prec if act_prec is not None else prec)
    a.side_k, a.side_v = ptr(side_k), ptr(side_v)
    a.ld_side = side_k.stride(0) if side_k is not None else 0
    a.dside_k, a.dside_v = ptr(dside_k), ptr(dside_v)
    a.ld_dside = dside_k.stride(0) if dside_k is not None else 0
    a.n_g, a.T, a.has_summary, a.n_q = n_g, T, int(has_summary), n_q
    ws = torch.empty(load().gava_attention_backward_workspace_bytes(batch, heads, n_q or n), dtype=torch.uint8, device=q.device)
    a.workspace = ptr(ws)
    a.q, a.k, a.v, a.ld_qkv = ptr(q), ptr(k), ptr(v), k.stride(0)
    a.dout, a.ld_dout = ptr(dout), dout.stride(0)
    a.dq, a.dk, a.dv, a.ld_dqkv = ptr(dq), ptr(dk), ptr(dv), dk.stride(0)
    a.batch, a.heads, a.n, a.causal, a.prec, a.q_scale = batch, heads, n, int(causal), prec, q_scale
    check(load().gava_attention_backward(C.byref(a), stream_ptr()), "gava_attention_backward")


def row_stats(rowsum, D):
    """float2 partial sums [rows][slots] of the folding producers -> (mean, rstd) float2 [rows]."""
    rows, slots = rowsum.shape[0], rowsum.shape[1]
    stats = torch.empty(rows, 2, dtype=torch.float32, device=rowsum.device)
    check(load().gava_row_stats(ptr(rowsum), slots, D, rows, ptr(stats), stream_ptr()), "gava_row_stats")
    return stats


def clip_descriptors(videos, *, T, rate, size, first_temporal_view=False, first_spatial_view=False):
    """uint8 videos [n_i][H_i][W_i][3] on the device -> (uint8 device tensor holding an array of gava_clip_desc, keep-alive
    list).  The geometry is computed by the library (gava_clip_geometry: dataset.py:124-129,163-186)."""
    lib = load()
    arr = (ClipDesc * len(videos))()
    for i, v in enumerate(videos):
        assert v.is_cuda and v.dtype == torch.uint8 and v.dim() == 4 and v.shape[-1] == 3 and v.is_contiguous()
        arr[i].frames, arr[i].n_frames, arr[i].height, arr[i].width = ptr(v), v.shape[0], v.shape[1], v.shape[2]
        check(lib.gava_clip_geometry(C.byref(arr[i]), T, rate, size, int(first_temporal_view), int(first_spatial_view)),
              "gava_clip_geometry")
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(videos[0].device), list(videos)
