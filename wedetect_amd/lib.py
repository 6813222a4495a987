"""ctypes binding of libwedetect_hip.so (declared in include/wedetect_hip.h).

There is NO fallback: if the library is not built, importing this module raises
``WedetectHipMissing``; every non-zero status raises ``WedetectHipError``.  Tensors are
passed as raw device pointers (``tensor.data_ptr()``) and the caller's current HIP
stream; torch is used only for memory and streams.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# $WEDETECT_LIB: another build of the same ABI (same-box A/B runs of two kernel versions); default = the in-tree library
LIB_PATH = os.environ.get("WEDETECT_LIB") or os.path.join(_HERE, "libwedetect_hip.so")

ACT_NONE, ACT_RELU, ACT_SILU, ACT_GELU = 0, 1, 2, 3
OUT_ROWS, OUT_DECONV2X2 = 0, 1
SPLIT_A, SPLIT_C = 1, 2
ABI_VERSION = 14
NMS_VANILLA, NMS_TORCHVISION, NMS_MMCV = 0, 1, 2
# torchvision/ops/boxes.py batched_nms: the per-class loop (_batched_nms_vanilla) above this many box coordinates
# (boxes.numel()).  4000 / 20000 are the values of torchvision 0.15 ... 0.21 (the releases contemporary with the
# reference's torch 2.x pins; newer releases raised the GPU limit to 100000 — set nms_param explicitly to reproduce those).
TV_TRICK_MAX_NUMEL = {"cpu": 4000, "cuda": 20000}
MMCV_SPLIT_THR = 10000                                 # mmcv/ops/nms.py batched_nms: per-class loop from this many candidates

EXPORTS = (
    "wd_abi_version", "wd_strerror", "wd_sizeof_conv_gemm", "wd_conv_gemm", "wd_conv_gemm_tuned", "wd_conv_gemm_config", "wd_stem_patchify", "wd_dwconv7", "wd_dwconv7_variant",
    "wd_layernorm_rows", "wd_l2norm_rows", "wd_dfl_decode", "wd_topk_workspace_bytes", "wd_topk_capacity",
    "wd_topk_candidates", "wd_nms_workspace_bytes", "wd_nms_gather", "wd_retrieval_max",
    "wd_split_weights_bytes", "wd_split_weights", "wd_split_weights_padded", "wd_dwconv7_stats", "wd_ln_stats_finalize", "wd_conv_gemm_split", "wd_conv_gemm_split_ws", "wd_conv_gemm_split_config", "wd_layernorm_rows_split", "wd_layernorm_rows_split_s2d", "wd_letterbox_u8", "wd_retrieval_max_split", "wd_similarity_split", "wd_mlp_fused_wide_ln", "wd_text_embed", "wd_attention_small", "wd_recall_scratch_floats", "wd_recall_match",
    "wd_max_sigmoid_attn", "wd_adaptive_maxpool_nhwc", "wd_cross_attention_small", "wd_time_next_gemm",
    "wd_cv_resize_paste_u8", "wd_chw_to_hwc_u8", "wd_p8_workspace_bytes", "wd_dwconv7_ln", "wd_probe_lds_dma", "wd_probe_issue", "wd_mlp_fused_split", "wd_mlp_fused_wide", "wd_stem_fused",
)


class WedetectHipMissing(ImportError):
    pass


class WedetectHipError(RuntimeError):
    pass


class ConvGemm(C.Structure):
    """Mirror of ``struct WdConvGemm``."""
    _fields_ = [
        ("a", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p), ("c", C.c_void_p),
        ("batch", C.c_int32), ("hin", C.c_int32), ("win", C.c_int32), ("cin", C.c_int32), ("lda", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("hout", C.c_int32), ("wout", C.c_int32),
        ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32),
        ("ldc", C.c_int32), ("ldres", C.c_int32),
        ("act", C.c_int32), ("out_mode", C.c_int32),
        ("res_alpha", C.c_float), ("out_scale", C.c_float), ("out_bias", C.c_float),
        ("sigmoid", C.c_int32),
        ("c_batch_stride", C.c_int32),
        ("seg_rows", C.c_int32), ("seg_end0", C.c_int32), ("seg_end1", C.c_int32),
        ("seg_scale", C.c_float * 3), ("seg_bias", C.c_float * 3),
        ("range_flag", C.c_void_p),
        ("c2", C.c_void_p), ("ldc2", C.c_int32),
        ("a_scale", C.c_float), ("c_split_scale", C.c_float),
        ("ln_stats", C.c_void_p), ("ln_u", C.c_void_p),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise WedetectHipMissing(
            f"{LIB_PATH} not found: build it with `python -m wedetect_amd.build` (hipcc, gfx950). "
            "wedetect_amd has no CPU or PyTorch fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise WedetectHipMissing(f"{LIB_PATH} does not export {name}")
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.wd_abi_version.restype = C.c_int
    lib.wd_strerror.restype = C.c_char_p
    lib.wd_strerror.argtypes = [C.c_int]
    lib.wd_conv_gemm.argtypes = [C.POINTER(ConvGemm), vp]
    lib.wd_conv_gemm_tuned.argtypes = [C.POINTER(ConvGemm), i32, vp]
    lib.wd_conv_gemm_config.restype = C.c_char_p
    lib.wd_conv_gemm_config.argtypes = [i32, i32, i32]
    lib.wd_stem_patchify.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.wd_dwconv7.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.wd_dwconv7_variant.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.wd_layernorm_rows.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, f32, vp]
    lib.wd_l2norm_rows.argtypes = [vp, vp, i64, i32, vp]
    lib.wd_dfl_decode.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.wd_topk_workspace_bytes.restype = i64
    lib.wd_topk_workspace_bytes.argtypes = [i32, i64, i32]
    lib.wd_topk_capacity.restype = i32
    lib.wd_topk_capacity.argtypes = [i32]
    lib.wd_topk_candidates.argtypes = [vp, i32, i64, f32, i32, vp, vp, vp, vp, i64, vp]
    lib.wd_nms_workspace_bytes.restype = i64
    lib.wd_nms_workspace_bytes.argtypes = [i32]
    lib.wd_nms_gather.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp, f32, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, i64, vp]
    lib.wd_retrieval_max.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.wd_split_weights_bytes.restype = i64
    lib.wd_split_weights_bytes.argtypes = [i32, i32]
    lib.wd_split_weights.argtypes = [vp, i32, i32, f32, vp, vp]
    lib.wd_split_weights_padded.argtypes = [vp, i32, i32, f32, vp, vp]
    lib.wd_dwconv7_stats.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]
    lib.wd_similarity_split.argtypes = [vp, i64, vp, f32, vp, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), i32, vp, vp]
    lib.wd_ln_stats_finalize.argtypes = [vp, vp, i64, i32, f32, vp]
    lib.wd_conv_gemm_split.argtypes = [C.POINTER(ConvGemm), vp, f32, i32, i32, vp]
    lib.wd_conv_gemm_split_ws.argtypes = [C.POINTER(ConvGemm), vp, f32, i32, i32, vp, i64, i32, vp]
    lib.wd_layernorm_rows_split.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, f32, vp]
    lib.wd_layernorm_rows_split_s2d.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]
    lib.wd_retrieval_max_split.argtypes = [vp, vp, f32, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.wd_recall_scratch_floats.restype = i64
    lib.wd_recall_scratch_floats.argtypes = [i32, i32]
    lib.wd_recall_match.argtypes = [vp, vp, vp, vp, i32, vp, i32, vp, i64, vp, i32, i32, vp]
    lib.wd_text_embed.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, vp]
    lib.wd_attention_small.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.wd_max_sigmoid_attn.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.wd_adaptive_maxpool_nhwc.argtypes = [vp, i32, vp, i32, i64, i32, i32, i32, i32, i32, vp]
    lib.wd_cross_attention_small.argtypes = [vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.wd_time_next_gemm.argtypes = [vp, vp]
    lib.wd_letterbox_u8.argtypes = [vp, i32, i32, vp, vp, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.wd_cv_resize_paste_u8.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.wd_chw_to_hwc_u8.argtypes = [vp, i32, vp, i32, i32, i32, vp]
    lib.wd_dwconv7_ln.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]
    lib.wd_mlp_fused_split.argtypes = [vp, i64, i32, i32, vp, f32, vp, vp, f32, vp, vp, f32, vp, vp]
    lib.wd_mlp_fused_wide.argtypes = [vp, i64, i32, i32, vp, f32, vp, vp, f32, vp, vp, f32, vp, vp, i64, vp]
    lib.wd_mlp_fused_wide_ln.argtypes = [vp, i64, i32, i32, vp, f32, vp, vp, vp, vp, f32, vp, vp, f32, vp, vp, i64, vp]
    lib.wd_stem_fused.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, i32, f32, vp, vp]
    lib.wd_probe_issue.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp]
    lib.wd_probe_lds_dma.argtypes = [vp, i64, i32, i32, i32, i32, vp, vp]
    lib.wd_p8_workspace_bytes.restype = i64
    lib.wd_p8_workspace_bytes.argtypes = []
    lib.wd_conv_gemm_split_config.restype = C.c_char_p
    lib.wd_conv_gemm_split_config.argtypes = [i32, i32, i32, i32]
    if lib.wd_sizeof_conv_gemm() != C.sizeof(ConvGemm):
        raise WedetectHipMissing("struct WdConvGemm layout differs between the library and lib.ConvGemm; rebuild")
    if lib.wd_abi_version() != ABI_VERSION:
        raise WedetectHipMissing(f"ABI mismatch: library {lib.wd_abi_version()} vs binding {ABI_VERSION}; rebuild")
    return lib


LIB = _load()


def check(status: int, what: str) -> None:
    if status != 0:
        raise WedetectHipError(f"{what} failed: {LIB.wd_strerror(status).decode()} ({status})")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t) -> int:
    return 0 if t is None else t.data_ptr()


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_cuda:
        raise WedetectHipError(f"{name}: expected a CUDA/HIP float32 tensor, got {t.dtype} on {t.device}")
    return t


# ------------------------------------------------------------------------------------------
# thin typed wrappers (pointer plumbing only)
# ------------------------------------------------------------------------------------------
def conv_gemm(a, w, bias, c, *, batch, hin, win, cin, lda, kh=1, kw=1, stride=1, pad=0, hout=None, wout=None,
              n, ldc, act=ACT_NONE, res=None, ldres=0, res_alpha=1.0, out_mode=OUT_ROWS,
              out_scale=1.0, out_bias=0.0, sigmoid=False, c_batch_stride=0, seg=None, tuned_cfg=None,
              w_split=None, split_cfg=-1, split_flags=0, workspace=None, k_splits=0, range_flag=None, c2=None, ldc2=0, a_scale=1.0, c_split_scale=1.0,
              ln_stats=None, ln_u=None) -> None:
    """``seg`` = (seg_rows, seg_end0, seg_end1, (s0, s1, s2), (b0, b1, b2)) or None.
    ``w_split`` = (split weight buffer, unscale) from :func:`split_weights` selects the fp16x3 kernel;
    ``split_flags`` = SPLIT_A / SPLIT_C: activations / output stored as fp16 hi/lo groups; ``workspace`` (a device
    tensor) lets under-filled launches split K (``k_splits`` = 0: library's choice, > 0: forced)."""
    hout = (hin + 2 * pad - kh) // stride + 1 if hout is None else hout
    wout = (win + 2 * pad - kw) // stride + 1 if wout is None else wout
    p = ConvGemm(a=_p(a), w=_p(w), bias=_p(bias), res=_p(res), c=_p(c), batch=batch, hin=hin, win=win, cin=cin,
                 lda=lda, kh=kh, kw=kw, stride=stride, pad=pad, hout=hout, wout=wout,
                 m=batch * hout * wout, n=n, k=kh * kw * cin, ldc=ldc, ldres=ldres, act=act, out_mode=out_mode,
                 res_alpha=res_alpha, out_scale=out_scale, out_bias=out_bias, sigmoid=int(bool(sigmoid)),
                 c_batch_stride=c_batch_stride, range_flag=_p(range_flag), c2=_p(c2), ldc2=ldc2, a_scale=a_scale, c_split_scale=c_split_scale,
                 ln_stats=_p(ln_stats), ln_u=_p(ln_u))
    if seg is not None:
        p.seg_rows, p.seg_end0, p.seg_end1 = int(seg[0]), int(seg[1]), int(seg[2])
        p.seg_scale = (C.c_float * 3)(*[float(v) for v in seg[3]])
        p.seg_bias = (C.c_float * 3)(*[float(v) for v in seg[4]])
    if w_split is not None:
        if workspace is not None:
            check(LIB.wd_conv_gemm_split_ws(C.byref(p), _p(w_split[0]), float(w_split[1]), int(split_flags), int(split_cfg),
                                            _p(workspace), workspace.numel() * workspace.element_size(), int(k_splits),
                                            stream_ptr()), f"wd_conv_gemm_split_ws[{split_cfg}]")
            return
        check(LIB.wd_conv_gemm_split(C.byref(p), _p(w_split[0]), float(w_split[1]), int(split_flags), int(split_cfg),
                                     stream_ptr()),
              f"wd_conv_gemm_split[{split_cfg}]")
        return
    if tuned_cfg is not None:
        check(LIB.wd_conv_gemm_tuned(C.byref(p), int(tuned_cfg), stream_ptr()), f"wd_conv_gemm_tuned[{tuned_cfg}]")
        return
    check(LIB.wd_conv_gemm(C.byref(p), stream_ptr()), "wd_conv_gemm")


def mlp_fused_supported(rows: int, c: int, hidden: int) -> bool:
    return c == 128 and hidden == 512 and rows > 0 and rows % 128 == 0


MLP_WIDE_WIDTHS = (256, 512)
MLP_WIDE_FOLD_WIDTHS = (256,)          # wd_mlp_fused_wide_ln


def mlp_wide_supported(rows: int, c: int, hidden: int) -> bool:
    return c in MLP_WIDE_WIDTHS and hidden == 4 * c and rows > 0 and rows % 128 == 0


def mlp_wide_pack(w_split: torch.Tensor, n: int, k: int) -> torch.Tensor:
    """wd_split_weights buffer of an [n][k] matrix ([n][k / 8][hi x8 | lo x8] 32-byte groups) -> FRAGMENT-MAJOR order for
    wd_mlp_fused_wide: [n / 32][k / 16][2 (hi, lo)][64 lanes = (k half) * 32 + (row in block)][16 B] — one MFMA operand of a
    wave is one contiguous 1 KB.  Pack time only."""
    if n % 32 or k % 16 or w_split.numel() != n * k * 4:
        raise WedetectHipError(f"mlp_wide_pack: [{n}][{k}] is not a whole number of 32 x 16 fragments")
    v = w_split.view(n // 32, 32, k // 16, 2, 2, 16)               # [block, row, k16 step, k half, part, bytes]
    return v.permute(0, 2, 4, 3, 1, 5).contiguous().view(-1)


def mlp_fused_wide(a_split, rows, c, hidden, w1_frag, b1, w2_frag, b2, x, hid_scale=1.0, range_flag=None, workspace=None) -> None:
    """x <- x + W2 GELU(W1 a + b1) + b2 in one kernel for c = 256 / 512 (wd_mlp_fused_wide); ``w*_frag`` = (fragment-major
    buffer from mlp_wide_pack, unscale) pairs.  ``workspace``: a zero-initialised fp32 tensor of p8_workspace_bytes() bytes
    selects the persistent form when there are more row blocks than CUs."""
    ws_bytes = 0 if workspace is None else workspace.numel() * workspace.element_size()
    check(LIB.wd_mlp_fused_wide(_p(a_split), rows, c, hidden, _p(w1_frag[0]), float(w1_frag[1]), _p(b1), _p(w2_frag[0]),
                                float(w2_frag[1]), _p(b2), _p(x), float(hid_scale), _p(range_flag), _p(workspace), ws_bytes,
                                stream_ptr()),
          "wd_mlp_fused_wide")


def mlp_fused_wide_ln(d_split, rows, c, hidden, w1g_frag, v, u, ln_stats, w2_frag, b2, x, hid_scale=1.0, range_flag=None, workspace=None) -> None:
    """The wide one-kernel block MLP with the block's LayerNorm folded into pwconv1 (wd_mlp_fused_wide_ln, c = 256):
    x <- x + W2 GELU(rstd (W1g d - mean u) + v) + b2; ``d_split`` / ``ln_stats`` from dwconv7_stats / ln_stats_finalize."""
    ws_bytes = 0 if workspace is None else workspace.numel() * workspace.element_size()
    check(LIB.wd_mlp_fused_wide_ln(_p(d_split), rows, c, hidden, _p(w1g_frag[0]), float(w1g_frag[1]), _p(v), _p(u), _p(ln_stats),
                                   _p(w2_frag[0]), float(w2_frag[1]), _p(b2), _p(x), float(hid_scale), _p(range_flag), _p(workspace),
                                   ws_bytes, stream_ptr()), "wd_mlp_fused_wide_ln")


def mlp_fused(a_split, rows, c, hidden, w1_split, b1, w2_split, b2, x, hid_scale=1.0, range_flag=None) -> None:
    """x <- x + W2 GELU(W1 a + b1) + b2 in one kernel (wd_mlp_fused_split); ``w*_split`` = (buffer, unscale) pairs."""
    check(LIB.wd_mlp_fused_split(_p(a_split), rows, c, hidden, _p(w1_split[0]), float(w1_split[1]), _p(b1), _p(w2_split[0]),
                                 float(w2_split[1]), _p(b2), _p(x), float(hid_scale), _p(range_flag), stream_ptr()),
          "wd_mlp_fused_split")


def split_weights(w: torch.Tensor):
    """fp32 weight rows [n, k] -> (fp16 hi/lo buffer for wd_conv_gemm_split, unscale).

    The weights are scaled by the power of two that brings max|w| just under 2^14 before
    splitting (keeps the low halves out of the fp16 subnormal range); ``unscale`` undoes it in
    the GEMM epilogue.  One device sync (max|w|): pack-time only."""
    _f32(w, "w")
    n, k = w.shape
    w = w.contiguous()
    amax = float(w.abs().max())
    scale = 1.0 if amax == 0.0 else 2.0 ** (13 - math.floor(math.log2(amax)))
    out = torch.empty(LIB.wd_split_weights_bytes(n, k), dtype=torch.uint8, device=w.device)
    check(LIB.wd_split_weights_padded(_p(w), n, k, scale, _p(out), stream_ptr()), "wd_split_weights_padded")   # zero rows up to a multiple of 8
    return out, 1.0 / scale


def p8_workspace_bytes() -> int:
    """Bytes of the park workspace of the persistent 256 x 256 fp16x3 kernel (needs a device: CU count)."""
    return int(LIB.wd_p8_workspace_bytes())


def gemm_config(m: int, n: int, k: int, split: bool = False, conv: bool = False, presplit: bool = False,
                park: bool = False, dma: bool = False, conv3: bool = False) -> str:
    """``dma``: a pre-split layer outside the plain-row 1x1 case (k x k / strided conv, scatter or batch-stride output,
    residual or dual-format output of a SPLIT_C layer): the implicit-GEMM LDS-DMA kernel of split_gemm_conv.hip; ``conv3``:
    3 x 3 / stride 1 / pad 1 among those — the row-sharing kernels of split_gemm_conv3.hip."""
    if split and dma and conv3:
        return LIB.wd_conv_gemm_split_config(m, n, k, 5).decode()
    if split and dma:
        return LIB.wd_conv_gemm_split_config(m, n, k, 4).decode()
    if split:
        return LIB.wd_conv_gemm_split_config(m, n, k, 1 if conv else ((3 if park else 2) if presplit else 0)).decode()
    return LIB.wd_conv_gemm_config(m, n, k).decode()


def stem_patchify(img_u8: torch.Tensor, out: torch.Tensor) -> None:
    b, h, w, _ = img_u8.shape
    check(LIB.wd_stem_patchify(_p(img_u8), _p(out), b, h, w, stream_ptr()), "wd_stem_patchify")


STEM_FUSED_WIDTHS = (64, 96, 128, 192)


def stem_fused(img_u8: torch.Tensor, wgt, bias, gamma, beta, out, eps=1e-6) -> None:
    """uint8 NHWC image -> / 255 -> 4 x 4 stride-4 conv + bias -> LayerNorm -> fp32 rows, one kernel (wd_stem_fused)."""
    b, h, w, _ = img_u8.shape
    check(LIB.wd_stem_fused(_p(img_u8), b, h, w, _p(wgt), _p(bias), _p(gamma), _p(beta), wgt.shape[0], eps, _p(out),
                            stream_ptr()), "wd_stem_fused")


def dwconv7(x, w7, bias, y, batch, h, w, c, variant: int = 0) -> None:
    """variant != 0 pins the kernel form (wd_dwconv7_variant; all forms are bit-identical)."""
    if variant:
        check(LIB.wd_dwconv7_variant(_p(x), _p(w7), _p(bias), _p(y), batch, h, w, c, variant, stream_ptr()), "wd_dwconv7_variant")
        return
    check(LIB.wd_dwconv7(_p(x), _p(w7), _p(bias), _p(y), batch, h, w, c, stream_ptr()), "wd_dwconv7")


def dwconv7_stats(x, w7, bias, y_split, part, batch, h, w, c, scale=1.0) -> None:
    """Depthwise 7 x 7 of a block whose LayerNorm is folded into the following GEMM: ``y_split`` <- d * scale as fp16 hi/lo groups,
    ``part`` [c/32, batch*h*w, 2] <- per-block (mean, centred sum of squares) of d (wd_dwconv7_stats)."""
    check(LIB.wd_dwconv7_stats(_p(x), _p(w7), _p(bias), _p(y_split), _p(part), batch, h, w, c, float(scale), stream_ptr()), "wd_dwconv7_stats")


def ln_stats_finalize(part, stats, rows, c, eps=1e-6) -> None:
    """``stats`` [rows, 2] <- (mean, 1 / sqrt(var + eps)) over all c channels from the per-block partials."""
    check(LIB.wd_ln_stats_finalize(_p(part), _p(stats), rows, c, float(eps), stream_ptr()), "wd_ln_stats_finalize")


def layernorm_rows(x, y, gamma, beta, rows, c, ldx=None, ldy=None, eps=1e-6, split=False) -> None:
    """``split``: write y as fp16 hi/lo groups for a ``conv_gemm(..., split_flags=SPLIT_A)`` consumer."""
    fn = LIB.wd_layernorm_rows_split if split else LIB.wd_layernorm_rows
    check(fn(_p(x), _p(y), _p(gamma), _p(beta), rows, c, ldx or c, ldy or c, eps, stream_ptr()),
          "wd_layernorm_rows_split" if split else "wd_layernorm_rows")


def layernorm_rows_split_s2d(x, y, gamma, beta, batch, h, w, c, eps=1e-6) -> None:
    """LayerNorm over the channels of an NHWC map, written as fp16 hi/lo groups in SPACE-TO-DEPTH order: the
    [batch * h/2 * w/2, 4 c] GEMM rows of a 2 x 2 / stride-2 convolution ((kh, kw, cin) column order)."""
    check(LIB.wd_layernorm_rows_split_s2d(_p(x), _p(y), _p(gamma), _p(beta), batch, h, w, c, eps, stream_ptr()),
          "wd_layernorm_rows_split_s2d")


def letterbox_u8(src, h, w, bounds_h, kk_h, ksize_h, bounds_v, kk_v, ksize_v, tmp, dst, dst_h, dst_w, new_w, new_h,
                 left, top, fill) -> None:
    check(LIB.wd_letterbox_u8(_p(src), h, w, _p(bounds_h), _p(kk_h), ksize_h, _p(bounds_v), _p(kk_v), ksize_v, _p(tmp),
                              _p(dst), dst_h, dst_w, new_w, new_h, left, top, int(fill[0]), int(fill[1]), int(fill[2]),
                              stream_ptr()), "wd_letterbox_u8")


CVRESIZE_COPY, CVRESIZE_AREA_FAST, CVRESIZE_AREA, CVRESIZE_LINEAR = 0, 1, 2, 3


def cv_resize_paste_u8(src, sh, sw, mode, xa, xidx, xw, ya, yidx, yw, p0, p1, p2, dst, dst_h, dst_w, new_h, new_w, top, left,
                       fill=114, swap_rb=False) -> None:
    check(LIB.wd_cv_resize_paste_u8(_p(src), sh, sw, mode, _p(xa), _p(xidx), _p(xw), _p(ya), _p(yidx), _p(yw), int(p0), int(p1),
                                    float(p2), _p(dst), dst_h, dst_w, new_h, new_w, top, left, int(fill), int(bool(swap_rb)),
                                    stream_ptr()), "wd_cv_resize_paste_u8")


def chw_to_hwc_u8(src, dst) -> None:
    """src [B, 3, H, W] uint8 / float32 (BGR, 0..255) -> dst [B, H, W, 3] uint8 (RGB)."""
    b, c, h, w = src.shape
    if c != 3 or tuple(dst.shape) != (b, h, w, 3) or dst.dtype != torch.uint8 or src.dtype not in (torch.uint8, torch.float32):
        raise WedetectHipError("chw_to_hwc_u8: src [B,3,H,W] uint8|float32, dst [B,H,W,3] uint8")
    if not src.is_contiguous() or not dst.is_contiguous():
        raise WedetectHipError("chw_to_hwc_u8: contiguous tensors only")
    check(LIB.wd_chw_to_hwc_u8(_p(src), int(src.dtype == torch.float32), _p(dst), b, h, w, stream_ptr()), "wd_chw_to_hwc_u8")


def dwconv7_ln(x, w7, bias, y, gamma, beta, batch, h, w, c, eps=1e-6, split=False) -> None:
    """Depthwise 7x7 + LayerNorm fused (bit-identical to dwconv7 then layernorm_rows in place)."""
    check(LIB.wd_dwconv7_ln(_p(x), _p(w7), _p(bias), _p(y), _p(gamma), _p(beta), batch, h, w, c, float(eps), int(bool(split)),
                            stream_ptr()), "wd_dwconv7_ln")


def l2norm_rows(x, y) -> None:
    rows, c = x.shape
    check(LIB.wd_l2norm_rows(_p(x), _p(y), rows, c, stream_ptr()), "wd_l2norm_rows")


def dfl_decode(dist, ld, boxes, batch, hl, wl, stride, anchor_off, anchors_total) -> None:
    check(LIB.wd_dfl_decode(_p(dist), ld, _p(boxes), batch, hl, wl, stride, anchor_off, anchors_total, stream_ptr()),
          "wd_dfl_decode")


def topk_capacity(nms_pre: int) -> int:
    return LIB.wd_topk_capacity(nms_pre)


def topk_workspace_bytes(batch: int, n: int, nms_pre: int) -> int:
    return LIB.wd_topk_workspace_bytes(batch, n, nms_pre)


def topk_candidates(scores, batch, n, thr, nms_pre, out_idx, out_score, out_count, workspace) -> None:
    check(LIB.wd_topk_candidates(_p(scores), batch, n, thr, nms_pre, _p(out_idx), _p(out_score), _p(out_count),
                                 _p(workspace), workspace.numel() * workspace.element_size(), stream_ptr()),
          "wd_topk_candidates")


def nms_workspace_bytes(batch: int) -> int:
    return LIB.wd_nms_workspace_bytes(batch)


def nms_threshold(iou_thr: float, nms_mode: int, device_kind: str = "cpu") -> float:
    """The fp32 value whose ``ovr > value`` decides like the library's own comparison: mmcv's nms_cpu takes a C++
    ``float`` (nearest); torchvision's CPU nms_kernel_impl compares the fp32 IoU with a C++ ``double``, i.e. like the
    largest fp32 <= the Python value (include/wedetect_hip.h, wd_nms_gather); torchvision's GPU kernel (nms_kernel.cu)
    takes the threshold as a ``float`` (``device_kind`` "cuda": nearest, like mmcv)."""
    import numpy as np
    if device_kind not in TV_TRICK_MAX_NUMEL:
        raise ValueError(f"device_kind must be one of {sorted(TV_TRICK_MAX_NUMEL)}, not {device_kind!r}")
    t = np.float32(iou_thr)
    if nms_mode == NMS_TORCHVISION and device_kind == "cpu" and float(t) > float(iou_thr):
        t = np.nextafter(t, np.float32(-np.inf), dtype=np.float32)
    return float(t)


def nms_gather(cand_idx, cand_score, cand_count, cand_stride, boxes, n_anchor, k, meta, iou_thr, max_out,
               embed, embed_dim, out_boxes, out_scores, out_labels, out_anchors, out_count, out_embed, batch,
               nms_mode: int = NMS_VANILLA, mode_param: int = 0, workspace=None) -> None:
    ws_bytes = 0 if workspace is None else workspace.numel() * workspace.element_size()
    check(LIB.wd_nms_gather(_p(cand_idx), _p(cand_score), _p(cand_count), cand_stride, _p(boxes), n_anchor, k,
                            _p(meta), iou_thr, max_out, nms_mode, mode_param, _p(embed), embed_dim, _p(out_boxes),
                            _p(out_scores), _p(out_labels), _p(out_anchors), _p(out_count), _p(out_embed), batch,
                            _p(workspace), ws_bytes, stream_ptr()),
          "wd_nms_gather")


def retrieval_max(e, t, scale, bias, count, out, n_img, rows_per_img, n_cls, dim) -> None:
    check(LIB.wd_retrieval_max(_p(e), _p(t), _p(scale), _p(bias), _p(count), _p(out), n_img, rows_per_img, n_cls, dim,
                               stream_ptr()), "wd_retrieval_max")


def retrieval_max_split(e, t_split, scale, bias, count, out, n_img, rows_per_img, n_cls, dim, range_flag=None) -> None:
    """fp16x3 variant of :func:`retrieval_max`.  ``e``: fp32 region rows [n_img, rows, dim] (split here, scale
    1: a few MB); ``t_split``: ``split_weights(bank)`` prepared once for the (possibly class-sharded) bank.
    ``range_flag``: int32 [1] device tensor, set to 1 when an operand left the fp16 range (inf / NaN accumulator)."""
    _f32(e, "e")
    rows = e.reshape(-1, dim).contiguous()
    es = torch.empty(LIB.wd_split_weights_bytes(rows.shape[0], dim), dtype=torch.uint8, device=e.device)
    check(LIB.wd_split_weights_padded(_p(rows), rows.shape[0], dim, 1.0, _p(es), stream_ptr()), "wd_split_weights_padded")
    check(LIB.wd_retrieval_max_split(_p(es), _p(t_split[0]), float(t_split[1]), _p(scale), _p(bias), _p(count), _p(out),
                                     n_img, rows_per_img, n_cls, dim, _p(range_flag), stream_ptr()), "wd_retrieval_max_split")


def similarity_split(e_split, rows, t_split, unscale, out, n_cls, dim, ldo, seg=None, sigmoid=True, range_flag=None) -> None:
    """Region x text similarity on the fp16x3 256 x 256 kernel (wd_similarity_split): ``e_split`` = embeddings as fp16 hi/lo
    groups (buffer padded to a multiple of 8 rows), ``t_split`` = the text rows from :func:`split_weights`; ``seg`` as in
    :func:`conv_gemm`; ``unscale`` = text unscale / embedding split scale."""
    sr, e0, e1 = (int(seg[0]), int(seg[1]), int(seg[2])) if seg is not None else (0, 0, 0)
    sc = (C.c_float * 3)(*[float(v) for v in (seg[3] if seg is not None else (1, 1, 1))])
    sb = (C.c_float * 3)(*[float(v) for v in (seg[4] if seg is not None else (0, 0, 0))])
    check(LIB.wd_similarity_split(_p(e_split), int(rows), _p(t_split), float(unscale), _p(out), int(n_cls), int(dim), int(ldo),
                                  sr, e0, e1, sc, sb, int(bool(sigmoid)), _p(range_flag), stream_ptr()), "wd_similarity_split")


def text_embed(ids, pos_ids, word, pos, type0, out) -> None:
    check(LIB.wd_text_embed(_p(ids), _p(pos_ids), _p(word), _p(pos), _p(type0), _p(out), ids.numel(), word.shape[1],
                            stream_ptr()), "wd_text_embed")


def attention_small(qkv, mask, out, n_seq, seq_len, heads, head_dim) -> None:
    check(LIB.wd_attention_small(_p(qkv), _p(mask), _p(out), n_seq, seq_len, heads, head_dim, qkv.shape[1], out.shape[1],
                                 stream_ptr()), "wd_attention_small")


def max_sigmoid_attn(embed, guide, head_bias, head_scale, x, n_img, hw, n_guide, heads, head_channels, out_head_channels) -> None:
    """``embed`` / ``x``: 2-d row views [n_img * hw, ld]; ``x`` is scaled in place."""
    check(LIB.wd_max_sigmoid_attn(_p(embed), embed.stride(0), _p(guide), _p(head_bias), _p(head_scale), _p(x), x.stride(0),
                                  n_img, hw, n_guide, heads, head_channels, out_head_channels, stream_ptr()),
          "wd_max_sigmoid_attn")


def adaptive_maxpool_nhwc(x, out, out_img_stride, n_img, h, w, channels, pool) -> None:
    """``x``: rows [n_img * h * w, ld]; ``out``: rows of the patch table starting at this level's first cell."""
    check(LIB.wd_adaptive_maxpool_nhwc(_p(x), x.stride(0), _p(out), out.stride(0), out_img_stride, n_img, h, w, channels, pool,
                                       stream_ptr()), "wd_adaptive_maxpool_nhwc")


def cross_attention_small(q, k, v, out, n_img, n_q, n_k, heads, head_dim) -> None:
    if k.stride(0) != v.stride(0):
        raise WedetectHipError("k and v must share their row pitch")
    check(LIB.wd_cross_attention_small(_p(q), q.stride(0), _p(k), _p(v), k.stride(0), _p(out), out.stride(0), n_img, n_q, n_k,
                                       heads, head_dim, stream_ptr()), "wd_cross_attention_small")


def time_next_gemm(start: "torch.cuda.Event", stop: "torch.cuda.Event") -> None:
    """The next GEMM launch of this thread records its own begin / end into the two (already created, timing-enabled)
    events; ``start.elapsed_time(stop)`` is then the kernel's duration.  See wd_time_next_gemm."""
    check(LIB.wd_time_next_gemm(start.cuda_event, stop.cuda_event), "wd_time_next_gemm")
