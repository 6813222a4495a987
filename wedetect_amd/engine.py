"""The image-tower execution plan: a fixed sequence of C-ABI kernel launches over
pre-allocated NHWC fp32 buffers resident in HBM.

One ``ImageTower`` = (architecture, batch, input H x W) on one GPU.  Nothing is
allocated per step; every ``torch.cat`` of the reference is replaced by producers writing
into channel slices of a wider buffer (row stride ``ld`` > C).  All arithmetic happens in
libwedetect_hip.so — this module only sequences launches on the current HIP stream.

Reference call stack being replaced:
  ConvNeXt.forward                  mm_backbone.py:233-255
  CSPRepBiFPANNeck.forward          yolo_world_pafpn.py:1114-1137
  YOLOWorldHeadModule.forward       yolo_world_head.py:263-294
  head_predict / predict_by_feat    generate_proposal.py:1150-1218 / yolo_world_head.py:578-749
"""
from __future__ import annotations

import contextlib
import copy
import math
import os
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as L
from .arch import ArchSpec, CLS_MID, EMBED_DIM, REG_MID, STRIDES, get_arch, level_sizes
from .pack import Packed


DEFAULT_PRECISION = "fp16x3"


def fold_layernorm_into_linear(w1: torch.Tensor, b1: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """LayerNorm (affine gamma, beta) followed by a linear layer (w1 [n, c], b1 [n]) as ONE contraction of the un-normalised row d:

        W1 LN(d) + b1 = rstd (W' d - mean u) + v,    W' = W1 diag(gamma),  u = W' 1,  v = W1 beta + b1

    (mm_backbone.py:114-118: norm -> pwconv1).  Returns fp32 (W', u, v), computed in float64; u is the row sum of the fp32 W' the
    GEMM actually multiplies, so that a constant row d = c 1 cancels to rounding."""
    w64, g64 = w1.double(), gamma.double()
    w1g = (w64 * g64[None, :]).float().contiguous()
    u = w1g.double().sum(dim=1).float().contiguous()
    v = (w64 @ beta.double() + b1.double()).float().contiguous()
    return w1g, u, v


_DEVICE_STREAMS: Dict[str, Dict[str, list]] = {}


def device_streams(dev: torch.device, role: str, n: int = 1) -> List[torch.cuda.Stream]:
    """The ``n`` streams of a ROLE on a device, shared by every tower of the process.  The HIP runtime multiplexes streams onto four
    hardware queues in the order they are created: a stream of batches wants caller, post, nh and the second backbone stream
    on four DIFFERENT queues (ImageTower.detect), so the roles are created once, in that order, the first time any tower pipelines —
    a tower built later (another architecture or batch size in the same process) then gets the very mapping the first one had
    instead of whatever the creation order of ITS streams would have landed on (a Large tower built after a Base tower in one
    process ran 9 % slower than alone: its nh and second backbone streams shared a queue).  Sharing is only a matter of order:
    towers used one after another, as the detectors use them, never wait for each other on it."""
    pool = _DEVICE_STREAMS.setdefault(str(dev), {})
    if not pool:                                   # fixed creation order of the pipeline roles
        for r in ("post", "nh", "bb2"):
            pool[r] = [torch.cuda.Stream(device=dev)]
    lst = pool.setdefault(role, [])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=dev))
    return lst[:n]


class ImageTower:
    PRECISIONS = ("fp32", "fp16x3")
    SPLIT_K_AUTO_PIXELS = 4 * 640 * 640
    SPLIT_K_MID_PIXELS = 8 * 640 * 640

    def __init__(self, arch, packed: Packed, batch: int, height: int, width: int, device="cuda",
                 max_classes: int = 256, nms_pre: int = 30000, max_out: int = 300,
                 precision: Optional[str] = None, split_k: Optional[bool] = None):
        """``precision``: arithmetic of the dense convs / linears of backbone, neck and head —
        "fp32" = v_mfma_f32_16x16x4_f32 (conv_gemm.hip); "fp16x3" = three fp16 MFMA passes on
        operands split into (hi, lo) halves, fp32-equivalent accuracy at 2-2.5x the speed
        (split_gemm.hip).  The region x text similarity GEMM, depthwise convs, LayerNorm and
        the post-process are fp32 in both modes.  None = $WEDETECT_PRECISION or DEFAULT_PRECISION.
        ``split_k`` (fp16x3 only; None = $WEDETECT_SPLIT_K == "1", default off): let under-filled launches —
        batch-1 inference, the coarsest maps — split their K loop over several workgroups and add the
        partial sums in a second pass (wd_conv_gemm_split_ws): Base batch 1 10.6 -> 6.3 ms.  Off by default
        because the split count depends on the batch size, so results are no longer bit-identical ACROSS
        batch sizes (they stay deterministic, and within 1e-6 of the unsplit ones)."""
        self.a: ArchSpec = get_arch(arch) if isinstance(arch, str) else arch
        self.P = packed
        if precision is None:
            precision = os.environ.get("WEDETECT_PRECISION", DEFAULT_PRECISION)
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {self.PRECISIONS}")
        self.precision = precision
        self.Ws: Dict[str, tuple] = {}          # weight name -> (split buffer, unscale), fp16x3 mode only
        # split-K workspace for under-filled fp16x3 launches (small batches, the coarsest maps): partial sums [S][m][n]
        auto_class = split_k is None and os.environ.get("WEDETECT_SPLIT_K", "auto") == "auto"
        if split_k is None:
            # round 6: "auto" (default) = on for SMALL towers — at most SPLIT_K_AUTO_PIXELS input pixels per batch (four 640 x 640
            # images) — where every long-K layer is one under-filled, K-sequential launch (Base batch 1: pwconv2 of a stage-3 block =
            # 52 workgroups walking 128 K stages, 75 us for 3 GFLOP): 7.92 -> 5.08 ms per image, profiles/r06_small_batch.txt.  The
            # reference runs batch 1 everywhere (infer_wedetect.py:117, generate_proposal.py:1261).  Large towers keep the unsplit
            # launches, so an image's low-order bits differ between a small and a large tower (1e-6; deterministic within each
            # class; the detectors' towers of one checkpoint agree bit for bit only inside a class).  "1" / "0" force.
            env = os.environ.get("WEDETECT_SPLIT_K", "auto")
            split_k = env == "1" or (env == "auto" and batch * height * width <= self.SPLIT_K_MID_PIXELS)
        # The split count of a layer comes from its per-image geometry times the REFERENCE batch of the tower's class, never from the
        # batch itself: 1 for the small class (<= SPLIT_K_AUTO_PIXELS per batch, and for split_k=True towers of any size), and — round
        # 6, later — the number of images of this size that fill the MID class (<= SPLIT_K_MID_PIXELS, eight 640 x 640 images: the
        # 3 x 3 convs of the 40 x 40 / 20 x 20 maps are 50 / 26 tiles on 256 CUs there).  Inside a class an image gets the same bits
        # in any batch; between classes they differ in the low-order bits (1e-6).
        self._split_ref = 1
        if split_k and auto_class and batch * height * width > self.SPLIT_K_AUTO_PIXELS:
            self._split_ref = max(1, self.SPLIT_K_MID_PIXELS // (height * width))
        self.kws = (torch.empty(16 << 20, dtype=torch.float32, device=torch.device(device))
                    if precision == "fp16x3" and split_k else None)
        self._kws_lane: Dict[int, torch.Tensor] = {}          # the side lanes' own split-K workspaces (DAG mode)
        # Batch-INVARIANT split-K (on by default, $WEDETECT_FIXED_SPLITK=0 turns it off): the 3x3 convs on maps of at most
        # 20 x 20 pixels with K >= 2304 (BepC3 of the coarsest level, the level-2 head convs) always split K in two —
        # decided by the layer alone, never by the batch, so results stay bit-identical across batch sizes.  Base B = 32:
        # 96.6 -> 71.8 us (256 -> 256) and 175.6 -> 125.1 us (512 -> 256) per launch (profiles/r02_smallmap_ab.txt).
        self.fixed_splitk = precision == "fp16x3" and os.environ.get("WEDETECT_FIXED_SPLITK", "1") == "1"
        self.fws = (torch.empty(2 * batch * 400 * 256 + 64, dtype=torch.float32, device=torch.device(device))
                    if self.fixed_splitk else None)
        # 256 x 256 fp16x3 kernels for the big pre-split layers (split_gemm_p8.hip).  $WEDETECT_P8: "tile" = one workgroup per
        # output tile, picked by the library; "persist" (default since round 4) = additionally offer the park workspace, which
        # selects the persistent work-unit form where it applies (K >= 1024, at least one gang tile per gang).  Rounds 2-3 dealt the
        # units to single workgroups: 12 % slower inside the step (profiles/r03_p8_tile_vs_persist.txt) because every workgroup
        # streamed its own row panel through the XCD's L2 (FETCH_SIZE 1.35 GB per launch against 0.33 - 0.58 GB); dealt to gangs
        # (round 4, profiles/r04_persist_pmc.txt) it is 0.4 - 0.8 % faster than the tile form in the step.  Its flag words must
        # start zero and it is never lent to split-K launches; "0" = neither
        # (the round-1 128 x 128 / ping-pong kernels, for A/B runs).
        self.p8_mode = os.environ.get("WEDETECT_P8", "persist")
        self.post_stream, self._post_ready, self._post_done = None, None, None      # detect(overlap_post=True)
        # Neck / head as a DAG on side streams (round 5; $WEDETECT_DAG = "auto" (default, see below), "1": always, "0": never = the
        # serial chain of rounds 1-4).  The neck + head are ~110 launches of which the 20 x 20 /
        # 40 x 40 ones fill 50 - 200 of 256 CUs, issued as ONE dependent chain although the graph is not one: the BiFusion
        # input branches only read backbone outputs, a BepC3's cv2 is independent of its 3 x 3 chain, head level 0 (the largest
        # head convs) needs only P3, which exists before downsample2 -> Rep_n3 -> downsample1 -> Rep_n4 run, and the cls / reg
        # branches of a level are independent (yolo_world_pafpn.py:1114-1137, yolo_world_head.py:271-294).  Same kernels, same
        # arguments, same buffers (disjoint channel slices where two lanes write one buffer): bit-identical results.
        # Measured at Base / Tiny, batches 1 ... 32 (profiles/r05_small_batch.txt): eager launches gain at EVERY batch size (Tiny batch
        # 1: 4.34 -> 3.93 ms, Base batch 32: +1.3 %), a captured hipGraph gains from 8 x 640 x 640 pixels and is neutral below — so
        # "auto" = always when launching eagerly, from that size under capture.
        dag = os.environ.get("WEDETECT_DAG", "auto")
        self.dag = dag != "0"
        self._dag_forced = dag == "1"
        self._dag_in_capture = dag == "1" or batch * height * width >= 8 * 640 * 640
        self._side: List[torch.cuda.Stream] = []
        self._events: List[torch.cuda.Event] = []
        self._ev_i = 0
        self._ev_after_neck = 0
        self._lane_i = 0
        self._fws_lane: Dict[int, torch.Tensor] = {}
        self._head_evs: List[torch.cuda.Event] = []
        # Backbone as INDEPENDENT IMAGE CHAINS on side streams (round 6, late; $WEDETECT_BB_CHAINS = "auto" (default), "1" = one
        # chain = rounds 1-5, "2" / "4": that many).  A ConvNeXt stage is one dependent chain of launches — depthwise 7 x 7 (HBM-bound,
        # no MFMA), statistics finalize, pwconv1, pwconv2 (MFMA-bound, the HBM idle) — but only ALONG an image: images never meet
        # before the result gather (mm_backbone.py:233-255 is batch-parallel throughout).  The batch is cut into contiguous image
        # groups, each group runs the whole backbone on its own stream over row slices of the SAME buffers, the neck starts when
        # all have arrived: one group's depthwise / LayerNorm launches then run beside another's GEMMs and the tail of one
        # launch is filled by the next group's head.  Same kernels on the same rows (every kernel form gives a row the same
        # bits whatever batch it arrives in): bit-identical to the one-chain step
        # (tests/test_gpu_network.py::test_backbone_image_chains_equal_the_single_chain).  "auto": two chains for the IN-LINE step of
        # towers without latency split-K (that class splits K by the batch's reference geometry) in the window in which they were
        # measured to pay (_n_chains: 32 to 63 x 640 x 640 pixels per batch, + 2.9 % at Base B = 32); a stream of batches keeps two
        # whole backbones in flight instead (bb_depth below).  $WEDETECT_BB_CHAIN_ORDER / _STAGES: the cross-chain phase orders and
        # stage ranges that were measured and lost (profiles/r06_pipeline.txt), kept for A/B runs.
        self.bb_chains = os.environ.get("WEDETECT_BB_CHAINS", "auto")
        self.bb_chain_order = os.environ.get("WEDETECT_BB_CHAIN_ORDER", "free")
        st = os.environ.get("WEDETECT_BB_CHAIN_STAGES", "0-3").split("-")       # the stages (0-based, inclusive) that run as chains
        self.bb_chain_stages = (max(0, int(st[0])), min(3, int(st[-1])))
        self._chain_streams: List[torch.cuda.Stream] = []
        # A stream of batches (detect(overlap_post=True)), round 6, late: neck + head + similarity of step i on the tower's own
        # "nh" stream beside the BACKBONE of step i + 1 ($WEDETECT_PIPE_NECK = "auto" (default: on, every size gains), "1", "0").
        # The neck / head launches fill 50 - 200 of 256 CUs at 0.15 - 0.45 of the MFMA roof; the next batch's backbone — which
        # touches none of their buffers once c1..c4 exist in bb_depth + 1 sets (0.8 GB each for Base, B = 32) — runs in the
        # gaps.  Same kernels, same arguments: bit-identical results
        # (tests/test_gpu_network.py::test_neck_head_pipelined_behind_the_next_backbone_equals_the_in_line_step).
        # $WEDETECT_BB_DEPTH ("auto" = 2 from two 640 x 640 images per batch): TWO backbones in flight — the steps of the stream
        # alternate between the caller's stream and a second backbone stream (_slot1_backbone).
        self.pipe_neck = os.environ.get("WEDETECT_PIPE_NECK", "auto")
        self._nh_stream: Optional[torch.cuda.Stream] = None
        self._nh_issue = False                    # neck / head launches are being issued on the nh stream (never lend them self.park)
        self._x_sets: List[List[torch.Tensor]] = []
        self._x_par = 0
        self._x_free: List[Optional[torch.cuda.Event]] = [None]
        self._bb_done: List[torch.cuda.Event] = []
        self.bb_depth = os.environ.get("WEDETECT_BB_DEPTH", "auto")
        self._depth2_issue = False                # a backbone of the two-in-flight schedule is being issued: no image chains by default
        # The persistent (stream-K) forms of the 256 x 256 kernel and of the wide block MLP hold every CU for the whole launch: the
        # fastest form of a launch that has the chip to itself (+ 0.4 - 0.8 % on the one-stream step, round 4), the wrong one where
        # launches of other streams are meant to run in its gaps — under the stream pipeline the TILE form wins at every size
        # (Base B = 16 / 32 / 64: + 0.5 / 1.0 / 1.4 %, profiles/r06_pipeline.txt) although the launch itself is slower alone.  Same
        # MFMA chain per output, bit-identical.  $WEDETECT_PIPE_PERSIST: "0" (default) = tile forms while a pipelined step's
        # backbone is issued, "p8" / "mlp" / "1" = keep the persistent form of that family / of both (A/B).
        self.pipe_persist = os.environ.get("WEDETECT_PIPE_PERSIST", "0")
        self._pipe_issue = False                  # the backbone of a pipelined step (detect(overlap_post=True)) is being issued
        # $WEDETECT_FORCE_TILE=1: the tile forms in EVERY step — what a pipelined step's backbone runs, for measuring those
        # kernels one at a time (bench.py's serial leg, scripts/profile_final.sh)
        self.force_tile_forms = os.environ.get("WEDETECT_FORCE_TILE", "0") == "1"
        self._bb_slot = 0
        self._slot1: Optional[dict] = None
        self._chain_parks: Dict[int, torch.Tensor] = {}
        self._chain_evs: List[torch.cuda.Event] = []
        self.s2d_down = os.environ.get("WEDETECT_S2D_DOWN", "1") != "0"     # downsample convs as plain GEMMs on space-to-depth LayerNorm rows
        self.park = None
        if precision == "fp16x3" and self.p8_mode == "persist" and L.p8_workspace_bytes() > 0:
            self.park = torch.zeros(L.p8_workspace_bytes() // 4, dtype=torch.float32, device=torch.device(device))
        # 3 x 3 / stride 1 convs of the pre-split neck / head on the row-sharing kernel (split_gemm_conv3.hip, round 4): one LDS
        # stage per (filter row, channel chunk) serves the row's three taps.  Another K order than the tap-per-stage kernel
        # (last-bit differences); "0" pins that one (cfg 70) for A/B runs and for the bit-identity tests against rounds 2-3
        self.conv3 = os.environ.get("WEDETECT_CONV3", "1") != "0"
        self._park_mlp = None                    # park workspace of the persistent wide fused MLP (the same layout; shared with self.park)
        self.B, self.H, self.W = batch, height, width
        self.fuse_stem = os.environ.get("WEDETECT_FUSE_STEM", "1") == "1"      # stem as one fp32 kernel (bit-identical to the fp32 three-launch form)
        self.fuse_mlp = os.environ.get("WEDETECT_FUSE_MLP", "1") == "1"        # stage-1 block MLP as one kernel (bit-identical; profiles/r03_mlp_fused.txt)
        # round 4: the block MLP of the 256 / 512-channel stages as one kernel too (wd_mlp_fused_wide, bit-identical):
        # comma-separated widths, "" = none
        # comma-separated widths, "" = none.  Default "256": on the stage-2 shape the fused persistent kernel is 7 % faster than
        # the two launches (750 vs 804 us per block), on the stage-3 shape (512 channels) still 2-3 % slower (670 vs 654 us) —
        # profiles/r04_mlp_wide.txt
        self.fuse_mlp_wide = tuple(int(v) for v in os.environ.get("WEDETECT_FUSE_MLP_WIDE", "256").split(",") if v.strip())
        self.Wf: Dict[str, tuple] = {}           # fragment-major weight copies of those layers (lib.mlp_wide_pack)
        # dwconv -> LayerNorm in one kernel: "auto" = the stages of <= 128 channels, where the pre-norm values stay in registers
        # (profiles/r03_dwln_reg.txt); "1" = every stage (the wide ones through L2: slower, profiles/r02_dwln_ab.txt); "0" = never
        self.fuse_dwln = os.environ.get("WEDETECT_FUSE_DWLN", "auto")
        # round 5: the same for 256 / 384 / 512 channels (dwconv7_ln_wide_kernel: the channel blocks dealt to one or two thread
        # groups of a workgroup, pre-norm values in registers, bit-identical to the pair); comma-separated widths, "" = none
        self.fuse_dwln_wide = tuple(int(v) for v in os.environ.get("WEDETECT_FUSE_DWLN_WIDE", "").split(",") if v.strip())
        # round 5: the block LayerNorm of the stages that run pwconv1 / pwconv2 as two launches FOLDED into pwconv1 (no LayerNorm
        # kernel, the normalised tensor is never written): wd_dwconv7_stats writes the raw depthwise output d as fp16 hi/lo groups
        # plus per-(pixel, 32-channel block) statistics, wd_ln_stats_finalize turns them into (mean, rstd) per row, and the GEMM's
        # epilogue computes rstd (W' d - mean u) + v with W' = W gamma, u = W' 1, v = W beta + b  ==  W LN(d) + b.  Not
        # bit-identical to the LayerNorm kernel (the centring happens after the contraction, in fp32; within 5e-5 / 1e-5 of it on
        # embeddings / scores, every golden and index-parity test green with it): $WEDETECT_LN_FOLD, default on — 30 LayerNorm
        # launches and 6 GB of HBM traffic per Base step less, +1.2 % (profiles/r05_ln_fold.txt).
        self.ln_fold = os.environ.get("WEDETECT_LN_FOLD", "1") == "1"
        # round 6: the fold inside the wide one-kernel block MLP too (stage 2 of Base: dwconv -> LayerNorm kernel -> fused MLP becomes
        # dwconv with statistics -> finalize -> fused MLP with the (mean, rstd, u, v) hidden epilogue); $WEDETECT_LN_FOLD_FUSED=0: A/B
        self.ln_fold_fused = os.environ.get("WEDETECT_LN_FOLD_FUSED", "1") == "1"
        self.ln_part = self.ln_stats = None
        # round 6: the region x text similarity GEMM on the fp16x3 256 x 256 kernel for LARGE text banks (wd_similarity_split;
        # yolo_world_head.py:90-108).  The 80-class launch is bound by its 86 MB of scores and 826 MB of embeddings, not by the fp32
        # MFMA rate, and keeps the fp32 kernel; from 256 classes the contraction dominates (1203 classes: 2.3 of a 41 ms step at
        # 0.67 of the fp32 MFMA peak = 106 TFLOP/s, against ~380 on the fp16x3 kernel).  The embedding conv then writes its output
        # twice — fp16 hi/lo groups for this GEMM, fp32 for the gather and the callers (WdConvGemm.c2 with the batch-stride row map).
        # $WEDETECT_SIM_SPLIT: "auto" (default) = banks of at least SIM_SPLIT_MIN rows, "1" = always, "0" = never.
        self.sim_split = os.environ.get("WEDETECT_SIM_SPLIT", "auto")
        self.embed_s: Optional[torch.Tensor] = None          # [round_up(B * anchors, 8), 768] hi/lo groups, allocated on first use
        self._embed_split_on = False                          # the head of the step being issued writes embed_s
        self._embed_split_valid = False                       # embed_s holds the split of self.embed
        self._text_split: list = []                           # (weakref to the bank tensor, version, normalize, (split buffer, unscale))
        self.overflowed = False
        self.fp16x3_trips = 0            # range-guard trips of this tower (each: one re-calibration attempt, then the fp32 fallback)
        self.fp16x3_retries = 0          # returns from the fp32 fallback to the fp16x3 kernels
        self._clean_fp32_steps, self._retry_after = 0, self.FALLBACK_RETRY
        # sticky range flag of the fp16x3 GEMMs (WdConvGemm.range_flag): set by a launch whose accumulators are inf / NaN
        self.range_flags = torch.zeros(2, dtype=torch.int32, device=torch.device(device))
        self.range_flag = self.range_flags[0:1]
        # The five neck layers that read the ConvNeXt residual streams c1..c4 directly are the only GEMMs whose inputs are
        # not bounded by construction (neither LayerNorm outputs nor activations of a BN-folded conv).  They run fp16x3
        # under their OWN range flag; if it ever trips, only they are pinned to the fp32 MFMA kernel (neck_pin) and the
        # step is repeated — the rest of the tower keeps fp16x3.  $WEDETECT_NECK_GUARD=0 pins them from the start.
        self.range_flag2 = self.range_flags[1:2]
        self.neck_pin = precision != "fp16x3" or os.environ.get("WEDETECT_NECK_GUARD", "1") == "0"
        self.dev = torch.device(device)
        if height % 32 or width % 32:
            raise ValueError("input size must be a multiple of 32")
        a, B = self.a, batch
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=self.dev)
        # ---- backbone buffers
        self.hw = [(height // (4 << i), width // (4 << i)) for i in range(4)]
        self.M = [B * h * w for h, w in self.hw]
        self.patches = f(self.M[0], 48)
        self.x = [f(self.M[i], a.dims[i]) for i in range(4)]            # c1..c4 (residual streams)
        self.tmp = f(max(self.M[i] * a.dims[i] for i in range(4)))
        self.hid = f(max(self.M[i] * 4 * a.dims[i] for i in range(4)))
        # ---- neck buffers
        nc = a.neck_channels
        (h3, w3), (h4, w4), (h5, w5) = self.hw[1], self.hw[2], self.hw[3]
        M2, M3, M4, M5 = self.M
        self.cat_n4 = f(M5, nc["d1"] + nc["p5r"])
        self.cat_b0 = f(M4, 3 * nc["p5r"])
        self.b0_t = f(M3, nc["p5r"])
        self.f0 = f(M4, nc["p5r"])
        self.f_out0 = f(M4, nc["p5r"])
        self.cat_n3 = f(M4, nc["d2"] + nc["p4r"])
        self.cat_b1 = f(M3, 3 * nc["p4r"])
        self.b1_t = f(M2, nc["p4r"])
        self.f1 = f(M3, nc["p4r"])
        self.p3 = f(M3, nc["p4r"])
        self.p4 = f(M4, nc["n3"])
        self.p5 = f(M5, nc["n4"])
        self._bep = {}
        for name, m, cout in (("Rep_p4", M4, nc["p5r"]), ("Rep_p3", M3, nc["p4r"]),
                              ("Rep_n3", M4, nc["n3"]), ("Rep_n4", M5, nc["n4"])):
            c_ = cout // 2
            # u0s / u1s: the fp16 hi/lo twins of u0 / u1 (pre-split neck: the 3x3 convs read the split form, the BottleRep
            # residual "+ alpha x" the fp32 form — the producing layer writes both)
            self._bep[name] = dict(cat=f(m, 2 * c_), u0=f(m, c_), u1=f(m, c_), t=f(m, c_), u0s=f(m, c_), u1s=f(m, c_), c_=c_)
        # Pre-split neck / head (round 3, $WEDETECT_NECK_PRESPLIT=0 restores the round-2 path for A/B runs): every neck / head
        # activation is written by its producer as fp16 hi/lo groups (WD_SPLIT_C epilogues) and every conv reads it through
        # the LDS-DMA implicit-GEMM kernel (split_gemm_conv.hip) — no per-tap re-splitting, no register staging, no LDS bank
        # conflicts.  Same halves, same K order: embeddings / boxes are bit-identical to the loader-split path.
        self.presplit_neck = (precision == "fp16x3" and os.environ.get("WEDETECT_NECK_PRESPLIT", "1") != "0"
                              and all(v % 32 == 0 for v in nc.values()) and CLS_MID % 16 == 0 and REG_MID % 16 == 0)
        # ---- head buffers
        self.lv = level_sizes(height, width)
        self.nl = [h * w for h, w in self.lv]
        self.ntot = sum(self.nl)
        self.off = [0, self.nl[0], self.nl[0] + self.nl[1]]
        self.embed = f(B, self.ntot, EMBED_DIM)
        self.boxes = f(B, self.ntot, 4)
        self.hc = [(f(B * n, CLS_MID), f(B * n, CLS_MID)) for n in self.nl]
        self.hr = [(f(B * n, REG_MID), f(B * n, REG_MID), f(B * n, 4 * 16)) for n in self.nl]
        # ---- similarity / post-process buffers (sized for max_classes; grown on demand)
        self.nms_pre, self.max_out = nms_pre, max_out
        self.cap = L.topk_capacity(nms_pre)
        # bumped whenever a buffer that captured hipGraphs hold raw pointers to is re-allocated (scores, top-k workspace,
        # normalised text bank): graphs captured under an older generation must be dropped, not replayed
        self.generation = 0
        # fp16x3 range management (calibrate()): power-of-two scale per split buffer / loader-split input, 1.0 unless a
        # calibration pass found the tensor outside the window in which fp16 (hi, lo) pairs carry fp32 accuracy
        self.sscale: Dict[str, float] = {}
        self._calib: Optional[Dict[str, torch.Tensor]] = None
        self._ln_scaled: Dict[tuple, tuple] = {}
        self.calibrated = False
        self.calibration_runs = 0
        self._alloc_post(max_classes)
        self.nms_ws = torch.zeros(max(1, L.nms_workspace_bytes(B) // 4), dtype=torch.int32, device=self.dev)
        self.cand_idx = torch.empty(B, self.cap, dtype=torch.int32, device=self.dev)
        self.cand_score = f(B, self.cap)
        self.cand_count = torch.empty(B, dtype=torch.int32, device=self.dev)
        self.out_boxes = f(B, max_out, 4)
        self.out_scores = f(B, max_out)
        self.out_labels = torch.empty(B, max_out, dtype=torch.int32, device=self.dev)
        self.out_anchors = torch.empty(B, max_out, dtype=torch.int32, device=self.dev)
        self.out_count = torch.empty(B, dtype=torch.int32, device=self.dev)
        self.out_embed = f(B, max_out, EMBED_DIM)
        self.text_norm = f(max_classes, EMBED_DIM)
        # per-level exp(logit_scale) in fp32 like ``logit_scale.exp()`` on a float32 parameter
        self.lvl_scale = [float(np.exp(np.float32(self.P.s[f"head{l}.logit_scale"]))) for l in range(3)]
        self.lvl_bias = [float(np.float32(self.P.s[f"head{l}.bias"])) for l in range(3)]
        self.lvl_logit_scale = [float(np.float32(self.P.s[f"head{l}.logit_scale"])) for l in range(3)]
        self._prepare_fold()

    def _alloc_post(self, k: int) -> None:
        self.max_classes = k
        self.generation += 1
        self.scores = torch.empty(self.B, self.ntot, k, dtype=torch.float32, device=self.dev)
        nbytes = L.topk_workspace_bytes(self.B, self.ntot * k, self.nms_pre)
        self.topk_ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.dev)
        off = (-self.topk_ws.data_ptr()) % 256
        self.topk_ws = self.topk_ws[off:off + nbytes]

    # ------------------------------------------------------------------ helpers
    def _record(self, key: Optional[str], t: torch.Tensor) -> None:
        """calibrate(): running max |x| of the tensor behind a scale key (fp32 pass: ``t`` holds fp32 values)."""
        if self._calib is None or key is None or t.numel() == 0:
            return
        m = t.abs().max()
        self._calib[key] = torch.maximum(self._calib[key], m) if key in self._calib else m

    def _ln_params(self, wname: str, bname: str, key: Optional[str]):
        """LayerNorm affine parameters with the output's split scale folded in (a power of two: y * s == LN with
        gamma * s, beta * s, exactly)."""
        sc = self.sscale.get(key, 1.0) if key else 1.0
        if sc == 1.0 or self.precision != "fp16x3":
            return self.P[wname], self.P[bname]
        ck = (wname, sc)
        if ck not in self._ln_scaled:
            self._ln_scaled[ck] = (self.P[wname] * sc, self.P[bname] * sc)
        return self._ln_scaled[ck]

    def _gemm(self, a, w: str, b: Optional[str], c, *, fp32: bool = False, a_key: Optional[str] = None,
              c_key: Optional[str] = None, **kw):
        """One dense layer with packed weight ``w`` / bias ``b`` in the tower's precision.
        ``fp32=True`` pins the layer to the fp32 MFMA kernel: the neck layers that read the ConvNeXt
        residual streams c1..c4 directly — the only GEMM inputs that are neither LayerNorm outputs nor
        activations of a BN-folded conv, hence not bounded by construction (fp16 halves overflow at 65504)."""
        ws = None
        guarded = fp32 and self.precision == "fp16x3" and not self.neck_pin
        if self.precision == "fp16x3" and (not fp32 or guarded):
            ws = self.Ws.get(w)
            if ws is None:                      # first use: split once, keep resident
                wt = self.P[w]
                ws = self.Ws[w] = L.split_weights(wt.view(wt.shape[0], -1))
        work = None
        if ws is not None and self.kws is not None:
            # latency mode: the split count comes from the layer's PER-IMAGE geometry (never from the batch), so that every tower
            # of the small class gives an image the same bits whatever batch it arrives in
            ks = self._latency_splits(kw)
            if ks > 1:
                kh_, st_, pd_ = kw.get("kh", 1), kw.get("stride", 1), kw.get("pad", 0)
                m_ = self.B * ((kw["hin"] + 2 * pd_ - kh_) // st_ + 1) * ((kw["win"] + 2 * pd_ - kh_) // st_ + 1)
                work, kw = self._lane_kws(ks * m_ * kw["n"] + 64), dict(kw, k_splits=ks)
        if (ws is not None and self.fixed_splitk and self.kws is None and kw.get("kh", 1) == 3 and kw.get("stride", 1) == 1
                and kw["hin"] * kw["win"] <= 400 and 9 * kw["cin"] >= 2304 and kw["n"] % 4 == 0
                and 2 * self.B * kw["hin"] * kw["win"] * kw["n"] <= self.fws.numel()):
            work, kw = self._lane_fws(), dict(kw, k_splits=2)
        if (ws is not None and (kw.get("split_flags", 0) & L.SPLIT_A) and not self.conv3 and kw.get("kh", 1) == 3
                and kw.get("stride", 1) == 1 and "split_cfg" not in kw):
            kw = dict(kw, split_cfg=70)
        if ws is not None and (kw.get("split_flags", 0) & L.SPLIT_A):
            plain = kw.get("kh", 1) == 1 and kw.get("kw", 1) == 1 and kw.get("stride", 1) == 1 and kw.get("pad", 0) == 0
            m = self.B * kw["hin"] * kw["win"]
            if plain and self.park is not None and self._lane_i == 0 and not self._nh_issue and self._persist_ok("p8") and L.gemm_config(m, kw["n"], kw["cin"], split=True, presplit=True,
                                                                                       park=True).endswith("/p8s"):
                work = self.park
            elif plain and self.p8_mode == "0" and kw["cin"] % 16 == 0:
                # A/B switch: the round-1 kernels (another K-tile order inside the launch, the same MFMA chain per output) for
                # the layers those kernels cover — plain rows, no scatter / batch-stride / dual output, no residual on a
                # split output; the rest stays on the implicit-GEMM kernel
                flags_ = kw.get("split_flags", 0)
                special = (kw.get("out_mode", 0) != 0 or kw.get("c_batch_stride", 0) > 0 or kw.get("seg") is not None
                           or kw.get("sigmoid") or kw.get("c2") is not None or ((flags_ & L.SPLIT_C) and kw.get("res") is not None))
                if not special:
                    kw = dict(kw, split_cfg=63 if m >= 131072 else 60)
        if ws is not None:
            flags = kw.get("split_flags", 0)
            sa = self.sscale.get(a_key, 1.0) if a_key else 1.0
            if sa != 1.0:                       # the operand was (or will be, by the loader) multiplied by sa: divide it out
                ws = (ws[0], ws[1] / sa)
                if not (flags & L.SPLIT_A):
                    kw = dict(kw, a_scale=sa)
            sc = self.sscale.get(c_key, 1.0) if (c_key and (flags & L.SPLIT_C)) else 1.0
            if sc != 1.0:
                kw = dict(kw, c_split_scale=sc)
        L.conv_gemm(a, None if ws is not None else self.P[w], self.P[b] if b else None, c, batch=self.B,
                    w_split=ws, workspace=work,
                    range_flag=(self.range_flag2 if guarded else self.range_flag) if ws is not None else None, **kw)
        if self._calib is not None:
            if fp32 and a_key:                  # a layer that reads an fp32 residual stream through the loader
                self._record(a_key, a[..., : kw["cin"]])
            if c_key:
                n, kh_, st_, pd_ = kw["n"], kw.get("kh", 1), kw.get("stride", 1), kw.get("pad", 0)
                m = self.B * ((kw["hin"] + 2 * pd_ - kh_) // st_ + 1) * ((kw["win"] + 2 * pd_ - kh_) // st_ + 1)
                if c.dim() == 1:                # flat scratch buffer (the MLP hidden tensor): rows are dense
                    self._record(c_key, c[: m * kw["ldc"]])
                else:
                    self._record(c_key, c[..., : n // 4] if kw.get("out_mode", 0) == L.OUT_DECONV2X2 else c[..., :n])

    def _latency_splits(self, kw) -> int:
        """K splits of one layer in latency mode, from its geometry at the class's reference batch (ONE image in the small class;
        the library's own rules — split_gemm.hip: pick_ksplits and the implicit-GEMM kernel's): 1 = launch unsplit."""
        flags = kw.get("split_flags", 0)
        kh, st, pd = kw.get("kh", 1), kw.get("stride", 1), kw.get("pad", 0)
        n, k = kw["n"], kh * kh * kw["cin"]
        if n % 4 or kw.get("ln_stats") is not None or "k_splits" in kw:
            return 1
        m1 = ((kw["hin"] + 2 * pd - kh) // st + 1) * ((kw["win"] + 2 * pd - kh) // st + 1) * self._split_ref   # rows of the class's reference batch
        plain = kh == 1 and st == 1 and pd == 0
        special = (kw.get("out_mode", 0) != 0 or kw.get("c_batch_stride", 0) > 0 or kw.get("seg") is not None or kw.get("sigmoid")
                   or kw.get("out_scale", 1.0) != 1.0 or kw.get("out_bias", 0.0) != 0.0)
        covered = plain and not special and kw.get("c2") is None and not ((flags & L.SPLIT_C) and kw.get("res") is not None)
        if (flags & L.SPLIT_A) and not covered:                    # implicit-GEMM LDS-DMA kernels (any epilogue): 256 x 128 tiles
            if kw["cin"] % 16 or k % 16 or n % 8:
                return 1
            tiles, nk = -(-m1 // 256) * -(-n // 128), k // 16
            if tiles >= 128 or nk < 32:
                return 1
            return max(1, min(256 // tiles, nk // 16, 8))
        if flags & L.SPLIT_C:                                      # GEMM kernels that write hi/lo groups do not split K
            return 1
        if (flags & L.SPLIT_A) and not L.gemm_config((self.B if self._split_ref == 1 else 1) * m1, n, k, split=True, presplit=True).endswith("/glds"):
            return 1                                               # a 256-tile / ping-pong launch (never in the small class's shapes)
        tiles, nk = -(-m1 // 128) * -(-n // 128), -(-k // 16)
        if tiles >= 128 or nk < 16:
            return 1
        return max(1, min(512 // tiles, nk // 8, 16))

    def _persist_ok(self, family: str) -> bool:
        """May a launch of ``family`` ("p8" / "mlp") take its persistent form now?  Not while a pipelined step's backbone is issued
        (see __init__: pipe_persist)."""
        return not (self._pipe_issue or self.force_tile_forms) or self.pipe_persist in ("1", family)

    def _fold_weights(self, q: str) -> None:
        """W', u, v of one block (fold_layernorm_into_linear), on the device, once (shared through the packed set)."""
        if q + "w1g" in self.P.t:
            return
        self.P.t[q + "w1g"], self.P.t[q + "u"], self.P.t[q + "v"] = fold_layernorm_into_linear(
            self.P[q + "w1"], self.P[q + "b1"], self.P[q + "ln_w"], self.P[q + "ln_b"])

    def _prepare_fold(self) -> None:
        """Statistics buffers and the folded weights (W', u, v) of every block whose LayerNorm is folded — in __init__, not lazily
        inside backbone(): an allocation or the torch math of the fold must never land in a captured hipGraph."""
        a = self.a
        stages = [i for i in range(4)
                  if self._fold_ok(i, self.precision == "fp16x3" and a.dims[i] % 8 == 0 and (i == 0 or a.dims[i - 1] % 8 == 0))]
        if not stages:
            return
        if self.ln_stats is None:
            rows = max(self.M[k] for k in range(4))
            blk = max(self.M[k] * (a.dims[k] // 32) for k in range(4) if a.dims[k] % 32 == 0)
            self.ln_part = torch.empty(2 * blk, dtype=torch.float32, device=self.dev)
            self.ln_stats = torch.empty(2 * rows, dtype=torch.float32, device=self.dev)
        if not torch.cuda.is_current_stream_capturing():
            for i in stages:
                for j in range(a.depths[i]):
                    self._fold_weights(f"s{i}.{j}.")

    def _fold_ok(self, i: int, pre: bool) -> bool:
        c = self.a.dims[i]
        if not (self.ln_fold and pre and self.precision == "fp16x3" and c % 32 == 0):
            return False
        if self.fuse_dwln == "1" or (self.fuse_dwln == "auto" and (c <= 128 or c in self.fuse_dwln_wide)):
            return False                        # dwconv + LayerNorm run as one kernel there
        if self.fuse_mlp and L.mlp_fused_supported(self.M[i], c, 4 * c):
            return False                        # the 128-channel one-kernel block MLP takes LayerNorm rows (its dwconv + LayerNorm is one kernel too)
        if c in self.fuse_mlp_wide and L.mlp_wide_supported(self.M[i], c, 4 * c):
            return self.ln_fold_fused and c in L.MLP_WIDE_FOLD_WIDTHS      # round 6: the wide one folds (256 channels)
        return True

    def _mlp_fused_fold(self, q: str, i: int) -> None:
        """Round 6: the wide one-kernel block MLP WITH the block's LayerNorm folded into its pwconv1 (wd_mlp_fused_wide_ln):
        self.tmp holds the raw depthwise output as hi/lo groups (wd_dwconv7_stats), self.ln_stats the rows' (mean, rstd)."""
        c = self.a.dims[i]
        self._fold_weights(q)
        ws = []
        for name, (n_, k_) in ((q + "w1g", (4 * c, c)), (q + "w2", (c, 4 * c))):
            f_ = self.Wf.get(name)
            if f_ is None:
                s_ = self.Ws.get(name)
                if s_ is None:
                    wt = self.P[name]
                    s_ = self.Ws[name] = L.split_weights(wt.view(wt.shape[0], -1))
                f_ = self.Wf[name] = (L.mlp_wide_pack(s_[0], n_, k_), s_[1])
            ws.append(f_)
        sa, sh = self.sscale.get(q + "dw", 1.0), self.sscale.get(q + "hid", 1.0)
        if self._park_mlp is None:
            self._park_mlp = self.park if self.park is not None else torch.zeros(
                max(1, L.p8_workspace_bytes() // 4), dtype=torch.float32, device=self.dev)
        L.mlp_fused_wide_ln(self.tmp, self.M[i], c, 4 * c, (ws[0][0], ws[0][1] / sa), self.P[q + "v"], self.P[q + "u"], self.ln_stats,
                            (ws[1][0], ws[1][1] / sh), self.P[q + "b2"], self.x[i], hid_scale=sh, range_flag=self.range_flag,
                            workspace=self._park_mlp if self._persist_ok("mlp") else None)

    def _mlp_fused(self, q: str, i: int, wide: bool = False) -> None:
        """One ConvNeXt block MLP as a single launch: the 4c hidden activation never leaves the CU (wd_mlp_fused_split for the
        128-channel stage, hidden chunk in registers; ``wide``: wd_mlp_fused_wide for 256 / 512 channels, hidden chunks through
        LDS, weights fragment-major).  The range scales are applied exactly as the two-kernel path does (``_gemm``): LN scale
        divided out of W1's unscale, the hidden scale applied before the split and divided out of W2's."""
        c = self.a.dims[i]
        ws = []
        for name in (q + "w1", q + "w2"):
            s_ = self.Ws.get(name)
            if s_ is None:
                wt = self.P[name]
                s_ = self.Ws[name] = L.split_weights(wt.view(wt.shape[0], -1))
            if wide:
                f_ = self.Wf.get(name)
                if f_ is None:
                    n_, k_ = (4 * c, c) if name.endswith("w1") else (c, 4 * c)
                    f_ = self.Wf[name] = (L.mlp_wide_pack(s_[0], n_, k_), s_[1])
                s_ = f_
            ws.append(s_)
        sa, sh = self.sscale.get(q + "ln", 1.0), self.sscale.get(q + "hid", 1.0)
        kw = {}
        if wide:
            if self._park_mlp is None:
                self._park_mlp = self.park if self.park is not None else torch.zeros(
                    max(1, L.p8_workspace_bytes() // 4), dtype=torch.float32, device=self.dev)
            kw["workspace"] = self._park_mlp if self._persist_ok("mlp") else None
        fn = L.mlp_fused_wide if wide else L.mlp_fused
        fn(self.tmp, self.M[i], c, 4 * c, (ws[0][0], ws[0][1] / sa), self.P[q + "b1"], (ws[1][0], ws[1][1] / sh),
           self.P[q + "b2"], self.x[i], hid_scale=sh, range_flag=self.range_flag, **kw)

    def _neck_split(self) -> bool:
        """Is the neck / head running on pre-split activations in this step?  Not when the tower fell back to fp32 kernels,
        nor when the five neck input layers are pinned to fp32 (their outputs would have to be re-split)."""
        return self.presplit_neck and self.precision == "fp16x3" and not self.neck_pin

    @staticmethod
    def unsplit(t: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
        """fp32 view of a tensor stored as fp16 hi/lo groups ([hi x8 | lo x8] per 8 channels): hi + lo, i.e. the producer's
        fp32 value to 2^-22 relative.  ``scale``: the split scale the producer multiplied by (calibrate(); a power of two,
        divided out exactly).  Diagnostics / tests only — the hot path never converts back."""
        rows, c = t.shape
        h = t.contiguous().view(torch.float16).view(rows, c // 8, 2, 8)
        v = (h[:, :, 0].float() + h[:, :, 1].float()).reshape(rows, c)
        return v if scale == 1.0 else v / scale

    def pyramid(self) -> List[torch.Tensor]:
        """P3, P4, P5 as fp32 rows whatever format the neck wrote them in (split scales divided out)."""
        ps = [self.p3, self.p4, self.p5]
        if not self._neck_split():
            return ps
        return [self.unsplit(p, self.sscale.get(k, 1.0)) for p, k in zip(ps, ("p3", "p4", "p5"))]

    def _conv(self, a, w, b, c, *, hin, win, cin, lda, n, ldc, k=1, stride=1, act=L.ACT_NONE, res=None, ldres=0,
              res_alpha=1.0, **kw):
        self._gemm(a, w, b, c, hin=hin, win=win, cin=cin, lda=lda, kh=k, kw=k, stride=stride,
                   pad=(k // 2 if k == 3 else 0), n=n, ldc=ldc, act=act, res=res, ldres=ldres, res_alpha=res_alpha, **kw)

    # ------------------------------------------------------------------ DAG lanes (side streams)
    N_LANES = 4            # lane 0 = the caller's stream; 1, 2 = branch work (BiFusion inputs, head cls / reg); 3 = BepC3 cv2

    def _dag_on(self) -> bool:
        """Side streams are used for a step only outside the calibration pass (its recorders are torch ops on the current
        stream) and without the opt-in latency split-K (one shared workspace)."""
        if self._nh_issue and not self._dag_forced:
            # round 6: a neck / head pipelined beside the NEXT batch's backbone (detect(overlap_post=True), _pipe_neck_on) is one
            # serial chain on the nh stream: the backbone fills its under-filled launches, and caller + image chain + nh + post are
            # exactly the runtime's four hardware queues — three more lane streams would share queues with them and serialise
            # behind the backbone (Base B = 32: 963 images/s against 934 with the lanes, profiles/r06_pipeline.txt)
            return False
        return (self.dag and self._calib is None
                and (self._dag_in_capture or not torch.cuda.is_current_stream_capturing()))

    def _lane_kws(self, need: int = 0) -> Optional[torch.Tensor]:
        """Latency-mode split-K workspace of the lane that is launching, at least ``need`` floats (None when the mode is off).
        Grown on demand — the split count of a layer does not depend on the batch, its partial sums do — and only outside
        stream capture (a captured step has run eagerly before: GraphedDetect warms up)."""
        if self.kws is None:
            return None
        # a neck / head issued on the nh stream runs beside the NEXT step's backbone (lane 0 of the caller's stream): its lanes,
        # lane 0 included, take workspaces of their own
        li = self._lane_i + (16 if self._nh_issue else 0)
        w = self.kws if li == 0 else self._kws_lane.get(li)
        if w is None or w.numel() < need:
            if torch.cuda.is_current_stream_capturing():
                raise L.WedetectHipError("split-K workspace must be grown before stream capture (run the step eagerly once)")
            w = torch.empty(max(need, self.kws.numel()), dtype=torch.float32, device=self.dev)
            if li == 0:
                self.kws = w
                self.generation += 1
            else:
                self._kws_lane[li] = w
        return w

    def _lane_fws(self) -> torch.Tensor:
        """Fixed split-K workspace of the lane that is launching: two lanes may run split-K convs at the same time."""
        if self._lane_i == 0:
            return self.fws
        w = self._fws_lane.get(self._lane_i)
        if w is None:
            w = self._fws_lane[self._lane_i] = torch.empty_like(self.fws)
        return w

    class _Lane:
        def __init__(self, tower, i):
            self.t, self.i, self.ctx = tower, i, None

        def __enter__(self):
            t = self.t
            self.prev = t._lane_i
            t._lane_i = self.i
            if self.i > 0:
                self.ctx = torch.cuda.stream(t._side[self.i - 1])
                self.ctx.__enter__()
            return self

        def __exit__(self, *a):
            if self.ctx is not None:
                self.ctx.__exit__(*a)
            self.t._lane_i = self.prev

    def _lane(self, i: int) -> "ImageTower._Lane":
        if len(self._side) < self.N_LANES - 1:
            self._side = device_streams(self.dev, "lanes", self.N_LANES - 1)
        return ImageTower._Lane(self, i)

    def _mark(self) -> torch.cuda.Event:
        """Event recorded on the CURRENT stream, from a per-tower pool (no allocation per step; re-recorded every step)."""
        if self._ev_i == len(self._events):
            self._events.append(torch.cuda.Event())
        ev = self._events[self._ev_i]
        self._ev_i += 1
        ev.record(torch.cuda.current_stream())
        return ev

    @staticmethod
    def _after(*evs) -> None:
        """The current stream waits for the given events."""
        st = torch.cuda.current_stream()
        for ev in evs:
            if ev is not None:
                st.wait_event(ev)

    # ------------------------------------------------------------------ backbone
    BB_CHAINS_MIN_PIXELS = 32 * 640 * 640
    PIPE_NECK_MIN_PIXELS = 0                 # every size gains (B = 1 ... 64, profiles/r06_pipeline.txt)

    def _n_chains(self) -> int:
        """Image chains of this step's backbone (see __init__): 1 inside the calibration pass (its recorders are torch ops on
        the current stream), in the latency split-K classes, and wherever the DAG rule keeps a captured step on one stream."""
        if self.bb_chains == "1" or self._calib is not None or self.kws is not None:
            return 1
        if torch.cuda.is_current_stream_capturing() and not self._dag_in_capture:
            return 1
        n = 2 if self.bb_chains == "auto" else int(self.bb_chains)
        if self.bb_chains == "auto" and self._depth2_issue:
            return 1                # two whole backbones are in flight instead (detect(overlap_post=True), _bb_depth)
        if self.bb_chains == "auto" and not (self.BB_CHAINS_MIN_PIXELS <= self.B * self.H * self.W < 2 * self.BB_CHAINS_MIN_PIXELS):
            return 1                # measured window (profiles/r06_pipeline.txt): pays at 32 x 640 x 640, loses at 16 x and 64 x
        while n > 1 and (self.B % n or any(t.numel() % (64 * n) for t in (self.tmp, self.hid, self.ln_part, self.ln_stats) if t is not None)):
            n -= 1
        return max(1, n)

    def _chain_views(self, n: int) -> List["ImageTower"]:
        """``n`` shallow copies of the tower, each seeing batch B / n: its images' ROWS of the residual streams c1..c4 (the very
        buffers the neck reads afterwards), its own n-th of the scratch buffers, its own park workspace (two persistent launches
        may be in flight at once).  Weights, split caches, scales and flags are shared by reference; built per step — the
        towers' switches (precision fallback, pins) may change between steps."""
        if self._park_mlp is None:
            self._park_mlp = self.park if self.park is not None else torch.zeros(
                max(1, L.p8_workspace_bytes() // 4), dtype=torch.float32, device=self.dev)
        part = lambda t, h: None if t is None else t[h * (t.numel() // n): (h + 1) * (t.numel() // n)]
        views = []
        for h in range(n):
            v = copy.copy(self)
            v.B = self.B // n
            v.M = [m // n for m in self.M]
            v.x = [t[h * (m // n): (h + 1) * (m // n)] for t, m in zip(self.x, self.M)]
            v.patches = self.patches[h * v.M[0]: (h + 1) * v.M[0]]
            v.tmp, v.hid, v.ln_part, v.ln_stats = part(self.tmp, h), part(self.hid, h), part(self.ln_part, h), part(self.ln_stats, h)
            v._lane_i = 0
            if h:
                if h not in self._chain_parks:
                    if torch.cuda.is_current_stream_capturing():
                        raise L.WedetectHipError("image-chain workspaces must exist before stream capture (run the step eagerly once)")
                    self._chain_parks[h] = torch.zeros(max(1, L.p8_workspace_bytes() // 4), dtype=torch.float32, device=self.dev)
                v.park = self._chain_parks[h] if self.park is not None else None
                v._park_mlp = self._chain_parks[h]
            views.append(v)
        return views

    def backbone(self, images_u8: torch.Tensor) -> List[torch.Tensor]:
        B = self.B
        if images_u8.dtype != torch.uint8 or tuple(images_u8.shape) != (B, self.H, self.W, 3):
            raise L.WedetectHipError(f"images must be uint8 [{B},{self.H},{self.W},3] (RGB, NHWC)")
        if not images_u8.is_contiguous():
            images_u8 = images_u8.contiguous()
        if self._x_sets and self._x_free[self._x_par] is not None and not torch.cuda.is_current_stream_capturing():
            # a pipelined step's neck (nh stream) may still be reading the c1..c4 set this backbone is about to overwrite
            torch.cuda.current_stream().wait_event(self._x_free[self._x_par])
        n = self._n_chains()
        first, last = self.bb_chain_stages if n > 1 else (0, 3)
        # the stages outside [first, last] run as ONE chain over the whole batch on the caller's stream
        if n == 1 or first > 0:
            for _ in self._backbone_blocks(images_u8, 0, 3 if n == 1 else first - 1):
                pass
        if n > 1:
            self._backbone_chains(images_u8, n, first, last)
            if last < 3:
                for _ in self._backbone_blocks(None, last + 1, 3):
                    pass
        return self.x

    def _backbone_chains(self, images_u8: torch.Tensor, n: int, first: int, last: int) -> None:
        """Stages ``first`` .. ``last`` (with the stem when first == 0) as ``n`` image chains: chain 0 on the caller's stream,
        chain h on its own stream; launches are issued phase by phase in turn so that no stream's queue runs dry while the host
        is busy with another's; the caller's stream continues when all chains have arrived."""
        views = self._chain_views(n)
        if len(self._chain_streams) < n - 1:           # their own streams: the DAG lanes may be running the previous step's neck
            self._chain_streams = device_streams(self.dev, "chains", n - 1)
        while len(self._chain_evs) < 3 * n:
            self._chain_evs.append(torch.cuda.Event())
        done, ev_dw, ev_mlp = self._chain_evs[:n], self._chain_evs[n:2 * n], self._chain_evs[2 * n:3 * n]
        main = torch.cuda.current_stream()
        streams = [main] + self._chain_streams[: n - 1]
        done[0].record(main)
        gens, pending = [], []
        for h, v in enumerate(views):
            if h:
                streams[h].wait_event(done[0])          # everything issued before (the previous neck reads c1..c4; the stages before ``first``)
            gens.append(ImageTower._backbone_blocks(v, images_u8[h * v.B: (h + 1) * v.B], first, last))
            pending.append(next(gens[h]))               # the phase the chain issues next
        order = self.bb_chain_order
        rec_dw, rec_mlp = [False] * n, [False] * n
        alive = list(range(n))
        while alive:
            for h in list(alive):
                tag, before = pending[h], (h - 1) % n
                with (torch.cuda.stream(streams[h]) if h else contextlib.nullcontext()):
                    # phase order ACROSS chains ($WEDETECT_BB_CHAIN_ORDER; measured, profiles/r06_pipeline.txt: "free" wins): "dw" =
                    # the depthwise phases take turns (a chain's HBM-bound phase then always runs beside another chain's GEMMs),
                    # "gemm" = the block MLPs take turns
                    if tag == "dw" and order in ("dw", "both") and rec_dw[before]:
                        streams[h].wait_event(ev_dw[before])
                    if tag == "mlp" and order in ("gemm", "both") and rec_mlp[before]:
                        streams[h].wait_event(ev_mlp[before])
                    try:
                        pending[h] = next(gens[h])
                    except StopIteration:
                        alive.remove(h)
                        if h:
                            done[h].record(streams[h])
                    if tag == "dw" and order in ("dw", "both"):
                        ev_dw[h].record(streams[h])
                        rec_dw[h] = True
                    if tag == "mlp" and order in ("gemm", "both"):
                        ev_mlp[h].record(streams[h])
                        rec_mlp[h] = True
        for h in range(1, n):
            main.wait_event(done[h])

    def _backbone_blocks(self, images_u8: Optional[torch.Tensor], first: int = 0, last: int = 3):
        """The backbone's launches on the current stream as a generator: yields the name of the phase it issues NEXT — "stem",
        "down" (downsample LayerNorm + conv), "dw" (depthwise 7 x 7 with its LayerNorm / statistics), "mlp" (the block MLP) —
        ``backbone`` interleaves the image chains' launches at these points."""
        a, B = self.a, self.B
        if first == 0:
            yield "stem"
            self._stem(images_u8)
        for i in range(first, last + 1):
            yield from self._stage(i)

    def _stem(self, images_u8: torch.Tensor) -> None:
        a = self.a
        if self.fuse_stem and a.dims[0] in L.STEM_FUSED_WIDTHS:
            # patchify + conv + LayerNorm in one fp32 kernel: the image is read once, the rows written once
            L.stem_fused(images_u8, self.P["stem.w"], self.P["stem.b"], self.P["stem.ln_w"], self.P["stem.ln_b"], self.x[0])
        else:
            L.stem_patchify(images_u8, self.patches)
            h0, w0 = self.hw[0]
            self._conv(self.patches, "stem.w", "stem.b", self.x[0], hin=h0, win=w0, cin=48, lda=48, n=a.dims[0],
                       ldc=a.dims[0])
            L.layernorm_rows(self.x[0], self.x[0], self.P["stem.ln_w"], self.P["stem.ln_b"], self.M[0], a.dims[0])

    def _stage(self, i: int):
        """Stage ``i`` of the backbone (downsample, then its blocks) as a generator, see ``_backbone_blocks``."""
        a, B = self.a, self.B
        c = a.dims[i]
        h, w = self.hw[i]
        # fp16x3: LayerNorm and pwconv1 write their outputs as fp16 hi/lo groups, so the GEMMs that
        # consume them copy both operands (no per-column-tile conversion of the same rows)
        pre = self.precision == "fp16x3" and c % 8 == 0 and (i == 0 or a.dims[i - 1] % 8 == 0)
        fa = L.SPLIT_A if pre else 0
        if i > 0:
            yield "down"
            cp = a.dims[i - 1]
            hp, wp = self.hw[i - 1]
            if pre and hp % 2 == 0 and wp % 2 == 0 and self.s2d_down:
                # LayerNorm writes the 2 x 2 / stride-2 convolution's GEMM rows directly (space-to-depth, (kh, kw, cin)
                # column order = the packed weight's): the downsample runs as a plain pre-split GEMM with K = 4 cp on
                # the DMA-fed kernels instead of the register-staged conv loader.  Same K order, same bits.
                g_, b_ = self._ln_params(f"down{i}.ln_w", f"down{i}.ln_b", f"down{i}.ln")
                L.layernorm_rows_split_s2d(self.x[i - 1], self.tmp, g_, b_, B, hp, wp, cp)
                self._gemm(self.tmp, f"down{i}.w", f"down{i}.b", self.x[i], hin=h, win=w, cin=4 * cp, lda=4 * cp, n=c,
                           ldc=c, split_flags=fa, a_key=f"down{i}.ln")
            else:
                g_, b_ = self._ln_params(f"down{i}.ln_w", f"down{i}.ln_b", f"down{i}.ln" if pre else None)
                L.layernorm_rows(self.x[i - 1], self.tmp, g_, b_, self.M[i - 1], cp, split=pre)
                self._record(f"down{i}.ln", self.tmp[: self.M[i - 1] * cp])
                self._gemm(self.tmp, f"down{i}.w", f"down{i}.b", self.x[i], hin=hp, win=wp, cin=cp, lda=cp, kh=2, kw=2,
                           stride=2, pad=0, n=c, ldc=c, split_flags=fa, a_key=f"down{i}.ln" if pre else None)
        for j in range(a.depths[i]):
            q = f"s{i}.{j}."
            yield "dw"
            if self._fold_ok(i, pre) and not self.sscale.get(q + "fold_off"):
                # LayerNorm folded into pwconv1: dwconv -> (split d, block statistics) -> row statistics -> GEMM
                self._prepare_fold()            # buffers and folded weights exist since __init__; a switch flipped later lands here
                self._fold_weights(q)
                L.dwconv7_stats(self.x[i], self.P[q + "dw_w"], self.P[q + "dw_b"], self.tmp, self.ln_part, B, h, w, c,
                                scale=self.sscale.get(q + "dw", 1.0))
                L.ln_stats_finalize(self.ln_part, self.ln_stats, self.M[i], c)
                yield "mlp"
                if c in self.fuse_mlp_wide and L.mlp_wide_supported(self.M[i], c, 4 * c):
                    self._mlp_fused_fold(q, i)
                    continue
                self._conv(self.tmp, q + "w1g", q + "v", self.hid, hin=h, win=w, cin=c, lda=c, n=4 * c, ldc=4 * c,
                           act=L.ACT_GELU, split_flags=L.SPLIT_A | L.SPLIT_C, a_key=q + "dw", c_key=q + "hid",
                           ln_stats=self.ln_stats, ln_u=self.P[q + "u"])
                self._conv(self.hid, q + "w2", q + "b2", self.x[i], hin=h, win=w, cin=4 * c, lda=4 * c, n=c, ldc=c,
                           res=self.x[i], ldres=c, split_flags=fa, a_key=q + "hid")
                continue
            g_, b_ = self._ln_params(q + "ln_w", q + "ln_b", q + "ln" if pre else None)
            if c % 32 == 0 and (self.fuse_dwln == "1" or (self.fuse_dwln == "auto" and (c <= 128 or c in self.fuse_dwln_wide))):   # bit-identical to the pair
                L.dwconv7_ln(self.x[i], self.P[q + "dw_w"], self.P[q + "dw_b"], self.tmp, g_, b_, B, h, w, c, split=pre)
            else:
                L.dwconv7(self.x[i], self.P[q + "dw_w"], self.P[q + "dw_b"], self.tmp, B, h, w, c)
                if self._calib is not None and self.ln_fold:
                    self._record(q + "dw", self.tmp[: self.M[i] * c])      # the folded path splits the pre-norm tensor itself
                    # and centres AFTER the contraction: rstd (W'd - mean u) loses |mean| / std x 2^-22 of the output to
                    # cancellation.  Record the worst row's |mean| / std; calibrate() keeps the LayerNorm kernel for a block
                    # where it exceeds FOLD_MAX_MEAN_OVER_STD
                    d = self.tmp[: self.M[i] * c].view(self.M[i], c)
                    self._calib[q + "dw.mr"] = (d.mean(dim=1).abs() / d.std(dim=1, unbiased=False).clamp_min(1e-30)).max()
                L.layernorm_rows(self.tmp, self.tmp, g_, b_, self.M[i], c, split=pre)
            self._record(q + "ln", self.tmp[: self.M[i] * c])
            yield "mlp"
            if pre and self.fuse_mlp and L.mlp_fused_supported(self.M[i], c, 4 * c):
                self._mlp_fused(q, i)       # pwconv1 -> GELU -> pwconv2 -> residual in one kernel: same bits
                continue
            if pre and c in self.fuse_mlp_wide and L.mlp_wide_supported(self.M[i], c, 4 * c):
                self._mlp_fused(q, i, wide=True)
                continue
            self._conv(self.tmp, q + "w1", q + "b1", self.hid, hin=h, win=w, cin=c, lda=c, n=4 * c, ldc=4 * c,
                       act=L.ACT_GELU, split_flags=(L.SPLIT_A | L.SPLIT_C) if pre else 0,
                       a_key=q + "ln" if pre else None, c_key=q + "hid")
            # x <- x + (gamma*W2) hid + gamma*b2   (in place: each element is read then written by one lane)
            self._conv(self.hid, q + "w2", q + "b2", self.x[i], hin=h, win=w, cin=4 * c, lda=4 * c, n=c, ldc=c,
                       res=self.x[i], ldres=c, split_flags=fa, a_key=q + "hid" if pre else None)

    # ------------------------------------------------------------------ neck
    def _bepc3(self, name: str, x, ldx: int, cin: int, hw: Tuple[int, int], out, cout: int, xkey: str, okey: str):
        """xkey / okey: split-scale keys of the input / output buffers (calibrate()).  DAG mode: cv2 (a 1 x 1 conv of the block
        input into the second half of ``cat``) runs on lane 3 beside the 3 x 3 chain; cv3 waits for both."""
        bf = self._bep[name]
        c_ = bf["c_"]
        h, w = hw
        nb = self.a.neck_repeats // 2
        cat = bf["cat"]
        S = self._neck_split()
        fl = (L.SPLIT_A | L.SPLIT_C) if S else 0
        sfx = "s" if S else ""
        D = self._dag_on()

        def cv2():
            self._conv(x, f"{name}.cv2.w", f"{name}.cv2.b", cat[:, c_:], hin=h, win=w, cin=cin, lda=ldx, n=c_,
                       ldc=2 * c_, act=L.ACT_SILU, split_flags=fl, a_key=xkey, c_key=name + ".cat")
        ev_cv2 = None
        if D:
            ev_in = self._mark()                    # the block input is complete on the caller's stream
            with self._lane(3):
                self._after(ev_in)
                cv2()
                ev_cv2 = self._mark()
        # cv1 output: the 3x3 chain reads it (split) AND the first BottleRep adds it back (fp32): dual write
        self._conv(x, f"{name}.cv1.w", f"{name}.cv1.b", bf["u0" + sfx], hin=h, win=w, cin=cin, lda=ldx, n=c_, ldc=c_,
                   act=L.ACT_SILU, split_flags=fl, c2=bf["u0"] if S else None, ldc2=c_ if S else 0, a_key=xkey, c_key=name + ".u0")
        cur, nxt = "u0", "u1"
        for j in range(nb):
            s = f"{name}.m{j}"
            self._conv(bf[cur + sfx], s + ".c1.w", s + ".c1.b", bf["t"], hin=h, win=w, cin=c_, lda=c_, n=c_, ldc=c_, k=3,
                       act=L.ACT_SILU, split_flags=fl, a_key=f"{name}.{cur}", c_key=name + ".t")
            last = j == nb - 1
            dst, ldd = (cat, 2 * c_) if last else (bf[nxt + sfx], c_)
            self._conv(bf["t"], s + ".c2.w", s + ".c2.b", dst, hin=h, win=w, cin=c_, lda=c_, n=c_, ldc=ldd, k=3,
                       act=L.ACT_SILU, res=bf[cur], ldres=c_, res_alpha=self.P.s[s + ".alpha"], split_flags=fl,
                       c2=bf[nxt] if (S and not last) else None, ldc2=c_ if (S and not last) else 0,
                       a_key=name + ".t", c_key=name + ".cat" if last else f"{name}.{nxt}")
            cur, nxt = nxt, cur
        if D:
            self._after(ev_cv2)
        else:
            cv2()
        self._conv(cat, f"{name}.cv3.w", f"{name}.cv3.b", out, hin=h, win=w, cin=2 * c_, lda=2 * c_, n=cout,
                   ldc=cout, act=L.ACT_SILU, split_flags=fl, a_key=name + ".cat", c_key=okey)

    def _bifusion_mid(self, name: str, mid, c_mid: int, hw_mid, cat, cout: int, midkey: str):
        hm, wm = hw_mid
        fc = L.SPLIT_C if self._neck_split() else 0       # fp32 residual stream in (split by the loader), hi/lo groups out
        self._conv(mid, name + ".cv1.w", name + ".cv1.b", cat[:, cout:], hin=hm, win=wm, cin=c_mid, lda=c_mid, n=cout,
                   ldc=3 * cout, act=L.ACT_RELU, fp32=True, split_flags=fc, a_key=midkey, c_key=name + ".cat")   # mid / low: residual streams

    def _bifusion_low(self, name: str, low, c_low: int, hw_low, cat, tbuf, cout: int, lowkey: str):
        hl, wl = hw_low
        S = self._neck_split()
        fl = (L.SPLIT_A | L.SPLIT_C) if S else 0
        fc = L.SPLIT_C if S else 0
        self._conv(low, name + ".cv2.w", name + ".cv2.b", tbuf, hin=hl, win=wl, cin=c_low, lda=c_low, n=cout, ldc=cout,
                   act=L.ACT_RELU, fp32=True, split_flags=fc, a_key=lowkey, c_key=name + ".t")
        self._conv(tbuf, name + ".downsample.w", name + ".downsample.b", cat[:, 2 * cout:], hin=hl, win=wl, cin=cout,
                   lda=cout, n=cout, ldc=3 * cout, k=3, stride=2, act=L.ACT_RELU, split_flags=fl, a_key=name + ".t",
                   c_key=name + ".cat")

    def _bifusion(self, name: str, top, ld_top: int, hw_top, mid, c_mid: int, hw_mid, low, c_low: int, hw_low,
                  cat, tbuf, out, cout: int, topkey: str, midkey: str, lowkey: str, okey: str, pre=None):
        """cat = [upsample(top) | cv1(mid) | downsample(cv2(low))] -> cv3 -> out  (yolo_world_pafpn.py:711-715).
        ``pre``: events of the mid / low branches when the caller already issued them on side lanes (DAG mode: they read
        backbone outputs only)."""
        ht, wt = hw_top
        hm, wm = hw_mid
        fl = (L.SPLIT_A | L.SPLIT_C) if self._neck_split() else 0
        self._gemm(top, name + ".up.w", name + ".up.b", cat, hin=ht, win=wt, cin=cout, lda=ld_top, n=4 * cout,
                   ldc=3 * cout, out_mode=L.OUT_DECONV2X2, split_flags=fl, a_key=topkey, c_key=name + ".cat")
        if pre is None:
            self._bifusion_mid(name, mid, c_mid, hw_mid, cat, cout, midkey)
            self._bifusion_low(name, low, c_low, hw_low, cat, tbuf, cout, lowkey)
        else:
            self._after(*pre)
        self._conv(cat, name + ".cv3.w", name + ".cv3.b", out, hin=hm, win=wm, cin=3 * cout, lda=3 * cout, n=cout,
                   ldc=cout, act=L.ACT_RELU, split_flags=fl, a_key=name + ".cat", c_key=okey)

    def neck(self, _level_ready=None) -> List[torch.Tensor]:
        """P3, P4, P5 buffers.  In the pre-split mode (see __init__) they hold fp16 hi/lo groups, like every other neck
        buffer: ``pyramid()`` gives fp32 views for diagnostics.  ``_level_ready(l)`` (features()): called the moment P3 / P4 /
        P5 is complete on the caller's stream, so that a head level can be issued beside the rest of the neck."""
        a = self.a
        nc = a.neck_channels
        c1, c2, c3, c4 = self.x
        hw2, hw3, hw4, hw5 = self.hw
        ld4 = nc["d1"] + nc["p5r"]
        ld3 = nc["d2"] + nc["p4r"]
        S = self._neck_split()
        fl = (L.SPLIT_A | L.SPLIT_C) if S else 0
        fpn_out0 = self.cat_n4[:, nc["d1"]:]
        fpn_out1 = self.cat_n3[:, nc["d2"]:]
        D = self._dag_on()
        self._ev_i = 0
        self._head_evs = []
        pre0 = pre1 = None
        if D:
            # the mid / low branches of BOTH BiFusion blocks read backbone outputs only: lanes 1 and 2, from the start
            ev0 = self._mark()
            with self._lane(1):
                self._after(ev0)
                self._bifusion_mid("Bifusion0", c3, nc["c3"], hw4, self.cat_b0, nc["p5r"], "c3")
                e_m0 = self._mark()
                self._bifusion_mid("Bifusion1", c2, nc["c2"], hw3, self.cat_b1, nc["p4r"], "c2")
                e_m1 = self._mark()
            with self._lane(2):
                self._after(ev0)
                self._bifusion_low("Bifusion0", c2, nc["c2"], hw3, self.cat_b0, self.b0_t, nc["p5r"], "c2")
                e_l0 = self._mark()
                self._bifusion_low("Bifusion1", c1, nc["c1"], hw2, self.cat_b1, self.b1_t, nc["p4r"], "c1")
                e_l1 = self._mark()
            pre0, pre1 = (e_m0, e_l0), (e_m1, e_l1)
        self._conv(c4, "reduce_layer0.w", "reduce_layer0.b", fpn_out0, hin=hw5[0], win=hw5[1], cin=nc["c4"],
                   lda=nc["c4"], n=nc["p5r"], ldc=ld4, act=L.ACT_RELU, fp32=True, split_flags=L.SPLIT_C if S else 0,
                   a_key="c4", c_key="cat_n4")
        self._bifusion("Bifusion0", fpn_out0, ld4, hw5, c3, nc["c3"], hw4, c2, nc["c2"], hw3, self.cat_b0, self.b0_t,
                       self.f0, nc["p5r"], "cat_n4", "c3", "c2", "f0", pre=pre0)
        self._bepc3("Rep_p4", self.f0, nc["p5r"], nc["p5r"], hw4, self.f_out0, nc["p5r"], "f0", "f_out0")
        self._conv(self.f_out0, "reduce_layer1.w", "reduce_layer1.b", fpn_out1, hin=hw4[0], win=hw4[1], cin=nc["p5r"],
                   lda=nc["p5r"], n=nc["p4r"], ldc=ld3, act=L.ACT_RELU, split_flags=fl, a_key="f_out0", c_key="cat_n3")
        self._bifusion("Bifusion1", fpn_out1, ld3, hw4, c2, nc["c2"], hw3, c1, nc["c1"], hw2, self.cat_b1, self.b1_t,
                       self.f1, nc["p4r"], "cat_n3", "c2", "c1", "f1", pre=pre1)
        self._bepc3("Rep_p3", self.f1, nc["p4r"], nc["p4r"], hw3, self.p3, nc["p4r"], "f1", "p3")
        if _level_ready is not None:
            _level_ready(0)
        self._conv(self.p3, "downsample2.w", "downsample2.b", self.cat_n3, hin=hw3[0], win=hw3[1], cin=nc["p4r"],
                   lda=nc["p4r"], n=nc["d2"], ldc=ld3, k=3, stride=2, act=L.ACT_RELU, split_flags=fl, a_key="p3", c_key="cat_n3")
        self._bepc3("Rep_n3", self.cat_n3, ld3, ld3, hw4, self.p4, nc["n3"], "cat_n3", "p4")
        if _level_ready is not None:
            _level_ready(1)
        self._conv(self.p4, "downsample1.w", "downsample1.b", self.cat_n4, hin=hw4[0], win=hw4[1], cin=nc["n3"],
                   lda=nc["n3"], n=nc["d1"], ldc=ld4, k=3, stride=2, act=L.ACT_RELU, split_flags=fl, a_key="p4", c_key="cat_n4")
        self._bepc3("Rep_n4", self.cat_n4, ld4, ld4, hw5, self.p5, nc["n4"], "cat_n4", "p5")
        self._ev_after_neck = self._ev_i
        if _level_ready is not None:
            _level_ready(2)
        return [self.p3, self.p4, self.p5]

    # ------------------------------------------------------------------ head
    def _head_level(self, l: int) -> None:
        """One level of the head.  DAG mode: the classification branch (3 x 3, 3 x 3, 256 -> 768 embedding conv) on lane 1 and
        the regression branch (3 x 3, 3 x 3, DFL logits, decode) on lane 2 — the last level keeps its classification branch on
        the caller's stream, which has nothing else left to run; ``_head_join`` makes the caller's stream wait for all of it."""
        feat, cin = (self.p3, self.p4, self.p5)[l], self.a.head_in[l]
        S = self._neck_split()                  # P3..P5 and the branch intermediates as fp16 hi/lo groups; embeddings and
        fl = (L.SPLIT_A | L.SPLIT_C) if S else 0     # DFL logits (read by fp32 kernels) stay fp32
        fa = L.SPLIT_A if S else 0
        h, w = self.lv[l]

        def cls_branch():
            c1, c2 = self.hc[l]
            self._conv(feat, f"head{l}.cls0.w", f"head{l}.cls0.b", c1, hin=h, win=w, cin=cin, lda=cin, n=CLS_MID,
                       ldc=CLS_MID, k=3, act=L.ACT_SILU, split_flags=fl, a_key=f"p{l + 3}", c_key=f"h{l}.c1")
            self._conv(c1, f"head{l}.cls1.w", f"head{l}.cls1.b", c2, hin=h, win=w, cin=CLS_MID, lda=CLS_MID, n=CLS_MID,
                       ldc=CLS_MID, k=3, act=L.ACT_SILU, split_flags=fl, a_key=f"h{l}.c1", c_key=f"h{l}.c2")
            dst = self.embed.view(-1, EMBED_DIM)[self.off[l]:]
            if self._embed_split_on and S:
                # hi/lo groups for the fp16x3 similarity GEMM + the fp32 rows everybody else reads, same per-image row map
                self._conv(c2, f"head{l}.embed.w", f"head{l}.embed.b", self.embed_s[self.off[l]:], hin=h, win=w, cin=CLS_MID,
                           lda=CLS_MID, n=EMBED_DIM, ldc=EMBED_DIM, c_batch_stride=self.ntot, split_flags=fl, a_key=f"h{l}.c2",
                           c_key="embed", c2=dst, ldc2=EMBED_DIM)
            else:
                self._conv(c2, f"head{l}.embed.w", f"head{l}.embed.b", dst, hin=h, win=w, cin=CLS_MID, lda=CLS_MID,
                           n=EMBED_DIM, ldc=EMBED_DIM, c_batch_stride=self.ntot, split_flags=fa, a_key=f"h{l}.c2")

        def reg_branch():
            r1, r2, dist = self.hr[l]
            self._conv(feat, f"head{l}.reg0.w", f"head{l}.reg0.b", r1, hin=h, win=w, cin=cin, lda=cin, n=REG_MID,
                       ldc=REG_MID, k=3, act=L.ACT_SILU, split_flags=fl, a_key=f"p{l + 3}", c_key=f"h{l}.r1")
            self._conv(r1, f"head{l}.reg1.w", f"head{l}.reg1.b", r2, hin=h, win=w, cin=REG_MID, lda=REG_MID, n=REG_MID,
                       ldc=REG_MID, k=3, act=L.ACT_SILU, split_flags=fl, a_key=f"h{l}.r1", c_key=f"h{l}.r2")
            self._conv(r2, f"head{l}.dist.w", f"head{l}.dist.b", dist, hin=h, win=w, cin=REG_MID, lda=REG_MID, n=64,
                       ldc=64, split_flags=fa, a_key=f"h{l}.r2")
            L.dfl_decode(dist, 64, self.boxes, self.B, h, w, STRIDES[l], self.off[l], self.ntot)
        if not self._dag_on():
            cls_branch()
            reg_branch()
            return
        ev = self._mark()                       # this level's pyramid map is complete on the caller's stream
        with self._lane(2):
            self._after(ev)
            reg_branch()
            self._head_evs.append(self._mark())
        if l == 2:
            cls_branch()
        else:
            with self._lane(1):
                self._after(ev)
                cls_branch()
                self._head_evs.append(self._mark())

    def _head_join(self) -> None:
        self._after(*self._head_evs)
        self._head_evs = []

    SIM_SPLIT_MIN = 256
    FALLBACK_RETRY, FALLBACK_RETRY_MAX = 64, 4096     # clean fp32 batches before a fallen-back tower tries fp16x3 again (doubles per relapse)
    # LayerNorm fold (rstd (W'd - mean u) + v): the centring after the contraction costs |mean| / std x 2^-22 of relative accuracy
    # on top of the fp16x3 kernels' own 2^-22; 64 keeps it at their level (1.5e-5).  A block beyond it keeps the LayerNorm kernel.
    FOLD_MAX_MEAN_OVER_STD = 64.0

    def want_sim_split(self, num_classes: Optional[int]) -> bool:
        """Will similarity() of a ``num_classes``-row bank run on the fp16x3 kernel (so that the head must write the split
        embeddings)?"""
        if num_classes is None or self.precision != "fp16x3" or not self._neck_split() or self.sim_split == "0" or self._calib is not None:
            return False
        return self.sim_split == "1" or num_classes >= self.SIM_SPLIT_MIN

    def _begin_head(self, num_classes: Optional[int], standalone: bool = False) -> None:
        if standalone:
            # a head() outside neck_head() (tests, diagnostics): rewind the event pool to where neck() left it, or repeated calls
            # would take fresh events forever
            self._ev_i = self._ev_after_neck
            self._head_evs = []
        self._embed_split_on = self.want_sim_split(num_classes)
        if self._embed_split_on and self.embed_s is None:
            rows = (self.B * self.ntot + 7) // 8 * 8
            self.embed_s = torch.empty(rows, EMBED_DIM, dtype=torch.float32, device=self.dev)
            self.generation += 1
        self._embed_split_valid = self._embed_split_on

    def head(self, num_classes: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Region embeddings (post contrastive-BN) [B, N, 768] and decoded boxes [B, N, 4].  ``num_classes``: size of the text
        bank similarity() will be called with (selects the fp16x3 similarity path for large banks, see __init__)."""
        self._begin_head(num_classes, standalone=True)
        for l in range(3):
            self._head_level(l)
        self._head_join()
        self._record("embed", self.embed)       # calibrate(): the split scale of the embeddings (every level written: the whole buffer)
        return self.embed, self.boxes

    # ------------------------------------------------------------------ similarity
    def similarity(self, text: torch.Tensor, normalize: bool, sigmoid: bool = True) -> torch.Tensor:
        """scores[b, n, k] = sigmoid(<embed[b,n], t_k> * exp(logit_scale_lvl) + bias_lvl).
        ``normalize``: L2-normalise the text rows first (BNContrastiveHead, yolo_world_head.py:101);
        the Uni path uses its prompt rows as stored (generate_proposal.py:1130)."""
        k = text.shape[0]
        if text.dtype != torch.float32 or text.shape[1] != EMBED_DIM or not text.is_cuda:
            raise L.WedetectHipError("text bank must be a device float32 [K, 768] tensor")
        if k > self.max_classes:
            self._alloc_post(k)
        if k > self.text_norm.shape[0]:
            self.text_norm = torch.empty(k, EMBED_DIM, dtype=torch.float32, device=self.dev)
            self.generation += 1
        t = text.contiguous()
        out = self.scores.view(-1)[: self.B * self.ntot * k].view(self.B, self.ntot, k)
        seg = (self.ntot, self.off[1], self.off[2], self.lvl_scale, self.lvl_bias)
        if self._embed_split_valid and self.precision == "fp16x3" and self.want_sim_split(k):
            ts = self._split_text(t, normalize)
            if ts is not None:
                L.similarity_split(self.embed_s, self.B * self.ntot, ts[0], ts[1] / self.sscale.get("embed", 1.0), out, k, EMBED_DIM, k,
                                   seg=seg, sigmoid=sigmoid, range_flag=self.range_flag)
                return out
        if normalize:
            L.l2norm_rows(t, self.text_norm[:k])
            t = self.text_norm[:k]
        L.conv_gemm(self.embed, t, None, out, batch=1, hin=1, win=self.B * self.ntot, cin=EMBED_DIM, lda=EMBED_DIM,
                    n=k, ldc=k, sigmoid=sigmoid, seg=seg)
        return out

    def _split_text(self, t: torch.Tensor, normalize: bool):
        """fp16 hi/lo groups of the (normalised) text rows, split ONCE per bank: an entry is valid for the same tensor OBJECT
        (held by weak reference: a new tensor that happens to land on a freed bank's address is another object) at the same
        version counter (an in-place update bumps it).  Splitting reads max |t| on the host — one sync per NEW bank — so a bank
        first seen under stream capture falls back to the fp32 kernel for that call (None).  Pass the bank as ONE contiguous
        tensor that lives across calls, as the detectors do; a temporary is split on every call."""
        for ref, ver, norm, ts in self._text_split:
            if ref() is t and ver == t._version and norm == bool(normalize):
                return ts
        if torch.cuda.is_current_stream_capturing():
            return None
        k = t.shape[0]
        src = t
        if normalize:
            L.l2norm_rows(t, self.text_norm[:k])
            src = self.text_norm[:k]
        ts = L.split_weights(src)
        self._text_split = [e for e in self._text_split if e[0]() is not None][-1:]     # the last two live banks
        self._text_split.append((weakref.ref(t), t._version, bool(normalize), ts))
        return ts

    # ------------------------------------------------------------------ post-process
    NMS_MODES = {"vanilla": L.NMS_VANILLA, "torchvision": L.NMS_TORCHVISION, "mmcv": L.NMS_MMCV}

    def postprocess(self, scores: torch.Tensor, score_thr: float, meta: torch.Tensor, iou_thr: float = 0.7,
                    with_embed: bool = True, nms: str = "vanilla", nms_param: Optional[int] = None,
                    nms_device: str = "cpu") -> Dict[str, torch.Tensor]:
        """scores [B, N, K] -> candidates (score desc, index asc, <= nms_pre) -> class-aware NMS
        -> <= max_out rows per image.  ``meta`` [B, 8] fp32 device tensor, see wd_nms_gather.
        ``nms``: which library's batched NMS is reproduced — "torchvision" (generate_proposal.py:1210; ``nms_param`` =
        the box-coordinate count above which it runs per class, 4000 for CPU tensors), "mmcv" (mmdet's
        _bbox_post_process, yolo_world_head.py:740-744; ``nms_param`` = split_thr, 10000) or "vanilla" (label test).
        ``nms_device``: the device kind the reference's tensors would be on — decides torchvision's branch limit default
        (4000 / 20000 box coordinates) and whether its kernel compares the IoU with a double ("cpu") or a float ("cuda")."""
        B, n, k = scores.shape
        mode = self.NMS_MODES[nms]
        if nms_param is None:
            nms_param = {L.NMS_VANILLA: 0, L.NMS_TORCHVISION: L.TV_TRICK_MAX_NUMEL[nms_device], L.NMS_MMCV: L.MMCV_SPLIT_THR}[mode]
        L.topk_candidates(scores, B, n * k, float(np.float32(score_thr)), self.nms_pre, self.cand_idx,
                          self.cand_score, self.cand_count, self.topk_ws)
        L.nms_gather(self.cand_idx, self.cand_score, self.cand_count, self.cap, self.boxes, n, k, meta,
                     L.nms_threshold(iou_thr, mode, nms_device), self.max_out, self.embed if with_embed else None, EMBED_DIM,
                     self.out_boxes, self.out_scores, self.out_labels, self.out_anchors, self.out_count,
                     self.out_embed if with_embed else None, B, nms_mode=mode, mode_param=int(nms_param),
                     workspace=self.nms_ws)
        res = dict(bboxes=self.out_boxes, scores=self.out_scores, labels=self.out_labels, anchors=self.out_anchors,
                   count=self.out_count)
        if with_embed:
            res["embeddings"] = self.out_embed
        return res

    # ------------------------------------------------------------------ fp16x3 range calibration
    SCALE_TARGET_LOG2 = 10       # calibrate() places max |x| of every split tensor at 2^10: 2^6 of headroom below the fp16 maximum

    def calibrate(self, images_u8: torch.Tensor, merge: bool = False) -> Dict[str, float]:
        """Choose the power-of-two split scales for this checkpoint's activation ranges from ONE batch.

        An fp32 value travels through the fp16x3 GEMMs as hi = fp16(x), lo = fp16(x - hi).  That pair reproduces x to 2^-22
        relative only while lo stays a NORMAL fp16 number, i.e. for |x| >~ 2^-3; below, lo is quantised to 2^-24 and the
        pair carries an ABSOLUTE error of ~3e-8.  That floor is harmless for a homogeneous O(1) tensor (it is below the
        fp32 rounding of the tensor's large elements), but a tensor that lives at 1e-4, or CHANNELS that do while others
        sit at 10 (residual streams with outlier channels, compensated by the next layer's weights), would be carried to
        3e-4 relative.  Above 65504 the hi half overflows (the run-time range guard catches that and falls back to fp32
        kernels).  This pass runs the tower once with the fp32 kernels on ``images_u8``, records max |x| of every tensor
        that is split (LayerNorm outputs, GELU hidden tensors, neck / head activations, the residual streams the neck
        reads) and gives each the power of two that moves its maximum to 2^10: elements down to 2^-13 of the maximum
        (four decades) keep full fp32 accuracy, 64 x of headroom remain before the guard would trip.  The producer
        multiplies by the scale right before the split (LayerNorm: folded into gamma / beta; GEMM epilogues:
        WdConvGemm.c_split_scale; loader-split layers: a_scale), the consumer's weight unscale divides it out — exact
        operations: values whose halves were normal before come out bit-identical, the others more accurate.
        Measured (tests/test_gpu_precision.py, error of the region embeddings against the fp32-MFMA tower, in units of
        their rms): streams at 1e-4: 5.4e-4 uncalibrated -> 7.5e-6; hidden activations at 1e-3: 9.4e-5 -> 8.4e-6; streams at
        3e4: guard trips -> 1.0e-5.  Returns the recorded maxima.  Detectors call it on the first batch a tower sees."""
        self.calibrated = True
        if self.precision != "fp16x3":
            return {}
        self.calibration_runs += 1
        saved = self.precision
        self.precision, self._calib = "fp32", {}
        try:
            self.features(images_u8)
            keys = list(self._calib)
            vals = torch.stack([self._calib[k] for k in keys]).tolist() if keys else []
        finally:
            self.precision, self._calib = saved, None
        amax = dict(zip(keys, vals))
        new = {}
        for k, m in amax.items():
            if k.endswith(".mr"):                             # worst |mean| / std of a block's pre-norm rows, not a range
                if not (m <= self.FOLD_MAX_MEAN_OVER_STD):
                    new[k[: -len("dw.mr")] + "fold_off"] = 2.0    # any value but 1.0: travels with the scales to every tower of the checkpoint
                continue
            if math.isfinite(m) and m > 0.0:
                sc = 2.0 ** (self.SCALE_TARGET_LOG2 - math.floor(math.log2(m)) - 1)      # max lands in [2^9, 2^10)
                if sc != 1.0:
                    new[k] = sc
        if merge:
            # re-calibration after the range guard tripped: never RAISE a scale the earlier batch chose (both batches must
            # stay below the fp16 maximum) — per tensor the smaller of the two
            keys = set(new) | set(self.sscale)
            off = {k for k in keys if k.endswith("fold_off")}                  # a block that lost its fold keeps the LayerNorm kernel
            new = {k: (2.0 if k in off else min(new.get(k, 1.0), self.sscale.get(k, 1.0))) for k in keys}
            new = {k: v for k, v in new.items() if v != 1.0}
        self.adopt_scales(new)
        self.fold_mean_over_std = {k[: -len(".mr")]: v for k, v in amax.items() if k.endswith(".mr")}      # diagnostics
        return {k: v for k, v in amax.items() if not k.endswith(".mr")}

    def adopt_scales(self, sscale: Dict[str, float]) -> None:
        """Take split scales chosen elsewhere (another tower of the same checkpoint: detector._TowerHolder calibrates ONCE per
        checkpoint so that an image gets the same bits whatever batch shape it arrives in)."""
        self.calibrated = True
        if dict(sscale) != self.sscale:
            self.sscale = dict(sscale)
            self._ln_scaled.clear()
            self.generation += 1               # scales are kernel arguments baked into captured graphs

    def identity_meta(self) -> torch.Tensor:
        """Letterbox metadata of a network-sized image: pad 0, scale 1, clamp to H x W."""
        m = torch.tensor([0.0, 0.0, 0.0, 1.0, 1.0, float(self.W), float(self.H), 0.0], dtype=torch.float32)
        return m.repeat(self.B, 1).to(self.dev)

    def level_of(self, anchors: torch.Tensor) -> torch.Tensor:
        a = anchors.clamp_min(0)
        return (a >= self.off[1]).to(torch.int64) + (a >= self.off[2]).to(torch.int64)

    # ------------------------------------------------------------------ whole steps
    def neck_head(self, num_classes: Optional[int] = None):
        """Neck and head as ONE schedule: in DAG mode a head level is issued on the side lanes the moment its pyramid map
        exists (level 0 beside downsample2 -> Rep_n3 -> downsample1 -> Rep_n4, level 1 beside Rep_n4); otherwise the serial
        chain neck(); head()."""
        if not self._dag_on():
            self.neck()
            return self.head(num_classes)
        self._begin_head(num_classes)
        self.neck(_level_ready=self._head_level)
        self._head_join()
        return self.embed, self.boxes

    def features(self, images_u8: torch.Tensor, num_classes: Optional[int] = None):
        self.wait_post()                          # a pipelined step may still be reading the buffers the head is about to write
        self.backbone(images_u8)
        return self.neck_head(num_classes)

    def detect(self, images_u8, text, meta, *, normalize_text: bool, score_thr: float, iou_thr: float = 0.7,
               with_embed: bool = False, nms: Optional[str] = None, nms_param: Optional[int] = None, nms_device: str = "cpu",
               overlap_post: bool = False):
        """The whole step.  ``nms`` None picks the library the reference's path of this text handling uses: normalised
        text = BNContrastiveHead of the mmdet path -> "mmcv"; prompts as stored = the Uni scripts -> "torchvision".

        ``overlap_post`` (a stream of batches; round 4): the post-process of this step — top-k over B x N x K scores and the
        class-aware NMS, 0.75 ms of kernels that occupy 32 waves to a few hundred workgroups of a 256-CU chip — is issued
        on the tower's own second stream behind the similarity GEMM, so that the NEXT call's backbone runs beside it
        instead of behind it.  Same kernels, same buffers, same results; what changes is the contract: the returned
        tensors are produced on ``self.post_stream`` — call ``wait_post()`` (or synchronise the device) before reading them
        from another stream — and they are overwritten by the next call's post-process, as ever.  The next call's head
        waits for this post-process before it overwrites the boxes / embeddings / scores it reads.
        Round 6: in this mode the neck, head and similarity GEMM of the step are issued on the tower's ``nh`` stream (``_pipe_neck_on``),
        so that they run beside the NEXT call's backbone too; ``self.x`` alternates between two c1..c4 sets.  The contract is the
        same — everything a step produces (``embed``, ``boxes``, ``scores`` included) is complete once ``wait_post()`` has been
        honoured; a later in-line call, ``backbone()`` or ``features()`` orders itself behind the pending work by itself."""
        if nms is None:
            nms = "mmcv" if normalize_text else "torchvision"
        if not overlap_post:
            self.wait_post()                      # a pipelined step may still be reading the buffers this one is about to write
            self.features(images_u8, num_classes=text.shape[0])
            scores = self.similarity(text, normalize=normalize_text)
            return self.postprocess(scores, score_thr, meta, iou_thr, with_embed, nms, nms_param, nms_device)
        if self.post_stream is None:
            self.post_stream = device_streams(self.dev, "post")[0]
            self._post_ready = torch.cuda.Event()
        main = torch.cuda.current_stream()
        pipe = self._pipe_neck_on()
        depth = self._bb_depth() if pipe else 1
        bb_stream = main
        if pipe:
            # c1..c4 in depth + 1 sets: this backbone writes the set the neck of step i - depth - 1 read
            if not self._x_sets:
                self._x_sets = [self.x]
                self._nh_stream = device_streams(self.dev, "nh")[0]
                self._bb_done = [torch.cuda.Event(), torch.cuda.Event()]
            while len(self._x_sets) < depth + 1:
                self._x_sets.append([torch.empty_like(t) for t in self.x])
                self._x_free.append(None)
            self._x_par = (self._x_par + 1) % len(self._x_sets)
            self.x = self._x_sets[self._x_par]        # backbone() waits for the neck that last read this set
        self._pipe_issue = pipe
        try:
            self._issue_pipelined_backbone(images_u8, main, depth)
        finally:
            self._pipe_issue = False
        bb_stream = self._slot1["stream"] if (depth == 2 and self._bb_slot) else main
        if pipe:
            self._bb_done[self._bb_slot].record(bb_stream)
        nh = self._nh_stream if pipe else main
        with (torch.cuda.stream(nh) if pipe else contextlib.nullcontext()):
            if pipe:
                nh.wait_event(self._bb_done[self._bb_slot])
                self._nh_issue = True
            try:
                if self._dag_on():
                    self.wait_post()                  # head level 0 starts inside the neck: the previous post-process must be done with
                    self.neck_head(text.shape[0])     # boxes / embeddings before ANY head kernel is issued (it had the whole backbone)
                else:
                    self.neck()
                    self.wait_post()
                    self.head(text.shape[0])
                scores = self.similarity(text, normalize=normalize_text)
            finally:
                self._nh_issue = False
            self._post_ready.record(nh)
            if pipe:
                if self._x_free[self._x_par] is None:
                    self._x_free[self._x_par] = torch.cuda.Event()
                self._x_free[self._x_par].record(nh)
        # the caller's tensors are read by kernels on post_stream (and the nh stream) after this call returns: tell the caching
        # allocator, or a caller that drops `meta` / `text` right away could see the block reused while they still read it
        for t_ in (meta, text):
            if isinstance(t_, torch.Tensor) and t_.is_cuda:
                t_.record_stream(self.post_stream)
                if pipe:
                    t_.record_stream(nh)
        with torch.cuda.stream(self.post_stream):
            self.post_stream.wait_event(self._post_ready)
            res = self.postprocess(scores, score_thr, meta, iou_thr, with_embed, nms, nms_param, nms_device)
            if self._post_done is None:
                self._post_done = torch.cuda.Event()
            self._post_done.record(self.post_stream)
        return res

    def _issue_pipelined_backbone(self, images_u8: torch.Tensor, main, depth: int) -> None:
        if depth == 2:
            # TWO backbones in flight: steps alternate between the caller's stream (slot 0: the tower's own scratch buffers) and
            # the tower's second backbone stream (slot 1: scratch, park workspace and an input staging buffer of its own)
            self._bb_slot ^= 1
            self._depth2_issue = True
            try:
                if self._bb_slot:
                    self._slot1_backbone(images_u8, main)
                else:
                    self.backbone(images_u8)
            finally:
                self._depth2_issue = False
        else:
            self._bb_slot = 0
            self.backbone(images_u8)

    def _bb_depth(self) -> int:
        """Backbones in flight in a stream of batches (``$WEDETECT_BB_DEPTH``: "auto" = 2 from two 640 x 640 images per batch, "1", "2"): 2 = steps alternate between two
        backbone streams, so that the backbone of step i + 1 runs beside the backbone of step i (and the neck / head of step
        i - 1).  Whole-batch launches on both — unlike the image chains, which it replaces in a stream of batches (caller, second
        backbone, nh, post = the four hardware queues; the chains stay for the in-line step): Base B = 8 712 -> 780 images/s,
        B = 16 893 -> 933, B = 4 575 -> 612, Tiny B = 32 1 925 -> 1 987; at Base B = 32 equal to the chains (972.7 / 971.3)."""
        if self.bb_depth == "auto":                  # measured (profiles/r06_pipeline.txt): pays from two 640 x 640 images per batch;
            return 2 if self.B * self.H * self.W >= 2 * 640 * 640 else 1      # a single image is bound by the host's launch rate
        return 2 if self.bb_depth == "2" else 1

    def _slot1_backbone(self, images_u8: torch.Tensor, main) -> torch.cuda.Stream:
        """The backbone of this step on the tower's SECOND backbone stream, over scratch buffers of its own; the batch is
        staged into a tower-owned buffer on the caller's stream first, so that the caller may reuse its tensor at once."""
        if self._slot1 is None:
            z = lambda t: None if t is None else torch.empty_like(t)
            self._slot1 = dict(stream=device_streams(self.dev, "bb2")[0], ready=torch.cuda.Event(), img=torch.empty_like(images_u8),
                               tmp=z(self.tmp), hid=z(self.hid), ln_part=z(self.ln_part), ln_stats=z(self.ln_stats), patches=z(self.patches),
                               park=torch.zeros(max(1, L.p8_workspace_bytes() // 4), dtype=torch.float32, device=self.dev))
        s1 = self._slot1
        main.wait_event(self._bb_done[1])             # the slot's previous backbone (two steps ago) has read the staging buffer
        s1["img"].copy_(images_u8, non_blocking=True)
        s1["ready"].record(main)
        v = copy.copy(self)
        v.tmp, v.hid, v.ln_part, v.ln_stats, v.patches = s1["tmp"], s1["hid"], s1["ln_part"], s1["ln_stats"], s1["patches"]
        v.park = s1["park"] if self.park is not None else None
        v._park_mlp = s1["park"]
        if v.bb_chains != "auto":
            v.bb_chains = "1"                         # forced image chains run on slot 0 only (one set of chain workspaces)
        with torch.cuda.stream(s1["stream"]):
            s1["stream"].wait_event(s1["ready"])
            v.backbone(s1["img"])
        return s1["stream"]

    def _pipe_neck_on(self) -> bool:
        """Neck / head of a pipelined step on the nh stream?  Never inside the calibration pass or stream capture."""
        if self.pipe_neck == "0" or self._calib is not None or torch.cuda.is_current_stream_capturing():
            return False
        return self.pipe_neck == "1" or self.B * self.H * self.W >= self.PIPE_NECK_MIN_PIXELS

    def wait_post(self) -> None:
        """Makes the CURRENT stream wait for the post-process of the last ``detect(overlap_post=True)`` call (no-op otherwise)."""
        if self._post_done is not None and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().wait_event(self._post_done)

    def checked_counts(self, res: Dict[str, torch.Tensor], rerun, recalibrate=None) -> List[int]:
        """Kept-row counts of a step on the host (the one D2H sync a caller needs anyway) with the fp16x3 range guard:
        the fp16x3 GEMMs carry fp32 operands as fp16 (hi, lo) pairs, so an activation beyond 65504 becomes inf.  Every
        fp16x3 launch checks its accumulators in the epilogue and raises the tower's sticky ``range_flag`` on inf / NaN
        (a later ReLU would otherwise turn the NaN into a plausible 0), and the top-k kernel reports non-finite score
        rows as count -1.  When either happens this tower switches to the fp32 MFMA kernels and ``rerun()`` (the caller's closure
        that repeats the step) is executed once more; ``fp16x3_trips`` counts it.  Round 6: after ``FALLBACK_RETRY`` clean fp32
        batches (doubling after every relapse, at most ``FALLBACK_RETRY_MAX``) the tower re-calibrates on the current batch and
        returns to the fp16x3 kernels (``fp16x3_retries``) — rounds 1-5 stayed in fp32 "for good", at 0.4 x the throughput.  Raises if fp32 produces non-finite scores too.  ``recalibrate``: the caller's closure that re-derives the split
        scales from THIS batch (detector._TowerHolder.recalibrate: scales only ever go down, every tower of the checkpoint
        adopts them); it is tried once per trip before the fp32 fallback — a first batch of blank images may have chosen
        scales under which an ordinary image overflows, which is a calibration problem, not a checkpoint that needs fp32."""
        self.wait_post()                         # a pipelined step's counts are produced on post_stream: order the read behind it
        counts = res["count"].tolist()
        if self.overflowed and self.precision == "fp32" and min(counts, default=0) >= 0:
            # round 6: the fallback is no longer for good.  After FALLBACK_RETRY clean fp32 batches (doubling after every further
            # trip, capped) the tower re-derives its split scales from the CURRENT batch (scales only go down) and returns to the
            # fp16x3 kernels for the next step: one odd image in a long-running process no longer costs the rest of it 2.5 x
            self._clean_fp32_steps += 1
            if self._clean_fp32_steps >= self._retry_after and recalibrate is not None:
                self.precision = "fp16x3"
                recalibrate()                        # calibrate(merge=True) through the caller: its fp32 pass restores self.precision
                self.overflowed, self._clean_fp32_steps = False, 0
                self.range_flags.zero_()
                self.fp16x3_retries += 1
            return counts
        flags = self.range_flags.tolist() if self.precision == "fp16x3" else [0, 0]
        if flags[1] and not self.neck_pin:
            # a residual stream left the fp16 range in one of the neck layers that read it directly: pin those five layers to
            # the fp32 kernel (as round 1 did unconditionally) and repeat; everything else stays fp16x3
            import warnings
            warnings.warn("wedetect_amd: a backbone residual stream left the fp16 range in a neck input layer; those layers now "
                          "run the fp32 MFMA kernel")
            self.neck_pin = True
            self.range_flags.zero_()
            return self.checked_counts(rerun(), rerun, recalibrate)
        tripped = flags[0] != 0
        if min(counts, default=0) >= 0 and not tripped:
            return counts
        self.range_flags.zero_()
        if self.precision == "fp16x3":
            self.fp16x3_trips += 1
        if self.precision == "fp16x3" and recalibrate is not None:     # also with all-unit scales: a later batch may need scales < 1
            before = dict(self.sscale)
            recalibrate()
            if self.sscale != before:                       # new scales: one more fp16x3 attempt, fp32 only if that trips too
                import warnings
                warnings.warn("wedetect_amd: an activation left the fp16 range under the split scales of an earlier batch; "
                              "re-calibrated on this batch")
                return self.checked_counts(rerun(), rerun, None)
        if self.precision == "fp32":
            raise L.WedetectHipError("non-finite scores in fp32 mode: the checkpoint or the inputs produce inf / NaN")
        import warnings
        warnings.warn("wedetect_amd: an activation left the fp16 range (|x| >= 65504) in an fp16x3 layer; this tower now runs "
                      "the fp32 MFMA kernels (ImageTower(precision='fp32') avoids the detour)")
        self.precision = "fp32"
        self.overflowed = True
        self._clean_fp32_steps = 0
        self._retry_after = min(self.FALLBACK_RETRY_MAX, self.FALLBACK_RETRY if self.fp16x3_retries == 0 else 2 * self._retry_after)
        counts = rerun()["count"].tolist()
        if min(counts, default=0) < 0:
            raise L.WedetectHipError("non-finite scores in fp32 mode: the checkpoint or the inputs produce inf / NaN")
        return counts


class GraphedDetect:
    """One whole hot-path step captured into a hipGraph (HIP streams and graphs instead of a
    tracing compiler): ~380 kernel launches become one ``hipGraphLaunch``.  Every C-ABI entry
    point is capture-safe (asynchronous, no allocation, no host synchronisation), inputs live
    in static buffers owned by this object.  Worth it when the step is launch-bound, i.e. at
    small batch (the reference runs batch 1 everywhere); at batch 32 the GPU is the bottleneck.

        g = GraphedDetect(tower, num_classes=80, normalize_text=True, score_thr=0.001)
        out = g(images_u8, text, meta)      # copies into the static inputs, replays, returns views
    """

    def __init__(self, tower: ImageTower, num_classes: int, *, normalize_text: bool, score_thr: float,
                 iou_thr: float = 0.7, with_embed: bool = True, warmup: int = 2, nms: Optional[str] = None,
                 nms_param: Optional[int] = None, nms_device: str = "cpu"):
        self.tower = tower
        dev = tower.dev
        self.images = torch.zeros(tower.B, tower.H, tower.W, 3, dtype=torch.uint8, device=dev)
        self.text = torch.zeros(num_classes, EMBED_DIM, dtype=torch.float32, device=dev)
        self.text[:, 0] = 1.0
        self.meta = tower.identity_meta()
        kw = dict(normalize_text=normalize_text, score_thr=score_thr, iou_thr=iou_thr, with_embed=with_embed, nms=nms,
                  nms_param=nms_param, nms_device=nms_device)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # one-time kernel attribute setup + buffer growth
            for _ in range(max(1, warmup)):
                tower.detect(self.images, self.text, self.meta, **kw)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = tower.detect(self.images, self.text, self.meta, **kw)
        # the graph holds raw pointers to tower.scores / topk_ws / text_norm: it is valid for this generation only
        self.generation = tower.generation

    def __call__(self, images_u8: torch.Tensor, text: torch.Tensor, meta: torch.Tensor):
        if self.generation != self.tower.generation:
            raise L.WedetectHipError("stale hipGraph: the tower re-allocated its similarity / top-k buffers (a larger class "
                                     "bank arrived) after this graph was captured; capture a new one")
        self.tower.wait_post()                 # a pipelined eager step of the same tower may still be reading its buffers
        self.images.copy_(images_u8, non_blocking=True)
        self.text.copy_(text, non_blocking=True)
        self.meta.copy_(meta, non_blocking=True)
        self.graph.replay()
        return self.out
