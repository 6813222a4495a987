"""TEST INFRASTRUCTURE — CPU restatement of the WeDetect hot path (the parity oracle).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package; nothing under ``wedetect_amd/`` does.  Every function cites
the reference file:line it follows.  Pinning status:

  * network (backbone / neck / head / contrast / DFL / decode / filter_scores_and_topk):
    pinned — bit-identical to an import of the reference on shared synthetic weights
    (tests/golden/make_golden.py, run in the build container; fixtures committed).
  * NMS: torchvision.ops.batched_nms / mmcv.ops.batched_nms are third-party native code
    absent from /root/reference and from this image (**parity unpinned**: no binary to
    run against).  oracle/postprocess.py restates BOTH published algorithms branch for
    branch (fp32 coordinate offsets, class-agnostic pass, the candidate-count branches,
    double vs float threshold) with the defined total order (score desc, candidate index
    asc), pinned to hand-derived vectors (tests/test_cpu.py).
"""
