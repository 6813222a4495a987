# round-6 GPU driver script: `gpurun -- bash scripts/r06_gpu.sh <step> [args]`; every step writes under gpurun_out/r06/<step>/
set -u
STEP=${1:-baseline}; shift || true
OUT=gpurun_out/r06/$STEP; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
Q="--steps 30 --warmup 8 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs"
line() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], 'dom', r['frac'], r.get('avg_launch_us'), 'sim', d.get('sim_gemm',{}).get('frac'))
    except Exception as e:
        print(f, 'FAILED', e)
PY
}
case $STEP in
dwfin)
  # statistics finalize inside the depthwise kernel: identity tests, step A/B at B = 32 and on single images (the code behind $WEDETECT_DW_FIN was
  # measured 1.8 x slower and removed: profiles/r06_dwfin.txt; the step is kept as the record of what was run)
  ( timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_network.py -q -m gpu -k "finalize_inside or layernorm or fold or image_chains or pipelined or goldens or batch_invariant" -x 2>&1 | tail -4 ) > $OUT/tests.log
  for i in 1 2 3; do
    for f in 0 1; do
      WEDETECT_DW_FIN=$f python bench.py $Q > $OUT/bench_b32_fin${f}_$i.json 2> $OUT/bench_b32_fin${f}_$i.err
      WEDETECT_DW_FIN=$f python bench.py $Q --batch 1 --no-overlap-post > $OUT/bench_b1inline_fin${f}_$i.json 2> $OUT/bench_b1inline_fin${f}_$i.err
      WEDETECT_DW_FIN=$f python bench.py $Q --batch 8 > $OUT/bench_b8_fin${f}_$i.json 2> $OUT/bench_b8_fin${f}_$i.err
    done
  done
  cat $OUT/tests.log; line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
forms2)
  for i in 1 2; do
    python bench.py $Q > $OUT/bench_default_$i.json 2> $OUT/bench_default_$i.err
    WEDETECT_CONV3_WS=0 python bench.py $Q > $OUT/bench_conv3ws0_$i.json 2> $OUT/bench_conv3ws0_$i.err
    WEDETECT_FUSE_MLP_WIDE=256,512 python bench.py $Q > $OUT/bench_wide512_$i.json 2> $OUT/bench_wide512_$i.err
    WEDETECT_DWCONV_DMA=0 python bench.py $Q > $OUT/bench_dwdma0_$i.json 2> $OUT/bench_dwdma0_$i.err
    WEDETECT_FIXED_SPLITK=0 python bench.py $Q > $OUT/bench_fsk0_$i.json 2> $OUT/bench_fsk0_$i.err
  done
  line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
persist)
  # which persistent forms to keep while a pipelined backbone is issued: none (0) / the 256 x 256 kernel's (p8) / the wide MLP's (mlp) / both (1)
  ( timeout 1200 python -m pytest tests/test_gpu_network.py -q -m gpu -k "pipelined or image_chains" -x 2>&1 | tail -4 ) > $OUT/tests.log
  for i in 1 2; do
    for m in 0 p8 mlp 1; do
      WEDETECT_PIPE_PERSIST=$m python bench.py $Q > $OUT/bench_b32_pp${m}_$i.json 2> $OUT/bench_b32_pp${m}_$i.err
      WEDETECT_PIPE_PERSIST=$m python bench.py $Q --batch 64 > $OUT/bench_b64_pp${m}_$i.json 2> $OUT/bench_b64_pp${m}_$i.err
      WEDETECT_PIPE_PERSIST=$m python bench.py $Q --batch 16 > $OUT/bench_b16_pp${m}_$i.json 2> $OUT/bench_b16_pp${m}_$i.err
    done
  done
  cat $OUT/tests.log; line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
p8tile)
  # persistent (stream-K gangs) against tile form of the 256 x 256 kernel under the stream pipeline, in line, other towers / batches
  for i in 1 2; do
    for m in persist tile; do
      WEDETECT_P8=$m python bench.py $Q > $OUT/bench_b32_${m}_$i.json 2> $OUT/bench_b32_${m}_$i.err
      WEDETECT_P8=$m python bench.py $Q --no-overlap-post > $OUT/bench_inline_${m}_$i.json 2> $OUT/bench_inline_${m}_$i.err
      WEDETECT_P8=$m python bench.py $Q --batch 16 > $OUT/bench_b16_${m}_$i.json 2> $OUT/bench_b16_${m}_$i.err
      WEDETECT_P8=$m python bench.py $Q --batch 64 > $OUT/bench_b64_${m}_$i.json 2> $OUT/bench_b64_${m}_$i.err
      WEDETECT_P8=$m python bench.py $Q --arch large --batch 16 --classes 1203 > $OUT/bench_large16_${m}_$i.json 2> $OUT/bench_large16_${m}_$i.err
      WEDETECT_P8=$m python bench.py $Q --arch tiny > $OUT/bench_tiny32_${m}_$i.json 2> $OUT/bench_tiny32_${m}_$i.err
    done
  done
  line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
wide512)
  # the 512-channel one-kernel block MLP as default?  in-line steps (no pipeline), other batches
  for i in 1 2; do
    for w in "256" "256,512"; do
      n=${w//,/_}
      WEDETECT_FUSE_MLP_WIDE=$w python bench.py $Q --no-overlap-post > $OUT/bench_inline_w${n}_$i.json 2> $OUT/bench_inline_w${n}_$i.err
      WEDETECT_FUSE_MLP_WIDE=$w python bench.py $Q --batch 16 > $OUT/bench_b16_w${n}_$i.json 2> $OUT/bench_b16_w${n}_$i.err
      WEDETECT_FUSE_MLP_WIDE=$w python bench.py $Q --batch 64 > $OUT/bench_b64_w${n}_$i.json 2> $OUT/bench_b64_w${n}_$i.err
      WEDETECT_FUSE_MLP_WIDE=$w python bench.py $Q --batch 24 > $OUT/bench_b24_w${n}_$i.json 2> $OUT/bench_b24_w${n}_$i.err
    done
  done
  line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
energy2)
  for i in 1 2 3; do
    python bench.py $Q > $OUT/bench_default_$i.json 2> $OUT/bench_default_$i.err
    WEDETECT_FUSE_MLP_WIDE=256,512 python bench.py $Q > $OUT/bench_wide512_$i.json 2> $OUT/bench_wide512_$i.err
    WEDETECT_FUSE_MLP_WIDE=256,512 WEDETECT_P8=tile python bench.py $Q > $OUT/bench_wide512_p8tile_$i.json 2> $OUT/bench_wide512_p8tile_$i.err
    WEDETECT_FUSE_MLP_WIDE=256,512 WEDETECT_LN_FOLD=0 python bench.py $Q > $OUT/bench_wide512_nofold_$i.json 2> $OUT/bench_wide512_nofold_$i.err
    WEDETECT_FUSE_MLP_WIDE= python bench.py $Q > $OUT/bench_nowide_$i.json 2> $OUT/bench_nowide_$i.err
  done
  line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
energy)
  # under the power cap a form that moves fewer bytes may win although it is slower alone: the 512-channel one-kernel block MLP
  # (hidden tensor never leaves the CU), the similarity on the fp16x3 kernel at K = 80, the LayerNorm kernel instead of the fold
  for i in 1 2 3; do
    python bench.py $Q > $OUT/bench_default_$i.json 2> $OUT/bench_default_$i.err
    WEDETECT_FUSE_MLP_WIDE=256,512 python bench.py $Q > $OUT/bench_wide512_$i.json 2> $OUT/bench_wide512_$i.err
    WEDETECT_SIM_SPLIT=1 python bench.py $Q > $OUT/bench_simsplit_$i.json 2> $OUT/bench_simsplit_$i.err
    WEDETECT_P8=tile python bench.py $Q > $OUT/bench_p8tile_$i.json 2> $OUT/bench_p8tile_$i.err
  done
  line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
depth32)
  for i in 1 2 3 4; do
    WEDETECT_BB_DEPTH=1 WEDETECT_BB_CHAINS=2 python bench.py $Q > $OUT/bench_d1c2_$i.json 2> $OUT/bench_d1c2_$i.err
    WEDETECT_BB_DEPTH=2 WEDETECT_BB_CHAINS=1 python bench.py $Q > $OUT/bench_d2c1_$i.json 2> $OUT/bench_d2c1_$i.err
  done
  line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
depthcfg)
  for i in 1 2; do
    for cfg in "base64:--batch 64" "base2:--batch 2" "large16:--arch large --batch 16 --classes 1203" "tiny32:--arch tiny" "large1280:--arch large --size 1280 --batch 4 --classes 1203"; do
      name=${cfg%%:*}; fl=${cfg#*:}
      WEDETECT_BB_DEPTH=1 python bench.py $Q $fl > $OUT/bench_${name}_d1_$i.json 2> $OUT/bench_${name}_d1_$i.err
      WEDETECT_BB_DEPTH=2 WEDETECT_BB_CHAINS=1 python bench.py $Q $fl > $OUT/bench_${name}_d2c1_$i.json 2> $OUT/bench_${name}_d2c1_$i.err
    done
  done
  line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
depth)
  # two backbones in flight (steps alternate between two backbone streams): identity tests, sweep over the batch
  ( timeout 1200 python -m pytest tests/test_gpu_network.py -q -m gpu -k "pipelined" -x 2>&1 | tail -8 ) > $OUT/tests.log
  for i in 1 2; do
    for b in ${BATCHES:-1 4 8 16 32}; do
      for dp in 1 2; do
        WEDETECT_BB_DEPTH=$dp python bench.py $Q --batch $b > $OUT/bench_b${b}_d${dp}_$i.json 2> $OUT/bench_b${b}_d${dp}_$i.err
      done
    done
    WEDETECT_BB_DEPTH=2 WEDETECT_BB_CHAINS=1 python bench.py $Q > $OUT/bench_b32_d2c1_$i.json 2> $OUT/bench_b32_d2c1_$i.err
  done
  cat $OUT/tests.log; line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
chainstages)
  ( timeout 1200 python -m pytest tests/test_gpu_network.py -q -m gpu -k "image_chains or pipelined" -x 2>&1 | tail -8 ) > $OUT/tests.log
  for i in 1 2; do
    for st in 0-3 2-3 2-2 0-1 1-3; do
      WEDETECT_BB_CHAIN_STAGES=$st python bench.py $Q > $OUT/bench_st${st}_$i.json 2> $OUT/bench_st${st}_$i.err
    done
  done
  cat $OUT/tests.log; line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
pipesmalldag)
  for i in 1 2; do
    for b in 1 4 8 16; do
      WEDETECT_PIPE_NECK=1 python bench.py $Q --batch $b > $OUT/bench_b${b}_p1_dagoff_$i.json 2> $OUT/bench_b${b}_p1_dagoff_$i.err
      WEDETECT_DAG=1 WEDETECT_PIPE_NECK=1 python bench.py $Q --batch $b > $OUT/bench_b${b}_p1_dag1_$i.json 2> $OUT/bench_b${b}_p1_dag1_$i.err
    done
  done
  line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
pipesmall)
  # the pipelined neck at small batches (latency split-K classes), tests first
  ( timeout 1200 python -m pytest tests/test_gpu_network.py -q -m gpu -k "image_chains or pipelined or latency or mid_class" -x 2>&1 | tail -8 ) > $OUT/tests.log
  for i in 1 2; do
    for b in 1 2 4 8; do
      for pn in 0 1; do
        WEDETECT_PIPE_NECK=$pn python bench.py $Q --batch $b > $OUT/bench_b${b}_p${pn}_$i.json 2> $OUT/bench_b${b}_p${pn}_$i.err
      done
    done
  done
  cat $OUT/tests.log; line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
pipecfg)
  # where do image chains / the pipelined neck pay?  other towers and batches, same-box alternating
  for i in 1 2; do
    for cfg in "tiny32:--arch tiny" "base64:--batch 64" "base16:--batch 16" "large16:--arch large --batch 16 --classes 1203" "uni32:--mode uni --classes 256"; do
      name=${cfg%%:*}; fl=${cfg#*:}
      for m in "1 0" "2 0" "1 1" "2 1"; do
        set -- $m
        WEDETECT_BB_CHAINS=$1 WEDETECT_PIPE_NECK=$2 python bench.py $Q $fl > $OUT/bench_${name}_c$1p$2_$i.json 2> $OUT/bench_${name}_c$1p$2_$i.err
      done
    done
  done
  line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
pipedag)
  # with the neck / head pipelined beside the next backbone, is the DAG (three more streams on four hardware queues) still worth it?
  for i in 1 2; do
    for d in auto 0; do
      for m in "1 1" "2 1" "2 0"; do
        set -- $m
        WEDETECT_DAG=$d WEDETECT_BB_CHAINS=$1 WEDETECT_PIPE_NECK=$2 python bench.py $Q > $OUT/bench_dag${d}_c$1p$2_$i.json 2> $OUT/bench_dag${d}_c$1p$2_$i.err
      done
    done
  done
  line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
hwq)
  # do the streams of a step (caller, post, three DAG lanes, image chain, nh) collide on the runtime's 4 hardware queues?
  for i in 1 2; do
    for q in ${HWQS:-4 8}; do
      for m in "1 0" "2 0" "1 1" "2 1"; do
        set -- $m
        GPU_MAX_HW_QUEUES=$q WEDETECT_BB_CHAINS=$1 WEDETECT_PIPE_NECK=$2 python bench.py $Q > $OUT/bench_q${q}_c$1p$2_$i.json 2> $OUT/bench_q${q}_c$1p$2_$i.err
      done
    done
  done
  line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
pipe)
  # neck / head of step i on the nh stream beside the backbone of step i + 1: identity tests, same-box alternating A/B
  ( timeout 1200 python -m pytest tests/test_gpu_network.py -q -m gpu -k "image_chains or pipelined" -x 2>&1 | tail -8 ) > $OUT/tests.log
  for i in 1 2 3; do
    WEDETECT_BB_CHAINS=1 WEDETECT_PIPE_NECK=0 python bench.py $Q > $OUT/bench_c1p0_$i.json 2> $OUT/bench_c1p0_$i.err
    WEDETECT_BB_CHAINS=2 WEDETECT_PIPE_NECK=0 python bench.py $Q > $OUT/bench_c2p0_$i.json 2> $OUT/bench_c2p0_$i.err
    WEDETECT_BB_CHAINS=1 WEDETECT_PIPE_NECK=1 python bench.py $Q > $OUT/bench_c1p1_$i.json 2> $OUT/bench_c1p1_$i.err
    WEDETECT_BB_CHAINS=2 WEDETECT_PIPE_NECK=1 python bench.py $Q > $OUT/bench_c2p1_$i.json 2> $OUT/bench_c2p1_$i.err
  done
  cat $OUT/tests.log; line $OUT/bench_*.json; cat $OUT/*.err | grep -v amdgpu.ids | tail -n 20
  ;;
chains)
  # backbone as independent image chains on side streams: identity test, then same-box alternating A/B of the step
  ( timeout 1200 python -m pytest tests/test_gpu_network.py -q -m gpu -k "image_chains" -x 2>&1 | tail -8 ) > $OUT/tests.log
  for i in 1 2 3; do
    WEDETECT_BB_CHAINS=1 python bench.py $Q > $OUT/bench_chains1_$i.json 2> $OUT/bench_chains1_$i.err
    for o in free dw gemm both; do
      WEDETECT_BB_CHAINS=2 WEDETECT_BB_CHAIN_ORDER=$o python bench.py $Q > $OUT/bench_chains2_${o}_$i.json 2> $OUT/bench_chains2_${o}_$i.err
    done
  done
  cat $OUT/tests.log; line $OUT/bench_*.json; tail -n 3 $OUT/*.err | tail -n 30
  ;;
baseline)
  python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  WEDETECT_DAG=0 python bench.py $Q > $OUT/bench_dag0.json 2> $OUT/bench_dag0.err
  OUT=$OUT/layers.json python scripts/neck_layer_times.py > $OUT/layers.txt 2>&1
  line $OUT/bench_default.json $OUT/bench_dag0.json
  grep -v amdgpu.ids $OUT/layers.txt
  ;;
dwpmc)
  # what the depthwise 7x7 kernels ARE bound by: SQ / TCC counters over scripts/dwconv_bench.py
  export TMPDIR=/tmp
  python scripts/dwconv_bench.py > $OUT/dwconv_bench.txt 2>&1
  python scripts/hbm_bw_probe.py > $OUT/hbm_probe.txt 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o p -- python scripts/dwconv_bench.py > /dev/null 2> $OUT/trace.err
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/pmc1 -o p -- python scripts/dwconv_bench.py > /dev/null 2> $OUT/pmc1.err
  timeout 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU -d $OUT/pmc2 -o p -- python scripts/dwconv_bench.py > /dev/null 2> $OUT/pmc2.err
  timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $OUT/pmc3 -o p -- python scripts/dwconv_bench.py > /dev/null 2> $OUT/pmc3.err
  python scripts/rocpd_summary.py $(ls $OUT/trace/*.db | head -1) --pmc $(ls $OUT/pmc1/*.db | head -1) --pmc $(ls $OUT/pmc2/*.db | head -1) --pmc $(ls $OUT/pmc3/*.db | head -1) > $OUT/summary.txt 2>&1
  rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
  grep -v amdgpu.ids $OUT/dwconv_bench.txt; tail -5 $OUT/hbm_probe.txt; grep -i "dwconv\|kernel  \|## PMC" $OUT/summary.txt | cut -c1-330; tail -3 $OUT/pmc*.err
  ;;
dwdma)
  # the LDS-DMA staged depthwise kernel: identity tests, isolated A/B (channel blocks per workgroup 1 / 2 / 4), step A/B
  ( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_split.py -q -m gpu -k "dwconv or stats or fold or layernorm" -x 2>&1 | tail -8 ) > $OUT/tests.log
  for n in 1 2 4; do echo "== WD_DWCONV_NCB=$n"; WD_DWCONV_NCB=$n python scripts/dwconv_bench.py; done > $OUT/dwconv_bench.txt 2>&1
  for i in 1 2; do
    WEDETECT_DWCONV_DMA=0 python bench.py $Q > $OUT/bench_dma0_$i.json 2> $OUT/bench_dma0_$i.err
    python bench.py $Q > $OUT/bench_dma1_$i.json 2> $OUT/bench_dma1_$i.err
  done
  cat $OUT/tests.log; grep -v amdgpu.ids $OUT/dwconv_bench.txt; line $OUT/bench_dma*.json
  ;;
dwdma2)
  export TMPDIR=/tmp
  ( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_split.py -q -m gpu -k "dwconv or stats or fold" -x 2>&1 | tail -5 ) > $OUT/tests.log
  python scripts/dwconv_bench.py > $OUT/dwconv_bench.txt 2>&1
  VARIANTS=4 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/pmc1 -o p -- python scripts/dwconv_bench.py > /dev/null 2> $OUT/pmc1.err
  VARIANTS=4 timeout 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU -d $OUT/pmc2 -o p -- python scripts/dwconv_bench.py > /dev/null 2> $OUT/pmc2.err
  VARIANTS=4 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o p -- python scripts/dwconv_bench.py > /dev/null 2> $OUT/trace.err
  python scripts/rocpd_summary.py $(ls $OUT/trace/*.db | head -1) --pmc $(ls $OUT/pmc1/*.db | head -1) --pmc $(ls $OUT/pmc2/*.db | head -1) > $OUT/summary.txt 2>&1
  rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2
  for i in 1 2; do
    WEDETECT_DWCONV_DMA=0 python bench.py $Q > $OUT/bench_dma0_$i.json 2> $OUT/bench_dma0_$i.err
    python bench.py $Q > $OUT/bench_dma1_$i.json 2> $OUT/bench_dma1_$i.err
  done
  cat $OUT/tests.log; grep -v amdgpu.ids $OUT/dwconv_bench.txt; grep -i "dwconv\|kernel  \|## PMC" $OUT/summary.txt | cut -c1-330; line $OUT/bench_dma*.json
  ;;
dwln)
  ( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_split.py -q -m gpu -k "dwconv or stats or fold" -x 2>&1 | tail -5 ) > $OUT/tests.log
  { echo "== WEDETECT_DWCONV_DMA=0"; WEDETECT_DWCONV_DMA=0 python scripts/dwln_bench.py; echo "== WEDETECT_DWCONV_DMA=1"; python scripts/dwln_bench.py; } > $OUT/dwln_bench.txt 2>&1
  for i in 1 2; do
    WEDETECT_DWCONV_DMA=0 python bench.py $Q > $OUT/bench_dma0_$i.json 2> $OUT/bench_dma0_$i.err
    python bench.py $Q > $OUT/bench_dma1_$i.json 2> $OUT/bench_dma1_$i.err
  done
  cat $OUT/tests.log; grep -v amdgpu.ids $OUT/dwln_bench.txt; line $OUT/bench_dma*.json
  ;;
bstride)
  ( timeout 1200 python -m pytest tests/test_gpu_split.py -q -m gpu -k "batch_stride or p8_kernel or special" -x 2>&1 | tail -5 ) > $OUT/tests.log
  ( timeout 1200 python -m pytest tests/test_gpu_network.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -5 ) > $OUT/tests_net.log
  OUT=$OUT/layers.json python scripts/neck_layer_times.py > $OUT/layers.txt 2>&1
  for i in 1 2; do
    WEDETECT_P8_BSTRIDE=0 python bench.py $Q > $OUT/bench_off_$i.json 2> $OUT/bench_off_$i.err
    python bench.py $Q > $OUT/bench_on_$i.json 2> $OUT/bench_on_$i.err
  done
  cat $OUT/tests.log $OUT/tests_net.log; grep "embed\|total" $OUT/layers.txt; line $OUT/bench_o*.json
  ;;
simsplit)
  ( timeout 1200 python -m pytest tests/test_gpu_split.py -q -m gpu -k "similarity or special or conv_pp or p8_kernel" -x 2>&1 | tail -8 ) > $OUT/tests.log
  ( timeout 1500 python -m pytest tests/test_gpu_network.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -8 ) > $OUT/tests_net.log
  for i in 1 2; do
    WEDETECT_SIM_SPLIT=0 python bench.py $Q --arch large --batch 16 --classes 1203 > $OUT/large_off_$i.json 2> $OUT/large_off_$i.err
    python bench.py $Q --arch large --batch 16 --classes 1203 > $OUT/large_on_$i.json 2> $OUT/large_on_$i.err
    WEDETECT_SIM_SPLIT=0 python bench.py $Q --mode uni --classes 256 > $OUT/uni_off_$i.json 2> $OUT/uni_off_$i.err
    python bench.py $Q --mode uni --classes 256 > $OUT/uni_on_$i.json 2> $OUT/uni_on_$i.err
  done
  cat $OUT/tests.log $OUT/tests_net.log; line $OUT/large_o*.json $OUT/uni_o*.json
  python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06/simsplit/*_on_1.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], json.dumps(d.get('sim_gemm')))
PY
  ;;
fullsuite)
  ( timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > $OUT/tests_full.log
  cp gpurun_out/parity_r05.jsonl $OUT/parity.jsonl 2>/dev/null
  tail -25 $OUT/tests_full.log
  ;;
smallb)
  export TMPDIR=/tmp
  BATCH=1 ALL=1 python scripts/neck_layer_times.py > $OUT/layers_b1.txt 2>&1
  BATCH=8 ALL=1 python scripts/neck_layer_times.py > $OUT/layers_b8.txt 2>&1
  python bench.py $Q --batch 1 > $OUT/bench_b1.json 2> $OUT/bench_b1.err
  python bench.py $Q --batch 8 > $OUT/bench_b8.json 2> $OUT/bench_b8.err
  WEDETECT_SPLIT_K=1 python bench.py $Q --batch 1 > $OUT/bench_b1_splitk.json 2> $OUT/bench_b1_splitk.err
  WEDETECT_SPLIT_K=1 python bench.py $Q --batch 8 > $OUT/bench_b8_splitk.json 2> $OUT/bench_b8_splitk.err
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o p -- python bench.py --steps 20 --warmup 5 --batch 1 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs > /dev/null 2> $OUT/trace.err
  python scripts/rocpd_summary.py $(ls $OUT/trace/*.db | head -1) > $OUT/summary_b1.txt 2>&1
  rm -rf $OUT/trace
  line $OUT/bench_b*.json; head -40 $OUT/summary_b1.txt | cut -c1-120; grep "total" $OUT/layers_b1.txt $OUT/layers_b8.txt
  ;;
latency)
  ( timeout 1200 python -m pytest tests/test_gpu_network.py -q -m gpu -k "latency or split_k" -x 2>&1 | tail -8 ) > $OUT/tests.log
  for b in 1 2 4 8; do
    WEDETECT_SPLIT_K=0 python bench.py $Q --batch $b > $OUT/bench_b${b}_off.json 2> $OUT/bench_b${b}_off.err
    python bench.py $Q --batch $b > $OUT/bench_b${b}_auto.json 2> $OUT/bench_b${b}_auto.err
  done
  WEDETECT_SPLIT_K=0 python bench.py $Q --arch tiny --batch 1 > $OUT/bench_tiny_b1_off.json 2>/dev/null
  python bench.py $Q --arch tiny --batch 1 > $OUT/bench_tiny_b1_auto.json 2>/dev/null
  cat $OUT/tests.log; line $OUT/bench_*.json
  ;;
foldfused)
  ( timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_network.py tests/test_gpu_precision.py -q -m gpu -k "fold or fused or precision or calibrat" -x 2>&1 | tail -8 ) > $OUT/tests.log
  for i in 1 2; do
    WEDETECT_LN_FOLD_FUSED=0 python bench.py $Q > $OUT/bench_off_$i.json 2> $OUT/bench_off_$i.err
    python bench.py $Q > $OUT/bench_on_$i.json 2> $OUT/bench_on_$i.err
  done
  cat $OUT/tests.log; line $OUT/bench_o*.json
  python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06/foldfused/bench_o*_1.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], {k:v for k,v in d['gemm_kernels'].items() if 'fused' in k})
PY
  ;;
fsplitk)
  for i in 1 2; do
    WEDETECT_FIXED_SPLITK=r5 python bench.py $Q > $OUT/bench_r5_$i.json 2> $OUT/bench_r5_$i.err
    python bench.py $Q > $OUT/bench_r6_$i.json 2> $OUT/bench_r6_$i.err
  done
  WEDETECT_FIXED_SPLITK=r5 python bench.py $Q --arch large --batch 16 --classes 1203 > $OUT/large_r5.json 2> $OUT/large_r5.err
  python bench.py $Q --arch large --batch 16 --classes 1203 > $OUT/large_r6.json 2> $OUT/large_r6.err
  OUT=$OUT/layers.json python scripts/neck_layer_times.py > $OUT/layers.txt 2>&1
  ( timeout 1500 python -m pytest tests/test_gpu_network.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -6 ) > $OUT/tests.log
  line $OUT/bench_r*.json $OUT/large_r*.json; grep "downsample1\|head2.reg0\|total" $OUT/layers.txt; cat $OUT/tests.log
  ;;
ceiling)
  WEDETECT_LIB=$GRAFT_REPO_ROOT/wedetect_amd/libwedetect_hip_abl.so python scripts/p8_ceiling.py > $OUT/p8_ceiling.jsonl 2> $OUT/p8_ceiling.err
  cat $OUT/p8_ceiling.jsonl; tail -3 $OUT/p8_ceiling.err
  ;;
dryrun)
  # the N > 1 code of bench.py with 8 / 2 gloo ranks time-sliced on the one GPU of this pool (not a scaling measurement)
  WEDETECT_BENCH_SHARE_GPU=1 WEDETECT_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs > $OUT/bench_n8_dry.json 2> $OUT/bench_n8_dry.err
  WEDETECT_BENCH_SHARE_GPU=1 WEDETECT_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-reference --no-host-fed --no-other-configs > $OUT/bench_n2_dry.json 2> $OUT/bench_n2_dry.err
  WEDETECT_BENCH_SHARE_GPU=1 WEDETECT_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --mode retrieval --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_n2_retrieval_dry.json 2> $OUT/bench_n2_retrieval_dry.err
  for f in $OUT/bench_n*_dry.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['n_gpus'], d.get('ranks_seen'), d.get('collective_backend'), d['value'], d['unit'])" || tail -n 5 ${f%.json}.err; done
  ;;
midclass)
  # mid-class latency split-K (5 - 8 images of 640 x 640): class-invariance test, B = 5 .. 8 with / without, B = 1 / 4 unchanged
  ( timeout 1200 python -m pytest tests/test_gpu_network.py -q -m gpu -k "latency or split_k or mid_class" -x 2>&1 | tail -6 ) > $OUT/tests.log
  for b in 5 6 8; do
    WEDETECT_SPLIT_K=0 python bench.py $Q --batch $b > $OUT/bench_b${b}_off.json 2> $OUT/bench_b${b}_off.err
    python bench.py $Q --batch $b > $OUT/bench_b${b}_auto.json 2> $OUT/bench_b${b}_auto.err
  done
  for b in 1 4 16; do python bench.py $Q --batch $b > $OUT/bench_b${b}_auto.json 2> $OUT/bench_b${b}_auto.err; done
  cat $OUT/tests.log; line $OUT/bench_*.json
  ;;
c3pmc)
  # what the 3 x 3 K loop is bound by: SQ / TCC / TCP counters per form (kernel names carry the form), one layer per run
  export TMPDIR=/tmp
  for sh in 0 1; do
    export SHAPE=$sh
    timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace$sh -o p -- python scripts/conv3_pmc.py > /dev/null 2> $OUT/trace$sh.err
    timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc1_$sh -o p -- python scripts/conv3_pmc.py > /dev/null 2> $OUT/pmc1_$sh.err
    timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $OUT/pmc2_$sh -o p -- python scripts/conv3_pmc.py > /dev/null 2> $OUT/pmc2_$sh.err
    timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum -d $OUT/pmc3_$sh -o p -- python scripts/conv3_pmc.py > /dev/null 2> $OUT/pmc3_$sh.err
    # (a --pmc TA_* pass aborted inside a torch kernel and hung the profiler for 18 minutes: not collected)
    P=""; for d in $OUT/pmc1_$sh $OUT/pmc2_$sh $OUT/pmc3_$sh; do f=$(ls $d/*.db 2>/dev/null | head -1); [ -n "$f" ] && P="$P --pmc $f"; done
    python scripts/rocpd_summary.py $(ls $OUT/trace$sh/*.db | head -1) $P > $OUT/summary$sh.txt 2>&1
    rm -rf $OUT/trace$sh $OUT/pmc?_$sh
    grep -i "conv3\|kernel  \|## PMC" $OUT/summary$sh.txt | cut -c1-330; tail -n 2 $OUT/pmc3_$sh.err
  done
  ;;
ws12)
  # the twelve-wave producer / consumer 3 x 3 kernel (cfg 79) against the eight-wave forms
  ( timeout 1200 python -m pytest tests/test_gpu_split.py -q -m gpu -k "conv3 or conv_pp" -x 2>&1 | tail -8 ) > $OUT/tests.log
  python scripts/conv3_bench.py > $OUT/conv3_bench.txt 2>&1
  WEDETECT_CONV3_WS=0 python scripts/neck_layer_times.py > $OUT/layers_off.txt 2>&1
  python scripts/neck_layer_times.py > $OUT/layers_on.txt 2>&1
  for i in 1 2; do
    WEDETECT_CONV3_WS=0 python bench.py $Q > $OUT/bench_off_$i.json 2> $OUT/bench_off_$i.err
    python bench.py $Q > $OUT/bench_on_$i.json 2> $OUT/bench_on_$i.err
  done
  WEDETECT_CONV3_WS=0 WEDETECT_DAG=0 python bench.py $Q > $OUT/bench_dag0_off.json 2> $OUT/bench_dag0_off.err
  WEDETECT_DAG=0 python bench.py $Q > $OUT/bench_dag0_on.json 2> $OUT/bench_dag0_on.err
  cat $OUT/tests.log; grep -v amdgpu.ids $OUT/conv3_bench.txt | cut -c1-400; grep total $OUT/layers_off.txt $OUT/layers_on.txt; line $OUT/bench_o*.json $OUT/bench_dag0_o*.json
  ;;
stag)
  # the staggered K loop of the 3 x 3 kernel (cfg 77) against the in-step loop (cfg 78): identity tests, isolated A/B, per-layer serial
  # chain, step A/B (DAG on and off)
  ( timeout 1200 python -m pytest tests/test_gpu_split.py -q -m gpu -k "conv3 or conv_pp" -x 2>&1 | tail -8 ) > $OUT/tests.log
  python scripts/conv3_bench.py > $OUT/conv3_bench.txt 2>&1
  WEDETECT_CONV3_STAG=0 python scripts/neck_layer_times.py > $OUT/layers_off.txt 2>&1
  python scripts/neck_layer_times.py > $OUT/layers_on.txt 2>&1
  for i in 1 2; do
    WEDETECT_CONV3_STAG=0 python bench.py $Q > $OUT/bench_off_$i.json 2> $OUT/bench_off_$i.err
    python bench.py $Q > $OUT/bench_on_$i.json 2> $OUT/bench_on_$i.err
  done
  WEDETECT_CONV3_STAG=0 WEDETECT_DAG=0 python bench.py $Q > $OUT/bench_dag0_off.json 2> $OUT/bench_dag0_off.err
  WEDETECT_DAG=0 python bench.py $Q > $OUT/bench_dag0_on.json 2> $OUT/bench_dag0_on.err
  cat $OUT/tests.log; grep -v amdgpu.ids $OUT/conv3_bench.txt | cut -c1-260; grep total $OUT/layers_off.txt $OUT/layers_on.txt; line $OUT/bench_o*.json $OUT/bench_dag0_o*.json
  ;;
*) echo "unknown step $STEP"; exit 2;;
esac
