#!/bin/bash
# rocprofv3 passes of one round on the GPU box (run from the repo root through gpurun); summaries land in gpurun_out/$1/ and are
# copied into profiles/<round>/ by hand.  Counters (--pmc) are collected in their own runs with --kernel-trace only, FETCH_SIZE and
# WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots).
set -uo pipefail
OUT=gpurun_out/${1:-prof}
mkdir -p $OUT
export TMPDIR=/tmp
# 1. the bench command, bounded, hipGraphs ON
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 bench.py --gpus 1 --steps 1 --warmup 0 --ccdm-steps 20 --max-slices 6 --no-cpu-baseline --no-extra > $OUT/bench_under_rocprof.log 2>&1
echo "bench pass rc=$?"
# 2. one eager latent-UNet forward, per-kernel timeline
GG_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -d $OUT/ldm -o ldm -- python3 tools/perf_probe.py ldm > $OUT/ldm_probe.log 2>&1
echo "ldm pass rc=$?"
python3 tools/ldm_timeline.py $(find $OUT/ldm -name "ldm_results.db" | head -1) --all > $OUT/ldm_unet_forward_timeline_eager.txt 2>&1
# 3. CCDM forward @128^3: kernel stats, FETCH / WRITE passes, one SQ pass
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ccdm -o ccdm -- python3 tools/perf_probe.py ccdm128 > $OUT/ccdm_probe.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 tools/perf_probe.py ccdm128 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 tools/perf_probe.py ccdm128 > /dev/null 2>&1
python3 tools/pmc_parse.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_conv3d.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq_ccdm -- python3 tools/perf_probe.py ccdm128 > /dev/null 2>&1
{ for k in "conv_halo_kernel<1, 2, 0, 2, 0" "conv_halo_kernel<1, 4, 0, 0, 0" "conv_halo_kernel<1, 4, 1, 0, 0" "conv_halo_kernel<1, 1, 0, 1, 0"; do echo "## $k> in python3 tools/perf_probe.py ccdm128"; python3 tools/pmc_sq.py "$k" $OUT/pmc_sq_ccdm; done; } > $OUT/pmc_sq_summary.txt
echo "ccdm passes done"
# 4. AE decode + cond-encode @512^2: kernel stats, FETCH / WRITE / SQ for the 2-D halo conv and the single-head attention
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ae -o ae -- python3 tools/perf_probe.py ae > $OUT/ae_probe.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_ae -- python3 tools/perf_probe.py ae > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_ae -- python3 tools/perf_probe.py ae > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq_ae -- python3 tools/perf_probe.py ae > /dev/null 2>&1
for k in "conv_halo_kernel<0, 4, 0, 0" "conv_halo_kernel<0, 4, 0, 1" "conv_halo_kernel<0, 3, 0, 0" "conv_halo_kernel<0, 3, 0, 1" "conv_halo_kernel<0, 4, 1, 0" "conv_halo_kernel<0, 4, 1, 1" "attn_kernel<512" "attn_kernel<384" "conv_box2d_kernel<16, 8, 2"; do
  { echo "## $k (AE decode + cond-encode @512^2, python3 tools/perf_probe.py ae; FETCH_SIZE raw KiB: x2 on gfx950 for wide reads)"; python3 tools/pmc_sq.py "$k" $OUT/pmc_sq_ae $OUT/pmc_fetch_ae $OUT/pmc_write_ae; } >> $OUT/pmc_ae_summary.txt
done
echo "ae passes done"
find $OUT -name "*_kernel_stats.csv" | head; find $OUT -name "*.csv" -size +20M -delete; find $OUT -name "*.db" -size +20M -delete
du -sh $OUT
