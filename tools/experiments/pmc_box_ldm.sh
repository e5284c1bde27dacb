# SQ / FETCH / instruction-mix counters of the box conv inside the eager latent-UNet forward (separate rocprofv3 --pmc passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_box_ldm
rm -rf $OUT; mkdir -p $OUT
GG_NO_GRAPH=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/tools/perf_probe.py ldm > /dev/null 2>&1
GG_NO_GRAPH=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/tools/perf_probe.py ldm > /dev/null 2>&1
GG_NO_GRAPH=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/tools/perf_probe.py ldm > /dev/null 2>&1
cd $R
for k in "conv_box2d_kernel<16, 6, 1, 0, 1, 0, 1>" "conv_box2d_kernel<16, 12, 1, 0, 1, 0, 1>" "conv_box2d_kernel<4, 1, 1, 0, 1, 0, 4>" "conv_box2d_kernel<8, 2, 1, 0, 1, 0, 2>" "gn_apply_acc_lean_kernel<2>" "gn_fused_small_kernel" "attn_kernel<32, 256, 2>"; do
  { echo "## $k (eager latent-UNet forwards N=1 @64x64 and N=4 @32x32, python3 tools/perf_probe.py ldm; FETCH_SIZE raw KiB: x2 on gfx950 for wide reads)"; python3 tools/pmc_sq.py "$k" $OUT/sq $OUT/fetch $OUT/write; } >> $OUT/pmc_box_ldm_summary.txt
done
cat $OUT/pmc_box_ldm_summary.txt | head -60
find $OUT -name "*.csv" -size +10M -delete
