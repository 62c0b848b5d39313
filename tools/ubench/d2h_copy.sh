# GPU box: which engine serves D2H copies (kernel trace shows __amd_rocclr_copyBuffer when it is the blit kernel)
R=$(pwd); OUT=$R/gpurun_out/d2h; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
$R/tools/ubench/d2h_copy 4
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- $R/tools/ubench/d2h_copy 4 > /dev/null 2>&1
f=$(find $OUT/t -name "t_kernel_stats.csv" | head -1); cut -c1-140 $f
rm -rf $OUT/t
