R=$(pwd); OUT=$R/gpurun_out/tcopy; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
AMD_LOG_LEVEL=4 python $R/tools/ubench/torch_copy2.py own_host_own_stream > $OUT/log_torch.txt 2>&1
grep -i "sdma\|HSA Copy\|engine\|blit\|copyBuffer" $OUT/log_torch.txt | head -40
echo ======= ubench
AMD_LOG_LEVEL=4 $R/tools/ubench/d2h_copy 4 > $OUT/log_ubench.txt 2>&1
grep -i "sdma\|HSA Copy\|engine\|blit\|copyBuffer" $OUT/log_ubench.txt | head -20
