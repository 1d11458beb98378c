mkdir -p gpurun_out/r2x
timeout -k 10 400 python -m pytest tests/test_embed_gpu.py tests/test_r100_full_gpu.py -m gpu -q -x > gpurun_out/r2x/tests.log 2>&1; tail -3 gpurun_out/r2x/tests.log
timeout -k 10 100 python tools/layer_profile.py iresnet100 256 > gpurun_out/r2x/r100.txt 2>&1 || exit 1
DIF_NO_YSUB=1 timeout -k 10 100 python tools/layer_profile.py iresnet100 256 > gpurun_out/r2x/r100_noysub.txt 2>&1 || exit 1
grep -E "^conv1|TOTAL" gpurun_out/r2x/r100.txt gpurun_out/r2x/r100_noysub.txt
timeout -k 10 100 python tools/layer_profile.py resnet 256 > gpurun_out/r2x/r50.txt 2>&1; grep -E "^conv1_conv|TOTAL" gpurun_out/r2x/r50.txt
