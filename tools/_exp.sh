timeout -k 10 300 python -m pytest tests/test_embed_gpu.py -m gpu -q -x -k "stem or vs_oracle or full_batch" 2>&1 | tail -2
timeout -k 10 100 python tools/layer_profile.py resnet 256 2>&1 | grep -E "^conv1_conv|TOTAL"
timeout -k 10 100 python tools/layer_profile.py iresnet100 256 2>&1 | grep -E "^conv1 |TOTAL"
