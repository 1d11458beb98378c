mkdir -p gpurun_out/r3b
(timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/r3b/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3b/tests.log); tail -3 gpurun_out/r3b/tests.log
