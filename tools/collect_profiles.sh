#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the
# exact bench.py commands, plus the un-profiled bench lines and per-layer tables.
# Outputs land in gpurun_out/; tools/check_profiles.py copies the summaries into profiles/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# Per-kernel durations are taken with ONE lane (DIF_STREAMS=1): with two lanes kernels of the two
# streams overlap and the sum of per-kernel durations double-counts the wall clock.
for w in r50 r100; do
  export DIF_STREAMS=1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks1_$w -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ks1_$w.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pm_$w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pm_$w.log 2>&1
  unset DIF_STREAMS
  export DIF_STREAMS=2     # the "two lanes" files (IResNet-100's default; ResNet50V2 defaults to one lane)
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$w -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ks_$w.log 2>&1
  unset DIF_STREAMS
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pf_$w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pf_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pw_$w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pw_$w.log 2>&1
done
DIF_STREAMS=1 python3 bench.py --no-cpu-baseline > gpurun_out/bench1_r50.json 2>/dev/null
DIF_STREAMS=1 python3 bench.py --workload r100 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench1_r100.json 2>/dev/null
python3 bench.py > gpurun_out/bench_r50.json 2>/dev/null
python3 bench.py --workload r100 --steps 10 --warmup 3 > gpurun_out/bench_r100.json 2>/dev/null
python3 tools/layer_profile.py resnet 256 > gpurun_out/layers_r50.txt
python3 tools/layer_profile.py iresnet100 256 > gpurun_out/layers_r100.txt
echo done
