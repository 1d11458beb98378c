#!/bin/bash
# Round-5 evidence run, on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the EXACT driver command
# (python3 bench.py --gpus 1 --steps 20 --warmup 5), separate PMC passes (never combined with tracing), the
# un-profiled bench lines of every workload, per-layer tables and the small-batch latency table.
# Outputs land in gpurun_out/r05/; tools/check_profiles.py copies the summaries into profiles/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05
mkdir -p $O
prof() { local tag=$1; shift; rocprofv3 "$@" --output-format csv -d $O/$tag -o p -- python3 bench.py --gpus 1 ${ARGS} > $O/$tag.json 2> $O/$tag.err; }
# 1. the driver's command, default executor (two lanes: kernel durations of the two streams overlap)
ARGS="--steps 20 --warmup 5" prof ks_default --kernel-trace --stats &&
# 2. the same workload on ONE lane: kernels run back to back, so per-kernel durations add up to the forward
DIF_STREAMS=1 ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-throughput-mode --no-latency" prof ks_default_1lane --kernel-trace --stats &&
# 3. counters, each in its own pass
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-latency" prof pf_default --pmc FETCH_SIZE &&
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-latency" prof pw_default --pmc WRITE_SIZE &&
DIF_STREAMS=1 ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-latency" prof pm_default --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE &&
ARGS="--workload r100 --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-latency" prof pf_r100 --pmc FETCH_SIZE &&
ARGS="--workload r100 --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-latency" prof pw_r100 --pmc WRITE_SIZE &&
DIF_STREAMS=1 ARGS="--workload r50 --steps 10 --warmup 3 --no-cpu-baseline --no-latency" prof ks_r50_1lane --kernel-trace --stats &&
ARGS="--workload r50 --steps 3 --warmup 1 --no-cpu-baseline --no-latency" prof pf_r50 --pmc FETCH_SIZE &&
ARGS="--workload r50 --steps 3 --warmup 1 --no-cpu-baseline --no-latency" prof pw_r50 --pmc WRITE_SIZE &&
# 4. un-profiled bench lines
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err &&
DIF_STREAMS=1 python3 bench.py --no-cpu-baseline > $O/bench_default_1lane.json 2>/dev/null &&
python3 bench.py --workload r100 > $O/bench_r100.json 2>/dev/null &&
python3 bench.py --workload r50 > $O/bench_r50.json 2>/dev/null &&
DIF_STREAMS=1 python3 bench.py --workload r50 --no-cpu-baseline > $O/bench_r50_1lane.json 2>/dev/null &&
python3 bench.py --workload r100_arc --no-cpu-baseline > $O/bench_r100_arc.json 2>/dev/null &&
python3 bench.py --workload r100_1m_bf16x3 --no-cpu-baseline > $O/bench_r100_1m_bf16x3.json 2>/dev/null &&
python3 bench.py --workload r100_1m_bf16x2 --no-cpu-baseline > $O/bench_r100_1m_bf16x2.json 2>/dev/null &&
python3 bench.py --force-collectives --no-cpu-baseline --no-throughput-mode > $O/bench_default_fc.json 2>/dev/null &&
python3 bench.py --workload frames --steps 3 --warmup 1 > $O/bench_frames.json 2>/dev/null &&
python3 bench.py --workload frames_mtcnn --steps 5 --warmup 2 > $O/bench_frames_mtcnn.json 2>/dev/null &&
# 5. per-layer tables, latency
python3 tools/layer_profile.py iresnet100 256 > $O/layers_r100.txt 2>&1 &&
python3 tools/layer_profile.py resnet 256 > $O/layers_r50.txt 2>&1 &&
python3 tools/layer_profile.py iresnet100 256 bf16x2 > $O/layers_r100_bf16x2.txt 2>&1 &&
python3 tools/layer_profile.py yolov3 64 > $O/layers_yolov3.txt 2>&1 &&
python3 tools/latency.py > $O/latency.txt 2>&1 &&
(rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_frames_mtcnn -o p -- python3 bench.py --workload frames_mtcnn --steps 3 --warmup 1 --no-cpu-baseline > $O/ks_frames_mtcnn.json 2> $O/ks_frames_mtcnn.err && python3 tools/kstats.py $O/ks_frames_mtcnn 4 > $O/ks_frames_mtcnn.txt) &&
# 5b. round 5: the small-batch regime -- per-layer tables, kernel-trace stats and block traces at batch 1 / 8 / 32
(for a in iresnet100 resnet; do for b in 1 8 32; do python3 tools/layer_profile.py $a $b > $O/layers_${a}_b$b.txt 2>&1 || exit 1; done; done) &&
(for c in "iresnet100 1" "iresnet100 8" "iresnet100 32" "resnet 1" "resnet 8"; do set -- $c; rocprofv3 --kernel-trace --stats --output-format csv -d $O/ksb_$1_$2 -o p -- python3 tools/small_batch.py $1 $2 50 > $O/ksb_$1_$2.log 2>&1 && python3 tools/kstats.py $O/ksb_$1_$2 55 > $O/ksb_$1_$2.txt || exit 1; done) &&
(for b in 1 8; do DIF_OPTIONS=dbg=256 python3 tools/bf3_trace.py $b f32 iresnet100 2> $O/trace_r100_b$b.txt > /dev/null; done; DIF_OPTIONS=dbg=256 python3 tools/bf3_trace.py 1 f32 resnet 2> $O/trace_r50_b1.txt > /dev/null) &&
python3 tools/small_batch_modes.py iresnet100 1 2>/dev/null > $O/launch_modes.txt && python3 tools/small_batch_modes.py resnet 1 2>/dev/null >> $O/launch_modes.txt &&
python3 tools/time_gallery_set.py 2>/dev/null > $O/gallery_set.txt &&
DIF_STREAMS=1 ARGS="--workload r100_1m_bf16x2 --steps 3 --warmup 1 --no-cpu-baseline --no-latency" prof pm_bf16x2 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE &&
DIF_OPTIONS=dbg=256 python3 tools/bf3_trace.py 256 f32 2> $O/trace_r100_f32.txt > /dev/null &&
DIF_OPTIONS=dbg=256 python3 tools/bf3_trace.py 256 bf16x2 2> $O/trace_r100_bf16x2.txt > /dev/null &&
DIF_OPTIONS=dbg=256 python3 tools/bf3_trace.py 256 f32 resnet 2> $O/trace_r50_f32.txt > /dev/null &&
DIF_OPTIONS=dbg=768 python3 tools/bf3_trace.py 256 f32 2> $O/place_r100_f32.txt > /dev/null &&
(for b in 64 128 256 512 1024; do python3 tools/time_embed.py iresnet100 $b 2>/dev/null | tail -1; done; for b in 64 128 256 512 1024 2048; do python3 tools/time_embed.py resnet $b 2>/dev/null | tail -1; done) > $O/batch_sweep.txt &&
python3 tools/bf_tier_gates.py 256 512 2>/dev/null | grep -v amdgpu > $O/bf_tier_gates.txt &&
python3 tools/match_ab.py 2>/dev/null | tail -4 > $O/match_ab.txt &&
(bash tools/pmc_match.sh > /dev/null 2>&1; cp gpurun_out/match_pmc/table.txt $O/match_pmc_table.txt) &&
(bash tools/pmc_match_traffic.sh 2>/dev/null | grep "per dif_match" > $O/match_traffic.txt)
echo "collect rc=$?"
