#!/bin/bash
# lanes on least-priority streams vs normal ones, with and without an RCCL process group created before the net (r04)
out=gpurun_out/r4c; mkdir -p $out
B="--no-cpu-baseline --no-throughput-mode --steps 15 --warmup 4"
run() { # tag, env, args
  DIF_OPTIONS="$2" timeout -k 10 200 python3 bench.py $B $3 > $out/$1.json 2> $out/$1.err || return 1
  python3 - "$out/$1.json" "$1" <<'PY'
import json,sys
s=open(sys.argv[1]).read(); d=json.loads(s[s.index('{"metric'):].splitlines()[0]); r=d['roofline']
print('%-28s %8.1f faces/s  embed %.3f ms  match %.3f ms  b256 %s' % (sys.argv[2], d['value'], d['phases_ms']['embed'], d['phases_ms']['match'], r.get('b256',{}).get('forward_ms_hip_events')), flush=True)
PY
}
run r100_prio_least "lane_prio=0" "" && run r100_prio_normal "lane_prio=1" "" && \
run r100_fc_prio_least "lane_prio=0" "--force-collectives" && run r100_fc_prio_normal "lane_prio=1" "--force-collectives" && \
run r50_prio_least "lane_prio=0" "--workload r50" && run r50_prio_normal "lane_prio=1" "--workload r50" && \
run r50_fc_prio_least "lane_prio=0" "--workload r50 --force-collectives" && run r50_fc_prio_normal "lane_prio=1" "--workload r50 --force-collectives"
