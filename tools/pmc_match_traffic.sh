#!/bin/bash
# Fabric bytes per dif_match (FETCH_SIZE doubled + WRITE_SIZE, gfx950 corrections: tools/pmc_traffic.py), separate --pmc passes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/match_traffic
mkdir -p $O
run() { # tag G B opts
  MATCH_OPTS="$4" rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$1_f -o p -- python3 tools/match_prof.py $2 $3 6 > $O/$1_f.log 2>&1 &&
  MATCH_OPTS="$4" rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$1_w -o p -- python3 tools/match_prof.py $2 $3 6 > $O/$1_w.log 2>&1 &&
  python3 tools/pmc_traffic.py $O/$1_f/p_counter_collection.csv $O/$1_w/p_counter_collection.csv 9 > $O/$1.json
}
run g1_512x1m 1000000 512 "filter=2" && run b1_512x1m 1000000 512 "filter=2,frag=0" && run b1_4096x125k 125000 4096 "filter=2" && run g1_32x1m 1000000 32 "filter=2" &&
run bd_512x1m 1000000 512 "filter=1,bd=1" && run tile_512x1m 1000000 512 "filter=1,bd=0"
python3 - <<'PY'
import json
for t in ('g1_512x1m', 'b1_512x1m', 'b1_4096x125k', 'g1_32x1m', 'bd_512x1m', 'tile_512x1m'):
    d = json.load(open('gpurun_out/match_traffic/%s.json' % t))['kernels']
    for k, v in d.items():
        if k.startswith('match_') and v['read_bytes_per_forward'] > 1e6:
            print('%-16s %-34s read %.3f GB  write %.4f GB per dif_match' % (t, k[:34], v['read_bytes_per_forward'] / 1e9, v['write_bytes_per_forward'] / 1e9))
PY
