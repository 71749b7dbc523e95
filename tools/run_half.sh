#!/bin/bash
# GPU box: the half-tile configuration of the two-plane kernel (conv_ws_half_kernel: 144 x 128 tiles, two workgroups per CU) against the
# full-tile one -- bit equality of every output first, then durations by start delay of a CU's second workgroup.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export BENCH_SHAPES="16,48,48,256,1024,1,1;16,48,48,1024,256,1,1;16,96,96,128,512,1,1;16,96,96,512,128,1,1;16,192,192,64,256,1,1"
rm -f /tmp/dump_full.txt /tmp/dump_half.txt
for e in 0 1 2 3; do
  DML_WS_HALF=0 BENCH_DUMP=/tmp/dump_full.txt BENCH_EPI=$e python3 $R/tools/bench_h2.py $([ $e = 0 ] && echo all || echo dgrad) only=h2 >/dev/null 2>&1
  DML_WS_HALF=2 BENCH_DUMP=/tmp/dump_half.txt BENCH_EPI=$e python3 $R/tools/bench_h2.py $([ $e = 0 ] && echo all || echo dgrad) only=h2 >/dev/null 2>&1
done
if cmp -s /tmp/dump_full.txt /tmp/dump_half.txt; then echo "BIT-EQUAL: $(wc -l < /tmp/dump_full.txt) launches, every output tensor identical"; else echo "DIFFERENT"; diff /tmp/dump_full.txt /tmp/dump_half.txt | head -20; fi
for cfg in "0 0" "1 0" "1 300" "1 600" "1 1000" "2 0" "2 600"; do
  set -- $cfg
  echo "== DML_WS_HALF=$1 stagger $2 x 10 ns: forward with statistics | data gradient EPI 3 | EPI 2"
  DML_WS_HALF=$1 DML_WS_HALF_STAGGER=$2 python3 $R/tools/bench_h2.py fwd only=h2 2>/dev/null
  DML_WS_HALF=$1 DML_WS_HALF_STAGGER=$2 BENCH_EPI=3 python3 $R/tools/bench_h2.py dgrad only=h2 2>/dev/null
  DML_WS_HALF=$1 DML_WS_HALF_STAGGER=$2 BENCH_EPI=2 python3 $R/tools/bench_h2.py dgrad only=h2 2>/dev/null
done
