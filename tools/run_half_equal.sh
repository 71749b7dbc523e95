#!/bin/bash
# GPU box: bit equality of the half-tile configuration (DML_WS_HALF=2) with the full-tile one on seeded inputs (tools/bench_h2.py digests)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export BENCH_SHAPES="16,48,48,256,1024,1,1;16,48,48,1024,256,1,1;16,96,96,128,512,1,1;16,96,96,512,128,1,1;16,192,192,64,256,1,1"
rm -f /tmp/dump_full.txt /tmp/dump_half.txt
for e in 0 1 2 3; do
  for h in 0 2; do
    f=/tmp/dump_full.txt; [ $h = 2 ] && f=/tmp/dump_half.txt
    DML_WS_HALF=$h BENCH_DUMP=$f BENCH_EPI=$e python3 $R/tools/bench_h2.py $([ $e = 0 ] && echo fwd || echo dgrad) only=h2 >/dev/null 2>&1
    [ $e = 0 ] && DML_WS_HALF=$h BENCH_DUMP=$f BENCH_EPI=0 python3 $R/tools/bench_h2.py dgrad only=h2 >/dev/null 2>&1
  done
done
if cmp -s /tmp/dump_full.txt /tmp/dump_half.txt; then echo "BIT-EQUAL: $(wc -l < /tmp/dump_full.txt) launches, every output tensor identical"; else echo "DIFFERENT"; diff /tmp/dump_full.txt /tmp/dump_half.txt | head -50; fi
