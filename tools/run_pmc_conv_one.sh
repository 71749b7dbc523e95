#!/bin/bash
# GPU box: HBM bytes of single conv launches (separate --pmc passes).  usage: run_pmc_conv_one.sh "H C N k mode [accum]" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "$@"; do
  rm -rf /tmp/po_rd /tmp/po_wr
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/po_rd -- python3 $R/tools/pmc_conv_one.py $spec > /tmp/po.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/po_wr -- python3 $R/tools/pmc_conv_one.py $spec > /tmp/po2.log 2>&1
  echo "== $spec: $(grep algorithmic /tmp/po.log)"
  python3 $R/tools/traffic_summary.py /tmp/po_rd /tmp/po_wr 12 | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['kernels'].items():
    if k.startswith('conv_'): print('   %s: read %.1f MB, write %.1f MB per launch (%.0f launches)' % (k, v['read_GB']*1e3, v['write_GB']*1e3, v['launches_per_step']*12))
"
done
