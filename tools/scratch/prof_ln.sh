#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-parity --no-roofline > /tmp/pl.json 2> /tmp/pl.err
f=$(find /tmp/pl -name "*kernel_stats.csv" | head -1)
grep -E "ln_bwd|ln_fwd|attn_fwd_bf16_dual|attn_bwd_bf16_dual" $f | awk -F, '{printf "%-60s calls %s avg %.1f us\n", substr($1,1,60), $2, $4/1000}'
cut -c1-160 /tmp/pl.json
