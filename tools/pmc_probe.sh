#!/bin/bash
# rocprofv3 counter pass of a stand-alone probe binary (tools/rowpass2_probe_ns and friends): the wait / LDS group only,
# --kernel-trace the only trace domain.  Usage (GPU box, repository root): tools/pmc_probe.sh <out-dir-under-gpurun_out> <binary> [args]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
BIN=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT -- $BIN "$@" > $OUT.log 2>&1 || echo "pass failed"
python3 - $OUT <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]; tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
for k, c in tot.items():
    print(k, 'launches', n[k], {a: '%.3g' % (b / max(n[k], 1)) for a, b in sorted(c.items())},
          'lds conflict share %.2f' % (c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1)),
          'wait share %.2f' % (c['SQ_WAIT_ANY'] / max(c['SQ_WAVE_CYCLES'], 1)))
PY
