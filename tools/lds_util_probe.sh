#!/bin/bash
# dev tool (GPU box): how busy is the LDS pipe, and at what clock, under the AES kernels?  SQ_LDS_IDX_ACTIVE = LDS-array cycles, GRBM_GUI_ACTIVE / 8 = cycles of the
# kernel (per XCD); the duration comes from the same pass's kernel trace is not available with --pmc, so the clock is GRBM_GUI_ACTIVE / 8 / the duration tools/eval_ab.py prints.
export TMPDIR=/tmp
raw=/tmp/lds_probe; rm -rf $raw; mkdir -p $raw
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $raw/eval -- python3 tools/eval_ab.py > $raw/eval.log 2>&1 || { tail -5 $raw/eval.log; exit 1; }
grep "path" $raw/eval.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/lds_probe/eval/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    if 'k_eval' in k and 'reduce' not in k:
        big = {c: [x for x in v if x >= 0.5 * max(v)] for c, v in cs.items()}
        m = {c: sum(v) / len(v) for c, v in big.items()}
        cyc = m['GRBM_GUI_ACTIVE'] / 8
        print(k, 'launches', len(big['GRBM_GUI_ACTIVE']), 'cycles per XCD %.3e' % cyc, 'LDS insts %.3e' % m['SQ_INSTS_LDS'], 'LDS_IDX_ACTIVE %.3e' % m['SQ_LDS_IDX_ACTIVE'],
              'bank conflict %.3e' % m['SQ_LDS_BANK_CONFLICT'], 'SQ_BUSY_CYCLES %.3e' % m['SQ_BUSY_CYCLES'],
              '-> LDS-array cycles per CU / kernel cycles = %.3f' % (m['SQ_LDS_IDX_ACTIVE'] / 256 / cyc))
PY
