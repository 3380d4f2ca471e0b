#!/bin/bash
# dev tool, run ON the GPU box through gpurun:   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r03'
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command (what bench.py's roofline objects must agree with);
# 2. SEPARATE --pmc passes (gpurun refuses --pmc together with the trace domains; MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass) over the
#    headline call (tools/batch_prof.py), the LWE batch (tools/enc_prof.py) and the decryption batch (tools/decrypt_time.py);
# everything under gpurun_out/<tag>_*; tools/summarize_profile.py <tag> then writes the summaries committed under profiles/.
set -o pipefail
tag=${1:-r03}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
if [ "${2:-all}" != "pmc" ]; then
  rocprofv3 --kernel-trace --stats -d $out/${tag}_stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/${tag}_stats.log 2> $out/${tag}_stats.err || { tail -5 $out/${tag}_stats.err; exit 1; }
  echo "[profile] stats done"
fi
declare -A counters=( [fetch]="FETCH_SIZE" [write]="WRITE_SIZE" [sq]="SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE" )
declare -A progs=( [batch]="tools/batch_prof.py" [enc]="tools/enc_prof.py" [dec]="tools/decrypt_time.py 65536" )
for name in fetch write sq; do
  for pn in batch enc dec; do
    rocprofv3 --pmc ${counters[$name]} -d $out/${tag}_pmc_${name}_${pn} -- python3 ${progs[$pn]} > $out/${tag}_pmc_${name}_${pn}.log 2>&1 || { echo "[profile] pmc $name $pn failed"; tail -5 $out/${tag}_pmc_${name}_${pn}.log; exit 1; }
    echo "[profile] pmc $name $pn done"
  done
done
