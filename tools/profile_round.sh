#!/bin/bash
# dev tool, run ON the GPU box through gpurun:   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r03 <commit>'
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command (what bench.py's roofline objects must agree with);
# 2. SEPARATE --pmc passes (gpurun refuses --pmc together with the trace domains; MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass) over the
#    headline call (tools/batch_prof.py), the LWE batch (tools/enc_prof.py) and the decryption batch (tools/decrypt_time.py).
# The raw rocprofv3 output stays in /tmp on the box (it exceeds what gpurun copies back); tools/summarize_profile.py condenses it into
# gpurun_out/<tag>_profiles/, whose files are then committed under profiles/.
set -o pipefail
tag=${1:-r03}
commit=${2:-unknown}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
raw=/tmp/prof_$tag
rm -rf $raw; mkdir -p $raw gpurun_out/${tag}_profiles
rocprofv3 --kernel-trace --stats --output-format csv -d $raw/${tag}_stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-drop-in > $raw/${tag}_stats.log 2> $raw/${tag}_stats.err || { tail -5 $raw/${tag}_stats.err; exit 1; }
echo "[profile] stats done: $(ls $raw/${tag}_stats/*/ | head -3 | tr '\n' ' ')"
declare -A counters=( [fetch]="FETCH_SIZE" [write]="WRITE_SIZE" [tcc]="TCC_HIT_sum TCC_MISS_sum" [sq]="SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE" )
declare -A progs=( [batch]="tools/batch_prof.py" [enc]="tools/enc_prof.py" [dec]="tools/decrypt_time.py 65536" )
for name in fetch write tcc sq; do
  for pn in batch enc dec; do
    rocprofv3 --pmc ${counters[$name]} --output-format csv -d $raw/${tag}_pmc_${name}_${pn} -- python3 ${progs[$pn]} > $raw/${tag}_pmc_${name}_${pn}.log 2>&1 || { echo "[profile] pmc $name $pn failed"; tail -5 $raw/${tag}_pmc_${name}_${pn}.log; exit 1; }
    echo "[profile] pmc $name $pn done: $(du -sh $raw/${tag}_pmc_${name}_${pn} | cut -f1)"
  done
done
PROF_RAW=$raw PROF_OUT=gpurun_out/${tag}_profiles python3 tools/summarize_profile.py $tag $commit
ls -la gpurun_out/${tag}_profiles
