set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/chain4
timeout -k 10 800 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/chain4 -o c -- python3 tools/chain_prof.py 248 1 config4 > gpurun_out/chain4.log 2>&1
tail -2 gpurun_out/chain4.log
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/chain4/**/c_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_witness_bits" in r["Kernel_Name"]]
# last call = after the last-but-(passes) witness_bits: take the second half of the list
half = idx[len(idx)//2]
last = rows[half:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:11.1f} us  +{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:10.1f} us  {n}')
PY
