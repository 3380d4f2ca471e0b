set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/chain_prof.py 248 5
rm -rf gpurun_out/chain_prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/chain_prof -o c -- python3 tools/chain_prof.py 248 5 > gpurun_out/chain_prof.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/chain_prof/**/c_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last call: from the last k_witness_bits on
idx = [i for i, r in enumerate(rows) if "k_witness_bits" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} us  +{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  {n}  grid={r["Grid_Size"]} wg={r["Workgroup_Size"]}')
PY
