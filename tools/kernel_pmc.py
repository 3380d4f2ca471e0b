"""Per-kernel PMC counters of any tool program: one rocprofv3 --pmc pass per counter group, the rows of kernels whose name contains <substr> averaged.  Run ON the
GPU box.  dev tool.   usage: python3 tools/kernel_pmc.py <kernel substring> <out.json> -- python3 tools/<prog>.py args..."""
import csv, glob, json, os, subprocess, sys
os.environ["TMPDIR"] = "/tmp"
sub, out = sys.argv[1], sys.argv[2]
cmd = sys.argv[sys.argv.index("--") + 1:]
groups = ["GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES", "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA",
          "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INST_CYCLES_VMEM",
          "SQ_WAIT_INST_LDS SQ_INSTS_WAVE32_LDS SQ_INST_LEVEL_LDS", "FETCH_SIZE", "TCC_HIT_sum TCC_MISS_sum"]
res = {}
for gi, g in enumerate(groups):
    d = f"/tmp/kpmc_{gi}"
    subprocess.run(["rm", "-rf", d])
    r = subprocess.run(["rocprofv3", "--pmc", *g.split(), "--output-format", "csv", "-d", d, "--", *cmd], capture_output=True, text=True, cwd="/tmp")
    if r.returncode:
        print("group", g, "failed:", r.stderr[-300:].replace("\n", " | "), flush=True)
        continue
    vals = {}
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if sub in row["Kernel_Name"]:
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for c, v in vals.items():
        res[c] = {"launches": len(v), "mean": sum(v) / len(v)}
    print(g, {c: round(sum(v) / len(v), 1) for c, v in vals.items()}, flush=True)
json.dump({"kernel": sub, "command": " ".join(cmd), "counters": res}, open(out, "w"), indent=1)
