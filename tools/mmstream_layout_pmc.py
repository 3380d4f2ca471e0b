"""L2-miss traffic and L2 hit rate of k_mmstream per launch layout: one rocprofv3 --pmc pass per (layout, counter group) over tools/mmstream_layout_one.py.
Run ON the GPU box.  FETCH_SIZE is in KiB and, on gfx950, half the bytes of a wide coalesced read stream (MI355X_MICROARCH.md): doubled here.  dev tool.
usage: python3 tools/mmstream_layout_pmc.py name:map,persistent,sync,spin[,ngl] ..."""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TMPDIR"] = "/tmp"
groups = {"fetch": "FETCH_SIZE", "write": "WRITE_SIZE", "tcc": "TCC_HIT_sum TCC_MISS_sum", "clk": "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"}
res = {}
for a in sys.argv[1:]:
    name, cfg = a.split(":")
    res[name] = {"cfg": cfg}
    for gname, ctrs in groups.items():
        d = f"/tmp/mmpmc_{name}_{gname}"
        subprocess.run(["rm", "-rf", d])
        r = subprocess.run(["rocprofv3", "--pmc", *ctrs.split(), "--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "tools", "mmstream_layout_one.py"), cfg],
                           capture_output=True, text=True, cwd="/tmp")
        if r.returncode:
            print(name, gname, "failed", r.stderr[-400:], flush=True)
            continue
        vals = {}
        for f in glob.glob(d + "/*/*_counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                if row["Kernel_Name"].startswith(("(anonymous namespace)::k_mmstream(", "(anonymous namespace)::k_mmstream_p(")):
                    vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for c, v in vals.items():
            res[name][c] = {"launches": len(v), "mean": sum(v) / len(v)}
        print(name, gname, {c: round(sum(v) / len(v), 1) for c, v in vals.items()}, flush=True)
    x = res[name]
    if "FETCH_SIZE" in x: x["fetch_GB_x2"] = x["FETCH_SIZE"]["mean"] * 1024 * 2 / 1e9
    if "WRITE_SIZE" in x: x["write_GB"] = x["WRITE_SIZE"]["mean"] * 1024 / 1e9
    if "TCC_HIT_sum" in x: x["l2_hit_rate"] = x["TCC_HIT_sum"]["mean"] / (x["TCC_HIT_sum"]["mean"] + x["TCC_MISS_sum"]["mean"])
    print(name, {k: v for k, v in x.items() if not isinstance(v, dict)}, flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r4_mmstream_layout_pmc.json"), "w"), indent=1)
