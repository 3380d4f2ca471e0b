"""Clock, matrix-pipe busy fraction and duration of k_mmstream_p for several builds of libmfhip.so (MFHIP_LIB), one box: two rocprofv3 --pmc passes and one
--kernel-trace --stats pass per build over tools/batch_prof.py.  Run ON the GPU box.  dev tool.   usage: python3 tools/mm_variant_clock.py <lib.so> ..."""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TMPDIR"] = "/tmp"
out = {}
for lib in sys.argv[1:]:
    lib, _, wave1 = lib.partition("@")  # lib.so@w: run with the one-wave-per-SIMD body (MFUOCO_MM_WAVE1=1, read by tools/batch_prof.py)
    name = os.path.basename(lib) + ("@" + wave1 if wave1 else "")
    env = dict(os.environ, MFHIP_LIB=os.path.abspath(lib), MFUOCO_MM_WAVE1="1" if wave1 else "0")
    vals = {}
    for gi, g in enumerate(["GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"]):
        d = f"/tmp/mvc_{name}_{gi}"
        subprocess.run(["rm", "-rf", d])
        r = subprocess.run(["rocprofv3", "--pmc", *g.split(), "--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "tools", "batch_prof.py")],
                           capture_output=True, text=True, cwd="/tmp", env=env)
        if r.returncode:
            print(name, "pmc failed", r.stderr[-300:], flush=True)
            continue
        for f in glob.glob(d + "/*/*_counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                if "k_mmstream_p(" in row["Kernel_Name"] or "k_mmstream_w(" in row["Kernel_Name"]:
                    vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    d = f"/tmp/mvc_{name}_st"
    subprocess.run(["rm", "-rf", d])
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "tools", "batch_prof.py")],
                       capture_output=True, text=True, cwd="/tmp", env=env)
    avg_ns = None
    for f in glob.glob(d + "/*/*_kernel_stats.csv"):
        for row in csv.DictReader(open(f)):
            if "k_mmstream_p(" in row["Name"] or "k_mmstream_w(" in row["Name"]:
                avg_ns = float(row["AverageNs"])
    m = {c: sum(v) / len(v) for c, v in vals.items()}
    res = {"avg_launch_ms": avg_ns / 1e6 if avg_ns else None}
    if avg_ns and "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        res.update(clock_ghz=cyc / avg_ns, mfma_busy_frac=m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc)
    if "SQ_WAVE_CYCLES" in m:
        res.update(wait_any=m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], wait_inst=m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], active=m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"])
    out[name] = res
    print(name, {k: (round(v, 4) if v is not None else None) for k, v in res.items()}, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r4_mm_variant_clock.json"), "w"), indent=1)
