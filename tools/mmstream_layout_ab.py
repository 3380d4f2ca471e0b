"""Same-process, same-box A/B of k_mmstream's launch layouts (mfh_set_mm_stream / mfh_set_batch_launch) on the headline call:
mfh_prove_batch, 1020 statements, default instance, CRS expanded inside the call.  Configurations are run round-robin REPS times so that
clock / box drift hits all of them alike; every configuration's proofs are compared bit for bit with the first one's.  dev tool.
usage: python tools/mmstream_layout_ab.py [--nb=1020] [--reps=3] [cfg ...]     cfg = name:map,persistent,sync,spin[,ngl[,bw_merged]]"""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import c_lwe_snarks_amd as mf

nb, reps, cfgs = 1020, 3, []
for a in sys.argv[1:]:
    if a.startswith("--nb="): nb = int(a[5:])
    elif a.startswith("--reps="): reps = int(a[7:])
    else:
        name, v = a.split(":")
        v = [int(x) for x in v.split(",")]
        v = v + [4] * (5 - len(v)) if len(v) < 5 else v
        cfgs.append((name, v + [1] * (6 - len(v))))
if not cfgs:
    cfgs = [("base", [0, 0, 0, 0, 4, 0]), ("map1", [1, 0, 0, 0, 4, 0]), ("pers_map1_nosync", [1, 1, 0, 0, 4, 0]), ("pers_map1_sync1", [1, 1, 1, 64, 4, 0]),
            ("pers_map1_sync2", [1, 1, 2, 64, 4, 0]), ("pers_map0_sync1", [0, 1, 1, 64, 4, 0])]
p = mf.DEFAULT
ctx = mf.Context(p, 0)
ctx.set_seed(bytes((37 * i + 11) & 0xFF for i in range(40)))
inst = bench.build_instance(mf, ctx, torch, p, 20260101)
ctx.ssp_prepare(inst["d_ssp"])
d_crs = ctx.setup(inst["d_ssp"], inst["alpha"], inst["beta"], inst["s"], inst["sk"], inst["err"])
rng = np.random.default_rng(5)
deltas = [int(x) for x in rng.integers(0, mf.P, size=nb, dtype=np.uint64)]
mags = [rng.integers(0, 256, size=400, dtype=np.uint8).tobytes() for _ in range(nb)]
signs = [bytes(5)] * nb
bits = [inst["bits"]] * nb
out = ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs)
ref = None
res = {name: {"ms_call": [], "ms_launch": []} for name, _ in cfgs}
for r in range(reps + 1):  # round 0 = warm-up + bit-identity
    for name, (mp, pers, sync, spin, ngl, bwm) in cfgs:
        ctx.set_batch_launch(ngl, True)
        ctx.set_batch_bw(bool(bwm))
        ctx.set_mm_stream(mp, int(pers), sync, spin)
        ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs, out=out)
        if r == 0:
            torch.cuda.synchronize()
            if ref is None: ref = out.clone()
            same = bool(torch.equal(out, ref))
            res[name]["bit_identical"] = same
            print(f"{name}: proofs bit-identical to {cfgs[0][0]}: {same}", flush=True)
            continue
        ctx.set_timing(True); ctx.timing_drain("mmstream_rounds")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 3
        for _ in range(n): ctx.prove_batch(d_crs, inst["d_ssp"], bits, deltas, mags, signs, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        cnt, ms, rows = ctx.timing_drain("mmstream_rounds"); ctx.set_timing(False)
        res[name]["ms_call"].append(round(dt * 1e3, 3)); res[name]["ms_launch"].append(round(ms / max(cnt, 1), 4)); res[name]["launches_per_call"] = cnt // n
        print(f"  round {r} {name:22s} {dt*1e3:8.2f} ms/call  k_mmstream {ms/max(cnt,1):7.3f} ms x {cnt//n} launches  -> {nb/dt:8.1f} proofs/s", flush=True)
for name, _ in cfgs:
    r = res[name]
    r["ms_call_min"] = min(r["ms_call"]); r["ms_launch_min"] = min(r["ms_launch"])
    r["ms_call_med"] = float(np.median(r["ms_call"])); r["ms_launch_med"] = float(np.median(r["ms_launch"]))
print(json.dumps({"nb": nb, "configs": {n: v for n, v in cfgs}, "results": res}))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"nb": nb, "configs": {n: v for n, v in cfgs}, "results": res}, open(os.path.join(ROOT, "gpurun_out", "r4_mmstream_layout_ab.json"), "w"), indent=1)
