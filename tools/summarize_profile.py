"""Condense rocprofv3 outputs under gpurun_out/ into the small summaries committed under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
RAW = os.environ.get("PROF_RAW", os.path.join(ROOT, "gpurun_out"))  # where the rocprofv3 output directories are (on the GPU box: /tmp, the raw CSVs exceed what gpurun copies back)
out_dir = os.environ.get("PROF_OUT", os.path.join(ROOT, "profiles"))
os.makedirs(out_dir, exist_ok=True)


def one(pattern):
    g = glob.glob(os.path.join(RAW, pattern))
    return max(g, key=os.path.getmtime) if g else None   # gpurun merges outputs of earlier calls: take the newest


# kernel stats
f = one(f"{tag}_stats/*/*_kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as o:
        commit_ = sys.argv[2] if len(sys.argv) > 2 else os.popen("git -C %s rev-parse --short HEAD" % ROOT).read().strip()
        o.write(f"# commit {commit_}: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-drop-in   (MI355X, 1 GPU; default batch mode: the single-proof path (regenerated and resident), then the headline run, then the regenerate-per-group and resident-image variants of the same batch, then the LWE encryption and decryption batches).  k_mmstream_p = the persistent 16-group S + AS launches, one per super-group of 255 proofs (bench.py's roofline kernel), k_mmstream1 = single-group launches (b_w)\n")
        o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            o.write(",".join(['"' + r["Name"][:110].replace('"', "'") + '"', r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]) + "\n")
    log = one(f"{tag}_stats.log")
    if log:
        for line in open(log):
            if line.startswith("{"):
                open(os.path.join(out_dir, f"{tag}_bench_under_rocprof.json"), "w").write(line)

# PMC: separate passes per counter group (fetch / write / sq) and per program (batch = the headline call, enc = the LWE batch, dec = the decryption batch;
# round 1/2 layout: one directory per counter group)
commit = sys.argv[2] if len(sys.argv) > 2 else os.popen("git -C %s rev-parse --short HEAD" % ROOT).read().strip()
res = {}
for name in ("fetch", "write", "tcc", "sq"):
    files = glob.glob(os.path.join(RAW, f"{tag}_pmc_{name}*/*/*_counter_collection.csv"))
    for f in files:
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                big = [x for x in v if x >= 0.5 * max(v)] if max(v) > 0 else v  # the launches of the larger shape
                res.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v), "max": max(v), "mean_of_large_launches": sum(big) / len(big),
                                            "large_launches": len(big)}
json.dump({"commit": commit, "kernels": res}, open(os.path.join(out_dir, f"{tag}_pmc_by_kernel.json"), "w"), indent=1)

# HBM traffic per launch (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read stream:
# doubled -- an upper bound where the reads are not wide streams)
TRAFFIC = [
    ("k_mmstream_p(", "traffic_mmstream.json", "k_mmstream_p", "persistent launches over the whole S and AS regions for 8 + 8 groups of 63 / 64 coefficient vectors (a super-group of 255 proofs).  Algorithmic: "
     "2 x 4.24 GB of A fragments once + 16 x 8 MB of digits read, 16 x 33 MB of recombined partial products written (16-byte records per byte-position quad and vector: round 6; int32: 16 x 133 MB).  The workgroups consume 16 x 4.24 GB of fragments and 8096 x 8 MB of digit "
     "fragments out of the L2s; what an XCD's 32 concurrent workgroups (4 tile groups x 8 groups of one region) can share bounds the L2 misses at 4048 x (8.39 / 8 + 8 / 4) MB = 12.3 GB per 8 groups "
     "(24.7 GB per launch): DESIGN.md 4.2c"),
    ("k_mmstream_pb(", "traffic_mmstream_bw.json", "k_mmstream_pb", "b_w of the call's super-groups (4 groups of 255 one-byte columns) in one persistent launch over the BT+BV image (2.83 GB of fragments)"),
    ("k_mmstream(", "traffic_mmstream_nonpersistent.json", "k_mmstream", "one workgroup per item (the layout of rounds 1-3, mfh_set_mm_stream(ctx, 0, 0, 0, 0)); only present when a profiled program selects it"),
    ("k_mmstream1(", "traffic_mmstream1.json", "k_mmstream1", "one group per launch: b_w's pass over the BT+BV image (2.83 GB of fragments), HBM-bound"),
    ("k_expand_mm", "traffic_expandmm.json", "k_expand_mm", "the barrier-free CRS expansion: reads the compressed CRS region (92 B per row), writes the image region (129 536 B per row)"),
    ("k_encrypt_mm", "traffic_encryptmm.json", "k_encrypt_mm", "reads the Toeplitz(sk) fragments (12.7 MB per head value, from L2 after the first workgroups), writes 384 B of int32 partial sums per row and column chunk"),
    ("k_decrypt_mm", "traffic_decryptmm.json", "k_decrypt_mm", "65 536 full ciphertexts of 141 216 B streamed once (9.25 GB), Toeplitz(sk) fragments through LDS (13.2 MB, L2), 384 B of int32 partial sums "
     "written per row and column chunk"),
    ("k_witness_mm8q(", "traffic_witnessmm.json", "k_witness_mm8q", "one read of the SSP image in fragment order (2.86 GB) per 255 statements, 255 x 128 KB of w written"),
    ("k_evalmm_finish_groups", "traffic_evalmm_finish.json", "k_evalmm_finish_groups", "reads the recombined partial products of a launch's 16 groups (16 x 33 MB of 16-byte records; 16 x 133 MB of int32 before round 6), writes their ciphertexts"),
    ("k_ntt_lds_mul8", "traffic_ntt_lds.json", "k_ntt_lds_mul8", "2048-point blocks of 248 x 3 transforms: forward low stages, pointwise product, inverse low stages"),
    ("void k_eval<736, 2>", "traffic_eval2.json", "k_eval<736,2>", "the single-proof kernel: ~0 HBM bytes by construction (CRS b's, coefficients, partials)"),
    ("void k_mac_resident<736, 2>", "traffic_mac2.json", "k_mac_resident<736,2>", "the resident single-proof regime: one read of the expanded region"),
]
for key, fname, kname, note in TRAFFIC:
    kk = next((v for k, v in res.items() if key in k), None)
    if kk and "FETCH_SIZE" in kk and "WRITE_SIZE" in kk:
        fetch = kk["FETCH_SIZE"]["mean_of_large_launches"] * 1024 * 2
        write = kk["WRITE_SIZE"]["mean_of_large_launches"] * 1024
        json.dump({"kernel": kname, "commit": commit, "profile": f"{tag}_pmc_by_kernel.json", "fetch_bytes_x2_corrected": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
                   "launches": kk["FETCH_SIZE"]["large_launches"],
                   "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE); FETCH_SIZE doubled per the gfx950 correction for wide coalesced reads; " + note},
                  open(os.path.join(out_dir, fname), "w"), indent=1)
        print("traffic", kname, f"{(fetch + write) / 1e9:.3f} GB per launch")
for k, v in res.items():
    if "SQ_INSTS_VALU_MFMA_I8" in v:
        print(k[:60], {c: round(x["mean_of_large_launches"]) for c, x in v.items()})
