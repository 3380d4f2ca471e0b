"""Kernel timeline of one mfh_prove_batch step from a rocprofv3 kernel trace: per stream (queue), start / duration of every launch. dev tool.
usage: python tools/batch_timeline.py '<glob of *_kernel_trace.csv>' [step index from the end, default 1]"""
import csv, glob, sys
f = max(glob.glob(sys.argv[1]), key=lambda p: len(open(p).read()))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a step of the transient-image path starts with the three expansion launches k_evalmm16<1, ...>
ex = [i for i, r in enumerate(rows) if "k_evalmm16<1" in r["Kernel_Name"] or "k_expand_mm" in r["Kernel_Name"]]
starts = ex[::3]  # three expansion launches (S, AS, BT+BV) per call
s0 = starts[-back - 1]; s1 = starts[-back]
t0 = int(rows[s0]["Start_Timestamp"])
qs = {}
for r in rows[s0:s1]:
    q = r.get("Queue_Id", "0")
    qs.setdefault(q, len(qs))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  q{qs[q]}  dur {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:90]}")
print(f"step span {(int(rows[s1]['Start_Timestamp']) - t0) / 1e3:.1f} us")
