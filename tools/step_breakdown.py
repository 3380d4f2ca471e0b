"""Where one mfh_prove_batch step spends its time, from a rocprofv3 --kernel-trace CSV (dev tool).
usage: python tools/step_breakdown.py '<glob of *_kernel_trace.csv>' [step index from the end, default 1]
Prints per kernel: launches, summed duration, and the time during which ONLY that kernel (possibly several launches of it) was on the GPU
-- what would be saved if it vanished -- plus the idle time of the step."""
import csv, glob, re, sys
from collections import defaultdict

f = max(glob.glob(sys.argv[1], recursive=True), key=lambda p: len(open(p).read()))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ex = [i for i, r in enumerate(rows) if "k_evalmm16<1" in r["Kernel_Name"] or "k_expand_mm" in r["Kernel_Name"]]
starts = ex[::3]
s0, s1 = starts[-back - 1], starts[-back]
step = rows[s0:s1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(rows[s1]["Start_Timestamp"])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:48]


ev = []
tot, cnt = defaultdict(int), defaultdict(int)
for r in step:
    s, e, k = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
    tot[k] += e - s
    cnt[k] += 1
    ev.append((s, 1, k))
    ev.append((e, -1, k))
ev.sort()
active = defaultdict(int)
alone = defaultdict(int)
idle = 0
last = t0
for t, dlt, k in ev:
    live = [x for x, v in active.items() if v > 0]
    if t > last:
        if not live:
            idle += t - last
        elif len(live) == 1:
            alone[live[0]] += t - last
    active[k] += dlt
    last = t
print(f"step span {(t1 - t0) / 1e6:.3f} ms, idle {idle / 1e6:.3f} ms (tail to next step {(t1 - last) / 1e6:.3f} ms)")
print(f"{'kernel':50s} {'n':>5s} {'sum ms':>9s} {'alone ms':>9s}")
for k in sorted(tot, key=lambda k: -tot[k]):
    print(f"{k:50s} {cnt[k]:5d} {tot[k] / 1e6:9.3f} {alone[k] / 1e6:9.3f}")
