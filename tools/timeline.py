"""Print the kernel timeline of the last proof in a rocprofv3 kernel trace (gaps between kernels)."""
import csv, glob, sys
f = glob.glob(sys.argv[1])[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last k_smudge marks the end of a proof; take the window between the 3rd-last and last pair of smudges
idx = [i for i, r in enumerate(rows) if "k_smudge" in r["Kernel_Name"]]
if len(sys.argv) > 2 and sys.argv[2] == "regen":  # last proof before the CRS expansion (regenerate regime)
    ex = [i for i, r in enumerate(rows) if "k_expand" in r["Kernel_Name"]][0]
    idx = [i for i in idx if i < ex]
end = idx[-1]; start = idx[-3] + 1
t0 = int(rows[start]["Start_Timestamp"]); prev_end = t0
tot_k = 0
for r in rows[start:end + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:70]}")
    prev_end = e; tot_k += e - s
print(f"total span {(prev_end - t0) / 1e3:.1f} us, kernel time {tot_k / 1e3:.1f} us")
