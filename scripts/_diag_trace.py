import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
fin = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "finalize_kernel" in r["Kernel_Name"]]
d = [(fin[i + 1][1] - fin[i][1]) / 1e6 for i in range(len(fin) - 1)]
print("iterations:", len(d))
print(" ".join(f"{x:.2f}" for x in d))
# longest kernels in slow iterations
slow = [i for i, x in enumerate(d) if x > 2.5 and x < 50]
print("slow iterations:", slow[:20])
for i in slow[:3]:
    a, b = fin[i][1], fin[i + 1][1]
    ks = [(r["Kernel_Name"].split("(anonymous namespace)::")[-1][:30], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(r["Start_Timestamp"]) - a) / 1e3)
          for r in rows if a <= int(r["Start_Timestamp"]) < b]
    print("iteration", i, [(k, round(t, 1), round(s, 1)) for k, t, s in ks])
