"""Summarise a rocprofv3 kernel trace: median duration per (kernel, grid) -- separates decode-step launches from prefill ones."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
g = collections.defaultdict(list)
for r in rows:
    g[(r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in g.values())
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k[0]:60s} grid {k[1]:>8s}x{k[2]:<4s} wg {k[3]:>4s} n {len(v):5d} med {v[len(v)//2]/1e3:8.1f} us  sum {sum(v)/1e6:8.2f} ms {100*sum(v)/tot:5.1f}%")
