"""Fold the two rocprofv3 PMC passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE; kilobytes per dispatch) into
profiles/rNN_hbm_traffic.json: memory-side bytes per launch and per kernel class. FETCH_SIZE is doubled -- on gfx950 it tallies
128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM section); WRITE_SIZE matched the algorithmic store volume as is."""
import collections, csv, glob, json, os, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
out_path = sys.argv[2] if len(sys.argv) > 2 else "profiles/r02_hbm_traffic.json"
agg = collections.defaultdict(lambda: {"launches": 0, "fetch_kb_raw": 0.0, "write_kb": 0.0})
for kind, key in (("fetch", "fetch_kb_raw"), ("write", "write_kb")):
    f = sorted(glob.glob(f"{src}/pmc_{kind}/*/*_counter_collection.csv"), key=os.path.getmtime)[-1]
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        cls = "gemm_bf16_t144" if ("gemm_bf16_t144" in name or "gemm_bf16_t288w" in name) else "sanm_qkv_attn_kernel" if "sanm_qkv_attn" in name else name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0][:48]
        for k in (cls, name[:96]):
            agg[k][key] += float(r["Counter_Value"])
            if kind == "fetch":
                agg[k]["launches"] += 1
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 1 --profile-steps 1",
       "correction": "fetch bytes = 2 x FETCH_SIZE (gfx950 counts 128-B requests at 64 B); write bytes = WRITE_SIZE; both reported in KiB by rocprofv3",
       "kernels": {}}
for k, v in sorted(agg.items(), key=lambda kv: -(2 * kv[1]["fetch_kb_raw"] + kv[1]["write_kb"])):
    if v["launches"] == 0:
        continue
    fetch = 2.0 * v["fetch_kb_raw"] * 1024 / v["launches"]
    write = v["write_kb"] * 1024 / v["launches"]
    out["kernels"][k] = {"launches": v["launches"], "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
                         "bytes_per_launch": round(fetch + write)}
json.dump(out, open(out_path, "w"), indent=1)
for k, v in list(out["kernels"].items())[:10]:
    print(f"{k[:80]:80s} {v['launches']:5d}  fetch {v['fetch_bytes_per_launch']/1e6:8.1f} MB  write {v['write_bytes_per_launch']/1e6:8.1f} MB")
