"""Fold one rocprofv3 SQ counter pass (tools/profile_round.sh: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE) into profiles/rNN_mfma_util.json: per
kernel class the MFMA pipe utilisation (gfx94x MfmaUtil formula, MFMA busy cycles / (GPU cycles x 256 CUs x 4 SIMDs); GRBM_GUI_ACTIVE comes
summed over the 8 XCDs, so GPU cycles = GRBM_GUI_ACTIVE / 8) and where the
wave cycles went (WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stalls, ACTIVE_INST_ANY = issuing; quad-cycles)."""
import collections, csv, glob, json, os, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof/pmc_sq"
out_path = sys.argv[2] if len(sys.argv) > 2 else "profiles/r02_mfma_util.json"
f = sorted(glob.glob(f"{src}/*/*_counter_collection.csv"), key=os.path.getmtime)[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
dur = collections.defaultdict(float)                  # kernel-trace durations (ns) per class, one per dispatch
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    cls = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:64]
    agg[cls][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen[cls] and "End_Timestamp" in r:
        dur[cls] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    seen[cls].add(r["Dispatch_Id"])
CUS, SIMDS, XCDS = 256, 4, 8
out = {"source": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS "
                 "SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline",
       "formulas": {"mfma_util": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs); kernels run serialised and ~10 % slower under counter collection", "wave shares": "X / SQ_WAVE_CYCLES",
                    "lds_bank_conflict_per_cu_cycle": "SQ_LDS_BANK_CONFLICT / (GPU cycles * 256 CUs): extra LDS cycles lost to bank conflicts per CU-cycle"},
       "kernels": {}}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
    n = len(seen[k])
    gui, wave = v["GRBM_GUI_ACTIVE"], max(v["SQ_WAVE_CYCLES"], 1.0)
    if gui <= 0:
        continue
    out["kernels"][k] = {"launches": n, "gpu_cycles_per_launch": round(gui / XCDS / n), "mfma_util": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / XCDS * CUS * SIMDS), 4),
                         "wait_any_share": round(v["SQ_WAIT_ANY"] / wave, 3), "wait_inst_any_share": round(v["SQ_WAIT_INST_ANY"] / wave, 3),
                         "active_inst_share": round(v["SQ_ACTIVE_INST_ANY"] / wave, 3), "wait_inst_lds_share": round(v["SQ_WAIT_INST_LDS"] / wave, 3),
                         "lds_bank_conflict_per_cu_cycle": round(v["SQ_LDS_BANK_CONFLICT"] / (gui / XCDS * CUS), 4)}
    if dur[k] > 0:       # GPU-active cycles / wall time of the same dispatches: the clock the chip actually ran at under this kernel (power management)
        out["kernels"][k]["effective_clock_ghz"] = round(gui / XCDS / dur[k], 3)
# the class the bench's roofline names: every 144-row-tile GEMM launch together
t = collections.defaultdict(float); nl = 0
for k, v in agg.items():
    if k.startswith("gemm_bf16_t144") or k.startswith("gemm_bf16_t288w"):
        nl += len(seen[k])
        for c, x in v.items():
            t[c] += x
if nl:
    out["kernels"]["gemm_bf16_t144"] = {"launches": nl, "gpu_cycles_per_launch": round(t["GRBM_GUI_ACTIVE"] / XCDS / nl),
                                        "mfma_util": round(t["SQ_VALU_MFMA_BUSY_CYCLES"] / (t["GRBM_GUI_ACTIVE"] / XCDS * CUS * SIMDS), 4),
                                        "wait_any_share": round(t["SQ_WAIT_ANY"] / t["SQ_WAVE_CYCLES"], 3),
                                        "wait_inst_any_share": round(t["SQ_WAIT_INST_ANY"] / t["SQ_WAVE_CYCLES"], 3),
                                        "active_inst_share": round(t["SQ_ACTIVE_INST_ANY"] / t["SQ_WAVE_CYCLES"], 3)}
json.dump(out, open(out_path, "w"), indent=1)
for k, v in list(out["kernels"].items())[:12]:
    print(f"{k[:60]:60s} n={v['launches']:5d} cyc={v['gpu_cycles_per_launch']:8d} mfma {v['mfma_util']:.3f} wait {v['wait_any_share']:.2f} istall {v['wait_inst_any_share']:.2f} "
          f"active {v['active_inst_share']:.2f} lds-stall {v['wait_inst_lds_share']:.3f} bankconf {v['lds_bank_conflict_per_cu_cycle']:.3f}")
