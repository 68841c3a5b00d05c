"""Condenses the rocprofv3 passes of tools/profile.sh TAG (gpurun_out/prof/TAG) into profiles/round2/:

    TAG_kernel_stats.csv        per-kernel totals / averages of the --kernel-trace --stats pass
    TAG_counters.csv            every counter of the step kernel's LAST launch (the timed one), one row per counter
    TAG.json                    what bench.py reads: HBM bytes per reach-timestep (FETCH_SIZE x 2 + WRITE_SIZE, see below),
                                flops per reach and Newton iteration, issue statistics, the instantiation that ran and the
                                sha256 of the library it ran from (bench.py ignores the file when either differs)

FETCH_SIZE correction (MI355X_MICROARCH.md, HBM section): on gfx950 the counter tallies 128-byte read requests at 64 bytes,
i.e. reports half of the bytes of a coalesced streaming read - "double it before comparing with a byte count; other access
widths: calibrate on a known byte count".  This kernel's mandatory reads are known exactly (hk, Qk, hg, Qg once per launch:
4 B N sizeof(real)); the file records them next to the doubled counter so that the calibration is visible.
usage: python tools/refresh_profiles.py TAG [--kernel preissmann|derive]"""
import argparse, csv, glob, json, os, shutil, sys

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--kernel", default="preissmann")
ap.add_argument("--round", default="round4")
ap.add_argument("--second", default=None,
                help="tag of a second run of the same workload with another number of levels in its launch: the two together give the "
                     "part of the traffic that does not depend on the levels (state in, state out) and the part per level")
a = ap.parse_args()
P, D = f"gpurun_out/prof/{a.tag}", f"profiles/{a.round}"
os.makedirs(D, exist_ok=True)
newest = lambda pat: sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)[-1]
shutil.copy(newest(f"{P}/trace/**/*_kernel_stats.csv"), f"{D}/{a.tag}_kernel_stats.csv")
bench = json.loads(open(f"{P}/bench_trace.json").read().strip().split("\n")[-1])
vals = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_wait", "pmc_flops"):
    try:
        rows = list(csv.reader(open(newest(f"{P}/{name}/**/*_counter_collection.csv"))))
    except IndexError:
        print("missing pass", name, file=sys.stderr)
        continue
    h = rows[0]
    ci, cv, di, kn = h.index("Counter_Name"), h.index("Counter_Value"), h.index("Dispatch_Id"), h.index("Kernel_Name")
    keep = [r for r in rows[1:] if a.kernel in r[kn]]
    last = max(int(r[di]) for r in keep)
    vals.update({r[ci]: float(r[cv]) for r in keep if int(r[di]) == last})
    kernel_name = keep[-1][kn]
with open(f"{D}/{a.tag}_counters.csv", "w") as f:
    f.write('"Kernel_Name","Counter_Name","Counter_Value (last launch)"\n')
    for k, v in sorted(vals.items()):
        f.write(f'"{kernel_name[:120]}","{k}",{v:.17g}\n')
stats = list(csv.DictReader(open(f"{D}/{a.tag}_kernel_stats.csv")))
krow = [r for r in stats if a.kernel in r["Name"]][0]
cfg = bench["config"]
B, K, N = cfg["reaches_per_gpu"], bench["steps"], cfg["nodes"]
real = 8 if bench["dtype"] == "f64" else 4
its = cfg["mean_newton_iterations_per_step"]
out = {"tag": a.tag, "source": f"rocprofv3 passes of tools/profile.sh {a.tag} " + open(f"{P}/args.txt").read().strip(),
       "library_sha256": open(f"{P}/library_sha256.txt").read().strip(), "kernel_name": kernel_name[:200],
       "kernel": cfg["kernel"], "nodes": N, "reaches": B, "levels_in_launch": K, "dtype": bench["dtype"],
       "mean_newton_iterations_per_step": its,
       "kernel_stats_avg_ns": float(krow["AverageNs"]), "kernel_stats_calls": int(krow["Calls"]),
       "kernel_stats_max_ns": float(krow["MaxNs"]), "kernel_stats_total_ns": float(krow["TotalDurationNs"]),
       "kernel_stats_note": f"{krow['Calls']} launches in the trace: the warm-up launch ({bench['warmup']} levels) and the timed launch "
                            f"({K} levels) - the timed one is MaxNs; per level, TotalDurationNs / {bench['warmup'] + K} levels "
                            f"= {float(krow['TotalDurationNs']) / (bench['warmup'] + K) / 1e6:.4f} ms against bench.py's "
                            f"{bench['roofline']['kernel_ms'] / K:.4f} ms",
       "bench_kernel_ms": bench["roofline"]["kernel_ms"], "bench_value": bench["value"]}
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    fetch_raw, write = vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
    mandatory = 4.0 * B * N * real
    out.update(fetch_bytes_raw_counter=fetch_raw, fetch_bytes_corrected=2 * fetch_raw, write_bytes=write,
               mandatory_read_bytes=mandatory, mandatory_write_bytes=mandatory,
               hbm_bytes_per_reach_timestep=(2 * fetch_raw + write) / (B * K),
               algorithmic_bytes_per_reach_timestep=4 * N * real + 40,
               traffic_note="rocprofv3 FETCH_SIZE x 2 (gfx950 tallies 128-B read requests at 64 B; the kernel's known reads - "
                            "hk, Qk, hg, Qg once per launch - calibrate it) + WRITE_SIZE, separate passes, per reach-timestep")
if a.second and "FETCH_SIZE" in vals:
    P2 = f"gpurun_out/prof/{a.second}"
    b2 = json.loads(open(f"{P2}/bench_trace.json").read().strip().split("\n")[-1])
    v2 = {}
    for name in ("pmc_fetch", "pmc_write"):
        rows = list(csv.reader(open(newest(f"{P2}/{name}/**/*_counter_collection.csv"))))
        h = rows[0]
        ci, cv, di, kn = h.index("Counter_Name"), h.index("Counter_Value"), h.index("Dispatch_Id"), h.index("Kernel_Name")
        keep = [r for r in rows[1:] if a.kernel in r[kn]]
        last = max(int(r[di]) for r in keep)
        v2.update({r[ci]: float(r[cv]) for r in keep if int(r[di]) == last})
    K2 = b2["steps"]
    assert b2["config"]["reaches_per_gpu"] == B and b2["config"]["nodes"] == N and K2 != K
    t1 = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / B            # per reach, launch of K levels
    t2 = (2 * v2["FETCH_SIZE"] + v2["WRITE_SIZE"]) * 1024 / B                # per reach, launch of K2 levels
    per_level = (t1 - t2) / (K - K2)
    out.update(hbm_bytes_per_reach_fixed=t1 - per_level * K, hbm_bytes_per_reach_per_level=per_level,
               second_profile=dict(tag=a.second, levels_in_launch=K2, hbm_bytes_per_reach=t2),
               traffic_note="rocprofv3 FETCH_SIZE x 2 (gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE, separate passes, of the "
                            f"timed launch at {K} and at {K2} levels: bytes per reach = fixed (state in, state out) + per_level x levels")
w = vals.get("SQ_WAVES")
if w and "SQ_INSTS_VALU" in vals:
    sfx = "F64" if real == 8 else "F32"
    fl = 64 * (2 * vals[f"SQ_INSTS_VALU_FMA_{sfx}"] + vals[f"SQ_INSTS_VALU_MUL_{sfx}"] + vals[f"SQ_INSTS_VALU_ADD_{sfx}"]
               + vals[f"SQ_INSTS_VALU_TRANS_{sfx}"])
    out.update(flops_per_launch=fl, flops_per_reach_iteration=fl / (B * K * its),
               valu_instructions_per_wave_iteration=vals["SQ_INSTS_VALU"] / w / (K * its),
               arithmetic_share_of_valu=(vals[f"SQ_INSTS_VALU_FMA_{sfx}"] + vals[f"SQ_INSTS_VALU_MUL_{sfx}"] + vals[f"SQ_INSTS_VALU_ADD_{sfx}"]
                + vals[f"SQ_INSTS_VALU_TRANS_{sfx}"]) / vals["SQ_INSTS_VALU"])
if "SQ_ACTIVE_INST_VALU" in vals:
    out.update(valu_active_per_wave_cycle=vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_WAVE_CYCLES"],
               insts_per_wave={k[9:].lower(): vals[k] / w for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM")})
if "SQ_WAIT_ANY" in vals:
    out.update(wait_any_share=vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"], wait_inst_any_share=vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"])
json.dump(out, open(f"{D}/{a.tag}.json", "w"), indent=1)
shutil.copy(f"{P}/bench_trace.json", f"{D}/{a.tag}_bench_under_rocprof.json")
print(json.dumps({k: v for k, v in out.items() if k not in ("source", "kernel_name", "traffic_note")}, indent=1))
