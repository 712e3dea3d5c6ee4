"""scratch/profile_r3_collect.py <tag> [round]: digest gpurun_out/prof_<tag>/ (scratch/profile_r3.sh) into the committed files under
profiles/: kernel-stats CSVs, PMC summaries, pmc_c2.json (what bench.py's roofline reads) and config_kernels.json."""
import collections, csv, glob, json, os, re, shutil, subprocess, sys, time
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
src = "gpurun_out/prof_%s" % tag
# on the GPU box (no .git there) the build id comes in by the environment and the digest goes under gpurun_out/
build = os.environ.get("RT_BUILD_ID") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dst = os.environ.get("RT_PROFILE_OUT", "profiles")
os.makedirs(dst, exist_ok=True)
for f in ("c2_kernel_stats_1inflight.csv", "c2_kernel_stats_default.csv", "c2_kernel_stats_batch16.csv", "batch1.txt", "bench_default.json", "batch16.txt"):
    shutil.copy(os.path.join(src, f), "%s/%s_%s" % (dst, rnd, f))

def kname(k):
    m = re.search(r"(k_[a-z_]+)", k)
    return m.group(1) if m else None

def pmc(prefix):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s_*/**/*counter_collection.csv" % (src, prefix), recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc

c2 = pmc("pmc")
with open("%s/%s_c2_pmc_summary.txt" % (dst, rnd), "w") as o:
    o.write("# rocprofv3 --pmc, one pass per run (scratch/profile_r3.sh), scratch/batch_launches.py 1 (one C2 frame per launch), build %s\n" % build)
    for k in c2:
        o.write("[%s] mean per launch over %d launches\n" % (k, len(next(iter(c2[k].values())))))
        for c, v in sorted(c2[k].items()):
            o.write("  %-36s %.6g\n" % (c, sum(v) / len(v)))
t = {c: sum(v) / len(v) for c, v in c2["k_traverse"].items()}
stats = {kname(r["Name"]): r for r in csv.DictReader(open(os.path.join(src, "c2_kernel_stats_1inflight.csv"))) if kname(r["Name"])}
kt = float(stats["k_traverse"]["AverageNs"]) * 1e-9
read_b = 128 * t.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * t.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * t.get("TCC_EA0_RDREQ_32B_sum", 0)
write_b = t.get("WRITE_SIZE", 0) * 1024
out = {"kernel": "k_traverse", "spheres": 1000000, "n_gpus": 1, "build": build, "collected": time.strftime("%Y-%m-%d"),
       "kernel_ms_in_trace": round(kt * 1e3, 4),
       "hbm_bytes_per_launch": int(read_b + write_b), "hbm_read_bytes": int(read_b), "hbm_write_bytes": int(write_b),
       "l2_hit_rate": round(t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]), 4),
       "l2_request_bytes_per_launch": int(128 * t["TCC_REQ_sum"]),
       "valu": {"wave_instructions_per_launch": int(t["SQ_INSTS_VALU"]),
                "active_lane_fraction": round(t["SQ_THREAD_CYCLES_VALU"] / (64.0 * t["SQ_INSTS_VALU"]), 4),
                "issue_fraction_in_trace": round(t["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9 * kt), 4)},
       "method": "rocprofv3 --pmc, separate passes: HBM reads = 128*TCC_EA0_RDREQ_128B + 64*..._64B + 32*..._32B (FETCH_SIZE reads half on gfx950, "
                 "MI355X_MICROARCH.md), writes = WRITE_SIZE*1024; active lanes = SQ_THREAD_CYCLES_VALU / (64*SQ_INSTS_VALU); "
                 "issue fraction = SQ_INSTS_VALU*4 cycles / (1024 SIMDs * 2.4 GHz * kernel time)",
       "source": "profiles/%s_c2_pmc_summary.txt, profiles/%s_c2_kernel_stats_1inflight.csv" % (rnd, rnd)}

def digest(t, kt):
    read_b = 128 * t.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * t.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * t.get("TCC_EA0_RDREQ_32B_sum", 0)
    write_b = t.get("WRITE_SIZE", 0) * 1024
    return {"kernel_ms_in_trace": round(kt * 1e3, 4), "hbm_bytes_per_launch": int(read_b + write_b), "hbm_read_bytes": int(read_b),
            "hbm_write_bytes": int(write_b), "l2_hit_rate": round(t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]), 4),
            "l2_request_bytes_per_launch": int(128 * t["TCC_REQ_sum"]),
            "valu": {"wave_instructions_per_launch": int(t["SQ_INSTS_VALU"]),
                     "active_lane_fraction": round(t["SQ_THREAD_CYCLES_VALU"] / (64.0 * t["SQ_INSTS_VALU"]), 4),
                     "issue_fraction_in_trace": round(t["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9 * kt), 4)}}

cb = pmc("pmcb")
if "k_traverse" in cb:
    tb = {c: sum(v) / len(v) for c, v in cb["k_traverse"].items()}
    sb = {kname(r["Name"]): r for r in csv.DictReader(open(os.path.join(src, "c2_kernel_stats_batch16.csv"))) if kname(r["Name"])}
    out["batch"] = dict(digest(tb, float(sb["k_traverse"]["AverageNs"]) * 1e-9), frames_per_launch=16,
                        source="profiles/%s_c2_batch16_pmc_summary.txt, profiles/%s_c2_kernel_stats_batch16.csv" % (rnd, rnd))
    with open("%s/%s_c2_batch16_pmc_summary.txt" % (dst, rnd), "w") as o:
        o.write("# rocprofv3 --pmc, one pass per run, scratch/batch_launches.py 16 (launches of 16 C2 frames, one at a time), build %s\n" % build)
        for k in ("k_raygen", "k_traverse", "k_shade"):
            if k in cb:
                o.write("[%s] mean per launch over %d launches\n" % (k, len(next(iter(cb[k].values())))))
                for c, v in sorted(cb[k].items()):
                    o.write("  %-36s %.6g\n" % (c, sum(v) / len(v)))
json.dump(out, open(dst + "/pmc_c2.json", "w"), indent=1)
print(json.dumps(out, indent=1))

cfg = {}
for f in sorted(glob.glob(os.path.join(src, "cfg_*_kernel_stats.csv"))):
    name = re.match(r"cfg_(.*)_kernel_stats.csv", os.path.basename(f)).group(1)
    shutil.copy(f, "%s/%s_cfg_%s_kernel_stats.csv" % (dst, rnd, name))
    rows = [(kname(r["Name"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open(f))]
    tot = sum(v for k, v in rows)
    ks = collections.Counter()
    for k, v in rows:
        ks[k or "copies and fills"] += v
    top, tv = ks.most_common(1)[0]
    cfg[name] = {"kernel": top, "fraction": round(tv / tot, 3), "shares": {k: round(v / tot, 3) for k, v in ks.most_common(5)},
                 "source": "profiles/%s_cfg_%s_kernel_stats.csv (rocprofv3 --kernel-trace --stats, build %s)" % (rnd, name, build)}
    j = os.path.join(src, "cfg_%s.json" % name)
    if os.path.exists(j):
        shutil.copy(j, "%s/%s_cfg_%s.json" % (dst, rnd, name))
def cfg_pmc(name):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/cfgpmc_%s_*/**/*counter_collection.csv" % (src, name), recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc

FRAMES_PER_RUN = 6   # scratch/bench_configs.py: 1 counting + 2 warm-up + 3 timed renders
for name in list(cfg):
    c = cfg_pmc(name)
    if not c:
        continue
    tot_i = sum(sum(v["SQ_INSTS_VALU"]) for v in c.values() if "SQ_INSTS_VALU" in v)
    tot_l = sum(sum(v["SQ_THREAD_CYCLES_VALU"]) for v in c.values() if "SQ_THREAD_CYCLES_VALU" in v)
    gui = sum(sum(v["GRBM_GUI_ACTIVE"]) for v in c.values() if "GRBM_GUI_ACTIVE" in v) / 8.0 / 2.4e9   # summed over the 8 XCDs
    rd = sum(128 * sum(v.get("TCC_EA0_RDREQ_128B_sum", [0])) + 64 * sum(v.get("TCC_EA0_RDREQ_64B_sum", [0])) + 32 * sum(v.get("TCC_EA0_RDREQ_32B_sum", [0])) for v in c.values())
    wr = sum(1024 * sum(v.get("WRITE_SIZE", [0])) for v in c.values())
    per = {}
    for k, v in c.items():
        if "SQ_INSTS_VALU" in v and sum(v["SQ_INSTS_VALU"]) > 0.01 * tot_i:
            i = sum(v["SQ_INSTS_VALU"])
            busy = sum(v["GRBM_GUI_ACTIVE"]) / 8.0 / 2.4e9
            per[k] = {"wave_instructions": int(i), "active_lane_fraction": round(sum(v["SQ_THREAD_CYCLES_VALU"]) / 64.0 / i, 3),
                      "issue_ms": round(i * 4 / (1024 * 2.4e9) * 1e3, 2), "busy_ms": round(busy * 1e3, 2),
                      "issue_fraction": round(i * 4 / (1024 * 2.4e9) / busy, 3) if busy > 0 else None,
                      "waves_waiting_fraction": round(sum(v["SQ_WAIT_ANY"]) / max(1.0, sum(v["SQ_WAVE_CYCLES"])), 3)}
    spp = json.load(open(os.path.join(src, "cfg_%s.json" % name)))["spp"] if os.path.exists(os.path.join(src, "cfg_%s.json" % name)) else None
    cfg[name]["valu"] = {"run": "%d frames per run (scratch/bench_configs.py %s%s), separate rocprofv3 --pmc passes, totals over the run" % (FRAMES_PER_RUN, name, ", RT_BENCH_SPP=8" if spp and spp > 1 else ""),
                         "issue_fraction": round(tot_i * 4 / (1024 * 2.4e9) / gui, 3), "active_lane_fraction": round(tot_l / 64.0 / tot_i, 3),
                         "hbm_bytes_per_frame": int((rd + wr) / FRAMES_PER_RUN), "hbm_GB_per_s": round((rd + wr) / gui / 1e9, 1),
                         "kernels": per, "source": "profiles/%s_cfg_%s_pmc_summary.txt" % (rnd, name)}
    with open("%s/%s_cfg_%s_pmc_summary.txt" % (dst, rnd, name), "w") as o:
        o.write("# rocprofv3 --pmc, one pass per run, scratch/bench_configs.py %s (%d frames per run%s), build %s: totals over the run\n" % (name, FRAMES_PER_RUN, ", 8 spp" if spp and spp > 1 else "", build))
        for k in c:
            o.write("[%s] %d launches\n" % (k, len(next(iter(c[k].values())))))
            for cn, v in sorted(c[k].items()):
                o.write("  %-28s %.6g\n" % (cn, sum(v)))
json.dump(cfg, open(dst + "/config_kernels.json", "w"), indent=1)
print(json.dumps(cfg, indent=1))
