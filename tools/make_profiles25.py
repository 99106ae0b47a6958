"""Turn gpurun_out/prof25 (tools/profile_fused25.sh) into profiles/rNN_*25000* summaries.
usage: python tools/make_profiles25.py <round-number> [samples = 25000] [traces = 1048576]
(4096: the one-wave-per-trace kernel k_wave, gpurun_out/prof4096 -> profiles/rNN_*4096*; 8192: k_wave2)"""
import csv, glob, json, sys, collections
rnd = int(sys.argv[1]); tag = f"r{rnd:02d}"
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 25000
TRACES = int(sys.argv[3]) if len(sys.argv) > 3 else 1048576
P = "gpurun_out/prof25" if NS == 25000 else f"gpurun_out/prof{NS}"; OUT = "profiles"
KERNEL = {25000: "k_fused25", 4096: "k_wave<", 8192: "k_wave2<2", 16384: "k_wave2<4"}.get(NS, "k_fused25")
KNAME = KERNEL.split("<")[0]
ALG = NS * 4 + 16


def counter(dirname, name):
    tot = collections.defaultdict(float)
    for f in glob.glob(f"{P}/{dirname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name:
                tot[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = sorted(tot.values())
    return (sum(v) / len(v), len(v)) if v else (None, 0)


for f in glob.glob(f"{P}/stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.reader(open(f)))
    csv.writer(open(f"{OUT}/{tag}_bench{NS}_kernel_stats.csv", "w", newline="")).writerows(rows)
    for r in rows[1:]:
        if KERNEL in r[0]:
            print("kernel stats:", r[0][:60], r[1:5])
line = [l for l in open(f"{P}/stats.log") if l.startswith('{"metric"')]
if line:
    open(f"{OUT}/{tag}_bench{NS}_config1_line.json", "w").write(line[-1])
# FETCH_SIZE correction: the factor calibrated on the load-only build of k_fused (same 8 B / lane
# buffer loads), profiles/rNN_traffic.json
corr = json.load(open(f"{OUT}/{tag}_traffic.json"))["fetch_correction"]
fetch, nf = counter("pmc_FETCH_SIZE", "FETCH_SIZE")
write, nw = counter("pmc_WRITE_SIZE", "WRITE_SIZE")
if fetch is not None and write is not None:
    hbm = fetch * 1024.0 * corr + write * 1024.0
    json.dump({"round": rnd, "workload": f"config1_n{NS}",
               "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --samples {NS} --traces {TRACES} --config 1 --steps 2 --warmup 1 --no-cpu-baseline",
               "kernel": f"{KNAME} (every launch of the pass averaged)", "engine": "fused", "traces_per_launch": TRACES,
               "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write, "fetch_correction": corr,
               "correction": f"factor of {OUT}/{tag}_traffic.json (load-only build of k_fused, the same 8 B / lane buffer loads)",
               "hbm_bytes_per_launch": hbm, "hbm_bytes_per_trace": hbm / TRACES, "algorithmic_bytes_per_trace": ALG},
              open(f"{OUT}/{tag}_traffic_{NS}.json", "w"), indent=1)
    print("traffic B/trace:", hbm / TRACES, "algorithmic", ALG)
sq = {}
for d in ("sq_1", "sq_2"):
    for f in glob.glob(f"{P}/{d}/**/*counter_collection.csv", recursive=True):
        names = {r["Counter_Name"] for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]}
        for n in names:
            v, k = counter(d, n)
            sq[n] = v / TRACES
json.dump({"round": rnd, "kernel": f"{KNAME}<0>" if NS in (4096, 8192, 16384) else "k_fused25<0,false>", "workload": f"config1_n{NS}", "traces_per_launch": TRACES, "per_trace": sq,
           "units": "SQ_ACTIVE_* / SQ_WAIT_* / SQ_WAVE_CYCLES in units of 4 cycles, summed over the waves of a trace; SQ_INSTS_* wave-instructions"},
          open(f"{OUT}/{tag}_sq_counters_{NS}.json", "w"), indent=1)
print(json.dumps(sq, indent=1))
