"""Turn gpurun_out/prof (tools/profile_round.sh) into the committed profiles/rNN_* summaries.
usage: python tools/make_profiles.py <round-number>"""
import csv, glob, json, os, re, sys, collections
rnd = int(sys.argv[1]); tag = f"r{rnd:02d}"
P = "gpurun_out/prof"; OUT = "profiles"
KERNEL = "k_fused"
TRACES = 1048576
ALG = 131088

def counter(dirname, name):
    tot = collections.defaultdict(float)
    for f in glob.glob(f"{P}/{dirname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name:
                tot[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = sorted(tot.values())
    return (sum(v) / len(v), len(v)) if v else (None, 0)

ALG_C = {1: 131088, 2: 131112, 3: 131280}
TPL = {1: 1048576, 2: 1048576, 3: 262144}          # traces per launch (config 3: one channel plan)
cal, nc = counter("cal_FETCH_SIZE", "FETCH_SIZE")
corr = 2.0
note = "gfx950 x2 (MI355X_MICROARCH.md, HBM section)"
if cal:
    corr = TRACES * 131072.0 / (cal * 1024.0)
    note = (f"calibrated on the load-only build of the same kernel (same 8 B/lane access pattern, known "
            f"131072 B/trace): FETCH_SIZE reads {cal * 1024 / TRACES:.0f} B/trace -> factor {corr:.3f} "
            f"(the guide's gfx950 factor for 16 B/lane streams is 2)")
for c in (1, 2, 3):
    # 1. kernel stats
    for f in glob.glob(f"{P}/stats_c{c}/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.reader(open(f)))
        name = f"{OUT}/{tag}_bench_kernel_stats.csv" if c == 1 else f"{OUT}/{tag}_bench_config{c}_kernel_stats.csv"
        with open(name, "w", newline="") as o:
            csv.writer(o).writerows(rows)
        for r in rows[1:]:
            if KERNEL in r[0]:
                print(f"config {c} kernel stats:", r[0][:70], r[1:5])
    if os.path.exists(f"{P}/stats_c{c}.log"):
        line = [l for l in open(f"{P}/stats_c{c}.log") if l.startswith('{"metric"')]
        if line:
            open(f"{OUT}/{tag}_bench_line.json" if c == 1 else f"{OUT}/{tag}_bench_config{c}_line.json",
                 "w").write(line[-1])
    # 2./3. traffic
    fetch, nf = counter(f"pmc_FETCH_SIZE_c{c}", "FETCH_SIZE")
    write, nw = counter(f"pmc_WRITE_SIZE_c{c}", "WRITE_SIZE")
    if fetch is None or write is None:
        continue
    hbm = fetch * 1024.0 * corr + write * 1024.0
    json.dump({"round": rnd, "workload": f"config{c}",
               "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --config {c} --steps 2 --warmup 1 --no-cpu-baseline",
               "kernel": "k_fused (every launch of the pass averaged)", "engine": "fused", "traces_per_launch": TPL[c],
               "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
               "fetch_correction": corr, "correction": note,
               "hbm_bytes_per_launch": hbm, "hbm_bytes_per_trace": hbm / TPL[c],
               "algorithmic_bytes_per_trace": ALG_C[c]},
              open(f"{OUT}/{tag}_traffic.json" if c == 1 else f"{OUT}/{tag}_traffic_config{c}.json", "w"), indent=1)
    print(f"config {c} traffic B/trace:", hbm / TPL[c], "corr", corr)
# 4. SQ counters per trace
sq = {}
for d in ("sq_1", "sq_2"):
    for f in glob.glob(f"{P}/{d}/**/*counter_collection.csv", recursive=True):
        names = {r["Counter_Name"] for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]}
        for n in names:
            v, k = counter(d, n)
            sq[n] = v / TRACES
clock = open(f"{P}/clock_probe.txt").read().strip().split("\n")[-2:] if os.path.exists(f"{P}/clock_probe.txt") else []
json.dump({"round": rnd, "workload": "config1", "kernel": "k_fused<0,false>", "traces_per_launch": TRACES,
           "per_trace": sq,
           "units": "SQ_ACTIVE_* / SQ_WAIT_* / SQ_WAVE_CYCLES in units of 4 cycles, summed over the waves of a trace; SQ_INSTS_* wave-instructions",
           "clock_probe": clock}, open(f"{OUT}/{tag}_sq_counters.json", "w"), indent=1)
print(json.dumps(sq, indent=1))
