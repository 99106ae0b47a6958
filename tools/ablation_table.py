"""Throughput / clock / power of the FUSED kernel with parts cut out (build-time ablations), the
measurements behind the bound model of DESIGN.md section 5.1.
   for each variant:  make -C detprocess_amd/csrc variant NAME=<v> EXTRA="-DOFX_QUICK <flags>"
   python tools/ablation_table.py [out.json]        (on the GPU box, from the repo root)
Variants: full (the product library), noexch (no LDS exchanges), notail (nothing after the last
inverse stage), nofft (no DFT butterflies / inter-stage twiddles: loads, middle step, exchanges and
tail remain), noexch_notail (arithmetic + loads only), nofft_notail (loads + exchanges + middle),
loadonly (the trace stream alone).  Results of the ablated builds are wrong by construction."""
import json, os, re, subprocess, sys

VARIANTS = ["full", "noexch", "notail", "nofft", "noexch_notail", "nofft_notail", "loadonly"]
dest = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ablation_table.json"
rows = {}
for v in VARIANTS:
    env = dict(os.environ)
    lib = os.path.join(os.getcwd(), f"gpurun_{v}.so")
    if v != "full":
        if not os.path.exists(lib):
            continue
        env["OFX_LIB"] = lib
    r = subprocess.run([sys.executable, "tools/clock_probe.py", "262144", "4"], env=env,
                       capture_output=True, text=True)
    m = re.search(r"([\d.]+) M traces/s", r.stdout)
    s = re.findall(r"\((-?\d+), (-?[\d.]+)\)", r.stdout)
    clk = sorted(int(a) for a, b in s if int(a) > 0)
    pw = sorted(float(b) for a, b in s if float(b) > 0)
    rows[v] = {"M_traces_per_s": float(m.group(1)) if m else None,
               "sclk_MHz_median": clk[len(clk) // 2] if clk else None,
               "power_W_median": pw[len(pw) // 2] if pw else None, "samples": len(s)}
    print(v, rows[v], flush=True)
json.dump(rows, open(dest, "w"), indent=1)
