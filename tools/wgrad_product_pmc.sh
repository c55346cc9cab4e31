#!/bin/bash
# HBM traffic of the weight-gradient launch PER PRODUCT (verdict r5, item 4): two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
# `tools/kernel_bench.py wgradp 4`: every product of a layer alone at the merged launch's chunking (three row chunks), then the merged
# launch.  bytes = (2 FETCH_SIZE + WRITE_SIZE) 1024 (gfx950 correction, MI355X_MICROARCH.md).  usage (GPU box): bash tools/wgrad_product_pmc.sh
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_wgp
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o t -- python3 $root/tools/kernel_bench.py wgradp 4 > $out.$c.log 2>&1 || { echo "pass $c failed"; tail -5 $out.$c.log; exit 1; }
done
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "wgrad_multi_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals[c] = [float(r["Counter_Value"]) for r in rows]
names = ["in_proj  dW[2304, 768]", "out_proj dW[768, 768]", "linear1  dW[1536, 768]", "linear2  dW[768, 1536]", "merged layer (144 tiles)"]
M, D, ff = 16672, 768, 1536
alg = [(M * n + M * k) * 2.0 + n * k * 4.0 for n, k in ((2304, 768), (768, 768), (1536, 768), (768, 1536))]
alg.append(sum(alg))
per = len(vals["FETCH_SIZE"]) // 5            # 1 warm-up + reps launches per entry
print(f"# {per} launches per entry (first = warm-up, dropped); M = {M} rows, three row chunks per tile")
print(f"{'product':28s} {'read MB':>9s} {'write MB':>9s} {'total':>9s} {'algorithmic':>12s} {'ratio':>6s}   (operands MB, gradient MB)")
for i, nm in enumerate(names):
    fs, ws = vals["FETCH_SIZE"][i * per + 1:(i + 1) * per], vals["WRITE_SIZE"][i * per + 1:(i + 1) * per]
    rd, wr = 2.0 * sum(fs) / len(fs) * 1024 / 1e6, sum(ws) / len(ws) * 1024 / 1e6
    n, k = ((2304, 768), (768, 768), (1536, 768), (768, 1536), (0, 0))[i]
    ops_mb = (M * n + M * k) * 2.0 / 1e6 if n else sum((M * a + M * b) * 2.0 for a, b in ((2304, 768), (768, 768), (1536, 768), (768, 1536))) / 1e6
    g_mb = n * k * 4.0 / 1e6 if n else 18.9
    print(f"{nm:28s} {rd:9.1f} {wr:9.1f} {rd + wr:9.1f} {alg[i] / 1e6:12.1f} {(rd + wr) / (alg[i] / 1e6):6.2f}   ({ops_mb:.1f}, {g_mb:.1f})")
PY
rm -rf $out
